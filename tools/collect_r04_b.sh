#!/bin/bash
# round 4 collection, part B: a quick test of the assembly, then the full bench line (extras + measured CPU baseline)
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/final_b
rm -rf $out && mkdir -p $out
cd $root
timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "assembly or labels or pair_build or multi_component or piled" > $out/pytest_asm.txt 2>&1
echo "pytest asm rc=$?" > $out/progress.txt
tail -3 $out/pytest_asm.txt
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
echo "bench rc=$?" >> $out/progress.txt
