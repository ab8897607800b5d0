#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_h
rm -rf $out && mkdir -p $out
cd $root
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1
echo "pytest rc=$?" > $out/progress.txt
tail -3 $out/pytest.txt
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
echo "bench rc=$?" >> $out/progress.txt
