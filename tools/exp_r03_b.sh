#!/bin/bash
# round-3 experiment B: two steps per exchange - bit identity, then timing
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r03_b
mkdir -p $out
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "resident" > $out/pytest.txt 2>&1
echo "pytest rc=$?" >> $out/pytest.txt
timeout -k 10 200 python3 tools/bench_cheb.py 250000 60000 --modes 2,1 > $out/cheb.txt 2>&1
echo "cheb rc=$?" >> $out/cheb.txt
