#!/bin/bash
# round 4: the m-space assembly - tests first (assembly, labels, pair build, then everything), then the bench line
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_g
rm -rf $out && mkdir -p $out
cd $root
timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "assembly or labels or pair_build or laplacian_view or spmv or mean_filter or null_vectors" > $out/pytest_asm.txt 2>&1
echo "pytest asm rc=$?" > $out/progress.txt
tail -15 $out/pytest_asm.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1
echo "pytest rc=$?" >> $out/progress.txt
tail -5 $out/pytest.txt
timeout -k 10 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench.json 2> $out/bench.err
echo "bench rc=$?" >> $out/progress.txt
