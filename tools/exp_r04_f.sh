#!/bin/bash
# round 4: the measurement-contract additions (stage roofline entries, KNN pair counting, full-size oracle test, sweep columns)
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_f
rm -rf $out && mkdir -p $out
cd $root
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "full_size_250k or knn_bit_exact or test_knn" > $out/pytest.txt 2>&1
echo "pytest rc=$?" > $out/progress.txt
tail -3 $out/pytest.txt
timeout -k 10 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
echo "bench rc=$?" >> $out/progress.txt
timeout -k 10 500 python3 tools/sweep.py 10000 30000 100000 --cpu-max 30000 > $out/sweep_small.md 2> $out/sweep.err
echo "sweep rc=$?" >> $out/progress.txt
