#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_k
rm -rf $out && mkdir -p $out
cd $root
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "orth or full_size_1m or vector_kernels or paired or pair_driver or rowpart" > $out/pytest.txt 2>&1
echo "pytest rc=$?" > $out/progress.txt
tail -3 $out/pytest.txt
cd /tmp && export TMPDIR=/tmp
args="--vertices 1000000 --k 10 --steps 2 --warmup 1 --no-extras --no-cpu-baseline"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py $args > $out/c5_bench_under_rocprof.json 2> $out/rocprof.err
cp $(ls $out/stats/*/*kernel_stats.csv | tail -1) $out/c5_kernel_stats.csv
rm -rf $out/stats
echo "c5 done" >> $out/progress.txt
cd $root
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
echo "bench rc=$?" >> $out/progress.txt
