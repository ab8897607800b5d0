#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_i
rm -rf $out && mkdir -p $out
cd $root
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "orth or vector_kernels or pair_driver or paired or spectrum_vs or resident_kernel_bit" > $out/pytest.txt 2>&1
echo "pytest rc=$?" > $out/progress.txt
tail -3 $out/pytest.txt
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
echo "bench rc=$?" >> $out/progress.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.err
cp $(ls $out/stats/*/*kernel_stats.csv | tail -1) $out/kernel_stats.csv
rm -rf $out/stats
echo "stats done" >> $out/progress.txt
