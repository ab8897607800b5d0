#!/bin/bash
# (1) the preamble experiment: base library against the RX_EXP_PRELOAD tuning build, pair launches of degree 8 / 32 / 111
# (2) kernel statistics of the C5 step with the Gram-Schmidt kernels' loads in flight
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_j
rm -rf $out && mkdir -p $out
cd $root
for deg in 8 32 111; do
  bash tools/exp_variants.sh r04_j/deg$deg --degree $deg --reps 40 --modes 1 250000
done
echo "variants done" > $out/progress.txt
cd /tmp && export TMPDIR=/tmp
args="--vertices 1000000 --k 10 --steps 2 --warmup 1 --no-extras --no-cpu-baseline"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py $args > $out/c5_bench_under_rocprof.json 2> $out/rocprof.err
cp $(ls $out/stats/*/*kernel_stats.csv | tail -1) $out/c5_kernel_stats.csv
rm -rf $out/stats
echo "c5 done" >> $out/progress.txt
