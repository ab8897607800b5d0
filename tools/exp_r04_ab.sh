#!/bin/bash
set -e
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/ab_trace
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > $out/bench.json 2> $out/rocprof.err
python3 $root/tools/trace_timeline.py $(ls $out/stats/*/*kernel_trace.csv | tail -1) > $out/timeline_last_step.txt
cp $(ls $out/stats/*/*kernel_stats.csv | tail -1) $out/kernel_stats.csv
rm -rf $out/stats
grep -n "k_orth\|k_cheb" $out/timeline_last_step.txt | sed -n 1,70p | cut -c1-140
head -12 $out/kernel_stats.csv | cut -c1-150
