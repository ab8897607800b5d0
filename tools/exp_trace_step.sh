#!/bin/bash
# kernel trace of the headline step: gaps and the timeline of the last step -> gpurun_out/$1/
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.err
tr=$(ls $out/trace/*/*kernel_trace.csv | tail -1)
cp $(ls $out/trace/*/*kernel_stats.csv | tail -1) $out/kernel_stats.csv
python3 $root/tools/trace_gaps.py $tr 0.6 > $out/gaps.txt
python3 $root/tools/trace_timeline.py $tr > $out/timeline_last_step.txt
rm -rf $out/trace
