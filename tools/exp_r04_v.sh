#!/bin/bash
# timeline of one step with the shared-launch assembly
set -e
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/v_trace
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > $out/bench.json 2> $out/rocprof.err
python3 $root/tools/trace_timeline.py $(ls $out/stats/*/*kernel_trace.csv | tail -1) > $out/timeline_last_step.txt
python3 $root/tools/trace_gaps.py $(ls $out/stats/*/*kernel_trace.csv | tail -1) 0.6 > $out/gaps.txt
rm -rf $out/stats
head -75 $out/timeline_last_step.txt | cut -c1-150
