#!/bin/bash
# kernel statistics + idle gaps of the 250k pair with scan defects (asymmetric W: restarted Arnoldi): gpurun_out/messy_prof/
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/messy_prof
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/tools/sweep_messy.py > $out/run.txt 2> $out/rocprof.err
cp $(ls $out/stats/*/*kernel_stats.csv | tail -1) $out/kernel_stats.csv
python3 $root/tools/trace_gaps.py $(ls $out/stats/*/*kernel_trace.csv | tail -1) 0.5 > $out/gaps.txt
rm -rf $out/stats
cat $out/run.txt
