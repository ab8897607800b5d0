#!/bin/bash
# round 4 collection, part C: kernel statistics of the C5 step (1M-vertex pair, k = 10) on one GPU, the size sweep
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/final_c
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="--vertices 1000000 --k 10 --steps 2 --warmup 1 --no-extras --no-cpu-baseline"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py $args > $out/c5_bench_under_rocprof.json 2> $out/rocprof.err
cp $(ls $out/stats/*/*kernel_stats.csv | tail -1) $out/c5_kernel_stats.csv
python3 $root/tools/trace_gaps.py $(ls $out/stats/*/*kernel_trace.csv | tail -1) 0.6 > $out/c5_gaps.txt
python3 $root/tools/trace_timeline.py $(ls $out/stats/*/*kernel_trace.csv | tail -1) > $out/c5_timeline_last_step.txt
rm -rf $out/stats
echo "c5 done" > $out/progress.txt
cd $root
timeout -k 10 700 python3 tools/sweep.py > $out/sweep.md 2> $out/sweep.err
echo "sweep rc=$?" >> $out/progress.txt
