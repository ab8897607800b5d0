#!/bin/bash
# round 4 collection, part A (kernel statistics of the headline workload + PMC passes): results in gpurun_out/final_a/
set -e
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/final_a
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="--steps 4 --warmup 2 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py $args > $out/bench_under_rocprof.json 2> $out/rocprof.err
cp $(ls $out/stats/*/*kernel_stats.csv | tail -1) $out/kernel_stats.csv
python3 $root/tools/trace_gaps.py $(ls $out/stats/*/*kernel_trace.csv | tail -1) 0.6 > $out/gaps.txt
python3 $root/tools/trace_timeline.py $(ls $out/stats/*/*kernel_trace.csv | tail -1) > $out/timeline_last_step.txt
rm -rf $out/stats
echo "stats done" > $out/progress.txt
sargs="--steps 1 --warmup 1 --no-extras --no-cpu-baseline"
PF_PERSIST=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_stream -- python3 $root/bench.py $sargs > $out/bench_stream_under_rocprof.json 2>> $out/rocprof.err
cp $(ls $out/stats_stream/*/*kernel_stats.csv | tail -1) $out/stream_kernel_stats.csv
rm -rf $out/stats_stream
echo "stream stats done" >> $out/progress.txt
pargs="--steps 2 --warmup 1 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py $pargs > $out/pmc_fetch.json 2>> $out/rocprof.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/bench.py $pargs > $out/pmc_write.json 2>> $out/rocprof.err
echo "pmc resident done" >> $out/progress.txt
PF_PERSIST=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_stream -- python3 $root/bench.py $sargs > $out/pmc_fetch_stream.json 2>> $out/rocprof.err
PF_PERSIST=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_stream -- python3 $root/bench.py $sargs > $out/pmc_write_stream.json 2>> $out/rocprof.err
echo "pmc streaming done" >> $out/progress.txt
cd $root
python3 tools/pmc_make_summary.py --fetch $out/pmc_fetch --write $out/pmc_write --fetch-stream $out/pmc_fetch_stream --write-stream $out/pmc_write_stream \
    --stats $out/kernel_stats.csv --bench $out/bench_under_rocprof.json --out $out/pmc_summary.json > /dev/null
python3 tools/pmc_list_all.py $out/pmc_fetch $out/pmc_write $out/kernel_stats.csv > $out/pmc_all_kernels.md
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_fetch_stream $out/pmc_write_stream
echo "all done" >> $out/progress.txt
