#!/bin/bash
# Only the PMC passes of tools/collect_profiles.sh (steps 4-5), summarised against a kernel-stats file and the bench line
# of the profiled run that are already in profiles/:   bash tools/collect_pmc.sh r03_e
set -e
tag=${1:-r03_e}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pargs="--steps 2 --warmup 1 --no-extras --no-cpu-baseline"
sargs="--steps 1 --warmup 1 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py $pargs > $out/pmc_fetch.json 2> $out/rocprof.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/bench.py $pargs > $out/pmc_write.json 2>> $out/rocprof.err
PF_PERSIST=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_stream -- python3 $root/bench.py $sargs > $out/pmc_fetch_stream.json 2>> $out/rocprof.err
PF_PERSIST=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_stream -- python3 $root/bench.py $sargs > $out/pmc_write_stream.json 2>> $out/rocprof.err
cd $root
python3 tools/pmc_make_summary.py --fetch $out/pmc_fetch --write $out/pmc_write --fetch-stream $out/pmc_fetch_stream --write-stream $out/pmc_write_stream \
    --stats profiles/${tag}_kernel_stats.csv --bench profiles/${tag}_bench_under_rocprof.json --out $out/pmc_summary.json > /dev/null
python3 tools/pmc_list_all.py $out/pmc_fetch $out/pmc_write profiles/${tag}_kernel_stats.csv > $out/all_kernels.md
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_fetch_stream $out/pmc_write_stream
