#!/bin/bash
# round 4: full GPU test selection, then the bench line (extras, no CPU leg), then the 15k trace
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_e
rm -rf $out && mkdir -p $out
cd $root
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1
echo "pytest rc=$?" > $out/progress.txt
tail -3 $out/pytest.txt
timeout -k 10 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
echo "bench rc=$?" >> $out/progress.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace15 -- python3 $root/tools/profile_15k.py > $out/profile_15k_traced.txt 2>&1
tr=$(ls $out/trace15/*/*kernel_trace.csv | tail -1)
python3 $root/tools/trace_timeline.py $tr > $out/timeline_15k.txt
python3 $root/tools/trace_gaps.py $tr 0.3 > $out/gaps_15k.txt
rm -rf $out/trace15
echo "trace done" >> $out/progress.txt
