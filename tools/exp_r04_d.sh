#!/bin/bash
# round 4: the asymmetric branch after the scheduling changes - tests, the extras of the bench line, a kernel trace of the 15k pair
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_d
rm -rf $out && mkdir -p $out
cd $root
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "orth or eigs or spectrum or pair_driver or timeout or open_mesh or messy or end_to_end or large_hole or recursive_eig or paired or multi_gpu" > $out/pytest.txt 2>&1
echo "pytest rc=$?" > $out/progress.txt
tail -3 $out/pytest.txt
python3 tools/profile_15k.py > $out/profile_15k.txt 2>&1
echo "15k rc=$?" >> $out/progress.txt
head -3 $out/profile_15k.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace15 -- python3 $root/tools/profile_15k.py > $out/profile_15k_traced.txt 2>&1
tr=$(ls $out/trace15/*/*kernel_trace.csv | tail -1)
python3 $root/tools/trace_timeline.py $tr > $out/timeline_15k.txt
python3 $root/tools/trace_gaps.py $tr 0.3 > $out/gaps_15k.txt
rm -rf $out/trace15
cd $root
timeout -k 10 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
echo "bench rc=$?" >> $out/progress.txt
