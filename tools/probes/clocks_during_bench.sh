#!/bin/bash
# Shader clock and socket power while bench.py's timed loop runs (is the step clock- or power-limited?).
#   bash tools/probes/clocks_during_bench.sh      -> gpurun_out/clocks/
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/clocks
rm -rf $out && mkdir -p $out
rocm-smi --showclocks --showpower --showmaxpower > $out/idle.txt 2>&1
python3 $root/bench.py --steps 1500 --warmup 5 --no-extras --no-cpu-baseline > $out/bench.json 2> $out/bench.err &
pid=$!
sleep 9
for i in 1 2 3 4 5 6; do
    rocm-smi --showclocks --showpower -t > $out/busy_$i.txt 2>&1
    sleep 0.3
done
wait $pid
