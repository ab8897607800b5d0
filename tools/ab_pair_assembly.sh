#!/bin/bash
# pair assembly: parity tests first, then A/B on one box (fork of the independent chains on / off, two-stream form)
set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -k "graph or assembl or golden or build or mesh or pair or soup or download or spectrum" > gpurun_out/u_tests.log 2>&1 || { tail -40 gpurun_out/u_tests.log; exit 1; }
tail -3 gpurun_out/u_tests.log
run() { python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['breakdown_ms_per_step'].items()})"; }
{
for rep in 1 2 3; do
echo "## shared launches, fork"; run
echo "## shared launches, no fork"; PF_BUILD_FORK=0 run
echo "## two streams"; PF_PAIR_BUILD_STREAMS=1 run
done
} 2>&1 | tee gpurun_out/u_ab.log
