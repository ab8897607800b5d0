#!/bin/bash
# A/B on one box: heads and tails switches
set -e
mkdir -p gpurun_out
run() { python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['breakdown_ms_per_step'].items()})"; }
{
for rep in 1 2; do
echo "## new (all on)"; run
echo "## PF_EXP_NOSTEPWISE"; PF_EXP_NOSTEPWISE=1 run
echo "## PF_EXP_NOWINBEGIN"; PF_EXP_NOWINBEGIN=1 run
echo "## PF_EXP_BUILDSYNC"; PF_EXP_BUILDSYNC=1 run
echo "## old (all three)"; PF_EXP_NOSTEPWISE=1 PF_EXP_NOWINBEGIN=1 PF_EXP_BUILDSYNC=1 run
done
} 2>&1 | tee gpurun_out/s_ab.log
