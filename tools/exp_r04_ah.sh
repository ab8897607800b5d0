#!/bin/bash
set -e
python -m pytest tests -x -q -m gpu -k "resident or persist or owner or two_contexts or ctx" > gpurun_out/ah_tests.log 2>&1 || { tail -30 gpurun_out/ah_tests.log; exit 1; }
tail -2 gpurun_out/ah_tests.log
run() { python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['breakdown_ms_per_step'].items()})"; }
for rep in 1 2 3; do run; done
