#!/bin/bash
set -e
mkdir -p gpurun_out
PF_BENCH_DETAIL=1 PF_DEBUG_BUILD=1 python bench.py --steps 12 --warmup 4 --no-extras --no-cpu-baseline 2> gpurun_out/w_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['ms_per_step'], d['breakdown_ms_per_step']); print({k:v for k,v in d.items() if 'detail' in k})"
tail -5 gpurun_out/w_err.log
