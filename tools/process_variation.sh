#!/bin/bash
# process-to-process variation of the bench step: 8 processes, breakdown + the pair build's own report
set -e
mkdir -p gpurun_out
for i in 1 2 3 4 5 6 7 8; do
  PF_DEBUG_BUILD=1 python bench.py --steps 12 --warmup 4 --no-extras --no-cpu-baseline 2> gpurun_out/t_err_$i.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$i: %.3f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['breakdown_ms_per_step'].items()})"
  grep "pf_build2" gpurun_out/t_err_$i.log | tail -3 | cut -c1-200
done 2>&1 | tee gpurun_out/t_var.log
rm -f gpurun_out/t_err_*.log
