#!/bin/bash
set -e
mkdir -p gpurun_out
python tools/exp_r04_y.py 103 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/aa_fuzz.log
python -m pytest tests -x -q -m gpu > gpurun_out/r_tests.log 2>&1 || { tail -40 gpurun_out/r_tests.log; exit 1; }
tail -3 gpurun_out/r_tests.log
run() { python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['breakdown_ms_per_step'].items()}, 'matvecs', d['matvecs_per_step'], 'resid %.1e' % d['max_eig_residual'])"; }
for rep in 1 2; do
echo "## partial reorthogonalisation"; run
echo "## PF_EIGS_PRO=0"; PF_EIGS_PRO=0 run
done 2>&1 | tee gpurun_out/aa_ab.log
