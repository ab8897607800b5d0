#!/bin/bash
# the local Gram-Schmidt step in one launch (k_orth_local) against two launches (PF_ORTH_LOCAL=0): tests, then the pair
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "orth or eigs or spectr or pair_driver or timeout or partial" > gpurun_out/ol_tests.log 2>&1 || { tail -30 gpurun_out/ol_tests.log; exit 1; }
tail -2 gpurun_out/ol_tests.log
run() { python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['breakdown_ms_per_step'].items()}, 'resid %.2e' % d['max_eig_residual'])"; }
for rep in 1 2 3; do
echo "## one launch"; run
echo "## PF_ORTH_LOCAL=0"; PF_ORTH_LOCAL=0 run
done
