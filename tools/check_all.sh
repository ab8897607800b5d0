#!/bin/bash
# full GPU suite, then the bench line (heads and tails of the solves reworked)
set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r_tests.log 2>&1 || { tail -40 gpurun_out/r_tests.log; exit 1; }
tail -3 gpurun_out/r_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r_bench.json 2> gpurun_out/r_bench.err || { tail -20 gpurun_out/r_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, d.get('breakdown_ms'))
for k in ('bundled_15k_pair', 'messy_250k_pair', 'c5_1m_k10'):
    v = d.get(k, {})
    print(k, v.get('ms'), v.get('breakdown_ms'))
PY
