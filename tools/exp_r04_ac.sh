#!/bin/bash
set -e
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/ac_bench.json 2> gpurun_out/ac_bench.err || { tail -20 gpurun_out/ac_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open('gpurun_out/ac_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, d.get('breakdown_ms_per_step'))
print('parity', d.get('parity_at_full_size'))
for k in ('bundled_15k_pair', 'messy_250k_pair', 'c5_1m_k10', 'single_graph_solve'):
    v = d.get(k, {})
    print(k, {kk: vv for kk, vv in v.items() if kk in ('ms', 'breakdown_ms', 'max_eig_residual', 'max_rel_eigenvalue_error_vs_oracle', 'knn_index_mismatches', 'solver_modes', 'outer_steps', 'ratio_to_clean_pair', 'error')})
print('roofline', d.get('roofline'))
print('cpu', d.get('cpu_baseline'))
PY
