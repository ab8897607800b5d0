#!/bin/bash
set -e
python - <<'PY' 2>/dev/null
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from pyfocusr_amd import _hip
from pyfocusr_amd.meshgen import blob_mesh
ctx = _hip.default_context()
r = bench.c5_1m_k10(ctx, reps=2)
print('fresh process        ', round(r['ms'], 1), {k: round(v, 1) for k, v in r['breakdown_ms'].items()})
meshes = [blob_mesh(250000, s) for s in (0, 1)]
for m in meshes:
    m._pf_device_mesh = _hip.DeviceMesh(m.points, m.faces, ctx=ctx)
timers = dict(assembly=0.0, eigensolve=0.0, eigsort=0.0, knn=0.0, matvecs=0)
for _ in range(10):
    bench.hot_path_step([ctx, ctx], meshes[0], meshes[1], 5, 5000, timers)
r = bench.c5_1m_k10(ctx, reps=2)
print('after ten 250k steps ', round(r['ms'], 1), {k: round(v, 1) for k, v in r['breakdown_ms'].items()})
print(_hip.persist_state(ctx))
r = bench.messy_250k_pair(ctx, check_cpu=False)
r = bench.c5_1m_k10(ctx, reps=2)
print('after the messy pair ', round(r['ms'], 1), {k: round(v, 1) for k, v in r['breakdown_ms'].items()})
print(_hip.persist_state(ctx))
PY
