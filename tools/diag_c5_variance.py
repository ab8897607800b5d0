#!/usr/bin/env python3
"""Where the C5 step's time differs from process to process: per pass the stage times, the resident filter kernel's own
clock per launch and the hold-back it ran with.  python tools/diag_c5_variance.py [passes]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from pyfocusr_amd import _hip  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = _hip.default_context()
meshes = [blob_mesh(1000000, seed=s) for s in (0, 1)]
for m in meshes:
    m._pf_device_mesh = _hip.DeviceMesh(m.points, m.faces, ctx=ctx)
np.random.seed(0)
for p in range(passes):
    _hip.persist_clock(ctx, reset=True)
    timers = dict(assembly=0.0, eigensolve=0.0, eigsort=0.0, knn=0.0, matvecs=0)
    t0 = time.perf_counter()
    bench.hot_path_step([ctx, ctx], meshes[0], meshes[1], 10, 5000, timers)
    dt = time.perf_counter() - t0
    ms, n = _hip.persist_clock(ctx)
    st = _hip.persist_state(ctx)
    print("pass %d: %.2f ms  asm %.2f eig %.2f sort %.2f knn %.2f | resident %d launches x %.1f us = %.2f ms, hold %d, matvecs %d" % (
        p, 1e3 * dt, 1e3 * timers["assembly"], 1e3 * timers["eigensolve"], 1e3 * timers["eigsort"], 1e3 * timers["knn"], n,
        1e3 * ms / max(n, 1), ms, st["hold_ticks"], timers["matvecs"]), flush=True)
