#!/usr/bin/env python3
"""Diagnostic: where does the eigsort stage of bench.py's step wait?  (per-call wall time of finalize_wait / final_remap /
eigsort_costs inside bench.hot_path_step, several steps)  python tools/diag_eigsort_wait.py"""
import os
import sys
import time
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import bench  # noqa: E402
from pyfocusr_amd import _hip  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

calls = defaultdict(list)


def wrap(cls, name):
    fn = getattr(cls, name)

    def wrapped(self, *a, **kw):
        t0 = time.perf_counter()
        try:
            return fn(self, *a, **kw)
        finally:
            calls[name].append(1e3 * (time.perf_counter() - t0))
    setattr(cls, name, wrapped)


for nm in ("finalize_wait", "final_remap", "eigs_smallest2"):
    wrap(_hip.DeviceLaplacian, nm)
for nm in ("eigsort_costs", "knn1_graphs"):
    wrap(_hip.Context, nm)
ctx = _hip.Context(0)
meshes = [blob_mesh(250000, seed=s) for s in (0, 1)]
for m in meshes:
    m._pf_device_mesh = _hip.DeviceMesh(m.points, m.faces, ctx=ctx)
np.random.seed(1234)
keep = None
for step in range(8):
    calls.clear()
    timers = dict(assembly=0.0, eigensolve=0.0, eigsort=0.0, knn=0.0, matvecs=0)
    keep = bench.hot_path_step([ctx, ctx], meshes[0], meshes[1], 5, 5000, timers)
    if step >= 3:
        print("eigsort %.2f ms | " % (1e3 * timers["eigsort"]) + " ; ".join("%s %s" % (k, ["%.2f" % x for x in v]) for k, v in calls.items()), flush=True)
