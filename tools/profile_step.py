#!/usr/bin/env python3
"""cProfile of one bench step on the GPU box: where the host thread spends its time (ctypes calls
include the time blocked on the device)."""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pyfocusr_amd import _hip  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
ctx = _hip.default_context()
meshes = [blob_mesh(n, s) for s in (0, 1)]
for m in meshes:
    m._pf_device_mesh = _hip.DeviceMesh(m.points, m.faces, ctx=ctx)
timers = dict(assembly=0.0, eigensolve=0.0, eigsort=0.0, knn=0.0, matvecs=0)
np.random.seed(0)
bench.hot_path_step([ctx, ctx], meshes[0], meshes[1], 5, 5000, timers)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    bench.hot_path_step([ctx, ctx], meshes[0], meshes[1], 5, 5000, timers)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
