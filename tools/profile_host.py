#!/usr/bin/env python3
"""cProfile of the host side of the bench step (python tools/profile_host.py [steps]): where the interpreter's time goes."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from pyfocusr_amd import _hip  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = _hip.default_context()
meshes = [blob_mesh(250000, s) for s in (0, 1)]
for m in meshes:
    m._pf_device_mesh = _hip.DeviceMesh(m.points, m.faces, ctx=ctx)
timers = dict(assembly=0.0, eigensolve=0.0, eigsort=0.0, knn=0.0, matvecs=0)
np.random.seed(0)
for _ in range(3):
    bench.hot_path_step([ctx, ctx], meshes[0], meshes[1], 5, 5000, timers)
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    bench.hot_path_step([ctx, ctx], meshes[0], meshes[1], 5, 5000, timers)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
