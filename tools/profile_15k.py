#!/usr/bin/env python3
"""cProfile of the bundled 15k pair (BASELINE config C2: asymmetric W, Arnoldi path) - where the host time goes."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pyfocusr_amd import PolyMesh, _hip  # noqa: E402

ctx = _hip.default_context()
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
zt, zs = np.load(os.path.join(gold, "target_mesh_15k.npz")), np.load(os.path.join(gold, "source_mesh_15k.npz"))
meshes = [PolyMesh(z["points"], z["faces"]) for z in (zt, zs)]
timers = dict(assembly=0.0, eigensolve=0.0, eigsort=0.0, knn=0.0, matvecs=0)
np.random.seed(0)
for _ in range(2):
    bench.hot_path_step([ctx, ctx], meshes[0], meshes[1], 5, 20000, timers)
for key in timers:
    timers[key] = 0
t0 = time.perf_counter()
for _ in range(5):
    bench.hot_path_step([ctx, ctx], meshes[0], meshes[1], 5, 20000, timers)
print("plain: ms per step %.2f" % (1e3 * (time.perf_counter() - t0) / 5), {k: round(1e3 * v / 5, 2) for k, v in timers.items()})
for key in timers:
    timers[key] = 0
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    bench.hot_path_step([ctx, ctx], meshes[0], meshes[1], 5, 20000, timers)
pr.disable()
print("under cProfile: ms per step %.2f" % (1e3 * (time.perf_counter() - t0) / 3), {k: round(1e3 * v / 3, 2) for k, v in timers.items()})
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
