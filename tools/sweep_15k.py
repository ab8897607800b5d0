#!/usr/bin/env python3
"""Bundled 15k pair (asymmetric W) against the C driver's filter knobs and the resident kernel's switches:
python tools/sweep_15k.py   (reads SWEEP_CUTS, SWEEP_STRENGTHS; environment as given)"""
import os
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
print("| cut | strength | ms per pair | assembly | eigensolve | eigsort | knn | matvecs | degree | outer steps | second passes | mode |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for cut in os.environ.get("SWEEP_CUTS", "default").split(","):
    for strength in os.environ.get("SWEEP_STRENGTHS", "default").split(","):
        if cut != "default":
            os.environ["PF_EIGS_CUT"] = cut
        if strength != "default":
            os.environ["PF_EIGS_STRENGTH"] = strength
        timers = dict(assembly=0.0, eigensolve=0.0, eigsort=0.0, knn=0.0, matvecs=0)
        np.random.seed(0)
        for _ in range(3):
            bench.hot_path_step([ctx, ctx], meshes[0], meshes[1], 5, 20000, timers)
        for key in timers:
            timers[key] = 0
        ctx.sync()
        t0 = time.perf_counter()
        R = 8
        for _ in range(R - 1):
            bench.hot_path_step([ctx, ctx], meshes[0], meshes[1], 5, 20000, timers)
        gs = bench.hot_path_step([ctx, ctx], meshes[0], meshes[1], 5, 20000, timers, keep_graphs=True)
        ctx.sync()
        ms = 1e3 * (time.perf_counter() - t0) / R
        st = [g.eigs_stats for g in gs]
        print("| %s | %s | %.2f | %.2f | %.2f | %.2f | %.2f | %d | %s | %s | %s | %s |" % (
            cut, strength, ms, 1e3 * timers["assembly"] / R, 1e3 * timers["eigensolve"] / R, 1e3 * timers["eigsort"] / R,
            1e3 * timers["knn"] / R, timers["matvecs"] // R, " / ".join(str(s.degree) for s in st), " / ".join(str(s.outer_steps) for s in st),
            " / ".join(str(getattr(s, "second_passes", "?")) for s in st), " / ".join(str(getattr(s, "mode", "?")) for s in st)), flush=True)
