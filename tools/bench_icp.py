#!/usr/bin/env python3
"""ICP timing on a synthetic pair: python tools/bench_icp.py [n_vertices]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import _hip, icp  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
ctx = _hip.default_context()
a, b = blob_mesh(n, seed=1), blob_mesh(n, seed=0)
for rep in range(3):
    t0 = time.perf_counter()
    surf = _hip.DeviceSurface(a.points, a.faces, ctx=ctx)
    t1 = time.perf_counter()
    q = b.points[::n // 1000][:1000]
    for _ in range(20):
        surf.closest(q)
    t2 = time.perf_counter()
    surf.close()
    tr = icp.icp_transform(a.points, a.faces, b.points, ctx=ctx)
    t3 = time.perf_counter()
    print("n=%d: surface build %.2f ms, closest(1000 landmarks) %.3f ms/call, full ICP (100 it) %.1f ms, mean distance %.4f"
          % (n, 1e3 * (t1 - t0), 1e3 * (t2 - t1) / 20, 1e3 * (t3 - t2), tr.mean_distance), flush=True)
