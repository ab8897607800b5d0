#!/usr/bin/env python3
"""cProfile of compute_spectra on the 250k pair: how much host time one outer step of the pair driver costs (it has to
stay below the device's ~0.25 ms per step, or the device starves).  python tools/profile_spectra.py [n]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import Graph, _hip  # noqa: E402
from pyfocusr_amd.graph import compute_spectra  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
ctx = _hip.default_context()
meshes = [blob_mesh(n, s) for s in (0, 1)]
for m in meshes:
    m._pf_device_mesh = _hip.DeviceMesh(m.points, m.faces, ctx=ctx)
pr = cProfile.Profile()
tot = 0.0
for it in range(8):
    graphs = [Graph(m, n_spectral_features=5, n_rand_samples=5000, ctx=ctx, verbose=False) for m in meshes]
    for g in graphs:
        _ = g.device
    ctx.sync()
    if it >= 3:
        pr.enable()
    t0 = time.perf_counter()
    compute_spectra(graphs)
    if it >= 3:
        tot += time.perf_counter() - t0
        pr.disable()
    for g in graphs:
        g.device.close()
print("compute_spectra: %.2f ms per pair (under cProfile)" % (1e3 * tot / 5))
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(25)
