#!/usr/bin/env python3
"""cProfile of the eigsort + KNN stages of the 250k bench step (host side): python tools/profile_eigsort.py [n]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import Graph, _hip, eigsort  # noqa: E402
from pyfocusr_amd.graph import compute_spectra, spectral_knn  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
k = 5
ctx = _hip.default_context()
meshes = [blob_mesh(n, s) for s in (0, 1)]
for m in meshes:
    m._pf_device_mesh = _hip.DeviceMesh(m.points, m.faces, ctx=ctx)
pr = cProfile.Profile()
total = 0.0
np.random.seed(1234)
for it in range(8):
    graphs = [Graph(m, n_spectral_features=k, n_rand_samples=5000, ctx=ctx, verbose=False) for m in meshes]
    compute_spectra(graphs)
    gt, gs = graphs
    if it >= 3:
        pr.enable()
    t0 = time.perf_counter()
    Q = eigsort(gt, gs, k, target_as_reference=True).sort_eigenmaps()
    w = Q[:k] * np.max((gs.eig_vals[:k], gt.eig_vals[:k]), axis=0)
    w = np.exp(-(w**2) / (2 * np.mean(w) ** 2))
    idx = spectral_knn(gt, gs, k, w)
    if it >= 3:
        total += time.perf_counter() - t0
        pr.disable()
    for g in graphs:
        g.device.close()
print("eigsort + knn: %.2f ms per step (under cProfile)" % (1e3 * total / 5))
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
