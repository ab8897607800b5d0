#!/usr/bin/env python3
"""How the 1-NN stage's time depends on the coordinate weights (which pick the two grid axes: the widest extents).
One 250k pair solved once; then eigsort's weights from fresh row samples, and forced weight orders.
python tools/probes/knn_axes.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401,E402
from pyfocusr_amd import Graph, _hip, eigsort  # noqa: E402
from pyfocusr_amd.graph import build_devices, compute_spectra, spectral_knn  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

k = 5
ctx = _hip.Context(0)
meshes = [blob_mesh(250000, seed=s) for s in (0, 1)]
for m in meshes:
    m._pf_device_mesh = _hip.DeviceMesh(m.points, m.faces, ctx=ctx)
np.random.seed(1234)
graphs = [Graph(m, n_spectral_features=k, n_rand_samples=5000, ctx=ctx, verbose=False) for m in meshes]
build_devices(graphs)
compute_spectra(graphs)
gt, gs = graphs


def timed_knn(w, reps=4):
    best = 1e9
    for _ in range(reps):
        ctx.sync()
        t0 = time.perf_counter()
        idx = spectral_knn(gt, gs, k, w)
        best = min(best, time.perf_counter() - t0)
    return 1e3 * best, idx


Q = eigsort(gt, gs, k, target_as_reference=True).sort_eigenmaps()
print("| weights | two widest | 1-NN stage ms (best of 4) |")
print("|---|---|---|")
for trial in range(8):
    for g in graphs:
        g.rand_idxs = g.get_list_rand_idxs(5000)
    es = eigsort(gt, gs, k, target_as_reference=True)
    Q = es.sort_eigenmaps()
    w = Q[:k] * np.max((gs.eig_vals[:k], gt.eig_vals[:k]), axis=0)
    w = np.exp(-(w**2) / (2 * np.mean(w) ** 2))
    ms, _ = timed_knn(w)
    print("| %s | %s | %.3f |" % (np.array2string(w, precision=3), sorted(np.argsort(-w)[:2].tolist()), ms))
base = w.copy()
ref_idx = None
for pair in ((0, 1), (0, 2), (1, 2), (0, 3), (0, 4), (1, 3), (2, 3), (3, 4)):
    # the same weights, two of them nudged to the top: which axes the grid takes, everything else equal
    w2 = base.copy()
    top = base.max()
    w2[list(pair)] = top * np.array([1.02, 1.01])
    ms, idx = timed_knn(w2)
    print("| forced %s | %s | %.3f |" % (np.array2string(w2, precision=3), list(pair), ms))
