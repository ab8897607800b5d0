#!/usr/bin/env python3
"""Diagnostic: where does the KNN time go at config C5 (1M vertices, k=10)?  Prints the spectral weights, the
distribution of nearest-neighbour distances relative to the coordinate extents and the KNN kernel time; also
times the GPU ICP on a 250k pair.  python tools/diag_c5_knn.py [n] [k]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import Graph, _hip, eigsort, icp  # noqa: E402
from pyfocusr_amd.graph import compute_spectra  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = _hip.default_context()
ctx.timing_enable(True)

m250 = [blob_mesh(250000, seed=s) for s in (0, 1)]
for rep in range(2):
    t0 = time.perf_counter()
    tr = icp.icp_transform(m250[0].points, m250[0].faces, m250[1].points, ctx=ctx)
    print("ICP 250k pair, 100 iterations x 1000 landmarks: %.1f ms (mean distance %.4f)" % (1e3 * (time.perf_counter() - t0), tr.mean_distance), flush=True)

meshes = [blob_mesh(n, seed=s) for s in (0, 1)]
gs = [Graph(m, n_spectral_features=k, n_rand_samples=5000, ctx=ctx, verbose=False) for m in meshes]
compute_spectra(gs)
print("eig_vals target", gs[0].eig_vals)
print("eig_vals source", gs[1].eig_vals)
Q = eigsort(gs[0], gs[1], k, target_as_reference=True).sort_eigenmaps()
w = Q[:k] * np.max((gs[1].eig_vals[:k], gs[0].eig_vals[:k]), axis=0)
w = np.exp(-(w**2) / (2 * np.mean(w) ** 2))
print("Q", Q, "\nweights", w, flush=True)
for label, wt in (("weighted", w), ("unweighted", np.ones(k))):
    T, S = gs[0].eig_vecs[:, :k] * wt[None, :], gs[1].eig_vecs[:, :k] * wt[None, :]
    for rep in range(2):
        ctx.timing(reset=True)
        t0 = time.perf_counter()
        idx, d2 = ctx.knn(T, S, 1)
        t1 = time.perf_counter()
    r = np.sqrt(d2.ravel())
    ext = np.sort(np.ptp(T, axis=0))[::-1]
    print("%s: knn %.1f ms (kernel %.1f ms); extents (sorted) %s" % (label, 1e3 * (t1 - t0), ctx.timing()["knn_ms"], np.round(ext, 3)))
    print("   NN distance quantiles 50/90/99/99.9/max: %s ; typical spacing ~ %.4f" % (
        np.round(np.quantile(r, [0.5, 0.9, 0.99, 0.999, 1.0]), 5), ext[0] / np.sqrt(n)), flush=True)
