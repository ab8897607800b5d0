"""cProfile of the host side of one hot-path step (eigsort + weights) at bench size; prints the top entries."""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from pyfocusr_amd import Graph, _hip, eigsort  # noqa: E402
from pyfocusr_amd.graph import compute_spectra  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
k = 5
ctx = _hip.Context()
meshes = [blob_mesh(n, seed=s) for s in (1, 0)]
for rep in range(3):
    graphs = [Graph(m, n_spectral_features=k, n_rand_samples=10000, ctx=ctx, verbose=False) for m in meshes]
    compute_spectra(graphs)
    gt, gs = graphs
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    es = eigsort(gt, gs, k, target_as_reference=True)
    Q = es.sort_eigenmaps()
    w = Q[:k] * np.max((gs.eig_vals[:k], gt.eig_vals[:k]), axis=0)
    w = np.exp(-(w**2) / (2 * np.mean(w) ** 2))
    src, tgt = gs.eig_vecs[:, :k] * w[None, :], gt.eig_vecs[:, :k] * w[None, :]
    pr.disable()
    print("rep %d: %.2f ms" % (rep, 1e3 * (time.perf_counter() - t0)))
    for g in graphs:
        g.device.close()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
