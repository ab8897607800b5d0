#!/usr/bin/env python3
"""The box-hierarchy 1-NN search (pf_knn_tree.hip) against a left-to-right numpy brute force (small random cases: ties,
duplicates, offsets, every depth) and against the grid search (large cases), then timings of both on the inputs that
matter: the spectral coordinates of synthetic blob pairs (250k k=5, 1M k=10 = BASELINE config C5) and uniform / noisy
clouds.  python tools/knn_tree_check.py [--no-c5]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import Graph, _hip, eigsort  # noqa: E402
from pyfocusr_amd.graph import compute_spectra  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

ctx = _hip.default_context()
ctx.timing_enable(True)


def brute(ref, qry):
    idx = np.empty(len(qry), dtype=np.int64)
    d2o = np.empty(len(qry))
    for i, q in enumerate(qry):
        s = np.zeros(len(ref))
        for c in range(ref.shape[1]):
            df = q[c] - ref[:, c]
            s = df * df if c == 0 else s + df * df
        j = int(np.argmin(s))  # first minimum = lowest index on ties
        idx[i], d2o[i] = j, s[j]
    return idx, d2o


rng = np.random.default_rng(11)
n_cases = 0
for rep in range(0 if "--depths" in sys.argv else 60):
    d = int(rng.integers(1, 17))
    n_ref, n_qry = int(rng.integers(1, 6000)), int(rng.integers(1, 400))
    kind = rep % 5
    ref = rng.uniform(-0.5, 0.5, (n_ref, d))
    qry = rng.uniform(-0.5, 0.5, (n_qry, d))
    if kind == 1:  # lattice: many exact ties
        ref, qry = np.round(ref * 4) / 4, np.round(qry * 4) / 4
    elif kind == 2:  # duplicates of references, queries ON references
        ref[n_ref // 2:] = ref[: n_ref - n_ref // 2]
        qry = ref[rng.integers(0, n_ref, n_qry)].copy()
    elif kind == 3:  # far apart
        qry = qry + 7.0
    elif kind == 4:  # a 2-manifold in d dimensions
        u, v = rng.uniform(0, 1, n_ref), rng.uniform(0, 1, n_ref)
        ref = np.stack([np.cos((c + 1) * u * 3) * np.sin((c % 3 + 1) * v * 2) for c in range(d)], axis=1)
        u, v = rng.uniform(0, 1, n_qry), rng.uniform(0, 1, n_qry)
        qry = np.stack([np.cos((c + 1) * u * 3) * np.sin((c % 3 + 1) * v * 2) for c in range(d)], axis=1) + 0.05
    bi, bd = brute(ref, qry)
    for mode in (2, 1):
        ctx.knn_mode(mode)
        gi, gd = ctx.knn1(ref, qry, return_d2=True)
        assert np.array_equal(gi, bi) and np.array_equal(gd, bd), (rep, mode, d, n_ref, n_qry, kind, int(np.sum(gi != bi)))
    n_cases += 1
print("brute-force parity: %d random cases, hierarchy and grid, indices and squared distances bit-identical" % n_cases, flush=True)


def timed(ref, qry, mode, reps=2):
    ctx.knn_mode(mode)
    best = None
    for _ in range(reps):
        ctx.timing(reset=True)
        t0 = time.perf_counter()
        idx, d2 = ctx.knn1(ref, qry, return_d2=True)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return idx, d2, 1e3 * best, ctx.timing()["knn_ms"]


def compare(name, ref, qry, grid_too=True):
    ctx.knn_tree_stats(True)
    ti, td, t_ms, t_k = timed(ref, qry, 2)
    leaves, supers = ctx.knn_tree_stats(False)
    _, _, t_ms, t_k = timed(ref, qry, 2)  # (without the counters)
    line = "| %s | %d x %d | %d | %.2f | %.2f | %.0f | %.1f |" % (name, len(ref), len(qry), ref.shape[1], t_ms, t_k,
                                                            64.0 * leaves / len(qry) * (4 if ref.shape[1] <= 8 else 2) / 1.0, supers / max(len(qry), 1) * (4 if ref.shape[1] <= 8 else 2))
    if grid_too:
        gi, gd, g_ms, g_k = timed(ref, qry, 1)
        assert np.array_equal(gi, ti) and np.array_equal(gd, td), name
        line += " %.2f | %.2f |" % (g_ms, g_k)
    else:
        rows = np.linspace(0, len(qry) - 1, 64).astype(np.int64)
        bi, bd = brute(ref, qry[rows])
        assert np.array_equal(bi, ti[rows]) and np.array_equal(bd, td[rows]), name
        line += " - | - |"
    print(line, flush=True)


print("| case | n_ref x n_qry | d | hierarchy ms (incl. H2D/D2H) | hierarchy events ms | candidates per query group / G | supers opened per query group / G | grid ms | grid events ms |")
print("|---|---|---|---|---|---|---|---|---|")
for n, d in (() if "--depths" in sys.argv else ((250000, 5), (250000, 6), (250000, 8), (250000, 10), (250000, 16))):
    compare("unrelated uniform clouds", np.random.default_rng(0).uniform(-0.5, 0.5, (n, d)), np.random.default_rng(1).uniform(-0.5, 0.5, (n, d)))
if "--depths" not in sys.argv:
    ref = np.random.default_rng(0).uniform(-0.5, 0.5, (250000, 5))
    compare("registered (noise 1e-3)", ref, ref[np.random.default_rng(2).permutation(250000)] + 1e-3 * np.random.default_rng(3).standard_normal((250000, 5)))
    ref = np.random.default_rng(0).uniform(-0.5, 0.5, (1000000, 10))
    compare("noisy copies (1e-3)", ref, ref[np.random.default_rng(2).permutation(1000000)] + 1e-3 * np.random.default_rng(3).standard_normal((1000000, 10)),
            grid_too=False)

sizes = [(250000, 5)] + ([] if "--no-c5" in sys.argv else [(1000000, 10)])
if "--depths" in sys.argv:  # where does the hierarchy start to pay on spectral coordinates?
    sizes = [(250000, 4), (250000, 6), (250000, 7), (250000, 8), (250000, 10), (60000, 6), (15000, 6)]
for n, k in sizes:
    meshes = [blob_mesh(n, seed=s) for s in (0, 1)]
    gs = [Graph(m, n_spectral_features=k, n_rand_samples=5000, ctx=ctx, verbose=False) for m in meshes]
    compute_spectra(gs)
    Q = eigsort(gs[0], gs[1], k, target_as_reference=True).sort_eigenmaps()
    w = Q[:k] * np.max((gs[1].eig_vals[:k], gs[0].eig_vals[:k]), axis=0)
    w = np.exp(-(w**2) / (2 * np.mean(w) ** 2))
    T, S = gs[0].eig_vecs[:, :k] * w[None, :], gs[1].eig_vecs[:, :k] * w[None, :]
    compare("spectral coordinates of a blob pair, weighted", T, S)
    compare("the same, unweighted", np.ascontiguousarray(gs[0].eig_vecs[:, :k]), np.ascontiguousarray(gs[1].eig_vecs[:, :k]))
    for g in gs:
        g.device.close()
ctx.knn_mode(0)
