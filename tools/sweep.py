#!/usr/bin/env python3
"""Size sweep of the hot path: per mesh size, GPU assembly / eigensolve / KNN times,
the filter kernels' time per step - the resident kernel (operators in registers: an EFFECTIVE rate, not a roofline
fraction) and, from one extra solve with the resident path off, the streaming kernel against the HBM roofline - and the
reference's scipy `eigs` call timed on the host for comparison (north_star: "eigenpairs/sec on synthetic meshes of
10k-1M ... as fraction of HBM roofline, next to the reference scipy path").
python tools/sweep.py [--cpu-max 100000] n1 n2 ...   -> markdown table on stdout."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy import sparse  # noqa: E402
from scipy.sparse.linalg import eigs  # noqa: E402  (comparison column: the reference's call, graph.py:372)
from pyfocusr_amd import Graph, _hip, eigsort  # noqa: E402
from pyfocusr_amd.graph import compute_spectra, spectral_knn  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("sizes", type=int, nargs="*", default=[10000, 30000, 100000, 250000, 500000, 1000000])
ap.add_argument("--cpu-max", type=int, default=100000, help="largest size whose scipy eigs column is measured here (250k: bench.py's cpu_baseline leg)")
ap.add_argument("--k", type=int, default=5)
args = ap.parse_args()

ctx = _hip.default_context()
ctx.timing_enable(True)
print("| n | k | assembly ms (pair) | eigensolve ms (pair) | matvecs/mesh | us per step of the pair | filter kernel | effective GB/s (algorithmic bytes / time) | "
      "streaming kernel: us per launch (both graphs) | streaming kernel: algorithmic GB/s | fraction of the 8 TB/s HBM peak | eigsort ms | KNN ms | "
      "eigenpairs/s (pair, all stages) | scipy eigs s/mesh (1 thread) | eigensolve speed-up vs scipy (pair) | max residual |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for n in args.sizes:
    k = args.k if n < 1000000 else 10  # BASELINE config C5: k = 10 at 1M
    meshes = [blob_mesh(n, seed=s) for s in (0, 1)]
    for m in meshes:
        m._pf_device_mesh = _hip.DeviceMesh(m.points, m.faces, ctx=ctx)
    best = None
    for rep in range(3):
        ctx.timing(reset=True)
        t0 = time.perf_counter()
        gs = [Graph(m, n_spectral_features=k, n_rand_samples=5000, ctx=ctx, verbose=False) for m in meshes]
        for g in gs:
            _ = g.device
        ctx.sync()
        t1 = time.perf_counter()
        compute_spectra(gs)
        t2 = time.perf_counter()
        Q = eigsort(gs[0], gs[1], k, target_as_reference=True).sort_eigenmaps()  # focusr.py:515-522
        w = Q[:k] * np.max((gs[1].eig_vals[:k], gs[0].eig_vals[:k]), axis=0)
        w = np.exp(-(w**2) / (2 * np.mean(w) ** 2))
        t2b = time.perf_counter()
        idx = spectral_knn(gs[0], gs[1], k, w)  # focusr.py:351-353 on the device-resident eigenvectors
        assert idx is not None
        t3 = time.perf_counter()
        tm = ctx.timing()
        # graph-steps: a resident launch runs many (the library counts them); a one-step launch advances both graphs
        steps = 2 * (tm["op_launches"] - tm["persist_launches"]) + tm["persist_steps"]
        row = dict(asm=t1 - t0, eig=t2 - t1, sort=t2b - t2, knn=t3 - t2b, us=2e3 * tm["op_ms"] / steps,
                   split=tm["persist_launches"] > 0 and tm["persist_steps"] / max(tm["persist_launches"], 1) < 1.5 * max(g.eigs_stats.degree for g in gs),
                   persist=tm["persist_steps"] > 0,
                   gbs=tm["op_bytes"] / tm["op_ms"] / 1e6, mv=sum(g.eigs_stats.matvecs for g in gs) / 2,
                   res=max(g.eigs_stats.residuals.max() for g in gs))
        for g in gs:
            g.device.close()
        if best is None or row["eig"] < best["eig"]:
            best = row
    # the kernel that does stream the operators through HBM: one extra solve with the resident path off
    _hip.persist_enable(False)
    try:
        ctx.timing(reset=True)
        gs = [Graph(m, n_spectral_features=k, n_rand_samples=5000, ctx=ctx, verbose=False) for m in meshes]
        compute_spectra(gs)
        for g in gs:
            _ = g.eig_vecs
            g.device.close()
        tm = ctx.timing()
        s_us = 1e3 * tm["op_ms"] / max(tm["op_launches"], 1)
        s_gbs = tm["op_bytes"] / max(tm["op_ms"], 1e-9) / 1e6
    finally:
        _hip.persist_enable(True)
    cpu, cpu_s = "", None
    if n <= args.cpu_max:
        dev = _hip.DeviceLaplacian(meshes[0].points, meshes[0].faces, ctx=ctx)
        d = dev.download()
        dev.close()
        W = sparse.csr_matrix((d["w"], d["colidx"], d["rowptr"]), shape=(n, n))
        L = (sparse.diags(1.0 / (d["deg"] + 1e-8)) @ (sparse.diags(d["deg"]) - W)).tocsr()
        t0 = time.perf_counter()
        eigs(L, k=k + 1, sigma=1e-10, which="LM", ncv=4 * (k + 1))
        cpu_s = time.perf_counter() - t0
        cpu = "%.2f" % cpu_s
    total = best["asm"] + best["eig"] + best["sort"] + best["knn"]
    kind = "one step per launch"
    if best["persist"]:
        kind = "resident, one launch per graph" if best["split"] else "resident, both graphs per launch"
    print("| %d | %d | %.2f | %.2f | %d | %.2f | %s | %.0f | %.2f | %.0f | %.2f | %.2f | %.2f | %.1f | %s | %s | %.1e |" % (
        n, k, 1e3 * best["asm"], 1e3 * best["eig"], best["mv"], best["us"], kind, best["gbs"], s_us, s_gbs, s_gbs / 8000.0,
        1e3 * best["sort"], 1e3 * best["knn"], 2 * k / total, cpu, "" if cpu_s is None else "%.0f x" % (2 * cpu_s / best["eig"]),
        best["res"]), flush=True)
    for m in meshes:
        m._pf_device_mesh.close()

# KNN stress input of SURVEY 8d: two unrelated uniform clouds U(-0.5, 0.5)^(N x 5), seeds 0 / 1
print()
print("| KNN stress: U(-0.5,0.5)^(N x d), seeds 0/1 | d | GPU ms (incl. H2D/D2H) | kernel ms | scipy KDTree s (1 thread, 20k-query sample scaled) |")
print("|---|---|---|---|---|")
from scipy.spatial import KDTree  # noqa: E402

for n, d in ((250000, 5), (1000000, 5), (250000, 10)):
    ref = np.random.default_rng(0).uniform(-0.5, 0.5, (n, d))
    qry = np.random.default_rng(1).uniform(-0.5, 0.5, (n, d))
    best = None
    for rep in range(3):
        ctx.timing(reset=True)
        t0 = time.perf_counter()
        idx = ctx.knn1(ref, qry)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
        kern = ctx.timing()["knn_ms"]
    t0 = time.perf_counter()
    tree = KDTree(ref)
    ii = tree.query(qry[:20000])[1]
    cpu = (time.perf_counter() - t0) * n / 20000
    assert np.array_equal(ii, idx[:20000])
    print("| %d x %d | %d | %.1f | %.1f | %.1f |" % (n, n, d, 1e3 * best, kern, cpu), flush=True)
