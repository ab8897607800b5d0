#!/usr/bin/env python3
"""Fine-grained wall-clock breakdown of one bench step (host perf_counter around every sub-stage, medians over
several steps): where the ~30 ms of a 250k pair go.  python tools/time_step_detail.py [n] [k]"""
import os
import sys
import time
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import Graph, _hip, eigsort  # noqa: E402
from pyfocusr_amd.graph import compute_spectra, spectral_knn  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = _hip.default_context()
ctx.timing_enable(True)
meshes = [blob_mesh(n, s) for s in (0, 1)]
for m in meshes:
    m._pf_device_mesh = _hip.DeviceMesh(m.points, m.faces, ctx=ctx)
acc = defaultdict(list)

# wrap the blocking ctypes entry points to see where the host waits
calls = defaultdict(float)
for name in ("orth_end", "orth_begin", "cheb2", "cheb", "finalize_vectors", "combine", "dots", "resnorm", "spmv", "copy",
             "lock_null_vectors", "start_vector", "orth", "scale", "ws_ensure", "sync", "finalize_wait"):
    fn = getattr(_hip.DeviceLaplacian, name)

    def make(fn, name):
        def wrapped(self, *a, **kw):
            t0 = time.perf_counter()
            try:
                return fn(self, *a, **kw)
            finally:
                calls[name] += time.perf_counter() - t0
        return wrapped
    setattr(_hip.DeviceLaplacian, name, make(fn, name))
for name in ("final_rows",):
    setattr(_hip.DeviceLaplacian, name, make(getattr(_hip.DeviceLaplacian, name), name))
for name in ("knn1_graphs", "knn1"):
    setattr(_hip.Context, name, make(getattr(_hip.Context, name), name))


def step():
    T = []
    mark = lambda name: T.append((name, time.perf_counter()))
    mark("start")
    graphs = [Graph(m, n_spectral_features=k, n_rand_samples=5000, ctx=ctx, verbose=False) for m in meshes]
    mark("graph ctor (host geometry)")
    for g in graphs:
        _ = g.device
    ctx.sync()
    mark("assembly (device build)")
    calls.clear()
    compute_spectra(graphs)
    mark("compute_spectra")
    spectra_calls = dict(calls)
    gt, gs = graphs
    es = eigsort(gt, gs, k, target_as_reference=True)
    mark("eigsort ctor (sampling)")
    es.calc_c_lambda()
    mark("c_lambda")
    es.calc_c_hist()
    mark("c_hist")
    es.calc_c_spatial()
    mark("c_spatial")
    es.eigen_sort()
    mark("eigen_sort")
    Q = es.Q
    w = Q[:k] * np.max((gs.eig_vals[:k], gt.eig_vals[:k]), axis=0)
    w = np.exp(-(w**2) / (2 * np.mean(w) ** 2))
    mark("weights")
    calls.clear()
    idx = spectral_knn(gt, gs, k, w)
    assert idx is not None
    mark("knn (device-resident coordinates)")
    knn_calls = dict(calls)
    knn_calls["library's events around pf_knn_run"] = 1e-3 * ctx.timing()["knn_ms"]
    for g in graphs:
        g.device.close()
    mark("close")
    for (a, ta), (b, tb) in zip(T[:-1], T[1:]):
        acc[b].append(tb - ta)
    for name, v in spectra_calls.items():
        acc["  spectra: " + name].append(v)
    for name, v in knn_calls.items():
        acc["  knn: " + name].append(v)
    acc["TOTAL"].append(T[-1][1] - T[0][1])


for _ in range(3):
    step()
acc.clear()
for _ in range(8):
    step()
for name, v in acc.items():
    print("%-36s %8.3f ms" % (name, 1e3 * float(np.median(v))))
