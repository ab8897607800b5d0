#!/usr/bin/env python3
"""KNN stress inputs: two unrelated uniform clouds, plus a registered pair (query = reference + noise) and
far-apart clouds; GPU ms and parity against scipy's KDTree on a sample.  python tools/knn_stress.py"""
import os
import sys
import time

import numpy as np
from scipy.spatial import KDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import _hip  # noqa: E402

ctx = _hip.default_context()
ctx.timing_enable(True)
print("| case | n x n | d | GPU ms (incl. H2D/D2H) | events around pf_knn_run ms | KDTree s (1 thread, 5k-query sample scaled) |")
print("|---|---|---|---|---|---|")
cases = []
for n, d in ((250000, 5), (1000000, 5), (250000, 10), (250000, 3), (250000, 2), (250000, 16)):
    cases.append(("unrelated uniform clouds", n, d, lambda n=n, d=d: (np.random.default_rng(0).uniform(-0.5, 0.5, (n, d)),
                                                                      np.random.default_rng(1).uniform(-0.5, 0.5, (n, d)))))
def registered(n, d):
    ref = np.random.default_rng(0).uniform(-0.5, 0.5, (n, d))
    return ref, ref[np.random.default_rng(2).permutation(n)] + 1e-3 * np.random.default_rng(3).standard_normal((n, d))
def apart(n, d):
    ref = np.random.default_rng(0).uniform(-0.5, 0.5, (n, d))
    return ref, np.random.default_rng(1).uniform(-0.5, 0.5, (n, d)) + 5.0
cases.append(("registered (noise 1e-3)", 250000, 5, lambda: registered(250000, 5)))
cases.append(("clouds 5 apart", 100000, 5, lambda: apart(100000, 5)))
for name, n, d, make in cases:
    ref, qry = make()
    best = None
    for rep in range(3):
        ctx.timing(reset=True)
        t0 = time.perf_counter()
        idx = ctx.knn1(ref, qry)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
        kern = ctx.timing()["knn_ms"]
    t0 = time.perf_counter()
    ii = KDTree(ref).query(qry[:5000])[1]
    cpu = (time.perf_counter() - t0) * n / 5000
    assert np.array_equal(ii, idx[:5000]), name
    print("| %s | %d x %d | %d | %.1f | %.1f | %.1f |" % (name, n, n, d, 1e3 * best, kern, cpu), flush=True)
