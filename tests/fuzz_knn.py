#!/usr/bin/env python3
"""Randomised sweep of the exact KNN kernel against a left-to-right numpy brute force (indices AND squared distances
bit-exact), over dimensions, sizes, K, alignment regimes, duplicates and degenerate extents.  Not collected by pytest:
python tests/fuzz_knn.py SEED N_CASES   on the GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import _hip  # noqa: E402

ctx = _hip.default_context()
rng = np.random.default_rng(int(sys.argv[1]))
N = int(sys.argv[2])


def brute(ref, qry, K):
    idx = np.empty((len(qry), K), dtype=np.int64)
    d2 = np.empty((len(qry), K))
    for lo in range(0, len(qry), 512):
        q = qry[lo:lo + 512]
        acc = None
        for c in range(ref.shape[1]):  # left to right, separate multiply and add
            df = q[:, None, c] - ref[None, :, c]
            sq = df * df
            acc = sq if acc is None else acc + sq
        order = np.lexsort((np.broadcast_to(np.arange(len(ref)), acc.shape), acc), axis=1)[:, :K]
        idx[lo:lo + 512] = order
        d2[lo:lo + 512] = np.take_along_axis(acc, order, axis=1)
    return idx, d2


fails, t0 = 0, time.time()
for it in range(N):
    d = int(rng.integers(1, 17))
    K = int(rng.integers(1, 5)) if d <= 4 else 1
    n_ref, n_qry = int(rng.integers(K, 6000)), int(rng.integers(1, 3000))
    regime = int(rng.integers(0, 6))
    ref = rng.uniform(-0.5, 0.5, (n_ref, d))
    if regime == 0:
        qry = rng.uniform(-0.5, 0.5, (n_qry, d))                       # unrelated clouds
    elif regime == 1:
        qry = ref[rng.integers(0, n_ref, n_qry)] + 1e-3 * rng.normal(size=(n_qry, d))   # well registered
    elif regime == 2:
        qry = ref[rng.integers(0, n_ref, n_qry)].copy()                # exact duplicates: distance 0, index ties
        ref[: n_ref // 2] = ref[n_ref // 2: n_ref // 2 * 2]            # duplicated references too
    elif regime == 3:
        qry = rng.uniform(5, 6, (n_qry, d))                            # far away from every reference
    elif regime == 4:
        ref[:, rng.integers(0, d)] = 0.25                              # a degenerate axis
        qry = rng.uniform(-0.5, 0.5, (n_qry, d))
    else:
        ref = np.round(ref * 8) / 8                                    # lattice: many exact ties
        qry = np.round(rng.uniform(-0.5, 0.5, (n_qry, d)) * 8) / 8
    idx, d2 = ctx.knn(ref, qry, K)
    widx, wd2 = brute(ref, qry, K)
    if not (np.array_equal(idx.reshape(n_qry, K), widx) and np.array_equal(d2.reshape(n_qry, K), wd2)):
        fails += 1
        print("FAIL d=%d K=%d n_ref=%d n_qry=%d regime=%d: %d index mismatches" % (
            d, K, n_ref, n_qry, regime, int(np.sum(idx.reshape(n_qry, K) != widx))), flush=True)
print("done: %d failures of %d, %.1fs" % (fails, N, time.time() - t0))
