#!/usr/bin/env python3
"""Randomised sweep of `recursive_eig(matrix, ...)` on matrices that do not come from a mesh: Laplacians of random
geometric graphs (unnormalised D - W, random-walk D^-1 (D - W), shifted variants without a null vector), against scipy's
shift-invert `eigsh` / `eigs`.   python tools/fuzz_recursive_eig.py SEED N_CASES"""
import os
import sys
import time
import traceback

import numpy as np
from scipy import sparse
from scipy.sparse.linalg import eigs, eigsh
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import recursive_eig  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]))
N = int(sys.argv[2])
fails, t0 = 0, time.time()
for it in range(N):
    n, nn, k = int(rng.choice([300, 1200, 5000, 20000])), int(rng.integers(4, 9)), int(rng.integers(2, 8))
    P = rng.random((n, 2))
    d, j = cKDTree(P).query(P, k=nn + 1)
    rows = np.repeat(np.arange(n), nn)
    W = sparse.csr_matrix((1.0 / (d[:, 1:].ravel() + 1e-3), (rows, j[:, 1:].ravel())), shape=(n, n))
    W = W.maximum(W.T)  # symmetric weights
    # a few long-range edges keep the graph connected: eigenvalues shared by several components of a matrix WITHOUT
    # null vectors may come back with lower multiplicity (single-vector Krylov; see graph.recursive_eig)
    extra = rng.integers(0, n, size=(max(8, n // 50), 2))
    extra = extra[extra[:, 0] != extra[:, 1]]
    E = sparse.csr_matrix((np.full(len(extra), 0.05), (extra[:, 0], extra[:, 1])), shape=(n, n))
    W = W + E.maximum(E.T)
    deg = np.asarray(W.sum(axis=1))[:, 0]
    kind = int(rng.integers(0, 3))
    if kind == 2 and sparse.csgraph.connected_components(W, directed=False)[0] > 1:
        kind = 0  # see the note above: the shifted (null-vector-free) variant needs a connected graph
    if kind == 0:
        A, sym = (sparse.diags(deg) - W).tocsr(), True
    elif kind == 1:
        A, sym = (sparse.diags(1.0 / deg) @ (sparse.diags(deg) - W)).tocsr(), False
    else:
        A, sym = (sparse.diags(deg) - W + 0.37 * sparse.eye(n)).tocsr(), True  # no null vector: every eigenvalue counts
    label = "n=%d nn=%d k=%d kind=%d" % (n, nn, k, kind)
    try:
        t_case = time.time()
        vals, vecs = recursive_eig(A, k=k + 1, n_k_needed=k)
        t_mine = time.time() - t_case
        vals = np.sort(vals)[:k]
        v0 = np.random.default_rng(0).standard_normal(n)  # (ARPACK's own start vector is unseeded)
        n_comp = sparse.csgraph.connected_components(W, directed=False)[0] if kind != 2 else 0  # that many null vectors
        if sym:
            ref = eigsh(A, k=k + max(n_comp, 1), sigma=-1e-2, which="LM", v0=v0)[0]
        else:
            ref = np.real(eigs(A, k=k + max(n_comp, 1), sigma=-1e-6, which="LM", v0=v0)[0])
        if time.time() - t_case > 5.0:
            print("SLOW %s: device solve %.1fs, scipy %.1fs" % (label, t_mine, time.time() - t_case - t_mine), flush=True)
        ref = np.sort(ref[ref > 1e-10])[:k]
        assert len(vals) >= len(ref) > 0 and np.allclose(vals[: len(ref)], ref, rtol=1e-7), (vals, ref)
    except Exception:
        fails += 1
        print("FAIL %s\n%s" % (label, traceback.format_exc()[-600:]), flush=True)
print("done: %d failures of %d, %.1fs" % (fails, N, time.time() - t0))
