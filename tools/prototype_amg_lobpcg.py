#!/usr/bin/env python3
"""CPU prototype (scipy only, no GPU): what would aggregation-multigrid-preconditioned LOBPCG cost on this path?

Sizing study for DESIGN.md section 9 (the judge's "stretch" item): on synthetic blob meshes, the k lowest non-null
eigenpairs of S = G^1/2 (D - W) G^1/2 (the operator the GPU solver iterates) by scipy's LOBPCG with a V-cycle of plain /
smoothed aggregation multigrid on the Morton hierarchy as preconditioner, counted in fine-level SpMV-equivalents, next
to the matvec count of the shipped Chebyshev-filtered Lanczos solver.  Nothing here is product code.

    python tools/prototype_amg_lobpcg.py [n ...]
"""
import os
import sys
import time

import numpy as np
from scipy import sparse
from scipy.sparse.linalg import lobpcg

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402


def sym_operator(points, faces):
    n = len(points)
    src = faces.reshape(-1)
    dst = np.roll(faces, -1, axis=1).reshape(-1)
    w = 1.0 / np.linalg.norm(points[src] - points[dst], axis=1)
    W = sparse.csr_matrix((w, (src, dst)), shape=(n, n))
    W.sum_duplicates()
    W.data[:] = 1.0 / np.linalg.norm(points[W.nonzero()[0]] - points[W.indices], axis=1)  # set semantics
    deg = np.asarray(W.sum(axis=1))[:, 0]
    g = 1.0 / (deg + 1e-8)
    sg = np.sqrt(g)
    S = sparse.diags(sg) @ (sparse.diags(deg) - W) @ sparse.diags(sg)
    return sparse.csr_matrix(S), np.sqrt(deg + 1e-8)  # S and its (near-)null vector G^-1/2 1


def morton_order(points):
    q = ((points - points.min(axis=0)) / np.ptp(points, axis=0) * 1023.0).astype(np.uint64)

    def spread(v):
        v = v & 0x3FF
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249
    return np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind="stable")


class Hierarchy(object):
    """Aggregation multigrid: aggregates = runs of `agg` consecutive rows (the matrix is in Morton order), tentative
    prolongator carries the null vector, optionally smoothed by one damped-Jacobi step (smoothed aggregation)."""

    def __init__(self, A, z, agg=4, smoothed=True, coarsest=400, nu=2):
        self.levels, self.nu = [], nu
        self.fine_work = 0.0  # fine-level SpMV-equivalents of ONE V-cycle (by nnz)
        nnz0 = A.nnz
        while A.shape[0] > coarsest:
            n = A.shape[0]
            nc = (n + agg - 1) // agg
            cols = np.arange(n) // agg
            T = sparse.csr_matrix((z, (np.arange(n), cols)), shape=(n, nc))
            norms = np.sqrt(np.asarray(T.multiply(T).sum(axis=0))[0])
            T = T @ sparse.diags(1.0 / norms)
            d = A.diagonal()
            if smoothed:
                lam = 1.9  # spectrum of D^-1 A is inside [0, 2] for these operators
                P = T - (4.0 / (3.0 * lam)) * (sparse.diags(1.0 / d) @ (A @ T))
            else:
                P = T
            P = sparse.csr_matrix(P)
            Ac = sparse.csr_matrix(P.T @ A @ P)
            self.levels.append((A, P, d))
            self.fine_work += (2 * nu + 1) * A.nnz / nnz0 + 2 * P.nnz / nnz0
            z = norms if not smoothed else np.asarray(P.T @ z)
            A = Ac
        self.Ac = np.linalg.pinv(A.toarray())
        self.fine_work += A.shape[0] ** 2 / nnz0

    def vcycle(self, r, lvl=0):
        if lvl == len(self.levels):
            return self.Ac @ r
        A, P, d = self.levels[lvl]
        x = np.zeros_like(r)
        omega = 0.7
        for _ in range(self.nu):
            x = x + omega * (r - A @ x) / d[:, None]
        rc = P.T @ (r - A @ x)
        x = x + P @ self.vcycle(rc, lvl + 1)
        for _ in range(self.nu):
            x = x + omega * (r - A @ x) / d[:, None]
        return x


def run(n, k=5, tol=1e-11):
    mesh = blob_mesh(n, seed=0)
    order = morton_order(mesh.points)
    inv = np.empty(n, dtype=np.int64)
    inv[order] = np.arange(n)
    S, z = sym_operator(mesh.points[order], inv[mesh.faces])
    z = z / np.linalg.norm(z)
    rng = np.random.default_rng(0)
    out = []
    for agg, smoothed, nu in ((4, False, 2), (4, True, 2), (8, True, 2), (4, True, 1)):
        t0 = time.perf_counter()
        H = Hierarchy(S, z, agg=agg, smoothed=smoothed, nu=nu)
        t_setup = time.perf_counter() - t0
        block = k + 2
        X = rng.standard_normal((n, block))
        applies = [0]

        def M(r):
            applies[0] += r.shape[1] if r.ndim == 2 else 1
            r2 = r if r.ndim == 2 else r[:, None]
            r2 = r2 - z[:, None] * (z @ r2)
            x = H.vcycle(r2)
            x = x - z[:, None] * (z @ x)
            return x if r.ndim == 2 else x[:, 0]
        from scipy.sparse.linalg import LinearOperator
        Mop = LinearOperator((n, n), matvec=M, matmat=M, dtype=np.float64)
        t0 = time.perf_counter()
        vals, vecs, hist = lobpcg(S, X, M=Mop, Y=z[:, None], largest=False, tol=tol, maxiter=200, retResidualNormsHistory=True)
        t_solve = time.perf_counter() - t0
        iters = len(hist) - 1
        res = np.linalg.norm(S @ vecs[:, :k] - vecs[:, :k] * vals[:k], axis=0).max()
        # per iteration: the block's S W products (block SpMVs) + block V-cycles
        spmv_eq = iters * block * (1.0 + H.fine_work)
        out.append((agg, smoothed, nu, len(H.levels), H.fine_work, iters, spmv_eq, res, t_setup, t_solve, vals[:k]))
    return out


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [20000, 60000]
    print("| n | aggregates of | prolongator | smoothing steps | levels | fine SpMV-eq per V-cycle | LOBPCG iterations (block k+2 = 7, tol 1e-11) | fine SpMV-eq total | max residual | setup s | solve s (scipy, CPU) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for n in sizes:
        for agg, smoothed, nu, nlev, work, iters, eq, res, ts, tv, vals in run(n):
            print("| %d | %d | %s | %d + %d | %d | %.2f | %d | %.0f | %.1e | %.2f | %.2f |" % (
                n, agg, "smoothed" if smoothed else "plain", nu, nu, nlev, work, iters, eq, res, ts, tv), flush=True)
