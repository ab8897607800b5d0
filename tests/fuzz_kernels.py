#!/usr/bin/env python3
"""Randomised sweep of the "next"-row kernels against their CPU restatements: CPD (E-step, affine and deformable
registrations), closest point on a surface, graph mean filter.  Uses oracle/ as the checker, hence lives under tests/
(not collected by pytest):  python tests/fuzz_kernels.py SEED N_ROUNDS   on the GPU box."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cpd_port, icp_port  # noqa: E402
from pyfocusr_amd import _hip, cpd  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

ctx = _hip.default_context()
rng = np.random.default_rng(int(sys.argv[1]))
N = int(sys.argv[2])
fails, t0 = 0, time.time()


def check(label, fn):
    global fails
    try:
        fn()
    except Exception:
        fails += 1
        print("FAIL %s\n%s" % (label, traceback.format_exc()[-500:]), flush=True)


for it in range(N):
    D = int(rng.integers(1, 17))
    Nx, M = int(rng.integers(1, 2500)), int(rng.integers(1, 2500))
    X = rng.normal(size=(Nx, D)) * rng.uniform(0.1, 3)
    Y = rng.normal(size=(M, D)) * rng.uniform(0.1, 3) + rng.uniform(-1, 1)
    s2, w = float(10 ** rng.uniform(-3, 1)), float(rng.choice([0.0, 0.1, 0.5]))

    def estep():
        dev = _hip.DeviceCpd(X, Y, ctx=ctx)
        P1, Pt1, PX = dev.estep(Y, s2, w)
        a, b, c, _ = cpd_port.expectation(X, Y, s2, w)
        dev.close()
        np.testing.assert_allclose(P1, a, rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(Pt1, b, rtol=1e-10, atol=1e-300)
        np.testing.assert_allclose(PX, c, rtol=1e-9, atol=1e-12)
    check("estep N=%d M=%d D=%d s2=%g w=%g" % (Nx, M, D, s2, w), estep)

    if Nx > D + 2 and M > D + 2:
        def affine():
            a = cpd.affine_registration(X=X, Y=Y, max_iterations=6, tolerance=0.0, ctx=ctx).register()[0]
            b = cpd_port.AffineRegistration(X, Y, max_iterations=6, tolerance=0.0).register()[0]
            np.testing.assert_allclose(a, b, atol=1e-7 * max(1.0, np.abs(b).max()))
        check("affine N=%d M=%d D=%d" % (Nx, M, D), affine)

        def deform():
            # kernel width of the order of the cloud: a well-separated leading spectrum.  (beta far below the point
            # spacing makes G ~ I, whose "leading" eigenpairs are an arbitrary choice inside one cluster — nothing to compare)
            scale = float(np.sqrt(np.mean(np.var(Y, axis=0)) * D))
            kw = dict(alpha=float(10 ** rng.uniform(-2, 1)), beta=scale * float(rng.uniform(0.7, 3.0)), num_eig=int(rng.integers(1, 150)),
                      max_iterations=5, tolerance=0.0)
            Ys = Y[: min(M, 600)]
            a = cpd.deformable_registration(X=X, Y=Ys, ctx=ctx, **kw).register()[0]
            b = cpd_port.DeformableRegistration(X, Ys, low_rank=True, **kw).register()[0]
            np.testing.assert_allclose(a, b, atol=1e-6 * max(1.0, np.abs(b).max()), err_msg=repr(kw))
        check("deformable N=%d M=%d D=%d" % (Nx, M, D), deform)

    def closest():
        n = int(rng.integers(4, 3000))
        pts = rng.normal(size=(n, 3)) * rng.uniform(0.01, 100)
        vpf = int(rng.choice([3, 3, 3, 4, 5]))
        faces = rng.integers(0, n, size=(int(rng.integers(1, 6000)), vpf)).astype(np.int32)
        q = rng.normal(size=(int(rng.integers(1, 60)), 3)) * rng.uniform(0.01, 200)
        surf = _hip.DeviceSurface(pts, faces, ctx=ctx)
        cp, face, d2 = surf.closest(q)
        surf.close()
        wcp, wface, wd2 = icp_port.closest_points_on_surface(pts, faces, q)
        assert np.array_equal(d2, wd2) and np.array_equal(face, wface) and np.array_equal(cp, wcp)
    check("closest", closest)

    def meanfilter():
        from scipy import sparse

        m = blob_mesh(int(rng.integers(50, 4000)), seed=int(rng.integers(0, 10**6)))
        dev = _hip.DeviceLaplacian(m.points, m.faces, ctx=ctx)
        d = dev.download()
        W = sparse.csr_matrix((d["w"], d["colidx"], d["rowptr"]), shape=(dev.n, dev.n))
        ncols, iters = int(rng.integers(1, 6)), int(rng.integers(0, 12))
        vals = rng.normal(size=(dev.n, ncols))
        got = dev.mean_filter(vals, iters)
        dev.close()
        A = sparse.diags(1.0 / (1 + np.asarray(W.sum(axis=1))[:, 0])) @ (W + sparse.eye(dev.n))
        want = vals
        for _ in range(iters):
            want = A @ want
        np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-15)
    check("mean filter", meanfilter)

print("done: %d failures in %d rounds, %.1fs" % (fails, N, time.time() - t0))
