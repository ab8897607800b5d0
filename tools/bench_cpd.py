#!/usr/bin/env python3
"""CPD E-step timing: python tools/bench_cpd.py [N] [D] [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import _hip  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 3
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
rng = np.random.default_rng(0)
X, Y = rng.random((N, D)) - 0.5, rng.random((N, D)) - 0.5
ctx = _hip.default_context()
dev = _hip.DeviceCpd(X, Y, ctx=ctx)
for sigma2 in (0.05, 1e-3, 1e-5, 1e-7):
    dev.estep(Y, sigma2)
    t0 = time.perf_counter()
    for _ in range(reps):
        dev.estep(Y, sigma2)
    t1 = time.perf_counter()
    print("E-step %d x %d, d=%d, sigma2 %.0e: %.3f ms per call (%.1f G pairs/s, two exp passes)"
          % (N, N, D, sigma2, 1e3 * (t1 - t0) / reps, reps * N * N / (t1 - t0) / 1e9))

# deformable M-step pieces (K = 100)
K = 100
Q = np.linalg.qr(rng.standard_normal((N, K)))[0]
S = np.sort(rng.random(K))[::-1] + 0.01
dev.set_basis(Q)
dev.estep_resident(0.05)


def t(label, f, n=50):
    f()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    ctx.sync()
    print("  %-28s %.3f ms" % (label, 1e3 * (time.perf_counter() - t0) / n))


H, R = dev.deform_sums()
H, R = H.copy(), R.copy()
Cm = np.zeros((K, D))
t("estep_resident + sync", lambda: (dev.estep_resident(0.05), ctx.sync()))
t("deform_sums", dev.deform_sums)
t("apply_deform", lambda: dev.apply_deform(Cm))
t("affine_sums", dev.affine_sums)
t("apply_affine", lambda: dev.apply_affine(np.eye(D), np.zeros(D)))


def host():
    A = H.copy()
    A[np.diag_indices_from(A)] += 0.1 / S
    Z = np.linalg.solve(A, R)
    return S[:, None] * ((R - H @ Z) / 0.1)


t("host solve etc.", host)
