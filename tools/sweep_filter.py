#!/usr/bin/env python3
"""Eigensolve time of the 250k pair over the two knobs of the Chebyshev filter (`_krylov._solve_gen`): the damped
interval's lower end `cut` = mult * (k+1) / n and the `strength` that sets the degree.
python tools/sweep_filter.py [n] [k]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import Graph, _hip  # noqa: E402
from pyfocusr_amd._krylov import drive_pair  # noqa: E402
from pyfocusr_amd.graph import _device_eigs_gen  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = _hip.default_context()
graphs = [Graph(blob_mesh(n, s), n_spectral_features=k, n_rand_samples=5000, ctx=ctx, verbose=False) for s in (0, 1)]
devs = [g.device for g in graphs]


def run(mult, strength):
    def solver(d):
        return lambda: _device_eigs_gen(d, k=k + 1, n_k_needed=k, k_buffer=1, minmax=True, cut=mult * (k + 1) / n, strength=strength)
    t0 = time.perf_counter()
    ra, rb = drive_pair(solver(devs[0]), devs[0], solver(devs[1]), devs[1])
    ctx.sync()
    return time.perf_counter() - t0, ra, rb


base = run(12.0, 2.0)[1][0]
print("%6s %8s %8s %8s %8s %8s %10s" % ("mult", "strength", "ms", "degree", "outer", "matvecs", "max dlam"))
for mult in (6.0, 9.0, 12.0, 16.0, 24.0):
    for strength in (1.5, 2.0, 2.5, 3.0, 4.0):
        run(mult, strength)
        ts = []
        for _ in range(3):
            t, ra, rb = run(mult, strength)
            ts.append(t)
        st = ra[2]
        print("%6.1f %8.2f %8.2f %8d %8d %8d %10.2e" % (mult, strength, 1e3 * np.median(ts), st.degree, st.outer_steps, st.matvecs,
                                                         np.max(np.abs(ra[0][:k] / base[:k] - 1))), flush=True)
