#!/usr/bin/env python3
"""Micro-benchmark of the fused SpMV/Chebyshev kernel: us per launch and algorithmic GB/s
(HIP events on the ctx stream) for synthetic blob meshes.  python tools/bench_spmv.py [n ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import _hip  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

ctx = _hip.default_context()
ctx.timing_enable(True)
for n in [int(a) for a in sys.argv[1:]] or [250000]:
    m = blob_mesh(n, 0)
    dev = _hip.DeviceLaplacian(m.points, m.faces, ctx=ctx)
    dev.ws_ensure(4)
    dev.upload(0, np.random.default_rng(0).standard_normal(n))
    dev.cheb(0, 1, 200, 1.0001, 0.9999)
    best = 1e9
    for rep in range(5):
        ctx.timing(reset=True)
        dev.cheb(0, 1, 400, 1.0001, 0.9999)
        t = ctx.timing()
        best = min(best, 1e3 * t["op_ms"] / t["op_launches"])
    nbytes = 12 * dev.nnz_l + 20 * n + 4
    print("n=%d nnz_l=%d sell_entries=%d (padding %.1f%%)  %.3f us/launch  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
        n, dev.nnz_l, dev.info.sell_entries, 100.0 * (dev.info.sell_entries / dev.nnz_w - 1), best,
        nbytes / best / 1e3, nbytes / best / 1e3 / 80.0))
    # two graphs per launch (pf_cheb2), as the pair solver issues them
    m2 = blob_mesh(n, 1)
    dev2 = _hip.DeviceLaplacian(m2.points, m2.faces, ctx=ctx)
    dev2.ws_ensure(4)
    dev2.upload(0, np.random.default_rng(1).standard_normal(n))
    req = (0, 1, 400, 1.0001, 0.9999, 1.0)
    dev.cheb2(req, dev2, req)
    best2 = 1e9
    for rep in range(5):
        ctx.timing(reset=True)
        dev.cheb2(req, dev2, req)
        t = ctx.timing()
        best2 = min(best2, 1e3 * t["op_ms"] / t["op_launches"])
    nbytes2 = nbytes + 12 * dev2.nnz_l + 20 * n + 4
    print("   paired: %.3f us/launch  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (best2, nbytes2 / best2 / 1e3, nbytes2 / best2 / 1e3 / 80.0))
    dev.close()
    dev2.close()
