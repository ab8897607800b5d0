#!/usr/bin/env python3
"""Micro-benchmark of the filter application (pf_cheb / pf_cheb2): us per recurrence step with the resident kernel and
one step per launch, for pairs and single graphs.  python tools/bench_cheb.py [--degree 145] [n ...]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import _hip  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("sizes", type=int, nargs="*", default=[250000])
ap.add_argument("--degree", type=int, default=145)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--modes", default="2,1,0", help="2: resident, two steps per exchange; 1: resident, one; 0: one step per launch")
args = ap.parse_args()
ctx = _hip.default_context()
ctx.timing_enable(True)
for n in args.sizes:
    devs = []
    for s in (0, 1):
        m = blob_mesh(n, s)
        d = _hip.DeviceLaplacian(m.points, m.faces, ctx=ctx)
        d.ws_ensure(4)
        d.upload(0, np.random.default_rng(s).standard_normal(n))
        devs.append(d)
    req = (0, 1, args.degree, 1.0001, 0.9999, 1.0)
    for mode in [int(x) for x in args.modes.split(",")]:
        _hip.persist_enable(bool(mode))
        _hip.persist_two_step(2 if mode == 2 else 0)
        for label, fn in (("pair", lambda: devs[0].cheb2(req, devs[1], req)), ("single", lambda: devs[0].cheb(*req))):
            fn()
            ctx.sync()
            ctx.timing(reset=True)
            for _ in range(args.reps):
                fn()
            t = ctx.timing(reset=True)
            steps = args.reps * args.degree
            print("n=%d %s %-6s: %.3f us per step (%d resident launches, %d kernel launches)" % (
                n, {2: "resident x2", 1: "resident", 0: "1 step/launch"}[mode], label, 1e3 * t["op_ms"] / steps, t["persist_launches"],
                t["op_launches"]), flush=True)
    for d in devs:
        d.close()
