#!/usr/bin/env python3
"""us per recurrence step of the resident kernels (one row per thread) against the hold-back of the first fetch of a step
(PF_PERSIST_HOLD: 10 ns ticks after the step began; 0 = the fixed s_sleep of round 2; "table" = the built-in table).
python tools/tune_hold.py [--holds 0,1800,...] [n ...]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import _hip  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("sizes", type=int, nargs="*", default=[250000])
ap.add_argument("--holds", default="table,0,60,65,70,75,80,85,90,95")
ap.add_argument("--degree", type=int, default=145)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--two-step", action="store_true", help="the two-steps-per-exchange kernels (PF_PERSIST_HOLD2) instead")
args = ap.parse_args()
ctx = _hip.default_context()
ctx.timing_enable(True)
_hip.persist_two_step(2 if args.two_step else 0)
VAR = "PF_PERSIST_HOLD2" if args.two_step else "PF_PERSIST_HOLD"
holds = args.holds.split(",")
print("| n | kernel | " + " | ".join(str(h) for h in holds) + " |")
print("|---|---|" + "---|" * len(holds))
for n in args.sizes:
    devs = []
    for s in (0, 1):
        m = blob_mesh(n, s)
        d = _hip.DeviceLaplacian(m.points, m.faces, ctx=ctx)
        d.ws_ensure(4)
        d.upload(0, np.random.default_rng(s).standard_normal(n))
        devs.append(d)
    req = (0, 1, args.degree, 1.0001, 0.9999, 1.0)
    for label, halves, fn in (("pair, halves", 1, lambda: devs[0].cheb2(req, devs[1], req)),
                              ("pair", 0, lambda: devs[0].cheb2(req, devs[1], req)), ("single", 0, lambda: devs[0].cheb(*req))):
        _hip.persist_pair_halves(bool(halves))
        cells = []
        for h in holds:
            if h == "table":
                os.environ.pop(VAR, None)
            else:
                os.environ[VAR] = h
            for _ in range(3):  # (a graph's rings for two steps per exchange are built at its third application)
                fn()
            ctx.sync()
            ctx.timing(reset=True)
            for _ in range(args.reps):
                fn()
            t = ctx.timing(reset=True)
            cells.append("%.3f" % (1e3 * t["op_ms"] / (args.reps * args.degree)))
        print("| %d | %s | " % (n, label) + " | ".join(cells) + " |", flush=True)
    for d in devs:
        d.close()
