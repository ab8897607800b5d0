#!/usr/bin/env python3
"""Eigensolve wall-clock of a mesh pair against the filter's placement and strength (experiment knobs PF_EIGS_CUT,
PF_EIGS_STRENGTH of pf_eigs.hip; defaults 8 and 1.8).  python tools/sweep_filter.py [n] [k]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import _hip  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = _hip.default_context()
meshes = [blob_mesh(n, s) for s in (0, 1)]
devs = [_hip.DeviceLaplacian(m.points, m.faces, ctx=ctx) for m in meshes]
ref = None
print("| cut | strength | ms per pair | degree | filter applications | matvecs | max rel eigenvalue difference to the default |")
print("|---|---|---|---|---|---|---|")
CUTS = os.environ.get("SWEEP_CUTS", "8,12,16,24,6").split(",")
STRENGTHS = os.environ.get("SWEEP_STRENGTHS", "1.8,1.6,2.0,2.4,2.8").split(",")
for cut in CUTS:
    for strength in STRENGTHS:
        os.environ["PF_EIGS_CUT"], os.environ["PF_EIGS_STRENGTH"] = cut, strength
        times = []
        for rep in range(6):
            ctx.sync()
            t0 = time.perf_counter()
            ra, rb = devs[0].eigs_smallest2(devs[1], k + 1, k + 1)
            ctx.sync()
            times.append(time.perf_counter() - t0)
        vals = np.concatenate([ra[0], rb[0]])
        if ref is None:
            ref = vals
        sa, sb = ra[-1], rb[-1]
        print("| %s | %s | %.2f | %d / %d | %d / %d | %d | %.1e |" % (
            cut, strength, 1e3 * np.median(times[1:]), sa["degree"], sb["degree"], sa["outer_steps"], sb["outer_steps"],
            sa["matvecs"] + sb["matvecs"], float(np.max(np.abs(vals - ref) / np.maximum(np.abs(ref), 1e-30)))), flush=True)
