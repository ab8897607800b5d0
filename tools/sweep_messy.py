#!/usr/bin/env python3
"""The 250k pair with scan defects (bench.messy_250k_pair) against the C driver's filter knobs (asymmetric W).
python tools/sweep_messy.py   (SWEEP_POINTS="12:2.0,8:2.5,...")"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pyfocusr_amd import _hip  # noqa: E402

ctx = _hip.default_context()
print("| cut | strength | ms per pair | breakdown | solver |")
print("|---|---|---|---|---|")
for point in os.environ.get("SWEEP_POINTS", "default").split(","):
    if point != "default":
        os.environ["PF_EIGS_CUT"], os.environ["PF_EIGS_STRENGTH"] = point.split(":")
    r = bench.messy_250k_pair(ctx, check_cpu=False)
    print("| %s | %.2f | %s | %s |" % (point.replace(":", " | "), r["ms"], r.get("breakdown_ms", r.get("breakdown")),
                                    {k: v for k, v in r.items() if k in ("solver_modes", "outer_steps", "degree", "matvecs", "second_passes")}), flush=True)
