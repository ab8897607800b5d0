#!/usr/bin/env python3
"""Diagnostic (tuning build with -DRX_EXP_STAMPS, PYFOCUSR_HIP_LIB=...): where the cycles of a step of k_cheb_resident<2,1,8>
go, per wave class, from s_memtime stamps accumulated inside the kernel.  python tools/stamp_resident.py [n] [degree]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import _hip  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
degree = int(sys.argv[2]) if len(sys.argv) > 2 else 145
ctx = _hip.default_context()
ctx.timing_enable(True)
devs = []
for s in (0, 1):
    m = blob_mesh(n, s)
    d = _hip.DeviceLaplacian(m.points, m.faces, ctx=ctx)
    d.ws_ensure(4)
    d.upload(0, np.random.default_rng(s).standard_normal(n))
    devs.append(d)
req = (0, 1, degree, 1.0001, 0.9999, 1.0)
_hip.persist_two_step(0)
for _ in range(5):
    devs[0].cheb2(req, devs[1], req)
ctx.sync()
ctx.timing(reset=True)
devs[0].cheb2(req, devs[1], req)
ctx.sync()
t = ctx.timing()
print("pair: %.3f us per step by events" % (1e3 * t["op_ms"] / degree))
lib = _hip.load_library()
buf = (C.c_uint64 * (256 * 16 * 10))()
lib.pf_persist_stamps.restype = C.c_int
assert lib.pf_persist_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16, 10).astype(np.float64) / degree  # cycles per step
nb = (devs[0].n + 4095) // 4096 * 4  # blocks that own windows
a = a[: ((nb + 7) // 8) * 8]
names = ["rows + stores issued", "hold-back", "polls", "store drain", "barrier", "loop overhead", "| hand-off 0 issued at",
         "hand-off 1 issued at", "repeats(lane 0)", "steps with a repeat"]
for label, waves in (("waves 0-3 (boundary rows, pollers of graph 0)", slice(0, 4)), ("waves 4-7 (interior)", slice(4, 8)),
                     ("waves 8-11 (interior, pollers of graph 1)", slice(8, 12)), ("waves 12-15 (interior)", slice(12, 16))):
    sub = a[:, waves, :].reshape(-1, 10)
    tot = sub[:, :6].sum(axis=1).mean()
    print("%-48s total %7.0f cycles/step: " % (label, tot) + ", ".join("%s %.0f" % (nm, v) for nm, v in zip(names, sub.mean(axis=0))))
print("(s_memtime ticks of the shader clock, ~2.4 GHz)" )
