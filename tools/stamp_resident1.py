#!/usr/bin/env python3
"""Diagnostic (tuning build with -DRX_EXP_STAMPS, PYFOCUSR_HIP_LIB=...): where the cycles of a step of the single-graph
resident kernel go (k_cheb_resident<1,NW,..>), per wave, from s_memtime stamps accumulated inside the kernel.
python tools/stamp_resident1.py [n] [degree]   (PF_PERSIST_T512 selects the block shape)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import _hip  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
degree = int(sys.argv[2]) if len(sys.argv) > 2 else 169
ctx = _hip.default_context()
ctx.timing_enable(True)
m = blob_mesh(n, 0)
d = _hip.DeviceLaplacian(m.points, m.faces, ctx=ctx)
d.ws_ensure(4)
d.upload(0, np.random.default_rng(0).standard_normal(n))
_hip.persist_two_step(0)
for _ in range(5):
    d.cheb(0, 1, degree, 1.0001, 0.9999, 1.0)
ctx.sync()
ctx.timing(reset=True)
d.cheb(0, 1, degree, 1.0001, 0.9999, 1.0)
ctx.sync()
t = ctx.timing()
print("single graph, %d rows: %.3f us per step by events (%d resident launches)" % (n, 1e3 * t["op_ms"] / degree, t["persist_launches"]))
lib = _hip.load_library()
buf = (C.c_uint64 * (256 * 16 * 10))()
lib.pf_persist_stamps.restype = C.c_int
assert lib.pf_persist_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16, 10).astype(np.float64) / degree  # cycles per step
names = ["rows", "hold", "polls", "drain", "barrier", "loop", "| hand-off at", "(g1)", "repeats(lane 0)", "steps with a repeat"]
print("wave: " + ", ".join(names))
for w in range(16):
    sub = a[:, w, :]
    if sub[:, :6].sum() == 0:
        continue
    print("%2d  total %6.0f: " % (w, sub[:, :6].sum(axis=1).mean()) + ", ".join("%7.1f" % v for v in sub.mean(axis=0)))
print("(s_memtime ticks: 100 MHz constant clock? see below) block 0 wave 0 raw:", a[0, 0])
