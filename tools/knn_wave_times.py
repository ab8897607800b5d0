#!/usr/bin/env python3
"""Where the 1-NN grid search's time is: the waves' own run times in a counted search of the bench pair's spectral
coordinates (pf_knn_wave_stats): average, slowest wave, and how much of the kernel a perfectly balanced launch would take."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pyfocusr_amd import _hip  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
k = 5
ctx = _hip.default_context()
meshes = [blob_mesh(n, s) for s in (0, 1)]
timers = dict(assembly=0.0, eigensolve=0.0, eigsort=0.0, knn=0.0, matvecs=0)
np.random.seed(0)
bench.hot_path_step([ctx, ctx], meshes[0], meshes[1], k, 5000, timers)
lib = _hip.load_library()
lib.pf_knn_wave_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double)]
ctx.knn_count(True)  # the next step's search is the counting instantiation
for rep in range(3):
    ctx.knn_count(True)
    timers = dict(assembly=0.0, eigensolve=0.0, eigsort=0.0, knn=0.0, matvecs=0)
    bench.hot_path_step([ctx, ctx], meshes[0], meshes[1], k, 5000, timers)
    print("counted step: knn stage %.2f ms, %.0f pairs per query" % (1e3 * timers["knn"], ctx.knn_count(True) / n))
dt = timers["knn"]
pairs = ctx.knn_count(False)
s, m, nw, mc = C.c_double(), C.c_double(), C.c_int64(), C.c_int64()
det = (C.c_double * 5)()
lib.pf_knn_wave_stats(ctx._h, C.byref(s), C.byref(m), C.byref(nw), C.byref(mc), det)
slots = 256 * 4 * 6
print("counted search %.2f ms wall; %d waves, %.0f pairs per query" % (1e3 * dt, nw.value, pairs / n))
print("wave run time: average %.1f us, slowest %.1f us (scanned %d candidates x queries); sum %.1f ms = %.3f ms over %d wave slots" % (
    s.value / max(nw.value, 1), m.value, mc.value, 1e-3 * s.value, 1e-3 * s.value / slots, slots))
print("per wave: %.1f chunks in %.1f scans; scans %.1f us, bounds %.1f us, rest %.1f us" % (det[0] / nw.value, det[1] / nw.value, det[2] / nw.value, det[3] / nw.value,
      (s.value - det[2] - det[3]) / nw.value))
print("candidates that pass the coordinates outside the grid plane (against the bounds at hand): %.1f per chunk of 64 lanes" % (det[4] / max(det[0], 1)))
