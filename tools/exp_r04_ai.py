import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyfocusr_amd import Graph, _hip
from pyfocusr_amd.graph import build_devices, compute_spectra
from pyfocusr_amd.meshgen import messy_blob_mesh
ctx = _hip.default_context()
meshes = [messy_blob_mesh(250000, s) for s in (0, 1)]
for rep in range(3):
    gs = [Graph(m, n_spectral_features=5, n_rand_samples=5000, ctx=ctx, verbose=False) for m in meshes]
    build_devices(gs)
    ctx.sync()
    t0 = time.perf_counter()
    compute_spectra(gs)
    ctx.sync()
    dt = time.perf_counter() - t0
    print("%.2f ms" % (1e3 * dt), [(g.eigs_stats.outer_steps, g.eigs_stats.second_passes, g.eigs_stats.degree, g.eigs_stats.restarts, g.eigs_stats.mode) for g in gs], flush=True)
    for g in gs:
        g.device.close()
