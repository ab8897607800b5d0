#!/usr/bin/env python3
"""How often does the Gram-Schmidt step of the Krylov driver need its second pass (pf_orth_end runs it on demand)?
Blob pairs of several sizes and k, plus the bundled 15k meshes (Arnoldi path, restarts)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import Graph, PolyMesh, _hip  # noqa: E402
from pyfocusr_amd.graph import compute_spectra  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

ctx = _hip.default_context()
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cases = [("blob %d" % n, [blob_mesh(n, s) for s in (0, 1)], k) for n, k in ((2000, 5), (20000, 5), (20000, 20), (100000, 10), (250000, 5), (500000, 5))]
z = [np.load(os.path.join(gold, f)) for f in ("target_mesh_15k.npz", "source_mesh_15k.npz")]
cases.append(("bundled 15k", [PolyMesh(a["points"], a["faces"]) for a in z], 5))
cases.append(("bundled 15k", [PolyMesh(a["points"], a["faces"]) for a in z], 12))
for name, meshes, k in cases:
    gs = [Graph(m, n_spectral_features=k, n_rand_samples=5000, ctx=ctx, verbose=False) for m in meshes]
    t0 = time.perf_counter()
    compute_spectra(gs)
    dt = time.perf_counter() - t0
    print("%-14s k=%-3d  %7.1f ms  outer steps %s  second passes %s  restarts %s  max residual %.1e" % (
        name, k, 1e3 * dt, [g.eigs_stats.outer_steps for g in gs], [g.eigs_stats.second_passes for g in gs],
        [g.eigs_stats.restarts for g in gs], max(g.eigs_stats.residuals.max() for g in gs)), flush=True)
