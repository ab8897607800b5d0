#!/usr/bin/env python3
"""Partial reorthogonalisation on the device: the fuzzer's cases (seed, count) with PF_EIGS_PRO on and off."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import Graph, PolyMesh, _hip  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

ctx = _hip.default_context()
rng = np.random.default_rng(int(sys.argv[1]))
for it in range(int(sys.argv[2])):
    n, k = int(rng.choice([400, 1500, 6000, 25000, 80000])), int(rng.integers(1, 9))
    m = blob_mesh(n, seed=int(rng.integers(0, 10**6)))
    pts, faces = m.points, m.faces
    two = bool(rng.integers(0, 3) == 0)
    if two:
        m2 = blob_mesh(max(200, n // 4), seed=int(rng.integers(0, 10**6)))
        pts, faces = np.concatenate([pts, m2.points + 300.0]), np.concatenate([faces, m2.faces + n])
    if not two:
        rng.integers(2, 5), rng.integers(1, 21), rng.integers(0, 2)  # (the fuzzer's draws for the row-partitioned leg)
    g = Graph(PolyMesh(pts, faces), n_spectral_features=k, norm_eig_vecs=False, n_rand_samples=10**9, ctx=ctx, verbose=False)
    g.get_graph_spectrum()
    out = []
    for pro in ("0", "1"):
        os.environ["PF_EIGS_PRO"] = pro
        try:
            vals, vecs, st = g.device.eigs_smallest(k)
            out.append("pro=%s steps %d local %d second %d restarts %d resid %.1e" % (pro, st["outer_steps"], st["local_steps"], st["second_passes"], st["restarts"], st["max_residual"]))
            if pro == "0":
                ref = vals
            else:
                out.append("rel diff %.1e" % np.max(np.abs(vals / ref - 1)))
        except Exception as e:  # noqa: BLE001
            out.append("pro=%s FAIL %s" % (pro, str(e)[:90]))
    print("n=%d k=%d two=%s | " % (len(pts), k, two) + " | ".join(out), flush=True)
