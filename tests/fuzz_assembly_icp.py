#!/usr/bin/env python3
"""Randomised sweep: (i) Laplacian assembly on random polygon soups (3-6 vertices per face, duplicates, degenerate
edges, unreferenced points) bit-exact against the oracle; (ii) the ICP loop on small random pairs against the CPU
restatement.  Not collected by pytest:  python tests/fuzz_assembly_icp.py SEED N_CASES   on the GPU box."""
import os
import sys
import time
import traceback

import numpy as np
from scipy.sparse.csgraph import connected_components

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import icp_port  # noqa: E402
from oracle import reference_port as orc  # noqa: E402
from pyfocusr_amd import _hip, icp  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

ctx = _hip.default_context()
rng = np.random.default_rng(int(sys.argv[1]))
N = int(sys.argv[2])
fails, t0 = 0, time.time()
for it in range(N):
    try:
        n, vpf = int(rng.integers(3, 4000)), int(rng.integers(3, 7))
        F = int(rng.integers(1, 9000))
        pts = rng.normal(size=(n, 3)) * 10 ** rng.uniform(-3, 3)
        faces = rng.integers(0, n, size=(F, vpf)).astype(np.int32)
        if rng.integers(0, 2):
            faces[: F // 3] = faces[F // 3: F // 3 * 2][: F // 3]  # repeated faces
        keep = np.array([len(set(f)) == vpf for f in faces])      # the reference divides by zero on repeated vertices
        faces = faces[keep] if keep.any() else faces[:0]
        if len(faces):
            dev = _hip.DeviceLaplacian(pts, faces, ctx=ctx)
            d = dev.download(labels=True)
            W, deg, d_inv, L = orc.graph_matrices(pts, faces)
            Wc = W.tocsr()
            Wc.sort_indices()
            assert np.array_equal(d["rowptr"], Wc.indptr) and np.array_equal(d["colidx"], Wc.indices), "pattern"
            assert np.array_equal(d["w"], Wc.data) and np.array_equal(d["deg"], deg), "values"
            ncomp, lab = connected_components(Wc, directed=True, connection="weak")
            smallest = np.full(ncomp, n, dtype=np.int64)
            np.minimum.at(smallest, lab, np.arange(n))
            if dev.n_components <= 4096:
                assert np.array_equal(d["labels"], smallest[lab]), "component labels"
            dev.close()
        a, b = blob_mesh(int(rng.integers(200, 1500)), seed=int(rng.integers(0, 10**6))), blob_mesh(
            int(rng.integers(200, 1500)), seed=int(rng.integers(0, 10**6)))
        mode, iters, lm = str(rng.choice(["rigid", "similarity"])), int(rng.integers(1, 8)), int(rng.choice([50, 300, 5000]))
        got = icp.icp_transform(a.points, a.faces, b.points, numberOfIterations=iters, number_landmarks=lm, transform_mode=mode, ctx=ctx)
        want = icp_port.icp(a.points, a.faces, b.points, n_iterations=iters, n_landmarks=lm, mode=mode)
        assert np.allclose(got.matrix, want, rtol=0, atol=1e-11), ("icp", np.abs(got.matrix - want).max())
    except Exception:
        fails += 1
        print("FAIL case %d\n%s" % (it, traceback.format_exc()[-600:]), flush=True)
print("done: %d failures of %d, %.1fs" % (fails, N, time.time() - t0))
