#!/usr/bin/env python3
"""Randomised sweep of the post-processing tail of `align_maps` (focusr.py:368-431) against the oracle: graph mean
filters, second NN query, 3-NN inverse-distance locations (including exactly coincident points), on random blob pairs
of different sizes with random initial correspondences.  Not collected by pytest:
python tests/fuzz_tail.py SEED N_CASES   on the GPU box."""
import os
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import reference_port as orc  # noqa: E402
from pyfocusr_amd import Focusr, Graph, _hip  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

ctx = _hip.default_context()
rng = np.random.default_rng(int(sys.argv[1]))
N = int(sys.argv[2])
fails, t0 = 0, time.time()
for it in range(N):
    try:
        nt, ns = int(rng.integers(100, 3000)), int(rng.integers(100, 3000))
        a, b = blob_mesh(nt, seed=int(rng.integers(0, 10**6))), blob_mesh(ns, seed=int(rng.integers(0, 10**6)))
        gi, pi = int(rng.choice([0, 1, 7, 60])), int(rng.choice([0, 1, 5, 40]))
        reg = object.__new__(Focusr)
        reg._ctx = ctx
        reg.graph_target = Graph(a, n_spectral_features=2, ctx=ctx, verbose=False)
        reg.graph_source = Graph(b, n_spectral_features=2, ctx=ctx, verbose=False)
        reg.graph_smoothing_iterations, reg.projection_smooth_iterations = gi, pi
        reg.initial_correspondence_type = reg.final_correspondence_type = "kd"
        idx0 = rng.integers(0, nt, ns)
        if pi == 0 and gi == 0:
            idx0[: ns // 3] = rng.integers(0, nt, ns // 3)  # projected points coincide with target vertices: zero distances
        reg.corresponding_target_idx_for_each_source_pt = idx0.copy()
        reg.get_smoothed_correspondences()
        reg.get_weighted_final_node_locations()
        reg.get_nearest_neighbour_final_node_locations()
        Wt = orc.weighted_adjacency(a.points, a.faces)
        Ws = orc.weighted_adjacency(b.points, b.faces)
        sm, proj, idx = orc.smoothed_correspondences(Wt, Ws, a.points, idx0, gi, pi)
        np.testing.assert_allclose(reg.smoothed_target_coords, sm, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(reg.source_projected_on_target, proj, rtol=1e-11, atol=1e-12)
        # the NN query runs on OUR coordinates (equal to the oracle's to ~1e-12): compare with a query on the same input
        assert np.array_equal(reg.corresponding_target_idx_for_each_source_pt,
                              orc.knn1(reg.smoothed_target_coords, reg.source_projected_on_target))
        want = orc.weighted_final_node_locations(reg.smoothed_target_coords, reg.source_projected_on_target, a.points)
        np.testing.assert_allclose(reg.weighted_avg_transformed_points, want, rtol=1e-9, atol=1e-10)
        assert np.array_equal(reg.nearest_neighbor_transformed_points, a.points[reg.corresponding_target_idx_for_each_source_pt])
        for g in (reg.graph_target, reg.graph_source):
            g.device.close()
    except Exception:
        fails += 1
        print("FAIL case %d (nt=%d ns=%d gi=%d pi=%d)\n%s" % (it, nt, ns, gi, pi, traceback.format_exc()[-700:]), flush=True)
print("done: %d failures of %d, %.1fs" % (fails, N, time.time() - t0))
