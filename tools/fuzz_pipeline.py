#!/usr/bin/env python3
"""Randomised end-to-end sweep: `Focusr(target, source)` + `align_maps()` on random blob pairs of different sizes with
random option combinations; checks that it runs and that the outputs are well-formed.
python tools/fuzz_pipeline.py SEED N_CASES"""
import contextlib
import io
import os
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import Focusr, _hip  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

ctx = _hip.default_context()
rng = np.random.default_rng(int(sys.argv[1]))
N = int(sys.argv[2])
fails, t0 = 0, time.time()
for it in range(N):
    nt, ns = (int(rng.choice([250, 600, 2000, 7000, 30000])) for _ in range(2))
    kw = dict(n_spectral_features=int(rng.integers(2, 6)), n_extra_spectral=int(rng.integers(0, 4)),  # (one eigenmap in total: the
              # eigenvalue gap is the mean of an empty difference -> NaN costs, in the reference as well)
              icp_register_first=bool(rng.integers(0, 2)), icp_registration_mode=str(rng.choice(["rigid", "similarity"])),
              icp_reg_target_to_source=bool(rng.integers(0, 2)), target_eigenmap_as_reference=bool(rng.integers(0, 2)),
              rigid_before_non_rigid_reg=bool(rng.integers(0, 2)), get_weighted_spectral_coords=bool(rng.integers(0, 2)),
              include_points_as_features=bool(rng.integers(0, 2)), smooth_correspondences=True,
              non_rigid_max_iterations=int(rng.choice([5, 60])), n_coords_spectral_registration=int(rng.choice([300, 5000])),
              n_coords_spectral_ordering=int(rng.choice([500, 5000])), graph_smoothing_iterations=int(rng.choice([3, 300])))
    a, b = blob_mesh(nt, seed=int(rng.integers(0, 10**6))), blob_mesh(ns, seed=int(rng.integers(0, 10**6)))
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            reg = Focusr(a, b, ctx=ctx, **kw)
            reg.align_maps()
        idx = reg.corresponding_target_idx_for_each_source_pt
        ok = (idx.shape == (ns,) and idx.min() >= 0 and idx.max() < nt and reg.weighted_avg_transformed_points.shape == (ns, 3)
              and np.all(np.isfinite(reg.weighted_avg_transformed_points)) and np.all(np.isfinite(reg.target_spectral_coords)))
        if not ok:
            fails += 1
            print("BAD OUTPUT nt=%d ns=%d %r" % (nt, ns, kw), flush=True)
    except Exception:
        fails += 1
        print("EXC nt=%d ns=%d %r\n%s" % (nt, ns, kw, traceback.format_exc()[-600:]), flush=True)
print("done: %d failures of %d, %.1fs" % (fails, N, time.time() - t0))
