#!/usr/bin/env python3
"""Wall time of `Focusr(target, source)` + `align_maps()` with EVERY argument at the reference's default (ICP,
affine + deformable CPD, smoothing, both outputs) on a synthetic blob pair, with a per-stage breakdown.
python tools/time_default_pipeline.py [n_vertices] [repeats]"""
import contextlib
import io
import os
import sys
import time
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyfocusr_amd  # noqa: E402
from pyfocusr_amd import Focusr, _hip, focusr as focusr_mod, vtk_functions  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = _hip.default_context()
stages = OrderedDict()


def timed(owner, name, label=None):
    fn = getattr(owner, name)

    def wrapper(*a, **kw):
        t0 = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            ctx.sync()
            stages[label or name] = stages.get(label or name, 0.0) + time.perf_counter() - t0

    setattr(owner, name, wrapper)


timed(focusr_mod, "icp_transform", "icp")
timed(focusr_mod, "compute_spectra", "spectra (assembly + eigensolve x2)")
timed(focusr_mod.eigsort, "sort_eigenmaps", "eigsort")
for m in ("register_target_to_source", "get_initial_correspondences", "get_smoothed_correspondences",
          "get_weighted_final_node_locations", "get_nearest_neighbour_final_node_locations",
          "get_source_mesh_transformed_weighted_avg", "get_source_mesh_transformed_nearest_neighbour"):
    timed(Focusr, m)

from pyfocusr_amd import cpd as cpd_mod  # noqa: E402

timed(cpd_mod, "low_rank_affinity", "  cpd: low-rank affinity")
timed(_hip.DeviceCpd, "estep", "  cpd: E-steps")
timed(_hip.DeviceCpd, "estep_resident", "  cpd: E-steps (enqueue only)")
timed(cpd_mod.affine_registration, "_maximisation", "  cpd: affine M-steps (incl. waiting for the E-step)")
timed(cpd_mod.deformable_registration, "_maximisation", "  cpd: deformable M-steps (incl. waiting for the E-step)")
timed(cpd_mod.deformable_registration, "transform_point_cloud", "  cpd: deformable transform of all points")
timed(_hip.DeviceCpd, "deform_sums", "    deform_sums")
timed(_hip.DeviceCpd, "apply_deform", "    apply_deform")
timed(_hip.DeviceCpd, "affine_sums", "    affine_sums")
timed(_hip.DeviceCpd, "apply_affine", "    apply_affine")
timed(cpd_mod.np.linalg, "solve", "    np.linalg.solve")
_orig_register = cpd_mod._ExpectationMaximisation.register


def _register(self, *a, **kw):
    out = _orig_register(self, *a, **kw)
    stages["  cpd: iterations %s" % type(self).__name__] = self.iteration * 1e-3  # printed as 'ms' = count
    return out


cpd_mod._ExpectationMaximisation.register = _register
meshes = [blob_mesh(n, seed=s) for s in (1, 0)]
for rep in range(reps):
    stages.clear()
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        reg = Focusr(meshes[0], meshes[1], ctx=ctx)
        t1 = time.perf_counter()
        reg.align_maps()
        ctx.sync()
        t2 = time.perf_counter()
    print("n=%d rep %d: ctor %.1f ms, align_maps %.1f ms, total %.1f ms" % (n, rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t2 - t0)))
    for k, v in stages.items():
        print("    %-48s %8.1f ms" % (k, 1e3 * v))
    for r in ("rigid", "non_rigid"):
        pass
    sys.stdout.flush()
