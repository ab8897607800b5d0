#!/usr/bin/env python3
"""Generate `tests/golden/*.npz` by RUNNING THE REFERENCE in the build container.

Runs only where `/root/reference` exists (never on the GPU box).  The reference
package is imported unmodified; the three third-party modules it imports that
are not installed here (`vtk`, `itkwidgets`, `cycpd`) are replaced by empty stub
modules, and meshes are handed to it as `pyfocusr_amd.vtk_functions.PolyMesh`
objects, which duck-type the vtkPolyData calls the hot path makes
(`graph.py:58-62,155-164`).  Everything written to the fixtures is either an
input (points, faces) or an output of reference code.

Because ARPACK's start vector makes eigenvector signs/order run-dependent
(SURVEY.md §0.6), eigenpairs are stored sorted ascending with the
largest-|entry|-positive sign convention, and the reference's `eigsort` /
`Focusr` steps are then driven from those canonical eigenvectors.

    python tools/make_golden.py            # writes tests/golden/*.npz
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PYFOCUSR_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")


def import_reference():
    sys.dont_write_bytecode = True
    vtk = types.ModuleType("vtk")
    vtk_util = types.ModuleType("vtk.util")
    ns = types.ModuleType("vtk.util.numpy_support")
    ns.numpy_to_vtk = lambda a, *x, **k: np.asarray(a)
    ns.vtk_to_numpy = lambda a: np.asarray(getattr(a, "values", a))
    vtk.util = vtk_util
    vtk_util.numpy_support = ns
    itk = types.ModuleType("itkwidgets")
    itk.Viewer = None
    cycpd = types.ModuleType("cycpd")
    for name, mod in (("vtk", vtk), ("vtk.util", vtk_util), ("vtk.util.numpy_support", ns),
                      ("itkwidgets", itk), ("cycpd", cycpd)):
        sys.modules.setdefault(name, mod)
    sys.path.insert(0, REF)
    import pyfocusr  # noqa: E402  (the reference)

    return pyfocusr


def canonicalize(vals, vecs):
    order = np.argsort(vals, kind="stable")
    vals = np.asarray(vals)[order]
    vecs = np.array(vecs[:, order], dtype=np.float64, copy=True)
    piv = np.argmax(np.abs(vecs), axis=0)
    sgn = np.sign(vecs[piv, np.arange(vecs.shape[1])])
    sgn[sgn == 0] = 1.0
    return vals, vecs * sgn[None, :]


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


def mesh_fixture(ref, mesh, ks):
    """Per-mesh golden: reference matrices + canonical eigenpairs for each k."""
    out = dict(points=mesh.points, faces=mesh.faces)
    big = mesh.GetNumberOfPoints() + 1
    first = True
    for k in ks:
        g = ref.Graph(mesh, n_spectral_features=k, norm_eig_vecs=False, n_rand_samples=big,
                      feature_weights=np.eye(0))
        _, log = quiet(g.get_graph_spectrum)
        if first:
            W = g.adjacency_matrix.tocsr()
            W.sort_indices()
            L = g.laplacian_matrix.tocsr()
            L.sort_indices()
            out.update(W_data=W.data, W_indices=W.indices.astype(np.int32), W_indptr=W.indptr.astype(np.int32),
                       deg=g.degree_matrix.diagonal(), d_inv=g.degree_matrix_inv.diagonal(),
                       L_data=L.data, L_indices=L.indices.astype(np.int32), L_indptr=L.indptr.astype(np.int32),
                       pts_scale_range=g.pts_scale_range, mean_pts_scale_range=g.mean_pts_scale_range,
                       max_pts_scale_range=g.max_pts_scale_range, normed_points=g.normed_points)
            first = False
        vals, vecs = canonicalize(g.eig_vals, g.eig_vecs)
        normed = (vecs - np.min(vecs, axis=0)) / np.ptp(vecs, axis=0) - 0.5  # graph.py:254-257
        out["k%d_eig_vals" % k] = vals
        out["k%d_eig_vecs_raw" % k] = vecs
        out["k%d_eig_vecs" % k] = normed
        out["k%d_n_retries" % k] = np.int64(log.count("Not enough eigenvalues"))
    return out


def pair_fixture(ref, mesh_t, mesh_s, fix_t, fix_s, k, ns):
    """Reference eigsort + spectral coords + kd correspondence, driven from the
    canonical eigenvectors of the per-mesh fixtures."""
    from scipy.optimize import linear_sum_assignment
    from scipy.spatial import KDTree

    def graph(mesh, fix):
        g = ref.Graph(mesh, n_spectral_features=k, n_rand_samples=mesh.GetNumberOfPoints() + 1,
                      feature_weights=np.eye(0))
        g.eig_vals = fix["k%d_eig_vals" % k].copy()
        g.eig_vecs = fix["k%d_eig_vecs" % k].copy()
        return g

    gt, gs = graph(mesh_t, fix_t), graph(mesh_s, fix_s)
    out = {}
    sorter = sys.modules["pyfocusr.eigsort"].eigsort(graph_target=gt, graph_source=gs, n_features=k, target_as_reference=True)
    out["rand_target_points"] = sorter.rand_target_points
    out["rand_source_points"] = sorter.rand_source_points
    _, out["idx_spatial"] = KDTree(sorter.rand_source_points).query(sorter.rand_target_points)
    Q, log = quiet(sorter.sort_eigenmaps)
    for name in ("c_lambda", "c_hist", "c_hist_f", "c_spatial", "c_spatial_f"):
        out[name] = getattr(sorter, name)
    out["Q"] = np.asarray(Q)
    c = sorter.c_spatial * sorter.c_lambda * sorter.c_hist
    c_f = sorter.c_spatial_f * sorter.c_lambda * sorter.c_hist_f
    tm, sm = linear_sum_assignment(np.min((c, c_f), axis=0))
    out["target_matches"], out["source_matches"] = tm, sm
    out["flipped"] = (c > c_f)[tm, sm]
    out["eigsort_log"] = np.array(log)
    out["eig_vecs_t_sorted"] = gt.eig_vecs
    out["eig_vecs_s_sorted"] = gs.eig_vecs

    reg = object.__new__(ref.Focusr)  # drive focusr.py:351-366,459-508 without ICP/CPD
    reg.graph_target, reg.graph_source = gt, gs
    reg.Q = np.asarray(Q)
    reg.n_spectral_features = ns
    reg.initial_correspondence_type = "kd"
    for weighted in (False, True):
        reg.get_weighted_spectral_coords = weighted
        reg.calc_spectral_coords()
        reg.get_initial_correspondences()
        tag = "w" if weighted else "u"
        out["coords_s_" + tag] = np.array(reg.source_spectral_coords)
        out["coords_t_" + tag] = np.array(reg.target_spectral_coords)
        out["knn_idx_" + tag] = np.asarray(reg.corresponding_target_idx_for_each_source_pt, dtype=np.int64)
        d2, _ = KDTree(reg.target_spectral_coords).query(reg.source_spectral_coords, k=2)
        out["knn_top2_" + tag] = d2
    out["spectral_weights"] = reg.spectral_weights
    return out


def tail_fixture(ref, mesh_t, mesh_s, idx_initial):
    """Reference outputs of the post-KNN tail ("next" rows f1/f2 of SURVEY §8f):
    `Graph.mean_filter_graph` (graph.py:320-354) on n x 3 and n x 1 values for 25 and 300
    iterations, `Focusr.get_smoothed_correspondences` (focusr.py:368-396, both correspondence
    types "kd") and `Focusr.get_weighted_final_node_locations` (focusr.py:401-426) including
    its coincident-point branch (:415-419), driven from the reference's own initial
    correspondences `idx_initial` (pair fixture, `knn_idx_w`)."""
    def graph(mesh):
        g = ref.Graph(mesh, n_spectral_features=3, n_rand_samples=mesh.GetNumberOfPoints() + 1,
                      feature_weights=np.eye(0))
        quiet(g.get_weighted_adjacency_matrix)
        return g

    gt, gs = graph(mesh_t), graph(mesh_s)
    out = {}
    scalar = np.cos(3.0 * gt.normed_points[:, 0]) + gt.normed_points[:, 1] ** 2  # any n x 1 signal
    out["scalar_in"] = scalar
    for it in (25, 300):
        out["mf_t_points_%d" % it] = np.asarray(gt.mean_filter_graph(gt.points, iterations=it))
        out["mf_t_scalar_%d" % it] = np.asarray(gt.mean_filter_graph(scalar[:, None], iterations=it))
        out["mf_s_points_%d" % it] = np.asarray(gs.mean_filter_graph(gs.points, iterations=it))

    reg = object.__new__(ref.Focusr)  # drive focusr.py:368-426 without ICP / CPD
    reg.graph_target, reg.graph_source = gt, gs
    reg.initial_correspondence_type = "kd"
    reg.final_correspondence_type = "kd"
    reg.graph_smoothing_iterations = 300  # focusr.py:49 default
    reg.projection_smooth_iterations = 40  # focusr.py:55 default
    reg.corresponding_target_idx_for_each_source_pt = np.array(idx_initial)
    out["idx_initial"] = np.asarray(idx_initial, dtype=np.int64)
    reg.get_smoothed_correspondences()
    out["smoothed_target_coords"] = np.asarray(reg.smoothed_target_coords)
    out["source_projected_on_target"] = np.asarray(reg.source_projected_on_target)
    out["idx_final"] = np.asarray(reg.corresponding_target_idx_for_each_source_pt, dtype=np.int64)
    reg.get_weighted_final_node_locations()
    out["weighted_avg_transformed_points"] = np.array(reg.weighted_avg_transformed_points)
    reg.get_nearest_neighbour_final_node_locations()
    out["nearest_neighbor_transformed_points"] = np.array(reg.nearest_neighbor_transformed_points)
    d3, i3 = __import__("scipy.spatial", fromlist=["KDTree"]).KDTree(reg.smoothed_target_coords).query(
        reg.source_projected_on_target, k=4)
    out["final_top4_dist"], out["final_top4_idx"] = d3, i3  # tie margins of the 3-NN sets

    # coincident-point branch (focusr.py:415-419): every 7th projected point is put exactly on a
    # smoothed target point (a different one each time), two of them on the SAME target point.
    proj = np.array(reg.source_projected_on_target)
    hit = np.arange(0, len(proj), 7)
    tgt = (hit * 13 + 5) % len(reg.smoothed_target_coords)
    tgt[1] = tgt[0]
    proj[hit, :] = reg.smoothed_target_coords[tgt, :]
    reg.source_projected_on_target = proj
    reg.get_weighted_final_node_locations()
    out["coincident_projected"] = proj
    out["coincident_rows"] = hit
    out["coincident_weighted_avg"] = np.array(reg.weighted_avg_transformed_points)
    return out


def main():
    ref = import_reference()
    sys.path.insert(0, REPO)
    from pyfocusr_amd.vtk_functions import read_vtk_mesh

    os.makedirs(OUT, exist_ok=True)
    if "--only-tail" in sys.argv:  # adds tail_5k.npz next to the existing fixtures, which stay as they are
        mt = read_vtk_mesh(os.path.join(REF, "data", "target_mesh.vtk"))
        ms = read_vtk_mesh(os.path.join(REF, "data", "source_mesh.vtk"))
        idx = np.load(os.path.join(OUT, "pair_5k.npz"))["knn_idx_w"]
        tf = tail_fixture(ref, mt, ms, idx)
        np.savez_compressed(os.path.join(OUT, "tail_5k.npz"), **tf)
        print("tail_5k", {k: v.shape for k, v in tf.items()})
        return
    meshes, fixtures = {}, {}
    for name, ks in (("target_mesh", (3, 6)), ("source_mesh", (3, 6)),
                     ("target_mesh_15k", (5,)), ("source_mesh_15k", (5,))):
        meshes[name] = read_vtk_mesh(os.path.join(REF, "data", name + ".vtk"))
        fixtures[name] = mesh_fixture(ref, meshes[name], ks)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **fixtures[name])
        print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in fixtures[name].items()
                     if k.endswith("eig_vals")})
    for tag, t, s, k, ns in (("pair_5k", "target_mesh", "source_mesh", 6, 3),
                             ("pair_15k", "target_mesh_15k", "source_mesh_15k", 5, 5)):
        pf = pair_fixture(ref, meshes[t], meshes[s], fixtures[t], fixtures[s], k, ns)
        np.savez_compressed(os.path.join(OUT, tag + ".npz"), **pf)
        print(tag, "Q", pf["Q"], "flipped", pf["flipped"], "matches", pf["source_matches"])
        if tag == "pair_5k":
            tf = tail_fixture(ref, meshes[t], meshes[s], pf["knn_idx_w"])
            np.savez_compressed(os.path.join(OUT, "tail_5k.npz"), **tf)


if __name__ == "__main__":
    main()
