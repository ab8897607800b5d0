"""CPU: the oracle (oracle/reference_port.py) pinned against fixtures produced by
running the reference itself (tools/make_golden.py) and against the eigenvalues
printed in the reference's notebook."""
import numpy as np
import pytest
from scipy import sparse

from oracle import reference_port as orc

MESHES = ["target_mesh", "source_mesh", "target_mesh_15k", "source_mesh_15k"]

NOTEBOOK = {  # examples/Example_registering_two_bone_meshes.ipynb cell 2 output
    "target_mesh": [8.39246263e-04, 1.63007145e-03, 2.12549101e-03, 3.13941439e-03, 3.77495258e-03, 4.01682329e-03],
    "source_mesh": [8.31236570e-04, 1.64152416e-03, 2.11362458e-03, 3.09029787e-03, 3.88535401e-03, 3.92405051e-03],
}


@pytest.mark.parametrize("name", MESHES)
def test_matrices_bit_exact(golden, name):
    g = golden(name)
    W, deg, d_inv, L = orc.graph_matrices(g["points"], g["faces"])
    assert np.array_equal(W.indptr, g["W_indptr"]) and np.array_equal(W.indices, g["W_indices"])
    assert np.array_equal(W.data, g["W_data"])
    assert np.array_equal(deg, g["deg"]) and np.array_equal(d_inv, g["d_inv"])
    assert np.array_equal(L.indptr, g["L_indptr"]) and np.array_equal(L.indices, g["L_indices"])
    assert np.array_equal(L.data, g["L_data"])


@pytest.mark.parametrize("name", MESHES)
def test_geometry(golden, name):
    g = golden(name)
    rng, mx, mean, normed = orc.geometry(g["points"])
    assert np.array_equal(rng, g["pts_scale_range"]) and mx == g["max_pts_scale_range"]
    assert mean == g["mean_pts_scale_range"] and np.array_equal(normed, g["normed_points"])


def test_mesh_quirks(golden):
    """5k meshes are clean; 15k meshes have one-way edges / isolated vertices."""
    for name, sym, iso in (("target_mesh", True, 0), ("source_mesh", True, 0),
                           ("target_mesh_15k", False, 0), ("source_mesh_15k", False, 2)):
        g = golden(name)
        n = len(g["points"])
        W = sparse.csr_matrix((g["W_data"], g["W_indices"], g["W_indptr"]), shape=(n, n))
        assert (abs(W - W.T).nnz == 0) == sym
        assert int(np.sum(g["deg"] == 0)) == iso
        assert len(g["L_data"]) == len(g["W_data"]) + n - iso


@pytest.mark.parametrize("name,k", [("target_mesh", 6), ("source_mesh", 6), ("target_mesh", 3),
                                    ("target_mesh_15k", 5), ("source_mesh_15k", 5)])
def test_spectrum(golden, name, k):
    g = golden(name)
    trace = []
    out = orc.graph_spectrum(g["points"], g["faces"], k, trace=trace)
    gv = g["k%d_eig_vals" % k]
    assert out["eig_vals"].shape == gv.shape  # incl. the widened 9 columns of source_mesh_15k
    np.testing.assert_allclose(out["eig_vals"], gv, rtol=1e-8)
    assert len(trace) - 1 == int(g["k%d_n_retries" % k])
    np.testing.assert_allclose(out["eig_vecs"], g["k%d_eig_vecs" % k], atol=5e-8)
    np.testing.assert_allclose(out["eig_vecs_raw"], g["k%d_eig_vecs_raw" % k], atol=5e-8)
    if name in NOTEBOOK and k == 6:
        np.testing.assert_allclose(gv, NOTEBOOK[name], rtol=0, atol=5.1e-12)  # 9 printed digits
        np.testing.assert_allclose(out["eig_vals"], NOTEBOOK[name], rtol=0, atol=5.1e-12)


@pytest.mark.parametrize("pair,t,s,k,ns", [("pair_5k", "target_mesh", "source_mesh", 6, 3),
                                           ("pair_15k", "target_mesh_15k", "source_mesh_15k", 5, 5)])
def test_eigsort_and_correspondence(golden, pair, t, s, k, ns):
    p, gt, gs = golden(pair), golden(t), golden(s)
    nt, nsrc = len(gt["points"]), len(gs["points"])
    vt, vs = gt["k%d_eig_vecs" % k], gs["k%d_eig_vecs" % k]
    lt, ls = gt["k%d_eig_vals" % k], gs["k%d_eig_vals" % k]
    out = orc.sort_eigenmaps(gt["points"], gs["points"], lt, ls, vt, vs,
                             np.arange(nt), np.arange(nsrc), k)
    for name in ("c_lambda", "c_hist", "c_hist_f", "c_spatial", "c_spatial_f", "Q"):
        np.testing.assert_allclose(out[name], p[name], rtol=1e-12, atol=0, err_msg=name)
    assert np.array_equal(out["idx_spatial"], p["idx_spatial"])
    assert np.array_equal(out["target_matches"], p["target_matches"])
    assert np.array_equal(out["source_matches"], p["source_matches"])
    flipped = np.zeros(k, dtype=bool)
    for a, b in out["flipped_pairs"]:
        flipped[a] = True
    assert np.array_equal(flipped, p["flipped"])
    assert np.array_equal(out["eig_vecs_s"], p["eig_vecs_s_sorted"])
    assert np.array_equal(out["eig_vecs_t"], p["eig_vecs_t_sorted"])

    w = orc.spectral_weights(out["Q"], ls, lt, ns)
    np.testing.assert_allclose(w, p["spectral_weights"], rtol=1e-12)
    for tag, weights in (("u", None), ("w", p["spectral_weights"])):
        cs, ct = orc.spectral_coords(out["eig_vecs_s"], out["eig_vecs_t"], ns, weights)
        assert np.array_equal(cs, p["coords_s_" + tag]) and np.array_equal(ct, p["coords_t_" + tag])
        idx = orc.knn1(ct, cs)
        assert np.array_equal(idx, p["knn_idx_" + tag])
        bidx, _ = orc.knn1_bruteforce(ct, cs)
        assert np.array_equal(bidx, p["knn_idx_" + tag])


def test_set_semantics_properties():
    """Property test (hypothesis) of the oracle's vectorised adjacency against the reference's literal
    per-edge loop (graph.py:156-178) on random polygon soups: directed set semantics, row sums of L."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    @settings(max_examples=60, deadline=None)
    @given(st.integers(3, 40), st.integers(1, 80), st.integers(3, 5), st.integers(0, 2**31 - 1))
    def run(n, n_faces, vpf, seed):
        rng = np.random.default_rng(seed)
        pts = rng.normal(size=(n, 3))
        if n < vpf:
            return
        faces = np.array([rng.choice(n, vpf, replace=False) for _ in range(n_faces)])
        lil = sparse.lil_matrix((n, n))
        for f in faces:  # the reference's loop: edge e of a cell joins vertices e and e+1 (mod v)
            for e in range(vpf):
                p1, p2 = int(f[e]), int(f[(e + 1) % vpf])
                lil[p1, p2] = 1.0 / np.sqrt(np.sum(np.square(pts[p1] - pts[p2])))
        W, deg, d_inv, L = orc.graph_matrices(pts, faces)
        ref = lil.tocsr()
        ref.sort_indices()
        assert np.array_equal(W.indptr, ref.indptr) and np.array_equal(W.indices, ref.indices)
        assert np.array_equal(W.data, ref.data)
        assert np.array_equal(deg, np.asarray(lil.sum(axis=1))[:, 0])
        assert np.max(np.abs(np.asarray(L.sum(axis=1)))) < 1e-14  # random-walk Laplacian: rows sum to ~0

    run()


def test_tail_fixture_mean_filter_and_final_locations(golden):
    """"Next" rows f1/f2 pinned: oracle `mean_filter_graph` (graph.py:320-354), smoothed correspondences
    (focusr.py:368-396) and weighted final locations incl. the coincident branch (focusr.py:401-426) against
    outputs of the REFERENCE's own methods (tools/make_golden.py: tail_fixture -> tests/golden/tail_5k.npz)."""
    t, gt, gs = golden("tail_5k"), golden("target_mesh"), golden("source_mesh")
    nt, ns = len(gt["points"]), len(gs["points"])
    Wt = sparse.csr_matrix((gt["W_data"], gt["W_indices"], gt["W_indptr"]), shape=(nt, nt))
    Ws = sparse.csr_matrix((gs["W_data"], gs["W_indices"], gs["W_indptr"]), shape=(ns, ns))
    for it in (25, 300):
        assert np.array_equal(orc.mean_filter_graph(Wt, gt["points"], it), t["mf_t_points_%d" % it])
        assert np.array_equal(orc.mean_filter_graph(Wt, t["scalar_in"][:, None], it), t["mf_t_scalar_%d" % it])
        assert np.array_equal(orc.mean_filter_graph(Ws, gs["points"], it), t["mf_s_points_%d" % it])
    assert np.array_equal(t["idx_initial"], golden("pair_5k")["knn_idx_w"])
    sm, proj, idx = orc.smoothed_correspondences(Wt, Ws, gt["points"], t["idx_initial"], 300, 40)
    assert np.array_equal(sm, t["smoothed_target_coords"])
    assert np.array_equal(proj, t["source_projected_on_target"])
    assert np.array_equal(idx, t["idx_final"])
    assert np.array_equal(gt["points"][idx], t["nearest_neighbor_transformed_points"])
    out = orc.weighted_final_node_locations(sm, proj, gt["points"])
    assert np.array_equal(out, t["weighted_avg_transformed_points"])
    out_c = orc.weighted_final_node_locations(sm, t["coincident_projected"], gt["points"])
    assert np.array_equal(out_c, t["coincident_weighted_avg"])
    rows = t["coincident_rows"]
    hit = (rows * 13 + 5) % nt
    hit[1] = hit[0]
    assert np.array_equal(out_c[rows], gt["points"][hit])  # focusr.py:415-419: the coincident target point itself
