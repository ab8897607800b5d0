"""CPU tests (`-m "not gpu"`): C-ABI surface, host logic (Krylov driver, Graph/eigsort
shells, VTK reader, mesh generator) against the oracle and the golden fixtures.  No device
compute happens here; the device is replaced by the numpy test double in tests/_numpy_ops.py."""
import os
import re
import sys

import numpy as np
import pytest
from scipy import sparse

from oracle import reference_port as orc

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _numpy_ops import NumpyOps  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------- C-ABI surface
def header_functions():
    text = open(os.path.join(REPO, "include", "pyfocusr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as entry
    from pyfocusr_amd import _hip

    entry.build()  # hipcc cross-compiles gfx950 without a GPU
    assert os.path.exists(_hip.LIB_PATH)
    lib = _hip.load_library()
    declared = header_functions()
    assert len(declared) >= 30
    assert sorted(_hip.SIGNATURES) == declared  # the binding covers exactly the header
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.pf_version() == 1


def test_product_fails_loudly_without_gpu_or_library(monkeypatch):
    from pyfocusr_amd import _hip

    if _hip.load_library().pf_device_count() == 0:
        with pytest.raises(_hip.HipUnavailable):
            _hip.Context()
        from pyfocusr_amd import Graph, PolyMesh
        from pyfocusr_amd.meshgen import blob_mesh

        g = Graph(blob_mesh(200, 0), n_spectral_features=2, verbose=False)
        with pytest.raises(_hip.HipUnavailable):
            g.get_graph_spectrum()  # no silent CPU fallback
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/libpyfocusr_hip.so")
    with pytest.raises(_hip.HipUnavailable):
        _hip.load_library()


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "pyfocusr_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("oracle-", ""), f
                assert "_numpy_ops" not in src, f
    # tools/ and examples/ are product-side too: no `import oracle`, no `from oracle ...`
    for folder in ("tools", "examples"):
        for f in os.listdir(os.path.join(REPO, folder)):
            if f.endswith(".py"):
                src = open(os.path.join(REPO, folder, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


# ------------------------------------------------------------------------------- mesh I/O
def test_vtk_reader_and_mesh_protocol(tmp_path, golden):
    from pyfocusr_amd import read_vtk_mesh
    from pyfocusr_amd.vtk_functions import PolyMesh, mesh_arrays

    g = golden("target_mesh")
    pts, faces = g["points"][:50], np.array([[0, 1, 2], [2, 1, 3], [4, 3, 1]], dtype=np.int32)
    path = tmp_path / "m.vtk"
    with open(path, "w") as fh:
        fh.write("# vtk DataFile Version 4.2\nvtk output\nASCII\nDATASET POLYDATA\nPOINTS %d double\n" % len(pts))
        flat = pts.reshape(-1)
        for i in range(0, len(flat), 9):
            fh.write(" ".join(repr(float(v)) for v in flat[i:i + 9]) + "\n")
        fh.write("POLYGONS %d %d\n" % (len(faces), 4 * len(faces)))
        for f in faces:
            fh.write("3 %d %d %d\n" % tuple(f))
        fh.write("POINT_DATA %d\nSCALARS thickness_change_(mm) double\nLOOKUP_TABLE default\n" % len(pts))
        fh.write(" ".join("0.5" for _ in range(len(pts))) + "\n")
    m = read_vtk_mesh(str(path))
    assert np.array_equal(m.points, pts) and np.array_equal(m.faces, faces)
    assert m.GetPointData().GetNumberOfArrays() == 1
    assert m.GetPointData().GetArray(0).GetName() == "thickness_change_(mm)"
    # vtkPolyData protocol the reference walks (graph.py:58-62,155-164): edges (0,1),(1,2),(2,0)
    assert m.GetNumberOfPoints() == 50 and m.GetNumberOfCells() == 3
    cell = m.GetCell(1)
    edges = [(cell.GetEdge(e).GetPointId(0), cell.GetEdge(e).GetPointId(1)) for e in range(cell.GetNumberOfEdges())]
    assert edges == [(2, 1), (1, 3), (3, 2)]
    assert m.GetPoint(3) == tuple(pts[3])

    class Duck(object):  # no .points/.faces: generic walk
        def __getattr__(self, name):
            if name in ("points", "faces"):
                raise AttributeError(name)
            return getattr(m, name)

    p2, f2 = mesh_arrays(Duck())
    assert np.array_equal(p2, pts) and np.array_equal(f2, faces)
    with pytest.raises(ValueError):
        PolyMesh(pts, np.array([[0, 1, 50]]))


def test_blob_mesh_is_closed_manifold():
    from pyfocusr_amd.meshgen import blob_mesh

    m = blob_mesh(2000, seed=5)
    assert m.points.shape == (2000, 3) and m.faces.shape == (2 * 2000 - 4, 3)
    W = orc.weighted_adjacency(m.points, m.faces)
    assert abs(W - W.T).nnz == 0 and W.nnz == 3 * len(m.faces)  # every directed edge once, both directions
    deg_count = np.diff(W.indptr)
    assert deg_count.min() >= 3 and deg_count.max() <= 10
    ncomp, _ = sparse.csgraph.connected_components(W, directed=False)
    assert ncomp == 1
    m2 = blob_mesh(2000, seed=5)
    assert np.array_equal(m.points, m2.points) and np.array_equal(m.faces, m2.faces)


# ------------------------------------------------------------------------------- Krylov driver
def solve(points, faces, k, **kw):
    from pyfocusr_amd._krylov import filtered_eigs

    W = orc.weighted_adjacency(points, faces)
    ops = NumpyOps(W)
    c0 = ops.lock_null_vectors()
    lam, first, st = filtered_eigs(ops, k, ops.symmetric, null_slots=c0, **kw)
    X = ops.download_slots(first, len(lam))
    if ops.symmetric:
        X = X * ops.s[:, None]
    X = X / np.linalg.norm(X, axis=0)
    lam, X = orc.canonicalize(lam, X)
    return lam, X, st, ops


@pytest.mark.parametrize("name,k", [("target_mesh", 6), ("source_mesh", 3), ("target_mesh_15k", 5), ("source_mesh_15k", 9)])
def test_filtered_krylov_schur_matches_reference(golden, name, k):
    g = golden(name)
    gk = {"target_mesh": 6, "source_mesh": 3, "target_mesh_15k": 5, "source_mesh_15k": 5}[name]
    lam, X, st, ops = solve(g["points"], g["faces"], k)
    gv = g["k%d_eig_vals" % gk]
    m = min(len(gv), len(lam))
    assert len(lam) == k
    np.testing.assert_allclose(lam[:m], gv[:m], rtol=1e-8)
    tol = 5e-7 if "15k" in name else 2e-9
    assert np.max(np.abs(orc.minmax_normalize(X)[:, :m] - g["k%d_eig_vecs" % gk][:, :m])) < tol
    assert st.residuals.max() < 1e-8 and st.filter_resets == 0
    if not ops.symmetric:
        assert st.degree <= 128  # complex outliers: degree capped


def test_krylov_components_isolated_and_bad_cut():
    from pyfocusr_amd.meshgen import blob_mesh

    a, b = blob_mesh(400, seed=3), blob_mesh(500, seed=4)
    pts = np.concatenate([a.points, b.points + 200.0, np.zeros((2, 3))])
    faces = np.concatenate([a.faces, b.faces + 400])
    ref = orc.graph_spectrum(pts, faces, 4)  # 2 + 2 nulls -> widened, 6 columns
    lam, X, st, ops = solve(pts, faces, len(ref["eig_vals"]))
    assert ops.n_isolated == 2 and st.n_null == 2
    np.testing.assert_allclose(lam, ref["eig_vals"], rtol=1e-8)
    assert np.all(X[-2:, :] == 0.0)  # isolated vertices stay out of every eigenvector
    # a cut far too small must be detected and enlarged, not silently converge to wrong pairs
    lam2, _, st2, _ = solve(pts, faces, len(ref["eig_vals"]), cut=1e-7)
    assert st2.filter_resets >= 1
    np.testing.assert_allclose(lam2, ref["eig_vals"], rtol=1e-8)
    # thick restart path: tiny basis
    lam3, _, st3, _ = solve(a.points, a.faces, 4, m_max=14)
    assert st3.restarts >= 1
    np.testing.assert_allclose(lam3, orc.graph_spectrum(a.points, a.faces, 4)["eig_vals"], rtol=1e-8)


def test_filter_overflow_is_reported_not_propagated(monkeypatch):
    """At very high degree (k = 1 on a large open mesh: a tiny cut inside a flat ellipse) an eigenvalue outside the
    damped set overflows the filtered vector before a Ritz value can expose it (fuzz case: 150k vertices, hole with
    2235 stranded vertices -> LinAlgError from numpy's eig on a NaN Hessenberg matrix).  Non-finite orthogonalisation
    coefficients must send a non-symmetric solve to a taller ellipse and fail a symmetric one with the reason."""
    from pyfocusr_amd import _krylov
    from pyfocusr_amd.meshgen import blob_mesh

    m = blob_mesh(1500, seed=2)
    faces = m.faces[np.linalg.norm(m.points[m.faces].mean(1) - m.points[0], axis=1) > 8.0]  # open: asymmetric W
    heights, poisoned = [], [0]
    orig_solve, orig_end = _krylov._solve_gen, NumpyOps.orth_end

    def spy(*a, **kw):
        heights.append(kw.get("half_height"))
        return (yield from orig_solve(*a, **kw))

    def poisoned_end(self):
        h, beta = orig_end(self)
        if poisoned[0] == 0:  # the very first step of the first attempt overflows
            poisoned[0] = 1
            return h, np.inf
        return h, beta

    monkeypatch.setattr(_krylov, "_solve_gen", spy)
    monkeypatch.setattr(NumpyOps, "orth_end", poisoned_end)
    lam, X, st, ops = solve(m.points, faces, 2, ellipse=True)
    assert not ops.symmetric and heights[:2] == [0.25, 0.4]
    np.testing.assert_allclose(lam, np.sort(orc.graph_spectrum(m.points, faces, 2)["eig_vals"])[:2], rtol=1e-7)
    poisoned[0] = 0
    with pytest.raises(RuntimeError, match="overflowed"):
        solve(m.points, m.faces, 2)


TET = (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.3]], float), np.array([[0, 2, 1], [0, 1, 3], [1, 2, 3], [0, 3, 2]]))
OCT = (np.array([[1, 0, 0], [-1, 0, 0], [0, 1.1, 0], [0, -1.2, 0], [0, 0, 1.3], [0, 0, -0.9]], float),
       np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]]))


def dense_nonnull(points, faces):
    W, deg, d_inv, L = orc.graph_matrices(points, faces)
    ev = np.sort(np.linalg.eigvals(L.toarray()).real)
    return ev[ev > 1e-10]


@pytest.mark.parametrize("mesh,k", [(TET, 1), (TET, 3), (OCT, 3), (OCT, 5)])
def test_krylov_tiny_meshes_plain_mode(mesh, k):
    """Meshes so small that the wanted eigenvalues are not a corner of the spectrum: the solver drops the
    filter (degree 1) and exhausts the Krylov space.  (scipy's eigs refuses k >= n-1 here.)"""
    lam, X, st, ops = solve(mesh[0], mesh[1], k)
    np.testing.assert_allclose(lam, dense_nonnull(*mesh)[:k], rtol=1e-10)
    assert st.degree == 1


def test_krylov_many_pairs_of_a_small_mesh():
    from pyfocusr_amd.meshgen import blob_mesh

    m = blob_mesh(40, seed=1)
    lam, X, st, ops = solve(m.points, m.faces, 30)
    np.testing.assert_allclose(lam, dense_nonnull(m.points, m.faces)[:30], rtol=1e-9)


def grid_mesh(nx, ny, seed=0):
    """Open surface: every boundary edge belongs to one face only, hence is one-way in W (graph.py:178)."""
    r = np.random.default_rng(seed)
    x, y = np.meshgrid(np.arange(nx, dtype=float), np.arange(ny, dtype=float), indexing="ij")
    z = 0.3 * np.sin(x / 5) + 0.2 * np.cos(y / 7)
    pts = np.stack([x, y, z], -1).reshape(-1, 3) + 0.05 * r.normal(size=(nx * ny, 3))
    idx = np.arange(nx * ny).reshape(nx, ny)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    return pts, np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)])


@pytest.mark.parametrize("nx,ny", [(40, 40), (100, 80)])
def test_krylov_open_mesh_complex_spectrum(nx, ny):
    """Open meshes make L non-normal with complex LOW eigenvalues; the reference keeps the real parts
    (graph.py:386), so conjugate pairs show up as repeated eigenvalues.  The interval filter cannot work
    here; the solver must switch to the ellipse filter and return the same real parts."""
    pts, faces = grid_mesh(nx, ny)
    ref = orc.graph_spectrum(pts, faces, 5)
    lam, X, st, ops = solve(pts, faces, 5)
    assert not ops.symmetric and st.filter_resets == 0
    m = min(len(lam), len(ref["eig_vals"]))
    assert m >= 5
    np.testing.assert_allclose(lam[:m], ref["eig_vals"][:m], rtol=1e-7)
    assert np.isclose(lam[0], lam[1], rtol=1e-9)  # a conjugate pair
    lam2, _, st2, _ = solve(pts, faces, 5, ellipse=True)  # what Graph asks for when it sees many one-way edges
    np.testing.assert_allclose(lam2[:m], ref["eig_vals"][:m], rtol=1e-7)


def test_lockstep_pair_driver_equals_single_solves(golden):
    """`drive_pair` (two solvers sharing kernel launches) must return what two separate solves
    return, also when the two need different degrees / step counts / a filter reset."""
    from pyfocusr_amd._krylov import drive, drive_pair, filtered_eigs_gen
    from pyfocusr_amd.meshgen import blob_mesh

    a, b = golden("target_mesh"), blob_mesh(1500, seed=9)
    single = []
    for pts, faces, kw in ((a["points"], a["faces"], {}), (b.points, b.faces, dict(cut=1e-7))):
        ops = NumpyOps(orc.weighted_adjacency(pts, faces))
        c0 = ops.lock_null_vectors()
        single.append(drive(filtered_eigs_gen(ops, 4, ops.symmetric, null_slots=c0, **kw), ops))
    oa, ob = NumpyOps(orc.weighted_adjacency(a["points"], a["faces"])), NumpyOps(orc.weighted_adjacency(b.points, b.faces))
    ga = filtered_eigs_gen(oa, 4, True, null_slots=oa.lock_null_vectors())
    gb = filtered_eigs_gen(ob, 4, True, null_slots=ob.lock_null_vectors(), cut=1e-7)
    ra, rb = drive_pair(ga, oa, gb, ob)
    for (lam, first, st), (lam1, first1, st1) in ((ra, single[0]), (rb, single[1])):
        assert np.array_equal(lam, lam1) and first == first1 and st.matvecs == st1.matvecs
    assert rb[2].filter_resets >= 1 and ra[2].degree != rb[2].degree
    # the same with the device class's pair entries (one call per outer step of both solvers: both Gram-Schmidt steps and
    # the next filter application; Gram-Schmidt steps alone where one solver does not speculate; single requests once
    # the solvers have drifted apart or one has finished)
    from _numpy_ops import PairedNumpyOps

    pa, pb = PairedNumpyOps(orc.weighted_adjacency(a["points"], a["faces"])), PairedNumpyOps(orc.weighted_adjacency(b.points, b.faces))
    ga = filtered_eigs_gen(pa, 4, True, null_slots=pa.lock_null_vectors())
    gb = filtered_eigs_gen(pb, 4, True, null_slots=pb.lock_null_vectors(), cut=1e-7)
    qa, qb = drive_pair(ga, pa, gb, pb)
    for (lam, first, st), (lam1, first1, st1) in ((qa, single[0]), (qb, single[1])):
        assert np.array_equal(lam, lam1) and first == first1 and st.matvecs == st1.matvecs
    assert pa.pair_calls["orth_cheb2"] >= 10 and np.array_equal(pa.download_slots(qa[1], 4), oa.download_slots(ra[1], 4))


def test_widen_rule_matches_reference_trace(golden):
    from pyfocusr_amd.graph import _widened_k

    assert _widened_k(6, 5, 1, 1, 15000) == (6, 0)
    assert _widened_k(6, 5, 1, 3, 15000) == (12, 1)  # source_mesh_15k: 9 columns returned
    assert _widened_k(4, 3, 1, 9, 15000) == (12, 2)
    g = golden("source_mesh_15k")
    assert int(g["k5_n_retries"]) == 1 and g["k5_eig_vals"].shape == (9,)


# ------------------------------------------------------------------------------- Graph / eigsort shells
class FakeDevice(NumpyOps):
    """NumpyOps + the download/finalize surface of DeviceLaplacian, for host-logic tests."""

    def __init__(self, points, faces):
        NumpyOps.__init__(self, orc.weighted_adjacency(points, faces))
        self.n_components = len([c for c in np.unique(self.labels) if np.sum(self.labels == c) > 1])
        self.nnz_w = self.W.nnz

    def download(self, labels=False):
        W = self.W
        g = 1.0 / (self.deg + 1e-8)
        rows = np.repeat(np.arange(self.n), np.diff(W.indptr))
        return dict(rowptr=W.indptr.astype(np.int32), colidx=W.indices.astype(np.int32), w=W.data,
                    l_offdiag=-(g[rows] * W.data), deg=self.deg, l_diag=g * self.deg)

    def finalize_wait(self):
        pass

    def finalize_vectors(self, first, count, minmax, wait=True):
        X = self.download_slots(first, count)
        if self.symmetric:
            X = X * self.s[:, None]
        X = X / np.linalg.norm(X, axis=0)
        _, X = orc.canonicalize(np.arange(float(count)), X)
        return orc.minmax_normalize(X) if minmax else X

    def close(self):
        pass


class FakeCtx(object):
    def knn1(self, ref, qry, return_d2=False):
        return orc.knn1(ref, qry)


def fake_graph(gold, k, **kw):
    from pyfocusr_amd import Graph, PolyMesh

    gr = Graph(PolyMesh(gold["points"], gold["faces"]), n_spectral_features=k, n_rand_samples=10**9,
               ctx=FakeCtx(), verbose=False, **kw)
    gr._device = FakeDevice(gold["points"], gold["faces"])
    return gr


@pytest.mark.parametrize("name", ["target_mesh", "source_mesh_15k"])
def test_graph_shell_matrices_and_spectrum(golden, name):
    g = golden(name)
    k = 6 if name == "target_mesh" else 5
    gr = fake_graph(g, k)
    assert gr.adjacency_matrix.nnz == 0  # empty lil until computed (graph.py:70-72)
    gr.get_weighted_adjacency_matrix()
    gr.get_degree_matrix()
    gr.get_G_matrix()
    gr.get_laplacian_matrix()
    L = gr.laplacian_matrix
    assert np.array_equal(L.indptr, g["L_indptr"]) and np.array_equal(L.indices, g["L_indices"])
    assert np.array_equal(L.data, g["L_data"])  # isolated rows carry no explicit zero diagonal
    assert np.array_equal(gr.degree_matrix_inv.diagonal(), g["d_inv"])
    assert np.array_equal(gr.normed_points, g["normed_points"])
    assert np.array_equal(gr.pts_scale_range, g["pts_scale_range"])
    gr.get_graph_spectrum()
    assert gr.eig_vals.shape == g["k%d_eig_vals" % k].shape
    np.testing.assert_allclose(gr.eig_vals, g["k%d_eig_vals" % k], rtol=1e-8)
    assert gr.get_rand_eig_vecs().shape == gr.eig_vecs.shape
    gr.get_eig_val_gap()
    assert gr.eig_val_gap == np.mean(np.diff(gr.eig_vals))
    with pytest.raises(NotImplementedError):
        fake_graph(g, 3, list_features_to_calc=["curvature"])
    np.random.seed(3)
    a = gr.get_list_rand_idxs(100)
    np.random.seed(3)
    assert np.array_equal(a, gr.get_list_rand_idxs(100)) and len(np.unique(a)) == 100


@pytest.mark.parametrize("pair,t,s,k,ns", [("pair_5k", "target_mesh", "source_mesh", 6, 3),
                                           ("pair_15k", "target_mesh_15k", "source_mesh_15k", 5, 5)])
def test_eigsort_and_focusr_host_logic(golden, pair, t, s, k, ns):
    from pyfocusr_amd import Focusr, eigsort

    p = golden(pair)
    graphs = []
    for name in (t, s):
        gold = golden(name)
        gr = fake_graph(gold, k)
        gr.eig_vals = gold["k%d_eig_vals" % k].copy()
        gr.eig_vecs = gold["k%d_eig_vecs" % k].copy()
        graphs.append(gr)
    gt, gs = graphs
    sorter = eigsort(graph_target=gt, graph_source=gs, n_features=k, target_as_reference=True)
    Q = sorter.sort_eigenmaps()
    for name in ("c_lambda", "c_hist", "c_hist_f", "c_spatial", "c_spatial_f"):
        np.testing.assert_allclose(getattr(sorter, name), p[name], rtol=1e-12, err_msg=name)
    np.testing.assert_allclose(Q, p["Q"], rtol=1e-12)
    assert np.array_equal(sorter.source_matches, p["source_matches"])
    assert np.array_equal(gs.eig_vecs, p["eig_vecs_s_sorted"])
    reg = object.__new__(Focusr)
    reg._ctx = FakeCtx()
    reg.graph_target, reg.graph_source, reg.Q, reg.n_spectral_features = gt, gs, Q, ns
    reg.get_weighted_spectral_coords = True
    reg.initial_correspondence_type = "kd"
    reg.calc_spectral_coords()
    reg.get_initial_correspondences()
    np.testing.assert_allclose(reg.spectral_weights, p["spectral_weights"], rtol=1e-12)
    assert np.array_equal(reg.corresponding_target_idx_for_each_source_pt, p["knn_idx_w"])
    # source-as-reference branch (eigsort.py:77-78,111-122) against the oracle
    gt2, gs2 = fake_graph(golden(t), k), fake_graph(golden(s), k)
    for gr, name in ((gt2, t), (gs2, s)):
        gr.eig_vals = golden(name)["k%d_eig_vals" % k].copy()
        gr.eig_vecs = golden(name)["k%d_eig_vecs" % k].copy()
    ref = orc.sort_eigenmaps(gt2.points, gs2.points, gt2.eig_vals, gs2.eig_vals, gt2.eig_vecs.copy(),
                             gs2.eig_vecs.copy(), np.arange(gt2.n_points), np.arange(gs2.n_points), k,
                             target_as_reference=False)
    Q2 = eigsort(gt2, gs2, k, target_as_reference=False).sort_eigenmaps()
    np.testing.assert_allclose(Q2, ref["Q"], rtol=1e-12)
    assert np.array_equal(gt2.eig_vecs, ref["eig_vecs_t"]) and np.array_equal(gs2.eig_vecs, ref["eig_vecs_s"])


def test_focusr_constructor_surface():
    """Same positional order / defaults as focusr.py:23-69."""
    import inspect

    from pyfocusr_amd import Focusr

    sig = inspect.signature(Focusr.__init__)
    names = list(sig.parameters)
    assert names[1:3] == ["vtk_mesh_target", "vtk_mesh_source"]
    d = {k: v.default for k, v in sig.parameters.items()}
    assert d["n_spectral_features"] == 3 and d["n_extra_spectral"] == 3 and d["icp_register_first"] is True
    assert d["n_coords_spectral_ordering"] == 5000 and d["get_weighted_spectral_coords"] is True
    assert d["graph_smoothing_iterations"] == 300 and d["projection_smooth_iterations"] == 40
    assert d["list_features_to_calc"] == ["curvature"] and d["initial_correspondence_type"] == "kd"


@pytest.mark.parametrize("seed", range(8))
def test_eigsort_random_inputs_vs_oracle(seed):
    """The eigsort mirror on random eigenmaps (different vertex counts, sampled subsets, both reference choices,
    columns that need flips and permutations) against the oracle's restatement of eigsort.py:54-249."""
    from pyfocusr_amd import Graph, PolyMesh, eigsort
    from pyfocusr_amd.meshgen import blob_mesh

    rng = np.random.default_rng(100 + seed)
    k = int(rng.integers(2, 7))
    nt, ns = int(rng.integers(200, 900)), int(rng.integers(200, 900))
    n_samples = int(rng.choice([150, 10**9]))  # subset sampling / all points
    target_ref = bool(rng.integers(0, 2))
    graphs = []
    for n in (nt, ns):
        m = blob_mesh(n, seed=int(rng.integers(0, 10**6)))
        gr = Graph(PolyMesh(m.points, m.faces), n_spectral_features=k, n_rand_samples=n_samples, ctx=FakeCtx(), verbose=False)
        gr.eig_vals = np.sort(rng.uniform(1e-4, 5e-3, k + int(rng.integers(0, 3))))  # extra columns as after widening
        base = rng.uniform(-0.5, 0.5, (n, 1)) * rng.uniform(0.2, 1.0, (1, len(gr.eig_vals))) + 0.2 * rng.uniform(-0.5, 0.5, (n, len(gr.eig_vals)))
        base = (base - base.min(0)) / np.ptp(base, axis=0) - 0.5  # min-max normalised like graph.py:254-257
        gr.eig_vecs = base * rng.choice([-1.0, 1.0], size=(1, base.shape[1]))
        gr.eig_vecs = (gr.eig_vecs - gr.eig_vecs.min(0)) / np.ptp(gr.eig_vecs, axis=0) - 0.5
        graphs.append(gr)
    gt, gs = graphs
    vt0, vs0 = gt.eig_vecs.copy(), gs.eig_vecs.copy()
    ref = orc.sort_eigenmaps(gt.points, gs.points, gt.eig_vals, gs.eig_vals, vt0.copy(), vs0.copy(), gt.rand_idxs, gs.rand_idxs, k,
                             target_as_reference=target_ref)
    sorter = eigsort(gt, gs, k, target_as_reference=target_ref)
    Q = sorter.sort_eigenmaps()
    for name in ("c_lambda", "c_hist", "c_hist_f", "c_spatial", "c_spatial_f"):
        np.testing.assert_allclose(getattr(sorter, name), ref[name], rtol=1e-12, atol=1e-300, err_msg=name)
    np.testing.assert_allclose(Q, ref["Q"], rtol=1e-12)
    assert np.array_equal(gt.eig_vecs, ref["eig_vecs_t"]) and np.array_equal(gs.eig_vecs, ref["eig_vecs_s"])
    assert np.array_equal(gt.eig_vals, np.sort(gt.eig_vals))  # eigenvalues are never permuted


def test_vtk_reader_ascii_binary_and_version_5(tmp_path):
    """Legacy VTK POLYDATA in the four flavours a user may have on disk: ASCII / BINARY (big-endian blocks), file
    versions <= 4.2 (`POLYGONS n size` records) and 5.1 (`OFFSETS` / `CONNECTIVITY`), with point scalars."""
    from pyfocusr_amd.vtk_functions import read_vtk_mesh

    rng = np.random.default_rng(0)
    pts = rng.normal(size=(7, 3))
    faces = np.array([[0, 1, 2], [2, 3, 4], [4, 5, 6]], dtype=np.int32)
    sc = rng.normal(size=7)
    head = "# vtk DataFile Version %s\nvtk output\n%s\nDATASET POLYDATA\n"
    files = {
        "a42": (head % ("4.2", "ASCII") + "POINTS 7 float\n" + "\n".join(" ".join(repr(float(v)) for v in p) for p in pts)
                + "\nPOLYGONS 3 12\n" + "\n".join("3 %d %d %d" % tuple(f) for f in faces)
                + "\n\nPOINT_DATA 7\nSCALARS thickness double\nLOOKUP_TABLE default\n" + " ".join(repr(float(v)) for v in sc) + "\n").encode(),
        "a51": (head % ("5.1", "ASCII") + "POINTS 7 double\n" + " ".join(repr(float(v)) for v in pts.ravel())
                + "\n\nMETADATA\nINFORMATION 0\n\nPOLYGONS 4 9\nOFFSETS vtktypeint64\n0 3 6 9\nCONNECTIVITY vtktypeint64\n"
                + " ".join(str(int(v)) for v in faces.ravel()) + "\n").encode(),
        "b42": (head % ("4.2", "BINARY")).encode() + b"POINTS 7 float\n" + pts.astype(">f4").tobytes() + b"\nPOLYGONS 3 12\n"
        + np.hstack([np.full((3, 1), 3), faces]).astype(">i4").tobytes()
        + b"\nPOINT_DATA 7\nSCALARS thickness double\nLOOKUP_TABLE default\n" + sc.astype(">f8").tobytes() + b"\n",
        "b51": (head % ("5.1", "BINARY")).encode() + b"POINTS 7 double\n" + pts.astype(">f8").tobytes()
        + b"\nPOLYGONS 4 9\nOFFSETS vtktypeint64\n" + np.array([0, 3, 6, 9]).astype(">i8").tobytes()
        + b"\nCONNECTIVITY vtktypeint64\n" + faces.astype(">i8").ravel().tobytes() + b"\n",
    }
    for name, blob in files.items():
        path = tmp_path / (name + ".vtk")
        path.write_bytes(blob)
        m = read_vtk_mesh(str(path))
        want = pts.astype(np.float32).astype(np.float64) if name == "b42" else pts
        np.testing.assert_allclose(m.points, want, rtol=1e-15 if name != "a42" else 1e-12, err_msg=name)
        assert np.array_equal(m.faces, faces), name
        if name in ("a42", "b42"):
            assert m.point_data[0][0] == "thickness" and np.allclose(m.point_data[0][1], sc)
    # writer round trip (coordinates exact, scalars kept)
    from pyfocusr_amd.vtk_functions import PolyMesh, set_mesh_scalars, write_vtk_mesh

    mesh = PolyMesh(pts, faces)
    set_mesh_scalars(mesh, np.arange(7))
    write_vtk_mesh(mesh, str(tmp_path / "out.vtk"))
    back = read_vtk_mesh(str(tmp_path / "out.vtk"))
    assert np.array_equal(back.points, pts) and np.array_equal(back.faces, faces)
    assert back.point_data[0][0] == "scalars" and np.array_equal(back.point_data[0][1], np.arange(7.0))
    bad = tmp_path / "bad.vtk"
    bad.write_bytes(b"# vtk DataFile Version 4.2\nx\nASCII\nDATASET UNSTRUCTURED_GRID\n")
    with pytest.raises(NotImplementedError):
        read_vtk_mesh(str(bad))


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The drop-in boundary is a C-ABI: `include/pyfocusr_hip.h` must compile as C99 and a C program must link against
    the in-tree library and call into it (no GPU needed for pf_version / pf_device_count)."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    from pyfocusr_amd import _hip

    header = os.path.join(REPO, "include", "pyfocusr_hip.h")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", header], check=True)
    if not os.path.exists(_hip.LIB_PATH):
        pytest.skip("library not built")
    src = tmp_path / "abi.c"
    src.write_text('#include "pyfocusr_hip.h"\n#include <stdio.h>\n'
                   'int main(void) { printf("%d %d\\n", pf_version(), pf_device_count() >= 0); return 0; }\n')
    exe = tmp_path / "abi"
    libdir = os.path.dirname(_hip.LIB_PATH)
    subprocess.run([gcc, "-std=c99", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe), "-L", libdir,
                    "-l:" + os.path.basename(_hip.LIB_PATH), "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) >= 1 and out[1] == "1"


@pytest.mark.parametrize("n,m", [(1, 1), (1, 7), (5, 3), (64, 64), (1000, 999), (14998, 14996), (4000, 5000)])
def test_w1_quantile_form_equals_scipy(n, m):
    """eigsort.calc_c_hist's W1 for samples of different size (the 14 998 / 14 996-vertex pair) against
    scipy.stats.wasserstein_distance, the function eigsort.py:176-187 calls; ties included."""
    from scipy.stats import wasserstein_distance

    from pyfocusr_amd.eigsort import _w1_quantile_plan

    rng = np.random.default_rng(n * 31 + m)
    u, v = np.sort(rng.normal(size=n)), np.sort(rng.normal(size=m) * 0.7 + 0.2)
    if n > 4:
        u[1], v[:2] = u[2], u[2]  # ties within and across the samples
        u.sort(), v.sort()
    lens, iu, iv = _w1_quantile_plan(n, m)
    assert abs(lens.sum() - 1.0) < 1e-12 and iu.max() == n - 1 and iv.max() == m - 1
    np.testing.assert_allclose((np.abs(u[iu] - v[iv]) * lens).sum(), wasserstein_distance(u, v), rtol=1e-12, atol=1e-15)
