"""The C++ Krylov driver of the product (pyfocusr_amd/csrc/pf_krylov.h + pf_dense.h: what pf_eigs_smallest /
pf_eigs_smallest2 run) on a CPU TEST DOUBLE of the device primitives (tests/csrc/krylov_double.cpp), against the oracle:
the reference's `recursive_eig` -> scipy `eigs` (graph.py:357-389) on the same meshes.  Runs without a GPU; the same
driver on the device is covered by tests/test_gpu_parity.py."""
import ctypes as C

import numpy as np
import pytest
from scipy import sparse

import _krylov_double as kd
from oracle import reference_port as orc
from pyfocusr_amd.graph import _widened_k
from test_cpu_host import grid_mesh

_dp = C.POINTER(C.c_double)


def P(a):
    return a.ctypes.data_as(_dp)


def W_of(points, faces):
    return orc.weighted_adjacency(points, faces).tocsr()


def wanted_columns(W, k):
    """Column count the reference's widen-and-retry loop ends with for `n_spectral_features = k` (graph.py:374-379)."""
    from scipy.sparse.csgraph import connected_components

    n = W.shape[0]
    deg = np.asarray(W.sum(axis=1)).ravel()
    _, labels = connected_components(W + W.T, directed=False)
    sizes = np.bincount(labels)
    n_null = int(np.sum(sizes >= 2) + np.sum(deg == 0))
    k_final, _ = _widened_k(k + 1, k, 1, n_null, n)
    return k_final - n_null


# ------------------------------------------------------------------------------------------------ dense algebra
def _schur(A):
    n = A.shape[0]
    T, Z, wr, wi = np.ascontiguousarray(A, dtype=float).copy(), np.zeros((n, n)), np.zeros(n), np.zeros(n)
    assert kd.lib().td_real_schur(n, P(T), P(Z), P(wr), P(wi)) == 0
    return T, Z, wr + 1j * wi


def _block_eigs(T):
    n, i, out = T.shape[0], 0, np.zeros(T.shape[0], complex)
    while i < n:
        if i + 1 < n and T[i + 1, i] != 0:
            out[i:i + 2] = np.linalg.eigvals(T[i:i + 2, i:i + 2])
            i += 2
        else:
            out[i] = T[i, i]
            i += 1
    return out


@pytest.mark.parametrize("kind", ["general", "hessenberg", "symmetric", "nearly_diagonal", "repeated_block"])
def test_real_schur_reorder_eigenvectors(kind):
    """pf_dense.h against numpy: A = Z T Z^T with T quasi-triangular, eigenvalues, selected blocks moved to the top
    (what scipy.linalg.schur(sort=...) does for _krylov._ordered_schur), complex eigenvectors from the Schur form."""
    rng = np.random.default_rng(hash(kind) % 1000)
    lib = kd.lib()
    for trial in range(40):
        n = int(rng.integers(1, 70))
        A = rng.standard_normal((n, n))
        if kind == "hessenberg":
            A = np.triu(A, -1)
        elif kind == "symmetric":
            A = A + A.T
        elif kind == "nearly_diagonal":
            A = np.diag(rng.standard_normal(n)) + 1e-3 * A
        elif kind == "repeated_block" and n > 3:  # the locked null vectors of the driver: theta0 I in the corner
            A[:3, :3] = 2.0 * np.eye(3)
            A[3:, :3] = 0
        scale = max(np.abs(A).max(), 1.0)
        T, Z, ev = _schur(A)
        assert np.abs(Z @ T @ Z.T - A).max() < 1e-12 * scale and np.abs(Z.T @ Z - np.eye(n)).max() < 1e-12
        assert n < 3 or np.abs(np.tril(T, -2)).max() == 0
        sd = np.diag(T, -1) != 0
        assert not np.any(sd[:-1] & sd[1:])
        ref = np.linalg.eigvals(A)
        assert max(np.min(np.abs(ref - e)) for e in ev) < 1e-9 * scale
        wr, wi = np.zeros(n), np.zeros(n)
        assert lib.td_eigenvalues(n, P(np.ascontiguousarray(A)), P(wr), P(wi)) == 0
        assert max(np.min(np.abs(ref - e)) for e in wr + 1j * wi) < 1e-9 * scale
        rows = _block_eigs(T)
        sel = np.abs(rows) > np.median(np.abs(rows))
        i = 0
        while i < n:  # both rows of a 2 x 2 block carry the same flag
            if i + 1 < n and T[i + 1, i] != 0:
                sel[i] = sel[i + 1] = sel[i] or sel[i + 1]
                i += 2
            else:
                i += 1
        T2, Z2, ok = T.copy(), Z.copy(), C.c_int32()
        top = lib.td_schur_reorder(n, P(T2), P(Z2), np.ascontiguousarray(sel.astype(np.int8)).ctypes.data_as(C.c_char_p), C.byref(ok))
        assert np.abs(Z2 @ T2 @ Z2.T - A).max() < 1e-11 * scale and np.abs(Z2.T @ Z2 - np.eye(n)).max() < 1e-12
        if ok.value:
            assert top == sel.sum()
            lead = np.linalg.eigvals(T2[:top, :top]) if top else np.zeros(0)
            np.testing.assert_allclose(np.sort(np.abs(lead)), np.sort(np.abs(rows[sel])), rtol=1e-8, atol=1e-10)
            assert top in (0, n) or np.abs(T2[top:, :top]).max() == 0
        V, evv = np.zeros((n, n, 2)), np.zeros((n, 2))
        lib.td_schur_eigenvectors(n, P(T), P(Z), P(V), P(evv))
        Vc, lam = V[..., 0] + 1j * V[..., 1], evv[:, 0] + 1j * evv[:, 1]
        assert np.abs(A @ Vc - Vc * lam[None, :]).max() < 1e-9 * scale * max(np.linalg.cond(Vc), 1.0)
        np.testing.assert_allclose(np.linalg.norm(Vc, axis=0), 1.0)


def test_eigh_sym_and_hessenberg_residual_estimate():
    rng = np.random.default_rng(5)
    lib = kd.lib()
    for n in (1, 2, 7, 33, 60):
        A = rng.standard_normal((n, n))
        A = A + A.T
        V, d = np.ascontiguousarray(A).copy(), np.zeros(n)
        lib.td_eigh_sym(n, P(V), P(d))
        assert np.abs(A @ V - V * d[None, :]).max() < 1e-12 * max(np.abs(A).max(), 1) * n
        np.testing.assert_allclose(np.sort(d), np.linalg.eigvalsh(A), atol=1e-12 * max(np.abs(A).max(), 1) * n)
    # Arnoldi matrix of a mildly non-symmetric operator: the O(n^2) recurrence gives the Ritz residual factors
    # |s_last| / |s| that the eigenvectors of H would give
    n, N = 40, 200
    A = rng.standard_normal((N, N))
    A = A + A.T + 0.3 * rng.standard_normal((N, N))
    Vs, H = [rng.standard_normal(N)], np.zeros((n + 1, n))
    Vs[0] /= np.linalg.norm(Vs[0])
    for j in range(n):
        w = A @ Vs[j]
        for _ in range(2):
            for i in range(j + 1):
                c = Vs[i] @ w
                H[i, j] += c
                w -= c * Vs[i]
        H[j + 1, j] = np.linalg.norm(w)
        Vs.append(w / H[j + 1, j])
    Hn = np.ascontiguousarray(H[:n, :n])
    ev, S = np.linalg.eig(Hn)
    for k in np.argsort(-np.abs(ev))[:12]:
        true = abs(S[-1, k]) / np.linalg.norm(S[:, k])
        est = lib.td_hessenberg_residual_factor(n, P(Hn), 0, ev[k].real, ev[k].imag)
        assert abs(est - true) < 1e-6 * true + 1e-15


# ------------------------------------------------------------------------------------------------ the driver
def canonical(vals, vecs):
    X = vecs / np.linalg.norm(vecs, axis=0)
    return orc.canonicalize(vals, X)


@pytest.mark.parametrize("name,k", [("target_mesh", 6), ("source_mesh", 3), ("target_mesh_15k", 5), ("source_mesh_15k", 9)])
def test_cpp_driver_matches_reference(golden, name, k):
    """Lanczos (5k meshes: symmetric W) and Arnoldi with carried complex outliers (15k meshes: one-way edges) against the
    reference-generated goldens."""
    g = golden(name)
    gk = {"target_mesh": 6, "source_mesh": 3, "target_mesh_15k": 5, "source_mesh_15k": 5}[name]
    W = W_of(g["points"], g["faces"])
    vals, vecs, st, res = kd.solve(W, k)
    gv = g["k%d_eig_vals" % gk]
    m = min(len(gv), len(vals))
    assert len(vals) == k and st["mode"] == (1 if "15k" in name else 0)
    np.testing.assert_allclose(vals[:m], gv[:m], rtol=1e-8)
    lam, X = canonical(vals, vecs)
    tol = 5e-7 if "15k" in name else 2e-9
    assert np.max(np.abs(orc.minmax_normalize(X)[:, :m] - g["k%d_eig_vecs" % gk][:, :m])) < tol
    assert res.max() < 1e-8 and st["filter_resets"] == 0
    if "15k" in name:
        assert st["degree"] <= 128
    # a Gram-Schmidt step that reports its second pass makes the driver repeat the filter application: same pairs
    vals2, _, st2, _ = kd.solve(W, k, redone_every=5)
    np.testing.assert_allclose(vals2, vals, rtol=1e-10)
    assert st2["second_passes"] >= 3 and st2["matvecs"] > st["matvecs"]


def test_cpp_driver_components_isolated_and_restart():
    from pyfocusr_amd.meshgen import blob_mesh

    parts = [blob_mesh(n, seed=30 + i) for i, n in enumerate((900, 700, 500))]
    pts = np.concatenate([p.points + 300.0 * i for i, p in enumerate(parts)] + [np.zeros((2, 3))])
    off = np.cumsum([0] + [len(p.points) for p in parts])
    faces = np.concatenate([p.faces + off[i] for i, p in enumerate(parts)])
    ref = orc.graph_spectrum(pts, faces, 4)  # 3 + 2 nulls -> widened twice
    W = W_of(pts, faces)
    assert wanted_columns(W, 4) == len(ref["eig_vals"])
    vals, vecs, st, res = kd.solve(W, len(ref["eig_vals"]))
    assert st["n_null"] == 3 and st["mode"] == 0
    np.testing.assert_allclose(vals, ref["eig_vals"], rtol=1e-8)
    assert np.all(vecs[-2:, :] == 0.0)  # isolated vertices stay out of every eigenvector
    L = ref["L"]
    assert np.abs(L @ vecs - vecs * vals[None, :]).max() < 1e-9
    # a small basis: thick restarts, symmetric and asymmetric
    m = blob_mesh(700, seed=2)
    refm = np.sort(np.linalg.eigvals(orc.graph_matrices(m.points, m.faces)[3].toarray()).real)
    vals, vecs, st, res = kd.solve(W_of(m.points, m.faces), 6, m_max_limit=14)
    assert st["restarts"] >= 1
    np.testing.assert_allclose(vals, refm[refm > 1e-10][:6], rtol=1e-8)
    faces = np.delete(m.faces, [5, 300, 900], axis=0)  # three one-way triangles
    refa = np.sort(np.linalg.eigvals(orc.graph_matrices(m.points, faces)[3].toarray()).real)
    vals, vecs, st, res = kd.solve(W_of(m.points, faces), 6, m_max_limit=26)
    assert st["restarts"] >= 1 and st["mode"] == 1
    np.testing.assert_allclose(vals, refa[refa > 1e-10][:6], rtol=1e-8)


def test_cpp_driver_refuses_what_it_does_not_cover():
    from pyfocusr_amd.meshgen import blob_mesh

    m = blob_mesh(40, seed=1)
    with pytest.raises(kd.DoubleError) as e:  # too small for the filtered iteration: the Python driver's plain mode
        kd.solve(W_of(m.points, m.faces), 5)
    assert e.value.code == -4 and "Python driver" in str(e.value)


@pytest.mark.parametrize("nx,ny", [(40, 40), (100, 80)])
def test_cpp_driver_open_mesh_ellipse_filter(nx, ny):
    """Open meshes: complex LOW eigenvalues (the reference keeps the real parts, graph.py:386: pairs show up as repeated
    values).  The interval filter cannot work; the driver must end up with the ellipse filter whichever way it starts."""
    pts, faces = grid_mesh(nx, ny)
    ref = orc.graph_spectrum(pts, faces, 5)
    W = W_of(pts, faces)
    for hint in (-1, 0, 1):  # by the count of one-way edges (here: ellipse at once) / interval first / ellipse at once
        vals, vecs, st, res = kd.solve(W, 6, ellipse_hint=hint)
        m = min(len(vals), len(ref["eig_vals"]))
        # (interval first: when the outliers can be carried, the complex low pairs come out of the interval filter's
        # subspace; otherwise the attempt is given up for the ellipse)
        assert m >= 5 and (st["mode"] == 2 or hint == 0)
        np.testing.assert_allclose(vals[:m], ref["eig_vals"][:m], rtol=1e-7)
        assert np.isclose(vals[0], vals[1], rtol=1e-9)  # a conjugate pair


def test_cpp_driver_holes_interval_filter_carries_complex_low_pairs():
    """A cap cut out of a closed blob (39 one-way boundary edges, 72 stranded vertices): the low eigenvalues feel the hole
    and come in complex pairs, the interval filter's subspace still holds them - same values as the ellipse filter's, at
    a third of the operator applications."""
    from pyfocusr_amd.meshgen import blob_mesh

    m = blob_mesh(8000, seed=5)
    faces = m.faces[np.linalg.norm(m.points[m.faces].mean(axis=1) - m.points[10], axis=1) > 6.0]
    W = W_of(m.points, faces)
    a, _, sa, _ = kd.solve(W, 5, ellipse_hint=0)
    b, _, sb, _ = kd.solve(W, 5, ellipse_hint=1)
    assert sa["mode"] == 1 and sb["mode"] == 2 and sa["matvecs"] < 0.5 * sb["matvecs"]
    np.testing.assert_allclose(a, b, rtol=1e-10)
    ev = np.linalg.eigvals(orc.graph_matrices(m.points, faces)[3].toarray())
    low = np.sort(ev.real[ev.real > 1e-10])[:5]
    np.testing.assert_allclose(a, low, rtol=1e-8)


@pytest.mark.parametrize("n,seed", [(6000, 0), (9000, 3)])
def test_cpp_driver_messy_blob(n, seed):
    """A closed mesh with the defect classes of the bundled scans (`messy_blob_mesh`: ~50 one-way edges, duplicated
    directed edges, edges in three faces, stranded vertices): Arnoldi with the interval filter, outliers carried."""
    from pyfocusr_amd.meshgen import messy_blob_mesh

    mesh = messy_blob_mesh(n, seed=seed)
    ref = orc.graph_spectrum(mesh.points, mesh.faces, 5)
    W = W_of(mesh.points, mesh.faces)
    assert abs(W - W.T).nnz > 64
    vals, vecs, st, res = kd.solve(W, wanted_columns(W, 5), ellipse_hint=0)
    assert len(vals) == len(ref["eig_vals"]) and st["mode"] == 1
    np.testing.assert_allclose(vals, ref["eig_vals"], rtol=1e-8)
    assert np.abs(ref["L"] @ vecs - vecs * vals[None, :]).max() < 1e-9


def test_cpp_pair_driver_equals_single_solves(golden):
    """`pfk::drive_pair` (two solvers in lockstep, fused requests) returns what two single solves return, for graphs of
    different size, symmetry and step count."""
    gs = [golden("target_mesh"), golden("source_mesh_15k")]
    Ws = [W_of(g["points"], g["faces"]) for g in gs]
    ra, rb, calls = kd.solve_pair(Ws[0], 6, Ws[1], 9)
    assert calls > 10
    for W, k, (vals, vecs, st) in zip(Ws, (6, 9), (ra, rb)):
        v1, x1, s1, _ = kd.solve(W, k)
        assert np.array_equal(vals, v1) and np.array_equal(vecs, x1) and st["matvecs"] == s1["matvecs"]


def test_cpp_driver_partial_reorthogonalisation(monkeypatch):
    """Symmetric graphs of 4096 vertices and more run Lanczos with PARTIAL reorthogonalisation (pf_krylov.h: most steps
    orthogonalise against the null vectors and the last two basis vectors only, estimates of the drift decide when two
    full Gram-Schmidt steps are due; the extraction orthonormalises its Ritz vectors): same eigenpairs as with full
    Gram-Schmidt in every step and as the oracle, most steps local - alone, in a pair, and with several components."""
    from pyfocusr_amd.meshgen import blob_mesh

    m = blob_mesh(9000, seed=51)
    W = W_of(m.points, m.faces)
    vals, vecs, st, res = kd.solve(W, 6)
    assert st["mode"] == 0 and st["restarts"] == 0
    assert st["local_steps"] >= 0.6 * st["outer_steps"] > 0, st
    # the device's steps take ONE Gram-Schmidt pass unless digits cancel; the full steps of such a run ask for two (one pass
    # against a basis that is orthogonal to 1e-9 only would leave the new vector there and the estimates wrong: seen as
    # "no convergence" on a two-component mesh of the GPU fuzzer before they did)
    monkeypatch.setenv("TD_ONE_PASS", "1")
    vals_1, vecs_1, st_1, res_1 = kd.solve(W, 6)
    monkeypatch.delenv("TD_ONE_PASS")
    assert st_1["restarts"] == 0 and st_1["local_steps"] > 0 and st_1["outer_steps"] == st["outer_steps"]
    np.testing.assert_allclose(vals_1, vals, rtol=1e-11)
    monkeypatch.setenv("PF_EIGS_PRO", "0")
    vals_f, vecs_f, st_f, res_f = kd.solve(W, 6)
    monkeypatch.delenv("PF_EIGS_PRO")
    assert st_f["local_steps"] == 0 and st_f["outer_steps"] == st["outer_steps"]
    np.testing.assert_allclose(vals, vals_f, rtol=2e-12)
    assert np.max(np.abs(np.abs(np.sum(vecs * vecs_f, axis=0)) / (np.linalg.norm(vecs, axis=0) * np.linalg.norm(vecs_f, axis=0)) - 1.0)) < 1e-10
    assert res.max() < 4 * max(res_f.max(), 1e-13)
    ref = orc.graph_spectrum(m.points, m.faces, 5)
    np.testing.assert_allclose(vals[: len(ref["eig_vals"])], ref["eig_vals"][: len(vals)], rtol=1e-9)
    L = ref["L"]
    assert np.abs(L @ vecs - vecs * vals[None, :]).max() < 1e-10
    # two components (two locked null vectors in every local step) beside a single one, as a pair
    parts = [blob_mesh(n, seed=60 + i) for i, n in enumerate((5000, 4200))]
    pts = np.concatenate([p.points + 400.0 * i for i, p in enumerate(parts)])
    faces = np.concatenate([parts[0].faces, parts[1].faces + len(parts[0].points)])
    W2 = W_of(pts, faces)
    monkeypatch.setenv("TD_ONE_PASS", "1")
    (va, xa, sa), (vb, xb, sb), _ = kd.solve_pair(W2, 7, W, 6)
    monkeypatch.delenv("TD_ONE_PASS")
    assert sa["n_null"] == 2 and sa["local_steps"] > 0 and sb["local_steps"] > 0
    np.testing.assert_allclose(vb, vals, rtol=1e-11)
    L2 = orc.graph_matrices(pts, faces)[3]
    assert np.abs(L2 @ xa - xa * va[None, :]).max() < 1e-10


def test_cpp_driver_partial_reorthogonalisation_degenerate_and_restart(monkeypatch):
    """A torus grid (every eigenvalue of its Laplacian twice, by symmetry) needs a thick restart: the Ritz vectors kept from
    a basis that partial reorthogonalisation left orthogonal to 1e-9 only are orthonormalised again before the iteration
    goes on with full steps - both copies of every eigenvalue, residuals at the level of a run with full Gram-Schmidt,
    also when the steps take one Gram-Schmidt pass like the device's (before: 78 instead of 61 steps, residual 3e-10)."""
    nu, nv, k = 128, 64, 9
    uu, vv = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    au, av = 2 * np.pi * uu / nu, 2 * np.pi * vv / nv
    pts = np.stack([(3 + np.cos(av)) * np.cos(au), (3 + np.cos(av)) * np.sin(au), np.sin(av)], axis=-1).reshape(-1, 3)
    a, b, c, d = uu * nv + vv, ((uu + 1) % nu) * nv + vv, ((uu + 1) % nu) * nv + (vv + 1) % nv, uu * nv + (vv + 1) % nv
    faces = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)]).astype(np.int32)
    W = W_of(pts, faces)
    monkeypatch.setenv("PF_EIGS_PRO", "0")
    vals_f, _, st_f, res_f = kd.solve(W, k)
    monkeypatch.delenv("PF_EIGS_PRO")
    for one_pass in (False, True):
        if one_pass:
            monkeypatch.setenv("TD_ONE_PASS", "1")
        vals, vecs, st, res = kd.solve(W, k)
        assert st["restarts"] >= 1 and st["local_steps"] > 0, st
        assert st["outer_steps"] <= st_f["outer_steps"] + 6, (st, st_f)
        np.testing.assert_allclose(vals, vals_f, rtol=1e-12)
        assert res.max() < 1e-11
        assert np.abs(vals[0] / vals[1] - 1.0) < 1e-9 and np.abs(vals[2] / vals[3] - 1.0) < 1e-9  # the pairs
