"""ORACLE — test infrastructure, NOT product code.

CPU restatement (numpy / scipy) of the reference's spectral hot path
(`/root/reference/pyfocusr/graph.py`, `eigsort.py`, `focusr.py:351-366,459-508`).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this module; nothing under `pyfocusr_amd/` (or `tools/`) does.

Parity pin: every function below is checked against the reference itself,
imported in the build container with stub `vtk`/`itkwidgets`/`cycpd` modules
(`tools/make_golden.py`), through the committed fixtures in `tests/golden/` and
the known-answer eigenvalues printed in the reference's notebook
(`examples/Example_registering_two_bone_meshes.ipynb`, cells 2 and 13).

The heavy arithmetic of the reference lives in third-party scipy (un-pinned in
the reference's `requirements.txt:1-8`; 1.15.3 here): ARPACK `eigs` in
shift-invert mode, `KDTree.query`, `wasserstein_distance`,
`linear_sum_assignment`.  The oracle issues the *same scipy calls with the same
parameters* at the reference's call sites and restates everything around them.
"""
import numpy as np
from scipy import sparse
from scipy.optimize import linear_sum_assignment
from scipy.sparse.linalg import eigs
from scipy.spatial import KDTree
from scipy.stats import wasserstein_distance

MIN_EIG_VAL = 1e-10  # graph.py:369


# --------------------------------------------------------------------------------------
# graph.py:58-82  geometry part of Graph.__init__
# --------------------------------------------------------------------------------------
def geometry(points):
    pts_scale_range = np.ptp(points, axis=0)  # graph.py:63
    max_range = np.max(pts_scale_range)  # :64
    mean_range = np.mean(pts_scale_range)  # :65
    normed = (points - np.min(points, axis=0)) / mean_range  # :67
    return pts_scale_range, max_range, mean_range, normed


def list_rand_idxs(n_points, n_rand_samples, rng=None):
    """graph.py:274-290.  Strict `>` (A7); the reference's draw is unseeded."""
    if n_rand_samples > n_points:
        return np.arange(n_points)
    rng = np.random if rng is None else rng
    return rng.choice(n_points, size=n_rand_samples, replace=False)


# --------------------------------------------------------------------------------------
# graph.py:148-178  weighted adjacency, directed "set" semantics
# --------------------------------------------------------------------------------------
def directed_edges(faces):
    """Directed edges in the reference's visiting order: for each cell, for each
    VTK polygon edge e: (v_e, v_{e+1 mod nv})  (graph.py:156-161)."""
    f = np.asarray(faces, dtype=np.int64)
    nv = f.shape[1]
    src = f.reshape(-1)  # cell-major, edge-minor: identical to the nested loops
    dst = np.roll(f, -1, axis=1).reshape(-1)
    assert nv >= 2
    return src, dst


def weighted_adjacency(points, faces):
    """W[p1,p2] = 1/sqrt(sum((X1-X2)^2)), ASSIGNED per directed face edge
    (graph.py:177-178): a directed edge seen twice is stored once (last write,
    same value), a one-way edge makes W asymmetric.  Returns scipy CSR with
    sorted indices (what `lil_matrix.tocsr()` yields)."""
    n = len(points)
    src, dst = directed_edges(faces)
    d = points[src] - points[dst]
    sq = np.square(d)
    dist = np.sqrt((sq[:, 0] + sq[:, 1]) + sq[:, 2])  # np.sum over 3 elements is left-to-right
    with np.errstate(divide="ignore"):
        w = 1.0 / dist
    key = src * n + dst
    # last write wins: walk reversed so np.unique keeps the final occurrence.
    ukey, first_rev = np.unique(key[::-1], return_index=True)
    vals = w[::-1][first_rev]
    rows = (ukey // n).astype(np.int64)
    cols = (ukey % n).astype(np.int64)
    W = sparse.csr_matrix((vals, (rows, cols)), shape=(n, n))
    W.sort_indices()
    return W


def degree_and_inverse(W):
    """graph.py:216-219: deg = row sums (out-degree); D_inv = (deg + 1e-8)^-1."""
    # `lil_matrix.sum(axis=1)` is the generic `self @ ones((n, 1))`: a strictly
    # left-to-right sum in column order (CSR's own `.sum` uses add.reduceat, whose
    # rounding differs), so restate it as the same product.
    deg = np.asarray(W @ np.ones((W.shape[1], 1)))[:, 0]
    d_inv = (deg + 1e-8) ** -1
    return deg, d_inv


def laplacian(W, deg, d_inv):
    """graph.py:221-226 with G = D_inv (graph.py:213-214): L = G @ (D - W).
    Issued through the same scipy operators so the stored pattern (zero diagonals
    of isolated vertices dropped) and the rounding are the reference's."""
    D = sparse.diags(deg)
    G = sparse.diags(d_inv)
    L = G @ (D - W)
    L = sparse.csr_matrix(L)
    L.sort_indices()
    return L


def graph_matrices(points, faces):
    W = weighted_adjacency(points, faces)
    deg, d_inv = degree_and_inverse(W)
    L = laplacian(W, deg, d_inv)
    return W, deg, d_inv, L


# --------------------------------------------------------------------------------------
# graph.py:357-389 recursive_eig  (scipy ARPACK shift-invert, as called by the reference)
# --------------------------------------------------------------------------------------
def recursive_eig(matrix, k, n_k_needed, k_buffer=1, sigma=1e-10, which="LM", trace=None):
    """Same widen-and-retry rule as the reference (`k += k_buffer + n_k_needed`),
    as a bounded loop.  Returns the reference's UNSORTED (vals, vecs) real parts."""
    n = matrix.shape[0]
    while True:
        eig_vals, eig_vecs = eigs(matrix, k=k, sigma=sigma, which=which, ncv=4 * k)
        if trace is not None:
            trace.append(int(k))
        n_good = int(np.sum(eig_vals > MIN_EIG_VAL))
        if n_good >= n_k_needed or k + k_buffer + n_k_needed >= n - 1:
            break
        k += k_buffer + n_k_needed
    keep = np.where(eig_vals > MIN_EIG_VAL)[0]
    return np.real(eig_vals[keep]), np.real(eig_vecs[:, keep])


def canonicalize(eig_vals, eig_vecs):
    """Strengthening shared by oracle, fixtures and product: ascending
    eigenvalues; each column's sign chosen so its largest-|entry| is positive."""
    order = np.argsort(eig_vals, kind="stable")
    vals = np.asarray(eig_vals)[order]
    vecs = np.array(eig_vecs[:, order], dtype=np.float64, copy=True)
    piv = np.argmax(np.abs(vecs), axis=0)
    sgn = np.sign(vecs[piv, np.arange(vecs.shape[1])])
    sgn[sgn == 0] = 1.0
    vecs *= sgn[None, :]
    return vals, vecs


def minmax_normalize(eig_vecs):
    """graph.py:254-257."""
    return (eig_vecs - np.min(eig_vecs, axis=0)) / np.ptp(eig_vecs, axis=0) - 0.5


def graph_spectrum(points, faces, n_spectral_features, norm_eig_vecs=True, trace=None):
    """graph.py:228-257 end to end; canonicalised (sorted, sign-fixed) output."""
    W, deg, d_inv, L = graph_matrices(points, faces)
    vals, vecs = recursive_eig(
        L, k=n_spectral_features + 1, n_k_needed=n_spectral_features, k_buffer=1, trace=trace
    )
    vals, vecs = canonicalize(vals, vecs)
    raw = vecs
    if norm_eig_vecs:
        vecs = minmax_normalize(vecs)
    return dict(W=W, deg=deg, d_inv=d_inv, L=L, eig_vals=vals, eig_vecs_raw=raw, eig_vecs=vecs)


# --------------------------------------------------------------------------------------
# graph.py:263-272 samplers
# --------------------------------------------------------------------------------------
def eig_val_gap(eig_vals):
    return np.mean(np.diff(eig_vals))


def rand_normalized_points(points, rand_idxs):
    p = points[rand_idxs, :]
    return (p - np.min(p, axis=0)) / np.ptp(p, axis=0)


# --------------------------------------------------------------------------------------
# graph.py:320-354 mean filter ("next" row f1)
# --------------------------------------------------------------------------------------
def mean_filter_graph(W, values, iterations=300):
    d_inv = sparse.diags(1.0 / (1 + np.asarray(W @ np.ones((W.shape[1], 1)))[:, 0]))
    out = values
    avg = d_inv @ (W + sparse.eye(W.shape[0]))
    for _ in range(iterations):
        out = avg @ out
    return out


# --------------------------------------------------------------------------------------
# eigsort.py
# --------------------------------------------------------------------------------------
def c_lambda(vals_t, vals_s, k):
    """eigsort.py:142-160.  The gap uses ALL returned eigenvalues (A6)."""
    gap = (eig_val_gap(vals_t) + eig_val_gap(vals_s)) / 2
    out = np.zeros((k, k))
    for i in range(k):
        for j in range(k):
            out[i, j] = np.exp((vals_t[i] - vals_s[j]) ** 2 / (2 * gap**2))
    return out


def c_hist(T, S, k):
    """eigsort.py:162-189."""
    eps = np.finfo(float).eps
    c = np.zeros((k, k))
    cf = np.zeros((k, k))
    for i in range(k):
        lt = np.log(T[:, i] + 0.5 + eps)
        for j in range(k):
            c[i, j] = wasserstein_distance(lt, np.log(S[:, j] + 0.5 + eps))
            cf[i, j] = wasserstein_distance(lt, np.log(-S[:, j] + 0.5 + eps))
    return c, cf


def knn1(ref_pts, qry_pts):
    """focusr.py:351-353 / eigsort.py:203-204: KDTree(ref).query(qry), k=1, p=2."""
    _, idx = KDTree(ref_pts).query(qry_pts)
    return idx


def knn1_bruteforce(ref_pts, qry_pts, chunk=2048):
    """Exhaustive 1-NN, sequential sum of squared differences, lowest index on
    ties — the arithmetic the HIP kernel restates; cross-check for `knn1`."""
    ref = np.ascontiguousarray(ref_pts, dtype=np.float64)
    qry = np.ascontiguousarray(qry_pts, dtype=np.float64)
    idx = np.empty(len(qry), dtype=np.int64)
    d2min = np.empty(len(qry), dtype=np.float64)
    for s in range(0, len(qry), chunk):
        q = qry[s : s + chunk]
        acc = np.zeros((len(q), len(ref)))
        for j in range(ref.shape[1]):
            diff = q[:, j][:, None] - ref[:, j][None, :]
            acc = acc + diff * diff
        idx[s : s + chunk] = np.argmin(acc, axis=1)
        d2min[s : s + chunk] = np.min(acc, axis=1)
    return idx, d2min


def c_spatial(T, S, rand_target_points, rand_source_points, k):
    """eigsort.py:191-233."""
    idx = knn1(rand_source_points, rand_target_points)
    m = T.shape[0]
    c = np.zeros((k, k))
    cf = np.zeros((k, k))
    for i in range(k):
        for j in range(k):
            c[i, j] = np.sqrt(np.sum((S[idx, j] - T[:, i]) ** 2)) / m
            cf[i, j] = np.sqrt(np.sum((-S[idx, j] - T[:, i]) ** 2)) / m
    return c, cf, idx


def eigen_sort(cl, ch, chf, cs, csf, eig_vecs_t, eig_vecs_s, target_as_reference=True):
    """eigsort.py:54-122.  Returns Q (k,), matches, flipped pairs and the
    flipped/permuted COPIES of the eigenvector matrices (the reference mutates in
    place)."""
    c = cs * cl * ch
    c_f = csf * cl * chf
    Q = np.min((c, c_f), axis=0)
    S = c > c_f
    t_flip, s_flip = np.where(S == True)  # noqa: E712
    if target_as_reference:
        t_match, s_match = linear_sum_assignment(Q)
    else:
        s_match, t_match = linear_sum_assignment(Q.T)
    Qm = Q[t_match, s_match]
    flipped = [p2 for p1 in zip(t_flip, s_flip) for p2 in zip(t_match, s_match) if p2 == p1]
    vt = np.array(eig_vecs_t, copy=True)
    vs = np.array(eig_vecs_s, copy=True)
    for m0, m1 in flipped:
        if target_as_reference:
            vs[:, m1] = vs[:, m1] * -1
        else:
            vt[:, m0] = vt[:, m0] * -1
    if target_as_reference:
        vs[:, t_match] = vs[:, s_match]
    else:
        vt[:, s_match] = vt[:, t_match]
    return Qm, np.asarray(t_match), np.asarray(s_match), flipped, vt, vs


def sort_eigenmaps(points_t, points_s, vals_t, vals_s, vecs_t, vecs_s, rand_t, rand_s, k,
                   target_as_reference=True):
    """eigsort.py:235-249 on explicit inputs."""
    T = vecs_t[rand_t, :]
    S = vecs_s[rand_s, :]
    pt = rand_normalized_points(points_t, rand_t)
    ps = rand_normalized_points(points_s, rand_s)
    cl = c_lambda(vals_t, vals_s, k)
    ch, chf = c_hist(T, S, k)
    cs, csf, idx3d = c_spatial(T, S, pt, ps, k)
    Q, tm, sm, flipped, vt, vs = eigen_sort(cl, ch, chf, cs, csf, vecs_t, vecs_s, target_as_reference)
    return dict(c_lambda=cl, c_hist=ch, c_hist_f=chf, c_spatial=cs, c_spatial_f=csf, Q=Q,
                target_matches=tm, source_matches=sm,
                flipped_pairs=np.asarray(flipped, dtype=np.int64).reshape(-1, 2),
                eig_vecs_t=vt, eig_vecs_s=vs, idx_spatial=idx3d)


# --------------------------------------------------------------------------------------
# focusr.py:459-508 spectral coordinates
# --------------------------------------------------------------------------------------
def spectral_weights(Q, vals_s, vals_t, ns):
    w = Q[:ns] * np.max((vals_s[:ns], vals_t[:ns]), axis=0)
    sigma = np.mean(w)
    return np.exp(-(w**2) / (2 * sigma**2))


def spectral_coords(vecs_s, vecs_t, ns, weights=None):
    if weights is None:
        return vecs_s[:, :ns], vecs_t[:, :ns]
    return vecs_s[:, :ns] * weights[None, :], vecs_t[:, :ns] * weights[None, :]


# --------------------------------------------------------------------------------------
# focusr.py:368-431 smoothed correspondences and final node locations ("next" rows f1/f2)
# --------------------------------------------------------------------------------------
def smoothed_correspondences(W_t, W_s, points_t, idx_initial, graph_iters=300, proj_iters=40):
    """focusr.py:368-396 with final_correspondence_type == "kd"."""
    smoothed_target = mean_filter_graph(W_t, points_t, iterations=graph_iters)
    projected = mean_filter_graph(W_s, smoothed_target[idx_initial, :], iterations=proj_iters)
    return smoothed_target, projected, knn1(smoothed_target, projected)


def weighted_final_node_locations(smoothed_target, projected, points_t, n_closest_pts=3):
    """focusr.py:401-426, the reference's per-point loop."""
    out = np.zeros((len(projected), 3))
    tree = KDTree(smoothed_target)
    for i in range(len(projected)):
        d, idx = tree.query(projected[i, :], k=n_closest_pts)
        if 0 in d:
            out[i, :] = points_t[idx[np.where(d == 0)[0][0]]]
        else:
            w = 1 / d[:, None]
            out[i, :] = np.sum(points_t[idx, :] * w, axis=0) / (sum(w))
    return out
