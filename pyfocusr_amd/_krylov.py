"""Host driver of the GPU eigensolver: Chebyshev-filtered Krylov-Schur.

Replaces the reference's `scipy.sparse.linalg.eigs(L, k, sigma=1e-10, which="LM",
ncv=4k)` call (`/root/reference/pyfocusr/graph.py:372`, ARPACK shift-invert on a
SuperLU factorisation) with an SpMV-only method that suits the GPU:

* the operator is either the symmetrised Laplacian `S = G^1/2 (D-W) G^1/2`
  (when W is symmetric; same eigenvalues as `L = G (D-W)`, `x_L = G^1/2 x_S`) or
  `L` itself (one-way edges make W asymmetric on the bundled 15k meshes);
* `B = T_p((c I - A)/e)` — a degree-p Chebyshev polynomial that maps the unwanted
  spectrum [a, 2] into [-1, 1] and the wanted low end to cosh-sized values — is
  applied by p fused SpMV+recurrence kernel launches;
* a Krylov-Schur (thick-restart Lanczos when A is symmetric, restarted Arnoldi
  otherwise) iteration on B with full CGS2 re-orthogonalisation finds the
  dominant invariant subspace of B in a few tens of steps;
* null vectors are known analytically per connected component (1_C for L,
  G^-1/2 1_C for S) and are locked in the basis from the start; isolated vertices
  (all-zero rows) are masked out and only counted;
* a final Rayleigh-Ritz step on A itself over the converged subspace yields the
  eigenvalues of A and the reference's `> 1e-10` null filter is applied to them.

Everything O(n) runs on the device through the `ops` object (the ctypes wrapper of
the C-ABI, `pyfocusr_amd/_hip.py`); this module only does the O(m^2)..O(m^3) dense
algebra on the small projected matrices and the control flow.
"""
import math

import numpy as np
import scipy.linalg as sla

MIN_EIG_VAL = 1e-10  # graph.py:369


class EigsStats(object):
    def __init__(self):
        self.matvecs = 0  # SpMV-equivalent launches of the fused Chebyshev/SpMV kernel
        self.outer_steps = 0
        self.restarts = 0
        self.filter_resets = 0
        self.degree = 0
        self.cut = 0.0
        self.residuals = None  # ||A x - lambda x||_2 of the returned pairs (operator actually iterated)
        self.n_null = 0
        self.second_passes = 0  # outer steps whose Gram-Schmidt projection cancelled digits (second pass run by orth_end)

    def as_dict(self):
        return dict(self.__dict__)


def _cheb_value(lam, c, e, p):
    """T_p((c - lam)/e) for real lam."""
    t = (c - lam) / e
    if abs(t) <= 1.0:
        return math.cos(p * math.acos(t))
    s = 1.0 if t > 0 or p % 2 == 0 else -1.0
    return s * math.cosh(p * math.acosh(abs(t)))


def _cheb_value_scaled(lam, c, e, p, rho):
    """T_p((c - lam)/e) / rho^p without overflow."""
    if rho == 1.0:
        return _cheb_value(lam, c, e, p)
    t = (c - lam) / e
    if abs(t) <= 1.0:
        return math.cos(p * math.acos(t)) * math.exp(-p * math.log(rho))
    u = math.acosh(abs(t))
    s = 1.0 if t > 0 or p % 2 == 0 else -1.0
    return s * 0.5 * math.exp(p * (u - math.log(rho))) * (1.0 + math.exp(-2.0 * p * u))


def _cheb_inverse(theta, c, e, p):
    """lam < a with T_p((c-lam)/e) = theta > 1."""
    return c - e * math.cosh(math.acosh(max(theta, 1.0)) / p)


def choose_filter(cut, hi=2.0, strength=2.0, min_degree=8, max_degree=4000):
    """Damped interval [cut, hi]; degree such that eigenvalues <= cut/2 are
    amplified by >= cosh(strength) relative to the damped part."""
    cut = min(max(cut, 1e-12), 0.5 * hi)
    c = 0.5 * (hi + cut)
    e = 0.5 * (hi - cut)
    growth_rate = math.acosh((c - 0.5 * cut) / e)  # acosh of the map of cut/2
    p = int(math.ceil(strength / growth_rate))
    p = max(min_degree, min(max_degree, p))
    return c, e, p


def choose_filter_ellipse(cut, half_height, hi=2.0, strength=2.0, min_degree=8, max_degree=4000):
    """Filter for a non-normal operator whose spectrum fills a strip around [0, hi] (open meshes: every
    boundary edge is one-way, graph.py:178, and L gets complex eigenvalues everywhere): the damped set is the
    ellipse with vertices cut and hi on the real axis and semi-minor axis `half_height`.  Same recurrence
    T_p((c - A)/e) with the foci c +- e pulled inwards; inside the ellipse |T_p| <= bulk, outside it grows.
    The recurrence is scaled by rho = (a + b)/e per step, so inside the ellipse |T_p / rho^p| <= ~0.5 whatever the
    degree.  Returns (c, e, p, rho)."""
    cut = min(max(cut, 1e-12), 0.5 * hi)
    c = 0.5 * (hi + cut)
    a = 0.5 * (hi - cut)
    b = min(half_height, 0.9 * a)
    e = math.sqrt(a * a - b * b)
    rho = (a + b) / e  # modulus of t + sqrt(t^2 - 1) on the ellipse
    t = (c - 0.5 * cut) / e
    growth_rate = math.log(t + math.sqrt(t * t - 1.0)) - math.log(rho)  # wanted (at cut/2) vs bulk, per degree
    p = int(math.ceil(strength / max(growth_rate, 1e-12)))
    p = max(min_degree, min(max_degree, p))
    return c, e, p, rho


def _ordered_schur(H, symmetric, n_real, n_extra=0, count_all=False):
    """Orthogonal U, (quasi-)triangular T with H = U T U^T, ordered so that the
    leading q columns span the dominant (largest-modulus) invariant subspace of
    the filtered operator that contains `n_real` real positive Ritz values — the
    images of the wanted low eigenvalues — plus `n_extra` more Ritz values.

    With a symmetric operator every Ritz value is real and q = n_real + n_extra.
    One-way mesh edges make L non-normal with a few genuinely complex eigenvalues
    (|Im| ~ 0.1) which the real Chebyshev polynomial amplifies even more than the
    wanted ones: they are dominant eigenvalues of B, must be carried (and
    converged) in the kept subspace, and are discarded after the final
    Rayleigh-Ritz step.  Returns (theta, U, T, q, n_real_found)."""
    m = H.shape[0]
    if symmetric:
        w, U = np.linalg.eigh(0.5 * (H + H.T))
        order = np.argsort(-w)
        q = min(n_real + n_extra, m)
        return w[order].astype(np.complex128), U[:, order], np.diag(w[order]), q, min(n_real, m)
    ev = np.linalg.eigvals(H)
    q, thr, n_found, _ = _select_dominant(ev, n_real, n_extra, count_all)
    T, U, sdim = sla.schur(H, output="real", sort=lambda r, i: math.hypot(r, i) > thr)
    theta = np.diag(T).astype(np.complex128)
    i = 0
    while i < m - 1:  # 2x2 blocks -> complex pair
        if T[i + 1, i] != 0.0:
            ev2 = np.linalg.eigvals(T[i:i + 2, i:i + 2])
            theta[i], theta[i + 1] = ev2[0], ev2[1]
            i += 2
        else:
            i += 1
    return theta, U, T, int(sdim), n_found


def _select_dominant(ev, n_real, n_extra, count_all):
    """How many of the largest-modulus Ritz values `ev` make up the wanted dominant set: enough to contain
    `n_real` images of low eigenvalues (+ `n_extra`), never splitting a conjugate pair.  Returns
    (q, modulus threshold between kept and dropped, number of wanted images found, order by modulus)."""
    m = len(ev)
    order = np.argsort(-np.abs(ev), kind="stable")
    ev = ev[order]
    if count_all:  # ellipse mode: every dominant Ritz value is an image of a low eigenvalue, real or complex
        is_real = np.ones(len(ev), dtype=bool)
    else:
        is_real = (np.abs(ev.imag) <= 1e-9 * np.abs(ev)) & (ev.real > 0)
    cnt = np.cumsum(is_real)
    hit = np.nonzero(cnt >= n_real)[0]
    q = int(hit[0]) + 1 if len(hit) else m
    extra = 0
    while q < m and extra < n_extra:
        # extend without splitting a conjugate pair
        step = 1 if is_real[q] or abs(ev[q].imag) <= 1e-9 * abs(ev[q]) else 2
        if q + step > m:
            break
        q += step
        extra += step
    if q < m:
        thr = 0.5 * (abs(ev[q - 1]) + abs(ev[q]))
        if abs(ev[q - 1]) == abs(ev[q]):  # conjugate pair straddling the cut
            q += 1
            thr = 0.5 * (abs(ev[q - 1]) + abs(ev[q])) if q < m else -1.0
    else:
        thr = -1.0
    return q, thr, int(min(cnt[-1], n_real)) if m else 0, order


RETRY_CODE = -5  # PF_E_PERSIST_TIMEOUT of the C-ABI: filter applications in flight were invalid, repeat the solve
MAX_RETRIES = 2


def _retryable(exc):
    return getattr(exc, "code", None) == RETRY_CODE


def _abandon(ops):
    """A solve was cut short: collect whatever orthogonalisation it left in flight."""
    fn = getattr(ops, "orth_abandon", None)
    if fn is not None:
        fn()


def drive(gen, ops):
    """Run a solver generator to completion, executing each filter request on `ops`.  `gen` is a generator, or a
    zero-argument callable that makes one: then a solve whose filter applications the device library declared invalid
    (error code RETRY_CODE: the library has drained its stream and switched to its other filter path by then) is
    repeated from the start."""
    make = gen if callable(gen) else None
    for attempt in range(MAX_RETRIES + 1):
        g = make() if make is not None else gen
        try:
            req = next(g)
            while True:
                if req[0] == "orth":
                    ops.orth_begin(*req[1:])
                elif req[0] == "orth+cheb":
                    ops.orth_begin(*req[1])
                    ops.cheb(*req[2])
                else:
                    ops.cheb(*req)
                req = g.send(None)
        except StopIteration as stop:
            return stop.value
        except Exception as exc:  # noqa: BLE001
            if make is None or not _retryable(exc) or attempt == MAX_RETRIES:
                raise
            g.close()
            _abandon(ops)


def drive_pair(gen_a, ops_a, gen_b, ops_b):
    """Run two solver generators in lockstep: while both have a filter application pending,
    the two Chebyshev recurrences advance in shared kernel launches (`ops.cheb2`); once one
    solver has finished the other continues alone.  Returns both results.  Generators or generator factories, as for
    `drive` (with factories BOTH solves are repeated after a RETRY_CODE error: they share their launches)."""
    makes = [gen_a if callable(gen_a) else None, gen_b if callable(gen_b) else None]
    can_retry = makes[0] is not None and makes[1] is not None
    for attempt in range(MAX_RETRIES + 1):
        gens = [makes[0]() if makes[0] is not None else gen_a, makes[1]() if makes[1] is not None else gen_b]
        try:
            return _drive_pair_once(gens, [ops_a, ops_b])
        except Exception as exc:  # noqa: BLE001
            if not can_retry or not _retryable(exc) or attempt == MAX_RETRIES:
                raise
            for g in gens:
                if g is not None:
                    g.close()
            _abandon(ops_a)
            _abandon(ops_b)


def _drive_pair_once(gens, ops):
    reqs, results = [None, None], [None, None]

    def advance(i, first=False):
        try:
            reqs[i] = next(gens[i]) if first else gens[i].send(None)
        except StopIteration as stop:
            reqs[i], results[i], gens[i] = None, stop.value, None

    advance(0, True)
    advance(1, True)
    def kind(i):
        return None if gens[i] is None else (reqs[i][0] if isinstance(reqs[i][0], str) else "cheb")

    pair_orth = hasattr(ops[0], "orth_begin2")
    fused = hasattr(ops[0], "orth_cheb2")
    while gens[0] is not None or gens[1] is not None:
        k0, k1 = kind(0), kind(1)
        if k0 == "orth+cheb" and k1 == "orth+cheb" and fused:
            # one outer step of both solvers in one library call: both Gram-Schmidt steps in shared launches and, right
            # behind them, the next filter application of both
            ops[0].orth_cheb2(reqs[0][1], reqs[0][2], ops[1], reqs[1][1], reqs[1][2])
            advance(0)
            advance(1)
        elif k0 in ("orth", "orth+cheb") or k1 in ("orth", "orth+cheb"):
            # Gram-Schmidt steps first (shared launches if both graphs have one); a fused request leaves its filter part
            both = k0 in ("orth", "orth+cheb") and k1 in ("orth", "orth+cheb")
            todo = [0, 1] if both else [0 if k0 in ("orth", "orth+cheb") else 1]
            parts = [reqs[i][1] if kind(i) == "orth+cheb" else reqs[i][1:] for i in todo]
            if both and pair_orth:
                ops[0].orth_begin2(parts[0], ops[1], parts[1])
            else:
                for i, part in zip(todo, parts):
                    ops[i].orth_begin(*part)
            for i in todo:
                if kind(i) == "orth+cheb":
                    reqs[i] = reqs[i][2]
                else:
                    advance(i)
        elif k0 == "cheb" and k1 == "cheb":
            ops[0].cheb2(reqs[0], ops[1], reqs[1])
            advance(0)
            advance(1)
        else:
            i = 0 if gens[0] is not None else 1
            ops[i].cheb(*reqs[i])
            advance(i)
    return results[0], results[1]


def filtered_eigs(ops, n_wanted, symmetric, **kw):
    """Smallest `n_wanted` non-null eigenpairs of the Laplacian held by `ops`
    (see `filtered_eigs_gen` for arguments and return value)."""
    return drive(lambda: filtered_eigs_gen(ops, n_wanted, symmetric, **kw), ops)


class _NeedEllipse(Exception):
    """The interval filter cannot handle this non-normal operator (complex wanted eigenvalues, or too many
    complex outliers to carry): switch to the ellipse filter."""


def filtered_eigs_gen(ops, n_wanted, symmetric, ellipse=None, nulls_fresh=False, **kw):
    """Solver generator (see `_solve_gen`).  For a non-symmetric operator the cheap strategy — interval filter,
    complex outliers carried as dominant Ritz values — is tried first unless `ellipse` is True (the caller knows
    the graph has many one-way edges); if it fails the ellipse filter takes over."""
    # (`nulls_fresh`: the caller has just written the null vectors into slots [0, null_slots) - the first attempt need not)
    if symmetric or ellipse is False:
        return (yield from _solve_gen(ops, n_wanted, symmetric, nulls_fresh=nulls_fresh, **kw))
    if ellipse is None:
        try:
            return (yield from _solve_gen(ops, n_wanted, symmetric, nulls_fresh=nulls_fresh, **kw))
        except _NeedEllipse:
            nulls_fresh = False
    half_height = 0.125 * kw.get("hi", 2.0)
    for _ in range(4):
        try:
            return (yield from _solve_gen(ops, n_wanted, symmetric, half_height=half_height, nulls_fresh=nulls_fresh, **kw))
        except _NeedEllipse:
            nulls_fresh = False
            half_height *= 1.6  # outliers above the assumed strip: make the ellipse taller
    raise RuntimeError("filtered Krylov-Schur: could not enclose the complex spectrum of this non-normal Laplacian in an "
                       "ellipse (one-way edges: an open or non-manifold mesh); the eigenpairs were NOT computed")


def _solve_gen(ops, n_wanted, symmetric, null_slots=0, cut=None, tol=1e-12, m_max=None,
               max_restarts=60, max_filter_resets=8, seed=0, strength=None, hi=2.0,
               nonsym_degree_cap=128, half_height=None, verbose=False, adapt_cut=False, nulls_fresh=False):
    """Generator form of the solver: yields `(src, dst, degree, c, e, rho)` whenever the Chebyshev filter has to be
    applied (the only expensive device operation), `("orth", w, first, count, normalize)` whenever a Gram-Schmidt
    step has to be started (`ops.orth_begin`), and `("orth+cheb", orth args, filter request)` for a Gram-Schmidt step with
    the next filter application right behind it; it receives nothing back.  `drive` / `drive_pair` execute the requests -
    the pair driver in launches that two graphs share.  Its return value is the solver result.

    Smallest `n_wanted` non-null eigenpairs of the Laplacian held by `ops`.

    `ops` must already hold `null_slots` orthonormal null vectors of the operator
    in workspace slots [0, null_slots) (one per non-trivial connected component).
    Returns (eig_vals ascending, first_slot, stats): eigenvalues > 1e-10 (the
    reference's null filter, graph.py:381), at least `n_wanted` of them unless the
    graph has fewer; the matching eigenvectors of the ITERATED operator (S if
    symmetric else L) sit in workspace slots [first_slot, first_slot+len(vals)).
    """
    n = ops.n
    n_active = n - ops.n_isolated
    stats = EigsStats()
    c0 = int(null_slots)
    if c0 > 0 and not nulls_fresh and ops.lock_null_vectors() != c0:  # (re)write slots [0, c0): a previous attempt's extraction reuses them
        raise RuntimeError("null_slots does not match the graph's component count")
    n_wanted = int(min(n_wanted, max(n_active - c0, 0)))
    if n_wanted <= 0:
        return np.zeros(0), 0, stats
    q_target = c0 + n_wanted  # real dominant Ritz values to converge (locked nulls included)
    if m_max is None:
        m_max = max(3 * q_target + 24, 48)
    m_max = int(min(m_max, n_active))
    reg = max(m_max + 1, 2 * q_target + 2)  # region A: Krylov basis + residual vector; region B: restart / Ritz products
    ops.ws_ensure(2 * reg)
    A0, B0 = 0, reg
    # Placement and strength of the filter.  Symmetric graphs (the resident kernel's case): a step of the recurrence
    # costs 1.5 us of a 250k pair against ~55 us for all that surrounds an application, which favours a somewhat
    # longer, weaker filter and fewer applications (tools/sweep_filter.py: 9.9 -> 9.2 ms per 250k pair).
    if strength is None:
        strength = 1.8 if symmetric else 2.0
    if cut is None:
        cut = (8.0 if symmetric else 12.0) * (n_wanted + 1) / max(n_active, 1)
    ellipse = half_height is not None and not symmetric
    degree_cap = 4000 if (symmetric or ellipse) else int(nonsym_degree_cap)
    plain = False  # no filter: B = (hi - A)/hi.  For tiny / dense-ish graphs whose wanted eigenvalues are not
    #                a small corner of [0, hi]; the Krylov space is then exhausted or restarted as usual.
    n_starts = [int(seed)]

    def start_vector(slot, nbasis):
        ops.start_vector(slot, n_starts[0])  # generated on the device, zero on isolated vertices
        n_starts[0] += 1
        _, nrm = ops.orth(slot, A0, nbasis)
        ops.scale(slot, 1.0 / nrm)

    def ritz(j, n_extra=0, full=True):
        if symmetric or full:
            theta, U, T, q, n_real = _ordered_schur(H[:j, :j], symmetric, q_target, n_extra, count_all=ellipse)
            res = np.abs(b[:j] @ U[:, :q])
        else:
            # convergence check only: eigenvector residuals |b^T s_i| from LAPACK's eig; the ordered real Schur
            # form (a Python callback per eigenvalue in scipy) is computed once, when it is needed
            ev, S = np.linalg.eig(H[:j, :j])
            q, _, n_real, order = _select_dominant(ev, q_target, n_extra, ellipse)
            theta, U, T = ev[order], None, None
            res = np.abs(b[:j] @ S[:, order[:q]])
        lead = theta[:q]
        if ellipse:
            theta_min = float(np.min(np.abs(lead))) if len(lead) else 0.0
        else:
            real_lead = lead[(np.abs(lead.imag) <= 1e-9 * np.abs(lead)) & (lead.real > 0)].real
            theta_min = float(np.min(real_lead)) if len(real_lead) else 0.0
            if not symmetric and q > q_target + 16:
                raise _NeedEllipse()  # too many complex outliers to carry along
        return theta, U, T, q, n_real, res, theta_min

    while True:
        if cut >= 0.5 * hi:
            plain = True
        rho = 1.0
        if plain:
            c, e, p = hi, hi, 1
        elif ellipse:
            c, e, p, rho = choose_filter_ellipse(cut, half_height, hi=hi, strength=strength, max_degree=degree_cap)
        else:
            c, e, p = choose_filter(cut, hi=hi, strength=strength, max_degree=degree_cap)
        stats.degree, stats.cut = p, cut
        if hasattr(ops, "orth_strict"):
            # an unfiltered iteration on a small graph runs until it nearly exhausts the space: the loose single-pass
            # criterion of the device's Gram-Schmidt step loses orthogonality there; on small graphs in general the
            # second pass costs nothing that matters
            ops.orth_strict(plain or n_active < 4096)
        theta0 = _cheb_value_scaled(0.0, c, e, p, rho)
        bulk = 0.5 * (1.0 + rho ** (-2.0 * p))  # bound of the scaled polynomial on the damped set (1 for the interval)
        band = -1.0 if plain else 1.5 * bulk  # wanted Ritz values must clear the damped set (none in plain mode)
        # Krylov-Schur state  B V_j = V_j H + v_j b^T ;  null vectors are locked exact Ritz pairs.
        j = c0
        H = np.zeros((m_max, m_max))
        H[:c0, :c0] = theta0 * np.eye(c0)
        b = np.zeros(m_max)
        start_vector(A0 + j, j)
        outcome = None  # "converged" | "cut" | "range"
        restarts = 0
        while outcome is None:
            spec = False  # the filter application of the current step was already queued speculatively
            near = False  # the last Ritz check was within a digit of the tolerance: the next step almost certainly converges
            next_check, seen = 0, None  # step of the next Ritz check; (step, residual / tolerance) of the last one
            while j < m_max and outcome is None:  # ---- expand
                if not spec:
                    yield (A0 + j, A0 + j + 1, p, c, e, rho)
                    stats.matvecs += p
                stats.outer_steps += 1
                # CGS2 + normalisation entirely on the device; the coefficients come back asynchronously
                # the Gram-Schmidt step (ops.orth_begin: CGS + normalisation on the device, results come back asynchronously)
                # and - to keep the device busy - the NEXT filter application, queued before this step's result is read
                spec = j + 1 < m_max and not near  # (a speculative application after the last step would be wasted)
                if spec:
                    yield ("orth+cheb", (A0 + j + 1, A0, j + 1, True), (A0 + j + 1, A0 + j + 2, p, c, e, rho))
                    stats.matvecs += p
                else:
                    yield ("orth", A0 + j + 1, A0, j + 1, True)
                h, beta = ops.orth_end()
                if getattr(ops, "orth_redone", False):
                    stats.second_passes += 1
                    spec = False  # w was refined after the speculative application had read it: apply the filter again
                if not (np.isfinite(beta) and np.all(np.isfinite(h))):
                    # eigenvalues outside the damped set grow like ratio^p: at high degree (small cut: k = 1 on a
                    # large open mesh) even a modest outlier overflows before any Ritz value could expose it
                    if not symmetric:
                        raise _NeedEllipse()
                    raise RuntimeError("filtered Krylov-Schur: the Chebyshev filter overflowed (degree %d): the operator "
                                       "has eigenvalues above the assumed bound %g" % (p, hi))
                H[:j + 1, j] = h
                H[j, :j] = b[:j]
                j += 1
                b[:] = 0.0
                b[j - 1] = beta
                exhausted = beta <= 1e-14 * max(abs(theta0), 1.0) or j >= n_active
                if exhausted or j == m_max or j >= max(q_target + 8, next_check):  # Ritz check: ~0.2 ms of host work
                    theta, U, T, q, n_real, res, theta_min = ritz(j, full=False)
                    # The check is host work per step (eigh of H: ~40 us for a symmetric graph, a general eig ~130 us), hidden
                    # behind the device's step only as long as the host keeps up: once two checks have shown the (roughly
                    # geometric) decay of the largest residual, half of the steps it still needs - at most 3 - are skipped
                    # before looking again (near convergence that is every step again)
                    worst = float(np.max(res)) / max(tol * max(theta_min, 1.0), 1e-300) if len(res) else 0.0
                    next_check = j + 1
                    if seen is not None and worst > 1.0 and seen[1] > worst and n_real >= q_target and theta_min > band:
                        per_step = np.log(seen[1] / worst) / (j - seen[0])
                        next_check = j + int(min(4, max(1, 0.5 * np.log(worst) / per_step)))
                    seen = (j, worst)
                    if verbose:
                        print("  j=%d q=%d theta_min=%.4g |theta|max=%.3g max res=%.3e" % (
                            j, q, theta_min, np.max(np.abs(theta)), np.max(res)))
                    # (within one digit: with the filter application at ~0.2 ms a stalled step costs about as much as a
                    # wasted application, so speculation is only given up when convergence is all but certain; measured at
                    # 250k: eigensolve 11.96 / 11.32 / 11.51 ms with 1e3 / 1e1 / never)
                    near = n_real >= q_target and theta_min > band and np.all(res <= 1e1 * tol * max(theta_min, 1.0))
                    if n_real >= q_target and np.all(res <= tol * max(theta_min, 1.0)) and theta_min > band:
                        outcome = "converged"
                    elif not symmetric and not ellipse and np.max(np.abs(theta)) > 1e7 * max(theta_min, 1.0) and p > 16:
                        outcome = "range"  # complex outliers eat the dynamic range: lower the degree
                    elif (j >= q + 12 or exhausted) and theta_min < band:
                        outcome = "cut"  # wanted eigenvalues sit inside the damped band
                    elif adapt_cut and j == m_max and not plain and not ellipse:
                        # General matrices (no a-priori scale for the low end, unlike mesh Laplacians): if the filter
                        # amplifies far more Ritz values than wanted, the wanted ones are crowded together near the top
                        # of the filter's range and Lanczos separates them slowly — narrow the undamped interval to
                        # just above the (q_target+1)-th lowest eigenvalue estimate.
                        amp = sorted(_cheb_inverse(t.real, c, e, p) for t in theta if abs(t.imag) <= 1e-9 * abs(t) and t.real > band)
                        if len(amp) >= 2 * q_target + 4 and amp[q_target] > 0 and 3.0 * amp[q_target] < cut:
                            shrink_to = 3.0 * amp[q_target]
                            outcome = "shrink"
                    elif exhausted:
                        outcome = "converged"
                    if outcome == "converged" and U is None:
                        theta, U, T, q, n_real, res, theta_min = ritz(j)  # the Schur vectors the extraction needs
            if outcome is not None:
                break
            if restarts >= max_restarts:
                if not symmetric and not ellipse:
                    raise _NeedEllipse()
                raise RuntimeError("filtered Krylov-Schur did not converge (max residual %.3e)" % np.max(res))
            # ---- thick restart: keep the dominant Schur vectors + a buffer
            theta, U, T, n_keep, _, _, _ = ritz(j, n_extra=max(4, q_target // 2))
            n_keep = min(n_keep, j - 1)
            if not symmetric and n_keep < j and T[n_keep, n_keep - 1] != 0.0:
                n_keep -= 1  # never split a 2x2 block
            ops.combine(A0, j, U[:, :n_keep], B0)
            ops.copy(A0 + j, B0 + n_keep, 1)  # the residual vector follows the kept block
            ops.copy(B0, A0, n_keep + 1)
            Hn = np.zeros((m_max, m_max))
            Hn[:n_keep, :n_keep] = T[:n_keep, :n_keep]
            bn = np.zeros(m_max)
            bn[:n_keep] = U[:, :n_keep].T @ b[:j]
            H, b, j = Hn, bn, n_keep
            restarts += 1
            stats.restarts += 1
        if outcome == "converged":
            break
        stats.filter_resets += 1
        if stats.filter_resets > max_filter_resets:
            if not symmetric and not ellipse:
                raise _NeedEllipse()
            raise RuntimeError("could not place the Chebyshev filter (cut %g, degree %d)" % (cut, p))
        if outcome == "range":
            degree_cap = max(16, p // 2)
        elif outcome == "shrink":
            cut = shrink_to
        else:
            lead = theta[:q]
            lam_est = [] if ellipse else sorted(_cheb_inverse(t.real, c, e, p) for t in lead
                                                if abs(t.imag) <= 1e-9 * abs(t) and t.real > 1.5)[c0:]
            if len(lam_est) >= 2:
                cut = max(4.0 * cut, 2.5 * lam_est[-1] * (n_wanted + 1) / len(lam_est))
            else:
                cut = 8.0 * cut
            cut = min(cut, hi)
        if verbose:
            print("  filter reset (%s): cut %.3e degree cap %d" % (outcome, cut, degree_cap))

    # ---- Rayleigh-Ritz on A itself over the converged orthonormal Schur vectors Z
    if 2 * q + 1 > reg:
        raise RuntimeError("workspace too small for Ritz extraction (q=%d, m_max=%d)" % (q, m_max))
    ops.combine(A0, j, U[:, :q], B0)  # Z -> region B
    HA = np.zeros((q, q))
    if hasattr(ops, "spmv_multi"):
        ops.spmv_multi(B0, A0, q)  # A Z -> region A (Krylov basis no longer needed), one library call
    else:
        for i in range(q):
            ops.spmv(B0 + i, A0 + i)
    stats.matvecs += q
    if hasattr(ops, "gram"):  # all q^2 inner products behind one synchronisation
        HA[:, :] = ops.gram(A0, q, B0, q).T
    else:
        for i in range(q):
            HA[:, i] = ops.dots(A0 + i, B0, q)
    if symmetric:
        lam, R = np.linalg.eigh(0.5 * (HA + HA.T))
    else:
        lam_c, R_c = np.linalg.eig(HA)
        order = np.argsort(lam_c.real, kind="stable")  # complex outliers carried by the interval filter have Re ~ 1
        take = q_target
        if take < q and abs(lam_c[order[take - 1]].imag) > 1e-9 and np.isclose(
                lam_c[order[take - 1]].real, lam_c[order[take]].real, rtol=1e-9, atol=0):
            take += 1  # never split a conjugate pair
        order = order[:take]
        sel = lam_c[order]
        is_cplx = np.abs(sel.imag) > 1e-9 * np.maximum(np.abs(sel), 1e-300)
        if np.any(is_cplx) and not ellipse:
            raise _NeedEllipse()  # open mesh: the low eigenvalues themselves are complex
        if ellipse and np.any(sel.real > cut):
            raise _NeedEllipse()  # junk from above the assumed strip crept into the dominant subspace
        lam = sel.real  # the reference keeps np.real(eig_vals): a conjugate pair shows up as a repeated value
        R = np.empty((q, take))
        for col, i in enumerate(order):
            y = R_c[:, i]
            if is_cplx[col]:
                # reference: np.real(eig_vecs) of BOTH conjugate vectors, i.e. the same real part twice, with
                # ARPACK's run-dependent phase.  Fix the phase (largest component real positive) instead.
                piv = np.argmax(np.abs(y))
                y = y * (np.conj(y[piv]) / abs(y[piv]))
            R[:, col] = np.real(y)
        R /= np.linalg.norm(R, axis=0, keepdims=True)
    keep = np.where(lam > MIN_EIG_VAL)[0]
    stats.n_null = int(min(q_target, len(lam)) - len(keep)) if not symmetric else int(np.sum(lam <= MIN_EIG_VAL))
    lam = lam[keep]
    R = R[:, keep]
    nk = len(keep)
    if nk == 0:
        # The dominant subspace of the filtered operator held nothing but (numerically) null directions.  Seen on
        # strongly non-normal Laplacians (a large hole: hundreds of one-way boundary edges next to >1000 stranded
        # vertices), where Ritz values of the filter keep growing instead of converging.  A taller ellipse may
        # still enclose the spectrum; if not, say so instead of handing an empty block to the device.
        if not symmetric:
            raise _NeedEllipse()
        raise RuntimeError("filtered Krylov-Schur converged onto the null space only (no eigenvalue > 1e-10 among "
                           "%d Ritz values)" % q)
    X0, AX0 = B0 + q, A0 + q
    if X0 + nk > 2 * reg or AX0 + nk > reg:
        raise RuntimeError("workspace too small for Ritz extraction")
    ops.combine(B0, q, R, X0)  # X = Z R
    ops.combine(A0, q, R, AX0)  # A X = (A Z) R
    # (for a complex pair the real part alone is not an eigenvector: its "residual" is |Im lambda| * |Im x|)
    if hasattr(ops, "resnorms"):
        stats.residuals = np.asarray(ops.resnorms(AX0, X0, lam[:nk]))
    else:
        stats.residuals = np.array([ops.resnorm(AX0 + i, X0 + i, lam[i]) for i in range(nk)])
    ops.sync()  # nothing of this solve is left in flight (a speculative filter application that failed surfaces here)
    return lam, X0, stats
