"""Coherent Point Drift registration of spectral coordinates without cycpd.

The reference fits `cycpd.affine_registration` and then `cycpd.deformable_registration` on
random subsets of the two spectral embeddings and applies the result to every target point
(`/root/reference/pyfocusr/focusr.py:297-334`).  cycpd is a third-party Cython package that is
not part of the reference tree; this module provides the slice of its interface the reference
uses — constructor keywords `X, Y, max_iterations, tolerance` (+ `num_eig, alpha, beta` for
the deformable model), `register() -> (TY, parameters)`, `transform_point_cloud(Y)` — running
the published algorithm (Myronenko & Song, TPAMI 2010; conventions of the pycpd code base cycpd
derives from, see the docstrings) with its O(M*N) work on the MI355X:

* E-step: `pf_cpd_estep` — P is never formed; column sums, row sums and P@X in two passes.
* Low-rank model of the Gaussian affinity G (M x M, `num_eig` leading eigenpairs): randomised
  subspace iteration whose products G@V come from `pf_cpd_gram` (G is never formed either);
  host work is a QR of an M x (num_eig+oversampling) block.
* `transform_point_cloud` of all n points of the mesh: `pf_cpd_gram` again (n x M affinity).

The M-steps are O(M d^2) (affine) and O(M K d + K^3) (deformable, Woodbury form) numpy on the
host.  Parity with cycpd itself is unpinned (absent from the build image); `tests/test_cpd.py`
checks this module against a dense CPU restatement of the same algorithm.
"""
import numpy as np
from scipy.fft import idct

from . import _hip


def initialize_sigma2(X, Y):
    """Mean squared distance between the two sets / d, without the (M, N, d) temporary:
    sum_mn |x_n - y_m|^2 = M sum|x|^2 + N sum|y|^2 - 2 (sum x).(sum y)."""
    (N, D), M = X.shape, Y.shape[0]
    xc, yc = X - X.mean(axis=0), Y - X.mean(axis=0)  # centre first: keeps the cancellation harmless
    total = M * np.sum(xc * xc) + N * np.sum(yc * yc) - 2.0 * np.dot(xc.sum(axis=0), yc.sum(axis=0))
    return float(total / (D * M * N))


class _ExpectationMaximisation(object):
    def __init__(self, X, Y, sigma2=None, max_iterations=None, tolerance=None, w=None, ctx=None, verbose=False, **_ignored):
        X = np.ascontiguousarray(X, dtype=np.float64)
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        if X.ndim != 2 or Y.ndim != 2:
            raise ValueError("The target and source point clouds must be 2D numpy arrays.")
        if X.shape[1] != Y.shape[1]:
            raise ValueError("Both point clouds need to have the same number of dimensions.")
        if sigma2 is not None and sigma2 <= 0:
            raise ValueError("Expected a positive value for sigma2 instead got: {}".format(sigma2))
        if w is not None and not (0 <= w < 1):
            raise ValueError("Expected a value between 0 (inclusive) and 1 (exclusive) for w instead got: {}".format(w))
        self.X, self.Y, self.TY = X, Y, Y.copy()
        (self.N, self.D), self.M = X.shape, Y.shape[0]
        self.sigma2 = initialize_sigma2(X, Y) if sigma2 is None else float(sigma2)
        self.tolerance = 0.001 if tolerance is None else tolerance
        self.w = 0.0 if w is None else w
        self.max_iterations = 100 if max_iterations is None else int(max_iterations)
        self.iteration, self.diff, self.q = 0, np.inf, np.inf
        self.verbose = verbose
        self._ctx = ctx
        self.P1 = self.Pt1 = self.PX = None
        self.Np = 0.0

    def register(self, callback=lambda **kwargs: None):
        dev = self._dev = _hip.DeviceCpd(self.X, self.Y, ctx=self._ctx)
        try:
            self._on_device(dev)
            self.transform_point_cloud()
            while self.iteration < self.max_iterations and self.diff > self.tolerance:
                self.P1, self.Pt1, self.PX = dev.estep(self.TY, self.sigma2, self.w)
                self.Np = float(self.P1.sum())
                self.update_transform()
                self.transform_point_cloud()
                self.update_variance()
                self.iteration += 1
                if callable(callback):
                    callback(iteration=self.iteration, error=self.q, X=self.X, Y=self.TY)
                if self.verbose:
                    print("CPD iteration %d: sigma2 %.3e, change %.3e" % (self.iteration, self.sigma2, self.diff))
        finally:
            self._dev = None
            dev.close()
        return self.TY, self.get_registration_parameters()

    def _on_device(self, dev):
        pass


class affine_registration(_ExpectationMaximisation):
    """TY = Y B + t  (M-step of Fig. 3 of the CPD paper)."""

    def __init__(self, B=None, t=None, *args, **kwargs):
        super(affine_registration, self).__init__(*args, **kwargs)
        self.B = np.eye(self.D) if B is None else np.asarray(B, dtype=np.float64)
        self.t = np.zeros(self.D) if t is None else np.asarray(t, dtype=np.float64).reshape(self.D)

    def update_transform(self):
        muX = self.PX.sum(axis=0) / self.Np
        muY = self.P1 @ self.Y / self.Np
        self.X_hat = self.X - muX
        Y_hat = self.Y - muY
        self.A = (self.PX - self.P1[:, None] * muX[None, :]).T @ Y_hat  # X_hat^T P^T Y_hat
        self.YPY = Y_hat.T @ (self.P1[:, None] * Y_hat)
        self.B = np.linalg.solve(self.YPY.T, self.A.T)
        self.t = muX - self.B.T @ muY

    def transform_point_cloud(self, Y=None):
        if Y is None:
            self.TY = self.Y @ self.B + self.t
            return None
        return np.asarray(Y, dtype=np.float64) @ self.B + self.t

    def update_variance(self):
        qprev = self.q
        trAB = np.trace(self.A @ self.B)
        xPx = self.Pt1 @ np.sum(self.X_hat * self.X_hat, axis=1)
        trBYPYP = np.trace(self.B @ self.YPY @ self.B)
        self.q = (xPx - 2 * trAB + trBYPYP) / (2 * self.sigma2) + self.D * self.Np / 2 * np.log(self.sigma2)
        self.diff = abs(self.q - qprev)
        self.sigma2 = (xPx - trAB) / (self.Np * self.D)
        if self.sigma2 <= 0:
            self.sigma2 = self.tolerance / 10

    def get_registration_parameters(self):
        return self.B, self.t


EIG_FLOOR = 1e-14  # eigenvalues of G below EIG_FLOOR * largest are rounding noise (G is positive semi-definite)


def low_rank_affinity(Y, beta, num_eig, ctx=None, oversample=28, max_iterations=30, rtol=1e-13, seed=0):
    """(Q (M,K'), S (K',)) with G(Y,Y) ~ Q diag(S) Q^T: the leading eigenpairs of the Gaussian affinity, at most
    `num_eig` of them and only those above EIG_FLOOR * the largest.  Every product G @ V runs on the device.

    A Rayleigh-Ritz step on a random (num_eig + oversample)-dimensional block first reveals the numerical rank.
    * Wide kernel (the reference's default beta = 3 on unit-sized coordinates): the spectrum decays
      geometrically and only ~15 eigenvalues exceed 1e-14 of the largest; the rest of the requested 100 are
      numerically G's null space — they carry no displacement (their 1/S term in the Woodbury system pins
      their coefficients to zero), so they are dropped, and subspace iteration in the span of the leading Ritz
      vectors converges in two or three more products.
    * Narrow kernel (rank >= num_eig): implicitly restarted Lanczos (ARPACK `eigsh`, machine-precision
      tolerance) with the device product as its operator."""
    M = Y.shape[0]
    K = int(min(num_eig, M))
    p = int(min(M, K + oversample))
    rng = np.random.default_rng(seed)
    # random orthonormal block without a QR: p columns of the orthonormal DCT matrix with random row signs
    E = np.zeros((M, p))
    E[rng.choice(M, p, replace=False), np.arange(p)] = 1.0
    V = idct(E, axis=0, norm="ortho") * rng.choice([-1.0, 1.0], size=M)[:, None]
    prev = None
    for it in range(max_iterations):
        GV = _hip.gaussian_gram_product(Y, Y, beta, V, ctx=ctx)
        H = V.T @ GV
        s, U = np.linalg.eigh((H + H.T) / 2)
        idx = np.argsort(np.abs(s))[::-1]
        rank = int(max(1, np.count_nonzero(s[idx] > EIG_FLOOR * abs(s[idx[0]]))))
        if V.shape[1] == M:
            keep = min(K, rank)
            return V @ U[:, idx[:keep]], s[idx[:keep]]
        if it == 0 and rank >= K and rank >= V.shape[1] - 2:
            break  # not rank-deficient within the block: Lanczos below
        keep = min(K, rank)
        top = s[idx[:keep]]
        m = keep if prev is None else min(keep, len(prev))  # a value sitting on the floor may come and go
        if prev is not None and np.max(np.abs(top[:m] - prev[:m])) <= rtol * abs(top[0]):
            return V @ U[:, idx[:keep]], top
        prev = top
        p_next = int(min(V.shape[1], rank + 12))
        if p_next < V.shape[1]:  # continue in the span of the leading Ritz vectors
            GV = GV @ U[:, idx[:p_next]]
        V = np.linalg.qr(GV)[0]
    from scipy.sparse.linalg import LinearOperator, eigsh

    def product(v):
        v = np.asarray(v, dtype=np.float64)
        return _hip.gaussian_gram_product(Y, Y, beta, v.reshape(M, -1), ctx=ctx).reshape(v.shape)

    op = LinearOperator((M, M), matvec=product, matmat=product, dtype=np.float64)
    s, Q = eigsh(op, k=min(K, M - 1), which="LA", v0=rng.standard_normal(M), ncv=min(M, max(2 * K + 1, K + 40)))
    order = np.argsort(s)[::-1]
    s, Q = s[order], Q[:, order]
    keep = int(max(1, np.count_nonzero(s > EIG_FLOOR * s[0])))
    return Q[:, :keep], s[:keep]


class deformable_registration(_ExpectationMaximisation):
    """TY = Y + G W with G the Gaussian affinity of Y (width beta) in the low-rank form G ~ Q S Q^T
    (`num_eig` eigenpairs; Section 6 of the CPD paper), regularisation weight alpha."""

    def __init__(self, alpha=None, beta=None, num_eig=100, low_rank=True, *args, **kwargs):
        super(deformable_registration, self).__init__(*args, **kwargs)
        if alpha is not None and alpha <= 0:
            raise ValueError("Expected a positive value for regularization parameter alpha. Instead got: {}".format(alpha))
        if beta is not None and beta <= 0:
            raise ValueError("Expected a positive value for the width of the coherent Gaussian kernel. Instead got: {}".format(beta))
        self.alpha = 2.0 if alpha is None else float(alpha)
        self.beta = 2.0 if beta is None else float(beta)
        self.num_eig = int(num_eig)
        self.W = np.zeros((self.M, self.D))
        self.Q, self.S = low_rank_affinity(self.Y, self.beta, self.num_eig, ctx=self._ctx)
        self.inv_S = 1.0 / self.S

    def _on_device(self, dev):
        dev.set_basis(self.Q)

    def update_transform(self):
        # Woodbury form of (diag(P1) G + alpha sigma2 I) W = PX - diag(P1) Y with G = Q S Q^T
        F = self.PX - self.P1[:, None] * self.Y
        dPQ = self.P1[:, None] * self.Q
        lam = self.alpha * self.sigma2
        QtPQ = self._dev.weighted_gram() if getattr(self, "_dev", None) is not None else self.Q.T @ dPQ
        Z = np.linalg.solve(lam * np.diag(self.inv_S) + QtPQ, self.Q.T @ F)
        self.W = (F - dPQ @ Z) / lam

    def transform_point_cloud(self, Y=None):
        if Y is None:
            self.TY = self.Y + self.Q @ (self.S[:, None] * (self.Q.T @ self.W))
            return None
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        return Y + _hip.gaussian_gram_product(Y, self.Y, self.beta, self.W, ctx=self._ctx)

    def update_variance(self):
        qprev = self.sigma2
        xPx = self.Pt1 @ np.sum(self.X * self.X, axis=1)
        yPy = self.P1 @ np.sum(self.TY * self.TY, axis=1)
        trPXY = np.sum(self.TY * self.PX)
        self.sigma2 = (xPx - 2 * trPXY + yPy) / (self.Np * self.D)
        if self.sigma2 <= 0:
            self.sigma2 = self.tolerance / 10
        self.diff = abs(self.sigma2 - qprev)

    def get_registration_parameters(self):
        return self.Q, self.S, self.W
