"""ORACLE — test infrastructure, NOT product code.  **Parity unpinned.**

CPU restatement of the Coherent Point Drift registration the reference runs on the spectral
coordinates between eigsort and the KNN correspondence
(`/root/reference/pyfocusr/focusr.py:297-334`: `cycpd.affine_registration` then
`cycpd.deformable_registration(num_eig=…, alpha=…, beta=…)`, fitted on random subsets of
`n_coords_spectral_registration` points and applied to all target points with
`transform_point_cloud`).

`cycpd` is a third-party package (not in `/root/reference`, not pinned in `requirements.txt:1-8`,
absent from the build image) and the reference's tests hold no vector for it, so this file
restates the *published* algorithm — Myronenko & Song, "Point Set Registration: Coherent Point
Drift", IEEE TPAMI 32(12), 2010: EM with a Gaussian mixture centred on the moving points; affine
M-step of its Fig. 3; non-rigid M-step of Fig. 4 with the low-rank approximation
G ~ Q S Q^T of its Section 6 (Woodbury solve) — with the conventions of the open-source pycpd code
base that cycpd derives from: sigma^2 initialised to the mean squared distance / D, outlier weight
w = 0, `tolerance` compared with the change of the objective (affine) / of sigma^2 (deformable),
sigma^2 <- tolerance/10 when the update is not positive.  The HIP path is tested against THIS
restatement; agreement with cycpd itself cannot be checked here.
"""
import numpy as np


def initialize_sigma2(X, Y):
    diff = X[None, :, :] - Y[:, None, :]
    return float(np.sum(diff ** 2) / (X.shape[1] * X.shape[0] * Y.shape[0]))


def gaussian_kernel(A, beta, B=None):
    B = A if B is None else B
    d2 = np.sum((A[:, None, :] - B[None, :, :]) ** 2, axis=2)
    return np.exp(-d2 / (2 * beta ** 2))


def expectation(X, TY, sigma2, w=0.0):
    """E-step: returns P1 (M,), Pt1 (N,), PX (M,D), Np."""
    M, D = TY.shape
    N = X.shape[0]
    P = np.exp(-np.sum((X[None, :, :] - TY[:, None, :]) ** 2, axis=2) / (2 * sigma2))  # (M, N)
    c = (2 * np.pi * sigma2) ** (D / 2) * w / (1 - w) * M / N
    den = np.sum(P, axis=0)
    den[den == 0] = np.finfo(float).eps
    den += c
    P = P / den[None, :]
    Pt1, P1 = P.sum(axis=0), P.sum(axis=1)
    return P1, Pt1, P @ X, float(P1.sum())


class _EM(object):
    def __init__(self, X, Y, sigma2=None, max_iterations=100, tolerance=0.001, w=0.0):
        self.X, self.Y, self.TY = np.asarray(X, float), np.asarray(Y, float), np.array(Y, float)
        (self.N, self.D), self.M = self.X.shape, self.Y.shape[0]
        self.sigma2 = initialize_sigma2(self.X, self.Y) if sigma2 is None else sigma2
        self.max_iterations, self.tolerance, self.w = max_iterations, tolerance, w
        self.iteration, self.diff, self.q = 0, np.inf, np.inf

    def register(self):
        self.transform_point_cloud()
        while self.iteration < self.max_iterations and self.diff > self.tolerance:
            self.P1, self.Pt1, self.PX, self.Np = expectation(self.X, self.TY, self.sigma2, self.w)
            self.update_transform()
            self.transform_point_cloud()
            self.update_variance()
            self.iteration += 1
        return self.TY, self.get_registration_parameters()


class AffineRegistration(_EM):
    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.B, self.t = np.eye(self.D), np.zeros(self.D)

    def update_transform(self):
        muX = self.PX.sum(axis=0) / self.Np
        muY = (self.P1[:, None] * self.Y).sum(axis=0) / self.Np
        self.X_hat = self.X - muX
        Y_hat = self.Y - muY
        # X_hat^T P^T Y_hat without P: P X_hat = PX - P1 muX^T
        self.A = (self.PX - self.P1[:, None] * muX[None, :]).T @ Y_hat
        self.YPY = Y_hat.T @ (self.P1[:, None] * Y_hat)
        self.B = np.linalg.solve(self.YPY.T, self.A.T)
        self.t = muX - self.B.T @ muY

    def transform_point_cloud(self, Y=None):
        if Y is None:
            self.TY = self.Y @ self.B + self.t
            return None
        return Y @ self.B + self.t

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


def low_rank_eigen(G, num_eig):
    S, Q = np.linalg.eigh(G)
    idx = np.argsort(np.abs(S))[::-1][:num_eig]
    return Q[:, idx], S[idx]


class DeformableRegistration(_EM):
    def __init__(self, *args, alpha=2.0, beta=2.0, low_rank=True, num_eig=100, eig_floor=1e-15, **kw):
        super().__init__(*args, **kw)
        self.alpha, self.beta, self.low_rank = alpha, beta, low_rank
        self.W = np.zeros((self.M, self.D))
        self.G = gaussian_kernel(self.Y, beta)
        if low_rank:
            self.Q, self.S = low_rank_eigen(self.G, min(num_eig, self.M))
            self.S = np.maximum(self.S, eig_floor * np.max(self.S))  # see pyfocusr_amd/cpd.py: keeps the K x K system SPD

    def update_transform(self):
        F = self.PX - self.P1[:, None] * self.Y
        if not self.low_rank:
            A = self.P1[:, None] * self.G + self.alpha * self.sigma2 * np.eye(self.M)
            self.W = np.linalg.solve(A, F)
        else:
            dPQ = self.P1[:, None] * self.Q
            lam = self.alpha * self.sigma2
            Z = np.linalg.solve(lam * np.diag(1.0 / self.S) + self.Q.T @ dPQ, self.Q.T @ F)
            self.W = (F - dPQ @ Z) / lam

    def transform_point_cloud(self, Y=None):
        if Y is not None:
            return Y + gaussian_kernel(Y, self.beta, self.Y) @ self.W
        if self.low_rank:
            self.TY = self.Y + self.Q @ (self.S[:, None] * (self.Q.T @ self.W))
        else:
            self.TY = self.Y + self.G @ self.W
        return None

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
        return self.G, self.W
