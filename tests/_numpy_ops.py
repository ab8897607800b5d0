"""TEST DOUBLE (CPU) for the device-ops object the Krylov driver talks to.

Implements the same method surface as `pyfocusr_amd._hip.DeviceLaplacian` with
numpy/scipy so that the host logic in `pyfocusr_amd/_krylov.py` can be exercised
by `-m "not gpu"` tests in a container without a GPU.  Lives under `tests/` and is
never imported by the product package (the product fails loudly without the HIP
library)."""
import numpy as np
from scipy import sparse
from scipy.sparse.csgraph import connected_components


class NumpyOps(object):
    def __init__(self, W):
        """W: scipy CSR weighted adjacency (directed set semantics)."""
        n = W.shape[0]
        self.n = n
        self.W = W.tocsr()
        deg = np.asarray(self.W @ np.ones((n, 1)))[:, 0]
        self.deg = deg
        g = 1.0 / (deg + 1e-8)
        self.g = g
        self.isolated = deg == 0
        self.n_isolated = int(self.isolated.sum())
        self.symmetric = abs(self.W - self.W.T).nnz == 0
        K = sparse.diags(deg) - self.W
        self.L = (sparse.diags(g) @ K).tocsr()
        s = np.sqrt(g)
        self.s = s
        self.S = (sparse.diags(s) @ K @ sparse.diags(s)).tocsr() if self.symmetric else None
        self.A = self.S if self.symmetric else self.L
        self.ws = np.zeros((n, 0), order="F")
        ncomp, labels = connected_components(self.W + self.W.T, directed=False)
        self.labels = labels
        self.launches = 0

    # -- workspace
    def ws_ensure(self, nslots):
        if self.ws.shape[1] < nslots:
            new = np.zeros((self.n, nslots), order="F")
            new[:, : self.ws.shape[1]] = self.ws
            self.ws = new

    def upload(self, slot, x):
        self.ws[:, slot] = x

    def download_slots(self, first, count):
        return np.array(self.ws[:, first : first + count])

    def copy(self, src, dst, count):
        self.ws[:, dst : dst + count] = self.ws[:, src : src + count].copy()

    def start_vector(self, slot, seed):
        x = np.random.default_rng(seed).standard_normal(self.n)
        x[self.isolated] = 0.0
        self.ws[:, slot] = x

    def mask_isolated(self, slot):
        self.ws[self.isolated, slot] = 0.0

    # -- null vectors of the iterated operator, one per non-trivial component
    def lock_null_vectors(self):
        comps = [c for c in np.unique(self.labels) if np.sum(self.labels == c) > 1]
        self.ws_ensure(len(comps) + 1)
        for i, c in enumerate(comps):
            v = (self.labels == c).astype(np.float64)
            if self.symmetric:
                v = v * np.sqrt(self.deg + 1e-8)
            self.ws[:, i] = v / np.linalg.norm(v)
        return len(comps)

    # -- operator applications
    def spmv(self, src, dst):
        self.ws[:, dst] = self.A @ self.ws[:, src]
        self.launches += 1

    def cheb(self, src, dst, p, c, e, rho=1.0):
        A = self.A
        y0 = self.ws[:, src]
        y1 = (c * y0 - A @ y0) / (e * rho)
        for _ in range(p - 1):
            y0, y1 = y1, (2.0 / (e * rho)) * (c * y1 - A @ y1) - y0 / (rho * rho)
        self.ws[:, dst] = y1
        self.launches += p

    def cheb2(self, req_self, other, req_other):
        self.cheb(*req_self)
        other.cheb(*req_other)

    # -- vector kernels
    def dots(self, w, first, count):
        return self.ws[:, first : first + count].T @ self.ws[:, w]

    def orth(self, w, first, count):
        V = self.ws[:, first : first + count]
        x = self.ws[:, w]
        h1 = V.T @ x
        x = x - V @ h1
        h2 = V.T @ x
        x = x - V @ h2
        self.ws[:, w] = x
        return h1 + h2, float(np.linalg.norm(x))

    def orth_begin(self, w, first, count, normalize=True):
        h, nrm = self.orth(w, first, count)
        if normalize and nrm > 1e-140:
            self.ws[:, w] /= nrm
        self._orth_result = (h, nrm)

    def orth_end(self):
        return self._orth_result

    def scale(self, slot, alpha):
        self.ws[:, slot] *= alpha

    def combine(self, src_first, m, Y, dst_first):
        Y = np.asarray(Y, dtype=np.float64)
        self.ws[:, dst_first : dst_first + Y.shape[1]] = self.ws[:, src_first : src_first + m] @ Y

    def sync(self):
        pass

    def resnorm(self, ax, x, lam):
        return float(np.linalg.norm(self.ws[:, ax] - lam * self.ws[:, x]))

    # -- primitives the row-partitioned solver adds (pyfocusr_amd/rowpart.py)
    def op_step(self, x, prev, out, alpha, c, beta):
        """out = alpha (c x - A x) - beta prev; `out` may be the `prev` slot."""
        y = alpha * (c * self.ws[:, x] - self.A @ self.ws[:, x])
        if prev is not None:
            y = y - beta * self.ws[:, prev]
        self.ws[:, out] = y
        self.launches += 1

    def cheb_steps(self, prev, cur, k_first, n_steps, c, e, rho=1.0):
        a, b = prev, cur
        for k in range(k_first, k_first + n_steps):
            if k == 1:
                self.op_step(b, None, a, 1.0 / (e * rho), c, 0.0)
            else:
                self.op_step(b, a, a, 2.0 / (e * rho), c, 1.0 / (rho * rho))
            a, b = b, a
        return a, b

    def axpy(self, w, first, count, coef):
        self.ws[:, w] += self.ws[:, first : first + count] @ np.asarray(coef, dtype=np.float64)

    def rows_create(self, idx):
        return np.asarray(idx, dtype=np.int64).copy()

    def rows_gather(self, slot, rows):
        return self.ws[rows, slot].copy()

    def rows_scatter(self, slot, rows, values):
        self.ws[rows, slot] = values

    def rows_fill(self, slot, rows, value):
        self.ws[rows, slot] = value


class MatrixOps(NumpyOps):
    """Test double over a general symmetric matrix (the local operator of one rank of the row-partitioned solve)."""

    def __init__(self, A):
        A = A.tocsr()
        self.n = A.shape[0]
        self.A = self.S = A
        self.symmetric = True
        self.isolated = np.zeros(self.n, dtype=bool)
        self.n_isolated = 0
        self.ws = np.zeros((self.n, 0), order="F")
        self.launches = 0

    def lock_null_vectors(self):
        return 0


class PairedNumpyOps(NumpyOps):
    """`NumpyOps` with the pair entries of the device class (`orth_begin2`, `orth_cheb2`), so that the fused branches of
    `_krylov.drive_pair` run in the CPU tests too; counts how often each was used."""

    def __init__(self, W):
        super().__init__(W)
        self.pair_calls = {"orth_begin2": 0, "orth_cheb2": 0}

    def orth_begin2(self, req, other, req_other):
        self.pair_calls["orth_begin2"] += 1
        self.orth_begin(*req)
        other.orth_begin(*req_other)

    def orth_cheb2(self, orth, req, other, orth_other, req_other):
        self.pair_calls["orth_cheb2"] += 1
        self.orth_begin(*orth)
        other.orth_begin(*orth_other)
        self.cheb2(req, other, req_other)
