"""`eigsort` — order and sign-flip the eigenmaps of two graphs so they correspond.

Drop-in mirror of `/root/reference/pyfocusr/eigsort.py` (`class eigsort` :9-249):
same constructor, public cost matrices (`c_lambda`, `c_hist`, `c_hist_f`,
`c_spatial`, `c_spatial_f`, `Q`) and in-place mutation of the non-reference
graph's `eig_vecs` (its `eig_vals` are never permuted).

Device work (`pf_eigsort_costs`, the default whenever both graphs still hold the eigenvector block their eigensolve
left in HBM): the four sample-based cost matrices - `c_hist` / `c_hist_f` (eigsort.py:162-189: 2k^2 one-dimensional
Wasserstein distances of log-transformed samples, as LDS sorts + order-statistic differences) and `c_spatial` /
`c_spatial_f` (eigsort.py:191-233: differences at spatially nearest sampled vertices, a 3-D 1-NN) - from the sampled
rows read where they are; the sign flips and column moves of `eigen_sort` (eigsort.py:108-122) are applied to the
device block's column map and to its pinned host image by one kernel + one DMA (`Graph._remap_host_image`).
Host work: `c_lambda` (k x k), the Hungarian assignment on a k x k matrix, and - when samples were assigned from
outside, a graph has no device block, or the device refuses (> 16384 samples) - the same four matrices in numpy
(sorted columns + one cached breakpoint plan for all W1 distances), which the reference-generated goldens pin to 1e-12.
"""
import functools

import numpy as np
from scipy.optimize import linear_sum_assignment

from . import _hip
from .main import print_header


@functools.lru_cache(maxsize=8)
def _w1_quantile_plan(n, m):
    """W1 between two empirical distributions = integral over t in (0, 1) of |F_u^-1(t) - F_v^-1(t)| (the area between
    the CDFs, which scipy integrates along x, equals the area between the quantile functions).  Both quantile functions
    are step functions - u_(i) on (i/n, (i+1)/n], v_(j) on (j/m, (j+1)/m] - so the integrand is constant between
    consecutive breakpoints of {i/n} U {j/m}.  Those breakpoints depend on (n, m) only: one plan serves all 2 k^2 pairs
    of `calc_c_hist`.  Breakpoints are merged on the integer grid of n*m (exact).  Returns (interval lengths, index of
    the active u order statistic, index of the active v order statistic); for n == m this is 1/n, arange, arange."""
    pts = np.union1d(np.arange(n + 1, dtype=np.int64) * m, np.arange(m + 1, dtype=np.int64) * n)
    left = pts[:-1]
    return np.diff(pts) / float(n * m), left // m, left // n


class eigsort(object):
    def __init__(self, graph_target, graph_source, n_features, target_as_reference=True):
        self.graph_target = graph_target
        self.graph_source = graph_source
        self.n_features = n_features
        self.target_as_reference = target_as_reference

        # eigsort.py:34-41: the sampled points and eigenvector rows.  Gathered on first use (properties below): when both
        # graphs still hold their eigenvectors and points on one device, the cost matrices are computed there
        # (`_device_costs`) and the samples never have to come to the host.
        self._samples = {}
        self._device_result = None

        self.c_lambda = np.zeros((self.n_features, self.n_features))
        self.c_hist = np.zeros_like(self.c_lambda)
        self.c_hist_f = np.zeros_like(self.c_lambda)
        self.c_spatial = np.zeros_like(self.c_lambda)
        self.c_spatial_f = np.zeros_like(self.c_lambda)

        self.Q = None
        self.target_matches = None
        self.source_matches = None
        self.flipped_pairs = None
        self.idx_source_for_each_target_pt = None
        self.verbose = getattr(graph_target, "verbose", True)

    def _sample(self, name, make):
        if name not in self._samples:
            self._samples[name] = make()
        return self._samples[name]

    rand_target_points = property(lambda self: self._sample("tp", self.graph_target.get_rand_normalized_points),
                                  lambda self, v: self._samples.__setitem__("tp", v))
    rand_source_points = property(lambda self: self._sample("sp", self.graph_source.get_rand_normalized_points),
                                  lambda self, v: self._samples.__setitem__("sp", v))
    rand_target_eig_vecs = property(lambda self: self._sample("tv", self.graph_target.get_rand_eig_vecs),
                                    lambda self, v: self._samples.__setitem__("tv", v))
    rand_source_eig_vecs = property(lambda self: self._sample("sv", self.graph_source.get_rand_eig_vecs),
                                    lambda self, v: self._samples.__setitem__("sv", v))

    def _device_costs(self):
        """c_hist, c_hist_f, c_spatial, c_spatial_f and the spatial 1-NN indices from `pf_eigsort_costs`, or None when the
        host path has to run: samples already in hand or assigned from outside, a graph without its device-resident
        block / points / an up-to-date column map, different contexts, samples of more than 16384 rows."""
        if self._device_result is not None:
            return self._device_result
        if self._samples:
            return None
        devs = []
        for g in (self.graph_target, self.graph_source):
            fm, dev = getattr(g, "_final_map", None), getattr(g, "_device", None)
            if (fm is None or dev is None or not getattr(dev, "_h", None) or not getattr(dev, "has_points", False)
                    or self.n_features > len(fm[0]) or getattr(g, "rand_idxs", None) is None):
                return None
            devs.append(dev)
        rt, rs = self.graph_target.rand_idxs, self.graph_source.rand_idxs
        if devs[0].ctx is not devs[1].ctx or not (1 <= len(rt) <= 16384 and 1 <= len(rs) <= 16384) or self.n_features > 16:
            return None
        from .graph import device_block_is_current

        if not (device_block_is_current(self.graph_target) and device_block_is_current(self.graph_source)):
            return None
        (ct, st), (cs, ss) = self.graph_target._final_map, self.graph_source._final_map
        k = self.n_features
        try:
            self._device_result = devs[0].ctx.eigsort_costs(devs[0], devs[1], rt, rs, k, ct[:k], st[:k], cs[:k], ss[:k])
        except _hip.PfError as exc:
            if getattr(exc, "code", None) == _hip.PF_E_PERSIST_TIMEOUT:
                raise
            return None  # e.g. a device whose LDS cannot hold one column's sort: the host path computes the same matrices
        return self._device_result

    def _ctx(self):
        ctx = getattr(self.graph_target, "_ctx", None)
        return ctx if ctx is not None else _hip.default_context()

    def eigen_sort(self):
        """eigsort.py:54-140."""
        c = self.c_spatial * self.c_lambda * self.c_hist
        c_f = self.c_spatial_f * self.c_lambda * self.c_hist_f
        self.Q = np.min((c, c_f), axis=0)
        S = c > c_f  # True where the FLIPPED pairing is cheaper
        (target_flipped, source_flipped) = np.where(S == True)  # noqa: E712

        if self.target_as_reference is True:
            target_matches, source_matches = linear_sum_assignment(self.Q)
        elif self.target_as_reference is False:
            source_matches, target_matches = linear_sum_assignment(self.Q.T)
        self.Q = self.Q[target_matches, source_matches]

        flipped_pairs = [
            p2
            for p1 in zip(target_flipped, source_flipped)
            for p2 in zip(target_matches, source_matches)
            if p2 == p1
        ]
        # eigsort.py:108-122: negate the flipped columns of the non-reference graph, then move its columns into
        # the reference's order.  Same result as the reference's two statements, without copying the whole
        # (n, k) array when the assignment is the identity (the usual case) — at 250k vertices that copy
        # costs more than the rest of eigsort together.
        if self.target_as_reference is True:
            flip_cols = [m1 for _, m1 in flipped_pairs]
            dst, src = np.asarray(target_matches), np.asarray(source_matches)
        else:
            flip_cols = [m0 for m0, _ in flipped_pairs]
            dst, src = np.asarray(source_matches), np.asarray(target_matches)
        mutated = self.graph_source if self.target_as_reference is True else self.graph_target
        changes = len(flip_cols) > 0 or not np.array_equal(dst, src)
        fmap = getattr(mutated, "_final_map", None)  # the same flips / permutation for the graph's device-resident block
        on_device = (fmap is not None and changes and getattr(mutated, "_eig_vecs", None) is not None
                     and getattr(mutated, "_remap_ready", lambda: False)())
        # the host array: rewritten from the device block below when that is possible (one DMA instead of strided host
        # passes); read - which collects a download that may still be owed or in flight - only if it has to be changed here
        vecs = mutated.eig_vecs if (changes and not on_device) else None
        for col in flip_cols:
            if vecs is not None:  # (None: a graph held on another rank / on the device only, see parallel.py)
                np.negative(vecs[:, col], out=vecs[:, col])  # one strided pass, no temporary (x * -1 == -x bit for bit)
            if fmap is not None:
                fmap[1][col] = -fmap[1][col]
        if not np.array_equal(dst, src):
            if vecs is not None:
                vecs[:, dst] = vecs[:, src]
            if fmap is not None:
                cols, signs = fmap[0].copy(), fmap[1].copy()
                cols[dst], signs[dst] = fmap[0][src], fmap[1][src]
                fmap = (cols, signs)
        if fmap is not None:
            mutated._final_map = fmap
        if on_device and not mutated._remap_host_image():
            raise RuntimeError("eigsort: the device-resident eigenvector block vanished during eigen_sort")
        self.target_matches, self.source_matches = np.asarray(target_matches), np.asarray(source_matches)
        self.flipped_pairs = flipped_pairs

        if self.verbose:
            print_header("Eigenvector Sorting Results")
            if self.target_as_reference is True:
                print("Using target eigenmaps as the reference")
            elif self.target_as_reference is False:
                print("Using source eigenmaps as the reference")
            print("The matches for eigenvectors were as follows:")
            print("Target\t|  Source")
            for matched_pair in zip(target_matches, source_matches):
                source_value = str(matched_pair[1])
                target_value = str(matched_pair[0])
                if matched_pair in flipped_pairs:
                    if self.target_as_reference is True:
                        source_value = "-" + source_value
                    elif self.target_as_reference is False:
                        target_value = "-" + target_value
                print("{:6}\t|  {:6}".format(target_value, source_value))
            print("*Negative source values means those eigenvectors were flipped*\n ")

    def calc_c_lambda(self):
        """eigsort.py:142-160."""
        for graph in [self.graph_source, self.graph_target]:
            if graph.eig_val_gap is None:
                graph.get_eig_val_gap()
        eigen_gap = (self.graph_target.eig_val_gap + self.graph_source.eig_val_gap) / 2
        for i in range(self.n_features):
            for j in range(self.n_features):
                self.c_lambda[i, j] = np.exp(
                    (self.graph_target.eig_vals[i] - self.graph_source.eig_vals[j]) ** 2 / (2 * eigen_gap**2)
                )

    def calc_c_hist(self):
        """eigsort.py:162-189: c_hist[i, j] = W1(log(T_i + 0.5 + eps), log(+-S_j + 0.5 + eps)).

        The reference calls scipy's `wasserstein_distance` 2 k^2 times, each sorting both
        samples again.  Every column is sorted once here; for equally sized samples the
        1-D earth mover's distance between two empirical distributions is the mean absolute
        difference of their order statistics (identical to scipy's CDF integral up to
        summation rounding, ~1e-16 relative).  Unequal sizes use the same identity with the merged breakpoints of the
        two step quantile functions (`_w1_quantile_plan`)."""
        dev = self._device_costs()
        if dev is not None:
            self.c_hist[:, :], self.c_hist_f[:, :] = dev[0][0], dev[0][1]
            return
        eps = np.finfo(float).eps
        k = self.n_features
        # log is monotone: the raw columns are sorted (2k sorts instead of 3k), log(v + c) of the ascending values is
        # ascending, and log(-v + c) ascending is the same read from the descending end - the same multisets of values
        raw_t = [np.sort(self.rand_target_eig_vecs[:, i]) for i in range(k)]
        raw_s = [np.sort(self.rand_source_eig_vecs[:, j]) for j in range(k)]
        log_t = [np.log(c + 0.5 + eps) for c in raw_t]
        log_s = [np.log(c + 0.5 + eps) for c in raw_s]
        log_sf = [np.log(-c[::-1] + 0.5 + eps) for c in raw_s]
        n_t, n_s = self.rand_target_eig_vecs.shape[0], self.rand_source_eig_vecs.shape[0]
        if n_t != n_s:
            # samples of different size (both meshes sampled completely: 14 998 vs 14 996 vertices): the quantile form,
            # with the breakpoint work shared by all pairs (12 ms -> 1.5 ms per pair against a merge per distance)
            lens, iu, iv = _w1_quantile_plan(n_t, n_s)
            log_s = np.stack([c[iv] for c in log_s])
            log_sf = np.stack([c[iv] for c in log_sf])
            for i in range(k):  # (no BLAS here: a threaded ddot of 30k values costs more than the whole matrix)
                col = log_t[i][iu]
                self.c_hist[i, :] = (np.abs(col - log_s) * lens).sum(axis=1)
                self.c_hist_f[i, :] = (np.abs(col - log_sf) * lens).sum(axis=1)
            return
        for i in range(k):
            for j in range(k):
                self.c_hist[i, j] = np.mean(np.abs(log_t[i] - log_s[j]))
                self.c_hist_f[i, j] = np.mean(np.abs(log_t[i] - log_sf[j]))

    def calc_c_spatial(self):
        """eigsort.py:191-233; the KDTree query runs on the GPU."""
        dev = self._device_costs()
        if dev is not None:
            self.c_spatial[:, :], self.c_spatial_f[:, :] = dev[0][2], dev[0][3]
            self.idx_source_for_each_target_pt = dev[1]
            return
        idx = self._ctx().knn1(self.rand_source_points, self.rand_target_points)
        self.idx_source_for_each_target_pt = idx
        m = self.rand_target_eig_vecs.shape[0]
        k = self.n_features
        src = [np.ascontiguousarray(self.rand_source_eig_vecs[idx, j]) for j in range(k)]  # gathered once per column
        tgt = [np.ascontiguousarray(self.rand_target_eig_vecs[:, i]) for i in range(k)]
        for i in range(k):
            for j in range(k):
                self.c_spatial[i, j] = np.sqrt(np.sum((src[j] - tgt[i]) ** 2)) / m
                self.c_spatial_f[i, j] = np.sqrt(np.sum((-src[j] - tgt[i]) ** 2)) / m

    def sort_eigenmaps(self):
        """eigsort.py:235-249."""
        self.calc_c_lambda()
        self.calc_c_hist()
        self.calc_c_spatial()
        self.eigen_sort()
        return self.Q
