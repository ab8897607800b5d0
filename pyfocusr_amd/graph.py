"""`Graph` — mesh -> weighted adjacency -> degree -> random-walk Laplacian ->
lowest-k non-null eigenpairs -> min-max-normalised spectral coordinates.

Drop-in mirror of `/root/reference/pyfocusr/graph.py` (`class Graph` :18-354,
`recursive_eig` :357-389): same constructor arguments, methods and attribute
names, so `eigsort` / `Focusr` / user code written against the reference keep
working.  The arithmetic runs on an MI355X through `libpyfocusr_hip.so`:

* `get_weighted_adjacency_matrix / get_degree_matrix / get_G_matrix /
  get_laplacian_matrix` (graph.py:148-226)  -> `pf_graph_build` (one call builds
  all of them on the device; the scipy attributes are host views of its output,
  bit-identical to the reference's matrices);
* `recursive_eig` -> scipy `eigs` ARPACK shift-invert (graph.py:372)  ->
  Chebyshev-filtered Krylov-Schur on the device (`_krylov.filtered_eigs`), with
  the reference's `> 1e-10` null filter and `k += k_buffer + n_k_needed` widening
  rule reproduced from the number of connected components;
* eigenvector normalisation (graph.py:254-257) -> `pf_finalize_vectors`;
* `mean_filter_graph` (graph.py:320-354) -> `pf_mean_filter`.

Deliberate strengthenings: `feature_weights=None` works
(A1); eigenpairs are returned sorted ascending with a deterministic sign —
largest-|entry| positive (A5); `get_list_rand_idxs(force_randomization=True)`
uses `np.random.shuffle` (A7).  Extra node features (curvature, point-data
arrays; graph.py:85-119,166-175,191-210) need VTK and are outside the hot path:
non-empty feature lists raise `NotImplementedError`.
"""
import os

import numpy as np
from scipy import sparse

from . import _hip
from ._krylov import MIN_EIG_VAL, drive, drive_pair, filtered_eigs_gen
from .vtk_functions import mesh_arrays, vtk_deep_copy  # noqa: F401

__all__ = ["Graph", "recursive_eig", "compute_spectra", "spectral_knn"]


class _DeviceBackedCSR(sparse.csr_matrix):
    """scipy CSR view of the device Laplacian that remembers the device graph, so
    `recursive_eig(graph.laplacian_matrix, ...)` can run on the GPU."""

    _pf_device = None


def _widened_k(k, n_k_needed, k_buffer, n_null, n_max):
    """graph.py:374-379: ARPACK is asked for k pairs; those <= 1e-10 are dropped;
    while fewer than n_k_needed remain, k += k_buffer + n_k_needed."""
    retries = 0
    while k - n_null < n_k_needed and k + k_buffer + n_k_needed < n_max:
        k += k_buffer + n_k_needed
        retries += 1
    return k, retries


class Graph(object):
    def __init__(
        self,
        vtk_mesh,
        n_spectral_features=3,
        norm_eig_vecs=True,
        n_rand_samples=10000,
        list_features_to_calc=[],
        list_features_to_get_from_mesh=[],
        feature_weights=None,
        include_features_in_adj_matrix=False,
        include_features_in_G_matrix=False,
        G_matrix_p_function="exp",
        norm_node_features_std=True,
        norm_node_features_cap_std=3,
        norm_node_features_0_1=True,
        ctx=None,
        verbose=True,
    ):
        # Inputs (graph.py:36-55)
        self.vtk_mesh = vtk_mesh
        self.n_spectral_features = n_spectral_features
        self.norm_eig_vecs = norm_eig_vecs
        self.include_features_in_adj_matrix = include_features_in_adj_matrix
        self.include_features_in_G_matrix = include_features_in_G_matrix
        self.G_matrix_p_function = G_matrix_p_function
        self.norm_node_features_std = norm_node_features_std
        self.norm_node_features_cap_std = norm_node_features_cap_std
        self.norm_node_features_0_1 = norm_node_features_0_1
        self.verbose = verbose
        self._ctx = ctx

        if len(list_features_to_calc) or len(list_features_to_get_from_mesh):
            raise NotImplementedError(
                "extra node features (graph.py:85-119) need VTK and are outside the MI355X hot path; "
                "pass list_features_to_calc=[] and list_features_to_get_from_mesh=[]")
        self.node_features = []
        self.n_extra_features = 0
        self.mean_xyz_range_scaled_features = []
        self.feature_weights = np.eye(self.n_extra_features) if feature_weights is None else feature_weights

        # Mesh/points characteristics (graph.py:58-67)
        self.points, self._faces = mesh_arrays(vtk_mesh)
        self.n_points = int(self.points.shape[0])
        # graph.py:63-67: bounding-box ranges and normed_points are only read by the optional
        # point/feature coordinates of Focusr; they are computed on first access.
        self._geometry = None
        self._normed_points = None

        # Matrices (graph.py:70-76): filled from the device graph on demand.
        self._device = None
        self._host = None
        self._adjacency_matrix = None
        self.degree_matrix = None
        self.degree_matrix_inv = None
        self.laplacian_matrix = None
        self.G = None

        self.eig_vals = None
        self._eig_vecs = None
        self._eig_pending = False
        self._final_map = None  # (cols, signs): eig_vecs[:, c] == signs[c] * (device-resident block)[:, cols[c]], or None
        self.eig_val_gap = None
        self.eigs_stats = None
        self.rand_idxs = self.get_list_rand_idxs(n_rand_samples)

    # ------------------------------------------------------------------ eigenvectors: host array + device-resident twin
    _eig_pending = False  # (class default: shells made with Graph.__new__ - parallel.py, bench.py - have no download in flight)
    _eig_vecs = None
    _final_map = None

    @property
    def eig_vecs(self):
        if self._eig_pending:  # the download the eigensolve queued is collected by the first reader
            try:
                self._device.finalize_wait()
            except Exception:
                # the image never arrived (a non-finite norm, a HIP error): no later reader may mistake the unfilled
                # pinned array for eigenvectors
                self._eig_pending = False
                self._eig_vecs = None
                self._final_map = None
                raise
            self._eig_pending = False
        return self._eig_vecs

    @eig_vecs.setter
    def eig_vecs(self, value):
        """Assigning eigenvectors from outside disconnects the host array from the block the solver left on the
        device (`_set_spectrum` is the one place that keeps them connected)."""
        if self._eig_pending:
            self._eig_pending = False
            self._device.finalize_wait()
        self._eig_vecs = value
        self._final_map = None

    def _remap_ready(self):
        fm, dev, vecs = self._final_map, self._device, self._eig_vecs
        return bool(fm is not None and dev is not None and getattr(dev, "_h", None) and vecs is not None
                    and hasattr(dev, "final_remap") and vecs.shape == (self.n_points, len(fm[0])) and vecs.flags.c_contiguous
                    and vecs.flags.writeable and getattr(dev, "_final_count", 0) == len(fm[0]))

    def _remap_host_image(self):
        """Rewrite the host array of `eig_vecs` IN PLACE as (device-resident block)[:, cols] * signs of `_final_map` - what
        eigsort's sign flips and column moves (eigsort.py:108-122) make of it - by one kernel and one DMA that the next
        reader collects.  False when the array is not the pinned image of the block (assigned from outside, closed device)."""
        if not self._remap_ready():
            return False
        fm, dev, vecs = self._final_map, self._device, self._eig_vecs
        # (an image of the block that is still owed to `vecs` is replaced by the remapped one; one in flight is collected
        # first - both inside pf_final_remap_begin)
        dev.final_remap(fm[0], fm[1], vecs)
        self._eig_pending = True
        return True

    def _set_spectrum(self, vals, vecs, stats):
        self.eig_vals, self._eig_vecs, self.eigs_stats = vals, vecs, stats
        m = 0 if vecs is None else vecs.shape[1]
        on_device = self._device is not None and getattr(self._device, "_final_count", 0) == m and m > 0
        self._final_map = (np.arange(m), np.ones(m)) if on_device else None
        # the (n, m) array is pinned memory that a copy stream is still filling (`DeviceLaplacian.finalize_vectors(wait=False)`):
        # eigsort's cost matrices and the KNN read the device-resident twin meanwhile
        self._eig_pending = bool(on_device and getattr(self._device, "_final_pending", False))

    # ------------------------------------------------------------------ device graph
    @property
    def device(self):
        """The `DeviceLaplacian` of this mesh (built on first use)."""
        if self._device is None:
            resident = getattr(self.vtk_mesh, "_pf_device_mesh", None)  # inputs already in HBM (bench, pipelines)
            if resident is not None:
                self._device = _hip.DeviceLaplacian(device_mesh=resident)
            else:
                self._device = _hip.DeviceLaplacian(self.points, self._faces, ctx=self._ctx)
        return self._device

    def _host_arrays(self):
        if self._host is None:
            self._host = self.device.download()
        return self._host

    def _geom(self):
        if self._geometry is None:
            rng = np.ptp(self.points, axis=0)
            self._geometry = (rng, np.max(rng), np.mean(rng))
        return self._geometry

    @property
    def pts_scale_range(self):
        return self._geom()[0]

    @property
    def max_pts_scale_range(self):
        return self._geom()[1]

    @property
    def mean_pts_scale_range(self):
        return self._geom()[2]

    @property
    def normed_points(self):
        if self._normed_points is None:
            self._normed_points = (self.points - np.min(self.points, axis=0)) / self.mean_pts_scale_range
        return self._normed_points

    @normed_points.setter
    def normed_points(self, value):
        self._normed_points = value

    @property
    def adjacency_matrix(self):
        if self._adjacency_matrix is None:
            return sparse.lil_matrix((self.n_points, self.n_points))  # graph.py:70-72 (empty until computed)
        return self._adjacency_matrix

    @adjacency_matrix.setter
    def adjacency_matrix(self, value):
        self._adjacency_matrix = value

    def norm_node_features(self, norm_using_std=True, norm_range_0_to_1=True, cap_std=3):
        """graph.py:121-142 (no features on the hot path: nothing to do)."""
        return None

    # ------------------------------------------------------------------ matrices (graph.py:148-226)
    def get_weighted_adjacency_matrix(self):
        h = self._host_arrays()
        n = self.n_points
        self._adjacency_matrix = sparse.csr_matrix((h["w"], h["colidx"], h["rowptr"]), shape=(n, n))

    def get_degree_matrix(self):
        h = self._host_arrays()
        self.degree_matrix = sparse.diags(h["deg"])
        self.degree_matrix_inv = sparse.diags((h["deg"] + 1e-8) ** -1)

    def get_G_matrix(self, p_function="exp"):
        if self.degree_matrix_inv is None:
            self.get_degree_matrix()
        self.G = self.degree_matrix_inv  # graph.py:213-214 (no extra features)

    def get_laplacian_matrix(self):
        if self.G is None:
            self.get_G_matrix()
        h = self._host_arrays()
        n = self.n_points
        rowptr = h["rowptr"].astype(np.int64)
        cnt = np.diff(rowptr)
        has_diag = cnt > 0  # scipy drops the explicit zero diagonal of isolated vertices
        rows = np.repeat(np.arange(n), cnt)
        below = np.zeros(n + 1, dtype=np.int64)  # entries of row i with column < i
        np.add.at(below, rows[h["colidx"] < rows] + 1, 1)
        new_ptr = rowptr + np.concatenate([[0], np.cumsum(has_diag)])
        nnz = int(new_ptr[-1])
        data = np.empty(nnz)
        idx = np.empty(nnz, dtype=np.int32)
        diag_pos = new_ptr[:-1] + below[1:]
        pos_in_row = np.arange(len(rows)) - rowptr[rows]
        shift = (h["colidx"] > rows).astype(np.int64)
        dest = new_ptr[rows] + pos_in_row + shift
        data[dest] = h["l_offdiag"]
        idx[dest] = h["colidx"]
        data[diag_pos[has_diag]] = h["l_diag"][has_diag]
        idx[diag_pos[has_diag]] = np.arange(n, dtype=np.int32)[has_diag]
        L = _DeviceBackedCSR((data, idx, new_ptr.astype(np.int32)), shape=(n, n))
        L._pf_device = self.device
        self.laplacian_matrix = L

    # ------------------------------------------------------------------ spectrum (graph.py:228-257)
    def get_graph_spectrum(self):
        dev = self.device
        if self.verbose:
            print("Beginning Eigen Decomposition")
        if PAIR_DRIVER == "c" and _spectra_c([self]):
            if self.verbose:
                print("All final eigenvalues are: \n{}".format(self.eig_vals))
                print("-" * 72)
                print("Final eigenvalues of interest are: \n{}".format(self.eig_vals))
            return
        self._set_spectrum(*_device_eigs(
            dev,
            k=self.n_spectral_features + 1,
            n_k_needed=self.n_spectral_features,
            k_buffer=1,
            minmax=self.norm_eig_vecs is True,
            verbose=self.verbose,
            wait=False,
        ))
        if self.verbose:
            print("All final eigenvalues are: \n{}".format(self.eig_vals))
            print("-" * 72)
            print("Final eigenvalues of interest are: \n{}".format(self.eig_vals))

    # ------------------------------------------------------------------ samplers (graph.py:263-290)
    def get_eig_val_gap(self):
        self.eig_val_gap = np.mean(np.diff(self.eig_vals))

    def get_rand_eig_vecs(self):
        fm = self._final_map
        if (fm is not None and self._device is not None and getattr(self._device, "_h", None)
                and device_block_is_current(self)):  # (an unannounced in-place edit of eig_vecs disowns the block)
            # the sampled rows straight from the block the solver left in HBM: a random gather of rows of the
            # freshly downloaded (cache-cold) host array costs more than the whole device round trip
            rows = self._device.final_rows(self.rand_idxs)
            cols, signs = self._final_map
            if np.array_equal(cols, np.arange(len(cols))) and np.all(signs == 1.0):
                return rows
            return rows[:, cols] * signs
        return self.eig_vecs[self.rand_idxs, :]

    def get_rand_normalized_points(self):
        dev = self._device
        if dev is not None and getattr(dev, "_h", None) and getattr(dev, "has_points", False) and len(self.rand_idxs) < self.n_points:
            sample = dev.point_rows(self.rand_idxs)  # the same rows from the copy in HBM: cheaper than a cache-cold host gather
        else:
            sample = self.points[self.rand_idxs, :]  # gathered once (the reference gathers the same rows three times)
        # (sample - min) / ptp per coordinate, as graph.py:269-272 - on the transposed copy: reductions along the long,
        # contiguous axis (numpy's axis-0 reductions of an (n, 3) array cost 0.1 ms each at n = 5000); same values
        st = np.ascontiguousarray(sample.T)
        lo = st.min(axis=1)
        return np.ascontiguousarray(((st - lo[:, None]) / (st.max(axis=1) - lo)[:, None]).T)

    def get_list_rand_idxs(self, n_rand_samples, replace=False, force_randomization=False):
        if n_rand_samples > self.n_points:
            list_points = np.arange(self.n_points)
            if force_randomization is True:
                np.random.shuffle(list_points)
            return list_points
        if replace:
            return np.random.choice(self.n_points, size=n_rand_samples, replace=True)
        # same distribution as np.random.choice(replace=False) (graph.py:290) without permuting all
        # n_points: a Generator seeded from the legacy global state (so np.random.seed still pins it).
        # (shuffle=False: the sample is the same uniform SET, only not in random order - every consumer of `rand_idxs` is a
        # sum, a sort or a per-point search over the sample; the shuffle was half of the 0.15 ms this draw costs per graph,
        # host time in front of the first kernel of a step)
        rng = np.random.default_rng(np.random.randint(0, 2**31 - 1))
        return rng.choice(self.n_points, size=n_rand_samples, replace=False, shuffle=False)

    # ------------------------------------------------------------------ viewers (graph.py:296-314)
    def _no_viewer(self, *a, **k):
        raise ImportError("itkwidgets viewers are not part of the MI355X hot path")

    view_mesh_existing_scalars = view_mesh_eig_vec = view_mesh_features = _no_viewer

    # ------------------------------------------------------------------ graph filter (graph.py:320-354)
    def mean_filter_graph(self, values, iterations=300):
        """out = ((D+I)^-1 (W+I))^iterations values, on the device."""
        return self.device.mean_filter(np.asarray(values, dtype=np.float64), iterations)


PAIRED_LAUNCHES = True  # two graphs of one context advance their Chebyshev recurrences in shared launches


def _pair_pays(ga, gb):
    """Should two graphs share kernel launches (`pf_cheb2`)?  Sharing amortises the ~3 us launch latency of a
    Chebyshev step: 2 x 50k vertices take 4.2 us shared vs 6.2 us apart, 2 x 1M 29.7 vs 32.9.  Around 250k
    vertices it is a wash (9.6 us shared vs 2 x 4.5 apart): ONE operator then just fits the eight 4 MiB XCD
    L2s and stays resident from launch to launch, two do not.  The shared form is kept everywhere: the gain of
    the split form at that size is ~2 % and disappears under a profiler, which flushes L2 between dispatches."""
    return bool(PAIRED_LAUNCHES)


def compute_spectra(graphs):
    """`get_graph_spectrum()` of several graphs at once (Focusr.__init__ does target then
    source, focusr.py:150,169; the two are independent).  Two graphs of one context run in
    lockstep on its stream (`_paired_spectra`); graphs of different contexts run from one host
    thread each, every one on its own HIP stream (ctypes releases the GIL during library calls)."""
    graphs = list(graphs)
    build_devices(graphs)  # (two graphs that are still to be assembled: side by side)
    if len(graphs) == 2 and graphs[0].device.ctx is graphs[1].device.ctx and _pair_pays(graphs[0], graphs[1]):
        _paired_spectra(graphs[0], graphs[1])
        return
    if len(graphs) <= 1 or len({id(g.device.ctx) for g in graphs}) < len(graphs):
        for g in graphs:
            g.get_graph_spectrum()
        return
    import threading

    errors = []

    def run(g):
        try:
            g.get_graph_spectrum()
        except BaseException as exc:  # noqa: BLE001 - re-raised in the caller's thread
            errors.append(exc)

    threads = [threading.Thread(target=run, args=(g,)) for g in graphs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]


def build_devices(graphs):
    """The device graphs of several `Graph`s; two that still have to be assembled, on one context, are assembled side
    by side (`DeviceLaplacian.build_pair`: two streams; an assembly is mostly launch latency).  Same results as touching
    `.device` of each."""
    graphs = list(graphs)
    todo = [g for g in graphs if g._device is None]
    if len(todo) == 2 and os.environ.get("PF_PAIR_BUILD", "1") != "0" and (todo[0]._ctx is todo[1]._ctx):
        ga, gb = todo
        meshes = []
        for g in todo:
            m = getattr(g.vtk_mesh, "_pf_device_mesh", None)  # inputs already in HBM (bench, pipelines)
            if m is None:
                m = _hip.DeviceMesh(g.points, g._faces, ctx=g._ctx)
            meshes.append(m)
        if meshes[0].ctx is meshes[1].ctx:
            ga._device, gb._device = _hip.DeviceLaplacian.build_pair(*meshes)
    return [g.device for g in graphs]


PAIR_DRIVER = os.environ.get("PF_PAIR_DRIVER", "c")  # "c": pf_eigs_smallest / pf_eigs_smallest2 where they apply; "python": always the generators of _krylov.py


def _c_plan(g, dev):
    """How many eigenpairs the C driver has to return for `g`: the column count the reference's widen-and-retry loop
    ends with (graph.py:374-379), given the null eigenvalues the assembler already knows."""
    n_null = dev.n_components + dev.n_isolated
    k_final, retries = _widened_k(g.n_spectral_features + 1, g.n_spectral_features, 1, n_null, dev.n)
    return max(min(k_final - n_null, dev.n - n_null), 0), retries


def _c_accept(g, dev, plan, result):
    """Store what the C driver returned for `g` (see `_spectra_c`)."""
    from ._krylov import EigsStats

    (m_out, retries), (vals, vecs, st) = plan, result
    if g.verbose:
        print("Starting!")
        for _ in range(retries):
            print("Not enough eigenvalues found, trying again with more eigenvalues!")
            print("Starting!")
    stats = EigsStats()
    for key in ("matvecs", "outer_steps", "restarts", "filter_resets", "degree", "cut", "n_null", "second_passes", "mode", "local_steps"):
        setattr(stats, key, st[key])
    stats.residuals = st["residuals"]
    g._set_spectrum(vals, vecs, stats)


def _spectra_c(graphs):
    """The eigensolve of one graph, or of the two graphs of a pair, behind ONE library call (`pf_eigs_smallest` /
    `pf_eigs_smallest2`: the Krylov driver restated in C++ - Lanczos for symmetric W, Arnoldi with carried outliers or
    the ellipse filter for asymmetric W): no interpreter between the launches of an outer step.  Returns False when the
    call does not cover the graphs (too small for the filtered iteration, a general matrix without analytic null
    vectors, null vectors the component count did not predict, fewer eigenpairs than the reference's widen-and-retry
    loop ends with): the Python driver then does the whole solve."""
    devs = [g.device for g in graphs]
    if not all(hasattr(d, "eigs_smallest2") and getattr(d, "lock_nulls", True) for d in devs):
        return False
    if len(graphs) == 2 and (devs[0].ctx is not devs[1].ctx or (graphs[0].norm_eig_vecs is True) != (graphs[1].norm_eig_vecs is True)):
        return False
    plans = [_c_plan(g, d) for g, d in zip(graphs, devs)]
    if any(m_out == 0 for m_out, _ in plans):
        return False
    minmax = graphs[0].norm_eig_vecs is True
    try:
        if len(graphs) == 2:
            results = devs[0].eigs_smallest2(devs[1], plans[0][0], plans[1][0], minmax=minmax, wait=False)
        else:
            results = (devs[0].eigs_smallest(plans[0][0], minmax=minmax),)
    except _hip.PfError as exc:
        if getattr(exc, "code", None) in (_hip.PF_E_STATE, _hip.PF_E_DEGENERATE):
            return False
        raise
    for dev, (m_out, _), (vals, vecs, st) in zip(devs, plans, results):
        if len(vals) != m_out or st["n_null"] != dev.n_components:
            for d in devs:
                d.finalize_wait()
            return False
    for g, dev, plan, result in zip(graphs, devs, plans, results):
        _c_accept(g, dev, plan, result)
    return True


def _paired_spectra(ga, gb):
    """Both spectra on ONE stream with the two Chebyshev recurrences advancing in shared kernel
    launches (`pf_cheb2`): a 250k-vertex filter step alone is a ~5 us kernel of which ~3 us is
    launch/ramp latency, so two graphs per launch cost ~1.5x one."""
    for g in (ga, gb):
        if g.verbose:
            print("Beginning Eigen Decomposition")
    if PAIR_DRIVER == "c" and _spectra_c([ga, gb]):
        for g in (ga, gb):
            if g.verbose:
                print("All final eigenvalues are: \n{}".format(g.eig_vals))
                print("-" * 72)
                print("Final eigenvalues of interest are: \n{}".format(g.eig_vals))
        return

    def solver(g):  # a factory: `drive_pair` repeats both solves if the resident filter kernel had to give up
        return lambda: _device_eigs_gen(g.device, k=g.n_spectral_features + 1, n_k_needed=g.n_spectral_features,
                                        k_buffer=1, minmax=g.norm_eig_vecs is True, verbose=g.verbose)

    ra, rb = drive_pair(solver(ga), ga.device, solver(gb), gb.device)
    for g, (vals, vecs, stats) in ((ga, ra), (gb, rb)):
        g._set_spectrum(vals, vecs, stats)
        if g.verbose:
            print("All final eigenvalues are: \n{}".format(g.eig_vals))
            print("-" * 72)
            print("Final eigenvalues of interest are: \n{}".format(g.eig_vals))


def device_block_is_current(g):
    """Guard against in-place edits of `g.eig_vecs` nobody told us about: 64 rows of the device-resident block (with the
    recorded column permutation / sign flips) must agree exactly with the host array; otherwise the block is disowned."""
    if getattr(g, "_eig_pending", False):
        return True  # the host array is still on its way: nobody can have edited it
    rows = np.linspace(0, g.n_points - 1, num=min(64, g.n_points)).astype(np.int64)
    cols, signs = g._final_map
    if np.array_equal(g._device.final_rows(rows)[:, cols] * signs, g.eig_vecs[rows][:, :len(cols)]):
        return True
    g._final_map = None
    return False


def spectral_knn(graph_target, graph_source, n_coords, weights=None):
    """focusr.py:351-353 on the spectral coordinates `eig_vecs[:, :n_coords] * weights` of two graphs WITHOUT the
    n x k coordinate arrays crossing PCIe: both graphs still hold the block their eigensolve left in HBM, and the
    column permutation / sign flips `eigsort` applied to the host arrays (eigsort.py:108-122) are folded, with the
    weights, into per-column scale factors of `pf_knn1_graphs`.  Bit-identical to `ctx.knn1(target_coords,
    source_coords)` on the host arrays ((-v) w = v (-w) exactly).  Returns the target index of every source vertex,
    or None when a graph's host eigenvectors are no longer the device block's (assigned from outside, device
    closed, graphs on different contexts): the caller then takes the host-array path."""
    devs = []
    for g in (graph_target, graph_source):
        fm, dev = getattr(g, "_final_map", None), getattr(g, "_device", None)
        if fm is None or dev is None or not getattr(dev, "_h", None) or n_coords > len(fm[0]):
            return None
        devs.append(dev)
    if devs[0].ctx is not devs[1].ctx:
        return None
    for g in (graph_target, graph_source):
        if not device_block_is_current(g):
            return None
    w = np.ones(n_coords) if weights is None else np.asarray(weights, dtype=np.float64)
    (ct, st), (cs, ss) = graph_target._final_map, graph_source._final_map
    return devs[0].ctx.knn1_graphs(devs[0], devs[1], ct[:n_coords], st[:n_coords] * w, cs[:n_coords], ss[:n_coords] * w)


def _device_eigs_gen(dev, k, n_k_needed, k_buffer=1, minmax=False, verbose=False, **solver_kw):
    """`recursive_eig` on a device graph, as a generator of filter requests (see
    `_krylov.filtered_eigs_gen`).  Result: (eig_vals ascending, eig_vecs (n, m), stats) with
    m = k_final - (#null eigenvalues) >= n_k_needed, exactly the column count the reference's
    widen-and-retry loop ends with."""
    n = dev.n
    lock = getattr(dev, "lock_nulls", True)  # False: general matrix whose rows do not sum to zero
    n_locked_null = dev.n_components + dev.n_isolated if lock else 0
    extra = 0  # null eigenvalues the solver itself found (they occupy Ritz slots: not locked)
    if verbose:
        print("Starting!")
    while True:
        n_null = n_locked_null + extra
        k_final, retries = _widened_k(k, n_k_needed, k_buffer, n_null, n)
        if verbose:
            for _ in range(retries):
                print("Not enough eigenvalues found, trying again with more eigenvalues!")
                print("Starting!")
        m_out = max(min(k_final - n_null, n - n_null), 0)
        if m_out == 0:
            return np.zeros(0), np.zeros((n, 0)), None
        c0 = dev.lock_null_vectors() if lock else 0
        # many one-way edges (an open mesh: every boundary edge) make L strongly non-normal: go straight to the
        # ellipse filter; a handful (the bundled 15k meshes) are cheaper to carry as outliers of the interval filter
        kw = dict(solver_kw)
        if dev.symmetric:
            # the filter damps [cut, hi]: hi = the operator's proven spectral bound (2 in general; ~1.6-1.7 for a closed
            # triangle mesh, pf_graph_info.spectral_bound) - the degree goes with the square root of the interval
            kw.setdefault("hi", getattr(dev, "spectral_bound", 2.0))
        if not dev.symmetric and "ellipse" not in kw:
            kw["ellipse"] = True if getattr(dev, "n_oneway", 0) > 32 else None
        lam, first, stats = yield from filtered_eigs_gen(dev, m_out + extra, dev.symmetric, null_slots=c0, nulls_fresh=lock, **kw)
        found = stats.n_null - c0
        if found > extra and len(lam) < m_out:  # null vectors the component count did not predict
            extra = found
            continue
        break
    vecs = dev.finalize_vectors(first, len(lam), minmax, wait=False)  # callers: Graph._set_spectrum, or dev.finalize_wait()
    return lam, vecs, stats


def _device_eigs(dev, k, n_k_needed, k_buffer=1, minmax=False, verbose=False, wait=True, **solver_kw):
    """`wait=False`: the eigenvector array is returned while its download is still in flight (`dev.finalize_wait()`)."""
    out = drive(lambda: _device_eigs_gen(dev, k, n_k_needed, k_buffer, minmax, verbose, **solver_kw), dev)
    if wait and hasattr(dev, "finalize_wait"):
        dev.finalize_wait()
    return out


def _device_from_matrix(matrix, ctx=None):
    """Upload a general sparse matrix (the argument of the reference's `recursive_eig`) and work out
    what the filtered solver needs to know about it: an upper bound of the spectrum (Gershgorin) and
    whether the indicator vectors of its connected components are null vectors (rows summing to 0)."""
    A = sparse.csr_matrix(matrix, dtype=np.float64, copy=True)
    if A.shape[0] != A.shape[1]:
        raise ValueError("expected a square matrix")
    A.sum_duplicates()
    A.sort_indices()
    n = A.shape[0]
    dev = _hip.DeviceLaplacian(matrix=(A.indptr, A.indices, A.data), ctx=ctx)
    absA = abs(A)
    hi = float(np.asarray(absA.sum(axis=1)).max()) if A.nnz else 1.0
    scale = float(absA.max()) if A.nnz else 1.0
    row_sums = np.abs(np.asarray(A @ np.ones(n)))
    dev.lock_nulls = bool(row_sums.max() <= 1e-12 * max(scale, 1e-300))
    offdiag = np.diff(A.indptr) - (A.diagonal() != 0)
    if np.any((offdiag == 0) & (A.diagonal() != 0)):
        raise NotImplementedError("rows whose only entry is a non-zero diagonal are not supported")
    return dev, dict(hi=max(hi, 1e-300), cut=8.0 * (1 + 1) / max(n, 1) * max(hi, 1e-300) / 2.0, adapt_cut=True)


def recursive_eig(matrix, k, n_k_needed, k_buffer=1, sigma=1e-10, which="LM"):
    """graph.py:357-389: the `n_k_needed` (or more, after widening) eigenpairs of `matrix` nearest zero
    with eigenvalue > 1e-10.

    A Laplacian produced by `Graph.get_laplacian_matrix()` carries its device graph and is solved in
    place; any other scipy sparse matrix (the reference's signature) is uploaded with
    `pf_graph_from_matrix`.  `sigma`/`which` select ARPACK's shift-invert mode in the reference and have no
    counterpart here: the eigenvalues nearest zero of a positive semi-definite operator are always the
    ones computed.  Returns raw (un-normalised, unit-2-norm) eigenvectors, sorted ascending.

    General matrices: null vectors are locked analytically when the rows sum to zero (one per connected component);
    the filter's undamped interval adapts itself when the low end of the spectrum is far below k/n of its top.
    Known limitation: for a DISCONNECTED matrix WITHOUT null vectors (e.g. a shifted Laplacian), an eigenvalue shared
    by several components may be returned with lower multiplicity (single-vector Krylov iteration; mesh Laplacians
    are not affected: their only degenerate eigenvalue by construction is 0, handled through the components)."""
    dev = getattr(matrix, "_pf_device", None)
    solver_kw = {}
    owned = False
    if dev is None or getattr(dev, "_h", None) is None:
        dev, solver_kw = _device_from_matrix(matrix)
        solver_kw["cut"] = solver_kw["cut"] * (n_k_needed + 1) / 2.0
        owned = True
    try:
        vals, vecs, _ = _device_eigs(dev, k, n_k_needed, k_buffer, minmax=False, verbose=True, **solver_kw)
    finally:
        if owned:
            dev.close()
    return vals, vecs
