"""Multi-GPU layout of the hot path: one process per GPU.

The per-mesh work (assembly + eigensolve) of a pair shards at mesh granularity —
target on rank 0, source on rank 1 — with ONE exchange: an all-gather of the
spectral coordinates (n x k float64, 10 MB at 250k x 5) so that eigsort (k x k,
replicated) and the query-sharded KNN can run.  The collective goes through
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests); tensors of different length are padded to the longest.  `split_pair_correspondence` is the
device-to-device form (the resident eigenvector blocks are sent as they sit in HBM); the host helpers above it
are kept for the CPU-side tests and tools.
"""
import numpy as np


def shard_rows(n_rows, world, rank):
    """Contiguous, balanced [lo, hi) slice of `n_rows` for `rank`."""
    base, rem = divmod(int(n_rows), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _device(dist, torch):
    return "cuda" if dist.get_backend() == "nccl" else "cpu"


def all_gather_rows(dist, torch, arr):
    """All-gather 2-D float64 arrays whose row counts differ per rank -> list of numpy arrays."""
    arr = np.ascontiguousarray(arr, dtype=np.float64)
    if arr.ndim == 1:
        arr = arr[:, None]
    dev = _device(dist, torch)
    world = dist.get_world_size()
    shape = torch.tensor([arr.shape[0], arr.shape[1]], dtype=torch.int64, device=dev)
    shapes = [torch.empty_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape)
    shapes = [tuple(int(v) for v in s.cpu()) for s in shapes]
    rows = max(s[0] for s in shapes)
    cols = max(s[1] for s in shapes)
    pad = torch.zeros((rows, cols), dtype=torch.float64, device=dev)
    pad[: arr.shape[0], : arr.shape[1]] = torch.from_numpy(arr).to(dev)
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return [o[: s[0], : s[1]].cpu().numpy() for o, s in zip(out, shapes)]


def gather_spectral(dist, torch, eig_vals, eig_vecs, points):
    """Every rank contributes its mesh's (eig_vals, eig_vecs, points); every rank gets all."""
    vals = [v[:, 0] for v in all_gather_rows(dist, torch, eig_vals)]
    vecs = all_gather_rows(dist, torch, eig_vecs)
    pts = all_gather_rows(dist, torch, points)
    return vals, vecs, pts


# ------------------------------------------------------------------------------------------------------------------
# Device-to-device form of the split pair (BASELINE config C4): the eigenvectors never visit the host
# ------------------------------------------------------------------------------------------------------------------
class _DeviceBlock(object):
    """A row-major float64 block in device memory, exposed through `__cuda_array_interface__` so that
    `torch.as_tensor` wraps it WITHOUT a copy (the block stays owned by its `DeviceLaplacian`)."""

    def __init__(self, ptr, n_rows, n_cols):
        self.__cuda_array_interface__ = {"shape": (int(n_rows), int(n_cols)), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def resident_block_tensor(torch, dev, device_index):
    """Zero-copy torch view of the block `dev.finalize_vectors` left in HBM ((n, m) float64)."""
    ptr, n_rows, n_cols = dev.final_device()
    return torch.as_tensor(_DeviceBlock(ptr, n_rows, n_cols), device=torch.device("cuda", device_index))


class _SampledGraph(object):
    """What `eigsort` reads of a graph (eigsort.py:34-41, 149-158), backed by sampled rows only: the mesh itself lives
    on another rank.  `eig_vecs` is None - there is no host array to flip or permute; `eigsort.eigen_sort` records
    its flips / permutation in `_final_map`, which the KNN on the gathered device block then applies."""

    def __init__(self, eig_vals, sample_vecs, sample_points, ctx):
        self.eig_vals = np.asarray(eig_vals, dtype=np.float64)
        self.eig_val_gap = None
        self.eig_vecs = None
        self._sample_vecs, self._sample_points = sample_vecs, sample_points
        self._final_map = (np.arange(sample_vecs.shape[1]), np.ones(sample_vecs.shape[1]))
        self._ctx = ctx
        self.verbose = False

    def get_eig_val_gap(self):
        self.eig_val_gap = np.mean(np.diff(self.eig_vals))

    def get_rand_eig_vecs(self):
        return self._sample_vecs

    def get_rand_normalized_points(self):
        return self._sample_points


def _all_gather_stack(dist, torch, t):
    """(world, *t.shape) tensor of every rank's `t` (equal shapes) with ONE `all_gather_into_tensor`, in the flat
    concatenating form every backend implements."""
    world = dist.get_world_size()
    flat = t.contiguous().reshape(-1)
    out = torch.empty(world * flat.numel(), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, flat)
    return out.view(world, *t.shape)


def sample_rows(n_points, n_rand_samples, rng):
    """graph.py:274-290 with an explicit generator (every rank must draw the SAME rows of both meshes)."""
    if n_rand_samples > n_points:
        return np.arange(n_points)
    return rng.choice(n_points, size=n_rand_samples, replace=False)


def split_pair_correspondence(dist, torch, graph, k, n_samples, seed=1234, block=None, knn_blocks=None, stream=None,
                              knn_ctx=None):
    """BASELINE config C4 / SURVEY 8e: rank 0 holds the TARGET graph, rank 1 the SOURCE graph, spectra computed.
    One all-gather of the two resident eigenvector blocks, device to device (RCCL over xGMI with backend "nccl"),
    enqueued on the library's own stream; eigsort replicated from sampled rows (the 5000 x m samples are the only
    eigenvector data that reach a host); weights; the 1-NN of this rank's shard of the source rows against all target
    rows straight from the gathered buffer (`pf_knn1_blocks`); the int64 index shards all-gathered.

    `block` (a (n, m) float64 tensor) and `knn_blocks(ref, n_ref, qry, n_qry, stride, ct, st, cs, ss)` replace the device
    pieces in the CPU test of this control flow (`knn_ctx`: the object whose `knn1` serves eigsort's 3-D query there).
    Returns (idx of every source row, Q, weights)."""
    from .eigsort import eigsort

    world, rank = dist.get_world_size(), dist.get_rank()
    if world != 2:
        raise ValueError("split_pair_correspondence is the two-rank layout (target on rank 0, source on rank 1)")
    dev = graph.device if block is None else None
    if block is None:
        block = resident_block_tensor(torch, dev, dev.ctx.device)
    tdev = block.device
    n_own, m_own = int(block.shape[0]), int(block.shape[1])

    def on_stream():
        import contextlib

        return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    with on_stream():
        # shapes + eigenvalues: one small gather (16 eigenvalue slots are plenty: m <= 16 = the KNN's dimension limit)
        meta = torch.zeros(2 + 16, dtype=torch.float64, device=tdev)
        meta[0], meta[1] = n_own, m_own
        meta[2:2 + len(graph.eig_vals)] = torch.as_tensor(np.asarray(graph.eig_vals, dtype=np.float64)).to(tdev)
        metas_h = _all_gather_stack(dist, torch, meta).cpu().numpy()
        ns = [int(r[0]) for r in metas_h]
        ms = [int(r[1]) for r in metas_h]
        vals = [r[2:2 + m].copy() for r, m in zip(metas_h, ms)]
        rows_max, cols_max = max(ns), max(ms)
        # the ONE large exchange: the resident blocks, padded to a common shape, device to device
        pad = torch.zeros((rows_max, cols_max), dtype=torch.float64, device=tdev)
        pad[:n_own, :m_own] = block
        both = _all_gather_stack(dist, torch, pad)
        # identical samples on both ranks; the sampled eigenvector rows come off the gathered buffer, the sampled,
        # normalised points (graph.py:269-272) of each mesh from the rank that owns it (a 5000 x 3 gather)
        rng = np.random.default_rng(seed)
        rows = [sample_rows(ns[r], n_samples, rng) for r in range(world)]
        n_s = max(len(r) for r in rows)
        own_pts = np.asarray(graph.points)[rows[rank], :]
        own_pts = (own_pts - np.min(own_pts, axis=0)) / np.ptp(own_pts, axis=0)
        pts_pad = torch.zeros((n_s, 3), dtype=torch.float64, device=tdev)
        pts_pad[:len(own_pts)] = torch.as_tensor(own_pts).to(tdev)
        pts_all = _all_gather_stack(dist, torch, pts_pad)
        samples = []
        for r in range(world):
            sel = torch.as_tensor(rows[r], dtype=torch.int64).to(tdev)
            vecs = both[r].index_select(0, sel)[:, :ms[r]].cpu().numpy()
            samples.append(_SampledGraph(vals[r], vecs, pts_all[r, :len(rows[r])].cpu().numpy(),
                                         knn_ctx if knn_ctx is not None else (dev.ctx if dev is not None else None)))
    gt, gs = samples
    Q = eigsort(gt, gs, k, target_as_reference=True).sort_eigenmaps()  # replicated: k x k work on the samples
    w = Q[:k] * np.max((gs.eig_vals[:k], gt.eig_vals[:k]), axis=0)  # focusr.py:481-490
    w = np.exp(-(w**2) / (2 * np.mean(w) ** 2))
    (ct, st), (cs, ss) = gt._final_map, gs._final_map
    lo, hi = shard_rows(ns[1], world, rank)
    if knn_blocks is None:
        ctx = dev.ctx
        torch.cuda.current_stream(tdev).synchronize() if stream is None else stream.synchronize()
        part = ctx.knn1_blocks(both[0].data_ptr(), ns[0], cols_max, both[1].data_ptr() + 8 * lo * cols_max, hi - lo, cols_max,
                               ct[:k], st[:k] * w, cs[:k], ss[:k] * w)
    else:
        part = knn_blocks(both[0], ns[0], both[1][lo:hi], hi - lo, cols_max, ct[:k], st[:k] * w, cs[:k], ss[:k] * w)
    with on_stream():
        span = max(shard_rows(ns[1], world, r)[1] - shard_rows(ns[1], world, r)[0] for r in range(world))
        part_pad = torch.zeros(span, dtype=torch.int64, device=tdev)
        part_pad[:hi - lo] = torch.as_tensor(np.asarray(part, dtype=np.int64)).to(tdev)
        parts_h = _all_gather_stack(dist, torch, part_pad).cpu().numpy()  # int64 indices as int64
    idx = np.concatenate([parts_h[r, :shard_rows(ns[1], world, r)[1] - shard_rows(ns[1], world, r)[0]] for r in range(world)])
    return idx, Q, w
