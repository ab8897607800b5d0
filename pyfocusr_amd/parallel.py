"""Multi-GPU layout of the hot path: one process per GPU.

The per-mesh work (assembly + eigensolve) of a pair shards at mesh granularity —
target on rank 0, source on rank 1 — with ONE exchange: an all-gather of the
spectral coordinates (n x k float64, 10 MB at 250k x 5) so that eigsort (k x k,
replicated) and the query-sharded KNN can run.  The collective goes through
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests); tensors of different length are padded to the longest.
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
