"""World-size-2 CPU test (gloo) of the multi-GPU layout helpers in pyfocusr_amd/parallel.py:
the spectral-coordinate all-gather (ragged row counts, as for the 14998/14996-vertex pair) and
the query-sharded correspondence that follows it.  On the GPU box the same code runs over
RCCL (backend "nccl")."""
import os
import socket
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import reference_port as orc
    from pyfocusr_amd.parallel import all_gather_rows, gather_spectral, shard_rows

    rng = np.random.default_rng(rank)
    n = 700 + 13 * rank  # ragged
    vals = np.sort(rng.uniform(1e-4, 1e-3, 5 + rank))  # source may carry extra columns (widening)
    vecs = rng.uniform(-0.5, 0.5, size=(n, len(vals)))
    pts = rng.normal(size=(n, 3))
    gvals, gvecs, gpts = gather_spectral(dist, torch, vals, vecs, pts)
    assert len(gvals) == world
    assert np.array_equal(gvals[rank], vals) and np.array_equal(gvecs[rank], vecs) and np.array_equal(gpts[rank], pts)
    other = 1 - rank
    exp = np.random.default_rng(other)
    n_o = 700 + 13 * other
    vals_o = np.sort(exp.uniform(1e-4, 1e-3, 5 + other))
    vecs_o = exp.uniform(-0.5, 0.5, size=(n_o, len(vals_o)))
    assert np.array_equal(gvals[other], vals_o) and np.array_equal(gvecs[other], vecs_o)

    # query-sharded correspondence: each rank matches its slice of the source rows against the full target
    tgt, src = gvecs[0][:, :5], gvecs[1][:, :5]
    lo, hi = shard_rows(len(src), world, rank)
    part = orc.knn1(tgt, src[lo:hi]).astype(np.float64)[:, None]
    parts = all_gather_rows(dist, torch, part)
    idx = np.concatenate([p[:, 0] for p in parts]).astype(np.int64)
    assert np.array_equal(idx, orc.knn1(tgt, src))
    covered = [shard_rows(len(src), world, r) for r in range(world)]
    assert covered[0][0] == 0 and covered[-1][1] == len(src) and covered[0][1] == covered[1][0]
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, "ok%d" % rank), "w").close()


class _HostKnn(object):
    """Stands in for the device context in the CPU run of `split_pair_correspondence`."""

    def knn1(self, ref, qry):
        from oracle import reference_port as orc

        return orc.knn1(ref, qry)


def _split_data(rank):
    rng = np.random.default_rng(100 + rank)
    n, m = 1500 + 21 * rank, 5 + 2 * rank  # ragged rows AND columns (the source of the 15k pair carries 9 columns)
    t = rng.uniform(0, 1, size=(n, 3))
    base = np.stack([np.cos(2 * np.pi * t[:, 0]), np.sin(2 * np.pi * t[:, 1]), t[:, 2] - 0.5, t[:, 0] * t[:, 1] - 0.25,
                     np.cos(3 * t[:, 2])], axis=1) * 0.45
    vecs = np.concatenate([base, rng.uniform(-0.5, 0.5, size=(n, m - 5))], axis=1)
    if rank == 1:
        vecs[:, [1, 3]] = -vecs[:, [3, 1]]  # the source's maps arrive permuted and flipped: eigsort has to undo it
    vals = np.sort(rng.uniform(1e-4, 1e-3, m)) + 1e-5 * np.arange(m)
    return vals, vecs, t


def _split_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import reference_port as orc
    from pyfocusr_amd.eigsort import eigsort
    from pyfocusr_amd.parallel import _SampledGraph, sample_rows, split_pair_correspondence

    vals, vecs, pts = _split_data(rank)

    class G(object):
        pass

    g = G()
    g.eig_vals, g.points = vals, pts

    def knn_blocks(ref, n_ref, qry, n_qry, stride, ct, st, cs, ss):
        r, q = ref.numpy()[:n_ref], qry.numpy()[:n_qry]
        return orc.knn1(r[:, ct] * st, q[:, cs] * ss)

    k, n_samples = 5, 400
    idx, Q, w = split_pair_correspondence(dist, torch, g, k, n_samples, seed=7, block=torch.from_numpy(vecs),
                                          knn_blocks=knn_blocks, knn_ctx=_HostKnn())
    # the same pipeline on one process with everything in hand
    data = [_split_data(r) for r in range(world)]
    rng = np.random.default_rng(7)
    rows = [sample_rows(len(d[1]), n_samples, rng) for d in data]
    graphs = []
    for d, rr in zip(data, rows):
        p = d[2][rr]
        graphs.append(_SampledGraph(d[0], d[1][rr], (p - p.min(axis=0)) / np.ptp(p, axis=0), _HostKnn()))
    Q_ref = eigsort(graphs[0], graphs[1], k, target_as_reference=True).sort_eigenmaps()
    assert np.array_equal(Q, Q_ref)
    cs, ss = graphs[1]._final_map
    assert not np.array_equal(cs[:k], np.arange(k)) and np.any(ss[:k] < 0)  # a real permutation with flips
    w_ref = Q_ref[:k] * np.max((data[1][0][:k], data[0][0][:k]), axis=0)
    w_ref = np.exp(-(w_ref**2) / (2 * np.mean(w_ref) ** 2))
    assert np.array_equal(w, w_ref)
    expect = orc.knn1(data[0][1][:, :k] * w_ref, (data[1][1][:, cs[:k]] * ss[:k]) * w_ref)
    assert idx.dtype == np.int64 and np.array_equal(idx, expect)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, "split_ok%d" % rank), "w").close()


def test_split_pair_correspondence_gloo(tmp_path):
    """The control flow of the device-to-device split pair (BASELINE config C4) with CPU tensors over gloo: ragged
    blocks, replicated eigsort from samples, query-sharded KNN with the flips / permutation folded in, int64 gather."""
    import torch.multiprocessing as mp

    mp.spawn(_split_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "split_ok0") and os.path.exists(tmp_path / "split_ok1")


def test_gather_and_sharded_correspondence_gloo(tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_shard_rows_balanced():
    from pyfocusr_amd.parallel import shard_rows

    for n in (0, 1, 7, 250000, 14996):
        for world in (1, 2, 3, 8):
            spans = [shard_rows(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
