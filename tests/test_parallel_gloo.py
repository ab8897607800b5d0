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
