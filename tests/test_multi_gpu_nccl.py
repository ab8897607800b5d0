"""The multi-GPU layouts of SURVEY.md 8e on REAL devices: one process per GPU, `torch.distributed` backend "nccl" (= RCCL
over xGMI), world size >= 2.  Every test here skips itself on a box with fewer devices than ranks (the build's boxes have
one); on a multi-GPU node `pytest -m gpu` runs them with no further arrangement:

* `test_split_pair_nccl_two_devices` - BASELINE config C4: target mesh on GPU 0, source mesh on GPU 1, ONE device-to-device
  all-gather of the resident eigenvector blocks on the library's own stream (`torch.cuda.ExternalStream`), ragged row counts
  (14 998 / 14 996) and column counts (5 / 9) padded on the device, eigsort replicated, KNN sharded by source rows; the result
  must be the reference's 14 996 correspondence indices of the bundled 15k pair (tests/golden/pair_15k.npz).
* `test_row_partitioned_solve_nccl` - BASELINE config C5's layout: ONE mesh's rows over 2 / 4 ranks, ghost zones, boundary
  rows exchanged device buffer -> all-gather -> device buffer every S steps; eigenpairs equal to the single-device solve.

What a one-GPU box CAN run of the same code is in tests/test_gpu_parity.py (`test_split_pair_device_to_device`: the two
ranks as threads; `test_rccl_all_gather_of_resident_block`: RCCL with a world of one), tests/test_rowpart.py (ranks sharing
the device over gloo, ranks as threads with the device-buffer exchange) and tests/test_parallel_gloo.py (CPU)."""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from test_parallel_gloo import _free_port  # noqa: E402


def _device_count():
    try:
        import torch

        return torch.cuda.device_count()  # (counting devices does not initialise the GPU in this process)
    except Exception:  # noqa: BLE001
        return 0


def _init(rank, world, port):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    return torch, dist


def _split_worker(rank, world, port, out_dir):
    torch, dist = _init(rank, world, port)
    from conftest import load_golden
    from pyfocusr_amd import Graph, PolyMesh, _hip
    from pyfocusr_amd.parallel import split_pair_correspondence

    p = load_golden("pair_15k")
    g_ = load_golden("target_mesh_15k" if rank == 0 else "source_mesh_15k")
    ctx = _hip.Context(rank)
    gr = Graph(PolyMesh(g_["points"], g_["faces"]), n_spectral_features=5, n_rand_samples=10**9, ctx=ctx, verbose=False)
    gr.get_graph_spectrum()
    assert gr.eig_vecs.shape[1] == (5 if rank == 0 else 9)  # ragged columns: the source mesh's widened solve
    stream = torch.cuda.ExternalStream(ctx.stream_ptr, device=torch.device("cuda", rank))
    idx, Q, w = split_pair_correspondence(dist, torch, gr, 5, 10**9, seed=3, stream=stream)
    np.testing.assert_allclose(Q, p["Q"], rtol=1e-5)
    np.testing.assert_allclose(w, p["spectral_weights"], rtol=1e-5)
    assert idx.dtype == np.int64 and len(idx) == 14996
    assert int(np.sum(idx != p["knn_idx_w"])) == 0
    # both ranks hold the same answer
    mine = torch.as_tensor(idx).to(torch.device("cuda", rank))
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    assert all(bool(torch.equal(b, mine)) for b in both)
    gr.device.close()
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, "ok%d" % rank), "w").close()


@pytest.mark.gpu
@pytest.mark.skipif(_device_count() < 2, reason="needs two MI355X (RCCL world of 2)")
def test_split_pair_nccl_two_devices(tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_split_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(2))


def _rowpart_worker(rank, world, port, out_dir, n, k, s):
    torch, dist = _init(rank, world, port)
    from pyfocusr_amd import Graph, _hip, rowpart
    from pyfocusr_amd.meshgen import blob_mesh

    ctx = _hip.Context(rank)
    m = blob_mesh(n, seed=9)
    full = _hip.DeviceLaplacian(m.points, m.faces, ctx=ctx)
    comm = rowpart.Comm(dist, torch)

    def make_local(S_local):
        return _hip.DeviceLaplacian(matrix=(S_local.indptr, S_local.indices, S_local.data), ctx=ctx)

    vals, vecs, own, stats, ops = rowpart.row_partitioned_eigs(m.points, m.faces, k, comm, make_local, s=s, device_graph=full,
                                                               device_exchange=True)
    # the same mesh on this rank's device alone
    one = Graph(m, n_spectral_features=k, n_rand_samples=10**9, norm_eig_vecs=False, ctx=ctx, verbose=False)
    one.get_graph_spectrum()
    np.testing.assert_allclose(vals, one.eig_vals[:k], rtol=1e-10)
    ref = one.eig_vecs[:, :k]
    sign = np.sign(comm.allreduce_sum(np.sum(vecs * ref[own], axis=0)))
    assert np.max(np.abs(vecs * sign - ref[own])) < 1e-7
    assert ops.exchanges >= stats.matvecs // s
    one.device.close()
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, "ok%d" % rank), "w").close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,k,s", [(2, 60000, 5, 16), (4, 250000, 5, 16)])
def test_row_partitioned_solve_nccl(tmp_path, world, n, k, s):
    if _device_count() < world:
        pytest.skip("needs %d MI355X (RCCL world of %d)" % (world, world))
    import torch.multiprocessing as mp

    mp.spawn(_rowpart_worker, args=(world, _free_port(), str(tmp_path), n, k, s), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` from a plain shell (no launcher, WORLD_SIZE unset) starts its N ranks itself, as children
    of a process that has made no GPU call: the launcher line of the contract (dry run: the command only)."""
    import json
    import subprocess

    env = dict(os.environ, PF_BENCH_SPAWN_DRYRUN="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
