"""Row-partitioned eigensolve: host-side layout logic, and the whole distributed solve
under gloo with the CPU test double as the per-rank local operator (world sizes 2 and 3); on the GPU box the same
code runs with `_hip.DeviceLaplacian` local operators (`-m gpu`, ranks sharing the one MI355X)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (first: torch and libpyfocusr_hip.so must share ONE HIP runtime — whichever copy of
#                libamdhip64 is loaded first serves both, and torch.cuda only comes up on its own copy)

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _mesh_operator(n, seed):
    from oracle import reference_port as orc
    from pyfocusr_amd import rowpart
    from pyfocusr_amd.meshgen import blob_mesh

    m = blob_mesh(n, seed=seed)
    W, deg, d_inv, L = orc.graph_matrices(m.points, m.faces)
    Wc = W.tocsr()
    Wc.sort_indices()
    S, sg = rowpart.symmetric_operator(Wc.indptr, Wc.indices, Wc.data, deg)
    return m, L, S, sg


def test_layouts_cover_and_ghost_zone_is_exact():
    """Own sets partition the rows; ring r is exactly the set at graph distance r; and `s` steps of the three-term
    recurrence on chunk + s rings reproduce the global recurrence on the own rows (the property the scheme rests on)."""
    from _numpy_ops import MatrixOps
    from pyfocusr_amd import rowpart
    from scipy.sparse.csgraph import shortest_path

    m, L, S, sg = _mesh_operator(1500, 2)
    order = rowpart.morton_order(m.points)
    assert sorted(order.tolist()) == list(range(1500))
    for world, s in ((2, 3), (3, 5), (5, 2)):
        layouts = rowpart.build_all_layouts(S, order, world, s)
        owned = np.concatenate([lay.local[: lay.n_own] for lay in layouts])
        assert sorted(owned.tolist()) == list(range(1500))
        pattern = (S != 0).astype(np.float64)
        for lay in layouts:
            dist = shortest_path(pattern, unweighted=True, indices=lay.local[: lay.n_own]).min(axis=0)
            for r in range(len(lay.ring_ptr) - 1):
                ring = lay.local[lay.ring_ptr[r]: lay.ring_ptr[r + 1]]
                assert np.all(dist[ring] == r + 1)
            assert abs(lay.S_local - S[lay.local][:, lay.local]).max() == 0
            assert np.all(lay.publish < lay.n_own)
            for q, (src, dst) in lay.fill.items():
                assert np.all(dst >= lay.n_own)
                assert np.array_equal(layouts[q].local[layouts[q].publish][src], lay.local[dst])
        # s steps without communication are exact on the own rows
        x = np.random.default_rng(0).standard_normal(1500)
        y0, y1 = x, (x - S @ x)
        for _ in range(s - 1):
            y0, y1 = y1, 2.0 * (y1 - S @ y1) - y0
        for lay in layouts:
            o = MatrixOps(lay.S_local)
            o.ws_ensure(3)
            o.ws[:, 0] = x[lay.local]
            o.op_step(0, None, 1, 1.0, 1.0, 0.0)
            prev, cur = 0, 1
            for _ in range(s - 1):
                o.op_step(cur, prev, prev, 2.0, 1.0, 1.0)
                prev, cur = cur, prev
            np.testing.assert_allclose(o.ws[: lay.n_own, cur], y1[lay.local[: lay.n_own]], rtol=0, atol=1e-12 * np.abs(y1).max())


def _worker(rank, world, port, out_dir, s):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _numpy_ops import MatrixOps
    from oracle import reference_port as orc
    from pyfocusr_amd import _krylov, rowpart

    m, L, S, sg = _mesh_operator(3000, 7)
    k = 4
    comm = rowpart.Comm(dist, torch)
    order = rowpart.morton_order(m.points)
    local, n_own, ring_ptr, S_local, spans, pos = rowpart.build_layout(S, order, world, rank, s)
    ghosts = [g[:, 0].astype(np.int64) for g in comm.allgather_ragged(local[n_own:].astype(np.float64)[:, None])]
    layout = rowpart.finish_layout(rank, world, local, n_own, ring_ptr, S_local, spans, pos, ghosts)
    ops = rowpart.RowPartitionedOps(MatrixOps(layout.S_local), layout, comm, S.shape[0], s)
    vals, first, stats = _krylov.filtered_eigs(ops, k + 1, True)
    vals = vals[:k]
    ref_vals, ref_vecs = orc.canonicalize(*orc.recursive_eig(L, k + 1, k))
    np.testing.assert_allclose(vals, ref_vals, rtol=1e-8)
    own = layout.local[:n_own]
    vecs = np.stack([ops.local.rows_gather(first + j, np.arange(n_own)) for j in range(k)], axis=1) * sg[own, None]
    nrm = np.sqrt(comm.allreduce_sum(np.sum(vecs * vecs, axis=0)))
    vecs = vecs / nrm
    sign = np.sign(comm.allreduce_sum(np.sum(vecs * ref_vecs[own], axis=0)))
    assert np.max(np.abs(vecs * sign - ref_vecs[own])) < 1e-7
    assert ops.exchanges > 0 and stats.matvecs > 0
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, "ok%d" % rank), "w").close()


@pytest.mark.parametrize("world,s", [(2, 6), (3, 16)])
def test_row_partitioned_solve_gloo(tmp_path, world, s):
    import torch.multiprocessing as mp

    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), s), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))


def _gpu_worker(rank, world, port, out_dir, n, k, s):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # ranks share GPU 0; collectives on the host
    from oracle import reference_port as orc
    from pyfocusr_amd import _hip, rowpart
    from pyfocusr_amd.meshgen import blob_mesh

    ctx = _hip.Context(0)
    m = blob_mesh(n, seed=9)
    full = _hip.DeviceLaplacian(m.points, m.faces, ctx=ctx)
    comm = rowpart.Comm(dist, torch)

    def make_local(S_local):
        return _hip.DeviceLaplacian(matrix=(S_local.indptr, S_local.indices, S_local.data), ctx=ctx)

    vals, vecs, own, stats, ops = rowpart.row_partitioned_eigs(m.points, m.faces, k, comm, make_local, s=s, device_graph=full)
    W, deg, d_inv, L = orc.graph_matrices(m.points, m.faces)
    ref_vals, ref_vecs = orc.canonicalize(*orc.recursive_eig(L, k + 1, k))
    np.testing.assert_allclose(vals, ref_vals, rtol=1e-8)
    sign = np.sign(comm.allreduce_sum(np.sum(vecs * ref_vecs[own], axis=0)))
    assert np.max(np.abs(vecs * sign - ref_vecs[own])) < 1e-7
    # true residual of the assembled eigenvectors against the oracle's L (needs all rows: gather)
    parts = comm.allgather_ragged(np.concatenate([own[:, None].astype(np.float64), vecs], axis=1))
    allv = np.zeros((n, k))
    for p in parts:
        allv[p[:, 0].astype(np.int64)] = p[:, 1:]
    R = L @ allv - allv * vals[None, :]
    assert np.max(np.linalg.norm(R, axis=0)) < 1e-9
    assert ops.exchanges >= stats.matvecs // s  # one refresh per s steps (plus the Rayleigh-Ritz products)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, "ok%d" % rank), "w").close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,k,s", [(2, 20000, 5, 8), (3, 60000, 4, 16)])
def test_row_partitioned_solve_on_device(tmp_path, world, n, k, s):
    """Ranks share the one MI355X (gloo for the collectives): every rank holds chunk + ghost rows as its own
    device graph; eigenpairs equal the oracle's."""
    import torch.multiprocessing as mp

    mp.spawn(_gpu_worker, args=(world, _free_port(), str(tmp_path), n, k, s), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))


def _thread_ranks(world, target):
    """Run `target(rank, comm_shared)` in `world` threads; re-raise the first failure."""
    import threading

    from pyfocusr_amd import rowpart

    shared = rowpart.ThreadComm.Shared(world)
    errors = []

    def run(rank):
        try:
            target(rank, shared)
        except BaseException as exc:  # noqa: BLE001
            errors.append(exc)
            shared.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]


def test_row_partitioned_solve_threads_cpu():
    """The in-process communicator (threads as ranks) with the CPU test double: same answer as under gloo."""
    from _numpy_ops import MatrixOps
    from oracle import reference_port as orc
    from pyfocusr_amd import _krylov, rowpart

    m, L, S, sg = _mesh_operator(2500, 4)
    ref_vals, _ = orc.canonicalize(*orc.recursive_eig(L, 4, 3))
    order = rowpart.morton_order(m.points)
    layouts = rowpart.build_all_layouts(S, order, 3, 5)
    out = {}

    def rank_main(rank, shared):
        comm = rowpart.ThreadComm(shared, rank)
        ops = rowpart.RowPartitionedOps(MatrixOps(layouts[rank].S_local), layouts[rank], comm, S.shape[0], 5)
        vals, first, stats = _krylov.filtered_eigs(ops, 4, True)
        out[rank] = vals[:3]

    _thread_ranks(3, rank_main)
    for r in range(3):
        np.testing.assert_allclose(out[r], ref_vals, rtol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("device_exchange", [False, True])
def test_row_partitioned_solve_threads_on_device(device_exchange):
    """Three ranks as threads of one process, one HIP stream each on the one MI355X.  With `device_exchange` the
    boundary rows go device buffer -> collective -> device buffer (torch CUDA tensors; RCCL's role on a GPU node is
    played by device-to-device copies here) — the path a multi-GPU node runs, minus the RCCL call itself."""
    import torch

    from oracle import reference_port as orc
    from pyfocusr_amd import _hip, rowpart
    from pyfocusr_amd.meshgen import blob_mesh

    n, k, s, world = 40000, 5, 12, 3
    m = blob_mesh(n, seed=12)
    W, deg, d_inv, L = orc.graph_matrices(m.points, m.faces)
    ref_vals, ref_vecs = orc.canonicalize(*orc.recursive_eig(L, k + 1, k))
    out = {}

    def rank_main(rank, shared):
        ctx = _hip.Context(0)
        comm = rowpart.ThreadComm(shared, rank, torch=torch)
        full = _hip.DeviceLaplacian(m.points, m.faces, ctx=ctx)

        def make_local(S_local):
            return _hip.DeviceLaplacian(matrix=(S_local.indptr, S_local.indices, S_local.data), ctx=ctx)

        vals, vecs, own, stats, ops = rowpart.row_partitioned_eigs(m.points, m.faces, k, comm, make_local, s=s,
                                                                   device_graph=full, device_exchange=device_exchange)
        out[rank] = (vals, vecs, own, ops.exchanges)

    _thread_ranks(world, rank_main)
    allv = np.zeros((n, k))
    for r in range(world):
        vals, vecs, own, exchanges = out[r]
        np.testing.assert_allclose(vals, ref_vals, rtol=1e-8)
        allv[own] = vecs
        assert exchanges > 0
    sign = np.sign(np.sum(allv * ref_vecs, axis=0))
    assert np.max(np.abs(allv * sign - ref_vecs)) < 1e-7
    R = L @ allv - allv * ref_vals[None, :]
    assert np.max(np.linalg.norm(R, axis=0)) < 1e-9
