"""Row-partitioned eigensolve of ONE large mesh across several GPUs (BASELINE config C5).

The reference has no counterpart (it runs scipy `eigs` on one core, `graph.py:357-389`); north_star asks for a
row-partitioned SpMV "reported only if it pays".  A Chebyshev step on a quarter of a 1M-vertex mesh is a ~5 us
kernel, a collective costs >= 10 us, so exchanging boundary values EVERY step (all-gather of x slices, halo
exchange) cannot pay.  What can pay is exchanging every `s` steps:

* rows are split into `world` contiguous chunks of the Morton order (compact patches of the surface);
* each rank also holds the `s` rings of rows around its chunk (ghost rows) and the principal submatrix of the
  symmetric operator S = G^1/2 (D - W) G^1/2 on chunk + ghosts;
* after an exchange every local row is exact; each application of the local operator invalidates one more ring
  from the outside (ring s first), so after `s` steps of the three-term recurrence the rank's OWN rows are
  still exact — then the ghost values of the two recurrence vectors are refreshed in ONE all-gather of the
  published boundary rows (a few hundred KB per rank) and the next `s` steps run without communication.
  A degree-145 filter application needs 10 exchanges at s = 16 instead of 145, at the price of ~13 % redundant
  rows (16 rings around a 500 x 500-vertex patch).
* inner products are taken over own rows (ghost rows are zeroed after every operator application) and summed
  with an all-reduce of the <= 49 coefficients of a Gram-Schmidt pass.

Everything above the device ops is the unchanged Krylov driver (`_krylov.filtered_eigs`): `RowPartitionedOps`
implements its `ops` interface on top of a per-rank local operator.  The collectives go through
`torch.distributed` ("nccl" = RCCL over xGMI on a GPU node, "gloo" in the CPU tests and in the one-GPU
rehearsal); exchanged values are staged through host memory in this version.

Status: correct by construction and tested with 2-3 ranks (CPU test double under gloo; ranks sharing one
MI355X); NOT measured on a multi-GPU node (none available to the build) — `bench.py --row-partition S` runs it.
"""
import numpy as np

from .parallel import all_gather_rows, shard_rows


def morton_order(points):
    """Permutation that sorts vertices along a 3-D Morton curve (10 bits per axis), stable."""
    p = np.asarray(points, dtype=np.float64)
    lo, ext = p.min(axis=0), np.ptp(p, axis=0)
    ext = np.where(ext > 0, ext, 1.0)
    q = np.minimum((p - lo) / ext * 1023.0, 1023.0).astype(np.uint64)

    def spread(v):
        v = v & np.uint64(0x3FF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x030000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x0300F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x030C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x09249249)
        return v

    code = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))
    return np.argsort(code, kind="stable")


class Layout(object):
    """What one rank holds: `local` = global ids of [own rows, ring 1, ..., ring s]; `n_own`; the principal
    submatrix `S_local` (CSR, local ids); `publish` = local ids (all own) whose values other ranks need, in the
    order they are published; `fill` = per owner rank q: (positions in q's published array, local ghost ids)."""

    def __init__(self, local, n_own, ring_ptr, S_local, publish, fill, owner_spans):
        self.local, self.n_own, self.ring_ptr = local, int(n_own), ring_ptr
        self.S_local, self.publish, self.fill, self.owner_spans = S_local, publish, fill, owner_spans

    @property
    def n_local(self):
        return len(self.local)


def ghost_rings(S, own, s):
    """Rows within `s` hops of the row set `own` (pattern of the symmetric CSR matrix S), ring by ring."""
    n = S.shape[0]
    seen = np.zeros(n, dtype=bool)
    seen[own] = True
    frontier, rings = np.asarray(own), []
    indptr, indices = S.indptr, S.indices
    for _ in range(int(s)):
        if len(frontier) == 0:
            break
        starts, ends = indptr[frontier], indptr[frontier + 1]
        total = int((ends - starts).sum())
        if total == 0:
            break
        # concatenated column indices of the frontier rows
        offs = np.repeat(starts - np.concatenate(([0], np.cumsum(ends - starts)[:-1])), ends - starts)
        cols = indices[np.arange(total) + offs]
        new = np.unique(cols[~seen[cols]])
        seen[new] = True
        rings.append(new)
        frontier = new
    return rings


def build_layout(S, order, world, rank, s):
    """First half of a rank's layout (everything that needs no other rank): own rows, ghost rings, local matrix.
    `order` = Morton permutation (same on every rank)."""
    n = S.shape[0]
    pos = np.empty(n, dtype=np.int64)
    pos[order] = np.arange(n)  # Morton position of every vertex
    spans = [shard_rows(n, world, r) for r in range(world)]
    lo, hi = spans[rank]
    own = order[lo:hi]
    rings = ghost_rings(S, own, s)
    rings = [r[np.argsort(pos[r], kind="stable")] for r in rings]
    local = np.concatenate([own] + rings) if rings else own.copy()
    ring_ptr = np.cumsum([len(own)] + [len(r) for r in rings])
    S_local = S[local][:, local].tocsr()
    S_local.sort_indices()
    return local, len(own), ring_ptr, S_local, spans, pos


def finish_layout(rank, world, local, n_own, ring_ptr, S_local, spans, pos, ghosts_by_rank):
    """Publication / fill index lists once every rank's ghost list is known."""
    owner_of = np.empty(len(pos), dtype=np.int64)
    for q, (a, b) in enumerate(spans):
        owner_of[a:b] = q
    owner_of_vertex = owner_of[pos]  # owner rank of every global vertex id
    # what each rank publishes: its own rows that are a ghost of somebody, in Morton order
    published = []
    for q in range(world):
        need = [g[owner_of_vertex[g] == q] for r, g in enumerate(ghosts_by_rank) if r != q]
        ids = np.unique(np.concatenate(need)) if need else np.zeros(0, dtype=np.int64)
        published.append(ids[np.argsort(pos[ids], kind="stable")])
    glob_to_local = np.full(len(pos), -1, dtype=np.int64)
    glob_to_local[local] = np.arange(len(local))
    publish = glob_to_local[published[rank]]
    fill = {}
    my_ghosts = local[n_own:]
    where = np.full(len(pos), -1, dtype=np.int64)
    for q in range(world):
        if q == rank:
            continue
        mine = my_ghosts[owner_of_vertex[my_ghosts] == q]
        if len(mine) == 0:
            continue
        where[published[q]] = np.arange(len(published[q]))
        fill[q] = (where[mine].copy(), glob_to_local[mine])
    return Layout(local, n_own, ring_ptr, S_local, publish, fill, spans)


def build_all_layouts(S, order, world, s):
    """Every rank's layout in one process (tests, and the reference for the distributed construction)."""
    parts = [build_layout(S, order, world, r, s) for r in range(world)]
    ghosts = [p[0][p[1]:] for p in parts]
    return [finish_layout(r, world, *parts[r][:4], parts[r][4], parts[r][5], ghosts) for r in range(world)]


class Comm(object):
    """The two collectives the solver needs, over `torch.distributed` (or none for world size 1)."""

    def __init__(self, dist=None, torch=None):
        self.dist, self.torch = dist, torch
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0

    def allreduce_sum(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        if self.world == 1:
            return arr
        dev = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
        t = self.torch.from_numpy(arr.copy()).to(dev)
        self.dist.all_reduce(t)
        return t.cpu().numpy()

    def allgather_ragged(self, arr):
        """list over ranks of each rank's 2-D float64 array (row counts may differ)."""
        if self.world == 1:
            return [np.ascontiguousarray(arr, dtype=np.float64)]
        return all_gather_rows(self.dist, self.torch, arr)

    def allgather_device(self, send, recv, stream, local):
        """recv[q] <- rank q's `send` (CUDA tensors of equal shape): one RCCL all-gather enqueued on the library's
        stream (`stream` = torch.cuda.ExternalStream over it), i.e. after the gather kernel and before the scatter
        kernel, with no host synchronisation."""
        if stream is None:
            local.sync()
            self.dist.all_gather_into_tensor(recv, send)
            self.torch.cuda.synchronize()
            return
        with self.torch.cuda.stream(stream):
            self.dist.all_gather_into_tensor(recv, send)


class ThreadComm(object):
    """The same two collectives between THREADS of one process (one "rank" per thread, each with its own HIP
    stream on the same device): the harness the tests use to run every rank of the solve on the one-GPU box,
    including the device-buffer exchange that RCCL serves on a multi-GPU node."""

    class Shared(object):
        def __init__(self, world):
            import threading

            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank, torch=None):
        self.shared, self.rank, self.world, self.torch = shared, rank, shared.world, torch

    def _exchange(self, item):
        sh = self.shared
        sh.slots[self.rank] = item
        sh.barrier.wait()
        items = list(sh.slots)
        sh.barrier.wait()  # nobody overwrites its slot before everybody has read
        return items

    def allreduce_sum(self, arr):
        return np.sum(self._exchange(np.ascontiguousarray(arr, dtype=np.float64)), axis=0)

    def allgather_ragged(self, arr):
        return self._exchange(np.ascontiguousarray(arr, dtype=np.float64))

    def allgather_device(self, send, recv, stream, local):
        """recv[q] <- rank q's `send` (device tensors, equal shapes).  Threads stand in for ranks, so the ordering
        RCCL gets from the stream is made with host synchronisation here."""
        local.sync()  # this rank's gather kernel has filled `send`
        for q, t in enumerate(self._exchange(send)):
            recv[q].copy_(t)
        self.torch.cuda.synchronize()
        self.shared.barrier.wait()  # every copy out of `send` is done before its owner refills it


class RowPartitionedOps(object):
    """The `ops` object of `_krylov.filtered_eigs` for a row-partitioned operator.

    `local` is a per-rank operator object over chunk + ghost rows (a `_hip.DeviceLaplacian` built from the local
    submatrix, or the CPU test double) offering the single-device ops plus `op_step`, `axpy`, `rows_*`."""

    def __init__(self, local, layout, comm, n_global, s, device_exchange=False):
        self.local, self.layout, self.comm = local, layout, comm
        self.n, self.n_isolated, self.s = int(n_global), 0, int(s)
        self._pub = local.rows_create(layout.publish)
        self._ghost = local.rows_create(np.arange(layout.n_own, layout.n_local, dtype=np.int64))
        self._fill = {q: (src, local.rows_create(dst)) for q, (src, dst) in layout.fill.items()}
        self.exchanges = 0
        self._pending = None
        self._dev = None
        if device_exchange and comm.world > 1:
            # Boundary values travel device buffer -> collective -> device buffer (RCCL over xGMI on a GPU node):
            # ONE gather launch for both recurrence vectors, one all-gather, ONE scatter launch for every owner —
            # all enqueued on the library's own stream (torch sees it as an ExternalStream), no host synchronisation.
            torch = comm.torch
            if not torch.cuda.is_available():
                raise RuntimeError("device_exchange needs torch.cuda; if a GPU is present, import torch BEFORE pyfocusr_amd "
                                   "loads libpyfocusr_hip.so (both must share one copy of the HIP runtime)")
            sizes = comm.allreduce_sum(np.eye(comm.world)[comm.rank] * len(layout.publish))
            n_pub = int(max(np.max(sizes), 1))
            dev = torch.device("cuda", torch.cuda.current_device())
            dst = np.concatenate([d for _, (_, d) in sorted(layout.fill.items())]) if layout.fill else np.zeros(0, np.int64)
            off = np.concatenate([q * 2 * n_pub + src for q, (src, _) in sorted(layout.fill.items())]) \
                if layout.fill else np.zeros(0, np.int64)
            rows = local.rows_create(dst)
            local.rows_set_sources(rows, off)
            stream = torch.cuda.ExternalStream(local.ctx.stream_ptr) if getattr(local, "ctx", None) is not None else None
            self._dev = dict(torch=torch, n_pub=n_pub, rows=rows, stream=stream,
                             send=torch.zeros((2, n_pub), dtype=torch.float64, device=dev),
                             recv=torch.zeros((comm.world, 2, n_pub), dtype=torch.float64, device=dev))

    # ---- communication
    def _refresh_ghosts_device(self, slots):
        d, loc = self._dev, self.local
        second = slots[1] if len(slots) > 1 else None
        loc.rows_gather2_dev(slots[0], second, self._pub, d["send"].data_ptr(), d["n_pub"])
        self.comm.allgather_device(d["send"], d["recv"], d["stream"], loc)
        loc.rows_scatter2_dev(slots[0], second, d["rows"], d["recv"].data_ptr(), d["n_pub"])
        self.exchanges += 1

    def _refresh_ghosts(self, slots):
        """Owners publish their boundary rows of `slots`; every rank overwrites its ghost rows."""
        if self.comm.world == 1:
            return
        if self._dev is not None:
            return self._refresh_ghosts_device(slots)
        mine = np.stack([self.local.rows_gather(sl, self._pub) for sl in slots], axis=1)  # (n_pub, len(slots))
        parts = self.comm.allgather_ragged(mine)
        for q, (src, dst) in self._fill.items():
            for j, sl in enumerate(slots):
                self.local.rows_scatter(sl, dst, np.ascontiguousarray(parts[q][src, j]))
        self.exchanges += 1

    def _zero_ghosts(self, slot):
        self.local.rows_fill(slot, self._ghost, 0.0)

    # ---- the ops interface of the Krylov driver
    def ws_ensure(self, nslots):
        self.local.ws_ensure(nslots + 2)  # two scratch slots for the chunked recurrence
        self._scratch = nslots

    def start_vector(self, slot, seed):
        self.local.start_vector(slot, seed + 7919 * self.comm.rank)
        self._zero_ghosts(slot)

    def lock_null_vectors(self):
        return 0

    def copy(self, src, dst, count):
        self.local.copy(src, dst, count)

    def scale(self, slot, alpha):
        self.local.scale(slot, alpha)

    def combine(self, src_first, m, Y, dst_first):
        self.local.combine(src_first, m, Y, dst_first)

    def dots(self, w, first, count):
        return self.comm.allreduce_sum(self.local.dots(w, first, count))

    def orth(self, w, first, count):
        """Two classical Gram-Schmidt passes of slot `w` against slots [first, first+count); returns the summed
        coefficients and the norm of what is left (the caller normalises)."""
        h = np.zeros(count)
        for _ in range(2):
            if count:
                c = self.dots(w, first, count)
                self.local.axpy(w, first, count, -c)
                h += c
        nrm = float(np.sqrt(max(self.comm.allreduce_sum(self.local.dots(w, w, 1))[0], 0.0)))
        return h, nrm

    def orth_begin(self, w, first, count, normalize=True):
        h, nrm = self.orth(w, first, count)
        if normalize and nrm > 0:
            self.local.scale(w, 1.0 / nrm)
        self._pending = (h, nrm)

    def orth_end(self):
        out, self._pending = self._pending, None
        return out

    def sync(self):
        fn = getattr(self.local, "sync", None)
        if fn is not None:
            fn()

    def resnorm(self, ax, x, lam):
        r = self.local.resnorm(ax, x, lam)
        return float(np.sqrt(self.comm.allreduce_sum(np.array([r * r]))[0]))

    def spmv(self, src, dst):
        self._refresh_ghosts([src])
        self.local.op_step(src, None, dst, -1.0, 0.0, 0.0)  # dst = A src
        self._zero_ghosts(dst)

    def cheb(self, src, dst, p, c, e, rho=1.0):
        """dst = T_p((c - A)/e) src / rho^p in chunks of `s` steps between ghost refreshes.  Same recurrence and
        coefficients as the single-device `pf_cheb` (first step alpha = 1/(e rho), then 2/(e rho) and 1/rho^2)."""
        loc, s = self.local, self.s
        a, b = self._scratch, self._scratch + 1
        loc.copy(src, b, 1)  # y_0 (its ghost rows are refreshed below)
        prev, cur, done = a, b, 0
        while done < p:
            self._refresh_ghosts([cur] if done == 0 else [cur, prev])
            n_steps = min(s, p - done)
            prev, cur = loc.cheb_steps(prev, cur, done + 1, n_steps, c, e, rho)
            done += n_steps
        loc.copy(cur, dst, 1)
        self._zero_ghosts(dst)


def symmetric_operator(rowptr, col, w, deg):
    """S = G^1/2 (D - W) G^1/2 with G = 1/(deg + 1e-8) (graph.py:216-226), as scipy CSR, from CSR(W) and deg."""
    from scipy import sparse

    n = len(deg)
    W = sparse.csr_matrix((w, col, rowptr), shape=(n, n))
    sg = np.sqrt(1.0 / (deg + 1e-8))
    S = sparse.diags(deg * sg * sg) - sparse.diags(sg) @ W @ sparse.diags(sg)
    S = S.tocsr()
    S.sort_indices()
    return S, sg


def row_partitioned_eigs(points, faces, k, comm, make_local, s=16, device_graph=None, verbose=False, timing=None,
                         device_exchange=False, **solver_kw):
    """Lowest `k` non-null eigenpairs of the mesh Laplacian with the rows split over `comm.world` ranks.

    `make_local(S_local)` builds the rank's local operator object; `device_graph` (a `DeviceLaplacian` of the
    whole mesh, built by the caller on this rank's device) supplies CSR(W) and deg.  Returns
    (eig_vals (k,), eig_vecs_own (n_own, k) unit-norm eigenvectors of L on this rank's rows, own global ids,
    stats, ops)."""
    from . import _krylov

    if device_graph.n_isolated or device_graph.n_components != 1:
        raise NotImplementedError("row-partitioned solve needs ONE connected component and no unreferenced vertices "
                                  "(this mesh: %d components, %d isolated vertices); use the single-device solve"
                                  % (device_graph.n_components, device_graph.n_isolated))
    d = device_graph.download()
    S, sg = symmetric_operator(d["rowptr"], d["colidx"], d["w"], d["deg"])
    deg = d["deg"]
    if abs(S - S.T).max() > 1e-12 * abs(S).max():
        raise NotImplementedError("row-partitioned solve needs a symmetric adjacency (no one-way edges)")
    order = morton_order(points)
    world, rank = comm.world, comm.rank
    local, n_own, ring_ptr, S_local, spans, pos = build_layout(S, order, world, rank, s)
    ghosts = comm.allgather_ragged(local[n_own:].astype(np.float64)[:, None])
    ghosts = [g[:, 0].astype(np.int64) for g in ghosts]
    layout = finish_layout(rank, world, local, n_own, ring_ptr, S_local, spans, pos, ghosts)
    ops = RowPartitionedOps(make_local(layout.S_local), layout, comm, len(deg), s, device_exchange=device_exchange)
    import time

    t0 = time.perf_counter()
    vals, first, stats = _krylov.filtered_eigs(ops, k + 1, True, verbose=verbose, **solver_kw)  # + the null vector
    if timing is not None:
        timing["solve"] = time.perf_counter() - t0
    keep = vals > 1e-10  # graph.py:381
    vals = vals[keep][:k]
    slots = (np.arange(len(keep))[keep] + first)[:k]
    own_local = np.arange(n_own)
    vecs = np.stack([ops.local.rows_gather(int(sl), ops.local.rows_create(own_local)) for sl in slots], axis=1)
    vecs = vecs * sg[local[:n_own], None]  # eigenvectors of L = G^1/2 (eigenvectors of S)
    nrm = np.sqrt(comm.allreduce_sum(np.sum(vecs * vecs, axis=0)))
    vecs = vecs / nrm[None, :]
    return vals, vecs, local[:n_own], stats, ops
