#!/usr/bin/env python3
"""Benchmark of the spectral hot path on MI355X (contract: see the build brief).

One "step" = one pass of the hot path over one synthetic mesh pair (BASELINE.json config C3:
250k-vertex closed-manifold blobs, seeds 2r / 2r+1 on rank r, k = 5):
    for target and source:  Laplacian assembly (inputs resident in HBM)  ->  lowest-k non-null
                            eigenpairs (Chebyshev-filtered Krylov-Schur)  ->  normalised eigenvectors
    eigsort (sign/order, 3-D NN on 5000 samples)  ->  weighted spectral coordinates
    1-NN of every source vertex among the target vertices (d = k)
metric value = eigenpairs per second = (2 k) / step time, whole job (sum over ranks).

N > 1 ranks: every rank owns an independent mesh pair (the unit that shards with no data-path
collective: template-to-many-targets pipelines) -> weak scaling.  With exactly 2 ranks the C4
layout is measured in addition (one pair split target/source over the two GPUs, RCCL all-gather
of the spectral coordinates, query-sharded KNN) and reported under "split_pair".
"""
import argparse
import json
import os
import sys
import time

# One process per GPU: the host side of a step is small dense algebra (<= 49 x 49) and numpy element-wise work; BLAS /
# OpenMP thread pools would only spin against the other ranks' on a shared host (torchrun sets OMP_NUM_THREADS=1 itself
# for N > 1; the same here for N = 1 keeps the per-rank figure comparable).
for _var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_var, "1")

import numpy as np  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

METRIC = "lowest-k eigenpairs/sec + KNN-correspondence wall-clock, 250k-vertex mesh pair k=5"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md, HBM)
# LDS peak (MI355X_MICROARCH.md, LDS table): ds_read_b64 conflict-free = 256 B per clock and CU; 256 CUs; 2.4 GHz
LDS_PEAK_GBS = 256.0 * 256 * 2.4
PMC_SUMMARY = os.path.join("profiles", "r04_pmc_summary.json")
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak (MI355X_MICROARCH.md; SURVEY 8d: what bounds the exact KNN)
TIMING_STRIDE = 4  # filter applications per HIP event pair


def spmv_algorithmic_bytes(n, nnz_l):
    """y = L x moves 12*nnz + 20*n + 4 bytes (f64 values, i32 columns,
    row pointers, x once, y once), nnz counted with the diagonal."""
    return 12 * nnz_l + 20 * n + 4


def hot_path_step(ctxs, mesh_t, mesh_s, k, n_samples, timers, keep_graphs=False):
    """Same calls as Focusr.__init__ + align_maps (focusr.py:134-170, 514-545) without ICP/CPD."""
    from pyfocusr_amd import Graph, eigsort
    from pyfocusr_amd.graph import build_devices, compute_spectra, spectral_knn

    ctx = ctxs[0]
    t0 = time.perf_counter()
    graphs = [Graph(mesh, n_spectral_features=k, n_rand_samples=n_samples, ctx=c, verbose=False)
              for mesh, c in zip((mesh_t, mesh_s), ctxs)]
    ta = time.perf_counter()
    build_devices(graphs)  # assembly from the resident meshes (one context: the two meshes side by side on two streams)
    tb = time.perf_counter()
    for c in ctxs:
        c.sync()
    t1 = time.perf_counter()
    if "assembly_detail" in timers:  # (PF_BENCH_DETAIL=1: where the assembly stage's time is, host side)
        for key, dt in (("graph_objects", ta - t0), ("build_devices", tb - ta), ("sync", t1 - tb)):
            timers["assembly_detail"][key] = timers["assembly_detail"].get(key, 0.0) + dt
    compute_spectra(graphs)  # target and source concurrently, one HIP stream each
    t2 = time.perf_counter()
    timers["assembly"] += t1 - t0
    timers["eigensolve"] += t2 - t1
    gt, gs = graphs
    t0 = time.perf_counter()
    Q = eigsort(gt, gs, k, target_as_reference=True).sort_eigenmaps()
    w = Q[:k] * np.max((gs.eig_vals[:k], gt.eig_vals[:k]), axis=0)  # focusr.py:481-490
    w = np.exp(-(w**2) / (2 * np.mean(w) ** 2))
    t1 = time.perf_counter()
    # focusr.py:351-353 on eig_vecs[:, :k] * w of both meshes, read where the eigensolves left them (HBM); what
    # Focusr.get_initial_correspondences does.  (Two contexts, --streams 2: the host arrays, as before.)
    idx = spectral_knn(gt, gs, k, w)
    if idx is None:
        idx = ctx.knn1(gt.eig_vecs[:, :k] * w[None, :], gs.eig_vecs[:, :k] * w[None, :])
    t2 = time.perf_counter()
    timers["eigsort"] += t1 - t0
    timers["knn"] += t2 - t1
    timers["matvecs"] += gt.eigs_stats.matvecs + gs.eigs_stats.matvecs
    res = max(gt.eigs_stats.residuals.max(), gs.eigs_stats.residuals.max())
    for g in graphs:
        g.device.close()
    if keep_graphs:
        return graphs
    return idx, res, (gt.device.nnz_l, gs.device.nnz_l), (gt.eig_vecs, gs.eig_vecs, w, gt.eig_vals, gs.eig_vals)


def cpu_baseline(meshes, k, coords, gpu_vals, gpu_idx, eigs_runs=3):
    """The oracle (reference calls restated: scipy eigs shift-invert + KDTree, 1 thread, exactly as the reference
    issues them: graph.py:372, focusr.py:351-353) on the WHOLE workload of one step, as BASELINE.md section 3 asks:
    both meshes' assembly and eigensolve (one untimed warm-up solve, then the median of `eigs_runs` timed runs per
    mesh) and the KDTree correspondence of ALL source rows - on the very coordinate arrays the GPU KNN of the last
    timed step consumed.  Nothing is scaled or extrapolated.  Because the same arrays are in hand, the leg doubles as
    the parity check at full size: relative eigenvalue error of the device solves against scipy's (both meshes), and
    every KNN index against the KDTree's."""
    from oracle import reference_port as orc
    from scipy.spatial import KDTree

    t_asm, t_eigs, eig_err, runs_txt = [], [], 0.0, []
    for m_i, mesh in enumerate(meshes):
        t0 = time.perf_counter()
        W, deg, d_inv, L = orc.graph_matrices(mesh.points, mesh.faces)
        t_asm.append(time.perf_counter() - t0)
        if m_i == 0:
            orc.recursive_eig(L, k + 1, k)  # warm-up (first-call costs of ARPACK / SuperLU), untimed
        t_runs = []
        for _ in range(max(1, eigs_runs)):
            t0 = time.perf_counter()
            vals, vecs = orc.recursive_eig(L, k + 1, k)
            t_runs.append(time.perf_counter() - t0)
        t_eigs.append(float(np.median(t_runs)))
        runs_txt.append("/".join("%.2f" % t for t in t_runs))
        vals = np.sort(vals)[:k]
        eig_err = max(eig_err, float(np.max(np.abs(np.asarray(gpu_vals[m_i])[:k] / vals - 1.0))))
        del W, L, vecs
    tgt, src = coords
    t0 = time.perf_counter()
    tree = KDTree(tgt)
    t_tree = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, kd_idx = tree.query(src)
    t_query = time.perf_counter() - t0
    knn_mismatches = int(np.sum(kd_idx != np.asarray(gpu_idx)))
    n = len(src)
    t_pair = sum(t_asm) + sum(t_eigs) + t_tree + t_query
    # SURVEY 8d "best-effort CPU" row: the same calls with KDTree.query(workers=-1) on every host core
    # (ARPACK / SuperLU have no threaded mode: the eigensolve term is unchanged).
    t0 = time.perf_counter()
    tree.query(src, workers=-1)
    t_query_mt = time.perf_counter() - t0
    t_pair_mt = sum(t_asm) + sum(t_eigs) + t_tree + t_query_mt
    best_effort = dict(value=2 * k / t_pair_mt, unit="eigenpairs/s", cores=os.cpu_count(), pair_seconds=t_pair_mt,
                       sample="as above with KDTree.query(workers=-1): %.3fs for all %d queries" % (t_query_mt, n))
    base = dict(best_effort_all_cores=best_effort, value=2 * k / t_pair, unit="eigenpairs/s", cores=1, cpu_model=cpu_model(),
                host_threads=os.cpu_count(), kind="port",
                sample="the whole step, measured: both meshes - vectorised assembly %s s, scipy eigs(sigma=1e-10, ncv=4(k+1)) "
                       "%s s (median of %d runs each after one warm-up solve); KDTree build %.2fs + query of all %d source "
                       "points %.2fs; pair: %.1fs"
                       % (" + ".join("%.2f" % t for t in t_asm), " | ".join(runs_txt), max(1, eigs_runs), t_tree, n, t_query, t_pair),
                pair_seconds=t_pair)
    parity = dict(max_rel_eigenvalue_error_vs_cpu=eig_err, knn_index_mismatches_vs_kdtree=knn_mismatches,
                  knn_rows_checked=int(n),
                  note="device eigenvalues of BOTH meshes of the last timed step against scipy eigs on the oracle's L; "
                       "every device 1-NN index of the last timed step against KDTree.query")
    return base, parity


def cpu_model():
    """The host CPU's model string (stated next to the core count of the CPU baseline)."""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or "unknown"


def device_copy_gbs(torch, dev, n_bytes=1 << 30, reps=10):
    """Achievable HBM figure of this box: a plain device-to-device copy of 1 GiB (4x the Infinity
    Cache), read + write bytes over the time between two events on torch's stream."""
    a = torch.empty(n_bytes // 8, dtype=torch.float64, device=dev).normal_()
    b = torch.empty_like(a)
    b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n_bytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def bundled_15k_pair(ctx, k=5, reps=3):
    """BASELINE config C2: the reference's own 15k bone meshes (asymmetric W, isolated vertices; points and
    faces travel as test fixtures), same hot path; reported next to the headline line, with the eigenvalue
    error against the reference-generated golden values."""
    from pyfocusr_amd import PolyMesh

    gold = os.path.join(REPO, "tests", "golden")
    try:
        zt, zs = np.load(os.path.join(gold, "target_mesh_15k.npz")), np.load(os.path.join(gold, "source_mesh_15k.npz"))
    except OSError:
        return None
    meshes = [PolyMesh(z["points"], z["faces"]) for z in (zt, zs)]
    best, err = None, 0.0
    for _ in range(reps + 1):
        timers = dict(assembly=0.0, eigensolve=0.0, eigsort=0.0, knn=0.0, matvecs=0)
        t0 = time.perf_counter()
        graphs = hot_path_step([ctx, ctx], meshes[0], meshes[1], k, 20000, timers, keep_graphs=True)
        dt = time.perf_counter() - t0
        for g, z in zip(graphs, (zt, zs)):
            gv = z["k5_eig_vals"]
            err = max(err, float(np.max(np.abs(g.eig_vals[:len(gv)] / gv - 1.0))))
        if best is None or dt < best[0]:
            best = (dt, timers)
    return dict(workload="C2: data/target_mesh_15k.vtk + data/source_mesh_15k.vtk (14998 / 14996 vertices), k=5; asymmetric W "
                         "(one-way edges): restarted Arnoldi on L, both graphs in shared launches",
                ms=1e3 * best[0], eigenpairs_per_s=2 * k / best[0],
                breakdown_ms={key: 1e3 * best[1][key] for key in ("assembly", "eigensolve", "eigsort", "knn")},
                matvecs=best[1]["matvecs"], solver_modes=[int(getattr(g.eigs_stats, "mode", -1)) for g in graphs],
                max_rel_eigenvalue_error_vs_reference=err)


def messy_250k_pair(ctx, n=250000, k=5, samples=5000, reps=3, check_cpu=True):
    """The headline pair with the defect classes of the reference's scanned meshes (`meshgen.messy_blob_mesh`:
    ten small holes, stranded vertices, reversed and duplicated faces: ~50 one-way edges per mesh): W asymmetric
    (graph.py:178), three extra null eigenvalues for the widen-and-retry rule (graph.py:374-379).  Untimed extra: stage
    times of the best of `reps` passes, the eigenvalue error of one mesh against the oracle's `recursive_eig` (scipy
    eigs), 96 correspondence rows against a brute-force scan."""
    from pyfocusr_amd import _hip
    from pyfocusr_amd.meshgen import messy_blob_mesh

    meshes = [messy_blob_mesh(n, seed=s) for s in (0, 1)]
    for m in meshes:
        m._pf_device_mesh = _hip.DeviceMesh(m.points, m.faces, ctx=ctx)
    best = None
    for _ in range(reps + 1):
        timers = dict(assembly=0.0, eigensolve=0.0, eigsort=0.0, knn=0.0, matvecs=0)
        t0 = time.perf_counter()
        graphs = hot_path_step([ctx, ctx], meshes[0], meshes[1], k, samples, timers, keep_graphs=True)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, timers, graphs)
    gt, gs = best[2]
    out = dict(workload="the C3 pair (%d vertices, k=%d) with scan defects: per mesh 7 triangular + 3 hexagonal holes, 3 stranded "
                        "vertices, 2 reversed faces, 2 edges in three faces" % (n, k),
               ms=1e3 * best[0], eigenpairs_per_s=2 * k / best[0],
               breakdown_ms={key: 1e3 * best[1][key] for key in ("assembly", "eigensolve", "eigsort", "knn")},
               matvecs=best[1]["matvecs"], columns=[int(len(g.eig_vals)) for g in (gt, gs)],
               solver_modes=[int(getattr(g.eigs_stats, "mode", -1)) for g in (gt, gs)],
               outer_steps=[int(g.eigs_stats.outer_steps) for g in (gt, gs)],
               max_eig_residual=float(max(g.eigs_stats.residuals.max() for g in (gt, gs))))
    # correspondences of 96 source rows against a brute force over all target rows (the coordinates the search used)
    np.random.seed(4321)
    timers = dict(assembly=0.0, eigensolve=0.0, eigsort=0.0, knn=0.0, matvecs=0)
    idx, _, _, (vt, vs, w, vals_t, vals_s) = hot_path_step([ctx, ctx], meshes[0], meshes[1], k, samples, timers)
    tgt, src = vt[:, :k] * w[None, :], vs[:, :k] * w[None, :]
    rows = np.linspace(0, n - 1, 96).astype(np.int64)
    bad = 0
    for r in rows:
        d2 = np.zeros(n)
        for c in range(k):
            d2 += (src[r, c] - tgt[:, c]) ** 2
        bad += int(np.argmin(d2) != idx[r])
    out["knn_rows_checked_bruteforce"], out["knn_index_mismatches"] = int(len(rows)), bad
    if check_cpu:
        from oracle import reference_port as orc

        t0 = time.perf_counter()
        W, deg, d_inv, L = orc.graph_matrices(meshes[0].points, meshes[0].faces)
        ref_vals, _ = orc.recursive_eig(L, k + 1, k)
        ref_vals = np.sort(ref_vals)
        out["cpu_oracle_seconds_one_mesh"] = time.perf_counter() - t0
        m = min(len(ref_vals), len(vals_t))
        out["columns_of_the_oracle"] = int(len(ref_vals))
        out["max_rel_eigenvalue_error_vs_oracle"] = float(np.max(np.abs(np.asarray(vals_t)[:m] / ref_vals[:m] - 1.0)))
    for m_ in meshes:
        del m_._pf_device_mesh
    return out


def c5_1m_k10(ctx, n=1000000, k=10, samples=5000, reps=3):
    """BASELINE config C5 on one GPU (untimed extra, like the bundled 15k pair): a 1M-vertex blob pair, k = 10 -
    stage times of the best of `reps` passes, the largest eigenpair residual, and the correspondence indices of 96
    source rows against a brute-force scan of all 1M target rows (left-to-right squared distances, as the kernel)."""
    from pyfocusr_amd import _hip
    from pyfocusr_amd.meshgen import blob_mesh

    meshes = [blob_mesh(n, seed=s) for s in (0, 1)]
    for m in meshes:
        m._pf_device_mesh = _hip.DeviceMesh(m.points, m.faces, ctx=ctx)
    best = None
    # (one untimed pass first: the ~2 GB of Krylov workspace come fresh from hipMalloc, and the resident kernel's hold-back
    # calibration runs its trial launches - a first pass took 108 ms where the following ones take 81)
    for rep in range(reps + 1):
        timers = dict(assembly=0.0, eigensolve=0.0, eigsort=0.0, knn=0.0, matvecs=0)
        _hip.persist_clock(ctx, reset=True)
        t0 = time.perf_counter()
        idx, res, _, (vt, vs, w, _, _) = hot_path_step([ctx, ctx], meshes[0], meshes[1], k, samples, timers)
        dt = time.perf_counter() - t0
        if rep > 0 and (best is None or dt < best[0]):
            best = (dt, timers, res, _hip.persist_clock(ctx), _hip.persist_state(ctx)["hold_ticks"])
    tgt, src = vt[:, :k] * w[None, :], vs[:, :k] * w[None, :]
    rows = np.linspace(0, n - 1, 96).astype(np.int64)
    bad = 0
    for r in rows:
        d2 = np.zeros(n)
        for c in range(k):
            d2 += (src[r, c] - tgt[:, c]) ** 2
        bad += int(np.argmin(d2) != idx[r])
    for m in meshes:
        del m._pf_device_mesh
    return dict(workload="C5 on ONE GPU: synthetic %d-vertex blob pair, k=%d (assembly + eigensolve x2, eigsort, 1-NN in d=%d)" % (n, k, k),
                ms=1e3 * best[0], eigenpairs_per_s=2 * k / best[0],
                breakdown_ms={key: 1e3 * best[1][key] for key in ("assembly", "eigensolve", "eigsort", "knn")},
                matvecs=best[1]["matvecs"], max_eig_residual=float(best[2]),
                resident_launches=int(best[3][1]), resident_us_per_launch=1e3 * best[3][0] / max(best[3][1], 1), hold_ticks=int(best[4]),
                knn_rows_checked_bruteforce=int(len(rows)), knn_index_mismatches=bad)


def single_graph_solve(ctx, mesh, k, reps=3):
    """One mesh alone (what `Graph.get_graph_spectrum()` / `pf_eigs_smallest` run when there is no partner): the resident
    filter kernel with two recurrence steps per exchange (the default for single graphs with 1024-row windows) against
    one step per exchange; best of `reps` solves each, eigenvalues compared."""
    from pyfocusr_amd import Graph, _hip

    out, vals = {}, {}
    try:
        for level, name in ((1, "two_steps_per_exchange_ms"), (0, "one_step_per_exchange_ms")):
            _hip.persist_two_step(level)
            best = None
            for _ in range(reps):
                g = Graph(mesh, n_spectral_features=k, n_rand_samples=5000, ctx=ctx, verbose=False)
                _ = g.device
                ctx.sync()
                t0 = time.perf_counter()
                g.get_graph_spectrum()
                _ = g.eig_vecs  # (collects the download)
                dt = time.perf_counter() - t0
                best = dt if best is None or dt < best else best
                vals[level] = g.eig_vals.copy()
                g.device.close()
            out[name] = 1e3 * best
    finally:
        _hip.persist_two_step(1)
    out["max_rel_eigenvalue_difference"] = float(np.max(np.abs(vals[1] / vals[0] - 1.0)))
    out["workload"] = "eigensolve of ONE %d-vertex mesh, k=%d (no partner graph in the launches)" % (len(mesh.points), k)
    return out


def row_partition_step(ctx, dist, torch, mesh, k, s):
    """BASELINE config C5 layout (opt-in, `--row-partition S`): ONE mesh's rows split over all ranks, ghost zones of
    depth S, the Chebyshev recurrence exchanging boundary rows every S steps (pyfocusr_amd/rowpart.py).  Returns
    (seconds of the eigensolve, seconds of the setup, stats)."""
    from pyfocusr_amd import _hip, rowpart

    comm = rowpart.Comm(dist, torch)
    t0 = time.perf_counter()
    full = _hip.DeviceLaplacian(mesh.points, mesh.faces, ctx=ctx)
    made = []

    def make_local(S_local):
        made.append(_hip.DeviceLaplacian(matrix=(S_local.indptr, S_local.indices, S_local.data), ctx=ctx))
        return made[-1]

    marks = {}  # row_partitioned_eigs records the seconds of the eigensolve proper under "solve"
    vals, vecs, own, stats, ops = rowpart.row_partitioned_eigs(mesh.points, mesh.faces, k, comm, make_local, s=s,
                                                               device_graph=full, timing=marks,
                                                               device_exchange=dist.get_backend() == "nccl")
    ctx.sync()
    total = time.perf_counter() - t0
    out = dict(solve_s=marks.get("solve", float("nan")), setup_s=total - marks.get("solve", 0.0), matvecs=int(stats.matvecs),
               exchanges=int(ops.exchanges), rows_own=int(len(own)), rows_local=int(ops.layout.n_local), eig_vals=vals.tolist())
    for g in made:
        g.close()
    full.close()
    return out


def split_pair_step(ctx, dist, torch, tdev, rank, mesh, k, n_samples):
    """BASELINE config C4: rank 0 = target, rank 1 = source; one all-gather of the eigenvector blocks as they sit in
    HBM (RCCL over xGMI, enqueued on the library's stream; nothing but 5000-row samples reaches a host); eigsort
    replicated (k x k work); KNN sharded by source rows, read from the gathered device buffer; int64 indices gathered."""
    from pyfocusr_amd import Graph
    from pyfocusr_amd.parallel import all_gather_rows, gather_spectral, shard_rows, split_pair_correspondence

    g = Graph(mesh, n_spectral_features=k, n_rand_samples=n_samples, ctx=ctx, verbose=False)
    g.get_graph_spectrum()
    if tdev == "cuda":
        stream = torch.cuda.ExternalStream(ctx.stream_ptr, device=torch.device("cuda", ctx.device))
        idx, _, _ = split_pair_correspondence(dist, torch, g, k, n_samples, seed=1234, stream=stream)
        g.device.close()
        return idx
    # rehearsal over gloo (CPU tensors): the host-staged form
    from pyfocusr_amd import eigsort

    vals, vecs, pts = gather_spectral(dist, torch, g.eig_vals, g.eig_vecs, g.points)
    graphs = []
    for r in range(2):
        h = Graph.__new__(Graph)
        h.points, h.n_points, h.eig_vals, h.eig_vecs = pts[r], len(pts[r]), vals[r], vecs[r].copy()
        h.eig_val_gap, h.verbose, h._ctx, h._device = None, False, ctx, None
        np.random.seed(1234)  # identical sample on both ranks
        h.rand_idxs = h.get_list_rand_idxs(n_samples)
        graphs.append(h)
    gt, gs = graphs
    Q = eigsort(gt, gs, k, target_as_reference=True).sort_eigenmaps()
    w = Q[:k] * np.max((gs.eig_vals[:k], gt.eig_vals[:k]), axis=0)
    w = np.exp(-(w**2) / (2 * np.mean(w) ** 2))
    src, tgt = gs.eig_vecs[:, :k] * w[None, :], gt.eig_vecs[:, :k] * w[None, :]
    lo, hi = shard_rows(len(src), 2, rank)
    part = ctx.knn1(tgt, src[lo:hi])
    parts = all_gather_rows(dist, torch, part.astype(np.float64))
    g.device.close()
    return np.concatenate([p[:, 0] for p in parts]).astype(np.int64)


def free_port():
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def spawn_command(n_gpus, argv, port):
    """The launcher line of the contract for N ranks on one node (what the driver itself runs for N > 1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(n_gpus, argv=None):
    """Run this script's ranks as children of a process that has not touched the GPU; returns the exit status to pass on
    (the launcher's: non-zero if any rank failed; rank 0's JSON line goes to stdout as it is printed)."""
    import subprocess

    cmd = spawn_command(n_gpus, sys.argv[1:] if argv is None else argv, free_port())
    if os.environ.get("PF_BENCH_SPAWN_DRYRUN") == "1":  # (tests: the command only)
        print(json.dumps(cmd))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: what RCCL needs on this pool's hosts)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--vertices", dest="n", type=int, default=250000, help="vertices per mesh")
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--samples", type=int, default=5000, help="n_coords_spectral_ordering (focusr.py:37)")
    ap.add_argument("--cpu-eigs-runs", type=int, default=3, help="timed scipy eigs runs per mesh of the CPU baseline (median)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the headline workload (no bundled-15k-pair measurement): keeps rocprofv3 kernel statistics "
                         "to the 250k launches")
    ap.add_argument("--row-partition", type=int, default=0, metavar="S",
                    help="opt-in extra at N > 1: additionally solve ONE mesh with its rows split over all ranks, ghost "
                         "zones of depth S (BASELINE config C5 layout); reported under 'row_partitioned'")
    ap.add_argument("--pair", choices=("on", "off"), default="on",
                    help="on: the target and source recurrences share kernel launches (pf_cheb2, the library default); "
                         "off: one graph per launch, the two solves one after the other")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="torch.distributed backend; gloo + --share-gpu rehearses the N>1 path on a 1-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses GPU 0")
    ap.add_argument("--trace-calls", action="store_true",
                    help="diagnostic: host wall time of the blocking library calls of the timed steps, summed per call name, "
                         "under 'host_call_ms_per_step'")
    ap.add_argument("--streams", type=int, default=1, choices=(1, 2),
                    help="1: target and source Chebyshev recurrences in lockstep, two graphs per launch on one stream; "
                         "2: two host threads, one HIP stream each")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.k < 2:
        raise SystemExit("bench.py: --k must be >= 2 (with a single eigenmap the reference's eigenvalue-gap cost, "
                         "eigsort.py:149-158, is the mean of an empty difference: NaN)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # a plain `python bench.py --gpus N`: start the N ranks ourselves, as fresh child processes (one per GPU over
        # torch.distributed.run), BEFORE this process has made any GPU call - it never will: it relays the ranks' output
        # and exits with their status
        raise SystemExit(spawn_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch

    if args.share_gpu:
        local = 0
        os.environ["PF_PERSIST"] = "0"  # several processes on one GPU cannot all keep a resident kernel's blocks on it
    torch.cuda.set_device(local)
    dist = None
    tdev = "cuda" if args.backend == "nccl" else "cpu"  # where the collectives' tensors live
    if world > 1:
        import torch.distributed as dist

        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")

    from pyfocusr_amd import _hip
    from pyfocusr_amd import graph as _graph
    from pyfocusr_amd.meshgen import blob_mesh

    _graph.PAIRED_LAUNCHES = args.pair == "on"
    call_ms = {}
    if args.trace_calls:
        def _wrap(cls, name):
            fn = getattr(cls, name)

            def wrapped(self, *a, **kw):
                t0 = time.perf_counter()
                try:
                    return fn(self, *a, **kw)
                finally:
                    call_ms[name] = call_ms.get(name, 0.0) + 1e3 * (time.perf_counter() - t0)
            setattr(cls, name, wrapped)
        for nm in ("finalize_wait", "final_remap", "eigs_smallest2", "final_rows", "point_rows"):
            _wrap(_hip.DeviceLaplacian, nm)
        for nm in ("eigsort_costs", "knn1_graphs", "sync"):
            _wrap(_hip.Context, nm)

    ctx = _hip.Context(local)
    ctxs = [ctx, _hip.Context(local) if args.streams == 2 else ctx]  # one stream per mesh of the pair
    for c in set(ctxs):
        # every TIMING_STRIDE-th filter application of the timed region carries a HIP event pair (the roofline entry's
        # average launch duration); an event record costs ~5 us of device time, and a pair around each of the 36 applications
        # of a step was 0.25 ms of the step it measures.  PF_BENCH_TIMING=0 / 1: none / all of them (A/B).
        c.timing_enable(int(os.environ.get("PF_BENCH_TIMING", TIMING_STRIDE)))
    mesh_t, mesh_s = blob_mesh(args.n, seed=2 * rank), blob_mesh(args.n, seed=2 * rank + 1)
    for m, c in zip((mesh_t, mesh_s), ctxs):
        m._pf_device_mesh = _hip.DeviceMesh(m.points, m.faces, ctx=c)  # inputs resident in HBM

    def barrier():
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    timers = dict(assembly=0.0, eigensolve=0.0, eigsort=0.0, knn=0.0, matvecs=0)
    if os.environ.get("PF_BENCH_DETAIL") == "1":
        timers["assembly_detail"] = {}
    np.random.seed(1234 + rank)
    for _ in range(args.warmup):
        # (the results are held until the next step returns, as in the timed loop: a step then finds the previous step's
        # pinned eigenvector arrays still in use and the pool of those grows to its steady four blocks HERE - two
        # hipHostMalloc calls of 10 MB, 1.7 ms, used to land in the second timed step)
        idx, max_res, nnz, coords = hot_path_step(ctxs, mesh_t, mesh_s, args.k, args.samples, timers)
    for key in timers:
        timers[key] = {} if key == "assembly_detail" else 0
    call_ms.clear()
    for c in set(ctxs):
        c.timing(reset=True)
        _hip.persist_clock(c, reset=True)
    barrier()
    t0 = time.perf_counter()
    step_ends, stage_marks = [], []
    for _ in range(args.steps):
        idx, max_res, nnz, coords = hot_path_step(ctxs, mesh_t, mesh_s, args.k, args.samples, timers)
        step_ends.append(time.perf_counter())
        stage_marks.append((timers["assembly"], timers["eigensolve"], timers["eigsort"], timers["knn"], timers["matvecs"]))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timing_stride = max(int(os.environ.get("PF_BENCH_TIMING", TIMING_STRIDE)), 1)
    tms = [c.timing() for c in set(ctxs)]
    clocks = [_hip.persist_clock(c) for c in set(ctxs)]  # every resident launch of the timed steps, on the device's own clock
    clock_ms, clock_n = sum(x[0] for x in clocks), sum(x[1] for x in clocks)
    tm = dict(knn_ms=tms[0]["knn_ms"])
    for key in ("op_ms", "op_launches", "op_bytes", "persist_ms", "persist_launches", "persist_steps", "persist_bytes",
                "persist_lds_bytes"):
        tm[key] = sum(t[key] for t in tms)

    # The kernel that DOES stream the operators through HBM (one step per launch, k_sell_op2): the path of graphs the
    # resident kernel does not cover.  One untimed extra step with the resident kernel switched off gives its HBM
    # roofline entry in the same run.
    stream_tm = None
    if rank == 0 and not args.no_extras:
        _hip.persist_enable(False)
        try:
            extra = dict(assembly=0.0, eigensolve=0.0, eigsort=0.0, knn=0.0, matvecs=0)
            for c in set(ctxs):
                c.timing(reset=True)
            hot_path_step(ctxs, mesh_t, mesh_s, args.k, args.samples, extra)
            sts = [c.timing(reset=True) for c in set(ctxs)]
            stream_tm = {key: sum(t[key] for t in sts) for key in ("op_ms", "op_launches", "op_bytes", "persist_launches")}
            stream_tm["eigensolve_ms"] = 1e3 * extra["eigensolve"]
        finally:
            _hip.persist_enable(True)

    failed = []  # extras that raised: the line still prints, the exit status is non-zero
    rowp = None
    if world > 1 and args.row_partition > 0:
        try:
            mesh_rp = mesh_t if rank == 0 else blob_mesh(args.n, seed=0)  # every rank needs the SAME mesh here
            barrier()
            r = row_partition_step(ctx, dist, torch, mesh_rp, args.k, args.row_partition)
            barrier()
            t = torch.tensor([r["solve_s"], r["setup_s"]], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            rows = torch.tensor([float(r["rows_local"])], dtype=torch.float64, device=tdev)
            dist.all_reduce(rows, op=dist.ReduceOp.MAX)
            rowp = dict(workload="C5 layout: ONE %d-vertex mesh, rows split over %d ranks, ghost depth %d, k=%d"
                                 % (args.n, world, args.row_partition, args.k),
                        eigensolve_ms=1e3 * float(t[0]), setup_ms=1e3 * float(t[1]), matvecs=r["matvecs"],
                        exchanges=r["exchanges"], rows_own_rank0=r["rows_own"], max_rows_with_ghosts=int(rows.item()),
                        eig_vals=r["eig_vals"])
            if rank == 0:  # the same mesh on one device, for comparison
                from pyfocusr_amd import Graph

                t0 = time.perf_counter()
                g1 = Graph(mesh_t, n_spectral_features=args.k, n_rand_samples=args.samples, ctx=ctx, verbose=False)
                _ = g1.device
                ctx.sync()
                t1 = time.perf_counter()
                g1.get_graph_spectrum()
                rowp["single_device_eigensolve_ms"] = 1e3 * (time.perf_counter() - t1)
                rowp["max_rel_eigenvalue_diff_vs_single_device"] = float(
                    np.max(np.abs(np.array(r["eig_vals"]) - g1.eig_vals[:args.k]) / g1.eig_vals[:args.k]))
                g1.device.close()
        except Exception as exc:  # noqa: BLE001 - an opt-in extra must never cost the headline line (but it costs the exit status)
            rowp = dict(error="%s: %s" % (type(exc).__name__, exc))
            failed.append("row_partitioned")

    if rank == 0:
        n = args.n
        pmc = {}
        if os.path.exists(os.path.join(REPO, PMC_SUMMARY)):
            with open(os.path.join(REPO, PMC_SUMMARY)) as fh:
                pmc = json.load(fh).get("kernels", {})  # separate --pmc passes of this command (tools/pmc_make_summary.py)
        # The dominant kernel.  With the resident kernel (the default for this workload) the operators live in registers
        # and x in LDS: its bytes are LDS bytes, its step time is set by one memory-side hand-off between neighbouring
        # windows plus the LDS gathers of the boundary rows, and what it moves through HBM is two orders of magnitude
        # below the algorithmic bytes of one step per launch.  So: `bound` "lds", `achieved` = LDS bytes per launch / launch
        # time against the LDS peak; `hbm_frac` from PMC traffic; the algorithmic figure stays as an EFFECTIVE rate,
        # never divided by a peak.  Otherwise (resident path off / not applicable): the streaming kernel, HBM roofline.
        resident = tm.get("persist_launches", 0) > 0 and tm["persist_ms"] > 0.5 * tm["op_ms"]
        if tm["op_ms"] <= 0.0:  # PF_BENCH_TIMING=0: no event pairs, no roofline entry (an A/B of what the events cost)
            roofline = {"bound": None, "note": "operator timing switched off (PF_BENCH_TIMING=0)"}
        elif resident:
            launches, kernel_ms = tm["persist_launches"], tm["persist_ms"]
            kernel_us = 1e3 * kernel_ms / max(launches, 1)
            lds_per_launch = tm["persist_lds_bytes"] / max(launches, 1)
            achieved = lds_per_launch / (kernel_us * 1e-6) / 1e9
            steps_per_launch = tm["persist_steps"] / 2.0 / max(launches, 1)  # the library counts graph-steps: two per pair-step
            entry = pmc.get("k_cheb_resident<2, 1, 8, true>", {})
            traffic = entry.get("hbm_bytes_per_step", None)
            traffic_refused = None
            if traffic is not None:
                # the committed PMC passes must be of THIS kernel in THIS shape: same steps per launch (the filter degree)
                # and the same LDS bytes per launch (rows, entries, outside rows), else the figure is not this run's
                rec_steps = entry.get("steps_per_launch_avg")
                rec_lds = entry.get("lds_bytes_per_launch")
                if rec_steps is None or abs(rec_steps - steps_per_launch) > 0.02 * steps_per_launch:
                    traffic_refused = "recorded steps per launch %s != %.1f of this run" % (rec_steps, steps_per_launch)
                elif rec_lds is not None and abs(rec_lds - lds_per_launch) > 0.02 * lds_per_launch:
                    traffic_refused = "recorded LDS bytes per launch %.4g != %.4g of this run" % (rec_lds, lds_per_launch)
                if traffic_refused:
                    traffic = None
            traffic = None if traffic is None else traffic * steps_per_launch + entry.get("hbm_bytes_per_launch_fixed", 0.0)
            roofline = {
                "bound": "lds",
                "kernel": "k_cheb_resident<2, 1, 8, true> (a whole Chebyshev recurrence of both graphs of the pair per launch: one "
                          "block per CU owns a 1024-row window of each graph - its two halves take the graphs in opposite order - "
                          "SELL-64 entries in registers, the window's x double-buffered in LDS, boundary rows handed to "
                          "neighbouring windows through memory, value = message, first fetch of a step at a tuned time; f64)",
                "achieved": achieved, "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": achieved / LDS_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": (PMC_SUMMARY + " (rocprofv3 --pmc passes of this command, committed; not measured in this run)")
                if traffic is not None else (None if traffic_refused is None else "refused: " + PMC_SUMMARY + ": " + traffic_refused),
                "hbm_frac": None if traffic is None else traffic / (kernel_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "lds_bytes_per_launch": lds_per_launch,
                # the same average on the device's 100 MHz clock inside the kernel (first to last instruction of block 0, ALL
                # launches of the timed steps): what rocprofv3's kernel trace measures; the event pair adds the dispatch of
                # the kernel and the event packets (~20 us)
                "avg_launch_us_in_kernel_clock": None if clock_n == 0 else 1e3 * clock_ms / clock_n, "launches_in_kernel_clock": clock_n,
                # when a block first asks for its neighbours' values, in 10 ns ticks after a step began: the library's table,
                # adjusted by what this process measured on its first launches
                "hold_back_ticks": _hip.persist_state(ctx)["hold_ticks"],
                "avg_launch_us_hip_events": kernel_us, "launches": launches, "launches_are": "the TIMED launches: one filter application in %d "
                "carries the event pair" % timing_stride, "launches_per_step": launches * timing_stride / args.steps,
                "steps_per_launch": steps_per_launch,
                "us_per_step_of_the_pair": 2e3 * kernel_ms / max(tm["persist_steps"], 1),
                "effective_algorithmic_GBps": tm["persist_bytes"] / max(launches, 1) / (kernel_us * 1e-6) / 1e9,
                # the bound that does apply: every step contains one dependent memory-side hand-off between workgroups
                # (MI355X_MICROARCH.md, price list, "handoff-1to1": 0.8-1.0 us on an idle chip for <= 4 KB)
                "handoff_floor_us_per_step": 1.0,
                "frac_of_handoff_floor": 1.0 / (2e3 * kernel_ms / max(tm["persist_steps"], 1)),
                "note": "frac is LDS bytes (8 B per gathered x + 8 B per row + 8 B per outside row, counted by the library) "
                        "against the conflict-free ds_read_b64 peak; random 8-byte gathers conflict ~3-4x, and each step waits "
                        "for one memory-side hand-off (~1 us): the kernel is latency-bound, neither LDS- nor HBM-bandwidth-bound "
                        "(DESIGN.md section 4).  effective_algorithmic_GBps = SURVEY 8d bytes (12 nnz + 20 n + 4 per graph and step, "
                        "what one step per launch streams) over the same time: an effective rate, not a roofline fraction.",
            }
        else:
            launches, kernel_ms, kernel_bytes = tm["op_launches"], tm["op_ms"], tm["op_bytes"]
            kernel_us = 1e3 * kernel_ms / max(launches, 1)
            alg_bytes = kernel_bytes / max(launches, 1)
            achieved = alg_bytes / (kernel_us * 1e-6) / 1e9
            entry = pmc.get("k_sell_op2<true>", {})
            traffic = entry.get("hbm_bytes_per_launch")
            roofline = {"bound": "hbm", "kernel": "k_sell_op2/k_sell_op (fused SpMV + Chebyshev recurrence, SELL-64, f64; both graphs "
                                                  "of the pair per launch)",
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_source": PMC_SUMMARY if traffic is not None else None,
                        "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us_hip_events": kernel_us, "launches": launches}
        roofline_stream = None
        if stream_tm is not None and stream_tm["op_launches"] > 0 and stream_tm["persist_launches"] == 0:
            us = 1e3 * stream_tm["op_ms"] / stream_tm["op_launches"]
            alg = stream_tm["op_bytes"] / stream_tm["op_launches"]
            entry = pmc.get("k_sell_op2<true>", {})
            roofline_stream = {
                "bound": "hbm", "kernel": "k_sell_op2 (one Chebyshev step of both graphs per launch, SELL-64 streamed from HBM / "
                                          "Infinity Cache; the path of graphs the resident kernel does not cover; measured in ONE extra, "
                                          "untimed step of this run with the resident kernel switched off)",
                "achieved": alg / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": entry.get("hbm_bytes_per_launch"),
                "traffic_source": PMC_SUMMARY if entry.get("hbm_bytes_per_launch") is not None else None,
                "algorithmic_bytes_per_launch": alg, "avg_launch_us_hip_events": us, "launches": stream_tm["op_launches"],
                "eigensolve_ms_of_that_step": stream_tm["eigensolve_ms"]}
        out = {
            "metric": METRIC,
            "value": world * 2 * args.k * args.steps / elapsed,
            "unit": "eigenpairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "C3: synthetic %d-vertex closed-manifold blob mesh pair per GPU (seeds 2r/2r+1), k=%d: "
                                   "assembly + eigensolve x2, eigsort, 1-NN correspondence" % (n, args.k),
                       "n_vertices": n, "n_faces": 2 * n - 4, "k": args.k, "parallelism": "%d independent pair(s)" % world},
            "breakdown_ms_per_step": {key: 1e3 * timers[key] / args.steps
                                      for key in ("assembly", "eigensolve", "eigsort", "knn")},
            **({"assembly_detail_ms_per_step": {k2: 1e3 * v2 / args.steps for k2, v2 in timers["assembly_detail"].items()}}
               if "assembly_detail" in timers else {}),
            **({"ms_of_each_step": [round(1e3 * (b - a), 3) for a, b in zip([t0] + step_ends[:-1], step_ends)],
                "stages_of_each_step_ms_and_matvecs": [[round((1e3 if i < 4 else 1) * (y - x), 3) for i, (x, y) in enumerate(zip(a, b))]
                                                       for a, b in zip([(0, 0, 0, 0, 0)] + stage_marks[:-1], stage_marks)]}
               if os.environ.get("PF_BENCH_DETAIL") == "1" else {}),
            "matvecs_per_step": timers["matvecs"] / args.steps,
            # eigenpairs/s of the eigensolve alone (Laplacian on the device -> normalised eigenpairs in
            # host memory), and the algorithmic bytes the operator kernel moved per step (sum over its launches)
            "eigensolve_only_eigenpairs_per_s": 2 * args.k * world / (timers["eigensolve"] / args.steps),
            # (the library accumulates the bytes of the TIMED applications: one in timing_stride)
            "operator_algorithmic_bytes_per_step": tm["op_bytes"] * timing_stride / args.steps,
            "filter_applications_timed": int(tm["op_launches"]) if tm.get("persist_launches", 0) > 0 else None,
            "timing_stride": timing_stride,
            "knn_kernel_ms": tm["knn_ms"],
            "max_eig_residual": float(max_res),
            "roofline": roofline,
        }
        # every stage against the roofline that bounds it.  Assembly: SURVEY 8d's algorithmic bytes
        # (144 n per mesh: points and faces in, CSR(W) / L values, degrees out) over the stage's wall time against the HBM
        # peak - the stage is ~110 small dependent launches, bound by their number and latency, not by bytes; `traffic` =
        # what its kernels really moved (PMC passes of this command, committed).  1-NN: floating-point operations of the
        # squared distances the search evaluates (3 d per candidate-query pair, pairs counted by one extra untimed search
        # with the counting instantiation) over the kernel's time against the FP64 vector peak.
        stage_pmc = {}
        if os.path.exists(os.path.join(REPO, PMC_SUMMARY)):
            with open(os.path.join(REPO, PMC_SUMMARY)) as fh:
                stage_pmc = json.load(fh).get("stages", {})
        asm_bytes = 2 * 144.0 * n
        asm_s = timers["assembly"] / args.steps
        asm_traffic = stage_pmc.get("assembly", {}).get("hbm_bytes_per_step")
        out["roofline_assembly"] = {
            "bound": "hbm", "achieved": asm_bytes / asm_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": asm_bytes / asm_s / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes_per_step": asm_bytes, "ms_per_step": 1e3 * asm_s, "traffic": asm_traffic,
            "traffic_over_algorithmic": None if asm_traffic is None else asm_traffic / asm_bytes,
            "dispatches_per_step": stage_pmc.get("assembly", {}).get("dispatches_per_step"),
            "traffic_source": PMC_SUMMARY if asm_traffic is not None else None,
            "note": "both meshes of the pair; wall time of the stage (the two builds share their launches: pf_launch.h)"}
        try:
            if args.no_extras:
                raise LookupError("skipped with --no-extras (one extra untimed step)")
            ctx.knn_count(True)
            np.random.seed(99)
            extra = dict(assembly=0.0, eigensolve=0.0, eigsort=0.0, knn=0.0, matvecs=0)
            hot_path_step(ctxs, mesh_t, mesh_s, args.k, args.samples, extra)
            pairs = ctx.knn_count(False)
            knn_ms = tm["knn_ms"] if tm["knn_ms"] > 0 else None  # (the search of the last timed step, by HIP events)
            flops = 3.0 * args.k * pairs
            out["roofline_knn"] = {
                "bound": "fp64 valu", "kernel": "k_knn_coop<%d> (exact 1-NN of every source row, grid over two axes, 64 candidates per wave step)" % args.k,
                "achieved": None if not knn_ms else flops / (knn_ms * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": None if not knn_ms else flops / (knn_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                "candidate_query_pairs": pairs, "candidates_per_query": pairs / float(n), "flops_per_pair": 3 * args.k,
                "brute_force_pairs": float(n) * n, "pruned_to_fraction": pairs / (float(n) * n), "kernel_ms": knn_ms,
                "traffic": stage_pmc.get("knn", {}).get("hbm_bytes_per_step"),
                "note": "pairs counted in one extra untimed step (same meshes, another eigsort sample: the count moves a few per cent "
                        "with the sampled weights); kernel_ms: HIP events around the search of the timed steps; the kernel is bound by "
                        "the latency of its per-chunk loads, not by the FP64 rate (DESIGN.md section 5)"}
        except LookupError as exc:
            out["roofline_knn"] = dict(skipped=str(exc))
        except Exception as exc:  # noqa: BLE001
            out["roofline_knn"] = dict(error="%s: %s" % (type(exc).__name__, exc))
            failed.append("roofline_knn")
        if args.trace_calls:
            out["host_call_ms_per_step"] = {k: v / args.steps for k, v in call_ms.items()}
        if roofline_stream is not None:
            out["roofline_streaming_kernel"] = roofline_stream
        if not args.no_extras:
            out["roofline"]["achievable_hbm_copy_GBps"] = device_copy_gbs(torch, torch.device("cuda", local))
        if rowp is not None:
            out["row_partitioned"] = rowp
        if world == 1 and not args.no_extras:
            c2 = bundled_15k_pair(ctx, 5)
            if c2 is not None:
                out["bundled_15k_pair"] = c2
            try:
                out["single_graph_solve"] = single_graph_solve(ctx, mesh_t, args.k)
            except Exception as exc:  # noqa: BLE001
                out["single_graph_solve"] = dict(error="%s: %s" % (type(exc).__name__, exc))
                failed.append("single_graph_solve")
            try:
                out["messy_250k_pair"] = messy_250k_pair(ctx, n=args.n, k=args.k, samples=args.samples, check_cpu=not args.no_cpu_baseline)
                out["messy_250k_pair"]["ratio_to_clean_pair"] = out["messy_250k_pair"]["ms"] / out["ms_per_step"]
            except Exception as exc:  # noqa: BLE001
                out["messy_250k_pair"] = dict(error="%s: %s" % (type(exc).__name__, exc))
                failed.append("messy_250k_pair")
            try:
                out["c5_1m_k10"] = c5_1m_k10(ctx)
            except Exception as exc:  # noqa: BLE001 - an extra: recorded, and the exit status says so
                out["c5_1m_k10"] = dict(error="%s: %s" % (type(exc).__name__, exc))
                failed.append("c5_1m_k10")
        if world == 1 and not args.no_cpu_baseline:
            vt, vs, w, vals_t, vals_s = coords  # last timed step; the coordinate arrays are built here, outside the timed region
            coords = (vt[:, :args.k] * w[None, :], vs[:, :args.k] * w[None, :])
            out["cpu_baseline"], out["parity_at_full_size"] = cpu_baseline((mesh_t, mesh_s), args.k, coords, (vals_t, vals_s), idx,
                                                                         eigs_runs=args.cpu_eigs_runs)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    else:
        out = {}
    # BASELINE config C4 (N = 2 only), measured AFTER the headline figures are in hand and under a watchdog: this is the
    # one place where the ranks talk to each other on the data path, and a collective that never completes must not
    # cost the headline line (the watchdog prints it, with the failure named, and ends the ranks).
    split = None
    if world == 2:
        import threading

        done = threading.Event()

        def watchdog():
            if not done.wait(240.0):
                if rank == 0:
                    out["split_pair"] = dict(error="no completion within 240 s (collective hang?)")
                    print(json.dumps(out), flush=True)
                os._exit(3)  # a collective that never completed: the line is out, the status says it failed

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            for it in range(2):  # one warm-up, one timed
                barrier()
                s0 = time.perf_counter()
                split_idx = split_pair_step(ctx, dist, torch, tdev, rank, mesh_t if rank == 0 else mesh_s, args.k, args.samples)
                barrier()
                split_s = time.perf_counter() - s0
            t = torch.tensor([split_s], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            split = dict(workload="C4: one %d-vertex pair, target on GPU0 / source on GPU1, ONE device-to-device RCCL all-gather of "
                                  "the resident eigenvector blocks, eigsort replicated from samples, query-sharded KNN on the "
                                  "gathered buffer" % args.n, ms=1e3 * float(t.item()),
                         eigenpairs_per_s=2 * args.k / float(t.item()), scaling="strong",
                         n_correspondences=int(len(split_idx)))
        except Exception as exc:  # noqa: BLE001 - an extra must never cost the headline line (but it costs the exit status)
            split = dict(error="%s: %s" % (type(exc).__name__, exc))
            failed.append("split_pair")
        done.set()
        if rank == 0:
            out["split_pair"] = split

    if rank == 0:
        print(json.dumps(out), flush=True)
    if split is not None and "error" in split:
        sys.stdout.flush()
        os._exit(4)  # the peer may be stuck in a collective this rank never reached: do not wait for it in a barrier
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.stderr.write("bench.py: extras failed: %s\n" % ", ".join(failed))
        sys.exit(5)


if __name__ == "__main__":
    main()
