#!/usr/bin/env python3
"""Randomised cross-check of the three eigensolver front ends on symmetric meshes: the Python Krylov driver (reference
point), `pf_eigs_smallest` (one C call) and the row-partitioned solve with thread-ranks on one GPU (random world size and
ghost depth, host-staged and device-buffer exchange).   python tools/fuzz_solvers.py SEED N_CASES"""
import os
import sys
import threading
import time
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import Graph, PolyMesh, _hip, rowpart  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

ctx = _hip.default_context()
rng = np.random.default_rng(int(sys.argv[1]))
N = int(sys.argv[2])
fails, t0 = 0, time.time()
for it in range(N):
    n, k = int(rng.choice([400, 1500, 6000, 25000, 80000])), int(rng.integers(1, 9))
    m = blob_mesh(n, seed=int(rng.integers(0, 10**6)))
    pts, faces = m.points, m.faces
    two = bool(rng.integers(0, 3) == 0)
    if two:  # second component: pf_eigs_smallest locks two null vectors (the row-partitioned mode needs one component)
        m2 = blob_mesh(max(200, n // 4), seed=int(rng.integers(0, 10**6)))
        pts, faces = np.concatenate([pts, m2.points + 300.0]), np.concatenate([faces, m2.faces + n])
    label = "n=%d k=%d two_components=%s" % (len(pts), k, two)
    try:
        g = Graph(PolyMesh(pts, faces), n_spectral_features=k, norm_eig_vecs=False, n_rand_samples=10**9, ctx=ctx, verbose=False)
        g.get_graph_spectrum()
        ref_vals, ref_vecs = g.eig_vals[:k], g.eig_vecs[:, :k]
        vals, vecs, st = g.device.eigs_smallest(k)
        assert len(vals) == k and np.allclose(vals, ref_vals, rtol=1e-9, atol=0), ("pf_eigs_smallest values", vals, ref_vals)
        gaps = np.minimum(np.diff(np.concatenate(([0.0], ref_vals))), np.diff(np.concatenate((ref_vals, [2 * ref_vals[-1]]))))
        assert np.all(np.max(np.abs(vecs - ref_vecs), axis=0) < 1e-7 + 1e-10 / gaps), "pf_eigs_smallest vectors"
        if not two:
            world, s, dev_x = int(rng.integers(2, 5)), int(rng.integers(1, 21)), bool(rng.integers(0, 2))
            label += " world=%d s=%d device_exchange=%s" % (world, s, dev_x)
            shared, out, errors = rowpart.ThreadComm.Shared(world), {}, []

            def rank_main(rank):
                try:
                    c = _hip.Context(0)
                    comm = rowpart.ThreadComm(shared, rank, torch=torch)
                    full = _hip.DeviceLaplacian(pts, faces, ctx=c)
                    made = []

                    def make_local(S_local):
                        made.append(_hip.DeviceLaplacian(matrix=(S_local.indptr, S_local.indices, S_local.data), ctx=c))
                        return made[-1]

                    v, x, own, stats, ops = rowpart.row_partitioned_eigs(pts, faces, k, comm, make_local, s=s, device_graph=full,
                                                                         device_exchange=dev_x)
                    out[rank] = (v, x, own)
                    for d in made + [full]:
                        d.close()
                    c.close()
                except BaseException as exc:  # noqa: BLE001
                    errors.append(exc)
                    shared.barrier.abort()

            threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            if errors:
                raise errors[0]
            allv = np.zeros((len(pts), k))
            for r in range(world):
                assert np.allclose(out[r][0], ref_vals, rtol=1e-9, atol=0), ("row-partitioned values", out[r][0], ref_vals)
                allv[out[r][2]] = out[r][1]
            sign = np.sign(np.sum(allv * ref_vecs, axis=0))
            assert np.all(np.max(np.abs(allv * sign - ref_vecs), axis=0) < 1e-7 + 1e-10 / gaps), "row-partitioned vectors"
        g.device.close()
    except Exception:
        fails += 1
        print("FAIL %s\n%s" % (label, traceback.format_exc()[-700:]), flush=True)
print("done: %d failures of %d, %.1fs" % (fails, N, time.time() - t0))
