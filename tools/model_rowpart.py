#!/usr/bin/env python3
"""Per-rank cost model of the row-partitioned solve from single-GPU measurements: for a mesh of N vertices split
over P ranks with ghost depth S, build rank 0's chunk + ghost graph and time (i) S recurrence steps on it, (ii) the
device-side part of one boundary exchange (gather -> copy -> index_select -> scatter; the RCCL transfer itself is
not included), against (iii) the single-device step on the whole mesh.   python tools/model_rowpart.py [N] [P] [S]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pyfocusr_amd import _hip, rowpart  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S_ = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ctx = _hip.default_context()
m = blob_mesh(N, seed=0)
full = _hip.DeviceLaplacian(m.points, m.faces, ctx=ctx)
d = full.download()
S, sg = rowpart.symmetric_operator(d["rowptr"], d["colidx"], d["w"], d["deg"])
order = rowpart.morton_order(m.points)
t0 = time.perf_counter()
layouts = rowpart.build_all_layouts(S, order, P, S_)
print("layouts for %d ranks built on the host in %.2f s" % (P, time.perf_counter() - t0))
lay = layouts[0]
loc = _hip.DeviceLaplacian(matrix=(lay.S_local.indptr, lay.S_local.indices, lay.S_local.data), ctx=ctx)


def steps_per_us(g, n_steps=320):
    g.ws_ensure(4)
    g.start_vector(0, 1)
    g.start_vector(1, 2)
    g.cheb_steps(0, 1, 2, 16, 1.0, 1.0, 1.0)
    ctx.sync()
    t0 = time.perf_counter()
    g.cheb_steps(0, 1, 2, n_steps, 1.0, 1.0, 1.0)
    ctx.sync()
    return 1e6 * (time.perf_counter() - t0) / n_steps


us_full, us_loc = steps_per_us(full), steps_per_us(loc)
pub = loc.rows_create(lay.publish)
n_pub = max(len(l.publish) for l in layouts)
dst = np.concatenate([d_ for _, (_, d_) in sorted(lay.fill.items())])
off = np.concatenate([q * 2 * n_pub + src for q, (src, _) in sorted(lay.fill.items())])
ghost_rows = loc.rows_create(dst)
loc.rows_set_sources(ghost_rows, off)
send = torch.zeros((2, n_pub), dtype=torch.float64, device="cuda")
recv = torch.zeros((P, 2, n_pub), dtype=torch.float64, device="cuda")
stream = torch.cuda.ExternalStream(ctx.stream_ptr)


def exchange():
    loc.rows_gather2_dev(0, 1, pub, send.data_ptr(), n_pub)
    with torch.cuda.stream(stream):  # stands in for the all-gather, enqueued on the library's stream
        recv.copy_(send.unsqueeze(0).expand(P, 2, n_pub))
    loc.rows_scatter2_dev(0, 1, ghost_rows, recv.data_ptr(), n_pub)


exchange()
t0 = time.perf_counter()
ctx.sync()
t0 = time.perf_counter()
for _ in range(200):
    exchange()
ctx.sync()
us_x = 1e6 * (time.perf_counter() - t0) / 200
bytes_x = 16 * len(lay.publish)
print("N=%d P=%d S=%d: rank 0 holds %d own + %d ghost rows (%.1f %% redundant), publishes %d rows (%.0f KB per exchange)"
      % (N, P, S_, lay.n_own, lay.n_local - lay.n_own, 100.0 * (lay.n_local - lay.n_own) / lay.n_own, len(lay.publish), bytes_x / 1e3))
print("  single device, whole mesh : %.2f us per step" % us_full)
print("  one rank, chunk + ghosts  : %.2f us per step" % us_loc)
print("  exchange, device side     : %.1f us (+ the RCCL all-gather of %.0f KB x %d ranks, ~10-20 us on xGMI)" % (us_x, bytes_x / 1e3, P))
model = us_loc + (us_x + 15.0) / S_
print("  model per step            : %.2f us  ->  %.2fx the single device on %d GPUs" % (model, us_full / model, P))
