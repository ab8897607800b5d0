#!/usr/bin/env python3
"""pf_orth_split against numpy: Gram-Schmidt over two slot ranges, single graph and pair, with and without the device's
second pass, with a step that cancels digits (host-side second pass)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfocusr_amd import _hip  # noqa: E402
from pyfocusr_amd.meshgen import blob_mesh  # noqa: E402

ctx = _hip.default_context()
rng = np.random.default_rng(1)
m = blob_mesh(31250, seed=5)
g = _hip.DeviceLaplacian(m.points, m.faces, ctx=ctx)
n = g.n
g.ws_ensure(16)
Q, _ = np.linalg.qr(rng.standard_normal((n, 8)))
for s in range(8):
    g.upload(s, Q[:, s])
for case, (wvec, passes) in enumerate([(rng.standard_normal(n), 0), (rng.standard_normal(n), 1),
                                       (Q[:, 6] * 1e3 + Q[:, 1] * 50 + 1e-3 * rng.standard_normal(n), 0),
                                       (Q[:, 6] * 1e3 + Q[:, 1] * 50 + 1e-3 * rng.standard_normal(n), 1)]):
    g.upload(9, wvec)
    g.orth_device_passes(bool(passes))
    g.orth_split(5, 2)           # basis: slots 0, 1 and 5, 6
    g.orth_begin(9, 0, 4, True)
    h, nrm = g.orth_end()
    got = g.download_slots(9, 1)[:, 0]
    B = Q[:, [0, 1, 5, 6]]
    href = B.T @ wvec
    wref = wvec - B @ href
    h2 = B.T @ wref
    wref -= B @ h2
    nref = np.linalg.norm(wref)
    print("case %d passes %d: h err %.1e nrm err %.1e vec err %.1e redone %s twice %s | left-over along basis %.1e, along skipped slots %.1e" % (
        case, passes, np.max(np.abs(h - (href + h2))) / np.max(np.abs(href)), abs(nrm - nref) / nref, np.max(np.abs(got - wref / nref)),
        g.orth_redone, g.orth_twice, np.max(np.abs(B.T @ got)), np.max(np.abs(Q[:, [2, 3, 4, 7]].T @ got))))
