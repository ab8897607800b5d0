"""ctypes wrapper of the CPU TEST DOUBLE of the C++ Krylov driver (tests/csrc/krylov_double.cpp): the driver headers of
the product (pyfocusr_amd/csrc/pf_krylov.h, pf_dense.h) compiled with g++ over plain host loops.  Test infrastructure:
never imported by the package."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "krylov_double.cpp")
LIB = os.path.join(HERE, "_build", "libpf_krylov_double.so")
HDRS = [os.path.join(HERE, "..", "pyfocusr_amd", "csrc", h) for h in ("pf_krylov.h", "pf_dense.h")] + \
       [os.path.join(HERE, "..", "include", "pyfocusr_hip.h")]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class Stats(C.Structure):  # pf_eigs_stats of include/pyfocusr_hip.h
    _fields_ = [("matvecs", C.c_int64), ("outer_steps", C.c_int32), ("restarts", C.c_int32), ("filter_resets", C.c_int32),
                ("degree", C.c_int32), ("n_null", C.c_int32), ("cut", C.c_double), ("max_residual", C.c_double),
                ("second_passes", C.c_int32), ("mode", C.c_int32), ("local_steps", C.c_int32), ("reserved", C.c_int32)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


def build(force=False):
    stale = force or not os.path.exists(LIB) or any(os.path.getmtime(p) > os.path.getmtime(LIB) for p in [SRC] + HDRS)
    if stale:
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-Wno-unused-but-set-variable", "-o", LIB, SRC], check=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.td_last_error.restype = C.c_char_p
        _lib.td_hessenberg_residual_factor.restype = C.c_double
        _lib.td_hessenberg_residual_factor.argtypes = [C.c_int32, _dp, C.c_int32, C.c_double, C.c_double]
    return _lib


class DoubleError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "%s (code %d)" % (msg, code))
        self.code = code


def _csr(W):
    W = W.tocsr().astype(np.float64)
    W.sort_indices()
    return (W.shape[0], np.ascontiguousarray(W.indptr, dtype=np.int32), np.ascontiguousarray(W.indices, dtype=np.int32),
            np.ascontiguousarray(W.data, dtype=np.float64))


def solve(W, n_wanted, ellipse_hint=-1, redone_every=0, m_max_limit=0):
    """The C++ driver on the test double: (vals, vecs (n, m) eigenvectors of L, stats dict, residuals)."""
    n, rp, ci, w = _csr(W)
    vals, vecs, res = np.zeros(n_wanted), np.zeros((n, n_wanted)), np.zeros(n_wanted)
    n_out, st = C.c_int32(), Stats()
    rc = lib().td_solve(C.c_int64(n), rp.ctypes.data_as(_ip), ci.ctypes.data_as(_ip), w.ctypes.data_as(_dp), n_wanted, ellipse_hint,
                        redone_every, m_max_limit, vals.ctypes.data_as(_dp), vecs.ctypes.data_as(_dp), res.ctypes.data_as(_dp), C.byref(n_out), C.byref(st))
    if rc != 0:
        raise DoubleError(rc, lib().td_last_error().decode())
    m = n_out.value
    return vals[:m].copy(), vecs.reshape(-1)[: n * m].reshape(n, m).copy(), st.as_dict(), res[:m].copy()


def solve_pair(Wa, wanted_a, Wb, wanted_b):
    a, b = _csr(Wa), _csr(Wb)
    outs = []
    for (n, _, _, _), m in ((a, wanted_a), (b, wanted_b)):
        outs.append((np.zeros(m), np.zeros((n, m)), C.c_int32(), Stats()))
    calls = C.c_int64()
    args = []
    for (n, rp, ci, w), m, (vals, vecs, n_out, st) in ((a, wanted_a, outs[0]), (b, wanted_b, outs[1])):
        args += [C.c_int64(n), rp.ctypes.data_as(_ip), ci.ctypes.data_as(_ip), w.ctypes.data_as(_dp), m, vals.ctypes.data_as(_dp),
                 vecs.ctypes.data_as(_dp), C.byref(n_out), C.byref(st)]
    rc = lib().td_solve_pair(*args, C.byref(calls))
    if rc != 0:
        raise DoubleError(rc, lib().td_last_error().decode())
    res = []
    for (n, _, _, _), (vals, vecs, n_out, st) in ((a, outs[0]), (b, outs[1])):
        m = n_out.value
        res.append((vals[:m].copy(), vecs.reshape(-1)[: n * m].reshape(n, m).copy(), st.as_dict()))
    return res[0], res[1], calls.value
