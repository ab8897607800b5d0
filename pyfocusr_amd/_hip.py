"""ctypes binding of libpyfocusr_hip.so (C-ABI in include/pyfocusr_hip.h).

There is no CPU fallback: if the shared library is missing, or no MI355X is
visible, the product path raises.  The library is built in-tree by
`__graft_entry__.build()` (hipcc --offload-arch=gfx950).
"""
import atexit
import ctypes as C
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PYFOCUSR_HIP_LIB", os.path.join(_HERE, "csrc", "libpyfocusr_hip.so"))  # env: tuning builds

PF_OP_RW = 0
PF_OP_SYM = 1

_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


class HipUnavailable(RuntimeError):
    pass


class PfError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libpyfocusr_hip error %d: %s" % (code, msg))
        self.code = code


class GraphInfo(C.Structure):
    _fields_ = [("n", C.c_int64), ("n_faces", C.c_int64), ("nnz_w", C.c_int64), ("nnz_l", C.c_int64),
                ("is_symmetric", C.c_int32), ("n_isolated", C.c_int32), ("n_components", C.c_int32),
                ("max_degree", C.c_int32), ("sell_entries", C.c_int64), ("n_pad", C.c_int64), ("n_oneway", C.c_int64),
                ("spectral_bound", C.c_double)]


class EigsStats(C.Structure):
    _fields_ = [("matvecs", C.c_int64), ("outer_steps", C.c_int32), ("restarts", C.c_int32), ("filter_resets", C.c_int32),
                ("degree", C.c_int32), ("n_null", C.c_int32), ("cut", C.c_double), ("max_residual", C.c_double),
                ("second_passes", C.c_int32), ("mode", C.c_int32), ("local_steps", C.c_int32), ("reserved", C.c_int32)]


class Timing(C.Structure):
    _fields_ = [("op_ms", C.c_double), ("op_launches", C.c_int64), ("op_bytes", C.c_double), ("knn_ms", C.c_double),
                ("build_ms", C.c_double), ("persist_ms", C.c_double), ("persist_launches", C.c_int64),
                ("persist_steps", C.c_int64), ("persist_bytes", C.c_double), ("persist_lds_bytes", C.c_double)]


# name -> (restype, argtypes): every symbol include/pyfocusr_hip.h declares.
SIGNATURES = {
    "pf_version": (C.c_int, []),
    "pf_last_error": (C.c_char_p, []),
    "pf_device_count": (C.c_int, []),
    "pf_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "pf_destroy": (None, [C.c_void_p]),
    "pf_sync": (C.c_int, [C.c_void_p]),
    "pf_stream": (C.c_void_p, [C.c_void_p]),
    "pf_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "pf_timing_get": (C.c_int, [C.c_void_p, C.POINTER(Timing), C.c_int]),
    "pf_graph_build": (C.c_int, [C.c_void_p, _f64p, C.c_int64, _i32p, C.c_int64, C.c_int32, C.POINTER(C.c_void_p)]),
    "pf_mesh_upload": (C.c_int, [C.c_void_p, _f64p, C.c_int64, _i32p, C.c_int64, C.c_int32, C.POINTER(C.c_void_p)]),
    "pf_mesh_free": (None, [C.c_void_p]),
    "pf_graph_build_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "pf_graph_build_device2": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "pf_graph_from_matrix": (C.c_int, [C.c_void_p, C.c_int64, _i32p, _i32p, _f64p, C.POINTER(C.c_void_p)]),
    "pf_graph_free": (None, [C.c_void_p]),
    "pf_graph_get_info": (C.c_int, [C.c_void_p, C.POINTER(GraphInfo)]),
    "pf_graph_download": (C.c_int, [C.c_void_p, _i32p, _i32p, _f64p, _f64p, _f64p, _f64p, _i32p]),
    "pf_ws_ensure": (C.c_int, [C.c_void_p, C.c_int32]),
    "pf_ws_upload": (C.c_int, [C.c_void_p, C.c_int32, _f64p]),
    "pf_ws_download": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _f64p]),
    "pf_ws_copy": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "pf_start_vector": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint64]),
    "pf_mask_isolated": (C.c_int, [C.c_void_p, C.c_int32]),
    "pf_lock_null_vectors": (C.c_int, [C.c_void_p, C.c_int32, _i32p]),
    "pf_spmv": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "pf_spmv_multi": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "pf_knn_mode": (C.c_int, [C.c_void_p, C.c_int32]),
    "pf_knn_tree_stats": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pf_persist_enable": (C.c_int, [C.c_int]),
    "pf_persist_two_step": (C.c_int, [C.c_int]),
    "pf_persist_pair_halves": (C.c_int, [C.c_int]),
    "pf_persist_clock": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
    "pf_persist_state": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pf_persist_test_hook": (C.c_int, [C.c_int]),
    "pf_cheb": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double]),
    "pf_cheb2": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double,
                           C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double]),
    "pf_dots": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _f64p]),
    "pf_orth": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _f64p, _f64p]),
    "pf_orth_begin": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "pf_orth_end": (C.c_int, [C.c_void_p, _f64p, _f64p]),
    "pf_orth_split": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "pf_orth_one_launch": (C.c_int, [C.c_int32]),
    "pf_orth_begin2": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_int32]),
    "pf_orth_cheb2": (C.c_int, [C.c_void_p, C.c_void_p, _i32p, _i32p, _f64p]),
    "pf_eigsort_costs": (C.c_int, [C.c_void_p, C.c_void_p, _i64p, C.c_int64, _i64p, C.c_int64, C.c_int32, _i32p, _f64p, _i32p, _f64p, _f64p,
                                   _i64p]),
    "pf_orth_redone": (C.c_int, [C.c_void_p]),
    "pf_orth_strict": (C.c_int, [C.c_void_p, C.c_int32]),
    "pf_scale": (C.c_int, [C.c_void_p, C.c_int32, C.c_double]),
    "pf_combine": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _f64p, C.c_int32, C.c_int32]),
    "pf_resnorm": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, _f64p]),
    "pf_gram": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f64p]),
    "pf_resnorms": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _f64p, C.c_int32, _f64p]),
    "pf_finalize_vectors": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f64p]),
    "pf_finalize_vectors_begin": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f64p]),
    "pf_finalize_vectors_end": (C.c_int, [C.c_void_p]),
    "pf_final_remap_begin": (C.c_int, [C.c_void_p, _i32p, _f64p, C.c_int32, _f64p]),
    "pf_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "pf_host_free": (C.c_int, [C.c_void_p]),
    "pf_final_rows": (C.c_int, [C.c_void_p, _i64p, C.c_int64, _f64p]),
    "pf_point_rows": (C.c_int, [C.c_void_p, _i64p, C.c_int64, _f64p]),
    "pf_spmv_host": (C.c_int, [C.c_void_p, C.c_int32, _f64p, _f64p]),
    "pf_mean_filter": (C.c_int, [C.c_void_p, _f64p, C.c_int32, C.c_int32, _f64p]),
    "pf_knn1": (C.c_int, [C.c_void_p, _f64p, C.c_int64, _f64p, C.c_int64, C.c_int32, _i64p, _f64p]),
    "pf_knn": (C.c_int, [C.c_void_p, _f64p, C.c_int64, _f64p, C.c_int64, C.c_int32, C.c_int32, _i64p, _f64p]),
    "pf_knn1_graphs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, _i32p, _f64p, _i32p, _f64p, _i64p, _f64p]),
    "pf_knn1_blocks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                 _i32p, _f64p, _i32p, _f64p, _i64p, _f64p]),
    "pf_final_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), _i64p, _i32p]),
    "pf_knn_upload": (C.c_int, [C.c_void_p, _f64p, C.c_int64, _f64p, C.c_int64, C.c_int32]),
    "pf_knn_run": (C.c_int, [C.c_void_p]),
    "pf_knn_download": (C.c_int, [C.c_void_p, _i64p, _f64p]),
    "pf_eigs_smallest": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _f64p, _f64p, C.POINTER(C.c_int32), C.POINTER(EigsStats)]),
    "pf_knn_count": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]),
    "pf_host_detach": (C.c_int, [C.c_void_p]),
    "pf_orth_device_passes": (C.c_int, [C.c_void_p, C.c_int32]),
    "pf_eigs_smallest_ex": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _f64p, _f64p, _f64p, C.POINTER(C.c_int32),
                                      C.POINTER(EigsStats)]),
    "pf_eigs_smallest2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    _f64p, _f64p, _f64p, C.POINTER(C.c_int32), C.POINTER(EigsStats),
                                    _f64p, _f64p, _f64p, C.POINTER(C.c_int32), C.POINTER(EigsStats)]),
    "pf_op_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double]),
    "pf_cheb_steps": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "pf_axpy": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _f64p]),
    "pf_rows_create": (C.c_int, [C.c_void_p, _i64p, C.c_int64, C.POINTER(C.c_void_p)]),
    "pf_rows_free": (None, [C.c_void_p]),
    "pf_rows_gather": (C.c_int, [C.c_void_p, C.c_int32, _f64p]),
    "pf_rows_scatter": (C.c_int, [C.c_void_p, C.c_int32, _f64p]),
    "pf_rows_gather_dev": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "pf_rows_scatter_dev": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "pf_rows_set_sources": (C.c_int, [C.c_void_p, _i64p]),
    "pf_rows_gather2_dev": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]),
    "pf_rows_scatter2_dev": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]),
    "pf_rows_fill": (C.c_int, [C.c_void_p, C.c_int32, C.c_double]),
    "pf_surface_create": (C.c_int, [C.c_void_p, _f64p, C.c_int64, _i32p, C.c_int64, C.c_int32, C.POINTER(C.c_void_p)]),
    "pf_surface_free": (None, [C.c_void_p]),
    "pf_surface_closest": (C.c_int, [C.c_void_p, _f64p, C.c_int64, _f64p, _i32p, _f64p]),
    "pf_cpd_create": (C.c_int, [C.c_void_p, _f64p, C.c_int64, _f64p, C.c_int64, C.c_int32, C.POINTER(C.c_void_p)]),
    "pf_cpd_free": (None, [C.c_void_p]),
    "pf_cpd_estep": (C.c_int, [C.c_void_p, _f64p, C.c_double, C.c_double, _f64p, _f64p, _f64p]),
    "pf_cpd_set_basis": (C.c_int, [C.c_void_p, _f64p, C.c_int32]),
    "pf_cpd_weighted_gram": (C.c_int, [C.c_void_p, _f64p]),
    "pf_cpd_affine_sums": (C.c_int, [C.c_void_p, _f64p, _f64p]),
    "pf_cpd_apply_affine": (C.c_int, [C.c_void_p, _f64p, _f64p]),
    "pf_cpd_deform_sums": (C.c_int, [C.c_void_p, _f64p, _f64p]),
    "pf_cpd_apply_deform": (C.c_int, [C.c_void_p, _f64p, _f64p]),
    "pf_cpd_download": (C.c_int, [C.c_void_p, _f64p, _f64p, _f64p, _f64p]),
    "pf_cpd_gram": (C.c_int, [C.c_void_p, _f64p, C.c_int64, _f64p, C.c_int64, C.c_int32, C.c_double, _f64p, C.c_int32, _f64p]),
}

_lib = None
_lib_lock = threading.Lock()
_live_graphs = weakref.WeakSet()
_live_contexts = weakref.WeakSet()


@atexit.register
def _shutdown():
    """Release device objects while the HIP runtime is still alive (its own static
    destructors run after Python's, and freeing into a torn-down runtime aborts)."""
    for g in list(_live_graphs):
        g.close()
    for c in list(_live_contexts):
        c.close()


def load_library():
    """dlopen the in-tree library and bind every declared symbol (no GPU needed)."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipUnavailable(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


PF_E_DEGENERATE = -3
PF_E_STATE = -4
PF_E_PERSIST_TIMEOUT = -5


class _PinnedBlock(object):
    """One hipHostMalloc block; goes back to the module's free list when the last array over it is collected."""
    __slots__ = ("ptr", "nbytes")

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes

    def __del__(self):
        try:
            _pinned_release(self.ptr, self.nbytes)
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


_pinned_free = {}  # nbytes -> [ptr]: blocks of collected arrays, reused by the next result of the same size
_pinned_cached = [0]
_pinned_lock = threading.Lock()  # (compute_spectra runs one host thread per context; arrays are collected on any thread)
# page-locked memory kept for reuse.  A 1M-vertex pair with k = 10 turns over four 84 MB blocks per pipeline step: with the
# 256 MiB of round 3 whether they fitted beside what smaller workloads had left behind decided between 80 and 90 ms per
# step (two hipHostMalloc / hipHostFree of 84 MB each) - 1 GiB, and blocks of OTHER sizes make room first.
_PINNED_CACHE_BYTES = 1 << 30


def _pinned_release(ptr, nbytes):
    if _pinned_free is None or _lib is None:
        return
    # a download the library still OWES to this block (pf_finalize_vectors_begin holds downloads back; a failed call may
    # have left one behind) must never be queued once the block can be handed out again; one in flight is waited for
    _lib.pf_host_detach(C.c_void_p(ptr))
    evicted = []
    with _pinned_lock:
        if nbytes <= _PINNED_CACHE_BYTES:
            # the sizes in use now are the ones worth keeping: blocks of other sizes go first (largest first)
            while _pinned_cached[0] + nbytes > _PINNED_CACHE_BYTES:
                others = [sz for sz, ptrs in _pinned_free.items() if ptrs and sz != nbytes]
                if not others:
                    break
                sz = max(others)
                evicted.append(_pinned_free[sz].pop())
                _pinned_cached[0] -= sz
        keep = _pinned_cached[0] + nbytes <= _PINNED_CACHE_BYTES
        if keep:
            _pinned_free.setdefault(nbytes, []).append(ptr)
            _pinned_cached[0] += nbytes
    for other in evicted:
        _lib.pf_host_free(C.c_void_p(other))
    if not keep:
        _lib.pf_host_free(C.c_void_p(ptr))


def pinned_trim():
    """Give the cached page-locked blocks back to the system (they are kept for reuse otherwise, up to 1 GiB)."""
    with _pinned_lock:
        blocks = [p for ptrs in _pinned_free.values() for p in ptrs]
        _pinned_free.clear()
        _pinned_cached[0] = 0
    for ptr in blocks:
        _lib.pf_host_free(C.c_void_p(ptr))


def pinned_empty(shape, dtype=np.float64):
    """`np.empty(shape, dtype)` over page-locked host memory (pf_host_alloc): device results land in it by one DMA
    instead of the runtime's chunked staging of a pageable destination (hipHostMalloc itself costs 0.1-1 ms: the blocks
    of collected arrays are kept and reused).  Ordinary cacheable host memory for every other purpose."""
    lib = load_library()
    dtype = np.dtype(dtype)
    count = int(np.prod(shape)) if len(shape) else 1
    nbytes = max(count * dtype.itemsize, 1)
    nbytes = (nbytes + 4095) & ~4095
    ptr = None
    with _pinned_lock:
        free = _pinned_free.get(nbytes)
        if free:
            ptr = free.pop()
            _pinned_cached[0] -= nbytes
    if ptr is None:
        p = C.c_void_p()
        _check(lib.pf_host_alloc(C.c_size_t(nbytes), C.byref(p)))
        ptr = int(p.value)
    buf = (C.c_char * nbytes).from_address(ptr)
    buf._pf_block = _PinnedBlock(ptr, nbytes)  # the array's base keeps `buf`, and with it the block, alive
    return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)


def persist_enable(on=True):
    """Process-wide switch of the resident Chebyshev kernel (operator in registers, x in LDS, one kernel per filter
    application; on by default, see csrc/pf_persist.hip).  Results are bit-identical either way."""
    _check(load_library().pf_persist_enable(int(bool(on))))


def orth_one_launch(on=True):
    """Process-wide switch of the one-launch local Gram-Schmidt step (csrc/pf_operator.hip: k_orth_local); off: dot
    products and projection as two launches, like every other step.  Results are bit-identical either way."""
    _check(load_library().pf_orth_one_launch(int(bool(on))))


def persist_two_step(level=1):
    """Process-wide level of the two-steps-per-exchange form of the resident kernel (csrc/pf_persist.hip:
    k_cheb_resident2): 0 off, 1 single-graph recurrences (default), 2 paired recurrences too.  Results are
    bit-identical at every level."""
    _check(load_library().pf_persist_two_step(int(level)))


def persist_pair_halves(on=True):
    """Process-wide switch of the pair kernel whose halves take the two graphs in opposite order (csrc/pf_persist.hip:
    k_cheb_resident<2,1,8,true>; on by default).  Results are bit-identical either way."""
    _check(load_library().pf_persist_pair_halves(int(bool(on))))


def persist_clock(ctx, reset=False):
    """(summed run time in ms, number) of ctx's completed resident launches since the last reset, on the device's own
    100 MHz clock inside the kernel (csrc/pf_persist.hip: pf_persist_clock).  Synchronises the ctx stream."""
    ms, n = C.c_double(0.0), C.c_int64(0)
    _check(load_library().pf_persist_clock(ctx._h, C.byref(ms), C.byref(n), int(bool(reset))))
    return ms.value, n.value


class _PersistInfo(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("two_step", C.c_int32), ("owner", C.c_int32), ("timeouts", C.c_int32),
                ("launches", C.c_int64), ("launches_two_step", C.c_int64), ("suspended_for", C.c_int32), ("rearms", C.c_int32),
                ("owner_switches", C.c_int32), ("hold_ticks", C.c_int32)]


def persist_state(ctx=None):
    """pf_persist_state as a dict: is the resident path on, who owns it, how many waits ran out, launch counters."""
    info = _PersistInfo()
    _check(load_library().pf_persist_state(None if ctx is None else ctx._h, C.byref(info)))
    return {name: int(getattr(info, name)) for name, _ in _PersistInfo._fields_}


def persist_test_hook(n_launches=1):
    """Test hook: the next `n_launches` resident launches give up at once (PF_E_PERSIST_TIMEOUT recovery)."""
    _check(load_library().pf_persist_test_hook(int(n_launches)))


def _check(code):
    if code != 0:
        raise PfError(code, load_library().pf_last_error().decode("utf-8", "replace"))


def _f64(a):
    return a.ctypes.data_as(_f64p)


def _c_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError("expected shape %r, got %r" % (shape, a.shape))
    return a


class Context(object):
    """One HIP stream on one device."""

    def __init__(self, device=0):
        lib = load_library()
        if lib.pf_device_count() <= 0:
            raise HipUnavailable("no HIP device visible: the pyfocusr_amd hot path needs an MI355X (no CPU fallback)")
        h = C.c_void_p()
        _check(lib.pf_create(int(device), C.byref(h)))
        self._lib = lib
        self._h = h
        self.device = int(device)
        self._children = weakref.WeakSet()  # graphs / meshes allocated from this context
        _live_contexts.add(self)

    def close(self):
        if getattr(self, "_h", None):
            for child in list(self._children):  # device objects must go before their allocator
                child.close()
            self._lib.pf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def sync(self):
        _check(self._lib.pf_sync(self._h))

    @property
    def stream_ptr(self):
        """Address of the ctx's hipStream_t (for `torch.cuda.ExternalStream`: a collective enqueued on it is ordered
        with the library's kernels without any host synchronisation)."""
        return int(self._lib.pf_stream(self._h) or 0)

    def timing_enable(self, on=True):
        """HIP-event timing of the filter applications: True / 1 every application, N > 1 every N-th, False off."""
        _check(self._lib.pf_timing_enable(self._h, int(on)))

    def timing(self, reset=False):
        t = Timing()
        _check(self._lib.pf_timing_get(self._h, C.byref(t), int(bool(reset))))
        return dict(op_ms=t.op_ms, op_launches=int(t.op_launches), op_bytes=t.op_bytes, knn_ms=t.knn_ms,
                    build_ms=t.build_ms, persist_ms=t.persist_ms, persist_launches=int(t.persist_launches),
                    persist_steps=int(t.persist_steps), persist_bytes=t.persist_bytes, persist_lds_bytes=t.persist_lds_bytes)

    # ---- nearest neighbour -------------------------------------------------------------
    def knn_mode(self, mode):
        """How 1-NN searches prune: 0 by depth (grid for d <= 6, box hierarchy for d >= 7), 1 always the grid, 2 always
        the hierarchy.  The results are the same bits."""
        _check(self._lib.pf_knn_mode(self._h, int(mode)))

    def knn_count(self, enable_counting=False):
        """Candidate-query pairs evaluated by the last counted grid search (`pf_knn_count`); sets the switch for the next."""
        pairs = C.c_int64()
        _check(self._lib.pf_knn_count(self._h, int(bool(enable_counting)), C.byref(pairs)))
        return int(pairs.value)

    def knn_tree_stats(self, enable_counting=False):
        """(leaves scanned, supers opened) of the last box-hierarchy search, if counting was on for it; sets the switch."""
        a, b = C.c_int64(), C.c_int64()
        _check(self._lib.pf_knn_tree_stats(self._h, int(bool(enable_counting)), C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def knn1(self, ref, qry, return_d2=False):
        """Index (int64) of the nearest `ref` row for every `qry` row."""
        ref, qry = _c_f64(ref), _c_f64(qry)
        if ref.ndim != 2 or qry.ndim != 2 or ref.shape[1] != qry.shape[1]:
            raise ValueError("ref and qry must be (n, d) arrays with equal d")
        idx = np.empty(qry.shape[0], dtype=np.int64)
        d2 = np.empty(qry.shape[0], dtype=np.float64) if return_d2 else None
        _check(self._lib.pf_knn1(self._h, _f64(ref), ref.shape[0], _f64(qry), qry.shape[0], ref.shape[1],
                                 idx.ctypes.data_as(_i64p), _f64(d2) if return_d2 else None))
        return (idx, d2) if return_d2 else idx

    def knn(self, ref, qry, k):
        """(idx (n_qry, k) int64, squared distances (n_qry, k)), ascending by (distance, index); k <= 4, d <= 4."""
        ref, qry = _c_f64(ref), _c_f64(qry)
        if ref.ndim != 2 or qry.ndim != 2 or ref.shape[1] != qry.shape[1]:
            raise ValueError("ref and qry must be (n, d) arrays with equal d")
        idx = np.empty((qry.shape[0], int(k)), dtype=np.int64)
        d2 = np.empty((qry.shape[0], int(k)), dtype=np.float64)
        _check(self._lib.pf_knn(self._h, _f64(ref), ref.shape[0], _f64(qry), qry.shape[0], ref.shape[1], int(k),
                                idx.ctypes.data_as(_i64p), _f64(d2)))
        return idx, d2

    def knn1_graphs(self, dev_ref, dev_qry, col_ref, scale_ref, col_qry, scale_qry, return_d2=False):
        """`knn1` on coordinates built on the device from the two graphs' resident eigenvector blocks:
        ref[:, c] = final_ref[:, col_ref[c]] * scale_ref[c], qry likewise."""
        col_ref = np.ascontiguousarray(col_ref, dtype=np.int32)
        col_qry = np.ascontiguousarray(col_qry, dtype=np.int32)
        scale_ref, scale_qry = _c_f64(scale_ref), _c_f64(scale_qry)
        d = len(col_ref)
        if not (len(col_qry) == len(scale_ref) == len(scale_qry) == d):
            raise ValueError("col / scale arrays must share one length d")
        idx = np.empty(dev_qry.n, dtype=np.int64)
        d2 = np.empty(dev_qry.n, dtype=np.float64) if return_d2 else None
        _check(self._lib.pf_knn1_graphs(dev_ref._h, dev_qry._h, d, col_ref.ctypes.data_as(_i32p), _f64(scale_ref),
                                        col_qry.ctypes.data_as(_i32p), _f64(scale_qry), idx.ctypes.data_as(_i64p),
                                        _f64(d2) if return_d2 else None))
        return (idx, d2) if return_d2 else idx

    def eigsort_costs(self, dev_t, dev_s, rows_t, rows_s, k, col_t, sign_t, col_s, sign_s):
        """eigsort's c_hist, c_hist_f, c_spatial, c_spatial_f (each k x k) and the 3-D 1-NN indices of the sampled points,
        computed on the device from the two graphs' resident eigenvector blocks (`pf_eigsort_costs`)."""
        rows_t = np.ascontiguousarray(rows_t, dtype=np.int64)
        rows_s = np.ascontiguousarray(rows_s, dtype=np.int64)
        col_t = np.ascontiguousarray(col_t, dtype=np.int32)
        col_s = np.ascontiguousarray(col_s, dtype=np.int32)
        sign_t, sign_s = _c_f64(sign_t), _c_f64(sign_s)
        if min(len(col_t), len(col_s), len(sign_t), len(sign_s)) < k:
            raise ValueError("eigsort_costs: k columns / signs per graph are needed")
        out = np.empty((4, int(k), int(k)), dtype=np.float64)
        idx = np.empty(len(rows_t), dtype=np.int64)
        _check(self._lib.pf_eigsort_costs(dev_t._h, dev_s._h, rows_t.ctypes.data_as(_i64p), len(rows_t), rows_s.ctypes.data_as(_i64p),
                                          len(rows_s), int(k), col_t.ctypes.data_as(_i32p), _f64(sign_t), col_s.ctypes.data_as(_i32p),
                                          _f64(sign_s), _f64(out), idx.ctypes.data_as(_i64p)))
        return out, idx

    def knn1_blocks(self, ref_ptr, n_ref, ref_stride, qry_ptr, n_qry, qry_stride, col_ref, scale_ref, col_qry, scale_qry,
                    return_d2=False):
        """`knn1` on coordinates built from two row-major float64 blocks already in this device's memory (integer
        addresses, row strides in doubles): block[:, col[c]] * scale[c] for c < d."""
        col_ref = np.ascontiguousarray(col_ref, dtype=np.int32)
        col_qry = np.ascontiguousarray(col_qry, dtype=np.int32)
        scale_ref, scale_qry = _c_f64(scale_ref), _c_f64(scale_qry)
        d = len(col_ref)
        if not (len(col_qry) == len(scale_ref) == len(scale_qry) == d):
            raise ValueError("col / scale arrays must share one length d")
        idx = np.empty(int(n_qry), dtype=np.int64)
        d2 = np.empty(int(n_qry), dtype=np.float64) if return_d2 else None
        _check(self._lib.pf_knn1_blocks(self._h, C.c_void_p(int(ref_ptr)), int(n_ref), int(ref_stride), C.c_void_p(int(qry_ptr)),
                                        int(n_qry), int(qry_stride), d, col_ref.ctypes.data_as(_i32p), _f64(scale_ref),
                                        col_qry.ctypes.data_as(_i32p), _f64(scale_qry), idx.ctypes.data_as(_i64p),
                                        _f64(d2) if return_d2 else None))
        return (idx, d2) if return_d2 else idx

    def knn_upload(self, ref, qry):
        ref, qry = _c_f64(ref), _c_f64(qry)
        _check(self._lib.pf_knn_upload(self._h, _f64(ref), ref.shape[0], _f64(qry), qry.shape[0], ref.shape[1]))
        self._knn_nq = qry.shape[0]

    def knn_run(self):
        _check(self._lib.pf_knn_run(self._h))

    def knn_download(self):
        idx = np.empty(self._knn_nq, dtype=np.int64)
        d2 = np.empty(self._knn_nq, dtype=np.float64)
        _check(self._lib.pf_knn_download(self._h, idx.ctypes.data_as(_i64p), _f64(d2)))
        return idx, d2


_default_ctx = {}


def default_context(device=None):
    """Process-wide context per device (LOCAL_RANK selects the device under torchrun)."""
    if device is None:
        device = int(os.environ.get("PYFOCUSR_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = load_library().pf_device_count()
        if n > 0:
            device %= n
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class DeviceMesh(object):
    """points (n,3) f64 + faces (F,v) i32 resident in HBM (input side of the assembler)."""

    def __init__(self, points, faces, ctx=None):
        self.ctx = ctx if ctx is not None else default_context()
        self._lib = self.ctx._lib
        pts = _c_f64(points).reshape(-1, 3)
        f = np.ascontiguousarray(faces, dtype=np.int32)
        if f.ndim != 2:
            raise ValueError("faces must be (F, verts_per_face)")
        h = C.c_void_p()
        _check(self._lib.pf_mesh_upload(self.ctx._h, _f64(pts), pts.shape[0], f.ctypes.data_as(_i32p), f.shape[0],
                                        f.shape[1] if f.shape[0] else 3, C.byref(h)))
        self._h = h
        self.n, self.n_faces = pts.shape[0], f.shape[0]
        _live_graphs.add(self)
        self.ctx._children.add(self)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pf_mesh_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class DeviceSurface(object):
    """Triangle soup of one mesh in HBM (Morton-sorted, chunk boxes) for exact closest-point queries
    (`pf_surface_*`): the search inside the ICP pre-alignment."""

    def __init__(self, points, faces, ctx=None):
        self.ctx = ctx if ctx is not None else default_context()
        self._lib = self.ctx._lib
        pts = _c_f64(points).reshape(-1, 3)
        f = np.ascontiguousarray(faces, dtype=np.int32)
        if f.ndim != 2 or f.shape[0] == 0:
            raise ValueError("faces must be a non-empty (F, verts_per_face) array")
        h = C.c_void_p()
        _check(self._lib.pf_surface_create(self.ctx._h, _f64(pts), pts.shape[0], f.ctypes.data_as(_i32p), f.shape[0],
                                           f.shape[1], C.byref(h)))
        self._h = h
        self.n, self.n_faces = pts.shape[0], f.shape[0]
        _live_graphs.add(self)
        self.ctx._children.add(self)

    def closest(self, queries):
        """(points (q,3) f64, face (q,) i32, squared distance (q,) f64) of the closest surface point of each query."""
        q = _c_f64(queries).reshape(-1, 3)
        pts = np.empty_like(q)
        face = np.empty(len(q), dtype=np.int32)
        d2 = np.empty(len(q), dtype=np.float64)
        _check(self._lib.pf_surface_closest(self._h, _f64(q), len(q), _f64(pts), face.ctypes.data_as(_i32p), _f64(d2)))
        return pts, face, d2

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pf_surface_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class DeviceCpd(object):
    """Fixed set X (N,d) and moving set Y (M,d) resident in HBM for the E-steps of a CPD registration."""

    def __init__(self, X, Y, ctx=None):
        self.ctx = ctx if ctx is not None else default_context()
        self._lib = self.ctx._lib
        X, Y = _c_f64(X), _c_f64(Y)
        if X.ndim != 2 or Y.ndim != 2 or X.shape[1] != Y.shape[1]:
            raise ValueError("X (N,d) and Y (M,d) must share d")
        self.N, self.M, self.D = X.shape[0], Y.shape[0], X.shape[1]
        h = C.c_void_p()
        _check(self._lib.pf_cpd_create(self.ctx._h, _f64(X), self.N, _f64(Y), self.M, self.D, C.byref(h)))
        self._h = h
        self._P1, self._Pt1, self._PX = np.empty(self.M), np.empty(self.N), np.empty((self.M, self.D))
        _live_graphs.add(self)
        self.ctx._children.add(self)

    def estep(self, TY, sigma2, w=0.0):
        """(P1 (M,), Pt1 (N,), PX (M,d)) for the moving set at TY; the arrays are reused by the next call."""
        TY = _c_f64(TY)
        if TY.shape != (self.M, self.D):
            raise ValueError("TY must be (%d, %d)" % (self.M, self.D))
        _check(self._lib.pf_cpd_estep(self._h, _f64(TY), float(sigma2), float(w), _f64(self._P1), _f64(self._Pt1),
                                      _f64(self._PX)))
        return self._P1, self._Pt1, self._PX

    # ---- device-resident iterations: the posterior sums stay in HBM, small moments out, parameters in
    def estep_resident(self, sigma2, w=0.0):
        _check(self._lib.pf_cpd_estep(self._h, None, float(sigma2), float(w), None, None, None))

    def affine_sums(self):
        """dict of the moment sums of `pf_cpd_affine_sums` (about the centres cx, cy)."""
        D = self.D
        shifts, sums = np.empty(32), np.empty(2 * D * D + 3 * D + 3)
        _check(self._lib.pf_cpd_affine_sums(self._h, _f64(shifts), _f64(sums)))
        o = 1 + 2 * D + 2 * D * D
        return dict(cx=shifts[:D], cy=shifts[16:16 + D], Np=sums[0], sPX=sums[1:1 + D], sP1Y=sums[1 + D:1 + 2 * D],
                    PXY=sums[1 + 2 * D:1 + 2 * D + D * D].reshape(D, D), YPY=sums[1 + 2 * D + D * D:o].reshape(D, D),
                    sPt1=sums[o], sPt1XX=sums[o + 1], sPt1X=sums[o + 2:o + 2 + D])

    def apply_affine(self, B, t):
        B, t = _c_f64(B), _c_f64(t)
        _check(self._lib.pf_cpd_apply_affine(self._h, _f64(B), _f64(t)))

    def deform_sums(self):
        """(H (K,K) = Q^T diag(P1) Q, R (K,d) = Q^T (PX - diag(P1) Y))."""
        K = self._H.shape[0]
        R = np.empty((K, self.D))
        _check(self._lib.pf_cpd_deform_sums(self._h, _f64(self._H), _f64(R)))
        return self._H, R

    def apply_deform(self, Cmat):
        """TY = Y + Q C; returns (Np, yPy, trPXY, sum Pt1, xPx) with the new TY."""
        Cmat = _c_f64(Cmat)
        sums = np.empty(5)
        _check(self._lib.pf_cpd_apply_deform(self._h, _f64(Cmat), _f64(sums)))
        return sums

    def download(self):
        """(TY, P1, Pt1, PX) as they stand on the device."""
        TY = np.empty((self.M, self.D))
        _check(self._lib.pf_cpd_download(self._h, _f64(TY), _f64(self._P1), _f64(self._Pt1), _f64(self._PX)))
        return TY, self._P1, self._Pt1, self._PX

    def set_basis(self, Q):
        """Keep the low-rank basis Q (M,K) on the device for `weighted_gram`."""
        Q = _c_f64(Q)
        if Q.ndim != 2 or Q.shape[0] != self.M:
            raise ValueError("Q must be (%d, K)" % self.M)
        _check(self._lib.pf_cpd_set_basis(self._h, _f64(Q), Q.shape[1]))
        self._H = np.empty((Q.shape[1], Q.shape[1]))

    def weighted_gram(self):
        """Q^T diag(P1) Q with the P1 of the last E-step."""
        _check(self._lib.pf_cpd_weighted_gram(self._h, _f64(self._H)))
        return self._H

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pf_cpd_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def gaussian_gram_product(A, B, beta, V, ctx=None):
    """G(A,B) @ V with G_ij = exp(-|a_i - b_j|^2 / (2 beta^2)), evaluated on the device without forming G."""
    ctx = ctx if ctx is not None else default_context()
    A, B, V = _c_f64(A), _c_f64(B), _c_f64(V)
    if A.ndim != 2 or B.ndim != 2 or A.shape[1] != B.shape[1] or V.ndim != 2 or V.shape[0] != B.shape[0]:
        raise ValueError("shapes: A (a,d), B (b,d), V (b,c)")
    out = np.empty((A.shape[0], V.shape[1]))
    _check(ctx._lib.pf_cpd_gram(ctx._h, _f64(A), A.shape[0], _f64(B), B.shape[0], A.shape[1], float(beta), _f64(V),
                                V.shape[1], _f64(out)))
    return out


class _Rows(object):
    """A fixed subset of a graph's rows on the device (`pf_rows_*`)."""

    def __init__(self, owner, h, n):
        self._lib, self._h, self.n = owner._lib, h, n

    def close(self):
        if self._h:
            self._lib.pf_rows_free(self._h)
            self._h = None


class DeviceLaplacian(object):
    """Device-resident graph of one mesh: CSR(W), deg, SELL-64 operators, workspace.
    Also the `ops` object the Krylov driver (`_krylov.filtered_eigs`) drives."""

    def __init__(self, points=None, faces=None, ctx=None, device_mesh=None, matrix=None, _handle=None):
        h = C.c_void_p()
        if _handle is not None:  # (ctx, pf_graph*) of a graph the library has already built (build_pair)
            self.ctx, h = _handle
            self._lib = self.ctx._lib
        elif matrix is not None:  # (rowptr, colidx, values) of a general CSR matrix, canonical format
            self.ctx = ctx if ctx is not None else default_context()
            self._lib = self.ctx._lib
            rp = np.ascontiguousarray(matrix[0], dtype=np.int32)
            ci = np.ascontiguousarray(matrix[1], dtype=np.int32)
            va = _c_f64(matrix[2])
            _check(self._lib.pf_graph_from_matrix(self.ctx._h, len(rp) - 1, rp.ctypes.data_as(_i32p),
                                                  ci.ctypes.data_as(_i32p), _f64(va), C.byref(h)))
        elif device_mesh is not None:
            self.ctx = device_mesh.ctx
            self._lib = self.ctx._lib
            _check(self._lib.pf_graph_build_device(device_mesh._h, C.byref(h)))
        else:
            self.ctx = ctx if ctx is not None else default_context()
            self._lib = self.ctx._lib
            pts = _c_f64(points).reshape(-1, 3)
            f = np.ascontiguousarray(faces, dtype=np.int32)
            if f.ndim != 2:
                raise ValueError("faces must be (F, verts_per_face)")
            _check(self._lib.pf_graph_build(self.ctx._h, _f64(pts), pts.shape[0], f.ctypes.data_as(_i32p), f.shape[0],
                                            f.shape[1] if f.shape[0] else 3, C.byref(h)))
        self._h = h
        _live_graphs.add(self)
        self.ctx._children.add(self)
        info = GraphInfo()
        _check(self._lib.pf_graph_get_info(h, C.byref(info)))
        self.info = info
        self.n = int(info.n)
        self.nnz_w = int(info.nnz_w)
        self.nnz_l = int(info.nnz_l)
        self.symmetric = bool(info.is_symmetric)
        self.n_isolated = int(info.n_isolated)
        self.n_components = int(info.n_components)
        self.max_degree = int(info.max_degree)
        self.n_oneway = int(info.n_oneway)
        self.spectral_bound = float(info.spectral_bound) if 0.0 < float(info.spectral_bound) <= 2.0 else 2.0
        self.op = PF_OP_SYM if self.symmetric else PF_OP_RW
        self.has_points = matrix is None
        self._rows = []

    @classmethod
    def build_pair(cls, mesh_a, mesh_b):
        """The graphs of two `DeviceMesh`es of one context assembled side by side on two streams
        (`pf_graph_build_device2`): same results as two constructor calls, in about the time of one."""
        if mesh_a.ctx is not mesh_b.ctx:
            raise ValueError("build_pair: the two meshes must belong to one context")
        ha, hb = C.c_void_p(), C.c_void_p()
        _check(mesh_a.ctx._lib.pf_graph_build_device2(mesh_a._h, mesh_b._h, C.byref(ha), C.byref(hb)))
        return cls(_handle=(mesh_a.ctx, ha)), cls(_handle=(mesh_b.ctx, hb))

    def close(self):
        if getattr(self, "_h", None):
            for rows in getattr(self, "_rows", []):
                rows.close()
            self._lib.pf_graph_free(self._h)  # (collects a download that is still owed or in flight)
            self._h = None
            self._final_out = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # ---- matrices back to the host (scipy views for the reference-style attributes) ----------
    def download(self, labels=False):
        n, nnz = self.n, self.nnz_w
        out = dict(rowptr=np.empty(n + 1, np.int32), colidx=np.empty(nnz, np.int32), w=np.empty(nnz),
                   l_offdiag=np.empty(nnz), deg=np.empty(n), l_diag=np.empty(n))
        lab = np.empty(n, np.int32) if labels else None
        _check(self._lib.pf_graph_download(self._h, out["rowptr"].ctypes.data_as(_i32p),
                                           out["colidx"].ctypes.data_as(_i32p), _f64(out["w"]), _f64(out["l_offdiag"]),
                                           _f64(out["deg"]), _f64(out["l_diag"]),
                                           lab.ctypes.data_as(_i32p) if labels else None))
        if labels:
            out["labels"] = lab
        return out

    # ---- ops interface ------------------------------------------------------------------------
    def ws_ensure(self, n_slots):
        _check(self._lib.pf_ws_ensure(self._h, int(n_slots)))

    def upload(self, slot, x):
        x = _c_f64(x, (self.n,))
        _check(self._lib.pf_ws_upload(self._h, int(slot), _f64(x)))

    def download_slots(self, first, count):
        out = np.empty((int(count), self.n), dtype=np.float64)
        _check(self._lib.pf_ws_download(self._h, int(first), int(count), _f64(out)))
        return out.T

    def copy(self, src, dst, count):
        _check(self._lib.pf_ws_copy(self._h, int(src), int(dst), int(count)))

    def mask_isolated(self, slot):
        _check(self._lib.pf_mask_isolated(self._h, int(slot)))

    def start_vector(self, slot, seed):
        _check(self._lib.pf_start_vector(self._h, int(slot), int(seed)))

    def lock_null_vectors(self):
        k = C.c_int32()
        _check(self._lib.pf_lock_null_vectors(self._h, self.op, C.byref(k)))
        return int(k.value)

    def spmv(self, src, dst):
        _check(self._lib.pf_spmv(self._h, self.op, int(src), int(dst)))

    def spmv_multi(self, src_first, dst_first, count):
        _check(self._lib.pf_spmv_multi(self._h, self.op, int(src_first), int(dst_first), int(count)))

    def cheb(self, src, dst, degree, c, e, rho=1.0):
        _check(self._lib.pf_cheb(self._h, self.op, int(src), int(dst), int(degree), float(c), float(e), float(rho)))

    def cheb2(self, req_self, other, req_other):
        """One lockstep filter application for two graphs of the same context:
        req = (src, dst, degree, c, e, rho)."""
        a, b = req_self, req_other
        _check(self._lib.pf_cheb2(self._h, self.op, int(a[0]), int(a[1]), int(a[2]), float(a[3]), float(a[4]), float(a[5]),
                                  other._h, other.op, int(b[0]), int(b[1]), int(b[2]), float(b[3]), float(b[4]), float(b[5])))

    def dots(self, w, first, count):
        out = np.empty(int(count), dtype=np.float64)
        _check(self._lib.pf_dots(self._h, int(w), int(first), int(count), _f64(out)))
        return out

    def orth(self, w, first, count):
        h = np.empty(max(int(count), 1), dtype=np.float64)
        nrm = C.c_double()
        _check(self._lib.pf_orth(self._h, int(w), int(first), int(count), _f64(h), C.byref(nrm)))
        return h[: int(count)], float(nrm.value)

    def orth_begin(self, w, first, count, normalize=True):
        _check(self._lib.pf_orth_begin(self._h, int(w), int(first), int(count), int(bool(normalize))))
        self._orth_count = int(count)

    def orth_begin2(self, req, other, req_other):
        """`orth_begin(*req)` of this graph and `other.orth_begin(*req_other)` in shared launches (requests:
        (w, first, count, normalize)); each graph collects with its own `orth_end`."""
        _check(self._lib.pf_orth_begin2(self._h, int(req[0]), int(req[1]), int(req[2]), int(bool(req[3])), other._h, int(req_other[0]),
                                        int(req_other[1]), int(req_other[2]), int(bool(req_other[3]))))
        self._orth_count, other._orth_count = int(req[2]), int(req_other[2])

    def orth_cheb2(self, orth, req, other, orth_other, req_other):
        """`orth_begin2` and, right behind it, `cheb2` in one library call (orth: (w, first, count, normalize); req as for
        `cheb2`)."""
        o = (C.c_int32 * 8)(int(orth[0]), int(orth[1]), int(orth[2]), int(bool(orth[3])), int(orth_other[0]), int(orth_other[1]),
                            int(orth_other[2]), int(bool(orth_other[3])))
        ci = (C.c_int32 * 8)(self.op, int(req[0]), int(req[1]), int(req[2]), other.op, int(req_other[0]), int(req_other[1]),
                             int(req_other[2]))
        cd = (C.c_double * 6)(float(req[3]), float(req[4]), float(req[5]), float(req_other[3]), float(req_other[4]), float(req_other[5]))
        _check(self._lib.pf_orth_cheb2(self._h, other._h, o, ci, cd))
        self._orth_count, other._orth_count = int(orth[2]), int(orth_other[2])

    def orth_split(self, first2, split):
        """The next `orth_begin` / `orth_begin2` / `orth_cheb2` step of this graph takes its basis from the slots
        [first, first + split) and [first2, first2 + count - split) (`pf_orth_split`)."""
        _check(self._lib.pf_orth_split(self._h, int(first2), int(split)))

    def orth_strict(self, on):
        """Second Gram-Schmidt pass at the classical threshold (|w'| < 0.71 |w|) instead of the loose one (0.3); on = 2:
        every step takes its second pass."""
        _check(self._lib.pf_orth_strict(self._h, 2 if on == 2 and on is not True else int(bool(on))))

    def orth_device_passes(self, on):
        """The second Gram-Schmidt pass queued with the first, run on the device's own verdict (`pf_orth_device_passes`):
        nothing queued behind a step ever reads a stale vector (`orth_redone` stays False, `orth_twice` tells)."""
        _check(self._lib.pf_orth_device_passes(self._h, int(bool(on))))

    def orth_end(self):
        h = np.empty(max(self._orth_count, 1), dtype=np.float64)
        nrm = C.c_double()
        _check(self._lib.pf_orth_end(self._h, _f64(h), C.byref(nrm)))
        # the step needed its second Gram-Schmidt pass, run only now: anything queued since orth_begin that read w is stale
        r = self._lib.pf_orth_redone(self._h)
        self.orth_redone = r == 1
        self.orth_twice = r == 2  # (both passes ran on the device, before anything queued behind them)
        return h[: self._orth_count], float(nrm.value)

    def orth_abandon(self):
        """Collect and discard an orthogonalisation left in flight by a solve that was aborted."""
        try:
            self.orth_end()
        except PfError:
            pass

    def scale(self, slot, alpha):
        _check(self._lib.pf_scale(self._h, int(slot), float(alpha)))

    def combine(self, src_first, m, Y, dst_first):
        Y = _c_f64(Y)
        if Y.ndim != 2 or Y.shape[0] != m:
            raise ValueError("Y must be (m, k)")
        _check(self._lib.pf_combine(self._h, int(src_first), int(m), _f64(Y), Y.shape[1], int(dst_first)))

    def resnorm(self, ax, x, lam):
        out = C.c_double()
        _check(self._lib.pf_resnorm(self._h, int(ax), int(x), float(lam), C.byref(out)))
        return float(out.value)

    def gram(self, first_a, count_a, first_b, count_b):
        """(count_a, count_b) matrix of inner products <slot first_a+i, slot first_b+j>, one synchronisation."""
        out = np.empty((int(count_a), int(count_b)), dtype=np.float64)
        _check(self._lib.pf_gram(self._h, int(first_a), int(count_a), int(first_b), int(count_b), _f64(out)))
        return out

    def resnorms(self, ax_first, x_first, lams):
        """||slot(ax_first+i) - lams[i] slot(x_first+i)||_2 for every i, one synchronisation."""
        lams = _c_f64(lams)
        out = np.empty(len(lams), dtype=np.float64)
        _check(self._lib.pf_resnorms(self._h, int(ax_first), int(x_first), _f64(lams), len(lams), _f64(out)))
        return out

    def finalize_vectors(self, first, count, minmax, wait=True):
        """Normalised eigenvectors of `count` slots -> (n, count) array in PINNED host memory (one DMA on the ctx's copy
        stream).  `wait=False`: the array is returned while the download is still in flight - `finalize_wait()` before
        anybody reads it (`Graph.eig_vecs` does); the device-resident twin of the block is usable at once."""
        out = pinned_empty((self.n, int(count)))
        _check(self._lib.pf_finalize_vectors_begin(self._h, int(first), int(count), int(self.op == PF_OP_SYM), int(bool(minmax)),
                                                   _f64(out)))
        self._final_count = int(count)  # the same block stays resident on the device (final_rows, Context.knn1_graphs)
        self._final_pending = True
        # the download may not even be queued yet (the library holds it back until a long kernel runs): the array must
        # outlive it whatever the caller does with its own reference - a pinned block that is freed under an owed
        # download is a write into unmapped memory from the device
        self._final_out = out
        if wait:
            self.finalize_wait()
        return out

    def final_remap(self, cols, signs, out):
        """out[:, c] <- (device-resident block)[:, cols[c]] * signs[c], by a kernel and one DMA on the copy stream (`out`: the
        pinned array `finalize_vectors` returned, all of its columns); `finalize_wait()` before anybody reads it."""
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        signs = np.ascontiguousarray(signs, dtype=np.float64)
        assert out.flags.c_contiguous and out.shape == (self.n, len(cols))
        _check(self._lib.pf_final_remap_begin(self._h, cols.ctypes.data_as(_i32p), _f64(signs), len(cols), _f64(out)))
        self._final_pending = True
        self._final_out = out

    def finalize_wait(self):
        """Collect the download a `finalize_vectors(..., wait=False)` left in flight (no-op otherwise)."""
        if getattr(self, "_final_pending", False) and self._h:
            self._final_pending = False
            try:
                _check(self._lib.pf_finalize_vectors_end(self._h))
            finally:
                self._final_out = None

    def final_rows(self, rows):
        """Rows of the block the last `finalize_vectors` left on the device -> (len(rows), count) array."""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        out = np.empty((len(rows), self._final_count), dtype=np.float64)
        _check(self._lib.pf_final_rows(self._h, rows.ctypes.data_as(_i64p), len(rows), _f64(out)))
        return out

    def final_device(self):
        """(device address, n_rows, n_cols) of the block the last `finalize_vectors` left in HBM (row-major float64)."""
        ptr, n_rows, n_cols = C.c_void_p(), C.c_int64(), C.c_int32()
        _check(self._lib.pf_final_device(self._h, C.byref(ptr), C.byref(n_rows), C.byref(n_cols)))
        return int(ptr.value), int(n_rows.value), int(n_cols.value)

    def point_rows(self, rows):
        """Rows of the mesh's points as they sit on the device -> (len(rows), 3) array (mesh-built graphs only)."""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        out = np.empty((len(rows), 3), dtype=np.float64)
        _check(self._lib.pf_point_rows(self._h, rows.ctypes.data_as(_i64p), len(rows), _f64(out)))
        return out

    def spmv_host(self, x, op=None):
        x = _c_f64(x, (self.n,))
        y = np.empty(self.n, dtype=np.float64)
        _check(self._lib.pf_spmv_host(self._h, self.op if op is None else int(op), _f64(x), _f64(y)))
        return y

    def eigs_smallest(self, n_wanted, minmax=False, wait=True):
        """`pf_eigs_smallest_ex`: the eigensolve as ONE C call -> (vals, vecs (n, m) in pinned memory, stats dict with a
        `residuals` array).  `wait=False`: the eigenvector download is still in flight (`finalize_wait()`)."""
        m = int(n_wanted)
        vals, vecs, resid, n_out, st = np.empty(m), pinned_empty((self.n, m)), np.zeros(m), C.c_int32(), EigsStats()
        _check(self._lib.pf_eigs_smallest_ex(self._h, m, int(bool(minmax)), 0 if wait else 1, _f64(vals), _f64(vecs), _f64(resid),
                                             C.byref(n_out), C.byref(st)))
        return self._eigs_result(vals, vecs, resid, n_out.value, st, wait)

    def _eigs_result(self, vals, vecs, resid, m, st, wait):
        self._final_count = m
        self._final_pending = (not wait) and m > 0
        self._final_out = vecs if self._final_pending else None  # (kept alive until the download has been collected)
        if m != vecs.shape[1]:  # fewer pairs than asked for: the library wrote an (n, m) block
            if self._final_pending:
                self.finalize_wait()
            vecs = np.ascontiguousarray(vecs.reshape(-1)[: self.n * m].reshape(self.n, m))
        stats = {f: getattr(st, f) for f, _ in EigsStats._fields_}
        stats["residuals"] = resid[:m].copy()
        return vals[:m].copy(), vecs, stats

    def eigs_smallest2(self, other, n_wanted, n_wanted_other, minmax=False, wait=True):
        """`pf_eigs_smallest2`: this graph and `other` (same ctx) solved together in ONE C call -
        the pipelined pair driver in C++.  Returns two tuples (vals, vecs (n, m) in pinned memory, stats dict with a
        `residuals` array).  `wait=False`: the eigenvector downloads are still in flight (`finalize_wait()` on each)."""
        outs = []
        for dev, m in ((self, int(n_wanted)), (other, int(n_wanted_other))):
            outs.append((np.empty(m), pinned_empty((dev.n, m)), np.zeros(m), C.c_int32(), EigsStats()))
        (va, xa, ra, na, sa), (vb, xb, rb, nb, sb) = outs
        _check(self._lib.pf_eigs_smallest2(self._h, other._h, int(n_wanted), int(n_wanted_other), int(bool(minmax)), 0 if wait else 1,
                                           _f64(va), _f64(xa), _f64(ra), C.byref(na), C.byref(sa),
                                           _f64(vb), _f64(xb), _f64(rb), C.byref(nb), C.byref(sb)))
        return tuple(dev._eigs_result(vals, vecs, resid, n_out.value, st, wait) for dev, (vals, vecs, resid, n_out, st) in zip((self, other), outs))

    # ---- primitives of the row-partitioned solve (pyfocusr_amd/rowpart.py)
    def op_step(self, x, prev, out, alpha, c, beta, op=None):
        """out = alpha (c x - A x) - beta prev (slots; prev None: no prev term; out may be the prev slot)."""
        _check(self._lib.pf_op_step(self._h, self.op if op is None else int(op), int(x), -1 if prev is None else int(prev),
                                    int(out), float(alpha), float(c), float(beta)))

    def cheb_steps(self, prev, cur, k_first, n_steps, c, e, rho=1.0, op=None):
        """Steps k_first.. of the Chebyshev recurrence on explicit state slots; returns (prev, cur) afterwards."""
        op_, oc = C.c_int32(), C.c_int32()
        _check(self._lib.pf_cheb_steps(self._h, self.op if op is None else int(op), int(prev), int(cur), int(k_first),
                                       int(n_steps), float(c), float(e), float(rho), C.byref(op_), C.byref(oc)))
        return op_.value, oc.value

    def axpy(self, w, first, count, coef):
        coef = _c_f64(coef)
        _check(self._lib.pf_axpy(self._h, int(w), int(first), int(count), _f64(coef)))

    def rows_create(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        h = C.c_void_p()
        _check(self._lib.pf_rows_create(self._h, idx.ctypes.data_as(_i64p), len(idx), C.byref(h)))
        rows = _Rows(self, h, len(idx))
        self._rows.append(rows)
        return rows

    def rows_gather(self, slot, rows):
        out = np.empty(rows.n)
        _check(self._lib.pf_rows_gather(rows._h, int(slot), _f64(out)))
        return out

    def rows_scatter(self, slot, rows, values):
        values = _c_f64(values, (rows.n,))
        _check(self._lib.pf_rows_scatter(rows._h, int(slot), _f64(values)))

    def rows_gather_dev(self, slot, rows, device_ptr):
        """Values of the row subset into a device buffer of the caller (int address); no host sync."""
        _check(self._lib.pf_rows_gather_dev(rows._h, int(slot), C.c_void_p(int(device_ptr))))

    def rows_scatter_dev(self, slot, rows, device_ptr):
        _check(self._lib.pf_rows_scatter_dev(rows._h, int(slot), C.c_void_p(int(device_ptr))))

    def rows_set_sources(self, rows, offsets):
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if len(offsets) != rows.n:
            raise ValueError("one offset per row")
        _check(self._lib.pf_rows_set_sources(rows._h, offsets.ctypes.data_as(_i64p)))

    def rows_gather2_dev(self, slot_a, slot_b, rows, device_ptr, stride):
        _check(self._lib.pf_rows_gather2_dev(rows._h, int(slot_a), -1 if slot_b is None else int(slot_b),
                                             C.c_void_p(int(device_ptr)), int(stride)))

    def rows_scatter2_dev(self, slot_a, slot_b, rows, device_ptr, stride):
        _check(self._lib.pf_rows_scatter2_dev(rows._h, int(slot_a), -1 if slot_b is None else int(slot_b),
                                              C.c_void_p(int(device_ptr)), int(stride)))

    def sync(self):
        self.ctx.sync()

    def rows_fill(self, slot, rows, value):
        _check(self._lib.pf_rows_fill(rows._h, int(slot), float(value)))

    def mean_filter(self, values, iterations):
        v = _c_f64(values)
        one_d = v.ndim == 1
        v2 = v.reshape(self.n, -1)
        out = np.empty_like(v2)
        _check(self._lib.pf_mean_filter(self._h, _f64(v2), v2.shape[1], int(iterations), _f64(out)))
        return out[:, 0] if one_d else out
