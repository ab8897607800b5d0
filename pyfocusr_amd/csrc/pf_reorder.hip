// Internal vertex renumbering for the eigensolver kernels.
//
// The reference's matrices keep the mesh's vertex order (and so does everything that crosses the
// C-ABI: CSR(W), deg, eigenvectors).  Inside the solver the order is free, and it decides how
// the SpMV behaves: x is gathered through 6-7 neighbour indices per row, and on a mesh whose
// index order carries no spatial locality every gather touches its own cache line (measured:
// 2.2 TB/s algorithmic).  So the operator is stored in a renumbered space:
//   1. Morton (Z-order) code of each vertex position, 10 bits per axis -> radix sort: vertices
//      that are close on the surface become close in index, a wave's 64 rows gather from a
//      handful of lines;
//   2. inside windows of g->win_rows (1024, 2048 or 4096) consecutive rows, a stable sort by
//      (boundary rows first, then the rows one hop behind them, then descending degree): the 64 rows of a SELL slice then share one width and the
//      padding disappears; and the rows of a window that other windows touch - or that touch other
//      windows - are its FIRST rows, so that the resident Chebyshev kernel (pf_persist.hip), whose
//      blocks own one window each, can compute and publish exactly those rows first and do the
//      interior rows while the published values travel.
// perm[new] = old (-1 on padding rows), iperm[old] = new.  The radix sorts are hipCUB's.
#include <hipcub/hipcub.hpp>

#include "pf_internal.h"
#include "pf_launch.h"

namespace {

inline unsigned nblk(int64_t n) { return (unsigned)((n + PF_BLOCK - 1) / PF_BLOCK); }

__device__ __forceinline__ unsigned long long enc_f64(double d) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | (1ull << 63));
}
__device__ __forceinline__ double dec_f64(unsigned long long u) {
    return __longlong_as_double((long long)((u >> 63) ? (u & ~(1ull << 63)) : ~u));
}

struct k_bbox {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const double* __restrict__ pts, int64_t n, unsigned long long* bbox) {
    unsigned long long lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0ull, 0ull, 0ull};
    for (int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * PF_BLOCK) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const unsigned long long e = enc_f64(pts[3 * i + a]);
            lo[a] = e < lo[a] ? e : lo[a];
            hi[a] = e > hi[a] ? e : hi[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int off = PF_WAVE / 2; off > 0; off >>= 1) {
            const unsigned long long l2 = __shfl_xor(lo[a], off, PF_WAVE), h2 = __shfl_xor(hi[a], off, PF_WAVE);
            lo[a] = l2 < lo[a] ? l2 : lo[a];
            hi[a] = h2 > hi[a] ? h2 : hi[a];
        }
    }
    // block-level merge first: 64-bit atomics on six neighbouring addresses serialise, one set per block is enough
    __shared__ unsigned long long part[PF_BLOCK / PF_WAVE][6];
    const int wave = threadIdx.x / PF_WAVE;
    if ((threadIdx.x & (PF_WAVE - 1)) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            part[wave][a] = lo[a];
            part[wave][3 + a] = hi[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        unsigned long long v = part[0][threadIdx.x];
        for (int w = 1; w < PF_BLOCK / PF_WAVE; ++w) {
            const unsigned long long o = part[w][threadIdx.x];
            v = threadIdx.x < 3 ? (o < v ? o : v) : (o > v ? o : v);
        }
        if (threadIdx.x < 3)
            atomicMax(&bbox[threadIdx.x], ~v);  // (minima are kept inverted: the box starts as all zeroes)
        else
            atomicMax(&bbox[threadIdx.x], v);
    }
}
};

__device__ __forceinline__ unsigned spread10(unsigned v) {  // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

struct k_morton_keys {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const double* __restrict__ pts, int64_t n,
                                                          const unsigned long long* __restrict__ bbox,
                                                          unsigned* __restrict__ keys, int32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= n) return;
    unsigned code = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double lo = dec_f64(~bbox[a]), hi = dec_f64(bbox[3 + a]);
        const double ext = hi - lo;
        double t = ext > 0.0 ? (pts[3 * i + a] - lo) / ext : 0.0;
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        unsigned q = (unsigned)(t * 1023.0);
        code |= spread10(q) << a;
    }
    keys[i] = code;
    vals[i] = (int32_t)i;
}
};

struct k_scatter_pos {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ order, int64_t n, int32_t* __restrict__ pos) {
    const int64_t r = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (r < n) pos[order[r]] = (int32_t)r;
}
};

// flag[r] = 1 for the Morton positions whose row has an entry in another window, and for the positions such an
// entry points at (W may be asymmetric: a row can be read from outside without reading outside itself)
struct k_boundary_flags {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ order, const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col, const int32_t* __restrict__ pos,
                                                             int64_t n, int32_t win_rows, unsigned* __restrict__ flag) {
    const int64_t r = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (r >= n) return;
    const int32_t old = order ? order[r] : (int32_t)r;  // (order == pos == nullptr: the rows stand in their base order already)
    const int32_t w = (int32_t)(r / win_rows);
    bool mine = false;
    for (int32_t a = rowptr[old]; a < rowptr[old + 1]; ++a) {
        const int32_t rj = pos ? pos[col[a]] : col[a];
        if (rj / win_rows != w) {
            mine = true;
            flag[rj] = 1u;  // (plain stores of the same value: a benign race)
        }
    }
    if (mine) flag[r] = 1u;
}
};

// flag 2 for the positions one hop behind the boundary rows (rows a boundary row reads, or that read one): with two
// recurrence steps per exchange (k_cheb_resident2) another window needs those as well, and a window publishes its leading
// rows.  Only a sort heuristic: what is really published is derived from the windows' lists (pf_windows.hip).
struct k_second_ring_flags {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ order, const int32_t* __restrict__ rowptr,
                                                                const int32_t* __restrict__ col, const int32_t* __restrict__ pos,
                                                                int64_t n, unsigned* __restrict__ flag) {
    const int64_t r = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (r >= n) return;
    const int32_t old = order ? order[r] : (int32_t)r;
    const bool boundary = flag[r] == 1u;
    bool near = false;
    for (int32_t a = rowptr[old]; a < rowptr[old + 1]; ++a) {
        const int32_t rj = pos ? pos[col[a]] : col[a];
        const unsigned f = flag[rj];  // (0 may turn into 2 meanwhile; 1 never changes in this kernel)
        if (boundary && f == 0u) flag[rj] = 2u;
        near |= f == 1u;
    }
    if (!boundary && near) flag[r] = 2u;
}
};

// second key: window of win_rows Morton-consecutive rows, then boundary rows first, then the rows next to them, then
// descending degree
struct k_degree_keys {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ order, const int32_t* __restrict__ rowptr,
                                                          const unsigned* __restrict__ flag, int64_t n, int32_t win_rows,
                                                          unsigned* __restrict__ keys) {
    const int64_t r = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (r >= n) return;
    const int32_t old = order ? order[r] : (int32_t)r;
    int32_t d = rowptr[old + 1] - rowptr[old];
    d = d > 1023 ? 1023 : d;
    const unsigned f = flag[r];
    keys[r] = ((unsigned)(r / win_rows) << 12) | ((f == 1u ? 0u : (f == 2u ? 1u : 2u)) << 10) | (unsigned)(1023 - d);
}
};

// The second sort is local: the key's high bits are the window, and the rows already stand in window order (Morton
// position).  One block per window sorts (boundary flag | degree, position in the window) in LDS - the position makes the
// keys unique, so the bitonic network gives exactly the stable sort by (flag, degree) a radix sort would (17 merge-sort
// launches, ~107 us at 250k rows; this: one launch, ~10 us).
struct k_sort_windows {
    static constexpr int BOUNDS = 1024;
    static __device__ __forceinline__ void run(const unsigned* __restrict__ keys, const int32_t* __restrict__ vals, int64_t n,
                                                       int32_t win_rows, int32_t n_pow2, int32_t* __restrict__ out) {
    extern __shared__ unsigned long long wbuf[];
    const int64_t r0 = (int64_t)blockIdx.x * win_rows;
    for (int i = threadIdx.x; i < n_pow2; i += 1024) {
        const int64_t r = r0 + i;
        wbuf[i] = (i < win_rows && r < n) ? (((unsigned long long)(keys[r] & 4095u) << 32) | (unsigned)i) : ~0ull;
    }
    __syncthreads();
    for (int size = 2; size <= n_pow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < n_pow2 / 2; t += 1024) {
                const int pos = 2 * t - (t & (stride - 1));
                const unsigned long long a = wbuf[pos], b = wbuf[pos + stride];
                const bool up = (pos & size) == 0;
                if ((a > b) == up) {
                    wbuf[pos] = b;
                    wbuf[pos + stride] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < win_rows; i += 1024) {
        const int64_t r = r0 + i;
        if (r < n) out[r] = vals ? vals[r0 + (int64_t)(wbuf[i] & 0xffffffffull)] : (int32_t)(r0 + (int64_t)(wbuf[i] & 0xffffffffull));
    }
}
};

struct k_iota {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(int32_t* __restrict__ v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i < n) v[i] = (int32_t)i;
}
};

struct k_finish_perm {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(int32_t* __restrict__ perm, int32_t* __restrict__ iperm, int64_t n,
                                                          int64_t n_pad) {
    const int64_t r = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (r >= n_pad) return;
    if (r < n) iperm[perm[r]] = (int32_t)r;
    else perm[r] = -1;
}
};

// perm[r] = morder[perm_m[r]] (-1 on padding rows), iperm[original] = r: what the C-ABI side of the library maps with
struct k_compose_perm {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ perm_m, const int32_t* __restrict__ morder,
                                                           int64_t n_pad, int32_t* __restrict__ perm, int32_t* __restrict__ iperm) {
    const int64_t r = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (r >= n_pad) return;
    const int32_t m = perm_m[r];
    const int32_t o = m >= 0 ? morder[m] : -1;
    perm[r] = o;
    if (o >= 0) iperm[o] = (int32_t)r;
}
};

// Krylov start vector, part 1: a low-order polynomial of the vertex position (coordinates mapped to
// [-1, 1] by the bounding box).  It is rich in the low eigenmodes the solver wants (a random vector
// carries ~n^-1/2 of each), which saves about one outer step in twenty.  Stored in solver order.
struct k_smooth_start {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const double* __restrict__ pts, const int32_t* __restrict__ perm,
                                                           const unsigned long long* __restrict__ bbox, int64_t n_pad,
                                                           double* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (r >= n_pad) return;
    const int32_t i = perm[r];
    double v = 0.0;
    if (i >= 0) {
        double c[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double lo = dec_f64(~bbox[a]), hi = dec_f64(bbox[3 + a]);
            const double h = 0.5 * (hi - lo);
            c[a] = h > 0.0 ? (pts[3 * (int64_t)i + a] - 0.5 * (hi + lo)) / h : 0.0;
        }
        const double x = c[0], y = c[1], z = c[2];  // fixed, unequal coefficients: no accidental cancellation
        v = 1.0 * x + 0.9 * y + 1.1 * z + 0.8 * x * y + 1.2 * y * z + 0.7 * x * z + 1.3 * (x * x - y * y) + 0.6 * z * z;
    }
    out[r] = v;
}
};


// ---- Morton order without a general sort (round 3) ------------------------------------------------------------------
// hipCUB hands arrays below 1M items to rocPRIM's merge sort: 16 launches, ~100 us for a 250k-vertex mesh.  The keys here
// are cell codes of points spread over a bounding box, so a counting sort by the key's leading bits leaves buckets of a few
// dozen vertices at most, and a vertex finds its final place by counting the bucket mates that precede it in (key,
// index) order - the order a stable sort of the keys gives, whatever order the atomics filled the bucket in.  Six
// launches.  A bucket of more than PF_ORDER_BUCKET_MAX vertices (points piled into one cell) raises a flag instead: the
// build learns of it in its one read-back and repeats the ordering with the general sort.
constexpr int32_t PF_ORDER_BUCKET_MAX = 1024;

struct k_order_hist {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const unsigned* __restrict__ keys, int64_t n, int shift, int32_t* __restrict__ hist) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i < n) atomicAdd(&hist[keys[i] >> shift], 1);
}
};

struct k_order_scatter {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const unsigned* __restrict__ keys, int64_t n, int shift,
                                                            const int32_t* __restrict__ start, int32_t* __restrict__ cursor,
                                                            unsigned* __restrict__ bkey, int32_t* __restrict__ bidx) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= n) return;
    const unsigned k = keys[i];
    const int32_t b = (int32_t)(k >> shift);
    const int32_t slot = start[b] + atomicAdd(&cursor[b], 1);
    bkey[slot] = k;
    bidx[slot] = (int32_t)i;
}
};

struct k_order_rank {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const unsigned* __restrict__ bkey, const int32_t* __restrict__ bidx, int64_t n,
                                                         int shift, const int32_t* __restrict__ start, int32_t* __restrict__ order,
                                                         int32_t* __restrict__ overflow) {
    const int64_t s = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (s >= n) return;
    const unsigned k = bkey[s];
    const int32_t me = bidx[s];
    const int32_t b = (int32_t)(k >> shift);
    const int32_t lo = start[b], hi = start[b + 1];
    if (hi - lo > PF_ORDER_BUCKET_MAX) {
        // the build will repeat the ordering - but the kernels queued behind this one still run on what is written here:
        // it has to be a permutation (the bucket as the atomics filled it), never what the buffer happened to hold
        if (s == lo) atomicOr(overflow, 1);
        order[s] = me;
        return;
    }
    int32_t before = 0;
    for (int32_t a = lo; a < hi; ++a) {
        const unsigned ka = bkey[a];
        before += (ka < k || (ka == k && bidx[a] < me)) ? 1 : 0;
    }
    order[lo + before] = me;
}
};

}  // namespace

// The Morton order of a mesh's points, computed BEFORE the mesh is assembled (round 4): g->morder[m] = original vertex,
// g->mrank[original] = m, g->order_bbox.  The assembler renumbers points and faces with it and builds everything in that
// space; pf_compute_order then only refines the order inside windows.
int pf_morton_order(pf_graph* g, const double* d_pts, int32_t* d_overflow) {
    hipStream_t st = g->build_stream ? g->build_stream : g->ctx->stream;
    const int64_t n = g->n;
    const int in = (int)n;
    unsigned *k0 = nullptr, *k1 = nullptr;
    int32_t* v0 = nullptr;
    void* tmp = nullptr;
    int32_t *hist = nullptr, *bstart = nullptr;
    int rc = PF_OK;
    auto fail = [&](hipError_t e) {
        if (e != hipSuccess && rc == PF_OK) {
            pf_set_error("pf_morton_order: %s", hipGetErrorString(e));
            rc = PF_E_HIP;
        }
        return e != hipSuccess;
    };
    do {
        if (fail(pf_malloc(st, (void**)&g->order_bbox, 6 * sizeof(unsigned long long)))) break;
        if (fail(pf_malloc(st, (void**)&g->morder, sizeof(int32_t) * std::max<int64_t>(n, 1)))) break;
        if (fail(pf_malloc(st, (void**)&g->mrank, sizeof(int32_t) * std::max<int64_t>(n, 1)))) break;
        if (fail(pf_malloc(st, (void**)&k0, sizeof(unsigned) * std::max<int64_t>(n, 1))) ||
            fail(pf_malloc(st, (void**)&k1, sizeof(unsigned) * std::max<int64_t>(n, 1))) ||
            fail(pf_malloc(st, (void**)&v0, sizeof(int32_t) * std::max<int64_t>(n, 1))))
            break;
        if (fail(pfl::memset_words(st, g->order_bbox, 0, 6 * sizeof(unsigned long long)))) break;
        if (n == 0) break;
        pfl::launch<k_bbox>(dim3(256), dim3(PF_BLOCK), 0, st, d_pts, n, g->order_bbox);
        pfl::launch<k_morton_keys>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, d_pts, n, g->order_bbox, k0, v0);
        if (fail(hipGetLastError())) break;
        const bool counting = d_overflow != nullptr && n >= 4096;  // (else: the general sort)
        if (counting) {
            int bits = 12;
            while (bits < 22 && ((int64_t)1 << bits) < 2 * n) ++bits;  // ~2 buckets per vertex, 4 M at most
            const int shift = 30 - bits;
            const int64_t nb = (int64_t)1 << bits;
            if (fail(pf_malloc(st, (void**)&hist, sizeof(int32_t) * (size_t)(2 * nb + 2)))) break;  // [nb + 1] counts, then [nb] cursors
            if (fail(pf_malloc(st, (void**)&bstart, sizeof(int32_t) * (size_t)(nb + 1)))) break;
            if (fail(pfl::memset_words(st, hist, 0, sizeof(int32_t) * (size_t)(2 * nb + 2)))) break;
            pfl::launch<k_order_hist>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, k0, n, shift, hist);
            if (fail(hipGetLastError())) break;
            if (pf_exclusive_scan_i32(st, hist, bstart, nb + 1) != PF_OK) {
                rc = PF_E_HIP;
                break;
            }
            // k1 / v0: the bucketed keys and vertices (v0's identity is not needed any more: the index is the thread's own)
            pfl::launch<k_order_scatter>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, k0, n, shift, bstart, hist + nb + 1, k1, v0);
            pfl::launch<k_order_rank>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, k1, v0, n, shift, bstart, g->morder, d_overflow);
            if (fail(hipGetLastError())) break;
        } else {
            size_t need = 0;
            pfl::flush_self();  // (a library sort launches at once: whatever this thread has recorded goes first)
            if (fail(hipcub::DeviceRadixSort::SortPairs(nullptr, need, k0, k1, v0, g->morder, in, 0, 30, st))) break;
            if (fail(pf_malloc(st, &tmp, need))) break;
            if (fail(hipcub::DeviceRadixSort::SortPairs(tmp, need, k0, k1, v0, g->morder, in, 0, 30, st))) break;
        }
        pfl::launch<k_scatter_pos>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, g->morder, n, g->mrank);
        if (fail(hipGetLastError())) break;
    } while (0);
    pf_free(st, k0);
    pf_free(st, k1);
    pf_free(st, v0);
    pf_free(st, tmp);
    pf_free(st, hist);
    pf_free(st, bstart);
    return rc;
}

// The order inside windows for a graph whose rows stand in Morton order already (g->morder set: a mesh assembled in
// m-space; d_pts: the points in that order): boundary rows first, then by degree; g->perm_m / g->iperm_m (solver <-> m),
// their compositions with morder in g->perm / g->iperm (solver <-> original), g->smooth.
static int order_in_windows(pf_graph* g, const double* d_pts) {
    hipStream_t st = g->build_stream ? g->build_stream : g->ctx->stream;  // blocks are taken and released on this one ...
    hipStream_t ls = g->side_stream ? g->side_stream : st;                // ... the kernels may run on the build's side stream
    const int64_t n = g->n;
    const int32_t win_rows = g->win_rows;
    unsigned *bflag = nullptr, *k0 = nullptr, *k1 = nullptr;
    int32_t* v1 = nullptr;
    void* tmp = nullptr;
    int rc = PF_OK;
    auto fail = [&](hipError_t e) {
        if (e != hipSuccess && rc == PF_OK) {
            pf_set_error("pf_compute_order: %s", hipGetErrorString(e));
            rc = PF_E_HIP;
        }
        return e != hipSuccess;
    };
    do {
        if (fail(pf_malloc(st, (void**)&bflag, sizeof(unsigned) * std::max<int64_t>(n, 1)))) break;
        if (fail(pf_malloc(st, (void**)&k0, sizeof(unsigned) * std::max<int64_t>(n, 1)))) break;
        if (fail(pfl::memset_words(ls, bflag, 0, sizeof(unsigned) * std::max<int64_t>(n, 1)))) break;
        pfl::launch<k_boundary_flags>(dim3(nblk(n)), dim3(PF_BLOCK), 0, ls, nullptr, g->rowptr, g->col, nullptr, n, win_rows, bflag);
        pfl::launch<k_second_ring_flags>(dim3(nblk(n)), dim3(PF_BLOCK), 0, ls, nullptr, g->rowptr, g->col, nullptr, n, bflag);
        pfl::launch<k_degree_keys>(dim3(nblk(n)), dim3(PF_BLOCK), 0, ls, nullptr, g->rowptr, bflag, n, win_rows, k0);
        if (fail(hipGetLastError())) break;
        if (win_rows <= 4096) {
            int32_t n_pow2 = 2;
            while (n_pow2 < win_rows) n_pow2 <<= 1;
            pfl::launch<k_sort_windows>(dim3((unsigned)((n + win_rows - 1) / win_rows)), dim3(1024), sizeof(unsigned long long) * (size_t)n_pow2, ls, k0, nullptr, n, win_rows, n_pow2, g->perm_m);
            if (fail(hipGetLastError())) break;
        } else {
            int bits2 = 12;
            for (int64_t w = (n + win_rows - 1) / win_rows; w > 0; w >>= 1) ++bits2;
            if (fail(pf_malloc(st, (void**)&k1, sizeof(unsigned) * n)) || fail(pf_malloc(st, (void**)&v1, sizeof(int32_t) * n))) break;
            pfl::launch<k_iota>(dim3(nblk(n)), dim3(PF_BLOCK), 0, ls, v1, n);
            size_t need = 0;
            pfl::flush_self();  // (a library sort launches at once: whatever this thread has recorded goes first)
            if (fail(hipcub::DeviceRadixSort::SortPairs(nullptr, need, k0, k1, v1, g->perm_m, (int)n, 0, bits2, ls))) break;
            if (fail(pf_malloc(st, &tmp, need))) break;
            if (fail(hipcub::DeviceRadixSort::SortPairs(tmp, need, k0, k1, v1, g->perm_m, (int)n, 0, bits2, ls))) break;
        }
        pfl::launch<k_finish_perm>(dim3(nblk(g->n_pad)), dim3(PF_BLOCK), 0, ls, g->perm_m, g->iperm_m, n, g->n_pad);
        pfl::launch<k_compose_perm>(dim3(nblk(g->n_pad)), dim3(PF_BLOCK), 0, ls, g->perm_m, g->morder, g->n_pad, g->perm, g->iperm);
        pfl::launch<k_smooth_start>(dim3(nblk(g->n_pad)), dim3(PF_BLOCK), 0, ls, d_pts, g->perm_m, g->order_bbox, g->n_pad, g->smooth);
        if (fail(hipGetLastError())) break;
    } while (0);
    pf_free(st, bflag);
    pf_free(st, k0);
    pf_free(st, k1);
    pf_free(st, v1);
    pf_free(st, tmp);
    return rc;
}

// Fills g->perm [n_pad], g->iperm [n] and g->smooth [n_pad] (all already allocated) - and g->perm_m / g->iperm_m for a
// mesh assembled in m-space.
int pf_compute_order(pf_graph* g, const double* d_pts, int32_t* d_overflow) {
    if (g->morder) return order_in_windows(g, d_pts);
    hipStream_t st = g->build_stream ? g->build_stream : g->ctx->stream;
    const int64_t n = g->n;
    const int in = (int)n;
    unsigned long long* bbox = nullptr;
    unsigned *k0 = nullptr, *k1 = nullptr;
    int32_t *v0 = nullptr, *v1 = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0, need = 0;
    int rc = PF_OK;
    auto fail = [&](hipError_t e) {
        if (e != hipSuccess && rc == PF_OK) {
            pf_set_error("pf_compute_order: %s", hipGetErrorString(e));
            rc = PF_E_HIP;
        }
        return e != hipSuccess;
    };
    do {
        // the box (minima inverted, so that it starts as zeroes) and the boundary flags: one block, one fill
        if (fail(pf_malloc(st, (void**)&bbox, 6 * sizeof(unsigned long long) + sizeof(unsigned) * (size_t)n))) break;
        unsigned* bflag = reinterpret_cast<unsigned*>(bbox + 6);
        if (fail(pf_malloc(st, (void**)&k0, sizeof(unsigned) * n)) || fail(pf_malloc(st, (void**)&k1, sizeof(unsigned) * n))) break;
        if (fail(pf_malloc(st, (void**)&v0, sizeof(int32_t) * n)) || fail(pf_malloc(st, (void**)&v1, sizeof(int32_t) * n))) break;
        if (fail(pfl::memset_words(st, bbox, 0, 6 * sizeof(unsigned long long) + sizeof(unsigned) * (size_t)n))) break;
        if (d_pts) {
            pfl::launch<k_bbox>(dim3(256), dim3(PF_BLOCK), 0, st, d_pts, n, bbox);
            pfl::launch<k_morton_keys>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, d_pts, n, bbox, k0, v0);
        } else {  // no geometry (graph handed in as a matrix): keep the caller's order, only sort degrees in windows
            if (fail(pfl::memset_words(st, k0, 0, sizeof(unsigned) * n))) break;
            pfl::launch<k_iota>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, v0, n);
        }
        if (fail(hipGetLastError())) break;
        const bool counting = d_pts != nullptr && d_overflow != nullptr && n >= 4096;  // (else: the general sort)
        pfl::flush_self();  // (a library sort launches at once: whatever this thread has recorded goes first)
        if (fail(hipcub::DeviceRadixSort::SortPairs(nullptr, need, k0, k1, v0, v1, in, 0, 30, st))) break;
        tmp_bytes = need;
        const int32_t win_rows = g->win_rows;
        int bits2 = 12;
        for (int64_t w = (n + win_rows - 1) / win_rows; w > 0; w >>= 1) ++bits2;
        pfl::flush_self();  // (a library sort launches at once: whatever this thread has recorded goes first)
        if (fail(hipcub::DeviceRadixSort::SortPairs(nullptr, need, k0, k1, v0, v1, in, 0, bits2, st))) break;
        tmp_bytes = need > tmp_bytes ? need : tmp_bytes;
        if (fail(pf_malloc(st, &tmp, tmp_bytes))) break;
        need = tmp_bytes;
        if (counting) {
            int bits = 12;
            while (bits < 22 && ((int64_t)1 << bits) < 2 * n) ++bits;  // ~2 buckets per vertex, 4 M at most
            const int shift = 30 - bits;
            const int64_t nb = (int64_t)1 << bits;
            int32_t* hist = nullptr;  // [nb + 1] counts, then [nb] cursors: one block, one fill
            int32_t* bstart = nullptr;
            if (fail(pf_malloc(st, (void**)&hist, sizeof(int32_t) * (size_t)(2 * nb + 2)))) break;
            if (fail(pf_malloc(st, (void**)&bstart, sizeof(int32_t) * (size_t)(nb + 1)))) {
                pf_free(st, hist);
                break;
            }
            bool bad = fail(pfl::memset_words(st, hist, 0, sizeof(int32_t) * (size_t)(2 * nb + 2)));
            if (!bad) {
                pfl::launch<k_order_hist>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, k0, n, shift, hist);
                bad = fail(hipGetLastError()) || pf_exclusive_scan_i32(st, hist, bstart, nb + 1) != PF_OK;
            }
            if (!bad) {
                // k1 / v0: the bucketed keys and vertices (v0's identity is not needed any more: the index is the thread's own)
                pfl::launch<k_order_scatter>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, k0, n, shift, bstart, hist + nb + 1, k1, v0);
                pfl::launch<k_order_rank>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, k1, v0, n, shift, bstart, v1, d_overflow);
                bad = fail(hipGetLastError());
            }
            pf_free(st, hist);
            pf_free(st, bstart);
            if (bad) {
                if (rc == PF_OK) rc = PF_E_HIP;
                break;
            }
        } else if (fail(hipcub::DeviceRadixSort::SortPairs(tmp, need, k0, k1, v0, v1, in, 0, 30, st))) {  // v1 = Morton order
            break;
        }
        // v0 <- Morton position of every vertex (scratch until the second sort); the boundary flags in their zeroed block
        pfl::launch<k_scatter_pos>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, v1, n, v0);
        pfl::launch<k_boundary_flags>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, v1, g->rowptr, g->col, v0, n, win_rows, bflag);
        pfl::launch<k_second_ring_flags>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, v1, g->rowptr, g->col, v0, n, bflag);
        pfl::launch<k_degree_keys>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, v1, g->rowptr, bflag, n, win_rows, k0);
        if (fail(hipGetLastError())) break;
        if (win_rows <= 4096) {
            int32_t n_pow2 = 2;
            while (n_pow2 < win_rows) n_pow2 <<= 1;
            pfl::launch<k_sort_windows>(dim3((unsigned)((n + win_rows - 1) / win_rows)), dim3(1024), sizeof(unsigned long long) * (size_t)n_pow2, st, k0, v1, n, win_rows, n_pow2, g->perm);
            if (fail(hipGetLastError())) break;
        } else {
            need = tmp_bytes;
            if (fail(hipcub::DeviceRadixSort::SortPairs(tmp, need, k0, k1, v1, g->perm, in, 0, bits2, st))) break;
        }
        pfl::launch<k_finish_perm>(dim3(nblk(g->n_pad)), dim3(PF_BLOCK), 0, st, g->perm, g->iperm, n, g->n_pad);
        if (d_pts) pfl::launch<k_smooth_start>(dim3(nblk(g->n_pad)), dim3(PF_BLOCK), 0, st, d_pts, g->perm, bbox, g->n_pad, g->smooth);
        else if (fail(pfl::memset_words(st, g->smooth, 0, sizeof(double) * g->n_pad))) break;  // start vector = noise only
        if (fail(hipGetLastError())) break;
    } while (0);
    pf_free(st, bbox);
    pf_free(st, k0);
    pf_free(st, k1);
    pf_free(st, v0);
    pf_free(st, v1);
    pf_free(st, tmp);
    return rc;
}
