// Exact 1-nearest-neighbour search in d <= 8 dimensions, float64.
//
// Replaces `scipy.spatial.KDTree(target).query(source)` at
// /root/reference/pyfocusr/focusr.py:351-353 (spectral coordinates, d = n_spectral_features)
// and /root/reference/pyfocusr/eigsort.py:203-204 (normalised xyz, d = 3).
//
// Arithmetic (what makes indices AND distances bit-identical to a numpy brute force): squared
// distance = sum over the d coordinates, left to right, of (q_c - r_c)^2 with separate multiply
// and add (file compiled with -ffp-contract=off); the smallest value wins, the lowest reference
// index on exact ties.  FP64-VALU-bound and not a dense contraction: depth d <= 8, and the
// |x|^2+|y|^2-2xy form MFMA would need loses ~5 digits at neighbour distances ~1e-3.
//
// Search (exact, output identical to exhaustive search): the reference points are binned on a
// uniform 2-D grid over the reference set's two widest axes and radix-sorted by cell (hipCUB),
// row-major, so the cells of one grid row that fall in an x-interval are one contiguous run of
// the sorted array.  Queries are sorted along a Morton curve of the same grid, so the 256
// queries of a block are neighbours in that plane.  Phase 1: each lane scans the cells around
// its own and gets an upper bound best0 on its nearest distance.  Every coordinate term of the
// squared distance is <= the rounded sum (adding non-negative terms is monotone in floating
// point), so only references within sqrt(best0) of the query on BOTH grid axes can win or tie:
// the block takes the bounding rectangle of those squares in cell coordinates and, row by row,
// streams the references inside it through LDS (coalesced loads, broadcast reads) for every lane
// to scan exhaustively; ties go to the lowest original index.  Spectral embeddings of surfaces
// are 2-manifolds, so the rectangle holds ~1 % of the references (~0.1 % once the clouds are
// registered); for unrelated clouds it grows to the whole grid and the kernel degenerates into
// the tiled brute force.
#include <hipcub/hipcub.hpp>

#include <string.h>

#include <algorithm>

#include "pf_internal.h"

namespace {

constexpr int KNN_LDS_DOUBLES = 4096;  // LDS tile budget: 32 KiB of coordinates -> 512 points at d = 8, 256 at d = 16

inline unsigned nblk(int64_t n) { return (unsigned)((n + PF_BLOCK - 1) / PF_BLOCK); }

struct KnnGrid {
    int a0, a1;      // grid axes (a1 == a0 when d == 1)
    int r0, r1;      // cells per axis
    double lo0, lo1; // grid origin
    double s0, s1;   // cells per unit length (0 when the extent is 0)
};

__device__ __forceinline__ unsigned long long enc_f64(double d) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | (1ull << 63));
}
__device__ __forceinline__ double dec_f64(unsigned long long u) {
    return __longlong_as_double((long long)((u >> 63) ? (u & ~(1ull << 63)) : ~u));
}

// monotone non-decreasing in x (same expression for references, queries and interval ends)
__device__ __forceinline__ int cell_of(double x, double lo, double scale, int r) {
    const double t = (x - lo) * scale;
    return t > 0.0 ? (t >= (double)r ? r - 1 : (int)t) : 0;
}

// per-axis [min, max] of the reference set (order-preserving integer encoding + 64-bit atomics)
__global__ __launch_bounds__(PF_BLOCK) void k_extent(const double* __restrict__ pts, int64_t n, int d, unsigned long long* ext /* [2][16] */) {
    __shared__ unsigned long long part[PF_BLOCK / PF_WAVE][32];
    const int wave = threadIdx.x / PF_WAVE;
    for (int a = 0; a < d; ++a) {
        unsigned long long lo = ~0ull, hi = 0ull;
        for (int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * PF_BLOCK) {
            const unsigned long long e = enc_f64(pts[i * d + a]);
            lo = e < lo ? e : lo;
            hi = e > hi ? e : hi;
        }
        for (int off = PF_WAVE / 2; off > 0; off >>= 1) {
            const unsigned long long l2 = __shfl_xor(lo, off, PF_WAVE), h2 = __shfl_xor(hi, off, PF_WAVE);
            lo = l2 < lo ? l2 : lo;
            hi = h2 > hi ? h2 : hi;
        }
        if ((threadIdx.x & (PF_WAVE - 1)) == 0) {
            part[wave][a] = lo;
            part[wave][16 + a] = hi;
        }
    }
    __syncthreads();  // block-level merge first: the 64-bit atomics of a whole grid on neighbouring addresses serialise
    if ((int)threadIdx.x < 2 * d) {
        const bool is_hi = (int)threadIdx.x >= d;
        const int a = is_hi ? (int)threadIdx.x - d + 16 : (int)threadIdx.x;
        unsigned long long v = part[0][a];
        for (int w = 1; w < PF_BLOCK / PF_WAVE; ++w) {
            const unsigned long long o = part[w][a];
            v = is_hi ? (o > v ? o : v) : (o < v ? o : v);
        }
        if (is_hi)
            atomicMax(&ext[a], v);
        else
            atomicMin(&ext[a], v);
    }
}

__global__ void k_make_grid(const unsigned long long* __restrict__ ext, int d, int res, KnnGrid* g) {
    if (threadIdx.x | blockIdx.x) return;
    int b0 = 0, b1 = 0;
    double w0 = -1.0, w1 = -1.0;
    for (int a = 0; a < d; ++a) {
        const double e = dec_f64(ext[16 + a]) - dec_f64(ext[a]);
        if (e > w0) {
            w1 = w0;
            b1 = b0;
            w0 = e;
            b0 = a;
        } else if (e > w1) {
            w1 = e;
            b1 = a;
        }
    }
    if (d == 1) {
        b1 = b0;
        w1 = 0.0;
    }
    g->a0 = b0;
    g->a1 = b1;
    g->r0 = res;
    g->r1 = d == 1 ? 1 : res;
    g->lo0 = dec_f64(ext[b0]);
    g->lo1 = dec_f64(ext[b1]);
    g->s0 = (w0 > 0.0 && isfinite(w0)) ? (double)res / w0 : 0.0;
    g->s1 = (d > 1 && w1 > 0.0 && isfinite(w1)) ? (double)res / w1 : 0.0;
}

__device__ __forceinline__ unsigned spread16(unsigned v) {
    v &= 0xffffu;
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

// morton = 0: row-major cell id (references); 1: Morton code of the cell (queries)
__global__ __launch_bounds__(PF_BLOCK) void k_cell_keys(const double* __restrict__ pts, int64_t n, int d,
                                                        const KnnGrid* __restrict__ gp, int morton, unsigned* __restrict__ keys,
                                                        int32_t* __restrict__ vals, int32_t* __restrict__ hist) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= n) return;
    const KnnGrid g = *gp;
    const int cx = cell_of(pts[i * d + g.a0], g.lo0, g.s0, g.r0);
    const int cy = cell_of(pts[i * d + g.a1], g.lo1, g.s1, g.r1);
    const unsigned key = morton ? (spread16((unsigned)cx) | (spread16((unsigned)cy) << 1)) : (unsigned)(cy * g.r0 + cx);
    keys[i] = key;
    if (hist) atomicAdd(&hist[key], 1);  // the counting sort's histogram, taken while the key is at hand
    else vals[i] = (int32_t)i;           // the radix sort's payload
}

__global__ __launch_bounds__(PF_BLOCK) void k_gather_rows(const double* __restrict__ pts, const int32_t* __restrict__ order,
                                                          int64_t n, int d, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int64_t src = order[i];
    for (int c = 0; c < d; ++c) out[i * d + c] = pts[src * d + c];
}

// cell_start[c] = first sorted reference with cell id >= c   (c in [0, n_cells])
__global__ __launch_bounds__(PF_BLOCK) void k_cell_start(const unsigned* __restrict__ keys, int64_t n, int64_t n_cells,
                                                         int32_t* __restrict__ cell_start) {
    const int64_t c = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (c > n_cells) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)keys[mid] < c) lo = mid + 1; else hi = mid;
    }
    cell_start[c] = (int32_t)lo;
}

template <int D>
__device__ __forceinline__ double dist2(const double (&q)[D], const double* __restrict__ r) {
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const double df = q[c] - r[c];
        const double sq = df * df;
        s = (c == 0) ? sq : s + sq;
    }
    return s;
}

// the same sum split at coordinate P: s = dist2_head, then dist2_tail continues the very same accumulation, so the
// result is bit-identical to dist2.  Adding non-negative terms is monotone in floating point: head <= full sum,
// which lets a wave drop a candidate after P coordinates when the head already exceeds every lane's bound.
template <int D, int P>
__device__ __forceinline__ double dist2_head(const double (&q)[D], const double* __restrict__ r) {
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < P; ++c) {
        const double df = q[c] - r[c];
        const double sq = df * df;
        s = (c == 0) ? sq : s + sq;
    }
    return s;
}
template <int D, int P>
__device__ __forceinline__ double dist2_tail(const double (&q)[D], const double* __restrict__ r, double s) {
#pragma unroll
    for (int c = P; c < D; ++c) {
        const double df = q[c] - r[c];
        s = s + df * df;
    }
    return s;
}

// sorted insertion of candidate (s, o) into the lane's K best, ordered by (distance, original index)
template <int K>
__device__ __forceinline__ void topk_insert(double (&best)[K], int32_t (&bidx)[K], double s, int32_t o) {
    if (!(s < best[K - 1] || (s == best[K - 1] && o < bidx[K - 1]))) return;
    best[K - 1] = s;
    bidx[K - 1] = o;
#pragma unroll
    for (int j = K - 1; j > 0; --j) {
        const bool sw = best[j] < best[j - 1] || (best[j] == best[j - 1] && bidx[j] < bidx[j - 1]);
        const double ts = best[j - 1];
        const int32_t to = bidx[j - 1];
        best[j - 1] = sw ? best[j] : ts;
        bidx[j - 1] = sw ? bidx[j] : to;
        best[j] = sw ? ts : best[j];
        bidx[j] = sw ? to : bidx[j];
    }
}

template <int D, int K, int BS, int LANE_CELLS>
__global__ __launch_bounds__(BS) void k_knn_grid(const double* __restrict__ ref_s /* rows sorted by cell */,
                                                       const int32_t* __restrict__ ref_orig,
                                                       const int32_t* __restrict__ cell_start, int64_t n_ref,
                                                       const double* __restrict__ qry_s, const int32_t* __restrict__ qry_orig,
                                                       int64_t n_qry, const KnnGrid* __restrict__ gp,
                                                       int64_t* __restrict__ idx_out, double* __restrict__ d2_out) {
    constexpr int KNN_TILE = (BS >= 256 ? KNN_LDS_DOUBLES : KNN_LDS_DOUBLES / 4) / (D <= 8 ? 8 : 16);
    __shared__ double tile[KNN_TILE * D];
    __shared__ int32_t tile_idx[KNN_TILE];
    __shared__ int box[BS / PF_WAVE][4];
    const KnnGrid g = *gp;
    const int64_t qi = (int64_t)blockIdx.x * BS + threadIdx.x;
    const bool mine = qi < n_qry;
    const int64_t qq = mine ? qi : n_qry - 1;  // tail lanes replay the last query, result discarded
    double q[D];
#pragma unroll
    for (int c = 0; c < D; ++c) q[c] = qry_s[qq * D + c];
    // (read by their run-time axis from memory: picked out of q[] the compiler makes them indexed reads of a scratch copy)
    const double qx = qry_s[qq * D + g.a0], qy = qry_s[qq * D + g.a1];
    const int cx = cell_of(qx, g.lo0, g.s0, g.r0), cy = cell_of(qy, g.lo1, g.s1, g.r1);

    // ---- phase 1: upper bound from the cells around the query (grow the ring until K candidates are found)
    double bestk[K];
    int32_t bidx[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        bestk[j] = INFINITY;
        bidx[j] = 0x7fffffff;
    }
    int rx0 = 0, rx1 = -1, ry0 = 0, ry1 = -1;  // cells phase 1 has scanned (the last ring)
    for (int ring = 1; ring <= 4 && bestk[K - 1] == INFINITY; ++ring) {
#pragma unroll
        for (int j = 0; j < K; ++j) {  // the larger ring re-visits the smaller one: start over
            bestk[j] = INFINITY;
            bidx[j] = 0x7fffffff;
        }
        const int y0 = cy - ring > 0 ? cy - ring : 0, y1 = cy + ring < g.r1 - 1 ? cy + ring : g.r1 - 1;
        const int x0 = cx - ring > 0 ? cx - ring : 0, x1 = cx + ring < g.r0 - 1 ? cx + ring : g.r0 - 1;
        rx0 = x0, rx1 = x1, ry0 = y0, ry1 = y1;
        for (int y = y0; y <= y1; ++y) {
            const int32_t b = cell_start[y * g.r0 + x0], e = cell_start[y * g.r0 + x1 + 1];
            for (int32_t r = b; r < e; ++r) {
                topk_insert<K>(bestk, bidx, dist2<D>(q, ref_s + (int64_t)r * D), ref_orig[r]);
            }
        }
    }
    // ---- block rectangle: every reference that can beat or tie a lane lies within rad = sqrt(its K-th best) of it on
    // both grid axes.  The rectangle is the union over the block's lanes; it is re-derived while the scan proceeds
    // (bounds only shrink), which matters when the two clouds are poorly aligned and the first bound is loose.
    auto block_rectangle = [&](int& bx0, int& bx1, int& by0, int& by1) {
        const double best = bestk[K - 1];
        const double rad = sqrt(best) * (1.0 + 1e-9) + 1e-300;  // inflated against the rounding of sqrt / the subtractions
        double xl = qx - rad, xh = qx + rad, yl = qy - rad, yh = qy + rad;
        xl -= fabs(xl) * 1e-15;
        xh += fabs(xh) * 1e-15;
        yl -= fabs(yl) * 1e-15;
        yh += fabs(yh) * 1e-15;
        bx0 = cell_of(xl, g.lo0, g.s0, g.r0), bx1 = cell_of(xh, g.lo0, g.s0, g.r0);
        by0 = cell_of(yl, g.lo1, g.s1, g.r1), by1 = cell_of(yh, g.lo1, g.s1, g.r1);
        if (!(best < INFINITY)) {  // nothing nearby: the whole grid
            bx0 = 0;
            by0 = 0;
            bx1 = g.r0 - 1;
            by1 = g.r1 - 1;
        }
#pragma unroll
        for (int off = PF_WAVE / 2; off > 0; off >>= 1) {
            bx0 = min(bx0, __shfl_xor(bx0, off, PF_WAVE));
            by0 = min(by0, __shfl_xor(by0, off, PF_WAVE));
            bx1 = max(bx1, __shfl_xor(bx1, off, PF_WAVE));
            by1 = max(by1, __shfl_xor(by1, off, PF_WAVE));
        }
        __syncthreads();  // the previous round's readers are done with box[]
        if ((threadIdx.x & (PF_WAVE - 1)) == 0) {
            int* b = box[threadIdx.x / PF_WAVE];
            b[0] = bx0;
            b[1] = bx1;
            b[2] = by0;
            b[3] = by1;
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < BS / PF_WAVE; ++w) {
            bx0 = min(bx0, box[w][0]);
            bx1 = max(bx1, box[w][1]);
            by0 = min(by0, box[w][2]);
            by1 = max(by1, box[w][3]);
        }
    };
    // ---- phase 2a: the lane's OWN square.  The bounding rectangle of the 256 squares of a block is several times larger
    // than any one of them (250k-vertex pair of the bench: a lane's square holds ~600 references, the block's rectangle
    // ~2000+; clouds that lie close together - registered meshes - have squares inside the ring phase 1 has already
    // scanned).  So while no lane's square exceeds LANE_CELLS cells, each lane scans just its own cells, straight from
    // memory like phase 1 (neighbouring lanes read neighbouring cells: L2 hits; measured 3.7 -> 2.3 ms for the bench's
    // 250k x 250k, d = 5), and the block-wide LDS scan below is kept for blocks in which some lane's square is large
    // (unrelated or poorly aligned clouds, where sharing the loads pays).
    {
        const double best = bestk[K - 1];
        const double rad = sqrt(best) * (1.0 + 1e-9) + 1e-300;
        double xl = qx - rad, xh = qx + rad, yl = qy - rad, yh = qy + rad;
        xl -= fabs(xl) * 1e-15;
        xh += fabs(xh) * 1e-15;
        yl -= fabs(yl) * 1e-15;
        yh += fabs(yh) * 1e-15;
        int lx0 = cell_of(xl, g.lo0, g.s0, g.r0), lx1 = cell_of(xh, g.lo0, g.s0, g.r0);
        int ly0 = cell_of(yl, g.lo1, g.s1, g.r1), ly1 = cell_of(yh, g.lo1, g.s1, g.r1);
        const bool bounded = best < INFINITY;
        const int cells = bounded ? (lx1 - lx0 + 1) * (ly1 - ly0 + 1) : 0x7fffffff;
        if (!__syncthreads_or(cells > LANE_CELLS)) {  // block-uniform
            if (!(lx0 >= rx0 && lx1 <= rx1 && ly0 >= ry0 && ly1 <= ry1)) {  // something outside the scanned ring
                for (int y = ly0; y <= ly1; ++y) {
                    const int32_t b = cell_start[y * g.r0 + lx0], e = cell_start[y * g.r0 + lx1 + 1];
                    for (int32_t r = b; r < e; ++r) {
                        const double s = dist2<D>(q, ref_s + (int64_t)r * D);
                        if (s <= bestk[K - 1]) {
                            const int32_t o = ref_orig[r];
                            bool seen = false;  // phase 1 already holds some of these points
#pragma unroll
                            for (int j = 0; j < K; ++j) seen |= (bidx[j] == o);
                            if (!seen) topk_insert<K>(bestk, bidx, s, o);
                        }
                    }
                }
            }
            if (mine) {
                const int64_t dst = qry_orig[qi];
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    idx_out[dst * K + j] = bidx[j];
                    d2_out[dst * K + j] = bestk[j];
                }
            }
            return;
        }
    }
    int bx0, bx1, by0, by1;
    block_rectangle(bx0, bx1, by0, by1);

    // ---- phase 2: exhaustive scan of the rectangle, one contiguous run of references per grid row; rows are taken
    // from the middle outwards so that the bounds tighten early and the outer rows drop out
    const int mid = (by0 + by1) / 2;
    const int half = max(mid - by0, by1 - mid);
    int since_refresh = 0;
    for (int t = 0; t <= 2 * half; ++t) {
        const int y = (t & 1) ? mid + (t + 1) / 2 : mid - t / 2;
        if (y < by0 || y > by1) continue;  // block-uniform
        const int32_t run_b = cell_start[y * g.r0 + bx0], run_e = cell_start[y * g.r0 + bx1 + 1];
        for (int32_t t0 = run_b; t0 < run_e; t0 += KNN_TILE) {
            const int cnt = (run_e - t0) < KNN_TILE ? (run_e - t0) : KNN_TILE;
            __syncthreads();
            for (int k = threadIdx.x; k < cnt * D; k += BS) tile[k] = ref_s[(int64_t)t0 * D + k];
            for (int k = threadIdx.x; k < cnt; k += BS) tile_idx[k] = ref_orig[t0 + k];
            __syncthreads();
            for (int r = 0; r < cnt; ++r) {
                double s;
                if constexpr (D >= 6) {  // deep embeddings: most of the rectangle is far away in the other coordinates
                    constexpr int P = D / 2;
                    s = dist2_head<D, P>(q, tile + r * D);
                    if (!__any(s <= bestk[K - 1])) continue;  // wave-uniform: no lane can take this candidate
                    s = dist2_tail<D, P>(q, tile + r * D, s);
                } else {
                    s = dist2<D>(q, tile + r * D);
                }
                if (s <= bestk[K - 1]) {
                    const int32_t o = tile_idx[r];
                    bool seen = false;  // phase 1 already holds some of the window's points
#pragma unroll
                    for (int j = 0; j < K; ++j) seen |= (bidx[j] == o);
                    if (!seen) topk_insert<K>(bestk, bidx, s, o);
                }
            }
        }
        since_refresh += run_e - run_b;
        if (since_refresh >= 4 * KNN_TILE && by1 - by0 >= 8) {  // worth a re-derivation (two barriers, ~30 shuffles)
            since_refresh = 0;
            int nx0, nx1, ny0, ny1;
            block_rectangle(nx0, nx1, ny0, ny1);
            bx0 = max(bx0, nx0);
            bx1 = min(bx1, nx1);
            by0 = max(by0, ny0);
            by1 = min(by1, ny1);
        }
    }
    if (mine) {
        const int64_t dst = qry_orig[qi];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            idx_out[dst * K + j] = bidx[j];
            d2_out[dst * K + j] = bestk[j];
        }
    }
}

// ---- 1-NN, a GROUP of queries per wave (the default for k = 1) -----------------------------------------------------
// k_knn_grid gives every lane a query; a wave then runs as long as its lane with the largest square, and the launch as
// long as its slowest wave: on the 250k x 250k bench pair (d = 5) a lane scans 760 candidates on average, the worst lane
// of a wave 2300, and the slowest wave ran 5x longer than the average one - 2.4 ms of which ~0.6 were work.  Here a wave
// owns knn_group(d) Morton-consecutive queries (neighbours in the grid plane: their squares nearly coincide) and its 64
// lanes scan the candidates of the squares' bounding rectangle, one candidate per lane and step, against all queries
// of the group: the coordinate-major copy of the sorted references makes each load 512 contiguous bytes, a candidate
// is fetched once per group instead of once per query, and the work of a wave is a sum over a rectangle instead of a
// maximum over 64 lanes.  Per query the lanes' (distance, original index) pairs are reduced lexicographically across
// the wave - the minimum of that pair is unique, so the result does not depend on who found it or on how many extra
// candidates were looked at: bit-identical to k_knn_grid and to a brute force.
// Clouds that are far apart make the rectangle the whole grid: the scan is then a brute force whose loads are shared by
// the group only - per 64 queries no more arithmetic than k_knn_grid's shared LDS tiles.  k_knn_grid serves k > 1 and d >= 10.
// queries per wave: their coordinates stay in registers (G x d doubles, wave-uniform)
constexpr int knn_group(int d) { return d <= 4 ? 8 : (d <= 8 ? 4 : 2); }

__device__ __forceinline__ void wave_argmin(double& s, int32_t& o) {
#pragma unroll
    for (int off = PF_WAVE / 2; off > 0; off >>= 1) {
        const double s2 = __shfl_xor(s, off, PF_WAVE);
        const int32_t o2 = __shfl_xor(o, off, PF_WAVE);
        const bool take = s2 < s || (s2 == s && o2 < o);
        s = take ? s2 : s;
        o = take ? o2 : o;
    }
}

// COUNT (pf_knn_count; a separate instantiation, the search itself never pays for it): the candidate-query pairs whose
// distance the search evaluates, summed into *visited - the work figure behind the roofline entry of the 1-NN stage.
template <int D, bool COUNT = false>
__global__ __launch_bounds__(PF_BLOCK) void k_knn_coop(const double* __restrict__ soa, int64_t ld, const int32_t* __restrict__ ref_orig,
                                                       const int32_t* __restrict__ cell_start, const double* __restrict__ qry_s,
                                                       const int32_t* __restrict__ qry_orig, int64_t n_qry,
                                                       const KnnGrid* __restrict__ gp, int64_t* __restrict__ idx_out,
                                                       double* __restrict__ d2_out, unsigned long long* __restrict__ visited = nullptr) {
    constexpr int G = knn_group(D);
    unsigned long long seen = 0;  // (wave-uniform)
    const unsigned long long t_begin = COUNT ? __builtin_amdgcn_s_memrealtime() : 0ull;  // (100 MHz)
    unsigned long long n_chunks = 0, n_scans = 0, t_scan = 0, t_bounds = 0, n_pass = 0;  // (COUNT only)
    const KnnGrid g = *gp;
    const int lane = threadIdx.x & (PF_WAVE - 1);
    const int64_t group = (int64_t)blockIdx.x * (PF_BLOCK / PF_WAVE) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x / PF_WAVE));
    const int64_t q0 = group * G;
    if (q0 >= n_qry) return;  // (wave-uniform)
    const int nq = n_qry - q0 < G ? (int)(n_qry - q0) : G;
    double q[G][D];  // wave-uniform (a short group replays its last query)
    // the two grid coordinates of each query once more, read by their (run-time) axis: picking them out of q[][] the
    // compiler turns into an indexed read of a copy of q in SCRATCH memory - 176 bytes per lane stored by every wave at its
    // start, 656 MB of writes per 250k x 250k search by the counters, and the scratch set-up of every wave launch
    double qx[G], qy[G];
    double best[G];
    int32_t bidx[G];
    int ux0 = 0x7fffffff, ux1 = -1, uy0 = 0x7fffffff, uy1 = -1;  // the cells of the group's queries
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int64_t qi = q0 + (i < nq ? i : nq - 1);
#pragma unroll
        for (int c = 0; c < D; ++c) q[i][c] = qry_s[qi * D + c];
        qx[i] = qry_s[qi * D + g.a0];
        qy[i] = qry_s[qi * D + g.a1];
        const int cx = cell_of(qx[i], g.lo0, g.s0, g.r0), cy = cell_of(qy[i], g.lo1, g.s1, g.r1);
        ux0 = min(ux0, cx), ux1 = max(ux1, cx);
        uy0 = min(uy0, cy), uy1 = max(uy1, cy);
        best[i] = INFINITY;
        bidx[i] = 0x7fffffff;
    }
    // One scan serves every stage.  The references of the cells [x0, x1] of up to 64 grid rows - lane l names its row in
    // `row_of` (-1: none) - are per row one contiguous run of the sorted array: lane l fetches the run of its row (one
    // round trip for all rows), the runs are cut into chunks of 64 candidates, numbered through the rows, and the wave
    // takes one chunk per step (other waves cover the latency of its loads).
    auto scan_rows = [&](int row_of, int x0, int x1) {
        int32_t b_l = 0, e_l = 0;
        if (row_of >= 0) {
            b_l = cell_start[row_of * g.r0 + x0];
            e_l = cell_start[row_of * g.r0 + x1 + 1];
        }
        int32_t inc = (e_l - b_l + PF_WAVE - 1) >> 6;  // chunks of this lane's row; below: inclusive prefix over the lanes
#pragma unroll
        for (int off = 1; off < PF_WAVE; off <<= 1) {
            const int32_t up = __shfl_up(inc, off, PF_WAVE);
            if (lane >= off) inc += up;
        }
        const int32_t total = __builtin_amdgcn_readlane(inc, PF_WAVE - 1);
        if constexpr (COUNT) n_chunks += (unsigned long long)total, n_scans += 1ull;
        for (int32_t j = 0; j < total; ++j) {
            const int row = (int)__popcll(__ballot(inc <= j));  // lanes whose rows end before chunk j
            const int32_t first = row > 0 ? __builtin_amdgcn_readlane(inc, row - 1) : 0;
            const int32_t r = __builtin_amdgcn_readlane(b_l, row) + (j - first) * PF_WAVE + lane;
            if constexpr (COUNT) seen += (unsigned long long)__popcll(__ballot(r < __builtin_amdgcn_readlane(e_l, row)));
            if (r < __builtin_amdgcn_readlane(e_l, row)) {
                double x[D];
#pragma unroll
                for (int c = 0; c < D; ++c) x[c] = soa[(int64_t)c * ld + r];
                const int32_t o = ref_orig[r];
                if constexpr (COUNT) {  // how many candidates survive the coordinates OUTSIDE the grid plane, against the bounds at hand
                    bool pass = false;
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        double s3 = 0.0;
#pragma unroll
                        for (int c = 0; c < D; ++c)
                            if (c != g.a0 && c != g.a1) s3 += (q[i][c] - x[c]) * (q[i][c] - x[c]);
                        pass = pass || s3 <= best[i];
                    }
                    n_pass += (unsigned long long)__popcll(__ballot(pass));
                }
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    double s = 0.0;
                    constexpr int P = D >= 6 ? D / 2 : D;  // deep coordinates: most candidates are out after half of them
#pragma unroll
                    for (int c = 0; c < P; ++c) {  // dist2's operations, in its order
                        const double df = q[i][c] - x[c];
                        const double sq = df * df;
                        s = (c == 0) ? sq : s + sq;
                    }
                    if constexpr (P < D) {
                        // the partial sum is a prefix of the very same accumulation and only grows: if no candidate of
                        // the chunk can still reach the query's bound, the other coordinates are not looked at
                        if (!__any(s <= best[i])) continue;
#pragma unroll
                        for (int c = P; c < D; ++c) {
                            const double df = q[i][c] - x[c];
                            s = s + df * df;
                        }
                    }
                    if (s < best[i] || (s == best[i] && o < bidx[i])) {  // (a candidate met twice changes nothing)
                        best[i] = s;
                        bidx[i] = o;
                    }
                }
            }
        }
    };
    // every lane <- the wave's best per query (a valid starting point for whatever is scanned next); the rectangle that
    // can still hold a winner or a tie: within sqrt(best) of a query on both grid axes, the whole grid without a bound
    auto bounds = [&](int& bx0, int& bx1, int& by0, int& by1) -> bool {
        bool all = true;
        bx0 = 0x7fffffff, bx1 = -1, by0 = 0x7fffffff, by1 = -1;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            wave_argmin(best[i], bidx[i]);
            all = all && best[i] < INFINITY;
            const double rad = sqrt(best[i]) * (1.0 + 1e-9) + 1e-300;  // inflated against the rounding of sqrt / the subtractions
            double xl = qx[i] - rad, xh = qx[i] + rad, yl = qy[i] - rad, yh = qy[i] + rad;
            xl -= fabs(xl) * 1e-15;
            xh += fabs(xh) * 1e-15;
            yl -= fabs(yl) * 1e-15;
            yh += fabs(yh) * 1e-15;
            bx0 = min(bx0, cell_of(xl, g.lo0, g.s0, g.r0)), bx1 = max(bx1, cell_of(xh, g.lo0, g.s0, g.r0));
            by0 = min(by0, cell_of(yl, g.lo1, g.s1, g.r1)), by1 = max(by1, cell_of(yh, g.lo1, g.s1, g.r1));
        }
        if (!all) bx0 = 0, by0 = 0, bx1 = g.r0 - 1, by1 = g.r1 - 1;
        return all;
    };
    // phase 1: the ring of cells around the group's queries, grown until every query has an upper bound (4 rings at most)
    int x0 = 0, x1 = -1, y0 = 0, y1 = -1;
    int rx0 = 0, rx1 = -1, ry0 = 0, ry1 = -1;
    for (int ring = 1; ring <= 4; ++ring) {
        rx0 = max(ux0 - ring, 0), rx1 = min(ux1 + ring, g.r0 - 1);
        ry0 = max(uy0 - ring, 0), ry1 = min(uy1 + ring, g.r1 - 1);
        unsigned long long ta = COUNT ? __builtin_amdgcn_s_memrealtime() : 0ull;
        for (int yb = ry0; yb <= ry1; yb += PF_WAVE) scan_rows(yb + lane <= ry1 ? yb + lane : -1, rx0, rx1);
        unsigned long long tb = COUNT ? __builtin_amdgcn_s_memrealtime() : 0ull;
        const bool all_bounded = bounds(x0, x1, y0, y1);
        if constexpr (COUNT) t_scan += tb - ta, t_bounds += __builtin_amdgcn_s_memrealtime() - tb;
        if (all_bounded) break;
    }
    // phase 2: the rows of the rectangle from the group's own row outwards, a few at a time (4 + 4, 8 + 8, ... 32 + 32), and
    // after each batch the rectangle is re-derived from what has been found: bounds only shrink, and when the first one
    // is loose (clouds that are poorly aligned) the outer rows and columns drop out before they are read
    if (!(x0 >= rx0 && x1 <= rx1 && y0 >= ry0 && y1 <= ry1)) {  // (else the ring already held every candidate)
        const int mid = min(max((uy0 + uy1) / 2, y0), y1);
        int lo = mid, hi = mid - 1;  // rows [lo, hi] are done
        int w = 4;
        while (lo > y0 || hi < y1) {
            const int na = max(min(w, y1 - hi), 0), nb = max(min(w, lo - y0), 0);  // (a side that is finished may be past its bound)
            unsigned long long ta = COUNT ? __builtin_amdgcn_s_memrealtime() : 0ull;
            scan_rows(lane < na ? hi + 1 + lane : (lane < na + nb ? lo - 1 - (lane - na) : -1), x0, x1);
            unsigned long long tb = COUNT ? __builtin_amdgcn_s_memrealtime() : 0ull;
            hi += na;
            lo -= nb;
            w = min(2 * w, PF_WAVE / 2);
            int nx0, nx1, ny0, ny1;
            bounds(nx0, nx1, ny0, ny1);
            if constexpr (COUNT) t_scan += tb - ta, t_bounds += __builtin_amdgcn_s_memrealtime() - tb;
            x0 = max(x0, nx0), x1 = min(x1, nx1);
            y0 = max(y0, ny0), y1 = min(y1, ny1);
        }
    }
#pragma unroll
    for (int i = 0; i < G; ++i) {
        if (lane == i && i < nq) {
            const int64_t dst = qry_orig[q0 + i];
            idx_out[dst] = bidx[i];
            d2_out[dst] = best[i];
        }
    }
    if constexpr (COUNT) {
        if (lane == 0 && visited) {
            atomicAdd(visited, seen * (unsigned long long)nq);
            // the wave's own figures (diagnostics: pf_knn_wave_stats), one record per wave behind the counter - no atomics
            // (nine of them per wave on one cache line made the counted search six times slower)
            unsigned long long* rec = visited + 8 + 8 * group;
            rec[0] = __builtin_amdgcn_s_memrealtime() - t_begin;
            rec[1] = seen;
            rec[2] = n_chunks;
            rec[3] = n_scans;
            rec[4] = t_scan;
            rec[5] = t_bounds;
            rec[6] = n_pass;
        }
    }
}

// ---- small problems (both sets <= PF_KNN_SMALL points, k = 1): the exhaustive scan in ONE launch -----------------------------
// eigsort's 3-D search among 5000 sample points (eigsort.py:203-204) spent ~20 launches on building a grid for a kernel
// of 38 us.  Here a wave owns a few queries (coordinates wave-uniform), its lanes scan every 64th reference against all
// of them, and a lexicographic (distance, index) reduction across the wave picks the winner: dist2's operations in its
// order, lowest index on ties - the same bits and indices as every other path.
constexpr int64_t PF_KNN_SMALL = 16384;
constexpr int knn_small_group(int d) { return d <= 4 ? 8 : 4; }

template <int D>
__global__ __launch_bounds__(PF_BLOCK) void k_knn_small(const double* __restrict__ ref, int64_t n_ref, const double* __restrict__ qry,
                                                        int64_t n_qry, int64_t* __restrict__ idx_out, double* __restrict__ d2_out) {
    constexpr int G = knn_small_group(D);
    const int lane = threadIdx.x & (PF_WAVE - 1);
    const int64_t group = (int64_t)blockIdx.x * (PF_BLOCK / PF_WAVE) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x / PF_WAVE));
    const int64_t q0 = group * G;
    if (q0 >= n_qry) return;  // (wave-uniform)
    const int nq = n_qry - q0 < G ? (int)(n_qry - q0) : G;
    double q[G][D], best[G];
    int32_t bidx[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int64_t qi = q0 + (i < nq ? i : nq - 1);
#pragma unroll
        for (int c = 0; c < D; ++c) q[i][c] = qry[qi * D + c];
        best[i] = INFINITY;
        bidx[i] = 0x7fffffff;
    }
    for (int64_t r = lane; r < n_ref; r += PF_WAVE) {
        double x[D];
#pragma unroll
        for (int c = 0; c < D; ++c) x[c] = ref[r * D + c];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < D; ++c) {  // dist2's operations, in its order
                const double df = q[i][c] - x[c];
                const double sq = df * df;
                s = (c == 0) ? sq : s + sq;
            }
            if (s < best[i]) {  // (a lane meets its references in rising order: the first of equals stays)
                best[i] = s;
                bidx[i] = (int32_t)r;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < G; ++i) {
        wave_argmin(best[i], bidx[i]);
        if (lane == i && i < nq) {
            idx_out[q0 + i] = bidx[i];
            d2_out[q0 + i] = best[i];
        }
    }
}

template <int D>
static int launch_knn_small(pf_ctx* c) {
    const int64_t groups = (c->knn_nqry + knn_small_group(D) - 1) / knn_small_group(D);
    k_knn_small<D><<<(unsigned)((groups + PF_BLOCK / PF_WAVE - 1) / (PF_BLOCK / PF_WAVE)), PF_BLOCK, 0, c->stream>>>(
        c->knn_ref, c->knn_nref, c->knn_qry, c->knn_nqry, c->knn_idx, c->knn_d2);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

__global__ __launch_bounds__(PF_BLOCK) void k_gather_rows_soa(const double* __restrict__ pts, const int32_t* __restrict__ order,
                                                              int64_t n, int d, int64_t ld, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int64_t src = order[i];
    for (int c = 0; c < d; ++c) out[(int64_t)c * ld + i] = pts[src * d + c];
}

template <int D, int K>
int launch_knn_k(pf_ctx* c) {
    // Block size: one wave per block for shallow coordinates (d <= 6: the usual spectral embeddings; whether the lane-
    // private scan applies is then decided per wave, and a stray lane with a large square drags only 63 others into the
    // shared scan: 1.60 -> 1.39 ms at 250k x 250k, d = 5), four waves for deep ones, whose large squares make the shared
    // LDS tiles the common case (1M x 1M, d = 10: 161 ms with 256 threads, 172 with 64).
    constexpr int BS = D <= 6 ? PF_WAVE : PF_BLOCK;
    // d >= 10 stays with the one-query-per-lane kernel: only two queries' coordinates fit a wave's scalar registers there, so
    // a candidate load is shared by two queries instead of by a block, and the lane-per-query kernel's wave-level early
    // exit after d/2 coordinates pays more (measured, grouped / lane-per-query: unrelated 250k x 250k d = 6: 4.2 / 18.3 ms,
    // d = 8: 12.6 / 25.6, d = 9: 29.7 / 35.7, d = 12: 114 / 56; 1M x 1M noisy copies d = 10: 545 / 333; keeping 8 queries'
    // coordinates and best lists in LDS instead was slower still: d = 10: 78 ms, d = 12: 139)
    if constexpr (K == 1 && D <= 9) {
        const int64_t waves = (c->knn_nqry + knn_group(D) - 1) / knn_group(D);
        if (c->knn_count_on) {
            // [0] pairs, then from [8] on eight words per wave (pf_knn_wave_stats)
            const size_t words = 8 + 8 * (size_t)waves;
            if (c->knn_visited && c->knn_visited_words < words) {
                pf_free(c->stream, c->knn_visited);
                c->knn_visited = nullptr;
            }
            if (!c->knn_visited) {
                PF_HIP(pf_malloc(c->stream, (void**)&c->knn_visited, words * sizeof(unsigned long long)));
                c->knn_visited_words = words;
            }
            c->knn_visited_waves = waves;
            PF_HIP(hipMemsetAsync(c->knn_visited, 0, words * sizeof(unsigned long long), c->stream));
            k_knn_coop<D, true><<<(unsigned)((waves + PF_BLOCK / PF_WAVE - 1) / (PF_BLOCK / PF_WAVE)), PF_BLOCK, 0, c->stream>>>(
                c->knn_ref_soa, c->knn_ref_ld, c->knn_ref_orig, c->knn_cell_start, c->knn_qry_s, c->knn_qry_orig, c->knn_nqry,
                (const KnnGrid*)c->knn_grid, c->knn_idx, c->knn_d2, c->knn_visited);
        } else {
            k_knn_coop<D><<<(unsigned)((waves + PF_BLOCK / PF_WAVE - 1) / (PF_BLOCK / PF_WAVE)), PF_BLOCK, 0, c->stream>>>(
                c->knn_ref_soa, c->knn_ref_ld, c->knn_ref_orig, c->knn_cell_start, c->knn_qry_s, c->knn_qry_orig, c->knn_nqry,
                (const KnnGrid*)c->knn_grid, c->knn_idx, c->knn_d2);
        }
    } else {
        k_knn_grid<D, K, BS, 512><<<(unsigned)((c->knn_nqry + BS - 1) / BS), BS, 0, c->stream>>>(
            c->knn_ref_s, c->knn_ref_orig, c->knn_cell_start, c->knn_nref, c->knn_qry_s, c->knn_qry_orig, c->knn_nqry,
            (const KnnGrid*)c->knn_grid, c->knn_idx, c->knn_d2);
    }
    PF_HIP(hipGetLastError());
    return PF_OK;
}

template <int D>
int launch_knn(pf_ctx* c) {
    if (c->knn_k == 1) return launch_knn_k<D, 1>(c);
    if constexpr (D <= 4) {  // k > 1: the 3-NN of focusr.py:409-412 lives in 3-D
        if (c->knn_k == 2) return launch_knn_k<D, 2>(c);
        if (c->knn_k == 3) return launch_knn_k<D, 3>(c);
        if (c->knn_k == 4) return launch_knn_k<D, 4>(c);
    }
    pf_set_error("pf_knn: k = %d with d = %d is not supported (k <= 4 needs d <= 4)", c->knn_k, D);
    return PF_E_ARG;
}

template <typename T>
int grow(hipStream_t st, T** p, int64_t* cap, int64_t need) {
    if (need <= *cap) return PF_OK;
    pf_free(st, *p);
    *p = nullptr;
    *cap = 0;
    PF_HIP(pf_malloc(st, (void**)p, sizeof(T) * (size_t)need));
    *cap = need;
    return PF_OK;
}

// counting sort by cell key (keys < n_buckets): histogram, exclusive scan, scatter through per-bucket cursors.  The
// order inside a bucket is whatever the atomics make it: the search does not care (the minimum of (distance, original
// index) does not depend on the order candidates or queries are visited in).
// (the histogram is taken by k_cell_keys; the scatter carries the point's row along, so no gather pass follows)
__global__ __launch_bounds__(PF_BLOCK) void k_bucket_scatter(const unsigned* __restrict__ keys, int64_t n, const int32_t* __restrict__ start,
                                                             int32_t* __restrict__ cursor, unsigned* __restrict__ key_out,
                                                             int32_t* __restrict__ orig_out, const double* __restrict__ pts, int d,
                                                             double* __restrict__ rows_out) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= n) return;
    const unsigned k = keys[i];
    const int64_t pos = start[k] + atomicAdd(&cursor[k], 1);
    key_out[pos] = k;
    orig_out[pos] = (int32_t)i;
    for (int c = 0; c < d; ++c) rows_out[pos * d + c] = pts[i * d + c];
}

// sort one point set by grid cell: sorted keys, original indices, gathered rows.  250k points: a memset and 4 small
// launches where hipCUB's radix sort dispatches a merge sort of 17 launches (~107 us) for arrays below 1M items.
int sort_points(pf_ctx* c, const double* pts, int64_t n, int d, int morton, int key_bits, unsigned* key_out, int32_t* orig_out,
                double* rows_out) {
    hipStream_t st = c->stream;
    unsigned* k0 = nullptr;
    int32_t* v0 = nullptr;
    void* tmp = nullptr;
    size_t bytes = 0;
    int rc = PF_OK;
    hipError_t e = pf_malloc(st, (void**)&k0, sizeof(unsigned) * n);
    // keys are cell ids (row-major: < res^2) or Morton codes of cells (< 4^ceil(log2 res))
    int64_t n_buckets = 1;
    {
        const int res = c->knn_res;
        if (morton) {
            int bits = 0;
            while ((1 << bits) < res) ++bits;
            n_buckets = (int64_t)1 << (2 * bits);
        } else {
            n_buckets = d == 1 ? res : (int64_t)res * res;
        }
    }
    // (PF_KNN_BUCKET_MAX, read per call: the tests send small inputs down the radix-sort path that only point sets of more
    // than 16M points take by themselves)
    int64_t bucket_max = (int64_t)1 << 22;
    if (const char* ev = getenv("PF_KNN_BUCKET_MAX")) bucket_max = atoll(ev);
    if (e == hipSuccess && n_buckets <= bucket_max) {
        int32_t* hist = nullptr;  // [n_buckets + 1] counts, [n_buckets + 1] starts, [n_buckets] cursors
        e = pf_malloc(st, (void**)&hist, sizeof(int32_t) * (size_t)(3 * n_buckets + 2));
        tmp = hist;
        if (e == hipSuccess) e = hipMemsetAsync(hist, 0, sizeof(int32_t) * (size_t)(3 * n_buckets + 2), st);
        if (e == hipSuccess) {
            int32_t* start = hist + n_buckets + 1;
            int32_t* cursor = start + n_buckets + 1;
            k_cell_keys<<<nblk(n), PF_BLOCK, 0, st>>>(pts, n, d, (const KnnGrid*)c->knn_grid, morton, k0, nullptr, hist);
            e = hipGetLastError();
            if (e == hipSuccess && pf_exclusive_scan_i32(st, hist, start, n_buckets + 1) != PF_OK) e = hipErrorUnknown;
            if (e == hipSuccess) {
                k_bucket_scatter<<<nblk(n), PF_BLOCK, 0, st>>>(k0, n, start, cursor, key_out, orig_out, pts, d, rows_out);
                e = hipGetLastError();
            }
        }
    } else {
        if (e == hipSuccess) e = pf_malloc(st, (void**)&v0, sizeof(int32_t) * n);
        if (e == hipSuccess) {
            k_cell_keys<<<nblk(n), PF_BLOCK, 0, st>>>(pts, n, d, (const KnnGrid*)c->knn_grid, morton, k0, v0, nullptr);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k0, key_out, v0, orig_out, (int)n, 0, key_bits, st);
        if (e == hipSuccess) e = pf_malloc(st, &tmp, bytes);
        if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(tmp, bytes, k0, key_out, v0, orig_out, (int)n, 0, key_bits, st);
        if (e == hipSuccess) {
            k_gather_rows<<<nblk(n), PF_BLOCK, 0, st>>>(pts, orig_out, n, d, rows_out);
            e = hipGetLastError();
        }
    }
    if (e != hipSuccess) {
        pf_set_error("pf_knn: %s", hipGetErrorString(e));
        rc = PF_E_HIP;
    }
    pf_free(st, k0);
    pf_free(st, v0);
    pf_free(st, tmp);
    return rc;
}

}  // namespace

extern "C" {

// device buffers and grid parameters of a (n_ref, n_qry, d) problem; the coordinates are filled in by the caller
static int knn_prepare(pf_ctx* c, int64_t n_ref, int64_t n_qry, int32_t d) {
    PF_CHECK(c != nullptr, PF_E_ARG, "pf_knn_upload: ctx is NULL");
    const int32_t k = c->knn_k_next;
    c->knn_k_next = 1;
    PF_CHECK(k >= 1 && k <= 4 && k <= n_ref, PF_E_ARG, "pf_knn: k = %d out of range (1..4, <= n_ref)", k);
    PF_CHECK(n_ref > 0 && n_ref < ((int64_t)1 << 31) && n_qry > 0 && n_qry < ((int64_t)1 << 31) && d >= 1 && d <= 16, PF_E_ARG,
             "pf_knn_upload: n_ref %lld, n_qry %lld, d %d out of range (1 <= d <= 16)", (long long)n_ref, (long long)n_qry, d);
    PF_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    c->knn_ready = c->knn_done = false;
    PF_TRY(grow(st, &c->knn_ref, &c->knn_cap_ref, n_ref * 16));
    PF_TRY(grow(st, &c->knn_ref_s, &c->knn_cap_ref_s, n_ref * 16));
    PF_TRY(grow(st, &c->knn_ref_key, &c->knn_cap_ref_key, n_ref));
    c->knn_ref_ld = (n_ref + PF_WAVE - 1) & ~(int64_t)(PF_WAVE - 1);
    PF_TRY(grow(st, &c->knn_ref_soa, &c->knn_cap_ref_soa, c->knn_ref_ld * d));
    // grid resolution: ~4 references per cell if they were spread over the plane
    static const double per_cell = [] { const char* e = getenv("PF_KNN_PER_CELL"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 4.0; }();
    int res = (int)sqrt((double)n_ref / per_cell);
    res = res < 4 ? 4 : (res > 2048 ? 2048 : res);
    c->knn_res = res;
    PF_TRY(grow(st, &c->knn_cell_start, &c->knn_cap_cell, (int64_t)res * res + 2));
    PF_TRY(grow(st, &c->knn_ref_orig, &c->knn_cap_ref_orig, n_ref));
    PF_TRY(grow(st, &c->knn_qry, &c->knn_cap_qry, n_qry * 16));
    PF_TRY(grow(st, &c->knn_qry_s, &c->knn_cap_qry_s, n_qry * 16));
    PF_TRY(grow(st, &c->knn_qry_key, &c->knn_cap_qry_key, n_qry));
    PF_TRY(grow(st, &c->knn_qry_orig, &c->knn_cap_qry_orig, n_qry));
    PF_TRY(grow(st, &c->knn_idx, &c->knn_cap_idx, n_qry * k));
    PF_TRY(grow(st, &c->knn_d2, &c->knn_cap_d2, n_qry * k));
    c->knn_k = k;
    if (!c->knn_ext) PF_HIP(pf_malloc(st, (void**)&c->knn_ext, 32 * sizeof(unsigned long long)));
    if (!c->knn_grid) PF_HIP(pf_malloc(st, &c->knn_grid, sizeof(KnnGrid)));
    c->knn_nref = n_ref;
    c->knn_nqry = n_qry;
    c->knn_d = d;
    return PF_OK;
}

// coord[i][c] = fin[i][col[c]] * scale[c]; the d <= 16 columns and scales travel as kernel arguments (four small
// host-to-device copies and a synchronisation otherwise: ~0.1 ms per call)
struct CoordMap {
    int32_t col[16];
    double scale[16];
};
__global__ __launch_bounds__(PF_BLOCK) void k_coords_from_final(const double* __restrict__ fin, int64_t n, int32_t fc, int32_t d,
                                                                CoordMap m, double* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (e >= n * d) return;
    const int64_t i = e / d;
    const int32_t c = (int32_t)(e - i * d);
    out[e] = fin[i * fc + m.col[c]] * m.scale[c];
}

int pf_knn_upload(pf_ctx* c, const double* ref, int64_t n_ref, const double* qry, int64_t n_qry, int32_t d) {
    PF_CHECK(c && ref && qry, PF_E_ARG, "pf_knn_upload: NULL argument");
    PF_TRY(knn_prepare(c, n_ref, n_qry, d));
    hipStream_t st = c->stream;
    PF_HIP(hipMemcpyAsync(c->knn_ref, ref, sizeof(double) * n_ref * d, hipMemcpyHostToDevice, st));
    PF_HIP(hipMemcpyAsync(c->knn_qry, qry, sizeof(double) * n_qry * d, hipMemcpyHostToDevice, st));
    PF_HIP(hipStreamSynchronize(st));
    c->knn_ready = true;
    return PF_OK;
}

int pf_knn_run(pf_ctx* c) {
    PF_CHECK(c != nullptr, PF_E_ARG, "pf_knn_run: ctx is NULL");
    PF_CHECK(c->knn_ready, PF_E_STATE, "pf_knn_run: no uploaded problem");
    PF_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int d = c->knn_d;
    PF_HIP(hipEventRecord(c->ev0, st));
    if (c->knn_k == 1 && c->knn_mode == 0 && d <= 8 && c->knn_nref <= PF_KNN_SMALL && c->knn_nqry <= PF_KNN_SMALL) {
        int r = PF_E_ARG;
        switch (d) {
            case 1: r = launch_knn_small<1>(c); break;
            case 2: r = launch_knn_small<2>(c); break;
            case 3: r = launch_knn_small<3>(c); break;
            case 4: r = launch_knn_small<4>(c); break;
            case 5: r = launch_knn_small<5>(c); break;
            case 6: r = launch_knn_small<6>(c); break;
            case 7: r = launch_knn_small<7>(c); break;
            case 8: r = launch_knn_small<8>(c); break;
            default: break;
        }
        PF_TRY(r);
        PF_HIP(hipEventRecord(c->ev1, st));
        PF_HIP(hipEventSynchronize(c->ev1));
        float sms = 0.f;
        PF_HIP(hipEventElapsedTime(&sms, c->ev0, c->ev1));
        c->knn_ms = sms;
        c->knn_done = true;
        return PF_OK;
    }
    PF_HIP(hipMemsetAsync(c->knn_ext, 0xff, 16 * sizeof(unsigned long long), st));
    PF_HIP(hipMemsetAsync(c->knn_ext + 16, 0x00, 16 * sizeof(unsigned long long), st));
    const int res = c->knn_res;
    const int64_t n_cells = d == 1 ? res : (int64_t)res * res;
    int cell_bits = 1;
    while (((int64_t)1 << cell_bits) < n_cells) ++cell_bits;
    k_extent<<<256, PF_BLOCK, 0, st>>>(c->knn_ref, c->knn_nref, d, c->knn_ext);
    // deep coordinates (k = 1): a grid over two axes prunes two of d coordinates; the box hierarchy prunes with all of them
    const bool tree = c->knn_k == 1 && (c->knn_mode == 2 || (c->knn_mode == 0 && d >= PF_KNN_TREE_MIN_D));
    if (tree) {
        if (c->knn_nqry >= 32768) PF_TRY(pf_downloads_release(c));
        PF_TRY(pf_knn_tree_run(c));
        PF_HIP(hipEventRecord(c->ev1, st));
        PF_HIP(hipEventSynchronize(c->ev1));
        float tms = 0.f;
        PF_HIP(hipEventElapsedTime(&tms, c->ev0, c->ev1));
        c->knn_ms = tms;
        c->knn_done = true;
        return PF_OK;
    }
    k_make_grid<<<1, 1, 0, st>>>(c->knn_ext, d, res, (KnnGrid*)c->knn_grid);
    PF_HIP(hipGetLastError());
    PF_TRY(sort_points(c, c->knn_ref, c->knn_nref, d, 0, cell_bits, c->knn_ref_key, c->knn_ref_orig, c->knn_ref_s));
    k_cell_start<<<nblk(n_cells + 1), PF_BLOCK, 0, st>>>(c->knn_ref_key, c->knn_nref, n_cells, c->knn_cell_start);
    PF_HIP(hipGetLastError());
    if (c->knn_k == 1) {
        k_gather_rows_soa<<<nblk(c->knn_nref), PF_BLOCK, 0, st>>>(c->knn_ref, c->knn_ref_orig, c->knn_nref, d, c->knn_ref_ld, c->knn_ref_soa);
        PF_HIP(hipGetLastError());
    }
    PF_TRY(sort_points(c, c->knn_qry, c->knn_nqry, d, 1, 32, c->knn_qry_key, c->knn_qry_orig, c->knn_qry_s));
    // a large search is the long kernel behind which the eigenvector downloads that were held back may leave
    if (c->knn_nqry >= 32768) PF_TRY(pf_downloads_release(c));
    int r = PF_E_ARG;
    switch (d) {
        case 1: r = launch_knn<1>(c); break;
        case 2: r = launch_knn<2>(c); break;
        case 3: r = launch_knn<3>(c); break;
        case 4: r = launch_knn<4>(c); break;
        case 5: r = launch_knn<5>(c); break;
        case 6: r = launch_knn<6>(c); break;
        case 7: r = launch_knn<7>(c); break;
        case 8: r = launch_knn<8>(c); break;
        case 9: r = launch_knn<9>(c); break;
        case 10: r = launch_knn<10>(c); break;
        case 11: r = launch_knn<11>(c); break;
        case 12: r = launch_knn<12>(c); break;
        case 13: r = launch_knn<13>(c); break;
        case 14: r = launch_knn<14>(c); break;
        case 15: r = launch_knn<15>(c); break;
        case 16: r = launch_knn<16>(c); break;
        default: break;
    }
    PF_TRY(r);
    PF_HIP(hipEventRecord(c->ev1, st));
    PF_HIP(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    PF_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->knn_ms = ms;
    c->knn_done = true;
    return PF_OK;
}

/* Candidate-query pairs the grid search (k = 1, d <= 9) evaluated in the last search run with counting on; the call also
 * sets the switch for the searches to come (a separate kernel instantiation: the default search does not count). */
int pf_knn_count(pf_ctx* c, int32_t enable_counting, int64_t* pairs) {
    PF_CHECK(c != nullptr, PF_E_ARG, "pf_knn_count: ctx is NULL");
    unsigned long long h = 0ull;
    if (c->knn_visited) {
        PF_HIP(hipMemcpyAsync(&h, c->knn_visited, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        PF_HIP(hipStreamSynchronize(c->stream));
    }
    if (pairs) *pairs = (int64_t)h;
    c->knn_count_on = enable_counting != 0;
    return PF_OK;
}

/* Diagnostics of the last COUNTED grid search: the waves' own run times (10 ns ticks of the device's constant clock) -
 * their sum, the slowest wave, the number of waves - and the candidates of the wave that scanned most. */
int pf_knn_wave_stats(pf_ctx* c, double* sum_us, double* max_us, int64_t* waves, int64_t* max_candidates, double* detail /* [5]: chunks, scans, scan us, bounds us, candidates that pass the off-plane coordinates (sums) */) {
    PF_CHECK(c != nullptr, PF_E_ARG, "pf_knn_wave_stats: ctx is NULL");
    const int64_t nw = c->knn_visited ? c->knn_visited_waves : 0;
    std::vector<unsigned long long> h((size_t)(8 * nw), 0ull);
    if (nw > 0) {
        PF_HIP(hipMemcpyAsync(h.data(), c->knn_visited + 8, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, c->stream));
        PF_HIP(hipStreamSynchronize(c->stream));
    }
    unsigned long long sum = 0, mx = 0, mc = 0, ch = 0, sc = 0, ts = 0, tb = 0, np = 0;
    for (int64_t w = 0; w < nw; ++w) {
        const unsigned long long* r = h.data() + 8 * w;
        sum += r[0];
        if (r[0] > mx) mx = r[0], mc = r[1];
        ch += r[2], sc += r[3], ts += r[4], tb += r[5], np += r[6];
    }
    if (sum_us) *sum_us = 0.01 * (double)sum;
    if (max_us) *max_us = 0.01 * (double)mx;
    if (waves) *waves = nw;
    if (max_candidates) *max_candidates = (int64_t)mc;
    if (detail) detail[0] = (double)ch, detail[1] = (double)sc, detail[2] = 0.01 * (double)ts, detail[3] = 0.01 * (double)tb, detail[4] = (double)np;
    return PF_OK;
}

int pf_knn_mode(pf_ctx* c, int32_t mode) {
    PF_CHECK(c != nullptr && mode >= 0 && mode <= 2, PF_E_ARG, "pf_knn_mode: bad argument");
    c->knn_mode = mode;
    return PF_OK;
}

int pf_knn_download(pf_ctx* c, int64_t* idx_out, double* d2_out) {
    PF_CHECK(c != nullptr && idx_out != nullptr, PF_E_ARG, "pf_knn_download: NULL argument");
    PF_CHECK(c->knn_done, PF_E_STATE, "pf_knn_download: pf_knn_run has not completed");
    PF_HIP(hipSetDevice(c->device));
    const size_t count = (size_t)c->knn_nqry * c->knn_k;
    const size_t bytes = sizeof(int64_t) * count;
    if (bytes * (d2_out ? 2 : 1) <= ((size_t)8 << 20)) {
        // results of up to 8 MB come back through pinned memory and a copy kernel: no DMA engine, which the eigenvector
        // downloads in flight on the copy stream may be holding (pf_copy_by_kernel)
        unsigned char* pin = nullptr;
        PF_TRY(pf_pinned_scratch(c, bytes * (d2_out ? 2 : 1), reinterpret_cast<void**>(&pin), 2));
        PF_TRY(pf_copy_by_kernel(c->stream, c->knn_idx, pin, bytes));
        if (d2_out) PF_TRY(pf_copy_by_kernel(c->stream, c->knn_d2, pin + bytes, bytes));
        PF_HIP(hipStreamSynchronize(c->stream));
        memcpy(idx_out, pin, bytes);
        if (d2_out) memcpy(d2_out, pin + bytes, bytes);
        return PF_OK;
    }
    PF_HIP(hipMemcpyAsync(idx_out, c->knn_idx, bytes, hipMemcpyDeviceToHost, c->stream));
    if (d2_out) PF_HIP(hipMemcpyAsync(d2_out, c->knn_d2, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    PF_HIP(hipStreamSynchronize(c->stream));
    return PF_OK;
}

int pf_knn(pf_ctx* c, const double* ref, int64_t n_ref, const double* qry, int64_t n_qry, int32_t d, int32_t k,
           int64_t* idx_out, double* d2_out) {
    PF_CHECK(c != nullptr, PF_E_ARG, "pf_knn: ctx is NULL");
    c->knn_k_next = k;
    PF_TRY(pf_knn_upload(c, ref, n_ref, qry, n_qry, d));
    PF_TRY(pf_knn_run(c));
    return pf_knn_download(c, idx_out, d2_out);
}

int pf_knn1(pf_ctx* c, const double* ref, int64_t n_ref, const double* qry, int64_t n_qry, int32_t d, int64_t* idx_out,
            double* d2_out) {
    PF_TRY(pf_knn_upload(c, ref, n_ref, qry, n_qry, d));
    PF_TRY(pf_knn_run(c));
    return pf_knn_download(c, idx_out, d2_out);
}

int pf_knn1_blocks(pf_ctx* c, const double* ref_block, int64_t n_ref, int32_t ref_stride, const double* qry_block, int64_t n_qry,
                   int32_t qry_stride, int32_t d, const int32_t* col_ref, const double* scale_ref, const int32_t* col_qry,
                   const double* scale_qry, int64_t* idx_out, double* d2_out) {
    PF_CHECK(c && ref_block && qry_block && col_ref && scale_ref && col_qry && scale_qry && idx_out, PF_E_ARG,
             "pf_knn1_blocks: NULL argument");
    PF_CHECK(d >= 1 && d <= 16 && ref_stride >= 1 && qry_stride >= 1, PF_E_ARG, "pf_knn1_blocks: d = %d / strides out of range", d);
    for (int32_t k = 0; k < d; ++k)
        PF_CHECK(col_ref[k] >= 0 && col_ref[k] < ref_stride && col_qry[k] >= 0 && col_qry[k] < qry_stride, PF_E_ARG,
                 "pf_knn1_blocks: column %d out of range", k);
    PF_TRY(knn_prepare(c, n_ref, n_qry, d));
    hipStream_t st = c->stream;
    CoordMap mr{}, mq{};
    for (int32_t k = 0; k < d; ++k) {
        mr.col[k] = col_ref[k], mr.scale[k] = scale_ref[k];
        mq.col[k] = col_qry[k], mq.scale[k] = scale_qry[k];
    }
    k_coords_from_final<<<nblk(n_ref * d), PF_BLOCK, 0, st>>>(ref_block, n_ref, ref_stride, d, mr, c->knn_ref);
    k_coords_from_final<<<nblk(n_qry * d), PF_BLOCK, 0, st>>>(qry_block, n_qry, qry_stride, d, mq, c->knn_qry);
    PF_HIP(hipGetLastError());
    c->knn_ready = true;
    PF_TRY(pf_knn_run(c));
    return pf_knn_download(c, idx_out, d2_out);
}

int pf_knn1_graphs(pf_graph* ref_g, pf_graph* qry_g, int32_t d, const int32_t* col_ref, const double* scale_ref,
                   const int32_t* col_qry, const double* scale_qry, int64_t* idx_out, double* d2_out) {
    PF_CHECK(ref_g && qry_g, PF_E_ARG, "pf_knn1_graphs: NULL argument");
    PF_CHECK(ref_g->ctx == qry_g->ctx, PF_E_ARG, "pf_knn1_graphs: the two graphs must share one ctx");
    PF_CHECK(ref_g->final_vecs && qry_g->final_vecs, PF_E_STATE, "pf_knn1_graphs: no pf_finalize_vectors result is resident");
    return pf_knn1_blocks(ref_g->ctx, ref_g->final_vecs, ref_g->n, ref_g->final_count, qry_g->final_vecs, qry_g->n, qry_g->final_count, d,
                          col_ref, scale_ref, col_qry, scale_qry, idx_out, d2_out);
}

int pf_final_device(pf_graph* g, double** block, int64_t* n_rows, int32_t* n_cols) {
    PF_CHECK(g && block && n_rows && n_cols, PF_E_ARG, "pf_final_device: NULL argument");
    PF_CHECK(g->final_vecs != nullptr, PF_E_STATE, "pf_final_device: no pf_finalize_vectors result is resident");
    *block = g->final_vecs;
    *n_rows = g->n;
    *n_cols = g->final_count;
    return PF_OK;
}

}  // extern "C"
