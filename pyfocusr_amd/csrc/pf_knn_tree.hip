// Exact 1-nearest-neighbour search for DEEP coordinates (d >= 7 by default): a bounding-box hierarchy over ALL d coordinates.
//
// Replaces `scipy.spatial.KDTree(target).query(source)` of /root/reference/pyfocusr/focusr.py:351-353 where the
// spectral embedding is deep (BASELINE config C5: 1M x 1M vertices, d = 10).  pf_knn.hip prunes with a grid over the
// reference set's two widest axes: every reference within sqrt(best) of a query ON THOSE TWO AXES is a candidate.  For a
// 2-manifold embedded in 10 dimensions whose two clouds are poorly aligned (nearest distances ~0.15 of the cube) that
// rectangle holds 50-100k references per query, of which the other eight coordinates would have excluded all but a few
// thousand - what scipy's k-d tree exploits (~2000 visited per query), and why the grid scan took 127 ms at C5.
//
// Structure (built per call, ~0.5 ms at 1M x 10):
//   * references sorted along a Morton curve over their (up to) five widest axes: points of one patch of the embedded
//     surface become neighbours in the order;
//   * LEAVES of 64 consecutive points, coordinate-major inside a leaf (a wave reads one coordinate of a leaf as 512
//     contiguous bytes), each with its axis-aligned bounding box in all d coordinates;
//   * SUPERS of 64 consecutive leaves with the box of their boxes; the supers' boxes are the top level (244 at 1M).
// Search: a wave owns G Morton-consecutive queries (their coordinates in scalar registers).  Lane l tests box l of the
// level at hand; the nearest super's nearest leaf gives the first bound; then every super / leaf whose box is within a
// query's current bound is visited, a leaf's 64 points one per lane against all G queries, and after every few leaves the
// lanes' (distance, original index) pairs are reduced lexicographically across the wave.
//
// Exactness.  The squared distance is pf_knn.hip's: sum over the coordinates, left to right, of (q_c - r_c)^2, separate
// multiply and add (-ffp-contract=off); smallest wins, lowest reference index on exact ties.  A box's distance is the SAME
// accumulation over gap_c = max(lo_c - q_c, q_c - hi_c, 0).  For every point r of the box |q_c - r_c| >= gap_c in real
// numbers, and every operation of the accumulation (the subtraction, the square, each addition) is monotone under
// rounding, so the computed box distance is <= the computed distance of every point in it: a box is skipped only when its
// distance is > a query's bound - no candidate that could win or tie is ever dropped, and the minimum of (distance, index)
// does not depend on the order of the visits.  Indices and distances are bit-identical to a brute force.
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "pf_internal.h"

namespace {

inline unsigned nblk(int64_t n) { return (unsigned)((n + PF_BLOCK - 1) / PF_BLOCK); }

constexpr int TREE_AXES = 5;

struct TreeGrid {
    int na;                 // axes in the key
    int bits;               // bits per axis
    int axis[TREE_AXES];
    double lo[TREE_AXES];
    double scale[TREE_AXES];  // cells per unit length (0 when the extent is 0)
};

__device__ __forceinline__ double tdec_f64(unsigned long long u) {
    return __longlong_as_double((long long)((u >> 63) ? (u & ~(1ull << 63)) : ~u));
}

// the (up to) five widest axes of the reference set, widest first; 30 key bits shared between them
__global__ void k_tree_grid(const unsigned long long* __restrict__ ext /* [2][16]: encoded min, max */, int d, TreeGrid* g) {
    if (threadIdx.x | blockIdx.x) return;
    double w[16];
    bool used[16];
    for (int a = 0; a < d; ++a) {
        w[a] = tdec_f64(ext[16 + a]) - tdec_f64(ext[a]);
        used[a] = false;
    }
    const int na = d < TREE_AXES ? d : TREE_AXES;
    g->na = na;
    g->bits = 30 / na;
    for (int k = 0; k < na; ++k) {
        int b = -1;
        for (int a = 0; a < d; ++a)
            if (!used[a] && (b < 0 || w[a] > w[b])) b = a;
        used[b] = true;
        g->axis[k] = b;
        g->lo[k] = tdec_f64(ext[b]);
        g->scale[k] = (w[b] > 0.0 && isfinite(w[b])) ? (double)(1 << g->bits) / w[b] : 0.0;
    }
}

// Morton key of a point over the grid's axes (bit j of axis k lands at position j * na + k); points outside the
// reference set's box (queries) are clamped
__global__ __launch_bounds__(PF_BLOCK) void k_tree_keys(const double* __restrict__ pts, int64_t n, int d, const TreeGrid* __restrict__ gp,
                                                        unsigned* __restrict__ keys, int32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= n) return;
    // (the grid is read where it lies: a local copy indexed by the run-time axis number k would live in SCRATCH memory -
    // 0.78 ms per million points instead of ~0.03)
    const int na = gp->na, bits = gp->bits;
    unsigned key = 0;
    const int cells = 1 << bits;
    for (int k = 0; k < na; ++k) {
        const double t = (pts[i * d + gp->axis[k]] - gp->lo[k]) * gp->scale[k];
        const unsigned c = t > 0.0 ? (t >= (double)cells ? (unsigned)(cells - 1) : (unsigned)t) : 0u;
        for (int j = 0; j < bits; ++j) key |= ((c >> j) & 1u) << (j * na + k);
    }
    keys[i] = key;
    vals[i] = (int32_t)i;
}

// leaf storage: point p of the sorted order -> leaf p / 64, place p % 64, coordinate-major inside the leaf; the last
// leaf is filled up with copies of the last point under the index INT_MAX (they lose every tie)
__global__ __launch_bounds__(PF_BLOCK) void k_tree_leaves(const double* __restrict__ ref, const int32_t* __restrict__ order, int64_t n,
                                                          int d, int64_t n_slots, double* __restrict__ pts, int32_t* __restrict__ orig) {
    const int64_t p = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (p >= n_slots) return;
    const int64_t src = order[p < n ? p : n - 1];
    orig[p] = p < n ? (int32_t)src : 0x7fffffff;
    double* out = pts + (p >> 6) * (int64_t)d * PF_WAVE + (p & (PF_WAVE - 1));
    for (int c = 0; c < d; ++c) out[(int64_t)c * PF_WAVE] = ref[src * d + c];
}

// boxes of the leaves, stored per super: leaf_lo[(S * d + c) * 64 + leaf % 64]; leaves past the end get an empty box
// (lo = +inf, hi = -inf: infinitely far from everything)
__global__ __launch_bounds__(PF_BLOCK) void k_tree_leaf_boxes(const double* __restrict__ pts, int32_t n_leaf, int32_t n_sup, int d,
                                                              double* __restrict__ leaf_lo, double* __restrict__ leaf_hi) {
    const int64_t e = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (e >= (int64_t)n_sup * PF_WAVE * d) return;
    const int32_t L = (int32_t)(e / d);
    const int c = (int)(e - (int64_t)L * d);
    double lo = INFINITY, hi = -INFINITY;
    if (L < n_leaf) {
        const double* p = pts + ((int64_t)L * d + c) * PF_WAVE;
        for (int j = 0; j < PF_WAVE; ++j) {
            const double v = p[j];
            lo = v < lo ? v : lo;
            hi = v > hi ? v : hi;
        }
    }
    const int64_t o = ((int64_t)(L >> 6) * d + c) * PF_WAVE + (L & (PF_WAVE - 1));
    leaf_lo[o] = lo;
    leaf_hi[o] = hi;
}

// boxes of the supers: sup_lo[c * ns_pad + S]
__global__ __launch_bounds__(PF_BLOCK) void k_tree_super_boxes(const double* __restrict__ leaf_lo, const double* __restrict__ leaf_hi,
                                                               int32_t n_sup, int32_t ns_pad, int d, double* __restrict__ sup_lo,
                                                               double* __restrict__ sup_hi) {
    const int64_t e = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (e >= (int64_t)ns_pad * d) return;
    const int32_t S = (int32_t)(e / d);
    const int c = (int)(e - (int64_t)S * d);
    double lo = INFINITY, hi = -INFINITY;
    if (S < n_sup) {
        const double* pl = leaf_lo + ((int64_t)S * d + c) * PF_WAVE;
        const double* ph = leaf_hi + ((int64_t)S * d + c) * PF_WAVE;
        for (int j = 0; j < PF_WAVE; ++j) {
            lo = pl[j] < lo ? pl[j] : lo;
            hi = ph[j] > hi ? ph[j] : hi;
        }
    }
    sup_lo[(int64_t)c * ns_pad + S] = lo;
    sup_hi[(int64_t)c * ns_pad + S] = hi;
}

struct TreeArgs {
    const double* pts;
    const int32_t* orig;
    const double* leaf_lo;
    const double* leaf_hi;
    const double* sup_lo;
    const double* sup_hi;
    const double* qry;
    const int32_t* qry_order;
    int64_t n_qry;
    int32_t n_leaf, n_sup, ns_pad;
    int64_t* idx_out;
    double* d2_out;
    unsigned long long* counters;  // nullable: [0] leaves scanned, [1] supers opened (diagnostics)
};

__device__ __forceinline__ void tree_argmin(double& s, int32_t& o) {
#pragma unroll
    for (int off = PF_WAVE / 2; off > 0; off >>= 1) {
        const double s2 = __shfl_xor(s, off, PF_WAVE);
        const int32_t o2 = __shfl_xor(o, off, PF_WAVE);
        const bool take = s2 < s || (s2 == s && o2 < o);
        s = take ? s2 : s;
        o = take ? o2 : o;
    }
}

// squared distance of q to the box [lo, hi] (one coordinate every `cs` doubles): the accumulation of the point distance
template <int D>
__device__ __forceinline__ double box_d2(const double (&q)[D], const double* __restrict__ lo, const double* __restrict__ hi, int64_t cs) {
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const double t1 = lo[(int64_t)c * cs] - q[c], t2 = q[c] - hi[(int64_t)c * cs];
        const double g = fmax(fmax(t1, t2), 0.0);
        const double sq = g * g;
        s = (c == 0) ? sq : s + sq;
    }
    return s;
}

constexpr int tree_group(int d) { return d <= 12 ? 4 : 2; }
constexpr int TREE_BATCH = 2;  // leaves scanned between two reductions of the lanes' bests

template <int D>
__global__ __launch_bounds__(PF_BLOCK) void k_knn_tree(TreeArgs t) {
    constexpr int G = tree_group(D);
    const int lane = threadIdx.x & (PF_WAVE - 1);
    const int64_t group = (int64_t)blockIdx.x * (PF_BLOCK / PF_WAVE) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x / PF_WAVE));
    const int64_t q0 = group * G;
    if (q0 >= t.n_qry) return;  // (wave-uniform)
    const int nq = t.n_qry - q0 < G ? (int)(t.n_qry - q0) : G;
    double q[G][D];  // wave-uniform (a short group replays its last query)
    int32_t qdst[G];
    double best[G];
    int32_t bidx[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        qdst[i] = t.qry_order[q0 + (i < nq ? i : nq - 1)];
#pragma unroll
        for (int c = 0; c < D; ++c) q[i][c] = t.qry[(int64_t)qdst[i] * D + c];
        best[i] = INFINITY;
        bidx[i] = 0x7fffffff;
    }
    unsigned long long n_scanned = 0, n_opened = 0;

    auto scan_leaf = [&](int32_t L) {
        const double* p = t.pts + (int64_t)L * D * PF_WAVE + lane;
        double x[D];
#pragma unroll
        for (int c = 0; c < D; ++c) x[c] = p[c * PF_WAVE];
        const int32_t o = t.orig[(int64_t)L * PF_WAVE + lane];
        ++n_scanned;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            double s = 0.0;
            constexpr int P = D >= 6 ? D / 2 : D;  // most candidates are out after half of the coordinates
#pragma unroll
            for (int c = 0; c < P; ++c) {
                const double df = q[i][c] - x[c];
                const double sq = df * df;
                s = (c == 0) ? sq : s + sq;
            }
            if constexpr (P < D) {
                if (!__any(s <= best[i])) continue;  // (the partial sum is a prefix of the same accumulation and only grows)
#pragma unroll
                for (int c = P; c < D; ++c) {
                    const double df = q[i][c] - x[c];
                    s = s + df * df;
                }
            }
            if (s < best[i] || (s == best[i] && o < bidx[i])) {
                best[i] = s;
                bidx[i] = o;
            }
        }
    };
    auto reduce = [&]() {
#pragma unroll
        for (int i = 0; i < G; ++i) tree_argmin(best[i], bidx[i]);
    };

    // ---- the first bound: the leaf nearest to the group's first query inside the super nearest to it
    int32_t L0;
    {
        double sb = INFINITY;
        int32_t sa = 0;
        for (int32_t s0 = 0; s0 < t.ns_pad; s0 += PF_WAVE) {
            const double dd = box_d2<D>(q[0], t.sup_lo + s0 + lane, t.sup_hi + s0 + lane, t.ns_pad);
            if (dd < sb) sb = dd, sa = s0 + lane;
        }
        tree_argmin(sb, sa);
        const int64_t lb = (int64_t)sa * D * PF_WAVE + lane;
        double lbest = box_d2<D>(q[0], t.leaf_lo + lb, t.leaf_hi + lb, PF_WAVE);
        int32_t la = lane;
        tree_argmin(lbest, la);
        L0 = sa * PF_WAVE + la;
        scan_leaf(L0);
        reduce();
    }
    // ---- every super whose box is within a query's bound, every leaf of it whose box is
    for (int32_t s0 = 0; s0 < t.ns_pad; s0 += PF_WAVE) {
        double sd[G];
#pragma unroll
        for (int i = 0; i < G; ++i) sd[i] = box_d2<D>(q[i], t.sup_lo + s0 + lane, t.sup_hi + s0 + lane, t.ns_pad);
        unsigned long long sdone = 0ull;
        while (true) {
            bool need = false;
#pragma unroll
            for (int i = 0; i < G; ++i) need = need || sd[i] <= best[i];
            const unsigned long long sm = __ballot(need) & ~sdone;
            if (!sm) break;
            const int sbit = __ffsll((long long)sm) - 1;
            sdone |= 1ull << sbit;
            const int32_t S = s0 + sbit;
            if (S >= t.n_sup) continue;
            ++n_opened;
            const int64_t lb = (int64_t)S * D * PF_WAVE + lane;
            double ld[G];
#pragma unroll
            for (int i = 0; i < G; ++i) ld[i] = box_d2<D>(q[i], t.leaf_lo + lb, t.leaf_hi + lb, PF_WAVE);
            unsigned long long ldone = 0ull;
            while (true) {
                bool ln = false;
#pragma unroll
                for (int i = 0; i < G; ++i) ln = ln || ld[i] <= best[i];
                unsigned long long lm = __ballot(ln) & ~ldone;
                if (!lm) break;
                for (int b = 0; b < TREE_BATCH && lm; ++b) {
                    const int lbit = __ffsll((long long)lm) - 1;
                    lm &= lm - 1ull;
                    ldone |= 1ull << lbit;
                    const int32_t L = S * PF_WAVE + lbit;
                    if (L < t.n_leaf && L != L0) scan_leaf(L);
                }
                reduce();
            }
        }
    }
#pragma unroll
    for (int i = 0; i < G; ++i) {
        if (lane == i && i < nq) {
            t.idx_out[qdst[i]] = bidx[i];
            t.d2_out[qdst[i]] = best[i];
        }
    }
    if (t.counters && lane == 0) {
        atomicAdd(t.counters, n_scanned);
        atomicAdd(t.counters + 1, n_opened);
    }
}

template <typename T>
int tgrow(hipStream_t st, T** p, int64_t* cap, int64_t need) {
    if (need <= *cap) return PF_OK;
    pf_free(st, *p);
    *p = nullptr;
    *cap = 0;
    PF_HIP(pf_malloc(st, (void**)p, sizeof(T) * (size_t)need));
    *cap = need;
    return PF_OK;
}

int sort_by_key(hipStream_t st, unsigned* k_in, int32_t* v_in, unsigned* k_out, int32_t* v_out, int64_t n) {
    size_t bytes = 0;
    void* tmp = nullptr;
    PF_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k_in, k_out, v_in, v_out, (int)n, 0, 30, st));
    PF_HIP(pf_malloc(st, &tmp, bytes));
    const hipError_t e = hipcub::DeviceRadixSort::SortPairs(tmp, bytes, k_in, k_out, v_in, v_out, (int)n, 0, 30, st);
    pf_free(st, tmp);
    PF_HIP(e);
    return PF_OK;
}

template <int D>
void launch_tree(pf_ctx* c, const TreeArgs& a) {
    const int64_t waves = (c->knn_nqry + tree_group(D) - 1) / tree_group(D);
    k_knn_tree<D><<<(unsigned)((waves + PF_BLOCK / PF_WAVE - 1) / (PF_BLOCK / PF_WAVE)), PF_BLOCK, 0, c->stream>>>(a);
}

}  // namespace

// The search of pf_knn_run for k = 1 through the box hierarchy: c->knn_ref / knn_qry hold the coordinates (row-major),
// c->knn_ext the encoded extents of the references; results into c->knn_idx / knn_d2.
int pf_knn_tree_run(pf_ctx* c) {
    hipStream_t st = c->stream;
    const int d = c->knn_d;
    const int64_t n = c->knn_nref, nq = c->knn_nqry;
    pf_knn_tree& T = c->knn_tree;
    const int64_t n_leaf = (n + PF_WAVE - 1) / PF_WAVE;
    const int64_t n_sup = (n_leaf + PF_WAVE - 1) / PF_WAVE;
    const int64_t ns_pad = (n_sup + PF_WAVE - 1) & ~(int64_t)(PF_WAVE - 1);
    PF_CHECK(n_leaf < ((int64_t)1 << 31) / PF_WAVE, PF_E_ARG, "pf_knn: too many references for the box hierarchy");
    PF_TRY(tgrow(st, &T.pts, &T.cap_pts, n_leaf * PF_WAVE * d));
    PF_TRY(tgrow(st, &T.orig, &T.cap_orig, n_leaf * PF_WAVE));
    PF_TRY(tgrow(st, &T.leaf_lo, &T.cap_leaf, 2 * n_sup * PF_WAVE * d));
    PF_TRY(tgrow(st, &T.sup_lo, &T.cap_sup, 2 * ns_pad * d));
    PF_TRY(tgrow(st, &T.qry_order, &T.cap_qry, nq));
    if (!T.grid) PF_HIP(pf_malloc(st, &T.grid, sizeof(TreeGrid)));
    if (!T.counters) {
        PF_HIP(pf_malloc(st, (void**)&T.counters, 2 * sizeof(unsigned long long)));
    }
    PF_HIP(hipMemsetAsync(T.counters, 0, 2 * sizeof(unsigned long long), st));
    double* leaf_hi = T.leaf_lo + n_sup * PF_WAVE * d;
    double* sup_hi = T.sup_lo + ns_pad * d;
    const int64_t nmax = std::max(n, nq);
    unsigned *k0 = nullptr, *k1 = nullptr;
    int32_t *v0 = nullptr, *v1 = nullptr;
    int rc = PF_OK;
    do {
        if (pf_malloc(st, (void**)&k0, sizeof(unsigned) * nmax) != hipSuccess || pf_malloc(st, (void**)&k1, sizeof(unsigned) * nmax) != hipSuccess ||
            pf_malloc(st, (void**)&v0, sizeof(int32_t) * nmax) != hipSuccess || pf_malloc(st, (void**)&v1, sizeof(int32_t) * nmax) != hipSuccess) {
            pf_set_error("pf_knn: out of device memory (box hierarchy)");
            rc = PF_E_HIP;
            break;
        }
        k_tree_grid<<<1, 1, 0, st>>>(c->knn_ext, d, (TreeGrid*)T.grid);
        k_tree_keys<<<nblk(n), PF_BLOCK, 0, st>>>(c->knn_ref, n, d, (const TreeGrid*)T.grid, k0, v0);
        if ((rc = sort_by_key(st, k0, v0, k1, v1, n)) != PF_OK) break;
        k_tree_leaves<<<nblk(n_leaf * PF_WAVE), PF_BLOCK, 0, st>>>(c->knn_ref, v1, n, d, n_leaf * PF_WAVE, T.pts, T.orig);
        k_tree_leaf_boxes<<<nblk(n_sup * PF_WAVE * d), PF_BLOCK, 0, st>>>(T.pts, (int32_t)n_leaf, (int32_t)n_sup, d, T.leaf_lo, leaf_hi);
        k_tree_super_boxes<<<nblk(ns_pad * d), PF_BLOCK, 0, st>>>(T.leaf_lo, leaf_hi, (int32_t)n_sup, (int32_t)ns_pad, d, T.sup_lo, sup_hi);
        k_tree_keys<<<nblk(nq), PF_BLOCK, 0, st>>>(c->knn_qry, nq, d, (const TreeGrid*)T.grid, k0, v0);
        if ((rc = sort_by_key(st, k0, v0, k1, T.qry_order, nq)) != PF_OK) break;
        TreeArgs a{T.pts, T.orig, T.leaf_lo, leaf_hi, T.sup_lo, sup_hi, c->knn_qry, T.qry_order, nq, (int32_t)n_leaf, (int32_t)n_sup,
                   (int32_t)ns_pad, c->knn_idx, c->knn_d2, T.count_visits ? T.counters : nullptr};
        switch (d) {
            case 1: launch_tree<1>(c, a); break;
            case 2: launch_tree<2>(c, a); break;
            case 3: launch_tree<3>(c, a); break;
            case 4: launch_tree<4>(c, a); break;
            case 5: launch_tree<5>(c, a); break;
            case 6: launch_tree<6>(c, a); break;
            case 7: launch_tree<7>(c, a); break;
            case 8: launch_tree<8>(c, a); break;
            case 9: launch_tree<9>(c, a); break;
            case 10: launch_tree<10>(c, a); break;
            case 11: launch_tree<11>(c, a); break;
            case 12: launch_tree<12>(c, a); break;
            case 13: launch_tree<13>(c, a); break;
            case 14: launch_tree<14>(c, a); break;
            case 15: launch_tree<15>(c, a); break;
            case 16: launch_tree<16>(c, a); break;
            default: rc = PF_E_ARG; break;
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            pf_set_error("pf_knn (box hierarchy): %s", hipGetErrorString(e));
            rc = PF_E_HIP;
        }
    } while (0);
    pf_free(st, k0);
    pf_free(st, k1);
    pf_free(st, v0);
    pf_free(st, v1);
    return rc;
}

// diagnostics of the last box-hierarchy search (pf_knn_tree_count(ctx, 1) switches the counting on): leaves scanned and
// supers opened, summed over the waves
extern "C" int pf_knn_tree_stats(pf_ctx* c, int32_t enable_counting, int64_t* leaves_scanned, int64_t* supers_opened) {
    PF_CHECK(c != nullptr, PF_E_ARG, "pf_knn_tree_stats: ctx is NULL");
    unsigned long long h[2] = {0ull, 0ull};
    if (c->knn_tree.counters) {
        PF_HIP(hipMemcpyAsync(h, c->knn_tree.counters, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        PF_HIP(hipStreamSynchronize(c->stream));
    }
    if (leaves_scanned) *leaves_scanned = (int64_t)h[0];
    if (supers_opened) *supers_opened = (int64_t)h[1];
    c->knn_tree.count_visits = enable_counting != 0;
    return PF_OK;
}
