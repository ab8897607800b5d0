// Laplacian assembly on the device.
//
// Replaces the per-edge Python loop of Graph.get_weighted_adjacency_matrix and the scipy
// sparse algebra of get_degree_matrix / get_G_matrix / get_laplacian_matrix
// (/root/reference/pyfocusr/graph.py:148-178, 216-226).  Semantics reproduced:
//   * W[i,j] = 1/sqrt((xi-xj)^2 summed left to right), ASSIGNED per directed polygon edge
//     (set semantics: duplicates collapse, one-way edges make W asymmetric);
//   * deg_i = sum_j W_ij accumulated left to right in column order (what lil.sum(axis=1) does);
//   * G = diag(1/(deg+1e-8)); L = G (D - W): L_ij = -(g_i W_ij), L_ii = g_i deg_i.
// This file is compiled with -ffp-contract=off so every product/sum rounds as numpy's does:
// W, deg, L come out bit-identical to the reference's matrices (tests/golden).
//
// Pipeline (all on the ctx stream): count directed edges per source vertex (coalesced sweep of
// the face list, one int atomic per edge) -> scan -> scatter (col, w) into per-vertex segments
// -> per-vertex sort + unique -> scan -> compact into CSR and reduce deg -> symmetry probe ->
// union-find components -> SELL-64 operator storage.
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <thread>

#include "pf_internal.h"
#include "pf_launch.h"

namespace {

struct k_count_edges {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ faces, int64_t n_edges,
                                                          int32_t vpf, int64_t n, int32_t* __restrict__ cnt,
                                                          int32_t* __restrict__ rank, int32_t* __restrict__ flags,
                                                          double* __restrict__ quarter) {
    const int64_t e = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (e == 0 && quarter) *quarter = 0.25;  // k_face_bound's running minimum starts there (instead of two fills of its own)
    if (e >= n_edges) return;
    const int64_t f = e / vpf;
    const int32_t k = (int32_t)(e - f * vpf);
    const int32_t src = faces[e];
    const int32_t dst = faces[f * vpf + (k + 1 == vpf ? 0 : k + 1)];
    if (src < 0 || src >= n || dst < 0 || dst >= n) {
        atomicOr(flags, 1);
        return;
    }
    if (src == dst) {
        atomicOr(flags, 2);
        return;
    }
    // the edge's place in its vertex's list is what the count was when it arrived: k_scatter_edges needs no atomic of its
    // own (1.5 M atomics on scattered addresses per 250k mesh and pass - the two meshes of a pair queue for the same units)
    rank[e] = atomicAdd(&cnt[src], 1);
}
};

struct k_scatter_edges {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ faces,
                                                            const double* __restrict__ pts, int64_t n_edges,
                                                            int32_t vpf, int64_t n, const int32_t* __restrict__ start,
                                                            const int32_t* __restrict__ rank, int32_t* __restrict__ rcol,
                                                            double* __restrict__ rw, int32_t* __restrict__ flags) {
    const int64_t e = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (e >= n_edges) return;
    const int64_t f = e / vpf;
    const int32_t k = (int32_t)(e - f * vpf);
    const int32_t src = faces[e];
    const int32_t dst = faces[f * vpf + (k + 1 == vpf ? 0 : k + 1)];
    if (src < 0 || src >= n || dst < 0 || dst >= n || src == dst) return;  // flagged by k_count_edges; the host looks later
    const double dx = pts[3 * (int64_t)src + 0] - pts[3 * (int64_t)dst + 0];
    const double dy = pts[3 * (int64_t)src + 1] - pts[3 * (int64_t)dst + 1];
    const double dz = pts[3 * (int64_t)src + 2] - pts[3 * (int64_t)dst + 2];
    const double d2 = (dx * dx + dy * dy) + dz * dz;  // np.sum(np.square(.)): left to right
    const double wv = 1.0 / sqrt(d2);                 // graph.py:177-178
    if (!isfinite(wv)) atomicOr(flags, 4);            // coincident or non-finite vertices: the reference stores inf / nan
    const int32_t slot = start[src] + rank[e];
    rcol[slot] = dst;
    rw[slot] = wv;
}
};

// A rigorous upper bound for the spectrum of the normalised Laplacian of a closed triangle mesh, face by face.
// If every undirected edge lies in exactly two triangles, W = sum over triangles t of A_t, the triangle's own adjacency
// with half of each edge weight.  The normalised adjacency of a weighted triangle (weights a, b, c) has eigenvalues 1, mu1,
// mu2 with mu1 + mu2 = -1 (zero trace) and mu1 mu2 = P := 2abc / ((a+b)(a+c)(b+c)) (the determinants; 0 < P <= 1/4, = 1/4
// for equal weights), so its smallest one is mu_t = -(1 + sqrt(1 - 4 P)) / 2, and x^T A_t x >= mu_t x^T D_t x.  Summing over
// the triangles (the D_t add up to D: each edge at a vertex is in two of them): x^T W x >= mu_min x^T D x, i.e.
//   lambda_max(G^1/2 (D - W) G^1/2) <= 1 - mu_min = 1 + (1 + sqrt(1 - 4 P_min)) / 2
// against the generic bound 2: 1.69 for the 250k blobs, 1.60 for the bundled 5k mesh (their lambda_max: 1.58, 1.49).  The
// Chebyshev filter's degree scales with the square root of the interval it has to damp.  P is homogeneous of degree 0: the
// weights themselves (1 / edge length) serve.  out: bits of min P (positive doubles order like their bit patterns).
struct k_face_bound {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ faces, const double* __restrict__ pts,
                                                         int64_t n_faces, int64_t n, unsigned long long* __restrict__ out) {
    __shared__ double red[PF_BLOCK / PF_WAVE];
    const int64_t f = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    double p = 0.25;
    if (f < n_faces) {
        const int32_t i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
        if (i0 >= 0 && i0 < n && i1 >= 0 && i1 < n && i2 >= 0 && i2 < n) {
            double w[3];
            const int32_t v[4] = {i0, i1, i2, i0};
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const double dx = pts[3 * (int64_t)v[e]] - pts[3 * (int64_t)v[e + 1]];
                const double dy = pts[3 * (int64_t)v[e] + 1] - pts[3 * (int64_t)v[e + 1] + 1];
                const double dz = pts[3 * (int64_t)v[e] + 2] - pts[3 * (int64_t)v[e + 1] + 2];
                w[e] = 1.0 / sqrt((dx * dx + dy * dy) + dz * dz);
            }
            const double q = 2.0 * w[0] * w[1] * w[2] / ((w[0] + w[1]) * (w[0] + w[2]) * (w[1] + w[2]));
            p = (q > 0.0 && q < 0.25) ? q : (q >= 0.25 ? 0.25 : 0.0);  // (degenerate or non-finite: no bound from this face)
        }
    }
#pragma unroll
    for (int off = PF_WAVE / 2; off > 0; off >>= 1) {
        const double o = __shfl_down(p, off, PF_WAVE);
        p = o < p ? o : p;
    }
    if ((threadIdx.x & (PF_WAVE - 1)) == 0) red[threadIdx.x / PF_WAVE] = p;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = red[0];
        for (int i = 1; i < PF_BLOCK / PF_WAVE; ++i) m = red[i] < m ? red[i] : m;
        atomicMin(out, (unsigned long long)__double_as_longlong(m));
    }
}
};

// ---- m-space (round 4): the mesh renumbered by the Morton rank of its points before anything is assembled ----------------
// pts_m[m] = pts[morder[m]]; faces_m[e] = mrank[faces[e]] (an index out of range stays out of range: k_count_edges flags it)
struct k_renumber_points {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const double* __restrict__ pts, const int32_t* __restrict__ morder,
                                                              int64_t n, double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (t >= 3 * n) return;
    const int64_t m = t / 3;
    out[t] = pts[3 * (int64_t)morder[m] + (t - 3 * m)];
}
};
struct k_renumber_faces {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ faces, const int32_t* __restrict__ mrank,
                                                             int64_t n_edges, int64_t n, int32_t* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (e >= n_edges) return;
    const int32_t v = faces[e];
    out[e] = (v >= 0 && v < n) ? mrank[v] : (v < 0 ? v : (int32_t)n);
}
};

// one thread per vertex: insertion-sort its (col, w) segment by column, drop duplicate columns
// (a directed edge listed by two faces carries the same weight), report the unique count.
// The segments of a block's 256 vertices are one contiguous piece of rcol / rw (~3000 entries): it is staged in LDS by
// the whole block (coalesced), every thread sorts its own segment there, and the piece goes back the same way - the
// sort's ~40 dependent moves per row are LDS accesses instead of global ones (89 -> ~25 us at 250k vertices).  A block
// whose piece does not fit (very high degrees) sorts in place in global memory, as before.
constexpr int PF_SORT_CAP = 4096;  // entries staged per block: 48 KB
// `key` (nullable): the ORIGINAL vertex number of an m-space column - a row's entries stand in the order of the caller's
// numbering whatever space they are stored in: the degree is their sum from left to right (lil_matrix.sum, graph.py:217)
// and the boundary format is CSR with sorted original columns.
template <typename C, typename W>
__device__ __forceinline__ int32_t sort_unique_segment(C* c, W* v, int32_t b, int32_t e, const int32_t* __restrict__ key = nullptr) {
    for (int32_t a = b + 1; a < e; ++a) {
        const int32_t cc = c[a];
        const double vv = v[a];
        const int32_t kc = key ? key[cc] : cc;
        int32_t p = a - 1;
        while (p >= b && (key ? key[c[p]] : c[p]) > kc) {
            c[p + 1] = c[p];
            v[p + 1] = v[p];
            --p;
        }
        c[p + 1] = cc;
        v[p + 1] = vv;
    }
    int32_t u = 0;
    for (int32_t a = b; a < e; ++a) {
        if (a == b || c[a] != c[b + u - 1]) {
            c[b + u] = c[a];
            v[b + u] = v[a];
            ++u;
        }
    }
    return u;
}

struct k_sort_unique_rows {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ start, int64_t n,
                                                               int32_t* __restrict__ rcol, double* __restrict__ rw,
                                                               int32_t* __restrict__ ucnt, const int32_t* __restrict__ key) {
    __shared__ int32_t s_c[PF_SORT_CAP];
    __shared__ double s_v[PF_SORT_CAP];
    const int64_t i0 = (int64_t)blockIdx.x * PF_BLOCK;
    const int64_t i = i0 + threadIdx.x;
    const int64_t i1 = i0 + PF_BLOCK < n ? i0 + PF_BLOCK : n;
    const int32_t lo = start[i0], hi = start[i1];  // (block-uniform)
    if (hi - lo > PF_SORT_CAP) {
        if (i < n) ucnt[i] = sort_unique_segment(rcol, rw, start[i], start[i + 1], key);
        return;
    }
    for (int32_t a = threadIdx.x; a < hi - lo; a += PF_BLOCK) {
        s_c[a] = rcol[lo + a];
        s_v[a] = rw[lo + a];
    }
    __syncthreads();
    if (i < n) ucnt[i] = sort_unique_segment(s_c, s_v, start[i] - lo, start[i + 1] - lo, key);
    __syncthreads();
    for (int32_t a = threadIdx.x; a < hi - lo; a += PF_BLOCK) {
        rcol[lo + a] = s_c[a];
        rw[lo + a] = s_v[a];
    }
}
};

struct k_compact_rows {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ start,
                                                           const int32_t* __restrict__ rowptr, int64_t n,
                                                           const int32_t* __restrict__ rcol,
                                                           const double* __restrict__ rw, int32_t* __restrict__ col,
                                                           double* __restrict__ w, double* __restrict__ deg,
                                                           double* __restrict__ g, double* __restrict__ sg) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int32_t src = start[i];
    const int32_t b = rowptr[i], cntu = rowptr[i + 1] - b;
    double d = 0.0;
    for (int32_t a = 0; a < cntu; ++a) {
        const double v = rw[src + a];
        col[b + a] = rcol[src + a];
        w[b + a] = v;
        d += v;  // left to right in column order: lil_matrix.sum(axis=1)
    }
    deg[i] = d;
    const double gi = 1.0 / (d + 1e-8);  // graph.py:219
    g[i] = gi;
    sg[i] = sqrt(gi);
}
};

struct k_row_stats {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ rowptr, int64_t n,
                                                        int32_t* __restrict__ stats /* [0]=n_isolated [1]=max_degree */) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    const int32_t c = i < n ? rowptr[i + 1] - rowptr[i] : -1;
    // atomics on one address serialise (~10 ns each): one pair per block, not one per row
    __shared__ int32_t s_max, s_iso;
    if (threadIdx.x == 0) {
        s_max = 0;
        s_iso = 0;
    }
    __syncthreads();
    int32_t m = c;
#pragma unroll
    for (int off = PF_WAVE / 2; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, PF_WAVE));
    const int iso = __popcll(__ballot(c == 0));
    if ((threadIdx.x & (PF_WAVE - 1)) == 0) {
        atomicMax(&s_max, m);
        if (iso) atomicAdd(&s_iso, iso);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_max > 0) atomicMax(&stats[1], s_max);
        if (s_iso) atomicAdd(&stats[0], s_iso);
    }
}
};

// values == nullptr: structural symmetry (mesh graphs: W_ij and W_ji are then equal bit for bit);
// otherwise the values must agree as well (general Laplacians handed in as CSR).
struct k_symmetry_probe {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col,
                                                             const double* __restrict__ values, int64_t n,
                                                             int32_t* __restrict__ asym, const int32_t* __restrict__ key = nullptr) {
    // eight threads per row, each with every eighth entry: the binary searches of a row (three dependent gathers each, in
    // rows that lie anywhere) are in flight together instead of one after the other (50 -> ~15 us at 250k vertices)
    const int64_t t = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    const int64_t i = t >> 3;
    if (i >= n) return;
    for (int32_t a = rowptr[i] + (int32_t)(t & 7), e = rowptr[i + 1]; a < e; a += 8) {
        const int32_t j = col[a];
        int32_t lo = rowptr[j], hi = rowptr[j + 1] - 1;
        bool found = false;
        const int32_t ki = key ? key[i] : (int32_t)i;  // (rows are sorted by the original number of their columns)
        while (lo <= hi) {
            const int32_t mid = (lo + hi) >> 1;
            const int32_t c = col[mid];
            if (c == (int32_t)i) {
                found = values == nullptr || values[mid] == values[a];
                break;
            }
            if ((key ? key[c] : c) < ki) lo = mid + 1; else hi = mid - 1;
        }
        if (!found) atomicAdd(asym, 1);  // one-way (or numerically unequal) entry
    }
}
};

// ---- weakly connected components -----------------------------------------------------------------------------
// Atomics on one address serialise at ~10 ns each on this part (measured: 4 k atomicMax on one word = 41 us), and every
// union-find on the GPU ends up hammering the few surviving roots: atomicMin hooking took 1.0 ms and a
// compare-and-swap union-find (the ECL-CC scheme) 1.6 ms for a 250k-vertex mesh.  So: no atomics at all.  Rounds of
// hooking (k_label_round below; until round 3 a hook and a flatten launch per round on a forest of stars, in the manner
// of Soman et al. 2010): every edge whose ends sit in different
// trees writes "larger root -> smaller root" with a plain store (any winner is a valid parent: it is smaller and in
// the same component, so the forest stays acyclic and every root with a smaller neighbouring star gets hooked), then
// every vertex is pointed at its root again.  The number of stars falls geometrically; the round that sees no
// differing edge proves the labelling, and the surviving root of a component is its smallest vertex index.
__device__ __forceinline__ int32_t uf_find(const int32_t* label, int32_t x) {
    int32_t p = label[x];
    while (p != x) {
        x = p;
        p = label[x];
    }
    return x;
}

// `key` (nullable: the vertex number itself): the ORDER the forest is built on.  In m-space the vertex numbers follow a
// space-filling curve, and trees that always hook towards the smaller number grow along the curve - long chains, slow
// rounds (13.8 us against 9.1 per round at 250k, measured); ordered by the caller's (unrelated) vertex numbers the forest
// is the one of round 3, its gathers are still m-space local, and the surviving root of a component is the vertex with
// the smallest ORIGINAL number - the label the boundary format hands out.
struct k_label_init {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ col, int64_t n, int64_t n_pad,
                                                         int32_t* __restrict__ label, const int32_t* __restrict__ key) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= n_pad) return;
    int32_t m = (int32_t)i;
    if (i < n) {
        int32_t km = key ? key[i] : (int32_t)i;
        for (int32_t a = rowptr[i]; a < rowptr[i + 1]; ++a) {
            const int32_t c = col[a];
            const int32_t kc = key ? key[c] : c;
            if (kc < km) m = c, km = kc;
        }
    }
    label[i] = m;  // parent <= child in the key's order, equality for roots only: a forest
}
};

struct k_label_compress {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(int32_t* label, int64_t n, const int32_t* prev = nullptr) {
    if (prev && *prev == 0) return;
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i < n) label[i] = uf_find(label, (int32_t)i);
}
};

// One ROUND in one launch (round 3): a vertex first points itself at its current root (the flattening that used to be a
// launch of its own), then hooks across its edges between ROOTS found by walking the pointers as they are at that moment.
// Pointers only ever move to smaller vertices of the same component, so a walk terminates and whatever it returns is
// an ancestor; a hook lost to a race is found again in the next round (the edge still joins two trees).  A round in
// which no edge joined two roots wrote nothing but flattenings: the forest is final (k_label_compress flattens it once
// more for the readers).  Half the launches of hook + compress, and fewer rounds (roots, not last round's stars).
#ifndef PF_CC_SWEEPS
#define PF_CC_SWEEPS 1
#endif
struct k_label_round {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t n,
                                                          int32_t* label, int32_t* differing, const int32_t* prev,
                                                          const int32_t* __restrict__ key) {
    if (prev && *prev == 0) return;  // the previous round changed nothing: converged (rounds are queued ahead, unasked)
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    bool any = false;
    if (i < n) {
        const int32_t b = rowptr[i], e = rowptr[i + 1];
        // (PF_CC_SWEEPS passes over the edges per launch; measured at 250k vertices: 3 passes make the first launches 96 / 78 us
        // instead of 25 / 25 and save no round - the tail is the ~log2(trees) halvings, not work per launch - so: 1.  A launch of
        // a nearly converged labelling is all launch latency,
        // and a second pass already sees what the first one - of every other vertex too - has joined)
        for (int sweep = 0; sweep < PF_CC_SWEEPS; ++sweep) {
            int32_t fu = uf_find(label, (int32_t)i);
            if (label[i] != fu) label[i] = fu;
            bool again = false;
            for (int32_t a = b; a < e; ++a) {
                const int32_t fv = uf_find(label, col[a]);
                if (fu != fv) {
                    const bool u_first = key ? key[fu] < key[fv] : fu < fv;
                    const int32_t lo = u_first ? fu : fv, hi = u_first ? fv : fu;
                    label[hi] = lo;
                    fu = lo;
                    again = true;
                }
            }
            any = any || again;
            if (!again) break;
        }
    }
    if (__any(any) && (threadIdx.x & (PF_WAVE - 1)) == 0) *differing = 1;
}
};

struct k_collect_roots {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ label,
                                                            const int32_t* __restrict__ rowptr, int64_t n,
                                                            int32_t* __restrict__ roots, int32_t* __restrict__ n_roots) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (label[i] == (int32_t)i && rowptr[i + 1] > rowptr[i]) {
        const int32_t k = atomicAdd(n_roots, 1);
        if (k < PF_MAX_ROOTS) roots[k] = (int32_t)i;
    }
}
};

// ---- SELL-64 ----------------------------------------------------------------------------------
struct k_slice_widths {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ perm, int64_t n,
                                                           int64_t n_slices, int64_t* __restrict__ width64) {
    const int64_t row = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;  // renumbered row
    int32_t c = 0;
    if (row < n) {
        const int32_t old = perm[row];
        c = rowptr[old + 1] - rowptr[old];
    }
#pragma unroll
    for (int off = PF_WAVE / 2; off > 0; off >>= 1) c = max(c, __shfl_xor(c, off, PF_WAVE));
    const int64_t s = row / PF_WAVE;
    if ((threadIdx.x & (PF_WAVE - 1)) == 0 && s < n_slices) width64[s] = (int64_t)c * PF_WAVE;
    if (row == 0) width64[n_slices] = 0;  // (the scan's extra element: no fill of its own)
}
};

// The same storage, one block per slice and one thread per stored ENTRY (round 3).  k_fill_sell walks a row's entries one
// after the other - two dependent gathers each (the entry, then iperm / sg of its column), seven times in a row per
// lane: 110-150 us for a 250k-vertex mesh whose vertices come in no particular order.  Here every entry of the slice is
// its own thread: the row's facts sit in LDS, entry q of the slice is written by thread q (coalesced, the layout of
// pf_sell_index inverted), and all gathers of a slice are in flight together.
struct k_fill_sell_entries {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                const double* __restrict__ w, const double* __restrict__ deg,
                                                                const double* __restrict__ g, const double* __restrict__ sg,
                                                                const int32_t* __restrict__ perm, const int32_t* __restrict__ iperm,
                                                                int64_t n, const int64_t* __restrict__ slice_ptr,
                                                                int32_t* __restrict__ scol, double* __restrict__ sval_rw,
                                                                double* __restrict__ sval_sym, double* __restrict__ diag) {
    __shared__ int32_t s_b[PF_WAVE], s_cnt[PF_WAVE];
    __shared__ double s_g[PF_WAVE], s_s[PF_WAVE];
    const int64_t s = blockIdx.x;
    const int64_t base = slice_ptr[s];
    const int32_t width = (int32_t)((slice_ptr[s + 1] - base) / PF_WAVE);
    const int t = threadIdx.x;
    if (t < PF_WAVE) {
        const int64_t row = s * PF_WAVE + t;
        int32_t b = 0, cnt = 0;
        double gi = 0.0, si = 0.0, d = 0.0;
        if (row < n) {
            const int32_t old = perm[row];
            b = rowptr[old];
            cnt = rowptr[old + 1] - b;
            gi = g[old];
            si = sg[old];
            d = gi * deg[old];  // L_ii = g_i deg_i  (graph.py:226)
        }
        diag[row] = d;
        s_b[t] = b;
        s_cnt[t] = cnt;
        s_g[t] = gi;
        s_s[t] = si;
    }
    __syncthreads();
    const int32_t pairs = width >> 1, total = width * PF_WAVE;
    for (int32_t q = t; q < total; q += PF_BLOCK) {
        int lane, j;
        if (q < pairs * 2 * PF_WAVE) {
            const int32_t r = q & (2 * PF_WAVE - 1);
            lane = r >> 1;
            j = 2 * (q / (2 * PF_WAVE)) + (r & 1);
        } else {
            lane = q - pairs * 2 * PF_WAVE;
            j = width - 1;
        }
        const int64_t idx = base + q;  // == pf_sell_index(base, width, j, lane)
        if (j < s_cnt[lane]) {
            const int32_t a = s_b[lane] + j;
            const int32_t c = col[a];
            const double wv = w[a];
            scol[idx] = iperm[c];
            sval_rw[idx] = -(s_g[lane] * wv);  // L_ij = g_i * (0 - W_ij)
            if (sval_sym) sval_sym[idx] = -(wv * (s_s[lane] * sg[c]));
        } else {
            scol[idx] = (int32_t)(s * PF_WAVE + lane);  // padding: zero weight on the row's own (in-window) column
            sval_rw[idx] = 0.0;
            if (sval_sym) sval_sym[idx] = 0.0;
        }
    }
}
};

struct k_l_offdiag {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ rowptr, const double* __restrict__ w,
                                                        const double* __restrict__ g, int64_t n, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= n) return;
    const double gi = g[i];
    for (int32_t a = rowptr[i]; a < rowptr[i + 1]; ++a) out[a] = -(gi * w[a]);
}
};

struct k_l_diag {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const double* __restrict__ deg, const double* __restrict__ g, int64_t n,
                                                     double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i < n) out[i] = g[i] * deg[i];
}
};

// ---- the boundary format of a graph assembled in m-space: CSR(W), deg, labels in the caller's vertex order, made on demand
// (pf_graph_download: reference-style views, tests; never on the timed path)
struct k_len_original {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ rowptr_m, const int32_t* __restrict__ mrank,
                                                           int64_t n, int32_t* __restrict__ len) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i > n) return;
    len[i] = i < n ? rowptr_m[mrank[i] + 1] - rowptr_m[mrank[i]] : 0;
}
};
// one thread per original row: its entries stand in the order of their original columns already (k_sort_unique_rows' key)
struct k_rows_original {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ rowptr_m, const int32_t* __restrict__ col_m,
                                                            const double* __restrict__ w_m, const double* __restrict__ g_m,
                                                            const double* __restrict__ deg_m, const int32_t* __restrict__ mrank,
                                                            const int32_t* __restrict__ morder, const int32_t* __restrict__ rowptr_o,
                                                            int64_t n, int32_t* __restrict__ col_o, double* __restrict__ w_o,
                                                            double* __restrict__ loff_o, double* __restrict__ deg_o,
                                                            double* __restrict__ ldiag_o) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int32_t m = mrank[i];
    const double gi = g_m[m];
    if (deg_o) deg_o[i] = deg_m[m];
    if (ldiag_o) ldiag_o[i] = gi * deg_m[m];
    const int32_t b = rowptr_m[m], cnt = rowptr_m[m + 1] - b, dst = rowptr_o[i];
    for (int32_t a = 0; a < cnt; ++a) {
        if (col_o) col_o[dst + a] = morder[col_m[b + a]];
        if (w_o) w_o[dst + a] = w_m[b + a];
        if (loff_o) loff_o[dst + a] = -(gi * w_m[b + a]);
    }
}
};
// component labels as the reference-side tests know them: the SMALLEST original vertex number of the component
struct k_label_min_original {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ label_m, const int32_t* __restrict__ morder,
                                                                 int64_t n, int32_t* __restrict__ smallest) {
    const int64_t m = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (m < n) atomicMin(&smallest[label_m[m]], morder[m]);
}
};
struct k_label_original {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ label_m, const int32_t* __restrict__ mrank,
                                                             const int32_t* __restrict__ smallest, int64_t n, int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i < n) out[i] = smallest[label_m[mrank[i]]];
}
};

inline unsigned nblk(int64_t n) { return (unsigned)((n + PF_BLOCK - 1) / PF_BLOCK); }

template <typename T>
int dev_alloc(hipStream_t st, T** p, int64_t count) {
    PF_HIP(pf_malloc(st, (void**)p, sizeof(T) * (size_t)std::max<int64_t>(count, 1)));
    return PF_OK;
}

}  // namespace

namespace {

// split a general CSR matrix into its diagonal and off-diagonals: w = -A_ij, deg = A_ii, g = sg = 1, so that the
// operator storage (-g_i w) reproduces A_ij and the dense diagonal g_i deg_i reproduces A_ii.
struct k_csr_count_offdiag {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                                int64_t n, int32_t* __restrict__ cnt, int32_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= n) return;
    int32_t c = 0, prev = -1;
    for (int32_t a = rp[i]; a < rp[i + 1]; ++a) {
        const int32_t j = ci[a];
        if (j < 0 || j >= n || j <= prev) atomicOr(flags, 1);  // out of range, unsorted or duplicated column
        prev = j;
        c += (j != (int32_t)i);
    }
    cnt[i] = c;
}
};

struct k_csr_split {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                        const double* __restrict__ va, const int32_t* __restrict__ rowptr,
                                                        int64_t n, int32_t* __restrict__ col, double* __restrict__ w,
                                                        double* __restrict__ deg, double* __restrict__ g, double* __restrict__ sg) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= n) return;
    int32_t o = rowptr[i];
    double d = 0.0;
    for (int32_t a = rp[i]; a < rp[i + 1]; ++a) {
        const int32_t j = ci[a];
        if (j == (int32_t)i) d = va[a];
        else {
            col[o] = j;
            w[o] = -va[a];
            ++o;
        }
    }
    deg[i] = d;
    g[i] = 1.0;
    sg[i] = 1.0;
}
};

// Common tail of the builders: CSR(W) (or the off-diagonals of a general Laplacian), deg, g, sg are in
// place; derive symmetry, statistics, components, the solver renumbering and the SELL-64 storage.
// everything small the host wants after the assembly kernels, packed into one block for one copy:
// [0..5) row statistics, [5] labelling still changing, [6] stored entries (rowptr[n]), [8..16) the caller's flags,
// [16..16 + PF_ROOTS_AHEAD) the first component roots
constexpr int PF_ROOTS_AHEAD = 16;
constexpr int PF_REPORT_INTS = 16 + PF_ROOTS_AHEAD + 2;  // (+ the 64 bits of the face bound's P_min)
struct k_report {
    static constexpr int BOUNDS = 1024;
    static __device__ __forceinline__ void run(const int32_t* __restrict__ stats, const int32_t* __restrict__ last_round, const int32_t* __restrict__ rowptr_n,
                         const int32_t* __restrict__ extra, const int32_t* __restrict__ roots, const int32_t* __restrict__ pmin_bits,
                         const int32_t* __restrict__ order_overflow, int32_t* __restrict__ out) {
    const int t = threadIdx.x;
    if (t < 5) out[t] = stats[t];
    if (t == 5) out[5] = *last_round;
    if (t == 6) out[6] = rowptr_n ? *rowptr_n : 0;
    if (t == 7) out[7] = *order_overflow;  // the renumbering by counting gave up (pf_compute_order): repeat it
    if (t >= 8 && t < 16) out[t] = extra ? extra[t - 8] : 0;
    if (t >= 16 && t < 16 + PF_ROOTS_AHEAD) out[t] = roots[t - 16];
    if (t >= 16 + PF_ROOTS_AHEAD && t < PF_REPORT_INTS) out[t] = pmin_bits ? pmin_bits[t - 16 - PF_ROOTS_AHEAD] : 0;
}
};

// `d_extra` / `h_extra` (8 ints, optional): device flags of the caller that ride in this job's one read-back; when
// any is set end() returns at once (PF_OK, extra_hit = true) and the caller reports ITS error.  `nnz_from_rowptr`:
// g->nnz_w is read back here too (the mesh path sizes col / w by their upper bound instead of waiting for the count).
// Two halves around the ONE synchronisation of a build: begin() queues everything up to the read-back on the graph's
// build stream and returns; end() waits, decides, and queues the SELL fill.  Two meshes of a pair run their halves
// interleaved on two streams (pf_graph_build_device2): their ~75 small kernels each overlap on the device.
struct FinishJob {
    pf_graph* g = nullptr;
    const double* d_pts = nullptr;
    bool numeric_symmetry = false;
    const int32_t* d_extra = nullptr;
    int32_t* h_extra = nullptr;
    bool extra_hit = false;
    bool nnz_from_rowptr = false;
    const unsigned long long* d_pmin = nullptr;
    double* h_pmin = nullptr;
    int sid = 0;  // 0: the ctx stream, 1: its second stream

    static constexpr int PF_CC_ROUNDS = 128;
    static constexpr int PF_CC_FIRST = 12;
    hipStream_t st = nullptr;
    int32_t *flags = nullptr, *d_roots = nullptr, *round_flags = nullptr, *stats = nullptr, *report = nullptr, *h_report = nullptr;
    int64_t* width64 = nullptr;
    void* pin = nullptr;
    size_t slice_bytes = 0;
    int round = 0;
    std::vector<void*> tmp;

    void release() {
        for (void* q : tmp) pf_free(st, q);
        tmp.clear();
    }
    ~FinishJob() { release(); }

    int begin() {
        PF_TRY(begin_stats_and_labels());
        return queue_order(false);
    }

    int begin_stats_and_labels() {
        st = g->build_stream ? g->build_stream : g->ctx->stream;
        const int64_t n = g->n;
        PF_TRY(dev_alloc(st, &flags, 8 + PF_CC_ROUNDS));  // the flags and the labelling rounds' flags: one block, one fill
        tmp.push_back(flags);
        round_flags = flags + 8;
        PF_TRY(dev_alloc(st, &width64, g->n_slices + 1));
        tmp.push_back(width64);
        PF_TRY(dev_alloc(st, &d_roots, PF_MAX_ROOTS));
        tmp.push_back(d_roots);
        PF_HIP(pfl::memset_words(st, flags, 0, sizeof(int32_t) * (8 + PF_CC_ROUNDS)));
        stats = flags + 2;  // [0] isolated, [1] max degree, [2] asym, [3] changed, [4] n_roots
        pfl::launch<k_row_stats>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, g->rowptr, n, stats);
        PF_HIP(hipGetLastError());
        pfl::launch<k_symmetry_probe>(dim3(nblk(8 * n)), dim3(PF_BLOCK), 0, st, g->rowptr, g->col, numeric_symmetry ? g->w : nullptr, n, stats + 2, g->morder);
        PF_HIP(hipGetLastError());

        // components
        pfl::launch<k_label_init>(dim3(nblk(g->n_pad)), dim3(PF_BLOCK), 0, st, g->rowptr, g->col, n, g->n_pad, g->label, g->morder);
        PF_HIP(hipGetLastError());
        // PF_CC_FIRST rounds are queued without asking: a round that follows a round without changes returns at once
        // (~2 us instead of ~15), and whether the last one still changed something is read back with everything else below
        for (round = 0; round < PF_CC_FIRST; ++round) {
            const int32_t* prev = round ? round_flags + round - 1 : nullptr;
            pfl::launch<k_label_round>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, g->rowptr, g->col, n, g->label, round_flags + round, prev, g->morder);
            PF_HIP(hipGetLastError());
        }
        pfl::launch<k_label_compress>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, g->label, n, nullptr);
        PF_HIP(hipGetLastError());
        pfl::launch<k_collect_roots>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, g->label, g->rowptr, n, d_roots, stats + 4);
        PF_HIP(hipGetLastError());
        return PF_OK;
    }

    // solver-internal renumbering (Morton order, degree-sorted windows), the slice widths in that order, and the ONE
    // read-back (two copies into pinned memory) for everything the host has to know before it can size the SELL storage:
    // the row statistics, whether the labelling had converged, the first few component roots (a mesh usually has one),
    // the caller's own flags, the number of stored entries (mesh path), whether the renumbering by counting gave up -
    // and the slice pointers.  `robust`: the second time round, with the general sort.
    int queue_order(bool robust) {
        const int64_t n = g->n;
        // (g->side_stream, set by a build that has forked: the order chain runs beside the labelling rounds, which it has
        // nothing to do with; the join comes before the report below)
        hipStream_t ls = g->side_stream ? g->side_stream : st;
        PF_TRY(pf_compute_order(g, d_pts, robust ? nullptr : flags));  // (flags[0]: free for this; the statistics start at flags + 2)
        if (robust) PF_HIP(pfl::memset_words(st, flags, 0, sizeof(int32_t)));
        if (!g->perm_m) g->perm_m = g->perm, g->iperm_m = g->iperm;  // (no m-space: a graph handed in as a matrix)
        pfl::launch<k_slice_widths>(dim3(nblk(g->n_pad)), dim3(PF_BLOCK), 0, ls, g->rowptr, g->perm_m, n, g->n_slices, width64);
        PF_HIP(hipGetLastError());
        PF_TRY(pf_exclusive_scan_i64(ls, width64, g->slice_ptr, g->n_slices + 1));
        if (g->side_stream) {  // join: the build stream waits for the side chain
            hipStream_t side = g->side_stream;
            hipEvent_t ev = g->ctx->join_ev;
            pfl::call(st, [=](hipStream_t s) {
                (void)hipEventRecord(ev, side);
                (void)hipStreamWaitEvent(s, ev, 0);
            });
            g->side_stream = nullptr;
        }
        if (!report) {
            PF_TRY(dev_alloc(st, &report, PF_REPORT_INTS));
            tmp.push_back(report);
        }
        pfl::launch<k_report>(dim3(1), dim3(PF_WAVE), 0, st, stats, round_flags + PF_CC_FIRST - 1, nnz_from_rowptr ? g->rowptr + n : nullptr, d_extra, d_roots,
                                        reinterpret_cast<const int32_t*>(d_pmin), flags, report);
        PF_HIP(hipGetLastError());
        slice_bytes = sizeof(int64_t) * (size_t)(g->n_slices + 1);
        PF_TRY(pf_pinned_scratch(g->ctx, slice_bytes + sizeof(int32_t) * PF_REPORT_INTS, &pin, sid));
        h_report = reinterpret_cast<int32_t*>(static_cast<unsigned char*>(pin) + slice_bytes);
        PF_HIP(pfl::memcpy_async(st, pin, g->slice_ptr, slice_bytes, hipMemcpyDeviceToHost));
        PF_HIP(pfl::memcpy_async(st, h_report, report, sizeof(int32_t) * PF_REPORT_INTS, hipMemcpyDeviceToHost));
        return PF_OK;
    }

    int end() {
        const int64_t n = g->n;
        PF_HIP(pfl::sync(st));
        if (h_report[7] != 0) {  // vertices piled into one cell of the Morton grid: once more, with the general sort
            if (getenv("PF_DEBUG_WINDOWS")) fprintf(stderr, "pyfocusr_hip: renumbering by counting gave up, repeating with the general sort\n");
            PF_TRY(queue_order(true));
            PF_HIP(pfl::sync(st));
        }
        g->h_slice_ptr.resize((size_t)g->n_slices + 1);  // the resident Chebyshev kernel sizes its LDS from this
        memcpy(g->h_slice_ptr.data(), pin, slice_bytes);
        int32_t h_stats[6], h_roots[PF_ROOTS_AHEAD];
        for (int i = 0; i < 5; ++i) h_stats[i] = h_report[i];
        int32_t differing = h_report[5];
        const int32_t nnz32 = h_report[6];
        if (d_extra)
            for (int i = 0; i < 8; ++i) h_extra[i] = h_report[8 + i];
        for (int i = 0; i < PF_ROOTS_AHEAD; ++i) h_roots[i] = h_report[16 + i];
        if (d_pmin && h_pmin) {
            unsigned long long bits = 0;
            memcpy(&bits, h_report + 16 + PF_ROOTS_AHEAD, sizeof(bits));
            memcpy(h_pmin, &bits, sizeof(bits));
        }
        g->sell_entries = g->h_slice_ptr[(size_t)g->n_slices];
        if (nnz_from_rowptr) g->nnz_w = nnz32;
        if (d_extra) {
            extra_hit = false;
            for (int i = 0; i < 8; ++i) extra_hit = extra_hit || h_extra[i] != 0;
            if (extra_hit) return PF_OK;
        }
        int32_t n_roots = h_stats[4];
        if (differing) {  // (rare: more than PF_CC_FIRST rounds) finish the labelling, collect the roots again
            for (;;) {
                PF_CHECK(round + 3 <= PF_CC_ROUNDS, PF_E_HIP, "pf_graph_build: component labelling did not converge");
                for (int b = 0; b < 3; ++b, ++round) {
                    pfl::launch<k_label_round>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, g->rowptr, g->col, n, g->label, round_flags + round, nullptr, g->morder);
                    PF_HIP(hipGetLastError());
                }
                pfl::launch<k_label_compress>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, g->label, n, nullptr);
                PF_HIP(hipGetLastError());
                PF_HIP(pfl::memcpy_async(st, &differing, round_flags + round - 1, sizeof(int32_t), hipMemcpyDeviceToHost));
                PF_HIP(pfl::sync(st));
                if (!differing) break;
            }
            PF_HIP(pfl::memset_words(st, stats + 4, 0, sizeof(int32_t)));
            pfl::launch<k_collect_roots>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, g->label, g->rowptr, n, d_roots, stats + 4);
            PF_HIP(hipGetLastError());
            PF_HIP(pfl::memcpy_async(st, &n_roots, stats + 4, sizeof(int32_t), hipMemcpyDeviceToHost));
            PF_HIP(pfl::memcpy_async(st, h_roots, d_roots, sizeof(h_roots), hipMemcpyDeviceToHost));
            PF_HIP(pfl::sync(st));
        }
        g->n_isolated = h_stats[0];
        g->max_degree = h_stats[1];
        g->is_symmetric = h_stats[2] ? 0 : 1;
        g->n_oneway = h_stats[2];
        PF_CHECK(n_roots <= PF_MAX_ROOTS, PF_E_ARG, "pf_graph_build: %d connected components exceed the supported %d",
                 n_roots, PF_MAX_ROOTS);
        g->n_components = n_roots;
        g->roots.resize(n_roots);
        if (n_roots > PF_ROOTS_AHEAD) {
            PF_HIP(pfl::memcpy_async(st, g->roots.data(), d_roots, sizeof(int32_t) * n_roots, hipMemcpyDeviceToHost));
            PF_HIP(pfl::sync(st));
        } else {
            for (int32_t i = 0; i < n_roots; ++i) g->roots[(size_t)i] = h_roots[i];
        }
        std::sort(g->roots.begin(), g->roots.end());
        PF_TRY(dev_alloc(st, &g->scol, g->sell_entries));
        PF_TRY(dev_alloc(st, &g->sval_rw, g->sell_entries));
        if (g->is_symmetric) PF_TRY(dev_alloc(st, &g->sval_sym, g->sell_entries));
        pfl::launch<k_fill_sell_entries>(dim3((unsigned)g->n_slices), dim3(PF_BLOCK), 0, st, g->rowptr, g->col, g->w, g->deg, g->g, g->sg, g->perm_m, g->iperm_m, n,
                                                                        g->slice_ptr, g->scol, g->sval_rw, g->sval_sym, g->diag);
        PF_HIP(hipGetLastError());
        return PF_OK;
    }
};

int finish_graph(pf_graph* g, const double* d_pts, bool numeric_symmetry, const int32_t* d_extra = nullptr, int32_t* h_extra = nullptr,
                 bool* extra_hit = nullptr, bool nnz_from_rowptr = false, const unsigned long long* d_pmin = nullptr,
                 double* h_pmin = nullptr) {
    FinishJob f;
    f.g = g, f.d_pts = d_pts, f.numeric_symmetry = numeric_symmetry, f.d_extra = d_extra, f.h_extra = h_extra;
    f.nnz_from_rowptr = nnz_from_rowptr, f.d_pmin = d_pmin, f.h_pmin = h_pmin;
    PF_TRY(f.begin());
    PF_TRY(f.end());
    if (extra_hit) *extra_hit = f.extra_hit;
    return PF_OK;
}

}  // namespace

struct pf_mesh {
    pf_ctx* ctx = nullptr;
    double* pts = nullptr;    // [n][3]
    int32_t* faces = nullptr; // [n_faces][vpf]
    int64_t n = 0, n_faces = 0;
    int32_t vpf = 0;
};

namespace {

// The assembly of one mesh in two halves around its one synchronisation (see FinishJob)
struct MeshBuild {
    pf_mesh* mesh = nullptr;
    pf_graph* g = nullptr;
    int sid = 0;
    hipStream_t st = nullptr;
    std::vector<void*> tmp;
    bool ok = false;
    FinishJob fin;
    int32_t h_flags[8] = {0};
    double h_pmin = 0.0;
    unsigned long long* pmin = nullptr;
    int64_t n_edges = 0;

    void release() {  // the temporaries, under the stream that used them
        for (void* p : tmp) pf_free(st, p);
        tmp.clear();
        fin.release();
    }
    ~MeshBuild() {
        release();
        if (!ok && g) {
            g->build_stream = nullptr;
            pf_graph_free(g);
        }
    }

    template <typename T>
    int scratch(T** p, int64_t count) {
        int r = dev_alloc(st, p, count);
        if (r == PF_OK) tmp.push_back((void*)*p);
        return r;
    }

    // The first half in four phases (a pair's two builds are queued phase by phase from ONE thread, see
    // pf_graph_build_device2); begin() runs them in a row.
    static constexpr int N_PHASES = 4;
    int32_t *b_cnt = nullptr, *b_start = nullptr, *b_rank = nullptr, *b_rcol = nullptr, *b_ucnt = nullptr, *b_flags = nullptr;
    double* b_rw = nullptr;
    double* pts_m = nullptr;     // the points and ...
    int32_t* faces_m = nullptr;  // ... faces renumbered by the Morton rank of the points (m-space)
    bool fork_ok = false;        // set by the pair build: independent chains of the build may run on the ctx's second stream
    bool robust = false;         // the Morton order by the general sort (second attempt: vertices piled into one cell)
    bool needs_robust = false;   // end(): the counting sort gave up - build once more with robust = true

    int begin(pf_mesh* m, int stream_id) {
        PF_TRY(prepare(m, stream_id));
        for (int k = 0; k < N_PHASES; ++k) PF_TRY(phase(k));
        return PF_OK;
    }

    // one_stream: the second mesh of a pair whose launches are shared (pf_launch.h) - everything on the ctx stream
    int prepare(pf_mesh* m, int stream_id, bool one_stream = false) {
        mesh = m;
        sid = stream_id;
        pf_ctx* ctx = mesh->ctx;
        st = (sid && !one_stream) ? ctx->stream_b : ctx->stream;
        const int64_t n = mesh->n, n_faces = mesh->n_faces;
        const int32_t vpf = mesh->vpf;
        n_edges = n_faces * vpf;
        g = new pf_graph();
        g->ctx = ctx;
        g->build_stream = st;
        g->n = n;
        g->n_faces = n_faces;
        g->vpf = vpf;
        g->n_pad = (n + 4095) / 4096 * 4096;  // whole blocks of up to 512 rows, a multiple of 8 of them (XCD remap)
        g->win_rows = pf_window_rows(g->n_pad);
        g->n_slices = g->n_pad / PF_WAVE;
        g->n_chunks = (g->n_pad + PF_DOT_CHUNK - 1) / PF_DOT_CHUNK;
        return PF_OK;
    }

    int phase(int k) {
        pf_ctx* ctx = mesh->ctx;
        const int64_t n = mesh->n, n_faces = mesh->n_faces;
        const int32_t vpf = mesh->vpf;
        // from phase 0 on everything reads the mesh in m-space
        const double* d_pts = pts_m;
        const int32_t* d_faces = faces_m;
        const bool face_bound = vpf == 3 && n_faces > 0;
        if (k == 0) {  // storage, the Morton order and the renumbered mesh, the edge list counted and scattered
            // (the counters that start from zero share one block, and so do deg / g / sg: two memsets instead of seven launches)
            int32_t* zeroed = nullptr;
            const int64_t zstride = (n + 1 + 7) & ~(int64_t)7;
            PF_TRY(scratch(&zeroed, 2 * zstride + 8));
            b_cnt = zeroed, b_ucnt = zeroed + zstride, b_flags = zeroed + 2 * zstride;
            PF_TRY(scratch(&b_rank, n_edges));
            PF_TRY(scratch(&b_start, n + 1));
            PF_TRY(scratch(&b_rcol, n_edges));
            PF_TRY(scratch(&b_rw, n_edges));
            PF_TRY(dev_alloc(st, &g->rowptr, n + 1));
            PF_TRY(dev_alloc(st, &g->deg, 3 * g->n_pad));  // deg, g, sg: one allocation (g->g and g->sg point into it)
            g->g = g->deg + g->n_pad;
            g->sg = g->deg + 2 * g->n_pad;
            g->deg_block = true;
            PF_TRY(dev_alloc(st, &g->diag, g->n_pad));
            PF_TRY(dev_alloc(st, &g->label, g->n_pad));
            PF_TRY(dev_alloc(st, &g->perm, g->n_pad));
            PF_TRY(dev_alloc(st, &g->iperm, g->n_pad));
            PF_TRY(dev_alloc(st, &g->smooth, g->n_pad));
            PF_TRY(dev_alloc(st, &g->slice_ptr, g->n_slices + 1));

            PF_TRY(dev_alloc(st, &g->perm_m, g->n_pad));
            PF_TRY(dev_alloc(st, &g->iperm_m, g->n_pad));
            PF_HIP(pfl::memset_words(st, zeroed, 0, sizeof(int32_t) * (size_t)(2 * zstride + 8)));
            PF_HIP(pfl::memset_words(st, g->deg, 0, sizeof(double) * 3 * g->n_pad));
            if (sid == 0) PF_HIP(pfl::event_record(st, ctx->ev0));
            // m-space first: the Morton rank of every point (positions only), points and faces renumbered by it.  Every
            // gather of the build from here on - edge ends, reverse edges, neighbours' degrees, window flags - lands in
            // lines that the neighbouring threads share, whatever order the caller's vertices came in (round 3: ~35 x
            // the algorithmic bytes in counter traffic on the shuffled synthetic meshes).  b_flags[1]: the counting
            // sort's overflow flag (piled vertices), read back with everything else.
            PF_TRY(pf_morton_order(g, mesh->pts, robust ? nullptr : b_flags + 1));
            PF_TRY(scratch(&pts_m, 3 * n));
            PF_TRY(scratch(&faces_m, n_edges));
            d_pts = pts_m, d_faces = faces_m;
            if (n) {
                pfl::launch<k_renumber_points>(dim3(nblk(3 * n)), dim3(PF_BLOCK), 0, st, mesh->pts, g->morder, n, pts_m);
                PF_HIP(hipGetLastError());
            }
            if (n_edges) {
                pfl::launch<k_renumber_faces>(dim3(nblk(n_edges)), dim3(PF_BLOCK), 0, st, mesh->faces, g->mrank, n_edges, n, faces_m);
                PF_HIP(hipGetLastError());
            }

            if (face_bound) PF_TRY(scratch(&pmin, 1));
            if (n_edges) {
                pfl::launch<k_count_edges>(dim3(nblk(n_edges)), dim3(PF_BLOCK), 0, st, d_faces, n_edges, vpf, n, b_cnt, b_rank, b_flags, reinterpret_cast<double*>(pmin));
                PF_HIP(hipGetLastError());
            }
            // No read-back on the way: faces the counting kernel flags (index out of range, repeated vertex) are skipped by
            // the kernels behind it, col / w are sized by their upper bound (one entry per face edge; duplicates only shrink
            // it), and the flags, the entry count and everything the second half needs come back in ONE synchronisation
            // (each one costs ~30 us of idle device: 8 per mesh at first, 2 now).
            PF_TRY(pf_exclusive_scan_i32(st, b_cnt, b_start, n + 1));
            if (n_edges) {
                pfl::launch<k_scatter_edges>(dim3(nblk(n_edges)), dim3(PF_BLOCK), 0, st, d_faces, d_pts, n_edges, vpf, n, b_start, b_rank, b_rcol, b_rw, b_flags);
                PF_HIP(hipGetLastError());
            }
            return PF_OK;
        }
        if (k == 1) {  // CSR(W), degrees, the face bound
            pfl::launch<k_sort_unique_rows>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, b_start, n, b_rcol, b_rw, b_ucnt, g->morder);
            PF_HIP(hipGetLastError());
            PF_TRY(pf_exclusive_scan_i32(st, b_ucnt, g->rowptr, n + 1));
            PF_TRY(dev_alloc(st, &g->col, n_edges));
            PF_TRY(dev_alloc(st, &g->w, n_edges));
            pfl::launch<k_compact_rows>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, b_start, g->rowptr, n, b_rcol, b_rw, g->col, g->w, g->deg, g->g, g->sg);
            PF_HIP(hipGetLastError());
            // the face-by-face bound of the spectrum (k_face_bound); it holds if every undirected edge lies in exactly two
            // triangles: W symmetric and no directed edge listed twice - both known after the read-back
            // Fork (pair builds: their launches are recorded, so blocks released below go back only after the join has been
            // queued): the face bound and, in the last phase, the order inside windows have nothing to do with the row
            // statistics and the 12 labelling rounds - they run beside them on the ctx's second stream.
            if (fork_ok && pfl::tl_rec && pf_stream_b(ctx) && ctx->fork_ev) {
                hipStream_t side = ctx->stream_b;
                hipEvent_t ev = ctx->fork_ev;
                pfl::call(st, [=](hipStream_t s) {
                    (void)hipEventRecord(ev, s);
                    (void)hipStreamWaitEvent(side, ev, 0);
                });
                g->side_stream = side;
            }
            if (face_bound) {
                pfl::launch<k_face_bound>(dim3(nblk(n_faces)), dim3(PF_BLOCK), 0, g->side_stream ? g->side_stream : st, d_faces, d_pts, n_faces, n, pmin);
                PF_HIP(hipGetLastError());
            }
            fin.g = g, fin.d_pts = d_pts, fin.numeric_symmetry = false, fin.d_extra = b_flags, fin.h_extra = h_flags;
            fin.nnz_from_rowptr = true, fin.d_pmin = pmin, fin.h_pmin = &h_pmin, fin.sid = sid;
            return PF_OK;
        }
        if (k == 2) return fin.begin_stats_and_labels();
        return fin.queue_order(false);
    }

    int end() {
        const int64_t n = mesh->n;
        PF_TRY(fin.end());
        if (h_flags[1] != 0 && !(h_flags[0] & 7)) {  // vertices piled into one cell of the Morton grid: the whole build once more,
            needs_robust = true;                      // with the general sort (the order the atomics left is not reproducible)
            return PF_OK;
        }
        PF_CHECK(!(h_flags[0] & 1), PF_E_ARG, "pf_graph_build: face index out of range [0,%lld)", (long long)n);
        PF_CHECK(!(h_flags[0] & 2), PF_E_DEGENERATE, "pf_graph_build: a face repeats a vertex on one edge");
        PF_CHECK(!(h_flags[0] & 4), PF_E_DEGENERATE,
                 "pf_graph_build: an edge has zero length or non-finite coordinates (the reference would store an "
                 "infinite weight, graph.py:177-178)");
        PF_CHECK(!fin.extra_hit, PF_E_HIP, "pf_graph_build: unexpected assembly flag");
        g->spectral_bound = 2.0;
        if (pmin && g->is_symmetric && g->nnz_w == n_edges && h_pmin > 0.0 && h_pmin <= 0.25) {
            const double disc = 1.0 - 4.0 * h_pmin;
            const double b = (1.0 + (1.0 + sqrt(disc > 0.0 ? disc : 0.0)) / 2.0) * (1.0 + 1e-12);
            g->spectral_bound = b < 2.0 ? b : 2.0;
        }
        PF_TRY(dev_alloc(st, &g->pts, 3 * n));  // kept for pf_point_rows (the mesh object may go away before the graph)
        PF_HIP(pfl::memcpy_async(st, g->pts, mesh->pts, sizeof(double) * 3 * n, hipMemcpyDeviceToDevice));
        return PF_OK;
    }
};

}  // namespace

extern "C" {

void pf_graph_free(pf_graph* g) {
    if (!g) return;
    if (!g->ctx) {
        delete g;
        return;
    }
    hipSetDevice(g->ctx->device);
    hipStream_t st = g->ctx->stream;
    (void)pf_finalize_vectors_end(g);  // a download still in flight reads final_vecs
    {
        std::lock_guard<std::mutex> lk(g->ctx->deferred_mutex);
        auto& dq = g->ctx->deferred;
        dq.erase(std::remove(dq.begin(), dq.end(), g), dq.end());
    }
    if (g->final_stats) g->ctx->pinned_pool.emplace_back(g->final_stats_cap, reinterpret_cast<double*>(g->final_stats));
    if (g->final_ready) g->ctx->event_pool.push_back(g->final_ready);
    if (g->final_done) g->ctx->event_pool.push_back(g->final_done);
    pf_free(st, g->persist_ring);
    pf_free(st, g->persist_ring2);
    pf_free(st, g->final_vecs);
    pf_free(st, g->pts);
    pf_window_slots_free(g);
    pf_free(st, g->rowptr);
    pf_free(st, g->col);
    pf_free(st, g->w);
    pf_free(st, g->deg);
    if (!g->deg_block) {
        pf_free(st, g->g);
        pf_free(st, g->sg);
    }
    pf_free(st, g->label);
    if (g->perm_m != g->perm) pf_free(st, g->perm_m);
    if (g->iperm_m != g->iperm) pf_free(st, g->iperm_m);
    pf_free(st, g->morder);
    pf_free(st, g->mrank);
    pf_free(st, g->order_bbox);
    pf_free(st, g->perm);
    pf_free(st, g->iperm);
    pf_free(st, g->smooth);
    pf_free(st, g->stage);
    pf_free(st, g->slice_ptr);
    pf_free(st, g->scol);
    pf_free(st, g->sval_rw);
    pf_free(st, g->sval_sym);
    pf_free(st, g->diag);
    pf_free(st, g->mf_col);
    pf_free(st, g->mf_val);
    pf_free(st, g->ws);
    pf_free(st, g->partials);
    pf_free(st, g->coef);
    pf_free(st, g->orth_counter);
    if (g->orth_pending >= 0 || g->small_pending > 0 || g->px_state == -2) pfl::sync(st);  // a step or a result nobody collected still writes its pinned buffer
    if (g->orth_host) g->ctx->pinned_pool.emplace_back(g->orth_host_cap, g->orth_host);
    if (g->orth_ev) g->ctx->event_pool.push_back(g->orth_ev);
    if (g->px_host) g->ctx->pinned_pool.emplace_back(g->px_host_cap, g->px_host);
    if (g->px_ev) g->ctx->event_pool.push_back(g->px_ev);
    if (g->small_host) g->ctx->pinned_pool.emplace_back(g->small_host_cap, g->small_host);
    if (g->small_ev) g->ctx->event_pool.push_back(g->small_ev);
    delete g;
}


int pf_mesh_upload(pf_ctx* ctx, const double* pts, int64_t n, const int32_t* faces, int64_t n_faces, int32_t vpf,
                   pf_mesh** out) {
    PF_CHECK(ctx && pts && out && (faces || n_faces == 0), PF_E_ARG, "pf_mesh_upload: NULL argument");
    PF_CHECK(n > 0 && n < (int64_t)1 << 31, PF_E_ARG, "pf_mesh_upload: n = %lld out of range", (long long)n);
    PF_CHECK(n_faces >= 0 && vpf >= 2 && n_faces * vpf < (int64_t)1 << 31, PF_E_ARG,
             "pf_mesh_upload: faces %lld x %d out of range", (long long)n_faces, vpf);
    *out = nullptr;
    PF_HIP(hipSetDevice(ctx->device));
    pf_mesh* m = new pf_mesh();
    m->ctx = ctx;
    m->n = n;
    m->n_faces = n_faces;
    m->vpf = vpf;
    hipStream_t st = ctx->stream;
    int r = dev_alloc(st, &m->pts, 3 * n);
    if (r == PF_OK) r = dev_alloc(st, &m->faces, n_faces * vpf);
    hipError_t e = hipSuccess;
    if (r == PF_OK) e = pfl::memcpy_async(ctx->stream, m->pts, pts, sizeof(double) * 3 * n, hipMemcpyHostToDevice);
    if (r == PF_OK && e == hipSuccess && n_faces)
        e = pfl::memcpy_async(ctx->stream, m->faces, faces, sizeof(int32_t) * n_faces * vpf, hipMemcpyHostToDevice);
    if (r == PF_OK && e == hipSuccess) e = pfl::sync(ctx->stream);
    if (r != PF_OK || e != hipSuccess) {
        if (e != hipSuccess) pf_set_error("pf_mesh_upload: %s", hipGetErrorString(e));
        pf_mesh_free(m);
        return r != PF_OK ? r : PF_E_HIP;
    }
    *out = m;
    return PF_OK;
}

void pf_mesh_free(pf_mesh* m) {
    if (!m) return;
    pf_free(m->ctx->stream, m->pts);
    pf_free(m->ctx->stream, m->faces);
    delete m;
}

int pf_graph_build(pf_ctx* ctx, const double* pts, int64_t n, const int32_t* faces, int64_t n_faces,
                   int32_t vpf, pf_graph** out) {
    PF_CHECK(out != nullptr, PF_E_ARG, "pf_graph_build: out is NULL");
    *out = nullptr;
    pf_mesh* m = nullptr;
    PF_TRY(pf_mesh_upload(ctx, pts, n, faces, n_faces, vpf, &m));
    int r = pf_graph_build_device(m, out);
    pf_mesh_free(m);
    return r;
}

int pf_graph_build_device(pf_mesh* mesh, pf_graph** out) {
    PF_CHECK(mesh && out, PF_E_ARG, "pf_graph_build_device: NULL argument");
    pf_ctx* ctx = mesh->ctx;
    *out = nullptr;
    PF_HIP(hipSetDevice(ctx->device));
    MeshBuild job;
    PF_TRY(job.begin(mesh, 0));
    PF_TRY(job.end());
    if (job.needs_robust) {
        if (getenv("PF_DEBUG_WINDOWS")) fprintf(stderr, "pyfocusr_hip: renumbering by counting gave up, building again with the general sort\n");
        MeshBuild again;
        again.robust = true;
        PF_TRY(again.begin(mesh, 0));
        PF_TRY(again.end());
        PF_HIP(pfl::sync(ctx->stream));
        again.g->build_stream = nullptr;
        again.ok = true;
        *out = again.g;
        return PF_OK;
    }
    PF_HIP(pfl::event_record(ctx->stream, ctx->ev1));  // (no wait for the SELL fill: see pf_graph_build_device2)
    ctx->build_pending = true;
    if (pf_persist_enabled()) PF_TRY(pf_window_slots_begin(job.g));
    job.g->build_stream = nullptr;
    job.ok = true;
    *out = job.g;
    return PF_OK;
}

// The two meshes of a pair (target and source of focusr.py:134-170) assembled SIDE BY SIDE: mesh a on the ctx stream, mesh
// b on the ctx's second stream, their halves interleaved (begin a, begin b, end a, end b).  An assembly is ~75 small
// kernels (4-80 us each, most of them launch latency and one wave of blocks); two of them overlap almost completely.
// Results are those of two pf_graph_build_device calls, bit for bit.  On return both graphs live on the ctx stream.
int pf_graph_build_device2(pf_mesh* mesh_a, pf_mesh* mesh_b, pf_graph** out_a, pf_graph** out_b) {
    PF_CHECK(mesh_a && mesh_b && out_a && out_b, PF_E_ARG, "pf_graph_build_device2: NULL argument");
    PF_CHECK(mesh_a->ctx == mesh_b->ctx, PF_E_ARG, "pf_graph_build_device2: the two meshes must belong to one ctx");
    pf_ctx* ctx = mesh_a->ctx;
    *out_a = *out_b = nullptr;
    PF_HIP(hipSetDevice(ctx->device));
    if (!pf_stream_b(ctx)) {  // no second stream: one after the other
        PF_TRY(pf_graph_build_device(mesh_a, out_a));
        const int r = pf_graph_build_device(mesh_b, out_b);
        if (r != PF_OK) {
            pf_graph_free(*out_a);
            *out_a = nullptr;
        }
        return r;
    }
    int rc = PF_OK;
    bool redo = false;
    static const bool two_streams = getenv("PF_PAIR_BUILD_STREAMS") != nullptr;  // (the form of rounds 2-4, for comparisons)
    if (!two_streams) {
        // SHARED LAUNCHES (pf_launch.h): the host code of every phase runs once per mesh, each into its own recorder, and
        // what both meshes ask for goes out as one launch with the mesh in blockIdx.z - half the dispatches of the two-stream
        // form, half the host time, one stream.
        MeshBuild a, b;
        pfl::Recorder ra, rb;
        struct Unhook {
            ~Unhook() { pfl::tl_rec = nullptr; }
        } unhook;
        hipStream_t st = ctx->stream;
        rc = a.prepare(mesh_a, 0);
        if (rc == PF_OK) rc = b.prepare(mesh_b, 1, true);
        a.fork_ok = b.fork_ok = getenv("PF_BUILD_FORK") == nullptr || atoi(getenv("PF_BUILD_FORK")) != 0;
        for (int k = 0; k < MeshBuild::N_PHASES && rc == PF_OK; ++k) {
            pfl::tl_rec = &ra;
            rc = a.phase(k);
            pfl::tl_rec = &rb;
            if (rc == PF_OK) rc = b.phase(k);
            pfl::tl_rec = nullptr;
            pfl::flush(ra, &rb);
        }
        // second halves: each waits for the read-back (one wait serves both: one stream), decides, and records its SELL fill
        if (rc == PF_OK) {
            pfl::tl_rec = &ra;
            rc = a.end();
            pfl::tl_rec = &rb;
            if (rc == PF_OK) rc = b.end();
            if (rc == PF_OK && !(a.needs_robust || b.needs_robust) && pf_persist_enabled()) {
                // the window structures of the resident filter kernel, behind the fills (collected by the first application)
                pfl::tl_rec = &ra;
                rc = pf_window_slots_begin(a.g);
                pfl::tl_rec = &rb;
                if (rc == PF_OK) rc = pf_window_slots_begin(b.g);
            }
        }
        pfl::tl_rec = nullptr;
        pfl::flush(ra, &rb);  // (also after an error: the recorded launches are harmless, the held-back frees are due)
        a.release();
        b.release();
        if (rc == PF_OK) {
            const hipError_t e1 = hipEventRecord(ctx->ev1, st);  // (no wait for the fills: pf_timing_get reads the events)
            if (e1 == hipSuccess) ctx->build_pending = true;
            else rc = PF_E_HIP, pf_set_error("pf_graph_build_device2: %s", hipGetErrorString(e1));
        }
        if (rc == PF_OK && (a.needs_robust || b.needs_robust)) {
            redo = true;  // (a pile of vertices in one Morton cell: the two builds once more, one after the other; below)
        } else if (rc == PF_OK) {
            a.g->build_stream = b.g->build_stream = nullptr;
            a.ok = b.ok = true;
            *out_a = a.g;
            *out_b = b.g;
        } else {
            (void)hipStreamSynchronize(st);
        }
    } else {
        MeshBuild a, b;
        rc = pf_streams_join(ctx, 1);  // the second stream sees the uploads and may reuse what the first has released
        if (rc == PF_OK) {
            // The first halves are ~55 launches each, ~4 us of host time apiece.  Queued one build after the other, the
            // second mesh would start when the first one's kernels are half through; queued from two threads (rounds 2-3)
            // they start together - until the host's scheduler leaves the second thread waiting for a core for 3-9 ms,
            // which it did in one step of six in about one process of twelve on this pool's loaded hosts (PF_DEBUG_BUILD:
            // "worker began 3045 us after the call", "ended 9344").  So: ONE thread, the two builds phase by phase (four
            // phases of ~15 launches): the second stream is never more than one phase behind, 0.45 ms of launching against
            // 0.97 ms of device time.
            static const bool dbg = getenv("PF_DEBUG_BUILD") != nullptr;
            using clk = std::chrono::steady_clock;
            const clk::time_point t0 = clk::now();
            rc = a.prepare(mesh_a, 0);
            if (rc == PF_OK) rc = b.prepare(mesh_b, 1);
            for (int k = 0; k < MeshBuild::N_PHASES && rc == PF_OK; ++k) {
                rc = a.phase(k);
                if (rc == PF_OK) rc = b.phase(k);
            }
            if (dbg) fprintf(stderr, "pf_build2: first halves queued in %.0f us (one thread, phase by phase)\n",
                             std::chrono::duration<double, std::micro>(clk::now() - t0).count());
        }
        const auto tq0 = std::chrono::steady_clock::now();
        if (rc == PF_OK) rc = a.end();
        const auto tq1 = std::chrono::steady_clock::now();
        if (rc == PF_OK) rc = b.end();
        const auto tq2 = std::chrono::steady_clock::now();
        if (getenv("PF_DEBUG_BUILD"))
            fprintf(stderr, "pf_build2: second halves (wait for the read-back, SELL fill queued): first mesh %.0f us, second %.0f us\n",
                    std::chrono::duration<double, std::micro>(tq1 - tq0).count(), std::chrono::duration<double, std::micro>(tq2 - tq1).count());
        // the temporaries go back before the join, so that the first stream may have the second one's from now on (a
        // block is visible across streams only if it was released before the join)
        a.release();
        b.release();
        // whatever happened, the first stream waits for the second before anything else is queued on it (and the graphs
        // move to the first stream)
        const int rj = pf_streams_join(ctx, 0);
        if (rc == PF_OK) rc = rj;
        if (rc == PF_OK) {
            // The call does NOT wait for the SELL fills it has queued: everything that reads the graphs is ordered behind
            // them on the ctx stream, and the host fields came with the read-back.  The build's device time (pf_timing_get:
            // build_ms) is taken from the events when somebody asks.
            const hipError_t e1 = pfl::event_record(ctx->stream, ctx->ev1);
            if (e1 == hipSuccess) ctx->build_pending = true;
            else rc = PF_E_HIP, pf_set_error("pf_graph_build_device2: %s", hipGetErrorString(e1));
        }
        if (rc == PF_OK && !(a.needs_robust || b.needs_robust) && pf_persist_enabled()) {
            // the window structures of the resident filter kernel, queued behind the fills (collected by the first application)
            rc = pf_window_slots_begin(a.g);
            if (rc == PF_OK) rc = pf_window_slots_begin(b.g);
        }
        if (rc == PF_OK && (a.needs_robust || b.needs_robust)) {
            redo = true;  // (a pile of vertices in one Morton cell: the two builds once more, one after the other; below)
        } else if (rc == PF_OK) {
            a.g->build_stream = b.g->build_stream = nullptr;
            a.ok = b.ok = true;
            *out_a = a.g;
            *out_b = b.g;
        } else {
            (void)pfl::sync(ctx->stream_b);
            (void)pfl::sync(ctx->stream);
        }
    }
    if (redo) {
        (void)pfl::sync(ctx->stream_b);
        (void)pfl::sync(ctx->stream);
        PF_TRY(pf_graph_build_device(mesh_a, out_a));
        const int r = pf_graph_build_device(mesh_b, out_b);
        if (r != PF_OK) {
            pf_graph_free(*out_a);
            *out_a = nullptr;
        }
        return r;
    }
    return rc;
}

int pf_graph_from_matrix(pf_ctx* ctx, int64_t n, const int32_t* rowptr, const int32_t* colidx, const double* values,
                         pf_graph** out) {
    PF_CHECK(ctx && rowptr && out, PF_E_ARG, "pf_graph_from_matrix: NULL argument");
    PF_CHECK(n > 0 && n < (int64_t)1 << 31, PF_E_ARG, "pf_graph_from_matrix: n = %lld out of range", (long long)n);
    const int64_t nnz = rowptr[n];
    PF_CHECK(nnz >= 0 && nnz < (int64_t)1 << 31 && (nnz == 0 || (colidx && values)), PF_E_ARG,
             "pf_graph_from_matrix: bad nnz %lld", (long long)nnz);
    *out = nullptr;
    PF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    pf_graph* g = new pf_graph();
    g->ctx = ctx;
    g->unit_g = 1;
    g->n = n;
    g->n_pad = (n + 4095) / 4096 * 4096;
    g->win_rows = pf_window_rows(g->n_pad);
    g->n_slices = g->n_pad / PF_WAVE;
    g->n_chunks = (g->n_pad + PF_DOT_CHUNK - 1) / PF_DOT_CHUNK;
    struct Guard {
        pf_graph* g;
        std::vector<void*> tmp;
        bool ok = false;
        ~Guard() {
            for (void* p : tmp) pf_free(g->ctx->stream, p);
            if (!ok) pf_graph_free(g);
        }
    } guard{g};
    int32_t *rp = nullptr, *ci = nullptr, *cnt = nullptr, *flags = nullptr;
    double* va = nullptr;
    auto scratch = [&](auto** p, int64_t count) -> int {
        int r = dev_alloc(st, p, count);
        if (r == PF_OK) guard.tmp.push_back((void*)*p);
        return r;
    };
    PF_TRY(scratch(&rp, n + 1));
    PF_TRY(scratch(&ci, nnz));
    PF_TRY(scratch(&va, nnz));
    PF_TRY(scratch(&cnt, n + 1));
    PF_TRY(scratch(&flags, 8));
    PF_TRY(dev_alloc(st, &g->rowptr, n + 1));
    PF_TRY(dev_alloc(st, &g->deg, g->n_pad));
    PF_TRY(dev_alloc(st, &g->g, g->n_pad));
    PF_TRY(dev_alloc(st, &g->sg, g->n_pad));
    PF_TRY(dev_alloc(st, &g->diag, g->n_pad));
    PF_TRY(dev_alloc(st, &g->label, g->n_pad));
    PF_TRY(dev_alloc(st, &g->perm, g->n_pad));
    PF_TRY(dev_alloc(st, &g->iperm, g->n_pad));
    PF_TRY(dev_alloc(st, &g->smooth, g->n_pad));
    PF_TRY(dev_alloc(st, &g->slice_ptr, g->n_slices + 1));
    PF_HIP(pfl::memcpy_async(st, rp, rowptr, sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice));
    if (nnz) {
        PF_HIP(pfl::memcpy_async(st, ci, colidx, sizeof(int32_t) * nnz, hipMemcpyHostToDevice));
        PF_HIP(pfl::memcpy_async(st, va, values, sizeof(double) * nnz, hipMemcpyHostToDevice));
    }
    PF_HIP(pfl::memset_words(st, cnt, 0, sizeof(int32_t) * (n + 1)));
    PF_HIP(pfl::memset_words(st, flags, 0, sizeof(int32_t) * 8));
    PF_HIP(pfl::memset_words(st, g->deg, 0, sizeof(double) * g->n_pad));
    PF_HIP(pfl::memset_words(st, g->g, 0, sizeof(double) * g->n_pad));
    PF_HIP(pfl::memset_words(st, g->sg, 0, sizeof(double) * g->n_pad));
    PF_HIP(pfl::event_record(st, ctx->ev0));
    pfl::launch<k_csr_count_offdiag>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, rp, ci, n, cnt, flags);
    PF_HIP(hipGetLastError());
    PF_TRY(pf_exclusive_scan_i32(st, cnt, g->rowptr, n + 1));
    int32_t h_flag = 0, nnz32 = 0;
    PF_HIP(pfl::memcpy_async(st, &h_flag, flags, sizeof(int32_t), hipMemcpyDeviceToHost));
    PF_HIP(pfl::memcpy_async(st, &nnz32, g->rowptr + n, sizeof(int32_t), hipMemcpyDeviceToHost));
    PF_HIP(pfl::sync(st));
    PF_CHECK(!h_flag, PF_E_ARG, "pf_graph_from_matrix: column indices must be in range, sorted and unique within each row");
    g->nnz_w = nnz32;
    PF_TRY(dev_alloc(st, &g->col, g->nnz_w));
    PF_TRY(dev_alloc(st, &g->w, g->nnz_w));
    pfl::launch<k_csr_split>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, rp, ci, va, g->rowptr, n, g->col, g->w, g->deg, g->g, g->sg);
    PF_HIP(hipGetLastError());
    PF_TRY(finish_graph(g, nullptr, true));
    PF_HIP(pfl::event_record(st, ctx->ev1));
    PF_HIP(pfl::sync(st));
    float ms = 0.f;
    PF_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->build_ms = ms;
    guard.ok = true;
    *out = g;
    return PF_OK;
}

int pf_graph_get_info(pf_graph* g, pf_graph_info* o) {
    PF_CHECK(g && o, PF_E_ARG, "pf_graph_get_info: NULL argument");
    o->n = g->n;
    o->n_faces = g->n_faces;
    o->nnz_w = g->nnz_w;
    o->nnz_l = g->nnz_w + (g->n - g->n_isolated);
    o->spectral_bound = g->spectral_bound;
    o->is_symmetric = g->is_symmetric;
    o->n_isolated = g->n_isolated;
    o->n_components = g->n_components;
    o->max_degree = g->max_degree;
    o->sell_entries = g->sell_entries;
    o->n_pad = g->n_pad;
    o->n_oneway = g->n_oneway;
    return PF_OK;
}

int pf_graph_download(pf_graph* g, int32_t* rowptr, int32_t* colidx, double* w, double* l_offdiag, double* deg,
                      double* l_diag, int32_t* component_label) {
    PF_CHECK(g, PF_E_ARG, "pf_graph_download: graph is NULL");
    PF_HIP(hipSetDevice(g->ctx->device));
    hipStream_t st = g->ctx->stream;
    if (g->morder) {  // assembled in m-space: the caller's order is made here
        const int64_t n = g->n, nnz = g->nnz_w;
        int32_t *len = nullptr, *rp = nullptr, *co = nullptr, *lab = nullptr, *small = nullptr;
        double *wo = nullptr, *lo = nullptr, *dg = nullptr, *ld = nullptr;
        int rc = PF_OK;
        auto bad = [&](hipError_t e) {
            if (e != hipSuccess && rc == PF_OK) {
                pf_set_error("pf_graph_download: %s", hipGetErrorString(e));
                rc = PF_E_HIP;
            }
            return e != hipSuccess;
        };
        do {
            if (bad(pf_malloc(st, (void**)&len, sizeof(int32_t) * (size_t)(n + 1))) || bad(pf_malloc(st, (void**)&rp, sizeof(int32_t) * (size_t)(n + 1)))) break;
            pfl::launch<k_len_original>(dim3(nblk(n + 1)), dim3(PF_BLOCK), 0, st, g->rowptr, g->mrank, n, len);
            if (bad(hipGetLastError())) break;
            if (pf_exclusive_scan_i32(st, len, rp, n + 1) != PF_OK) {
                rc = PF_E_HIP;
                break;
            }
            const size_t ne = (size_t)std::max<int64_t>(nnz, 1);
            if (colidx && bad(pf_malloc(st, (void**)&co, sizeof(int32_t) * ne))) break;
            if (w && bad(pf_malloc(st, (void**)&wo, sizeof(double) * ne))) break;
            if (l_offdiag && bad(pf_malloc(st, (void**)&lo, sizeof(double) * ne))) break;
            if (deg && bad(pf_malloc(st, (void**)&dg, sizeof(double) * (size_t)n))) break;
            if (l_diag && bad(pf_malloc(st, (void**)&ld, sizeof(double) * (size_t)n))) break;
            pfl::launch<k_rows_original>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, g->rowptr, g->col, g->w, g->g, g->deg, g->mrank, g->morder, rp, n, co, wo, lo, dg, ld);
            if (bad(hipGetLastError())) break;
            if (component_label) {
                if (bad(pf_malloc(st, (void**)&lab, sizeof(int32_t) * (size_t)n)) || bad(pf_malloc(st, (void**)&small, sizeof(int32_t) * (size_t)n))) break;
                if (bad(pfl::memset_words(st, small, 0x7f, sizeof(int32_t) * (size_t)n))) break;
                pfl::launch<k_label_min_original>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, g->label, g->morder, n, small);
                pfl::launch<k_label_original>(dim3(nblk(n)), dim3(PF_BLOCK), 0, st, g->label, g->mrank, small, n, lab);
                if (bad(hipGetLastError())) break;
                if (bad(pfl::memcpy_async(st, component_label, lab, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost))) break;
            }
            if (rowptr && bad(pfl::memcpy_async(st, rowptr, rp, sizeof(int32_t) * (size_t)(n + 1), hipMemcpyDeviceToHost))) break;
            if (colidx && nnz && bad(pfl::memcpy_async(st, colidx, co, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost))) break;
            if (w && nnz && bad(pfl::memcpy_async(st, w, wo, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost))) break;
            if (l_offdiag && nnz && bad(pfl::memcpy_async(st, l_offdiag, lo, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost))) break;
            if (deg && bad(pfl::memcpy_async(st, deg, dg, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost))) break;
            if (l_diag && bad(pfl::memcpy_async(st, l_diag, ld, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost))) break;
        } while (0);
        bad(pfl::sync(st));
        for (void* q : {(void*)len, (void*)rp, (void*)co, (void*)lab, (void*)small, (void*)wo, (void*)lo, (void*)dg, (void*)ld}) pf_free(st, q);
        return rc;
    }
    double* tmp = nullptr;
    if (rowptr) PF_HIP(pfl::memcpy_async(st, rowptr, g->rowptr, sizeof(int32_t) * (g->n + 1), hipMemcpyDeviceToHost));
    if (colidx && g->nnz_w) PF_HIP(pfl::memcpy_async(st, colidx, g->col, sizeof(int32_t) * g->nnz_w, hipMemcpyDeviceToHost));
    if (w && g->nnz_w) PF_HIP(pfl::memcpy_async(st, w, g->w, sizeof(double) * g->nnz_w, hipMemcpyDeviceToHost));
    if (deg) PF_HIP(pfl::memcpy_async(st, deg, g->deg, sizeof(double) * g->n, hipMemcpyDeviceToHost));
    double* tmp_diag = nullptr;
    if (l_diag) {  // g->diag is stored in solver order: rebuild g_i deg_i in mesh order
        PF_HIP(pf_malloc(st, (void**)&tmp_diag, sizeof(double) * g->n));
        pfl::launch<k_l_diag>(dim3(nblk(g->n)), dim3(PF_BLOCK), 0, st, g->deg, g->g, g->n, tmp_diag);
        hipError_t ed = hipGetLastError();
        if (ed == hipSuccess) ed = pfl::memcpy_async(st, l_diag, tmp_diag, sizeof(double) * g->n, hipMemcpyDeviceToHost);
        if (ed != hipSuccess) {
            pfl::sync(st);
            pf_free(st, tmp_diag);
            pf_set_error("pf_graph_download: %s", hipGetErrorString(ed));
            return PF_E_HIP;
        }
    }
    if (component_label) PF_HIP(pfl::memcpy_async(st, component_label, g->label, sizeof(int32_t) * g->n, hipMemcpyDeviceToHost));
    if (l_offdiag && g->nnz_w) {
        PF_HIP(pf_malloc(st, (void**)&tmp, sizeof(double) * g->nnz_w));
        pfl::launch<k_l_offdiag>(dim3(nblk(g->n)), dim3(PF_BLOCK), 0, st, g->rowptr, g->w, g->g, g->n, tmp);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = pfl::memcpy_async(st, l_offdiag, tmp, sizeof(double) * g->nnz_w, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            pfl::sync(st);
            pf_free(st, tmp);
            pf_set_error("pf_graph_download: %s", hipGetErrorString(e));
            return PF_E_HIP;
        }
    }
    hipError_t e = pfl::sync(st);
    pf_free(st, tmp);
    pf_free(st, tmp_diag);
    PF_HIP(e);
    return PF_OK;
}

}  // extern "C"
