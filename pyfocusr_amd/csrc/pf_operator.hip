// Eigensolver kernels: the fused SpMV + Chebyshev three-term recurrence over the SELL-64
// operator storage, and the dense-vector kernels of the Krylov-Schur driver (multi-dot,
// multi-axpy, basis rotation, residual norm, eigenvector post-processing).
//
// These replace the ARPACK/SuperLU machinery behind `scipy.sparse.linalg.eigs(L, k,
// sigma=1e-10, which="LM", ncv=4k)` (/root/reference/pyfocusr/graph.py:372).  Everything here
// is HBM/L2-bandwidth-bound: one thread per matrix row, 64 consecutive rows per wave reading
// 64 consecutive SELL entries per step (512 B of values + 256 B of column indices per wave
// instruction), x gathered through L2.  Reductions are two-stage and atomic-free, so results
// are bitwise reproducible run to run.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <vector>

#include "pf_internal.h"

namespace {

inline unsigned nblk(int64_t n) { return (unsigned)((n + PF_BLOCK - 1) / PF_BLOCK); }

// tuning knobs of the operator kernel (tools/tune_spmv.sh builds variants with -D...)
#ifndef PF_OP_BLOCK
#define PF_OP_BLOCK 256
#endif
#ifndef PF_OP_NT
#define PF_OP_NT 0
#endif
#if PF_OP_NT
#define PF_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#else
#define PF_STREAM_LOAD(p) (*(p))
#endif

// out = alpha * (shift * x - A x) - beta * prev          (A = diag + SELL off-diagonals)
//   SpMV:            alpha = -1, shift = 0, beta = 0      -> out = A x
//   Chebyshev k = 1: alpha = 1/e, shift = c, beta = 0
//   Chebyshev k > 1: alpha = 2/e, shift = c, beta = 1     (out may alias prev)
struct OpArgs {
    const int64_t* slice_ptr;
    const int32_t* scol;
    const double* sval;
    const double* diag;
    const double* x;
    const double* prev;
    double* out;
    double alpha, shift, beta;
    unsigned n_blocks;  // multiple of 8
};

template <bool HAS_PREV>
__device__ __forceinline__ void sell_op_block(const OpArgs& a, unsigned bid) {
#pragma clang fp contract(fast)
    // Blocks are dealt round-robin over the 8 XCDs (private 4 MiB L2 each).  Give every XCD one
    // contiguous eighth of the (Morton-ordered) rows: its slice of the matrix (~2.6 MB at 250k
    // vertices) and of x then stays in its own L2 from one launch to the next.  n_blocks is a
    // multiple of 8 (n_pad is a multiple of 8 * PF_BLOCK); placement affects speed only.
    const unsigned per_xcd = a.n_blocks >> 3;
    const unsigned blk = (bid & 7u) * per_xcd + (bid >> 3);
    const int64_t row = (int64_t)blk * PF_OP_BLOCK + threadIdx.x;
    const int64_t s = row >> 6;
    const int lane = threadIdx.x & (PF_WAVE - 1);
    const int64_t base = a.slice_ptr[s];
    const int width = (int)((a.slice_ptr[s + 1] - base) >> 6);
    const double* __restrict__ x = a.x;
    const double xi = x[row];
    double acc = a.diag[row] * xi;
    // pairs of entries per lane (pf_sell_index): one 16-byte load of two values + one 8-byte load of two columns
    const int pairs = width >> 1;
    const double2* __restrict__ vp2 = reinterpret_cast<const double2*>(a.sval + base) + lane;
    const int2* __restrict__ cp2 = reinterpret_cast<const int2*>(a.scol + base) + lane;
    int j = 0;
    for (; j + 2 <= pairs; j += 2) {
        const int2 c0 = PF_STREAM_LOAD(cp2 + (int64_t)(j + 0) * PF_WAVE), c1 = PF_STREAM_LOAD(cp2 + (int64_t)(j + 1) * PF_WAVE);
        const double2 v0 = PF_STREAM_LOAD(vp2 + (int64_t)(j + 0) * PF_WAVE), v1 = PF_STREAM_LOAD(vp2 + (int64_t)(j + 1) * PF_WAVE);
        const double x0 = x[c0.x], x1 = x[c0.y], x2 = x[c1.x], x3 = x[c1.y];
        acc += v0.x * x0;
        acc += v0.y * x1;
        acc += v1.x * x2;
        acc += v1.y * x3;
    }
    for (; j < pairs; ++j) {
        const int2 c0 = PF_STREAM_LOAD(cp2 + (int64_t)j * PF_WAVE);
        const double2 v0 = PF_STREAM_LOAD(vp2 + (int64_t)j * PF_WAVE);
        acc += v0.x * x[c0.x];
        acc += v0.y * x[c0.y];
    }
    if (width & 1) {
        const int64_t t = base + (int64_t)pairs * (2 * PF_WAVE) + lane;
        acc += PF_STREAM_LOAD(a.sval + t) * x[PF_STREAM_LOAD(a.scol + t)];
    }
    double r = a.alpha * (a.shift * xi - acc);
    if (HAS_PREV) r -= a.beta * a.prev[row];
    a.out[row] = r;
}

template <bool HAS_PREV>
__global__ __launch_bounds__(PF_OP_BLOCK) void k_sell_op(OpArgs a) {
    sell_op_block<HAS_PREV>(a, blockIdx.x);
}

// the plain product for `gridDim.y` consecutive workspace slots in one launch (the Rayleigh-Ritz tail of a solve: S Z for
// the 6-11 Ritz vectors was one launch per vector, each mostly launch latency): block row y takes slot y
__global__ __launch_bounds__(PF_OP_BLOCK) void k_sell_op_slots(OpArgs a, int64_t slot_stride) {
    a.x += (int64_t)blockIdx.y * slot_stride;
    a.out += (int64_t)blockIdx.y * slot_stride;
    sell_op_block<false>(a, blockIdx.x);
}

// the same step for two independent graphs in one launch (target and source mesh of a pair run
// their Chebyshev recurrences in lockstep): a 250k-vertex step alone is ~5 us, of which ~3 us is
// launch/ramp latency; two per launch amortise it.
template <bool HAS_PREV>
__global__ __launch_bounds__(PF_OP_BLOCK) void k_sell_op2(OpArgs a, OpArgs b) {
    if (blockIdx.x < a.n_blocks) sell_op_block<HAS_PREV>(a, blockIdx.x);
    else sell_op_block<HAS_PREV>(b, blockIdx.x - a.n_blocks);
}

// partial[b][chunk] = sum over the chunk's rows of V_b[i] * w[i]   (fixed order -> deterministic)
// partial[b][chunk] = <slot first+b, slot wslot> over the chunk; with self_col >= 0 column b == self_col is w itself
// (|w|^2 rides along)
__global__ __launch_bounds__(PF_BLOCK) void k_dot_partial(const double* __restrict__ ws, int64_t n_pad, int32_t first,
                                                          int32_t wslot, int64_t n_chunks, double* __restrict__ partial,
                                                          int32_t self_col = -1) {
    __shared__ double red[PF_BLOCK / PF_WAVE];
    const int b = blockIdx.y;
    const int64_t chunk = blockIdx.x;
    const double* v = ws + (int64_t)(b == self_col ? wslot : first + b) * n_pad;
    const double* w = ws + (int64_t)wslot * n_pad;
    const int64_t lo = chunk * PF_DOT_CHUNK;
    const int64_t hi = lo + PF_DOT_CHUNK < n_pad ? lo + PF_DOT_CHUNK : n_pad;
    double s = 0.0;
    for (int64_t i = lo + 2 * threadIdx.x; i < hi; i += 2 * PF_BLOCK) {
        const double2 a = *reinterpret_cast<const double2*>(v + i);
        const double2 c = *reinterpret_cast<const double2*>(w + i);
        s += a.x * c.x;
        s += a.y * c.y;
    }
#pragma unroll
    for (int off = PF_WAVE / 2; off > 0; off >>= 1) s += __shfl_down(s, off, PF_WAVE);
    if ((threadIdx.x & (PF_WAVE - 1)) == 0) red[threadIdx.x / PF_WAVE] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)b * n_chunks + chunk] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[b] = sum_chunks partial[b][chunk];  acc[b] (+)= out[b]
__global__ __launch_bounds__(PF_WAVE) void k_dot_finish(const double* __restrict__ partial, int64_t n_chunks,
                                                        double* __restrict__ out, double* __restrict__ acc, int accumulate) {
    const int b = blockIdx.x;
    double s = 0.0;
    for (int64_t k = threadIdx.x; k < n_chunks; k += PF_WAVE) s += partial[(int64_t)b * n_chunks + k];
#pragma unroll
    for (int off = PF_WAVE / 2; off > 0; off >>= 1) s += __shfl_down(s, off, PF_WAVE);
    if (threadIdx.x == 0) {
        out[b] = s;
        if (acc) acc[b] = accumulate ? acc[b] + s : s;
    }
}

// w -= sum_b h[b] V_b
__global__ __launch_bounds__(PF_BLOCK) void k_multi_axpy(double* __restrict__ ws, int64_t n_pad, int32_t first,
                                                         int32_t count, int32_t wslot, const double* __restrict__ h) {
    const int64_t i = 2 * ((int64_t)blockIdx.x * PF_BLOCK + threadIdx.x);
    if (i >= n_pad) return;
    double2 acc = *reinterpret_cast<double2*>(ws + (int64_t)wslot * n_pad + i);
    for (int b = 0; b < count; ++b) {
        const double hb = h[b];
        const double2 v = *reinterpret_cast<const double2*>(ws + (int64_t)(first + b) * n_pad + i);
        acc.x -= hb * v.x;
        acc.y -= hb * v.y;
    }
    *reinterpret_cast<double2*>(ws + (int64_t)wslot * n_pad + i) = acc;
}

// Fused forms for pf_orth_begin (one Gram-Schmidt pass = k_dot_partial + one of these; no separate finish launch):
// every block finishes the per-chunk partial sums itself - in k_dot_finish's order, so all blocks (and the host) see
// the same bits - then subtracts.  Block 0 also publishes the coefficients.
constexpr int PF_ORTH_MAX = 256;
// k_axpy_finishing: a plain pass, hsum[b] (+)= h[b] (the second pass of the rare case below).
__global__ __launch_bounds__(PF_BLOCK) void k_axpy_finishing(double* __restrict__ ws, int64_t n_pad, int32_t first, int32_t count,
                                                             int32_t wslot, const double* __restrict__ partial, int64_t n_chunks,
                                                             double* __restrict__ hsum, int accumulate) {
    __shared__ double hs[PF_ORTH_MAX];
    const int lane = threadIdx.x & (PF_WAVE - 1);
    for (int b = threadIdx.x / PF_WAVE; b < count; b += PF_BLOCK / PF_WAVE) {
        double s = 0.0;
        for (int64_t k = lane; k < n_chunks; k += PF_WAVE) s += partial[(int64_t)b * n_chunks + k];
#pragma unroll
        for (int off = PF_WAVE / 2; off > 0; off >>= 1) s += __shfl_down(s, off, PF_WAVE);
        if (lane == 0) hs[b] = s;
    }
    __syncthreads();
    if (blockIdx.x == 0)
        for (int b = threadIdx.x; b < count; b += PF_BLOCK) hsum[b] = accumulate ? hsum[b] + hs[b] : hs[b];
    const int64_t i = 2 * ((int64_t)blockIdx.x * PF_BLOCK + threadIdx.x);
    if (i >= n_pad) return;
    double2 acc = *reinterpret_cast<double2*>(ws + (int64_t)wslot * n_pad + i);
    for (int b = 0; b < count; ++b) {
        const double hb = hs[b];
        const double2 v = *reinterpret_cast<const double2*>(ws + (int64_t)(first + b) * n_pad + i);
        acc.x -= hb * v.x;
        acc.y -= hb * v.y;
    }
    *reinterpret_cast<double2*>(ws + (int64_t)wslot * n_pad + i) = acc;
}
// The orthogonalisation step of one graph - or of the two graphs of a pair in the same two launches (blockIdx.z) - is
// k_orth_dots + k_orth_project:
//   h = V^T w and |w|^2 (partial sums per chunk);   w' = w - V h;   |w'|^2 = |w|^2 - sum h^2  (Pythagoras);
//   w' /= |w'| if `normalize`
// and h, |w'|^2 and a verdict go straight to the host's pinned buffer: host_out = [h (count), |w'|^2, redo].
// If |w'| < 0.3 |w| (0.71 |w| in strict mode, pf_orth_strict) the projection cancelled digits - the second Gram-Schmidt
// pass is due ("twice is enough", Daniel, Gragg, Kaufman, Stewart; ARPACK's criterion, by default with a looser constant) and Pythagoras is no longer a fair norm: redo = 1,
// w' stays un-normalised and pf_orth_end runs the second pass itself.  Otherwise the basis stays orthogonal to ~eps / 0.3
// and |w'| carries a relative error <= ~eps / 0.09: far inside what Lanczos needs (semi-orthogonality sqrt(eps) already
// preserves the Ritz values; the eigenvalues handed out come from a Rayleigh-Ritz step on the operator itself).
// Measured on the 250k blobs: the ratio is 0.35-0.78 in all steps but one or two per solve, and a step costs 2 launches
// where it took 6 (the two gated passes and the separate norm were ~4.4 us of launch floor each).
struct OrthArgs {
    double* ws;
    int64_t n_pad, n_chunks;
    int32_t first, count, wslot, normalize;
    // basis vector b < count lives in slot first + b for b < split and in slot first2 + (b - split) from there on: one range
    // (split = count), or - the local steps of Lanczos with partial reorthogonalisation, pf_orth_split - the locked null
    // vectors and the last two basis vectors
    int32_t split, first2;
    double* partial;   // [count + 1][n_chunks]
    double* hsum;      // device copy of h
    double* nrm2;      // device copy of |w'|^2
    double* host_out;  // pinned: h, |w'|^2, redo, then the ticket (written last: the host polls it)
    double thresh;     // second pass when |w'|^2 < thresh |w|^2
    double ticket;     // this step's number (> 0)
    // The second pass ON THE DEVICE (pf_orth_device_passes): the two kernels are queued twice; pass 1 leaves its verdict in
    // device memory and reports to the host only if it was fine, pass 2 returns at once unless the verdict asks for it and
    // reports h1 + h2 and the norm itself.  Whatever is queued behind (the next filter application of a pipelined driver)
    // then always reads the final vector - no pf_orth_redone, no repeated application.
    int32_t pass;      // 0: one pass, a raised verdict is the host's business (pf_orth_end); 1 / 2: first / second of two; -1: not this graph
    double* verdict;   // device copy of pass 1's verdict
    // k_orth_local (the whole step in one launch): arrivals at its grid-wide wait, the count that ends the wait, where a
    // wait that ran out is reported (the resident filter kernel's abort words)
    unsigned long long* counter;
    unsigned long long target;
    uint32_t* abort_flag;
    int32_t* host_abort;
};
struct OrthArgs2 {
    OrthArgs g[2];
};
__device__ __forceinline__ int32_t orth_slot(const OrthArgs& a, int b) { return b < a.split ? a.first + b : a.first2 + (b - a.split); }

// partial[b][chunk] = <slot first+b, slot wslot> over the chunk, b < count; partial[count][chunk] = |w|^2 over the chunk
// (k_dot_partial's sums, in its order)
// VB: basis vectors per block.  1: the shape tuned for bases that live in the Infinity Cache (250k pairs: 6 TB/s).  4 (bases
// beyond it: 1M rows x 50 vectors = 400 MB): the chunk of w is loaded ONCE for four vectors - with one vector per block it
// came back from HBM once per vector, half of the kernel's traffic - and four vectors' 32 KB pieces are in flight per block.
// Per (vector, chunk) the sums are formed by the same threads in the same order either way: the same bits.
template <int VB>
__global__ __launch_bounds__(PF_BLOCK) void k_orth_dots(OrthArgs2 a2) {
    __shared__ double red[VB][PF_BLOCK / PF_WAVE];
    const OrthArgs& a = a2.g[blockIdx.z];
    const int b0 = blockIdx.y * VB;
    const int64_t chunk = blockIdx.x;
    if (a.pass < 0 || b0 > a.count || chunk >= a.n_chunks) return;  // (block-uniform: the grid is sized for the larger graph of a pair)
    if (a.pass == 2 && *a.verdict == 0.0) return;                   // (the first pass was fine)
    const double* w = a.ws + (int64_t)a.wslot * a.n_pad;
    const int64_t lo = chunk * PF_DOT_CHUNK;
    // n_pad is a multiple of the chunk: every chunk is whole, every thread has PF_DOT_CHUNK / (2 PF_BLOCK) = 8 pairs, all
    // of whose loads are issued before the first product waits for one (the sums keep their order)
    constexpr int PAIRS = PF_DOT_CHUNK / (2 * PF_BLOCK);
    double2 cs[PAIRS], xs[VB][PAIRS];
#pragma unroll
    for (int it = 0; it < PAIRS; ++it) cs[it] = *reinterpret_cast<const double2*>(w + lo + 2 * threadIdx.x + (int64_t)it * (2 * PF_BLOCK));
#pragma unroll
    for (int u = 0; u < VB; ++u) {
        const int b = b0 + u;
        if (b > a.count) continue;
        const double* v = a.ws + (int64_t)(b == a.count ? a.wslot : orth_slot(a, b)) * a.n_pad;
#pragma unroll
        for (int it = 0; it < PAIRS; ++it) xs[u][it] = *reinterpret_cast<const double2*>(v + lo + 2 * threadIdx.x + (int64_t)it * (2 * PF_BLOCK));
    }
#pragma unroll
    for (int u = 0; u < VB; ++u) {
        if (b0 + u > a.count) continue;
        double s = 0.0;
#pragma unroll
        for (int it = 0; it < PAIRS; ++it) {
            s += xs[u][it].x * cs[it].x;
            s += xs[u][it].y * cs[it].y;
        }
#pragma unroll
        for (int off = PF_WAVE / 2; off > 0; off >>= 1) s += __shfl_down(s, off, PF_WAVE);
        if ((threadIdx.x & (PF_WAVE - 1)) == 0) red[u][threadIdx.x / PF_WAVE] = s;
    }
    __syncthreads();
    if (threadIdx.x < VB && b0 + (int)threadIdx.x <= a.count)
        a.partial[(int64_t)(b0 + threadIdx.x) * a.n_chunks + chunk] =
            (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// P: pairs of rows per thread.  1: a block owns 512 rows (many blocks: what fills the chip at 250k rows).  8 (bases beyond the
// Infinity Cache): a block owns a whole 4096-row chunk, so that it reads 32 KB of every basis vector in one piece instead of
// 4 KB pieces of fifty streams (DRAM pages), and the sums of the partial dot products are redone by an eighth of the blocks.
// Per row the subtractions are the same, in the same order.
template <int P>
__global__ __launch_bounds__(PF_BLOCK) void k_orth_project(OrthArgs2 a2) {
    __shared__ double hs[PF_ORTH_MAX + 1];
    __shared__ double s_scale, s_after, s_redo;
    const OrthArgs& a = a2.g[blockIdx.z];
    if (a.pass < 0 || 2 * (int64_t)blockIdx.x * PF_BLOCK * P >= a.n_pad) return;  // (block-uniform)
    if (a.pass == 2 && *a.verdict == 0.0) return;
    const int lane = threadIdx.x & (PF_WAVE - 1);
    const int32_t count = a.count;
    for (int b = threadIdx.x / PF_WAVE; b < count + 1; b += PF_BLOCK / PF_WAVE) {
        double s = 0.0;
        for (int64_t k = lane; k < a.n_chunks; k += PF_WAVE) s += a.partial[(int64_t)b * a.n_chunks + k];
#pragma unroll
        for (int off = PF_WAVE / 2; off > 0; off >>= 1) s += __shfl_down(s, off, PF_WAVE);
        if (lane == 0) hs[b] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // the same sequence in every block: the same bits
        double sum = 0.0;
        for (int b = 0; b < count; ++b) sum += hs[b] * hs[b];
        const double before = hs[count], after = before - sum;
        // (false for NaN and for a vanished vector: take the second pass.  The second pass itself is final - "twice is
        // enough" - unless it produced no number at all: verdict 1 then hands the step to pf_orth_end's own passes)
        const bool fine = a.pass == 2 ? (after == after) : after >= a.thresh * before;
        s_redo = fine ? (a.pass == 2 ? 2.0 : 0.0) : 1.0;
        s_after = after;
        s_scale = (fine && a.normalize && after > 1e-280) ? 1.0 / sqrt(after) : 1.0;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        const bool report = !(a.pass == 1 && s_redo != 0.0);  // (a first pass that asks for the second leaves the report to it)
        for (int b = threadIdx.x; b < count; b += PF_BLOCK) {
            const double hb = a.pass == 2 ? a.hsum[b] + hs[b] : hs[b];
            a.hsum[b] = hb;
            if (report) a.host_out[b] = hb;
        }
        if (threadIdx.x == 0) {
            if (a.pass == 1) *a.verdict = s_redo;
            *a.nrm2 = s_after;
            if (report) {
                a.host_out[count] = s_after;
                a.host_out[count + 1] = s_redo;
            }
        }
        // the ticket goes out behind everything else (system-scope fences + the block's barrier): the host polls it
        // instead of waiting for an event behind the kernel - the coefficients are known long before the projection
        // below has finished, and an event record would cost ~5 us of device time per step
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0 && report) {
            __hip_atomic_store(a.host_out + count + 2, a.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    const double sc = s_scale;
    if constexpr (P == 1) {
        const int64_t i = 2 * ((int64_t)blockIdx.x * PF_BLOCK + threadIdx.x);
        if (i >= a.n_pad) return;
        double2 acc = *reinterpret_cast<double2*>(a.ws + (int64_t)a.wslot * a.n_pad + i);
        // eight basis vectors' loads in flight per thread (the subtractions keep their order: the same bits)
        int b = 0;
        for (; b + 8 <= count; b += 8) {
            double2 vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vv[u] = *reinterpret_cast<const double2*>(a.ws + (int64_t)orth_slot(a, b + u) * a.n_pad + i);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double hb = hs[b + u];
                acc.x -= hb * vv[u].x;
                acc.y -= hb * vv[u].y;
            }
        }
        if (b < count) {
            // the last 1-7 vectors (a local step of Lanczos with partial reorthogonalisation has 3-4 in all): their loads in
            // flight together as well - one round trip instead of one per vector; a vector past the end is the last one
            // again with coefficient 0 (x - 0 v = x exactly)
            double2 vv[7];
            double hh[7];
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const int bu = b + u < count ? b + u : count - 1;
                vv[u] = *reinterpret_cast<const double2*>(a.ws + (int64_t)orth_slot(a, bu) * a.n_pad + i);
                hh[u] = b + u < count ? hs[bu] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                acc.x -= hh[u] * vv[u].x;
                acc.y -= hh[u] * vv[u].y;
            }
        }
        acc.x *= sc;
        acc.y *= sc;
        *reinterpret_cast<double2*>(a.ws + (int64_t)a.wslot * a.n_pad + i) = acc;
    } else {
        const int64_t i0 = (int64_t)blockIdx.x * (2 * PF_BLOCK * P) + 2 * threadIdx.x;  // (n_pad is a multiple of 4096: whole blocks)
        double2 acc[P];
#pragma unroll
        for (int it = 0; it < P; ++it) acc[it] = *reinterpret_cast<double2*>(a.ws + (int64_t)a.wslot * a.n_pad + i0 + (int64_t)it * (2 * PF_BLOCK));
        // (one block per CU at this shape: two vectors' pieces - 2 x 8 x 16 bytes per thread - are requested before the
        // first is used; the subtractions keep the order b, b + 1)
        int b = 0;
        for (; b + 2 <= count; b += 2) {
            const double h0 = hs[b], h1 = hs[b + 1];
            const double* v0 = a.ws + (int64_t)orth_slot(a, b) * a.n_pad + i0;
            const double* v1 = a.ws + (int64_t)orth_slot(a, b + 1) * a.n_pad + i0;
            double2 x0[P], x1[P];
#pragma unroll
            for (int it = 0; it < P; ++it) x0[it] = *reinterpret_cast<const double2*>(v0 + (int64_t)it * (2 * PF_BLOCK));
#pragma unroll
            for (int it = 0; it < P; ++it) x1[it] = *reinterpret_cast<const double2*>(v1 + (int64_t)it * (2 * PF_BLOCK));
#pragma unroll
            for (int it = 0; it < P; ++it) {
                acc[it].x -= h0 * x0[it].x;
                acc[it].y -= h0 * x0[it].y;
                acc[it].x -= h1 * x1[it].x;
                acc[it].y -= h1 * x1[it].y;
            }
        }
        for (; b < count; ++b) {
            const double hb = hs[b];
            const double* vb = a.ws + (int64_t)orth_slot(a, b) * a.n_pad + i0;
            double2 vv[P];
#pragma unroll
            for (int it = 0; it < P; ++it) vv[it] = *reinterpret_cast<const double2*>(vb + (int64_t)it * (2 * PF_BLOCK));
#pragma unroll
            for (int it = 0; it < P; ++it) {
                acc[it].x -= hb * vv[it].x;
                acc[it].y -= hb * vv[it].y;
            }
        }
#pragma unroll
        for (int it = 0; it < P; ++it) {
            acc[it].x *= sc;
            acc[it].y *= sc;
            *reinterpret_cast<double2*>(a.ws + (int64_t)a.wslot * a.n_pad + i0 + (int64_t)it * (2 * PF_BLOCK)) = acc[it];
        }
    }
}
// One lane per block: arrival at the meeting of the blocks of a graph inside a launch, then a bounded wait until the counter
// has reached `want`.  What the blocks exchange travels in 8-byte agent-scope atomic stores and loads (the partial sums):
// performed at the memory side, the stores waited for before the arrival, the loads issued behind the poll that matched -
// no release in front and no acquire behind (an L2 write-back and an invalidation: ~3 us of a 14 us step), and the poll
// itself is a relaxed load.  0: the wait ran out (abort words raised).
__device__ __forceinline__ int orth_meet(const OrthArgs& a, unsigned long long want) {
    __hip_atomic_fetch_add(a.counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000ull ||  // 0.2 s of the 100 MHz clock
            __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *a.host_abort = 1;
            return 0;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    return 1;
}

// A LOCAL Gram-Schmidt step (Lanczos with partial reorthogonalisation: the null vectors and the last two basis vectors, 4
// at most) in ONE launch.  A block owns a 4096-row chunk: it loads w and the basis vectors' pieces once, forms the chunk's
// partial sums exactly as k_orth_dots does, publishes them, waits until every chunk of the graph has (a counter, bounded),
// sums the partials exactly as k_orth_project does and projects the rows it still holds in registers - the same
// operations in the same order on every number: the same bits as the two launches, one kernel boundary less (~5 us of a
// ~16 us step).  The grid is one block per chunk and graph (122 at 250k, 490 at 1M rows): all resident.
constexpr int PF_ORTH_LOCAL_MAX = 4;
__global__ __launch_bounds__(PF_BLOCK) void k_orth_local(OrthArgs2 a2) {
    __shared__ double red[PF_ORTH_LOCAL_MAX + 1][PF_BLOCK / PF_WAVE];
    __shared__ double hs[PF_ORTH_LOCAL_MAX + 1];
    __shared__ double s_scale, s_after, s_redo;
    __shared__ int s_ok;
    const OrthArgs& a = a2.g[blockIdx.z];
    const int64_t chunk = blockIdx.x;
    if (a.pass < 0 || chunk >= a.n_chunks) return;  // (block-uniform: the grid is sized for the larger graph of a pair)
    const int32_t count = a.count;
    const int lane = threadIdx.x & (PF_WAVE - 1);
    double* w = a.ws + (int64_t)a.wslot * a.n_pad;
    const int64_t lo = chunk * PF_DOT_CHUNK;
    constexpr int PAIRS = PF_DOT_CHUNK / (2 * PF_BLOCK);
    double2 cs[PAIRS], xs[PF_ORTH_LOCAL_MAX][PAIRS];
#pragma unroll
    for (int it = 0; it < PAIRS; ++it) cs[it] = *reinterpret_cast<const double2*>(w + lo + 2 * threadIdx.x + (int64_t)it * (2 * PF_BLOCK));
#pragma unroll
    for (int u = 0; u < PF_ORTH_LOCAL_MAX; ++u) {
        const double* v = a.ws + (int64_t)orth_slot(a, u < count ? u : count - 1) * a.n_pad;  // (past the end: the last one again, unused)
#pragma unroll
        for (int it = 0; it < PAIRS; ++it) xs[u][it] = *reinterpret_cast<const double2*>(v + lo + 2 * threadIdx.x + (int64_t)it * (2 * PF_BLOCK));
    }
    // ---- the chunk's partial sums: k_orth_dots' operations in its order (column `count`: |w|^2)
#pragma unroll
    for (int u = 0; u <= PF_ORTH_LOCAL_MAX; ++u) {
        if (u > count) continue;
        double s = 0.0;
#pragma unroll
        for (int it = 0; it < PAIRS; ++it) {
            const double2 x = u == count ? cs[it] : xs[u < PF_ORTH_LOCAL_MAX ? u : 0][it];
            s += x.x * cs[it].x;
            s += x.y * cs[it].y;
        }
#pragma unroll
        for (int off = PF_WAVE / 2; off > 0; off >>= 1) s += __shfl_down(s, off, PF_WAVE);
        if (lane == 0) red[u][threadIdx.x / PF_WAVE] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x <= count) {
        const double p = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.partial) + (int64_t)threadIdx.x * a.n_chunks + chunk,
                           (unsigned long long)__double_as_longlong(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // in memory before the block says it has arrived
    }
    __syncthreads();
    // ---- every chunk of this graph has published: one arrival per block, a bounded wait
    if (threadIdx.x == 0) s_ok = orth_meet(a, a.target);
    __syncthreads();
    if (!s_ok) return;
    // ---- the sums of the partials: k_orth_project's operations in its order (read from where the other XCDs wrote them)
    {  // (the two vectors of a wave - b and b + 4 - with their loads in flight together; per vector the order of the sum is kept)
        const int b0 = threadIdx.x / PF_WAVE, b1 = b0 + PF_BLOCK / PF_WAVE;
        double s0 = 0.0, s1 = 0.0;
        for (int64_t k = lane; k < a.n_chunks; k += PF_WAVE) {
            const double x0 = b0 < count + 1 ? __longlong_as_double((long long)__hip_atomic_load(
                                                   reinterpret_cast<unsigned long long*>(a.partial) + (int64_t)b0 * a.n_chunks + k, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT))
                                             : 0.0;
            const double x1 = b1 < count + 1 ? __longlong_as_double((long long)__hip_atomic_load(
                                                   reinterpret_cast<unsigned long long*>(a.partial) + (int64_t)b1 * a.n_chunks + k, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT))
                                             : 0.0;
            s0 += x0;
            s1 += x1;
        }
#pragma unroll
        for (int off = PF_WAVE / 2; off > 0; off >>= 1) {
            s0 += __shfl_down(s0, off, PF_WAVE);
            s1 += __shfl_down(s1, off, PF_WAVE);
        }
        if (lane == 0 && b0 < count + 1) hs[b0] = s0;
        if (lane == 0 && b1 < count + 1) hs[b1] = s1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0;
        for (int b = 0; b < count; ++b) sum += hs[b] * hs[b];
        const double before = hs[count], after = before - sum;
        // (one pass: a raised verdict is pf_orth_end's business.  The flag goes through a vector register on purpose: this
        // compiler turned `fine ? 0.0 : 1.0` into an s_cselect on SCC behind a v_cmp that sets VCC - the verdict then was
        // whatever the last scalar compare had left)
        int fine = after >= a.thresh * before ? 1 : 0;
        asm volatile("" : "+v"(fine));
        s_redo = fine ? 0.0 : 1.0;
        s_after = after;
        s_scale = (fine && a.normalize && after > 1e-280) ? 1.0 / sqrt(after) : 1.0;
    }
    __syncthreads();
    if (chunk == 0) {
        for (int b = threadIdx.x; b < count; b += PF_BLOCK) {
            a.hsum[b] = hs[b];
            a.host_out[b] = hs[b];
        }
        if (threadIdx.x == 0) {
            *a.nrm2 = s_after;
            a.host_out[count] = s_after;
            a.host_out[count + 1] = s_redo;
        }
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(a.host_out + count + 2, a.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const double sc = s_scale;
#pragma unroll
    for (int it = 0; it < PAIRS; ++it) {
        double2 acc = cs[it];
#pragma unroll
        for (int u = 0; u < PF_ORTH_LOCAL_MAX; ++u) {
            if (u >= count) continue;
            const double hb = hs[u];
            acc.x -= hb * xs[u][it].x;
            acc.y -= hb * xs[u][it].y;
        }
        acc.x *= sc;
        acc.y *= sc;
        *reinterpret_cast<double2*>(w + lo + 2 * threadIdx.x + (int64_t)it * (2 * PF_BLOCK)) = acc;
    }
}
__global__ __launch_bounds__(PF_BLOCK) void k_scale_finishing(double* __restrict__ x, int64_t n_pad, const double* __restrict__ partial,
                                                              int64_t n_chunks, int normalize, const double* __restrict__ hsum,
                                                              int32_t count, double* __restrict__ nrm2, double* __restrict__ host_out) {
    __shared__ double s_v;
    if (threadIdx.x < PF_WAVE) {
        double s = 0.0;
        for (int64_t k = threadIdx.x; k < n_chunks; k += PF_WAVE) s += partial[k];
#pragma unroll
        for (int off = PF_WAVE / 2; off > 0; off >>= 1) s += __shfl_down(s, off, PF_WAVE);
        if (threadIdx.x == 0) s_v = s;
    }
    __syncthreads();
    const double v = s_v;
    if (blockIdx.x == 0) {
        for (int b = threadIdx.x; b < count; b += PF_BLOCK) host_out[b] = hsum[b];
        if (threadIdx.x == 0) {
            *nrm2 = v;
            host_out[count] = v;
            host_out[count + 1] = 0.0;  // (the verdict slot of k_orth_project: nothing left to redo)
        }
    }
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (normalize && i < n_pad && v > 1e-280) x[i] *= 1.0 / sqrt(v);
}

__global__ __launch_bounds__(PF_BLOCK) void k_scale(double* __restrict__ x, int64_t n_pad, double alpha) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i < n_pad) x[i] *= alpha;
}

// x *= 1/sqrt(*nrm2) with the norm still on the device (no host round trip between a Lanczos step and the next
// filter application); left alone when the norm is ~0 (invariant subspace: the host stops the iteration)
__global__ __launch_bounds__(PF_BLOCK) void k_scale_rsqrt(double* __restrict__ x, int64_t n_pad, const double* __restrict__ nrm2) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    const double v = *nrm2;
    if (i < n_pad && v > 1e-280) x[i] *= 1.0 / sqrt(v);
}

__global__ __launch_bounds__(PF_BLOCK) void k_mask_isolated(double* __restrict__ x, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ perm, int64_t n) {
    const int64_t r = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (r >= n) return;
    const int32_t i = perm[r];
    if (rowptr[i + 1] == rowptr[i]) x[r] = 0.0;
}

// Krylov start vector, part 2: smooth part + counter-based noise (splitmix64 of (seed, vertex)), zero
// on isolated vertices and padding.  Noise keeps every eigenmode present whatever the geometry.
__global__ __launch_bounds__(PF_BLOCK) void k_start_vector(double* __restrict__ x, const double* __restrict__ smooth,
                                                           const int32_t* __restrict__ perm, const int32_t* __restrict__ perm_m,
                                                           const int32_t* __restrict__ rowptr,
                                                           int64_t n_pad, unsigned long long seed, double noise) {
    const int64_t r = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (r >= n_pad) return;
    const int32_t i = perm[r];    // the caller's vertex number: what the noise is keyed by, whatever the internal order
    const int32_t m = perm_m[r];  // the row of CSR(W)
    double v = 0.0;
    if (i >= 0 && rowptr[m + 1] > rowptr[m]) {
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;  // uniform in [-0.5, 0.5)
        v = smooth[r] + noise * u;
    }
    x[r] = v;
}

// host order <-> solver order
__global__ __launch_bounds__(PF_BLOCK) void k_permute_in(const double* __restrict__ src_old, const int32_t* __restrict__ perm,
                                                         int64_t n_pad, double* __restrict__ dst_new) {
    const int64_t r = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (r >= n_pad) return;
    const int32_t i = perm[r];
    dst_new[r] = i >= 0 ? src_old[i] : 0.0;
}

__global__ __launch_bounds__(PF_BLOCK) void k_permute_out(const double* __restrict__ ws, int64_t n_pad, int32_t first,
                                                          int32_t count, const int32_t* __restrict__ iperm, int64_t n,
                                                          double* __restrict__ dst_old /* [count][n] */) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int64_t r = iperm[i];
    for (int c = 0; c < count; ++c) dst_old[(int64_t)c * n + i] = ws[(int64_t)(first + c) * n_pad + r];
}

__global__ __launch_bounds__(PF_BLOCK) void k_null_vector(double* __restrict__ x, const int32_t* __restrict__ label,
                                                          const int32_t* __restrict__ rowptr,
                                                          const double* __restrict__ deg, const int32_t* __restrict__ perm,
                                                          int64_t n_pad, int32_t root, int32_t sym) {
    // null vector of the iterated operator on one component: 1_C for L = G (D - W) and for a general matrix with zero
    // row sums (G = I: sym == 0 is passed), G^-1/2 1_C = sqrt(deg + 1e-8) 1_C for S = G^1/2 (D - W) G^1/2 of a mesh graph
    const int64_t r = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (r >= n_pad) return;
    const int32_t i = perm[r];
    double v = 0.0;
    if (i >= 0 && label[i] == root && rowptr[i + 1] > rowptr[i]) v = sym ? sqrt(deg[i] + 1e-8) : 1.0;
    x[r] = v;
}

// dst_c = sum_b src_b Y[b][c] for up to 8 output columns per launch (Y row-major m x k, in device memory)
constexpr int COMBINE_COLS = 8;
// (blockIdx.y = 1: the same rotation of a second block of vectors, src_first2 -> dst_first2)
__global__ __launch_bounds__(PF_BLOCK) void k_combine(double* __restrict__ ws, int64_t n_pad, int32_t src_first, int32_t m,
                                                      const double* __restrict__ Y, int32_t k, int32_t c0, int32_t ncols,
                                                      int32_t dst_first, int32_t src_first2, int32_t dst_first2) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= n_pad) return;
    if (blockIdx.y) src_first = src_first2, dst_first = dst_first2;
    double acc[COMBINE_COLS];
#pragma unroll
    for (int c = 0; c < COMBINE_COLS; ++c) acc[c] = 0.0;
    for (int b = 0; b < m; ++b) {
        const double v = ws[(int64_t)(src_first + b) * n_pad + i];
        const double* yr = Y + (int64_t)b * k + c0;
#pragma unroll
        for (int c = 0; c < COMBINE_COLS; ++c)
            if (c < ncols) acc[c] += v * yr[c];
    }
#pragma unroll
    for (int c = 0; c < COMBINE_COLS; ++c)
        if (c < ncols) ws[(int64_t)(dst_first + c0 + c) * n_pad + i] = acc[c];
}

// partial sums of (ax - lam x)^2
__global__ __launch_bounds__(PF_BLOCK) void k_resnorm_partial(const double* __restrict__ ax, const double* __restrict__ x,
                                                              double lam, int64_t n_pad, int64_t n_chunks,
                                                              double* __restrict__ partial) {
    __shared__ double red[PF_BLOCK / PF_WAVE];
    const int64_t lo = (int64_t)blockIdx.x * PF_DOT_CHUNK;
    const int64_t hi = lo + PF_DOT_CHUNK < n_pad ? lo + PF_DOT_CHUNK : n_pad;
    double s = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += PF_BLOCK) {
        const double r = ax[i] - lam * x[i];
        s += r * r;
    }
#pragma unroll
    for (int off = PF_WAVE / 2; off > 0; off >>= 1) s += __shfl_down(s, off, PF_WAVE);
    if ((threadIdx.x & (PF_WAVE - 1)) == 0) red[threadIdx.x / PF_WAVE] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// the batched forms (pf_gram, pf_resnorms): one launch for all pairs / all vectors, k_dot_partial's and
// k_resnorm_partial's sums in their order
__global__ __launch_bounds__(PF_BLOCK) void k_gram_partial(const double* __restrict__ ws, int64_t n_pad, int32_t first_a, int32_t first_b,
                                                           int32_t count_b, int64_t n_chunks, double* __restrict__ partial) {
    __shared__ double red[PF_BLOCK / PF_WAVE];
    const int p = blockIdx.y;  // pair (i, j): row i of the first block against row j of the second
    const int i = p / count_b, j = p - i * count_b;
    const double* v = ws + (int64_t)(first_b + j) * n_pad;
    const double* w = ws + (int64_t)(first_a + i) * n_pad;
    const int64_t lo = (int64_t)blockIdx.x * PF_DOT_CHUNK;
    const int64_t hi = lo + PF_DOT_CHUNK < n_pad ? lo + PF_DOT_CHUNK : n_pad;
    double s = 0.0;
    for (int64_t r = lo + 2 * threadIdx.x; r < hi; r += 2 * PF_BLOCK) {
        const double2 a = *reinterpret_cast<const double2*>(v + r);
        const double2 c = *reinterpret_cast<const double2*>(w + r);
        s += a.x * c.x;
        s += a.y * c.y;
    }
#pragma unroll
    for (int off = PF_WAVE / 2; off > 0; off >>= 1) s += __shfl_down(s, off, PF_WAVE);
    if ((threadIdx.x & (PF_WAVE - 1)) == 0) red[threadIdx.x / PF_WAVE] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)p * n_chunks + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

constexpr int PF_RESNORMS_MAX = 64;
struct LamArgs {
    double lam[PF_RESNORMS_MAX];
};
__global__ __launch_bounds__(PF_BLOCK) void k_resnorms_partial(const double* __restrict__ ws, int64_t n_pad, int32_t ax_first,
                                                               int32_t x_first, LamArgs la, int64_t n_chunks,
                                                               double* __restrict__ partial) {
    __shared__ double red[PF_BLOCK / PF_WAVE];
    const int b = blockIdx.y;
    const double* ax = ws + (int64_t)(ax_first + b) * n_pad;
    const double* x = ws + (int64_t)(x_first + b) * n_pad;
    const double lam = la.lam[b];
    const int64_t lo = (int64_t)blockIdx.x * PF_DOT_CHUNK;
    const int64_t hi = lo + PF_DOT_CHUNK < n_pad ? lo + PF_DOT_CHUNK : n_pad;
    double s = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += PF_BLOCK) {
        const double r = ax[i] - lam * x[i];
        s += r * r;
    }
#pragma unroll
    for (int off = PF_WAVE / 2; off > 0; off >>= 1) s += __shfl_down(s, off, PF_WAVE);
    if ((threadIdx.x & (PF_WAVE - 1)) == 0) red[threadIdx.x / PF_WAVE] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)b * n_chunks + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- eigenvector post-processing -----------------------------------------------------------------
struct VecStats {
    double sumsq, vmin, vmax, absmax, at_absmax;
    int64_t arg;
};

__device__ __forceinline__ void stats_merge(VecStats& a, const VecStats& b) {
    a.sumsq += b.sumsq;
    a.vmin = fmin(a.vmin, b.vmin);
    a.vmax = fmax(a.vmax, b.vmax);
    if (b.absmax > a.absmax || (b.absmax == a.absmax && b.arg < a.arg)) {
        a.absmax = b.absmax;
        a.at_absmax = b.at_absmax;
        a.arg = b.arg;
    }
}

__device__ __forceinline__ VecStats stats_shfl(const VecStats& v, int off) {
    VecStats o;
    o.sumsq = __shfl_down(v.sumsq, off, PF_WAVE);
    o.vmin = __shfl_down(v.vmin, off, PF_WAVE);
    o.vmax = __shfl_down(v.vmax, off, PF_WAVE);
    o.absmax = __shfl_down(v.absmax, off, PF_WAVE);
    o.at_absmax = __shfl_down(v.at_absmax, off, PF_WAVE);
    o.arg = __shfl_down(v.arg, off, PF_WAVE);
    return o;
}

__global__ __launch_bounds__(PF_BLOCK) void k_vec_stats_partial(const double* __restrict__ ws, int64_t n_pad, int64_t n,
                                                                int32_t first, const double* __restrict__ sg,
                                                                const int32_t* __restrict__ perm, const int32_t* __restrict__ perm_m,
                                                                int32_t from_sym, int64_t n_chunks, VecStats* __restrict__ partial) {
    __shared__ VecStats red[PF_BLOCK / PF_WAVE];
    const int b = blockIdx.y;
    const double* x = ws + (int64_t)(first + b) * n_pad;
    const int64_t lo = (int64_t)blockIdx.x * PF_DOT_CHUNK;
    const int64_t hi0 = lo + PF_DOT_CHUNK;
    const int64_t hi = hi0 < n ? hi0 : n;
    VecStats st{0.0, INFINITY, -INFINITY, -1.0, 0.0, (int64_t)1 << 62};
    for (int64_t r = lo + threadIdx.x; r < hi; r += PF_BLOCK) {
        const int64_t i = perm[r];  // mesh-order index: the tie-break key of the sign convention
        double v = x[r];
        if (from_sym) v *= sg[perm_m[r]];  // (per-vertex arrays live in m-space)
        VecStats e{v * v, v, v, fabs(v), v, i};
        stats_merge(st, e);
    }
#pragma unroll
    for (int off = PF_WAVE / 2; off > 0; off >>= 1) {
        VecStats o = stats_shfl(st, off);
        stats_merge(st, o);
    }
    if ((threadIdx.x & (PF_WAVE - 1)) == 0) red[threadIdx.x / PF_WAVE] = st;
    __syncthreads();
    if (threadIdx.x == 0) {
        VecStats t = red[0];
        stats_merge(t, red[1]);
        stats_merge(t, red[2]);
        stats_merge(t, red[3]);
        partial[(int64_t)b * n_chunks + blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(PF_WAVE) void k_vec_stats_finish(const VecStats* __restrict__ partial, int64_t n_chunks,
                                                              VecStats* __restrict__ out) {
    const int b = blockIdx.x;
    VecStats st{0.0, INFINITY, -INFINITY, -1.0, 0.0, (int64_t)1 << 62};
    for (int64_t k = threadIdx.x; k < n_chunks; k += PF_WAVE) stats_merge(st, partial[(int64_t)b * n_chunks + k]);
#pragma unroll
    for (int off = PF_WAVE / 2; off > 0; off >>= 1) {
        VecStats o = stats_shfl(st, off);
        stats_merge(st, o);
    }
    if (threadIdx.x == 0) out[b] = st;
}

// params[c] = {scale, vmin, vmax - vmin, shift} of vector c from its statistics (what pf_finalize_vectors documents:
// unit 2-norm, the largest-|entry| positive, min-max to [-0.5, 0.5] if asked)
__global__ void k_vec_params(const VecStats* __restrict__ fin, int32_t count, int32_t minmax, double* __restrict__ params) {
    for (int c = threadIdx.x; c < count; c += blockDim.x) {
        const VecStats s = fin[c];
        const double sgn = s.at_absmax < 0.0 ? -1.0 : 1.0;
        const double scale = sgn / sqrt(s.sumsq);
        const double vmin = (sgn > 0 ? s.vmin : s.vmax) * scale;  // monotone map: exact min/max of the scaled vector
        const double vmax = (sgn > 0 ? s.vmax : s.vmin) * scale;
        params[4 * c + 0] = scale;
        params[4 * c + 1] = minmax ? vmin : 0.0;
        params[4 * c + 2] = minmax ? (vmax - vmin) : 0.0;
        params[4 * c + 3] = minmax ? 0.5 : 0.0;
    }
}

// out[i][c] = (x_c[i] * scale_c - off_c) * inv_c - half_c       (row-major n x count)
__global__ __launch_bounds__(PF_BLOCK) void k_vec_apply(const double* __restrict__ ws, int64_t n_pad, int64_t n,
                                                        int32_t first, int32_t count, const double* __restrict__ sg,
                                                        const int32_t* __restrict__ iperm, const int32_t* __restrict__ mrank,
                                                        int32_t from_sym, const double* __restrict__ params, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int64_t r = iperm[i];
    const double s = from_sym ? sg[mrank ? mrank[i] : i] : 1.0;
    for (int c = 0; c < count; ++c) {
        const double scale = params[4 * c + 0], off = params[4 * c + 1], ptp = params[4 * c + 2], half = params[4 * c + 3];
        double v = (ws[(int64_t)(first + c) * n_pad + r] * s) * scale;
        if (ptp != 0.0) v = (v - off) / ptp - half;  // graph.py:254-257
        out[i * count + c] = v;
    }
}

// out[t][c] = final[rows[t]][c]: sampled rows of the resident eigenvector block (graph.py:266-267 on the device)
__global__ __launch_bounds__(PF_BLOCK) void k_final_rows(const double* __restrict__ fin, const int64_t* __restrict__ rows, int64_t n_rows,
                                                         int32_t fc, double* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (e >= n_rows * fc) return;
    const int64_t t = e / fc;
    out[e] = fin[rows[t] * fc + (e - t * fc)];
}

// Mean filter (graph.py:349-353): out = ((D+I)^-1 (W+I)) in, applied `iterations` times to an n x ncols array.
// scipy forms average_mat = D_inv @ (W + I) as a sparse-sparse product whose rows come out in DESCENDING column
// order (SMMP linked list), and average_mat @ v then sums in that order with the products (dinv*w)*v.  The same
// terms in the same order are kept here, but rows and gathers live in the solver's Morton/degree order (SELL-64,
// coalesced entry loads, neighbours close in memory) instead of the mesh's own vertex order.
__global__ __launch_bounds__(PF_BLOCK) void k_fill_mean_filter(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                               const double* __restrict__ w, const double* __restrict__ deg,
                                                               const int32_t* __restrict__ perm, const int32_t* __restrict__ iperm,
                                                               const int32_t* __restrict__ key, int64_t n_pad,
                                                               const int64_t* __restrict__ slice_ptr,
                                                               int32_t* __restrict__ mf_col, double* __restrict__ mf_val) {
    const int64_t row = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (row >= n_pad) return;
    const int64_t s = row / PF_WAVE;
    const int lane = (int)(row & (PF_WAVE - 1));
    const int32_t width = (int32_t)((slice_ptr[s + 1] - slice_ptr[s]) / PF_WAVE) + 1;
    const int64_t base = slice_ptr[s] + (int64_t)PF_WAVE * s;
    const int32_t i = perm[row];
    int32_t e = 0;
    if (i >= 0) {
        const double dinv = 1.0 / (1.0 + deg[i]);
        bool diag_done = false;
        const int32_t ki = key ? key[i] : i;  // (scipy's order is that of the caller's vertex numbers; perm / iperm / col: m-space)
        for (int32_t a = rowptr[i + 1] - 1; a >= rowptr[i]; --a) {
            const int32_t j = col[a];
            if (!diag_done && (key ? key[j] : j) < ki) {
                mf_col[base + (int64_t)PF_WAVE * e + lane] = (int32_t)row;
                mf_val[base + (int64_t)PF_WAVE * e + lane] = dinv;
                ++e;
                diag_done = true;
            }
            mf_col[base + (int64_t)PF_WAVE * e + lane] = iperm[j];
            mf_val[base + (int64_t)PF_WAVE * e + lane] = dinv * w[a];
            ++e;
        }
        if (!diag_done) {
            mf_col[base + (int64_t)PF_WAVE * e + lane] = (int32_t)row;
            mf_val[base + (int64_t)PF_WAVE * e + lane] = dinv;
            ++e;
        }
    }
    for (; e < width; ++e) {  // padding: 0 * own value
        mf_col[base + (int64_t)PF_WAVE * e + lane] = (int32_t)row;
        mf_val[base + (int64_t)PF_WAVE * e + lane] = 0.0;
    }
}

// NC > 0: compile-time column count; NC == 0: run-time ncols (one pass over the entries per column)
template <int NC>
__global__ __launch_bounds__(PF_BLOCK) void k_mean_filter(const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ mf_col,
                                                          const double* __restrict__ mf_val, const int32_t* __restrict__ perm,
                                                          int64_t n_pad, int32_t ncols, const double* __restrict__ in,
                                                          double* __restrict__ out) {
    const int64_t row = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (row >= n_pad) return;
    const int64_t s = row / PF_WAVE;
    const int lane = (int)(row & (PF_WAVE - 1));
    const int32_t width = (int32_t)((slice_ptr[s + 1] - slice_ptr[s]) / PF_WAVE) + 1;
    const int64_t base = slice_ptr[s] + (int64_t)PF_WAVE * s + lane;
    const bool real = perm[row] >= 0;
    if constexpr (NC > 0) {
        double acc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = 0.0;
        for (int32_t e = 0; e < width; ++e) {
            const int64_t j = mf_col[base + (int64_t)PF_WAVE * e];
            const double v = mf_val[base + (int64_t)PF_WAVE * e];
            // padding entries (0 * own value) must not touch the sum: 0 * inf or -0.0 would change bits
            if (v != 0.0 || j != row) {
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] += v * in[j * NC + c];
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) out[row * NC + c] = real ? acc[c] : 0.0;
    } else {
        for (int32_t c = 0; c < ncols; ++c) {
            double acc = 0.0;
            for (int32_t e = 0; e < width; ++e) {
                const double v = mf_val[base + (int64_t)PF_WAVE * e];
                const int64_t j = mf_col[base + (int64_t)PF_WAVE * e];
                if (v != 0.0 || j != row) acc += v * in[j * ncols + c];
            }
            out[row * ncols + c] = real ? acc : 0.0;
        }
    }
}

// rows between mesh order [n][ncols] and solver order [n_pad][ncols]
__global__ __launch_bounds__(PF_BLOCK) void k_rows_in(const double* __restrict__ src, const int32_t* __restrict__ perm, int64_t n_pad,
                                                      int32_t ncols, double* __restrict__ dst) {
    const int64_t t = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (t >= n_pad * ncols) return;
    const int64_t row = t / ncols;
    const int32_t old = perm[row];
    dst[t] = old >= 0 ? src[(int64_t)old * ncols + (t - row * ncols)] : 0.0;
}
__global__ __launch_bounds__(PF_BLOCK) void k_rows_out(const double* __restrict__ src, const int32_t* __restrict__ iperm, int64_t n,
                                                       int32_t ncols, double* __restrict__ dst) {
    const int64_t t = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (t >= n * ncols) return;
    const int64_t old = t / ncols;
    dst[t] = src[(int64_t)iperm[old] * ncols + (t - old * ncols)];
}

// ---- row subsets (pf_rows_*): boundary / ghost rows of a row-partitioned solve, addressed in solver order
__global__ __launch_bounds__(PF_BLOCK) void k_rows_to_new(const int64_t* __restrict__ old_idx, const int32_t* __restrict__ iperm,
                                                          int64_t cnt, int32_t* __restrict__ new_idx) {
    const int64_t t = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (t < cnt) new_idx[t] = iperm[old_idx[t]];
}
__global__ __launch_bounds__(PF_BLOCK) void k_rows_gather(const double* __restrict__ x, const int32_t* __restrict__ idx, int64_t cnt,
                                                          double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (t < cnt) out[t] = x[idx[t]];
}
__global__ __launch_bounds__(PF_BLOCK) void k_rows_scatter(double* __restrict__ x, const int32_t* __restrict__ idx, int64_t cnt,
                                                           const double* __restrict__ in) {
    const int64_t t = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (t < cnt) x[idx[t]] = in[t];
}
// both recurrence vectors of a boundary exchange in one launch each way
__global__ __launch_bounds__(PF_BLOCK) void k_rows_gather2(const double* __restrict__ xa, const double* __restrict__ xb,
                                                           const int32_t* __restrict__ idx, int64_t cnt, int64_t stride,
                                                           double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (t >= cnt) return;
    const int32_t r = idx[t];
    out[t] = xa[r];
    if (xb) out[stride + t] = xb[r];
}
__global__ __launch_bounds__(PF_BLOCK) void k_rows_scatter2(double* __restrict__ xa, double* __restrict__ xb,
                                                            const int32_t* __restrict__ idx, const int64_t* __restrict__ off,
                                                            int64_t cnt, int64_t stride, const double* __restrict__ src) {
    const int64_t t = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (t >= cnt) return;
    const int32_t r = idx[t];
    const int64_t o = off[t];
    xa[r] = src[o];
    if (xb) xb[r] = src[o + stride];
}
__global__ __launch_bounds__(PF_BLOCK) void k_rows_fill(double* __restrict__ x, const int32_t* __restrict__ idx, int64_t cnt,
                                                        double value) {
    const int64_t t = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (t < cnt) x[idx[t]] = value;
}

int stage_ensure(pf_graph* g, int64_t elems) {
    if (elems <= g->stage_cap) return PF_OK;
    pf_free(g->ctx->stream, g->stage);
    g->stage = nullptr;
    g->stage_cap = 0;
    PF_HIP(pf_malloc(g->ctx->stream, (void**)&g->stage, sizeof(double) * (size_t)elems));
    g->stage_cap = elems;
    return PF_OK;
}

int check_slots(pf_graph* g, int32_t first, int32_t count, const char* who) {
    PF_CHECK(g != nullptr, PF_E_ARG, "%s: graph is NULL", who);
    PF_HIP(hipSetDevice(g->ctx->device));  // the calling host thread may have another current device
    PF_CHECK(first >= 0 && count >= 0 && first + count <= g->n_slots, PF_E_ARG, "%s: slots [%d,%d) outside workspace of %d",
             who, first, first + count, g->n_slots);
    return PF_OK;
}

const double* op_values(pf_graph* g, int32_t op) {
    if (op == PF_OP_RW) return g->sval_rw;
    if (op == PF_OP_SYM) return g->sval_sym;
    return nullptr;
}

OpArgs op_args(pf_graph* g, const double* vals, const double* x, const double* prev, double* out, double alpha, double shift,
               double beta) {
    return OpArgs{g->slice_ptr, g->scol, vals, g->diag, x, prev, out, alpha, shift, beta, (unsigned)(g->n_pad / PF_OP_BLOCK)};
}

int64_t op_bytes(const pf_graph* g) {  // 12 nnz + 20 n + 4, nnz counted with the diagonal
    return 12 * (g->nnz_w + g->n - g->n_isolated) + 20 * g->n + 4;
}

int launch_op(pf_graph* g, const double* vals, const double* x, const double* prev, double* out, double alpha, double shift,
              double beta) {
    hipStream_t st = g->ctx->stream;
    const OpArgs a = op_args(g, vals, x, prev, out, alpha, shift, beta);
    if (prev)
        k_sell_op<true><<<a.n_blocks, PF_OP_BLOCK, 0, st>>>(a);
    else
        k_sell_op<false><<<a.n_blocks, PF_OP_BLOCK, 0, st>>>(a);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

int launch_op2(pf_graph* ga, const OpArgs& a, const OpArgs& b, bool has_prev) {
    hipStream_t st = ga->ctx->stream;
    if (has_prev)
        k_sell_op2<true><<<a.n_blocks + b.n_blocks, PF_OP_BLOCK, 0, st>>>(a, b);
    else
        k_sell_op2<false><<<a.n_blocks + b.n_blocks, PF_OP_BLOCK, 0, st>>>(a, b);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

// One Chebyshev recurrence: y0 = src; y1 = (c y0 - A y0)/(e rho); y_{k+1} = (2/(e rho))(c y_k - A y_k) - y_{k-1}/rho^2
// (rho > 1: scaled by rho^-k, so the result is T_p(.)/rho^p and high degrees cannot overflow).  Steps run one per
// launch (sell_op_block) through rotating temporaries; the caller's src is never overwritten and the last step
// lands in dst.
struct ChebRun {
    pf_graph* g;
    const double* vals;
    const double* src;
    double* dst;
    int32_t degree;
    double c, e, rho;
    const double* yp = nullptr;  // y_{done-1}
    const double* yc = nullptr;  // y_done
    int32_t done = 0;

    int32_t left() const { return degree - done; }
    double* free_tmp(int which) const {  // the which-th temporary that holds neither y_{done-1} nor y_done
        for (int t = 0; t < PF_WS_TMPS; ++t) {
            double* b = pf_tmp(g, t);
            if (b != yp && b != yc && which-- == 0) return b;
        }
        return nullptr;
    }
    OpArgs first() {  // step 1
        double* target = degree == 1 ? dst : pf_tmp(g, 0);
        const OpArgs a = op_args(g, vals, src, nullptr, target, 1.0 / (e * rho), c, 0.0);
        yp = src, yc = target, done = 1;
        return a;
    }
    OpArgs single() {  // one step k > 1
        double* target = left() == 1 ? dst : free_tmp(0);
        const OpArgs a = op_args(g, vals, yc, yp, target, 2.0 / (e * rho), c, 1.0 / (rho * rho));
        yp = yc, yc = target, done += 1;
        return a;
    }
};

struct OpTimer {
    pf_ctx* c;
    int64_t launches;
    double bytes;
    bool on;
    int64_t persist_steps = 0;
    double lds_bytes = 0.0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    OpTimer(pf_ctx* ctx, int64_t n, double b) : c(ctx), launches(n), bytes(b), on(ctx->timing) {
        if (on && ctx->timing_stride > 1) on = (ctx->timing_count++ % ctx->timing_stride) == 0;
        if (!on) return;
        if (c->spans_pending.size() >= 4096) (void)pf_timing_collect(c);  // bound the number of live events
        if (!c->spans_free.empty()) {
            e0 = c->spans_free.back().first;
            e1 = c->spans_free.back().second;
            c->spans_free.pop_back();
        } else if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
            on = false;
            return;
        }
        hipEventRecord(e0, c->stream);
    }
    int finish() {  // never blocks: the span is resolved in pf_timing_get
        if (!on) return PF_OK;
        PF_HIP(hipEventRecord(e1, c->stream));
        c->spans_pending.push_back({e0, e1, launches, bytes, persist_steps, lds_bytes});
        return PF_OK;
    }
};

// `count` doubles of device memory to the host behind whatever is queued on the ctx stream, and wait for them: through
// pinned memory and a copy kernel (pf_copy_by_kernel) - a copy COMMAND may queue behind the partner graph's eigenvector
// download on a DMA engine
int small_to_host(pf_graph* g, const double* d_src, double* out, size_t count) {
    pf_ctx* c = g->ctx;
    hipStream_t st = c->stream;
    const size_t bytes = sizeof(double) * count;
    void* pin = nullptr;
    if (bytes <= ((size_t)1 << 16) && pf_pinned_scratch(c, bytes, &pin) == PF_OK) {
        PF_TRY(pf_copy_by_kernel(st, d_src, pin, bytes));
        PF_HIP(hipStreamSynchronize(st));
        memcpy(out, pin, bytes);
        return PF_OK;
    }
    PF_HIP(hipMemcpyAsync(out, d_src, bytes, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    return PF_OK;
}

int dots_device(pf_graph* g, int32_t w, int32_t first, int32_t count, double* d_out, double* d_acc, int accumulate) {
    hipStream_t st = g->ctx->stream;
    PF_TRY(pf_reduce_ensure(g, count));
    dim3 grid((unsigned)g->n_chunks, (unsigned)count);
    k_dot_partial<<<grid, PF_BLOCK, 0, st>>>(g->ws, g->n_pad, first, w, g->n_chunks, g->partials);
    PF_HIP(hipGetLastError());
    k_dot_finish<<<(unsigned)count, PF_WAVE, 0, st>>>(g->partials, g->n_chunks, d_out, d_acc, accumulate);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

}  // namespace

int pf_reduce_ensure(pf_graph* g, int32_t count) {
    if (count > g->partial_cap) {
        pf_free(g->ctx->stream, g->partials);
        g->partials = nullptr;
        const int32_t cap = std::max(count, 64);
        // sized for VecStats partials (48 B) as well as plain doubles
        PF_HIP(pf_malloc(g->ctx->stream, (void**)&g->partials, sizeof(VecStats) * (size_t)cap * (size_t)(g->n_chunks + 1)));
        g->partial_cap = cap;
    }
    if (count > g->coef_cap) {
        pf_free(g->ctx->stream, g->coef);
        g->coef = nullptr;
        const int32_t cap = std::max(count, 64);
        PF_HIP(pf_malloc(g->ctx->stream, (void**)&g->coef, sizeof(double) * 8 * (size_t)cap));
        g->coef_cap = cap;
    }
    return PF_OK;
}

extern "C" {

int pf_ws_ensure(pf_graph* g, int32_t n_slots) {
    PF_CHECK(g != nullptr && n_slots > 0, PF_E_ARG, "pf_ws_ensure: bad argument");
    if (n_slots <= g->n_slots) return PF_OK;
    PF_HIP(hipSetDevice(g->ctx->device));
    hipStream_t st = g->ctx->stream;
    double* nw = nullptr;
    const size_t bytes = sizeof(double) * (size_t)(n_slots + PF_WS_TMPS) * (size_t)g->n_pad;
    PF_HIP(pf_malloc(st, (void**)&nw, bytes));
    PF_HIP(hipMemsetAsync(nw, 0, bytes, st));
    if (g->ws && g->n_slots > 0)
        PF_HIP(hipMemcpyAsync(nw, g->ws, sizeof(double) * (size_t)g->n_slots * g->n_pad, hipMemcpyDeviceToDevice, st));
    pf_free(st, g->ws);
    g->ws = nw;
    g->n_slots = n_slots;
    return PF_OK;
}

int pf_ws_upload(pf_graph* g, int32_t slot, const double* x) {
    PF_TRY(check_slots(g, slot, 1, "pf_ws_upload"));
    PF_CHECK(x != nullptr, PF_E_ARG, "pf_ws_upload: x is NULL");
    hipStream_t st = g->ctx->stream;
    PF_TRY(stage_ensure(g, g->n));
    PF_HIP(hipMemcpyAsync(g->stage, x, sizeof(double) * g->n, hipMemcpyHostToDevice, st));
    k_permute_in<<<nblk(g->n_pad), PF_BLOCK, 0, st>>>(g->stage, g->perm, g->n_pad, pf_slot(g, slot));
    PF_HIP(hipGetLastError());
    PF_HIP(hipStreamSynchronize(st));  // x may be a temporary on the host side
    return PF_OK;
}

int pf_ws_download(pf_graph* g, int32_t first, int32_t count, double* out) {
    PF_TRY(check_slots(g, first, count, "pf_ws_download"));
    PF_CHECK(out != nullptr, PF_E_ARG, "pf_ws_download: out is NULL");
    if (count == 0) return PF_OK;
    hipStream_t st = g->ctx->stream;
    PF_TRY(stage_ensure(g, (int64_t)count * g->n));
    k_permute_out<<<nblk(g->n), PF_BLOCK, 0, st>>>(g->ws, g->n_pad, first, count, g->iperm, g->n, g->stage);
    PF_HIP(hipGetLastError());
    PF_HIP(hipMemcpyAsync(out, g->stage, sizeof(double) * (size_t)count * g->n, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    return pf_persist_check(g->ctx);
}

int pf_ws_copy(pf_graph* g, int32_t src, int32_t dst, int32_t count) {
    PF_TRY(check_slots(g, src, count, "pf_ws_copy"));
    PF_TRY(check_slots(g, dst, count, "pf_ws_copy"));
    PF_CHECK(src + count <= dst || dst + count <= src || src == dst, PF_E_ARG, "pf_ws_copy: overlapping ranges");
    if (src == dst || count == 0) return PF_OK;
    PF_HIP(hipMemcpyAsync(pf_slot(g, dst), pf_slot(g, src), sizeof(double) * (size_t)count * g->n_pad,
                          hipMemcpyDeviceToDevice, g->ctx->stream));
    return PF_OK;
}

int pf_mask_isolated(pf_graph* g, int32_t slot) {
    PF_TRY(check_slots(g, slot, 1, "pf_mask_isolated"));
    if (g->n_isolated == 0) return PF_OK;
    k_mask_isolated<<<nblk(g->n), PF_BLOCK, 0, g->ctx->stream>>>(pf_slot(g, slot), g->rowptr, g->perm_m, g->n);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

int pf_start_vector(pf_graph* g, int32_t slot, uint64_t seed) {
    PF_TRY(check_slots(g, slot, 1, "pf_start_vector"));
    k_start_vector<<<nblk(g->n_pad), PF_BLOCK, 0, g->ctx->stream>>>(pf_slot(g, slot), g->smooth, g->perm, g->perm_m, g->rowptr, g->n_pad,
                                                                    (unsigned long long)seed, 0.5);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

int pf_lock_null_vectors(pf_graph* g, int32_t op, int32_t* n_locked) {
    PF_CHECK(g != nullptr && n_locked != nullptr, PF_E_ARG, "pf_lock_null_vectors: NULL argument");
    PF_HIP(hipSetDevice(g->ctx->device));
    PF_CHECK(op == PF_OP_RW || (op == PF_OP_SYM && g->is_symmetric), PF_E_ARG,
             "pf_lock_null_vectors: operator %d not available (W symmetric: %d)", op, g->is_symmetric);
    const int32_t nc = g->n_components;
    PF_TRY(pf_ws_ensure(g, nc + 1));
    hipStream_t st = g->ctx->stream;
    PF_TRY(pf_reduce_ensure(g, 1));
    for (int32_t c = 0; c < nc; ++c) {
        k_null_vector<<<nblk(g->n_pad), PF_BLOCK, 0, st>>>(pf_slot(g, c), g->label, g->rowptr, g->deg, g->perm_m, g->n_pad,
                                                          g->roots[c], op == PF_OP_SYM && !g->unit_g);
        PF_HIP(hipGetLastError());
        // normalised with the norm still on the device (the same 1 / sqrt as on the host: the same bits; a component has at
        // least two vertices, so the norm is positive) - no wait at the head of a solve
        PF_TRY(dots_device(g, c, c, 1, g->coef, nullptr, 0));
        k_scale_rsqrt<<<nblk(g->n_pad), PF_BLOCK, 0, st>>>(pf_slot(g, c), g->n_pad, g->coef);
        PF_HIP(hipGetLastError());
    }
    *n_locked = nc;
    return PF_OK;
}

int pf_spmv(pf_graph* g, int32_t op, int32_t src, int32_t dst) {
    PF_TRY(check_slots(g, src, 1, "pf_spmv"));
    PF_TRY(check_slots(g, dst, 1, "pf_spmv"));
    const double* vals = op_values(g, op);
    PF_CHECK(vals != nullptr && src != dst, PF_E_ARG, "pf_spmv: operator %d unavailable or src == dst", op);
    OpTimer t(g->ctx, 1, (double)op_bytes(g));
    PF_TRY(launch_op(g, vals, pf_slot(g, src), nullptr, pf_slot(g, dst), -1.0, 0.0, 0.0));
    return t.finish();
}

int pf_spmv_multi(pf_graph* g, int32_t op, int32_t src_first, int32_t dst_first, int32_t count) {
    PF_CHECK(count >= 0, PF_E_ARG, "pf_spmv_multi: negative count");
    PF_TRY(check_slots(g, src_first, count, "pf_spmv_multi"));
    PF_TRY(check_slots(g, dst_first, count, "pf_spmv_multi"));
    const double* vals = op_values(g, op);
    PF_CHECK(vals != nullptr && (src_first + count <= dst_first || dst_first + count <= src_first), PF_E_ARG,
             "pf_spmv_multi: operator %d unavailable or the slot ranges overlap", op);
    if (count == 0) return PF_OK;
    OpTimer t(g->ctx, count, (double)count * (double)op_bytes(g));
    {
        const OpArgs a = op_args(g, vals, pf_slot(g, src_first), nullptr, pf_slot(g, dst_first), -1.0, 0.0, 0.0);
        k_sell_op_slots<<<dim3(a.n_blocks, (unsigned)count), PF_OP_BLOCK, 0, g->ctx->stream>>>(a, g->n_pad);
        PF_HIP(hipGetLastError());
    }
    return t.finish();
}

int pf_cheb(pf_graph* g, int32_t op, int32_t src, int32_t dst, int32_t degree, double c, double e, double rho) {
    PF_TRY(check_slots(g, src, 1, "pf_cheb"));
    PF_TRY(check_slots(g, dst, 1, "pf_cheb"));
    const double* vals = op_values(g, op);
    PF_CHECK(vals != nullptr && src != dst, PF_E_ARG, "pf_cheb: operator %d unavailable or src == dst", op);
    PF_CHECK(degree >= 1 && e > 0.0 && rho >= 1.0, PF_E_ARG, "pf_cheb: degree %d / half-width %g / rho %g invalid", degree, e, rho);
    OpTimer t(g->ctx, degree, (double)degree * op_bytes(g));
    {  // the whole recurrence in one kernel with the operator in LDS, when it fits (pf_persist.hip)
        const pf_persist_args pa{g, vals, pf_slot(g, src), pf_slot(g, dst), degree, c, e, rho};
        int done = 0;
        PF_TRY(pf_persist_cheb(&pa, nullptr, &done, &t.lds_bytes));
        if (done) {
            t.launches = 1;
            t.persist_steps = degree;
            return t.finish();
        }
    }
    ChebRun r{g, vals, pf_slot(g, src), pf_slot(g, dst), degree, c, e, rho};
    int64_t launches = 1;
    {
        const OpArgs a = r.first();
        PF_TRY(launch_op(g, a.sval, a.x, a.prev, a.out, a.alpha, a.shift, a.beta));
    }
    while (r.left() > 0) {
        const OpArgs a = r.single();
        PF_TRY(launch_op(g, a.sval, a.x, a.prev, a.out, a.alpha, a.shift, a.beta));
        ++launches;
    }
    t.launches = launches;
    return t.finish();
}

int pf_cheb2(pf_graph* ga, int32_t op_a, int32_t src_a, int32_t dst_a, int32_t degree_a, double c_a, double e_a, double rho_a,
             pf_graph* gb, int32_t op_b, int32_t src_b, int32_t dst_b, int32_t degree_b, double c_b, double e_b, double rho_b) {
    PF_TRY(check_slots(ga, src_a, 1, "pf_cheb2"));
    PF_TRY(check_slots(ga, dst_a, 1, "pf_cheb2"));
    PF_TRY(check_slots(gb, src_b, 1, "pf_cheb2"));
    PF_TRY(check_slots(gb, dst_b, 1, "pf_cheb2"));
    PF_CHECK(ga != gb && ga->ctx == gb->ctx, PF_E_ARG, "pf_cheb2: the two graphs must differ and share one ctx (stream)");
    const double* va = op_values(ga, op_a);
    const double* vb = op_values(gb, op_b);
    PF_CHECK(va && vb && src_a != dst_a && src_b != dst_b, PF_E_ARG, "pf_cheb2: operator unavailable or src == dst");
    PF_CHECK(degree_a >= 1 && degree_b >= 1 && e_a > 0.0 && e_b > 0.0 && rho_a >= 1.0 && rho_b >= 1.0, PF_E_ARG,
             "pf_cheb2: bad degree / half-width / rho");
    OpTimer t(ga->ctx, std::max(degree_a, degree_b), (double)degree_a * op_bytes(ga) + (double)degree_b * op_bytes(gb));
    {
        const pf_persist_args pa{ga, va, pf_slot(ga, src_a), pf_slot(ga, dst_a), degree_a, c_a, e_a, rho_a};
        const pf_persist_args pb{gb, vb, pf_slot(gb, src_b), pf_slot(gb, dst_b), degree_b, c_b, e_b, rho_b};
        int done = 0;
        PF_TRY(pf_persist_cheb(&pa, &pb, &done, &t.lds_bytes));
        if (done) {
            t.launches = 1;
            t.persist_steps = degree_a + degree_b;
            return t.finish();
        }
        // The pair does not fit one resident launch (windows of 2048+ rows: registers / LDS): each graph in its own
        // resident launch still beats one step per launch by far (1M rows: 2 x 4 us per step against 30 us shared).
        int done_a = 0, done_b = 0;
        PF_TRY(pf_persist_cheb(&pa, nullptr, &done_a, &t.lds_bytes, false));
        PF_TRY(pf_persist_cheb(&pb, nullptr, &done_b, &t.lds_bytes, false));
        if (done_a || done_b) {
            t.launches = done_a + done_b;
            t.persist_steps = (done_a ? degree_a : 0) + (done_b ? degree_b : 0);
            for (int q = 0; q < 2; ++q) {
                if (q == 0 ? done_a : done_b) continue;
                ChebRun r = q == 0 ? ChebRun{ga, va, pf_slot(ga, src_a), pf_slot(ga, dst_a), degree_a, c_a, e_a, rho_a}
                                   : ChebRun{gb, vb, pf_slot(gb, src_b), pf_slot(gb, dst_b), degree_b, c_b, e_b, rho_b};
                OpArgs a = r.first();
                PF_TRY(launch_op(r.g, a.sval, a.x, a.prev, a.out, a.alpha, a.shift, a.beta));
                ++t.launches;
                while (r.left() > 0) {
                    a = r.single();
                    PF_TRY(launch_op(r.g, a.sval, a.x, a.prev, a.out, a.alpha, a.shift, a.beta));
                    ++t.launches;
                }
            }
            return t.finish();
        }
    }
    ChebRun ra{ga, va, pf_slot(ga, src_a), pf_slot(ga, dst_a), degree_a, c_a, e_a, rho_a};
    ChebRun rb{gb, vb, pf_slot(gb, src_b), pf_slot(gb, dst_b), degree_b, c_b, e_b, rho_b};
    int64_t launches = 1;
    {
        const OpArgs a = ra.first(), b = rb.first();
        PF_TRY(launch_op2(ga, a, b, false));
    }
    while (ra.left() > 0 || rb.left() > 0) {
        if (ra.left() > 0 && rb.left() > 0) {
            const OpArgs a = ra.single(), b = rb.single();
            PF_TRY(launch_op2(ga, a, b, true));
        } else {  // the longer recurrence finishes alone
            ChebRun& r = ra.left() > 0 ? ra : rb;
            const OpArgs a = r.single();
            PF_TRY(launch_op(r.g, a.sval, a.x, a.prev, a.out, a.alpha, a.shift, a.beta));
        }
        ++launches;
    }
    t.launches = launches;
    return t.finish();
}

// One outer step of a pipelined pair driver in ONE library call: the Gram-Schmidt step of both graphs (pf_orth_begin2) and,
// queued right behind it, the next filter application of both (pf_cheb2) - the device then never waits for the host
// between the two (two calls from Python leave it idle for ~15-25 us per step, a tenth of the filter application).
//   orth[8]   = {w, first, count, normalize} of graph a, then of graph b
//   cheb_i[8] = {op, src, dst, degree} of graph a, then of graph b;   cheb_d[6] = {c, e, rho} of a, then of b
int pf_orth_cheb2(pf_graph* ga, pf_graph* gb, const int32_t* orth, const int32_t* cheb_i, const double* cheb_d) {
    PF_CHECK(ga && gb && orth && cheb_i && cheb_d, PF_E_ARG, "pf_orth_cheb2: NULL argument");
    PF_TRY(pf_orth_begin2(ga, orth[0], orth[1], orth[2], orth[3], gb, orth[4], orth[5], orth[6], orth[7]));
    return pf_cheb2(ga, cheb_i[0], cheb_i[1], cheb_i[2], cheb_i[3], cheb_d[0], cheb_d[1], cheb_d[2], gb, cheb_i[4], cheb_i[5], cheb_i[6],
                    cheb_i[7], cheb_d[3], cheb_d[4], cheb_d[5]);
}

int pf_dots(pf_graph* g, int32_t w, int32_t first, int32_t count, double* out) {
    PF_TRY(check_slots(g, w, 1, "pf_dots"));
    PF_TRY(check_slots(g, first, count, "pf_dots"));
    PF_CHECK(out != nullptr, PF_E_ARG, "pf_dots: out is NULL");
    if (count == 0) return PF_OK;
    PF_TRY(pf_reduce_ensure(g, count));  // g->coef must exist before its address is taken
    PF_TRY(dots_device(g, w, first, count, g->coef, nullptr, 0));
    return small_to_host(g, g->coef, out, (size_t)count);
}

// the two launches of one Gram-Schmidt pass (of one graph, or of both graphs of a pair: ng = 2); bases beyond the Infinity
// Cache (>= 400k rows: 1M x 50 vectors = 400 MB) take the shapes with several vectors / a whole chunk per block
// blocks of k_orth_local the device holds at once (0: the one-launch step is switched off - PF_ORTH_LOCAL=0 - or unknown)
// pf_orth_one_launch: -1 as the environment says (PF_ORTH_LOCAL; on by default), 0 the separate launches, 1 on
static std::atomic<int> g_orth_one_launch{-1};

static int64_t orth_local_capacity(int device) {
    static std::mutex m;
    static std::map<int, int64_t> cap;
    std::lock_guard<std::mutex> lk(m);
    auto it = cap.find(device);
    if (it != cap.end()) return it->second;
    int64_t c = 0;
    const char* e = getenv("PF_ORTH_LOCAL");
    if (!(e && e[0] == '0')) {
        int per_cu = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_orth_local), PF_BLOCK, 0) == hipSuccess &&
            hipGetDeviceProperties(&prop, device) == hipSuccess)
            c = (int64_t)per_cu * prop.multiProcessorCount;
        else
            (void)hipGetLastError();
    }
    cap[device] = c;
    return c;
}

// the counter of the meeting inside k_orth_local: `arrivals` more per launch
static int orth_meetings_prepare(pf_graph* g, OrthArgs& a, unsigned long long arrivals, hipStream_t st) {
    pf_ctx* ctx = g->ctx;
    const uint64_t epoch = pf_persist_abort_epoch();
    if (!g->orth_counter) PF_HIP(pf_malloc(st, (void**)&g->orth_counter, sizeof(unsigned long long)));
    if (g->orth_epoch != epoch) {  // new, or a bounded wait ran out since: whatever had arrived then is void
        PF_HIP(hipMemsetAsync(g->orth_counter, 0, sizeof(unsigned long long), st));
        g->orth_arrivals = 0;
        g->orth_epoch = epoch;
    }
    g->orth_arrivals += arrivals;
    a.counter = g->orth_counter;
    a.target = g->orth_arrivals;
    a.abort_flag = ctx->persist_sync;
    a.host_abort = ctx->persist_abort;
    return PF_OK;
}

static int orth_launch_pass(OrthArgs2& a2, int ng, int64_t n_chunks, int64_t n_pad, int32_t count, hipStream_t st, pf_graph* const* gs = nullptr) {
    // a local step (4 vectors at most, one pass) of every graph in the launch: the whole step in ONE kernel
    bool local = gs != nullptr && count <= PF_ORTH_LOCAL_MAX && count >= 1;
    for (int q = 0; q < ng && local; ++q) local = a2.g[q].pass == 0 && a2.g[q].count <= PF_ORTH_LOCAL_MAX && a2.g[q].count >= 1;
    if (local && g_orth_one_launch.load() != 0 && pf_persist_trusted() && n_chunks * ng <= orth_local_capacity(gs[0]->ctx->device)) {
        PF_TRY(pf_persist_sync_ensure(gs[0]->ctx));
        for (int q = 0; q < ng; ++q) PF_TRY(orth_meetings_prepare(gs[q], a2.g[q], (unsigned long long)gs[q]->n_chunks, st));
        k_orth_local<<<dim3((unsigned)n_chunks, 1u, (unsigned)ng), PF_BLOCK, 0, st>>>(a2);
        PF_HIP(hipGetLastError());
        return PF_OK;
    }
    if (n_pad >= 400000) {
        k_orth_dots<4><<<dim3((unsigned)n_chunks, (unsigned)((count + 1 + 3) / 4), (unsigned)ng), PF_BLOCK, 0, st>>>(a2);
        PF_HIP(hipGetLastError());
        k_orth_project<8><<<dim3((unsigned)(n_pad / (2 * PF_BLOCK * 8)), 1u, (unsigned)ng), PF_BLOCK, 0, st>>>(a2);
    } else {
        k_orth_dots<1><<<dim3((unsigned)n_chunks, (unsigned)(count + 1), (unsigned)ng), PF_BLOCK, 0, st>>>(a2);
        PF_HIP(hipGetLastError());
        k_orth_project<1><<<dim3(nblk(n_pad / 2), 1u, (unsigned)ng), PF_BLOCK, 0, st>>>(a2);
    }
    PF_HIP(hipGetLastError());
    return PF_OK;
}

// checks, pinned result buffer, event: everything of pf_orth_begin that comes before the launches
static int orth_prepare(pf_graph* g, int32_t w, int32_t first, int32_t count, int32_t normalize) {
    // pf_orth_split (this step only, whatever becomes of it): the basis is slots [first, first + split) and
    // [first2, first2 + count - split)
    const int32_t split_req = g->orth_split, first2_req = g->orth_first2;
    g->orth_split = -1;
    PF_TRY(check_slots(g, w, 1, "pf_orth_begin"));
    if (split_req < 0) PF_TRY(check_slots(g, first, count, "pf_orth_begin"));
    PF_CHECK(g->orth_pending < 0, PF_E_STATE, "pf_orth_begin: a previous pf_orth_begin has not been collected");
    g->orth_split_now = -1;
    if (split_req >= 0) {
        const int32_t split = split_req, first2 = first2_req;
        PF_CHECK(split <= count && count > 0 && count < PF_ORTH_MAX, PF_E_ARG, "pf_orth_split: %d of %d vectors in the first range", split, count);
        PF_TRY(check_slots(g, first, split, "pf_orth_begin"));
        PF_TRY(check_slots(g, first2, count - split, "pf_orth_begin"));
        PF_CHECK((w < first || w >= first + split) && (w < first2 || w >= first2 + count - split), PF_E_ARG, "pf_orth_begin: w inside the basis ranges");
        PF_CHECK(first + split <= first2 || first2 + count - split <= first, PF_E_ARG, "pf_orth_split: overlapping ranges");
        g->orth_split_now = split;
        g->orth_first2_now = first2;
    } else {
        PF_CHECK(w < first || w >= first + count, PF_E_ARG, "pf_orth_begin: w inside the basis range");
    }
    hipStream_t st = g->ctx->stream;
    PF_TRY(pf_reduce_ensure(g, count + 1));  // (+ the |w|^2 column)
    if (count + 1 > g->orth_host_cap || !g->orth_host) {
        pf_ctx* c = g->ctx;
        if (g->orth_host) {
            PF_HIP(hipStreamSynchronize(st));
            c->pinned_pool.emplace_back(g->orth_host_cap, g->orth_host);
            g->orth_host = nullptr;
        }
        const int32_t cap = std::max(count + 1, 64);  // (h, |w'|^2, verdict, ticket: count + 3 doubles of cap + 2)
        for (size_t i = 0; i < c->pinned_pool.size(); ++i)
            if (c->pinned_pool[i].first >= cap) {  // a buffer a freed graph left behind
                g->orth_host_cap = c->pinned_pool[i].first;
                g->orth_host = c->pinned_pool[i].second;
                c->pinned_pool.erase(c->pinned_pool.begin() + (long)i);
                break;
            }
        if (!g->orth_host) {
            PF_HIP(hipHostMalloc((void**)&g->orth_host, sizeof(double) * (size_t)(cap + 2), hipHostMallocDefault));
            g->orth_host_cap = cap;
        }
    }
    if (!g->orth_ev) {
        pf_ctx* c = g->ctx;
        if (!c->event_pool.empty()) {
            g->orth_ev = c->event_pool.back();
            c->event_pool.pop_back();
        } else {
            PF_HIP(hipEventCreateWithFlags(&g->orth_ev, hipEventDisableTiming));
        }
    }
    g->orth_host[count + 1] = 0.0;     // the verdict slot; only k_orth_project ever raises it
    g->orth_host[count + 2] = 0.0;     // the ticket slot
    g->orth_ticket = 0.0;              // (set by the launch paths whose kernel writes a ticket)
    g->orth_w = w, g->orth_first = first, g->orth_normalize = normalize ? 1 : 0;
    return PF_OK;
}

static OrthArgs orth_args(pf_graph* g, int32_t w, int32_t first, int32_t count, int32_t normalize) {
    OrthArgs a{};
    a.ws = g->ws;
    a.n_pad = g->n_pad;
    a.n_chunks = g->n_chunks;
    a.first = first, a.count = count, a.wslot = w, a.normalize = normalize ? 1 : 0;
    a.split = g->orth_split_now >= 0 ? g->orth_split_now : count;
    a.first2 = g->orth_first2_now;
    a.partial = g->partials;
    a.hsum = g->coef + g->coef_cap;
    a.nrm2 = g->coef + 2 * g->coef_cap;
    a.host_out = g->orth_host;
    a.thresh = g->orth_thresh;
    g->orth_serial += 1.0;
    a.ticket = g->orth_serial;
    g->orth_ticket = a.ticket;  // pf_orth_end polls for it
    a.pass = g->orth_device_passes ? 1 : 0;
    a.verdict = g->coef + 3 * g->coef_cap;
    return a;
}

int pf_orth_begin(pf_graph* g, int32_t w, int32_t first, int32_t count, int32_t normalize) {
    PF_TRY(orth_prepare(g, w, first, count, normalize));
    hipStream_t st = g->ctx->stream;
    const int32_t cap = g->coef_cap;
    double* hpass = g->coef;           // coefficients of the current pass
    double* hsum = g->coef + cap;      // h1 + h2
    double* nrm2 = g->coef + 2 * cap;  // ||w||^2
    if (count > 0 && count < PF_ORTH_MAX) {
        // 2 launches and no copies: the projection, its norm (Pythagoras) and the normalisation ride behind one batch of
        // dot products; the second Gram-Schmidt pass is pf_orth_end's business in the rare step that needs it
        OrthArgs2 a2{};
        a2.g[0] = orth_args(g, w, first, count, normalize);
        pf_graph* const one[1] = {g};
        PF_TRY(orth_launch_pass(a2, 1, g->n_chunks, g->n_pad, count, st, one));
        if (a2.g[0].pass == 1) {  // the second pass, on the device's own verdict
            a2.g[0].pass = 2;
            PF_TRY(orth_launch_pass(a2, 1, g->n_chunks, g->n_pad, count, st));
        }
    } else if (count == 0) {
        k_dot_partial<<<dim3((unsigned)g->n_chunks, 1u), PF_BLOCK, 0, st>>>(g->ws, g->n_pad, w, w, g->n_chunks, g->partials);
        PF_HIP(hipGetLastError());
        k_scale_finishing<<<nblk(g->n_pad), PF_BLOCK, 0, st>>>(pf_slot(g, w), g->n_pad, g->partials, g->n_chunks, normalize ? 1 : 0,
                                                              hsum, count, nrm2, g->orth_host);
        PF_HIP(hipGetLastError());
    } else {
        for (int pass = 0; pass < 2; ++pass) {
            PF_TRY(dots_device(g, w, first, count, hpass, hsum, pass));
            k_multi_axpy<<<nblk(g->n_pad / 2), PF_BLOCK, 0, st>>>(g->ws, g->n_pad, first, count, w, hpass);
            PF_HIP(hipGetLastError());
        }
        PF_TRY(dots_device(g, w, w, 1, nrm2, nullptr, 0));
        if (normalize) {
            k_scale_rsqrt<<<nblk(g->n_pad), PF_BLOCK, 0, st>>>(pf_slot(g, w), g->n_pad, nrm2);
            PF_HIP(hipGetLastError());
        }
        PF_HIP(hipMemcpyAsync(g->orth_host, hsum, sizeof(double) * count, hipMemcpyDeviceToHost, st));
        PF_HIP(hipMemcpyAsync(g->orth_host + count, nrm2, sizeof(double), hipMemcpyDeviceToHost, st));
    }
    if (g->orth_ticket == 0.0) PF_HIP(hipEventRecord(g->orth_ev, st));  // (a ticketed step is polled for, not waited on)
    g->orth_wait = g->orth_ev;
    g->orth_pending = count;
    return PF_OK;
}

// pf_orth_begin for the two graphs of a pair (one ctx) behind the same two launches; each graph's result is collected
// with its own pf_orth_end.  Shapes the fused kernels do not cover take one pf_orth_begin each.
int pf_orth_begin2(pf_graph* ga, int32_t w_a, int32_t first_a, int32_t count_a, int32_t normalize_a, pf_graph* gb, int32_t w_b,
                   int32_t first_b, int32_t count_b, int32_t normalize_b) {
    PF_CHECK(ga != nullptr && gb != nullptr && ga != gb, PF_E_ARG, "pf_orth_begin2: two different graphs are needed");
    if (ga->ctx != gb->ctx || count_a <= 0 || count_b <= 0 || count_a >= PF_ORTH_MAX || count_b >= PF_ORTH_MAX) {
        PF_TRY(pf_orth_begin(ga, w_a, first_a, count_a, normalize_a));
        return pf_orth_begin(gb, w_b, first_b, count_b, normalize_b);
    }
    PF_TRY(orth_prepare(ga, w_a, first_a, count_a, normalize_a));
    PF_TRY(orth_prepare(gb, w_b, first_b, count_b, normalize_b));
    hipStream_t st = ga->ctx->stream;
    OrthArgs2 a2{};
    a2.g[0] = orth_args(ga, w_a, first_a, count_a, normalize_a);
    a2.g[1] = orth_args(gb, w_b, first_b, count_b, normalize_b);
    const int64_t chunks2 = std::max(ga->n_chunks, gb->n_chunks), pad2 = std::max(ga->n_pad, gb->n_pad);
    const int32_t count2 = std::max(count_a, count_b);
    pf_graph* const two[2] = {ga, gb};
    PF_TRY(orth_launch_pass(a2, 2, chunks2, pad2, count2, st, two));
    if (a2.g[0].pass == 1 || a2.g[1].pass == 1) {  // the second pass of the graph(s) that asked for it, on the device's own verdict
        for (int q = 0; q < 2; ++q) a2.g[q].pass = a2.g[q].pass == 1 ? 2 : -1;
        PF_TRY(orth_launch_pass(a2, 2, chunks2, pad2, count2, st));
    }
    // (no event: both results carry tickets, pf_orth_end polls for them)
    ga->orth_wait = ga->orth_ev;
    gb->orth_wait = gb->orth_ev;
    ga->orth_pending = count_a;
    gb->orth_pending = count_b;
    return PF_OK;
}

int pf_orth_end(pf_graph* g, double* h, double* nrm) {
    PF_CHECK(g != nullptr && nrm != nullptr, PF_E_ARG, "pf_orth_end: NULL argument");
    PF_CHECK(g->orth_pending >= 0, PF_E_STATE, "pf_orth_end: no pf_orth_begin in flight");
    const int32_t count = g->orth_pending;
    PF_CHECK(h != nullptr || count == 0, PF_E_ARG, "pf_orth_end: h is NULL");
    PF_HIP(hipSetDevice(g->ctx->device));
    if (g->orth_ticket != 0.0) {
        // the kernel's block 0 writes the results and then the ticket into the pinned buffer: poll for it (the stream is asked
        // now and then whether it has run dry without a ticket - a failed launch must not hang the host)
        volatile double* ticket = g->orth_host + count + 2;
        bool got = false;
        for (uint64_t spin = 0;; ++spin) {
            if (__atomic_load_n(reinterpret_cast<volatile uint64_t*>(ticket), __ATOMIC_ACQUIRE) ==
                *reinterpret_cast<const uint64_t*>(&g->orth_ticket)) {
                got = true;
                break;
            }
            if ((spin & 0xfff) == 0xfff) {
                const hipError_t q = hipStreamQuery(g->ctx->stream);
                if (q == hipSuccess) {  // everything queued has completed: the ticket is there, or never will be
                    got = __atomic_load_n(reinterpret_cast<volatile uint64_t*>(ticket), __ATOMIC_ACQUIRE) ==
                          *reinterpret_cast<const uint64_t*>(&g->orth_ticket);
                    break;
                }
                if (q != hipErrorNotReady) {
                    g->orth_pending = -1;
                    PF_HIP(q);
                }
            }
            __builtin_ia32_pause();
        }
        if (!got) {
            g->orth_pending = -1;
            PF_TRY(pf_persist_check(g->ctx));
            PF_CHECK(false, PF_E_HIP, "pf_orth_end: the Gram-Schmidt step never reported (launch failure?)");
        }
    } else {
        PF_HIP(hipEventSynchronize(g->orth_wait));
    }
    g->orth_pending = -1;
    g->orth_redone = 0;
    PF_TRY(pf_persist_check(g->ctx));
    g->orth_twice = g->orth_host[count + 1] == 2.0 ? 1 : 0;  // (both passes ran on the device: nothing behind them read a stale w)
    if (g->orth_host[count + 1] == 1.0) {
        // the first pass cancelled digits (or w vanished): second pass, exact norm, normalisation - synchronously.
        // Whatever was queued behind pf_orth_begin read a w that is only now final: pf_orth_redone tells the caller.
        hipStream_t st = g->ctx->stream;
        double* hsum = g->coef + g->coef_cap;
        double* nrm2 = g->coef + 2 * g->coef_cap;
        const int32_t w = g->orth_w, first = g->orth_first;
        const int32_t split = g->orth_split_now >= 0 ? g->orth_split_now : count;  // (two ranges: all dot products first, as in one)
        if (split > 0) {
            k_dot_partial<<<dim3((unsigned)g->n_chunks, (unsigned)split), PF_BLOCK, 0, st>>>(g->ws, g->n_pad, first, w, g->n_chunks, g->partials);
            PF_HIP(hipGetLastError());
        }
        if (count > split) {
            k_dot_partial<<<dim3((unsigned)g->n_chunks, (unsigned)(count - split)), PF_BLOCK, 0, st>>>(g->ws, g->n_pad, g->orth_first2_now, w, g->n_chunks,
                                                                                                   g->partials + (size_t)split * g->n_chunks);
            PF_HIP(hipGetLastError());
        }
        if (split > 0) {
            k_axpy_finishing<<<nblk(g->n_pad / 2), PF_BLOCK, 0, st>>>(g->ws, g->n_pad, first, split, w, g->partials, g->n_chunks, hsum, 1);
            PF_HIP(hipGetLastError());
        }
        if (count > split) {
            k_axpy_finishing<<<nblk(g->n_pad / 2), PF_BLOCK, 0, st>>>(g->ws, g->n_pad, g->orth_first2_now, count - split, w,
                                                                      g->partials + (size_t)split * g->n_chunks, g->n_chunks, hsum + split, 1);
            PF_HIP(hipGetLastError());
        }
        k_dot_partial<<<dim3((unsigned)g->n_chunks, 1u), PF_BLOCK, 0, st>>>(g->ws, g->n_pad, w, w, g->n_chunks, g->partials);
        PF_HIP(hipGetLastError());
        k_scale_finishing<<<nblk(g->n_pad), PF_BLOCK, 0, st>>>(pf_slot(g, w), g->n_pad, g->partials, g->n_chunks, g->orth_normalize,
                                                              hsum, count, nrm2, g->orth_host);
        PF_HIP(hipGetLastError());
        PF_HIP(hipStreamSynchronize(st));
        PF_TRY(pf_persist_check(g->ctx));
        g->orth_redone = 1;
    }
    for (int32_t b = 0; b < count; ++b) h[b] = g->orth_host[b];
    const double v = g->orth_host[count];
    *nrm = sqrt(v > 0.0 ? v : 0.0);
    return PF_OK;
}

// The criterion for the second Gram-Schmidt pass: |w'| < 0.3 |w| by default; strict: |w'| < 0.71 |w| (the classical
// "twice is enough" constant).  The loose one is measured safe where it matters for speed - the filtered iteration of a
// large graph: ~35 steps, ratios 0.35-0.78 - and is NOT safe for an unfiltered iteration that comes close to exhausting a
// small space (a 120-vertex mesh, 47 steps: ratios around 0.3-0.5 step after step, orthogonality lost, Ritz values
// above the spectrum), where the second pass costs nothing that matters.  Drivers switch to strict there.
int pf_orth_strict(pf_graph* g, int32_t on) {
    PF_CHECK(g != nullptr, PF_E_ARG, "pf_orth_strict: NULL graph");
    g->orth_thresh = on >= 2 ? 4.0 : (on ? 0.5 : 0.09);  // (2: |w'|^2 < 4 |w|^2 holds always - every step takes its second pass)
    return PF_OK;
}

int pf_orth_device_passes(pf_graph* g, int32_t on) {
    PF_CHECK(g != nullptr, PF_E_ARG, "pf_orth_device_passes: NULL graph");
    g->orth_device_passes = on ? 1 : 0;
    return PF_OK;
}

// The NEXT pf_orth_begin / _begin2 / pf_orth_cheb2 step of this graph (and only that one) takes its basis from two ranges
// of slots: [first, first + split) and [first2, first2 + count - split), first / count as passed to that call.
int pf_orth_one_launch(int32_t on) {
    g_orth_one_launch.store(on < 0 ? -1 : (on ? 1 : 0));
    return PF_OK;
}

int pf_orth_split(pf_graph* g, int32_t first2, int32_t split) {
    PF_CHECK(g != nullptr && split >= 0 && first2 >= 0, PF_E_ARG, "pf_orth_split: bad argument");
    g->orth_split = split;
    g->orth_first2 = first2;
    return PF_OK;
}

int pf_orth_redone(pf_graph* g) {
    PF_CHECK(g != nullptr, PF_E_ARG, "pf_orth_redone: NULL graph");
    return g->orth_redone ? 1 : (g->orth_twice ? 2 : 0);
}

int pf_orth(pf_graph* g, int32_t w, int32_t first, int32_t count, double* h, double* nrm) {
    PF_CHECK(nrm != nullptr && (h != nullptr || count == 0), PF_E_ARG, "pf_orth: NULL output");
    PF_TRY(pf_orth_begin(g, w, first, count, 0));
    return pf_orth_end(g, h, nrm);
}

int pf_scale(pf_graph* g, int32_t slot, double alpha) {
    PF_TRY(check_slots(g, slot, 1, "pf_scale"));
    k_scale<<<nblk(g->n_pad), PF_BLOCK, 0, g->ctx->stream>>>(pf_slot(g, slot), g->n_pad, alpha);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

int pf_combine(pf_graph* g, int32_t src_first, int32_t m, const double* Y, int32_t k, int32_t dst_first) {
    return pf_combine2(g, src_first, m, Y, k, dst_first, -1, -1);
}

// dst = src Y and, when src_first2 >= 0, dst2 = src2 Y in the same launches (one upload of Y)
int pf_combine2(pf_graph* g, int32_t src_first, int32_t m, const double* Y, int32_t k, int32_t dst_first, int32_t src_first2,
                int32_t dst_first2) {
    PF_TRY(check_slots(g, src_first, m, "pf_combine"));
    PF_TRY(check_slots(g, dst_first, k, "pf_combine"));
    PF_CHECK(Y != nullptr && m > 0 && k > 0, PF_E_ARG, "pf_combine: bad argument");
    PF_CHECK(src_first + m <= dst_first || dst_first + k <= src_first, PF_E_ARG, "pf_combine: overlapping ranges");
    const bool two = src_first2 >= 0;
    if (two) {
        PF_TRY(check_slots(g, src_first2, m, "pf_combine"));
        PF_TRY(check_slots(g, dst_first2, k, "pf_combine"));
        PF_CHECK((src_first2 + m <= dst_first2 || dst_first2 + k <= src_first2) && (dst_first + k <= dst_first2 || dst_first2 + k <= dst_first) &&
                     (src_first + m <= dst_first2 || dst_first2 + k <= src_first) && (src_first2 + m <= dst_first || dst_first + k <= src_first2),
                 PF_E_ARG, "pf_combine: overlapping ranges");
    }
    pf_ctx* ctx = g->ctx;
    hipStream_t st = ctx->stream;
    double* dY = nullptr;
    PF_HIP(pf_malloc(st, (void**)&dY, sizeof(double) * (size_t)m * k));
    // Y goes through one of a few pinned staging slots, so that the call need not wait for the copy (a synchronisation costs
    // 30-50 us of idle device, and the tail of a solve has three of these calls); a slot is reused only after the copy
    // that read it last has completed
    const size_t bytes = sizeof(double) * (size_t)m * k;
    const void* src = Y;
    bool staged = false;
    if (bytes <= PF_STAGE_BYTES) {
        if (!ctx->stage_ring) {
            if (hipHostMalloc(&ctx->stage_ring, (size_t)PF_STAGE_SLOTS * PF_STAGE_BYTES, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                ctx->stage_ring = nullptr;
            } else {
                for (int i = 0; i < PF_STAGE_SLOTS; ++i) ctx->stage_ev[i] = nullptr;
            }
        }
        if (ctx->stage_ring) {
            const int slot = ctx->stage_next;
            ctx->stage_next = (slot + 1) % PF_STAGE_SLOTS;
            bool ok = true;
            if (!ctx->stage_ev[slot]) ok = hipEventCreateWithFlags(&ctx->stage_ev[slot], hipEventDisableTiming) == hipSuccess;
            else ok = hipEventSynchronize(ctx->stage_ev[slot]) == hipSuccess;
            if (ok) {
                void* dst = static_cast<unsigned char*>(ctx->stage_ring) + (size_t)slot * PF_STAGE_BYTES;
                memcpy(dst, Y, bytes);
                src = dst;
                staged = true;
                hipError_t es = hipMemcpyAsync(dY, src, bytes, hipMemcpyHostToDevice, st);
                if (es == hipSuccess) es = hipEventRecord(ctx->stage_ev[slot], st);
                if (es != hipSuccess) {
                    pf_free(st, dY);
                    PF_HIP(es);
                }
            } else {
                (void)hipGetLastError();
            }
        }
    }
    hipError_t e = staged ? hipSuccess : hipMemcpyAsync(dY, Y, bytes, hipMemcpyHostToDevice, st);
    for (int32_t c0 = 0; c0 < k && e == hipSuccess; c0 += COMBINE_COLS) {
        const int32_t nc = std::min(COMBINE_COLS, k - c0);
        k_combine<<<dim3(nblk(g->n_pad), two ? 2u : 1u), PF_BLOCK, 0, st>>>(g->ws, g->n_pad, src_first, m, dY, k, c0, nc, dst_first, src_first2,
                                                                          dst_first2);
        e = hipGetLastError();
    }
    hipError_t e2 = staged ? hipSuccess : hipStreamSynchronize(st);  // (not staged: Y is the caller's host buffer)
    pf_free(st, dY);
    PF_HIP(e);
    PF_HIP(e2);
    return PF_OK;
}

int pf_resnorm(pf_graph* g, int32_t ax, int32_t x, double lam, double* out) {
    PF_TRY(check_slots(g, ax, 1, "pf_resnorm"));
    PF_TRY(check_slots(g, x, 1, "pf_resnorm"));
    PF_CHECK(out != nullptr, PF_E_ARG, "pf_resnorm: out is NULL");
    hipStream_t st = g->ctx->stream;
    PF_TRY(pf_reduce_ensure(g, 1));
    k_resnorm_partial<<<(unsigned)g->n_chunks, PF_BLOCK, 0, st>>>(pf_slot(g, ax), pf_slot(g, x), lam, g->n_pad, g->n_chunks, g->partials);
    PF_HIP(hipGetLastError());
    k_dot_finish<<<1, PF_WAVE, 0, st>>>(g->partials, g->n_chunks, g->coef, nullptr, 0);
    PF_HIP(hipGetLastError());
    double r2 = 0.0;
    PF_HIP(hipMemcpyAsync(&r2, g->coef, sizeof(double), hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    *out = sqrt(r2 > 0.0 ? r2 : 0.0);
    return PF_OK;
}

int pf_gram(pf_graph* g, int32_t first_a, int32_t count_a, int32_t first_b, int32_t count_b, double* out) {
    PF_TRY(check_slots(g, first_a, count_a, "pf_gram"));
    PF_TRY(check_slots(g, first_b, count_b, "pf_gram"));
    PF_CHECK(out != nullptr && count_a > 0 && count_b > 0, PF_E_ARG, "pf_gram: bad argument");
    PF_TRY(pf_reduce_ensure(g, count_a * count_b));  // partials: one column per pair
    hipStream_t st = g->ctx->stream;
    k_gram_partial<<<dim3((unsigned)g->n_chunks, (unsigned)(count_a * count_b)), PF_BLOCK, 0, st>>>(g->ws, g->n_pad, first_a, first_b, count_b,
                                                                                                 g->n_chunks, g->partials);
    PF_HIP(hipGetLastError());
    k_dot_finish<<<(unsigned)(count_a * count_b), PF_WAVE, 0, st>>>(g->partials, g->n_chunks, g->coef, nullptr, 0);
    PF_HIP(hipGetLastError());
    return small_to_host(g, g->coef, out, (size_t)count_a * count_b);
}

// pf_gram / pf_resnorms in two halves: _begin queues the kernels and a copy of the few results into a pinned block of the
// graph's own, with an event behind it; pf_small_end waits for that event only - what the stream holds behind it (the
// partner graph's extraction) keeps running.  One collection in flight per graph.
static int small_begin(pf_graph* g, const double* d_src, size_t count, bool root, bool append = false) {
    PF_CHECK(append ? g->small_pending > 0 && !g->small_root && !root : g->small_pending == 0, PF_E_STATE,
             "pf_gram_begin / pf_resnorms_begin: a previous result has not been collected");
    pf_ctx* c = g->ctx;
    hipStream_t st = c->stream;
    const size_t offset = append ? (size_t)g->small_pending : 0;
    if (append) {
        PF_CHECK((int64_t)(offset + count) <= g->small_host_cap, PF_E_STATE, "pf_gram_begin: no room to append");
    } else if ((int64_t)(2 * count) > g->small_host_cap || !g->small_host) {  // (room for an appended block of the same size)
        if (g->small_host) {
            PF_HIP(hipStreamSynchronize(st));
            c->pinned_pool.emplace_back(g->small_host_cap, g->small_host);
            g->small_host = nullptr;
        }
        const int32_t cap = std::max((int32_t)(2 * count), 128);
        for (size_t i = 0; i < c->pinned_pool.size(); ++i)
            if (c->pinned_pool[i].first >= cap) {
                g->small_host_cap = c->pinned_pool[i].first;
                g->small_host = c->pinned_pool[i].second;
                c->pinned_pool.erase(c->pinned_pool.begin() + (long)i);
                break;
            }
        if (!g->small_host) {
            PF_HIP(hipHostMalloc((void**)&g->small_host, sizeof(double) * (size_t)(cap + 2), hipHostMallocDefault));
            g->small_host_cap = cap;
        }
    }
    if (!g->small_ev) {
        if (!c->event_pool.empty()) {
            g->small_ev = c->event_pool.back();
            c->event_pool.pop_back();
        } else {
            PF_HIP(hipEventCreateWithFlags(&g->small_ev, hipEventDisableTiming));
        }
    }
    PF_TRY(pf_copy_by_kernel(st, d_src, g->small_host + offset, sizeof(double) * count));
    PF_HIP(hipEventRecord(g->small_ev, st));
    g->small_pending = (int32_t)(offset + count);
    g->small_root = root;
    return PF_OK;
}

int pf_small_end(pf_graph* g, double* out) {
    PF_CHECK(g != nullptr && out != nullptr && g->small_pending > 0, PF_E_STATE, "pf_small_end: nothing to collect");
    PF_HIP(hipEventSynchronize(g->small_ev));
    const int32_t count = g->small_pending;
    g->small_pending = 0;
    for (int32_t i = 0; i < count; ++i) {
        const double v = g->small_host[i];
        out[i] = g->small_root ? sqrt(v > 0.0 ? v : 0.0) : v;
    }
    return PF_OK;
}

// (append: the result goes behind the one already waiting - of the same size at most - and pf_small_end returns both)
int pf_gram_begin(pf_graph* g, int32_t first_a, int32_t count_a, int32_t first_b, int32_t count_b, int32_t append) {
    PF_TRY(check_slots(g, first_a, count_a, "pf_gram"));
    PF_TRY(check_slots(g, first_b, count_b, "pf_gram"));
    PF_CHECK(count_a > 0 && count_b > 0, PF_E_ARG, "pf_gram: bad argument");
    PF_TRY(pf_reduce_ensure(g, count_a * count_b));  // partials: one column per pair
    hipStream_t st = g->ctx->stream;
    k_gram_partial<<<dim3((unsigned)g->n_chunks, (unsigned)(count_a * count_b)), PF_BLOCK, 0, st>>>(g->ws, g->n_pad, first_a, first_b, count_b,
                                                                                                 g->n_chunks, g->partials);
    PF_HIP(hipGetLastError());
    k_dot_finish<<<(unsigned)(count_a * count_b), PF_WAVE, 0, st>>>(g->partials, g->n_chunks, g->coef, nullptr, 0);
    PF_HIP(hipGetLastError());
    return small_begin(g, g->coef, (size_t)count_a * count_b, false, append != 0);
}

int pf_resnorms_begin(pf_graph* g, int32_t ax_first, int32_t x_first, const double* lam, int32_t count) {
    PF_TRY(check_slots(g, ax_first, count, "pf_resnorms"));
    PF_TRY(check_slots(g, x_first, count, "pf_resnorms"));
    PF_CHECK(lam != nullptr && count > 0, PF_E_ARG, "pf_resnorms: bad argument");
    hipStream_t st = g->ctx->stream;
    PF_TRY(pf_reduce_ensure(g, count));
    for (int32_t at = 0; at < count; at += PF_RESNORMS_MAX) {
        const int32_t nb = std::min(count - at, PF_RESNORMS_MAX);
        LamArgs la{};
        for (int32_t i = 0; i < nb; ++i) la.lam[i] = lam[at + i];
        k_resnorms_partial<<<dim3((unsigned)g->n_chunks, (unsigned)nb), PF_BLOCK, 0, st>>>(g->ws, g->n_pad, ax_first + at, x_first + at, la,
                                                                                        g->n_chunks, g->partials);
        PF_HIP(hipGetLastError());
        k_dot_finish<<<(unsigned)nb, PF_WAVE, 0, st>>>(g->partials, g->n_chunks, g->coef + at, nullptr, 0);
        PF_HIP(hipGetLastError());
    }
    return small_begin(g, g->coef, (size_t)count, true);
}

int pf_resnorms(pf_graph* g, int32_t ax_first, int32_t x_first, const double* lam, int32_t count, double* out) {
    PF_TRY(check_slots(g, ax_first, count, "pf_resnorms"));
    PF_TRY(check_slots(g, x_first, count, "pf_resnorms"));
    PF_CHECK(out != nullptr && lam != nullptr && count > 0, PF_E_ARG, "pf_resnorms: bad argument");
    hipStream_t st = g->ctx->stream;
    PF_TRY(pf_reduce_ensure(g, count));
    for (int32_t at = 0; at < count; at += PF_RESNORMS_MAX) {
        const int32_t nb = std::min(count - at, PF_RESNORMS_MAX);
        LamArgs la{};
        for (int32_t i = 0; i < nb; ++i) la.lam[i] = lam[at + i];
        k_resnorms_partial<<<dim3((unsigned)g->n_chunks, (unsigned)nb), PF_BLOCK, 0, st>>>(g->ws, g->n_pad, ax_first + at, x_first + at, la,
                                                                                        g->n_chunks, g->partials);
        PF_HIP(hipGetLastError());
        k_dot_finish<<<(unsigned)nb, PF_WAVE, 0, st>>>(g->partials, g->n_chunks, g->coef + at, nullptr, 0);
        PF_HIP(hipGetLastError());
    }
    PF_TRY(small_to_host(g, g->coef, out, (size_t)count));
    for (int32_t i = 0; i < count; ++i) out[i] = sqrt(out[i] > 0.0 ? out[i] : 0.0);
    return PF_OK;
}

// The eigenvector post-processing in two halves, so that the n x count download (10 MB per 250k-vertex graph) leaves the
// critical path: _begin queues the kernels on the ctx stream and the copies on the ctx's COPY stream behind an event -
// later work on the ctx stream (eigsort's cost matrices, the KNN, the partner graph's solve) overlaps with them - and
// _end waits for the copies and checks the statistics.  The block itself is resident and usable on the device as soon as
// _begin returns.  `out` should be pinned memory (pf_host_alloc): a pageable destination is staged in chunks by the
// runtime (measured: ~12 gaps of 82 us per 250k pair).
static bool downloads_deferred() {
    static const bool on = [] { const char* v = getenv("PF_DOWNLOAD_DEFER"); return !(v && v[0] == '0'); }();
    return on;
}

// the owed image of g goes onto the copy stream, behind everything the ctx stream holds at this moment
static int queue_download(pf_graph* g) {
    pf_ctx* ctx = g->ctx;
    const double* src = nullptr;
    {
        std::lock_guard<std::mutex> lk(ctx->deferred_mutex);
        auto it = std::find(ctx->deferred.begin(), ctx->deferred.end(), g);
        if (it != ctx->deferred.end()) ctx->deferred.erase(it);
        src = g->dl_src;
        g->dl_src = nullptr;
    }
    if (!src) return PF_OK;
    PF_HIP(hipEventRecord(g->final_ready, ctx->stream));
    PF_HIP(hipStreamWaitEvent(ctx->copy_stream, g->final_ready, 0));
    // (the runtime's copy: a download kernel of our own with a few blocks, at the copy stream's low priority, only got CUs
    // when the search had finished - KNN stage 1.50 ms instead of 1.20)
    PF_HIP(hipMemcpyAsync(g->dl_dst, src, g->dl_bytes, hipMemcpyDeviceToHost, ctx->copy_stream));
    PF_HIP(hipEventRecord(g->final_done, ctx->copy_stream));
    return PF_OK;
}

// forget the image owed to the caller's buffer (never queued: nothing will be written), or wait for the one in flight;
// the resident block on the device stays usable.  For callers that are about to report a failure.
extern "C++" int pf_download_cancel(pf_graph* g) {
    pf_ctx* ctx = g->ctx;
    bool owed = false;
    {
        std::lock_guard<std::mutex> lk(ctx->deferred_mutex);
        auto it = std::find(ctx->deferred.begin(), ctx->deferred.end(), g);
        if (it != ctx->deferred.end()) ctx->deferred.erase(it);
        owed = g->dl_src != nullptr;
        g->dl_src = nullptr;
    }
    if (g->final_pending != 0) {
        g->final_pending = 0;
        g->final_check = 0;
        if (!owed) (void)hipEventSynchronize(g->final_done);
        pf_free(ctx->stream, g->final_params);
        g->final_params = nullptr;
        pf_free(ctx->stream, g->final_tmp);
        g->final_tmp = nullptr;
    }
    return PF_OK;
}

extern "C++" int pf_downloads_release(pf_ctx* c) {
    for (;;) {
        pf_graph* g = nullptr;
        {
            std::lock_guard<std::mutex> lk(c->deferred_mutex);
            if (c->deferred.empty()) break;
            g = c->deferred.back();
        }
        PF_TRY(queue_download(g));
    }
    return PF_OK;
}

int pf_finalize_vectors_begin(pf_graph* g, int32_t first, int32_t count, int32_t from_sym, int32_t minmax, double* out) {
    PF_TRY(check_slots(g, first, count, "pf_finalize_vectors"));
    PF_CHECK(out != nullptr && count > 0, PF_E_ARG, "pf_finalize_vectors: bad argument");
    PF_CHECK(!from_sym || g->is_symmetric, PF_E_ARG, "pf_finalize_vectors: from_sym on an asymmetric graph");
    PF_TRY(pf_finalize_vectors_end(g));  // an earlier result still on its way
    pf_ctx* ctx = g->ctx;
    hipStream_t st = ctx->stream;
    if (!ctx->copy_stream) PF_HIP(pf_create_side_stream(&ctx->copy_stream, true));
    // events and the pinned landing place of the statistics come from the ctx's pools (what freed graphs left behind:
    // hipHostMalloc / hipHostFree and event creation cost 0.1-0.2 ms each)
    for (hipEvent_t* ev : {&g->final_ready, &g->final_done}) {
        if (*ev) continue;
        if (!ctx->event_pool.empty()) {
            *ev = ctx->event_pool.back();
            ctx->event_pool.pop_back();
        } else {
            PF_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
        }
    }
    const int32_t stat_doubles = (int32_t)(sizeof(VecStats) / sizeof(double)) * count;
    if (stat_doubles > g->final_stats_cap + 2 || !g->final_stats) {
        if (g->final_stats) ctx->pinned_pool.emplace_back(g->final_stats_cap, reinterpret_cast<double*>(g->final_stats));
        g->final_stats = nullptr;
        for (size_t i = 0; i < ctx->pinned_pool.size(); ++i)
            if (ctx->pinned_pool[i].first + 2 >= stat_doubles) {
                g->final_stats_cap = ctx->pinned_pool[i].first;
                g->final_stats = ctx->pinned_pool[i].second;
                ctx->pinned_pool.erase(ctx->pinned_pool.begin() + (long)i);
                break;
            }
        if (!g->final_stats) {
            const int32_t cap = std::max(stat_doubles, 384);
            PF_HIP(hipHostMalloc((void**)&g->final_stats, sizeof(double) * (size_t)(cap + 2), hipHostMallocDefault));
            g->final_stats_cap = cap;
        }
    }
    PF_TRY(pf_reduce_ensure(g, count));
    VecStats* part = reinterpret_cast<VecStats*>(g->partials);
    VecStats* fin = part + (size_t)count * g->n_chunks;
    dim3 grid((unsigned)g->n_chunks, (unsigned)count);
    k_vec_stats_partial<<<grid, PF_BLOCK, 0, st>>>(g->ws, g->n_pad, g->n, first, g->sg, g->perm, g->perm_m, from_sym, g->n_chunks, part);
    PF_HIP(hipGetLastError());
    k_vec_stats_finish<<<(unsigned)count, PF_WAVE, 0, st>>>(part, g->n_chunks, fin);
    PF_HIP(hipGetLastError());
    // scale / sign / min-max parameters straight from the statistics, on the device (no read-back in between: the
    // statistics come to the host with the result and are checked then)
    double *d_params = nullptr, *d_out = nullptr;
    pf_free(st, g->final_vecs);  // the result stays resident (pf_final_rows, pf_knn1_graphs) until the next call
    g->final_vecs = nullptr;
    g->final_count = 0;
    // parameters, and a copy of the statistics that later reductions on the ctx stream cannot overwrite
    PF_HIP(pf_malloc(st, (void**)&d_params, sizeof(double) * 4 * (size_t)count + sizeof(VecStats) * (size_t)count));
    VecStats* d_stats = reinterpret_cast<VecStats*>(d_params + 4 * (size_t)count);
    hipError_t e = pf_malloc(st, (void**)&d_out, sizeof(double) * (size_t)g->n * count);
    if (e == hipSuccess) {
        k_vec_params<<<1, PF_WAVE, 0, st>>>(fin, count, minmax, d_params);
        k_vec_apply<<<nblk(g->n), PF_BLOCK, 0, st>>>(g->ws, g->n_pad, g->n, first, count, g->sg, g->iperm, g->mrank, from_sym, d_params, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(d_stats, fin, sizeof(VecStats) * count, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipEventRecord(g->final_ready, st);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->copy_stream, g->final_ready, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(g->final_stats, d_stats, sizeof(VecStats) * count, hipMemcpyDeviceToHost, ctx->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(g->final_done, ctx->copy_stream);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        pf_free(st, d_params);
        pf_free(st, d_out);
        PF_HIP(e);
    }
    g->final_vecs = d_out;
    g->final_count = count;
    g->final_params = d_params;
    g->final_pending = count;
    g->final_check = count;
    // The n x count image itself is OWED, not queued: while a 10 MB download runs, the next kernel boundary of the ctx
    // stream waits for it to drain (kernel trace: the partner's k_vec_params, eigsort's k_es_minmax and the search's first
    // kernel each took the download's 180 us) - beside small dependent kernels a download costs what it would cost in
    // line.  It leaves behind the next LONG kernel (pf_downloads_release: the 1-NN search), or when somebody collects it.
    g->dl_src = d_out;
    g->dl_dst = out;
    g->dl_bytes = sizeof(double) * (size_t)g->n * count;
    if (downloads_deferred()) {
        std::lock_guard<std::mutex> lk(ctx->deferred_mutex);
        if (std::find(ctx->deferred.begin(), ctx->deferred.end(), g) == ctx->deferred.end()) ctx->deferred.push_back(g);
    } else {
        PF_TRY(queue_download(g));
    }
    return PF_OK;
}

int pf_finalize_vectors_end(pf_graph* g) {
    PF_CHECK(g != nullptr, PF_E_ARG, "pf_finalize_vectors_end: NULL graph");
    if (g->final_pending == 0) return PF_OK;
    g->final_pending = 0;
    const int32_t count = g->final_check;
    g->final_check = 0;
    int rc = queue_download(g);  // (nobody released it: now)
    const hipError_t e = hipEventSynchronize(g->final_done);
    pf_free(g->ctx->stream, g->final_params);
    g->final_params = nullptr;
    pf_free(g->ctx->stream, g->final_tmp);
    g->final_tmp = nullptr;
    PF_TRY(rc);
    if (count <= 0) {  // only a remapped image of a block whose statistics have been looked at
        PF_HIP(e);
        return PF_OK;
    }
    const VecStats* hs = reinterpret_cast<const VecStats*>(g->final_stats);
    bool sane = e == hipSuccess;
    int32_t bad = 0;
    for (int32_t c = 0; c < count && sane; ++c)
        if (!(hs[c].sumsq > 0.0 && isfinite(hs[c].sumsq))) sane = false, bad = c;
    if (!sane) {
        pf_free(g->ctx->stream, g->final_vecs);
        g->final_vecs = nullptr;
        g->final_count = 0;
    }
    PF_HIP(e);
    PF_CHECK(sane, PF_E_STATE, "pf_finalize_vectors: vector %d has norm^2 %g", bad, hs[(size_t)bad].sumsq);
    return PF_OK;
}

int pf_finalize_vectors(pf_graph* g, int32_t first, int32_t count, int32_t from_sym, int32_t minmax, double* out) {
    PF_TRY(pf_finalize_vectors_begin(g, first, count, from_sym, minmax, out));
    return pf_finalize_vectors_end(g);
}

struct RemapArgs {
    int32_t col[64];
    double sign[64];
};
__global__ __launch_bounds__(PF_BLOCK) void k_final_remap(const double* __restrict__ fin, int64_t n, int32_t fc, int32_t count, RemapArgs m,
                                                          double* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (e >= n * count) return;
    const int64_t i = e / count;
    const int32_t c = (int32_t)(e - i * count);
    out[e] = fin[i * fc + m.col[c]] * m.sign[c];  // (sign = +-1: exact)
}

// eigsort's sign flips and column moves (eigsort.py:108-122) applied to the resident block's image on the host:
// out[i][c] = block[i][col[c]] * sign[c], computed on the device and copied on the copy stream - O(n k) strided host
// work (0.25 ms at 250k x 5, several ms at 1M x 10) becomes one DMA that overlaps with the KNN.  Collected by
// pf_finalize_vectors_end like the first download; the resident block itself is unchanged.
int pf_final_remap_begin(pf_graph* g, const int32_t* col, const double* sign, int32_t count, double* out) {
    PF_CHECK(g && col && sign && out, PF_E_ARG, "pf_final_remap_begin: NULL argument");
    // An image that is still owed to the same array is simply replaced by the remapped one (its statistics are looked at
    // when that is collected); one that is in flight, or owed to another array, is collected first.
    if (!(g->dl_src && g->dl_dst == out)) PF_TRY(pf_finalize_vectors_end(g));
    PF_CHECK(g->final_vecs != nullptr, PF_E_STATE, "pf_final_remap_begin: no pf_finalize_vectors result is resident");
    PF_CHECK(count >= 1 && count <= 64 && count <= g->final_count, PF_E_ARG, "pf_final_remap_begin: count %d out of range", count);
    RemapArgs m{};
    for (int32_t c = 0; c < count; ++c) {
        PF_CHECK(col[c] >= 0 && col[c] < g->final_count, PF_E_ARG, "pf_final_remap_begin: column %d out of range", col[c]);
        m.col[c] = col[c];
        m.sign[c] = sign[c];
    }
    pf_ctx* ctx = g->ctx;
    hipStream_t st = ctx->stream;
    PF_CHECK(ctx->copy_stream && g->final_ready && g->final_done, PF_E_STATE, "pf_final_remap_begin: no download machinery (call after pf_finalize_vectors_begin)");
    double* d_tmp = nullptr;
    PF_HIP(pf_malloc(st, (void**)&d_tmp, sizeof(double) * (size_t)g->n * count));
    k_final_remap<<<nblk(g->n * count), PF_BLOCK, 0, st>>>(g->final_vecs, g->n, g->final_count, count, m, d_tmp);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        pf_free(st, d_tmp);
        PF_HIP(e);
    }
    pf_free(st, g->final_tmp);  // (an earlier remapped image that was never sent: behind its last use in stream order)
    g->final_tmp = d_tmp;
    g->dl_src = d_tmp;
    g->dl_dst = out;
    g->dl_bytes = sizeof(double) * (size_t)g->n * count;
    g->final_pending = count;
    if (downloads_deferred()) {
        std::lock_guard<std::mutex> lk(ctx->deferred_mutex);
        if (std::find(ctx->deferred.begin(), ctx->deferred.end(), g) == ctx->deferred.end()) ctx->deferred.push_back(g);
    } else {
        PF_TRY(queue_download(g));
    }
    return PF_OK;
}

// out[t][c] = src[rows[t]][c] for a resident [n][width] block of the graph
static int rows_to_host(pf_graph* g, const double* src, int32_t width, const int64_t* rows, int64_t n_rows, double* out,
                        const char* who) {
    if (n_rows == 0) return PF_OK;
    PF_HIP(hipSetDevice(g->ctx->device));
    hipStream_t st = g->ctx->stream;
    for (int64_t i = 0; i < n_rows; ++i)
        PF_CHECK(rows[i] >= 0 && rows[i] < g->n, PF_E_ARG, "%s: row %lld out of range", who, (long long)rows[i]);
    int64_t* d_rows = nullptr;
    double* d_out = nullptr;
    PF_HIP(pf_malloc(st, (void**)&d_rows, sizeof(int64_t) * n_rows));
    hipError_t e = pf_malloc(st, (void**)&d_out, sizeof(double) * (size_t)n_rows * width);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rows, rows, sizeof(int64_t) * n_rows, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        k_final_rows<<<nblk(n_rows * width), PF_BLOCK, 0, st>>>(src, d_rows, n_rows, width, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, sizeof(double) * (size_t)n_rows * width, hipMemcpyDeviceToHost, st);
    hipError_t e2 = hipStreamSynchronize(st);
    pf_free(st, d_rows);
    pf_free(st, d_out);
    PF_HIP(e);
    PF_HIP(e2);
    return PF_OK;
}

int pf_final_rows(pf_graph* g, const int64_t* rows, int64_t n_rows, double* out) {
    PF_CHECK(g != nullptr && rows != nullptr && out != nullptr && n_rows >= 0, PF_E_ARG, "pf_final_rows: bad argument");
    PF_CHECK(g->final_vecs != nullptr, PF_E_STATE, "pf_final_rows: no pf_finalize_vectors result is resident");
    return rows_to_host(g, g->final_vecs, g->final_count, rows, n_rows, out, "pf_final_rows");
}

int pf_point_rows(pf_graph* g, const int64_t* rows, int64_t n_rows, double* out) {
    PF_CHECK(g != nullptr && rows != nullptr && out != nullptr && n_rows >= 0, PF_E_ARG, "pf_point_rows: bad argument");
    PF_CHECK(g->pts != nullptr, PF_E_STATE, "pf_point_rows: this graph was not built from a mesh");
    return rows_to_host(g, g->pts, 3, rows, n_rows, out, "pf_point_rows");
}

int pf_spmv_host(pf_graph* g, int32_t op, const double* x, double* y) {
    PF_CHECK(g != nullptr && x != nullptr && y != nullptr, PF_E_ARG, "pf_spmv_host: NULL argument");
    const double* vals = op_values(g, op);
    PF_CHECK(vals != nullptr, PF_E_ARG, "pf_spmv_host: operator %d unavailable", op);
    PF_TRY(pf_ws_ensure(g, 2));
    PF_TRY(pf_ws_upload(g, 0, x));
    PF_TRY(launch_op(g, vals, pf_slot(g, 0), nullptr, pf_slot(g, 1), -1.0, 0.0, 0.0));
    return pf_ws_download(g, 1, 1, y);
}

int pf_mean_filter(pf_graph* g, const double* values, int32_t ncols, int32_t iterations, double* out) {
    PF_CHECK(g != nullptr && values != nullptr && out != nullptr && ncols > 0 && iterations >= 0, PF_E_ARG,
             "pf_mean_filter: bad argument");
    PF_HIP(hipSetDevice(g->ctx->device));
    hipStream_t st = g->ctx->stream;
    if (!g->mf_col) {  // first use: the filter's rows in solver order
        const size_t entries = (size_t)(g->sell_entries + g->n_pad);
        PF_HIP(pf_malloc(st, (void**)&g->mf_col, sizeof(int32_t) * entries));
        PF_HIP(pf_malloc(st, (void**)&g->mf_val, sizeof(double) * entries));
        k_fill_mean_filter<<<nblk(g->n_pad), PF_BLOCK, 0, st>>>(g->rowptr, g->col, g->w, g->deg, g->perm_m, g->iperm_m, g->morder, g->n_pad,
                                                                g->slice_ptr, g->mf_col, g->mf_val);
        PF_HIP(hipGetLastError());
    }
    const size_t bytes_mesh = sizeof(double) * (size_t)g->n * ncols, bytes_pad = sizeof(double) * (size_t)g->n_pad * ncols;
    double *m = nullptr, *a = nullptr, *b = nullptr;
    PF_HIP(pf_malloc(st, (void**)&m, bytes_mesh));
    hipError_t e = pf_malloc(st, (void**)&a, bytes_pad);
    if (e == hipSuccess) e = pf_malloc(st, (void**)&b, bytes_pad);
    if (e == hipSuccess) e = hipMemcpyAsync(m, values, bytes_mesh, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        k_rows_in<<<nblk(g->n_pad * ncols), PF_BLOCK, 0, st>>>(m, g->perm, g->n_pad, ncols, a);
        for (int32_t it = 0; it < iterations; ++it) {
            if (ncols == 3) k_mean_filter<3><<<nblk(g->n_pad), PF_BLOCK, 0, st>>>(g->slice_ptr, g->mf_col, g->mf_val, g->perm, g->n_pad, ncols, a, b);
            else if (ncols == 1) k_mean_filter<1><<<nblk(g->n_pad), PF_BLOCK, 0, st>>>(g->slice_ptr, g->mf_col, g->mf_val, g->perm, g->n_pad, ncols, a, b);
            else k_mean_filter<0><<<nblk(g->n_pad), PF_BLOCK, 0, st>>>(g->slice_ptr, g->mf_col, g->mf_val, g->perm, g->n_pad, ncols, a, b);
            std::swap(a, b);
        }
        k_rows_out<<<nblk(g->n * ncols), PF_BLOCK, 0, st>>>(a, g->iperm, g->n, ncols, m);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, m, bytes_mesh, hipMemcpyDeviceToHost, st);
    hipError_t e2 = hipStreamSynchronize(st);
    pf_free(st, m);
    pf_free(st, a);
    pf_free(st, b);
    PF_HIP(e);
    PF_HIP(e2);
    return PF_OK;
}

// ---- primitives of the row-partitioned solve (pyfocusr_amd/rowpart.py)
int pf_op_step(pf_graph* g, int32_t op, int32_t x, int32_t prev, int32_t out, double alpha, double shift, double beta) {
    PF_TRY(check_slots(g, x, 1, "pf_op_step"));
    PF_TRY(check_slots(g, out, 1, "pf_op_step"));
    if (prev >= 0) PF_TRY(check_slots(g, prev, 1, "pf_op_step"));
    const double* vals = op_values(g, op);
    PF_CHECK(vals != nullptr && x != out, PF_E_ARG, "pf_op_step: operator %d unavailable or x == out", op);
    OpTimer t(g->ctx, 1, (double)op_bytes(g));
    PF_TRY(launch_op(g, vals, pf_slot(g, x), prev >= 0 ? pf_slot(g, prev) : nullptr, pf_slot(g, out), alpha, shift, beta));
    return t.finish();
}

int pf_cheb_steps(pf_graph* g, int32_t op, int32_t prev, int32_t cur, int32_t k_first, int32_t n_steps, double c, double e,
                  double rho, int32_t* out_prev, int32_t* out_cur) {
    PF_TRY(check_slots(g, prev, 1, "pf_cheb_steps"));
    PF_TRY(check_slots(g, cur, 1, "pf_cheb_steps"));
    const double* vals = op_values(g, op);
    PF_CHECK(vals != nullptr && prev != cur && out_prev && out_cur, PF_E_ARG, "pf_cheb_steps: bad operator / slots");
    PF_CHECK(k_first >= 1 && n_steps >= 0 && e > 0.0 && rho >= 1.0, PF_E_ARG, "pf_cheb_steps: k_first %d / n_steps %d invalid",
             k_first, n_steps);
    OpTimer t(g->ctx, n_steps, (double)n_steps * op_bytes(g));
    int32_t a = prev, b = cur;  // b holds y_{k-1} of the step about to run, a receives y_k (step 1: a is overwritten, no prev term)
    for (int32_t k = k_first; k < k_first + n_steps; ++k) {
        if (k == 1) PF_TRY(launch_op(g, vals, pf_slot(g, b), nullptr, pf_slot(g, a), 1.0 / (e * rho), c, 0.0));
        else PF_TRY(launch_op(g, vals, pf_slot(g, b), pf_slot(g, a), pf_slot(g, a), 2.0 / (e * rho), c, 1.0 / (rho * rho)));
        std::swap(a, b);
    }
    *out_prev = a;
    *out_cur = b;
    return t.finish();
}

int pf_axpy(pf_graph* g, int32_t w, int32_t first, int32_t count, const double* coef) {
    PF_TRY(check_slots(g, w, 1, "pf_axpy"));
    PF_TRY(check_slots(g, first, count, "pf_axpy"));
    PF_CHECK(coef != nullptr || count == 0, PF_E_ARG, "pf_axpy: coef is NULL");
    PF_CHECK(w < first || w >= first + count, PF_E_ARG, "pf_axpy: w inside the basis range");
    if (count == 0) return PF_OK;
    hipStream_t st = g->ctx->stream;
    PF_TRY(pf_reduce_ensure(g, count));
    std::vector<double> neg(coef, coef + count);
    for (double& v : neg) v = -v;  // k_multi_axpy subtracts
    PF_HIP(hipMemcpyAsync(g->coef, neg.data(), sizeof(double) * count, hipMemcpyHostToDevice, st));
    k_multi_axpy<<<nblk(g->n_pad / 2), PF_BLOCK, 0, st>>>(g->ws, g->n_pad, first, count, w, g->coef);
    PF_HIP(hipGetLastError());
    PF_HIP(hipStreamSynchronize(st));  // neg goes out of scope
    return PF_OK;
}

struct pf_rows {
    pf_graph* g = nullptr;
    int32_t* idx = nullptr;  // solver-order row of every entry
    double* buf = nullptr;   // device staging [n]
    int64_t* off = nullptr;  // optional: where each row's value sits in a receive buffer (pf_rows_set_sources)
    int64_t n = 0;
};

void pf_rows_free(pf_rows* r) {
    if (!r) return;
    hipSetDevice(r->g->ctx->device);
    hipStream_t st = r->g->ctx->stream;
    hipStreamSynchronize(st);
    pf_free(st, r->idx);
    pf_free(st, r->buf);
    pf_free(st, r->off);
    delete r;
}

int pf_rows_create(pf_graph* g, const int64_t* rows, int64_t n, pf_rows** out) {
    PF_CHECK(g && out && (rows || n == 0) && n >= 0, PF_E_ARG, "pf_rows_create: bad argument");
    for (int64_t i = 0; i < n; ++i)
        PF_CHECK(rows[i] >= 0 && rows[i] < g->n, PF_E_ARG, "pf_rows_create: row %lld outside [0, %lld)", (long long)rows[i],
                 (long long)g->n);
    PF_HIP(hipSetDevice(g->ctx->device));
    hipStream_t st = g->ctx->stream;
    pf_rows* r = new pf_rows();
    r->g = g;
    r->n = n;
    if (n > 0) {
        int64_t* tmp = nullptr;
        hipError_t e = pf_malloc(st, (void**)&r->idx, sizeof(int32_t) * n);
        if (e == hipSuccess) e = pf_malloc(st, (void**)&r->buf, sizeof(double) * n);
        if (e == hipSuccess) e = pf_malloc(st, (void**)&tmp, sizeof(int64_t) * n);
        if (e == hipSuccess) e = hipMemcpyAsync(tmp, rows, sizeof(int64_t) * n, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            k_rows_to_new<<<nblk(n), PF_BLOCK, 0, st>>>(tmp, g->iperm, n, r->idx);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        pf_free(st, tmp);
        if (e != hipSuccess) {
            pf_set_error("pf_rows_create: %s", hipGetErrorString(e));
            pf_rows_free(r);
            return PF_E_HIP;
        }
    }
    *out = r;
    return PF_OK;
}

int pf_rows_gather(pf_rows* r, int32_t slot, double* out) {
    PF_CHECK(r && (out || r->n == 0), PF_E_ARG, "pf_rows_gather: NULL argument");
    PF_TRY(check_slots(r->g, slot, 1, "pf_rows_gather"));
    if (r->n == 0) return PF_OK;
    hipStream_t st = r->g->ctx->stream;
    k_rows_gather<<<nblk(r->n), PF_BLOCK, 0, st>>>(pf_slot(r->g, slot), r->idx, r->n, r->buf);
    PF_HIP(hipGetLastError());
    PF_HIP(hipMemcpyAsync(out, r->buf, sizeof(double) * r->n, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    return PF_OK;
}

int pf_rows_scatter(pf_rows* r, int32_t slot, const double* in) {
    PF_CHECK(r && (in || r->n == 0), PF_E_ARG, "pf_rows_scatter: NULL argument");
    PF_TRY(check_slots(r->g, slot, 1, "pf_rows_scatter"));
    if (r->n == 0) return PF_OK;
    hipStream_t st = r->g->ctx->stream;
    PF_HIP(hipMemcpyAsync(r->buf, in, sizeof(double) * r->n, hipMemcpyHostToDevice, st));
    k_rows_scatter<<<nblk(r->n), PF_BLOCK, 0, st>>>(pf_slot(r->g, slot), r->idx, r->n, r->buf);
    PF_HIP(hipGetLastError());
    PF_HIP(hipStreamSynchronize(st));  // `in` is the caller's again
    return PF_OK;
}

// device-pointer forms: `dst` / `src` are device buffers of r->n doubles owned by the caller (e.g. a torch tensor that
// RCCL sends / has received).  Enqueued on the ctx stream without a host sync: the caller orders the two streams
// (pf_sync before the peer library reads dst; the peer library's own sync before src is read here).
int pf_rows_gather_dev(pf_rows* r, int32_t slot, double* dst) {
    PF_CHECK(r && (dst || r->n == 0), PF_E_ARG, "pf_rows_gather_dev: NULL argument");
    PF_TRY(check_slots(r->g, slot, 1, "pf_rows_gather_dev"));
    if (r->n == 0) return PF_OK;
    k_rows_gather<<<nblk(r->n), PF_BLOCK, 0, r->g->ctx->stream>>>(pf_slot(r->g, slot), r->idx, r->n, dst);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

int pf_rows_scatter_dev(pf_rows* r, int32_t slot, const double* src) {
    PF_CHECK(r && (src || r->n == 0), PF_E_ARG, "pf_rows_scatter_dev: NULL argument");
    PF_TRY(check_slots(r->g, slot, 1, "pf_rows_scatter_dev"));
    if (r->n == 0) return PF_OK;
    k_rows_scatter<<<nblk(r->n), PF_BLOCK, 0, r->g->ctx->stream>>>(pf_slot(r->g, slot), r->idx, r->n, src);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

int pf_rows_set_sources(pf_rows* r, const int64_t* offsets) {
    PF_CHECK(r && (offsets || r->n == 0), PF_E_ARG, "pf_rows_set_sources: NULL argument");
    if (r->n == 0) return PF_OK;
    hipStream_t st = r->g->ctx->stream;
    PF_HIP(hipSetDevice(r->g->ctx->device));
    if (!r->off) PF_HIP(pf_malloc(st, (void**)&r->off, sizeof(int64_t) * r->n));
    PF_HIP(hipMemcpyAsync(r->off, offsets, sizeof(int64_t) * r->n, hipMemcpyHostToDevice, st));
    PF_HIP(hipStreamSynchronize(st));
    return PF_OK;
}

int pf_rows_gather2_dev(pf_rows* r, int32_t slot_a, int32_t slot_b, double* dst, int64_t stride) {
    PF_CHECK(r && (dst || r->n == 0) && stride >= r->n, PF_E_ARG, "pf_rows_gather2_dev: bad argument");
    PF_TRY(check_slots(r->g, slot_a, 1, "pf_rows_gather2_dev"));
    if (slot_b >= 0) PF_TRY(check_slots(r->g, slot_b, 1, "pf_rows_gather2_dev"));
    if (r->n == 0) return PF_OK;
    k_rows_gather2<<<nblk(r->n), PF_BLOCK, 0, r->g->ctx->stream>>>(pf_slot(r->g, slot_a), slot_b >= 0 ? pf_slot(r->g, slot_b) : nullptr,
                                                                  r->idx, r->n, stride, dst);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

int pf_rows_scatter2_dev(pf_rows* r, int32_t slot_a, int32_t slot_b, const double* src, int64_t stride) {
    PF_CHECK(r && (src || r->n == 0), PF_E_ARG, "pf_rows_scatter2_dev: NULL argument");
    PF_CHECK(r->off || r->n == 0, PF_E_STATE, "pf_rows_scatter2_dev: no source offsets (pf_rows_set_sources)");
    PF_TRY(check_slots(r->g, slot_a, 1, "pf_rows_scatter2_dev"));
    if (slot_b >= 0) PF_TRY(check_slots(r->g, slot_b, 1, "pf_rows_scatter2_dev"));
    if (r->n == 0) return PF_OK;
    k_rows_scatter2<<<nblk(r->n), PF_BLOCK, 0, r->g->ctx->stream>>>(pf_slot(r->g, slot_a), slot_b >= 0 ? pf_slot(r->g, slot_b) : nullptr,
                                                                   r->idx, r->off, r->n, stride, src);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

int pf_rows_fill(pf_rows* r, int32_t slot, double value) {
    PF_CHECK(r != nullptr, PF_E_ARG, "pf_rows_fill: NULL argument");
    PF_TRY(check_slots(r->g, slot, 1, "pf_rows_fill"));
    if (r->n == 0) return PF_OK;
    k_rows_fill<<<nblk(r->n), PF_BLOCK, 0, r->g->ctx->stream>>>(pf_slot(r->g, slot), r->idx, r->n, value);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

}  // extern "C"
