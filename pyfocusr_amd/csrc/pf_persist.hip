// A whole Chebyshev recurrence in ONE kernel: the operator stays in LDS, only the vectors travel.
//
// One step per launch (pf_operator.hip) streams the SELL-64 matrix of the graph(s) through the chip on every step:
// 26 MB per 250k-vertex mesh, ~5 us, and the L2s do not keep it from one launch to the next (PMC: the fetched bytes
// per launch equal the algorithmic bytes).  But 1/256 of that matrix is ~74 KB - less than the 160 KB of LDS a CU
// has.  So: 256 blocks (one per CU, 1024 threads), each copies its contiguous run of SELL slices of both graphs into
// LDS once, then the steps of  y_{k+1} = (2/(e rho)) (c y_k - A y_k) - y_{k-1}/rho^2  run inside the kernel with a
// grid-wide barrier between steps.  Per step only x (gathered through L2), y_{k-1}, the diagonal and the output cross
// the fabric: 8 MB instead of 26 MB per graph.  The arithmetic per row is the one of sell_op_block, operation for
// operation: results are bit-identical to the one-step-per-launch path (tests/test_gpu_parity.py).
//
// Coherence.  The 8 XCDs have private L2s and every CU a private L1; neither snoops the others.  Measured on MI355X
// (250k rows, per step): compute 2.3 us; a barrier of atomics 1.5 us; but an agent-scope release (L2 write-back) and
// acquire (L2 + L1 invalidate) by every block 5.8 + 3.7 us, and `buffer_inv sc0` does not drop the L1.  Two measures
// remove all invalidation and all write-backs:
//   * every step writes its result to a buffer NOBODY HAS READ OR WRITTEN in this kernel (a ring of PS_RING vectors per
//     graph, at most PS_RING - 2 steps per launch): no cache can hold a stale copy of a line that was never touched,
//     so readers need no invalidate; kernel boundaries (which do invalidate) recycle the ring;
//   * writers store their results with agent scope (`global_store ... sc1`: written through the XCD's L2 to memory)
//     and every wave waits for the acknowledgement before its block arrives at the barrier, so no L2 write-back is
//     needed either (one `buffer_wbl2` per XCD and step by the last block to arrive was 0.4 us slower).
// The barrier itself: blocks arrive on the counter of their XCD (32 arrivals each; same-address atomics serialise at
// ~10 ns), the last arrival of an XCD bumps the device-wide counter, which everybody polls.  6.5 us per step of a
// 250k-vertex pair, of which ~4 us are the x gathers (bound by the L1's line rate, not by latency: issuing both
// graphs' gathers together changed nothing) and ~2.5 us the chain store-acknowledge -> arrive -> count -> poll.
// Two kernels follow.  k_sell_persist is the scheme above as described (any partition of the slices over the blocks,
// x gathered through L1/L2, grid barrier).  k_sell_persist_x, preferred when the graph's 1024-row windows fit, also
// keeps x in LDS and replaces the barrier by point-to-point flags between neighbouring windows (4.3 us per step of the
// pair); its own comment explains the differences.  Both write results to fresh ring buffers with write-through stores.
// Tried and dropped: one kernel per graph (512 threads, half the LDS, two blocks per CU) on two streams, hoping one
// graph's barrier chain would hide behind the other graph's gathers: 6.9 us per step of the pair against 6.5 us for
// both graphs in one kernel (bit-identical results either way).
// Every wait is bounded: a block that waits longer than a few seconds raises the abort flag, every other block sees
// it in its own wait loop, and the kernel drains; the host reports PF_E_HIP at its next synchronisation.  One block
// per CU (grid <= CU count, checked against the occupancy query) makes all blocks resident on an idle device.
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <vector>

#include "pf_internal.h"

namespace {

// Two persistent kernels in flight on different streams could each be given part of the CUs and wait for the rest
// forever (until the bounded waits give up): only one ctx of the process uses this path at a time.
std::atomic<pf_ctx*> g_owner{nullptr};

constexpr int PS_THREADS = 1024;
constexpr int PS_SYNC_STRIDE = 32;       // uint32 words between counters (128 B: one cache line each)
constexpr int PS_SYNC_WORDS = 10 * PS_SYNC_STRIDE;  // 8 XCD counters, device counter, abort flag
constexpr int PS_MAX_WINDOWS = 256;       // blocks of the x-in-LDS kernel (one window of 1024 rows each)
constexpr int PS_RING = 256;             // result buffers per graph; a launch runs at most PS_RING - 2 steps
constexpr unsigned PS_SPIN_LIMIT = 4000000u;
constexpr size_t PS_LDS_LIMIT = 160 * 1024 - 2048;  // static __shared__ of the kernels (neighbour list, state) lives in the remainder

struct PsGraph {
    const int64_t* slice_ptr;
    const int32_t* scol;
    const double* sval;
    const double* diag;
    const double* y_prev;  // y_{k_begin-2} (unused when k_begin == 1)
    const double* y_cur;   // y_{k_begin-1}
    double* dst;           // y_degree
    double* ring;          // [PS_RING][n_pad]: y_k lives in ring[(k-1) % PS_RING] for k < degree
    int64_t n_pad;
    int64_t n_slices;
    int32_t k_begin, k_end;  // steps of this launch (1-based, inclusive); k_end < k_begin: nothing left for this graph
    int32_t degree;
    double a1, a2, shift, beta;  // step 1: a1 (c x - A x); later: a2 (c x - A x) - beta prev
};

struct PsArgs {
    PsGraph g[2];
    uint32_t* sync;
    int32_t* host_abort;  // pinned: set to 1 when a barrier wait ran out
};

// Bounded wait for *word >= target; false (and the abort flag raised) if it ran out or another block gave up.
__device__ __forceinline__ bool wait_for(uint32_t* word, unsigned target, uint32_t* abort_flag) {
    unsigned spins = 0;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > PS_SPIN_LIMIT || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    return true;
}

// Grid barrier (results were stored write-through; this wave has waited for their acknowledgement).
__device__ __forceinline__ bool grid_barrier(uint32_t* sync, unsigned xcd, unsigned per_xcd, unsigned epoch, int* s_state) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's results have been acknowledged by the L2
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t* xc = sync + xcd * PS_SYNC_STRIDE;
        uint32_t* dc = sync + 8 * PS_SYNC_STRIDE;
        uint32_t* ab = sync + 9 * PS_SYNC_STRIDE;
        if (atomicAdd(xc, 1u) + 1u == per_xcd * epoch) atomicAdd(dc, 1u);
        *s_state = wait_for(dc, 8u * epoch, ab) ? 0 : 1;
    }
    __syncthreads();
    return *s_state == 0;
}

template <int NG>
__global__ __launch_bounds__(PS_THREADS) void k_sell_persist(PsArgs a) {
    // The floating-point operations are spelled out (and contraction is off) so that they are the ones the compiler
    // forms for sell_op_block: acc = d x; acc = fma(v, x, acc)...; t = fma(c, x, -acc); r = fma(alpha, t, -(beta prev)).
#pragma clang fp contract(off)
    extern __shared__ __align__(16) unsigned char lds[];
    __shared__ int s_state;
    const unsigned G = gridDim.x, per_xcd = G >> 3;
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned blk = xcd * per_xcd + (blockIdx.x >> 3);  // every XCD owns one contiguous run of (Morton-ordered) rows
    const int tid = threadIdx.x;
    const int lane = tid & (PF_WAVE - 1);

    // ---- stage this block's slices: values, then columns, then slice offsets
    int64_t s_lo[2] = {0, 0};
    int32_t n_sl[2] = {0, 0};
    double* lval[2] = {nullptr, nullptr};
    int32_t* lcol[2] = {nullptr, nullptr};
    int32_t* lbase[2] = {nullptr, nullptr};
    size_t off = 0;
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const PsGraph& g = a.g[q];
        s_lo[q] = (int64_t)blk * g.n_slices / G;
        n_sl[q] = (int32_t)((int64_t)(blk + 1) * g.n_slices / G - s_lo[q]);
        const int64_t e_lo = g.slice_ptr[s_lo[q]];
        const int64_t cnt = g.slice_ptr[s_lo[q] + n_sl[q]] - e_lo;
        lval[q] = reinterpret_cast<double*>(lds + off);
        off += (size_t)cnt * sizeof(double);
        for (int64_t i = tid; i < cnt; i += PS_THREADS) lval[q][i] = g.sval[e_lo + i];
    }
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const PsGraph& g = a.g[q];
        const int64_t e_lo = g.slice_ptr[s_lo[q]];
        const int64_t cnt = g.slice_ptr[s_lo[q] + n_sl[q]] - e_lo;
        lcol[q] = reinterpret_cast<int32_t*>(lds + off);
        off += (size_t)cnt * sizeof(int32_t);
        for (int64_t i = tid; i < cnt; i += PS_THREADS) lcol[q][i] = g.scol[e_lo + i];
    }
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const PsGraph& g = a.g[q];
        const int64_t e_lo = g.slice_ptr[s_lo[q]];
        lbase[q] = reinterpret_cast<int32_t*>(lds + off);
        off += (size_t)(n_sl[q] + 1) * sizeof(int32_t);
        for (int i = tid; i <= n_sl[q]; i += PS_THREADS) lbase[q][i] = (int32_t)(g.slice_ptr[s_lo[q] + i] - e_lo);
    }
    __syncthreads();

    int32_t n_steps = a.g[0].k_end - a.g[0].k_begin + 1;
    if (NG > 1 && a.g[1].k_end - a.g[1].k_begin + 1 > n_steps) n_steps = a.g[1].k_end - a.g[1].k_begin + 1;

    constexpr int JP = 4;  // pairs of entries gathered up front per row (widths up to 9); wider rows finish in a loop
    for (int32_t t = 0; t < n_steps; ++t) {
        // per-graph vectors of this step
        const double* xq[NG];
        const double* pq[NG];
        double* oq[NG];
        double alq[NG];
        int32_t rows[NG];
        int32_t most_rows = 0;
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const PsGraph& g = a.g[q];
            const int32_t k = g.k_begin + t;
            xq[q] = t == 0 ? g.y_cur : g.ring + (int64_t)((k - 2) % PS_RING) * g.n_pad;
            pq[q] = k == 1 ? nullptr : (t == 0 ? g.y_prev : (t == 1 ? g.y_cur : g.ring + (int64_t)((k - 3) % PS_RING) * g.n_pad));
            oq[q] = k == g.degree ? g.dst : g.ring + (int64_t)((k - 1) % PS_RING) * g.n_pad;
            alq[q] = k == 1 ? g.a1 : g.a2;
            rows[q] = k <= g.k_end ? n_sl[q] * PF_WAVE : 0;  // a graph whose recurrence is over sits the step out
            most_rows = rows[q] > most_rows ? rows[q] : most_rows;
        }
        // The rows of BOTH graphs a thread owns are gathered before any of them is summed: the step is bound by the
        // latency of the x gathers, and this keeps twice as many in flight.  Per row the operations and their order
        // are those of sell_op_block.
        for (int32_t r = tid; r < most_rows; r += PS_THREADS) {
            bool act[NG];
            int64_t row[NG];
            int32_t base[NG];
            int width[NG], pairs[NG];
            double xi[NG], dg[NG], pv[NG];
            double2 vv[NG][JP];
            double xa[NG][JP], xb[NG][JP];
            double vt[NG], xt[NG];
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                act[q] = r < rows[q];
                const int32_t rr = act[q] ? r : lane;  // an idle lane reads (and discards) a row of the first slice
                const int32_t sl = rr >> 6;
                row[q] = (s_lo[q] << 6) + rr;
                base[q] = lbase[q][sl];
                width[q] = act[q] ? (lbase[q][sl + 1] - base[q]) >> 6 : 0;
                pairs[q] = width[q] >> 1;
                xi[q] = xq[q][row[q]];
                dg[q] = a.g[q].diag[row[q]];
                pv[q] = pq[q] ? pq[q][row[q]] : 0.0;
#pragma unroll
                for (int j = 0; j < JP; ++j) {
                    const bool on = j < pairs[q];
                    const int32_t e = on ? base[q] + j * (2 * PF_WAVE) + 2 * lane : 0;
                    const int2 c0 = *reinterpret_cast<const int2*>(lcol[q] + e);
                    vv[q][j] = *reinterpret_cast<const double2*>(lval[q] + e);
                    xa[q][j] = xq[q][on ? c0.x : (int32_t)row[q]];
                    xb[q][j] = xq[q][on ? c0.y : (int32_t)row[q]];
                }
                const bool odd = (width[q] & 1) && pairs[q] <= JP;
                const int32_t e = odd ? base[q] + pairs[q] * (2 * PF_WAVE) + lane : 0;
                vt[q] = lval[q][e];
                xt[q] = xq[q][odd ? lcol[q][e] : (int32_t)row[q]];
            }
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                double acc = dg[q] * xi[q];
#pragma unroll
                for (int j = 0; j < JP; ++j) {
                    const double s0 = __builtin_fma(vv[q][j].x, xa[q][j], acc);
                    const double s1 = __builtin_fma(vv[q][j].y, xb[q][j], s0);
                    acc = j < pairs[q] ? s1 : acc;
                }
                if (pairs[q] > JP) {  // wide rows: the remaining pairs (and the odd entry) in order
                    const double2* vp2 = reinterpret_cast<const double2*>(lval[q] + base[q]) + lane;
                    const int2* cp2 = reinterpret_cast<const int2*>(lcol[q] + base[q]) + lane;
                    for (int j = JP; j < pairs[q]; ++j) {
                        const int2 c0 = cp2[j * PF_WAVE];
                        const double2 v0 = vp2[j * PF_WAVE];
                        acc = __builtin_fma(v0.x, xq[q][c0.x], acc);
                        acc = __builtin_fma(v0.y, xq[q][c0.y], acc);
                    }
                    if (width[q] & 1) {
                        const int32_t e = base[q] + pairs[q] * (2 * PF_WAVE) + lane;
                        acc = __builtin_fma(lval[q][e], xq[q][lcol[q][e]], acc);
                    }
                } else if (width[q] & 1) {
                    acc = __builtin_fma(vt[q], xt[q], acc);
                }
                const double u = __builtin_fma(a.g[q].shift, xi[q], -acc);
                double res;
                if (pq[q]) {
                    const double w = a.g[q].beta * pv[q];
                    res = __builtin_fma(alq[q], u, -w);
                } else {
                    res = alq[q] * u;
                }
                // agent-scope store: written through the XCD's L2 to memory, so the barrier needs no L2 write-back
                if (act[q]) __hip_atomic_store(&oq[q][row[q]], res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (t + 1 < n_steps && !grid_barrier(a.sync, xcd, per_xcd, (unsigned)(t + 1), &s_state)) {
            if (tid == 0) *a.host_abort = 1;
            return;
        }
    }
}

// ---- variant with x in LDS as well ------------------------------------------------------------------------------
// A block owns a WINDOW of 1024 consecutive rows (one row per thread and graph).  Besides the window's SELL entries
// (columns as 16-bit window-local slots: own row, or 1024 + index into the window's sorted list of outside rows,
// pf_window_slots_prepare) LDS holds the x values of the window's own rows - updated in place from the results, after
// the barrier's first __syncthreads - and of its outside rows, fetched from global memory at the start of a step
// (~220 values per window).  All gathers of a step are LDS reads; y_{k-1} of a row is the x the same thread used one
// step earlier (a register); the diagonal is a register.  Per step a block reads only its outside rows from memory.
// Same operations in the same order as sell_op_block: bit-identical results.
struct PxGraph {
    const int64_t* slice_ptr;
    const int32_t* slot;     // [sell_entries] window-local slots
    const int32_t* gh_cnt;   // [windows]
    const int32_t* gh_row;   // [windows][PF_TS_GHOSTS]
    const double* sval;
    const double* diag;
    const double* y_prev;
    const double* y_cur;
    double* dst;
    double* ring;
    int64_t n_pad;
    int32_t n_windows;
    int32_t k_begin, k_end, degree;
    double a1, a2, shift, beta;
};

struct PxArgs {
    PxGraph g[2];
    uint32_t* sync;
    int32_t* host_abort;
};

template <int NG>
__global__ __launch_bounds__(PS_THREADS) void k_sell_persist_x(PxArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __align__(16) unsigned char lds[];
    __shared__ int s_state;
    const unsigned G = gridDim.x, per_xcd = G >> 3;
    const unsigned xcd = blockIdx.x & 7u;
    const int32_t win = (int32_t)(xcd * per_xcd + (blockIdx.x >> 3));
    const int tid = threadIdx.x;
    const int lane = tid & (PF_WAVE - 1);
    const int sl = tid >> 6;

    bool have[NG];
    int32_t ghosts[NG];
    double* lval[NG];
    double* xl[NG];
    int32_t* lbase[NG];
    unsigned short* lslot[NG];
    const int32_t* ghr[NG];
    int64_t row[NG];
    double dg[NG], pv[NG];
    size_t off = 0;
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const PxGraph& g = a.g[q];
        have[q] = win < g.n_windows;
        const int64_t s0 = (int64_t)(have[q] ? win : 0) * (PF_TS_ROWS / PF_WAVE);
        const int64_t e_lo = g.slice_ptr[s0];
        const int64_t cnt = have[q] ? g.slice_ptr[s0 + PF_TS_ROWS / PF_WAVE] - e_lo : 0;
        ghosts[q] = have[q] ? g.gh_cnt[win] : 0;
        ghr[q] = g.gh_row + (int64_t)(have[q] ? win : 0) * PF_TS_GHOSTS;
        row[q] = (int64_t)(have[q] ? win : 0) * PF_TS_ROWS + tid;
        lval[q] = reinterpret_cast<double*>(lds + off);
        off += (size_t)cnt * sizeof(double);
        xl[q] = reinterpret_cast<double*>(lds + off);
        off += (size_t)(have[q] ? PF_TS_ROWS + ghosts[q] : 0) * sizeof(double);
        lbase[q] = reinterpret_cast<int32_t*>(lds + off);
        off += (size_t)(PF_TS_ROWS / PF_WAVE + 2) * sizeof(int32_t);  // 18 words: keeps the 8-byte alignment
        lslot[q] = reinterpret_cast<unsigned short*>(lds + off);
        off = (off + (size_t)cnt * sizeof(unsigned short) + 15) & ~(size_t)15;  // the next graph's values: 16-byte aligned
        for (int64_t i = tid; i < cnt; i += PS_THREADS) {
            lval[q][i] = g.sval[e_lo + i];
            lslot[q][i] = (unsigned short)g.slot[e_lo + i];
        }
        if (tid <= PF_TS_ROWS / PF_WAVE) lbase[q][tid] = have[q] ? (int32_t)(g.slice_ptr[s0 + tid] - e_lo) : 0;
        dg[q] = have[q] ? g.diag[row[q]] : 0.0;
        pv[q] = (have[q] && g.y_prev) ? g.y_prev[row[q]] : 0.0;
        if (have[q]) xl[q][tid] = g.y_cur[row[q]];
    }
    // the windows whose results this one reads (owners of its outside rows, either graph): the only blocks it has to
    // wait for - no grid-wide barrier
    __shared__ uint32_t nb_bits[PS_MAX_WINDOWS / 32];
    __shared__ int32_t nb_list[PS_MAX_WINDOWS];
    __shared__ int32_t nb_count;
    if (tid < PS_MAX_WINDOWS / 32) nb_bits[tid] = 0u;
    if (tid == 0) {
        nb_count = 0;
        s_state = 0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NG; ++q)
        for (int h = tid; h < ghosts[q]; h += PS_THREADS) {
            const int32_t w = ghr[q][h] / PF_TS_ROWS;
            atomicOr(&nb_bits[w >> 5], 1u << (w & 31));
        }
    __syncthreads();
    if (tid < PS_MAX_WINDOWS && ((nb_bits[tid >> 5] >> (tid & 31)) & 1u)) nb_list[atomicAdd(&nb_count, 1)] = tid;
    __syncthreads();
    const int32_t n_nb = nb_count;
    uint32_t* flags = a.sync + PS_SYNC_WORDS;            // [PS_MAX_WINDOWS] one 128-byte line each: steps finished
    uint32_t* ab = a.sync + 9 * PS_SYNC_STRIDE;

    int32_t n_steps = a.g[0].k_end - a.g[0].k_begin + 1;
    if (NG > 1 && a.g[1].k_end - a.g[1].k_begin + 1 > n_steps) n_steps = a.g[1].k_end - a.g[1].k_begin + 1;
    constexpr int JP = 4;

    for (int32_t t = 0; t < n_steps; ++t) {
        bool step[NG];
        double res[NG], xi[NG];
        // outside rows of this step: y_{k-1} of other windows, written (write-through) before the last barrier
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const PxGraph& g = a.g[q];
            const int32_t k = g.k_begin + t;
            step[q] = have[q] && k <= g.k_end;
            const double* xg = t == 0 ? g.y_cur : g.ring + (int64_t)((k - 2) % PS_RING) * g.n_pad;
            if (step[q])
                for (int h = tid; h < ghosts[q]; h += PS_THREADS) xl[q][PF_TS_ROWS + h] = xg[ghr[q][h]];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const PxGraph& g = a.g[q];
            const int32_t k = g.k_begin + t;
            const int32_t base = lbase[q][sl];
            const int width = step[q] ? (lbase[q][sl + 1] - base) >> 6 : 0;
            const int pairs = width >> 1;
            xi[q] = step[q] ? xl[q][tid] : 0.0;
            double acc = dg[q] * xi[q];
#pragma unroll
            for (int j = 0; j < JP; ++j) {
                const bool on = j < pairs;
                const int32_t e = on ? base + j * (2 * PF_WAVE) + 2 * lane : 0;
                const unsigned int two = *reinterpret_cast<const unsigned int*>(lslot[q] + e);
                const double2 v = *reinterpret_cast<const double2*>(lval[q] + e);
                const double x0 = xl[q][on ? (two & 0xffffu) : 0], x1 = xl[q][on ? (two >> 16) : 0];
                const double s0 = __builtin_fma(v.x, x0, acc);
                const double s1 = __builtin_fma(v.y, x1, s0);
                acc = on ? s1 : acc;
            }
            for (int j = JP; j < pairs; ++j) {  // wide rows
                const int32_t e = base + j * (2 * PF_WAVE) + 2 * lane;
                const unsigned int two = *reinterpret_cast<const unsigned int*>(lslot[q] + e);
                const double2 v = *reinterpret_cast<const double2*>(lval[q] + e);
                acc = __builtin_fma(v.x, xl[q][two & 0xffffu], acc);
                acc = __builtin_fma(v.y, xl[q][two >> 16], acc);
            }
            if (width & 1) {
                const int32_t e = base + pairs * (2 * PF_WAVE) + lane;
                acc = __builtin_fma(lval[q][e], xl[q][lslot[q][e]], acc);
            }
            const double u = __builtin_fma(g.shift, xi[q], -acc);
            if (k == 1) {
                res[q] = g.a1 * u;
            } else {
                const double w = g.beta * pv[q];
                res[q] = __builtin_fma(g.a2, u, -w);
            }
            double* out = k == g.degree ? g.dst : g.ring + (int64_t)((k - 1) % PS_RING) * g.n_pad;
            if (step[q]) __hip_atomic_store(&out[row[q]], res[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (t + 1 == n_steps) break;
        // ---- publish "step t done" and wait for the neighbours' (point to point: a window only ever reads the windows
        // of its outside rows); the window's own x is replaced by the results in between.  Results are in buffers
        // nobody touched before, so a window running a step ahead of a distant one harms nobody.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's results have been acknowledged
        __syncthreads();  // ... every wave's; and every thread has finished reading xl for this step
#pragma unroll
        for (int q = 0; q < NG; ++q)
            if (step[q]) {
                xl[q][tid] = res[q];
                pv[q] = xi[q];
            }
        if (tid == 0) __hip_atomic_store(flags + (int64_t)win * PS_SYNC_STRIDE, (uint32_t)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = tid; i < n_nb; i += PS_THREADS)
            if (!wait_for(flags + (int64_t)nb_list[i] * PS_SYNC_STRIDE, (uint32_t)(t + 1), ab)) s_state = 1;
        __syncthreads();
        if (s_state != 0) {
            if (tid == 0) *a.host_abort = 1;
            return;
        }
    }
}

// -1 undecided (environment PF_PERSIST=0 disables), 0 off, 1 on
int g_persist = -1;

bool persist_enabled() {
    if (g_persist < 0) {
        const char* v = getenv("PF_PERSIST");
        g_persist = (v && v[0] == '0') ? 0 : ((v && v[0] == '2') ? 2 : 1);
    }
    return g_persist >= 1;
}

struct DeviceFacts {
    int grid = 0;  // 0: persistent path unavailable on this device
    bool ready = false;
};
DeviceFacts g_facts[64];

int device_grid(int device) {
    if (device < 0 || device >= 64) return 0;
    DeviceFacts& f = g_facts[device];
    if (!f.ready) {
        f.ready = true;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_sell_persist<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)PS_LDS_LIMIT) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_sell_persist<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)PS_LDS_LIMIT) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_sell_persist_x<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)PS_LDS_LIMIT) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_sell_persist_x<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)PS_LDS_LIMIT) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sell_persist<2>, PS_THREADS, PS_LDS_LIMIT) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            return 0;
        }
        f.grid = std::min(prop.multiProcessorCount, 256) & ~7;  // one block per CU, a multiple of the 8 XCDs
    }
    return f.grid;
}

// -1 undecided (environment PF_PERSIST_X=0 disables the variant with x in LDS), 0 off, 1 on
int g_x_state = -1;
bool x_enabled() {
    if (g_x_state < 0) {
        const char* v = getenv("PF_PERSIST_X");
        g_x_state = (v && v[0] == '0') ? 0 : 1;
    }
    return g_x_state == 1;
}

// LDS bytes of the fullest window for the x-in-LDS kernel (layout as in k_sell_persist_x), both graphs together
int64_t lds_need_x(pf_graph* ga, pf_graph* gb) {
    pf_graph* gs[2] = {ga, gb};
    const int64_t wa = ga->n_pad / PF_TS_ROWS, wb = gb ? gb->n_pad / PF_TS_ROWS : 0;
    int64_t worst = 0;
    for (int64_t w = 0; w < std::max(wa, wb); ++w) {
        int64_t need = 0;
        for (int q = 0; q < 2; ++q) {
            pf_graph* g = gs[q];
            if (!g) continue;
            if (g->h_slice_ptr.empty() || (int64_t)g->h_px_gh_cnt.size() * PF_TS_ROWS != g->n_pad) return -1;
            need += (PF_TS_ROWS / PF_WAVE + 2) * 4;
            if (w < g->n_pad / PF_TS_ROWS) {
                const int64_t s0 = w * (PF_TS_ROWS / PF_WAVE);
                const int64_t cnt = g->h_slice_ptr[(size_t)(s0 + PF_TS_ROWS / PF_WAVE)] - g->h_slice_ptr[(size_t)s0];
                need += cnt * 8 + (PF_TS_ROWS + g->h_px_gh_cnt[(size_t)w]) * 8 + cnt * 2;
            }
            need = (need + 15) & ~(int64_t)15;
        }
        worst = std::max(worst, need);
    }
    return worst;
}

// LDS bytes the fullest block needs for g when the slices are split over `grid` blocks
int64_t lds_need(pf_graph* g, int grid) {
    if (g->h_slice_ptr.empty()) {
        g->h_slice_ptr.resize((size_t)g->n_slices + 1);
        if (hipMemcpyAsync(g->h_slice_ptr.data(), g->slice_ptr, sizeof(int64_t) * (g->n_slices + 1), hipMemcpyDeviceToHost,
                           g->ctx->stream) != hipSuccess ||
            hipStreamSynchronize(g->ctx->stream) != hipSuccess) {
            (void)hipGetLastError();
            g->h_slice_ptr.clear();
            return -1;
        }
        g->persist_grid = 0;
    }
    if (g->persist_grid != grid) {
        int64_t worst = 0;
        for (int b = 0; b < grid; ++b) {
            const int64_t lo = (int64_t)b * g->n_slices / grid, hi = (int64_t)(b + 1) * g->n_slices / grid;
            const int64_t cnt = g->h_slice_ptr[(size_t)hi] - g->h_slice_ptr[(size_t)lo];
            worst = std::max(worst, cnt * 12 + (hi - lo + 1) * 4);
        }
        g->persist_lds = worst;
        g->persist_grid = grid;
    }
    return g->persist_lds;
}

}  // namespace

int pf_persist_set(int on) {
    g_persist = on == 2 ? 2 : (on ? 1 : 0);
    return PF_OK;
}

extern "C" int pf_persist_enable(int on) { return pf_persist_set(on); }

// Runs the recurrence(s) in one kernel if the device, the sizes and the switch allow it.  *done = 1 when it was
// launched; 0 means "use the one-step-per-launch path" (never an error by itself).
int pf_persist_cheb(const pf_persist_args* a, const pf_persist_args* b, int* done) {
    *done = 0;
    if (!persist_enabled()) return PF_OK;
    pf_graph* ga = a->g;
    pf_ctx* ctx = ga->ctx;
    const int32_t longest = std::max(a->degree, b ? b->degree : 0);
    if (longest < 8) return PF_OK;  // staging the matrix must pay for itself
    PF_HIP(hipSetDevice(ctx->device));  // function attributes, occupancy queries and launches below are per device
    int grid = device_grid(ctx->device);
    if (grid < 8) return PF_OK;
    // Preferred: x in LDS as well (windows of 1024 rows; needs the window-local slots of the graph(s), built once)
    int64_t need = 0;
    bool use_x = false;
    if (x_enabled()) {
        const int64_t wa = ga->n_pad / PF_TS_ROWS, wb = b ? b->g->n_pad / PF_TS_ROWS : 0;
        const int64_t gx = (std::max(wa, wb) + 7) & ~(int64_t)7;
        const bool in_range = gx >= 8 && gx <= grid;  // (do not build window slots for graphs that cannot use them)
        if (in_range) {
            PF_TRY(pf_window_slots_prepare(ga));
            if (b) PF_TRY(pf_window_slots_prepare(b->g));
        }
        if (in_range && ga->px_state == 1 && (!b || b->g->px_state == 1)) {
            {
                const int64_t nx = lds_need_x(ga, b ? b->g : nullptr);
                if (nx > 0 && (size_t)nx + 64 <= PS_LDS_LIMIT) {
                    use_x = true;
                    need = nx;
                    grid = (int)gx;
                }
            }
        }
    }
    if (!use_x) {
        // Without x in LDS one graph alone gains nothing (250k rows: 5.0 us per step here, 4.7 us per launch there: the
        // barrier costs what the matrix traffic saves; with x in LDS it is 3.0 us); two graphs share every barrier
        // (6.5 vs 10.1 us).  pf_persist_enable(2) forces it.
        if (!b && g_persist != 2) return PF_OK;
        if (ga->n_slices < grid || (b && b->g->n_slices < grid)) return PF_OK;
        need = lds_need(ga, grid);
        if (need < 0) return PF_OK;
        if (b) {
            const int64_t nb = lds_need(b->g, grid);
            if (nb < 0) return PF_OK;
            need += nb;
        }
        if ((size_t)need + 64 > PS_LDS_LIMIT) return PF_OK;
    }
    {
        pf_ctx* expected = nullptr;
        if (!g_owner.compare_exchange_strong(expected, ctx) && expected != ctx) return PF_OK;  // another ctx owns the path
    }
    hipStream_t st = ctx->stream;
    if (!ctx->persist_sync) {
        PF_HIP(pf_malloc(st, (void**)&ctx->persist_sync, sizeof(uint32_t) * (PS_SYNC_WORDS + PS_MAX_WINDOWS * PS_SYNC_STRIDE)));
        PF_HIP(hipHostMalloc((void**)&ctx->persist_abort, sizeof(int32_t), hipHostMallocDefault));
        *ctx->persist_abort = 0;
    }
    const pf_persist_args* in[2] = {a, b};
    const int ng = b ? 2 : 1;
    for (int q = 0; q < ng; ++q) {
        pf_graph* g = in[q]->g;
        if (!g->persist_ring) PF_HIP(pf_malloc(st, (void**)&g->persist_ring, sizeof(double) * (size_t)PS_RING * (size_t)g->n_pad));
    }
    int32_t finished[2] = {0, 0};
    bool launched = false;
    while (finished[0] < a->degree || (b && finished[1] < b->degree)) {
        PsArgs args{};
        PxArgs xargs{};
        for (int q = 0; q < ng; ++q) {
            pf_graph* g = in[q]->g;
            PsGraph& p = args.g[q];
            auto where = [&](int32_t k) -> const double* {  // y_k: the caller's src, or its ring slot
                return k == 0 ? in[q]->src : g->persist_ring + (int64_t)((k - 1) % PS_RING) * g->n_pad;
            };
            p.slice_ptr = g->slice_ptr;
            p.scol = g->scol;
            p.sval = in[q]->vals;
            p.diag = g->diag;
            p.k_begin = finished[q] + 1;
            p.k_end = std::min(in[q]->degree, finished[q] + PS_RING - 2);
            p.y_cur = where(p.k_begin - 1);
            p.y_prev = p.k_begin >= 2 ? where(p.k_begin - 2) : nullptr;
            p.dst = in[q]->dst;
            p.ring = g->persist_ring;
            p.n_pad = g->n_pad;
            p.n_slices = g->n_slices;
            p.degree = in[q]->degree;
            p.a1 = 1.0 / (in[q]->e * in[q]->rho);
            p.a2 = 2.0 / (in[q]->e * in[q]->rho);
            p.shift = in[q]->c;
            p.beta = 1.0 / (in[q]->rho * in[q]->rho);
            finished[q] = std::max(finished[q], p.k_end);
            PxGraph& x = xargs.g[q];
            x.slice_ptr = g->slice_ptr;
            x.slot = g->px_slot;
            x.gh_cnt = g->px_gh_cnt;
            x.gh_row = g->px_gh_row;
            x.sval = p.sval;
            x.diag = p.diag;
            x.y_prev = p.y_prev;
            x.y_cur = p.y_cur;
            x.dst = p.dst;
            x.ring = p.ring;
            x.n_pad = p.n_pad;
            x.n_windows = (int32_t)(g->n_pad / PF_TS_ROWS);
            x.k_begin = p.k_begin;
            x.k_end = p.k_end;
            x.degree = p.degree;
            x.a1 = p.a1;
            x.a2 = p.a2;
            x.shift = p.shift;
            x.beta = p.beta;
        }
        args.sync = xargs.sync = ctx->persist_sync;
        args.host_abort = xargs.host_abort = ctx->persist_abort;
        PF_HIP(hipMemsetAsync(ctx->persist_sync, 0, sizeof(uint32_t) * (PS_SYNC_WORDS + (use_x ? PS_MAX_WINDOWS * PS_SYNC_STRIDE : 0)), st));
        if (getenv("PF_PERSIST_TEST_ABORT"))  // test hook: the first barrier finds the abort flag raised (tests/test_gpu_parity.py)
            PF_HIP(hipMemsetAsync(ctx->persist_sync + 9 * PS_SYNC_STRIDE, 1, sizeof(uint32_t), st));
        // A plain launch, not hipLaunchCooperativeKernel: one block per CU is resident-able by construction (grid <=
        // CU count, the occupancy query above says one block fits a CU), a block that has to wait for a CU another
        // stream is using starts as soon as that kernel ends, and every barrier wait is bounded anyway.  (The
        // cooperative entry point runs on a separate queue whose teardown crashes rocprofv3 at process exit.)
        if (use_x && b)
            k_sell_persist_x<2><<<dim3((unsigned)grid), dim3(PS_THREADS), (size_t)(need + 64), st>>>(xargs);
        else if (use_x)
            k_sell_persist_x<1><<<dim3((unsigned)grid), dim3(PS_THREADS), (size_t)(need + 64), st>>>(xargs);
        else if (b)
            k_sell_persist<2><<<dim3((unsigned)grid), dim3(PS_THREADS), (size_t)(need + 64), st>>>(args);
        else
            k_sell_persist<1><<<dim3((unsigned)grid), dim3(PS_THREADS), (size_t)(need + 64), st>>>(args);
        const hipError_t err = hipGetLastError();
        if (err != hipSuccess) {
            g_facts[ctx->device].grid = 0;  // the classic path from now on
            PF_CHECK(!launched, PF_E_HIP, "persistent Chebyshev kernel: launch of a later segment failed: %s", hipGetErrorString(err));
            return PF_OK;
        }
        launched = true;
    }
    *done = 1;
    return PF_OK;
}

void pf_persist_release(pf_ctx* ctx) {
    pf_ctx* expected = ctx;
    g_owner.compare_exchange_strong(expected, nullptr);
}

int pf_persist_check(pf_ctx* ctx) {
    if (ctx->persist_abort && *ctx->persist_abort) {
        *ctx->persist_abort = 0;
        g_persist = 0;  // whatever kept the blocks apart, do not try again in this process
        PF_CHECK(false, PF_E_HIP, "persistent Chebyshev kernel: a grid barrier timed out (the results of that filter "
                                  "application are invalid); the one-step-per-launch path is used from now on");
    }
    return PF_OK;
}
