// A whole Chebyshev recurrence in ONE kernel: the operator stays in registers, x in LDS, only boundary values travel.
//
// One step per launch (pf_operator.hip) streams the SELL-64 matrix of the graph(s) through the chip on every step
// (26 MB per 250k-vertex mesh, ~5 us) behind a ~1.5 us kernel boundary, and a filter application is ~145 dependent
// steps.  But 1/256 of that matrix is ~60 KB - and a CU has 512 KB of vector registers.  So: one block of 1024 threads
// per CU owns a WINDOW of NW x 1024 consecutive solver-order rows of each graph (a compact patch of the surface, Morton
// order, pf_reorder.hip), every thread keeps the entries of its NW rows in registers (values + 16-bit window-local
// column slots, pf_windows.hip; rows wider than JR entries keep the rest in LDS), the window's x and the ~220 outside
// values its rows read live in LDS (double buffered: one block barrier per step), and all steps of
//   y_{k+1} = (2/(e rho)) (c y_k - A y_k) - y_{k-1}/rho^2
// run inside the kernel.  Every gather of a step is an LDS read; y_{k-1} of a row is the x the same thread used one
// step earlier (a register).  The arithmetic per row is the one of sell_op_block, operation for operation (contraction
// off, explicit fma): results are bit-identical to the one-step-per-launch path (tests/test_gpu_parity.py).
//
// Hand-off between windows.  The 8 XCDs have private L2s and every CU a private L1; neither snoops the others.  What
// crosses windows per step is small: the rows of a window that other windows read (its BOUNDARY rows, ~25 %, sorted to
// the front of the window by pf_reorder.hip).  Per graph a ring of 4 vectors carries them: step k of a launch lives in
// slot (k + phase) & 3.
//   * producer: boundary rows are computed FIRST (their waves run at raised priority) and stored with agent scope
//     (`global_store ... sc1`: written through the XCD's L2 to memory); interior rows - which nobody else reads - are
//     computed while those values travel, and never leave the CU;
//   * consumer: one lane per outside row polls that row's slot in the ring with agent-scope loads (`sc1`: always served
//     from memory side, never from a stale L1/L2 line) until the value is no longer the EMPTY sentinel (a NaN payload no
//     arithmetic produces); the value itself is the message - no flag, no fence, no second round trip.  8-byte
//     naturally aligned accesses are single-copy atomic, so there is no tearing to guard against;
//   * recycling: at step k a window writes EMPTY over its own step k-2 values.  Safe: it could only start step k after
//     receiving step k-1 from every window it reads; those windows are exactly the windows that read it
//     (pf_windows.hip makes the relation symmetric), and each of them sent step k-1 after it had consumed step k-2.
//     Every wave drains its stores (s_waitcnt vmcnt(0)) before the step's barrier, so the EMPTY of step k is in memory
//     before the same thread's step k+1 value is even issued: a reader that sees step k+1 can never afterwards see the
//     stale step k-2 value in the slot it polls for step k+2;
//   * across launches: the only slot left non-empty is the one of step degree-1; the next launch of the graph starts at
//     phase' = (phase + degree) & 3, which makes that slot "step 3" of the new launch - emptied by every window in its
//     step 1, two hand-offs before anybody polls it.
// Measured on MI355X (250k-vertex pair): see DESIGN.md section 4 - the step is bound by one memory-side hand-off
// (~1 us) plus the boundary rows' LDS gathers, not by HBM or LDS bandwidth.
//
// Every wait is bounded: a lane that polls longer than a few seconds raises the abort flag, every other block sees it
// in its own wait loop, the kernel drains, and the host reports PF_E_PERSIST_TIMEOUT at its next synchronising call
// (pf_persist_check), with the stream drained and the path suspended for a while (see g_suspended), so the caller can simply repeat the solve.
// A plain launch with grid <= CU count and one block per CU (checked against the occupancy query): all blocks are
// resident on an idle device.  hipLaunchCooperativeKernel would add only the same size check at +15-19 us of host time
// per launch (MI355X_MICROARCH.md, "coop-launch": identical residency), and cannot protect against another tenant either.
#include <cstring>
#include <map>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <climits>
#include <cstdint>
#include <atomic>
#include <mutex>
#include <vector>

#include "pf_internal.h"

namespace {

// Two resident kernels in flight on different streams could each be given part of the CUs and wait for the rest until
// the bounded waits give up: only one ctx of the process uses this path at a time.
// who may have resident kernels in flight, per device; `m` is held from the decision to launch until the launch has been
// queued, so that a second ctx - which asks the owner's STREAM whether it has run dry - can never judge the path free in between
struct DeviceOwner {
    std::mutex m;
    pf_ctx* owner = nullptr;
};
constexpr int PF_MAX_DEVICES = 64;
DeviceOwner g_owners[PF_MAX_DEVICES];
DeviceOwner& owner_of(const pf_ctx* ctx) { return g_owners[(unsigned)ctx->device % PF_MAX_DEVICES]; }
std::atomic<int> g_persist{-1};            // -1 undecided (environment PF_PERSIST=0 disables), 0 off, 1 on
std::atomic<uint64_t> g_abort_epoch{1};    // bumped when a launch aborted: every graph's ring is refilled before reuse
std::atomic<int> g_test_aborts{0};         // pf_persist_test_hook: launches that start with the abort flag raised
std::atomic<int64_t> g_launches{0}, g_launches2{0};  // resident launches of this process (all; two steps per exchange)
std::atomic<int> g_timeouts{0};            // waits that ran out
// After a timeout the path is SUSPENDED, not switched off for good: the next g_suspend_len filter applications run one step
// per launch (enough for the repeated solve and the one after it), then the resident path is tried again; every further
// timeout doubles the suspension (a device that stays shared ends up streaming almost always, one that was shared for a
// moment gets its fast path back).  pf_persist_enable(1) lifts a suspension at once, PF_PERSIST_REARM=0 makes it permanent.
std::atomic<int64_t> g_suspended{0};       // applications left before the path is tried again (0: not suspended)
std::atomic<int> g_rearms{0}, g_owner_switches{0};
int64_t suspend_length() {
    static const int64_t base = [] {
        const char* e = getenv("PF_PERSIST_REARM");
        return e ? (int64_t)atoll(e) : (int64_t)64;
    }();
    if (base <= 0) return INT64_MAX / 2;
    const int t = std::min(std::max(g_timeouts.load() - 1, 0), 10);
    return base << t;
}

constexpr int RX_THREADS = PF_WIN_THREADS;
constexpr unsigned long long RX_EMPTY = 0xFFFFDEADFFFFDEADull;  // quiet NaN with a payload arithmetic never produces
constexpr unsigned RX_EMPTY32 = 0xFFFFDEADu;
// a wait gives up after this many repeated fetches (~0.5 us each: ~0.1 s; a launch lasts ~0.2 ms, and every block of the
// grid is resident before the first wait - what a wait can run into is a device shared with another tenant)
constexpr unsigned RX_SPIN_LIMIT = 200000u;
#ifdef RX_EXP_STAMPS
constexpr size_t RX_LDS_LIMIT = 160 * 1024 - 512 - 1024;
#else
constexpr size_t RX_LDS_LIMIT = 160 * 1024 - 512;  // the kernels' static __shared__ lives in the remainder
#endif

struct RxGraph {
    const int64_t* slice_ptr;
    const int32_t* slot;     // [sell_entries] window-local slots
    const int32_t* gh_cnt;   // [windows]
    const int32_t* gh_row;   // [windows][PF_WIN_GHOSTS]
    const int32_t* need;     // [windows] leading rows other windows read
    const double* sval;
    const double* diag;
    const double* src;       // y_0
    double* dst;             // y_degree
    double* ring;            // [4][n_pad]
    int64_t n_pad;
    int32_t n_windows;
    int32_t degree;
    int32_t phase;
    double a1, a2, shift, beta;  // step 1: a1 (c x - A x); later: a2 (c x - A x) - beta prev
};

struct RxArgs {
    RxGraph g[2];
    uint32_t* abort_flag;  // device word: some wait ran out (or the test hook raised it)
    int32_t* host_abort;   // pinned: set to 1 when this launch gave up
    unsigned long long* clock;  // [2] device: sum of block 0's run times (10 ns ticks), launches counted
    unsigned long long* report;  // pinned {ticks, id} of THIS launch for the hold-back calibration, or nullptr
    unsigned long long report_id;
    int32_t hold;          // first fetch of a step not before this many 10 ns ticks (s_memrealtime) after the step began;
                           // 0: a fixed s_sleep behind the wave's rows instead
};

#ifdef RX_EXP_STAMPS  // diagnostic build only: cycle stamps of the one-step kernel's phases, per block and wave
__device__ unsigned long long g_rx_stamps[256 * 16 * 10];
// (accumulated in LDS by each wave's first lane, clock values cut to 32 bits: the kernel has no registers to spare)
#define RX_STAMP(i) \
    do { \
        const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime(); \
        if (lane == 0) s_st[wave][i] += now_ - st_prev; \
        st_prev = now_; \
    } while (0)
#define RX_MARK(i) \
    do { \
        const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime(); \
        if (lane == 0) s_st[wave][i] += now_ - st_top; \
    } while (0)
#else
#define RX_STAMP(i)
#define RX_MARK(i)
#endif
#ifndef RX_HOLD
#define RX_HOLD 16
#endif

template <int NG, int NW>
constexpr int rx_table_bytes() {
    return ((NG * NW * 16 + 1) * 4 + 15) & ~15;
}

// d x_i + sum_{j < W} v[j] x[slot j], entries in order (x_i: the caller's register); all W LDS reads are issued before the first fma waits for one
// (a loop with a per-lane bound makes the compiler wait for every LDS read separately: 9 round trips per row)
template <int W, int JR>
__device__ __forceinline__ double rx_row(const double* x, double dg, double xi, const double (&v)[JR],
                                         const unsigned (&slp)[JR / 2]) {
#pragma clang fp contract(off)
    double xs[W > 0 ? W : 1];
#pragma unroll
    for (int j = 0; j < W; ++j) xs[j] = x[(j & 1) ? (slp[j >> 1] >> 16) : (slp[j >> 1] & 0xffffu)];
    double acc = dg * xi;
#pragma unroll
    for (int j = 0; j < W; ++j) acc = __builtin_fma(v[j], xs[j], acc);
    return acc;
}

template <int JR>
__device__ __forceinline__ double rx_row_dispatch(int w_uniform, const double* x, double dg, double xi,
                                                  const double (&v)[JR], const unsigned (&slp)[JR / 2]) {
    static_assert(JR == 4 || JR == 6 || JR == 8, "register entries per row");
    switch (w_uniform) {  // a slice (= wave) has one width; at most JR of its entries are in registers
        case 0: return rx_row<0, JR>(x, dg, xi, v, slp);
        case 1: return rx_row<1, JR>(x, dg, xi, v, slp);
        case 2: return rx_row<2, JR>(x, dg, xi, v, slp);
        case 3: return rx_row<3, JR>(x, dg, xi, v, slp);
        case 4: return rx_row<4, JR>(x, dg, xi, v, slp);
        case 5: return rx_row<(JR > 5 ? 5 : JR), JR>(x, dg, xi, v, slp);
        case 6: return rx_row<(JR > 6 ? 6 : JR), JR>(x, dg, xi, v, slp);
        case 7: return rx_row<(JR > 7 ? 7 : JR), JR>(x, dg, xi, v, slp);
        default: return rx_row<JR, JR>(x, dg, xi, v, slp);
    }
}

// SW (a pair with one row per thread only): the block's two halves take the graphs in opposite order.  Graph 1's rows are
// turned by half a window (thread t has row (t + 512) mod 1024), so each half's FIRST row is a low row of "its" graph -
// the rows that are handed over - and both graphs' hand-offs are issued one row's latency into the step instead of
// graph 1's behind graph 0's (a wave's row is one dependent chain: LDS reads, 8 fma, stores; ~0.3 us).  Each half also
// fetches its own graph's outside rows.  Per-row arithmetic is untouched, so results are bit-identical.
template <int NG, int NW, int JR, bool SW = false>
__global__ __launch_bounds__(RX_THREADS) void k_cheb_resident(RxArgs a) {

    // The floating-point operations are spelled out (and contraction is off) so that they are the ones the compiler
    // forms for sell_op_block: acc = d x; acc = fma(v, x, acc)...; t = fma(c, x, -acc); r = fma(alpha, t, -(beta prev)).
#pragma clang fp contract(off)
    static_assert(!SW || (NG == 2 && NW == 1), "the halves swap graphs in the pair kernel with one row per thread");
    extern __shared__ __align__(16) unsigned char lds[];
    __shared__ int s_state;
    __shared__ unsigned long long s_began;  // block 0: the launch's run time on the device's 100 MHz clock, for pf_persist_clock
    if (blockIdx.x == 0 && threadIdx.x == 0) s_began = __builtin_amdgcn_s_memrealtime();
    constexpr int RB = NW * RX_THREADS;
    const unsigned G = gridDim.x, per_xcd = G >> 3;
    const unsigned xcd = blockIdx.x & 7u;
    const int32_t win = (int32_t)(xcd * per_xcd + (blockIdx.x >> 3));  // every XCD owns one contiguous run of windows
    const int tid = threadIdx.x;
    const int lane = tid & (PF_WAVE - 1);
    const int wave = tid >> 6;
    (void)wave;  // (the stamps of the diagnostic build)

    // slot q of this wave is graph gq(q); its row of that graph's window is rt[q] (+ w * RX_THREADS)
    const int half = SW ? __builtin_amdgcn_readfirstlane(tid >> 9) : 0;
#define RX_GQ(q) (SW ? ((q) ^ half) : (q))
    bool have[NG];
    int32_t ghosts[NG], need_r[NG], xlen[NG], ghrow[NG], gh_lane[NG], rt[NG];
    int64_t row0[NG];
    double* xb[NG];
    double v[NG][NW][JR];
    unsigned slp[NG][NW][JR / 2];
    double dg[NG][NW], pv[NG][NW], xc[NG][NW];
    int32_t width[NG][NW], ovoff[NG][NW];
    int64_t sbase[NG][NW];

    int32_t* ovtab = reinterpret_cast<int32_t*>(lds);
    // the graphs' x buffers lie in graph order behind the table, whichever half looks at them
    int32_t gcount[NG];
    size_t goff[NG + 1];
    goff[0] = rx_table_bytes<NG, NW>();
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const bool hj = win < a.g[j].n_windows;
        gcount[j] = hj ? a.g[j].gh_cnt[win] : 0;
        goff[j + 1] = goff[j] + (hj ? (size_t)2 * (RB + ((gcount[j] + 1) & ~1)) * sizeof(double) : 0);
    }
    const size_t off = goff[NG];
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const RxGraph& g = a.g[RX_GQ(q)];
        have[q] = win < g.n_windows;
        const int32_t wq = have[q] ? win : 0;
        ghosts[q] = SW ? (RX_GQ(q) ? gcount[NG - 1] : gcount[0]) : gcount[q];
        need_r[q] = have[q] ? ((g.need[wq] + PF_WAVE - 1) & ~(PF_WAVE - 1)) : 0;
        row0[q] = (int64_t)wq * RB;
        rt[q] = SW ? ((tid + (RX_THREADS / 2) * RX_GQ(q)) & (RX_THREADS - 1)) : tid;
        // Which thread fetches which outside row.  One graph: thread t fetches row t.  Two graphs: the second graph's
        // rows are fetched by the waves of the block's upper half (thread 512 + t fetches row t) when both lists fit a
        // half, so that the two graphs' polls are in flight TOGETHER instead of one round trip after the other.
        // (SW: each half fetches the rows of its first graph; the host launches SW only when every list fits a half)
        gh_lane[q] = tid;
        if (SW) gh_lane[q] = q == 0 ? (tid & (RX_THREADS / 2 - 1)) : -1;
        if (!SW && NG == 2) {
            if (q == 1 && gcount[0] <= RX_THREADS / 2 && gcount[NG - 1] <= RX_THREADS / 2) gh_lane[q] = tid - RX_THREADS / 2;
        }
        ghrow[q] = (gh_lane[q] >= 0 && gh_lane[q] < ghosts[q]) ? g.gh_row[(int64_t)wq * PF_WIN_GHOSTS + gh_lane[q]] : 0;
        xlen[q] = have[q] ? RB + ((ghosts[q] + 1) & ~1) : 0;
        xb[q] = reinterpret_cast<double*>(lds + (SW ? (RX_GQ(q) ? goff[NG - 1] : goff[0]) : goff[q]));
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int64_t row = row0[q] + w * RX_THREADS + rt[q];
            const int64_t s = row >> 6;
            sbase[q][w] = g.slice_ptr[s];
            width[q][w] = have[q] ? (int32_t)((g.slice_ptr[s + 1] - sbase[q][w]) >> 6) : 0;
            dg[q][w] = have[q] ? g.diag[row] : 0.0;
            pv[q][w] = 0.0;
            if (lane == 0) ovtab[(RX_GQ(q) * NW + w) * 16 + (rt[q] >> 6)] = width[q][w] > JR ? (width[q][w] - JR) * PF_WAVE : 0;
        }
    }
    if (tid == 0) s_state = (int)__hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid < PF_WAVE) {  // exclusive prefix of the waves' overflow counts (<= 128 entries): one wave, two entries per lane
        constexpr int CNT = NG * NW * 16;
        const int32_t c0 = 2 * lane < CNT ? ovtab[2 * lane] : 0, c1 = 2 * lane + 1 < CNT ? ovtab[2 * lane + 1] : 0;
        int32_t inc = c0 + c1;
#pragma unroll
        for (int off = 1; off < PF_WAVE; off <<= 1) {
            const int32_t up = __shfl_up(inc, off, PF_WAVE);
            if (lane >= off) inc += up;
        }
        const int32_t before = inc - c0 - c1;
        if (2 * lane < CNT) ovtab[2 * lane] = before;
        if (2 * lane + 1 < CNT) ovtab[2 * lane + 1] = before + c0;
        if (lane == PF_WAVE - 1) ovtab[CNT] = inc;
    }
    __syncthreads();
    if (s_state != 0) {  // aborted before it began (test hook, or an earlier launch that is still being drained)
        if (tid == 0) *a.host_abort = 1;
        return;
    }
    double* ov_val = reinterpret_cast<double*>(lds + off);
    unsigned short* ov_slot = reinterpret_cast<unsigned short*>(lds + off + (size_t)ovtab[NG * NW * 16] * sizeof(double));

    // ---- this thread's rows: entries into registers (and, beyond JR per row, into LDS), x_0 into LDS
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const RxGraph& g = a.g[RX_GQ(q)];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int32_t wd = width[q][w];
            const int32_t pairs = wd >> 1;
            const int64_t base = sbase[q][w];
            ovoff[q][w] = ovtab[(RX_GQ(q) * NW + w) * 16 + (rt[q] >> 6)];
#pragma unroll
            for (int p = 0; p < JR / 2; ++p) {
                double2 vv = make_double2(0.0, 0.0);
                int2 ss = make_int2(0, 0);
                if (p < pairs) {
                    vv = *reinterpret_cast<const double2*>(g.sval + base + (int64_t)p * (2 * PF_WAVE) + 2 * lane);
                    ss = *reinterpret_cast<const int2*>(g.slot + base + (int64_t)p * (2 * PF_WAVE) + 2 * lane);
                }
                v[q][w][2 * p] = vv.x;
                v[q][w][2 * p + 1] = vv.y;
                slp[q][w][p] = (unsigned)ss.x | ((unsigned)ss.y << 16);
            }
            if (wd & 1) {  // the odd last entry sits behind the pairs, one per lane
                const int64_t idx = base + (int64_t)pairs * (2 * PF_WAVE) + lane;
                const double tv = g.sval[idx];
                const unsigned ts = (unsigned)g.slot[idx];
                if (wd - 1 < JR) {
#pragma unroll
                    for (int p = 0; p < JR / 2; ++p)
                        if (2 * p == wd - 1) {
                            v[q][w][2 * p] = tv;
                            slp[q][w][p] = ts;
                        }
                } else {
                    const int32_t o = ovoff[q][w] + (wd - 1 - JR) * PF_WAVE + lane;
                    ov_val[o] = tv;
                    ov_slot[o] = (unsigned short)ts;
                }
            }
            for (int p = JR / 2; p < pairs; ++p) {  // wide rows: the remaining pairs, entry-major in LDS
                const int64_t idx = base + (int64_t)p * (2 * PF_WAVE) + 2 * lane;
                const double2 vv = *reinterpret_cast<const double2*>(g.sval + idx);
                const int2 ss = *reinterpret_cast<const int2*>(g.slot + idx);
                const int32_t o = ovoff[q][w] + (2 * p - JR) * PF_WAVE + lane;
                ov_val[o] = vv.x;
                ov_slot[o] = (unsigned short)ss.x;
                ov_val[o + PF_WAVE] = vv.y;
                ov_slot[o + PF_WAVE] = (unsigned short)ss.y;
            }
            xc[q][w] = have[q] ? g.src[row0[q] + w * RX_THREADS + rt[q]] : 0.0;
            if (have[q]) xb[q][w * RX_THREADS + rt[q]] = xc[q][w];
        }
        if (gh_lane[q] >= 0 && gh_lane[q] < ghosts[q]) xb[q][RB + gh_lane[q]] = g.src[ghrow[q]];
    }
    __syncthreads();

    // the recurrences' scalars, per slot.  (SW: read through a wave-dependent index, they would be fetched from the
    // argument segment again in every step - they are pinned to scalar registers here, once)
    struct RxLoop {
        double* ring;
        double* dst;
        int64_t n_pad;
        int32_t degree, phase;
        double shift, a1, a2, beta;
    } L[NG];
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const RxGraph& gg = a.g[RX_GQ(q)];
        L[q].ring = gg.ring;
        L[q].dst = gg.dst;
        L[q].n_pad = gg.n_pad;
        L[q].degree = gg.degree;
        L[q].phase = gg.phase;
        L[q].shift = gg.shift;
        L[q].a1 = gg.a1;
        L[q].a2 = gg.a2;
        L[q].beta = gg.beta;
        if (SW) {
            long long s0 = __double_as_longlong(L[q].shift), s1 = __double_as_longlong(L[q].a1), s2 = __double_as_longlong(L[q].a2),
                      s3 = __double_as_longlong(L[q].beta);
            asm volatile("" : "+s"(L[q].ring), "+s"(L[q].dst), "+s"(L[q].n_pad), "+s"(L[q].degree), "+s"(L[q].phase));
            asm volatile("" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
            L[q].shift = __longlong_as_double(s0);
            L[q].a1 = __longlong_as_double(s1);
            L[q].a2 = __longlong_as_double(s2);
            L[q].beta = __longlong_as_double(s3);
        }
    }
    int32_t n_steps = a.g[0].degree;
    if (NG > 1 && a.g[1].degree > n_steps) n_steps = a.g[1].degree;
    bool early = (rt[0] & ~(PF_WAVE - 1)) < need_r[0];  // this wave owns boundary rows (its w = 0 rows)
    if (NG > 1) early = early || (rt[NG - 1] & ~(PF_WAVE - 1)) < need_r[NG - 1];
    int cur = 0;
#ifdef RX_EXP_STAMPS
    __shared__ unsigned s_st[16][10];
    if (lane < 10) s_st[wave][lane] = 0;
    unsigned st_prev = (unsigned)__builtin_amdgcn_s_memtime(), st_top = 0;
#endif

    for (int32_t k = 1; k <= n_steps; ++k) {
        RX_STAMP(5);  // loop overhead (state check, branch)
#ifdef RX_EXP_STAMPS
        st_top = st_prev;
#endif
        if (early) __builtin_amdgcn_s_setprio(3);
        const unsigned step_top = (unsigned)__builtin_amdgcn_s_memrealtime();
        // (threads with four rows: the row number is "new" in every step, so that the compiler forms a row's LDS and memory
        // addresses where it uses them instead of keeping ~6 registers per row alive across the whole recurrence, and
        // spilling; 1M-row pair, k = 10: eigensolves 73.0 -> 69.8 ms)
        int32_t rtk[NG];
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            rtk[q] = rt[q];
            if (NG * NW >= 4) asm volatile("" : "+v"(rtk[q]));
        }
#pragma unroll
        for (int w = 0; w < NW; ++w) {
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                const RxLoop& g = L[q];
                if (!have[q] || k > g.degree) continue;  // (wave-uniform) a graph whose recurrence is over sits the step out
                const double* x = xb[q] + (size_t)cur * xlen[q];
                double* xn = xb[q] + (size_t)(cur ^ 1) * xlen[q];
                const int32_t lr = w * RX_THREADS + rtk[q];
                const int32_t wd = width[q][w];
                const double xi = xc[q][w];  // the row's own x: last step's result, still in its register
                double acc = rx_row_dispatch<JR>(__builtin_amdgcn_readfirstlane(wd < JR ? wd : JR), x, dg[q][w], xi, v[q][w], slp[q][w]);
#ifndef RX_OV_PAIRS
#define RX_OV_PAIRS 1
#endif
                int j = JR;
#if RX_OV_PAIRS
                for (; j + 1 < wd; j += 2) {  // (two entries' value, slot and x reads in flight together; fmas in entry order)
                    const int32_t o = ovoff[q][w] + (j - JR) * PF_WAVE + lane;
                    const double a0 = ov_val[o], a1 = ov_val[o + PF_WAVE];
                    const unsigned s0 = ov_slot[o], s1 = ov_slot[o + PF_WAVE];
                    const double x0 = x[s0], x1 = x[s1];
                    acc = __builtin_fma(a0, x0, acc);
                    acc = __builtin_fma(a1, x1, acc);
                }
#endif
                for (; j < wd; ++j) {
                    const int32_t o = ovoff[q][w] + (j - JR) * PF_WAVE + lane;
                    acc = __builtin_fma(ov_val[o], x[ov_slot[o]], acc);
                }
                const double u = __builtin_fma(g.shift, xi, -acc);
                double res;
                if (k == 1) {
                    res = g.a1 * u;
                } else {
                    const double wp = g.beta * pv[q][w];
                    res = __builtin_fma(g.a2, u, -wp);
                }
                pv[q][w] = xi;
                xc[q][w] = res;
                xn[lr] = res;
                const int64_t row = row0[q] + lr;
                if (k == g.degree) g.dst[row] = res;
                if (lr < need_r[q]) {  // (wave-uniform) a boundary row: hand it over, and recycle the slot of step k-2
                    unsigned long long* ring = reinterpret_cast<unsigned long long*>(g.ring);
                    if (k < g.degree)
                        __hip_atomic_store(ring + (int64_t)((k + g.phase) & 3) * g.n_pad + row,
                                           (unsigned long long)__double_as_longlong(res), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(ring + (int64_t)((k + 2 + g.phase) & 3) * g.n_pad + row, RX_EMPTY, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                    RX_MARK(6 + q);  // hand-off of graph q issued
                }
            }
            if (w == 0 && early) __builtin_amdgcn_s_setprio(0);
        }
        RX_STAMP(0);  // rows computed, hand-off stores issued
        if (k == n_steps) break;
        // ---- the outside values of step k, straight from their owners' stores.  The owners stored them about when this
        // block stored its own, and an agent-scope store takes ~0.5 us to land: a poll issued at once would just miss it
        // and cost a second round trip, so the first poll is held back by about that long - the interior rows of the
        // other waves are being computed meanwhile.
        // (measured at 250k rows, s_sleep units of 64 cycles: one graph 2.22 us per step polling at once, 1.42 held back by
        // 16; a pair with the two graphs' polls on the two halves of the block: 2.46 / 2.25 / 2.04 / 1.93 / 2.03 / 2.14
        // held back by 4 / 8 / 12 / 16 / 20 / 24)
        bool slept = false;
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const RxLoop& g = L[q];
            if (have[q] && k < g.degree && gh_lane[q] >= 0 && gh_lane[q] < ghosts[q]) {
                if (a.hold > 0) {
                    // (asleep while more than 60 ns remain, then watching the clock)
                    while ((int)((unsigned)a.hold - ((unsigned)__builtin_amdgcn_s_memrealtime() - step_top)) > 6) __builtin_amdgcn_s_sleep(1);
                    while ((unsigned)__builtin_amdgcn_s_memrealtime() - step_top < (unsigned)a.hold) {
                    }
                } else if (RX_HOLD > 0 && !slept) {
                    __builtin_amdgcn_s_sleep(RX_HOLD);  // (every wave that fetches, once)
                }
                slept = true;
                RX_STAMP(1);  // hold-back
                const unsigned long long* p =
                    reinterpret_cast<const unsigned long long*>(g.ring) + (int64_t)((k + g.phase) & 3) * g.n_pad + ghrow[q];
                unsigned long long bits;
                unsigned spins = 0;
                while ((bits = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == RX_EMPTY) {
                    ++spins;
                    if (spins > RX_SPIN_LIMIT ||
                        ((spins & 63u) == 0u && __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                        __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s_state = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                xb[q][(size_t)(cur ^ 1) * xlen[q] + RB + gh_lane[q]] = __longlong_as_double((long long)bits);
#ifdef RX_EXP_STAMPS
                if (lane == 0) {
                    s_st[wave][8] += spins;                          // repeats of the wave's first lane
                }
                if (__ballot(spins > 0) != 0 && lane == 0) s_st[wave][9] += 1;  // steps in which some lane had to repeat
#endif
            }
        }
        RX_STAMP(2);  // polls returned (first tries and repeats)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's hand-offs and EMPTY stores are in memory (see header)
        RX_STAMP(3);  // own stores drained
        __syncthreads();
        RX_STAMP(4);  // barrier
        if (s_state != 0) {
            if (tid == 0) *a.host_abort = 1;
            return;
        }
        cur ^= 1;
    }
    if (blockIdx.x == 0 && tid == 0) {
        const unsigned long long ran = (unsigned long long)__builtin_amdgcn_s_memrealtime() - s_began;
        atomicAdd(a.clock, ran);
        atomicAdd(a.clock + 1, 1ull);
        if (a.report) {  // the run time, then the id that says it is there (both 8-byte stores to pinned host memory)
            __hip_atomic_store(a.report, ran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(a.report + 1, a.report_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
#ifdef RX_EXP_STAMPS
    if (lane == 0 && blockIdx.x < 256)
        for (int i = 0; i < 10; ++i) g_rx_stamps[((size_t)blockIdx.x * 16 + wave) * 10 + i] = s_st[wave][i];
#endif
}

#ifdef RX_EXP_STAMPS
extern "C" int pf_persist_stamps(unsigned long long* out /* [256][16][10] */) {
    PF_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rx_stamps), sizeof(unsigned long long) * 256 * 16 * 10));
    return PF_OK;
}
#endif

// ---------------------------------------------------------------------------------------------------------------
// Two recurrence steps per exchange (k_cheb_resident2).
//
// A step of k_cheb_resident is bound by one memory-side hand-off (store lands ~0.5 us, load returns ~0.4 us) that the
// interior rows cover only in part.  Here a window also keeps its RING 1 - the outside rows its own rows read - as
// rows it computes itself (their entries and slots in LDS, pf_windows.hip: k_win_rings), and RING 2, the rows those
// read.  A round = two steps:
//   phase A  y_{2r-1} on the own rows AND on ring 1, from buffer 0 (own | ring 1 | ring 2 of y_{2r-2}) into buffer 1
//   phase B  y_{2r} on the own rows, from buffer 1 (own | ring 1) into buffer 0; the leading rows that some other window
//            holds in either ring are published (boundary rows and the rows behind them come first in a window,
//            pf_reorder.hip), the interior rows are computed while they travel, then rings 1 and 2 of y_{2r} are fetched.
// One hand-off per TWO steps, for ~20 % more rows in phase A.  A ring-1 row is computed from its owner's entries in its
// owner's order (padding included): the same bits as the owner's.  The protocol of the hand-off is k_cheb_resident's,
// with rounds in place of steps: the ring slot of round r is (r + phase) & 3, a window empties its slot of round r-2 in
// round r, and a launch of R rounds (R = ceil(degree / 2); the last one publishes nothing) leaves exactly the slot of
// round R-1 filled, which is "round 3" of the next launch at phase' = (phase + R) & 3.  Needs "A holds rows of B <=> B
// holds rows of A" for both rings (k_win_rings_check; true whenever W is symmetric).  Own ring buffers: the set of
// published rows differs from k_cheb_resident's, and the two kernels may alternate on one graph.
struct Rx2Graph {
    const int64_t* slice_ptr;
    const int32_t* slot;      // [sell_entries] slots of the own rows' entries (own | ring 1)
    const int32_t* gh_cnt;    // [windows] rows of ring 1
    const int32_t* gh_cnt2;   // [windows] rows of ring 2
    const int32_t* gh_row;    // [windows][PF_WIN_GHOSTS] ring 1, then ring 2
    const int32_t* need2;     // [windows] leading rows other windows hold in a ring
    const uint8_t* g1_w;      // [windows][PF_WIN_G1]
    const int32_t* g1_pos;    // [windows][PF_WIN_GW][PF_WIN_G1]
    const uint16_t* g1_slot;  // same shape
    const int32_t* g1_gw;     // [windows]
    const double* sval;
    const double* diag;
    const double* src;
    double* dst;
    double* ring;             // [4][n_pad]
    int64_t n_pad;
    int32_t n_windows;
    int32_t degree;
    int32_t phase;
    double a1, a2, shift, beta;
};

struct Rx2Args {
    Rx2Graph g[2];
    uint32_t* abort_flag;
    int32_t* host_abort;
    unsigned long long* clock;  // as RxArgs::clock
    int32_t hold;  // first fetch of a round not before this many 10 ns ticks after its phase B began; 0: a fixed s_sleep
};

#ifndef RX2_JR_PAIR
#define RX2_JR_PAIR 8  // entries of a row kept in registers when a thread holds a row of each of two graphs
#endif
#define RX2_JR(ng) ((ng) == 2 ? RX2_JR_PAIR : 8)
#ifndef RX2_GB
#define RX2_GB 4  // entries of a ring-1 row whose LDS reads are in flight together
#endif
#ifndef RX2_HOLD1
#define RX2_HOLD1 16  // one graph: first poll of a round held back by this many x 64 cycles
#endif
#ifndef RX2_HOLD2
#define RX2_HOLD2 8   // two graphs: the same for the lower half's waves (they have just published; the upper half's arrive late anyway)
#endif

template <int NG>
__global__ __launch_bounds__(RX_THREADS) void k_cheb_resident2(Rx2Args a) {
#pragma clang fp contract(off)
    constexpr int JR = RX2_JR(NG), RB = RX_THREADS, NW = 1;
    constexpr int PH = RX_THREADS / NG;  // threads that look after one graph's outside rows (NG == 2: the halves of the block)
    extern __shared__ __align__(16) unsigned char lds[];
    __shared__ int s_state;
    __shared__ unsigned long long s_began;
    if (blockIdx.x == 0 && threadIdx.x == 0) s_began = __builtin_amdgcn_s_memrealtime();
    const unsigned G = gridDim.x, per_xcd = G >> 3;
    const unsigned xcd = blockIdx.x & 7u;
    const int32_t win = (int32_t)(xcd * per_xcd + (blockIdx.x >> 3));
    const int tid = threadIdx.x;
    const int lane = tid & (PF_WAVE - 1);
    const int wave = tid >> 6;
    const int qg = NG == 2 ? (tid >= PH ? 1 : 0) : 0;  // (wave-uniform) the graph whose outside rows this thread serves
    const int lt = tid - qg * PH;

    bool have[NG];
    int32_t g1n[NG], gall[NG], need_r[NG], gwid[NG], g1p[NG], o0[NG], o1[NG];
    int64_t row0[NG];
    double *gval[NG], *gdiag[NG];
    uint4* gslot8[NG];
    unsigned short* gslotx[NG];
    int32_t* glist[NG];
    double v[NG][JR];
    unsigned slp[NG][JR / 2];
    double dg[NG], xc[NG];
    int32_t width[NG], ovoff[NG];
    int64_t sbase[NG];

    int32_t* ovtab = reinterpret_cast<int32_t*>(lds);
    size_t off = rx_table_bytes<NG, NW>();
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const Rx2Graph& g = a.g[q];
        have[q] = win < g.n_windows;
        const int32_t wq = have[q] ? win : 0;
        g1n[q] = have[q] ? g.gh_cnt[wq] : 0;
        gall[q] = g1n[q] + (have[q] ? g.gh_cnt2[wq] : 0);
        need_r[q] = have[q] ? ((g.need2[wq] + PF_WAVE - 1) & ~(PF_WAVE - 1)) : 0;
        gwid[q] = have[q] ? g.g1_gw[wq] : 0;
        g1p[q] = (g1n[q] + PF_WAVE - 1) & ~(PF_WAVE - 1);
        row0[q] = (int64_t)wq * RB;
        o0[q] = (int32_t)off;  // buffer 0: own | ring 1 | ring 2
        off += have[q] ? (size_t)(RB + ((gall[q] + 1) & ~1)) * sizeof(double) : 0;
        o1[q] = (int32_t)off;  // buffer 1: own | ring 1
        off += have[q] ? (size_t)(RB + ((g1n[q] + 1) & ~1)) * sizeof(double) : 0;
        gslot8[q] = reinterpret_cast<uint4*>(lds + off);  // (16-byte aligned: everything before it is a multiple of 16)
        off += (size_t)g1p[q] * sizeof(uint4);
        gval[q] = reinterpret_cast<double*>(lds + off);
        off += (size_t)gwid[q] * g1p[q] * sizeof(double);
        gdiag[q] = reinterpret_cast<double*>(lds + off);
        off += (size_t)g1p[q] * sizeof(double);
        glist[q] = reinterpret_cast<int32_t*>(lds + off);
        off += (size_t)((gall[q] + 3) & ~3) * sizeof(int32_t);
        gslotx[q] = reinterpret_cast<unsigned short*>(lds + off);
        off += (size_t)(gwid[q] > 8 ? gwid[q] - 8 : 0) * g1p[q] * sizeof(unsigned short);
        const int64_t row = row0[q] + tid;
        const int64_t s = row >> 6;
        sbase[q] = g.slice_ptr[s];
        width[q] = __builtin_amdgcn_readfirstlane(have[q] ? (int32_t)((g.slice_ptr[s + 1] - sbase[q]) >> 6) : 0);
        dg[q] = have[q] ? g.diag[row] : 0.0;
        if (lane == 0) ovtab[q * 16 + wave] = width[q] > JR ? (width[q] - JR) * PF_WAVE : 0;
    }
    if (tid == 0) s_state = (int)__hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid == 0) {
        int32_t run = 0;
        for (int i = 0; i < NG * 16; ++i) {
            const int32_t c = ovtab[i];
            ovtab[i] = run;
            run += c;
        }
        ovtab[NG * 16] = run;
    }
    __syncthreads();
    if (s_state != 0) {
        if (tid == 0) *a.host_abort = 1;
        return;
    }
    double* ov_val = reinterpret_cast<double*>(lds + off);
    unsigned short* ov_slot = reinterpret_cast<unsigned short*>(lds + off + (size_t)ovtab[NG * 16] * sizeof(double));

    // ---- own rows: entries into registers (beyond JR per row: LDS), y_0 into buffer 0
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const Rx2Graph& g = a.g[q];
        const int32_t wd = width[q];
        const int32_t pairs = wd >> 1;
        const int64_t base = sbase[q];
        ovoff[q] = __builtin_amdgcn_readfirstlane(ovtab[q * 16 + wave]);
#pragma unroll
        for (int p = 0; p < JR / 2; ++p) {
            double2 vv = make_double2(0.0, 0.0);
            int2 ss = make_int2(0, 0);
            if (p < pairs) {
                vv = *reinterpret_cast<const double2*>(g.sval + base + (int64_t)p * (2 * PF_WAVE) + 2 * lane);
                ss = *reinterpret_cast<const int2*>(g.slot + base + (int64_t)p * (2 * PF_WAVE) + 2 * lane);
            }
            v[q][2 * p] = vv.x;
            v[q][2 * p + 1] = vv.y;
            slp[q][p] = (unsigned)ss.x | ((unsigned)ss.y << 16);
        }
        if (wd & 1) {
            const int64_t idx = base + (int64_t)pairs * (2 * PF_WAVE) + lane;
            const double tv = g.sval[idx];
            const unsigned ts = (unsigned)g.slot[idx];
            if (wd - 1 < JR) {
#pragma unroll
                for (int p = 0; p < JR / 2; ++p)
                    if (2 * p == wd - 1) {
                        v[q][2 * p] = tv;
                        slp[q][p] = ts;
                    }
            } else {
                const int32_t o = ovoff[q] + (wd - 1 - JR) * PF_WAVE + lane;
                ov_val[o] = tv;
                ov_slot[o] = (unsigned short)ts;
            }
        }
        for (int p = JR / 2; p < pairs; ++p) {
            const int64_t idx = base + (int64_t)p * (2 * PF_WAVE) + 2 * lane;
            const double2 vv = *reinterpret_cast<const double2*>(g.sval + idx);
            const int2 ss = *reinterpret_cast<const int2*>(g.slot + idx);
            const int32_t o = ovoff[q] + (2 * p - JR) * PF_WAVE + lane;
            ov_val[o] = vv.x;
            ov_slot[o] = (unsigned short)ss.x;
            ov_val[o + PF_WAVE] = vv.y;
            ov_slot[o + PF_WAVE] = (unsigned short)ss.y;
        }
        xc[q] = have[q] ? g.src[row0[q] + tid] : 0.0;
        if (have[q]) reinterpret_cast<double*>(lds + o0[q])[tid] = xc[q];
    }
    // ---- outside rows this thread serves: it fetches rows lt and lt + PH of graph qg's list and repeats the recurrence of
    // ring-1 row lt.  That row's entries live in LDS (values entry-major, the first eight slots as one 16-byte record),
    // as do its diagonal and the list; its two latest values are the row's places in the two buffers.
    int32_t gwi = 0;
    bool gf0 = false, gf1 = false, grow = false;
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        if (qg != q || !have[q]) continue;  // (wave-uniform)
        const Rx2Graph& g = a.g[q];
        double* b0 = reinterpret_cast<double*>(lds + o0[q]);
        const int64_t lbase = (int64_t)win * PF_WIN_GHOSTS;
        gf0 = lt < gall[q];
        gf1 = lt + PH < gall[q];
        grow = lt < g1n[q];
        if (gf0) {
            const int32_t r0 = g.gh_row[lbase + lt];
            glist[q][lt] = r0;
            b0[RB + lt] = g.src[r0];
            if (grow) {
                gwi = g.g1_w[(int64_t)win * PF_WIN_G1 + lt];
                gdiag[q][lt] = g.diag[r0];
                unsigned sl[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) sl[j] = 0u;
                for (int32_t j = 0; j < gwi; ++j) {
                    const int64_t o = ((int64_t)win * PF_WIN_GW + j) * PF_WIN_G1 + lt;
                    gval[q][j * g1p[q] + lt] = g.sval[g.g1_pos[o]];
                    const unsigned sj = g.g1_slot[o];
                    if (j >= 8) gslotx[q][(j - 8) * g1p[q] + lt] = (unsigned short)sj;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj)
                        if (jj == j) sl[jj] = sj;
                }
                gslot8[q][lt] = make_uint4(sl[0] | (sl[1] << 16), sl[2] | (sl[3] << 16), sl[4] | (sl[5] << 16), sl[6] | (sl[7] << 16));
            }
        }
        if (gf1) {
            const int32_t r1 = g.gh_row[lbase + lt + PH];
            glist[q][lt + PH] = r1;
            b0[RB + lt + PH] = g.src[r1];
        }
    }
    __syncthreads();

    int32_t n_steps = a.g[0].degree;
    if (NG > 1 && a.g[1].degree > n_steps) n_steps = a.g[1].degree;
    const int32_t rounds = (n_steps + 1) >> 1;
    int32_t need_max = need_r[0];
    if (NG > 1 && need_r[1] > need_max) need_max = need_r[1];
    const bool early = (tid & ~(PF_WAVE - 1)) < need_max;

    // one step of this thread's row of graph q: gathers from the buffer at LDS offset xo, y_{k-2} is what the target
    // buffer still holds at the row's place.  (The buffer's address is hidden from the optimiser once per phase: with two
    // fixed buffers it would keep both sets of gather addresses in registers across the loop and spill the entries.)
    auto own_step = [&](int q, int32_t xo_in, int32_t yo_in, bool first) -> double {
        const Rx2Graph& g = a.g[q];
        int32_t xo = xo_in, yo = yo_in;
        asm volatile("" : "+s"(xo), "+s"(yo));
        const double* x = reinterpret_cast<const double*>(lds + xo);
        double* y = reinterpret_cast<double*>(lds + yo);
        const int32_t wd = width[q];
        const double xi = xc[q];
        const double prev = y[tid];
        double acc = rx_row_dispatch<JR>(wd < JR ? wd : JR, x, dg[q], xi, v[q], slp[q]);
        for (int j = JR; j < wd; ++j) {
            const int32_t o = ovoff[q] + (j - JR) * PF_WAVE + lane;
            acc = __builtin_fma(ov_val[o], x[ov_slot[o]], acc);
        }
        const double u = __builtin_fma(g.shift, xi, -acc);
        double res;
        if (first) {
            res = g.a1 * u;
        } else {
            const double wp = g.beta * prev;
            res = __builtin_fma(g.a2, u, -wp);
        }
        xc[q] = res;
        y[tid] = res;
        return res;
    };

    for (int32_t r = 1; r <= rounds; ++r) {
        const int32_t kA = 2 * r - 1, kB = 2 * r;
        // ---- phase A: step kA of ring 1 and of the own rows, buffer 0 -> buffer 1.  The ring-1 rows come first: their reads
        // are two dependent LDS round trips (slots, then x), which the waves without such rows cover with their own rows
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const Rx2Graph& g = a.g[q];
            if (qg != q || !have[q] || kA >= g.degree) continue;  // (wave-uniform) ring 1 only if a phase B follows
            if (grow) {
                int32_t xo = o0[q], yo = o1[q];
                asm volatile("" : "+s"(xo), "+s"(yo));
                const double* x = reinterpret_cast<const double*>(lds + xo);
                double* y = reinterpret_cast<double*>(lds + yo);
                int32_t ld = g1p[q];
                asm volatile("" : "+s"(ld));  // (entry addresses formed here, not kept in registers across the rounds)
                const double* gv = gval[q] + lt;
                const uint4 gsl = gslot8[q][lt];
                const double gx = x[RB + lt];
                const double gprev = y[RB + lt];
                const unsigned sw[4] = {gsl.x, gsl.y, gsl.z, gsl.w};
                double acc = gdiag[q][lt] * gx;
#pragma unroll
                for (int h = 0; h < 8; h += RX2_GB) {  // RX2_GB entries' reads in flight together (unused places read slot 0 / entry 0)
                    double xs[RX2_GB], vs[RX2_GB];
#pragma unroll
                    for (int j = 0; j < RX2_GB; ++j) {
                        xs[j] = x[((h + j) & 1) ? (sw[(h + j) >> 1] >> 16) : (sw[(h + j) >> 1] & 0xffffu)];
                        vs[j] = gv[(h + j < gwi ? h + j : 0) * ld];
                    }
#pragma unroll
                    for (int j = 0; j < RX2_GB; ++j)
                        if (h + j < gwi) acc = __builtin_fma(vs[j], xs[j], acc);
                }
                for (int32_t j = 8; j < gwi; ++j) acc = __builtin_fma(gv[j * ld], x[gslotx[q][(j - 8) * ld + lt]], acc);
                const double u = __builtin_fma(g.shift, gx, -acc);
                double res;
                if (kA == 1) {
                    res = g.a1 * u;
                } else {
                    const double wp = g.beta * gprev;
                    res = __builtin_fma(g.a2, u, -wp);
                }
                y[RB + lt] = res;
            }
        }
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const Rx2Graph& g = a.g[q];
            if (!have[q] || kA > g.degree) continue;  // (block-uniform)
            const double res = own_step(q, o0[q], o1[q], kA == 1);
            if (kA == g.degree) {  // an odd degree ends here
                (g.dst + row0[q])[tid] = res;
                if (tid < need_r[q])
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(g.ring) + ((int64_t)((r + 2 + g.phase) & 3) * g.n_pad + row0[q]) + tid,
                                       RX_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        // ---- phase B: step kB of the own rows, buffer 1 -> buffer 0; leading rows published first
        const unsigned b_top = (unsigned)__builtin_amdgcn_s_memrealtime();
        if (early) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const Rx2Graph& g = a.g[q];
            if (!have[q] || kB > g.degree) continue;  // (block-uniform)
            const double res = own_step(q, o1[q], o0[q], false);
            if (kB == g.degree) (g.dst + row0[q])[tid] = res;
            if (tid < need_r[q]) {  // (wave-uniform)
                unsigned long long* ring = reinterpret_cast<unsigned long long*>(g.ring) + row0[q];  // (uniform bases + the thread's index)
                if (kB < g.degree)
                    __hip_atomic_store(ring + (int64_t)((r + g.phase) & 3) * g.n_pad + tid, (unsigned long long)__double_as_longlong(res),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ring + (int64_t)((r + 2 + g.phase) & 3) * g.n_pad + tid, RX_EMPTY, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (early) __builtin_amdgcn_s_setprio(0);
        if (r == rounds) break;
        // ---- rings 1 and 2 of step kB, straight from their owners' stores (two per thread in flight)
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const Rx2Graph& g = a.g[q];
            if (qg != q || !have[q] || kB >= g.degree) continue;  // (wave-uniform)
            if (gf0) {
                if (a.hold > 0) {
                    while ((int)((unsigned)a.hold - ((unsigned)__builtin_amdgcn_s_memrealtime() - b_top)) > 6) __builtin_amdgcn_s_sleep(1);
                    while ((unsigned)__builtin_amdgcn_s_memrealtime() - b_top < (unsigned)a.hold) {
                    }
                } else if (q == 0 && (NG == 1 ? RX2_HOLD1 : RX2_HOLD2) > 0) {
                    __builtin_amdgcn_s_sleep(NG == 1 ? RX2_HOLD1 : RX2_HOLD2);
                }
                const unsigned long long* p0 = reinterpret_cast<const unsigned long long*>(g.ring) + (int64_t)((r + g.phase) & 3) * g.n_pad;
                const int32_t ghr0 = glist[q][lt];
                const unsigned long long* p1 = p0 + (gf1 ? glist[q][lt + PH] : ghr0);
                p0 += ghr0;
                unsigned long long v0 = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long v1 = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (v0 == RX_EMPTY || v1 == RX_EMPTY) {
                    ++spins;
                    if (spins > RX_SPIN_LIMIT ||
                        ((spins & 63u) == 0u && __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                        __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s_state = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    if (v0 == RX_EMPTY) v0 = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v1 == RX_EMPTY) v1 = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                double* b0 = reinterpret_cast<double*>(lds + o0[q]);
                b0[RB + lt] = __longlong_as_double((long long)v0);
                if (gf1) b0[RB + lt + PH] = __longlong_as_double((long long)v1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s_state != 0) {
            if (tid == 0) *a.host_abort = 1;
            return;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        atomicAdd(a.clock, (unsigned long long)__builtin_amdgcn_s_memrealtime() - s_began);
        atomicAdd(a.clock + 1, 1ull);
    }
}

bool persist_enabled() {
    int v = g_persist.load();
    if (v < 0) {
        const char* e = getenv("PF_PERSIST");
        v = (e && e[0] == '0') ? 0 : 1;
        g_persist.store(v);
    }
    return v >= 1;
}

}  // namespace
bool pf_persist_enabled() { return persist_enabled(); }
uint64_t pf_persist_abort_epoch() { return g_abort_epoch.load(); }
bool pf_persist_trusted() { return persist_enabled() && g_suspended.load() <= 0; }  // (no bounded wait has run out lately)
namespace {

// a filter application is about to run: false while a suspension lasts (counted down here, once per application:
// `first_try` is false for the further attempts pf_cheb2 makes for the same application)
bool persist_available(bool first_try) {
    if (!persist_enabled()) return false;
    int64_t left = g_suspended.load();
    if (left <= 0) return true;
    if (first_try) {
        left = g_suspended.fetch_sub(1) - 1;
        if (left == 0) g_rearms.fetch_add(1);  // the NEXT application tries the resident path again
    }
    return false;
}

// Only one ctx per DEVICE may have resident kernels in flight (two grids would each get part of the CUs and wait for the
// rest).  The path belongs to the ctx that used it last; another ctx of the same device takes it over when the owner's
// last resident launch has completed, and runs this application one step per launch otherwise - it does not wait.
// The guard holds the device's lock until it goes out of scope - behind the launch and the record of its completion
// event - so that "the owner's last launch has completed" cannot be judged between another thread's decision to launch
// and its launch (round 3's check-then-act: two resident grids, each waiting for the other's CUs until the spin limit).
struct OwnerGuard {
    std::unique_lock<std::mutex> lk;
    bool ok = false;
    explicit OwnerGuard(pf_ctx* ctx) : lk(owner_of(ctx).m) {
        DeviceOwner& d = owner_of(ctx);
        if (d.owner == ctx) {
            ok = true;
        } else if (d.owner == nullptr) {
            d.owner = ctx;
            ok = true;
        } else if (hipStreamQuery(d.owner->stream) == hipSuccess) {
            // (the owner's stream has run dry: none of its resident launches is in flight.  Asking the stream instead of an
            // event recorded behind every resident launch - rounds 3-4 - keeps ~3 us of event packet out of every outer
            // step of the owner; the price is that a would-be owner waits for whatever else that stream holds too)
            d.owner = ctx;
            g_owner_switches.fetch_add(1);
            ok = true;
        } else {
            (void)hipGetLastError();  // still running (or queued)
        }
    }
};

int persist_launched(pf_ctx* ctx) {  // (nothing to mark: a would-be owner asks the owner's stream, see OwnerGuard)
    (void)ctx;
    return PF_OK;
}

using RxKernel = void (*)(RxArgs);
// entries of a row kept in registers, by kernel shape: 8 where the register file allows it (a thread holds NG x NW rows)
constexpr int rx_jr(int ng, int nw) { return ng * nw <= 2 ? 8 : 4; }
RxKernel rx_kernel(int ng, int nw) {
    if (ng == 1 && nw == 1) return k_cheb_resident<1, 1, rx_jr(1, 1)>;
    if (ng == 2 && nw == 1) return k_cheb_resident<2, 1, rx_jr(2, 1)>;
    if (ng == 1 && nw == 2) return k_cheb_resident<1, 2, rx_jr(1, 2)>;
    if (ng == 2 && nw == 2) return k_cheb_resident<2, 2, rx_jr(2, 2)>;
    if (ng == 1 && nw == 4) return k_cheb_resident<1, 4, rx_jr(1, 4)>;
    return nullptr;
}

struct DeviceFacts {
    std::once_flag once;
    int grid = 0;  // 0: resident path unavailable on this device
};
DeviceFacts g_facts[64];

int device_grid(int device) {
    if (device < 0 || device >= 64) return 0;
    DeviceFacts& f = g_facts[device];
    std::call_once(f.once, [&] {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) return;
        const int shapes[5][2] = {{1, 1}, {2, 1}, {1, 2}, {2, 2}, {1, 4}};
        for (const auto& s : shapes) {
            const void* fn = reinterpret_cast<const void*>(rx_kernel(s[0], s[1]));
            int per_cu = 0;
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RX_LDS_LIMIT) != hipSuccess ||
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, RX_THREADS, RX_LDS_LIMIT) != hipSuccess || per_cu < 1) {
                (void)hipGetLastError();
                return;
            }
        }
        for (int ng = 1; ng <= 3; ++ng) {
            const void* fn = ng == 1   ? reinterpret_cast<const void*>(k_cheb_resident2<1>)
                             : ng == 2 ? reinterpret_cast<const void*>(k_cheb_resident2<2>)
                                       : reinterpret_cast<const void*>(k_cheb_resident<2, 1, rx_jr(2, 1), true>);
            int per_cu = 0;
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RX_LDS_LIMIT) != hipSuccess ||
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, RX_THREADS, RX_LDS_LIMIT) != hipSuccess || per_cu < 1) {
                (void)hipGetLastError();
                return;
            }
        }
        f.grid = std::min(prop.multiProcessorCount, 256) & ~7;  // one block per CU, a multiple of the 8 XCDs
    });
    return f.grid;
}

// -1 undecided (environment PF_PERSIST_S2 = 0 / 1 / 2), 0 off, 1 single-graph launches (default), 2 paired launches too.
// Measured on MI355X at 250k rows (profiles/r03_two_step_ablation.md): one graph 1.41 -> 1.24 us per step; a pair 1.85 ->
// 2.08: with two graphs per block the step is bound by LDS issue and vector ALU work, not by the hand-off, and the
// repeated ring-1 rows (20 LDS reads each against 10 of an own row) cost more than the saved hand-off.
std::atomic<int> g_last_hold{0};  // the hold-back of the latest one-step resident launch (pf_persist_state)
std::atomic<int> g_two_step{-1};
int two_step_level() {
    int v = g_two_step.load();
    if (v < 0) {
        const char* e = getenv("PF_PERSIST_S2");
        v = e ? (e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 1)) : 1;
        g_two_step.store(v);
    }
    return v;
}
bool two_step_enabled() { return two_step_level() >= 1; }

// LDS bytes of the fullest window of k_cheb_resident2 (layout as in the kernel); -1 if the graph(s) cannot use it
int64_t lds_need2(pf_graph* ga, pf_graph* gb) {
    const int jr = RX2_JR(gb ? 2 : 1);
    pf_graph* gs[2] = {ga, gb};
    const int ng = gb ? 2 : 1;
    const int64_t RB = RX_THREADS;
    const int64_t ph = RX_THREADS / ng;
    const int64_t wa = ga->n_pad / RB, wb = gb ? gb->n_pad / RB : 0;
    int64_t worst = 0;
    for (int64_t w = 0; w < std::max(wa, wb); ++w) {
        int64_t need = ((ng * 16 + 1) * 4 + 15) & ~15, ov = 0;
        for (int q = 0; q < ng; ++q) {
            pf_graph* g = gs[q];
            if (g->px2_state != 1 || g->win_rows != RB || g->h_slice_ptr.empty() || (int64_t)g->h_px_gh_cnt.size() * RB != g->n_pad) return -1;
            if (w >= g->n_pad / RB) continue;
            const int64_t g1 = g->h_px_gh_cnt[(size_t)w], g2 = g->h_px_gh_cnt2[(size_t)w], gw = g->h_px_g1_gw[(size_t)w];
            if (g1 > ph || g1 + g2 > 2 * ph) return -1;  // a thread repeats one ring-1 row and fetches two outside rows
            const int64_t g1p = (g1 + PF_WAVE - 1) & ~(int64_t)(PF_WAVE - 1);
            need += (RB + ((g1 + g2 + 1) & ~1)) * 8 + (RB + ((g1 + 1) & ~1)) * 8 + g1p * 16 + gw * g1p * 8 + g1p * 8 +
                    ((g1 + g2 + 3) & ~3) * 4 + (gw > 8 ? gw - 8 : 0) * g1p * 2;
            for (int64_t s = w * (RB / PF_WAVE); s < (w + 1) * (RB / PF_WAVE); ++s) {
                const int64_t width = (g->h_slice_ptr[(size_t)s + 1] - g->h_slice_ptr[(size_t)s]) / PF_WAVE;
                if (width > jr) ov += (width - jr) * PF_WAVE;
            }
        }
        need += ov * 10 + 16;
        worst = std::max(worst, need);
    }
    return worst;
}

// two steps per exchange, if both graphs and the sizes allow it: *done = 1 when launched
// When a block first asks for the outside values of a round (see hold_ticks below): here for the two-step kernel, counted from the start of its phase B (environment PF_PERSIST_HOLD2)
int hold_ticks2(int ng, int windows) {
    if (const char* e = getenv("PF_PERSIST_HOLD2")) return atoi(e);
    (void)ng;
    (void)windows;
    return 0;
}

int persist_cheb2(const pf_persist_args* a, const pf_persist_args* b, int64_t grid, int* done, double* lds_bytes) {
    *done = 0;
    pf_graph* ga = a->g;
    pf_graph* gb = b ? b->g : nullptr;
    pf_ctx* ctx = ga->ctx;
    if (two_step_level() < (gb ? 2 : 1) || ga->win_rows != RX_THREADS) return PF_OK;
    // the second-ring structures cost ~0.1 ms to build (a kernel and a read-back): a graph pays that at its THIRD
    // single-graph application, not for the one or two a paired solve leaves over when its partner converges first
    if (!gb && ga->px2_state < 0 && ++ga->single_applications < 3) return PF_OK;
    PF_TRY(pf_window_rings_prepare(ga));
    if (gb) PF_TRY(pf_window_rings_prepare(gb));
    if (ga->px2_state != 1 || (gb && gb->px2_state != 1)) return PF_OK;
    if (ga->lds_need2_partner != (gb ? gb->uid : ga->uid) || ga->lds_need2_value == -2) {
        ga->lds_need2_value = lds_need2(ga, gb);
        ga->lds_need2_partner = gb ? gb->uid : ga->uid;
    }
    const int64_t need = ga->lds_need2_value;
    if (need < 0 || (size_t)need > RX_LDS_LIMIT) return PF_OK;
    OwnerGuard own(ctx);
    if (!own.ok) return PF_OK;
    hipStream_t st = ctx->stream;
    PF_TRY(pf_persist_sync_ensure(ctx));
    const uint64_t epoch = g_abort_epoch.load();
    const pf_persist_args* in[2] = {a, b};
    const int ng = gb ? 2 : 1;
    Rx2Args args{};
    for (int q = 0; q < ng; ++q) {
        pf_graph* g = in[q]->g;
        if (!g->persist_ring2) {
            PF_HIP(pf_malloc(st, (void**)&g->persist_ring2, sizeof(double) * 4 * (size_t)g->n_pad));
            g->persist_epoch2 = 0;
        }
        if (g->persist_epoch2 != epoch) {
            PF_HIP(hipMemsetD32Async((hipDeviceptr_t)g->persist_ring2, (int)RX_EMPTY32, (size_t)8 * (size_t)g->n_pad, st));
            g->persist_epoch2 = epoch;
            g->persist_phase2 = 0;
        }
        Rx2Graph& p = args.g[q];
        p.slice_ptr = g->slice_ptr;
        p.slot = g->px_slot;
        p.gh_cnt = g->px_gh_cnt;
        p.gh_cnt2 = g->px_gh_cnt2;
        p.gh_row = g->px_gh_row;
        p.need2 = g->px_need2;
        p.g1_w = g->px_g1_w;
        p.g1_pos = g->px_g1_pos;
        p.g1_slot = g->px_g1_slot;
        p.g1_gw = g->px_g1_gw;
        p.sval = in[q]->vals;
        p.diag = g->diag;
        p.src = in[q]->src;
        p.dst = in[q]->dst;
        p.ring = g->persist_ring2;
        p.n_pad = g->n_pad;
        p.n_windows = (int32_t)(g->n_pad / g->win_rows);
        p.degree = in[q]->degree;
        p.phase = g->persist_phase2;
        p.a1 = 1.0 / (in[q]->e * in[q]->rho);
        p.a2 = 2.0 / (in[q]->e * in[q]->rho);
        p.shift = in[q]->c;
        p.beta = 1.0 / (in[q]->rho * in[q]->rho);
    }
    args.abort_flag = ctx->persist_sync;
    args.host_abort = ctx->persist_abort;
    args.clock = reinterpret_cast<unsigned long long*>(ctx->persist_sync + 8);
    if (g_test_aborts.load() > 0) {
        g_test_aborts.fetch_sub(1);
        PF_HIP(hipMemsetAsync(ctx->persist_sync, 1, sizeof(uint32_t), st));
    }
    args.hold = hold_ticks2(ng, (int)grid);
    if (ng == 2) k_cheb_resident2<2><<<dim3((unsigned)grid), dim3(RX_THREADS), (size_t)need, st>>>(args);
    else k_cheb_resident2<1><<<dim3((unsigned)grid), dim3(RX_THREADS), (size_t)need, st>>>(args);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        g_two_step.store(0);  // one step per exchange from now on
        (void)hipGetLastError();
        return PF_OK;
    }
    PF_TRY(persist_launched(ctx));
    g_launches.fetch_add(1);
    g_launches2.fetch_add(1);
    for (int q = 0; q < ng; ++q) {
        pf_graph* g = in[q]->g;
        const int32_t d = in[q]->degree, rounds = (d + 1) / 2;
        g->persist_phase2 = (g->persist_phase2 + rounds) & 3;
        // LDS bytes: per step and own row the gathered x (8 per stored entry) and the result (8); per round the ring-1 rows'
        // entries (value 8 + slot 2 + gathered x 8 each) and results (8), and the fetched rows of both rings (8)
        if (lds_bytes)
            *lds_bytes += (double)d * (8.0 * (double)g->sell_entries + 8.0 * (double)g->n_pad) +
                          (double)(d / 2) * (18.0 * (double)g->px_g1_entries + 8.0 * (double)g->px_gh_total) +
                          (double)((d - 1) / 2) * 8.0 * (double)(g->px_gh_total + g->px_gh2_total);
    }
    *done = 1;
    return PF_OK;
}

// LDS bytes of the fullest window (layout as in k_cheb_resident); -1 if the graph(s) cannot use the kernel
int64_t lds_need(pf_graph* ga, pf_graph* gb, int nw) {
    const int jr = rx_jr(gb ? 2 : 1, nw);
    pf_graph* gs[2] = {ga, gb};
    const int ng = gb ? 2 : 1;
    const int64_t RB = (int64_t)nw * RX_THREADS;
    const int64_t wa = ga->n_pad / RB, wb = gb ? gb->n_pad / RB : 0;
    int64_t worst = 0;
    for (int64_t w = 0; w < std::max(wa, wb); ++w) {
        int64_t need = ((ng * nw * 16 + 1) * 4 + 15) & ~15, ov = 0;
        for (int q = 0; q < ng; ++q) {
            pf_graph* g = gs[q];
            if (g->h_slice_ptr.empty() || (int64_t)g->h_px_gh_cnt.size() * RB != g->n_pad) return -1;
            if (w >= g->n_pad / RB) continue;
            need += 2 * (RB + ((g->h_px_gh_cnt[(size_t)w] + 1) & ~1)) * 8;
            for (int64_t s = w * (RB / PF_WAVE); s < (w + 1) * (RB / PF_WAVE); ++s) {
                const int64_t width = (g->h_slice_ptr[(size_t)s + 1] - g->h_slice_ptr[(size_t)s]) / PF_WAVE;
                if (width > jr) ov += (width - jr) * PF_WAVE;
            }
        }
        need += ov * 10 + 16;
        worst = std::max(worst, need);
    }
    return worst;
}

}  // namespace

int pf_persist_set(int on) {
    g_persist.store(on ? 1 : 0);
    if (on) g_suspended.store(0);  // an explicit "on" lifts a suspension
    return PF_OK;
}

extern "C" int pf_persist_enable(int on) { return pf_persist_set(on); }

// 1: two recurrence steps per exchange where a graph allows it (default), 0: one (k_cheb_resident)
extern "C" int pf_persist_two_step(int level) {
    g_two_step.store(level <= 0 ? 0 : (level >= 2 ? 2 : 1));
    return PF_OK;
}

// 1 (default): paired recurrences on windows of 1024 rows use the kernel whose halves take the graphs in opposite order
std::atomic<int> g_pair_halves{-1};
bool pair_halves_enabled() {
    int v = g_pair_halves.load();
    if (v < 0) {
        const char* e = getenv("PF_PERSIST_HALVES");
        v = (e && e[0] == '0') ? 0 : 1;
        g_pair_halves.store(v);
    }
    return v != 0;
}
extern "C" int pf_persist_pair_halves(int on) {
    g_pair_halves.store(on ? 1 : 0);
    return PF_OK;
}

extern "C" int pf_persist_state(pf_ctx* ctx, pf_persist_info* out) {
    PF_CHECK(out != nullptr, PF_E_ARG, "pf_persist_state: NULL argument");
    pf_ctx* owner = nullptr;
    if (ctx) {
        std::lock_guard<std::mutex> lk(owner_of(ctx).m);
        owner = owner_of(ctx).owner;
    }
    out->enabled = persist_enabled() ? 1 : 0;
    out->two_step = two_step_level();
    out->owner = owner == nullptr ? 0 : (owner == ctx ? 1 : -1);
    out->timeouts = g_timeouts.load();
    out->launches = g_launches.load();
    out->launches_two_step = g_launches2.load();
    const int64_t left = g_suspended.load();
    out->suspended_for = (int32_t)std::min<int64_t>(std::max<int64_t>(left, 0), INT32_MAX);
    out->rearms = g_rearms.load();
    out->owner_switches = g_owner_switches.load();
    out->hold_ticks = g_last_hold.load();
    return PF_OK;
}

// When a block first asks for the outside values of a step, in 10 ns ticks after the step began (RxArgs::hold).
// The values a neighbour handed over in this step land ~0.8 us into it; a fetch issued before that does not only come
// back empty, it also delays the landing (250k pair: 1.58 us per step at the best hold, 1.81 one 85 ns earlier), while a
// late one costs just its lateness - so the table sits a little to the right of the measured optimum
// (tools/tune_hold.py; profiles/r03_hold_sweep.md).  Fewer windows: less traffic, earlier landing.
// Environment PF_PERSIST_HOLD overrides (ticks; 0: the fixed s_sleep behind a wave's rows, as in round 2).
struct HoldPoint {
    int windows, ticks;
};
int hold_from(const HoldPoint* t, int n, int windows) {
    if (windows <= t[0].windows) return t[0].ticks;
    for (int i = 1; i < n; ++i)
        if (windows <= t[i].windows)
            return t[i - 1].ticks + (t[i].ticks - t[i - 1].ticks) * (windows - t[i - 1].windows) / (t[i].windows - t[i - 1].windows);
    return t[n - 1].ticks;
}
int hold_ticks(int ng, int nw, bool halves, int windows) {
    if (const char* e = getenv("PF_PERSIST_HOLD")) return atoi(e);
    if (nw == 4) return 70;  // (700k..1M rows: a wave's four rows end later than that - the fetch follows them at once)
    if (nw == 2) {
        static const HoldPoint two_rows[] = {{147, 70}, {245, 74}};  // 300k..500k rows
        return hold_from(two_rows, 2, windows);
    }
    static const HoldPoint single[] = {{20, 58}, {59, 64}, {147, 68}, {245, 72}};
    static const HoldPoint pair[] = {{20, 80}, {59, 84}, {147, 89}, {245, 94}};
    static const HoldPoint pair_halves[] = {{20, 72}, {59, 74}, {147, 82}, {245, 88}};
    return ng == 1 ? hold_from(single, 4, windows) : (halves ? hold_from(pair_halves, 4, windows) : hold_from(pair, 4, windows));
}

// ---- the hold-back, measured.  The table above is the mean of a few boxes; where a box's own optimum sits 40 ns away,
// a 250k pair pays 1-3 %.  Every ctx therefore TRIES, for each kernel shape and window count it meets, the table's value
// and its neighbours at +-4 ticks (then further out on the side that wins, to +-12) on its first launches - two launches
// each, block 0 reports its run time into a pinned slot that the host reads a launch or two later, never waiting - and
// keeps the fastest.  Results do not depend on the hold-back (test_resident_kernel_bit_identical), so the trial launches
// are ordinary filter applications.  PF_PERSIST_CAL=0 keeps the table.
struct HoldCalibration {
    static constexpr int SLOTS = 16, MAX_CAND = 7, STEP = 4, REACH = 12, SAMPLES = 3;
    struct State {
        int table = 0, n_cand = 0, next = 0, chosen = -1;
        int cand[MAX_CAND];
        double best_us[MAX_CAND];
        int seen[MAX_CAND], asked[MAX_CAND];
    };
    struct Meta {
        unsigned long long id = 0;  // 0: slot free
        uint64_t key = 0;
        int cand = 0;
        int32_t steps = 0;
    };
    unsigned long long* ring = nullptr;  // pinned: SLOTS x {ticks, id}
    Meta meta[SLOTS];
    unsigned long long next_id = 1;
    std::map<uint64_t, State> states;

    void add(State& s, int hold) {
        s.cand[s.n_cand] = hold;
        s.best_us[s.n_cand] = 1e30;
        s.seen[s.n_cand] = s.asked[s.n_cand] = 0;
        ++s.n_cand;
    }
    void harvest() {
        for (int i = 0; i < SLOTS; ++i) {
            Meta& m = meta[i];
            if (m.id == 0) continue;
            if (__atomic_load_n(ring + 2 * i + 1, __ATOMIC_ACQUIRE) != m.id) continue;  // not finished yet
            const double us = (double)ring[2 * i] * 0.01 / (double)std::max(m.steps, 1);
            auto it = states.find(m.key);
            if (it != states.end() && it->second.chosen < 0 && m.cand < it->second.n_cand) {
                State& s = it->second;
                s.best_us[m.cand] = std::min(s.best_us[m.cand], us);
                ++s.seen[m.cand];
                decide(s);
            }
            m.id = 0;
        }
    }
    void decide(State& s) {
        for (int c = 0; c < s.n_cand; ++c)
            if (s.seen[c] < SAMPLES) return;
        int best = 0;
        for (int c = 1; c < s.n_cand; ++c)
            if (s.best_us[c] < s.best_us[best]) best = c;
        int lo = s.cand[0], hi = s.cand[0];
        for (int c = 1; c < s.n_cand; ++c) lo = std::min(lo, s.cand[c]), hi = std::max(hi, s.cand[c]);
        const int b = s.cand[best];
        if (s.n_cand < MAX_CAND && b == hi && b + STEP <= s.table + REACH) {
            add(s, b + STEP);
        } else if (s.n_cand < MAX_CAND && b == lo && b - STEP >= std::max(s.table - REACH, 1)) {
            add(s, b - STEP);
        } else {
            // a neighbour of the table's value has to beat it by more than the noise of three launches (a first launch
            // beside another tenant, clocks still ramping); otherwise the table stands
            s.chosen = (b != s.table && s.best_us[best] > 0.99 * s.best_us[0]) ? s.table : b;
        }
    }
    void release(int slot) {  // the launch that was to report into `slot` was never queued
        if (slot < 0 || slot >= SLOTS || meta[slot].id == 0) return;
        auto it = states.find(meta[slot].key);
        if (it != states.end() && meta[slot].cand < it->second.n_cand && it->second.asked[meta[slot].cand] > 0) --it->second.asked[meta[slot].cand];
        meta[slot].id = 0;
    }
    // the hold-back of the launch that is about to be queued; *slot / *id: where and as what it reports (-1: it does not)
    int pick(uint64_t key, int table, int32_t steps, int* slot, unsigned long long* id) {
        *slot = -1;
        *id = 0;
        harvest();
        State& s = states[key];
        if (s.n_cand == 0) {
            s.table = table;
            add(s, table);
            add(s, table + STEP);
            if (table - STEP >= 1) add(s, table - STEP);
        }
        if (s.chosen >= 0) return s.chosen;
        int c = -1;
        for (int t = 0; t < s.n_cand; ++t) {  // the next candidate that has not been asked often enough
            const int k = (s.next + t) % s.n_cand;
            if (s.asked[k] < SAMPLES) {
                c = k;
                break;
            }
        }
        if (c < 0) return s.table;  // every sample is on its way
        int free_slot = -1;
        for (int i = 0; i < SLOTS; ++i)
            if (meta[i].id == 0) {
                free_slot = i;
                break;
            }
        if (free_slot < 0) return s.table;
        s.next = (c + 1) % s.n_cand;
        ++s.asked[c];
        Meta& m = meta[free_slot];
        m.id = next_id++;
        m.key = key;
        m.cand = c;
        m.steps = steps;
        ring[2 * free_slot] = 0;
        ring[2 * free_slot + 1] = 0;
        *slot = free_slot;
        *id = m.id;
        return s.cand[c];
    }
    void forget_pending() {  // launches that gave up never report
        for (int i = 0; i < SLOTS; ++i) {
            if (meta[i].id == 0) continue;
            auto it = states.find(meta[i].key);
            if (it != states.end() && meta[i].cand < it->second.n_cand && it->second.asked[meta[i].cand] > 0) --it->second.asked[meta[i].cand];
            meta[i].id = 0;
        }
    }
};

bool calibration_enabled() {
    static const bool on = [] { const char* e = getenv("PF_PERSIST_CAL"); return !(e && e[0] == '0'); }();
    return on && getenv("PF_PERSIST_HOLD") == nullptr;
}

HoldCalibration* calibration_of(pf_ctx* ctx) {
    if (!ctx->persist_cal) {
        auto* cal = new HoldCalibration();
        if (hipHostMalloc((void**)&cal->ring, sizeof(unsigned long long) * 2 * HoldCalibration::SLOTS, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            delete cal;
            return nullptr;
        }
        memset(cal->ring, 0, sizeof(unsigned long long) * 2 * HoldCalibration::SLOTS);
        ctx->persist_cal = cal;
    }
    return static_cast<HoldCalibration*>(ctx->persist_cal);
}

// Block 0's own view of the resident launches since the last reset: their number and their summed run time on the
// device's 100 MHz clock (s_memrealtime at the first and last instruction of the block) - the kernel without the
// dispatch and event packets that a HIP event pair around a launch also measures.  Synchronises the ctx stream.
extern "C" int pf_persist_clock(pf_ctx* ctx, double* kernel_ms, int64_t* launches, int reset) {
    PF_CHECK(ctx != nullptr && kernel_ms != nullptr && launches != nullptr, PF_E_ARG, "pf_persist_clock: NULL argument");
    *kernel_ms = 0.0;
    *launches = 0;
    if (!ctx->persist_sync) return PF_OK;
    unsigned long long h[2] = {0, 0};
    PF_HIP(hipSetDevice(ctx->device));
    PF_HIP(hipMemcpyAsync(h, ctx->persist_sync + 8, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    if (reset) PF_HIP(hipMemsetAsync(ctx->persist_sync + 8, 0, sizeof(h), ctx->stream));
    *kernel_ms = (double)h[0] * 1e-5;  // 10 ns ticks
    *launches = (int64_t)h[1];
    return PF_OK;
}

extern "C" int pf_persist_test_hook(int n_launches) {
    g_test_aborts.store(n_launches > 0 ? n_launches : 0);
    return PF_OK;
}

// Runs the recurrence(s) in one kernel if the device, the sizes and the switch allow it.  *done = 1 when it was
// launched; 0 means "use the one-step-per-launch path" (never an error by itself).
int pf_persist_cheb(const pf_persist_args* a, const pf_persist_args* b, int* done, double* lds_bytes, bool first_try) {
    *done = 0;
    if (!persist_available(first_try)) return PF_OK;
    pf_graph* ga = a->g;
    pf_graph* gb = b ? b->g : nullptr;
    pf_ctx* ctx = ga->ctx;
    const int32_t longest = std::max(a->degree, b ? b->degree : 0);
    if (longest < 8) return PF_OK;  // loading the matrix must pay for itself
    if (gb && gb->win_rows != ga->win_rows) return PF_OK;
    const int nw = ga->win_rows / RX_THREADS, ng = gb ? 2 : 1;
    RxKernel kernel = rx_kernel(ng, nw);
    if (!kernel) return PF_OK;
    PF_HIP(hipSetDevice(ctx->device));  // function attributes, occupancy queries and launches below are per device
    const int dev_grid = device_grid(ctx->device);
    if (dev_grid < 8) return PF_OK;
    const int64_t wa = ga->n_pad / ga->win_rows, wb = gb ? gb->n_pad / gb->win_rows : 0;
    const int64_t grid = (std::max(wa, wb) + 7) & ~(int64_t)7;
    if (grid > dev_grid) return PF_OK;
    if (nw == 1) {
        PF_TRY(persist_cheb2(a, b, grid, done, lds_bytes));
        if (*done) return PF_OK;
    }
    PF_TRY(pf_window_slots_prepare(ga));
    if (gb) PF_TRY(pf_window_slots_prepare(gb));
    if (ga->px_state != 1 || (gb && gb->px_state != 1)) return PF_OK;
    // (the LDS need of a graph or pair is a walk over every window's slices on the host: remembered per graph and partner)
    if (ga->lds_need_partner != (gb ? gb->uid : ga->uid) || ga->lds_need_value == -2) {
        ga->lds_need_value = lds_need(ga, gb, nw);
        ga->lds_need_partner = gb ? gb->uid : ga->uid;
    }
    const int64_t need = ga->lds_need_value;
    if (need < 0 || (size_t)need > RX_LDS_LIMIT) return PF_OK;
    OwnerGuard own(ctx);
    if (!own.ok) return PF_OK;  // another ctx has resident kernels in flight
    if (ng == 2 && nw == 1 && pair_halves_enabled()) {  // each half of a block fetches one graph's outside rows: they must fit
        int32_t most = 0;
        for (int32_t c : ga->h_px_gh_cnt) most = std::max(most, c);
        for (int32_t c : gb->h_px_gh_cnt) most = std::max(most, c);
        if (most <= RX_THREADS / 2) kernel = k_cheb_resident<2, 1, rx_jr(2, 1), true>;
    }
    hipStream_t st = ctx->stream;
    PF_TRY(pf_persist_sync_ensure(ctx));
    const uint64_t epoch = g_abort_epoch.load();
    const pf_persist_args* in[2] = {a, b};
    RxArgs args{};
    for (int q = 0; q < ng; ++q) {
        pf_graph* g = in[q]->g;
        if (!g->persist_ring) {
            PF_HIP(pf_malloc(st, (void**)&g->persist_ring, sizeof(double) * 4 * (size_t)g->n_pad));
            g->persist_epoch = 0;
        }
        if (g->persist_epoch != epoch) {  // new, or left in an unknown state by an aborted launch: all slots empty
            PF_HIP(hipMemsetD32Async((hipDeviceptr_t)g->persist_ring, (int)RX_EMPTY32, (size_t)8 * (size_t)g->n_pad, st));
            g->persist_epoch = epoch;
            g->persist_phase = 0;
        }
        RxGraph& p = args.g[q];
        p.slice_ptr = g->slice_ptr;
        p.slot = g->px_slot;
        p.gh_cnt = g->px_gh_cnt;
        p.gh_row = g->px_gh_row;
        p.need = g->px_need;
        p.sval = in[q]->vals;
        p.diag = g->diag;
        p.src = in[q]->src;
        p.dst = in[q]->dst;
        p.ring = g->persist_ring;
        p.n_pad = g->n_pad;
        p.n_windows = (int32_t)(g->n_pad / g->win_rows);
        p.degree = in[q]->degree;
        p.phase = g->persist_phase;
        p.a1 = 1.0 / (in[q]->e * in[q]->rho);
        p.a2 = 2.0 / (in[q]->e * in[q]->rho);
        p.shift = in[q]->c;
        p.beta = 1.0 / (in[q]->rho * in[q]->rho);
    }
    args.abort_flag = ctx->persist_sync;
    args.host_abort = ctx->persist_abort;
    args.clock = reinterpret_cast<unsigned long long*>(ctx->persist_sync + 8);
    const bool halves = kernel == k_cheb_resident<2, 1, rx_jr(2, 1), true>;
    args.hold = hold_ticks(ng, nw, halves, (int)std::max(wa, wb));
    args.report = nullptr;
    args.report_id = 0;
    if (args.hold > 0 && calibration_enabled()) {
        if (HoldCalibration* cal = calibration_of(ctx)) {
            const uint64_t key = ((uint64_t)ng << 40) | ((uint64_t)nw << 32) | ((uint64_t)(halves ? 1 : 0) << 31) | (uint64_t)std::max(wa, wb);
            int slot = -1;
            unsigned long long id = 0;
            args.hold = cal->pick(key, args.hold, longest, &slot, &id);
            if (slot >= 0) {
                args.report = cal->ring + 2 * slot;
                args.report_id = id;
            }
        }
    }
    g_last_hold.store(args.hold);
    if (g_test_aborts.load() > 0) {  // pf_persist_test_hook: this launch finds the abort flag raised
        g_test_aborts.fetch_sub(1);
        PF_HIP(hipMemsetAsync(ctx->persist_sync, 1, sizeof(uint32_t), st));
    }
    kernel<<<dim3((unsigned)grid), dim3(RX_THREADS), (size_t)need, st>>>(args);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        if (args.report && ctx->persist_cal) {
            auto* cal = static_cast<HoldCalibration*>(ctx->persist_cal);
            cal->release((int)((args.report - cal->ring) / 2));
        }
        g_persist.store(0);  // the classic path from now on
        (void)hipGetLastError();
        return PF_OK;
    }
    PF_TRY(persist_launched(ctx));
    g_launches.fetch_add(1);
    for (int q = 0; q < ng; ++q) {
        pf_graph* g = in[q]->g;
        g->persist_phase = (g->persist_phase + in[q]->degree) & 3;
        // LDS bytes of this launch: per step and row the result written (8; the row's own x stays in a register), per
        // stored entry the gathered x (8), per outside row its value written once (8); the entries live in registers
        if (lds_bytes) *lds_bytes += (double)in[q]->degree * (8.0 * (double)g->sell_entries + 8.0 * (double)g->n_pad + 8.0 * (double)g->px_gh_total);
    }
    *done = 1;
    return PF_OK;
}

void pf_persist_release(pf_ctx* ctx) {
    {
        std::lock_guard<std::mutex> lk(owner_of(ctx).m);
        if (owner_of(ctx).owner == ctx) owner_of(ctx).owner = nullptr;
    }
    if (ctx->persist_cal) {  // (pf_destroy has synchronised the stream: no launch is left to report)
        auto* cal = static_cast<HoldCalibration*>(ctx->persist_cal);
        if (cal->ring) (void)hipHostFree(cal->ring);
        delete cal;
        ctx->persist_cal = nullptr;
    }
}

// the abort word on the device and its pinned twin (also raised by the one-launch Gram-Schmidt step of pf_operator.hip,
// whose grid-wide wait is bounded like the resident kernels')
int pf_persist_sync_ensure(pf_ctx* ctx) {
    if (!ctx->persist_sync) {
        hipStream_t st = ctx->stream;
        PF_HIP(pf_malloc(st, (void**)&ctx->persist_sync, sizeof(uint32_t) * 32));
        PF_HIP(hipMemsetAsync(ctx->persist_sync, 0, sizeof(uint32_t) * 32, st));
        PF_HIP(hipHostMalloc((void**)&ctx->persist_abort, sizeof(int32_t), hipHostMallocDefault));
        *ctx->persist_abort = 0;
    }
    return PF_OK;
}

// Called wherever the library has just synchronised with the stream.  If a resident launch gave up: drain the stream
// (launches queued behind the failed one give up at once: the flag is still raised), lower the flags, declare every
// ring unknown, SUSPEND the path (persist_available counts the suspension down) and report it - the results of the filter applications since the
// last clean check are invalid, and repeating the solve (now one step per launch) is all the caller has to do.
int pf_persist_check(pf_ctx* ctx) {
    if (ctx->persist_abort && *ctx->persist_abort) {
        (void)hipStreamSynchronize(ctx->stream);
        *ctx->persist_abort = 0;
        (void)hipMemsetAsync(ctx->persist_sync, 0, sizeof(uint32_t) * 8, ctx->stream);  // (the abort words; the clock counters behind them stay)
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipGetLastError();
        g_abort_epoch.fetch_add(1);
        g_timeouts.fetch_add(1);
        g_suspended.store(suspend_length());
        if (ctx->persist_cal) static_cast<HoldCalibration*>(ctx->persist_cal)->forget_pending();
        static std::atomic<bool> said{false};
        if (!said.exchange(true))
            fprintf(stderr, "libpyfocusr_hip: a wait inside the resident Chebyshev kernel ran out (device shared with another tenant?); "
                            "filter applications run one step per launch for a while (pf_persist_state tells)\n");
        PF_CHECK(false, PF_E_PERSIST_TIMEOUT,
                 "resident Chebyshev kernel: a wait for a neighbouring window ran out (device shared with another tenant?); "
                 "the filter applications since the last synchronisation are invalid, the stream is drained and the next "
                 "filter applications run one step per launch (the resident path is tried again later): repeat the solve");
    }
    return PF_OK;
}
