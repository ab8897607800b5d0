// Window structures of the resident Chebyshev kernel (pf_persist.hip), built once per graph on first use.
//
// A window is a run of g->win_rows (1024, 2048 or 4096) consecutive solver-order rows - a compact patch of the surface
// (Morton order) with its boundary rows first (pf_reorder.hip) - owned by one block of the resident kernel.  Per window:
//   * the sorted list of OUTSIDE rows its rows read (px_gh_row, at most PF_WIN_GHOSTS);
//   * for every SELL entry of its rows, the window-local slot of the entry's column: the row's index inside the
//     window, or win_rows + index into that list (px_slot) - 16 bits are enough, and every gather of a step becomes an
//     LDS read;
//   * how many of its leading rows some other window reads (px_need): only those are published every step.
// The hand-off protocol of the kernel recycles a slot of a window's published rows once every reader is known to be a
// step further, and it knows that from the values it receives itself; so "A reads B" has to imply "B reads A".  On
// meshes whose W is symmetric it does.  One-way edges (graph.py:178 on open or non-manifold meshes) can break it at
// window level; then B gets row 0 of A as an extra outside row (read every step, never used).
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <climits>

#include "pf_internal.h"
#include "pf_launch.h"

namespace {

constexpr int WB_CAP = 4096;  // outside-column candidates of a window before deduplication
constexpr int WB_ADJ_WORDS = PF_WIN_MAX / 32;

// pass 1: which windows read which (bit matrix), and the extent of each window's externally read rows.  A window's
// outside entries hit a handful of neighbours: merged in LDS first (same-address atomics in memory serialise at ~10 ns)
struct k_win_scan {
    static constexpr int BOUNDS = PF_WIN_THREADS;
    static __device__ __forceinline__ void run(const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ scol,
                                                             int32_t win_rows, int32_t n_windows, uint32_t* __restrict__ adj,
                                                             int32_t* __restrict__ need) {
    __shared__ uint32_t l_adj[WB_ADJ_WORDS];
    __shared__ int32_t l_need[PF_WIN_MAX];
    const int tid = threadIdx.x;
    const int lane = tid & (PF_WAVE - 1);
    const int32_t A = (int32_t)blockIdx.x;
    const int64_t r0 = (int64_t)A * win_rows;
    for (int i = tid; i < WB_ADJ_WORDS; i += PF_WIN_THREADS) l_adj[i] = 0u;
    for (int i = tid; i < PF_WIN_MAX; i += PF_WIN_THREADS) l_need[i] = 0;
    __syncthreads();
    for (int32_t lr = tid; lr < win_rows; lr += PF_WIN_THREADS) {
        const int64_t s = (r0 + lr) >> 6;
        const int64_t base = slice_ptr[s];
        const int32_t width = (int32_t)((slice_ptr[s + 1] - base) >> 6);
        for (int32_t j = 0; j < width; ++j) {
            const int32_t c = scol[pf_sell_index(base, width, j, lane)];
            const int32_t B = c / win_rows;
            if (B != A) {
                atomicOr(&l_adj[B >> 5], 1u << (B & 31));
                atomicMax(&l_need[B], c - B * win_rows + 1);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < WB_ADJ_WORDS; i += PF_WIN_THREADS)
        if (l_adj[i]) adj[(int64_t)A * WB_ADJ_WORDS + i] = l_adj[i];  // row A of the matrix is this block's alone
    for (int32_t B = tid; B < n_windows; B += PF_WIN_THREADS)
        if (l_need[B] > 0) atomicMax(&need[B], l_need[B]);
    if (tid == 0) atomicMax(&need[A], 1);  // row 0 of every window is always published (extra outside rows point at it)
}
};

// pass 2: one block per window: sorted unique outside rows (+ row 0 of the windows that read this one without being
// read by it), then the window-local slot of every entry
struct k_win_build {
    static constexpr int BOUNDS = PF_WIN_THREADS;
    static __device__ __forceinline__ void run(const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ scol,
                                                              int32_t win_rows, int32_t n_windows, const uint32_t* __restrict__ adj,
                                                              int32_t* __restrict__ slot_out, int32_t* __restrict__ gh_cnt,
                                                              int32_t* __restrict__ gh_row, int32_t* __restrict__ flags) {
    __shared__ int32_t cand[WB_CAP];
    __shared__ int32_t ghost[PF_WIN_GHOSTS];
    __shared__ int32_t cnt, hcount;
    const int tid = threadIdx.x;
    const int lane = tid & (PF_WAVE - 1);
    const int32_t A = (int32_t)blockIdx.x;
    const int64_t r0 = (int64_t)A * win_rows;
    if (tid == 0) cnt = 0, hcount = 0;
    for (int k = tid; k < WB_CAP; k += PF_WIN_THREADS) cand[k] = INT_MAX;
    __syncthreads();
    for (int32_t lr = tid; lr < win_rows; lr += PF_WIN_THREADS) {
        const int64_t s = (r0 + lr) >> 6;
        const int64_t base = slice_ptr[s];
        const int32_t width = (int32_t)((slice_ptr[s + 1] - base) >> 6);
        for (int32_t j = 0; j < width; ++j) {
            const int32_t c = scol[pf_sell_index(base, width, j, lane)];
            if (c < r0 || c >= r0 + win_rows) {
                const int p = atomicAdd(&cnt, 1);
                if (p < WB_CAP) cand[p] = c;
            }
        }
    }
    for (int32_t B = tid; B < n_windows; B += PF_WIN_THREADS) {
        if (B == A) continue;
        const bool b_reads_a = (adj[(int64_t)B * WB_ADJ_WORDS + (A >> 5)] >> (A & 31)) & 1u;
        const bool a_reads_b = (adj[(int64_t)A * WB_ADJ_WORDS + (B >> 5)] >> (B & 31)) & 1u;
        if (b_reads_a && !a_reads_b) {
            const int p = atomicAdd(&cnt, 1);
            if (p < WB_CAP) cand[p] = B * win_rows;
        }
    }
    __syncthreads();
    if (cnt > WB_CAP) {  // block-uniform
        if (tid == 0) atomicOr(flags, 1);
        return;
    }
    // bitonic sort of cand[0, cap) ascending, cap = the power of two that holds the candidates (INT_MAX padding sorts to
    // the end; a window has ~500 of them, not WB_CAP)
    int cap = 64;
    while (cap < cnt) cap <<= 1;  // block-uniform
    for (int k = 2; k <= cap; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < cap; i += PF_WIN_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const int32_t a = cand[i], b = cand[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        cand[i] = b;
                        cand[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    // unique: an entry that differs from its predecessor starts a new outside row; positions by ballot / popcount within a
    // wave and a running total over the waves (a serial pass by one thread cost ~40 us of the kernel's 95)
    {
        __shared__ int wave_tot[PF_WIN_THREADS / PF_WAVE];
        __shared__ int running;
        if (tid == 0) running = 0;
        __syncthreads();
        for (int base = 0; base < cap; base += PF_WIN_THREADS) {  // (block-uniform trip count)
            const int i = base + tid;
            const int32_t c = i < cap ? cand[i] : INT_MAX;
            const bool fresh = c != INT_MAX && (i == 0 || cand[i - 1] != c);
            const unsigned long long m = __ballot(fresh);
            const int before = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wave_tot[tid / PF_WAVE] = __popcll(m);
            __syncthreads();
            int off = running;
            for (int w = 0; w < tid / PF_WAVE; ++w) off += wave_tot[w];
            if (fresh && off + before < PF_WIN_GHOSTS) ghost[off + before] = c;
            __syncthreads();
            if (tid == 0) {
                int t = running;
                for (int w = 0; w < PF_WIN_THREADS / PF_WAVE; ++w) t += wave_tot[w];
                running = t;
            }
            __syncthreads();
        }
        if (tid == 0) hcount = running;
    }
    __syncthreads();
    const int h = hcount;
    if (h > PF_WIN_GHOSTS) {
        if (tid == 0) atomicOr(flags, 1);
        return;
    }
    if (tid == 0) gh_cnt[A] = h;
    for (int i = tid; i < PF_WIN_GHOSTS; i += PF_WIN_THREADS) gh_row[(int64_t)A * PF_WIN_GHOSTS + i] = i < h ? ghost[i] : 0;
    for (int32_t lr = tid; lr < win_rows; lr += PF_WIN_THREADS) {
        const int64_t s = (r0 + lr) >> 6;
        const int64_t base = slice_ptr[s];
        const int32_t width = (int32_t)((slice_ptr[s + 1] - base) >> 6);
        for (int32_t j = 0; j < width; ++j) {
            const int64_t idx = pf_sell_index(base, width, j, lane);
            const int32_t c = scol[idx];
            int32_t slot;
            if (c >= r0 && c < r0 + win_rows) {
                slot = (int32_t)(c - r0);
            } else {
                int lo = 0, hi = h;  // lower bound in the sorted list (c is in it)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (ghost[mid] < c) lo = mid + 1;
                    else hi = mid;
                }
                slot = win_rows + lo;
            }
            slot_out[idx] = slot;
        }
    }
}
};


// ---- second ring (k_cheb_resident2: two recurrence steps per exchange)
//
// pass 3: one block per window A.  Ring 1 = the sorted outside rows A's own rows read (k_win_build).  Ring 2 = the rows
// those read that are neither A's nor ring 1: sorted, appended to the window's list.  For every ring-1 row: the SELL
// indices of its entries (padding included: the same operations in the same order as its owner's) and their
// window-local slots (own row | win_rows + index in ring 1 | win_rows + |ring 1| + index in ring 2).  Which of a
// window's leading rows some other window holds in either ring (need2), and who holds rows of whom (adj2).
struct k_win_rings {
    static constexpr int BOUNDS = PF_WIN_THREADS;
    static __device__ __forceinline__ void run(const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ scol,
                                                              int32_t win_rows, int32_t n_windows, const int32_t* __restrict__ gh_cnt,
                                                              int32_t* __restrict__ gh_row, int32_t* __restrict__ gh_cnt2,
                                                              int32_t* __restrict__ need2, uint32_t* __restrict__ adj2,
                                                              uint8_t* __restrict__ g1_w, int32_t* __restrict__ g1_pos,
                                                              uint16_t* __restrict__ g1_slot, int32_t* __restrict__ g1_gw,
                                                              unsigned long long* __restrict__ totals, int32_t* __restrict__ flags) {
    __shared__ int32_t ring1[PF_WIN_G1];
    __shared__ int32_t cand[WB_CAP];
    __shared__ int32_t ring2[PF_WIN_GHOSTS];
    __shared__ uint32_t l_adj[WB_ADJ_WORDS];
    __shared__ int32_t l_need[PF_WIN_MAX];
    __shared__ int32_t cnt, hcount, wmax, entries;
    const int tid = threadIdx.x;
    const int lane = tid & (PF_WAVE - 1);
    const int32_t A = (int32_t)blockIdx.x;
    const int64_t r0 = (int64_t)A * win_rows;
    const int32_t g1 = gh_cnt[A];
    if (g1 > PF_WIN_G1) {  // block-uniform
        if (tid == 0) atomicOr(flags, 1);
        return;
    }
    if (tid == 0) cnt = 0, hcount = 0, wmax = 0, entries = 0;
    for (int i = tid; i < g1; i += PF_WIN_THREADS) ring1[i] = gh_row[(int64_t)A * PF_WIN_GHOSTS + i];
    for (int k = tid; k < WB_CAP; k += PF_WIN_THREADS) cand[k] = INT_MAX;
    for (int i = tid; i < WB_ADJ_WORDS; i += PF_WIN_THREADS) l_adj[i] = 0u;
    for (int i = tid; i < PF_WIN_MAX; i += PF_WIN_THREADS) l_need[i] = 0;
    __syncthreads();
    auto in_ring1 = [&](int32_t c, int* where) {
        int lo = 0, hi = g1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ring1[mid] < c) lo = mid + 1;
            else hi = mid;
        }
        *where = lo;
        return lo < g1 && ring1[lo] == c;
    };
    for (int i = tid; i < g1; i += PF_WIN_THREADS) {
        const int32_t row = ring1[i];
        const int64_t s = row >> 6;
        const int64_t base = slice_ptr[s];
        const int32_t width = (int32_t)((slice_ptr[s + 1] - base) >> 6);
        for (int32_t j = 0; j < width; ++j) {
            const int32_t c = scol[pf_sell_index(base, width, j, row & (PF_WAVE - 1))];
            int where;
            if ((c < r0 || c >= r0 + win_rows) && !in_ring1(c, &where)) {
                const int p = atomicAdd(&cnt, 1);
                if (p < WB_CAP) cand[p] = c;
            }
        }
        atomicMax(&wmax, width);
        atomicAdd(&entries, width);
    }
    __syncthreads();
    if (cnt > WB_CAP || wmax > PF_WIN_GW) {  // block-uniform
        if (tid == 0) atomicOr(flags, 1);
        return;
    }
    int cap = 64;
    while (cap < cnt) cap <<= 1;
    for (int k = 2; k <= cap; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < cap; i += PF_WIN_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const int32_t a = cand[i], b = cand[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        cand[i] = b;
                        cand[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    {
        __shared__ int wave_tot[PF_WIN_THREADS / PF_WAVE];
        __shared__ int running;
        if (tid == 0) running = 0;
        __syncthreads();
        for (int base = 0; base < cap; base += PF_WIN_THREADS) {
            const int i = base + tid;
            const int32_t c = i < cap ? cand[i] : INT_MAX;
            const bool fresh = c != INT_MAX && (i == 0 || cand[i - 1] != c);
            const unsigned long long m = __ballot(fresh);
            const int before = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wave_tot[tid / PF_WAVE] = __popcll(m);
            __syncthreads();
            int off = running;
            for (int w = 0; w < tid / PF_WAVE; ++w) off += wave_tot[w];
            if (fresh && off + before < PF_WIN_GHOSTS) ring2[off + before] = c;
            __syncthreads();
            if (tid == 0) {
                int t = running;
                for (int w = 0; w < PF_WIN_THREADS / PF_WAVE; ++w) t += wave_tot[w];
                running = t;
            }
            __syncthreads();
        }
        if (tid == 0) hcount = running;
    }
    __syncthreads();
    const int h = hcount;
    if (g1 + h > PF_WIN_GHOSTS) {
        if (tid == 0) atomicOr(flags, 1);
        return;
    }
    if (tid == 0) {
        gh_cnt2[A] = h;
        g1_gw[A] = wmax;
        atomicAdd(totals, (unsigned long long)h);
        atomicAdd(totals + 1, (unsigned long long)entries);
    }
    for (int i = tid; i < h; i += PF_WIN_THREADS) gh_row[(int64_t)A * PF_WIN_GHOSTS + g1 + i] = ring2[i];
    for (int i = tid; i < g1; i += PF_WIN_THREADS) {
        const int32_t row = ring1[i];
        const int64_t s = row >> 6;
        const int64_t base = slice_ptr[s];
        const int32_t width = (int32_t)((slice_ptr[s + 1] - base) >> 6);
        g1_w[(int64_t)A * PF_WIN_G1 + i] = (uint8_t)width;
        for (int32_t j = 0; j < width; ++j) {
            const int64_t idx = pf_sell_index(base, width, j, row & (PF_WAVE - 1));
            const int32_t c = scol[idx];
            int32_t slot;
            int where;
            if (c >= r0 && c < r0 + win_rows) {
                slot = (int32_t)(c - r0);
            } else if (in_ring1(c, &where)) {
                slot = win_rows + where;
            } else {
                int lo = 0, hi = h;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (ring2[mid] < c) lo = mid + 1;
                    else hi = mid;
                }
                slot = win_rows + g1 + lo;
            }
            const int64_t o = ((int64_t)A * PF_WIN_GW + j) * PF_WIN_G1 + i;
            g1_pos[o] = (int32_t)idx;
            g1_slot[o] = (uint16_t)slot;
        }
    }
    for (int i = tid; i < g1 + h; i += PF_WIN_THREADS) {
        const int32_t c = i < g1 ? ring1[i] : ring2[i - g1];
        const int32_t B = c / win_rows;
        atomicOr(&l_adj[B >> 5], 1u << (B & 31));
        atomicMax(&l_need[B], c - B * win_rows + 1);
    }
    __syncthreads();
    for (int i = tid; i < WB_ADJ_WORDS; i += PF_WIN_THREADS)
        if (l_adj[i]) adj2[(int64_t)A * WB_ADJ_WORDS + i] = l_adj[i];
    for (int32_t B = tid; B < n_windows; B += PF_WIN_THREADS)
        if (l_need[B] > 0) atomicMax(&need2[B], l_need[B]);
    if (tid == 0) atomicMax(&need2[A], 1);
}
};

// the hand-off protocol of the resident kernels needs "A holds rows of B" <=> "B holds rows of A" (see the header);
// with a symmetric W it follows for both rings.  Anything else: this graph keeps one step per exchange.
struct k_win_rings_check {
    static constexpr int BOUNDS = PF_WIN_THREADS;
    static __device__ __forceinline__ void run(const uint32_t* __restrict__ adj2, int32_t n_windows,
                                                                    int32_t* __restrict__ flags) {
    for (int64_t p = (int64_t)blockIdx.x * PF_WIN_THREADS + threadIdx.x; p < (int64_t)n_windows * n_windows;
         p += (int64_t)gridDim.x * PF_WIN_THREADS) {
        const int32_t A = (int32_t)(p / n_windows), B = (int32_t)(p % n_windows);
        const bool ab = (adj2[(int64_t)A * WB_ADJ_WORDS + (B >> 5)] >> (B & 31)) & 1u;
        const bool ba = (adj2[(int64_t)B * WB_ADJ_WORDS + (A >> 5)] >> (A & 31)) & 1u;
        if (ab != ba) atomicOr(flags, 2);
    }
}
};

}  // namespace

// g->px_state: -1 not tried, 0 this graph is not covered (the callers have other paths), 1 ready
// The window structures in two halves: _begin queues the two kernels and the read-back of the per-window counts (pinned
// block of the graph's own, event behind it) - the assembler calls it behind its SELL fill, so that the first filter
// application of a solve finds them ready instead of building them with two synchronisations at its head (round 3:
// ~70 us per graph) - and _prepare collects.  px_state: -1 not tried, -2 in flight, 0 not covered, 1 ready.
int pf_window_slots_begin(pf_graph* g) {
    if (g->px_state != -1) return PF_OK;
    g->px_state = 0;
    if (g->win_rows <= 0 || g->n_pad % g->win_rows != 0 || g->sell_entries <= 0) return PF_OK;
    const int64_t nw = g->n_pad / g->win_rows;
    if (nw > 256) return PF_OK;  // one window per block, one block per CU
    pf_ctx* c = g->ctx;
    hipStream_t st = c->stream;
    const int64_t tail = (nw + 1) & ~(int64_t)1;  // (the builder's flag: 8-byte aligned behind the counts)
    const size_t bytes = sizeof(int32_t) * (size_t)(tail + 2);
    const int32_t need_doubles = (int32_t)((bytes + 7) / 8);
    if (!g->px_host) {  // a pinned block: from the pool a freed graph left behind, or new
        const int32_t cap = std::max(need_doubles, 64);
        for (size_t i = 0; i < c->pinned_pool.size(); ++i)
            if (c->pinned_pool[i].first >= cap) {
                g->px_host_cap = c->pinned_pool[i].first;
                g->px_host = c->pinned_pool[i].second;
                c->pinned_pool.erase(c->pinned_pool.begin() + (long)i);
                break;
            }
        if (!g->px_host) {
            PF_HIP(hipHostMalloc((void**)&g->px_host, sizeof(double) * (size_t)(cap + 2), hipHostMallocDefault));
            g->px_host_cap = cap;
        }
    }
    if (!g->px_ev) {
        if (!c->event_pool.empty()) {
            g->px_ev = c->event_pool.back();
            c->event_pool.pop_back();
        } else {
            PF_HIP(hipEventCreateWithFlags(&g->px_ev, hipEventDisableTiming));
        }
    }
    int32_t* flags = nullptr;
    uint32_t* adj = nullptr;
    hipError_t e = pf_malloc(st, (void**)&flags, sizeof(int32_t));
    if (e == hipSuccess) e = pf_malloc(st, (void**)&adj, sizeof(uint32_t) * (size_t)nw * WB_ADJ_WORDS);
    if (e == hipSuccess) e = pf_malloc(st, (void**)&g->px_slot, sizeof(int32_t) * (size_t)g->sell_entries);
    if (e == hipSuccess) e = pf_malloc(st, (void**)&g->px_gh_cnt, sizeof(int32_t) * (size_t)(nw + 4));  // [nw] counts, then the builder's flag
    if (e == hipSuccess) e = pf_malloc(st, (void**)&g->px_need, sizeof(int32_t) * nw);
    if (e == hipSuccess) e = pf_malloc(st, (void**)&g->px_gh_row, sizeof(int32_t) * nw * PF_WIN_GHOSTS);
    if (e == hipSuccess) e = pfl::memset_words(st, flags, 0, sizeof(int32_t));
    if (e == hipSuccess) e = pfl::memset_words(st, adj, 0, sizeof(uint32_t) * (size_t)nw * WB_ADJ_WORDS);
    if (e == hipSuccess) e = pfl::memset_words(st, g->px_gh_cnt, 0, sizeof(int32_t) * (size_t)(nw + 4));
    if (e == hipSuccess) e = pfl::memset_words(st, g->px_need, 0, sizeof(int32_t) * nw);
    if (e == hipSuccess) {
        pfl::launch<k_win_scan>(dim3((unsigned)nw), dim3(PF_WIN_THREADS), 0, st, g->slice_ptr, g->scol, g->win_rows, (int32_t)nw, adj, g->px_need);
        pfl::launch<k_win_build>(dim3((unsigned)nw), dim3(PF_WIN_THREADS), 0, st, g->slice_ptr, g->scol, g->win_rows, (int32_t)nw, adj, g->px_slot,
                                                             g->px_gh_cnt, g->px_gh_row, flags);
        e = hipGetLastError();
    }
    // the counts and the flag in ONE read-back through pinned memory and a copy kernel
    if (e == hipSuccess) e = pfl::memcpy_async(st, g->px_gh_cnt + tail, flags, sizeof(int32_t), hipMemcpyDeviceToDevice);
    if (e == hipSuccess) {
        const int32_t* src = g->px_gh_cnt;
        double* dst = g->px_host;
        pfl::call(st, [=](hipStream_t s) { (void)pf_copy_by_kernel(s, src, dst, bytes); });
    }
    if (e == hipSuccess) e = pfl::event_record(st, g->px_ev);
    pf_free(st, flags);
    pf_free(st, adj);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        pf_window_slots_free(g);
        return PF_OK;
    }
    g->px_state = -2;
    return PF_OK;
}

int pf_window_slots_prepare(pf_graph* g) {
    if (g->px_state == -1) PF_TRY(pf_window_slots_begin(g));
    if (g->px_state != -2) return PF_OK;
    g->px_state = 0;
    const int64_t nw = g->n_pad / g->win_rows;
    const int64_t tail = (nw + 1) & ~(int64_t)1;
    if (hipEventSynchronize(g->px_ev) != hipSuccess) {
        (void)hipGetLastError();
        pf_window_slots_free(g);
        return PF_OK;
    }
    const int32_t* pin = reinterpret_cast<const int32_t*>(g->px_host);
    g->h_px_gh_cnt.assign(pin, pin + nw);
    if (pin[tail]) {  // the builder's flag: a window reads more outside rows than the structures hold
        pf_window_slots_free(g);
        return PF_OK;
    }
    g->px_gh_total = 0;
    for (int32_t c : g->h_px_gh_cnt) g->px_gh_total += c;
    if (getenv("PF_DEBUG_WINDOWS")) {
        int32_t lo = 1 << 30, hi = 0;
        for (int32_t c : g->h_px_gh_cnt) { lo = c < lo ? c : lo; hi = c > hi ? c : hi; }
        fprintf(stderr, "pyfocusr_hip: %lld windows, outside rows per window %d..%d (mean %.0f)\n", (long long)nw, (int)lo, (int)hi,
                (double)g->px_gh_total / (double)nw);
    }
    g->px_state = 1;
    return PF_OK;
}

static void window_rings_free(pf_graph* g) {
    hipStream_t st = g->ctx->stream;
    pf_free(st, g->px_gh_cnt2);
    pf_free(st, g->px_need2);
    pf_free(st, g->px_g1_w);
    pf_free(st, g->px_g1_pos);
    pf_free(st, g->px_g1_slot);
    pf_free(st, g->px_g1_gw);
    g->px_gh_cnt2 = g->px_need2 = g->px_g1_pos = g->px_g1_gw = nullptr;
    g->px_g1_w = nullptr;
    g->px_g1_slot = nullptr;
    g->h_px_gh_cnt2.clear();
    g->h_px_g1_gw.clear();
}

// g->px2_state: -1 not tried, 0 not covered (one step per exchange stays available), 1 ready
int pf_window_rings_prepare(pf_graph* g) {
    if (g->px2_state >= 0) return PF_OK;
    g->px2_state = 0;
    PF_TRY(pf_window_slots_prepare(g));
    if (g->px_state != 1 || g->win_rows != PF_WIN_THREADS) return PF_OK;
    const int64_t nw = g->n_pad / g->win_rows;
    hipStream_t st = g->ctx->stream;
    int32_t* flags = nullptr;
    uint32_t* adj2 = nullptr;
    unsigned long long* totals = nullptr;
    hipError_t e = pf_malloc(st, (void**)&flags, sizeof(int32_t));
    if (e == hipSuccess) e = pf_malloc(st, (void**)&totals, 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = pf_malloc(st, (void**)&adj2, sizeof(uint32_t) * (size_t)nw * WB_ADJ_WORDS);
    if (e == hipSuccess) e = pf_malloc(st, (void**)&g->px_gh_cnt2, sizeof(int32_t) * nw);
    if (e == hipSuccess) e = pf_malloc(st, (void**)&g->px_need2, sizeof(int32_t) * nw);
    if (e == hipSuccess) e = pf_malloc(st, (void**)&g->px_g1_gw, sizeof(int32_t) * nw);
    if (e == hipSuccess) e = pf_malloc(st, (void**)&g->px_g1_w, (size_t)nw * PF_WIN_G1);
    if (e == hipSuccess) e = pf_malloc(st, (void**)&g->px_g1_pos, sizeof(int32_t) * (size_t)nw * PF_WIN_GW * PF_WIN_G1);
    if (e == hipSuccess) e = pf_malloc(st, (void**)&g->px_g1_slot, sizeof(uint16_t) * (size_t)nw * PF_WIN_GW * PF_WIN_G1);
    if (e == hipSuccess) e = pfl::memset_words(st, flags, 0, sizeof(int32_t));
    if (e == hipSuccess) e = pfl::memset_words(st, totals, 0, 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = pfl::memset_words(st, adj2, 0, sizeof(uint32_t) * (size_t)nw * WB_ADJ_WORDS);
    if (e == hipSuccess) e = pfl::memset_words(st, g->px_gh_cnt2, 0, sizeof(int32_t) * nw);
    if (e == hipSuccess) e = pfl::memset_words(st, g->px_need2, 0, sizeof(int32_t) * nw);
    if (e == hipSuccess) e = pfl::memset_words(st, g->px_g1_gw, 0, sizeof(int32_t) * nw);
    // one pinned block for everything the host wants back: flag, totals, ring-2 counts, widest ring-1 rows
    const size_t back = sizeof(int32_t) * (2 + 2 * (size_t)nw) + 2 * sizeof(unsigned long long);
    void* pin = nullptr;
    if (e == hipSuccess && pf_pinned_scratch(g->ctx, back, &pin) != PF_OK) e = hipErrorOutOfMemory;
    if (e == hipSuccess) {
        pfl::launch<k_win_rings>(dim3((unsigned)nw), dim3(PF_WIN_THREADS), 0, st, g->slice_ptr, g->scol, g->win_rows, (int32_t)nw, g->px_gh_cnt, g->px_gh_row,
                                                             g->px_gh_cnt2, g->px_need2, adj2, g->px_g1_w, g->px_g1_pos, g->px_g1_slot,
                                                             g->px_g1_gw, totals, flags);
        pfl::launch<k_win_rings_check>(dim3((unsigned)std::min<int64_t>((nw * nw + PF_WIN_THREADS - 1) / PF_WIN_THREADS, 64)), dim3(PF_WIN_THREADS), 0, st, adj2, (int32_t)nw, flags);
        e = hipGetLastError();
    }
    unsigned long long* p_tot = reinterpret_cast<unsigned long long*>(pin);
    int32_t* p_i = reinterpret_cast<int32_t*>(p_tot + 2);
    if (e == hipSuccess) e = pfl::memcpy_async(st, p_tot, totals, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = pfl::memcpy_async(st, p_i, flags, sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = pfl::memcpy_async(st, p_i + 2, g->px_gh_cnt2, sizeof(int32_t) * nw, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = pfl::memcpy_async(st, p_i + 2 + nw, g->px_g1_gw, sizeof(int32_t) * nw, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = pfl::sync(st);
    pf_free(st, flags);
    pf_free(st, adj2);
    pf_free(st, totals);
    if (e != hipSuccess || p_i[0] != 0) {
        (void)hipGetLastError();
        window_rings_free(g);
        return PF_OK;
    }
    g->h_px_gh_cnt2.assign(p_i + 2, p_i + 2 + nw);
    g->h_px_g1_gw.assign(p_i + 2 + nw, p_i + 2 + 2 * nw);
    g->px_gh2_total = (int64_t)p_tot[0];
    g->px_g1_entries = (int64_t)p_tot[1];
    g->px2_state = 1;
    return PF_OK;
}

void pf_window_slots_free(pf_graph* g) {
    window_rings_free(g);
    hipStream_t st = g->ctx->stream;
    pf_free(st, g->px_slot);
    pf_free(st, g->px_gh_cnt);
    pf_free(st, g->px_gh_row);
    pf_free(st, g->px_need);
    g->px_slot = g->px_gh_cnt = g->px_gh_row = g->px_need = nullptr;
    g->h_px_gh_cnt.clear();
}
