// Internal declarations shared by the HIP translation units of libpyfocusr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/pyfocusr_hip.h"

void pf_set_error(const char* fmt, ...);

#define PF_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            pf_set_error("%s:%d %s: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_));   \
            (void)hipGetLastError(); /* clear the sticky error: later, unrelated calls must not report it again */ \
            return PF_E_HIP;                                                                  \
        }                                                                                     \
    } while (0)

#define PF_CHECK(cond, code, ...)        \
    do {                                 \
        if (!(cond)) {                   \
            pf_set_error(__VA_ARGS__);   \
            return (code);               \
        }                                \
    } while (0)

#define PF_TRY(call)            \
    do {                        \
        int r_ = (call);        \
        if (r_ != PF_OK) return r_; \
    } while (0)

constexpr int PF_WAVE = 64;        // gfx950 wavefront
constexpr int PF_BLOCK = 256;      // 4 waves: one per SIMD of a CU
constexpr int PF_DOT_CHUNK = 4096; // rows per block in the reduction kernels
constexpr int PF_MAX_ROOTS = 4096; // components tracked explicitly
constexpr int PF_WIN_THREADS = 1024; // threads of a window block of the resident Chebyshev kernel (pf_persist.hip)
constexpr int PF_WIN_GHOSTS = 1024;  // outside rows a window may read (one per thread)
constexpr int PF_WIN_MAX = 1024;     // windows per graph the window structures cover
constexpr int PF_WIN_G1 = 512;       // ring-1 rows of a window whose recurrence the window repeats itself (k_cheb_resident2)
constexpr int PF_WIN_GW = 16;        // entries per such row
constexpr int PF_STAGE_SLOTS = 8;   // pinned staging slots per ctx ...
constexpr size_t PF_STAGE_BYTES = 32768;  // ... of this size each
constexpr int PF_WS_TMPS = 4;      // temporaries behind the workspace slots (Chebyshev rotation)

// device buffers of the box hierarchy of pf_knn_tree.hip (1-NN for deep coordinates); they grow and stay with the ctx
struct pf_knn_tree {
    double* pts = nullptr;       // [n_leaf][d][64] leaves, coordinate-major
    int32_t* orig = nullptr;     // [n_leaf][64] original reference indices
    double* leaf_lo = nullptr;   // [2][n_sup][d][64] leaf boxes: lo, then hi
    double* sup_lo = nullptr;    // [2][d][ns_pad] boxes of the supers
    int32_t* qry_order = nullptr;  // queries along the references' Morton curve
    void* grid = nullptr;        // TreeGrid
    unsigned long long* counters = nullptr;
    int64_t cap_pts = 0, cap_orig = 0, cap_leaf = 0, cap_sup = 0, cap_qry = 0;
    bool count_visits = false;
};

struct pf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    void* stage_ring = nullptr;         // pinned staging slots for small host-to-device payloads (pf_combine's coefficients)
    hipEvent_t stage_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int stage_next = 0;
    hipStream_t copy_stream = nullptr;  // downloads that overlap with work on `stream` (pf_finalize_vectors_begin)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing = false;
    int timing_stride = 1;     // time every timing_stride-th filter application (an event record costs ~5 us of device time)
    int64_t timing_count = 0;
    uint32_t* persist_sync = nullptr;  // device word(s) of the resident Chebyshev kernel: its abort flag (pf_persist.hip)
    int32_t* persist_abort = nullptr;  // pinned host word: a barrier wait ran out
    void* persist_cal = nullptr;  // the hold-back calibration of this ctx's resident launches (pf_persist.hip: HoldCalibration)
    double op_ms = 0.0;
    int64_t op_launches = 0;
    double op_bytes = 0.0;
    double persist_ms = 0.0, persist_bytes = 0.0, persist_lds_bytes = 0.0;
    int64_t persist_launches = 0, persist_steps = 0;
    double knn_ms = 0.0;
    double build_ms = 0.0;
    bool build_pending = false;  // ev0 / ev1 bracket a build whose device time has not been read yet
    // nearest-neighbour state (pf_knn_upload / run / download)
    double* knn_ref = nullptr;   // [n_ref][d] as uploaded
    double* knn_qry = nullptr;   // [n_qry][d]
    double* knn_ref_s = nullptr; // rows sorted along the search axis
    double* knn_ref_soa = nullptr;  // the same rows coordinate-major, [d][knn_ref_ld] (k_knn_coop)
    int64_t knn_cap_ref_soa = 0, knn_ref_ld = 0;
    double* knn_qry_s = nullptr;
    unsigned* knn_ref_key = nullptr; // sorted grid-cell ids (references: row-major; queries: Morton)
    unsigned* knn_qry_key = nullptr;
    int32_t* knn_cell_start = nullptr; // [res*res + 1] first sorted reference of each cell
    int64_t knn_cap_cell = 0;
    int knn_res = 0;
    void* knn_grid = nullptr; // KnnGrid (pf_knn.hip)
    int32_t* knn_ref_orig = nullptr; // sorted position -> original index
    int32_t* knn_qry_orig = nullptr;
    unsigned long long* knn_ext = nullptr; // [16] per-axis min / max (encoded)
    int64_t knn_nref = 0, knn_nqry = 0;
    int64_t knn_cap_ref = 0, knn_cap_qry = 0, knn_cap_ref_s = 0, knn_cap_qry_s = 0, knn_cap_ref_key = 0,
            knn_cap_qry_key = 0, knn_cap_ref_orig = 0, knn_cap_qry_orig = 0, knn_cap_idx = 0, knn_cap_d2 = 0;
    int32_t knn_d = 0;
    int32_t knn_k = 1, knn_k_next = 1; // neighbours per query (pf_knn sets knn_k_next before the upload)
    int64_t* knn_idx = nullptr; // [n_qry]
    double* knn_d2 = nullptr;   // [n_qry]
    bool knn_ready = false, knn_done = false;
    pf_knn_tree knn_tree;
    bool knn_count_on = false;              // pf_knn_count: the counting instantiation of k_knn_coop
    unsigned long long* knn_visited = nullptr;
    size_t knn_visited_words = 0;
    int64_t knn_visited_waves = 0;
    int32_t knn_mode = 0;  // 0: by depth (box hierarchy for k = 1, d >= PF_KNN_TREE_MIN_D = 7), 1: always the grid, 2: always the hierarchy
    // operator timing: event pairs recorded around filter applications, resolved lazily in pf_timing_get so
    // that timing never blocks the host (the solver queues the next application while this one runs)
    struct TimedSpan {
        hipEvent_t e0, e1;
        int64_t launches;
        double bytes;
        int64_t persist_steps;  // > 0: resident launch(es) that ran this many steps
        double lds_bytes;       // LDS bytes those steps moved (pf_persist.hip)
    };
    std::vector<TimedSpan> spans_pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans_free;
    // pinned host buffers and events of freed graphs, handed to the next graph (hipHostMalloc / hipHostFree and event
    // creation cost 0.1-0.2 ms each: more than a 250k-vertex assembly kernel)
    std::vector<std::pair<int32_t, double*>> pinned_pool;  // (capacity in doubles - 1, buffer)
    void* pinned_scratch = nullptr;  // small read-backs land here (pinned: one DMA instead of a staged copy each)
    size_t pinned_scratch_bytes = 0;
    std::vector<hipEvent_t> event_pool;                    // created with hipEventDisableTiming
    // allocator state.  Two streams may allocate: `stream`, and `stream_b` while a pair of meshes is assembled side by
    // side (pf_graph_build_device2).  A cached block remembers the stream it was released on and the allocator epoch of
    // that moment; the other stream may take it only once it has waited for the releasing stream after that release
    // (pf_streams_join bumps the epoch and records what the waiter may now see).
    struct FreeBlock {
        void* p;
        int sid;          // 0: stream, 1: stream_b
        uint64_t epoch;   // alloc_epoch when it was released
    };
    std::multimap<size_t, FreeBlock> free_blocks;   // size -> block
    std::unordered_map<void*, size_t> live_blocks;
    hipStream_t stream_b = nullptr;
    std::mutex alloc_mutex;  // pf_malloc / pf_free: the second stream's job runs its first half on a thread of its own
    // that thread: created once per ctx (a new thread's first HIP call costs ~0.3 ms of per-thread runtime set-up)
    std::thread worker;
    std::mutex worker_mutex;
    std::condition_variable worker_cv;
    std::function<void()> worker_task;
    bool worker_busy = false, worker_stop = false;
    hipEvent_t join_ev = nullptr, fork_ev = nullptr;
    std::vector<pf_graph*> deferred;  // graphs with a download that is not queued yet ...
    std::mutex deferred_mutex;        // ... guarded: pf_host_free / pf_host_detach walk every ctx's list from whatever thread collects an array
    int64_t alloc_misses = 0;  // allocations the cache could not serve (hipMalloc: 0.1-1 ms each)
    uint64_t alloc_epoch = 1;
    uint64_t visible[2] = {0, 0};  // stream sid may take the OTHER stream's blocks released before this epoch
    void* pinned_scratch_b = nullptr;  // pf_pinned_scratch of stream_b's job
    size_t pinned_scratch_b_bytes = 0;
    void* pinned_scratch_knn = nullptr;  // ... and of the KNN's result read-back (which runs inside other users of the first)
    size_t pinned_scratch_knn_bytes = 0;
};

inline uint64_t pf_next_uid() {
    static std::atomic<uint64_t> next{1};
    return next.fetch_add(1);
}

struct pf_graph {
    hipStream_t build_stream = nullptr;  // while the graph is being assembled: the stream its kernels and blocks belong to
    hipStream_t side_stream = nullptr;   // ... and, between a fork and a join of the build, the stream of its independent chain
    uint64_t uid = pf_next_uid();  // never reused (unlike an address): remembered facts about a PAIR of graphs are keyed by it
    pf_ctx* ctx = nullptr;
    int64_t n = 0, n_pad = 0, n_faces = 0;
    int32_t vpf = 0;
    // CSR(W), sorted columns, unique directed edges
    int32_t* rowptr = nullptr; // [n+1]
    int32_t* col = nullptr;    // [nnz_w]
    double* w = nullptr;       // [nnz_w]
    int64_t nnz_w = 0;
    double* deg = nullptr;  // [n_pad]
    double* g = nullptr;    // 1/(deg+1e-8)
    double* sg = nullptr;   // sqrt(g)
    int32_t* label = nullptr; // component root per vertex
    // solver-internal renumbering (pf_reorder.hip): operator storage and workspace vectors live in
    // "new" order; everything that crosses the C-ABI is in the mesh's own ("old") order.
    // Three vertex numberings.  ORIGINAL: the caller's (everything that crosses the C-ABI).  M-SPACE (round 4; mesh graphs
    // only): the Morton rank of the vertex position - the assembler renumbers points and faces FIRST and builds CSR(W), deg,
    // labels ... in that space, so that every per-row gather of the build hits lines its neighbours share (the synthetic
    // meshes shuffle their vertices on purpose).  SOLVER: m-space rows permuted inside windows (boundary rows first, then
    // by degree).  morder == nullptr (a graph handed in as a matrix): m-space is the original numbering.
    int32_t* morder = nullptr; // [n]     m -> original
    int32_t* mrank = nullptr;  // [n]     original -> m
    int32_t* perm_m = nullptr; // [n_pad] solver row -> m (-1 on padding rows); == perm when morder == nullptr
    int32_t* iperm_m = nullptr; // [n]    m -> solver row;                        == iperm when morder == nullptr
    int32_t* perm = nullptr;  // [n_pad] solver row -> original vertex, -1 on padding rows
    int32_t* iperm = nullptr; // [n]     original vertex -> solver row
    unsigned long long* order_bbox = nullptr;  // (during a build) the bounding box of the points, encoded (pf_reorder.hip)
    double* smooth = nullptr; // [n_pad] solver order: low-order polynomial of the vertex position (Krylov start vector)
    double* stage = nullptr;  // [stage_cap] staging for permuted uploads/downloads
    int64_t stage_cap = 0;
    // SELL-64 operator storage (off-diagonals) + dense diagonal
    int64_t n_slices = 0, sell_entries = 0;
    int64_t* slice_ptr = nullptr; // [n_slices+1]
    int32_t* scol = nullptr;      // [sell_entries]
    double* sval_rw = nullptr;    // -g_i W_ij
    double* sval_sym = nullptr;   // -W_ij sqrt(g_i g_j)   (only when symmetric)
    double* diag = nullptr;       // deg_i g_i  (both operators)
    // mean-filter operator (pf_mean_filter, built on first use): rows of (D+I)^-1 (W+I) in solver order, SELL-64
    // with slice s at slice_ptr[s] + 64 s and width + 1 entries per row, in DESCENDING mesh column order
    int32_t* mf_col = nullptr;    // [sell_entries + n_pad]
    double* mf_val = nullptr;
    // windows of the resident Chebyshev kernel (pf_windows.hip / pf_persist.hip): win_rows (1024, 2048 or 4096, from
    // n_pad) consecutive solver-order rows, boundary rows first (pf_reorder.hip).  Built on first use (px_state).
    int32_t win_rows = 1024;
    std::vector<int64_t> h_slice_ptr;  // host copy of slice_ptr (LDS sizing)
    int32_t px_state = -1;         // -1 not tried, -2 in flight (pf_window_slots_begin), 0 this graph is not covered, 1 ready
    double* px_host = nullptr;     // pinned: the per-window counts + the builder's flag on their way to the host
    int32_t px_host_cap = 0;
    hipEvent_t px_ev = nullptr;
    int32_t* px_slot = nullptr;    // [sell_entries] window-local slot of every SELL column: own row, or win_rows + index
                                   // into the window's sorted list of outside rows
    int32_t* px_gh_cnt = nullptr;  // [windows]
    int32_t* px_gh_row = nullptr;  // [windows][PF_WIN_GHOSTS]
    int32_t* px_need = nullptr;    // [windows] leading rows of the window that other windows read (>= 1)
    std::vector<int32_t> h_px_gh_cnt;
    int64_t px_gh_total = 0;       // sum of h_px_gh_cnt
    double* persist_ring = nullptr;  // [4][n_pad] hand-off buffers of the windows' boundary rows (sentinel when empty)
    int32_t persist_phase = 0;       // ring slot of step k of the next launch = (k + phase) & 3
    uint64_t persist_epoch = 0;      // ring known good for this value of the library's abort epoch
    // two recurrence steps per exchange (k_cheb_resident2; windows of 1024 rows, symmetric neighbourhoods): a window
    // also keeps the rows its ring-1 rows read (ring 2: px_gh_row continues with them behind ring 1) and the ring-1
    // rows' own entries, computes the odd steps of ring 1 itself and exchanges every second step only
    int32_t px2_state = -1;          // -1 not tried, 0 not covered, 1 ready
    int32_t* px_gh_cnt2 = nullptr;   // [windows] rows of ring 2
    int32_t* px_need2 = nullptr;     // [windows] leading rows some other window holds in ring 1 or 2
    uint8_t* px_g1_w = nullptr;      // [windows][PF_WIN_G1] slice width of each ring-1 row
    int32_t* px_g1_pos = nullptr;    // [windows][PF_WIN_GW][PF_WIN_G1] SELL index of its entries
    uint16_t* px_g1_slot = nullptr;  // same shape: window-local slot of the entry's column (own | ring 1 | ring 2)
    int32_t* px_g1_gw = nullptr;     // [windows] widest ring-1 row
    std::vector<int32_t> h_px_gh_cnt2, h_px_g1_gw;
    int64_t px_gh2_total = 0, px_g1_entries = 0;  // sums over the windows: ring-2 rows, entries of ring-1 rows
    int32_t single_applications = 0;   // single-graph resident applications so far (the rings are built at the third)
    uint64_t lds_need_partner = 0;   // uid of the partner (own uid: alone) the remembered LDS need belongs to
    uint64_t lds_need2_partner = 0;
    int64_t lds_need_value = -2, lds_need2_value = -2;
    double* persist_ring2 = nullptr; // [4][n_pad] hand-off buffers of k_cheb_resident2 (slot of ROUND r = (r + phase2) & 3)
    int32_t persist_phase2 = 0;
    uint64_t persist_epoch2 = 0;
    int32_t is_symmetric = 0, n_isolated = 0, n_components = 0, max_degree = 0, n_oneway = 0;
    int32_t unit_g = 0;  // graph handed in as a matrix (pf_graph_from_matrix): G = I, the operator is the matrix itself
    std::vector<int32_t> roots; // roots of components with >= 2 vertices, ascending
    // workspace: n_slots vectors + 2 Chebyshev temporaries, stride n_pad
    double* ws = nullptr;
    int32_t n_slots = 0;
    // reduction scratch
    double* partials = nullptr; // [max_count][n_chunks] (+ stats)
    int32_t partial_cap = 0;    // in vectors
    double* coef = nullptr;     // device coefficients: [3][coef_cap]
    int32_t coef_cap = 0;
    int64_t n_chunks = 0;
    // split-phase orthogonalisation (pf_orth_begin / pf_orth_end): results land in pinned host memory
    double* orth_host = nullptr; // [orth_host_cap + 2] pinned: h, |w'|^2, verdict
    int32_t orth_host_cap = 0;
    int32_t orth_pending = -1;   // count of the orth in flight, -1 if none
    hipEvent_t orth_ev = nullptr;
    // pf_gram_begin / pf_resnorms_begin -> pf_small_end: a few doubles on their way to the host
    double* small_host = nullptr;  // [small_host_cap + 2] pinned
    int32_t small_host_cap = 0, small_pending = 0;
    bool small_root = false;       // pf_small_end returns square roots (residual norms)
    hipEvent_t small_ev = nullptr;
    hipEvent_t orth_wait = nullptr;  // the event pf_orth_end waits on: orth_ev, or the partner graph's after pf_orth_begin2
    int32_t orth_w = 0, orth_first = 0, orth_normalize = 0;  // arguments of the orth in flight (pf_orth_end's second pass)
    int32_t orth_redone = 0;     // the last pf_orth_end ran the second Gram-Schmidt pass itself
    double orth_serial = 0.0;    // tickets handed to the fused Gram-Schmidt kernels so far
    unsigned long long* orth_counter = nullptr;  // arrivals at the grid-wide wait of k_orth_local, summed over the graph's life
    unsigned long long orth_arrivals = 0;        // ... as the host counts them
    uint64_t orth_epoch = 0;                     // pf_persist_abort_epoch() the counter was last zeroed under
    int32_t orth_split = -1, orth_first2 = 0;          // pf_orth_split: for the next step ...
    int32_t orth_split_now = -1, orth_first2_now = 0;  // ... and the step in flight
    double orth_ticket = 0.0;    // ticket of the step in flight (0: that step reports through orth_ev instead)
    double orth_thresh = 0.09;   // second pass when |w'|^2 < orth_thresh |w|^2 (0.5: strict, pf_orth_strict)
    int32_t orth_device_passes = 0;  // 1: the second Gram-Schmidt pass is queued with the first and runs on the device's own verdict
    int32_t orth_twice = 0;          // the last collected step took both passes on the device
    // the last pf_finalize_vectors result stays in HBM (mesh order, [n][final_count] row-major) for pf_final_rows and
    // pf_knn1_graphs: the spectral coordinates never have to come back from the host
    double* final_vecs = nullptr;
    int32_t final_count = 0;
    // its download in flight (pf_finalize_vectors_begin / _end): copies on the ctx's copy stream between two events
    int32_t final_pending = 0;         // > 0: column count of the result whose download pf_finalize_vectors_end has to collect; < 0: a remapped image
    double* final_params = nullptr;    // device: per-column parameters + a private copy of the statistics
    void* final_stats = nullptr;       // pinned: the statistics as they arrive
    int32_t final_stats_cap = 0;
    hipEvent_t final_ready = nullptr, final_done = nullptr;
    // the host image that is still owed: a download that has not been queued yet (released behind the next long kernel,
    // pf_downloads_release; or by whoever collects it first)
    const double* dl_src = nullptr;
    double* dl_dst = nullptr;
    size_t dl_bytes = 0;
    int32_t final_check = 0;      // columns whose statistics pf_finalize_vectors_end still has to look at
    double* final_tmp = nullptr;  // device: the remapped image on its way to the host
    double* pts = nullptr;  // [n][3] the mesh's points (graphs built from a mesh): pf_point_rows
    bool deg_block = false; // g and sg live in deg's allocation (mesh path: one memset for the three)
    double spectral_bound = 2.0; // proven upper bound of the operator's spectrum (2: Gershgorin; less for closed triangle meshes)
};

// Caching device allocator, one cache per ctx (pf_api.hip).  Every use of a block is enqueued on
// the ctx's single stream, so a block released at enqueue time can be handed out again at once:
// later work is ordered behind earlier work by the stream.  After the first build the assembler's
// ~30 temporaries are recycled without a driver call (hipMalloc/hipFree cost 0.1-1 ms each and
// hipFree synchronises the device).  Blocks go back to the driver in pf_destroy.
int pf_timing_collect(pf_ctx* c);  // pf_api.hip: fold finished spans into op_ms / op_launches / op_bytes
hipError_t pf_malloc(hipStream_t st, void** p, size_t bytes);
// the ctx's pinned host block for small transfers, at least `bytes` large (valid until the next call that asks for more;
// users synchronise with the stream before they return)
int pf_pinned_scratch(pf_ctx* c, size_t bytes, void** out, int sid = 0);  // sid 0: ctx stream, 1: stream_b's job, 2: the KNN read-back
// src -> dst by a copy KERNEL on `st` (one of them pinned host memory, read or written in place: no DMA engine involved);
// sizes rounded up to 8 bytes
int pf_copy_by_kernel(hipStream_t st, const void* src, void* dst, size_t bytes);
int pf_downloads_release(pf_ctx* c);  // queue every download of the ctx that was held back (behind what the ctx stream holds now)
int pf_download_cancel(pf_graph* g);  // forget the image owed to the caller's buffer / wait for the one in flight (a call is about to fail)
// `waiter_sid` (0: stream, 1: stream_b) waits for everything queued on the other stream so far; afterwards it may reuse
// the blocks the other stream has released, and use what the other stream has written
int pf_streams_join(pf_ctx* c, int waiter_sid);
hipStream_t pf_stream_b(pf_ctx* c);  // created on first use (nullptr on failure)
hipError_t pf_create_side_stream(hipStream_t* s, bool low = false);  // low: the least priority (a third pool of queues: the copy stream)  // a stream that never shares a hardware queue with a ctx's main stream
void pf_worker_run(pf_ctx* c, std::function<void()> task);  // starts `task` on the ctx's worker thread (one at a time)
void pf_worker_wait(pf_ctx* c);                             // until that task has returned
void pf_free(hipStream_t st, void* p);

// SELL-64 entry layout inside a slice of `width` entries per row: entries come in PAIRS per lane, so that one
// lane reads two values with one 16-byte load and two column indices with one 8-byte load (the widest
// coalesced access: 1 KiB of values per wave instruction); a slice of odd width keeps its last entry in a
// plain 64-lane row behind the pairs.  No padding is added by the pairing.
__host__ __device__ static inline int64_t pf_sell_index(int64_t base, int32_t width, int32_t j, int lane) {
    const int32_t pairs = width >> 1;
    return j < 2 * pairs ? base + (int64_t)(j >> 1) * (2 * PF_WAVE) + 2 * lane + (j & 1)
                         : base + (int64_t)pairs * (2 * PF_WAVE) + lane;
}

static inline double* pf_slot(pf_graph* g, int32_t s) { return g->ws + (int64_t)s * g->n_pad; }
static inline double* pf_tmp(pf_graph* g, int which) { return g->ws + (int64_t)(g->n_slots + which) * g->n_pad; }

// pf_scan.hip
int pf_exclusive_scan_i32(hipStream_t st, const int32_t* in, int32_t* out, int64_t n);
int pf_exclusive_scan_i64(hipStream_t st, const int64_t* in, int64_t* out, int64_t n);

// pf_operator.hip
int pf_reduce_ensure(pf_graph* g, int32_t count);

// pf_reorder.hip
// d_overflow (device int, nullable): the Morton order by counting (pf_reorder.hip); set to 1 when a cell holds too many
// vertices for that - the caller then repeats the call with nullptr (the general sort)
int pf_compute_order(pf_graph* g, const double* d_pts, int32_t* d_overflow = nullptr);
// the Morton order of the mesh's points, before anything else of a build: fills g->morder / g->mrank (allocated here) and
// g->order_bbox; d_overflow (nullable: general sort) is raised when the counting sort met a pile of vertices in one cell
int pf_morton_order(pf_graph* g, const double* d_pts, int32_t* d_overflow);

// Rows per window of the resident Chebyshev kernel: one window per block, at most 256 blocks.
static inline int32_t pf_window_rows(int64_t n_pad) { return n_pad <= 262144 ? 1024 : (n_pad <= 524288 ? 2048 : 4096); }

// pf_knn_tree.hip: the 1-NN search of pf_knn_run through a bounding-box hierarchy over all d coordinates
// (spectral coordinates of 250k blob pairs, ms grid / hierarchy: d = 6: 1.15 / 1.90, 7: 1.98 / 1.91, 8: 29.5 / 8.8, 10: 6.3 / 2.1;
// 1M pair, d = 10: 133 / 19.6 - profiles/r03_knn_hierarchy.md)
constexpr int PF_KNN_TREE_MIN_D = 7;
int pf_knn_tree_run(pf_ctx* c);
extern "C" {
int pf_gram_begin(pf_graph* g, int32_t first_a, int32_t count_a, int32_t first_b, int32_t count_b, int32_t append);
int pf_resnorms_begin(pf_graph* g, int32_t ax_first, int32_t x_first, const double* lam, int32_t count);
int pf_small_end(pf_graph* g, double* out);
int pf_combine2(pf_graph* g, int32_t src_first, int32_t m, const double* Y, int32_t k, int32_t dst_first, int32_t src_first2,
                int32_t dst_first2);
}

// pf_persist.hip: a whole recurrence T_degree((c - A)/e)/rho^degree src -> dst in ONE kernel (operator in registers,
// x in LDS, neighbouring windows hand their boundary rows over through memory)
struct pf_persist_args {
    pf_graph* g;
    const double* vals;
    const double* src;
    double* dst;
    int32_t degree;
    double c, e, rho;
};
int pf_persist_cheb(const pf_persist_args* a, const pf_persist_args* b /* nullable */, int* done, double* lds_bytes /* += */,
                    bool first_try = true /* false: a further attempt for the same filter application (pf_cheb2's single-graph launches) */);
int pf_persist_check(pf_ctx* ctx);  // PF_E_PERSIST_TIMEOUT (stream drained, path switched off) if a wait of an earlier launch ran out
int pf_persist_set(int on);
void pf_persist_release(pf_ctx* ctx);  // pf_destroy: another ctx may take the resident path over
// pf_windows.hip
int pf_window_slots_prepare(pf_graph* g);  // px_* of the graph (see pf_graph)
int pf_window_slots_begin(pf_graph* g);    // ... queued only; _prepare collects
bool pf_persist_enabled();
bool pf_persist_trusted();  // kernels with grid-wide waits may be used (enabled, not suspended after a timeout)
uint64_t pf_persist_abort_epoch();  // bumped by every bounded wait that ran out (device counters of that time are void)
int pf_persist_sync_ensure(pf_ctx* ctx);  // ctx->persist_sync / persist_abort exist from here on
int pf_window_rings_prepare(pf_graph* g);  // the second-ring structures on top of them (px2_state)
void pf_window_slots_free(pf_graph* g);
