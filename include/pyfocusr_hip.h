/*
 * pyfocusr_hip.h — C-ABI of libpyfocusr_hip.so: the MI355X (gfx950) implementation of
 * pyfocusr's spectral-embedding hot path.
 *
 * The reference (gattia/pyfocusr) is pure Python with no FFI layer of its own; its boundary
 * for this path is the Python class API (Graph / eigsort / Focusr).  The entry points below
 * are what a ctypes binding of that path needs; each one names the reference code it
 * replaces (file:line in /root/reference/pyfocusr).  The Python mirror of the reference
 * classes that calls them lives in pyfocusr_amd/{graph,eigsort,focusr}.py; INTEGRATION.md
 * shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - plain C: pointers + sizes, no C++/torch types; all functions return 0 on success or a
 *    negative PF_E_* code (message via pf_last_error()); nothing throws.
 *  - host arrays are caller-owned, C-contiguous; the library never keeps a host pointer.
 *  - one pf_ctx per (device, HIP stream); a pf_graph belongs to the ctx that built it.  Calls
 *    on one ctx must not overlap in time; distinct ctxs are independent.
 *  - "slot" = one length-n float64 vector in the graph's device workspace.
 */
#ifndef PYFOCUSR_HIP_H
#define PYFOCUSR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_VERSION 1

#define PF_OK 0
#define PF_E_ARG (-1)        /* bad argument (null pointer, size, slot range, face index out of range) */
#define PF_E_HIP (-2)        /* HIP runtime error (allocation, launch, copy); see pf_last_error() */
#define PF_E_DEGENERATE (-3) /* a face repeats a vertex: the reference would store W_ii = inf */
#define PF_E_STATE (-4)      /* call order violated (e.g. knn run before upload) */
#define PF_E_PERSIST_TIMEOUT (-5) /* a wait inside the resident Chebyshev kernel ran out (device shared?): the filter
                                     applications since the last synchronising call are invalid; the stream has been
                                     drained and the path switched off - repeat the solve (pf_eigs_smallest and the
                                     Python drivers do so by themselves) */

/* operator selector for the eigensolver kernels */
#define PF_OP_RW 0  /* L = G (D - W), G = diag(1/(deg+1e-8))   graph.py:216-226 (as stored by the reference) */
#define PF_OP_SYM 1 /* S = G^1/2 (D - W) G^1/2: same spectrum, only when W is symmetric */

typedef struct pf_ctx pf_ctx;
typedef struct pf_graph pf_graph;
typedef struct pf_mesh pf_mesh; /* points + faces resident in HBM */

typedef struct pf_graph_info {
    int64_t n;             /* vertices */
    int64_t n_faces;
    int64_t nnz_w;         /* unique directed edges = nnz(W)                       graph.py:178 */
    int64_t nnz_l;         /* nnz of scipy's L = nnz_w + #(vertices with deg>0)     graph.py:226 */
    int32_t is_symmetric;  /* W == W^T (structure; values then agree bit for bit)  */
    int32_t n_isolated;    /* vertices referenced by no face (deg == 0)            */
    int32_t n_components;  /* weakly connected components with >= 2 vertices       */
    int32_t max_degree;    /* longest row of W                                      */
    int64_t sell_entries;  /* stored off-diagonal slots incl. padding (SELL-64)     */
    int64_t n_pad;         /* workspace slot stride in elements                     */
    int64_t n_oneway;      /* entries (i,j) of W without a matching (j,i): 0 iff symmetric;
                              every boundary edge of an open mesh is one (graph.py:178)      */
    double spectral_bound; /* proven upper bound of the spectrum of L (and of S): 2 in general; for a closed triangle
                              mesh (every edge in exactly two faces) 1 + (1 + sqrt(1 - 4 P_min)) / 2 with P_min the
                              smallest 2abc / ((a+b)(a+c)(b+c)) over the faces' edge weights - the Chebyshev
                              filter damps [cut, spectral_bound] instead of [cut, 2]                  */
} pf_graph_info;

typedef struct pf_timing {
    double op_ms;        /* accumulated device time of fused SpMV/Chebyshev launches (HIP events) */
    int64_t op_launches; /* number of those launches                                              */
    double op_bytes;     /* algorithmic bytes they moved: sum of 12 nnz + 20 n + 4 per graph and step */
    double knn_ms;       /* device time of the last pf_knn_run                                    */
    double build_ms;     /* device time of the last pf_graph_build (kernels only)                 */
    /* the part of the above done by the resident kernel (pf_persist_enable): one launch = a whole recurrence */
    double persist_ms;
    int64_t persist_launches;
    int64_t persist_steps; /* recurrence steps those launches ran, summed over the graphs of a launch (graph-steps) */
    double persist_bytes;  /* algorithmic bytes, counted as for op_bytes                                   */
    double persist_lds_bytes; /* LDS bytes the resident launches moved: per graph and step 8 B per stored SELL entry
                                 (the gathered x) + 16 B per row (own x read, result written) + 8 B per outside row */
} pf_timing;

/* ---- context -------------------------------------------------------------------------- */
int pf_version(void);
const char* pf_last_error(void);
int pf_device_count(void);
int pf_create(int device, pf_ctx** out);
void pf_destroy(pf_ctx* ctx);
int pf_sync(pf_ctx* ctx);
void* pf_stream(pf_ctx* ctx); /* the ctx's hipStream_t, so that a peer library (RCCL through torch) can enqueue on it */
/* time operator launches with HIP events on the ctx stream: on = 1 every filter application, on = N > 1 every N-th (an
 * event record costs ~5 us of device time: two per application are 0.25 ms of a 13.5 ms pair step); the accumulated
 * figures of pf_timing_get then cover the timed applications only */
int pf_timing_enable(pf_ctx* ctx, int on);
int pf_timing_get(pf_ctx* ctx, pf_timing* out, int reset);

/* ---- Laplacian assembly --------------------------------------------------------------- */
/* Replaces Graph.get_weighted_adjacency_matrix / get_degree_matrix / get_G_matrix (no
 * features) / get_laplacian_matrix   (graph.py:148-178, 216-219, 213-214, 221-226).
 * pts: n x 3 float64; faces: F x verts_per_face int32 (VTK polygon edge order
 * (0,1),(1,2),...,(v-1,0)).  Directed "set" semantics: W[i,j] = 1/||xi-xj|| once per unique
 * directed edge.  Builds, on the device: CSR(W) with sorted columns, deg (row sums, left to
 * right in column order), the SELL-64 operator storage for L (and S when W is symmetric), and
 * connected-component labels. */
int pf_graph_build(pf_ctx* ctx, const double* pts, int64_t n, const int32_t* faces, int64_t n_faces,
                   int32_t verts_per_face, pf_graph** out);
/* the same in two steps: host -> HBM copy of the mesh, then assembly from resident inputs */
int pf_mesh_upload(pf_ctx* ctx, const double* pts, int64_t n, const int32_t* faces, int64_t n_faces,
                   int32_t verts_per_face, pf_mesh** out);
void pf_mesh_free(pf_mesh* mesh);
int pf_graph_build_device(pf_mesh* mesh, pf_graph** out);
/* The two meshes of a pair assembled side by side (mesh a on the ctx stream, mesh b on a second stream of the ctx, the
 * two builds' halves interleaved around their one synchronisation each): an assembly is ~75 small kernels that are
 * mostly launch latency, and two of them overlap almost completely.  Same results as two pf_graph_build_device calls. */
int pf_graph_build_device2(pf_mesh* mesh_a, pf_mesh* mesh_b, pf_graph** out_a, pf_graph** out_b);
/* A device graph from a general sparse matrix A in CSR (sorted, unique columns; explicit diagonal optional)
 * instead of a mesh: what `recursive_eig(matrix, ...)` (graph.py:357-389) needs when it is handed a scipy
 * matrix.  PF_OP_RW applies A itself; is_symmetric reports A == A^T numerically (PF_OP_SYM is then the same
 * operator); deg/l_diag hold the diagonal; rows without off-diagonal entries count as isolated. */
int pf_graph_from_matrix(pf_ctx* ctx, int64_t n, const int32_t* rowptr, const int32_t* colidx, const double* values,
                         pf_graph** out);
void pf_graph_free(pf_graph* g);
int pf_graph_get_info(pf_graph* g, pf_graph_info* out);
/* CSR(W): rowptr[n+1], colidx[nnz_w], w[nnz_w]; l_offdiag[nnz_w] = -w/(deg_i+1e-8);
 * per-vertex deg[n], l_diag[n] = deg/(deg+1e-8).  Any output may be NULL. */
int pf_graph_download(pf_graph* g, int32_t* rowptr, int32_t* colidx, double* w, double* l_offdiag,
                      double* deg, double* l_diag, int32_t* component_label);

/* ---- device workspace ------------------------------------------------------------------ */
int pf_ws_ensure(pf_graph* g, int32_t n_slots);
int pf_ws_upload(pf_graph* g, int32_t slot, const double* x);            /* x[n]            */
int pf_ws_download(pf_graph* g, int32_t first, int32_t count, double* out); /* out[count][n]  */
int pf_ws_copy(pf_graph* g, int32_t src, int32_t dst, int32_t count);
/* Krylov start vector into `slot`: a low-order polynomial of the vertex positions (rich in the low
 * eigenmodes) plus counter-based noise from `seed`; zero on isolated vertices.  Deterministic. */
int pf_start_vector(pf_graph* g, int32_t slot, uint64_t seed);
int pf_mask_isolated(pf_graph* g, int32_t slot);                          /* x[i] = 0 where deg_i == 0 */
/* unit-norm null vectors of `op`, one per component with >= 2 vertices, into slots
 * [0, n_components): 1_C for PF_OP_RW, sqrt(deg+1e-8) on C for PF_OP_SYM. */
int pf_lock_null_vectors(pf_graph* g, int32_t op, int32_t* n_locked);

/* ---- eigensolver kernels (replace scipy eigs/ARPACK+SuperLU at graph.py:372) ------------- */
int pf_spmv(pf_graph* g, int32_t op, int32_t src, int32_t dst);           /* dst = A src     */
/* slots [dst_first, dst_first+count) = A slots [src_first, src_first+count), one call (the Rayleigh-Ritz step's A Z) */
int pf_spmv_multi(pf_graph* g, int32_t op, int32_t src_first, int32_t dst_first, int32_t count);
/* dst = T_degree((c I - A)/e) src / rho^degree : `degree` launches of the fused SpMV + three-term
 * recurrence kernel (scaled by rho >= 1 per step so that high degrees cannot overflow; rho = 1 is the
 * plain Chebyshev polynomial).  src is preserved; dst != src. */
/* ON by default: pf_cheb / pf_cheb2 run the WHOLE recurrence in one resident kernel (one block per CU) when the
 * graph(s) fit: each block owns a window of 1024 / 2048 / 4096 consecutive rows of every graph, keeps the rows' entries
 * in registers and the window's x in LDS, and neighbouring windows hand their boundary rows over through memory, one
 * value per outside row, no flags (pf_persist.hip; graphs up to 256 windows: ~1M rows, pairs up to ~524k rows each).
 * Everything else runs one step per launch.  Results are bit-identical in all cases.  0 switches it off (also:
 * environment PF_PERSIST=0); process-wide.  Needs 4 x n_pad doubles of scratch per graph.  All blocks must be resident
 * together (grid <= CU count; an idle device): a wait that runs out (another tenant on the device) is reported as
 * PF_E_PERSIST_TIMEOUT at the next synchronising call, after the library has drained the stream and SUSPENDED the path:
 * the next 64 filter applications run one step per launch, then the resident path is tried again (every further timeout
 * doubles the suspension; pf_persist_enable(1) lifts it; pf_persist_state tells).  One ctx has resident kernels in
 * flight at a time: the path belongs to the ctx that used it last, another ctx takes it over as soon as the owner's
 * last resident launch has completed and runs one step per launch until then. */
int pf_persist_enable(int on);
/* Level 1 by default: single-graph recurrences (pf_cheb) on graphs with windows of 1024 rows (up to ~262k rows) whose
 * windows see each other symmetrically (any symmetric W) exchange boundary values every SECOND step: a window repeats
 * the odd steps of the outside rows it reads itself (k_cheb_resident2: one memory-side hand-off per two steps; 250k
 * rows: 1.41 -> 1.24 us per step).  Level 2: paired recurrences (pf_cheb2) as well - measured slower than one step
 * per exchange there (2.08 against 1.85 us per step of a 250k pair), kept for the record.  0: one step per exchange
 * everywhere.  Bit-identical results at every level.  Environment: PF_PERSIST_S2=0/1/2; process-wide. */
int pf_persist_two_step(int level);
/* On by default: a paired recurrence (pf_cheb2) on windows of 1024 rows whose outside-row lists fit half a block runs the
 * kernel whose two halves take the graphs in opposite order (k_cheb_resident<2,1,8,true>), so that both graphs'
 * boundary rows are handed over one row's latency into a step.  0: both graphs in the same order everywhere.
 * Bit-identical results either way.  Environment: PF_PERSIST_HALVES=0/1; process-wide. */
int pf_persist_pair_halves(int on);
/* The resident launches of `ctx` since the last reset as their first block saw them: how many completed, and their summed
 * run time on the device's constant 100 MHz clock (read at the block's first and last instruction).  A cross-check of
 * pf_timing.persist_ms, whose HIP event pairs also contain the dispatch of the kernel and the event packets themselves
 * (~20 us per launch).  Synchronises the ctx stream. */
int pf_persist_clock(pf_ctx* ctx, double* kernel_ms, int64_t* launches, int reset);
/* What the resident path is doing, for callers that want to know whether they are on the fast path. */
typedef struct pf_persist_info {
    int32_t enabled;           /* 1: filter applications use the resident kernels where a graph allows it             */
    int32_t two_step;          /* the pf_persist_two_step level: 0, 1 (single-graph recurrences) or 2 (pairs too)      */
    int32_t owner;             /* 1: `ctx` owns the path, 0: no ctx has used it yet, -1: another ctx of the process  */
    int32_t timeouts;          /* waits that ran out since the process started (each reported as PF_E_PERSIST_TIMEOUT) */
    int64_t launches;          /* resident launches of this process                                                   */
    int64_t launches_two_step; /* ... of them with two steps per exchange                                             */
    int32_t suspended_for;     /* > 0: a wait ran out; this many filter applications still run one step per launch
                                  before the resident path is tried again (64 after the first timeout, doubled by every
                                  further one; environment PF_PERSIST_REARM = the base, 0 = never again)              */
    int32_t rearms;            /* suspensions that have ended                                                         */
    int32_t owner_switches;    /* times the path moved from one ctx to another (the previous owner's launches had
                                  completed; while they are in flight the other ctx runs one step per launch)         */
    int32_t hold_ticks;        /* the hold-back of the latest resident launch, 10 ns ticks after a step began (0: fixed sleep) */
} pf_persist_info;
int pf_persist_state(pf_ctx* ctx /* nullable */, pf_persist_info* out);
/* Test hook: the next n resident launches start with their abort flag raised (they give up at once and the
 * PF_E_PERSIST_TIMEOUT recovery runs).  Never needed in production. */
int pf_persist_test_hook(int n_launches);
int pf_cheb(pf_graph* g, int32_t op, int32_t src, int32_t dst, int32_t degree, double c, double e, double rho);
/* Two independent recurrences (graphs a and b of one ctx) advanced in lockstep: step k of both in
 * ONE launch while both have steps left, the longer one alone afterwards. */
int pf_cheb2(pf_graph* ga, int32_t op_a, int32_t src_a, int32_t dst_a, int32_t degree_a, double c_a, double e_a, double rho_a,
             pf_graph* gb, int32_t op_b, int32_t src_b, int32_t dst_b, int32_t degree_b, double c_b, double e_b, double rho_b);
int pf_dots(pf_graph* g, int32_t w, int32_t first, int32_t count, double* out);  /* out[b] = <slot first+b, slot w> */
/* classical Gram-Schmidt of slot w against slots [first, first+count), a second pass when the first one cancelled
 * digits: h[count] = summed coefficients, *nrm = ||w|| afterwards (w is left un-normalised). */
int pf_orth(pf_graph* g, int32_t w, int32_t first, int32_t count, double* h, double* nrm);
/* The same in two phases, so that the host can queue the next filter application before it reads the
 * coefficients: begin enqueues the kernels (and, if `normalize`, w <- w/||w|| with the norm taken on the
 * device) plus an asynchronous copy of the results; end waits for them.  One orth in flight per graph. */
int pf_orth_begin(pf_graph* g, int32_t w, int32_t first, int32_t count, int32_t normalize);
int pf_orth_end(pf_graph* g, double* h, double* nrm);
/* pf_orth_begin for the two graphs of a pair (one ctx) in shared launches, as pf_cheb2 does for the filter; each
 * graph's coefficients are collected with its own pf_orth_end. */
int pf_orth_begin2(pf_graph* ga, int32_t w_a, int32_t first_a, int32_t count_a, int32_t normalize_a, pf_graph* gb, int32_t w_b,
                   int32_t first_b, int32_t count_b, int32_t normalize_b);
/* pf_orth_begin2 and, queued right behind it, pf_cheb2 in one call (one outer step of a pipelined pair driver: the device
 * does not wait for the host between the two).  orth[8] = {w, first, count, normalize} of a, then of b; cheb_i[8] =
 * {op, src, dst, degree} of a, then of b; cheb_d[6] = {c, e, rho} of a, then of b. */
int pf_orth_cheb2(pf_graph* ga, pf_graph* gb, const int32_t* orth, const int32_t* cheb_i, const double* cheb_d);
/* 1 (2: see pf_orth_device_passes) if the last pf_orth_end found that the first Gram-Schmidt pass had cancelled digits (|w'| < 0.3 |w|) and ran the
 * second pass itself, after everything queued behind pf_orth_begin: work queued in between that READ slot w (the next
 * filter application of a pipelined driver) saw the un-refined, un-normalised vector and has to be repeated.  Rare:
 * never on the 250k blobs, a few times per solve right after a restart on small graphs. */
int pf_orth_redone(pf_graph* g);
/* on != 0: the second Gram-Schmidt pass runs whenever |w'| < 0.71 |w| (the classical constant) instead of 0.3 |w|: for
 * iterations that come close to exhausting a small space (unfiltered solves of tiny graphs), where the loose criterion
 * loses orthogonality.  on = 2: EVERY step takes the second pass (the full steps of Lanczos with partial
 * reorthogonalisation: one pass against a basis that is orthogonal to 1e-9 only would leave the new vector at that level).
 * Per graph; off by default. */
int pf_orth_strict(pf_graph* g, int32_t on);
/* on != 0: pf_orth_begin / pf_orth_begin2 / pf_orth_cheb2 queue the second pass together with the first; it runs on the
 * device's own verdict (two launches that return at once when the first pass was fine) and reports h1 + h2 itself, so
 * work queued behind the step never reads a stale w: pf_orth_redone then returns 2 ("two passes, nothing to repeat")
 * instead of 1.  For iterations whose steps often cancel digits - restarted Arnoldi with strongly amplified outliers in
 * the basis (asymmetric W): a repeated filter application costs more than the ~5 us of the two idle launches.  Per graph;
 * off by default. */
int pf_orth_device_passes(pf_graph* g, int32_t on);
/* The NEXT pf_orth_begin / pf_orth_begin2 / pf_orth_cheb2 step of this graph - and only that one - takes its basis from TWO
 * ranges of slots: [first, first + split) and [first2, first2 + count - split), with `first` and `count` as passed to that
 * call; the coefficients come back in that order.  For Lanczos with partial reorthogonalisation (pf_eigs_smallest on
 * symmetric W): most steps orthogonalise against the locked null vectors and the last two basis vectors only. */
int pf_orth_split(pf_graph* g, int32_t first2, int32_t split);
/* Process-wide: the LOCAL Gram-Schmidt steps (four vectors at most, one pass: Lanczos with partial reorthogonalisation) as
 * ONE launch (k_orth_local: the blocks of a graph meet at a counter between the dot products and the projection) where
 * the device holds the whole grid at once and the resident path is trusted (pf_persist_state).  0: dot products and
 * projection as two launches, like every other step; 1 / -1: one launch (the default; PF_ORTH_LOCAL=0 in the
 * environment switches it off for the whole process).  The results are bit-identical either way. */
int pf_orth_one_launch(int32_t on);
int pf_scale(pf_graph* g, int32_t slot, double alpha);
/* slots [dst_first, dst_first+k) = slots [src_first, src_first+m) * Y, Y row-major m x k; ranges must not overlap */
int pf_combine(pf_graph* g, int32_t src_first, int32_t m, const double* Y, int32_t k, int32_t dst_first);
int pf_resnorm(pf_graph* g, int32_t ax, int32_t x, double lam, double* out);     /* ||ax - lam x||_2 */
/* the batched forms the Rayleigh-Ritz step uses (one copy and one synchronisation instead of one per vector):
 * out[i][j] = <slot first_a+i, slot first_b+j> (count_a x count_b, row-major); out[i] = ||slot ax_first+i - lam[i] slot x_first+i||_2 */
int pf_gram(pf_graph* g, int32_t first_a, int32_t count_a, int32_t first_b, int32_t count_b, double* out);
int pf_resnorms(pf_graph* g, int32_t ax_first, int32_t x_first, const double* lam, int32_t count, double* out);
/* Eigenvector post-processing (graph.py:254-257 + the sign/scale convention): for each of
 * `count` slots from `first`: x <- sqrt(g) .* x if from_sym; scale to unit 2-norm; flip so the
 * largest-|entry| (lowest index on ties) is positive; if minmax: (v - min)/(max - min) - 0.5.
 * out: host n x count row-major (numpy (n, count) C-order). */
int pf_finalize_vectors(pf_graph* g, int32_t first, int32_t count, int32_t from_sym, int32_t minmax,
                        double* out);
/* The same in two halves: _begin queues the kernels and - on the ctx's copy stream, behind an event - the download
 * into `out` (pinned memory from pf_host_alloc: one DMA; a pageable destination is staged in chunks), and returns;
 * the block is resident and usable by pf_final_rows / pf_knn1_graphs / pf_eigsort_costs at once, and later work on the
 * ctx stream overlaps with the download.  _end waits for the download and checks the result (PF_E_STATE for a
 * vanished vector); `out` must not be read before it returns.  No-op when nothing is pending.  pf_graph_free and the
 * next _begin collect a pending download themselves.  Since round 3 the n x count image may be held back: it is queued
 * behind the next long kernel of the ctx (a 1-NN search of >= 32768 queries) or by _end, whichever comes first - beside
 * the small kernels that follow a solve a 10 MB download costs what it would cost in line.  `out` must therefore stay
 * allocated until _end (or pf_graph_free) has returned; pf_host_free of a block that is still owed its image cancels
 * the download.  PF_DOWNLOAD_DEFER=0: queued at once, as before. */
int pf_finalize_vectors_begin(pf_graph* g, int32_t first, int32_t count, int32_t from_sym, int32_t minmax, double* out);
int pf_finalize_vectors_end(pf_graph* g);
/* out[i][c] = block[i][col[c]] * sign[c] (c < count <= the block's column count, sign = +-1): the image on the host of
 * eigsort's sign flips and column moves (eigsort.py:108-122), computed on the device from the resident block and copied
 * like the first download (pinned `out`, copy stream; collected by pf_finalize_vectors_end).  The block stays as it is. */
int pf_final_remap_begin(pf_graph* g, const int32_t* col, const double* sign, int32_t count, double* out);
/* Pinned (page-locked) host memory for results that should arrive by one DMA; independent of any ctx. */
int pf_host_alloc(size_t bytes, void** out);
int pf_host_free(void* p);
/* Before a pinned block is reused without being freed (a caller-side pool): forget any eigenvector download the library
 * still owes to it (held back by pf_finalize_vectors_begin; left behind by a call that failed) and wait for one in flight.
 * pf_host_free does the same before it unmaps the block. */
int pf_host_detach(void* p);
/* The block written by the last pf_finalize_vectors stays resident in HBM.  out[t][c] = that block's row rows[t]
 * (n_rows x count, row-major): the sampled eigenvector rows of Graph.get_rand_eig_vecs (graph.py:266-267) without
 * touching the host copy. */
int pf_final_rows(pf_graph* g, const int64_t* rows, int64_t n_rows, double* out);
/* the same for the mesh's points (graphs built from a mesh keep them): out[t] = pts[rows[t]] (n_rows x 3), the gather
 * inside Graph.get_rand_normalized_points (graph.py:269-272) */
int pf_point_rows(pf_graph* g, const int64_t* rows, int64_t n_rows, double* out);
/* y = A x on host vectors (tests / roofline probes) */
int pf_spmv_host(pf_graph* g, int32_t op, const double* x, double* y);
/* Repeated mean filter out = (D+I)^-1 (W+I) applied `iterations` times to values[n][ncols]
 * (row-major), graph.py:320-354. */
int pf_mean_filter(pf_graph* g, const double* values, int32_t ncols, int32_t iterations, double* out);

/* ---- nearest neighbour (replaces scipy KDTree(ref).query(qry), k=1, p=2) ------------------ */
/* focusr.py:351-353 (spectral coordinates, d = n_spectral_features) and eigsort.py:203-204
 * (d = 3); d <= 16.  Exact search (uniform-grid pruning, result identical to exhaustive search): squared
 * distance accumulated left to right over the d
 * coordinates without FMA contraction, lowest reference index wins ties.
 * ref: n_ref x d, qry: n_qry x d row-major float64; idx_out[n_qry] int64; d2_out nullable. */
int pf_knn1(pf_ctx* ctx, const double* ref, int64_t n_ref, const double* qry, int64_t n_qry, int32_t d,
            int64_t* idx_out, double* d2_out);
/* How k = 1 searches prune (the results are the same bits either way).  0 (default): by depth - d <= 6 through a grid
 * over the references' two widest axes (pf_knn.hip), d >= 7 through a hierarchy of bounding boxes over ALL d coordinates
 * (pf_knn_tree.hip: leaves of 64 Morton-ordered points, supers of 64 leaves; what keeps deep, poorly aligned embeddings
 * - BASELINE config C5, 1M x 1M, d = 10 - from degenerating into a brute force inside a 2-D rectangle); 1: always the
 * grid; 2: always the hierarchy. */
int pf_knn_mode(pf_ctx* ctx, int32_t mode);
/* Visits of the last hierarchy search, summed over the waves (counted only while enable_counting was 1 for that search;
 * the call also sets the switch for the searches to come). */
int pf_knn_tree_stats(pf_ctx* ctx, int32_t enable_counting, int64_t* leaves_scanned, int64_t* supers_opened);
/* The same for the grid search (k = 1, d <= 9): candidate-query pairs whose squared distance the last counted search
 * evaluated (3 d floating-point operations each): the work behind the 1-NN stage's roofline entry in bench.py. */
int pf_knn_count(pf_ctx* ctx, int32_t enable_counting, int64_t* pairs);
/* k nearest neighbours (1 <= k <= 4, d <= 4), ascending by (distance, index): the 3-NN of
 * Focusr.get_weighted_final_node_locations (focusr.py:409-412).  idx_out / d2_out: n_qry x k row-major. */
int pf_knn(pf_ctx* ctx, const double* ref, int64_t n_ref, const double* qry, int64_t n_qry, int32_t d, int32_t k,
           int64_t* idx_out, double* d2_out);
/* focusr.py:351-353 with the coordinates taken from the two graphs' resident pf_finalize_vectors blocks instead of
 * host arrays: ref[i][c] = final_ref[i][col_ref[c]] * scale_ref[c], qry likewise (c < d) - the column selection, sign
 * flips and permutation of eigsort (eigsort.py:108-122) and the spectral weights (focusr.py:481-501) are folded into
 * col / scale, so the n x k coordinates never cross PCIe.  Same search, same arithmetic as pf_knn1. */
int pf_knn1_graphs(pf_graph* ref_g, pf_graph* qry_g, int32_t d, const int32_t* col_ref, const double* scale_ref,
                   const int32_t* col_qry, const double* scale_qry, int64_t* idx_out, double* d2_out);
/* The same on any two row-major blocks in device memory (row strides in doubles): what the ranks of a split pair hold
 * after the RCCL all-gather of their resident blocks - qry_block may point at a row range of a block, the
 * query shard of this rank.  The pointers must be valid on ctx's device; work is enqueued on ctx's stream. */
int pf_knn1_blocks(pf_ctx* ctx, const double* ref_block, int64_t n_ref, int32_t ref_stride, const double* qry_block,
                   int64_t n_qry, int32_t qry_stride, int32_t d, const int32_t* col_ref, const double* scale_ref,
                   const int32_t* col_qry, const double* scale_qry, int64_t* idx_out, double* d2_out);
/* eigsort's cost matrices (eigsort.py:162-233) from the graphs' device-resident eigenvector blocks and point copies:
 * m_t / m_s <= 16384 sampled rows of the target / source graph (rows_t / rows_s), the first k eigenmaps of each as
 * final[:, col[c]] * sign[c] (the column permutation and sign flips earlier eigsort calls left, identity at first).
 * out[4][k][k] = c_hist, c_hist_f, c_spatial, c_spatial_f (row = target map, column = source map); idx_out[m_t] = the
 * sampled source point nearest to each sampled target point (min-max normalised xyz, eigsort.py:203-204). */
int pf_eigsort_costs(pf_graph* g_target, pf_graph* g_source, const int64_t* rows_t, int64_t m_t, const int64_t* rows_s, int64_t m_s,
                     int32_t k, const int32_t* col_t, const double* sign_t, const int32_t* col_s, const double* sign_s, double* out,
                     int64_t* idx_out);
/* Device address and shape of the resident block ([n_rows][n_cols] row-major float64, owned by the graph, valid until
 * the next pf_finalize_vectors on it or pf_graph_free): lets a peer library (RCCL through torch) send it without a
 * host round trip. */
int pf_final_device(pf_graph* g, double** block, int64_t* n_rows, int32_t* n_cols);
/* split form (inputs resident in HBM across the timed region) */
int pf_knn_upload(pf_ctx* ctx, const double* ref, int64_t n_ref, const double* qry, int64_t n_qry, int32_t d);
int pf_knn_run(pf_ctx* ctx);
int pf_knn_download(pf_ctx* ctx, int64_t* idx_out, double* d2_out);

/* ---- the whole eigensolve in one call --------------------------------------------------------------------------
 * Replaces scipy.sparse.linalg.eigs(L, k, sigma=1e-10, which="LM", ncv=4k) at graph.py:372 (called from
 * recursive_eig, graph.py:357-389) for callers that bind the C-ABI without the Python driver: the n_wanted lowest
 * eigenpairs with eigenvalue > 1e-10 (graph.py:381) of the graph's random-walk Laplacian, ascending; vecs is
 * [n][*n_out] row-major, unit 2-norm, sign fixed (largest-|entry| positive), min-max normalised to [-0.5, 0.5] when
 * minmax != 0 (graph.py:254-257).  The reference's widen-and-retry rule (graph.py:369-384) only changes HOW MANY pairs
 * it asks for: a caller that wants its column count asks for k_final - #nulls, with #nulls = n_components +
 * n_isolated of pf_graph_get_info.  Symmetric W: thick-restart Lanczos on S = G^1/2 (D - W) G^1/2.  Asymmetric W
 * (one-way edges, graph.py:178: both bundled 15k meshes): restarted Arnoldi on L itself - the complex eigenvalues of
 * the non-normal L are carried as dominant Ritz values of the interval filter, or, when the low eigenvalues are complex
 * themselves (open surfaces), enclosed by the ellipse filter; real parts are returned as the reference does
 * (graph.py:386-389), a conjugate pair as a repeated value with a fixed phase.  Graphs of fewer than ~100 vertices and
 * operators whose wanted eigenvalues are no corner of the spectrum are refused with PF_E_STATE (pyfocusr_amd/_krylov.py
 * keeps the unfiltered mode for them).  vals needs room for n_wanted values, vecs for n * n_wanted. */
typedef struct pf_eigs_stats {
    int64_t matvecs;      /* SpMV-equivalent launches */
    int32_t outer_steps;  /* Lanczos steps */
    int32_t restarts;
    int32_t filter_resets;
    int32_t degree;       /* of the last Chebyshev filter */
    int32_t n_null;       /* eigenvalues <= 1e-10 found among the computed ones (= locked null vectors) */
    double cut;           /* lower end of the damped interval */
    double max_residual;  /* max ||S x - lambda x||_2 of the returned pairs */
    int32_t second_passes; /* outer steps whose Gram-Schmidt projection cancelled digits (second pass run) */
    int32_t mode;          /* 0: symmetric W (Lanczos on S); 1: asymmetric W, interval filter (Arnoldi on L, complex outliers carried); 2: ellipse filter */
    int32_t local_steps;   /* symmetric W: outer steps that orthogonalised against the null vectors and the last two basis vectors only
                              (partial reorthogonalisation; the others were full Gram-Schmidt steps) */
    int32_t reserved;
} pf_eigs_stats;
int pf_eigs_smallest(pf_graph* g, int32_t n_wanted, int32_t minmax, double* vals, double* vecs, int32_t* n_out,
                     pf_eigs_stats* stats);
/* The same with the options of the pair call: residuals (nullable) = ||A x - lambda x||_2 per returned pair;
 * async_download = 1: the call returns with the eigenvector download still in flight into vecs (pinned: pf_host_alloc) -
 * pf_finalize_vectors_end(g) before reading it. */
int pf_eigs_smallest_ex(pf_graph* g, int32_t n_wanted, int32_t minmax, int32_t async_download, double* vals, double* vecs,
                        double* residuals, int32_t* n_out, pf_eigs_stats* stats);
/* The two graphs of a pair (target and source mesh of focusr.py:134-170; one ctx; symmetric or not, each by its own) solved TOGETHER: the
 * two iterations advance in lockstep, every Gram-Schmidt step and filter application that both have pending runs in
 * launches the graphs share (pf_orth_cheb2: one library call and three launches per outer step of the pair), one graph
 * finishes alone once its partner has converged.  res_a / res_b (nullable): ||S x - lambda x||_2 per returned pair.
 * async_download = 1: the call returns with the eigenvector downloads still in flight (pf_finalize_vectors_begin into
 * vecs_a / vecs_b, which should be pinned: pf_host_alloc) - pf_finalize_vectors_end(g) before reading them; the
 * device-resident blocks are usable at once.  vals need room for n_wanted values, vecs for n * n_wanted; when a graph
 * returns fewer pairs (*n_out < n_wanted) its vecs hold an n x *n_out block. */
int pf_eigs_smallest2(pf_graph* ga, pf_graph* gb, int32_t n_wanted_a, int32_t n_wanted_b, int32_t minmax, int32_t async_download,
                      double* vals_a, double* vecs_a, double* res_a, int32_t* n_out_a, pf_eigs_stats* stats_a,
                      double* vals_b, double* vecs_b, double* res_b, int32_t* n_out_b, pf_eigs_stats* stats_b);

/* ---- primitives of the row-partitioned solve (one large mesh over several GPUs; SURVEY 8e / BASELINE config C5).
 * The reference has no counterpart (scipy eigs on one core, graph.py:357-389).  pyfocusr_amd/rowpart.py drives them.
 *   pf_op_step     one step of a three-term recurrence: out = alpha (shift x - A x) - beta prev   (slots; prev = -1:
 *                  no prev term; out may be the prev slot)
 *   pf_cheb_steps  steps k_first .. k_first+n_steps-1 of the Chebyshev recurrence of pf_cheb on explicit state: slot
 *                  `cur` holds y_{k_first-1}, slot `prev` holds y_{k_first-2} (ignored and overwritten when
 *                  k_first = 1); each step writes y_k over y_{k-2}; *out_cur / *out_prev tell where y_last and
 *                  y_{last-1} ended up
 *   pf_axpy        slot w += sum_i coef[i] * slot (first + i)
 *   pf_rows_*      a fixed subset of rows (mesh-order indices): gather its values of a slot to the host, scatter
 *                  host values into it, or fill it with a constant — the boundary / ghost rows of a partition. */
typedef struct pf_rows pf_rows;
int pf_op_step(pf_graph* g, int32_t op, int32_t x, int32_t prev, int32_t out, double alpha, double shift, double beta);
int pf_cheb_steps(pf_graph* g, int32_t op, int32_t prev, int32_t cur, int32_t k_first, int32_t n_steps, double c, double e,
                  double rho, int32_t* out_prev, int32_t* out_cur);
int pf_axpy(pf_graph* g, int32_t w, int32_t first, int32_t count, const double* coef);
int pf_rows_create(pf_graph* g, const int64_t* rows, int64_t n, pf_rows** out);
void pf_rows_free(pf_rows* r);
int pf_rows_gather(pf_rows* r, int32_t slot, double* out);
int pf_rows_scatter(pf_rows* r, int32_t slot, const double* in);
int pf_rows_gather_dev(pf_rows* r, int32_t slot, double* dst_device);        /* device buffers of the caller (e.g. a */
int pf_rows_scatter_dev(pf_rows* r, int32_t slot, const double* src_device); /* tensor RCCL sends / received); no sync */
/* one launch each way for BOTH vectors of a boundary exchange: dst[t] = slot_a[row t], dst[stride + t] = slot_b[row t]
 * (slot_b = -1: only a);  slot_a[row t] = src[off t], slot_b[row t] = src[off t + stride] with the per-row offsets
 * of pf_rows_set_sources (where each ghost row's value sits in the all-gathered receive buffer) */
int pf_rows_set_sources(pf_rows* r, const int64_t* offsets);
int pf_rows_gather2_dev(pf_rows* r, int32_t slot_a, int32_t slot_b, double* dst_device, int64_t stride);
int pf_rows_scatter2_dev(pf_rows* r, int32_t slot_a, int32_t slot_b, const double* src_device, int64_t stride);
int pf_rows_fill(pf_rows* r, int32_t slot, double value);

/* ---- closest point on a triangulated surface (ICP pre-alignment, "next" row f3) --------------------------
 * Replaces the vtkCellLocator::FindClosestPoint loop inside vtkIterativeClosestPointTransform, which the
 * reference runs through vtk_functions.py:12-29 (called from focusr.py:110-131).  Polygons with more than three
 * vertices are fan-triangulated (0,j+1,j+2).  Exact: the minimum over all triangles of the exact point-triangle
 * distance; lowest face index on exact ties.
 *   pf_surface_create   points [n][3] f64, faces [n_faces][verts_per_face] i32 (host) -> device search structure
 *   pf_surface_closest  qry [n_qry][3] f64 (host) -> out_pts [n_qry][3] closest surface points, out_face [n_qry]
 *                       face index (-1 and NaN point for a NaN query), out_d2 [n_qry] squared distances; each
 *                       output may be NULL. */
typedef struct pf_surface pf_surface;
int pf_surface_create(pf_ctx* ctx, const double* points, int64_t n, const int32_t* faces, int64_t n_faces,
                      int32_t verts_per_face, pf_surface** out);
void pf_surface_free(pf_surface* s);
int pf_surface_closest(pf_surface* s, const double* qry, int64_t n_qry, double* out_pts, int32_t* out_face,
                       double* out_d2);

/* ---- Coherent Point Drift pieces ("next" row f4) ------------------------------------------------------------
 * The reference registers the spectral coordinates with the third-party cycpd package (focusr.py:297-334).
 * These are the two O(M*N) operations of its EM iteration, matrix-free (P and G are never stored):
 *   pf_cpd_create  X [N][d] fixed set, Y [M][d] moving set (host) -> handle; the moving set's current position
 *                  TY starts as Y.
 *   pf_cpd_estep   TY [M][d] (host; NULL = keep the device copy), sigma2 > 0, outlier weight 0 <= w < 1 ->
 *                  P1 [M] = P 1, Pt1 [N] = P^T 1, PX [M][d] = P X with
 *                  P_mn = exp(-|x_n - ty_m|^2 / 2 sigma2) / (sum_m' exp(..) + (2 pi sigma2)^(d/2) w/(1-w) M/N)
 *                  (a zero column sum is replaced by DBL_EPSILON first).  Outputs may be NULL.
 *   pf_cpd_set_basis / pf_cpd_weighted_gram   Q [M][K] (host) stays on the device; H [K][K] = Q^T diag(P1) Q with
 *                  the P1 of the last pf_cpd_estep: the K x K matrix of the low-rank (Woodbury) deformable M-step.
 *   pf_cpd_gram    out [n_a][n_cols] = G(A,B) V,  G_ij = exp(-|a_i - b_j|^2 / 2 beta^2), V [n_b][n_cols] (all host). */
typedef struct pf_cpd pf_cpd;
int pf_cpd_create(pf_ctx* ctx, const double* X, int64_t N, const double* Y, int64_t M, int32_t d, pf_cpd** out);
void pf_cpd_free(pf_cpd* h);
int pf_cpd_estep(pf_cpd* h, const double* TY, double sigma2, double w, double* P1, double* Pt1, double* PX);
int pf_cpd_set_basis(pf_cpd* h, const double* Q, int32_t K);
int pf_cpd_weighted_gram(pf_cpd* h, double* H);
/* Device-resident EM iterations: after pf_cpd_estep(h, NULL, sigma2, w, NULL, NULL, NULL) the posterior sums stay
 * on the device; only the small moment sums an M-step needs come back, and only the new parameters go in.
 *   pf_cpd_affine_sums   shifts [32] = cx[16] | cy[16] (means of X and Y, the centre the sums are taken about);
 *                        sums [2 d^2 + 3 d + 3], with xc = x - cx, yc = y - cy, PXc_m = PX_m - P1_m cx:
 *                        sum P1 | sum PXc [d] | sum P1 yc [d] | sum PXc yc^T [d][d] | sum P1 yc yc^T [d][d] |
 *                        sum Pt1 | sum Pt1 |xc|^2 | sum Pt1 xc [d]
 *   pf_cpd_apply_affine  TY = Y B + t
 *   pf_cpd_deform_sums   H [K][K] = Q^T diag(P1) Q,  R [K][d] = Q^T (PX - diag(P1) Y)      (needs pf_cpd_set_basis)
 *   pf_cpd_apply_deform  TY = Y + Q C (C [K][d]), then sums [5] = sum P1 | sum P1 |ty|^2 | sum ty.PX | sum Pt1 |
 *                        sum Pt1 |x|^2 with the new TY
 *   pf_cpd_download      current TY [M][d] and the last posterior sums (each may be NULL) */
int pf_cpd_affine_sums(pf_cpd* h, double* shifts, double* sums);
int pf_cpd_apply_affine(pf_cpd* h, const double* B, const double* t);
int pf_cpd_deform_sums(pf_cpd* h, double* H, double* R);
int pf_cpd_apply_deform(pf_cpd* h, const double* C, double* sums);
int pf_cpd_download(pf_cpd* h, double* TY, double* P1, double* Pt1, double* PX);
int pf_cpd_gram(pf_ctx* ctx, const double* A, int64_t n_a, const double* B, int64_t n_b, int32_t d, double beta,
                const double* V, int32_t n_cols, double* out);

#ifdef __cplusplus
}
#endif
#endif /* PYFOCUSR_HIP_H */
