// pf_eigs_smallest / pf_eigs_smallest2: the whole eigensolve of one mesh graph, or of the two graphs of a pair, behind ONE
// C call.
//
// Replaces scipy.sparse.linalg.eigs(L, k, sigma=1e-10, which="LM", ncv=4k) of the reference (graph.py:372; called once
// per mesh by Graph.get_graph_spectrum, graph.py:243-248).  The driver itself - Chebyshev-filtered thick-restart Lanczos
// for symmetric W, restarted Arnoldi with carried complex outliers or the ellipse filter for asymmetric W (one-way
// edges: both bundled 15k meshes) - is pf_krylov.h, a HIP-free header that the CPU tests run on a test double; this file
// binds it to the device primitives of the C-ABI (pf_cheb / pf_cheb2, pf_orth_begin / pf_orth_begin2 / pf_orth_cheb2,
// pf_combine, pf_spmv_multi, pf_gram, pf_resnorms ...) and adds what surrounds a solve: the repeat after a resident-kernel
// timeout, the eigenvector finalisation and downloads.  Graphs the filtered iteration does not cover (fewer than ~100
// vertices, wanted eigenvalues that are no corner of the spectrum) are refused with PF_E_STATE: pyfocusr_amd/_krylov.py
// has the unfiltered mode for them.
#include <math.h>

#include <algorithm>
#include <vector>

#include "pf_internal.h"
#include "pf_krylov.h"

namespace {

// pfk::Ops over one device graph
struct DeviceOps final : pfk::Ops {
    pf_graph* g;
    explicit DeviceOps(pf_graph* graph) : g(graph) {}
    int64_t n() const override { return g->n; }
    int64_t n_isolated() const override { return g->n_isolated; }
    int32_t n_components() const override { return g->n_components; }
    int32_t n_oneway() const override { return g->n_oneway; }
    bool symmetric() const override { return g->is_symmetric; }
    double spectral_bound() const override { return g->spectral_bound; }
    int ws_ensure(int32_t slots) override { return pf_ws_ensure(g, slots); }
    int lock_nulls(int32_t op, int32_t* locked) override { return pf_lock_null_vectors(g, op, locked); }
    int orth_strict(bool on) override { return pf_orth_strict(g, on ? 1 : 0); }
    int orth_always_twice(bool on, bool strict_otherwise) override { return pf_orth_strict(g, on ? 2 : (strict_otherwise ? 1 : 0)); }
    int orth_device_passes(bool on) override { return pf_orth_device_passes(g, on ? 1 : 0); }
    int start_vector(int32_t slot, uint64_t seed) override { return pf_start_vector(g, slot, seed); }
    int orth_begin(int32_t w, int32_t first, int32_t count) override { return pf_orth_begin(g, w, first, count, 1); }
    int orth_end(double* h, double* nrm, bool* redone) override {
        PF_TRY(pf_orth_end(g, h, nrm));
        const int r = pf_orth_redone(g);
        *redone = r == 1;
        last_twice = r == 2;  // (both passes ran on the device: nothing queued behind read a stale vector)
        return PF_OK;
    }
    int cheb(int32_t op, int32_t src, int32_t dst, int32_t degree, double c, double e, double rho) override {
        return pf_cheb(g, op, src, dst, degree, c, e, rho);
    }
    int combine(int32_t src_first, int32_t m, const double* Y, int32_t k, int32_t dst_first) override {
        return pf_combine(g, src_first, m, Y, k, dst_first);
    }
    int combine2(int32_t src_first, int32_t m, const double* Y, int32_t k, int32_t dst_first, int32_t src_first2, int32_t dst_first2) override {
        return pf_combine2(g, src_first, m, Y, k, dst_first, src_first2, dst_first2);
    }
    int copy(int32_t src, int32_t dst, int32_t count) override { return pf_ws_copy(g, src, dst, count); }
    int spmv_multi(int32_t op, int32_t src_first, int32_t dst_first, int32_t count) override {
        return pf_spmv_multi(g, op, src_first, dst_first, count);
    }
    int gram(int32_t first_a, int32_t count_a, int32_t first_b, int32_t count_b, double* out) override {
        return pf_gram(g, first_a, count_a, first_b, count_b, out);
    }
    int resnorms(int32_t ax_first, int32_t x_first, const double* lam, int32_t count, double* out) override {
        return pf_resnorms(g, ax_first, x_first, lam, count, out);
    }
    int gram_begin(int32_t first_a, int32_t count_a, int32_t first_b, int32_t count_b, bool self_b) override {
        PF_TRY(pf_gram_begin(g, first_a, count_a, first_b, count_b, 0));
        return self_b ? pf_gram_begin(g, first_b, count_b, first_b, count_b, 1) : PF_OK;
    }
    bool orth_split(int32_t first2, int32_t split) override { return pf_orth_split(g, first2, split) == PF_OK; }
    int gram_end(double* out) override { return pf_small_end(g, out); }
    int resnorms_begin(int32_t ax_first, int32_t x_first, const double* lam, int32_t count) override {
        return pf_resnorms_begin(g, ax_first, x_first, lam, count);
    }
    int resnorms_end(double* out) override { return pf_small_end(g, out); }
    // the two graphs of a pair in shared launches (the partner is a DeviceOps of the same ctx: pf_eigs_smallest2 checks)
    int orth_begin_pair(pfk::Ops& other, const int32_t* o) override {
        return pf_orth_begin2(g, o[0], o[1], o[2], o[3], static_cast<DeviceOps&>(other).g, o[4], o[5], o[6], o[7]);
    }
    int cheb_pair(pfk::Ops& other, const int32_t* ci, const double* cd) override {
        return pf_cheb2(g, ci[0], ci[1], ci[2], ci[3], cd[0], cd[1], cd[2], static_cast<DeviceOps&>(other).g, ci[4], ci[5], ci[6], ci[7], cd[3],
                        cd[4], cd[5]);
    }
    int orth_cheb_pair(pfk::Ops& other, const int32_t* orth, const int32_t* ci, const double* cd) override {
        return pf_orth_cheb2(g, static_cast<DeviceOps&>(other).g, orth, ci, cd);
    }
};

int ellipse_hint() {  // PF_EIGS_ELLIPSE: -1 (default) by the graph's one-way edges, 0 interval filter first, 1 ellipse at once
    const char* ev = getenv("PF_EIGS_ELLIPSE");
    return ev ? atoi(ev) : -1;
}

// a solve was cut short: collect the Gram-Schmidt step it may have left in flight, and cancel the eigenvector download
// that a partner's finish() may already owe to the caller's buffer (the caller is about to be told that the call failed:
// its buffers may be gone before a held-back download would be released)
void give_up(pf_graph* g) {
    if (!g) return;
    if (g->orth_pending >= 0) {
        std::vector<double> h((size_t)g->orth_pending + 1);
        double nrm = 0.0;
        (void)pf_orth_end(g, h.data(), &nrm);
    }
    if (g->small_pending > 0) {  // an extraction was cut short between its halves
        (void)hipEventSynchronize(g->small_ev);
        g->small_pending = 0;
    }
    (void)pf_download_cancel(g);
}

int finish(pf_graph* g, pfk::Solver& s, int32_t minmax, double* vals, double* vecs, double* residuals, int32_t* n_out, pf_eigs_stats* stats,
           bool async) {
    *n_out = s.n_out;
    for (int i = 0; i < s.n_out; ++i) {
        vals[i] = s.vals[(size_t)i];
        if (residuals) residuals[i] = s.residuals[(size_t)i];
    }
    if (stats) *stats = s.st;
    if (s.n_out > 0) {
        // (eigenvectors of L: x_L = G^1/2 x_S when the symmetrised operator was iterated)
        PF_TRY(pf_finalize_vectors_begin(g, s.first_slot, s.n_out, s.sym ? 1 : 0, minmax ? 1 : 0, vecs));
        if (!async) PF_TRY(pf_finalize_vectors_end(g));
    }
    return PF_OK;
}

}  // namespace

// A wait of the resident filter kernel that ran out (PF_E_PERSIST_TIMEOUT: the stream is drained and the path switched
// off by then) invalidates the filter applications in flight; the solve is simply repeated, one step per launch.
extern "C" int pf_eigs_smallest_ex(pf_graph* g, int32_t n_wanted, int32_t minmax, int32_t async_download, double* vals, double* vecs,
                                   double* residuals, int32_t* n_out, pf_eigs_stats* stats_out) {
    PF_CHECK(g && vals && vecs && n_out && n_wanted >= 1, PF_E_ARG, "pf_eigs_smallest: bad argument");
    int rc = PF_OK;
    for (int attempt = 0; attempt < 3; ++attempt) {
        DeviceOps ops(g);
        pfk::Solver s;
        *n_out = 0;
        rc = s.init(&ops, n_wanted, ellipse_hint());
        if (rc == PF_OK) rc = pfk::drive_single(s);
        if (rc == PF_OK) rc = pf_sync(g->ctx);  // nothing of this solve is left in flight (and a late PF_E_PERSIST_TIMEOUT surfaces here)
        if (rc == PF_OK) rc = finish(g, s, minmax, vals, vecs, residuals, n_out, stats_out, async_download != 0);
        if (rc != PF_OK) give_up(g);
        if (rc != PF_E_PERSIST_TIMEOUT) break;
    }
    return rc;
}

extern "C" int pf_eigs_smallest(pf_graph* g, int32_t n_wanted, int32_t minmax, double* vals, double* vecs, int32_t* n_out,
                                pf_eigs_stats* stats_out) {
    return pf_eigs_smallest_ex(g, n_wanted, minmax, 0, vals, vecs, nullptr, n_out, stats_out);
}

extern "C" int pf_eigs_smallest2(pf_graph* ga, pf_graph* gb, int32_t n_wanted_a, int32_t n_wanted_b, int32_t minmax, int32_t async_download,
                                 double* vals_a, double* vecs_a, double* res_a, int32_t* n_out_a, pf_eigs_stats* stats_a,
                                 double* vals_b, double* vecs_b, double* res_b, int32_t* n_out_b, pf_eigs_stats* stats_b) {
    PF_CHECK(ga && gb && ga != gb && vals_a && vecs_a && n_out_a && vals_b && vecs_b && n_out_b && n_wanted_a >= 1 && n_wanted_b >= 1,
             PF_E_ARG, "pf_eigs_smallest2: bad argument");
    PF_CHECK(ga->ctx == gb->ctx, PF_E_ARG, "pf_eigs_smallest2: the two graphs must share one ctx (stream)");
    int rc = PF_OK;
    for (int attempt = 0; attempt < 3; ++attempt) {
        DeviceOps oa(ga), ob(gb);
        pfk::Solver a, b;
        *n_out_a = *n_out_b = 0;
        rc = a.init(&oa, n_wanted_a, ellipse_hint());
        if (rc == PF_OK) rc = b.init(&ob, n_wanted_b, ellipse_hint());
        if (rc == PF_OK) rc = pfk::drive_pair(a, b);
        // (both solves have synchronised with the stream in their extraction; a late PF_E_PERSIST_TIMEOUT of the partner's
        // last applications surfaces in pf_sync)
        if (rc == PF_OK) rc = pf_sync(ga->ctx);
        if (rc == PF_OK) rc = finish(ga, a, minmax, vals_a, vecs_a, res_a, n_out_a, stats_a, async_download != 0);
        if (rc == PF_OK) rc = finish(gb, b, minmax, vals_b, vecs_b, res_b, n_out_b, stats_b, async_download != 0);
        if (rc != PF_OK) {
            give_up(ga);
            give_up(gb);
        }
        if (rc != PF_E_PERSIST_TIMEOUT) break;
    }
    return rc;
}
