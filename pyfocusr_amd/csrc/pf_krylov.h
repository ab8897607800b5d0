// The Krylov driver of the eigensolve: one graph's solve as a resumable state machine, and the lockstep driver of a pair.
//
// Replaces scipy.sparse.linalg.eigs(L, k, sigma=1e-10, which="LM", ncv=4k) of the reference (graph.py:372; called once
// per mesh by Graph.get_graph_spectrum, graph.py:243-248): the Chebyshev-filtered Krylov-Schur iteration of
// pyfocusr_amd/_krylov.py restated in C++ -
//   * W symmetric: operator S = G^1/2 (D - W) G^1/2, thick-restart Lanczos (symmetric Ritz problem: pfd::eigh_sym);
//   * W asymmetric (one-way edges, graph.py:178 - both bundled 15k meshes, every scanned surface with a hole): operator
//     L = G (D - W) itself, restarted Arnoldi on the ordered real Schur form (pfd::real_schur / schur_reorder).  L is then
//     non-normal with complex eigenvalues off the axis that a real Chebyshev polynomial amplifies more than the wanted low
//     ones: with the INTERVAL filter they are carried as dominant Ritz values (largest modulus) next to the wanted ones
//     and dropped after the final Rayleigh-Ritz step on L; when the low eigenvalues are complex themselves (open
//     surfaces) or the outliers are too many, the ELLIPSE filter takes over (foci pulled inwards, recurrence scaled per
//     step), its height grown until it encloses the spectrum;
//   B = T_p((c - A)/e) damps [cut, hi]; a Gram-Schmidt step against the whole basis on the device (second pass on demand),
//   analytic null vectors locked per connected component, isolated vertices masked, final Rayleigh-Ritz on A itself,
//   eigenvalues > 1e-10 kept (graph.py:381).
// Pipelined: the Gram-Schmidt step of an outer step and the NEXT filter application are queued together before the
// step's coefficients are read, the Ritz check runs one step behind the device and is skipped on steps that the
// geometric decay of the residuals predicts to be far from convergence; for the asymmetric case the check costs
// O(j^3 / 4) + O(q j^2) as long as the projected matrix is upper Hessenberg (eigenvalues without vectors, residuals
// by pfd::hessenberg_residual_factor) and the full ordered Schur form only when that estimate says "converged".
//
// The device work goes through the `Ops` interface: pf_eigs.hip implements it with the C-ABI primitives (pf_cheb,
// pf_orth_begin, ...); tests/csrc/krylov_double.cpp is a CPU TEST DOUBLE of it so that this logic runs in `-m "not gpu"`
// tests.  No HIP in this header.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include <algorithm>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/pyfocusr_hip.h"
#include "pf_dense.h"

void pf_set_error(const char* fmt, ...);

#define PFK_CHECK(cond, code, ...)       \
    do {                                 \
        if (!(cond)) {                   \
            pf_set_error(__VA_ARGS__);   \
            return (code);               \
        }                                \
    } while (0)
#define PFK_TRY(call)                \
    do {                             \
        int r_ = (call);             \
        if (r_ != PF_OK) return r_;  \
    } while (0)

namespace pfk {

const int NEED_ELLIPSE = 1000;  // internal status: the interval filter cannot handle this non-normal operator

// ---- what a solver needs from the device, per graph.  Slots are workspace vectors of the graph.
struct Ops {
    virtual ~Ops() {}
    bool last_twice = false;  // the step orth_end just collected took its second pass on the device (nothing to repeat)
    virtual int64_t n() const = 0;
    virtual int64_t n_isolated() const = 0;
    virtual int32_t n_components() const = 0;  // components of >= 2 vertices: one analytic null vector each
    virtual int32_t n_oneway() const = 0;      // directed edges without their reverse
    virtual bool symmetric() const = 0;
    virtual double spectral_bound() const = 0;  // proven upper bound of the spectrum of S (symmetric graphs); 2 otherwise
    virtual int ws_ensure(int32_t slots) = 0;
    virtual int lock_nulls(int32_t op, int32_t* locked) = 0;  // unit null vectors of `op` into slots [0, n_components)
    virtual int orth_strict(bool on) = 0;
    virtual int orth_device_passes(bool on) { return on ? PF_OK : PF_OK; }  // second Gram-Schmidt pass queued with the first (pf_orth_device_passes)
    virtual int start_vector(int32_t slot, uint64_t seed) = 0;
    virtual int orth_begin(int32_t w, int32_t first, int32_t count) = 0;  // Gram-Schmidt + normalisation, asynchronous
    virtual int orth_end(double* h, double* nrm, bool* redone) = 0;
    virtual int cheb(int32_t op, int32_t src, int32_t dst, int32_t degree, double c, double e, double rho) = 0;
    virtual int combine(int32_t src_first, int32_t m, const double* Y, int32_t k, int32_t dst_first) = 0;
    // dst = src Y and dst2 = src2 Y (the same rotation of two blocks); default: two calls
    virtual int combine2(int32_t src_first, int32_t m, const double* Y, int32_t k, int32_t dst_first, int32_t src_first2, int32_t dst_first2) {
        const int rc = combine(src_first, m, Y, k, dst_first);
        return rc ? rc : combine(src_first2, m, Y, k, dst_first2);
    }
    virtual int copy(int32_t src, int32_t dst, int32_t count) = 0;
    virtual int spmv_multi(int32_t op, int32_t src_first, int32_t dst_first, int32_t count) = 0;
    virtual int gram(int32_t first_a, int32_t count_a, int32_t first_b, int32_t count_b, double* out) = 0;
    virtual int resnorms(int32_t ax_first, int32_t x_first, const double* lam, int32_t count, double* out) = 0;
    // The same in two halves - queue, collect - so that the extractions of the two graphs of a pair overlap (one graph's
    // host algebra and waits beside the other's launches).  Defaults: the synchronous call at the first half.
    // (self_b: the Gram matrix of the b vectors themselves, count_b x count_b, is appended to the result)
    virtual int gram_begin(int32_t first_a, int32_t count_a, int32_t first_b, int32_t count_b, bool self_b = false) {
        held.assign((size_t)count_a * count_b + (self_b ? (size_t)count_b * count_b : 0), 0.0);
        const int rc = gram(first_a, count_a, first_b, count_b, held.data());
        if (rc || !self_b) return rc;
        return gram(first_b, count_b, first_b, count_b, held.data() + (size_t)count_a * count_b);
    }
    virtual int gram_end(double* out) {
        std::copy(held.begin(), held.end(), out);
        return 0;
    }
    virtual int resnorms_begin(int32_t ax_first, int32_t x_first, const double* lam, int32_t count) {
        held.assign((size_t)count, 0.0);
        return resnorms(ax_first, x_first, lam, count, held.data());
    }
    virtual int resnorms_end(double* out) {
        std::copy(held.begin(), held.end(), out);
        return 0;
    }
    std::vector<double> held;
    // on: every Gram-Schmidt step from now on takes its second pass; off: back to the criterion orth_strict(strict_otherwise)
    // names.  (A test double whose steps always take two passes has nothing to do.)
    virtual int orth_always_twice(bool on, bool strict_otherwise) {
        (void)on, (void)strict_otherwise;
        return 0;
    }
    // The NEXT orth_begin (alone or in a pair's shared launch) takes its basis from the slots [first, first + split) and
    // [first2, first2 + count - split) instead of [first, first + count).  false: not supported (every step is a full one).
    virtual bool orth_split(int32_t first2, int32_t split) {
        (void)first2, (void)split;
        return false;
    }
    // launches that two graphs of a pair share; the defaults run them one after the other
    virtual int orth_begin_pair(Ops& other, const int32_t* orth /*[8]: w, first, count, normalize per graph*/) {
        PFK_TRY(orth_begin(orth[0], orth[1], orth[2]));
        return other.orth_begin(orth[4], orth[5], orth[6]);
    }
    virtual int cheb_pair(Ops& other, const int32_t* ci /*[8]: op, src, dst, degree per graph*/, const double* cd /*[6]: c, e, rho*/) {
        PFK_TRY(cheb(ci[0], ci[1], ci[2], ci[3], cd[0], cd[1], cd[2]));
        return other.cheb(ci[4], ci[5], ci[6], ci[7], cd[3], cd[4], cd[5]);
    }
    virtual int orth_cheb_pair(Ops& other, const int32_t* orth, const int32_t* ci, const double* cd) {
        PFK_TRY(orth_begin_pair(other, orth));
        return cheb_pair(other, ci, cd);
    }
};

inline double cheb_value(double lam, double c, double e, int p) {
    const double t = (c - lam) / e;
    if (fabs(t) <= 1.0) return cos(p * acos(t));
    const double s = (t > 0 || p % 2 == 0) ? 1.0 : -1.0;
    return s * cosh(p * acosh(fabs(t)));
}

// T_p((c - lam)/e) / rho^p without overflow
inline double cheb_value_scaled(double lam, double c, double e, int p, double rho) {
    if (rho == 1.0) return cheb_value(lam, c, e, p);
    const double t = (c - lam) / e;
    if (fabs(t) <= 1.0) return cos(p * acos(t)) * exp(-p * log(rho));
    const double u = acosh(fabs(t));
    const double s = (t > 0 || p % 2 == 0) ? 1.0 : -1.0;
    return s * 0.5 * exp(p * (u - log(rho))) * (1.0 + exp(-2.0 * p * u));
}

inline double cheb_inverse(double theta, double c, double e, int p) { return c - e * cosh(acosh(std::max(theta, 1.0)) / p); }

// damped interval [cut, hi]; degree such that eigenvalues <= cut/2 are amplified by >= cosh(strength) relative to it
inline void choose_filter(double cut, double hi, double strength, int max_degree, double* c, double* e, int* p) {
    cut = std::min(std::max(cut, 1e-12), 0.5 * hi);
    *c = 0.5 * (hi + cut);
    *e = 0.5 * (hi - cut);
    const double growth = acosh((*c - 0.5 * cut) / *e);
    *p = std::max(8, std::min(max_degree, (int)ceil(strength / growth)));
}

// The damped set is the ellipse with vertices cut and hi on the real axis and semi-minor axis half_height: the same
// recurrence with the foci c +- e pulled inwards, scaled by rho = (a + b)/e per step (|T_p / rho^p| <= ~0.5 inside).
inline void choose_filter_ellipse(double cut, double half_height, double hi, double strength, int max_degree, double* c, double* e,
                                  int* p, double* rho) {
    cut = std::min(std::max(cut, 1e-12), 0.5 * hi);
    *c = 0.5 * (hi + cut);
    const double a = 0.5 * (hi - cut);
    const double b = std::min(half_height, 0.9 * a);
    *e = sqrt(a * a - b * b);
    *rho = (a + b) / *e;
    const double t = (*c - 0.5 * cut) / *e;
    const double growth = log(t + sqrt(t * t - 1.0)) - log(*rho);
    *p = std::max(8, std::min(max_degree, (int)ceil(strength / std::max(growth, 1e-12))));
}

// How many of the Ritz values `ev` (any order) make up the wanted dominant set: the largest-modulus ones up to and
// including `n_real` images of low eigenvalues (real, positive - or any, in ellipse mode) plus n_extra more, never
// splitting a conjugate pair.  order: indices by decreasing modulus (stable); thr: modulus between kept and dropped
// (-1: everything kept).
struct Dominant {
    int q = 0, n_found = 0;
    double thr = -1.0;
    std::vector<int> order;
};

inline Dominant select_dominant(const std::vector<pfd::cplx>& ev, int n_real, int n_extra, bool count_all) {
    Dominant d;
    const int m = (int)ev.size();
    d.order.resize(m);
    for (int i = 0; i < m; ++i) d.order[i] = i;
    std::stable_sort(d.order.begin(), d.order.end(), [&](int a, int b) { return std::abs(ev[a]) > std::abs(ev[b]); });
    std::vector<char> is_real(m);
    std::vector<int> cnt(m);
    int run = 0;
    for (int i = 0; i < m; ++i) {
        const pfd::cplx z = ev[d.order[i]];
        is_real[i] = count_all || (fabs(z.imag()) <= 1e-9 * std::abs(z) && z.real() > 0);
        run += is_real[i] ? 1 : 0;
        cnt[i] = run;
    }
    int q = m;
    for (int i = 0; i < m; ++i)
        if (cnt[i] >= n_real) {
            q = i + 1;
            break;
        }
    int extra = 0;
    while (q < m && extra < n_extra) {
        const pfd::cplx z = ev[d.order[q]];
        const int step = (is_real[q] || fabs(z.imag()) <= 1e-9 * std::abs(z)) ? 1 : 2;
        if (q + step > m) break;
        q += step;
        extra += step;
    }
    if (q < m) {
        double lo = std::abs(ev[d.order[q - 1]]), nx = std::abs(ev[d.order[q]]);
        d.thr = 0.5 * (lo + nx);
        if (lo == nx) {  // a conjugate pair straddling the cut
            q += 1;
            d.thr = q < m ? 0.5 * (std::abs(ev[d.order[q - 1]]) + std::abs(ev[d.order[q]])) : -1.0;
        }
    }
    d.q = q;
    d.n_found = m ? std::min(cnt[m - 1], n_real) : 0;
    return d;
}

enum ReqKind { REQ_NONE = 0, REQ_CHEB, REQ_ORTH, REQ_ORTH_CHEB };

struct Request {
    ReqKind kind = REQ_NONE;
    int32_t cheb_src = 0, cheb_dst = 0;                  // filter application: slot src -> slot dst
    int32_t orth_w = 0, orth_first = 0, orth_count = 0;  // Gram-Schmidt step of slot w against slots [first, first + count), normalised
};

// ---- one graph's solve: advance() runs the host side up to the next device request that a driver may want to share
// with a partner graph (a filter application, a Gram-Schmidt step, or both) - the yield points of _krylov._solve_gen
struct Solver {
    Ops* ops = nullptr;
    int32_t n_wanted = 0;
    // results
    std::vector<double> vals, residuals;
    int32_t n_out = 0, first_slot = 0;
    pf_eigs_stats st{};
    bool done = false;
    Request req;
    bool req_launched = false;  // the pair driver has queued `req` already (ahead of its partner's extraction)
    // configuration
    bool sym = true;
    int32_t op = PF_OP_SYM;
    double hi = 2.0, strength = 1.8, tol = 1e-12;
    int64_t n_active = 0;
    int c0 = 0, want = 0, q_target = 0, m_max = 0, reg = 0, A0 = 0, B0 = 0;
    int out_cap = 16;  // complex outliers the interval filter may carry next to the wanted Ritz values
    int m_max_limit = 0;  // > 0: a smaller basis than the default (tests: forces thick restarts)
    // filter
    bool ellipse = false, interval_tried = false;
    double half_height = 0.0;
    int ellipse_tries = 0, degree_cap = 4000;
    double cut = 0.0, cut0 = 0.0, c = 0.0, e = 0.0, rho = 1.0, theta0 = 0.0, band = 1.5;
    int p = 0;
    uint64_t seed = 0;
    // Krylov state  B V_j = V_j H + v_j b^T ; null vectors are locked exact Ritz pairs
    std::vector<double> H, b, h;
    int j = 0, q = 0, restarts = 0, resets = 0;
    bool hess = true;  // H[c0:j, c0:j] is upper Hessenberg and b = beta e_j (no thick restart since the filter was placed)
    bool spec = false, near_conv = false, have_seen = false, start_pending = false;
    bool expect_final = false;  // (asymmetric graphs) the trend of the residuals says the check that comes next is the last one
    int next_check = 0, seen_j = 0;
    double seen_worst = 0.0;
    // the last Ritz analysis
    std::vector<double> theta_re, theta_im;  // Ritz values, dominant first
    std::vector<double> U, T;               // j x j: Ritz vectors (symmetric) / ordered Schur vectors and form
    double theta_min = 0.0, theta_absmax = 0.0;
    int n_real = 0;
    std::vector<double> res;
    // the Ritz analysis of a step is pure host work on this solver's own members: a pair driver runs the two graphs'
    // analyses side by side (analysis_due is raised, advance() returns, analyse() may be called from any ONE thread,
    // then advance() continues); a single driver lets advance() run it in line
    bool analysis_due = false, inline_analysis = true, exhausted = false;
    int outcome = 0, analysis_rc = PF_OK;  // outcome: 1 converged, 2 the cut has to move, 3 complex outliers eat the dynamic range
    enum { S_TOP, S_AFTER_CHEB, S_AFTER_ORTH, S_AFTER_ANALYSIS, S_EXTRACT_B, S_EXTRACT_C, S_DONE } state = S_TOP;
    // the extraction in three phases (queue Z, A Z and their Gram matrix | collect it, rotate, queue X, A X and the residual
    // norms | collect them); stepwise_extract: advance() returns between the phases with no request (drive_pair alternates
    // the two graphs' phases), `extracting` says so
    bool stepwise_extract = false, extracting = false;
    // Lanczos with PARTIAL reorthogonalisation (symmetric graphs; H. Simon 1984): a step orthogonalises against the locked
    // null vectors and the last two basis vectors only - the three-term recurrence - while a recurrence over the
    // coefficients estimates w_{j+1,k} = <v_{j+1}, v_k>; when an estimate passes pro_thresh the next TWO steps are full
    // Gram-Schmidt steps and the estimates start again from the unit roundoff.  The basis stays semi-orthogonal (1e-9),
    // the tridiagonal matrix is the projection of the operator to working precision, and the extraction orthonormalises
    // the few Ritz vectors it keeps (generalised Rayleigh-Ritz).  250k blobs: 4 full steps of ~36; a full step reads
    // the whole basis twice (avg. 18 vectors x 2 MB), a local one three vectors.
    bool pro = false, pro_local = false;   // enabled for this solve; the step in flight is a local one
    int pro_force = 0;                     // full steps still owed
    double pro_thresh = 1e-9;
    std::vector<double> om_prev, om_cur;   // estimates for v_{j-1}, v_j against v_k (basis index k)
    std::vector<double> ex_Rk;
    int ex_nk = 0;

    // ellipse_hint: -1 by the number of one-way edges, 0 interval filter first, 1 ellipse filter at once
    int init(Ops* o, int32_t wanted, int ellipse_hint = -1) {
        ops = o;
        n_wanted = wanted;
        PFK_CHECK(ops && n_wanted >= 1, PF_E_ARG, "pf_eigs_smallest: bad argument");
        sym = ops->symmetric();
        op = sym ? PF_OP_SYM : PF_OP_RW;
        hi = sym ? ops->spectral_bound() : 2.0;  // (2, or the face-by-face bound of a closed triangle mesh)
        strength = 1.8;  // (symmetric graphs: placement and strength as _krylov._solve_gen; asymmetric ones: below)
        n_active = ops->n() - ops->n_isolated();
        vals.assign((size_t)n_wanted, 0.0);
        residuals.assign((size_t)n_wanted, 0.0);
        // The loose single-pass criterion of the device's Gram-Schmidt step is for the filtered iteration of large graphs; on
        // small ones it can lose orthogonality.  The setting is sticky per graph: set here either way.
        PFK_TRY(ops->orth_strict(n_active < 4096));
        // Arnoldi with strongly amplified outliers in the basis cancels digits in every sixth step or so (15k bundled meshes;
        // nearly every step at higher degree): the second pass rides with the first instead of costing a repeated application.
        // Lanczos: only the first two steps of a solve (see step(): the first product with the start vector always cancels).
        PFK_TRY(ops->orth_device_passes(true));
        pro = sym && (!getenv("PF_EIGS_PRO") || atoi(getenv("PF_EIGS_PRO")) != 0) && n_active >= 4096;
        c0 = ops->n_components();
        want = (int)std::min<int64_t>(n_wanted, std::max<int64_t>(n_active - c0, 0));
        st.mode = sym ? 0 : 1;
        if (want <= 0) {
            done = true;
            state = S_DONE;
            return PF_OK;
        }
        q_target = c0 + want;
        out_cap = sym ? 0 : std::min(std::max(16, (int)ops->n_oneway()), 64);
        PFK_CHECK(n_active >= 4 * (q_target + 8 + out_cap), PF_E_STATE,
                  "pf_eigs_smallest: graph too small for the filtered iteration (%lld active vertices): use the Python driver",
                  (long long)n_active);
        // (asymmetric graphs: room for the carried outliers - the solve should end before the basis is full: a restart means
        // an ordered Schur form of the full matrix, and full Schur forms at every check from then on)
        m_max = (int)std::min<int64_t>(std::max(3 * q_target + 24, 48) + out_cap, n_active);
        if (m_max_limit > 0) m_max = std::min(m_max, std::max(m_max_limit, q_target + 4));
        reg = std::max(m_max + 1, 2 * (q_target + out_cap) + 2);
        PFK_TRY(ops->ws_ensure(2 * reg));
        A0 = 0, B0 = reg;
        // Asymmetric graphs: placement 12 (want + 1) / n and strength 2.0 where that gives a degree of 64 or more (the messy 250k
        // pair: 15.4 ms; 16.0-16.3 with stronger filters, which only run into the degree cap).  Small graphs gain from
        // fewer, longer applications - an outer step costs ~46 us besides its recurrence (launch preamble, Gram-Schmidt with
        // its device-side second pass) and the step count has a floor, the carried outliers: bundled 15k pair 6.85 ms at
        // (12, 2.0) with degrees 29 / 23, 6.1-6.2 anywhere in (6..8, 2.5..3.0) with degrees ~52 / 40 - so placement and
        // strength move towards (7, 2.75) as the degree at (12, 2.0) falls from 64 to 32.
        double cut_factor = sym ? 8.0 : 12.0;
        if (!sym) {
            strength = 2.0;
            double c_, e_;
            int p0 = 0;
            choose_filter(cut_factor * (want + 1) / (double)std::max<int64_t>(n_active, 1), hi, strength, 128, &c_, &e_, &p0);
            const double f = std::min(std::max((64.0 - p0) / 32.0, 0.0), 1.0);
            cut_factor -= 5.0 * f;
            strength += 0.75 * f;
        }
        // (a stronger filter for the graph of a pair that wants more columns - 9 against 5 for the bundled 15k pair - evens the
        // two step counts out, 44 / 48 instead of 48 / 53 on the messy 250k pair, at more recurrence steps per shared launch:
        // measured neutral, 12.94 against 12.95 ms and 5.11 against 5.15; not kept)
        cut0 = cut_factor * (want + 1) / (double)std::max<int64_t>(n_active, 1);
        if (const char* ev = getenv("PF_EIGS_CUT")) cut0 *= atof(ev) / cut_factor;  // (experiments: the filter's placement ...
        if (const char* ev = getenv("PF_EIGS_STRENGTH")) strength = atof(ev);  // ... and strength; results agree to tol)
        H.assign((size_t)m_max * m_max, 0.0);
        b.assign((size_t)m_max, 0.0);
        h.assign((size_t)m_max + 1, 0.0);
        // Which filter first.  Scattered defects and holes that are small against the surface (up to ~100 one-way edges in
        // the experiments behind DESIGN.md 4) leave a few dozen complex outliers that the interval filter carries at a fifth
        // of the ellipse filter's operator applications; a long open boundary (hundreds of one-way edges) leaves more
        // outliers than the basis has room for, and the attempt would only be noticed to fail ~70 steps in.
        if (!sym && (ellipse_hint == 1 || (ellipse_hint < 0 && ops->n_oneway() > 128))) return begin_mode(true);
        return begin_mode(false);
    }

    // The estimates of <v_{j+1}, v_k>, k = c0 .. j, from those of v_j and v_{j-1} (Simon's recurrence; alpha_k, beta_k read
    // from H; beta_next = the norm the step that produced v_{j+1} found).  Called with j still the index of the vector the
    // filter was applied to.  A step that was full resets its vector's estimates to the unit roundoff.
    void pro_update(double beta_next) {
        const double eps = 1.1e-16;
        const size_t ld = (size_t)m_max;
        auto alpha = [&](int k) { return H[(size_t)k * ld + k]; };
        auto beta_of = [&](int k) { return k > c0 ? H[(size_t)k * ld + k - 1] : 0.0; };  // beta_k couples v_{k-1} and v_k
        std::vector<double> nw((size_t)m_max + 2, 0.0);
        double worst = 0.0;
        if (!pro_local) {
            for (int k = c0; k <= j; ++k) nw[(size_t)k] = eps;
        } else if (beta_next > 0.0) {
            const double aj = alpha(j), bj = beta_of(j);
            for (int k = c0; k < j; ++k) {
                double t = beta_of(k + 1) * om_cur[(size_t)k + 1] + (alpha(k) - aj) * om_cur[(size_t)k] - bj * om_prev[(size_t)k];
                if (k > c0) t += beta_of(k) * om_cur[(size_t)k - 1];
                t += (t < 0.0 ? -1.0 : 1.0) * 2.0 * eps * (fabs(beta_of(k + 1)) + fabs(beta_next));
                nw[(size_t)k] = t / beta_next;
                worst = std::max(worst, fabs(nw[(size_t)k]));
            }
            nw[(size_t)j] = eps * sqrt((double)std::max<int64_t>(n_active, 1)) * fabs(beta_of(c0 + 1)) / beta_next;
            worst = std::max(worst, fabs(nw[(size_t)j]));
        }
        nw[(size_t)j + 1] = 1.0;
        om_prev.swap(om_cur);
        om_cur.swap(nw);
        if (pro_local && worst > pro_thresh) pro_force = 2;
        if (getenv("PF_EIGS_DEBUG")) fprintf(stderr, "pro: step j=%d local=%d estimate %.2e alpha %.3e beta %.3e\n", j, (int)pro_local, worst, alpha(j), beta_next);
    }

    // (re)start the solve with the interval filter, or with the ellipse filter at its next height
    int begin_mode(bool want_ellipse) {
        if (!want_ellipse) {
            ellipse = false;
            interval_tried = true;
            degree_cap = sym ? 4000 : 128;
        } else {
            PFK_CHECK(ellipse_tries < 4, PF_E_DEGENERATE,
                      "filtered Krylov-Schur: could not enclose the complex spectrum of this non-normal Laplacian in an ellipse "
                      "(one-way edges: an open or non-manifold mesh); the eigenpairs were NOT computed");
            half_height = ellipse_tries == 0 ? 0.125 * hi : 1.6 * half_height;
            ++ellipse_tries;
            ellipse = true;
            degree_cap = 4000;
            st.mode = 2;
        }
        cut = cut0;
        resets = 0;
        return begin_filter();
    }

    int begin_filter() {
        PFK_CHECK(cut < 0.5 * hi, PF_E_STATE, "pf_eigs_smallest: the wanted eigenvalues are not a corner of the spectrum (cut %g): "
                  "use the Python driver", cut);
        rho = 1.0;
        if (ellipse) choose_filter_ellipse(cut, half_height, hi, strength, degree_cap, &c, &e, &p, &rho);
        else choose_filter(cut, hi, strength, degree_cap, &c, &e, &p);
        st.degree = p;
        st.cut = cut;
        theta0 = cheb_value_scaled(0.0, c, e, p, rho);
        band = 1.5 * 0.5 * (1.0 + pow(rho, -2.0 * p));  // wanted Ritz values must clear the damped set (1.5 for the interval)
        std::fill(H.begin(), H.end(), 0.0);
        std::fill(b.begin(), b.end(), 0.0);
        for (int i = 0; i < c0; ++i) H[(size_t)i * m_max + i] = theta0;
        j = c0;
        hess = true;
        pro_force = 0;
        pro_local = false;
        om_prev.assign((size_t)m_max + 2, 0.0);
        om_cur.assign((size_t)m_max + 2, 0.0);
        // the null vectors (an earlier attempt's restart or extraction reuses their slots), then the start vector,
        // orthogonal to them and normalised on the device; its coefficients are collected when the first filter
        // application has been queued behind it (no synchronisation at the head of the solve)
        int32_t locked = 0;
        PFK_TRY(ops->lock_nulls(op, &locked));  // slots [0, c0)
        PFK_CHECK(locked == c0, PF_E_STATE, "pf_eigs_smallest: %d null vectors locked, %d components", locked, c0);
        PFK_TRY(ops->start_vector(A0 + j, seed++));
        PFK_TRY(ops->orth_begin(A0 + j, A0, j));
        start_pending = true;
        restarts = 0;
        begin_expand();
        return PF_OK;
    }

    void begin_expand() {
        spec = false;
        near_conv = false;
        have_seen = false;
        expect_final = false;
        next_check = 0;
        state = S_TOP;
    }

    // ---- Ritz analysis of the leading j x j block of H.  full = false (asymmetric only): residual ESTIMATES, no vectors
    // (valid while `hess`); otherwise U (and T) are filled for a restart or the extraction.
    int ritz(int n_extra, bool full) {
        const int ld = m_max;
        struct Tm { int j; bool full; timespec t0; Tm(int j_, bool f) : j(j_), full(f) { clock_gettime(CLOCK_MONOTONIC, &t0); }
            ~Tm() { if (getenv("PF_EIGS_DEBUG")) { timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
                fprintf(stderr, "ritz j=%d full=%d %.0f us\n", j, (int)full, (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3); } } } tm_(j, full);
        if (sym) {
            std::vector<double> A((size_t)j * j), w;
            for (int a = 0; a < j; ++a)
                for (int bb = 0; bb < j; ++bb) A[(size_t)a * j + bb] = 0.5 * (H[(size_t)a * ld + bb] + H[(size_t)bb * ld + a]);
            pfd::eigh_sym(A, j, w);
            std::vector<int> order(j);
            for (int i = 0; i < j; ++i) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](int a, int bb) { return w[a] > w[bb]; });
            theta_re.resize(j);
            theta_im.assign(j, 0.0);
            U.assign((size_t)j * j, 0.0);
            for (int col = 0; col < j; ++col) {
                theta_re[col] = w[order[col]];
                for (int r = 0; r < j; ++r) U[(size_t)r * j + col] = A[(size_t)r * j + order[col]];
            }
            q = std::min(q_target + n_extra, j);
            n_real = std::min(q_target, j);
            res.assign(q, 0.0);
            theta_min = INFINITY;
            bool any_pos = false;
            const int lead = std::min(q_target, j);
            for (int col = 0; col < lead; ++col)
                if (theta_re[col] > 0.0) theta_min = std::min(theta_min, theta_re[col]), any_pos = true;
            if (!any_pos) theta_min = 0.0;
            for (int col = 0; col < q; ++col) {
                double r = 0.0;
                for (int i = 0; i < j; ++i) r += b[i] * U[(size_t)i * j + col];
                res[col] = fabs(r);
            }
            theta_absmax = j ? std::max(fabs(theta_re[0]), fabs(theta_re[j - 1])) : 0.0;
            return PF_OK;
        }
        std::vector<pfd::cplx> ev;
        if (!full && hess) {
            // eigenvalues only; H = [theta0 I, X; 0, H22] with H22 upper Hessenberg
            const int m2 = j - c0;
            std::vector<double> A((size_t)m2 * m2), wr, wi;
            for (int a = 0; a < m2; ++a)
                for (int bb = 0; bb < m2; ++bb) A[(size_t)a * m2 + bb] = H[(size_t)(c0 + a) * ld + c0 + bb];
            PFK_CHECK(pfd::hessenberg_schur(A, m2, nullptr, wr, wi), PF_E_DEGENERATE, "pf_eigs_smallest: QR iteration of the projected matrix failed");
            ev.resize(j);
            for (int i = 0; i < c0; ++i) ev[i] = theta0;
            for (int i = 0; i < m2; ++i) ev[c0 + i] = pfd::cplx(wr[i], wi[i]);
            const Dominant d = select_dominant(ev, q_target, n_extra, ellipse);
            q = d.q;
            n_real = d.n_found;
            theta_re.resize(j);
            theta_im.resize(j);
            for (int i = 0; i < j; ++i) theta_re[i] = ev[d.order[i]].real(), theta_im[i] = ev[d.order[i]].imag();
            res.assign(q, 0.0);
            const double beta = fabs(b[j - 1]);
            for (int col = 0; col < q; ++col) {
                if (d.order[col] < c0) continue;  // a locked null vector: exact
                res[col] = beta * pfd::hessenberg_residual_factor(H, ld, j, c0, ev[d.order[col]]);
            }
            U.clear();
            T.clear();
        } else {
            T.assign((size_t)j * j, 0.0);
            for (int a = 0; a < j; ++a)
                for (int bb = 0; bb < j; ++bb) T[(size_t)a * j + bb] = H[(size_t)a * ld + bb];
            std::vector<double> wr, wi;
            PFK_CHECK(pfd::real_schur(T, j, U, wr, wi), PF_E_DEGENERATE, "pf_eigs_smallest: QR iteration of the projected matrix failed");
            pfd::schur_eigenvalues(T, j, ev);
            const Dominant d = select_dominant(ev, q_target, n_extra, ellipse);
            n_real = d.n_found;
            std::vector<char> sel(j, 0);
            for (int i = 0; i < j; ++i) sel[i] = std::abs(ev[i]) > d.thr ? 1 : 0;
            bool all_moved = true;
            q = pfd::schur_reorder(T, j, &U, j, sel, &all_moved);
            // (a refused swap - two blocks with numerically equal eigenvalues - leaves a valid Schur form whose leading q
            // columns span an invariant subspace all the same; the next analysis sorts it out)
            pfd::schur_eigenvalues(T, j, ev);
            theta_re.resize(j);
            theta_im.resize(j);
            for (int i = 0; i < j; ++i) theta_re[i] = ev[i].real(), theta_im[i] = ev[i].imag();
            res.assign(q, 0.0);
            for (int col = 0; col < q; ++col) {
                double r = 0.0;
                for (int i = 0; i < j; ++i) r += b[i] * U[(size_t)i * j + col];
                res[col] = fabs(r);
            }
        }
        theta_absmax = 0.0;
        for (int i = 0; i < j; ++i) theta_absmax = std::max(theta_absmax, hypot(theta_re[i], theta_im[i]));
        theta_min = INFINITY;
        bool any = false;
        for (int i = 0; i < q; ++i) {
            const double mod = hypot(theta_re[i], theta_im[i]);
            if (ellipse) theta_min = std::min(theta_min, mod), any = true;
            else if (fabs(theta_im[i]) <= 1e-9 * mod && theta_re[i] > 0) theta_min = std::min(theta_min, theta_re[i]), any = true;
        }
        if (!any) theta_min = 0.0;
        if (!ellipse && q > q_target + out_cap) return NEED_ELLIPSE;  // too many complex outliers to carry along
        return PF_OK;
    }

    // The Ritz check of the step just collected: pure host work (touches nothing but this solver's members; the error
    // text of a failure goes through pf_set_error, thread-local, so the status travels in analysis_rc and the caller's
    // thread reports it).
    void analyse() {
        analysis_due = false;
        analysis_rc = analyse_step();
    }

    int analyse_step() {
        // Asymmetric graphs: a check is the eigenvalues of the projected matrix (100-140 us at 50 columns) and, if they say
        // "converged", the ordered Schur form with its vectors (200-240 us more), which decides.  Where the trend of the
        // last two checks announced this one as the last, the Schur form is taken at once: it gives the true residuals
        // either way, and the solve's last check - nothing on the device can run beside it - loses the first pass.
        PFK_TRY(ritz(0, !sym && expect_final));
        double worst_res = 0.0;
        for (int col = 0; col < q; ++col) worst_res = std::max(worst_res, res[col]);
        const double scale = tol * std::max(theta_min, 1.0);
        const double worst = worst_res / std::max(scale, 1e-300);
        // Once two checks have shown the (roughly geometric) decay of the largest residual, part of the steps it still needs
        // are skipped before looking again.  Symmetric graphs (a check is ~50 us): half of them, at most 3.  Asymmetric
        // graphs (a check is 100-400 us of QR sweeps against a device step of 60-250 us): 0.9 of them, at most 6 - looking
        // a step late costs one step of device time, looking in vain costs more than that of host time - and every other
        // step while no trend is known.
        next_check = j + (sym ? 1 : 2);
        if (have_seen && worst > 1.0 && seen_worst > worst && n_real >= q_target && theta_min > band) {
            const double per_step = log(seen_worst / worst) / (j - seen_j);
            if (sym) next_check = j + (int)std::min(4.0, std::max(1.0, 0.5 * log(worst) / per_step));
            else next_check = j + (int)std::min(6.0, std::max(1.0, ceil(0.9 * log(worst) / per_step)));
        }
        expect_final = !sym && have_seen && n_real >= q_target && theta_min > band &&
                       (worst <= 10.0 || (worst > 1.0 && seen_worst > worst && log(worst) / (log(seen_worst / worst) / (j - seen_j)) <= next_check - j));
        if (getenv("PF_EIGS_DEBUG")) fprintf(stderr, "  check j=%d q=%d n_real=%d theta_min=%.3g worst=%.3g next=%d\n", j, q, n_real, theta_min, worst, next_check);
        have_seen = true;
        seen_j = j;
        seen_worst = worst;
        near_conv = n_real >= q_target && theta_min > band && worst_res <= 10.0 * scale;
        if (n_real >= q_target && worst_res <= scale && theta_min > band) outcome = 1;
        else if (!sym && !ellipse && theta_absmax > 1e7 * std::max(theta_min, 1.0) && p > 16) outcome = 3;
        else if ((j >= q + 12 || exhausted) && theta_min < band) outcome = 2;
        else if (exhausted) outcome = 1;
        if (outcome == 1 && !sym && U.empty()) {
            // the estimate says converged: the ordered Schur form decides (and provides the vectors)
            PFK_TRY(ritz(0, true));
            worst_res = 0.0;
            for (int col = 0; col < q; ++col) worst_res = std::max(worst_res, res[col]);
            const double scale2 = tol * std::max(theta_min, 1.0);
            if (!exhausted && !(n_real >= q_target && worst_res <= scale2 && theta_min > band)) {
                outcome = 0;  // not yet: look again next step
                near_conv = true;
                next_check = j + 1;
            }
        }
        return PF_OK;
    }

    // host side up to the next request; done == true when the solve is over (eigenvectors in slots [first_slot, + n_out))
    int advance() {
        for (;;) {
            int rc = step();
            if (rc == PF_OK && analysis_due && inline_analysis) {
                analyse();
                continue;
            }
            if (rc == NEED_ELLIPSE) {
                PFK_CHECK(!sym, PF_E_DEGENERATE, "pf_eigs_smallest: internal error (ellipse filter asked for a symmetric graph)");
                rc = begin_mode(true);  // (nothing of the abandoned attempt is left in flight but a speculative filter application)
                if (rc != PF_OK) return rc;
                continue;
            }
            return rc;
        }
    }

    int step() {
        for (;;) {
            switch (state) {
                case S_TOP:
                    if (!spec) {  // (else the filter application of this step was queued with the last Gram-Schmidt step)
                        req = Request{};
                        req.kind = REQ_CHEB;
                        req.cheb_src = A0 + j;
                        req.cheb_dst = A0 + j + 1;
                        st.matvecs += p;
                        state = S_AFTER_CHEB;
                        return PF_OK;
                    }
                    state = S_AFTER_CHEB;
                    break;
                case S_AFTER_CHEB: {
                    if (start_pending) {
                        start_pending = false;
                        double nrm = 0.0;
                        bool redone = false;
                        PFK_TRY(ops->orth_end(h.data(), &nrm, &redone));
                        PFK_CHECK(nrm > 0.0 && isfinite(nrm), PF_E_DEGENERATE, "pf_eigs_smallest: start vector vanished");
                        if (redone) {  // refined after the filter application had read it: apply the filter again
                            state = S_TOP;
                            break;
                        }
                    }
                    st.outer_steps += 1;
                    // Symmetric graphs: B v0 lies mostly along v0 and the locked null vectors - the first Gram-Schmidt step of
                    // every solve (250k blobs: always; the second sometimes) cancels digits, and its second pass, run by
                    // orth_end AFTER the speculative application had read the vector, cost that application (200 us of a
                    // 10.7 ms step, both graphs).  The first two steps take the device's own second pass; later steps (ratios
                    // 0.35-0.78) the single pass without the two idle launches.

                    // the Gram-Schmidt step and - to keep the device busy - the NEXT filter application, queued before this
                    // step's coefficients are read (a speculative application after the last step would be wasted)
                    spec = j + 1 < m_max && !near_conv;
                    req = Request{};
                    req.orth_w = A0 + j + 1;
                    req.orth_first = A0;
                    req.orth_count = j + 1;
                    // (local: not before the basis has two Lanczos vectors, never after a restart - the kept Ritz vectors
                    // are no Lanczos sequence -, not while full steps are owed)
                    pro_local = pro && hess && restarts == 0 && pro_force == 0 && j >= c0 + 2 && ops->orth_split(A0 + j - 1, c0);
                    if (pro_local) req.orth_count = c0 + 2;
                    else if (pro_force > 0) --pro_force;
                    if (sym) {
                        // A FULL step of a run with local steps: two passes, always - one pass of classical Gram-Schmidt
                        // against a basis that is orthogonal to 1e-9 only leaves the new vector at 1e-9, and the estimates
                        // (which restart from the unit roundoff there) would be wrong from then on.  Both on the device.
                        const bool pro_full = pro && hess && restarts == 0 && !pro_local && j >= c0 + 2;
                        PFK_TRY(ops->orth_device_passes(j <= c0 + 1 || pro_full));
                        if (pro) PFK_TRY(ops->orth_always_twice(pro_full, n_active < 4096));
                    }
                    if (spec) {
                        req.kind = REQ_ORTH_CHEB;
                        req.cheb_src = A0 + j + 1;
                        req.cheb_dst = A0 + j + 2;
                        st.matvecs += p;
                    } else {
                        req.kind = REQ_ORTH;
                    }
                    state = S_AFTER_ORTH;
                    return PF_OK;
                }
                case S_AFTER_ORTH: {
                    double beta = 0.0;
                    bool redone = false;
                    PFK_TRY(ops->orth_end(h.data(), &beta, &redone));
                    if (redone || ops->last_twice) st.second_passes += 1;
                    if (redone) spec = false;  // w was refined after the speculative application had read it: apply the filter again
                    bool finite = isfinite(beta);
                    for (int i = 0; i < (pro_local ? c0 + 2 : j + 1) && finite; ++i) finite = isfinite(h[i]);
                    if (!finite && !sym) return NEED_ELLIPSE;  // an outlier outside the damped set overflowed at this degree
                    PFK_CHECK(finite, PF_E_DEGENERATE, "pf_eigs_smallest: the Chebyshev filter overflowed (degree %d): the operator has "
                              "eigenvalues above the assumed bound %g", p, hi);
                    if (pro_local) st.local_steps += 1;
                    if (pro_local) {  // h = [nulls (c0), v_{j-1}, v_j]
                        for (int i = 0; i <= j; ++i) H[(size_t)i * m_max + j] = 0.0;
                        for (int i = 0; i < c0; ++i) H[(size_t)i * m_max + j] = h[i];
                        H[(size_t)(j - 1) * m_max + j] = h[c0];
                        H[(size_t)j * m_max + j] = h[c0 + 1];
                    } else {
                        for (int i = 0; i <= j; ++i) H[(size_t)i * m_max + j] = h[i];
                        // (partial reorthogonalisation keeps the matrix tridiagonal: what a full step removes from the
                        // older vectors is the drift the estimates track, not a coefficient of the recurrence)
                        if (pro && hess && restarts == 0)
                            for (int i = c0; i + 1 < j; ++i) H[(size_t)i * m_max + j] = 0.0;
                    }
                    for (int i = 0; i < j; ++i) H[(size_t)j * m_max + i] = b[i];
                    if (pro && hess && restarts == 0) pro_update(beta);
                    ++j;
                    std::fill(b.begin(), b.end(), 0.0);
                    b[j - 1] = beta;
                    exhausted = beta <= 1e-14 * std::max(fabs(theta0), 1.0) || j >= n_active;
                    outcome = 0;
                    analysis_rc = PF_OK;
                    state = S_AFTER_ANALYSIS;
                    if (exhausted || j == m_max || j >= std::max(q_target + (sym ? 8 : 12), next_check)) {
                        analysis_due = true;  // (the caller runs analyse(), here or on a helper thread, and calls again)
                        req = Request{};
                        return PF_OK;
                    }
                    break;
                }
                case S_AFTER_ANALYSIS: {
                    if (analysis_rc != PF_OK && analysis_rc != NEED_ELLIPSE)  // (the analysis may have run on another thread: the text is set here)
                        pf_set_error("pf_eigs_smallest: the QR iteration of the projected %d x %d matrix failed", j, j);
                    if (analysis_rc != PF_OK) return analysis_rc;
                    if (outcome == 1) {
                        PFK_TRY(extract_a());
                        extracting = true;
                        state = S_EXTRACT_B;
                        req = Request{};
                        if (stepwise_extract) return PF_OK;
                        break;
                    }
                    if (outcome == 2 || outcome == 3) {
                        if (resets >= 8) {
                            if (!sym && !ellipse) return NEED_ELLIPSE;
                            PFK_CHECK(false, PF_E_DEGENERATE, "pf_eigs_smallest: could not place the Chebyshev filter (cut %g, degree %d)", cut, p);
                        }
                        if (outcome == 3) {
                            degree_cap = std::max(16, p / 2);  // lower the degree: less dynamic range between outliers and wanted values
                        } else {  // wanted eigenvalues sit inside the damped band: widen the undamped interval
                            std::vector<double> est;
                            if (!ellipse)
                                for (int col = 0; col < std::min(q, (int)theta_re.size()); ++col)
                                    if (fabs(theta_im[col]) <= 1e-9 * hypot(theta_re[col], theta_im[col]) && theta_re[col] > 1.5)
                                        est.push_back(cheb_inverse(theta_re[col], c, e, p));
                            std::sort(est.begin(), est.end());
                            if ((int)est.size() > c0) est.erase(est.begin(), est.begin() + c0);
                            else est.clear();
                            cut = est.size() >= 2 ? std::max(4.0 * cut, 2.5 * est.back() * (want + 1) / (double)est.size()) : 8.0 * cut;
                            cut = std::min(cut, hi);
                        }
                        ++resets;
                        st.filter_resets += 1;
                        PFK_TRY(begin_filter());
                        break;
                    }
                    if (j < m_max) {
                        state = S_TOP;
                        break;
                    }
                    // ---- thick restart: the dominant Ritz / Schur vectors + a buffer, then the residual vector
                    if (restarts >= 60) {
                        if (!sym && !ellipse) return NEED_ELLIPSE;
                        PFK_CHECK(false, PF_E_DEGENERATE, "pf_eigs_smallest: no convergence after 60 restarts");
                    }
                    {
                        const int rc = ritz(std::max(4, q_target / 2), true);
                        if (rc != PF_OK) return rc;
                        int n_keep = std::min(q, j - 1);
                        if (!sym && n_keep >= 1 && n_keep < j && T[(size_t)n_keep * j + n_keep - 1] != 0.0) n_keep -= 1;  // never split a 2 x 2 block
                        PFK_CHECK(n_keep >= 1, PF_E_DEGENERATE, "pf_eigs_smallest: nothing to keep at a restart");
                        std::vector<double> Y((size_t)j * n_keep), bn((size_t)m_max, 0.0);
                        for (int i = 0; i < j; ++i)
                            for (int col = 0; col < n_keep; ++col) Y[(size_t)i * n_keep + col] = U[(size_t)i * j + col];
                        PFK_TRY(ops->combine(A0, j, Y.data(), n_keep, B0));
                        PFK_TRY(ops->copy(A0 + j, B0 + n_keep, 1));  // the residual vector follows the kept block
                        PFK_TRY(ops->copy(B0, A0, n_keep + 1));
                        if (pro && st.local_steps > 0 && restarts == 0) {
                            // the basis of a run with local steps is orthogonal to 1e-9 only, and so are the Ritz vectors
                            // kept from it: from here on every step is a full one, against a basis that is orthonormal
                            // again - column by column, two passes each (a rare event: ~20 small steps)
                            PFK_TRY(ops->orth_device_passes(true));
                            PFK_TRY(ops->orth_always_twice(true, n_active < 4096));
                            std::vector<double> hh((size_t)n_keep + 1);
                            for (int col = 1; col <= n_keep; ++col) {
                                double nrm_c = 0.0;
                                bool redone_c = false;
                                PFK_TRY(ops->orth_begin(A0 + col, A0, col));
                                PFK_TRY(ops->orth_end(hh.data(), &nrm_c, &redone_c));
                                PFK_CHECK(nrm_c > 0.5, PF_E_DEGENERATE, "pf_eigs_smallest: the kept Ritz vectors lost their independence (%g)", nrm_c);
                            }
                            PFK_TRY(ops->orth_always_twice(false, n_active < 4096));
                        }
                        for (int col = 0; col < n_keep; ++col) {
                            double r = 0.0;
                            for (int i = 0; i < j; ++i) r += U[(size_t)i * j + col] * b[i];
                            bn[col] = r;
                        }
                        std::fill(H.begin(), H.end(), 0.0);
                        if (sym) {
                            for (int i = 0; i < n_keep; ++i) H[(size_t)i * m_max + i] = theta_re[i];
                        } else {
                            for (int a = 0; a < n_keep; ++a)
                                for (int bb = 0; bb < n_keep; ++bb) H[(size_t)a * m_max + bb] = T[(size_t)a * j + bb];
                        }
                        b = bn;
                        j = n_keep;
                        hess = false;
                    }
                    ++restarts;
                    st.restarts += 1;
                    begin_expand();
                    break;
                }
                case S_EXTRACT_B: {
                    const int rc = extract_b();
                    if (rc != PF_OK) {
                        extracting = false;
                        return rc;  // (NEED_ELLIPSE: advance() begins the other mode)
                    }
                    state = S_EXTRACT_C;
                    if (stepwise_extract) return PF_OK;
                    break;
                }
                case S_EXTRACT_C:
                    PFK_TRY(extract_c());
                    extracting = false;
                    done = true;
                    state = S_DONE;
                    req = Request{};
                    return PF_OK;
                case S_DONE:
                    done = true;
                    req = Request{};
                    return PF_OK;
            }
        }
    }

    // Rayleigh-Ritz on A itself over the converged Ritz / Schur vectors
    int extract_a() {
        PFK_CHECK(2 * q + 1 <= reg, PF_E_STATE, "pf_eigs_smallest: workspace too small for the extraction (q = %d)", q);
        {
            std::vector<double> Y((size_t)j * q);
            for (int i = 0; i < j; ++i)
                for (int col = 0; col < q; ++col) Y[(size_t)i * q + col] = U[(size_t)i * j + col];
            PFK_TRY(ops->combine(A0, j, Y.data(), q, B0));  // Z -> region B
        }
        PFK_TRY(ops->spmv_multi(op, B0, A0, q));  // A Z -> region A (the Krylov basis is no longer needed)
        st.matvecs += q;
        // G[i][r] = <A z_i, z_r>; after partial reorthogonalisation also <z_i, z_r>: the Ritz vectors of a basis that is
        // orthogonal to 1e-9 only are orthonormalised by the Rayleigh-Ritz step itself (generalised problem)
        return ops->gram_begin(A0, q, B0, q, pro);
    }

    int extract_b() {
        std::vector<double> G((size_t)q * q * (pro ? 2 : 1)), HA((size_t)q * q), lam;
        std::vector<double>& Rk = ex_Rk;
        PFK_TRY(ops->gram_end(G.data()));
        int nk = 0;
        if (sym) {
            for (int a = 0; a < q; ++a)
                for (int bb = 0; bb < q; ++bb) HA[(size_t)a * q + bb] = 0.5 * (G[(size_t)a * q + bb] + G[(size_t)bb * q + a]);
            std::vector<double> Lc;  // pro: M = Z^T Z = Lc Lc^T (lower), HA <- Lc^-1 HA Lc^-T
            if (pro) {
                const double* M = G.data() + (size_t)q * q;
                Lc.assign((size_t)q * q, 0.0);
                for (int a = 0; a < q; ++a)
                    for (int bb = 0; bb <= a; ++bb) {
                        double v = 0.5 * (M[(size_t)a * q + bb] + M[(size_t)bb * q + a]);
                        for (int t = 0; t < bb; ++t) v -= Lc[(size_t)a * q + t] * Lc[(size_t)bb * q + t];
                        if (a == bb) {
                            PFK_CHECK(v > 0.25, PF_E_DEGENERATE, "pf_eigs_smallest: the Ritz vectors lost their independence (%g)", v);
                            Lc[(size_t)a * q + a] = sqrt(v);
                        } else {
                            Lc[(size_t)a * q + bb] = v / Lc[(size_t)bb * q + bb];
                        }
                    }
                // C = Lc^-1 HA Lc^-T: forward substitution on the rows, then on the columns
                for (int col = 0; col < q; ++col)
                    for (int a = 0; a < q; ++a) {
                        double v = HA[(size_t)a * q + col];
                        for (int t = 0; t < a; ++t) v -= Lc[(size_t)a * q + t] * HA[(size_t)t * q + col];
                        HA[(size_t)a * q + col] = v / Lc[(size_t)a * q + a];
                    }
                for (int row = 0; row < q; ++row)
                    for (int a = 0; a < q; ++a) {
                        double v = HA[(size_t)row * q + a];
                        for (int t = 0; t < a; ++t) v -= Lc[(size_t)a * q + t] * HA[(size_t)row * q + t];
                        HA[(size_t)row * q + a] = v / Lc[(size_t)a * q + a];
                    }
                for (int a = 0; a < q; ++a)
                    for (int bb = 0; bb < a; ++bb) HA[(size_t)a * q + bb] = HA[(size_t)bb * q + a] = 0.5 * (HA[(size_t)a * q + bb] + HA[(size_t)bb * q + a]);
            }
            pfd::eigh_sym(HA, q, lam);  // HA <- eigenvectors (columns)
            if (pro) {  // R = Lc^-T Y: back substitution, column by column
                for (int col = 0; col < q; ++col)
                    for (int a = q - 1; a >= 0; --a) {
                        double v = HA[(size_t)a * q + col];
                        for (int t = a + 1; t < q; ++t) v -= Lc[(size_t)t * q + a] * HA[(size_t)t * q + col];
                        HA[(size_t)a * q + col] = v / Lc[(size_t)a * q + a];
                    }
            }
            std::vector<int> order(q);
            for (int i = 0; i < q; ++i) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](int a, int bb) { return lam[a] < lam[bb]; });
            std::vector<int> keep;
            for (int i : order)
                if (lam[i] > 1e-10) keep.push_back(i);  // graph.py:381
            st.n_null = q - (int)keep.size();
            nk = std::min((int)keep.size(), (int)n_wanted);
            Rk.assign((size_t)q * std::max(nk, 1), 0.0);
            for (int r = 0; r < q; ++r)
                for (int cidx = 0; cidx < nk; ++cidx) Rk[(size_t)r * nk + cidx] = HA[(size_t)r * q + keep[cidx]];
            for (int i = 0; i < nk; ++i) vals[i] = lam[keep[i]];
        } else {
            for (int a = 0; a < q; ++a)
                for (int bb = 0; bb < q; ++bb) HA[(size_t)a * q + bb] = G[(size_t)bb * q + a];  // (Z^T A Z)[a][b] = <A z_b, z_a>
            std::vector<double> Zs, wr, wi;
            PFK_CHECK(pfd::real_schur(HA, q, Zs, wr, wi), PF_E_DEGENERATE, "pf_eigs_smallest: QR iteration of the Rayleigh quotient failed");
            std::vector<int> all(q);
            for (int i = 0; i < q; ++i) all[i] = i;
            std::vector<pfd::cplx> Vc, lam_c;
            pfd::schur_eigenvectors(HA, Zs, q, all, Vc, lam_c);
            std::vector<int> order(q);
            for (int i = 0; i < q; ++i) order[i] = i;
            // complex outliers carried by the interval filter have Re ~ 1: the wanted ones are the lowest real parts
            std::stable_sort(order.begin(), order.end(), [&](int a, int bb) { return lam_c[a].real() < lam_c[bb].real(); });
            int take = std::min(q_target, q);
            if (take < q && fabs(lam_c[order[take - 1]].imag()) > 1e-9 &&
                fabs(lam_c[order[take - 1]].real() - lam_c[order[take]].real()) <= 1e-9 * fabs(lam_c[order[take]].real()))
                take += 1;  // never split a conjugate pair
            std::vector<char> is_cplx(take);
            bool any_cplx = false, above = false;
            for (int i = 0; i < take; ++i) {
                const pfd::cplx z = lam_c[order[i]];
                is_cplx[i] = fabs(z.imag()) > 1e-9 * std::max(std::abs(z), 1e-300);
                any_cplx = any_cplx || is_cplx[i];
                above = above || z.real() > cut;
            }
            // (complex LOW eigenvalues - a hole large enough for the low modes to feel its one-way boundary - are no reason to
            // give the interval filter's result up: the dominant set holds every eigenvalue that the polynomial amplifies at
            // least as much as the last wanted real one, and |T_p| grows both to the left and away from the axis, so every
            // eigenvalue with a smaller real part is in the converged subspace, complex or not)
            (void)any_cplx;
            if (ellipse && above) return NEED_ELLIPSE;      // junk from above the assumed strip crept into the dominant subspace
            // the reference keeps np.real(eig_vals) and np.real(eig_vecs) (graph.py:386-389): a conjugate pair shows up as a
            // repeated value with the same real part twice - with ARPACK's run-dependent phase; here the phase is fixed
            // (largest component real positive)
            std::vector<int> keep;
            for (int i = 0; i < take; ++i)
                if (lam_c[order[i]].real() > 1e-10) keep.push_back(i);
            st.n_null = std::min(q_target, take) - (int)keep.size();
            nk = (int)keep.size();
            if (nk == 0) return NEED_ELLIPSE;  // nothing but (numerically) null directions in the dominant subspace
            nk = std::min(nk, (int)n_wanted);
            Rk.assign((size_t)q * nk, 0.0);
            for (int cidx = 0; cidx < nk; ++cidx) {
                const int col = order[keep[cidx]];
                pfd::cplx phase = 1.0;
                if (is_cplx[keep[cidx]]) {
                    int piv = 0;
                    double best = -1.0;
                    for (int r = 0; r < q; ++r)
                        if (std::abs(Vc[(size_t)r * q + col]) > best) best = std::abs(Vc[(size_t)r * q + col]), piv = r;
                    phase = std::conj(Vc[(size_t)piv * q + col]) / std::abs(Vc[(size_t)piv * q + col]);
                }
                double nrm = 0.0;
                for (int r = 0; r < q; ++r) {
                    const double v = (Vc[(size_t)r * q + col] * phase).real();
                    Rk[(size_t)r * nk + cidx] = v;
                    nrm += v * v;
                }
                nrm = sqrt(nrm);
                for (int r = 0; r < q; ++r) Rk[(size_t)r * nk + cidx] /= nrm;
                vals[cidx] = lam_c[col].real();
            }
        }
        const int X0 = B0 + q, AX0 = A0 + q;
        PFK_CHECK(X0 + nk <= 2 * reg && AX0 + nk <= reg, PF_E_STATE, "pf_eigs_smallest: workspace too small for the extraction");
        n_out = nk;
        first_slot = X0;
        ex_nk = nk;
        if (nk > 0) {
            PFK_TRY(ops->combine2(B0, q, Rk.data(), nk, X0, A0, AX0));  // X = Z R, A X = (A Z) R
            // (for a complex pair the real part alone is not an eigenvector: its "residual" is |Im lambda| |Im x|)
            PFK_TRY(ops->resnorms_begin(AX0, X0, vals.data(), nk));
        }
        return PF_OK;
    }

    int extract_c() {
        if (ex_nk > 0) {
            PFK_TRY(ops->resnorms_end(residuals.data()));
            st.max_residual = 0.0;
            for (int i = 0; i < ex_nk; ++i) st.max_residual = std::max(st.max_residual, residuals[i]);
        }
        return PF_OK;
    }

    void cheb_args(int32_t* ci, double* cd) const {
        ci[0] = op, ci[1] = req.cheb_src, ci[2] = req.cheb_dst, ci[3] = p;
        cd[0] = c, cd[1] = e, cd[2] = rho;
    }
    void orth_args(int32_t* o) const { o[0] = req.orth_w, o[1] = req.orth_first, o[2] = req.orth_count, o[3] = 1; }
    int run_cheb() { return ops->cheb(op, req.cheb_src, req.cheb_dst, p, c, e, rho); }
    int run_orth() { return ops->orth_begin(req.orth_w, req.orth_first, req.orth_count); }
};

inline int drive_single(Solver& s) {
    for (;;) {
        PFK_TRY(s.advance());
        if (s.done) return PF_OK;
        switch (s.req.kind) {
            case REQ_CHEB: PFK_TRY(s.run_cheb()); break;
            case REQ_ORTH: PFK_TRY(s.run_orth()); break;
            case REQ_ORTH_CHEB:
                PFK_TRY(s.run_orth());
                PFK_TRY(s.run_cheb());
                break;
            default: break;
        }
    }
}

// One helper thread that runs a solver's Ritz analysis beside the caller's: the analyses of the two graphs of a pair are
// independent host work (100-400 us each for asymmetric graphs) on the critical path of every step that has one.
struct AnalysisHelper {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    Solver* job = nullptr;
    bool busy = false, quit = false;
    AnalysisHelper() { th = std::thread([this] { loop(); }); }
    ~AnalysisHelper() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv.notify_all();
        th.join();
    }
    void submit(Solver* s) {
        {
            std::lock_guard<std::mutex> lk(m);
            job = s;
            busy = true;
        }
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !busy; });
    }
    void loop() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return job != nullptr || quit; });
            if (quit) return;
            Solver* s = job;
            job = nullptr;
            lk.unlock();
            s->analyse();
            lk.lock();
            busy = false;
            cv.notify_all();
        }
    }
};

inline int launch_request(Solver& s) {
    switch (s.req.kind) {
        case REQ_CHEB: return s.run_cheb();
        case REQ_ORTH: return s.run_orth();
        case REQ_ORTH_CHEB:
            PFK_TRY(s.run_orth());
            return s.run_cheb();
        default: return PF_OK;
    }
}

// advance the solvers named (both, or one) to their next device request, running the analyses that come due on the way -
// two at a time on two threads when a helper is at hand.  When ONE of them has just converged, its extraction (a chain of
// small launches and synchronisations, ~0.5 ms) would leave the partner's next step unqueued meanwhile: the partner goes
// first and its request is queued at once (req_launched), the extraction runs beside it.
inline int advance_pair(Solver& a, Solver& b, bool do_a, bool do_b, AnalysisHelper* helper) {
    if (do_a) PFK_TRY(a.advance());
    if (do_b) PFK_TRY(b.advance());
    // the phases of the extraction(s) in flight, the two graphs' alternating: each phase ends in launches, the next begins
    // with the wait for them - which the partner's launches fill
    auto extractions = [&]() -> int {
        for (;;) {
            bool any = false;
            if (do_a && !a.done && a.extracting) {
                PFK_TRY(a.advance());
                any = true;
            }
            if (do_b && !b.done && b.extracting) {
                PFK_TRY(b.advance());
                any = true;
            }
            if (!any) return PF_OK;
        }
    };
    PFK_TRY(extractions());
    for (;;) {
        const bool da = do_a && !a.done && a.analysis_due, db = do_b && !b.done && b.analysis_due;
        if (!da && !db) return PF_OK;
        if (da && db && helper) {
            helper->submit(&b);
            a.analyse();
            helper->wait();
        } else {
            if (da) a.analyse();
            if (db) b.analyse();
        }
        const bool ea = da && a.analysis_rc == PF_OK && a.outcome == 1, eb = db && b.analysis_rc == PF_OK && b.outcome == 1;
        if (ea != eb) {
            Solver& ex = ea ? a : b;
            Solver& other = ea ? b : a;
            const bool other_due = ea ? db : da, other_active = ea ? do_b : do_a;
            if (other_due) PFK_TRY(other.advance());
            if (other_active && !other.done && !other.analysis_due && !other.req_launched && other.req.kind != REQ_NONE) {
                PFK_TRY(launch_request(other));
                other.req_launched = true;
            }
            PFK_TRY(ex.advance());
            PFK_TRY(extractions());
            continue;
        }
        if (da) PFK_TRY(a.advance());
        if (db) PFK_TRY(b.advance());
        PFK_TRY(extractions());
    }
}

// the two solvers of a pair in lockstep: whatever both have pending runs in launches the two graphs share
inline int drive_pair(Solver& a, Solver& b) {
    // (symmetric pairs keep their ~50 us analyses in line: a helper would cost more than it saves)
    std::unique_ptr<AnalysisHelper> helper;
    if ((!a.done && !a.sym) || (!b.done && !b.sym)) {
        if (!getenv("PF_EIGS_HELPER") || atoi(getenv("PF_EIGS_HELPER")) != 0) helper.reset(new AnalysisHelper());
    }
    a.inline_analysis = b.inline_analysis = false;
    a.stepwise_extract = b.stepwise_extract = true;
    PFK_TRY(advance_pair(a, b, !a.done, !b.done, helper.get()));
    while (!a.done || !b.done) {
        if ((!a.done && a.req_launched) || (!b.done && b.req_launched)) {  // queued ahead of the partner's extraction: collect
            Solver& s = (!a.done && a.req_launched) ? a : b;
            s.req_launched = false;
            PFK_TRY(advance_pair(a, b, &s == &a, &s == &b, helper.get()));
            continue;
        }
        const ReqKind ka = a.done ? REQ_NONE : a.req.kind, kb = b.done ? REQ_NONE : b.req.kind;
        const bool oa = ka == REQ_ORTH || ka == REQ_ORTH_CHEB, ob = kb == REQ_ORTH || kb == REQ_ORTH_CHEB;
        int32_t orth[8], ci[8];
        double cd[6];
        if (ka == REQ_ORTH_CHEB && kb == REQ_ORTH_CHEB) {
            // one outer step of both solvers in one library call: both Gram-Schmidt steps in shared launches and, right
            // behind them, the next filter application of both
            a.orth_args(orth);
            b.orth_args(orth + 4);
            a.cheb_args(ci, cd);
            b.cheb_args(ci + 4, cd + 3);
            PFK_TRY(a.ops->orth_cheb_pair(*b.ops, orth, ci, cd));
            PFK_TRY(advance_pair(a, b, true, true, helper.get()));
        } else if (oa || ob) {
            // Gram-Schmidt steps first (shared launches if both graphs have one); a fused request leaves its filter part
            if (oa && ob) {
                a.orth_args(orth);
                b.orth_args(orth + 4);
                PFK_TRY(a.ops->orth_begin_pair(*b.ops, orth));
            } else {
                PFK_TRY((oa ? a : b).run_orth());
            }
            bool adv[2] = {false, false};
            int i = 0;
            for (Solver* s : {&a, &b}) {
                const ReqKind k = s->done ? REQ_NONE : s->req.kind;
                if (k == REQ_ORTH_CHEB && (s == &a ? oa : ob)) s->req.kind = REQ_CHEB;  // (its coefficients are read once the filter part is queued too)
                else if (k == REQ_ORTH) adv[i] = true;
                ++i;
            }
            PFK_TRY(advance_pair(a, b, adv[0], adv[1], helper.get()));
        } else if (ka == REQ_CHEB && kb == REQ_CHEB) {
            a.cheb_args(ci, cd);
            b.cheb_args(ci + 4, cd + 3);
            PFK_TRY(a.ops->cheb_pair(*b.ops, ci, cd));
            PFK_TRY(advance_pair(a, b, true, true, helper.get()));
        } else {
            Solver& s = ka == REQ_CHEB ? a : b;
            PFK_CHECK(!s.done && s.req.kind == REQ_CHEB, PF_E_STATE, "pf_eigs_smallest2: driver out of step");
            PFK_TRY(s.run_cheb());
            PFK_TRY(advance_pair(a, b, &s == &a, &s == &b, helper.get()));
        }
    }
    return PF_OK;
}

}  // namespace pfk
