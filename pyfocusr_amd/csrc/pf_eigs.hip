// pf_eigs_smallest / pf_eigs_smallest2: the whole eigensolve of one symmetric mesh graph, or of the two graphs of a
// pair, behind ONE C call.
//
// Replaces scipy.sparse.linalg.eigs(L, k, sigma=1e-10, which="LM", ncv=4k) of the reference (graph.py:372; called once
// per mesh by Graph.get_graph_spectrum, graph.py:243-248) for callers that bind the C-ABI without Python, and serves the
// Python mirror's paired solve of symmetric graphs: the same Chebyshev-filtered thick-restart Lanczos iteration as
// pyfocusr_amd/_krylov.py (its symmetric branch, restated in C++ on top of the same device primitives) -
//   operator S = G^1/2 (D - W) G^1/2, B = T_p((c - S)/e) with the damped interval [cut, hi] (cut starts at
//   12 (k+1)/n and is enlarged if wanted Ritz values sit in the damped band; hi = the graph's proven spectral bound), a
//   Gram-Schmidt step against the whole basis on the device (second pass on demand), analytic null vectors locked per
//   connected component, isolated vertices masked, thick restart on the dominant Ritz vectors, final Rayleigh-Ritz on S
//   itself, eigenvalues > 1e-10 kept (graph.py:381).
// Pipelined like the Python driver: the Gram-Schmidt step of an outer step and the NEXT filter application are queued
// together (pf_orth_begin + pf_cheb; for a pair pf_orth_cheb2: both graphs in shared launches) before the step's
// coefficients are read, so the device never waits for the host's Ritz check; the check itself (a symmetric eigenproblem
// of <= 49 x 49: Householder + implicit QL, ~50 us) runs one step behind and is skipped on steps that the geometric
// decay of the residuals predicts to be far from convergence; speculation is given up within one digit of convergence.
// The Python driver remains the general one (asymmetric W, complex spectra); these entry points refuse what they do
// not cover (PF_E_STATE) instead of guessing.
#include <math.h>

#include <algorithm>
#include <vector>

#include "pf_internal.h"

namespace {

// ---- symmetric eigenproblem of the projected matrix: Householder tridiagonalisation + implicit QL (the classical
// tred2 / tql2 pair).  V: n x n row-major symmetric on entry, the eigenvectors (columns) on return; d: eigenvalues
// (in no particular order).
void eigh_sym(std::vector<double>& V, int n, std::vector<double>& d) {
    std::vector<double> e((size_t)n, 0.0);
    d.assign((size_t)n, 0.0);
    auto at = [&](int r, int c) -> double& { return V[(size_t)r * n + c]; };
    if (n == 1) {
        d[0] = at(0, 0);
        at(0, 0) = 1.0;
        return;
    }
    for (int j = 0; j < n; ++j) d[j] = at(n - 1, j);
    for (int i = n - 1; i > 0; --i) {  // Householder reduction to tridiagonal form
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; ++k) scale += fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; ++j) {
                d[j] = at(i - 1, j);
                at(i, j) = 0.0;
                at(j, i) = 0.0;
            }
        } else {
            for (int k = 0; k < i; ++k) {
                d[k] /= scale;
                h += d[k] * d[k];
            }
            double f = d[i - 1];
            double g = sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h -= f * g;
            d[i - 1] = f - g;
            for (int j = 0; j < i; ++j) e[j] = 0.0;
            for (int j = 0; j < i; ++j) {
                f = d[j];
                at(j, i) = f;
                g = e[j] + at(j, j) * f;
                for (int k = j + 1; k <= i - 1; ++k) {
                    g += at(k, j) * d[k];
                    e[k] += at(k, j) * f;
                }
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; ++j) {
                e[j] /= h;
                f += e[j] * d[j];
            }
            const double hh = f / (h + h);
            for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
            for (int j = 0; j < i; ++j) {
                f = d[j];
                g = e[j];
                for (int k = j; k <= i - 1; ++k) at(k, j) -= (f * e[k] + g * d[k]);
                d[j] = at(i - 1, j);
                at(i, j) = 0.0;
            }
        }
        d[i] = h;
    }
    for (int i = 0; i < n - 1; ++i) {  // accumulate the transformations
        at(n - 1, i) = at(i, i);
        at(i, i) = 1.0;
        const double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; ++k) d[k] = at(k, i + 1) / h;
            for (int j = 0; j <= i; ++j) {
                double g = 0.0;
                for (int k = 0; k <= i; ++k) g += at(k, i + 1) * at(k, j);
                for (int k = 0; k <= i; ++k) at(k, j) -= g * d[k];
            }
        }
        for (int k = 0; k <= i; ++k) at(k, i + 1) = 0.0;
    }
    for (int j = 0; j < n; ++j) {
        d[j] = at(n - 1, j);
        at(n - 1, j) = 0.0;
    }
    at(n - 1, n - 1) = 1.0;
    e[0] = 0.0;
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];  // implicit QL
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = 2.220446049250313e-16;
    for (int l = 0; l < n; ++l) {
        tst1 = std::max(tst1, fabs(d[l]) + fabs(e[l]));
        int m = l;
        while (m < n - 1 && fabs(e[m]) > eps * tst1) ++m;
        if (m > l) {
            int iter = 0;
            do {
                ++iter;
                double g = d[l];
                double p = (d[l + 1] - g) / (2.0 * e[l]);
                double r = hypot(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                const double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; ++i) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c;
                const double el1 = e[l + 1];
                double s = 0.0, s2 = 0.0;
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2;
                    c2 = c;
                    s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = hypot(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (int k = 0; k < n; ++k) {
                        h = at(k, i + 1);
                        at(k, i + 1) = s * at(k, i) + c * h;
                        at(k, i) = c * at(k, i) - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (fabs(e[l]) > eps * tst1 && iter < 80);
        }
        d[l] = d[l] + f;
        e[l] = 0.0;
    }
}

double cheb_value(double lam, double c, double e, int p) {
    const double t = (c - lam) / e;
    if (fabs(t) <= 1.0) return cos(p * acos(t));
    const double s = (t > 0 || p % 2 == 0) ? 1.0 : -1.0;
    return s * cosh(p * acosh(fabs(t)));
}

double cheb_inverse(double theta, double c, double e, int p) { return c - e * cosh(acosh(std::max(theta, 1.0)) / p); }

void choose_filter(double cut, double hi, double strength, double* c, double* e, int* p) {
    cut = std::min(std::max(cut, 1e-12), 0.5 * hi);
    *c = 0.5 * (hi + cut);
    *e = 0.5 * (hi - cut);
    const double growth = acosh((*c - 0.5 * cut) / *e);
    *p = std::max(8, std::min(4000, (int)ceil(strength / growth)));
}

// Ritz pairs of the symmetric j x j matrix H (leading block of an ld x ld array), dominant first
void ritz_sorted(const std::vector<double>& H, int ld, int j, std::vector<double>& theta, std::vector<double>& U) {
    std::vector<double> A((size_t)j * j), w;
    for (int a = 0; a < j; ++a)
        for (int b = 0; b < j; ++b) A[(size_t)a * j + b] = 0.5 * (H[(size_t)a * ld + b] + H[(size_t)b * ld + a]);
    eigh_sym(A, j, w);
    std::vector<int> order(j);
    for (int i = 0; i < j; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return w[a] > w[b]; });
    theta.resize(j);
    U.assign((size_t)j * j, 0.0);
    for (int col = 0; col < j; ++col) {
        theta[col] = w[order[col]];
        for (int r = 0; r < j; ++r) U[(size_t)r * j + col] = A[(size_t)r * j + order[col]];
    }
}

// ---- one graph's solve as a resumable state machine: advance() runs the host side up to the next device request that
// a driver may want to share with a partner graph (a filter application, a Gram-Schmidt step, or both) - exactly the
// yield points of _krylov._solve_gen
enum ReqKind { REQ_NONE = 0, REQ_CHEB, REQ_ORTH, REQ_ORTH_CHEB };

struct Request {
    ReqKind kind = REQ_NONE;
    int32_t cheb_src = 0, cheb_dst = 0;                  // filter application: slot src -> slot dst
    int32_t orth_w = 0, orth_first = 0, orth_count = 0;  // Gram-Schmidt step of slot w against slots [first, first + count), normalised
};

struct Solver {
    pf_graph* g = nullptr;
    int32_t n_wanted = 0;
    // results
    std::vector<double> vals, residuals;
    int32_t n_out = 0, first_slot = 0;
    pf_eigs_stats st{};
    bool done = false;
    Request req;
    // configuration
    double hi = 2.0, strength = 1.8, tol = 1e-12;  // (strength and the cut below: as _krylov._solve_gen for symmetric graphs)
    int64_t n_active = 0;
    int c0 = 0, want = 0, q_target = 0, m_max = 0, reg = 0, A0 = 0, B0 = 0;
    // filter
    double cut = 0.0, c = 0.0, e = 0.0, theta0 = 0.0, band = 1.5;
    int p = 0;
    uint64_t seed = 0;
    // Krylov state  B V_j = V_j H + v_j b^T ; null vectors are locked exact Ritz pairs
    std::vector<double> H, b, theta, U, h;
    int j = 0, q = 0, restarts = 0, resets = 0;
    bool spec = false, near_conv = false, have_seen = false, start_pending = false;
    int next_check = 0, seen_j = 0;
    double seen_worst = 0.0;
    enum { S_TOP, S_AFTER_CHEB, S_AFTER_ORTH, S_DONE } state = S_TOP;

    int init(pf_graph* graph, int32_t wanted) {
        g = graph;
        n_wanted = wanted;
        PF_CHECK(g && n_wanted >= 1, PF_E_ARG, "pf_eigs_smallest: bad argument");
        PF_CHECK(g->is_symmetric, PF_E_STATE, "pf_eigs_smallest: W is not symmetric (one-way edges): use the Python driver");
        hi = g->spectral_bound;  // (2, or the face-by-face bound of a closed triangle mesh)
        n_active = g->n - g->n_isolated;
        vals.assign((size_t)n_wanted, 0.0);
        residuals.assign((size_t)n_wanted, 0.0);
        // The loose single-pass criterion of the device's Gram-Schmidt step is for the filtered iteration of large graphs; on
        // small ones it can lose orthogonality (pf_orth_strict).  The setting is sticky per graph: set here either way, so
        // that this solve does not inherit what an earlier driver left behind.
        PF_TRY(pf_orth_strict(g, n_active < 4096));
        c0 = g->n_components;  // one analytic null vector per component of >= 2 vertices (written below, once the workspace has its size)
        want = (int)std::min<int64_t>(n_wanted, std::max<int64_t>(n_active - c0, 0));
        if (want <= 0) {
            done = true;
            state = S_DONE;
            return PF_OK;
        }
        q_target = c0 + want;
        PF_CHECK(n_active >= 4 * (q_target + 8), PF_E_STATE,
                 "pf_eigs_smallest: graph too small for the filtered iteration (%lld active vertices): use the Python driver",
                 (long long)n_active);
        m_max = (int)std::min<int64_t>(std::max(3 * q_target + 24, 48), n_active);
        reg = std::max(m_max + 1, 2 * q_target + 2);
        PF_TRY(pf_ws_ensure(g, 2 * reg));
        int32_t locked = 0;
        PF_TRY(pf_lock_null_vectors(g, PF_OP_SYM, &locked));  // slots [0, c0)
        PF_CHECK(locked == c0, PF_E_STATE, "pf_eigs_smallest: %d null vectors locked, %d components", locked, c0);
        A0 = 0, B0 = reg;
        cut = 8.0 * (want + 1) / (double)std::max<int64_t>(n_active, 1);
        if (const char* ev = getenv("PF_EIGS_CUT")) cut *= atof(ev) / 8.0;  // (experiments: the filter's placement ...
        if (const char* ev = getenv("PF_EIGS_STRENGTH")) strength = atof(ev);  // ... and strength; results agree to tol)
        H.assign((size_t)m_max * m_max, 0.0);
        b.assign((size_t)m_max, 0.0);
        h.assign((size_t)m_max + 1, 0.0);
        return begin_filter();
    }

    int begin_filter() {
        PF_CHECK(cut < 0.5 * hi, PF_E_STATE, "pf_eigs_smallest: the wanted eigenvalues are not a corner of the spectrum (cut %g): "
                 "use the Python driver", cut);
        choose_filter(cut, hi, strength, &c, &e, &p);
        st.degree = p;
        st.cut = cut;
        theta0 = cheb_value(0.0, c, e, p);
        std::fill(H.begin(), H.end(), 0.0);
        std::fill(b.begin(), b.end(), 0.0);
        for (int i = 0; i < c0; ++i) H[(size_t)i * m_max + i] = theta0;
        j = c0;
        // start vector, orthogonal to the locked null vectors and normalised on the device; its coefficients are collected
        // when the first filter application has been queued behind it (no synchronisation at the head of the solve)
        PF_TRY(pf_start_vector(g, A0 + j, seed++));
        PF_TRY(pf_orth_begin(g, A0 + j, A0, j, 1));
        start_pending = true;
        restarts = 0;
        begin_expand();
        return PF_OK;
    }

    void begin_expand() {
        spec = false;
        near_conv = false;
        have_seen = false;
        next_check = 0;
        state = S_TOP;
    }

    // host side up to the next request; done == true when the solve is over (eigenvectors in slots [first_slot, + n_out))
    int advance() {
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
                        PF_TRY(pf_orth_end(g, h.data(), &nrm));
                        PF_CHECK(nrm > 0.0 && isfinite(nrm), PF_E_DEGENERATE, "pf_eigs_smallest: start vector vanished");
                        if (pf_orth_redone(g)) {  // refined after the filter application had read it: apply the filter again
                            state = S_TOP;
                            break;
                        }
                    }
                    st.outer_steps += 1;
                    // the Gram-Schmidt step and - to keep the device busy - the NEXT filter application, queued before this
                    // step's coefficients are read (a speculative application after the last step would be wasted)
                    spec = j + 1 < m_max && !near_conv;
                    req = Request{};
                    req.orth_w = A0 + j + 1;
                    req.orth_first = A0;
                    req.orth_count = j + 1;
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
                    PF_TRY(pf_orth_end(g, h.data(), &beta));
                    if (pf_orth_redone(g)) {
                        st.second_passes += 1;
                        spec = false;  // w was refined after the speculative application had read it: apply the filter again
                    }
                    bool finite = isfinite(beta);
                    for (int i = 0; i <= j && finite; ++i) finite = isfinite(h[i]);
                    PF_CHECK(finite, PF_E_DEGENERATE, "pf_eigs_smallest: the Chebyshev filter overflowed (degree %d): the operator has "
                             "eigenvalues above the assumed bound %g", p, hi);
                    for (int i = 0; i <= j; ++i) H[(size_t)i * m_max + j] = h[i];
                    for (int i = 0; i < j; ++i) H[(size_t)j * m_max + i] = b[i];
                    ++j;
                    std::fill(b.begin(), b.end(), 0.0);
                    b[j - 1] = beta;
                    const bool exhausted = beta <= 1e-14 * std::max(fabs(theta0), 1.0) || j >= n_active;
                    int outcome = 0;  // 1 converged, 2 the cut has to move
                    if (exhausted || j == m_max || j >= std::max(q_target + 8, next_check)) {
                        ritz_sorted(H, m_max, j, theta, U);
                        q = std::min(q_target, j);
                        double theta_min = INFINITY, worst_res = 0.0;
                        bool any_pos = false;
                        for (int col = 0; col < q; ++col)
                            if (theta[col] > 0.0) theta_min = std::min(theta_min, theta[col]), any_pos = true;
                        if (!any_pos) theta_min = 0.0;
                        for (int col = 0; col < q; ++col) {
                            double r = 0.0;
                            for (int i = 0; i < j; ++i) r += b[i] * U[(size_t)i * j + col];
                            worst_res = std::max(worst_res, fabs(r));
                        }
                        const double scale = tol * std::max(theta_min, 1.0);
                        const double worst = worst_res / std::max(scale, 1e-300);
                        // once two checks have shown the (roughly geometric) decay of the largest residual, half of the steps
                        // it still needs - at most 3 - are skipped before looking again
                        next_check = j + 1;
                        if (have_seen && worst > 1.0 && seen_worst > worst && q >= q_target && theta_min > band) {
                            const double per_step = log(seen_worst / worst) / (j - seen_j);
                            next_check = j + (int)std::min(4.0, std::max(1.0, 0.5 * log(worst) / per_step));
                        }
                        have_seen = true;
                        seen_j = j;
                        seen_worst = worst;
                        near_conv = q >= q_target && theta_min > band && worst_res <= 10.0 * scale;
                        if (q >= q_target && worst_res <= scale && theta_min > band) outcome = 1;
                        else if ((j >= q + 12 || exhausted) && theta_min < band) outcome = 2;
                        else if (exhausted) outcome = 1;
                    }
                    if (outcome == 1) {
                        PF_TRY(extract());
                        done = true;
                        state = S_DONE;
                        req = Request{};
                        return PF_OK;
                    }
                    if (outcome == 2) {  // wanted eigenvalues sit inside the damped band: widen the undamped interval
                        PF_CHECK(resets < 8, PF_E_DEGENERATE, "pf_eigs_smallest: could not place the Chebyshev filter (cut %g, degree %d)", cut, p);
                        std::vector<double> est;
                        for (int col = c0; col < std::min(q, (int)theta.size()); ++col)
                            if (theta[col] > 1.5) est.push_back(cheb_inverse(theta[col], c, e, p));
                        std::sort(est.begin(), est.end());
                        cut = est.size() >= 2 ? std::max(4.0 * cut, 2.5 * est.back() * (want + 1) / (double)est.size()) : 8.0 * cut;
                        cut = std::min(cut, hi);
                        ++resets;
                        st.filter_resets += 1;
                        int32_t locked = 0;
                        PF_TRY(pf_lock_null_vectors(g, PF_OP_SYM, &locked));
                        PF_TRY(begin_filter());
                        break;
                    }
                    if (j < m_max) {
                        state = S_TOP;
                        break;
                    }
                    // ---- thick restart: dominant Ritz vectors + a buffer, then the residual vector
                    PF_CHECK(restarts < 60, PF_E_DEGENERATE, "pf_eigs_smallest: no convergence after 60 restarts");
                    ritz_sorted(H, m_max, j, theta, U);
                    {
                        const int n_keep = std::min(std::min(q_target + std::max(4, q_target / 2), j), j - 1);
                        std::vector<double> Y((size_t)j * n_keep), bn((size_t)m_max, 0.0);
                        for (int i = 0; i < j; ++i)
                            for (int col = 0; col < n_keep; ++col) Y[(size_t)i * n_keep + col] = U[(size_t)i * j + col];
                        PF_TRY(pf_combine(g, A0, j, Y.data(), n_keep, B0));
                        PF_TRY(pf_ws_copy(g, A0 + j, B0 + n_keep, 1));  // the residual vector follows the kept block
                        PF_TRY(pf_ws_copy(g, B0, A0, n_keep + 1));
                        for (int col = 0; col < n_keep; ++col) {
                            double r = 0.0;
                            for (int i = 0; i < j; ++i) r += U[(size_t)i * j + col] * b[i];
                            bn[col] = r;
                        }
                        std::fill(H.begin(), H.end(), 0.0);
                        for (int i = 0; i < n_keep; ++i) H[(size_t)i * m_max + i] = theta[i];
                        b = bn;
                        j = n_keep;
                    }
                    ++restarts;
                    st.restarts += 1;
                    begin_expand();
                    break;
                }
                case S_DONE:
                    done = true;
                    req = Request{};
                    return PF_OK;
            }
        }
    }

    // Rayleigh-Ritz on S itself over the converged Ritz vectors
    int extract() {
        PF_CHECK(2 * q + 1 <= reg, PF_E_STATE, "pf_eigs_smallest: workspace too small for the extraction");
        {
            std::vector<double> Y((size_t)j * q);
            for (int i = 0; i < j; ++i)
                for (int col = 0; col < q; ++col) Y[(size_t)i * q + col] = U[(size_t)i * j + col];
            PF_TRY(pf_combine(g, A0, j, Y.data(), q, B0));  // Z -> region B
        }
        PF_TRY(pf_spmv_multi(g, PF_OP_SYM, B0, A0, q));  // S Z -> region A (the Krylov basis is no longer needed)
        st.matvecs += q;
        std::vector<double> G((size_t)q * q), HA((size_t)q * q), lam;
        PF_TRY(pf_gram(g, A0, q, B0, q, G.data()));  // G[i][r] = <S z_i, z_r>
        for (int a = 0; a < q; ++a)
            for (int bb = 0; bb < q; ++bb) HA[(size_t)a * q + bb] = 0.5 * (G[(size_t)a * q + bb] + G[(size_t)bb * q + a]);
        eigh_sym(HA, q, lam);  // HA <- eigenvectors (columns)
        std::vector<int> order(q);
        for (int i = 0; i < q; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int bb) { return lam[a] < lam[bb]; });
        std::vector<int> keep;
        for (int i : order)
            if (lam[i] > 1e-10) keep.push_back(i);  // graph.py:381
        st.n_null = q - (int)keep.size();
        const int nk = std::min((int)keep.size(), (int)n_wanted);
        const int X0 = B0 + q, AX0 = A0 + q;
        PF_CHECK(X0 + nk <= 2 * reg && AX0 + nk <= reg, PF_E_STATE, "pf_eigs_smallest: workspace too small for the extraction");
        n_out = nk;
        first_slot = X0;
        if (nk > 0) {
            std::vector<double> Rk((size_t)q * nk);
            for (int r = 0; r < q; ++r)
                for (int cidx = 0; cidx < nk; ++cidx) Rk[(size_t)r * nk + cidx] = HA[(size_t)r * q + keep[cidx]];
            PF_TRY(pf_combine(g, B0, q, Rk.data(), nk, X0));
            PF_TRY(pf_combine(g, A0, q, Rk.data(), nk, AX0));
            for (int i = 0; i < nk; ++i) vals[i] = lam[keep[i]];
            PF_TRY(pf_resnorms(g, AX0, X0, vals.data(), nk, residuals.data()));
            for (int i = 0; i < nk; ++i) st.max_residual = std::max(st.max_residual, residuals[i]);
        }
        return PF_OK;
    }

    void cheb_args(int32_t* ci, double* cd) const {
        ci[0] = PF_OP_SYM, ci[1] = req.cheb_src, ci[2] = req.cheb_dst, ci[3] = p;
        cd[0] = c, cd[1] = e, cd[2] = 1.0;
    }
    int run_cheb() { return pf_cheb(g, PF_OP_SYM, req.cheb_src, req.cheb_dst, p, c, e, 1.0); }
    int run_orth() { return pf_orth_begin(g, req.orth_w, req.orth_first, req.orth_count, 1); }
};

int drive_single(Solver& s) {
    for (;;) {
        PF_TRY(s.advance());
        if (s.done) return PF_OK;
        switch (s.req.kind) {
            case REQ_CHEB: PF_TRY(s.run_cheb()); break;
            case REQ_ORTH: PF_TRY(s.run_orth()); break;
            case REQ_ORTH_CHEB:
                PF_TRY(s.run_orth());
                PF_TRY(s.run_cheb());
                break;
            default: break;
        }
    }
}

// the two solvers of a pair in lockstep: whatever both have pending runs in launches the two graphs share
int drive_pair(Solver& a, Solver& b) {
    if (!a.done) PF_TRY(a.advance());
    if (!b.done) PF_TRY(b.advance());
    while (!a.done || !b.done) {
        const ReqKind ka = a.done ? REQ_NONE : a.req.kind, kb = b.done ? REQ_NONE : b.req.kind;
        const bool oa = ka == REQ_ORTH || ka == REQ_ORTH_CHEB, ob = kb == REQ_ORTH || kb == REQ_ORTH_CHEB;
        if (ka == REQ_ORTH_CHEB && kb == REQ_ORTH_CHEB) {
            // one outer step of both solvers in one library call: both Gram-Schmidt steps in shared launches and, right
            // behind them, the next filter application of both
            const int32_t orth[8] = {a.req.orth_w, a.req.orth_first, a.req.orth_count, 1, b.req.orth_w, b.req.orth_first, b.req.orth_count, 1};
            int32_t ci[8];
            double cd[6];
            a.cheb_args(ci, cd);
            b.cheb_args(ci + 4, cd + 3);
            PF_TRY(pf_orth_cheb2(a.g, b.g, orth, ci, cd));
            PF_TRY(a.advance());
            PF_TRY(b.advance());
        } else if (oa || ob) {
            // Gram-Schmidt steps first (shared launches if both graphs have one); a fused request leaves its filter part
            if (oa && ob) {
                PF_TRY(pf_orth_begin2(a.g, a.req.orth_w, a.req.orth_first, a.req.orth_count, 1, b.g, b.req.orth_w, b.req.orth_first,
                                      b.req.orth_count, 1));
            } else {
                PF_TRY((oa ? a : b).run_orth());
            }
            for (Solver* s : {&a, &b}) {
                const ReqKind k = s->done ? REQ_NONE : s->req.kind;
                if (k == REQ_ORTH_CHEB) s->req.kind = REQ_CHEB;  // (its coefficients are read once the filter part is queued too)
                else if (k == REQ_ORTH) PF_TRY(s->advance());
            }
        } else if (ka == REQ_CHEB && kb == REQ_CHEB) {
            PF_TRY(pf_cheb2(a.g, PF_OP_SYM, a.req.cheb_src, a.req.cheb_dst, a.p, a.c, a.e, 1.0, b.g, PF_OP_SYM, b.req.cheb_src, b.req.cheb_dst,
                            b.p, b.c, b.e, 1.0));
            PF_TRY(a.advance());
            PF_TRY(b.advance());
        } else {
            Solver& s = ka == REQ_CHEB ? a : b;
            PF_CHECK(!s.done && s.req.kind == REQ_CHEB, PF_E_STATE, "pf_eigs_smallest2: driver out of step");
            PF_TRY(s.run_cheb());
            PF_TRY(s.advance());
        }
    }
    return PF_OK;
}

void abandon(pf_graph* g) {  // a solve was cut short: collect the Gram-Schmidt step it may have left in flight
    if (g && g->orth_pending >= 0) {
        std::vector<double> h((size_t)g->orth_pending + 1);
        double nrm = 0.0;
        (void)pf_orth_end(g, h.data(), &nrm);
    }
}

int finish(Solver& s, int32_t minmax, double* vals, double* vecs, double* residuals, int32_t* n_out, pf_eigs_stats* stats, bool async) {
    *n_out = s.n_out;
    for (int i = 0; i < s.n_out; ++i) {
        vals[i] = s.vals[(size_t)i];
        if (residuals) residuals[i] = s.residuals[(size_t)i];
    }
    if (stats) *stats = s.st;
    if (s.n_out > 0) {
        PF_TRY(pf_finalize_vectors_begin(s.g, s.first_slot, s.n_out, 1, minmax ? 1 : 0, vecs));
        if (!async) PF_TRY(pf_finalize_vectors_end(s.g));
    }
    return PF_OK;
}

}  // namespace

// A wait of the resident filter kernel that ran out (PF_E_PERSIST_TIMEOUT: the stream is drained and the path switched
// off by then) invalidates the filter applications in flight; the solve is simply repeated, one step per launch.
extern "C" int pf_eigs_smallest(pf_graph* g, int32_t n_wanted, int32_t minmax, double* vals, double* vecs, int32_t* n_out,
                                pf_eigs_stats* stats_out) {
    PF_CHECK(g && vals && vecs && n_out && n_wanted >= 1, PF_E_ARG, "pf_eigs_smallest: bad argument");
    int rc = PF_OK;
    for (int attempt = 0; attempt < 3; ++attempt) {
        Solver s;
        *n_out = 0;
        rc = s.init(g, n_wanted);
        if (rc == PF_OK) rc = drive_single(s);
        if (rc == PF_OK) rc = finish(s, minmax, vals, vecs, nullptr, n_out, stats_out, false);
        if (rc == PF_OK) rc = pf_sync(g->ctx);  // nothing of this solve is left in flight (and a late PF_E_PERSIST_TIMEOUT surfaces here)
        if (rc != PF_OK) abandon(g);
        if (rc != PF_E_PERSIST_TIMEOUT) break;
    }
    return rc;
}

extern "C" int pf_eigs_smallest2(pf_graph* ga, pf_graph* gb, int32_t n_wanted_a, int32_t n_wanted_b, int32_t minmax, int32_t async_download,
                                 double* vals_a, double* vecs_a, double* res_a, int32_t* n_out_a, pf_eigs_stats* stats_a,
                                 double* vals_b, double* vecs_b, double* res_b, int32_t* n_out_b, pf_eigs_stats* stats_b) {
    PF_CHECK(ga && gb && ga != gb && vals_a && vecs_a && n_out_a && vals_b && vecs_b && n_out_b && n_wanted_a >= 1 && n_wanted_b >= 1,
             PF_E_ARG, "pf_eigs_smallest2: bad argument");
    PF_CHECK(ga->ctx == gb->ctx, PF_E_ARG, "pf_eigs_smallest2: the two graphs must share one ctx (stream)");
    int rc = PF_OK;
    for (int attempt = 0; attempt < 3; ++attempt) {
        Solver a, b;
        *n_out_a = *n_out_b = 0;
        rc = a.init(ga, n_wanted_a);
        if (rc == PF_OK) rc = b.init(gb, n_wanted_b);
        if (rc == PF_OK) rc = drive_pair(a, b);
        // (both solves have synchronised with the stream in their extraction; a late PF_E_PERSIST_TIMEOUT of the partner's
        // last applications surfaces in pf_sync)
        if (rc == PF_OK) rc = pf_sync(ga->ctx);
        if (rc == PF_OK) rc = finish(a, minmax, vals_a, vecs_a, res_a, n_out_a, stats_a, async_download != 0);
        if (rc == PF_OK) rc = finish(b, minmax, vals_b, vecs_b, res_b, n_out_b, stats_b, async_download != 0);
        if (rc != PF_OK) {
            abandon(ga);
            abandon(gb);
        }
        if (rc != PF_E_PERSIST_TIMEOUT) break;
    }
    return rc;
}
