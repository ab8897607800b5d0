// pf_eigs_smallest: the whole eigensolve of a symmetric mesh graph behind ONE C call.
//
// Replaces scipy.sparse.linalg.eigs(L, k, sigma=1e-10, which="LM", ncv=4k) of the reference (graph.py:372) for callers
// that bind the C-ABI without Python: the same Chebyshev-filtered thick-restart Lanczos iteration as
// pyfocusr_amd/_krylov.py (its symmetric branch, restated here in C++ on top of the same device primitives) —
//   operator S = G^1/2 (D - W) G^1/2, B = T_p((c - S)/e) with the damped interval [cut, 2] (cut starts at
//   12 (k+1)/n and is enlarged if wanted Ritz values sit in the damped band), full CGS2 re-orthogonalisation,
//   analytic null vectors locked per connected component, isolated vertices masked, thick restart on the dominant
//   Ritz vectors, final Rayleigh-Ritz on S itself, eigenvalues > 1e-10 kept (graph.py:381).
// The Python driver remains the general one (asymmetric W, complex spectra, two graphs per launch, pipelined steps);
// this entry point refuses what it does not cover (PF_E_STATE) instead of guessing.
#include <math.h>

#include <algorithm>
#include <vector>

#include "pf_internal.h"

namespace {

// cyclic Jacobi for a symmetric m x m matrix (row-major A is destroyed); eigenvalues in w, eigenvectors in the columns of V
void jacobi_eigh(std::vector<double>& A, int m, std::vector<double>& w, std::vector<double>& V) {
    V.assign((size_t)m * m, 0.0);
    for (int i = 0; i < m; ++i) V[(size_t)i * m + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < m; ++i) {
            diag += A[(size_t)i * m + i] * A[(size_t)i * m + i];
            for (int j = i + 1; j < m; ++j) off += A[(size_t)i * m + j] * A[(size_t)i * m + j];
        }
        if (off <= 1e-32 * (diag + off)) break;
        for (int p = 0; p < m - 1; ++p) {
            for (int q = p + 1; q < m; ++q) {
                const double apq = A[(size_t)p * m + q];
                if (apq == 0.0) continue;
                const double app = A[(size_t)p * m + p], aqq = A[(size_t)q * m + q];
                const double tau = (aqq - app) / (2.0 * apq);
                const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                for (int k = 0; k < m; ++k) {  // rotate columns p, q of A and V
                    const double akp = A[(size_t)k * m + p], akq = A[(size_t)k * m + q];
                    A[(size_t)k * m + p] = c * akp - s * akq;
                    A[(size_t)k * m + q] = s * akp + c * akq;
                    const double vkp = V[(size_t)k * m + p], vkq = V[(size_t)k * m + q];
                    V[(size_t)k * m + p] = c * vkp - s * vkq;
                    V[(size_t)k * m + q] = s * vkp + c * vkq;
                }
                for (int k = 0; k < m; ++k) {  // rotate rows p, q of A
                    const double apk = A[(size_t)p * m + k], aqk = A[(size_t)q * m + k];
                    A[(size_t)p * m + k] = c * apk - s * aqk;
                    A[(size_t)q * m + k] = s * apk + c * aqk;
                }
            }
        }
    }
    w.resize(m);
    for (int i = 0; i < m; ++i) w[i] = A[(size_t)i * m + i];
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
    std::vector<double> A((size_t)j * j), w, V;
    for (int a = 0; a < j; ++a)
        for (int b = 0; b < j; ++b) A[(size_t)a * j + b] = 0.5 * (H[(size_t)a * ld + b] + H[(size_t)b * ld + a]);
    jacobi_eigh(A, j, w, V);
    std::vector<int> order(j);
    for (int i = 0; i < j; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return w[a] > w[b]; });
    theta.resize(j);
    U.assign((size_t)j * j, 0.0);
    for (int col = 0; col < j; ++col) {
        theta[col] = w[order[col]];
        for (int r = 0; r < j; ++r) U[(size_t)r * j + col] = V[(size_t)r * j + order[col]];
    }
}

int eigs_smallest_once(pf_graph* g, int32_t n_wanted, int32_t minmax, double* vals, double* vecs, int32_t* n_out,
                       pf_eigs_stats* stats_out) {
    PF_CHECK(g && vals && vecs && n_out && n_wanted >= 1, PF_E_ARG, "pf_eigs_smallest: bad argument");
    PF_CHECK(g->is_symmetric, PF_E_STATE, "pf_eigs_smallest: W is not symmetric (one-way edges): use the Python driver");
    const double hi = g->spectral_bound, strength = 2.0, tol = 1e-12;  // (2, or the face-by-face bound of a closed triangle mesh)
    const int64_t n_active = g->n - g->n_isolated;
    pf_eigs_stats st{};
    *n_out = 0;
    int32_t c0 = 0;
    // The loose single-pass criterion of the device's Gram-Schmidt step is for the filtered iteration of large graphs; on
    // small ones it can lose orthogonality (pf_orth_strict).  The setting is sticky per graph: set it here either way, so
    // that this solve does not inherit what an earlier driver left behind.
    PF_TRY(pf_orth_strict(g, n_active < 4096));
    PF_TRY(pf_lock_null_vectors(g, PF_OP_SYM, &c0));
    const int want = (int)std::min<int64_t>(n_wanted, std::max<int64_t>(n_active - c0, 0));
    if (want <= 0) {
        if (stats_out) *stats_out = st;
        return PF_OK;
    }
    const int q_target = c0 + want;
    PF_CHECK(n_active >= 4 * (q_target + 8), PF_E_STATE,
             "pf_eigs_smallest: graph too small for the filtered iteration (%lld active vertices): use the Python driver",
             (long long)n_active);
    const int m_max = (int)std::min<int64_t>(std::max(3 * q_target + 24, 48), n_active);
    const int reg = std::max(m_max + 1, 2 * q_target + 2);
    PF_TRY(pf_ws_ensure(g, 2 * reg));
    PF_TRY(pf_lock_null_vectors(g, PF_OP_SYM, &c0));  // the workspace may have moved: rewrite slots [0, c0)
    const int A0 = 0, B0 = reg;
    double cut = 12.0 * (want + 1) / (double)std::max<int64_t>(n_active, 1);
    std::vector<double> H((size_t)m_max * m_max), b(m_max), theta, U, h(m_max + 1);
    uint64_t seed = 0;
    int j = 0, q = 0, p = 0;
    double c = 0, e = 0;
    bool converged = false;
    for (int reset = 0; reset <= 8 && !converged; ++reset) {
        PF_CHECK(cut < 0.5 * hi, PF_E_STATE, "pf_eigs_smallest: the wanted eigenvalues are not a corner of the spectrum (cut %g): "
                 "use the Python driver", cut);
        choose_filter(cut, hi, strength, &c, &e, &p);
        st.degree = p;
        st.cut = cut;
        const double theta0 = cheb_value(0.0, c, e, p), band = 1.5;
        std::fill(H.begin(), H.end(), 0.0);
        std::fill(b.begin(), b.end(), 0.0);
        for (int i = 0; i < c0; ++i) H[(size_t)i * m_max + i] = theta0;
        j = c0;
        {  // start vector, orthogonal to the locked null vectors
            double nrm = 0.0;
            PF_TRY(pf_start_vector(g, A0 + j, seed++));
            PF_TRY(pf_orth(g, A0 + j, A0, j, h.data(), &nrm));
            PF_CHECK(nrm > 0.0, PF_E_DEGENERATE, "pf_eigs_smallest: start vector vanished");
            PF_TRY(pf_scale(g, A0 + j, 1.0 / nrm));
        }
        bool reset_cut = false;
        for (int restarts = 0; !converged && !reset_cut; ++restarts) {
            PF_CHECK(restarts <= 60, PF_E_DEGENERATE, "pf_eigs_smallest: no convergence after 60 restarts");
            while (j < m_max && !converged && !reset_cut) {  // ---- expand
                double beta = 0.0;
                PF_TRY(pf_cheb(g, PF_OP_SYM, A0 + j, A0 + j + 1, p, c, e, 1.0));
                st.matvecs += p;
                st.outer_steps += 1;
                PF_TRY(pf_orth(g, A0 + j + 1, A0, j + 1, h.data(), &beta));
                for (int i = 0; i <= j; ++i) H[(size_t)i * m_max + j] = h[i];
                for (int i = 0; i < j; ++i) H[(size_t)j * m_max + i] = b[i];
                ++j;
                std::fill(b.begin(), b.end(), 0.0);
                b[j - 1] = beta;
                const bool exhausted = beta <= 1e-14 * std::max(fabs(theta0), 1.0) || j >= n_active;
                if (!exhausted) PF_TRY(pf_scale(g, A0 + j, 1.0 / beta));
                if (exhausted || j == m_max || j >= q_target + 8) {
                    ritz_sorted(H, m_max, j, theta, U);
                    q = std::min(q_target, j);
                    double theta_min = theta[q - 1], worst = 0.0;
                    for (int col = 0; col < q; ++col) {
                        double r = 0.0;
                        for (int i = 0; i < j; ++i) r += b[i] * U[(size_t)i * j + col];
                        worst = std::max(worst, fabs(r));
                    }
                    if (q >= q_target && worst <= tol * std::max(theta_min, 1.0) && theta_min > band) converged = true;
                    else if ((j >= q + 12 || exhausted) && theta_min < band) reset_cut = true;
                    else if (exhausted) converged = true;
                }
            }
            if (converged || reset_cut) break;
            // ---- thick restart: dominant Ritz vectors + a buffer, then the residual vector
            ritz_sorted(H, m_max, j, theta, U);
            int n_keep = std::min(std::min(q_target + std::max(4, q_target / 2), j), j - 1);
            std::vector<double> Y((size_t)j * n_keep), bn(m_max, 0.0);
            for (int i = 0; i < j; ++i)
                for (int col = 0; col < n_keep; ++col) Y[(size_t)i * n_keep + col] = U[(size_t)i * j + col];
            PF_TRY(pf_combine(g, A0, j, Y.data(), n_keep, B0));
            PF_TRY(pf_ws_copy(g, A0 + j, B0 + n_keep, 1));
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
            st.restarts += 1;
        }
        if (reset_cut) {  // wanted eigenvalues sit inside the damped band: widen the undamped interval
            std::vector<double> est;
            for (int col = c0; col < std::min(q, (int)theta.size()); ++col)
                if (theta[col] > 1.5) est.push_back(cheb_inverse(theta[col], c, e, p));
            std::sort(est.begin(), est.end());
            cut = est.size() >= 2 ? std::max(4.0 * cut, 2.5 * est.back() * (want + 1) / (double)est.size()) : 8.0 * cut;
            cut = std::min(cut, hi);
            st.filter_resets += 1;
        }
    }
    PF_CHECK(converged, PF_E_DEGENERATE, "pf_eigs_smallest: could not place the Chebyshev filter (cut %g, degree %d)", cut, p);

    // ---- Rayleigh-Ritz on S itself over the converged Ritz vectors
    PF_CHECK(2 * q + 1 <= reg, PF_E_STATE, "pf_eigs_smallest: workspace too small for the extraction");
    {
        std::vector<double> Y((size_t)j * q);
        for (int i = 0; i < j; ++i)
            for (int col = 0; col < q; ++col) Y[(size_t)i * q + col] = U[(size_t)i * j + col];
        PF_TRY(pf_combine(g, A0, j, Y.data(), q, B0));  // Z -> region B
    }
    for (int i = 0; i < q; ++i) {
        PF_TRY(pf_spmv(g, PF_OP_SYM, B0 + i, A0 + i));
        st.matvecs += 1;
    }
    std::vector<double> HA((size_t)q * q), col(q), lam, R;
    for (int i = 0; i < q; ++i) {
        PF_TRY(pf_dots(g, A0 + i, B0, q, col.data()));
        for (int r = 0; r < q; ++r) HA[(size_t)r * q + i] = col[r];
    }
    for (int a = 0; a < q; ++a)
        for (int bb = a + 1; bb < q; ++bb) HA[(size_t)a * q + bb] = HA[(size_t)bb * q + a] = 0.5 * (HA[(size_t)a * q + bb] + HA[(size_t)bb * q + a]);
    jacobi_eigh(HA, q, lam, R);
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
    std::vector<double> Rk((size_t)q * std::max(nk, 1));
    for (int r = 0; r < q; ++r)
        for (int cidx = 0; cidx < nk; ++cidx) Rk[(size_t)r * nk + cidx] = R[(size_t)r * q + keep[cidx]];
    if (nk > 0) {
        PF_TRY(pf_combine(g, B0, q, Rk.data(), nk, X0));
        PF_TRY(pf_combine(g, A0, q, Rk.data(), nk, AX0));
        for (int i = 0; i < nk; ++i) {
            double r = 0.0;
            vals[i] = lam[keep[i]];
            PF_TRY(pf_resnorm(g, AX0 + i, X0 + i, vals[i], &r));
            st.max_residual = std::max(st.max_residual, r);
        }
        PF_TRY(pf_finalize_vectors(g, X0, nk, 1, minmax ? 1 : 0, vecs));
    }
    *n_out = nk;
    if (stats_out) *stats_out = st;
    return pf_sync(g->ctx);  // nothing of this solve is left in flight (and a late PF_E_PERSIST_TIMEOUT surfaces here)
}

}  // namespace

// A wait of the resident filter kernel that ran out (PF_E_PERSIST_TIMEOUT: the stream is drained and the path switched
// off by then) invalidates the filter applications in flight; the solve is simply repeated, one step per launch.
extern "C" int pf_eigs_smallest(pf_graph* g, int32_t n_wanted, int32_t minmax, double* vals, double* vecs, int32_t* n_out,
                                pf_eigs_stats* stats_out) {
    int rc = PF_OK;
    for (int attempt = 0; attempt < 3; ++attempt) {
        rc = eigs_smallest_once(g, n_wanted, minmax, vals, vecs, n_out, stats_out);
        if (rc != PF_E_PERSIST_TIMEOUT) break;
    }
    return rc;
}
