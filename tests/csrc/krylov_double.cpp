// TEST DOUBLE (CPU) of the device side of the C++ Krylov driver - test infrastructure, never part of the product.
//
// pyfocusr_amd/csrc/pf_krylov.h (the eigensolve's driver: Lanczos / Arnoldi state machines, pair lockstep) and
// pf_dense.h (its dense algebra) are HIP-free headers; the product library drives them with the C-ABI device primitives
// (pf_eigs.hip).  This file implements the same `pfk::Ops` interface with plain loops over a CSR matrix so that
// `-m "not gpu"` tests can run the driver's logic against the oracle in a container without a GPU, and exports the
// dense routines one by one.  Built by __graft_entry__.build() / tests/conftest.py with g++ into
// tests/_build/libpf_krylov_double.so; nothing under pyfocusr_amd/ loads it.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <numeric>
#include <string>

#include "../../pyfocusr_amd/csrc/pf_krylov.h"

static thread_local std::string g_err;
void pf_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

namespace {

struct HostOps : pfk::Ops {
    int64_t nn = 0;
    std::vector<int32_t> rowptr, col;
    std::vector<double> w, deg, g, s;
    std::vector<char> isolated;
    std::vector<int32_t> label;
    std::vector<std::vector<int64_t>> comps;  // components of >= 2 vertices
    int64_t n_iso = 0;
    int32_t oneway = 0;
    bool is_sym = true;
    double bound = 2.0;
    std::vector<std::vector<double>> ws;
    std::vector<double> pend_h;
    double pend_nrm = 0.0;
    int redone_every = 0, orth_calls = 0;  // > 0: every such Gram-Schmidt step reports "redone" (the driver must repeat the filter)
    bool pend_redone = false;
    int64_t matvecs = 0, pair_calls = 0;

    void build(int64_t n, const int32_t* rp, const int32_t* ci, const double* wv) {
        nn = n;
        rowptr.assign(rp, rp + n + 1);
        col.assign(ci, ci + rp[n]);
        w.assign(wv, wv + rp[n]);
        deg.assign(n, 0.0);
        for (int64_t i = 0; i < n; ++i)
            for (int32_t k = rp[i]; k < rp[i + 1]; ++k) deg[i] += w[k];  // (left to right, like lil_matrix.sum)
        g.resize(n);
        s.resize(n);
        isolated.assign(n, 0);
        for (int64_t i = 0; i < n; ++i) {
            g[i] = 1.0 / (deg[i] + 1e-8);
            s[i] = sqrt(g[i]);
            if (rp[i + 1] == rp[i]) isolated[i] = 1;
        }
        // symmetry, one-way entries, components of the symmetrised pattern (union-find)
        std::vector<int32_t> parent(n);
        std::iota(parent.begin(), parent.end(), 0);
        auto find = [&](int32_t x) {
            while (parent[x] != x) x = parent[x] = parent[parent[x]];
            return x;
        };
        std::vector<char> has_in(n, 0);
        is_sym = true;
        oneway = 0;
        for (int64_t i = 0; i < n; ++i)
            for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
                const int32_t jn = ci[k];
                has_in[jn] = 1;
                const int32_t a = find((int32_t)i), b = find(jn);
                if (a != b) parent[a] = b;
                const int32_t* lo = std::lower_bound(ci + rp[jn], ci + rp[jn + 1], (int32_t)i);
                if (lo == ci + rp[jn + 1] || *lo != i) {
                    is_sym = false;
                    ++oneway;
                } else if (w[lo - ci] != w[k]) {
                    is_sym = false;
                }
            }
        n_iso = 0;
        for (int64_t i = 0; i < n; ++i)
            if (isolated[i]) {
                // a vertex with no outgoing entries has an all-zero row; the product's assembler counts it isolated when
                // nothing points at it either, and meshes give no other kind
                ++n_iso;
            }
        label.assign(n, -1);
        std::vector<int32_t> root_comp(n, -1);
        std::vector<int64_t> size(n, 0);
        for (int64_t i = 0; i < n; ++i) ++size[find((int32_t)i)];
        comps.clear();
        for (int64_t i = 0; i < n; ++i) {
            const int32_t r = find((int32_t)i);
            if (size[r] < 2) continue;
            if (root_comp[r] < 0) {
                root_comp[r] = (int32_t)comps.size();
                comps.emplace_back();
            }
            comps[root_comp[r]].push_back(i);
        }
    }

    int64_t n() const override { return nn; }
    int64_t n_isolated() const override { return n_iso; }
    int32_t n_components() const override { return (int32_t)comps.size(); }
    int32_t n_oneway() const override { return oneway; }
    bool symmetric() const override { return is_sym; }
    double spectral_bound() const override { return bound; }

    int ws_ensure(int32_t slots) override {
        while ((int32_t)ws.size() < slots) ws.emplace_back((size_t)nn, 0.0);
        return PF_OK;
    }
    int lock_nulls(int32_t op, int32_t* locked) override {
        ws_ensure((int32_t)comps.size() + 1);
        for (size_t c = 0; c < comps.size(); ++c) {
            std::vector<double>& v = ws[c];
            std::fill(v.begin(), v.end(), 0.0);
            double nrm = 0.0;
            for (int64_t i : comps[c]) {
                v[i] = op == PF_OP_SYM ? sqrt(deg[i] + 1e-8) : 1.0;
                nrm += v[i] * v[i];
            }
            nrm = sqrt(nrm);
            for (int64_t i : comps[c]) v[i] /= nrm;
        }
        *locked = (int32_t)comps.size();
        return PF_OK;
    }
    int orth_strict(bool) override { return PF_OK; }
    int start_vector(int32_t slot, uint64_t seed) override {
        uint64_t x = 0x9E3779B97F4A7C15ull * (seed + 1);
        auto next = [&]() {
            x += 0x9E3779B97F4A7C15ull;
            uint64_t z = x;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            return (double)((z ^ (z >> 31)) >> 11) / 9007199254740992.0;
        };
        for (int64_t i = 0; i < nn; ++i) {
            const double u1 = std::max(next(), 1e-300), u2 = next();
            ws[slot][i] = isolated[i] ? 0.0 : sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        }
        return PF_OK;
    }
    // the next orth_begin's basis: slots [first, first + split) and [first2, first2 + count - split)
    int32_t split_next = -1, first2_next = 0;
    int64_t local_calls = 0;
    bool always_twice = false;
    int orth_always_twice(bool on, bool) override {
        always_twice = on;
        return PF_OK;
    }
    bool orth_split(int32_t first2, int32_t split) override {
        split_next = split;
        first2_next = first2;
        return true;
    }
    int orth_begin(int32_t wslot, int32_t first, int32_t count) override {
        std::vector<double>& x = ws[wslot];
        const int32_t split = split_next >= 0 ? split_next : count, first2 = first2_next;
        if (split_next >= 0) ++local_calls;
        split_next = -1;
        auto slot = [&](int b) { return b < split ? first + b : first2 + (b - split); };
        pend_h.assign((size_t)std::max(count, 1), 0.0);
        // (TD_ONE_PASS: one pass unless the driver asked for two - what the device's steps usually are)
        const bool one_pass = getenv("TD_ONE_PASS") != nullptr;
        const int n_pass = one_pass && !always_twice ? 1 : 2;
        for (int pass = 0; pass < n_pass; ++pass) {
            std::vector<double> hh((size_t)count, 0.0);
            for (int b = 0; b < count; ++b) {
                const std::vector<double>& v = ws[slot(b)];
                double d = 0.0;
                for (int64_t i = 0; i < nn; ++i) d += v[i] * x[i];
                hh[b] = d;
            }
            for (int b = 0; b < count; ++b) {
                const std::vector<double>& v = ws[slot(b)];
                for (int64_t i = 0; i < nn; ++i) x[i] -= hh[b] * v[i];
                pend_h[b] += hh[b];
            }
        }
        double nrm = 0.0;
        for (int64_t i = 0; i < nn; ++i) nrm += x[i] * x[i];
        nrm = sqrt(nrm);
        if (nrm > 1e-140)
            for (int64_t i = 0; i < nn; ++i) x[i] /= nrm;
        pend_nrm = nrm;
        if (getenv("TD_DEBUG_ORTH")) {  // the orthogonality the step really left, against every slot in front of w
            double worst = 0.0;
            int at = -1;
            for (int sl = first; sl < wslot; ++sl) {
                double d = 0.0;
                for (int64_t i = 0; i < nn; ++i) d += ws[sl][i] * x[i];
                if (fabs(d) > worst) worst = fabs(d), at = sl;
            }
            fprintf(stderr, "orth w=%d count=%d local=%d: max |<w, v>| = %.2e (slot %d), nrm %.3e\n", wslot, count, (int)(split != count), worst, at, nrm);
        }
        ++orth_calls;
        pend_redone = redone_every > 0 && orth_calls % redone_every == 0;
        return PF_OK;
    }
    int orth_end(double* h, double* nrm, bool* redone) override {
        for (size_t i = 0; i < pend_h.size(); ++i) h[i] = pend_h[i];
        *nrm = pend_nrm;
        *redone = pend_redone;
        pend_redone = false;
        return PF_OK;
    }
    void apply(int32_t op, const std::vector<double>& x, std::vector<double>& y) {
        for (int64_t i = 0; i < nn; ++i) {
            double acc = 0.0;
            if (op == PF_OP_SYM) {
                for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) acc += w[k] * s[col[k]] * x[col[k]];
                y[i] = s[i] * (deg[i] * s[i] * x[i] - acc);
            } else {
                for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) acc += w[k] * x[col[k]];
                y[i] = g[i] * (deg[i] * x[i] - acc);
            }
        }
        ++matvecs;
    }
    int cheb(int32_t op, int32_t src, int32_t dst, int32_t p, double c, double e, double rho) override {
        std::vector<double> y0 = ws[src], y1((size_t)nn), t((size_t)nn);
        apply(op, y0, t);
        for (int64_t i = 0; i < nn; ++i) y1[i] = (c * y0[i] - t[i]) / (e * rho);
        for (int k = 1; k < p; ++k) {
            apply(op, y1, t);
            for (int64_t i = 0; i < nn; ++i) {
                const double v = (2.0 / (e * rho)) * (c * y1[i] - t[i]) - y0[i] / (rho * rho);
                y0[i] = y1[i];
                y1[i] = v;
            }
        }
        ws[dst] = y1;
        return PF_OK;
    }
    int combine(int32_t src_first, int32_t m, const double* Y, int32_t k, int32_t dst_first) override {
        std::vector<std::vector<double>> out((size_t)k, std::vector<double>((size_t)nn, 0.0));
        for (int a = 0; a < m; ++a)
            for (int b = 0; b < k; ++b) {
                const double y = Y[(size_t)a * k + b];
                if (y == 0.0) continue;
                const std::vector<double>& v = ws[src_first + a];
                for (int64_t i = 0; i < nn; ++i) out[b][i] += y * v[i];
            }
        for (int b = 0; b < k; ++b) ws[dst_first + b] = out[b];
        return PF_OK;
    }
    int copy(int32_t src, int32_t dst, int32_t count) override {
        std::vector<std::vector<double>> tmp(ws.begin() + src, ws.begin() + src + count);
        for (int b = 0; b < count; ++b) ws[dst + b] = tmp[b];
        return PF_OK;
    }
    int spmv_multi(int32_t op, int32_t src_first, int32_t dst_first, int32_t count) override {
        for (int b = 0; b < count; ++b) apply(op, ws[src_first + b], ws[dst_first + b]);
        return PF_OK;
    }
    int gram(int32_t fa, int32_t na, int32_t fb, int32_t nb, double* out) override {
        for (int a = 0; a < na; ++a)
            for (int b = 0; b < nb; ++b) {
                double d = 0.0;
                for (int64_t i = 0; i < nn; ++i) d += ws[fa + a][i] * ws[fb + b][i];
                out[(size_t)a * nb + b] = d;
            }
        return PF_OK;
    }
    int resnorms(int32_t ax, int32_t x, const double* lam, int32_t count, double* out) override {
        for (int b = 0; b < count; ++b) {
            double d = 0.0;
            for (int64_t i = 0; i < nn; ++i) {
                const double r = ws[ax + b][i] - lam[b] * ws[x + b][i];
                d += r * r;
            }
            out[b] = sqrt(d);
        }
        return PF_OK;
    }
    int orth_cheb_pair(pfk::Ops& other, const int32_t* orth, const int32_t* ci, const double* cd) override {
        ++pair_calls;
        return pfk::Ops::orth_cheb_pair(other, orth, ci, cd);
    }
};

void export_result(HostOps& ops, pfk::Solver& s, double* vals, double* vecs, double* residuals, int32_t* n_out, pf_eigs_stats* st) {
    *n_out = s.n_out;
    for (int i = 0; i < s.n_out; ++i) {
        vals[i] = s.vals[i];
        if (residuals) residuals[i] = s.residuals[i];
        const std::vector<double>& x = ops.ws[s.first_slot + i];
        for (int64_t r = 0; r < ops.nn; ++r) {
            // eigenvectors of L: x_L = G^1/2 x_S when the symmetrised operator was iterated (as pf_finalize_vectors does)
            vecs[(size_t)r * s.n_out + i] = ops.is_sym ? ops.s[r] * x[r] : x[r];
        }
    }
    if (st) *st = s.st;
}

}  // namespace

extern "C" {

const char* td_last_error() { return g_err.c_str(); }

// the eigensolve of one graph given CSR(W) (canonical: sorted columns, directed "set" entries).  vecs: [n][*n_out] row-major,
// eigenvectors of L (not normalised further).  info[0] = matvecs of the double, info[1] = paired calls.
int td_solve(int64_t n, const int32_t* rowptr, const int32_t* col, const double* w, int32_t n_wanted, int32_t ellipse_hint,
             int32_t redone_every, int32_t m_max_limit, double* vals, double* vecs, double* residuals, int32_t* n_out, pf_eigs_stats* stats) {
    HostOps ops;
    ops.build(n, rowptr, col, w);
    ops.redone_every = redone_every;
    pfk::Solver s;
    s.m_max_limit = m_max_limit;
    int rc = s.init(&ops, n_wanted, ellipse_hint);
    if (rc == PF_OK) rc = pfk::drive_single(s);
    if (rc == PF_OK) export_result(ops, s, vals, vecs, residuals, n_out, stats);
    return rc;
}

int td_solve_pair(int64_t na, const int32_t* rpa, const int32_t* cia, const double* wa, int32_t wanted_a, double* vals_a, double* vecs_a,
                  int32_t* n_out_a, pf_eigs_stats* st_a, int64_t nb, const int32_t* rpb, const int32_t* cib, const double* wb,
                  int32_t wanted_b, double* vals_b, double* vecs_b, int32_t* n_out_b, pf_eigs_stats* st_b, int64_t* pair_calls) {
    HostOps a, b;
    a.build(na, rpa, cia, wa);
    b.build(nb, rpb, cib, wb);
    pfk::Solver sa, sb;
    int rc = sa.init(&a, wanted_a);
    if (rc == PF_OK) rc = sb.init(&b, wanted_b);
    if (rc == PF_OK) rc = pfk::drive_pair(sa, sb);
    if (rc == PF_OK) {
        export_result(a, sa, vals_a, vecs_a, nullptr, n_out_a, st_a);
        export_result(b, sb, vals_b, vecs_b, nullptr, n_out_b, st_b);
        *pair_calls = a.pair_calls;
    }
    return rc;
}

// ---- the dense routines, one by one (row-major n x n arrays)
void td_eigh_sym(int32_t n, double* A /* in: symmetric; out: eigenvectors */, double* d) {
    std::vector<double> V(A, A + (size_t)n * n), w;
    pfd::eigh_sym(V, n, w);
    memcpy(A, V.data(), sizeof(double) * (size_t)n * n);
    memcpy(d, w.data(), sizeof(double) * (size_t)n);
}

int td_real_schur(int32_t n, double* A /* in: matrix; out: T */, double* Z, double* wr, double* wi) {
    std::vector<double> T(A, A + (size_t)n * n), Zv, r, i;
    const bool ok = pfd::real_schur(T, n, Zv, r, i);
    memcpy(A, T.data(), sizeof(double) * (size_t)n * n);
    memcpy(Z, Zv.data(), sizeof(double) * (size_t)n * n);
    memcpy(wr, r.data(), sizeof(double) * (size_t)n);
    memcpy(wi, i.data(), sizeof(double) * (size_t)n);
    return ok ? 0 : 1;
}

int td_eigenvalues(int32_t n, const double* A, double* wr, double* wi) {
    std::vector<double> T(A, A + (size_t)n * n), r, i;
    const bool ok = pfd::eigenvalues(T, n, r, i);
    memcpy(wr, r.data(), sizeof(double) * (size_t)n);
    memcpy(wi, i.data(), sizeof(double) * (size_t)n);
    return ok ? 0 : 1;
}

// T, Z in and out; select[n]; returns the leading dimension, *all_moved
int td_schur_reorder(int32_t n, double* T, double* Z, const char* select, int32_t* all_moved) {
    std::vector<double> Tv(T, T + (size_t)n * n), Zv(Z, Z + (size_t)n * n);
    std::vector<char> sel(select, select + n);
    bool ok = true;
    const int top = pfd::schur_reorder(Tv, n, &Zv, n, sel, &ok);
    memcpy(T, Tv.data(), sizeof(double) * (size_t)n * n);
    memcpy(Z, Zv.data(), sizeof(double) * (size_t)n * n);
    *all_moved = ok ? 1 : 0;
    return top;
}

// all eigenvectors of Z T Z^T: V [n][n] complex interleaved (re, im), ev [n] interleaved
void td_schur_eigenvectors(int32_t n, const double* T, const double* Z, double* V, double* ev) {
    std::vector<double> Tv(T, T + (size_t)n * n), Zv(Z, Z + (size_t)n * n);
    std::vector<int> all(n);
    std::iota(all.begin(), all.end(), 0);
    std::vector<pfd::cplx> Vc, lam;
    pfd::schur_eigenvectors(Tv, Zv, n, all, Vc, lam);
    for (size_t i = 0; i < (size_t)n * n; ++i) V[2 * i] = Vc[i].real(), V[2 * i + 1] = Vc[i].imag();
    for (int i = 0; i < n; ++i) ev[2 * i] = lam[i].real(), ev[2 * i + 1] = lam[i].imag();
}

double td_hessenberg_residual_factor(int32_t n, const double* H, int32_t first_row, double re, double im) {
    std::vector<double> Hv(H, H + (size_t)n * n);
    return pfd::hessenberg_residual_factor(Hv, n, n, first_row, pfd::cplx(re, im));
}

}  // extern "C"
