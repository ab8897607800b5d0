// Small dense host linear algebra of the Krylov drivers (pf_krylov.h): the projected matrices of the eigensolve are at
// most ~120 x 120, so this is plain scalar C++ - no HIP, no LAPACK - in a header that the device library
// (pf_eigs.hip) and the CPU test double of the drivers (tests/csrc/krylov_double.cpp) both include.
//
//   eigh_sym        symmetric eigenproblem: Householder tridiagonalisation + implicit QL (the classical tred2 / tql2 pair)
//   real_schur      A = Z T Z^T, T upper quasi-triangular: Householder reduction to Hessenberg form + Francis double-shift
//                   QR (the classical orthes / ortran / hqr2 scheme); eigenvalues only when Z is not asked for
//   schur_eigenvalues, schur_reorder (selected diagonal blocks moved to the top by swaps of adjacent blocks: a small
//                   Sylvester equation + a QR factorisation per swap, after Bai & Demmel), schur_eigenvectors (complex:
//                   the 2 x 2 blocks rotated to triangular form, then back substitution)
//   hessenberg_residual_factors   |last component| / ||.|| of the eigenvectors of an upper Hessenberg matrix for given
//                   eigenvalues by the O(n^2) recurrence from the bottom row - the Ritz residual estimates of an Arnoldi
//                   iteration without the O(n^3) vectors
// What the reference gets from ARPACK's dneupd inside scipy.sparse.linalg.eigs (graph.py:372) and what
// pyfocusr_amd/_krylov.py takes from numpy.linalg.eig / scipy.linalg.schur(sort=...).
// All matrices are row-major std::vector<double>, leading dimension = n unless stated.
#pragma once
#include <math.h>

#include <algorithm>
#include <complex>
#include <vector>

namespace pfd {

typedef std::complex<double> cplx;
static const double EPS = 2.220446049250313e-16;

// ---- symmetric eigenproblem.  V: n x n symmetric on entry, the eigenvectors (columns) on return; d: eigenvalues (in no
// particular order).
inline void eigh_sym(std::vector<double>& V, int n, std::vector<double>& d) {
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
    for (int l = 0; l < n; ++l) {
        tst1 = std::max(tst1, fabs(d[l]) + fabs(e[l]));
        int m = l;
        while (m < n - 1 && fabs(e[m]) > EPS * tst1) ++m;
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
            } while (fabs(e[l]) > EPS * tst1 && iter < 80);
        }
        d[l] = d[l] + f;
        e[l] = 0.0;
    }
}

// ---- reduction to upper Hessenberg form by Householder reflections, A <- Q^T A Q; Z <- Q when asked for.  Columns that
// are in Hessenberg form already (an Arnoldi matrix before its first restart) cost one scan each.
inline void hessenberg_reduce(std::vector<double>& A, int n, std::vector<double>* Z) {
    auto a = [&](int r, int c) -> double& { return A[(size_t)r * n + c]; };
    std::vector<double> ort((size_t)n, 0.0);
    std::vector<std::vector<double>> vs;  // the reflectors, for the accumulation
    std::vector<int> vm;
    for (int m = 1; m < n - 1; ++m) {
        double scale = 0.0;
        for (int i = m + 1; i < n; ++i) scale += fabs(a(i, m - 1));  // (below the subdiagonal only: nothing to do if zero)
        if (scale == 0.0) continue;
        scale += fabs(a(m, m - 1));
        double h = 0.0;
        for (int i = n - 1; i >= m; --i) {
            ort[i] = a(i, m - 1) / scale;
            h += ort[i] * ort[i];
        }
        double g = sqrt(h);
        if (ort[m] > 0) g = -g;
        h -= ort[m] * g;
        ort[m] -= g;
        // (I - u u^T / h) A (I - u u^T / h)
        for (int j = m; j < n; ++j) {
            double f = 0.0;
            for (int i = n - 1; i >= m; --i) f += ort[i] * a(i, j);
            f /= h;
            for (int i = m; i < n; ++i) a(i, j) -= f * ort[i];
        }
        for (int i = 0; i < n; ++i) {
            double f = 0.0;
            for (int j = n - 1; j >= m; --j) f += ort[j] * a(i, j);
            f /= h;
            for (int j = m; j < n; ++j) a(i, j) -= f * ort[j];
        }
        a(m, m - 1) = scale * g;
        for (int i = m + 1; i < n; ++i) a(i, m - 1) = 0.0;
        if (Z) {
            std::vector<double> u((size_t)n, 0.0);
            for (int i = m; i < n; ++i) u[i] = ort[i] / sqrt(h);  // I - u u^T with this scaling
            vs.push_back(u);
            vm.push_back(m);
        }
    }
    if (Z) {
        Z->assign((size_t)n * n, 0.0);
        for (int i = 0; i < n; ++i) (*Z)[(size_t)i * n + i] = 1.0;
        for (size_t t = 0; t < vs.size(); ++t) {  // Z = P_1 P_2 ... : apply from the right in order
            const std::vector<double>& u = vs[t];
            const int m = vm[t];
            for (int i = 0; i < n; ++i) {
                double f = 0.0;
                for (int j = m; j < n; ++j) f += (*Z)[(size_t)i * n + j] * u[j];
                for (int j = m; j < n; ++j) (*Z)[(size_t)i * n + j] -= f * u[j];
            }
        }
    }
}

// ---- real Schur form of an upper Hessenberg matrix by Francis double-shift QR steps (the iteration of EISPACK's hqr2
// without its back substitution).  H (n x n, entries below the subdiagonal zero) becomes T; Z (may be null: eigenvalues
// only, and the transformations then touch the active block alone) is multiplied from the right by the
// transformations.  On return T is upper quasi-triangular: a non-zero T[i+1][i] marks a 2 x 2 block with a complex
// conjugate pair; wr / wi receive the eigenvalues.  False if an eigenvalue did not converge in 60 sweeps.
inline bool hessenberg_schur(std::vector<double>& H, int nn, std::vector<double>* Zp, std::vector<double>& wr, std::vector<double>& wi) {
    auto h = [&](int r, int c) -> double& { return H[(size_t)r * nn + c]; };
    const bool wantz = Zp != nullptr;
    wr.assign((size_t)nn, 0.0);
    wi.assign((size_t)nn, 0.0);
    if (nn == 0) return true;
    double norm = 0.0;
    for (int i = 0; i < nn; ++i)
        for (int j = std::max(i - 1, 0); j < nn; ++j) norm += fabs(h(i, j));
    if (norm == 0.0) return true;
    int n = nn - 1, iter = 0;
    double exshift = 0.0, p = 0, q = 0, r = 0, s = 0, z = 0, w, x, y;
    while (n >= 0) {
        int l = n;
        while (l > 0) {  // a negligible subdiagonal entry splits the matrix
            s = fabs(h(l - 1, l - 1)) + fabs(h(l, l));
            if (s == 0.0) s = norm;
            if (fabs(h(l, l - 1)) < EPS * s) break;
            --l;
        }
        if (l > 0) h(l, l - 1) = 0.0;
        if (l == n) {  // one real eigenvalue
            h(n, n) += exshift;
            wr[n] = h(n, n);
            wi[n] = 0.0;
            --n;
            iter = 0;
        } else if (l == n - 1) {  // a pair
            w = h(n, n - 1) * h(n - 1, n);
            p = (h(n - 1, n - 1) - h(n, n)) / 2.0;
            q = p * p + w;
            z = sqrt(fabs(q));
            h(n, n) += exshift;
            h(n - 1, n - 1) += exshift;
            x = h(n, n);
            if (q >= 0) {  // two real eigenvalues: rotate the block to triangular form
                z = p >= 0 ? p + z : p - z;
                wr[n - 1] = x + z;
                wr[n] = z != 0.0 ? x - w / z : wr[n - 1];
                wi[n - 1] = wi[n] = 0.0;
                x = h(n, n - 1);
                s = fabs(x) + fabs(z);
                p = x / s;
                q = z / s;
                r = sqrt(p * p + q * q);
                p /= r;
                q /= r;
                const int jlo = n - 1, jhi = wantz ? nn - 1 : n;
                for (int j = jlo; j <= jhi; ++j) {
                    z = h(n - 1, j);
                    h(n - 1, j) = q * z + p * h(n, j);
                    h(n, j) = q * h(n, j) - p * z;
                }
                for (int i = wantz ? 0 : l; i <= n; ++i) {
                    z = h(i, n - 1);
                    h(i, n - 1) = q * z + p * h(i, n);
                    h(i, n) = q * h(i, n) - p * z;
                }
                if (wantz) {
                    std::vector<double>& Z = *Zp;
                    for (int i = 0; i < nn; ++i) {
                        z = Z[(size_t)i * nn + n - 1];
                        Z[(size_t)i * nn + n - 1] = q * z + p * Z[(size_t)i * nn + n];
                        Z[(size_t)i * nn + n] = q * Z[(size_t)i * nn + n] - p * z;
                    }
                }
                h(n, n - 1) = 0.0;
            } else {  // complex conjugate pair: the 2 x 2 block stays
                wr[n - 1] = wr[n] = x + p;
                wi[n - 1] = z;
                wi[n] = -z;
            }
            n -= 2;
            iter = 0;
        } else {
            x = h(n, n);
            y = h(n - 1, n - 1);
            w = h(n, n - 1) * h(n - 1, n);
            if (iter == 10) {  // exceptional shifts when the iteration stalls
                exshift += x;
                for (int i = 0; i <= n; ++i) h(i, i) -= x;
                s = fabs(h(n, n - 1)) + fabs(h(n - 1, n - 2));
                x = y = 0.75 * s;
                w = -0.4375 * s * s;
            }
            if (iter == 30) {
                s = (y - x) / 2.0;
                s = s * s + w;
                if (s > 0) {
                    s = sqrt(s);
                    if (y < x) s = -s;
                    s = x - w / ((y - x) / 2.0 + s);
                    for (int i = 0; i <= n; ++i) h(i, i) -= s;
                    exshift += s;
                    x = y = w = 0.964;
                }
            }
            if (++iter > 60) return false;
            int m = n - 2;  // two consecutive small subdiagonal entries: start the sweep there
            while (m >= l) {
                z = h(m, m);
                r = x - z;
                s = y - z;
                p = (r * s - w) / h(m + 1, m) + h(m, m + 1);
                q = h(m + 1, m + 1) - z - r - s;
                r = h(m + 2, m + 1);
                s = fabs(p) + fabs(q) + fabs(r);
                p /= s;
                q /= s;
                r /= s;
                if (m == l) break;
                if (fabs(h(m, m - 1)) * (fabs(q) + fabs(r)) < EPS * (fabs(p) * (fabs(h(m - 1, m - 1)) + fabs(z) + fabs(h(m + 1, m + 1))))) break;
                --m;
            }
            for (int i = m + 2; i <= n; ++i) {
                h(i, i - 2) = 0.0;
                if (i > m + 2) h(i, i - 3) = 0.0;
            }
            for (int k = m; k <= n - 1; ++k) {  // the double QR step on rows l..n and columns m..n
                const bool notlast = k != n - 1;
                if (k != m) {
                    p = h(k, k - 1);
                    q = h(k + 1, k - 1);
                    r = notlast ? h(k + 2, k - 1) : 0.0;
                    x = fabs(p) + fabs(q) + fabs(r);
                    if (x == 0.0) continue;
                    p /= x;
                    q /= x;
                    r /= x;
                }
                s = sqrt(p * p + q * q + r * r);
                if (p < 0) s = -s;
                if (s != 0.0) {
                    if (k != m) h(k, k - 1) = -s * x;
                    else if (l != m) h(k, k - 1) = -h(k, k - 1);
                    p += s;
                    x = p / s;
                    y = q / s;
                    z = r / s;
                    q /= p;
                    r /= p;
                    const int jhi = wantz ? nn - 1 : n;
                    for (int j = k; j <= jhi; ++j) {
                        p = h(k, j) + q * h(k + 1, j);
                        if (notlast) {
                            p += r * h(k + 2, j);
                            h(k + 2, j) -= p * z;
                        }
                        h(k, j) -= p * x;
                        h(k + 1, j) -= p * y;
                    }
                    const int ihi = std::min(n, k + 3);
                    for (int i = wantz ? 0 : l; i <= ihi; ++i) {
                        p = x * h(i, k) + y * h(i, k + 1);
                        if (notlast) {
                            p += z * h(i, k + 2);
                            h(i, k + 2) -= p * r;
                        }
                        h(i, k) -= p;
                        h(i, k + 1) -= p * q;
                    }
                    if (wantz) {
                        std::vector<double>& Z = *Zp;
                        for (int i = 0; i < nn; ++i) {
                            double* zr = &Z[(size_t)i * nn];
                            p = x * zr[k] + y * zr[k + 1];
                            if (notlast) {
                                p += z * zr[k + 2];
                                zr[k + 2] -= p * r;
                            }
                            zr[k] -= p;
                            zr[k + 1] -= p * q;
                        }
                    }
                }
            }
        }
    }
    if (wantz)  // what the sweeps left below the quasi-triangle is rounding noise of entries that were set to zero
        for (int i = 2; i < nn; ++i)
            for (int j = 0; j < i - 1; ++j) h(i, j) = 0.0;
    return true;
}

// A (n x n) -> T in place, Z: A = Z T Z^T.  Eigenvalues in wr / wi.
inline bool real_schur(std::vector<double>& A, int n, std::vector<double>& Z, std::vector<double>& wr, std::vector<double>& wi) {
    hessenberg_reduce(A, n, &Z);
    return hessenberg_schur(A, n, &Z, wr, wi);
}

// eigenvalues only (A is destroyed)
inline bool eigenvalues(std::vector<double>& A, int n, std::vector<double>& wr, std::vector<double>& wi) {
    hessenberg_reduce(A, n, nullptr);
    return hessenberg_schur(A, n, nullptr, wr, wi);
}

// eigenvalues of the diagonal blocks of a quasi-triangular T (a 2 x 2 block is recognised by its non-zero subdiagonal)
inline void schur_eigenvalues(const std::vector<double>& T, int n, std::vector<cplx>& ev) {
    ev.assign((size_t)n, cplx(0.0, 0.0));
    int i = 0;
    while (i < n) {
        if (i + 1 < n && T[(size_t)(i + 1) * n + i] != 0.0) {
            const double a = T[(size_t)i * n + i], b = T[(size_t)i * n + i + 1], c = T[(size_t)(i + 1) * n + i], d = T[(size_t)(i + 1) * n + i + 1];
            const double p = 0.5 * (a - d), q = p * p + b * c, m = 0.5 * (a + d);
            if (q >= 0) {
                const double z = sqrt(q);
                ev[i] = cplx(m + z, 0.0);
                ev[i + 1] = cplx(m - z, 0.0);
            } else {
                const double z = sqrt(-q);
                ev[i] = cplx(m, z);
                ev[i + 1] = cplx(m, -z);
            }
            i += 2;
        } else {
            ev[i] = cplx(T[(size_t)i * n + i], 0.0);
            i += 1;
        }
    }
}

namespace detail {

// solve the k x k system M x = rhs (k <= 4) by Gaussian elimination with complete pivoting; false when a pivot vanishes
inline bool solve_small(double* M, double* rhs, int k, double* x) {
    int colp[4] = {0, 1, 2, 3};
    for (int s = 0; s < k; ++s) {
        int pr = s, pc = s;
        double best = 0.0;
        for (int i = s; i < k; ++i)
            for (int j = s; j < k; ++j)
                if (fabs(M[i * k + j]) > best) best = fabs(M[i * k + j]), pr = i, pc = j;
        if (best == 0.0) return false;
        if (pr != s) {
            for (int j = 0; j < k; ++j) std::swap(M[s * k + j], M[pr * k + j]);
            std::swap(rhs[s], rhs[pr]);
        }
        if (pc != s) {
            for (int i = 0; i < k; ++i) std::swap(M[i * k + s], M[i * k + pc]);
            std::swap(colp[s], colp[pc]);
        }
        for (int i = s + 1; i < k; ++i) {
            const double f = M[i * k + s] / M[s * k + s];
            for (int j = s; j < k; ++j) M[i * k + j] -= f * M[s * k + j];
            rhs[i] -= f * rhs[s];
        }
    }
    double y[4];
    for (int i = k - 1; i >= 0; --i) {
        double v = rhs[i];
        for (int j = i + 1; j < k; ++j) v -= M[i * k + j] * y[j];
        y[i] = v / M[i * k + i];
    }
    for (int i = 0; i < k; ++i) x[colp[i]] = y[i];
    return true;
}

}  // namespace detail

// Swap the adjacent diagonal blocks of sizes n1, n2 (1 or 2 each) that start at row j1 of the quasi-triangular T
// (ld n), updating Z (rows x n, may be null).  The orthogonal Q of the swap comes from the QR factorisation of
// [-X; s I], X the solution of T11 X - X T22 = s T12.  False (nothing changed) when the swap would perturb T by more than
// rounding (eigenvalues of the two blocks too close to tell apart).
inline bool schur_swap(std::vector<double>& T, int n, std::vector<double>* Z, int zrows, int j1, int n1, int n2) {
    const int N = n1 + n2;
    auto t = [&](int r, int c) -> double& { return T[(size_t)r * n + c]; };
    double D[16], dnorm = 0.0;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            D[i * N + j] = t(j1 + i, j1 + j);
            dnorm = std::max(dnorm, fabs(D[i * N + j]));
        }
    if (dnorm == 0.0) return true;
    // Sylvester equation by its Kronecker form: unknowns X (n1 x n2) row-major
    const int k = n1 * n2;
    double M[16] = {0}, rhs[4], X[4];
    for (int i = 0; i < n1; ++i)
        for (int j = 0; j < n2; ++j) {
            const int row = i * n2 + j;
            for (int a = 0; a < n1; ++a) M[row * k + a * n2 + j] += D[i * N + a];                   // T11 X
            for (int b = 0; b < n2; ++b) M[row * k + i * n2 + b] -= D[(n1 + b) * N + (n1 + j)];     // - X T22
            rhs[row] = D[i * N + n1 + j];
        }
    if (!detail::solve_small(M, rhs, k, X)) return false;
    double xnorm = 0.0;
    for (int i = 0; i < k; ++i) xnorm = std::max(xnorm, fabs(X[i]));
    const double scale = 1.0 / std::max(1.0, xnorm);  // [-X; I] scaled so that its entries stay <= 1
    // QR of the N x n2 matrix G = [-scale X; scale I] by Householder reflections; Q (N x N) accumulated explicitly
    double G[8], Q[16];
    for (int i = 0; i < n1; ++i)
        for (int j = 0; j < n2; ++j) G[i * n2 + j] = -scale * X[i * n2 + j];
    for (int i = 0; i < n2; ++i)
        for (int j = 0; j < n2; ++j) G[(n1 + i) * n2 + j] = i == j ? scale : 0.0;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) Q[i * N + j] = i == j ? 1.0 : 0.0;
    for (int c = 0; c < n2; ++c) {
        double u[4] = {0, 0, 0, 0}, nrm = 0.0;
        for (int i = c; i < N; ++i) nrm += G[i * n2 + c] * G[i * n2 + c];
        nrm = sqrt(nrm);
        if (nrm == 0.0) continue;
        const double alpha = G[c * n2 + c] > 0 ? -nrm : nrm;
        for (int i = c; i < N; ++i) u[i] = G[i * n2 + c];
        u[c] -= alpha;
        double un = 0.0;
        for (int i = c; i < N; ++i) un += u[i] * u[i];
        if (un == 0.0) continue;
        for (int j = 0; j < n2; ++j) {  // G <- (I - 2 u u^T / un) G
            double f = 0.0;
            for (int i = c; i < N; ++i) f += u[i] * G[i * n2 + j];
            f *= 2.0 / un;
            for (int i = c; i < N; ++i) G[i * n2 + j] -= f * u[i];
        }
        for (int i = 0; i < N; ++i) {  // Q <- Q (I - 2 u u^T / un)
            double f = 0.0;
            for (int j = c; j < N; ++j) f += Q[i * N + j] * u[j];
            f *= 2.0 / un;
            for (int j = c; j < N; ++j) Q[i * N + j] -= f * u[j];
        }
    }
    // trial on the local block: Q^T D Q must be block upper triangular with the blocks swapped
    double QD[16], E[16];
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double v = 0.0;
            for (int a = 0; a < N; ++a) v += Q[a * N + i] * D[a * N + j];
            QD[i * N + j] = v;
        }
    double low = 0.0;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double v = 0.0;
            for (int a = 0; a < N; ++a) v += QD[i * N + a] * Q[a * N + j];
            E[i * N + j] = v;
            if (i >= n2 && j < n2) low = std::max(low, fabs(v));
        }
    if (low > 20.0 * EPS * dnorm) return false;
    // apply to the rows and columns of T and to Z
    for (int c = j1; c < n; ++c) {
        double col[4];
        for (int i = 0; i < N; ++i) {
            double v = 0.0;
            for (int a = 0; a < N; ++a) v += Q[a * N + i] * t(j1 + a, c);
            col[i] = v;
        }
        for (int i = 0; i < N; ++i) t(j1 + i, c) = col[i];
    }
    for (int r = 0; r < j1 + N; ++r) {
        double row[4];
        for (int j = 0; j < N; ++j) {
            double v = 0.0;
            for (int a = 0; a < N; ++a) v += t(r, j1 + a) * Q[a * N + j];
            row[j] = v;
        }
        for (int j = 0; j < N; ++j) t(r, j1 + j) = row[j];
    }
    for (int i = n2; i < N; ++i)
        for (int j = 0; j < n2; ++j) t(j1 + i, j1 + j) = 0.0;
    // a 2 x 2 block whose subdiagonal came out as rounding noise of a real pair keeps its entry: schur_eigenvalues
    // handles both signs of the discriminant
    if (Z) {
        std::vector<double>& Zm = *Z;
        for (int r = 0; r < zrows; ++r) {
            double row[4];
            for (int j = 0; j < N; ++j) {
                double v = 0.0;
                for (int a = 0; a < N; ++a) v += Zm[(size_t)r * n + j1 + a] * Q[a * N + j];
                row[j] = v;
            }
            for (int j = 0; j < N; ++j) Zm[(size_t)r * n + j1 + j] = row[j];
        }
    }
    return true;
}

// Move the diagonal blocks whose rows have select[i] != 0 to the top of T, keeping their relative order (what
// scipy.linalg.schur(sort=...) / LAPACK's trsen do).  `select` is indexed by the rows of T on entry (both rows of a
// 2 x 2 block carry the same flag).  Returns the dimension of the leading selected subspace; *all_moved = false when a
// swap was refused (T, Z then hold a valid Schur form in which some selected blocks stayed behind).
inline int schur_reorder(std::vector<double>& T, int n, std::vector<double>* Z, int zrows, const std::vector<char>& select, bool* all_moved) {
    int top = 0;  // rows [0, top) hold selected blocks already
    if (all_moved) *all_moved = true;
    int i = 0;
    while (i < n) {
        const int sz = (i + 1 < n && T[(size_t)(i + 1) * n + i] != 0.0) ? 2 : 1;
        if (!select[i]) {
            i += sz;
            continue;
        }
        int pos = i;  // bubble the block at `pos` up to `top`; everything in [top, i) is unselected
        bool ok = true;
        while (pos > top) {
            const int psz = (pos >= top + 2 && T[(size_t)(pos - 1) * n + pos - 2] != 0.0) ? 2 : 1;  // the block just above
            if (!schur_swap(T, n, Z, zrows, pos - psz, psz, sz)) {
                ok = false;
                break;
            }
            pos -= psz;
        }
        if (ok) top += sz;
        else if (all_moved) *all_moved = false;
        i += sz;  // (the unselected blocks between top and i moved down by sz: row i + sz is the next unvisited one)
    }
    return top;
}

// Complex eigenvectors of A = Z T Z^T for the eigenvalues at the rows `which` of T: the 2 x 2 blocks are rotated to
// (complex) triangular form, then one back substitution per vector.  V: n x which.size() column-major blocks stored
// row-major as V[r * m + c]; every vector has unit 2-norm.  ev_out receives the matching eigenvalues.
inline void schur_eigenvectors(const std::vector<double>& T, const std::vector<double>& Z, int n, const std::vector<int>& which,
                               std::vector<cplx>& V, std::vector<cplx>& ev_out) {
    std::vector<cplx> Tc((size_t)n * n), Zc((size_t)n * n);
    for (size_t i = 0; i < (size_t)n * n; ++i) Tc[i] = T[i], Zc[i] = Z[i];
    double tnorm = 0.0;
    for (size_t i = 0; i < (size_t)n * n; ++i) tnorm = std::max(tnorm, fabs(T[i]));
    for (int i = 0; i + 1 < n; ++i) {
        if (T[(size_t)(i + 1) * n + i] == 0.0) continue;
        const double a = T[(size_t)i * n + i], b = T[(size_t)i * n + i + 1], c = T[(size_t)(i + 1) * n + i], d = T[(size_t)(i + 1) * n + i + 1];
        const double p = 0.5 * (a - d), q = p * p + b * c, m = 0.5 * (a + d);
        const cplx mu = q >= 0 ? cplx(m + sqrt(q), 0.0) : cplx(m, sqrt(-q));
        // eigenvector of the block for mu: (b, mu - a) or (mu - d, c), whichever is better scaled
        cplx v0 = b, v1 = mu - a;
        if (std::abs(mu - d) + fabs(c) > std::abs(v0) + std::abs(v1)) v0 = mu - d, v1 = c;
        const double vn = sqrt(std::norm(v0) + std::norm(v1));
        v0 /= vn;
        v1 /= vn;
        // G = [v, w], w = (-conj v1, conj v0): unitary; T <- G^H T G, Z <- Z G
        const cplx g00 = v0, g10 = v1, g01 = -std::conj(v1), g11 = std::conj(v0);
        for (int col = 0; col < n; ++col) {
            const cplx x0 = Tc[(size_t)i * n + col], x1 = Tc[(size_t)(i + 1) * n + col];
            Tc[(size_t)i * n + col] = std::conj(g00) * x0 + std::conj(g10) * x1;
            Tc[(size_t)(i + 1) * n + col] = std::conj(g01) * x0 + std::conj(g11) * x1;
        }
        for (int row = 0; row < n; ++row) {
            const cplx x0 = Tc[(size_t)row * n + i], x1 = Tc[(size_t)row * n + i + 1];
            Tc[(size_t)row * n + i] = x0 * g00 + x1 * g10;
            Tc[(size_t)row * n + i + 1] = x0 * g01 + x1 * g11;
            const cplx z0 = Zc[(size_t)row * n + i], z1 = Zc[(size_t)row * n + i + 1];
            Zc[(size_t)row * n + i] = z0 * g00 + z1 * g10;
            Zc[(size_t)row * n + i + 1] = z0 * g01 + z1 * g11;
        }
        Tc[(size_t)(i + 1) * n + i] = 0.0;
        ++i;
    }
    const int m = (int)which.size();
    V.assign((size_t)n * m, cplx(0.0, 0.0));
    ev_out.assign((size_t)m, cplx(0.0, 0.0));
    std::vector<cplx> x((size_t)n);
    const double tiny = EPS * std::max(tnorm, 1e-300);
    for (int c = 0; c < m; ++c) {
        const int k = which[c];
        const cplx lam = Tc[(size_t)k * n + k];
        ev_out[c] = lam;
        std::fill(x.begin(), x.end(), cplx(0.0, 0.0));
        x[k] = 1.0;
        for (int i = k - 1; i >= 0; --i) {
            cplx s = 0.0;
            for (int j = i + 1; j <= k; ++j) s += Tc[(size_t)i * n + j] * x[j];
            cplx den = Tc[(size_t)i * n + i] - lam;
            if (std::abs(den) < tiny) den = tiny;
            x[i] = -s / den;
        }
        double nrm = 0.0;
        for (int r = 0; r < n; ++r) {
            cplx v = 0.0;
            for (int j = 0; j <= k; ++j) v += Zc[(size_t)r * n + j] * x[j];
            V[(size_t)r * m + c] = v;
            nrm += std::norm(v);
        }
        nrm = sqrt(nrm);
        if (nrm > 0.0)
            for (int r = 0; r < n; ++r) V[(size_t)r * m + c] /= nrm;
    }
}

// For an upper Hessenberg H (n x n) whose subdiagonal entries H[i][i-1], i >= first_row + 1, are non-zero, and an
// eigenvalue theta of its trailing block H[first_row:, first_row:]: |last component| / ||s||_2 of the eigenvector s of
// that block, from the recurrence that rows n-1 ... first_row+1 of (H - theta) s = 0 define once s[n-1] = 1 (it runs in
// the direction in which the components of a converged Ritz vector grow).  The Ritz residual of an Arnoldi pair is beta
// times this factor.
inline double hessenberg_residual_factor(const std::vector<double>& H, int ld, int n, int first_row, cplx theta) {
    std::vector<cplx> s((size_t)n, cplx(0.0, 0.0));
    s[n - 1] = 1.0;
    double norm2 = 1.0;
    for (int i = n - 1; i > first_row; --i) {  // row i determines s[i - 1]
        cplx acc = (H[(size_t)i * ld + i] - theta) * s[i];
        for (int j = i + 1; j < n; ++j) acc += H[(size_t)i * ld + j] * s[j];
        const double sub = H[(size_t)i * ld + i - 1];
        if (sub == 0.0) return 1.0;  // reduced matrix: no estimate
        s[i - 1] = -acc / sub;
        norm2 += std::norm(s[i - 1]);
        if (!(norm2 < 1e280)) return 0.0;
    }
    return 1.0 / sqrt(norm2);
}

}  // namespace pfd
