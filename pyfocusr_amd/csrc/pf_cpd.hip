// Coherent Point Drift on the spectral coordinates (reference: focusr.py:297-334 -> the third-party cycpd
// package).  The two O(M*N) pieces of the EM iteration live here, matrix-free:
//
//   pf_cpd_estep   P_mn = exp(-|x_n - ty_m|^2 / 2 sigma^2) / (sum_m' exp(..) + c) is never stored (M x N doubles =
//                  200 MB at 5000 x 5000, read twice per iteration).  Pass 1 sums the columns (den_n), pass 2 sums
//                  the rows (P1_m) and the weighted points (PX_m = sum_n P_mn x_n), both recomputing the exponential:
//                  2*M*N exp evaluations per iteration against ~0 bytes — FP64-VALU/transcendental-bound by design.
//   pf_cpd_gram    out = G(A,B) V with G_ij = exp(-|a_i - b_j|^2 / 2 beta^2): the products the low-rank
//                  eigen-decomposition of G (subspace iteration) and `transform_point_cloud` need.
//
// One thread owns one output row and walks the other set through LDS tiles (broadcast reads); the walk is cut into
// chunks across blockIdx.y to fill 256 CUs at M, N ~ 5000, partial sums land in a [chunks][rows] scratch and are
// added in chunk order — no atomics, bitwise reproducible.  D <= 16 (spectral coordinates + optional xyz).
#include <algorithm>
#include <cmath>

#include "pf_internal.h"

// No bit-exactness contract on this path (floating-point tolerance against the CPU restatement): allow FMA here,
// unlike the rest of the library, which is built with -ffp-contract=off.
#pragma clang fp contract(fast)

struct pf_cpd {
    pf_ctx* ctx = nullptr;
    int64_t N = 0, M = 0;
    int32_t D = 0;
    double* X = nullptr;   // [N][D] fixed set
    double* TY = nullptr;  // [M][D] moving set, current position
    double* den = nullptr; // [N] 1 / (column sum + c)
    double* part = nullptr; // scratch: max(chunks_m * N, chunks_n * M * (D + 1))
    double* out = nullptr;  // [N] Pt1 | [M] P1 | [M][D] PX
    int32_t chunks_m = 0, chunks_n = 0;
    // low-rank basis of the deformable model (pf_cpd_set_basis): H = Q^T diag(P1) Q per iteration
    double* Q = nullptr;      // [M][K]
    double* hpart = nullptr;  // [chunks_h][K][K]
    double* H = nullptr;      // [K][K]
    int32_t K = 0, chunks_h = 0;
    // device-resident M-step support: moment sums of the E-step results (pf_cpd_affine_sums / pf_cpd_deform_sums /
    // pf_cpd_variance_sums) and the transform kernels (pf_cpd_apply_affine / pf_cpd_apply_deform)
    double* Y = nullptr;      // [M][D] moving set as given (TY is its transformed copy)
    double* shift = nullptr;  // [3][16]: cx (mean of X), cy (mean of Y), zeros
    double* mpart = nullptr;  // per-block partial sums of the moment kernels
    double* msum = nullptr;   // their totals (device), also the staging area of small uploads
    int64_t mpart_cap = 0, msum_cap = 0;
};

namespace {

constexpr int CPD_TILE = 128;   // points of the walked set per LDS tile
constexpr int CPD_CHUNK = 128;  // points of the walked set per block (blockIdx.y): 5000 x 5000 -> 20 x 40 blocks = 3200
                                // waves; with 1024-point chunks the 400 waves left most SIMDs idle behind exp()'s latency
constexpr int GRAM_COLS = 8;     // columns of V per thread

inline unsigned nblk(int64_t n) { return (unsigned)((n + PF_BLOCK - 1) / PF_BLOCK); }

template <int D>
__device__ __forceinline__ double sqdist(const double (&a)[D], const double* __restrict__ b) {
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const double df = a[c] - b[c];
        s += df * df;
    }
    return s;
}

// pass 1: part[chunk][n] = sum over the chunk's m of exp(-|x_n - ty_m|^2 * inv2s)
template <int D>
__global__ __launch_bounds__(PF_BLOCK) void k_cpd_colsum(const double* __restrict__ X, int64_t N, const double* __restrict__ TY,
                                                         int64_t M, double inv2s, double* __restrict__ part) {
    __shared__ double tile[CPD_TILE * D];
    const int64_t n = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    const int64_t nn = n < N ? n : N - 1;
    double x[D];
#pragma unroll
    for (int c = 0; c < D; ++c) x[c] = X[nn * D + c];
    const int64_t m0 = (int64_t)blockIdx.y * CPD_CHUNK, m1 = m0 + CPD_CHUNK < M ? m0 + CPD_CHUNK : M;
    double acc = 0.0;
    for (int64_t t0 = m0; t0 < m1; t0 += CPD_TILE) {
        const int cnt = (int)(m1 - t0 < CPD_TILE ? m1 - t0 : CPD_TILE);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt * D; k += PF_BLOCK) tile[k] = TY[t0 * D + k];
        __syncthreads();
        for (int r = 0; r < cnt; ++r) acc += exp(-sqdist<D>(x, tile + r * D) * inv2s);
    }
    if (n < N) part[(int64_t)blockIdx.y * N + n] = acc;
}

// den[n] <- 1 / ((sum == 0 ? eps : sum) + c),  Pt1[n] = sum * den[n]
__global__ __launch_bounds__(PF_BLOCK) void k_cpd_colfinish(const double* __restrict__ part, int chunks, int64_t N, double c,
                                                            double* __restrict__ den, double* __restrict__ Pt1) {
    const int64_t n = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (n >= N) return;
    double s = 0.0;
    for (int k = 0; k < chunks; ++k) s += part[(int64_t)k * N + n];
    // den[n] > 0: the reciprocal of the column sum (+ c).  A column whose Gaussians all underflow to denormals has a
    // sum whose reciprocal overflows; it is stored as -sum instead and the row pass divides (e / sum <= 1 is exact
    // enough and finite, as in numpy's P / den).
    const double total = (s == 0.0 ? 2.220446049250313e-16 : s) + c;
    den[n] = total >= 1e-280 ? 1.0 / total : -total;
    Pt1[n] = s / total;
}

// pass 2: part[chunk][m][0] = sum_n P_mn, part[chunk][m][1 + c] = sum_n P_mn x_nc over the chunk's n
template <int D>
__global__ __launch_bounds__(PF_BLOCK) void k_cpd_rowsum(const double* __restrict__ X, int64_t N, const double* __restrict__ TY,
                                                         int64_t M, double inv2s, const double* __restrict__ den,
                                                         double* __restrict__ part) {
    __shared__ double tile[CPD_TILE * D];
    __shared__ double tden[CPD_TILE];
    const int64_t m = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    const int64_t mm = m < M ? m : M - 1;
    double y[D];
#pragma unroll
    for (int c = 0; c < D; ++c) y[c] = TY[mm * D + c];
    const int64_t n0 = (int64_t)blockIdx.y * CPD_CHUNK, n1 = n0 + CPD_CHUNK < N ? n0 + CPD_CHUNK : N;
    double p1 = 0.0, px[D];
#pragma unroll
    for (int c = 0; c < D; ++c) px[c] = 0.0;
    for (int64_t t0 = n0; t0 < n1; t0 += CPD_TILE) {
        const int cnt = (int)(n1 - t0 < CPD_TILE ? n1 - t0 : CPD_TILE);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt * D; k += PF_BLOCK) tile[k] = X[t0 * D + k];
        for (int k = threadIdx.x; k < cnt; k += PF_BLOCK) tden[k] = den[t0 + k];
        __syncthreads();
        for (int r = 0; r < cnt; ++r) {
            const double ev = exp(-sqdist<D>(y, tile + r * D) * inv2s), dn = tden[r];
            const double p = dn > 0.0 ? ev * dn : ev / -dn;
            p1 += p;
#pragma unroll
            for (int c = 0; c < D; ++c) px[c] += p * tile[r * D + c];
        }
    }
    if (m < M) {
        double* o = part + ((int64_t)blockIdx.y * M + m) * (D + 1);
        o[0] = p1;
#pragma unroll
        for (int c = 0; c < D; ++c) o[1 + c] = px[c];
    }
}

__global__ __launch_bounds__(PF_BLOCK) void k_cpd_rowfinish(const double* __restrict__ part, int chunks, int64_t M, int D,
                                                            double* __restrict__ P1, double* __restrict__ PX) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;  // over M * (D + 1)
    if (i >= M * (D + 1)) return;
    double s = 0.0;
    for (int k = 0; k < chunks; ++k) s += part[(int64_t)k * M * (D + 1) + i];
    const int64_t m = i / (D + 1);
    const int c = (int)(i - m * (D + 1));
    if (c == 0) P1[m] = s;
    else PX[m * D + c - 1] = s;
}

// out[i][col0 .. col0+GRAM_COLS) = sum_j exp(-|a_i - b_j|^2 * inv2b) V[j][col]
template <int D>
__global__ __launch_bounds__(PF_BLOCK) void k_gram(const double* __restrict__ A, int64_t n_a, const double* __restrict__ B,
                                                   int64_t n_b, double inv2b, const double* __restrict__ V, int32_t C,
                                                   double* __restrict__ out) {
    __shared__ double tile[CPD_TILE * D];
    __shared__ double tv[CPD_TILE * GRAM_COLS];
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    const int64_t ii = i < n_a ? i : n_a - 1;
    const int col0 = blockIdx.y * GRAM_COLS;
    double a[D];
#pragma unroll
    for (int c = 0; c < D; ++c) a[c] = A[ii * D + c];
    double acc[GRAM_COLS];
#pragma unroll
    for (int c = 0; c < GRAM_COLS; ++c) acc[c] = 0.0;
    for (int64_t t0 = 0; t0 < n_b; t0 += CPD_TILE) {
        const int cnt = (int)(n_b - t0 < CPD_TILE ? n_b - t0 : CPD_TILE);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt * D; k += PF_BLOCK) tile[k] = B[t0 * D + k];
        for (int k = threadIdx.x; k < cnt * GRAM_COLS; k += PF_BLOCK) {
            const int r = k / GRAM_COLS, c = k - r * GRAM_COLS;
            tv[k] = col0 + c < C ? V[(t0 + r) * C + col0 + c] : 0.0;
        }
        __syncthreads();
        for (int r = 0; r < cnt; ++r) {
            const double g = exp(-sqdist<D>(a, tile + r * D) * inv2b);
#pragma unroll
            for (int c = 0; c < GRAM_COLS; ++c) acc[c] += g * tv[r * GRAM_COLS + c];
        }
    }
    if (i < n_a) {
#pragma unroll
        for (int c = 0; c < GRAM_COLS; ++c)
            if (col0 + c < C) out[i * C + col0 + c] = acc[c];
    }
}

// hpart[chunk][i][j] = sum over the chunk's m of w_m Q[m][i] Q[m][j]; 16 x 16 outputs per block
constexpr int GRAM_TILE = 16;
constexpr int GRAM_CHUNK = 512;
__global__ __launch_bounds__(GRAM_TILE* GRAM_TILE) void k_weighted_gram(const double* __restrict__ Q, const double* __restrict__ w,
                                                                         int64_t M, int32_t K, double* __restrict__ part) {
    __shared__ double qi[GRAM_TILE][GRAM_TILE + 1], qj[GRAM_TILE][GRAM_TILE + 1];  // [m within step][column]
    const int tx = threadIdx.x % GRAM_TILE, ty = threadIdx.x / GRAM_TILE;
    const int i0 = blockIdx.y * GRAM_TILE, j0 = blockIdx.x * GRAM_TILE;
    const int64_t m0 = (int64_t)blockIdx.z * GRAM_CHUNK, m1 = m0 + GRAM_CHUNK < M ? m0 + GRAM_CHUNK : M;
    double acc = 0.0;
    for (int64_t mb = m0; mb < m1; mb += GRAM_TILE) {
        const int64_t m = mb + ty;  // thread (ty, tx) stages row m, columns i0 + tx and j0 + tx
        const bool ok = m < m1;
        __syncthreads();
        qi[ty][tx] = (ok && i0 + tx < K) ? Q[m * K + i0 + tx] * w[m] : 0.0;
        qj[ty][tx] = (ok && j0 + tx < K) ? Q[m * K + j0 + tx] : 0.0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < GRAM_TILE; ++r) acc += qi[r][ty] * qj[r][tx];
    }
    if (i0 + ty < K && j0 + tx < K) part[((int64_t)blockIdx.z * K + i0 + ty) * K + j0 + tx] = acc;
}

__global__ __launch_bounds__(PF_BLOCK) void k_sum_chunks(const double* __restrict__ part, int chunks, int64_t n, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int k = 0; k < chunks; ++k) s += part[(int64_t)k * n + i];
    out[i] = s;
}

// ---- moment sums for the device-resident M-steps.  One block per MOM_ROWS rows: the rows' data go to LDS, thread t
// sums output t (, t + 256, ...) over them; block partials are added in block order by k_sum_chunks.
constexpr int MOM_ROWS = 64;

// m side, L = 1 + 2 D + 2 D^2 outputs with xc = x - cx, yc = y - cy, PXc_m = PX_m - P1_m cx:
//   [0] sum P1 | [1..] sum PXc[d] | sum P1 yc[d] | sum PXc[d] yc[e] | sum P1 yc[d] yc[e]
__global__ __launch_bounds__(PF_BLOCK) void k_affine_moments_m(const double* __restrict__ P1, const double* __restrict__ PX,
                                                               const double* __restrict__ Y, const double* __restrict__ cx,
                                                               const double* __restrict__ cy, int64_t M, int D,
                                                               double* __restrict__ part) {
    __shared__ double sp[MOM_ROWS], spx[MOM_ROWS * 16], sy[MOM_ROWS * 16];
    const int64_t m0 = (int64_t)blockIdx.x * MOM_ROWS;
    const int cnt = (int)(M - m0 < MOM_ROWS ? M - m0 : MOM_ROWS);
    for (int k = threadIdx.x; k < cnt; k += PF_BLOCK) sp[k] = P1[m0 + k];
    for (int k = threadIdx.x; k < cnt * D; k += PF_BLOCK) {
        const int r = k / D, d = k - r * D;
        spx[k] = PX[m0 * D + k] - P1[m0 + r] * cx[d];
        sy[k] = Y[m0 * D + k] - cy[d];
    }
    __syncthreads();
    const int L = 1 + 2 * D + 2 * D * D;
    for (int o = threadIdx.x; o < L; o += PF_BLOCK) {
        double acc = 0.0;
        if (o == 0) {
            for (int r = 0; r < cnt; ++r) acc += sp[r];
        } else if (o < 1 + D) {
            for (int r = 0; r < cnt; ++r) acc += spx[r * D + o - 1];
        } else if (o < 1 + 2 * D) {
            for (int r = 0; r < cnt; ++r) acc += sp[r] * sy[r * D + o - 1 - D];
        } else if (o < 1 + 2 * D + D * D) {
            const int q = o - 1 - 2 * D, d = q / D, e = q - d * D;
            for (int r = 0; r < cnt; ++r) acc += spx[r * D + d] * sy[r * D + e];
        } else {
            const int q = o - 1 - 2 * D - D * D, d = q / D, e = q - d * D;
            for (int r = 0; r < cnt; ++r) acc += sp[r] * sy[r * D + d] * sy[r * D + e];
        }
        part[(int64_t)blockIdx.x * L + o] = acc;
    }
}

// n side, L = 2 + D outputs with xc = x - shift: [0] sum Pt1 | [1] sum Pt1 |xc|^2 | [2..] sum Pt1 xc[d]
__global__ __launch_bounds__(PF_BLOCK) void k_moments_n(const double* __restrict__ Pt1, const double* __restrict__ X,
                                                        const double* __restrict__ shift, int64_t N, int D, double* __restrict__ part) {
    __shared__ double sp[MOM_ROWS], sx[MOM_ROWS * 16];
    const int64_t n0 = (int64_t)blockIdx.x * MOM_ROWS;
    const int cnt = (int)(N - n0 < MOM_ROWS ? N - n0 : MOM_ROWS);
    for (int k = threadIdx.x; k < cnt; k += PF_BLOCK) sp[k] = Pt1[n0 + k];
    for (int k = threadIdx.x; k < cnt * D; k += PF_BLOCK) sx[k] = X[n0 * D + k] - shift[k % D];
    __syncthreads();
    const int L = 2 + D;
    for (int o = threadIdx.x; o < L; o += PF_BLOCK) {
        double acc = 0.0;
        if (o == 0) {
            for (int r = 0; r < cnt; ++r) acc += sp[r];
        } else if (o == 1) {
            for (int r = 0; r < cnt; ++r) {
                double q = 0.0;
                for (int d = 0; d < D; ++d) q += sx[r * D + d] * sx[r * D + d];
                acc += sp[r] * q;
            }
        } else {
            for (int r = 0; r < cnt; ++r) acc += sp[r] * sx[r * D + o - 2];
        }
        part[(int64_t)blockIdx.x * L + o] = acc;
    }
}

// m side of the deformable variance update, 3 outputs: [0] sum P1 | [1] sum P1 |ty|^2 | [2] sum ty . PX
__global__ __launch_bounds__(PF_BLOCK) void k_variance_m(const double* __restrict__ P1, const double* __restrict__ PX,
                                                         const double* __restrict__ TY, int64_t M, int D, double* __restrict__ part) {
    __shared__ double red[3][PF_BLOCK];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    const int64_t m = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (m < M) {
        double q = 0.0, t = 0.0;
        for (int d = 0; d < D; ++d) {
            const double ty = TY[m * D + d];
            q += ty * ty;
            t += ty * PX[m * D + d];
        }
        a0 = P1[m], a1 = P1[m] * q, a2 = t;
    }
    red[0][threadIdx.x] = a0, red[1][threadIdx.x] = a1, red[2][threadIdx.x] = a2;
    __syncthreads();
    for (int off = PF_BLOCK / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x < 3) part[(int64_t)blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0];
}

// R[i][d] = sum_m Q[m][i] (PX[m][d] - P1[m] Y[m][d]) over the block's rows; output o = i * D + d
__global__ __launch_bounds__(PF_BLOCK) void k_deform_rhs(const double* __restrict__ Q, const double* __restrict__ P1,
                                                         const double* __restrict__ PX, const double* __restrict__ Y, int64_t M,
                                                         int K, int D, double* __restrict__ part) {
    __shared__ double sf[MOM_ROWS * 16];
    const int64_t m0 = (int64_t)blockIdx.x * MOM_ROWS;
    const int cnt = (int)(M - m0 < MOM_ROWS ? M - m0 : MOM_ROWS);
    for (int k = threadIdx.x; k < cnt * D; k += PF_BLOCK) sf[k] = PX[m0 * D + k] - P1[m0 + k / D] * Y[m0 * D + k];
    __syncthreads();
    const int L = K * D;
    for (int o = threadIdx.x; o < L; o += PF_BLOCK) {
        const int i = o / D, d = o - i * D;
        double acc = 0.0;
        for (int r = 0; r < cnt; ++r) acc += Q[(m0 + r) * K + i] * sf[r * D + d];
        part[(int64_t)blockIdx.x * L + o] = acc;
    }
}

// TY = Y B + t
__global__ __launch_bounds__(PF_BLOCK) void k_apply_affine(const double* __restrict__ Y, const double* __restrict__ Bt /* [D][D] | [D] */,
                                                           int64_t M, int D, double* __restrict__ TY) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= M * D) return;
    const int64_t m = i / D;
    const int e = (int)(i - m * D);
    double acc = Bt[D * D + e];
    for (int d = 0; d < D; ++d) acc += Y[m * D + d] * Bt[d * D + e];
    TY[i] = acc;
}

// TY = Y + Q C,  C [K][D]
__global__ __launch_bounds__(PF_BLOCK) void k_apply_deform(const double* __restrict__ Y, const double* __restrict__ Q,
                                                           const double* __restrict__ C, int64_t M, int K, int D,
                                                           double* __restrict__ TY) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= M * D) return;
    const int64_t m = i / D;
    const int d = (int)(i - m * D);
    double acc = 0.0;
    for (int k = 0; k < K; ++k) acc += Q[m * K + k] * C[k * D + d];
    TY[i] = Y[i] + acc;
}

template <int D>
int run_estep(pf_cpd* h, double inv2s, double c) {
    hipStream_t st = h->ctx->stream;
    double *Pt1 = h->out, *P1 = h->out + h->N, *PX = h->out + h->N + h->M;
    k_cpd_colsum<D><<<dim3(nblk(h->N), h->chunks_m), PF_BLOCK, 0, st>>>(h->X, h->N, h->TY, h->M, inv2s, h->part);
    k_cpd_colfinish<<<nblk(h->N), PF_BLOCK, 0, st>>>(h->part, h->chunks_m, h->N, c, h->den, Pt1);
    k_cpd_rowsum<D><<<dim3(nblk(h->M), h->chunks_n), PF_BLOCK, 0, st>>>(h->X, h->N, h->TY, h->M, inv2s, h->den, h->part);
    k_cpd_rowfinish<<<nblk(h->M * (D + 1)), PF_BLOCK, 0, st>>>(h->part, h->chunks_n, h->M, D, P1, PX);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

template <int D>
int run_gram(hipStream_t st, const double* A, int64_t n_a, const double* B, int64_t n_b, double inv2b, const double* V, int32_t C,
             double* out) {
    k_gram<D><<<dim3(nblk(n_a), (unsigned)((C + GRAM_COLS - 1) / GRAM_COLS)), PF_BLOCK, 0, st>>>(A, n_a, B, n_b, inv2b, V, C, out);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

#define PF_DISPATCH_D(D_, CALL)                                           \
    switch (D_) {                                                         \
        case 1: return CALL(1);                                           \
        case 2: return CALL(2);                                           \
        case 3: return CALL(3);                                           \
        case 4: return CALL(4);                                           \
        case 5: return CALL(5);                                           \
        case 6: return CALL(6);                                           \
        case 7: return CALL(7);                                           \
        case 8: return CALL(8);                                           \
        case 9: return CALL(9);                                           \
        case 10: return CALL(10);                                         \
        case 11: return CALL(11);                                         \
        case 12: return CALL(12);                                         \
        case 13: return CALL(13);                                         \
        case 14: return CALL(14);                                         \
        case 15: return CALL(15);                                         \
        case 16: return CALL(16);                                         \
        default: pf_set_error("pf_cpd: d = %d out of range (1..16)", D_); \
            return PF_E_ARG;                                              \
    }

int dispatch_estep(pf_cpd* h, double inv2s, double c) {
#define CALL_E(D_) run_estep<D_>(h, inv2s, c)
    PF_DISPATCH_D(h->D, CALL_E)
#undef CALL_E
}

int dispatch_gram(hipStream_t st, int d, const double* A, int64_t n_a, const double* B, int64_t n_b, double inv2b, const double* V,
                  int32_t C, double* out) {
#define CALL_G(D_) run_gram<D_>(st, A, n_a, B, n_b, inv2b, V, C, out)
    PF_DISPATCH_D(d, CALL_G)
#undef CALL_G
}

}  // namespace

extern "C" {

void pf_cpd_free(pf_cpd* h) {
    if (!h) return;
    hipSetDevice(h->ctx->device);
    hipStreamSynchronize(h->ctx->stream);
    hipStream_t st = h->ctx->stream;
    pf_free(st, h->X);
    pf_free(st, h->TY);
    pf_free(st, h->den);
    pf_free(st, h->part);
    pf_free(st, h->out);
    pf_free(st, h->Q);
    pf_free(st, h->hpart);
    pf_free(st, h->H);
    pf_free(st, h->Y);
    pf_free(st, h->shift);
    pf_free(st, h->mpart);
    pf_free(st, h->msum);
    delete h;
}

int pf_cpd_create(pf_ctx* ctx, const double* X, int64_t N, const double* Y, int64_t M, int32_t D, pf_cpd** out) {
    PF_CHECK(ctx && X && Y && out, PF_E_ARG, "pf_cpd_create: NULL argument");
    PF_CHECK(N > 0 && M > 0 && N < ((int64_t)1 << 28) && M < ((int64_t)1 << 28) && D >= 1 && D <= 16, PF_E_ARG,
             "pf_cpd_create: N %lld, M %lld, d %d out of range (1 <= d <= 16)", (long long)N, (long long)M, D);
    PF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    pf_cpd* h = new pf_cpd();
    h->ctx = ctx;
    h->N = N, h->M = M, h->D = D;
    h->chunks_m = (int32_t)((M + CPD_CHUNK - 1) / CPD_CHUNK);
    h->chunks_n = (int32_t)((N + CPD_CHUNK - 1) / CPD_CHUNK);
    const int64_t part = std::max<int64_t>((int64_t)h->chunks_m * N, (int64_t)h->chunks_n * M * (D + 1));
    hipError_t e = hipSuccess;
    do {
        if ((e = pf_malloc(st, (void**)&h->X, sizeof(double) * N * D)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&h->TY, sizeof(double) * M * D)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&h->den, sizeof(double) * N)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&h->part, sizeof(double) * part)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&h->out, sizeof(double) * (N + M + M * D))) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&h->Y, sizeof(double) * M * D)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&h->shift, sizeof(double) * 48)) != hipSuccess) break;
        h->mpart_cap = (std::max(N, M) / MOM_ROWS + 1) * (1 + 2 * 16 + 2 * 256);
        h->msum_cap = 1 + 2 * 16 + 2 * 256 + 16;
        if ((e = pf_malloc(st, (void**)&h->mpart, sizeof(double) * h->mpart_cap)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&h->msum, sizeof(double) * h->msum_cap)) != hipSuccess) break;
        double sh[48] = {0.0};
        for (int64_t i = 0; i < N; ++i)
            for (int d = 0; d < D; ++d) sh[d] += X[i * D + d];
        for (int64_t i = 0; i < M; ++i)
            for (int d = 0; d < D; ++d) sh[16 + d] += Y[i * D + d];
        for (int d = 0; d < D; ++d) sh[d] /= (double)N, sh[16 + d] /= (double)M;
        if ((e = hipMemcpyAsync(h->shift, sh, sizeof(sh), hipMemcpyHostToDevice, st)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(h->X, X, sizeof(double) * N * D, hipMemcpyHostToDevice, st)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(h->Y, Y, sizeof(double) * M * D, hipMemcpyHostToDevice, st)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(h->TY, Y, sizeof(double) * M * D, hipMemcpyHostToDevice, st)) != hipSuccess) break;
        e = hipStreamSynchronize(st);
    } while (0);
    if (e != hipSuccess) {
        pf_set_error("pf_cpd_create: %s", hipGetErrorString(e));
        pf_cpd_free(h);
        return PF_E_HIP;
    }
    *out = h;
    return PF_OK;
}

int pf_cpd_set_basis(pf_cpd* h, const double* Q, int32_t K) {
    PF_CHECK(h && Q, PF_E_ARG, "pf_cpd_set_basis: NULL argument");
    PF_CHECK(K >= 1 && K <= 4096, PF_E_ARG, "pf_cpd_set_basis: K = %d out of range (1..4096)", K);
    PF_HIP(hipSetDevice(h->ctx->device));
    hipStream_t st = h->ctx->stream;
    pf_free(st, h->Q);
    pf_free(st, h->hpart);
    pf_free(st, h->H);
    h->Q = h->hpart = h->H = nullptr;
    h->K = 0;
    h->chunks_h = (int32_t)((h->M + GRAM_CHUNK - 1) / GRAM_CHUNK);
    PF_HIP(pf_malloc(st, (void**)&h->Q, sizeof(double) * h->M * K));
    PF_HIP(pf_malloc(st, (void**)&h->hpart, sizeof(double) * (size_t)h->chunks_h * K * K));
    PF_HIP(pf_malloc(st, (void**)&h->H, sizeof(double) * (size_t)K * K));
    PF_HIP(hipMemcpyAsync(h->Q, Q, sizeof(double) * h->M * K, hipMemcpyHostToDevice, st));
    // scratch of the device-resident M-step: K x d coefficients / right-hand side and their per-block partial sums
    const int64_t need_sum = std::max<int64_t>(h->msum_cap, (int64_t)K * h->D);
    const int64_t need_part = std::max<int64_t>(h->mpart_cap, ((h->M + MOM_ROWS - 1) / MOM_ROWS) * (int64_t)K * h->D);
    if (need_sum > h->msum_cap) {
        pf_free(st, h->msum);
        h->msum = nullptr, h->msum_cap = 0;
        PF_HIP(pf_malloc(st, (void**)&h->msum, sizeof(double) * need_sum));
        h->msum_cap = need_sum;
    }
    if (need_part > h->mpart_cap) {
        pf_free(st, h->mpart);
        h->mpart = nullptr, h->mpart_cap = 0;
        PF_HIP(pf_malloc(st, (void**)&h->mpart, sizeof(double) * need_part));
        h->mpart_cap = need_part;
    }
    PF_HIP(hipStreamSynchronize(st));
    h->K = K;
    return PF_OK;
}

int pf_cpd_weighted_gram(pf_cpd* h, double* H) {
    PF_CHECK(h && H, PF_E_ARG, "pf_cpd_weighted_gram: NULL argument");
    PF_CHECK(h->K > 0, PF_E_STATE, "pf_cpd_weighted_gram: no basis (pf_cpd_set_basis) on this handle");
    PF_HIP(hipSetDevice(h->ctx->device));
    hipStream_t st = h->ctx->stream;
    const unsigned tiles = (unsigned)((h->K + GRAM_TILE - 1) / GRAM_TILE);
    k_weighted_gram<<<dim3(tiles, tiles, (unsigned)h->chunks_h), GRAM_TILE * GRAM_TILE, 0, st>>>(h->Q, h->out + h->N, h->M, h->K, h->hpart);
    k_sum_chunks<<<nblk((int64_t)h->K * h->K), PF_BLOCK, 0, st>>>(h->hpart, h->chunks_h, (int64_t)h->K * h->K, h->H);
    PF_HIP(hipGetLastError());
    PF_HIP(hipMemcpyAsync(H, h->H, sizeof(double) * (size_t)h->K * h->K, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    return PF_OK;
}

int pf_cpd_estep(pf_cpd* h, const double* TY, double sigma2, double w, double* P1, double* Pt1, double* PX) {
    PF_CHECK(h != nullptr, PF_E_ARG, "pf_cpd_estep: NULL handle");
    PF_CHECK(sigma2 > 0.0 && std::isfinite(sigma2) && w >= 0.0 && w < 1.0, PF_E_ARG, "pf_cpd_estep: sigma2 %g, w %g out of range",
             sigma2, w);
    PF_HIP(hipSetDevice(h->ctx->device));
    hipStream_t st = h->ctx->stream;
    if (TY) PF_HIP(hipMemcpyAsync(h->TY, TY, sizeof(double) * h->M * h->D, hipMemcpyHostToDevice, st));
    const double c = std::pow(2.0 * M_PI * sigma2, 0.5 * h->D) * w / (1.0 - w) * (double)h->M / (double)h->N;
    PF_TRY(dispatch_estep(h, 1.0 / (2.0 * sigma2), c));
    if (Pt1) PF_HIP(hipMemcpyAsync(Pt1, h->out, sizeof(double) * h->N, hipMemcpyDeviceToHost, st));
    if (P1) PF_HIP(hipMemcpyAsync(P1, h->out + h->N, sizeof(double) * h->M, hipMemcpyDeviceToHost, st));
    if (PX) PF_HIP(hipMemcpyAsync(PX, h->out + h->N + h->M, sizeof(double) * h->M * h->D, hipMemcpyDeviceToHost, st));
    if (Pt1 || P1 || PX) PF_HIP(hipStreamSynchronize(st));  // device-resident loops read the results with the *_sums calls
    return PF_OK;
}

// ---- device-resident M-steps: small sums out, small parameters in; P1 / Pt1 / PX / TY never leave the device
int pf_cpd_affine_sums(pf_cpd* h, double* shifts, double* sums) {
    PF_CHECK(h && shifts && sums, PF_E_ARG, "pf_cpd_affine_sums: NULL argument");
    PF_HIP(hipSetDevice(h->ctx->device));
    hipStream_t st = h->ctx->stream;
    const int D = h->D, Lm = 1 + 2 * D + 2 * D * D, Ln = 2 + D;
    const unsigned bm = (unsigned)((h->M + MOM_ROWS - 1) / MOM_ROWS), bn = (unsigned)((h->N + MOM_ROWS - 1) / MOM_ROWS);
    double *Pt1 = h->out, *P1 = h->out + h->N, *PX = h->out + h->N + h->M;
    k_affine_moments_m<<<bm, PF_BLOCK, 0, st>>>(P1, PX, h->Y, h->shift, h->shift + 16, h->M, D, h->mpart);
    k_sum_chunks<<<nblk(Lm), PF_BLOCK, 0, st>>>(h->mpart, (int)bm, Lm, h->msum);
    PF_HIP(hipMemcpyAsync(sums, h->msum, sizeof(double) * Lm, hipMemcpyDeviceToHost, st));
    k_moments_n<<<bn, PF_BLOCK, 0, st>>>(Pt1, h->X, h->shift, h->N, D, h->mpart);
    k_sum_chunks<<<nblk(Ln), PF_BLOCK, 0, st>>>(h->mpart, (int)bn, Ln, h->msum);
    PF_HIP(hipGetLastError());
    PF_HIP(hipMemcpyAsync(sums + Lm, h->msum, sizeof(double) * Ln, hipMemcpyDeviceToHost, st));
    PF_HIP(hipMemcpyAsync(shifts, h->shift, sizeof(double) * 32, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    return PF_OK;
}

int pf_cpd_apply_affine(pf_cpd* h, const double* B, const double* t) {
    PF_CHECK(h && B && t, PF_E_ARG, "pf_cpd_apply_affine: NULL argument");
    PF_HIP(hipSetDevice(h->ctx->device));
    hipStream_t st = h->ctx->stream;
    const int D = h->D;
    PF_HIP(hipMemcpyAsync(h->msum, B, sizeof(double) * D * D, hipMemcpyHostToDevice, st));
    PF_HIP(hipMemcpyAsync(h->msum + D * D, t, sizeof(double) * D, hipMemcpyHostToDevice, st));
    k_apply_affine<<<nblk(h->M * D), PF_BLOCK, 0, st>>>(h->Y, h->msum, h->M, D, h->TY);
    PF_HIP(hipGetLastError());
    PF_HIP(hipStreamSynchronize(st));  // B and t are the caller's again
    return PF_OK;
}

int pf_cpd_deform_sums(pf_cpd* h, double* H, double* R) {
    PF_CHECK(h && H && R, PF_E_ARG, "pf_cpd_deform_sums: NULL argument");
    PF_CHECK(h->K > 0, PF_E_STATE, "pf_cpd_deform_sums: no basis (pf_cpd_set_basis) on this handle");
    PF_HIP(hipSetDevice(h->ctx->device));
    hipStream_t st = h->ctx->stream;
    const int D = h->D, K = h->K;
    double *P1 = h->out + h->N, *PX = h->out + h->N + h->M;
    const unsigned tiles = (unsigned)((K + GRAM_TILE - 1) / GRAM_TILE);
    k_weighted_gram<<<dim3(tiles, tiles, (unsigned)h->chunks_h), GRAM_TILE * GRAM_TILE, 0, st>>>(h->Q, P1, h->M, K, h->hpart);
    k_sum_chunks<<<nblk((int64_t)K * K), PF_BLOCK, 0, st>>>(h->hpart, h->chunks_h, (int64_t)K * K, h->H);
    PF_HIP(hipMemcpyAsync(H, h->H, sizeof(double) * (size_t)K * K, hipMemcpyDeviceToHost, st));
    const unsigned bm = (unsigned)((h->M + MOM_ROWS - 1) / MOM_ROWS);
    const int64_t need = (int64_t)bm * K * D;
    if (need > h->mpart_cap) {
        pf_free(st, h->mpart);
        h->mpart = nullptr, h->mpart_cap = 0;
        PF_HIP(pf_malloc(st, (void**)&h->mpart, sizeof(double) * need));
        h->mpart_cap = need;
    }
    if ((int64_t)K * D > h->msum_cap) {
        pf_free(st, h->msum);
        h->msum = nullptr, h->msum_cap = 0;
        PF_HIP(pf_malloc(st, (void**)&h->msum, sizeof(double) * K * D));
        h->msum_cap = (int64_t)K * D;
    }
    k_deform_rhs<<<bm, PF_BLOCK, 0, st>>>(h->Q, P1, PX, h->Y, h->M, K, D, h->mpart);
    k_sum_chunks<<<nblk((int64_t)K * D), PF_BLOCK, 0, st>>>(h->mpart, (int)bm, (int64_t)K * D, h->msum);
    PF_HIP(hipGetLastError());
    PF_HIP(hipMemcpyAsync(R, h->msum, sizeof(double) * K * D, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    return PF_OK;
}

int pf_cpd_apply_deform(pf_cpd* h, const double* C, double* sums) {
    PF_CHECK(h && C && sums, PF_E_ARG, "pf_cpd_apply_deform: NULL argument");
    PF_CHECK(h->K > 0, PF_E_STATE, "pf_cpd_apply_deform: no basis (pf_cpd_set_basis) on this handle");
    PF_CHECK((int64_t)h->K * h->D <= h->msum_cap, PF_E_STATE, "pf_cpd_apply_deform: scratch smaller than K x d");
    PF_HIP(hipSetDevice(h->ctx->device));
    hipStream_t st = h->ctx->stream;
    const int D = h->D, K = h->K;
    double *Pt1 = h->out, *P1 = h->out + h->N, *PX = h->out + h->N + h->M;
    PF_HIP(hipMemcpyAsync(h->msum, C, sizeof(double) * K * D, hipMemcpyHostToDevice, st));
    k_apply_deform<<<nblk(h->M * D), PF_BLOCK, 0, st>>>(h->Y, h->Q, h->msum, h->M, K, D, h->TY);
    // variance sums with the new TY and the posterior of this iteration: [Np, yPy, trPXY | sum Pt1, xPx]
    const unsigned bm = nblk(h->M), bn = (unsigned)((h->N + MOM_ROWS - 1) / MOM_ROWS);
    k_variance_m<<<bm, PF_BLOCK, 0, st>>>(P1, PX, h->TY, h->M, D, h->mpart);
    k_sum_chunks<<<1, PF_BLOCK, 0, st>>>(h->mpart, (int)bm, 3, h->msum);  // C has been consumed (stream order)
    PF_HIP(hipMemcpyAsync(sums, h->msum, sizeof(double) * 3, hipMemcpyDeviceToHost, st));
    k_moments_n<<<bn, PF_BLOCK, 0, st>>>(Pt1, h->X, h->shift + 32, h->N, D, h->mpart);
    k_sum_chunks<<<1, PF_BLOCK, 0, st>>>(h->mpart, (int)bn, 2 + D, h->msum);
    PF_HIP(hipGetLastError());
    PF_HIP(hipMemcpyAsync(sums + 3, h->msum, sizeof(double) * 2, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    return PF_OK;
}

int pf_cpd_download(pf_cpd* h, double* TY, double* P1, double* Pt1, double* PX) {
    PF_CHECK(h != nullptr, PF_E_ARG, "pf_cpd_download: NULL handle");
    PF_HIP(hipSetDevice(h->ctx->device));
    hipStream_t st = h->ctx->stream;
    if (TY) PF_HIP(hipMemcpyAsync(TY, h->TY, sizeof(double) * h->M * h->D, hipMemcpyDeviceToHost, st));
    if (Pt1) PF_HIP(hipMemcpyAsync(Pt1, h->out, sizeof(double) * h->N, hipMemcpyDeviceToHost, st));
    if (P1) PF_HIP(hipMemcpyAsync(P1, h->out + h->N, sizeof(double) * h->M, hipMemcpyDeviceToHost, st));
    if (PX) PF_HIP(hipMemcpyAsync(PX, h->out + h->N + h->M, sizeof(double) * h->M * h->D, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    return PF_OK;
}

int pf_cpd_gram(pf_ctx* ctx, const double* A, int64_t n_a, const double* B, int64_t n_b, int32_t d, double beta, const double* V,
                int32_t n_cols, double* out) {
    PF_CHECK(ctx && A && B && V && out, PF_E_ARG, "pf_cpd_gram: NULL argument");
    PF_CHECK(n_a > 0 && n_b > 0 && n_a < ((int64_t)1 << 31) && n_b < ((int64_t)1 << 31) && d >= 1 && d <= 16 && n_cols >= 1 &&
                 n_cols <= 4096 && beta > 0.0 && std::isfinite(beta),
             PF_E_ARG, "pf_cpd_gram: n_a %lld, n_b %lld, d %d, n_cols %d, beta %g out of range", (long long)n_a, (long long)n_b, d,
             n_cols, beta);
    PF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    double *dA = nullptr, *dB = nullptr, *dV = nullptr, *dO = nullptr;
    hipError_t e = hipSuccess;
    int rc = PF_OK;
    do {
        if ((e = pf_malloc(st, (void**)&dA, sizeof(double) * n_a * d)) != hipSuccess) break;
        if (A != B && (e = pf_malloc(st, (void**)&dB, sizeof(double) * n_b * d)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&dV, sizeof(double) * n_b * n_cols)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&dO, sizeof(double) * n_a * n_cols)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(dA, A, sizeof(double) * n_a * d, hipMemcpyHostToDevice, st)) != hipSuccess) break;
        if (A != B && (e = hipMemcpyAsync(dB, B, sizeof(double) * n_b * d, hipMemcpyHostToDevice, st)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(dV, V, sizeof(double) * n_b * n_cols, hipMemcpyHostToDevice, st)) != hipSuccess) break;
        rc = dispatch_gram(st, d, dA, n_a, A != B ? dB : dA, n_b, 1.0 / (2.0 * beta * beta), dV, n_cols, dO);
        if (rc != PF_OK) break;
        if ((e = hipMemcpyAsync(out, dO, sizeof(double) * n_a * n_cols, hipMemcpyDeviceToHost, st)) != hipSuccess) break;
        e = hipStreamSynchronize(st);
    } while (0);
    pf_free(st, dA);
    pf_free(st, dB);
    pf_free(st, dV);
    pf_free(st, dO);
    if (e != hipSuccess) {
        pf_set_error("pf_cpd_gram: %s", hipGetErrorString(e));
        return PF_E_HIP;
    }
    return rc;
}

}  // extern "C"
