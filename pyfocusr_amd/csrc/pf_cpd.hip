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
    const double d = 1.0 / ((s == 0.0 ? 2.220446049250313e-16 : s) + c);
    den[n] = d;
    Pt1[n] = s * d;
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
            const double p = exp(-sqdist<D>(y, tile + r * D) * inv2s) * tden[r];
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
        if ((e = hipMemcpyAsync(h->X, X, sizeof(double) * N * D, hipMemcpyHostToDevice, st)) != hipSuccess) break;
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
