// eigsort's cost matrices on the device (eigsort.py:162-233 of the reference).
//
// eigsort compares the eigenmaps of two meshes on a random sample of vertices (n_rand_samples = 5000): c_hist[i][j] is the
// 1-D earth mover's distance between the sampled values of target map i and source map j (after log(v + 0.5 + eps)),
// c_spatial[i][j] the RMS-like difference between target map i at a sampled vertex and source map j at the spatially
// nearest sampled source vertex (a 3-D 1-NN on the min-max normalised sample points); both also for the flipped source
// map (-v).  On the host that is 2 k^2 numpy reductions, 3 k sorts and 3 k logs over 5000 values plus four gathers and
// their downloads: 1.3 ms per pair at 250k vertices against ~0.3 ms of launches here - the sampled rows are read where
// the eigensolve left them (the graphs' resident eigenvector blocks and point copies), and what comes back is
// 4 k^2 numbers and the 1-NN indices.
//
//   c_hist[i][j]    = W1( log(T_i + 0.5 + eps), log(+-S_j + 0.5 + eps) )      (eigsort.py:176-187, scipy's wasserstein_distance)
//                     = the area between the two step quantile functions: target order statistic a holds on
//                     (a/mt, (a+1)/mt], source order statistic b on (b/ms, (b+1)/ms] - every thread takes one a and the few
//                     b it overlaps, on the integer grid of mt*ms; for mt == ms that is the mean absolute difference of
//                     the order statistics.  log is monotone: the raw values are sorted, log(-v + c) read in reverse
//   c_spatial[i][j] = sqrt( sum_r (+-S_j[idx[r]] - T_i[r])^2 ) / m,   idx = 1-NN of target sample point r among the
//                     source sample points (eigsort.py:203-233)
// Differences to the host path: the device's log (<= 1 ulp from libm's) and the order of the sums: ~1e-15 relative
// (tests/test_gpu_parity.py::test_eigsort_costs_on_device).
#include <float.h>
#include <string.h>

#include <vector>

#include "pf_internal.h"

extern "C" int pf_knn1_blocks(pf_ctx* c, const double* ref_block, int64_t n_ref, int32_t ref_stride, const double* qry_block, int64_t n_qry,
                              int32_t qry_stride, int32_t d, const int32_t* col_ref, const double* scale_ref, const int32_t* col_qry,
                              const double* scale_qry, int64_t* idx_out, double* d2_out);

namespace {

constexpr int ES_SORT_THREADS = 1024;
constexpr int ES_MAX_SAMPLES = 16384;  // one column is sorted by one block in LDS (128 KB at the limit)
constexpr int ES_RED_THREADS = 256;

inline unsigned es_blocks(int64_t n) { return (unsigned)((n + PF_BLOCK - 1) / PF_BLOCK); }

// vals[c][r] = fin[rows[r]][col[c]] * sign[c]  (coordinate-major);  raw[r][0..3) = pts[rows[r]]
__global__ __launch_bounds__(PF_BLOCK) void k_es_gather(const double* __restrict__ fin, int32_t fc, const double* __restrict__ pts,
                                                        const int64_t* __restrict__ rows, int64_t m, int32_t k,
                                                        const int32_t* __restrict__ col, const double* __restrict__ sign,
                                                        double* __restrict__ vals, double* __restrict__ raw) {
    const int64_t e = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (e >= m * (k + 3)) return;
    const int64_t c = e / m, r = e - c * m;
    if (c < k) vals[c * m + r] = fin[rows[r] * fc + col[c]] * sign[c];
    else raw[r * 3 + (c - k)] = pts[rows[r] * 3 + (c - k)];
}

// lohi[c] = min, lohi[3 + c] = max of coordinate c over the sample (one block per coordinate)
__global__ __launch_bounds__(ES_RED_THREADS) void k_es_minmax(const double* __restrict__ raw, int64_t m, double* __restrict__ lohi) {
    __shared__ double slo[ES_RED_THREADS], shi[ES_RED_THREADS];
    const int c = blockIdx.x;
    double lo = INFINITY, hi = -INFINITY;
    for (int64_t r = threadIdx.x; r < m; r += ES_RED_THREADS) {
        const double v = raw[r * 3 + c];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    slo[threadIdx.x] = lo;
    shi[threadIdx.x] = hi;
    __syncthreads();
    for (int s = ES_RED_THREADS / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            slo[threadIdx.x] = slo[threadIdx.x + s] < slo[threadIdx.x] ? slo[threadIdx.x + s] : slo[threadIdx.x];
            shi[threadIdx.x] = shi[threadIdx.x + s] > shi[threadIdx.x] ? shi[threadIdx.x + s] : shi[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        lohi[c] = slo[0];
        lohi[3 + c] = shi[0];
    }
}

// (x - min) / (max - min) per coordinate, graph.py:269-272
__global__ __launch_bounds__(PF_BLOCK) void k_es_normalize(const double* __restrict__ raw, const double* __restrict__ lohi, int64_t m,
                                                           double* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (e >= m * 3) return;
    const int c = (int)(e % 3);
    out[e] = (raw[e] - lohi[c]) / (lohi[3 + c] - lohi[c]);
}

// block b < k: target column b, block b >= k: source column b - k.  Bitonic sort of the raw values in LDS (padded with
// +inf to a power of two), then the logs: lt / ls = log(v + 0.5 + eps) ascending, lsf = log(-v + 0.5 + eps) ascending
// (= read from the descending end).
__global__ __launch_bounds__(ES_SORT_THREADS) void k_es_sort_log(const double* __restrict__ vals_t, const double* __restrict__ vals_s,
                                                                 int64_t mt, int64_t ms, int32_t k, double* __restrict__ lt,
                                                                 double* __restrict__ ls, double* __restrict__ lsf) {
    extern __shared__ double buf[];
    const int b = blockIdx.x;
    const int64_t m = b < k ? mt : ms;
    int n_pow2 = 2;
    while (n_pow2 < m) n_pow2 <<= 1;
    const double* src = b < k ? vals_t + (int64_t)b * m : vals_s + (int64_t)(b - k) * m;
    for (int i = threadIdx.x; i < n_pow2; i += ES_SORT_THREADS) buf[i] = i < m ? src[i] : INFINITY;
    __syncthreads();
    for (int size = 2; size <= n_pow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < n_pow2 / 2; t += ES_SORT_THREADS) {
                const int pos = 2 * t - (t & (stride - 1));
                const double a = buf[pos], c = buf[pos + stride];
                const bool up = (pos & size) == 0;
                if ((a > c) == up) {
                    buf[pos] = c;
                    buf[pos + stride] = a;
                }
            }
            __syncthreads();
        }
    }
    const double eps = DBL_EPSILON;
    if (b < k) {
        for (int64_t i = threadIdx.x; i < m; i += ES_SORT_THREADS) lt[(int64_t)b * m + i] = log(buf[i] + 0.5 + eps);
    } else {
        const int64_t o = (int64_t)(b - k) * m;
        for (int64_t i = threadIdx.x; i < m; i += ES_SORT_THREADS) {
            ls[o + i] = log(buf[i] + 0.5 + eps);
            lsf[o + i] = log(-buf[m - 1 - i] + 0.5 + eps);
        }
    }
}

__device__ __forceinline__ double es_block_sum(double v, double* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = ES_RED_THREADS / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// block (i, j): out[0][i][j] = W1(lt_i, ls_j), out[1][i][j] = W1(lt_i, lsf_j) (sorted samples of mt and ms values): thread
// -> target order statistic a, which holds on [a ms, (a+1) ms) of the grid of mt*ms; the source statistics b it overlaps
__global__ __launch_bounds__(ES_RED_THREADS) void k_es_w1(const double* __restrict__ lt, const double* __restrict__ ls,
                                                          const double* __restrict__ lsf, int64_t mt, int64_t ms, int32_t k,
                                                          double* __restrict__ out) {
    __shared__ double sh[ES_RED_THREADS];
    const int i = blockIdx.x / k, j = blockIdx.x - i * k;
    const double unit = 1.0 / ((double)mt * (double)ms);
    double acc = 0.0, accf = 0.0;
    for (int64_t a = threadIdx.x; a < mt; a += ES_RED_THREADS) {
        const double t = lt[(int64_t)i * mt + a];
        const int64_t lo = a * ms, hi = lo + ms;
        for (int64_t b = lo / mt; b < ms && b * mt < hi; ++b) {
            const int64_t from = b * mt > lo ? b * mt : lo, to = (b + 1) * mt < hi ? (b + 1) * mt : hi;
            const double len = (double)(to - from) * unit;
            acc += len * fabs(t - ls[(int64_t)j * ms + b]);
            accf += len * fabs(t - lsf[(int64_t)j * ms + b]);
        }
    }
    acc = es_block_sum(acc, sh);
    accf = es_block_sum(accf, sh);
    if (threadIdx.x == 0) {
        out[i * k + j] = acc;
        out[k * k + i * k + j] = accf;
    }
}

// block (i, j): out[2][i][j] = sqrt(sum (S_j[idx] - T_i)^2) / m, out[3][i][j] the same with -S_j
__global__ __launch_bounds__(ES_RED_THREADS) void k_es_spatial(const double* __restrict__ vals_t, const double* __restrict__ vals_s,
                                                               const int64_t* __restrict__ idx, int64_t m, int64_t ms, int32_t k,
                                                               double* __restrict__ out) {
    __shared__ double sh[ES_RED_THREADS];
    const int i = blockIdx.x / k, j = blockIdx.x - i * k;
    double a = 0.0, f = 0.0;
    for (int64_t r = threadIdx.x; r < m; r += ES_RED_THREADS) {
        const double t = vals_t[(int64_t)i * m + r], s = vals_s[(int64_t)j * ms + idx[r]];
        const double d0 = s - t, d1 = -s - t;
        a += d0 * d0;
        f += d1 * d1;
    }
    a = es_block_sum(a, sh);
    f = es_block_sum(f, sh);
    if (threadIdx.x == 0) {
        out[2 * k * k + i * k + j] = sqrt(a) / (double)m;
        out[3 * k * k + i * k + j] = sqrt(f) / (double)m;
    }
}

}  // namespace

extern "C" int pf_eigsort_costs(pf_graph* gt, pf_graph* gs, const int64_t* rows_t, int64_t mt, const int64_t* rows_s, int64_t ms, int32_t k,
                                const int32_t* col_t, const double* sign_t, const int32_t* col_s, const double* sign_s, double* out,
                                int64_t* idx_out) {
    PF_CHECK(gt && gs && rows_t && rows_s && col_t && sign_t && col_s && sign_s && out && idx_out, PF_E_ARG,
             "pf_eigsort_costs: NULL argument");
    PF_CHECK(gt->ctx == gs->ctx, PF_E_ARG, "pf_eigsort_costs: the two graphs must share one ctx");
    PF_CHECK(mt >= 1 && mt <= ES_MAX_SAMPLES && ms >= 1 && ms <= ES_MAX_SAMPLES && k >= 1 && k <= 16, PF_E_ARG,
             "pf_eigsort_costs: %lld / %lld samples (1..%d), k = %d (1..16)", (long long)mt, (long long)ms, ES_MAX_SAMPLES, k);
    PF_CHECK(gt->final_vecs && gs->final_vecs, PF_E_STATE, "pf_eigsort_costs: no pf_finalize_vectors result is resident");
    PF_CHECK(gt->pts && gs->pts, PF_E_STATE, "pf_eigsort_costs: the graphs were not built from meshes");
    for (int32_t c = 0; c < k; ++c)
        PF_CHECK(col_t[c] >= 0 && col_t[c] < gt->final_count && col_s[c] >= 0 && col_s[c] < gs->final_count, PF_E_ARG,
                 "pf_eigsort_costs: column out of range");
    for (int64_t r = 0; r < mt; ++r) PF_CHECK(rows_t[r] >= 0 && rows_t[r] < gt->n, PF_E_ARG, "pf_eigsort_costs: sample row out of range");
    for (int64_t r = 0; r < ms; ++r) PF_CHECK(rows_s[r] >= 0 && rows_s[r] < gs->n, PF_E_ARG, "pf_eigsort_costs: sample row out of range");
    pf_ctx* c = gt->ctx;
    PF_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    int32_t n_pow2 = 2;
    while (n_pow2 < (mt > ms ? mt : ms)) n_pow2 <<= 1;
    const size_t sort_lds = sizeof(double) * (size_t)n_pow2;
    if (sort_lds > 64 * 1024)  // (per device: set whenever it is needed, a host-side call of microseconds)
        PF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_es_sort_log), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(sizeof(double) * ES_MAX_SAMPLES)));

    // one host block -> one upload: rows of both graphs, then columns and signs
    const size_t head = sizeof(int64_t) * (size_t)(mt + ms);
    const size_t small = ((sizeof(int32_t) * 2 * k + 7) & ~(size_t)7) + sizeof(double) * 2 * k;
    std::vector<unsigned char> host(head + small);
    memcpy(host.data(), rows_t, sizeof(int64_t) * (size_t)mt);
    memcpy(host.data() + sizeof(int64_t) * (size_t)mt, rows_s, sizeof(int64_t) * (size_t)ms);
    int32_t* hcol = reinterpret_cast<int32_t*>(host.data() + head);
    double* hsign = reinterpret_cast<double*>(host.data() + head + ((sizeof(int32_t) * 2 * k + 7) & ~(size_t)7));
    for (int32_t i = 0; i < k; ++i) {
        hcol[i] = col_t[i];
        hcol[k + i] = col_s[i];
        hsign[i] = sign_t[i];
        hsign[k + i] = sign_s[i];
    }
    // device scratch: [upload][vals_t k mt][vals_s k ms][raw_t 3 mt][raw_s 3 ms][norm_t][norm_s][lohi 12][lt k mt][ls k ms][lsf k ms][out 4 k k]
    const size_t kt = (size_t)k * (size_t)mt, ks = (size_t)k * (size_t)ms;
    const size_t doubles = 2 * kt + 3 * ks + 6 * (size_t)(mt + ms) + 12 + 4 * (size_t)k * k;
    unsigned char* d = nullptr;
    // both small transfers of this call go through pinned memory and copy kernels (pf_copy_by_kernel: no DMA engine)
    const size_t up_bytes = (host.size() + 7) & ~(size_t)7, out_bytes = sizeof(double) * 4 * (size_t)k * k;
    unsigned char* pin = nullptr;
    PF_TRY(pf_pinned_scratch(c, up_bytes + out_bytes, reinterpret_cast<void**>(&pin)));
    memcpy(pin, host.data(), host.size());
    PF_HIP(pf_malloc(st, (void**)&d, up_bytes + sizeof(double) * doubles));
    int rc = PF_OK;
    do {
        auto fail = [&](hipError_t e) {
            if (e == hipSuccess) return false;
            pf_set_error("pf_eigsort_costs: %s", hipGetErrorString(e));
            rc = PF_E_HIP;
            return true;
        };
        if ((rc = pf_copy_by_kernel(st, pin, d, up_bytes)) != PF_OK) break;
        const int64_t* d_rows_t = reinterpret_cast<const int64_t*>(d);
        const int64_t* d_rows_s = d_rows_t + mt;
        const int32_t* d_col = reinterpret_cast<const int32_t*>(d + head);
        const double* d_sign = reinterpret_cast<const double*>(d + head + ((sizeof(int32_t) * 2 * k + 7) & ~(size_t)7));
        double* base = reinterpret_cast<double*>(d + up_bytes);
        double* vals_t = base;
        double* vals_s = vals_t + kt;
        double* raw_t = vals_s + ks;
        double* raw_s = raw_t + 3 * mt;
        double* norm_t = raw_s + 3 * ms;
        double* norm_s = norm_t + 3 * mt;
        double* lohi = norm_s + 3 * ms;
        double* lt = lohi + 12;
        double* ls = lt + kt;
        double* lsf = ls + ks;
        double* d_out = lsf + ks;
        k_es_gather<<<es_blocks(mt * (k + 3)), PF_BLOCK, 0, st>>>(gt->final_vecs, gt->final_count, gt->pts, d_rows_t, mt, k, d_col, d_sign,
                                                                  vals_t, raw_t);
        k_es_gather<<<es_blocks(ms * (k + 3)), PF_BLOCK, 0, st>>>(gs->final_vecs, gs->final_count, gs->pts, d_rows_s, ms, k, d_col + k,
                                                                  d_sign + k, vals_s, raw_s);
        k_es_minmax<<<3, ES_RED_THREADS, 0, st>>>(raw_t, mt, lohi);
        k_es_minmax<<<3, ES_RED_THREADS, 0, st>>>(raw_s, ms, lohi + 6);
        k_es_normalize<<<es_blocks(3 * mt), PF_BLOCK, 0, st>>>(raw_t, lohi, mt, norm_t);
        k_es_normalize<<<es_blocks(3 * ms), PF_BLOCK, 0, st>>>(raw_s, lohi + 6, ms, norm_s);
        k_es_sort_log<<<(unsigned)(2 * k), ES_SORT_THREADS, sort_lds, st>>>(vals_t, vals_s, mt, ms, k, lt, ls, lsf);
        k_es_w1<<<(unsigned)(k * k), ES_RED_THREADS, 0, st>>>(lt, ls, lsf, mt, ms, k, d_out);
        if (fail(hipGetLastError())) break;
        // the 3-D 1-NN of every target sample point among the source sample points (eigsort.py:203-204); its
        // synchronisation also covers everything queued above
        const int32_t cols3[3] = {0, 1, 2};
        const double ones3[3] = {1.0, 1.0, 1.0};
        rc = pf_knn1_blocks(c, norm_s, ms, 3, norm_t, mt, 3, 3, cols3, ones3, cols3, ones3, idx_out, nullptr);
        if (rc != PF_OK) break;
        k_es_spatial<<<(unsigned)(k * k), ES_RED_THREADS, 0, st>>>(vals_t, vals_s, c->knn_idx, mt, ms, k, d_out);
        if (fail(hipGetLastError())) break;
        if ((rc = pf_copy_by_kernel(st, d_out, pin + up_bytes, out_bytes)) != PF_OK) break;
        if (fail(hipStreamSynchronize(st))) break;
        memcpy(out, pin + up_bytes, out_bytes);
    } while (0);
    if (rc != PF_OK) (void)hipStreamSynchronize(st);
    pf_free(st, d);
    return rc;
}
