// Exhaustive 1-nearest-neighbour search in d <= 8 dimensions, float64.
//
// Replaces `scipy.spatial.KDTree(target).query(source)` at
// /root/reference/pyfocusr/focusr.py:351-353 (spectral coordinates, d = n_spectral_features)
// and /root/reference/pyfocusr/eigsort.py:203-204 (normalised xyz, d = 3).
//
// FP64-VALU-bound, not a dense contraction: depth d <= 8, and the |x|^2+|y|^2-2xy expansion
// that MFMA would need loses ~5 digits to cancellation at neighbour distances ~1e-3, which
// breaks index parity.  So: direct sum of squared differences, accumulated left to right with
// separate multiply and add (file compiled with -ffp-contract=off) — the same roundings as a
// numpy brute force — strict '<' while scanning references in ascending index order, so the
// lowest index wins ties.
//
// Layout: one query per lane held in registers; reference points streamed through LDS in
// tiles (coalesced global reads, broadcast LDS reads: every lane reads the same reference).
// The reference range is split over gridDim.y so small query sets still fill 256 CUs; a
// second kernel merges the per-split minima in split order.
#include <algorithm>

#include "pf_internal.h"

namespace {

constexpr int KNN_TILE = 512;  // reference points per LDS tile (d=8: 32 KiB)

template <int D>
__global__ __launch_bounds__(PF_BLOCK) void k_knn_partial(const double* __restrict__ ref, int64_t n_ref,
                                                          const double* __restrict__ qry, int64_t n_qry,
                                                          int64_t refs_per_split, double* __restrict__ part_d2,
                                                          int32_t* __restrict__ part_idx) {
    __shared__ double tile[KNN_TILE * D];
    const int64_t q = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    const int64_t r_begin = (int64_t)blockIdx.y * refs_per_split;
    const int64_t r_end = r_begin + refs_per_split < n_ref ? r_begin + refs_per_split : n_ref;
    double qc[D];
    const int64_t qq = q < n_qry ? q : n_qry - 1;  // tail lanes replay the last query, result discarded
#pragma unroll
    for (int c = 0; c < D; ++c) qc[c] = qry[qq * D + c];
    double best = INFINITY;
    int32_t best_idx = (int32_t)r_begin;
    for (int64_t t0 = r_begin; t0 < r_end; t0 += KNN_TILE) {
        const int cnt = (int)((r_end - t0) < KNN_TILE ? (r_end - t0) : KNN_TILE);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt * D; k += PF_BLOCK) tile[k] = ref[t0 * D + k];
        __syncthreads();
        for (int r = 0; r < cnt; ++r) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < D; ++c) {
                const double df = qc[c] - tile[r * D + c];
                const double sq = df * df;
                s = (c == 0) ? sq : s + sq;
            }
            if (s < best) {
                best = s;
                best_idx = (int32_t)(t0 + r);
            }
        }
    }
    if (q < n_qry) {
        part_d2[(int64_t)blockIdx.y * n_qry + q] = best;
        part_idx[(int64_t)blockIdx.y * n_qry + q] = best_idx;
    }
}

__global__ __launch_bounds__(PF_BLOCK) void k_knn_merge(const double* __restrict__ part_d2,
                                                        const int32_t* __restrict__ part_idx, int64_t n_qry,
                                                        int32_t splits, int64_t* __restrict__ idx, double* __restrict__ d2) {
    const int64_t q = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (q >= n_qry) return;
    double best = part_d2[q];
    int32_t bi = part_idx[q];
    for (int32_t s = 1; s < splits; ++s) {
        const double v = part_d2[(int64_t)s * n_qry + q];
        if (v < best) {  // strict: earlier split (lower indices) wins ties
            best = v;
            bi = part_idx[(int64_t)s * n_qry + q];
        }
    }
    idx[q] = bi;
    d2[q] = best;
}

template <int D>
int launch_knn(pf_ctx* c, dim3 grid, int64_t refs_per_split) {
    k_knn_partial<D><<<grid, PF_BLOCK, 0, c->stream>>>(c->knn_ref, c->knn_nref, c->knn_qry, c->knn_nqry, refs_per_split,
                                                       c->knn_part_d2, c->knn_part_idx);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

template <typename T>
int grow(hipStream_t st, T** p, int64_t* cap, int64_t need) {
    if (need <= *cap) return PF_OK;
    pf_free(st, *p);
    *p = nullptr;
    *cap = 0;
    PF_HIP(pf_malloc(st, (void**)p, sizeof(T) * (size_t)need));
    *cap = need;
    return PF_OK;
}

}  // namespace

extern "C" {

int pf_knn_upload(pf_ctx* c, const double* ref, int64_t n_ref, const double* qry, int64_t n_qry, int32_t d) {
    PF_CHECK(c && ref && qry, PF_E_ARG, "pf_knn_upload: NULL argument");
    PF_CHECK(n_ref > 0 && n_ref < ((int64_t)1 << 31) && n_qry > 0 && d >= 1 && d <= 8, PF_E_ARG,
             "pf_knn_upload: n_ref %lld, n_qry %lld, d %d out of range (1 <= d <= 8)", (long long)n_ref, (long long)n_qry, d);
    PF_HIP(hipSetDevice(c->device));
    c->knn_ready = c->knn_done = false;
    PF_TRY(grow(c->stream, &c->knn_ref, &c->knn_cap_ref, n_ref * 8));
    PF_TRY(grow(c->stream, &c->knn_qry, &c->knn_cap_qry, n_qry * 8));
    // enough (query-block x split) work items to cover 256 CUs several times
    const int64_t q_blocks = (n_qry + PF_BLOCK - 1) / PF_BLOCK;
    int64_t splits = (2048 + q_blocks - 1) / q_blocks;
    splits = std::max<int64_t>(1, std::min<int64_t>(splits, (n_ref + KNN_TILE - 1) / KNN_TILE));
    splits = std::min<int64_t>(splits, 65535);
    const int64_t need = splits * n_qry;
    if (need > c->knn_cap_part) {
        pf_free(c->stream, c->knn_part_d2);
        pf_free(c->stream, c->knn_part_idx);
        pf_free(c->stream, c->knn_idx);
        pf_free(c->stream, c->knn_d2);
        c->knn_part_d2 = nullptr;
        c->knn_part_idx = nullptr;
        c->knn_idx = nullptr;
        c->knn_d2 = nullptr;
        c->knn_cap_part = 0;
        PF_HIP(pf_malloc(c->stream, (void**)&c->knn_part_d2, sizeof(double) * (size_t)need));
        PF_HIP(pf_malloc(c->stream, (void**)&c->knn_part_idx, sizeof(int32_t) * (size_t)need));
        PF_HIP(pf_malloc(c->stream, (void**)&c->knn_idx, sizeof(int64_t) * (size_t)need));
        PF_HIP(pf_malloc(c->stream, (void**)&c->knn_d2, sizeof(double) * (size_t)need));
        c->knn_cap_part = (int32_t)std::min<int64_t>(need, INT32_MAX);
        PF_CHECK(need <= INT32_MAX, PF_E_ARG, "pf_knn_upload: problem too large");
    }
    c->knn_splits = (int32_t)splits;
    c->knn_nref = n_ref;
    c->knn_nqry = n_qry;
    c->knn_d = d;
    PF_HIP(hipMemcpyAsync(c->knn_ref, ref, sizeof(double) * n_ref * d, hipMemcpyHostToDevice, c->stream));
    PF_HIP(hipMemcpyAsync(c->knn_qry, qry, sizeof(double) * n_qry * d, hipMemcpyHostToDevice, c->stream));
    PF_HIP(hipStreamSynchronize(c->stream));
    c->knn_ready = true;
    return PF_OK;
}

int pf_knn_run(pf_ctx* c) {
    PF_CHECK(c != nullptr, PF_E_ARG, "pf_knn_run: ctx is NULL");
    PF_CHECK(c->knn_ready, PF_E_STATE, "pf_knn_run: no uploaded problem");
    PF_HIP(hipSetDevice(c->device));
    const int64_t q_blocks = (c->knn_nqry + PF_BLOCK - 1) / PF_BLOCK;
    const int64_t per = (c->knn_nref + c->knn_splits - 1) / c->knn_splits;
    dim3 grid((unsigned)q_blocks, (unsigned)c->knn_splits);
    PF_HIP(hipEventRecord(c->ev0, c->stream));
    int r = PF_E_ARG;
    switch (c->knn_d) {
        case 1: r = launch_knn<1>(c, grid, per); break;
        case 2: r = launch_knn<2>(c, grid, per); break;
        case 3: r = launch_knn<3>(c, grid, per); break;
        case 4: r = launch_knn<4>(c, grid, per); break;
        case 5: r = launch_knn<5>(c, grid, per); break;
        case 6: r = launch_knn<6>(c, grid, per); break;
        case 7: r = launch_knn<7>(c, grid, per); break;
        case 8: r = launch_knn<8>(c, grid, per); break;
        default: break;
    }
    PF_TRY(r);
    k_knn_merge<<<(unsigned)q_blocks, PF_BLOCK, 0, c->stream>>>(c->knn_part_d2, c->knn_part_idx, c->knn_nqry, c->knn_splits,
                                                               c->knn_idx, c->knn_d2);
    PF_HIP(hipGetLastError());
    PF_HIP(hipEventRecord(c->ev1, c->stream));
    PF_HIP(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    PF_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->knn_ms = ms;
    c->knn_done = true;
    return PF_OK;
}

int pf_knn_download(pf_ctx* c, int64_t* idx_out, double* d2_out) {
    PF_CHECK(c != nullptr && idx_out != nullptr, PF_E_ARG, "pf_knn_download: NULL argument");
    PF_CHECK(c->knn_done, PF_E_STATE, "pf_knn_download: pf_knn_run has not completed");
    PF_HIP(hipMemcpyAsync(idx_out, c->knn_idx, sizeof(int64_t) * c->knn_nqry, hipMemcpyDeviceToHost, c->stream));
    if (d2_out) PF_HIP(hipMemcpyAsync(d2_out, c->knn_d2, sizeof(double) * c->knn_nqry, hipMemcpyDeviceToHost, c->stream));
    PF_HIP(hipStreamSynchronize(c->stream));
    return PF_OK;
}

int pf_knn1(pf_ctx* c, const double* ref, int64_t n_ref, const double* qry, int64_t n_qry, int32_t d, int64_t* idx_out,
            double* d2_out) {
    PF_TRY(pf_knn_upload(c, ref, n_ref, qry, n_qry, d));
    PF_TRY(pf_knn_run(c));
    return pf_knn_download(c, idx_out, d2_out);
}

}  // extern "C"
