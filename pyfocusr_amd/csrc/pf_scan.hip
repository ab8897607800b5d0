// Exclusive prefix sums (int32 / int64) used by the Laplacian assembler: row pointers from
// per-vertex edge counts, SELL slice offsets from slice widths.  Three-phase scan: per-block
// totals -> (recursive) scan of the totals -> per-block rescan with its offset.  Wave-level
// scans use 64-lane shuffles; the four waves of a block meet through LDS.
#include "pf_internal.h"
#include "pf_launch.h"

namespace {

constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = PF_BLOCK * SCAN_ITEMS;  // 2048 elements per block

template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v, int lane) {
#pragma unroll
    for (int off = 1; off < PF_WAVE; off <<= 1) {
        T up = __shfl_up(v, off, PF_WAVE);
        if (lane >= off) v += up;
    }
    return v;
}

// exclusive scan of one value per thread over the 256-thread block; returns the block total via `total`
template <typename T>
__device__ __forceinline__ T block_exclusive_scan(T v, T* total) {
    __shared__ T wave_sum[PF_BLOCK / PF_WAVE];
    const int lane = threadIdx.x & (PF_WAVE - 1);
    const int wid = threadIdx.x / PF_WAVE;
    T inc = wave_inclusive_scan(v, lane);
    if (lane == PF_WAVE - 1) wave_sum[wid] = inc;
    __syncthreads();
    T offset = 0, all = 0;
#pragma unroll
    for (int wv = 0; wv < PF_BLOCK / PF_WAVE; ++wv) {
        T s = wave_sum[wv];
        if (wv < wid) offset += s;
        all += s;
    }
    __syncthreads();
    *total = all;
    return offset + inc - v;
}

template <typename T>
struct scan_block_totals {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const T* __restrict__ in, T* __restrict__ totals, int64_t n) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    T s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        int64_t i = base + k;
        if (i < n) s += in[i];
    }
    T total;
    (void)block_exclusive_scan(s, &total);
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
}
};

template <typename T>
struct scan_block_apply {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const T* __restrict__ in, T* __restrict__ out,
                                                             const T* __restrict__ block_offset, int64_t n) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS];
    T s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        int64_t i = base + k;
        v[k] = (i < n) ? in[i] : T(0);
        s += v[k];
    }
    T total;
    T run = block_exclusive_scan(s, &total) + (block_offset ? block_offset[blockIdx.x] : T(0));
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        int64_t i = base + k;
        if (i < n) out[i] = run;
        run += v[k];
    }
}
};

// the same, with the block's offset summed by the block itself from the (unscanned) totals of the blocks before it:
// up to SCAN_DIRECT_BLOCKS totals are few enough to be re-read by every block (123 for 250k elements), which saves the
// launch that scanned them
constexpr int64_t SCAN_DIRECT_BLOCKS = 4096;
template <typename T>
struct scan_block_apply_direct {
    static constexpr int BOUNDS = PF_BLOCK;
    static __device__ __forceinline__ void run(const T* __restrict__ in, T* __restrict__ out,
                                                                    const T* __restrict__ totals, int64_t n) {
    __shared__ T s_off;
    T mine = 0;
    for (int64_t b = threadIdx.x; b < blockIdx.x; b += PF_BLOCK) mine += totals[b];
    T before;
    (void)block_exclusive_scan(mine, &before);  // (integers: the order of the additions does not matter)
    if (threadIdx.x == 0) s_off = before;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS];
    T s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        int64_t i = base + k;
        v[k] = (i < n) ? in[i] : T(0);
        s += v[k];
    }
    T total;
    T run = block_exclusive_scan(s, &total) + s_off;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        int64_t i = base + k;
        if (i < n) out[i] = run;
        run += v[k];
    }
}
};

template <typename T>
int exclusive_scan(hipStream_t st, const T* in, T* out, int64_t n) {
    if (n <= 0) return PF_OK;
    const int64_t blocks = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (blocks == 1) {
        pfl::launch<scan_block_apply<T>>(dim3(1), dim3(PF_BLOCK), 0, st, in, out, nullptr, n);
        PF_HIP(hipGetLastError());
        return PF_OK;
    }
    T* totals = nullptr;
    PF_HIP(pf_malloc(st, (void**)&totals, sizeof(T) * blocks));
    pfl::launch<scan_block_totals<T>>(dim3((unsigned)blocks), dim3(PF_BLOCK), 0, st, in, totals, n);
    PF_HIP(hipGetLastError());
    int r = PF_OK;
    if (blocks <= SCAN_DIRECT_BLOCKS) {
        pfl::launch<scan_block_apply_direct<T>>(dim3((unsigned)blocks), dim3(PF_BLOCK), 0, st, in, out, totals, n);
        if (hipGetLastError() != hipSuccess) r = PF_E_HIP;
    } else {
        r = exclusive_scan<T>(st, totals, totals, blocks);
        if (r == PF_OK) {
            pfl::launch<scan_block_apply<T>>(dim3((unsigned)blocks), dim3(PF_BLOCK), 0, st, in, out, totals, n);
            if (hipGetLastError() != hipSuccess) r = PF_E_HIP;
        }
    }
    pf_free(st, totals);
    return r;
}

}  // namespace

int pf_exclusive_scan_i32(hipStream_t st, const int32_t* in, int32_t* out, int64_t n) {
    return exclusive_scan<int32_t>(st, in, out, n);
}
int pf_exclusive_scan_i64(hipStream_t st, const int64_t* in, int64_t* out, int64_t n) {
    return exclusive_scan<int64_t>(st, in, out, n);
}
