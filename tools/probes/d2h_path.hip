// Which engine carries a 10 MB device-to-pinned-host hipMemcpyAsync, by stream priority and source allocation, and what a
// latency-bound kernel on another stream pays while it runs.  Build: hipcc --offload-arch=gfx950 -O2 d2h_path.hip -o d2h_path
// Run under:  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d out -- ./d2h_path
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_chase(const int* __restrict__ next, int n, int steps, int* __restrict__ out) {
    int i = (blockIdx.x * blockDim.x + threadIdx.x) % n;
    for (int s = 0; s < steps; ++s) i = next[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = i;
}
__global__ void k_mark(int* p) { if (threadIdx.x == 0) atomicAdd(p, 1); }

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    const size_t bytes = 10u << 20;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("priority range: least %d greatest %d\n", lo, hi);
    hipStream_t s_main, s_norm, s_low, s_high;
    CK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&s_norm, hipStreamNonBlocking, 0));
    CK(hipStreamCreateWithPriority(&s_low, hipStreamNonBlocking, lo));
    CK(hipStreamCreateWithPriority(&s_high, hipStreamNonBlocking, hi));
    void *h = nullptr, *d = nullptr, *dp = nullptr;
    CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    CK(hipMalloc(&d, bytes));
    CK(hipMallocAsync(&dp, bytes, s_main));
    CK(hipMemsetAsync(d, 1, bytes, s_main));
    CK(hipMemsetAsync(dp, 2, bytes, s_main));
    // pointer-chase table of 64 MB: a random cycle
    const int n = 16 << 20;
    std::vector<int> nx(n);
    {
        std::vector<int> perm(n);
        for (int i = 0; i < n; ++i) perm[i] = i;
        unsigned long long r = 88172645463325252ull;
        for (int i = n - 1; i > 0; --i) {
            r ^= r << 13; r ^= r >> 7; r ^= r << 17;
            int j = (int)(r % (unsigned long long)(i + 1));
            int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
        }
        for (int i = 0; i < n; ++i) nx[perm[i]] = perm[(i + 1) % n];
    }
    int *d_next, *d_out, *d_mark;
    CK(hipMalloc((void**)&d_next, sizeof(int) * (size_t)n));
    CK(hipMalloc((void**)&d_out, sizeof(int) * 256 * 2048));
    CK(hipMalloc((void**)&d_mark, 64));
    CK(hipMemcpy(d_next, nx.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1, ready;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    // What precedes the copy on its stream decides the engine (kernel trace: __amd_rocclr_copyBuffer = the runtime's blit
    // kernel; memory-copy trace: an SDMA engine).  One k_chase on the main stream per case separates them in the trace.
    void* h_small = nullptr;
    CK(hipHostMalloc(&h_small, 4096, hipHostMallocDefault));
    enum { PLAIN, WAIT_DONE_EVENT, WAIT_PENDING_EVENT, AFTER_SMALL_COPY, AFTER_SMALL_COPY_BEHIND_WAIT, AFTER_OWN_KERNEL, N_CASES };
    const char* names[N_CASES] = {"nothing before it on the stream", "a wait for an event that has completed",
                                  "a wait for an event behind the running kernel of the main stream",
                                  "a 128-byte download (completed long ago)", "a wait for a pending event + 128-byte download (completed long ago)",
                                  "a kernel of the same stream (completed long ago)"};
    for (int rep = 0; rep < 2; ++rep)
        for (int c = 0; c < N_CASES; ++c) {
            hipStream_t cs = s_low;
            CK(hipDeviceSynchronize());
            if (c == AFTER_SMALL_COPY) CK(hipMemcpyAsync(h_small, d, 128, hipMemcpyDeviceToHost, cs));
            if (c == AFTER_OWN_KERNEL) k_mark<<<1, 64, 0, cs>>>(d_mark);
            if (c == AFTER_SMALL_COPY_BEHIND_WAIT) {
                k_chase<<<2048, 256, 0, s_main>>>(d_next, n, 20, d_out);
                CK(hipEventRecord(ready, s_main));
                CK(hipStreamWaitEvent(cs, ready, 0));
                CK(hipMemcpyAsync(h_small, d, 128, hipMemcpyDeviceToHost, cs));
            }
            if (c == AFTER_SMALL_COPY || c == AFTER_OWN_KERNEL || c == AFTER_SMALL_COPY_BEHIND_WAIT) {
                const double w = now_us();
                while (now_us() - w < 2000.0) {}  // the device finishes; the host does not wait on anything
            }
            if (c == WAIT_DONE_EVENT) {
                k_mark<<<1, 64, 0, s_main>>>(d_mark);
                CK(hipEventRecord(ready, s_main));
                CK(hipEventSynchronize(ready));
            }
            CK(hipEventRecord(e0, s_main));
            k_chase<<<2048, 256, 0, s_main>>>(d_next, n, 600, d_out);
            CK(hipEventRecord(e1, s_main));
            if (c == WAIT_PENDING_EVENT) CK(hipEventRecord(ready, s_main));
            const double t0 = now_us();
            if (c == WAIT_DONE_EVENT || c == WAIT_PENDING_EVENT) CK(hipStreamWaitEvent(cs, ready, 0));
            for (int k = 0; k < 2; ++k) CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, cs));
            CK(hipStreamSynchronize(cs));
            const double t1 = now_us();
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("case %d: before the copies: %-70s chase kernel %7.1f us   2 copies done after %7.1f us\n", c, names[c], ms * 1e3, t1 - t0);
        }
    return 0;
}
