// Launches that two meshes can SHARE (pf_graph_build_device2).
//
// An assembly is a chain of ~50 dependent dispatches per mesh, most of them a few microseconds of work behind the launch
// latency.  Rounds 2-4 ran the two meshes of a pair on two streams: ~100 launches of ~3 us of host time each, queued from
// one thread, so that the second mesh's chain trails the first one's and the build's length follows the host's pace (0.90
// to 1.05 ms from process to process on this pool).  Here the kernels of the assembly path are written as functors
//     struct k_foo { static constexpr int BOUNDS = ...; static __device__ void run(args...); };
// and started through pfl::launch<k_foo>(grid, block, lds, stream, args...).  Normally that is one ordinary launch.  While
// a pfl::Recorder is installed in the calling thread (pfl::tl_rec) the launch is only RECORDED; the pair build runs every
// phase of its host code once per mesh, each into its own recorder, and pfl::flush zips the two lists: where both meshes
// ask for the same kernel, ONE launch with gridDim.z = 2 runs both (block (x, y, z) works for mesh z if (x, y) lies inside
// that mesh's own grid), everything else (copies, event records, a kernel only one mesh needs) runs one after the other.
// Half the dispatches, half the host time, no second stream - and the arithmetic of every kernel is untouched.
//
// While a recorder is installed, pf_free defers (pf_api.hip): the host code of mesh a would otherwise hand a block back
// that mesh b's host code takes for a kernel which runs BEFORE the last kernel of mesh a that uses it.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <functional>
#include <utility>
#include <vector>

namespace pfl {

// ---- the argument pack of a kernel: an aggregate (trivially copyable whatever the standard library's tuple is)
template <typename... A>
struct Pack;
template <>
struct Pack<> {};
template <typename H, typename... T>
struct Pack<H, T...> {
    H head;
    Pack<T...> tail;
};

inline Pack<> make_pack() { return {}; }
template <typename H, typename... T>
Pack<H, T...> make_pack(H h, T... t) {
    return Pack<H, T...>{h, make_pack(t...)};
}

template <typename K, typename... D>
__device__ __forceinline__ void unpack(const Pack<>&, D... d) {
    K::run(d...);
}
template <typename K, typename H, typename... T, typename... D>
__device__ __forceinline__ void unpack(const Pack<H, T...>& p, D... d) {
    unpack<K>(p.tail, d..., p.head);
}

template <typename T>
struct ident {
    using type = T;
};
// the parameter types come from the kernel's own signature (a nullptr or an int at the call converts as in a direct call)
template <typename F>
struct sig;
template <typename R, typename... A>
struct sig<R (*)(A...)> {
    using pack = Pack<A...>;
    static pack make(typename ident<A>::type... a) { return make_pack<A...>(a...); }
};

template <typename K, typename P>
__global__ __launch_bounds__(K::BOUNDS) void k_one(P p) {
    unpack<K>(p);
}
// two meshes in one launch: z = 0 works with a inside grid ga, z = 1 with b inside gb
template <typename K, typename P>
__global__ __launch_bounds__(K::BOUNDS) void k_two(P a, P b, uint2 ga, uint2 gb) {
    const bool second = blockIdx.z != 0;
    const uint2 own = second ? gb : ga;
    if (blockIdx.x >= own.x || blockIdx.y >= own.y) return;
    const P& p = second ? b : a;
    unpack<K>(p);
}

struct Op {
    const void* key = nullptr;  // what may share a launch: the address of the two-mesh kernel
    dim3 grid, block;
    size_t lds = 0;
    hipStream_t st = nullptr;  // the stream the host code named (a build forks its independent chains onto a side stream)
    std::vector<unsigned char> args;
    void (*run1)(const Op&, hipStream_t) = nullptr;
    void (*run2)(const Op&, const Op&, hipStream_t) = nullptr;
    std::function<void(hipStream_t)> call;  // (key == nullptr) anything else that has to keep its place in the order
};

struct Recorder {
    std::vector<Op> ops;
    std::vector<std::pair<hipStream_t, void*>> frees;  // pf_free calls held back until the recorded work is queued
};

inline thread_local Recorder* tl_rec = nullptr;

template <typename K, typename... X>
void launch(dim3 grid, dim3 block, size_t lds, hipStream_t st, X... x) {
    using S = sig<decltype(&K::run)>;
    using P = typename S::pack;
    const P p = S::make(x...);
    Recorder* r = tl_rec;
    if (!r) {
        k_one<K, P><<<grid, block, lds, st>>>(p);
        return;
    }
    Op op;
    op.key = reinterpret_cast<const void*>(&k_two<K, P>);
    op.grid = grid, op.block = block, op.lds = lds, op.st = st;
    op.args.resize(sizeof(P));
    memcpy(op.args.data(), &p, sizeof(P));
    op.run1 = [](const Op& o, hipStream_t s) {
        P q;
        memcpy(&q, o.args.data(), sizeof(P));
        k_one<K, P><<<o.grid, o.block, o.lds, s>>>(q);
    };
    op.run2 = [](const Op& a, const Op& b, hipStream_t s) {
        P qa, qb;
        memcpy(&qa, a.args.data(), sizeof(P));
        memcpy(&qb, b.args.data(), sizeof(P));
        const dim3 g(a.grid.x > b.grid.x ? a.grid.x : b.grid.x, a.grid.y > b.grid.y ? a.grid.y : b.grid.y, 2u);
        k_two<K, P><<<g, a.block, a.lds > b.lds ? a.lds : b.lds, s>>>(qa, qb, make_uint2(a.grid.x, a.grid.y), make_uint2(b.grid.x, b.grid.y));
    };
    r->ops.push_back(std::move(op));
}

// anything else on the stream (copies, event records): runs at its place in the order, never shared
inline void call(hipStream_t st, std::function<void(hipStream_t)> f) {
    Recorder* r = tl_rec;
    if (!r) {
        f(st);
        return;
    }
    Op op;
    op.call = std::move(f);
    op.st = st;
    r->ops.push_back(std::move(op));
}

struct k_fill_words {
    static constexpr int BOUNDS = 256;
    static __device__ __forceinline__ void run(uint32_t* p, int64_t words, uint32_t value) {
        const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
        if (i + 3 < words) {
            *reinterpret_cast<uint4*>(p + i) = make_uint4(value, value, value, value);
        } else {
            for (int64_t j = i; j < words; ++j) p[j] = value;
        }
    }
};

// hipMemsetAsync, or - recorded - a fill kernel that the partner's fill shares a launch with (p 16-byte aligned, whole words)
inline hipError_t memset_words(hipStream_t st, void* p, int byte_value, size_t bytes) {
    if (!tl_rec || (bytes & 3) || (reinterpret_cast<uintptr_t>(p) & 15)) {
        if (!tl_rec) return hipMemsetAsync(p, byte_value, bytes, st);
        call(st, [=](hipStream_t s) { (void)hipMemsetAsync(p, byte_value, bytes, s); });
        return hipSuccess;
    }
    const uint32_t b = (uint32_t)(byte_value & 0xff);
    const int64_t words = (int64_t)(bytes / 4);
    launch<k_fill_words>(dim3((unsigned)((words + 1023) / 1024)), dim3(256), 0, st, reinterpret_cast<uint32_t*>(p), words, b * 0x01010101u);
    return hipSuccess;
}

// Queue what has been recorded, every record on the stream it names: shared launches where the two lists ask for the same
// kernel with the same block shape.  `b` may be null (one list, as it stands).  Then the held-back frees.  The recorders
// are left empty.
void flush(Recorder& a, Recorder* b);

// the calling thread's own list, one launch per record (a host-side wait is about to follow)
void flush_self();

}  // namespace pfl

namespace pfl {

inline hipError_t memcpy_async(hipStream_t st, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (!tl_rec) return hipMemcpyAsync(dst, src, bytes, kind, st);
    call(st, [=](hipStream_t s) { (void)hipMemcpyAsync(dst, src, bytes, kind, s); });
    return hipSuccess;
}

inline hipError_t event_record(hipStream_t st, hipEvent_t ev) {
    if (!tl_rec) return hipEventRecord(ev, st);
    call(st, [=](hipStream_t s) { (void)hipEventRecord(ev, s); });
    return hipSuccess;
}

// a host-side wait: whatever the calling thread has recorded goes out first (unshared)
inline hipError_t sync(hipStream_t st) {
    flush_self();
    return hipStreamSynchronize(st);
}

}  // namespace pfl
