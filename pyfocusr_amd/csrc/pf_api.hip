// Context, error reporting and timing for libpyfocusr_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <mutex>

#include "pf_internal.h"
#include "pf_launch.h"

static thread_local char g_err[1024] = "";
static std::mutex g_ctx_mutex;
static std::vector<pf_ctx*> g_ctxs;

// Small transfers of the hot path go through kernels that read or write PINNED host memory directly, not through
// hipMemcpyAsync: a copy command shares the DMA engines with the eigenvector downloads (2 x 10 MB per 250k pair, in
// flight on the copy stream while eigsort and the KNN run) and, depending on which engine the runtime picks in a given
// process, waits behind them: 40 KB of eigsort results then cost 0.6 ms (seen in one process of three).
__global__ __launch_bounds__(PF_BLOCK) void k_copy_words(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst,
                                                         size_t n) {
    const size_t i = (size_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

int pf_copy_by_kernel(hipStream_t st, const void* src, void* dst, size_t bytes) {
    if (bytes == 0) return PF_OK;
    const size_t n = (bytes + 7) / 8;  // (both buffers are allocated in multiples of 8 bytes by their owners)
    k_copy_words<<<(unsigned)((n + PF_BLOCK - 1) / PF_BLOCK), PF_BLOCK, 0, st>>>(static_cast<const unsigned long long*>(src),
                                                                                 static_cast<unsigned long long*>(dst), n);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

static pf_ctx* ctx_of_stream(hipStream_t st, int* sid = nullptr) {
    std::lock_guard<std::mutex> lk(g_ctx_mutex);
    for (pf_ctx* c : g_ctxs) {
        if (c->stream == st || (c->stream_b && c->stream_b == st)) {
            if (sid) *sid = c->stream == st ? 0 : 1;
            return c;
        }
    }
    return nullptr;
}

int pf_pinned_scratch(pf_ctx* c, size_t bytes, void** out, int sid) {
    void*& buf = sid == 2 ? c->pinned_scratch_knn : (sid ? c->pinned_scratch_b : c->pinned_scratch);
    size_t& have = sid == 2 ? c->pinned_scratch_knn_bytes : (sid ? c->pinned_scratch_b_bytes : c->pinned_scratch_bytes);
    if (bytes > have) {
        if (buf) {
            PF_HIP(hipStreamSynchronize(sid == 1 ? c->stream_b : c->stream));
            PF_HIP(hipHostFree(buf));
            buf = nullptr;
            have = 0;
        }
        const size_t cap = bytes > ((size_t)1 << 16) ? bytes : ((size_t)1 << 16);
        PF_HIP(hipHostMalloc(&buf, cap, hipHostMallocDefault));
        have = cap;
    }
    *out = buf;
    return PF_OK;
}

// A secondary stream of a ctx (second assembly, downloads) must not share a hardware queue with the ctx's main stream, or
// what it carries simply queues up between the main stream's kernels.  The runtime hands out hardware queues from one
// pool per priority level, shared by every stream of the process (torch's included), and which streams end up on one queue
// differs from process to process (seen as: one process in three assembled the two meshes one after the other and ran the
// eigenvector downloads in front of eigsort's kernels, +0.9 ms per step).  A stream of another priority comes from another
// pool: never the main stream's queue.  The same holds among the side streams: the second assembly's stream and the
// copy stream, both at the greatest priority, shared a queue in one process of three again once the step had become
// short enough to show it (round 3: 12.1-12.9 ms instead of 11.4) - the copy stream now takes the LEAST priority, a third
// pool: 20 of 20 processes within 11.27-11.56 ms, and the downloads no longer get in front of eigsort's kernels at all.
hipError_t pf_create_side_stream(hipStream_t* s, bool low) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least) {
        return hipStreamCreateWithPriority(s, hipStreamNonBlocking, low ? least : greatest);
    }
    (void)hipGetLastError();
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

hipStream_t pf_stream_b(pf_ctx* c) {
    if (!c->stream_b) {
        hipStream_t s = nullptr;
        // (the least priority, like the copy stream - they are never busy together: the first mesh's kernels, at the main
        // stream's normal priority, keep their pace and the second's fill the gaps: pair assembly 1.17 -> 1.14 ms)
        if (pf_create_side_stream(&s, true) != hipSuccess ||
            hipEventCreateWithFlags(&c->join_ev, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (s) (void)hipStreamDestroy(s);
            return nullptr;
        }
        std::lock_guard<std::mutex> lk(g_ctx_mutex);
        c->stream_b = s;
    }
    return c->stream_b;
}

void pf_worker_run(pf_ctx* c, std::function<void()> task) {
    std::unique_lock<std::mutex> lk(c->worker_mutex);
    if (!c->worker.joinable()) {
        c->worker = std::thread([c] {
            (void)hipSetDevice(c->device);
            std::unique_lock<std::mutex> l(c->worker_mutex);
            for (;;) {
                c->worker_cv.wait(l, [c] { return c->worker_stop || (c->worker_busy && c->worker_task); });
                if (c->worker_stop) return;
                std::function<void()> t = std::move(c->worker_task);
                c->worker_task = nullptr;
                l.unlock();
                t();
                l.lock();
                c->worker_busy = false;
                c->worker_cv.notify_all();
            }
        });
    }
    c->worker_cv.wait(lk, [c] { return !c->worker_busy; });
    c->worker_task = std::move(task);
    c->worker_busy = true;
    c->worker_cv.notify_all();
}

void pf_worker_wait(pf_ctx* c) {
    std::unique_lock<std::mutex> lk(c->worker_mutex);
    c->worker_cv.wait(lk, [c] { return !c->worker_busy; });
}

int pf_streams_join(pf_ctx* c, int waiter_sid) {
    PF_CHECK(c->stream_b != nullptr, PF_E_STATE, "pf_streams_join: no second stream");
    hipStream_t waiter = waiter_sid ? c->stream_b : c->stream, other = waiter_sid ? c->stream : c->stream_b;
    PF_HIP(hipEventRecord(c->join_ev, other));
    PF_HIP(hipStreamWaitEvent(waiter, c->join_ev, 0));
    c->alloc_epoch += 1;
    c->visible[waiter_sid] = c->alloc_epoch;
    return PF_OK;
}

hipError_t pf_malloc(hipStream_t st, void** p, size_t bytes) {
    *p = nullptr;
    int sid = 0;
    pf_ctx* c = ctx_of_stream(st, &sid);
    if (!c) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lk(c->alloc_mutex);
    bytes = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
    auto range = c->free_blocks.equal_range(bytes);
    // a block released on this stream (ordered by it) first; else one the other stream released before this stream last
    // waited for it (taking those first would drain the other stream's supply of exactly the sizes both use)
    auto pick = range.second;
    for (auto it = range.first; it != range.second; ++it) {
        if (it->second.sid == sid) {
            pick = it;
            break;
        }
        if (pick == range.second && it->second.epoch < c->visible[sid]) pick = it;
    }
    if (pick != range.second) {
        *p = pick->second.p;
        c->free_blocks.erase(pick);
    }
    if (!*p) {
        c->alloc_misses += 1;
        static const bool dbg_alloc = getenv("PF_DEBUG_ALLOC") != nullptr;
        const auto t_miss = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(p, bytes);
        if (dbg_alloc)
            fprintf(stderr, "libpyfocusr_hip: allocation %lld of ctx %p went to the driver: %zu bytes, stream %d, %.0f us\n", (long long)c->alloc_misses,
                    (void*)c, bytes, sid, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_miss).count());
        if (e != hipSuccess) {  // give cached blocks back to the driver and retry once
            (void)hipStreamSynchronize(c->stream);
            if (c->stream_b) (void)hipStreamSynchronize(c->stream_b);
            for (auto& kv : c->free_blocks) (void)hipFree(kv.second.p);
            c->free_blocks.clear();
            e = hipMalloc(p, bytes);
            if (e != hipSuccess) return e;
        }
    }
    c->live_blocks[*p] = bytes;
    return hipSuccess;
}

void pf_free(hipStream_t st, void* p) {
    if (!p) return;
    if (pfl::tl_rec) {  // launches of this thread are being recorded: the block stays taken until they have been queued
        pfl::tl_rec->frees.emplace_back(st, p);
        return;
    }
    int sid = 0;
    pf_ctx* c = ctx_of_stream(st, &sid);
    if (!c) return;
    std::lock_guard<std::mutex> lk(c->alloc_mutex);
    auto it = c->live_blocks.find(p);
    if (it == c->live_blocks.end()) return;  // not ours (or already released)
    c->free_blocks.emplace(it->second, pf_ctx::FreeBlock{p, sid, c->alloc_epoch});
    c->live_blocks.erase(it);
}

namespace pfl {

void flush(Recorder& a, Recorder* b) {
    Recorder* const saved = tl_rec;
    tl_rec = nullptr;  // what runs now is launched, and frees are frees
    size_t i = 0, j = 0;
    const size_t na = a.ops.size(), nb = b ? b->ops.size() : 0;
    auto alone = [](const Op& o) { o.key ? o.run1(o, o.st) : o.call(o.st); };
    while (i < na && j < nb) {
        const Op& x = a.ops[i];
        const Op& y = b->ops[j];
        if (x.key != y.key) {
            // a copy or an event record that only one mesh has (the first one records the build's start): it runs alone
            if (!x.key) {
                alone(x);
                ++i;
                continue;
            }
            if (!y.key) {
                alone(y);
                ++j;
                continue;
            }
            break;  // different kernels: the two host codes took different turns - the rest one after the other
        }
        if (x.key && x.st == y.st && x.block.x == y.block.x && x.block.y == y.block.y && x.block.z == y.block.z) {
            x.run2(x, y, x.st);
        } else {
            alone(x);
            alone(y);
        }
        ++i, ++j;
    }
    for (; i < na; ++i) alone(a.ops[i]);
    for (; j < nb; ++j) alone(b->ops[j]);
    a.ops.clear();
    for (auto& f : a.frees) pf_free(f.first, f.second);
    a.frees.clear();
    if (b) {
        b->ops.clear();
        for (auto& f : b->frees) pf_free(f.first, f.second);
        b->frees.clear();
    }
    tl_rec = saved;
}

void flush_self() {
    if (tl_rec) flush(*tl_rec, nullptr);
}

}  // namespace pfl

void pf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int pf_timing_collect(pf_ctx* c) {
    if (c->spans_pending.empty()) return PF_OK;
    PF_HIP(hipSetDevice(c->device));
    PF_HIP(hipStreamSynchronize(c->stream));
    for (auto& sp : c->spans_pending) {
        float ms = 0.f;
        PF_HIP(hipEventElapsedTime(&ms, sp.e0, sp.e1));
        c->op_ms += ms;
        c->op_launches += sp.launches;
        c->op_bytes += sp.bytes;
        if (sp.persist_steps > 0) {
            c->persist_ms += ms;
            c->persist_launches += sp.launches;
            c->persist_steps += sp.persist_steps;
            c->persist_bytes += sp.bytes;
            c->persist_lds_bytes += sp.lds_bytes;
        }
        c->spans_free.emplace_back(sp.e0, sp.e1);
    }
    c->spans_pending.clear();
    return PF_OK;
}

extern "C" {

int pf_version(void) { return PF_VERSION; }

int pf_host_alloc(size_t bytes, void** out) {
    PF_CHECK(out != nullptr && bytes > 0, PF_E_ARG, "pf_host_alloc: bad argument");
    *out = nullptr;
    PF_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return PF_OK;
}

// a download that is still OWED to this block (pf_finalize_vectors_begin holds it back) must never be queued: the device
// would write into memory that is about to be unmapped or handed to somebody else; one that is in flight is waited for
int pf_host_detach(void* p) {
    if (!p) return PF_OK;
    std::lock_guard<std::mutex> lk(g_ctx_mutex);
    for (pf_ctx* c : g_ctxs) {
        {
            std::lock_guard<std::mutex> lk2(c->deferred_mutex);
            for (size_t i = 0; i < c->deferred.size();) {
                pf_graph* g = c->deferred[i];
                if (g->dl_src && g->dl_dst == p) {
                    g->dl_src = nullptr;
                    c->deferred.erase(c->deferred.begin() + (long)i);
                } else {
                    ++i;
                }
            }
        }
        if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    }
    return PF_OK;
}

int pf_host_free(void* p) {
    if (!p) return PF_OK;
    PF_TRY(pf_host_detach(p));
    PF_HIP(hipHostFree(p));
    return PF_OK;
}

const char* pf_last_error(void) { return g_err; }

int pf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pf_create(int device, pf_ctx** out) {
    PF_CHECK(out != nullptr, PF_E_ARG, "pf_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        pf_set_error("pf_create: no HIP device visible (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return PF_E_HIP;
    }
    PF_CHECK(device >= 0 && device < count, PF_E_ARG, "pf_create: device %d out of range [0,%d)", device, count);
    PF_HIP(hipSetDevice(device));
    pf_ctx* c = new pf_ctx();
    c->device = device;
    PF_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    {
        std::lock_guard<std::mutex> lk(g_ctx_mutex);
        g_ctxs.push_back(c);
    }
    PF_HIP(hipEventCreate(&c->ev0));
    PF_HIP(hipEventCreate(&c->ev1));
    *out = c;
    return PF_OK;
}

void pf_destroy(pf_ctx* c) {
    if (!c) return;
    if (getenv("PF_DEBUG_ALLOC"))
        fprintf(stderr, "libpyfocusr_hip: ctx %p: %lld device allocations went to the driver, %zu blocks cached at the end\n", (void*)c,
                (long long)c->alloc_misses, c->free_blocks.size());
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->copy_stream) {
        hipStreamSynchronize(c->copy_stream);
        hipStreamDestroy(c->copy_stream);
    }
    if (c->worker.joinable()) {
        {
            std::unique_lock<std::mutex> lk(c->worker_mutex);
            c->worker_cv.wait(lk, [c] { return !c->worker_busy; });
            c->worker_stop = true;
            c->worker_cv.notify_all();
        }
        c->worker.join();
    }
    if (c->stream_b) {
        hipStreamSynchronize(c->stream_b);
        hipStreamDestroy(c->stream_b);
        hipEventDestroy(c->join_ev);
        if (c->fork_ev) hipEventDestroy(c->fork_ev);
        c->stream_b = nullptr;
    }
    if (c->pinned_scratch_b) hipHostFree(c->pinned_scratch_b);
    if (c->pinned_scratch_knn) hipHostFree(c->pinned_scratch_knn);
    for (auto& kv : c->free_blocks) hipFree(kv.second.p);
    for (auto& kv : c->live_blocks) hipFree(kv.first);  // graphs the caller forgot to free
    c->free_blocks.clear();
    c->live_blocks.clear();
    {
        std::lock_guard<std::mutex> lk(g_ctx_mutex);
        g_ctxs.erase(std::remove(g_ctxs.begin(), g_ctxs.end(), c), g_ctxs.end());
    }
    for (auto& sp : c->spans_pending) {
        hipEventDestroy(sp.e0);
        hipEventDestroy(sp.e1);
    }
    for (auto& pr : c->spans_free) {
        hipEventDestroy(pr.first);
        hipEventDestroy(pr.second);
    }
    if (c->stage_ring) hipHostFree(c->stage_ring);
    for (hipEvent_t ev : c->stage_ev)
        if (ev) hipEventDestroy(ev);
    for (auto& pb : c->pinned_pool) hipHostFree(pb.second);
    if (c->pinned_scratch) hipHostFree(c->pinned_scratch);
    for (hipEvent_t ev : c->event_pool) hipEventDestroy(ev);
    pf_persist_release(c);
    if (c->persist_abort) hipHostFree(c->persist_abort);
    hipEventDestroy(c->ev0);
    hipEventDestroy(c->ev1);
    hipStreamDestroy(c->stream);
    delete c;
}

void* pf_stream(pf_ctx* c) { return c ? (void*)c->stream : nullptr; }

int pf_sync(pf_ctx* c) {
    PF_CHECK(c != nullptr, PF_E_ARG, "pf_sync: ctx is NULL");
    PF_HIP(hipStreamSynchronize(c->stream));
    if (c->copy_stream) PF_HIP(hipStreamSynchronize(c->copy_stream));
    return pf_persist_check(c);
}

int pf_timing_enable(pf_ctx* c, int on) {
    PF_CHECK(c != nullptr, PF_E_ARG, "pf_timing_enable: ctx is NULL");
    c->timing = on != 0;
    c->timing_stride = on > 1 ? on : 1;  // on = N > 1: every N-th filter application carries the event pair
    c->timing_count = 0;
    return PF_OK;
}

int pf_timing_get(pf_ctx* c, pf_timing* out, int reset) {
    PF_CHECK(c != nullptr && out != nullptr, PF_E_ARG, "pf_timing_get: NULL argument");
    PF_TRY(pf_timing_collect(c));
    if (c->build_pending) {
        c->build_pending = false;
        float ms = 0.f;
        if (hipEventSynchronize(c->ev1) == hipSuccess && hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->build_ms = ms;
        else (void)hipGetLastError();
    }
    out->op_ms = c->op_ms;
    out->op_launches = c->op_launches;
    out->op_bytes = c->op_bytes;
    out->knn_ms = c->knn_ms;
    out->build_ms = c->build_ms;
    out->persist_ms = c->persist_ms;
    out->persist_launches = c->persist_launches;
    out->persist_steps = c->persist_steps;
    out->persist_bytes = c->persist_bytes;
    out->persist_lds_bytes = c->persist_lds_bytes;
    if (reset) {
        c->persist_ms = c->persist_bytes = c->persist_lds_bytes = 0.0;
        c->persist_launches = c->persist_steps = 0;
        c->op_ms = 0.0;
        c->op_launches = 0;
        c->op_bytes = 0.0;
    }
    return PF_OK;
}

}  // extern "C"
