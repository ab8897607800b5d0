// Context, error reporting and timing for libpyfocusr_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <mutex>

#include "pf_internal.h"

static thread_local char g_err[1024] = "";
static std::mutex g_ctx_mutex;
static std::vector<pf_ctx*> g_ctxs;

static pf_ctx* ctx_of_stream(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_ctx_mutex);
    for (pf_ctx* c : g_ctxs)
        if (c->stream == st) return c;
    return nullptr;
}

int pf_pinned_scratch(pf_ctx* c, size_t bytes, void** out) {
    if (bytes > c->pinned_scratch_bytes) {
        if (c->pinned_scratch) {
            PF_HIP(hipStreamSynchronize(c->stream));
            PF_HIP(hipHostFree(c->pinned_scratch));
            c->pinned_scratch = nullptr;
            c->pinned_scratch_bytes = 0;
        }
        const size_t cap = bytes > ((size_t)1 << 16) ? bytes : ((size_t)1 << 16);
        PF_HIP(hipHostMalloc(&c->pinned_scratch, cap, hipHostMallocDefault));
        c->pinned_scratch_bytes = cap;
    }
    *out = c->pinned_scratch;
    return PF_OK;
}

hipError_t pf_malloc(hipStream_t st, void** p, size_t bytes) {
    *p = nullptr;
    pf_ctx* c = ctx_of_stream(st);
    if (!c) return hipErrorInvalidValue;
    bytes = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
    auto it = c->free_blocks.find(bytes);
    if (it != c->free_blocks.end()) {
        *p = it->second;
        c->free_blocks.erase(it);
    } else {
        hipError_t e = hipMalloc(p, bytes);
        if (e != hipSuccess) {  // give cached blocks back to the driver and retry once
            (void)hipStreamSynchronize(st);
            for (auto& kv : c->free_blocks) (void)hipFree(kv.second);
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
    pf_ctx* c = ctx_of_stream(st);
    if (!c) return;
    auto it = c->live_blocks.find(p);
    if (it == c->live_blocks.end()) return;  // not ours (or already released)
    c->free_blocks.emplace(it->second, p);
    c->live_blocks.erase(it);
}

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

int pf_host_free(void* p) {
    if (p) PF_HIP(hipHostFree(p));
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
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->copy_stream) {
        hipStreamSynchronize(c->copy_stream);
        hipStreamDestroy(c->copy_stream);
    }
    for (auto& kv : c->free_blocks) hipFree(kv.second);
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
    return PF_OK;
}

int pf_timing_get(pf_ctx* c, pf_timing* out, int reset) {
    PF_CHECK(c != nullptr && out != nullptr, PF_E_ARG, "pf_timing_get: NULL argument");
    PF_TRY(pf_timing_collect(c));
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
