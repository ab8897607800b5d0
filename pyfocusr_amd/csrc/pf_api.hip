// Context, error reporting and timing for libpyfocusr_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "pf_internal.h"

static thread_local char g_err[1024] = "";

void pf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int pf_version(void) { return PF_VERSION; }

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
    PF_HIP(hipEventCreate(&c->ev0));
    PF_HIP(hipEventCreate(&c->ev1));
    *out = c;
    return PF_OK;
}

void pf_destroy(pf_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    hipFree(c->knn_ref);
    hipFree(c->knn_qry);
    hipFree(c->knn_part_d2);
    hipFree(c->knn_part_idx);
    hipFree(c->knn_idx);
    hipFree(c->knn_d2);
    hipEventDestroy(c->ev0);
    hipEventDestroy(c->ev1);
    hipStreamDestroy(c->stream);
    delete c;
}

int pf_sync(pf_ctx* c) {
    PF_CHECK(c != nullptr, PF_E_ARG, "pf_sync: ctx is NULL");
    PF_HIP(hipStreamSynchronize(c->stream));
    return PF_OK;
}

int pf_timing_enable(pf_ctx* c, int on) {
    PF_CHECK(c != nullptr, PF_E_ARG, "pf_timing_enable: ctx is NULL");
    c->timing = on != 0;
    return PF_OK;
}

int pf_timing_get(pf_ctx* c, pf_timing* out, int reset) {
    PF_CHECK(c != nullptr && out != nullptr, PF_E_ARG, "pf_timing_get: NULL argument");
    out->op_ms = c->op_ms;
    out->op_launches = c->op_launches;
    out->knn_ms = c->knn_ms;
    out->build_ms = c->build_ms;
    if (reset) {
        c->op_ms = 0.0;
        c->op_launches = 0;
    }
    return PF_OK;
}

}  // extern "C"
