// api.hip -- context, device buffers and error reporting of the C ABI (include/vbnn_hip.h).
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void vbnn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int vbnn_cu_count() {
    static int cus = 0;
    if (cus <= 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        else
            return 256;                                       // no device visible (host-side shape queries): MI355X
    }
    return cus;
}

extern "C" int vbnn_abi_version(void) { return VBNN_ABI_VERSION; }
extern "C" const char* vbnn_last_error(void) { return g_err; }

extern "C" int vbnn_ctx_create(int device, void* hip_stream, vbnn_ctx** out) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(out, "null out");
    int ndev = 0;
    VBNN_CHECK_HIP(hipGetDeviceCount(&ndev));
    if (ndev <= 0) { vbnn_set_error("no HIP device visible"); return VBNN_ERR_HIP; }
    VBNN_REQUIRE(device >= 0 && device < ndev, "device index");
    VBNN_CHECK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    VBNN_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        vbnn_set_error("libvbnn_hip is built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
        return VBNN_ERR_UNSUPPORTED;
    }
    vbnn_ctx* c = new vbnn_ctx();
    c->device = device;
    c->own_stream = false;
    c->stream = (hipStream_t)hip_stream;     // NULL = the device's default (null) stream
    c->scratch_doubles = 1u << 21;       // 16 MiB: block partials of the reductions
    hipError_t e = hipMalloc((void**)&c->scratch, c->scratch_doubles * sizeof(double));
    if (e != hipSuccess) { delete c; vbnn_set_error("hipMalloc(scratch): %s", hipGetErrorString(e)); return VBNN_ERR_NOMEM; }
    e = hipMalloc((void**)&c->counters, VBNN_CNT_TOTAL * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(c->counters, 0, VBNN_CNT_TOTAL * sizeof(unsigned));
    if (e != hipSuccess) {
        (void)hipFree(c->scratch); if (c->counters) (void)hipFree(c->counters); delete c;
        vbnn_set_error("hipMalloc(counters): %s", hipGetErrorString(e)); return VBNN_ERR_NOMEM;
    }
    *out = c;
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_ctx_destroy(vbnn_ctx* ctx) {
    VBNN_API_BEGIN
    if (!ctx) return VBNN_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->counters) (void)hipFree(ctx->counters);
    if (ctx->park) (void)hipFree(ctx->park);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_ctx_set_stream(vbnn_ctx* ctx, void* hip_stream) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx, "null ctx");
    if (ctx->own_stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); ctx->own_stream = false; }
    ctx->stream = (hipStream_t)hip_stream;
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_sync(vbnn_ctx* ctx) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx, "null ctx");
    VBNN_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_buf_alloc(vbnn_ctx* ctx, size_t bytes, void** dptr) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && dptr && bytes > 0, "argument");
    VBNN_CHECK_HIP(hipSetDevice(ctx->device));
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { vbnn_set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return VBNN_ERR_NOMEM; }
    e = hipMemsetAsync(p, 0, bytes, ctx->stream);
    if (e != hipSuccess) { (void)hipFree(p); vbnn_set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return VBNN_ERR_HIP; }
    *dptr = p;
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_buf_free(vbnn_ctx* ctx, void* dptr) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx, "null ctx");
    if (!dptr) return VBNN_OK;
    VBNN_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    VBNN_CHECK_HIP(hipFree(dptr));
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_buf_zero(vbnn_ctx* ctx, void* dptr, size_t bytes) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && dptr, "argument");
    if (bytes == 0) return VBNN_OK;
    VBNN_CHECK_HIP(hipMemsetAsync(dptr, 0, bytes, ctx->stream));
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_buf_upload(vbnn_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && dst_dev && src_host, "argument");
    VBNN_CHECK_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    // the host buffer is borrowed for the call only (pageable memory): do not return before it is consumed
    VBNN_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_buf_download(vbnn_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && dst_host && src_dev, "argument");
    VBNN_CHECK_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VBNN_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return VBNN_OK;
    VBNN_API_END
}
