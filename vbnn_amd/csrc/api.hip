// api.hip -- context, device buffers and error reporting of the C ABI (include/vbnn_hip.h).
#include "common.h"
#include <stdarg.h>
#include <vector>

static thread_local char g_err[512] = "";

void vbnn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// compute units the shape heuristics plan for: the budget of the CONTEXT whose call is running on this thread
// (vbnn_cu_scope, entered by every entry point that selects a kernel by shape), otherwise the device's. No process-wide
// state: a full-device context beside a budgeted one plans for the whole device.
static thread_local int t_cu_plan = 0;
vbnn_cu_scope::vbnn_cu_scope(const vbnn_ctx* c) : prev(t_cu_plan) { t_cu_plan = c ? c->cu_budget : 0; }
vbnn_cu_scope::~vbnn_cu_scope() { t_cu_plan = prev; }
int vbnn_cu_count() {
    if (t_cu_plan > 0) return t_cu_plan;
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

extern "C" int vbnn_ctx_create_cu_budget(int device, int n_cus, vbnn_ctx** out) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(out, "null out");
    vbnn_ctx* c = nullptr;
    int st = vbnn_ctx_create(device, nullptr, &c);
    if (st != VBNN_OK) return st;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    const int total = e == hipSuccess ? prop.multiProcessorCount : 0;
    hipStream_t s = nullptr;
    if (n_cus > 0 && n_cus < total) {
        // bit i of the mask enables compute unit i; consecutive indices go round the XCDs, so the first n_cus bits are an
        // even share of every XCD (checked on the box by timing: a 128-unit budget doubles a 256-tile launch)
        std::vector<uint32_t> mask((size_t)(total + 31) / 32, 0u);
        for (int i = 0; i < n_cus; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
        e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
        if (e == hipSuccess) c->cu_budget = n_cus;               // this context's launches are tiled for the units they really have
    } else {
        e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    }
    if (e != hipSuccess) { (void)vbnn_ctx_destroy(c); vbnn_set_error("stream creation: %s", hipGetErrorString(e)); return VBNN_ERR_HIP; }
    c->stream = s;
    c->own_stream = true;
    *out = c;
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_ctx_stream(vbnn_ctx* ctx, void** hip_stream_out, int* n_cus_out) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx, "null ctx");
    if (hip_stream_out) *hip_stream_out = (void*)ctx->stream;
    vbnn_cu_scope plan(ctx);
    if (n_cus_out) *n_cus_out = vbnn_cu_count();
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

// ---- a step as one graph launch (include/vbnn_hip.h) -----------------------------------------------------------------
__global__ void k_sample_advance(uint32_t* draw_dev, uint32_t by) { *draw_dev += by; }

extern "C" int vbnn_sample(vbnn_ctx* ctx, uint32_t* draw_dev, uint32_t by) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && draw_dev, "argument");
    hipLaunchKernelGGL(k_sample_advance, dim3(1), dim3(1), 0, ctx->stream, draw_dev, by);
    return vbnn_check_launch("k_sample_advance");
    VBNN_API_END
}

struct vbnn_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
    hipStream_t stream;
    int device;
    int kernel_nodes, nodes;
};

extern "C" int vbnn_capture_begin(vbnn_ctx* ctx) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx, "null ctx");
    VBNN_REQUIRE(ctx->stream != nullptr, "the NULL stream cannot be captured: give the context a stream of its own");
    // relaxed: other threads of the host (an allocator, a data loader) may go on calling HIP while this stream records
    VBNN_CHECK_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_capture_end(vbnn_ctx* ctx, vbnn_graph** out) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && out, "argument");
    hipGraph_t g = nullptr;
    VBNN_CHECK_HIP(hipStreamEndCapture(ctx->stream, &g));
    if (!g) { vbnn_set_error("the capture was invalidated (an operation that cannot be recorded was issued on the stream)"); return VBNN_ERR_HIP; }
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(g); vbnn_set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return VBNN_ERR_HIP; }
    vbnn_graph* r = new vbnn_graph();
    r->graph = g; r->exec = ex; r->stream = ctx->stream; r->device = ctx->device; r->kernel_nodes = 0; r->nodes = 0;
    size_t n = 0;
    if (hipGraphGetNodes(g, nullptr, &n) == hipSuccess && n > 0) {
        std::vector<hipGraphNode_t> nodes(n);
        if (hipGraphGetNodes(g, nodes.data(), &n) == hipSuccess) {
            r->nodes = (int)n;
            for (size_t i = 0; i < n; ++i) {
                hipGraphNodeType t;
                if (hipGraphNodeGetType(nodes[i], &t) == hipSuccess && t == hipGraphNodeTypeKernel) r->kernel_nodes += 1;
            }
        }
    }
    *out = r;
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_graph_launch(vbnn_graph* g) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(g && g->exec, "null graph");
    VBNN_CHECK_HIP(hipGraphLaunch(g->exec, g->stream));
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_graph_info(vbnn_graph* g, int* kernel_nodes, int* nodes) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(g, "null graph");
    if (kernel_nodes) *kernel_nodes = g->kernel_nodes;
    if (nodes) *nodes = g->nodes;
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_graph_destroy(vbnn_graph* g) {
    VBNN_API_BEGIN
    if (!g) return VBNN_OK;
    (void)hipStreamSynchronize(g->stream);
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return VBNN_OK;
    VBNN_API_END
}
