// calib.hip -- vbnn_box_calibrate: what THIS device holds under a pure matrix load and under a pure stream, measured in the run
// that reports a throughput (bench.py's `box` block; VERDICT r04 "a box-speed reference inside the bench line").
//
// The boxes of a pool differ: the same binary has measured 0.7275 and 0.7815 ms per wide step within one day, every kernel moving
// together -- the clock a power-bound chip holds (MI355X_MICROARCH.md, DVFS give-back: devices 12 % apart on an MFMA loop with
// no memory traffic at all). A bench line that carries two fixed probes of its box lets a reader split a round-to-round delta
// into "the code" and "the box":
//   mfma   every SIMD of every CU runs 2 waves x 2^15 v_mfma_f32_16x16x32_bf16 on operands that stay in registers (random
//          bit patterns in the bf16 normal range: all-zero operands run 15-20 % faster than real data, ibid. (1)) -- no LDS, no
//          memory. Each workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its loop:
//          mfma_clock_ghz = median over workgroups of d(cycles) / d(ticks) x 0.1; mfma_tflops = flops / event time of the launch.
//          2.5 PFLOP/s is 2.4 GHz x 1024 SIMDs x 1024 flop per cycle: at the clock this probe holds the chip's ceiling is
//          2.5 x mfma_clock_ghz / 2.4 -- `roofline.frac_at_held_clock` divides by that.
//   hbm    a 512 MiB -> 512 MiB copy (1 GiB moved), 16-byte nontemporal loads and stores, grid-stride: hbm_TBps.
// Several launches each, the last ones timed (the first ones ramp the clock). Blocking, on the context's stream: measurement,
// not a part of any step. The reference has no counterpart (main.lua:20 has a commented-out sys.clock()).
#include "common.h"
#include <algorithm>
#include <vector>

namespace {
constexpr int CAL_MFMA_PER_WAVE = 1 << 15;
constexpr int CAL_WAVES = 8;                     // 512 threads: two waves per SIMD, as the GEMM kernels run

__global__ __launch_bounds__(512) void k_calib_mfma(unsigned long long* stamps, float* sink, unsigned seed) {
    // operands: pseudo-random bf16 values in [1, 2) x +-1 from a per-lane LCG -- live bit patterns, finite products
    unsigned s = seed ^ (blockIdx.x * 512u + threadIdx.x) * 2654435761u;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s = s * 1664525u + 1013904223u;
            const unsigned short ua = (unsigned short)(0x3F80u | ((s >> 9) & 0x7Fu) | ((s >> 3) & 0x8000u));
            s = s * 1664525u + 1013904223u;
            const unsigned short ub = (unsigned short)(0x3F80u | ((s >> 9) & 0x7Fu) | ((s >> 3) & 0x8000u));
            a[k][e] = __builtin_bit_cast(bf16_t, ua);
            b[k][e] = __builtin_bit_cast(bf16_t, ub);
        }
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < CAL_MFMA_PER_WAVE / 32; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(u + i) & 3], b[u], acc[i], 0, 0, 0);
        // (no rescaling: products of +-[1, 4) summed with random signs walk to ~1e3 over the 4096 MFMAs an accumulator takes)
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (t == 1.2345e-30f) sink[0] = t;                          // (never: keeps the accumulators alive)
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void k_calib_copy(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n16) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n16) __builtin_nontemporal_store(v[u], dst + i + u * stride);
    }
}
}  // namespace

extern "C" int vbnn_box_calibrate(vbnn_ctx* ctx, vbnn_box_info* out) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && out, "null ctx/out");
    VBNN_CHECK_HIP(hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    VBNN_CHECK_HIP(hipGetDeviceProperties(&prop, ctx->device));
    const int cus = prop.multiProcessorCount;
    memset(out, 0, sizeof *out);
    out->cus = cus;
    hipEvent_t e0, e1;
    VBNN_CHECK_HIP(hipEventCreate(&e0));
    VBNN_CHECK_HIP(hipEventCreate(&e1));
    int status = VBNN_OK;
    unsigned long long* stamps = nullptr;
    float* sink = nullptr;
    char* buf = nullptr;
    const size_t half = (size_t)512 << 20;
    do {
        if (hipMalloc((void**)&stamps, (size_t)cus * 2 * sizeof(unsigned long long)) != hipSuccess ||
            hipMalloc((void**)&sink, 256) != hipSuccess || hipMalloc((void**)&buf, 2 * half) != hipSuccess) {
            vbnn_set_error("vbnn_box_calibrate: scratch allocation (1 GiB + stamps) failed");
            status = VBNN_ERR_NOMEM;
            break;
        }
        // ---- matrix pipe: 60 launches back to back (~0.5 ms each: ~30 ms, the clock has settled), events around the last 20, stamps of the last
        constexpr int WARM = 40, TIMED = 20;
        for (int r = 0; r < WARM + TIMED; ++r) {
            if (r == WARM && hipEventRecord(e0, ctx->stream) != hipSuccess) { status = VBNN_ERR_HIP; break; }
            hipLaunchKernelGGL(k_calib_mfma, dim3(cus), dim3(512), 0, ctx->stream, stamps, sink, 0x9E3779B9u + r);
        }
        if (status != VBNN_OK || hipEventRecord(e1, ctx->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) {
            vbnn_set_error("vbnn_box_calibrate: matrix probe: %s", hipGetErrorString(hipGetLastError()));
            status = VBNN_ERR_HIP;
            break;
        }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> st((size_t)cus * 2);
        if (hipMemcpy(st.data(), stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) { status = VBNN_ERR_HIP; break; }
        std::vector<double> ghz;
        for (int b = 0; b < cus; ++b)
            if (st[2 * b + 1] > 0) ghz.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 0.1);
        std::sort(ghz.begin(), ghz.end());
        out->mfma_clock_ghz = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
        out->mfma_ms = ms / TIMED;
        const double flop = (double)cus * CAL_WAVES * CAL_MFMA_PER_WAVE * (2.0 * 16 * 16 * 32);
        out->mfma_tflops = out->mfma_ms > 0 ? flop / (out->mfma_ms * 1e-3) / 1e12 : 0.0;
        // ---- memory: 512 MiB -> 512 MiB, 8 copies, events around the last 5
        if (hipMemsetAsync(buf, 0x3c, 2 * half, ctx->stream) != hipSuccess) { status = VBNN_ERR_HIP; break; }
        const int64_t n16 = (int64_t)(half / 16);
        for (int r = 0; r < 8; ++r) {
            if (r == 3 && hipEventRecord(e0, ctx->stream) != hipSuccess) { status = VBNN_ERR_HIP; break; }
            // (lab, one box: 4 nontemporal 16-byte loads in flight per lane; 8 workgroups per CU 4.3 TB/s, 32: 5.0, 64: 5.1; plain loads / stores
            // or 8 in flight: 4.0-4.9; torch's copy_ of the same buffers: 5.3)
            hipLaunchKernelGGL(k_calib_copy, dim3(cus * 64), dim3(256), 0, ctx->stream, (const f32x4*)buf, (f32x4*)(buf + half), n16);
        }
        if (status != VBNN_OK || hipEventRecord(e1, ctx->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) {
            vbnn_set_error("vbnn_box_calibrate: stream probe: %s", hipGetErrorString(hipGetLastError()));
            status = VBNN_ERR_HIP;
            break;
        }
        (void)hipEventElapsedTime(&ms, e0, e1);
        out->hbm_ms = ms / 5;
        out->hbm_bytes = (int64_t)(2 * half);
        out->hbm_TBps = out->hbm_ms > 0 ? (double)out->hbm_bytes / (out->hbm_ms * 1e-3) / 1e12 : 0.0;
    } while (false);
    if (stamps) (void)hipFree(stamps);
    if (sink) (void)hipFree(sink);
    if (buf) (void)hipFree(buf);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return status;
    VBNN_API_END
}
