// gap_lab.hip -- LAB: which property of a kernel makes the NEXT dependent launch on the same stream start ~5.6 us after it ends
// (rocprofv3 --kernel-trace: start(B) - end(A)), where most kernel pairs show 0.0? Each variant kernel runs ~30 us and is followed by
// an empty kernel; read the trace with tools/gap_probe_read.py.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gap_lab.hip -o tools/bin/gap_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ void spin(unsigned long long ticks) {       // s_memrealtime: 100 MHz
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void k_empty(float* p) { if (p && threadIdx.x == 1000) p[0] = 1.f; }
__global__ __launch_bounds__(256) void k_a_plain256(float* p) { spin(3000); if (p && threadIdx.x == 1000) p[0] = 1.f; }
__global__ __launch_bounds__(512) void k_b_plain512(float* p) { spin(3000); if (p && threadIdx.x == 1000) p[0] = 1.f; }
__global__ __launch_bounds__(1024) void k_c_plain1024(float* p) { spin(3000); if (p && threadIdx.x == 2000) p[0] = 1.f; }
__global__ __launch_bounds__(512) void k_d_dynlds(float* p) {
    extern __shared__ float sm[];
    sm[threadIdx.x] = (float)threadIdx.x; __syncthreads();
    spin(3000);
    if (p && sm[(threadIdx.x + 1) & 511] < 0.f) p[0] = 1.f;
}
__global__ __launch_bounds__(512) void k_e_prio(float* p) { __builtin_amdgcn_s_setprio(1); spin(3000); if (p && threadIdx.x == 1000) p[0] = 1.f; }
__global__ __launch_bounds__(256) void k_f_stores(float* p, size_t n) {          // plain stores, 128 MB
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 1.f;
}
__global__ __launch_bounds__(256) void k_g_wt_stores(float* p, size_t n) {       // write-through (agent-scope atomic) stores
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __hip_atomic_store(p + i, 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void k_h_nt_stores(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(1.f, p + i);
}
__global__ __launch_bounds__(256) void k_i_fence(float* p) { spin(3000); __threadfence(); if (p && threadIdx.x == 1000) p[0] = 1.f; }
__global__ __launch_bounds__(256) void k_j_atomic(unsigned* c) { spin(3000); if (threadIdx.x == 0) atomicAdd(c, 1u); }
__global__ __launch_bounds__(512, 2) void k_k_bigregs(float* p) {                  // many registers (2 waves per SIMD)
    float v[160];
#pragma unroll
    for (int i = 0; i < 160; ++i) v[i] = p ? p[i] : (float)i;
    spin(3000);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 160; ++i) s += v[i] * v[(i + 7) % 160];
    if (p && s == 12345.f) p[0] = s;
}
// a whole chip's stores in the launch's LAST microseconds: every workgroup spins, then stores 384 KB (what a two-pass GEMM's epilogue does)
template <int MODE>
__global__ __launch_bounds__(512) void k_l_endburst(float* p) {
    spin(3000);
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4* q = reinterpret_cast<f4*>(p) + (size_t)blockIdx.x * (384 * 1024 / 16);
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (int i = threadIdx.x; i < 384 * 1024 / 16; i += 512) {
        if (MODE == 0) q[i] = v;
        else if (MODE == 1) __builtin_nontemporal_store(v, q + i);
        else { __hip_atomic_store(reinterpret_cast<unsigned long long*>(q + i), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
               __hip_atomic_store(reinterpret_cast<unsigned long long*>(q + i) + 1, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
}
int main() {
    float* buf; unsigned* cnt;
    const size_t n = 32u << 20;
    CK(hipMalloc(&buf, n * 4)); CK(hipMalloc(&cnt, 64)); CK(hipMemset(cnt, 0, 64)); CK(hipMemset(buf, 0, n * 4));
    CK(hipFuncSetAttribute((const void*)k_d_dynlds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int rep = 0; rep < 12; ++rep) {
        hipLaunchKernelGGL(k_a_plain256, dim3(1024), dim3(256), 0, s, (float*)nullptr); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (float*)nullptr);
        hipLaunchKernelGGL(k_b_plain512, dim3(256), dim3(512), 0, s, (float*)nullptr); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (float*)nullptr);
        hipLaunchKernelGGL(k_c_plain1024, dim3(256), dim3(1024), 0, s, (float*)nullptr); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (float*)nullptr);
        hipLaunchKernelGGL(k_d_dynlds, dim3(256), dim3(512), 160 * 1024, s, (float*)nullptr); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (float*)nullptr);
        hipLaunchKernelGGL(k_e_prio, dim3(256), dim3(512), 0, s, (float*)nullptr); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (float*)nullptr);
        hipLaunchKernelGGL(k_f_stores, dim3(2048), dim3(256), 0, s, buf, n); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (float*)nullptr);
        hipLaunchKernelGGL(k_g_wt_stores, dim3(2048), dim3(256), 0, s, buf, n); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (float*)nullptr);
        hipLaunchKernelGGL(k_h_nt_stores, dim3(2048), dim3(256), 0, s, buf, n); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (float*)nullptr);
        hipLaunchKernelGGL(k_i_fence, dim3(1024), dim3(256), 0, s, (float*)nullptr); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (float*)nullptr);
        hipLaunchKernelGGL(k_j_atomic, dim3(1024), dim3(256), 0, s, cnt); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (float*)nullptr);
        hipLaunchKernelGGL(k_k_bigregs, dim3(256), dim3(512), 0, s, (float*)nullptr); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (float*)nullptr);
        hipLaunchKernelGGL(k_l_endburst<0>, dim3(256), dim3(512), 0, s, buf); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (float*)nullptr);
        hipLaunchKernelGGL(k_l_endburst<1>, dim3(256), dim3(512), 0, s, buf); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (float*)nullptr);
        hipLaunchKernelGGL(k_l_endburst<2>, dim3(256), dim3(512), 0, s, buf); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (float*)nullptr);
        CK(hipStreamSynchronize(s));
    }
    printf("done\n");
    return 0;
}
