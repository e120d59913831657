// launch_floor.hip -- what does a LAUNCH cost on this box, and what does a GRID BARRIER cost? (lab, not product)
// The small configuration (784-400-400-10, batch 256, fp32) is nine launches of 5-17 us each: this prices the two ways out.
//   1. N dependent launches of an empty kernel on one stream            -> us per launch (the floor a launch cannot go below)
//   2. the same nine launches captured in a hipGraph, replayed            -> us per node
//   3. a persistent kernel of G workgroups that meets at a grid barrier K times, each workgroup writing BYTES of payload
//      before the barrier and reading its neighbour's (other XCD) after it, checked:
//        a. bare barrier (no payload): relaxed agent atomics only
//        b. payload with plain stores + release fence / acquire fence (buffer_wbl2 sc1 / buffer_inv sc1)
//        c. payload with write-through (sc0 sc1) stores + vmcnt(0), loads with sc0 sc1 (no L2 write-back, no invalidate)
//   hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o tools/bin/launch_floor && tools/bin/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_empty() {}
__global__ void k_chain(const float* in, float* out) {            // one dependent load -> store per thread
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    out[i] = in[i] + 1.0f;
}

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (bounded: a workgroup that is never scheduled beside the others must not hang the box -- every wave exits)
        for (int spin = 0; spin < (1 << 22) && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spin)
            __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// MODE 0: bare; 1: plain stores + release / acquire fences; 2: write-through stores + sc1 loads
template <int MODE>
__global__ __launch_bounds__(256) void k_persist(unsigned* ctr, float* buf, int words_per_wg, int K, unsigned* bad) {
    const int G = gridDim.x, b = blockIdx.x;
    float* mine = buf + (size_t)b * words_per_wg;
    const float* other = buf + (size_t)((b + 1) % G) * words_per_wg;          // neighbouring block id = another XCD
    unsigned errs = 0;
    for (int k = 0; k < K; ++k) {
        if (MODE != 0) {
            for (int i = threadIdx.x; i < words_per_wg; i += 256) {
                const float v = (float)(k * 1000 + b);
                if (MODE == 1) mine[i] = v;
                else __builtin_nontemporal_store(v, mine + i), (void)0;
            }
        }
        if (MODE == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        }
        if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        grid_barrier(ctr, (unsigned)(k + 1) * G);
        if (MODE == 1) {
            if (threadIdx.x == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __syncthreads();
        }
        if (MODE != 0) {
            const float want = (float)(k * 1000 + (b + 1) % G);
            for (int i = threadIdx.x; i < words_per_wg; i += 256) {
                float v;
                if (MODE == 1) v = other[i];
                else v = __builtin_nontemporal_load(other + i);
                if (v != want) ++errs;
            }
        }
    }
    if (errs) atomicAdd(bad, errs);
}

// MODE 2 done properly: sc0 sc1 stores / loads through inline asm (nontemporal is not the same thing)
__global__ __launch_bounds__(256) void k_persist_sc(unsigned* ctr, float* buf, int words_per_wg, int K, unsigned* bad) {
    const int G = gridDim.x, b = blockIdx.x;
    float* mine = buf + (size_t)b * words_per_wg;
    const float* other = buf + (size_t)((b + 1) % G) * words_per_wg;
    unsigned errs = 0;
    for (int k = 0; k < K; ++k) {
        for (int i = threadIdx.x; i < words_per_wg; i += 256) {
            const float v = (float)(k * 1000 + b);
            asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(mine + i), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        grid_barrier(ctr, (unsigned)(k + 1) * G);
        const float want = (float)(k * 1000 + (b + 1) % G);
        for (int i = threadIdx.x; i < words_per_wg; i += 256) {
            float v;
            asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(other + i) : "memory");
            if (v != want) ++errs;
        }
    }
    if (errs) atomicAdd(bad, errs);
}

static float elapsed(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float *in, *out;
    CK(hipMalloc(&in, 1 << 20)); CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(in, 0, 1 << 20));
    const int N = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(13), dim3(256), 0, s);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    }
    printf("empty kernel, 13 x 256, %d dependent launches on a stream: %.2f us per launch\n", N, elapsed(e0, e1) * 1e3 / N);
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_chain, dim3(104), dim3(256), 0, s, in, out);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    }
    printf("load -> store kernel, 104 x 256: %.2f us per launch\n", elapsed(e0, e1) * 1e3 / N);
    // graph of nine nodes
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 9; ++i) hipLaunchKernelGGL(k_chain, dim3(104), dim3(256), 0, s, in, out);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 200; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    }
    printf("hipGraph of nine load -> store nodes, 200 replays: %.2f us per replay = %.2f us per node\n", elapsed(e0, e1) * 1e3 / 200,
           elapsed(e0, e1) * 1e3 / 200 / 9);
    // persistent kernel + grid barriers
    unsigned *ctr, *bad;
    CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&bad, 4));
    float* buf;
    const int G = 256, K = 200;
    CK(hipMalloc(&buf, (size_t)G * 65536));
    for (int bytes : {0, 4096, 16384, 65536}) {
        for (int mode = 0; mode < 3; ++mode) {
            if ((bytes == 0) != (mode == 0)) continue;
            float best = 1e9f;
            unsigned hb = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(ctr, 0, 4, s)); CK(hipMemsetAsync(bad, 0, 4, s));
                CK(hipEventRecord(e0, s));
                const int words = bytes / 4;
                if (mode == 0) hipLaunchKernelGGL(k_persist<0>, dim3(G), dim3(256), 0, s, ctr, buf, words, K, bad);
                if (mode == 1) hipLaunchKernelGGL(k_persist<1>, dim3(G), dim3(256), 0, s, ctr, buf, words, K, bad);
                if (mode == 2) hipLaunchKernelGGL(k_persist_sc, dim3(G), dim3(256), 0, s, ctr, buf, words, K, bad);
                CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
                const float us = elapsed(e0, e1) * 1e3f / K;
                if (us < best) best = us;
            }
            printf("persistent %d x 256, %d phases, %6d B payload per workgroup, %s: %.2f us per phase (stale words seen: %u)\n", G, K, bytes,
                   mode == 0 ? "bare barrier" : mode == 1 ? "plain stores + release/acquire fences" : "sc0 sc1 stores and loads", best, hb);
        }
    }
    return 0;
}
