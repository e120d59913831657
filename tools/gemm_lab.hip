// gemm_lab.hip -- stand-alone playground for the pipelined GEMM main loop (diagnostic build, not product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DV2_DIAG tools/gemm_lab.hip -o gpurun_out/gemm_lab && gpurun_out/gemm_lab
// Random bf16 operands, trivial epilogue, event timing of both schedules and s_memtime segment shares.
#ifndef LAB_NO_DIAG      // -DLAB_NO_DIAG: no stamps at all (timings comparable with the library build)
#define V2_DIAG 1
#endif
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <vector>
#include <algorithm>
#include "lab/common.h"
void vbnn_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int vbnn_cu_count() { return 256; }
#include "lab/gemm_v2.h"
#include "lab/gemm_v3.h"

struct EpiSum {          // keeps both accumulators live with one 16-byte store per four outputs
    typedef bf16_t elem_t;
    float* out; int M, N;
    __host__ __device__ __forceinline__ bf16_t* t1_ptr() const { return nullptr; }
    __host__ __device__ __forceinline__ bf16_t* t2_ptr() const { return nullptr; }
    __host__ __device__ __forceinline__ int64_t t_ld() const { return 0; }
    __device__ __forceinline__ int m_dim() const { return M; }
    __device__ __forceinline__ int n_dim() const { return N; }
    template <bool ST>
    __device__ __forceinline__ void apply(int m, int n, f32x4 a1, f32x4 a2, float (&t1)[4], float (&t2)[4]) const {
        if (m < M && n < N) *reinterpret_cast<f32x4*>(out + (size_t)n * M + m) = a1 + a2;
    }
    static constexpr int FAST_BATCH = 8;
    static constexpr int FAST_BATCH_V2 = 4;
    static constexpr bool PARK = false;
    static constexpr bool SPLITTABLE = false;
    static constexpr bool EDGE_FAST = false;
    __device__ __forceinline__ void edge_row(int, float) const {}
    __device__ __forceinline__ void set_part(int) {}
    static constexpr int FOLD_BATCH = 8;
    static constexpr int FOLD_SERIAL = 0;
    static constexpr int FOLD_STAGE = 0;
    struct Pre {};
    struct FPre {};
    struct Lane { unsigned o; };
    __device__ __forceinline__ FPre fold_load(int, int, const Lane&) const { return FPre{}; }
    __device__ __forceinline__ f32x4 fold(int, int, const Lane&, f32x4 a2, const FPre&) const { return a2; }
    __device__ __forceinline__ Pre load_folded(int, int, const Lane&) const { return Pre{}; }
    __device__ __forceinline__ void apply_folded(int um, int un, const Lane& ln, f32x4 a, f32x4, const Pre&, float (&)[4], float (&)[4]) const {
        *reinterpret_cast<f32x4*>(out + ((size_t)un * M + um) + ln.o) = a;
    }
    __host__ __device__ bool fast_ok() const { return true; }
    __host__ __device__ bool v3_ok() const { return true; }
    __device__ __forceinline__ Lane lane_init(int nl, int ml) const { return Lane{(unsigned)(nl * M + ml)}; }
    __device__ __forceinline__ Pre load_fast(int, int, const Lane&) const { return Pre{}; }
    __device__ __forceinline__ void apply_fast(int um, int un, const Lane& ln, f32x4 a1, f32x4 a2, const Pre&, float (&)[4], float (&)[4]) const {
        *reinterpret_cast<f32x4*>(out + ((size_t)un * M + um) + ln.o) = a1 + a2;
    }
};
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int lab_half(int M, int N, int K, const bf16_t* A, const bf16_t* A2, const bf16_t* B, const bf16_t* B2, float* out);

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    const size_t ea = (size_t)M * K, eb = (size_t)N * K;
    std::vector<unsigned short> ha(ea), hb(eb);
    srand(1);
    auto rnd = [] { float f = (float)rand() / RAND_MAX * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
    for (auto& v : ha) v = rnd();
    for (auto& v : hb) v = rnd();
    bf16_t *A, *A2, *B, *B2; float* out; unsigned long long* diag;
    CK(hipMalloc(&A, ea * 2)); CK(hipMalloc(&A2, ea * 2)); CK(hipMalloc(&B, eb * 2)); CK(hipMalloc(&B2, eb * 2));
    CK(hipMalloc(&out, (size_t)M * N * 4));
    CK(hipMemcpy(A, ha.data(), ea * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(A2, hb.data(), std::min(ea, eb) * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hb.data(), eb * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B2, ha.data(), std::min(ea, eb) * 2, hipMemcpyHostToDevice));
    const int tiles = ((M + 255) / 256) * ((N + 127) / 128);
    CK(hipMalloc(&diag, (size_t)tiles * 8 * 4 * 8)); CK(hipMemset(diag, 0, (size_t)tiles * 8 * 4 * 8));
#ifdef V2_DIAG
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_v2_diag), &diag, sizeof(diag)));
#endif
    EpiSum epi{out, M, N};
    hipStream_t st = 0;
    vbnn_ctx lab_ctx{};
    lab_ctx.stream = st;
    CK(hipMalloc((void**)&lab_ctx.counters, VBNN_CNT_TOTAL * 4)); CK(hipMemset(lab_ctx.counters, 0, VBNN_CNT_TOTAL * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int dual = 0; dual < 2; ++dual)
        for (int sched = 0; sched < 5; sched += 2) {
            g_v2_sched = sched;
            auto launch = [&] {
                if (dual) launch_gemm_v2<bf16_t, true, EpiSum>(&lab_ctx, A, A2, K, B, B2, K, M, N, K, epi);
                else launch_gemm_v2<bf16_t, false, EpiSum>(&lab_ctx, A, nullptr, K, B, nullptr, K, M, N, K, epi);
            };
            for (int i = 0; i < 3; ++i) launch();
            CK(hipDeviceSynchronize());
            std::vector<float> ts;
            for (int r = 0; r < 5; ++r) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < 10; ++i) launch();
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / 10 * 1e3f);
            }
            std::sort(ts.begin(), ts.end());
            const double fl = 2.0 * M * N * K * (dual ? 2 : 1);
            printf("M=%d N=%d K=%d %s sched%d: median %.1f us  %.0f TF", M, N, K, dual ? "dual  " : "single", sched, ts[2], fl / ts[2] / 1e6);
#ifdef V2_DIAG
            {
                std::vector<unsigned long long> hd((size_t)tiles * 8 * 4);
                CK(hipMemcpy(hd.data(), diag, hd.size() * 8, hipMemcpyDeviceToHost));
                double s[4] = {0, 0, 0, 0};
                for (size_t w = 0; w < (size_t)tiles * 8; ++w) for (int k = 0; k < 4; ++k) s[k] += (double)hd[w * 4 + k];
                const double tot = s[0] + s[1] + s[2] + s[3], nw = (double)tiles * 8;
                printf("   cycles/wave: wait_dma %.0f (%.0f%%) barrier %.0f (%.0f%%) issue %.0f (%.0f%%) read+mfma %.0f (%.0f%%)",
                       s[0] / nw, 100 * s[0] / tot, s[1] / nw, 100 * s[1] / tot, s[2] / nw, 100 * s[2] / tot, s[3] / nw, 100 * s[3] / tot);
            }
#endif
            printf("\n");
        }
    {   // the two-pass 256 x 256 kernel (gemm_v3.h)
        vbnn_ctx ctx{};
        ctx.stream = st;
        for (int dual = 0; dual < 2; ++dual) {
            const int km = getenv("LAB_KMAJOR") ? atoi(getenv("LAB_KMAJOR")) : 0;      // 0 NT, 1 A K-major, 2 both (timing only)
            auto launch = [&] {
                if (km == 1) launch_gemm_v3<bf16_t, true, true, false, EpiSum>(&ctx, A, A2, M, B, B2, K, M, N, K, epi);
                else if (km == 2) launch_gemm_v3<bf16_t, true, true, true, EpiSum>(&ctx, A, A2, M, B, B2, N, M, N, K, epi);
                else if (dual) launch_gemm_v3<bf16_t, true, false, false, EpiSum>(&ctx, A, A2, K, B, B2, K, M, N, K, epi);
                else launch_gemm_v3<bf16_t, false, false, false, EpiSum>(&ctx, A, nullptr, K, B, nullptr, K, M, N, K, epi);
            };
            for (int i = 0; i < 3; ++i) launch();
            CK(hipDeviceSynchronize());
            std::vector<float> ts;
            for (int r = 0; r < 5; ++r) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < 10; ++i) launch();
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / 10 * 1e3f);
            }
            std::sort(ts.begin(), ts.end());
            const double fl = 2.0 * M * N * K * (dual ? 2 : 1);
            unsigned long long hsh = 1469598103934665603ull;       // FNV-1a over the output bits: equal across builds = bitwise equal
            {
                std::vector<unsigned> ho((size_t)M * N);
                CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
                for (unsigned v : ho) { hsh ^= v; hsh *= 1099511628211ull; }
            }
            printf("M=%d N=%d K=%d %s v3 (256x256%s): median %.1f us  %.0f TF  out hash %016llx\n", M, N, K, dual ? "dual  " : "single",
                   dual ? ", two passes" : "", ts[2], fl / ts[2] / 1e6, hsh);
        }
    }
    return lab_half(M, N, K, A, A2, B, B2, out);
}

// ---------------------------------------------------------------------------------------------------------------
// Experiment (lab only): "loader half". Waves 0-3 issue ALL twelve DMAs of a tile (their own groups and their SIMD
// partners'), waves 4-7 issue none: does a wave's DMA issue overlap its partner's MFMAs when only one of the two
// carries DMAs? Same ring / swizzle / waits as gemm_nt_v2<true, 2, 4>, direct (unstaged) epilogue.
template <class Epi>
__global__ __launch_bounds__(512, 2) void gemm_lab_half(const bf16_t* __restrict__ A, const bf16_t* __restrict__ A2, int64_t lda,
                                                        const bf16_t* __restrict__ B, const bf16_t* __restrict__ B2, int64_t ldb,
                                                        int M, int N, int nk, int tiles_n, Epi epi) {
    constexpr int A_BYTES = v2_a_bytes(4), STAGE = v2_stage(4);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * 256, n0 = tn * 128;
    const int lw = wave & 3;                         // loader index
    const bf16_t* a_src[2][8];
    const bf16_t* b_src[2][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 8 * (lw + 4 * i) + (lane >> 3);
        const int64_t off = (int64_t)min(m0 + row, M - 1) * lda + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        a_src[0][i] = A + off; a_src[1][i] = A2 + off;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * (lw + 4 * i) + (lane >> 3);
        const int64_t off = (int64_t)min(n0 + row, N - 1) * ldb + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        b_src[0][i] = B + off; b_src[1][i] = B2 + off;
    }
    const int U = 2 * nk;
    const bool loader = __builtin_amdgcn_readfirstlane(wave) < 4;
    auto issue_one = [&](int u, auto pair_c, auto idx_c) {
        constexpr int P = decltype(pair_c)::value;
        constexpr int IDX = decltype(idx_c)::value;
        const int64_t koff = (int64_t)(u >> 1) * V2_BK;
        unsigned char* base = lds + (u % 3) * STAGE;
        if constexpr (IDX < 8)
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[P][IDX] + koff), (lptr_t)(base + (lw + 4 * IDX) * 1024), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((gptr_t)(b_src[P][IDX - 8] + koff), (lptr_t)(base + A_BYTES + (lw + 4 * (IDX - 8)) * 1024), 16, 0, 0);
    };
    auto issue_all = [&](int u, auto pair_c) {
        issue_one(u, pair_c, std::integral_constant<int, 0>()); issue_one(u, pair_c, std::integral_constant<int, 1>());
        issue_one(u, pair_c, std::integral_constant<int, 2>()); issue_one(u, pair_c, std::integral_constant<int, 3>());
        issue_one(u, pair_c, std::integral_constant<int, 4>()); issue_one(u, pair_c, std::integral_constant<int, 5>());
        issue_one(u, pair_c, std::integral_constant<int, 6>()); issue_one(u, pair_c, std::integral_constant<int, 7>());
        issue_one(u, pair_c, std::integral_constant<int, 8>()); issue_one(u, pair_c, std::integral_constant<int, 9>());
        issue_one(u, pair_c, std::integral_constant<int, 10>()); issue_one(u, pair_c, std::integral_constant<int, 11>());
    };
    const int rsw = (lane & 15) >> 1, q = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int csw = ((4 * s + q) ^ rsw) * 16;
        a_off[s] = (wm * 64 + (lane & 15)) * 128 + csw;
        b_off[s] = A_BYTES + (wn * 64 + (lane & 15)) * 128 + csw;
    }
    f32x4 acc1[4][4], acc2[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    std::integral_constant<int, 0> c0; std::integral_constant<int, 1> c1;
    if (loader) { issue_all(0, c0); issue_all(1, c1); }
    auto step = [&](int u, auto pair_c, f32x4 (&acc)[4][4]) {
        if (loader) {
            if (u + 1 < U) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else           asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const bool more = loader && (u + 2 < U);
        const unsigned char* stage = lds + (u % 3) * STAGE;
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
            bf16x8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const bf16x8*>(stage + a_off[sidx] + i * 16 * 128);
                bf[i] = *reinterpret_cast<const bf16x8*>(stage + b_off[sidx] + i * 16 * 128);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                    if (more && (j == 1 || j == 3)) {       // a DMA after every second MFMA of the first 24
                        const int slot = sidx * 8 + i * 2 + (j >> 1);
                        if (slot == 0) issue_one(u + 2, pair_c, std::integral_constant<int, 0>());
                        if (slot == 1) issue_one(u + 2, pair_c, std::integral_constant<int, 1>());
                        if (slot == 2) issue_one(u + 2, pair_c, std::integral_constant<int, 2>());
                        if (slot == 3) issue_one(u + 2, pair_c, std::integral_constant<int, 3>());
                        if (slot == 4) issue_one(u + 2, pair_c, std::integral_constant<int, 4>());
                        if (slot == 5) issue_one(u + 2, pair_c, std::integral_constant<int, 5>());
                        if (slot == 6) issue_one(u + 2, pair_c, std::integral_constant<int, 6>());
                        if (slot == 7) issue_one(u + 2, pair_c, std::integral_constant<int, 7>());
                        if (slot == 8) issue_one(u + 2, pair_c, std::integral_constant<int, 8>());
                        if (slot == 9) issue_one(u + 2, pair_c, std::integral_constant<int, 9>());
                        if (slot == 10) issue_one(u + 2, pair_c, std::integral_constant<int, 10>());
                        if (slot == 11) issue_one(u + 2, pair_c, std::integral_constant<int, 11>());
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    };
    for (int u = 0; u < U; u += 2) { step(u, c0, acc1); step(u + 1, c1, acc2); }
    float t1[4], t2[4];
    const int em = m0 + wm * 64 + (lane >> 4) * 4, en = n0 + wn * 64 + (lane & 15);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) epi.template apply<true>(em + i * 16, en + j * 16, acc1[i][j], acc2[i][j], t1, t2);
}

int lab_half(int M, int N, int K, const bf16_t* A, const bf16_t* A2, const bf16_t* B, const bf16_t* B2, float* out) {
    EpiSum epi{out, M, N};
    auto kern = gemm_lab_half<EpiSum>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, v2_lds(4, 3)));
    const int tm = (M + 255) / 256, tn = (N + 127) / 128, nk = K / 64;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(tm * tn), dim3(512), v2_lds(4, 3), 0, A, A2, (int64_t)K, B, B2, (int64_t)K, M, N, nk, tn, epi);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) ts.push_back(ms / 10 * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    printf("M=%d N=%d K=%d dual   loader-half: median %.1f us  %.0f TF\n", M, N, K, ts[2], 4.0 * M * N * K / ts[2] / 1e6);
    return 0;
}
