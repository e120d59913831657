// v4_lab.hip -- LAB ONLY (not product): how fast is the main loop of a 256 x 256 x 64 bf16 tile with FOUR waves of
// 128 x 128 (one wave per SIMD, the whole 512-register file, accumulators for the compiler to place in AGPRs) instead of
// gemm_v3.h's eight waves of 128 x 64? Same LDS image (rows x 128 B, 16-byte chunks XOR-swizzled by (row >> 1) & 7),
// same LDS-DMA pieces, one barrier per K step (A of step t + 1 and B of step t + 2 issued during step t). NT operands,
// one accumulator, trivial epilogue (the sums, for a check against the host on small integers).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/v4_lab.hip -o tools/bin/v4_lab && tools/bin/v4_lab [M N K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <type_traits>
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int ATILE = BM * 128, BTILE = BN * 128;            // bytes per K step
constexpr int LDS_BYTES = 2 * ATILE + 3 * BTILE;             // 163840

template <int N> __device__ __forceinline__ void wait_vm();
template <> __device__ __forceinline__ void wait_vm<0>() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm<8>() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }

__global__ __launch_bounds__(256) void gemm_v4_lab(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
                                                   int M, int N, int nk, int tiles_n, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware block -> tile mapping of gemm_v3.h in its simplest form: blocks sharing an XCD take a compact run of tiles
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // LDS-DMA pieces: 32 per operand per K step (8 rows x 128 B each), 8 per wave: piece g = wave + 4 d, d = 0..7
    int a_off[8], b_off[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const int row = 8 * (wave + 4 * d) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        a_off[d] = (m0 + row) * lda + chunk * 8;
        b_off[d] = (n0 + row) * ldb + chunk * 8;
    }
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7fffffff, 0x00020000);
    auto dma_a = [&](unsigned stage_off, int k_bytes, auto d_c) {
        constexpr int D = decltype(d_c)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(lds + stage_off + (wave + 4 * D) * 1024), 16, 2 * a_off[D], k_bytes, 0, 0);
    };
    auto dma_b = [&](unsigned stage_off, int k_bytes, auto d_c) {
        constexpr int D = decltype(d_c)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(lds + 2 * ATILE + stage_off + (wave + 4 * D) * 1024), 16, 2 * b_off[D], k_bytes, 0, 0);
    };
    std::integral_constant<int, 0> c0; std::integral_constant<int, 1> c1; std::integral_constant<int, 2> c2; std::integral_constant<int, 3> c3;
    std::integral_constant<int, 4> c4; std::integral_constant<int, 5> c5; std::integral_constant<int, 6> c6; std::integral_constant<int, 7> c7;

    // fragment reads: lane (r = lane & 15, q = lane >> 4): row r of the 16-row block, 16-byte chunk (4 s + q) ^ ((r >> 1) & 7)
    const int rsw = (lane & 15) >> 1, q = lane >> 4;
    int a_rd[2], b_rd[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int csw = ((4 * s + q) ^ rsw) * 16;
        a_rd[s] = (wr * 128 + (lane & 15)) * 128 + csw;
        b_rd[s] = (wc * 128 + (lane & 15)) * 128 + csw;
    }
#ifdef LAB_MFMA32
    // 32 x 32 x 16: lane (r = lane & 31, h = lane >> 5) reads row r of a 32-row block, 16-byte chunk (2 kk + h) ^ ((r >> 1) & 7)
    int a_rd32[4], b_rd32[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int csw = ((2 * kk + (lane >> 5)) ^ ((lane & 31) >> 1 & 7)) * 16;
        a_rd32[kk] = (wr * 128 + (lane & 31)) * 128 + csw;
        b_rd32[kk] = (wc * 128 + (lane & 31)) * 128 + csw;
    }
    f32x16 acc32[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
#endif

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: A(0), B(0), B(1)
    dma_a(0, 0, c0); dma_a(0, 0, c1); dma_a(0, 0, c2); dma_a(0, 0, c3); dma_a(0, 0, c4); dma_a(0, 0, c5); dma_a(0, 0, c6); dma_a(0, 0, c7);
    dma_b(0, 0, c0); dma_b(0, 0, c1); dma_b(0, 0, c2); dma_b(0, 0, c3); dma_b(0, 0, c4); dma_b(0, 0, c5); dma_b(0, 0, c6); dma_b(0, 0, c7);
    if (nk > 1) {
        dma_b(BTILE, 128, c0); dma_b(BTILE, 128, c1); dma_b(BTILE, 128, c2); dma_b(BTILE, 128, c3);
        dma_b(BTILE, 128, c4); dma_b(BTILE, 128, c5); dma_b(BTILE, 128, c6); dma_b(BTILE, 128, c7);
    }
    unsigned oa = 0, ob = 0, ob2 = 2 * BTILE;
    auto kstep = [&](const int t, auto tail_c) {
        constexpr bool TAIL = decltype(tail_c)::value;          // steady state: every DMA unconditional, constant waits
        const bool n1 = TAIL ? (t + 1 < nk) : true, n2 = TAIL ? (t + 2 < nk) : true;
        if (n1) wait_vm<8>(); else wait_vm<0>();               // at most B(t + 1)'s eight pieces still in flight
        __builtin_amdgcn_s_barrier();
        const unsigned oa_next = oa ^ ATILE;
        const int ka1 = (t + 1) * 128, kb2 = (t + 2) * 128;
        const unsigned char* sa = lds + oa;
        const unsigned char* sb = lds + 2 * ATILE + ob;
        auto slot = [&](int g) {                               // sixteen DMA slots per step: A(t + 1) first, then B(t + 2)
            if (n1) {
                if (g == 0) dma_a(oa_next, ka1, c0); if (g == 1) dma_a(oa_next, ka1, c1); if (g == 2) dma_a(oa_next, ka1, c2); if (g == 3) dma_a(oa_next, ka1, c3);
                if (g == 4) dma_a(oa_next, ka1, c4); if (g == 5) dma_a(oa_next, ka1, c5); if (g == 6) dma_a(oa_next, ka1, c6); if (g == 7) dma_a(oa_next, ka1, c7);
            }
            if (n2) {
                if (g == 8) dma_b(ob2, kb2, c0); if (g == 9) dma_b(ob2, kb2, c1); if (g == 10) dma_b(ob2, kb2, c2); if (g == 11) dma_b(ob2, kb2, c3);
                if (g == 12) dma_b(ob2, kb2, c4); if (g == 13) dma_b(ob2, kb2, c5); if (g == 14) dma_b(ob2, kb2, c6); if (g == 15) dma_b(ob2, kb2, c7);
            }
        };
#ifdef LAB_MFMA32
        // four k-substeps of 16; fragments of substep kk + 1 read under the sixteen MFMAs of substep kk
        bf16x8 af[2][4], bf[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { bf[0][j] = *reinterpret_cast<const bf16x8*>(sb + b_rd32[0] + j * 4096); af[0][j] = *reinterpret_cast<const bf16x8*>(sa + a_rd32[0] + j * 4096); }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (kk < 3) {
                    af[(kk + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(sa + a_rd32[kk < 3 ? kk + 1 : 0] + i * 4096);
                    bf[(kk + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(sb + b_rd32[kk < 3 ? kk + 1 : 0] + i * 4096);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][i], bf[kk & 1][j], acc32[i][j], 0, 0, 0);
                slot(4 * kk + i);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#else
        // both k-halves' fragments double-buffered in registers: the second half's sixteen reads fly under the first
        // half's 64 MFMAs; only the first half's reads are exposed (behind the barrier, once per K step)
        bf16x8 af[2][8], bf[2][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bf[0][j] = *reinterpret_cast<const bf16x8*>(sb + b_rd[0] + j * 2048);
#pragma unroll
        for (int i = 0; i < 8; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(sa + a_rd[0] + i * 2048);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (s == 0) {                                   // one A and one B fragment of the other k-half per group
                    af[1][i] = *reinterpret_cast<const bf16x8*>(sa + a_rd[1] + i * 2048);
                    bf[1][i] = *reinterpret_cast<const bf16x8*>(sb + b_rd[1] + i * 2048);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s][i], bf[s][j], acc[i][j], 0, 0, 0);
                slot(8 * s + i);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
        oa = oa_next;
        ob = ob == 2 * BTILE ? 0 : ob + BTILE;
        ob2 = ob2 == 2 * BTILE ? 0 : ob2 + BTILE;
    };
    int t = 0;
    for (; t + 2 < nk; ++t) kstep(t, std::false_type());
    for (; t < nk; ++t) kstep(t, std::true_type());
#ifdef LAB_MFMA32
    // 32 x 32 accumulator: lane l holds column n = l & 31, rows m = 8 (e >> 2) + 4 (l >> 5) + (e & 3), e = 0..15
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const int m = m0 + wr * 128 + 32 * i + 8 * e4 + 4 * (lane >> 5), n = n0 + wc * 128 + 32 * j + (lane & 31);
                *reinterpret_cast<f32x4*>(out + (size_t)n * M + m) = f32x4{acc32[i][j][4 * e4], acc32[i][j][4 * e4 + 1], acc32[i][j][4 * e4 + 2], acc32[i][j][4 * e4 + 3]};
            }
    (void)acc;
#else
    // epilogue: lane (c16 = lane & 15 -> n, q4 = lane >> 4 -> 4 consecutive m): out[n][m .. m + 3]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = m0 + wr * 128 + 16 * i + (lane >> 4) * 4, n = n0 + wc * 128 + 16 * j + (lane & 15);
            *reinterpret_cast<f32x4*>(out + (size_t)n * M + m) = acc[i][j];
        }
#endif
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    if (M % BM || N % BN || K % BK) { printf("sizes must be multiples of %d x %d x %d\n", BM, BN, BK); return 1; }
    std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K);
    srand(7);
    auto small = [] { float f = (float)(rand() % 5 - 2); unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };   // -2..2, exact in bf16
    auto rnd = [] { float f = (float)rand() / (float)RAND_MAX * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
    const bool check = (size_t)M * N * K <= (size_t)1024 * 1024 * 1024;
    for (auto& v : ha) v = check ? small() : rnd();
    for (auto& v : hb) v = check ? small() : rnd();
    bf16_t *a, *b; float* out;
    CK(hipMalloc(&a, ha.size() * 2 + 4096)); CK(hipMalloc(&b, hb.size() * 2 + 4096)); CK(hipMalloc(&out, (size_t)M * N * 4));
    CK(hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)gemm_v4_lab, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    const int tiles_m = M / BM, tiles_n = N / BN;
    auto launch = [&] { hipLaunchKernelGGL(gemm_v4_lab, dim3(tiles_m * tiles_n), dim3(256), LDS_BYTES, 0, a, K, b, K, M, N, K / BK, tiles_n, out); };
    launch();
    CK(hipDeviceSynchronize());
    if (check) {
        std::vector<float> ho((size_t)M * N);
        CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
        auto f = [](unsigned short h) { unsigned u = (unsigned)h << 16; float x; memcpy(&x, &u, 4); return x; };
        size_t bad = 0;
        for (int n = 0; n < N; n += 7)
            for (int m = 0; m < M; m += 5) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += (double)f(ha[(size_t)m * K + k]) * f(hb[(size_t)n * K + k]);
                if ((double)ho[(size_t)n * M + m] != s) { if (bad < 5) printf("mismatch at m %d n %d: %f vs %f\n", m, n, ho[(size_t)n * M + m], s); ++bad; }
            }
        printf("check %d x %d x %d: %zu mismatches\n", M, N, K, bad);
        if (bad) return 2;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / 10 * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    printf("v4 lab %d x %d x %d: median %.1f us = %.3f PFLOP/s\n", M, N, K, ts[2], 2.0 * M * N * K / (ts[2] * 1e-6) / 1e15);
    return 0;
}
