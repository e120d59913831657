// gemm_v1.h -- register-staged, LDS-tiled MFMA GEMM  C[m][n] = sum_k A[m][k] * Bt[n][k]
// (optionally a second accumulator from A2 / Bt2 sharing the tile) with a fused epilogue.
//
// This is the general kernel: any M, N (row-clamped loads, masked epilogue), element type
// f32 (v_mfma_f32_16x16x4_f32, exact fp32 products and sums) or bf16
// (v_mfma_f32_16x16x32_bf16, fp32 accumulate). It is THE kernel of the fp32 parity
// configuration and the fallback of the bf16 path for shapes the pipelined kernel
// (gemm_v2.h) does not take.
//
// Geometry: 256 threads = 4 waves (2 x 2), block tile (32*WR) x (32*WR), WR = 2 or 4 MFMA
// tiles per wave per dimension. K step = KS x 64 bytes of K per row: KS = 1 for the 128 x 128 tile;
// KS = 2 for the 64 x 64 tile of small problems (tens of blocks on 256 CUs, so every K step is a full
// global-load latency: half as many barrier / latency round trips).
// Operands are "packed": K contiguous, leading dimension padded to VBNN_KPAD = 64 elements with
// zeros, so the K loop needs no tail handling in either geometry.
#pragma once
#include "common.h"

template <typename T> struct Frag;
template <> struct Frag<float> { typedef f32x4 type; };
template <> struct Frag<bf16_t> { typedef bf16x8 type; };

template <typename T>
__device__ __forceinline__ f32x4 mfma_step(const typename Frag<T>::type& a, const typename Frag<T>::type& b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mfma_step<float>(const f32x4& a, const f32x4& b, f32x4 c) {
    // lane (i = l&15, q = l>>4) holds k = 4q..4q+3 of its row; MFMA step j contracts the four
    // k values {4q + j}: a permutation of k inside the 16-wide K step, identical for A and B.
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
    return c;
}
template <>
__device__ __forceinline__ f32x4 mfma_step<bf16_t>(const bf16x8& a, const bf16x8& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <typename T, bool DUAL, int WR, int KS, class Epi>
__global__ __launch_bounds__(256) void gemm_nt_v1(const T* __restrict__ A, const T* __restrict__ A2, int64_t lda,
                                                  const T* __restrict__ B, const T* __restrict__ B2, int64_t ldb,
                                                  int M, int N, int Kp, Epi epi) {
    constexpr int BT = 32 * WR;                       // block tile rows (M and N)
    constexpr int RB = 64 * KS;                       // bytes of K per row per step
    constexpr int PITCH = RB + 16;                    // + 16 B pad: conflict-free ds_read_b128 down a column
    constexpr int KE = RB / (int)sizeof(T);           // K elements per step
    constexpr int CE = 16 / (int)sizeof(T);           // elements per 16-byte chunk
    constexpr int CPR = RB / 16;                      // chunks per row
    constexpr int CHUNKS = BT * CPR;                  // 16-byte chunks per operand tile
    constexpr int CPT = CHUNKS / 256;                 // chunks per thread per operand tile
    constexpr int NOP = DUAL ? 4 : 2;
    static_assert(CHUNKS % 256 == 0, "tile does not divide over 256 threads");
    typedef typename Frag<T>::type frag_t;

    __shared__ __attribute__((aligned(16))) unsigned char lds[NOP * BT * PITCH];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BT, n0 = blockIdx.y * BT;

    // PD: register prefetch depth, in K steps. The 32 x 32 tile of latency-bound sizes does so little per step that each
    // step is one exposed global-load latency (the 784-deep forward of the small MLP: 25 steps, 37 us); four steps of
    // loads in flight cost 16 registers each and hide most of it. Larger tiles keep one step ahead.
    constexpr int PD = (WR == 1) ? 4 : 1;
    const T* src[4] = {A, B, A2, B2};
    // (a native vector type, not HIP's uint4 struct: an array of the latter is not promoted to registers -- it lived in scratch)
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t stage[PD][NOP][CPT];
    const T* gptr[NOP][CPT];
    int lds_off[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int chunk = tid + c * 256;
        const int row = chunk / CPR, kc = chunk % CPR;
        lds_off[c] = row * PITCH + kc * 16;
#pragma unroll
        for (int op = 0; op < NOP; ++op) {
            const bool isA = (op & 1) == 0;
            const int grow = isA ? min(m0 + row, M - 1) : min(n0 + row, N - 1);
            gptr[op][c] = src[op] + (int64_t)grow * (isA ? lda : ldb) + kc * CE;
        }
    }

    f32x4 acc1[WR][WR], acc2[WR][WR];
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int j = 0; j < WR; ++j) { acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int nk = Kp / KE;
#pragma unroll
    for (int d = 0; d < PD; ++d)
#pragma unroll
        for (int op = 0; op < NOP; ++op)
#pragma unroll
            for (int c = 0; c < CPT; ++c)
                stage[d][op][c] = *reinterpret_cast<const u32x4_t*>(gptr[op][c] + (int64_t)min(d, nk - 1) * KE);

    const int a_row = (wm * WR * 16 + (lane & 15)) * PITCH + (lane >> 4) * 16;
    const int b_row = (wn * WR * 16 + (lane & 15)) * PITCH + (lane >> 4) * 16;

    // (the K walk is unrolled by PD so that every register set has a compile-time index; same k order as ever)
    for (int kt0 = 0; kt0 < nk; kt0 += PD) {
#pragma unroll
      for (int d = 0; d < PD; ++d) {
        const int kt = kt0 + d;
        if (kt < nk) {                                // block-uniform (no `break`: the unrolled body must keep constant indices)
#pragma unroll
        for (int op = 0; op < NOP; ++op)
#pragma unroll
            for (int c = 0; c < CPT; ++c)
                *reinterpret_cast<u32x4_t*>(lds + op * BT * PITCH + lds_off[c]) = stage[d][op][c];
        __syncthreads();
        if (kt + PD < nk) {
#pragma unroll
            for (int op = 0; op < NOP; ++op)
#pragma unroll
                for (int c = 0; c < CPT; ++c)
                    stage[d][op][c] = *reinterpret_cast<const u32x4_t*>(gptr[op][c] + (int64_t)(kt + PD) * KE);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {             // 64 bytes of K per row per sub-step
            frag_t af[WR], bf[WR];
#pragma unroll
            for (int i = 0; i < WR; ++i) {
                af[i] = *reinterpret_cast<const frag_t*>(lds + 0 * BT * PITCH + a_row + i * 16 * PITCH + ks * 64);
                bf[i] = *reinterpret_cast<const frag_t*>(lds + 1 * BT * PITCH + b_row + i * 16 * PITCH + ks * 64);
            }
#pragma unroll
            for (int i = 0; i < WR; ++i)
#pragma unroll
                for (int j = 0; j < WR; ++j) acc1[i][j] = mfma_step<T>(af[i], bf[j], acc1[i][j]);
            if (DUAL) {
#pragma unroll
                for (int i = 0; i < WR; ++i) {
                    af[i] = *reinterpret_cast<const frag_t*>(lds + 2 * BT * PITCH + a_row + i * 16 * PITCH + ks * 64);
                    bf[i] = *reinterpret_cast<const frag_t*>(lds + 3 * BT * PITCH + b_row + i * 16 * PITCH + ks * 64);
                }
#pragma unroll
                for (int i = 0; i < WR; ++i)
#pragma unroll
                    for (int j = 0; j < WR; ++j) acc2[i][j] = mfma_step<T>(af[i], bf[j], acc2[i][j]);
            }
        }
        __syncthreads();
        }
      }
    }

    const int em = m0 + wm * WR * 16 + (lane >> 4) * 4;
    const int en = n0 + wn * WR * 16 + (lane & 15);
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int j = 0; j < WR; ++j) epi(em + i * 16, en + j * 16, acc1[i][j], acc2[i][j]);
}

template <typename T, bool DUAL, class Epi>
static int launch_gemm_v1(hipStream_t stream, const T* A, const T* A2, int64_t lda, const T* B, const T* B2,
                          int64_t ldb, int M, int N, int K, const Epi& epi) {
    const long blocks128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (blocks128 >= 128) {
        const int KE = 64 / (int)sizeof(T);
        const int Kp = (K + KE - 1) / KE * KE;
        if (lda < Kp || ldb < Kp) {
            vbnn_set_error("packed leading dimension too small: lda=%lld ldb=%lld need >= %d", (long long)lda, (long long)ldb, Kp);
            return VBNN_ERR_INVALID;
        }
        dim3 grid((M + 127) / 128, (N + 127) / 128);
        hipLaunchKernelGGL((gemm_nt_v1<T, DUAL, 4, 1, Epi>), grid, dim3(256), 0, stream, A, A2, lda, B, B2, ldb, M, N, Kp, epi);
    } else {
        constexpr int KS = 2;                                     // 128 B of K per row per step: 32 f32 / 64 bf16
        constexpr int KE2 = 128 / (int)sizeof(T);                 // (4 x 64 x 144 B of LDS for the dual tile)
        const int Kp = (K + KE2 - 1) / KE2 * KE2;
        if (lda < Kp || ldb < Kp) {
            vbnn_set_error("packed leading dimension too small: lda=%lld ldb=%lld need >= %d", (long long)lda, (long long)ldb, Kp);
            return VBNN_ERR_INVALID;
        }
        const long blocks64 = (long)((M + 63) / 64) * ((N + 63) / 64);
        if (blocks64 >= 96) {
            dim3 grid((M + 63) / 64, (N + 63) / 64);
            hipLaunchKernelGGL((gemm_nt_v1<T, DUAL, 2, KS, Epi>), grid, dim3(256), 0, stream, A, A2, lda, B, B2, ldb, M, N, Kp, epi);
        } else {        // latency-bound sizes (the 256 x 400 outputs of the small MLP): 32 x 32 tiles, 4x the blocks
            dim3 grid((M + 31) / 32, (N + 31) / 32);
            hipLaunchKernelGGL((gemm_nt_v1<T, DUAL, 1, KS, Epi>), grid, dim3(256), 0, stream, A, A2, lda, B, B2, ldb, M, N, Kp, epi);
        }
    }
    return vbnn_check_launch("gemm_nt_v1");
}
