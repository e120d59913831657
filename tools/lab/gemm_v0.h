// gemm_v0.h -- the LATENCY kernel of the fp32 launch-bound sizes (BASELINE configs[1]: 784-400-400-10 at batch 256).
//
// At these sizes a GEMM is a few hundred MFLOP -- two microseconds of the chip's fp32 matrix rate -- and what a launch
// costs is (a) the LENGTH OF ITS DEPENDENCY CHAIN and (b) the bytes its busiest CU pulls through its vector L1
// (measured, tools/v0_lab: ~25 B/clk per CU whatever the access pattern, hit or miss). gemm_v1's 32 x 32 tile walks K in
// 25 steps of (global load -> LDS write -> barrier -> LDS read -> 16 MFMAs), 0.6 us each, on 104 of the 256 CUs. This
// kernel shortens the chain and spreads the bytes instead:
//   * one (16 FM) x (16 FN) output tile per workgroup, FM, FN in {1, 2} chosen so that there are about as many tiles
//     as CUs (200 tiles of 16 x 32 for a 400 x 256 output);
//   * the K walk is SPLIT over the workgroup's four waves (wave w takes the 16-wide k groups w, w + 4, ...), so a wave's
//     chain is a quarter as long;
//   * a wave loads its MFMA fragments STRAIGHT FROM GLOBAL MEMORY in the layout v_mfma_f32_16x16x4_f32 wants them (lane
//     (i, q) holds row i, k = 4 q .. 4 q + 3 of a group: one 16-byte load of a K-contiguous operand, four 4-byte loads of
//     a K-major one): no LDS staging and no barrier anywhere in the walk, V0_PD groups of loads in flight per wave;
//   * the four partial tiles meet in LDS once (one barrier) and are added in wave order -- a fixed order: results are
//     reproducible bit for bit; wave f then runs the functor's epilogue on fragment f, whose own operands it fetched
//     before the walk (the functor's FAST protocol, as gemm_v1.h's 32 x 32 tile does).
// gemm_v1.h's launcher sends the fp32 shapes of its 32 x 32 geometry here (vbnn_debug_set(VBNN_DEBUG_V0, 0) keeps them on
// gemm_v1). Same operand forms as gemm_v1.h (TA / TB K-major, SQ in-register squares, K mask, synthetic ones row) and the same
// k permutation inside a 16-wide group; the SUMMATION ORDER differs from gemm_v1's (four interleaved chains instead of
// one), so the two kernels agree to rounding, not bitwise. fp32 only.
#pragma once
#include <type_traits>
#include "common.h"

// lane (i = l & 15, q = l >> 4) holds k = 4 q .. 4 q + 3 of its row; MFMA step j contracts the four k values {4 q + j}:
// a permutation of k inside the 16-wide group, identical for A and B (and the one gemm_v1.h uses)
__device__ __forceinline__ f32x4 v0_mfma16(const f32x4& a, const f32x4& b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
    return c;
}

#ifndef V0_LAB_PD
#define V0_LAB_PD 2          // (lab: 2 <= 4 < 8 < 16 in time at every shape of the small MLP: the walk is issue-bound, not latency-bound)
#endif
#ifndef V0_LAB_W
#define V0_LAB_W 4          // (lab, r03: 8 waves = an eighth of the K walk each: K = 784 forward 6.39 against 6.53 us, every other shape +-0.1 -- the walk is not what a launch waits for)
#endif
constexpr int V0_W = V0_LAB_W;       // waves of a workgroup = K splits
constexpr int V0_PD = V0_LAB_PD;     // k groups of loads in flight per wave
// functors whose epilogue draws noise that depends on indices alone offer draw_fast / apply_fast_z (EpiFwd): detected here
template <class E, class = void> struct v0_pre_noise : std::false_type { struct type {}; };
template <class E> struct v0_pre_noise<E, std::void_t<typename E::Noise>> : std::true_type { typedef typename E::Noise type; };

template <int FM, int FN> constexpr int v0_red_slots() { return V0_W * FM * FN * 2 * 64; }      // f32x4 slots of the reduction buffer

template <int FM, int FN, bool DUAL, class Epi, bool TA, bool TB, int SQ>
__device__ __forceinline__ void gemm_v0_tile(const float* __restrict__ A, const float* __restrict__ A2, int64_t lda,
                                             const float* __restrict__ B, const float* __restrict__ B2, int64_t ldb,
                                             int M, int N, int K, int ones_row, Epi& epi, int bx, int by,
                                             f32x4* __restrict__ red) {
    static_assert(SQ == 0 || DUAL, "SQ derives the pair's second operand");
    static_assert(FM * FN <= V0_W, "one epilogue wave per fragment");
    constexpr int NOP = DUAL ? 4 : 2;
    constexpr int FX = FM > FN ? FM : FN;
    constexpr int NF = FM * FN;
    epi.bind_draw();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int li = lane & 15, kq = lane >> 4;
    const int m0 = bx * 16 * FM, n0 = by * 16 * FN;
    const int ng = (K + 15) >> 4;                                 // 16-wide k groups
    const int ns = (ng - wave + V0_W - 1) / V0_W;                 // ... of which this wave takes w, w + 4, ...: wave-uniform, may be 0

    auto loaded = [](int op) { return !((SQ == 1 && op == 3) || (SQ == 2 && op == 2)); };
    auto is_t = [](int op) { return (op & 1) == 0 ? TA : TB; };
    auto frags = [](int op) { return (op & 1) == 0 ? FM : FN; };
    const float* src[4] = {A, B, A2, B2};
    // Addressing: buffer loads, byte offset = the lane's fixed part (a VGPR per operand fragment, set up once) + the K
    // step's part (an SGPR): no vector arithmetic per load. K-contiguous operand: lane part (row * ld + 4 q) * 4, step part
    // 64 g; K-major: lane part (4 q * ld + row) * 4, step part (16 g + t) * ld * 4. (The launcher keeps operands below 2 GiB.)
    __amdgpu_buffer_rsrc_t rs[NOP];
    int voff[NOP][FX], safe[NOP][FX];                             // the lane's fixed part; a readable offset of its row (first chunk / k row 0)
    bool ones[FM];
#pragma unroll
    for (int f = 0; f < FM; ++f) ones[f] = false;
#pragma unroll
    for (int op = 0; op < NOP; ++op) {
        if (!loaded(op)) continue;
        rs[op] = __builtin_amdgcn_make_buffer_rsrc((void*)src[op], 0, 0x7fffffff, 0x00020000);
        const bool isA = (op & 1) == 0;
        const int ld = (int)(isA ? lda : ldb);
        const int rows = isA ? M : N;
#pragma unroll
        for (int f = 0; f < FX; ++f) {
            if (f >= frags(op)) continue;
            const int r = (isA ? m0 : n0) + f * 16 + li;
            if (is_t(op)) {                                       // a row past the pitch is outside the matrix: the epilogue masks it
                safe[op][f] = (r < ld ? r : 0) * 4;
                voff[op][f] = safe[op][f] + 4 * kq * ld * 4;
            } else {
                safe[op][f] = min(r, rows - 1) * ld * 4;
                voff[op][f] = safe[op][f] + 4 * kq * 4;
            }
            if (TA && op == 0 && r == ones_row) ones[f] = true;
        }
    }
    // K-contiguous side: chunks of a row are read up to min(16 ng, ld) -- the operand's own zero padding when the leading
    // dimension holds it, the row's end when it is a raw matrix (gemm_v1.h's rule with the walk padded to 16, not 32).
    // Groups below `gfull` are whole on both sides and take the unmasked path; the last group may be partial: the TAIL
    // step of the wave that owns it, loaded with per-lane guards and zero-filled.
    const int klim_a = (int)min((int64_t)ng * 16, lda), klim_b = (int)min((int64_t)ng * 16, ldb);
    const int gfull = min(TA ? (K >> 4) : (klim_a >> 4), TB ? (K >> 4) : (klim_b >> 4));
    const int nsf = max((gfull - wave + V0_W - 1) / V0_W, 0);     // this wave's whole steps; ns - nsf is 0 or 1
    f32x4 st[V0_PD][NOP][FX], tl[NOP][FX];
    auto load_step = [&](auto d_c, int s) {
        constexpr int d = decltype(d_c)::value;
        const int g = wave + V0_W * s;
#pragma unroll
        for (int op = 0; op < NOP; ++op) {
            if (!loaded(op)) continue;
            const int ld = (int)(((op & 1) == 0) ? lda : ldb);
#pragma unroll
            for (int f = 0; f < FX; ++f) {
                if (f >= frags(op)) continue;
                if (is_t(op)) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        st[d][op][f][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs[op], voff[op][f], (g * 16 + t) * ld * 4, 0));
                } else {
                    st[d][op][f] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs[op], voff[op][f], g * 64, 0));
                }
            }
        }
    };
    auto load_tail = [&]() {
        const int k0 = (ng - 1) * 16 + 4 * kq;
#pragma unroll
        for (int op = 0; op < NOP; ++op) {
            if (!loaded(op)) continue;
            const bool isA = (op & 1) == 0;
            const int ld = (int)(isA ? lda : ldb);
#pragma unroll
            for (int f = 0; f < FX; ++f) {
                if (f >= frags(op)) continue;
                if (is_t(op)) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        tl[op][f][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rs[op], (k0 + t < K) ? voff[op][f] + ((ng - 1) * 16 + t) * ld * 4 : safe[op][f], 0, 0));
                } else {
                    tl[op][f] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        rs[op], (k0 + 4 <= (isA ? klim_a : klim_b)) ? voff[op][f] + (ng - 1) * 64 : safe[op][f], 0, 0));
                }
            }
        }
    };
    auto fixed_tail = [&](int op, f32x4 v) -> f32x4 {             // zero what the tail's guards replaced
        const bool isA = (op & 1) == 0;
        const int k0 = (ng - 1) * 16 + 4 * kq;
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = (is_t(op) ? (k0 + t < K) : (k0 + 4 <= (isA ? klim_a : klim_b))) ? v[t] : 0.f;
        return v;
    };
    f32x4 acc1[FM][FN], acc2[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) { acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // one K group's fragments (x[op][f]) into the accumulators; `tail`: the guarded group
    auto use = [&](const f32x4 (&x)[NOP][FX], bool tail) {
        f32x4 a[FM], b[FN], a2[FM], b2[FN];
        const int k0 = (ng - 1) * 16 + 4 * kq;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            a[i] = tail ? fixed_tail(0, x[0][i]) : x[0][i];
            if constexpr (DUAL && SQ != 2) a2[i] = tail ? fixed_tail(2, x[2][i]) : x[2][i];
            if constexpr (TA) {                                   // the synthetic row of ones (its square is one as well)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float one = (!tail || k0 + t < K) ? 1.f : 0.f;
                    a[i][t] = ones[i] ? one : a[i][t];
                    if constexpr (DUAL && SQ != 2) a2[i][t] = ones[i] ? one : a2[i][t];
                }
            }
            if constexpr (DUAL && SQ == 2) a2[i] = a[i] * a[i];
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            b[j] = tail ? fixed_tail(1, x[1][j]) : x[1][j];
            if constexpr (DUAL) b2[j] = SQ == 1 ? b[j] * b[j] : (tail ? fixed_tail(3, x[SQ == 1 ? 1 : 3][j]) : x[SQ == 1 ? 1 : 3][j]);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                acc1[i][j] = v0_mfma16(a[i], b[j], acc1[i][j]);
                if constexpr (DUAL) acc2[i][j] = v0_mfma16(a2[i], b2[j], acc2[i][j]);
            }
    };

    vbnn_static_for<0, V0_PD>([&](auto D) __attribute__((always_inline)) {
        if ((int)decltype(D)::value < nsf) load_step(D, (int)decltype(D)::value);     // wave-uniform
    });
    if (ns > nsf) load_tail();
    // the epilogue's own operands, fetched before the walk by the wave that will run it: wave f owns fragment f
    const int fm = wave / FN, fn = wave % FN;
    const int um = __builtin_amdgcn_readfirstlane(m0 + fm * 16), un = __builtin_amdgcn_readfirstlane(n0 + fn * 16);
    bool fast = false;
    typename Epi::Lane eln = {};
    typename Epi::Pre epre = {};
    typename v0_pre_noise<Epi>::type enoise = {};
    if (wave < NF) {
        fast = epi.fast_ok() && !epi.t1_ptr() && !epi.t2_ptr() && um + 16 <= epi.m_dim() && un + 16 <= epi.n_dim();   // wave-uniform
        if (fast) {
            eln = epi.lane_init(li, kq * 4);
            epre = epi.load_fast(um, un, eln);
            if constexpr (v0_pre_noise<Epi>::value) enoise = epi.draw_fast(um, un, eln);     // ~300 VALU instructions under the first loads' latency
        }
    }
    for (int s0 = 0; s0 < nsf; s0 += V0_PD) {
        vbnn_static_for<0, V0_PD>([&](auto D) __attribute__((always_inline)) {
            const int s = s0 + decltype(D)::value;
            if (s < nsf) {                                        // wave-uniform
                use(st[decltype(D)::value], false);
                if (s + V0_PD < nsf) load_step(D, s + V0_PD);
            }
        });
    }
    if (ns > nsf) use(tl, true);

    // red[wave][fragment][accumulator][lane]
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            red[((wave * NF + i * FN + j) * 2 + 0) * 64 + lane] = acc1[i][j];
            if constexpr (DUAL) red[((wave * NF + i * FN + j) * 2 + 1) * 64 + lane] = acc2[i][j];
        }
    __syncthreads();
    if (wave >= NF) return;
    f32x4 s1 = red[((0 * NF + wave) * 2 + 0) * 64 + lane], s2 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (DUAL) s2 = red[((0 * NF + wave) * 2 + 1) * 64 + lane];
#pragma unroll
    for (int w = 1; w < V0_W; ++w) {
        s1 += red[((w * NF + wave) * 2 + 0) * 64 + lane];
        if constexpr (DUAL) s2 += red[((w * NF + wave) * 2 + 1) * 64 + lane];
    }
    if (fast) {
        float t1[4], t2[4];
        if constexpr (v0_pre_noise<Epi>::value) epi.apply_fast_z(um, un, eln, s1, s2, epre, enoise, t1, t2);
        else epi.apply_fast(um, un, eln, s1, s2, epre, t1, t2);
        return;
    }
    epi(um + kq * 4, un + li, s1, s2);
}

template <int FM, int FN, bool DUAL, class Epi, bool TA, bool TB, int SQ>
__global__ __launch_bounds__(64 * V0_W) void gemm_nt_v0(const float* __restrict__ A, const float* __restrict__ A2, int64_t lda,
                                                         const float* __restrict__ B, const float* __restrict__ B2, int64_t ldb,
                                                         int M, int N, int K, int ones_row, int gx, Epi epi) {
    __shared__ f32x4 red[v0_red_slots<FM, FN>()];
    // (an XCD-aware tile order -- the blocks that share an L2 on consecutive rows of the larger operand -- measured the
    // same as this one: the operands of these sizes sit in every L2 after the first touch)
    gemm_v0_tile<FM, FN, DUAL, Epi, TA, TB, SQ>(A, A2, lda, B, B2, ldb, M, N, K, ones_row, epi, (int)blockIdx.x % gx, (int)blockIdx.x / gx, red);
}

// ---- two INDEPENDENT GEMMs in one launch: updateGradInput and accGradParameters of a layer both consume g and neither
// reads what the other writes, but as two launches the second waits for the first. Workgroups [0, a.blocks) compute
// problem A's tiles, the rest problem B's; each tile is computed exactly as its own launch would (bitwise the same).
template <class Epi>
struct V0Problem {
    const float* A; const float* A2; int64_t lda; const float* B; const float* B2; int64_t ldb;
    int M, N, K, ones_row, gx, blocks;
    Epi epi;
};
template <int FMA, int FNA, bool DUAL_A, class EpiA, bool TA_A, bool TB_A, int SQ_A, int FMB, int FNB, bool DUAL_B, class EpiB, bool TA_B,
          bool TB_B, int SQ_B>
__global__ __launch_bounds__(64 * V0_W) void gemm_nt_v0_pair(V0Problem<EpiA> a, V0Problem<EpiB> b) {
    constexpr int LA = v0_red_slots<FMA, FNA>(), LB = v0_red_slots<FMB, FNB>();
    __shared__ f32x4 red[LA > LB ? LA : LB];
    const int bid = (int)blockIdx.x;
    if (bid < a.blocks)
        gemm_v0_tile<FMA, FNA, DUAL_A, EpiA, TA_A, TB_A, SQ_A>(a.A, a.A2, a.lda, a.B, a.B2, a.ldb, a.M, a.N, a.K, a.ones_row, a.epi, bid % a.gx,
                                                               bid / a.gx, red);
    else
        gemm_v0_tile<FMB, FNB, DUAL_B, EpiB, TA_B, TB_B, SQ_B>(b.A, b.A2, b.lda, b.B, b.B2, b.ldb, b.M, b.N, b.K, b.ones_row, b.epi,
                                                               (bid - a.blocks) % b.gx, (bid - a.blocks) / b.gx, red);
}

// the kernel addresses an operand with 32-bit byte offsets, reads K-contiguous rows in 16-byte chunks and K-major rows of
// pitch ld; VBNN_ERR_UNSUPPORTED (nothing launched, no error text): the caller falls back to gemm_v1
template <bool TA, bool TB>
static inline bool v0_operands_ok(const float* A, const float* A2, int64_t lda, const float* B, const float* B2, int64_t ldb, int M, int N,
                                  int K, int ones_row) {
    const int64_t lim = (1ll << 31) - 64;
    if ((((uintptr_t)A | (uintptr_t)A2 | (uintptr_t)B | (uintptr_t)B2) & 15u) != 0 || lda % 4 != 0 || ldb % 4 != 0) return false;
    if ((TA ? (int64_t)K : (int64_t)M) * lda * 4 >= lim || (TB ? (int64_t)K : (int64_t)N) * ldb * 4 >= lim) return false;
    if (TA ? lda < M - (ones_row >= 0 ? 1 : 0) : lda < K) return false;
    if (TB ? ldb < N : ldb < K) return false;
    return K >= 1 && M >= 1 && N >= 1;
}
// tile of a launch: two fragments along N (the side whose rows are loaded once per array) when that still leaves a tile
// for most CUs, one fragment otherwise
static inline bool v0_wide_tile(int M, int N) { return (long)((M + 15) / 16) * ((N + 31) / 32) >= 160; }

template <bool DUAL, class Epi, bool TA, bool TB, int SQ>
static int launch_gemm_v0(hipStream_t stream, const float* A, const float* A2, int64_t lda, const float* B, const float* B2, int64_t ldb,
                          int M, int N, int K, const Epi& epi, int ones_row) {
    if (!v0_operands_ok<TA, TB>(A, A2, lda, B, B2, ldb, M, N, K, ones_row)) return VBNN_ERR_UNSUPPORTED;
    // accGradParameters form (both sides K-major: every fragment is four 4-byte loads per group): the single fragment;
    // forward / gradInput forms: 16 x 32 when that fills the chip (lab, 784-400-400-10 at batch 256: 6.6 vs 7.5 us, 4.9 vs 5.2)
    if (!(TA && TB) && v0_wide_tile(M, N)) {
        const int gx = (M + 15) / 16, gy = (N + 31) / 32;
        hipLaunchKernelGGL((gemm_nt_v0<1, 2, DUAL, Epi, TA, TB, SQ>), dim3(gx * gy), dim3(64 * V0_W), 0, stream, A, A2, lda, B, B2, ldb, M, N, K,
                           ones_row, gx, epi);
    } else {
        const int gx = (M + 15) / 16, gy = (N + 15) / 16;
        hipLaunchKernelGGL((gemm_nt_v0<1, 1, DUAL, Epi, TA, TB, SQ>), dim3(gx * gy), dim3(64 * V0_W), 0, stream, A, A2, lda, B, B2, ldb, M, N, K,
                           ones_row, gx, epi);
    }
    return vbnn_check_launch("gemm_nt_v0");
}
