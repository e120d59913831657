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
//
// r03, fp32 only (the launch-bound configurations, BASELINE configs[1]): operand FORMS beyond the packed one, so that the
// fp32 step needs no packing launch and no transposed copy of anything --
//   TA / TB   the operand is stored K-MAJOR, element (row, k) at X[k * ld + row] (the untransposed activation, gradient or
//             weight matrix): a thread's 16-byte chunk is then four ROWS of one k, written to the same LDS image with four
//             ds_write_b32. K rows past the true K are zero-filled (a K-major operand has no zero padding along K).
//   SQ        the pair's second operand on one side IS the square of the first (x.x beside x): it is formed in registers
//             while staging (bit for bit what the packer stored) and never loaded. SQ = 1: B2 = B.B; SQ = 2: A2 = A.A.
//   K mask    a K-contiguous operand whose leading dimension is shorter than the padded K walk (the raw minibatch, ld = I) is
//             zero-filled past its row end instead of read.
//   ones row  (TA) A row `ones_row` is all ones without being stored anywhere: its output row is the column sum of B --
//             the bias gradient from the parameter-gradient GEMM (vbnn_dw_args.gradBias).
// The K order of every accumulation chain is unchanged: results are bitwise those of the packed / transposed operands.
#pragma once
#include "common.h"
#include "gemm_v0.h"

// 1 (default): the fp32 shapes of the 32 x 32 geometry run on the latency kernel (gemm_v0.h); 0: on this file's tile.
// Test / A-B hook: vbnn_debug_set(VBNN_DEBUG_V0, ..).
static int g_v0 = 1;

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

// LDS bytes of one workgroup's tile(s)
template <typename T, bool DUAL, int WR, int KS>
constexpr int v1_lds_bytes() { return (WR == 1 ? 2 : 1) * (DUAL ? 4 : 2) * (32 * WR) * (64 * KS + 16); }

// one block tile (bx, by) of the GEMM; `lds` = v1_lds_bytes() of the workgroup's shared memory. A device function so that
// one launch can carry the tiles of TWO independent GEMMs (gemm_nt_v1_pair below).
template <typename T, bool DUAL, int WR, int KS, class Epi, bool TA = false, bool TB = false, int SQ = 0>
__device__ __forceinline__ void gemm_v1_tile(const T* __restrict__ A, const T* __restrict__ A2, int64_t lda,
                                             const T* __restrict__ B, const T* __restrict__ B2, int64_t ldb,
                                             int M, int N, int Kp, int K, int ones_row, Epi& epi, int bx, int by,
                                             unsigned char* __restrict__ lds) {
    constexpr int BT = 32 * WR;                       // block tile rows (M and N)
    constexpr int RB = 64 * KS;                       // bytes of K per row per step
    constexpr int PITCH = RB + 16;                    // + 16 B pad: conflict-free ds_read_b128 down a column
    constexpr int KE = RB / (int)sizeof(T);           // K elements per step
    constexpr int CE = 16 / (int)sizeof(T);           // elements per 16-byte chunk
    constexpr int CPR = RB / 16;                      // chunks per row
    constexpr int CHUNKS = BT * CPR;                  // 16-byte chunks per operand tile
    constexpr int CPT = CHUNKS / 256;                 // chunks per thread per operand tile
    constexpr int NOP = DUAL ? 4 : 2;
    constexpr int RC = BT / CE;                       // (K-major) 16-byte row chunks per k row
    static_assert(CHUNKS % 256 == 0, "tile does not divide over 256 threads");
    static_assert(!(TA || TB || SQ) || sizeof(T) == 4, "K-major / squared operand forms: fp32 only");
    static_assert(SQ == 0 || DUAL, "SQ derives the pair's second operand");
    static_assert(RC * KE == CHUNKS, "a K-major tile has as many chunks as a K-contiguous one");
    typedef typename Frag<T>::type frag_t;

    constexpr bool DB = (WR == 1);                    // double-buffered LDS: see the K walk
    static_assert(v1_lds_bytes<T, DUAL, WR, KS>() == (DB ? 2 : 1) * NOP * BT * PITCH, "LDS size helper out of step");

    epi.bind_draw();                                  // device-resident draw counter, if the host gave one (vbnn_fwd_args.draw_dev)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = bx * BT, n0 = by * BT;

    // PD: register prefetch depth, in K steps. The 32 x 32 tile of latency-bound sizes does so little per step that each
    // step is one exposed global-load latency (the 784-deep forward of the small MLP: 25 steps, 37 us); four steps of
    // loads in flight cost 16 registers each and hide most of it. Larger tiles keep one step ahead.
    constexpr int PD = (WR == 1) ? 4 : 1;
    const T* src[4] = {A, B, A2, B2};
    // (a native vector type, not HIP's uint4 struct: an array of the latter is not promoted to registers -- it lived in scratch)
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    // operand `op` (0 A, 1 B, 2 A2, 3 B2) is loaded at all -- not derived from its partner by SQ; and its storage form
    auto loaded = [](int op) { return !((SQ == 1 && op == 3) || (SQ == 2 && op == 2)); };
    auto is_t = [](int op) { return (op & 1) == 0 ? TA : TB; };
    u32x4_t stage[PD][NOP][CPT];
    const T* gptr[NOP][CPT];                          // a readable 16-byte chunk of the thread's row(s): the row start / k row 0
    int goff[NOP][CPT];                               // element offset from it to the thread's chunk at K step 0
    int lds_off[2][CPT];                              // byte offset of the chunk in an operand tile [K-contiguous | K-major]
    int kofs[2][CPT];                                 // element offset along K of the chunk inside a step [kc * CE | k row]
    int one_e[CPT];                                   // (TA) which element of the chunk is the ones row, or -1
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int chunk = tid + c * 256;
        lds_off[0][c] = (chunk / CPR) * PITCH + (chunk % CPR) * 16;               // row chunk / CPR, 16-byte chunk chunk % CPR
        kofs[0][c] = (chunk % CPR) * CE;
        lds_off[1][c] = ((chunk % RC) * CE) * PITCH + (chunk / RC) * (int)sizeof(T);   // rows 4 rc .. 4 rc + 3, k row chunk / RC
        kofs[1][c] = chunk / RC;
        one_e[c] = -1;
#pragma unroll
        for (int op = 0; op < NOP; ++op) {
            const bool isA = (op & 1) == 0;
            const int64_t ld = isA ? lda : ldb;
            const int rows = isA ? M : N, t0 = isA ? m0 : n0;
            if (is_t(op)) {
                // four consecutive ROWS of one k. The pitch is a multiple of 4 and so is r0: a chunk either lies inside
                // the pitch (stored rows, then the operand's zero padding) or wholly past it -- then every row of it is
                // outside the matrix, the epilogue masks what it feeds, and any readable finite data will do: chunk 0
                const int r0 = t0 + (chunk % RC) * CE;
                gptr[op][c] = src[op] + ((int64_t)r0 + CE <= ld ? r0 : 0);
                goff[op][c] = (chunk / RC) * (int)ld;
                if (isA && ones_row >= r0 && ones_row < r0 + CE) one_e[c] = ones_row - r0;
            } else {
                const int grow = min(t0 + chunk / CPR, rows - 1);
                gptr[op][c] = src[op] + (int64_t)grow * ld;
                goff[op][c] = (chunk % CPR) * CE;
            }
        }
    }
    // K limit of each side's loads. K-major: the true K (k rows past it do not exist). K-contiguous: the operand's own
    // zero padding is read up to Kp when the leading dimension holds it (the packed operands: exactly the loads of
    // before); a shorter row (a raw matrix such as the minibatch itself, ld = K) is zero-filled past its end.
    const int klim_a = TA ? K : (int)min((int64_t)Kp, lda), klim_b = TB ? K : (int)min((int64_t)Kp, ldb);
    const int64_t kstr_a = TA ? lda : 1, kstr_b = TB ? ldb : 1;
    // A chunk's LOAD is unconditional and its result is not touched until it is written to LDS PD steps later (the loads of
    // PD steps fly together: a load inside `if (valid)`, or a select right behind it, makes hipcc wait for each on the spot).
    // A chunk past the K limit reads the row's first chunk instead; `fixed` zeroes it -- and plants the ones -- at the LDS write.
    auto chunk_valid = [&](int op, int c, int kt) -> bool {
        const bool t = is_t(op);
        const int k = kt * KE + kofs[t ? 1 : 0][c];
        const int klim = (op & 1) == 0 ? klim_a : klim_b;
        return t ? (k < klim) : (k + CE <= klim);
    };
    auto load_chunk = [&](int op, int c, int kt) -> u32x4_t {
        const bool isA = (op & 1) == 0;
        const int64_t off = chunk_valid(op, c, kt) ? goff[op][c] + (int64_t)(kt * KE) * (isA ? kstr_a : kstr_b) : 0;
        return *reinterpret_cast<const u32x4_t*>(gptr[op][c] + off);
    };
    auto fixed = [&](int op, int c, int kt, u32x4_t v) -> u32x4_t {
        const bool valid = chunk_valid(op, c, kt);
        v = valid ? v : u32x4_t{0u, 0u, 0u, 0u};
        if (TA && (op & 1) == 0 && one_e[c] >= 0) {               // the synthetic row of ones (its square is one as well)
            const unsigned one = valid ? 0x3f800000u : 0u;
            v[0] = one_e[c] == 0 ? one : v[0]; v[1] = one_e[c] == 1 ? one : v[1];
            v[2] = one_e[c] == 2 ? one : v[2]; v[3] = one_e[c] == 3 ? one : v[3];
        }
        return v;
    };
    // one chunk into its operand tile: K-contiguous = one ds_write_b128; K-major = four ds_write_b32 (rows 4 rc + e, k row kr)
    auto put = [&](int op, int c, int buf, u32x4_t v) {
        unsigned char* base = lds + buf * (NOP * BT * PITCH) + op * BT * PITCH;
        if (is_t(op)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) *reinterpret_cast<unsigned*>(base + lds_off[1][c] + e * PITCH) = v[e];
        } else {
            *reinterpret_cast<u32x4_t*>(base + lds_off[0][c]) = v;
        }
    };
    auto squared = [](u32x4_t v) -> u32x4_t {
        u32x4_t r;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float f = __uint_as_float(v[e]); r[e] = __float_as_uint(f * f); }
        return r;
    };

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
                if (loaded(op)) stage[d][op][c] = load_chunk(op, c, min(d, nk - 1));

    // The epilogue's own operands (bias; x and r of the layer below; ...) are fetched NOW, before the K walk, through the
    // functor's FAST protocol when it applies (packed outputs, no transposed copies) and the wave's 16 x 16 tile lies inside
    // the matrix: at the launch-bound sizes the K walk is a handful of steps and an epilogue that starts its loads after it
    // adds a full memory round trip to every launch. Same arithmetic as the guarded form (the pipelined kernels' protocol).
    const int um = __builtin_amdgcn_readfirstlane(m0 + wm * WR * 16), un = __builtin_amdgcn_readfirstlane(n0 + wn * WR * 16);
    bool fast = false;
    typename Epi::Lane eln = {};
    typename Epi::Pre epre = {};
    if constexpr (WR == 1) {
        fast = epi.fast_ok() && !epi.t1_ptr() && !epi.t2_ptr() && um + 16 <= epi.m_dim() && un + 16 <= epi.n_dim();   // wave-uniform
        if (fast) {
            eln = epi.lane_init(lane & 15, (lane >> 4) * 4);
            epre = epi.load_fast(um, un, eln);
        }
    }
    const int a_row = (wm * WR * 16 + (lane & 15)) * PITCH + (lane >> 4) * 16;
    const int b_row = (wn * WR * 16 + (lane & 15)) * PITCH + (lane >> 4) * 16;

    auto put_step = [&](auto d_c, int kt, int buf) {          // the staged chunks of K step kt (register set d) into LDS buffer `buf`
        constexpr int d = decltype(d_c)::value;
#pragma unroll
        for (int op = 0; op < NOP; ++op)
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                if (loaded(op)) put(op, c, buf, fixed(op, c, kt, stage[d][op][c]));
                else put(op, c, buf, squared(fixed(op - 2, c, kt, stage[d][op - 2][c])));     // SQ: the pair's second operand is the first one squared
            }
    };
    auto load_step = [&](auto d_c, int kt) {
        constexpr int d = decltype(d_c)::value;
#pragma unroll
        for (int op = 0; op < NOP; ++op)
#pragma unroll
            for (int c = 0; c < CPT; ++c)
                if (loaded(op)) stage[d][op][c] = load_chunk(op, c, kt);
    };
    auto compute = [&](int buf) {
        const unsigned char* l = lds + buf * (NOP * BT * PITCH);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {             // 64 bytes of K per row per sub-step
            frag_t af[WR], bf[WR];
#pragma unroll
            for (int i = 0; i < WR; ++i) {
                af[i] = *reinterpret_cast<const frag_t*>(l + 0 * BT * PITCH + a_row + i * 16 * PITCH + ks * 64);
                bf[i] = *reinterpret_cast<const frag_t*>(l + 1 * BT * PITCH + b_row + i * 16 * PITCH + ks * 64);
            }
#pragma unroll
            for (int i = 0; i < WR; ++i)
#pragma unroll
                for (int j = 0; j < WR; ++j) acc1[i][j] = mfma_step<T>(af[i], bf[j], acc1[i][j]);
            if (DUAL) {
#pragma unroll
                for (int i = 0; i < WR; ++i) {
                    af[i] = *reinterpret_cast<const frag_t*>(l + 2 * BT * PITCH + a_row + i * 16 * PITCH + ks * 64);
                    bf[i] = *reinterpret_cast<const frag_t*>(l + 3 * BT * PITCH + b_row + i * 16 * PITCH + ks * 64);
                }
#pragma unroll
                for (int i = 0; i < WR; ++i)
#pragma unroll
                    for (int j = 0; j < WR; ++j) acc2[i][j] = mfma_step<T>(af[i], bf[j], acc2[i][j]);
            }
        }
    };
    // (the K walk is unrolled by PD so that every register set has a compile-time index; same k order as ever)
    if constexpr (DB) {
        // Double-buffered LDS (the 32 x 32 tile of the launch-bound sizes: one workgroup per CU, so nothing else hides a
        // step's LDS write + barrier): step kt + 1 is written into the other buffer while step kt is computed, ONE barrier
        // per step. The arithmetic and its order are those of the single-buffered loop.
        static_assert(PD % 2 == 0, "buffer parity = register set parity");
        put_step(std::integral_constant<int, 0>(), 0, 0);
        if (PD < nk) load_step(std::integral_constant<int, 0>(), PD);
        __syncthreads();
        for (int kt0 = 0; kt0 < nk; kt0 += PD) {
            vbnn_static_for<0, PD>([&](auto D) __attribute__((always_inline)) {
                constexpr int d = decltype(D)::value, dn = (d + 1) % PD;
                const int kt = kt0 + d;
                if (kt < nk) {                        // block-uniform
                    if (kt + 1 < nk) {
                        put_step(std::integral_constant<int, dn>(), kt + 1, (d + 1) & 1);
                        if (kt + 1 + PD < nk) load_step(std::integral_constant<int, dn>(), kt + 1 + PD);
                    }
                    compute(d & 1);
                    __syncthreads();                  // buffer d & 1 is free; buffer (d + 1) & 1 is complete
                }
            });
        }
    } else {
        for (int kt0 = 0; kt0 < nk; kt0 += PD) {
            vbnn_static_for<0, PD>([&](auto D) __attribute__((always_inline)) {
                constexpr int d = decltype(D)::value;
                const int kt = kt0 + d;
                if (kt < nk) {                        // block-uniform (the unrolled body keeps constant register-set indices)
                    put_step(D, kt, 0);
                    __syncthreads();
                    if (kt + PD < nk) load_step(D, kt + PD);
                    compute(0);
                    __syncthreads();
                }
            });
        }
    }

    if constexpr (WR == 1) {
        if (fast) {
            float t1[4], t2[4];
            epi.apply_fast(um, un, eln, acc1[0][0], acc2[0][0], epre, t1, t2);
            return;
        }
    }
    const int em = m0 + wm * WR * 16 + (lane >> 4) * 4;
    const int en = n0 + wn * WR * 16 + (lane & 15);
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int j = 0; j < WR; ++j) epi(em + i * 16, en + j * 16, acc1[i][j], acc2[i][j]);
}

template <typename T, bool DUAL, int WR, int KS, class Epi, bool TA = false, bool TB = false, int SQ = 0>
__global__ __launch_bounds__(256) void gemm_nt_v1(const T* __restrict__ A, const T* __restrict__ A2, int64_t lda,
                                                  const T* __restrict__ B, const T* __restrict__ B2, int64_t ldb,
                                                  int M, int N, int Kp, int K, int ones_row, Epi epi) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[v1_lds_bytes<T, DUAL, WR, KS>()];
    gemm_v1_tile<T, DUAL, WR, KS, Epi, TA, TB, SQ>(A, A2, lda, B, B2, ldb, M, N, Kp, K, ones_row, epi, (int)blockIdx.x, (int)blockIdx.y, lds);
}

// ---- two INDEPENDENT GEMMs in one launch (fp32, the 32 x 32 tile of the launch-bound sizes): updateGradInput and
// accGradParameters of a layer both consume g and neither reads what the other writes, but as two launches the second
// waits for the first -- 10 us each at 784-400-400-10 / batch 256, of which the chip is busy a fraction. Workgroups
// [0, a.blocks) compute problem A's tiles, the rest problem B's; each tile is computed exactly as its own launch would
// (same code, same K order: bitwise the two-launch results).
template <typename T, class Epi>
struct V1Problem {
    const T* A; const T* A2; int64_t lda; const T* B; const T* B2; int64_t ldb;
    int M, N, Kp, K, ones_row, gx, blocks;
    Epi epi;
};
template <typename T, bool DUAL_A, class EpiA, bool TA_A, bool TB_A, int SQ_A, bool DUAL_B, class EpiB, bool TA_B, bool TB_B, int SQ_B>
__global__ __launch_bounds__(256) void gemm_nt_v1_pair(V1Problem<T, EpiA> a, V1Problem<T, EpiB> b) {
    constexpr int LA = v1_lds_bytes<T, DUAL_A, 1, 2>(), LB = v1_lds_bytes<T, DUAL_B, 1, 2>();
    __shared__ __attribute__((aligned(16))) unsigned char lds[LA > LB ? LA : LB];
    const int bid = (int)blockIdx.x;
    if (bid < a.blocks)
        gemm_v1_tile<T, DUAL_A, 1, 2, EpiA, TA_A, TB_A, SQ_A>(a.A, a.A2, a.lda, a.B, a.B2, a.ldb, a.M, a.N, a.Kp, a.K, a.ones_row, a.epi,
                                                               bid % a.gx, bid / a.gx, lds);
    else
        gemm_v1_tile<T, DUAL_B, 1, 2, EpiB, TA_B, TB_B, SQ_B>(b.A, b.A2, b.lda, b.B, b.B2, b.ldb, b.M, b.N, b.Kp, b.K, b.ones_row, b.epi,
                                                               (bid - a.blocks) % b.gx, (bid - a.blocks) / b.gx, lds);
}

// form of a launch's operands beyond the packed default (fp32 only): see the head of this file
struct V1Form {
    bool ta = false, tb = false;      // A / B side stored K-major
    int sq = 0;                       // 1: B2 = B.B, 2: A2 = A.A (never loaded)
    int ones_row = -1;                // (ta) A row that is all ones
};

template <typename T, bool DUAL, class Epi, bool TA, bool TB, int SQ>
static int launch_gemm_v1_form(hipStream_t stream, const T* A, const T* A2, int64_t lda, const T* B, const T* B2,
                               int64_t ldb, int M, int N, int K, const Epi& epi, int ones_row) {
    // a K-contiguous operand holds the padded K walk (packed: zero fill) or is exactly a raw matrix of row length >= K,
    // whose columns [K, ld) -- if any -- must be finite (they meet the other side's zero padding)
    auto need = [&](int Kp) -> bool {
        const bool a_ok = TA || lda >= Kp || lda >= K, b_ok = TB || ldb >= Kp || ldb >= K;
        if (!a_ok || !b_ok) vbnn_set_error("leading dimension too small: lda=%lld ldb=%lld K=%d", (long long)lda, (long long)ldb, K);
        return a_ok && b_ok;
    };
    if ((TA && (lda % 4 != 0 || lda < M - (ones_row >= 0 ? 1 : 0))) || (TB && (ldb % 4 != 0 || ldb < N))) {
        vbnn_set_error("K-major fp32 operands need a row pitch that is a multiple of 4 and holds the rows: lda=%lld ldb=%lld",
                       (long long)lda, (long long)ldb);
        return VBNN_ERR_INVALID;
    }
    if ((((uintptr_t)A | (uintptr_t)B | (uintptr_t)A2 | (uintptr_t)B2) & 15u) != 0 || (!TA && lda % (16 / (int)sizeof(T)) != 0) ||
        (!TB && ldb % (16 / (int)sizeof(T)) != 0)) {
        vbnn_set_error("gemm_v1 operands must be 16-byte aligned with 16-byte row pitches");
        return VBNN_ERR_INVALID;
    }
    const long blocks128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (blocks128 >= 128) {
        const int KE = 64 / (int)sizeof(T);
        const int Kp = (K + KE - 1) / KE * KE;
        if (!need(Kp)) return VBNN_ERR_INVALID;
        dim3 grid((M + 127) / 128, (N + 127) / 128);
        hipLaunchKernelGGL((gemm_nt_v1<T, DUAL, 4, 1, Epi, TA, TB, SQ>), grid, dim3(256), 0, stream, A, A2, lda, B, B2, ldb, M, N, Kp, K, ones_row, epi);
    } else {
        constexpr int KS = 2;                                     // 128 B of K per row per step: 32 f32 / 64 bf16
        constexpr int KE2 = 128 / (int)sizeof(T);                 // (4 x 64 x 144 B of LDS for the dual tile)
        const int Kp = (K + KE2 - 1) / KE2 * KE2;
        if (!need(Kp)) return VBNN_ERR_INVALID;
        const long blocks64 = (long)((M + 63) / 64) * ((N + 63) / 64);
        if (blocks64 >= 96) {
            dim3 grid((M + 63) / 64, (N + 63) / 64);
            hipLaunchKernelGGL((gemm_nt_v1<T, DUAL, 2, KS, Epi, TA, TB, SQ>), grid, dim3(256), 0, stream, A, A2, lda, B, B2, ldb, M, N, Kp, K, ones_row, epi);
        } else {        // latency-bound sizes (the 256 x 400 outputs of the small MLP): the latency kernel, or 32 x 32 tiles, 4x the blocks
            if constexpr (sizeof(T) == 4) {
                if (g_v0) {
                    const int st = launch_gemm_v0<DUAL, Epi, TA, TB, SQ>(stream, A, A2, lda, B, B2, ldb, M, N, K, epi, ones_row);
                    if (st != VBNN_ERR_UNSUPPORTED) return st;
                }
            }
            dim3 grid((M + 31) / 32, (N + 31) / 32);
            hipLaunchKernelGGL((gemm_nt_v1<T, DUAL, 1, KS, Epi, TA, TB, SQ>), grid, dim3(256), 0, stream, A, A2, lda, B, B2, ldb, M, N, Kp, K, ones_row, epi);
        }
    }
    return vbnn_check_launch("gemm_nt_v1");
}

template <typename T, bool DUAL, class Epi>
static int launch_gemm_v1(hipStream_t stream, const T* A, const T* A2, int64_t lda, const T* B, const T* B2,
                          int64_t ldb, int M, int N, int K, const Epi& epi, const V1Form& f = V1Form()) {
    if constexpr (sizeof(T) == 4) {
        // the forms the fp32 step uses: forward (B2 = B.B), gradInput (A K-major), accGradParameters (both K-major, A2 = A.A)
        if constexpr (DUAL) {
            if (!f.ta && !f.tb && f.sq == 1) return launch_gemm_v1_form<T, true, Epi, false, false, 1>(stream, A, A2, lda, B, B2, ldb, M, N, K, epi, -1);
            if (f.ta && f.tb && f.sq == 2) return launch_gemm_v1_form<T, true, Epi, true, true, 2>(stream, A, A2, lda, B, B2, ldb, M, N, K, epi, f.ones_row);
        }
        if (f.ta && !f.tb && f.sq == 0) return launch_gemm_v1_form<T, DUAL, Epi, true, false, 0>(stream, A, A2, lda, B, B2, ldb, M, N, K, epi, -1);
        if (f.ta && f.tb && f.sq == 0) return launch_gemm_v1_form<T, DUAL, Epi, true, true, 0>(stream, A, A2, lda, B, B2, ldb, M, N, K, epi, f.ones_row);
    }
    if (f.ta || f.tb || f.sq) {
        vbnn_set_error("operand form not instantiated (K-major %d / %d, squares %d): fp32 only", (int)f.ta, (int)f.tb, f.sq);
        return VBNN_ERR_UNSUPPORTED;
    }
    return launch_gemm_v1_form<T, DUAL, Epi, false, false, 0>(stream, A, A2, lda, B, B2, ldb, M, N, K, epi, -1);
}

// would launch_gemm_v1 give this shape the 32 x 32 tile (the launch-bound geometry the pair launch exists for)?
static inline bool v1_small_geometry(int M, int N) {
    return (long)((M + 127) / 128) * ((N + 127) / 128) < 128 && (long)((M + 63) / 64) * ((N + 63) / 64) < 96;
}
// problem A: gradInput form (A K-major: TA), DUAL or not; problem B: accGradParameters form (both K-major, A2 = A.A when
// DUAL: SQ = 2). fp32 only. VBNN_ERR_UNSUPPORTED (nothing launched) when a shape wants another geometry.
template <bool DUAL, class EpiA, class EpiB>
static int launch_gemm_v1_pair(hipStream_t stream, const float* A, const float* A2, int64_t lda, const float* B, const float* B2, int64_t ldb,
                               int M, int N, int K, const EpiA& epi_a,
                               const float* xA, int64_t ldx, const float* gB, const float* gvB, int64_t ldg, int M2, int N2, int K2,
                               int ones_row, const EpiB& epi_b) {
    if (!v1_small_geometry(M, N) || !v1_small_geometry(M2, N2)) return VBNN_ERR_UNSUPPORTED;
    if ((((uintptr_t)A | (uintptr_t)A2 | (uintptr_t)B | (uintptr_t)B2 | (uintptr_t)xA | (uintptr_t)gB | (uintptr_t)gvB) & 15u) != 0 ||
        lda % 4 || ldb % 4 || ldx % 4 || ldg % 4 || lda < M || ldx < M2 - (ones_row >= 0 ? 1 : 0) || ldg < N2)
        return VBNN_ERR_UNSUPPORTED;
    if (g_v0 && v0_operands_ok<true, false>(A, A2, lda, B, B2, ldb, M, N, K, -1) &&
        v0_operands_ok<true, true>(xA, nullptr, ldx, gB, gvB, ldg, M2, N2, K2, ones_row)) {
        // the latency kernel's tiles, each problem with the tile its own launch would take (launch_gemm_v0)
        const bool wide = v0_wide_tile(M, N);
        const int gxa = (M + 15) / 16, gya = wide ? (N + 31) / 32 : (N + 15) / 16, gxb = (M2 + 15) / 16, gyb = (N2 + 15) / 16;
        V0Problem<EpiA> qa{A, A2, lda, B, B2, ldb, M, N, K, -1, gxa, gxa * gya, epi_a};
        V0Problem<EpiB> qb{xA, nullptr, ldx, gB, gvB, ldg, M2, N2, K2, ones_row, gxb, gxb * gyb, epi_b};
        if (wide)
            hipLaunchKernelGGL((gemm_nt_v0_pair<1, 2, DUAL, EpiA, true, false, 0, 1, 1, DUAL, EpiB, true, true, (DUAL ? 2 : 0)>),
                               dim3(qa.blocks + qb.blocks), dim3(64 * V0_W), 0, stream, qa, qb);
        else
            hipLaunchKernelGGL((gemm_nt_v0_pair<1, 1, DUAL, EpiA, true, false, 0, 1, 1, DUAL, EpiB, true, true, (DUAL ? 2 : 0)>),
                               dim3(qa.blocks + qb.blocks), dim3(64 * V0_W), 0, stream, qa, qb);
        return vbnn_check_launch("gemm_nt_v0_pair");
    }
    constexpr int KE2 = 32;
    V1Problem<float, EpiA> pa{A, A2, lda, B, B2, ldb, M, N, (K + KE2 - 1) / KE2 * KE2, K, -1, (M + 31) / 32, ((M + 31) / 32) * ((N + 31) / 32), epi_a};
    V1Problem<float, EpiB> pb{xA, nullptr, ldx, gB, gvB, ldg, M2, N2, (K2 + KE2 - 1) / KE2 * KE2, K2, ones_row, (M2 + 31) / 32,
                              ((M2 + 31) / 32) * ((N2 + 31) / 32), epi_b};
    if (ldb < pa.Kp && ldb < K) return VBNN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((gemm_nt_v1_pair<float, DUAL, EpiA, true, false, 0, DUAL, EpiB, true, true, (DUAL ? 2 : 0)>), dim3(pa.blocks + pb.blocks),
                       dim3(256), 0, stream, pa, pb);
    return vbnn_check_launch("gemm_nt_v1_pair");
}
