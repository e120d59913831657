// gemm_v2.h -- the pipelined bf16 MFMA GEMM of the wide configurations (gfx950 only).
//
//   C1[m][n] = sum_k A[m][k] Bt[n][k]      and, when DUAL,      C2[m][n] = sum_k A2[m][k] Bt2[n][k]
//
// Structure (cdna_hip_programming.md section 5, "What does break it"): ~1 workgroup per CU, LDS
// filled by LDS-DMA (global_load_lds_dwordx4) that stays in flight ACROSS the barrier, counted
// s_waitcnt vmcnt(N), raw s_barrier, all LDS in one array.
//
//   workgroup   2 WM waves as WM (M) x 2 (N); wave tile 64 x 64; block tile (64 WM) x 128
//               WM = 4: 512 threads, 256 x 128 -- the default; WM = 2: 256 threads, 128 x 128 -- for outputs
//               too small to give every CU a 256 x 128 tile (e.g. the 784 x 4096 gradient of layer 1)
//   K step      64 bf16 = 128 B per row; stage = A tile (8 KiB per 64 rows) + B tile (16 KiB)
//   ring        3 stages (144 KiB at WM = 4): DMAs run two tiles ahead of the MFMAs
//   tile stream DUAL alternates the pairs: (A,B,k0) -> acc1, (A2,B2,k0) -> acc2, (A,B,k0+64) ...
//               so the register cost of the second GEMM is only its accumulator
//   MFMA        v_mfma_f32_16x16x32_bf16, 32 per wave per tile (4 x 4 output tiles x 2 k-halves)
//   LDS image   lane-linear as the DMA writes it (8 rows x 128 B per wave-instruction); the bank
//               swizzle chunk' = chunk ^ ((row >> 1) & 7) is applied on the per-lane SOURCE address
//               and again on the ds_read_b128 address (rule 21: both sides or neither). With it the
//               16 rows x 1 chunk a lane group reads fall on 16 distinct 16-byte bank slots
//               (SQ_LDS_BANK_CONFLICT = 0 measured).
//   per tile    (schedules 0 / 2; schedule 4: step_pp below)
//               s_waitcnt vmcnt(G) lgkmcnt(0) -> this wave's G DMAs of tile u have landed (tile u+1's may fly) and its
//                                     LDS reads of tile u-1 have RETURNED (v2_wait_barrier: one asm with the barrier)
//               s_barrier           -> everybody's have; everybody is done reading tile u-1
//               16 ds_read_b128 + 32 MFMA on tile u, with the G DMAs of tile u+2 (into the buffer tile
//               u-1 occupied) issued one after every four MFMAs (SCHED 2). Bursting them right after the
//               barrier instead (SCHED 0) serialises the CU's 64 B/clk texture path in front of the matrix
//               pipe: s_memtime stamps put 32 % of a wave's time in that burst; interleaving is +20 %.
//   epilogue    accumulators re-laid through LDS into rows of n, so a wave-instruction stores 4 rows x 64
//               contiguous m; transposed outputs take a second LDS trip (see the end of the kernel).
//
// Any M, N >= 1 (row-clamped sources, masked epilogue); K is padded by the packed-operand
// convention (ld % 64 == 0, zero fill).
#pragma once
#include "common.h"
#include <type_traits>

constexpr int V2_BN = 128, V2_BK = 64;
constexpr int V2_B_BYTES = V2_BN * V2_BK * 2;           // 16384
constexpr int v2_a_bytes(int WM) { return 64 * WM * V2_BK * 2; }
constexpr int v2_stage(int WM) { return v2_a_bytes(WM) + V2_B_BYTES; }
// LDS per workgroup: the ring, or the epilogue's 18 KiB-per-wave staging if that is larger (2 stages, 4 waves)
constexpr int v2_lds(int WM, int ST) { return (v2_stage(WM) * ST > 2 * WM * 18432 ? v2_stage(WM) * ST : 2 * WM * 18432) + 16; }   // 147456 / 98304 / 73728 (+ 16 spare)

// Schedule (vbnn_debug_set key 1): 0 = the tile's DMAs in a burst after the barrier; 2 = interleaved with its MFMAs; 4 (r04) = the tile
// step as ALTERNATING clusters, the workgroup's halves one cluster apart (step_pp: gemm_v3.h's K step on this ring -- at 4096^3 forward
// 217.7 -> 200.6 us, gradInput 227.6 -> 216.0, accGradParameters 223.9 -> 214.8, bitwise equal); -1 = auto: 4 for the 8-wave 256 x 128
// tile, 0 for the 4-wave 128 x 128 tile (one wave per SIMD has no partner: no clusters to alternate, and a burst is 8 % faster than
// interleaving there).
inline int g_v2_sched = -1;
inline int g_v2_tile = 0;         // 0: pick 256 x 128 or 128 x 128 by shape; 128 / 256: force (vbnn_debug_set key 2)
inline int g_v2_psplit = -1;      // pair split of the 256 x 128 tiling: -1 by shape, 0 never, 1 whenever the functor allows (key 5)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N> __device__ __forceinline__ void v2_wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// THE phase boundary of the pipelined loops (gemm_v2.h, gemm_v3.h): ONE asm statement that
//   (1) waits until all but this wave's N youngest LDS-DMA pieces have landed        -- RAW, with the barrier
//   (2) RETIRES every LDS read this wave has issued (lgkmcnt(0))                     -- WAR, with the barrier
//   (3) s_barrier
// (2) is what makes it legal to refill a stage ONE phase after its last read (cdna_hip_programming.md, "WAR: restage a
// buffer >= 2 phases after its last ds_read, or 1 phase after when an lgkmcnt before the reading phase's first barrier
// retired those reads"). It used to be left to the compiler: a fragment's ds_read is waited for in front of the MFMA
// that consumes it, and the MFMAs of a step precede the next barrier in the source. But an MFMA is not a memory
// operation: in the DUAL SCHED-0 instantiations (no sched_barrier pins; the loop body is two steps) hipcc 7.2 sank the
// last eight MFMAs of a step -- and with them the wait for the last two A-fragment reads -- BELOW the next step's
// s_barrier: `ds_read_b128 x2; s_waitcnt vmcnt(8); s_barrier; buffer_load ... lds; ...; s_waitcnt lgkmcnt(1); v_mfma`.
// Those reads were in flight across the barrier while the other waves, released by it, issued the refill of the very
// stage they read (tile u + 3 -> stage u % 3). Harmless while an LDS read returns long before a DMA piece can land; with a
// second process's kernels contending for the CU's LDS and texture path it is a race, and it is the r02 "late piece":
// one 8-row A piece of one K step wrong, 1 run in 20 of the two-rank rehearsal, whose 512-row shard takes exactly these
// instantiations (128 x 128 tiles, SCHED 0). Found from the ISA (tools/check_barrier_lgkm.py lists the LDS reads
// outstanding at every s_barrier of every kernel; it runs as a CPU test). The barrier is part of the statement so that
// no LDS read of the NEXT phase can be scheduled between the wait and the barrier either (s_barrier is IntrNoMem to
// LLVM: a plain load may cross it; an asm with a memory clobber it may not).
template <int N> __device__ __forceinline__ void v2_wait_barrier() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
// a barrier with no LDS-DMA condition: retire this wave's LDS reads, then meet
__device__ __forceinline__ void v2_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// KM (pair-split launches only): both operands K-MAJOR -- element (row, k) at X[k * ld + row], i.e. x and g as the
// forward / gradInput GEMMs hold them (see gemm_v3.h: tile image [64 k][rows], chunk swizzle 2 h(k), fragments by
// ds_read_b64_tr_b16 through inline asm). Columns past the matrix are clamped to its last whole chunk: the rows of C they
// feed are masked by the epilogue. K must be whole 64-row steps.
template <bool DUAL, int SCHED, int WM, int ST, class Epi, bool KM = false>
__global__ __launch_bounds__(128 * WM, 2) void gemm_nt_v2(const bf16_t* __restrict__ A, const bf16_t* __restrict__ A2, int64_t lda,
                                                          const bf16_t* __restrict__ B, const bf16_t* __restrict__ B2, int64_t ldb,
                                                          int M, int N, int nk, int tiles_m, int tiles_n, int psplit, Epi epi_in) {
    constexpr int NW = 2 * WM;                 // waves per workgroup
    constexpr int BM = 64 * WM;
    constexpr int A_BYTES = v2_a_bytes(WM), STAGE = v2_stage(WM);
    constexpr int AG = (BM / 8) / NW;          // 8-row DMA groups of the A tile per wave (4)
    constexpr int BG = (V2_BN / 8) / NW;       // ... of the B tile (2 at WM = 4, 4 at WM = 2)
    constexpr int G = AG + BG;                 // DMAs per wave per tile
    static_assert(AG == 4 && (G == 6 || G == 8), "unsupported geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: keep it in an SGPR (LDS-DMA destinations)
    const int wm = wave >> 1, wn = wave & 1;

    // ---- block -> tile mapping. Blocks that share an XCD (bid % 8, T1) get a contiguous chunk of
    // the tile list; inside it tiles are walked in 4 (M) x 8 (N) groups so the ~32 blocks running
    // together on an XCD share 4 A panels and 8 B panels in its L2 (measured hit rate 81 % = the ideal
    // 1 - 12/64 of that grouping).
    const int nblk = tiles_m * tiles_n * ((!DUAL && psplit) ? 2 : 1);
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;      // bijective remap
    }
    // Pair split (psplit = 1, non-DUAL instantiation, splittable functors only): the grid is tiles x 2 and a tile's two
    // blocks compute ONE GEMM of the pair each -- (A, B) or (A2, B2) -- with the single-accumulator main loop; the
    // functor is told which one it holds and writes only the outputs that depend on it (epilogues.h, EpiDw::part).
    // For outputs too few to give every CU a dual tile this doubles the blocks with no exchange of partial sums.
    Epi epi = epi_in;
    const int part = (!DUAL && psplit) ? (bid & 1) : 0;
    if (!DUAL && psplit) {
        bid >>= 1;
        epi.set_part(1 + part);
        if (part) { A = A2; B = B2; }
    }
    int tm, tn;
    {
        constexpr int GM = 4, GN = 8;
        const int per_band = GM * tiles_n;                // tiles in a band of GM tile-rows (all bands but the last are full)
        const int band = bid / per_band;
        const int in_band = bid - band * per_band;
        const int band_rows = min(GM, tiles_m - band * GM);
        const int full = band_rows * GN;                  // tiles in a full group of GN tile-columns
        const int grp = in_band / full;                   // only the last group of a band can be narrower
        const int in_grp = in_band - grp * full;
        const int grp_cols = min(GN, tiles_n - grp * GN);
        tm = min(band * GM + in_grp / grp_cols, tiles_m - 1);
        tn = grp * GN + in_grp % grp_cols;
    }
    const int m0 = tm * BM, n0 = tn * V2_BN;

    // ---- LDS-DMA source pointers. Lane l of the instruction for 8-row group g fills LDS position
    // (row 8g + (l>>3), chunk slot l&7) with global chunk (l&7) ^ f(row), f(row) = (row >> 1) & 7.
    const bf16_t* a_src[2][AG];
    const bf16_t* b_src[2][BG];
#if defined(__HIP_DEVICE_COMPILE__)       // the LDS-DMA pieces go out in buffer form (see gemm_v3.h): SGPR resource + 32-bit lane byte offset + SGPR K offset
    int dma_avo[AG], dma_bvo[BG];
    const __amdgpu_buffer_rsrc_t dma_ra[2] = {__builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000),
                                              __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? A2 : A), 0, 0x7fffffff, 0x00020000)};
    const __amdgpu_buffer_rsrc_t dma_rb[2] = {__builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7fffffff, 0x00020000),
                                              __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? B2 : B), 0, 0x7fffffff, 0x00020000)};
#endif
    static_assert(!KM || (WM == 4 && !DUAL), "the K-major form exists for the 256 x 128 single-accumulator tile only");
    auto hk = [](int kr) { return (kr & 3) | (((kr >> 3) & 1) << 2); };
#pragma unroll
    for (int i = 0; i < AG; ++i) {
        int64_t off;
        if (KM) {                              // instruction g = wave + 8 i: k-rows 2 g, 2 g + 1 of the [64 k][256 m] tile
            const int kr = 2 * (wave + NW * i) + (lane >> 5);
            const int cs = (lane & 31) ^ (hk(kr) << 1);
            off = (int64_t)kr * lda + min(m0 + cs * 8, (int)lda - 8);
        } else {
            const int row = 8 * (wave + NW * i) + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            off = (int64_t)min(m0 + row, M - 1) * lda + chunk * 8;
        }
        a_src[0][i] = A + off;
        a_src[1][i] = DUAL ? A2 + off : A + off;
#if defined(__HIP_DEVICE_COMPILE__)
        dma_avo[i] = (int)(off * 2);
#endif
    }
#pragma unroll
    for (int i = 0; i < BG; ++i) {
        int64_t off;
        if (KM) {                              // instruction g = wave + 8 i: k-rows 4 g .. 4 g + 3 of the [64 k][128 n] tile
            const int kr = 4 * (wave + NW * i) + (lane >> 4);
            const int cs = (lane & 15) ^ (hk(kr) << 1);
            off = (int64_t)kr * ldb + min(n0 + cs * 8, (int)ldb - 8);
        } else {
            const int row = 8 * (wave + NW * i) + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            off = (int64_t)min(n0 + row, N - 1) * ldb + chunk * 8;
        }
        b_src[0][i] = B + off;
        b_src[1][i] = DUAL ? B2 + off : B + off;
#if defined(__HIP_DEVICE_COMPILE__)
        dma_bvo[i] = (int)(off * 2);
#endif
    }
    const int U = DUAL ? 2 * nk : nk;       // tiles in the stream

    // `pair` / `idx` are compile-time constants at every call site (the loop below is unrolled by the pair
    // period), so the pointer arrays stay in registers (runtime-indexed arrays would go to scratch).
    auto issue_one = [&](int u, auto pair_c, auto idx_c) {      // DMA number IDX (0..AG-1: A groups, then B groups) of tile u
        constexpr int P = decltype(pair_c)::value;
        constexpr int IDX = decltype(idx_c)::value;
        const int64_t kstep = (int64_t)(DUAL ? (u >> 1) : u) * V2_BK;
        const int64_t koff = KM ? kstep * (IDX < AG ? lda : ldb) : kstep;
        unsigned char* base = lds + (u % ST) * STAGE;
#if defined(__HIP_DEVICE_COMPILE__)
        const int dma_so = __builtin_amdgcn_readfirstlane((int)(koff * 2));
        if constexpr (IDX < AG)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dma_ra[P], (lptr_t)(base + (wave + NW * IDX) * 1024), 16, dma_avo[IDX], dma_so, 0, 0);
        else if constexpr (IDX < G)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dma_rb[P], (lptr_t)(base + A_BYTES + (wave + NW * (IDX - AG)) * 1024), 16,
                                                     dma_bvo[IDX - AG], dma_so, 0, 0);
#else
        if constexpr (IDX < AG)
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[P][IDX] + koff), (lptr_t)(base + (wave + NW * IDX) * 1024), 16, 0, 0);
        else if constexpr (IDX < G)
            __builtin_amdgcn_global_load_lds((gptr_t)(b_src[P][IDX - AG] + koff),
                                             (lptr_t)(base + A_BYTES + (wave + NW * (IDX - AG)) * 1024), 16, 0, 0);
#endif
    };
    auto issue = [&](int u, auto pair_c) {
        issue_one(u, pair_c, std::integral_constant<int, 0>()); issue_one(u, pair_c, std::integral_constant<int, 1>());
        issue_one(u, pair_c, std::integral_constant<int, 2>()); issue_one(u, pair_c, std::integral_constant<int, 3>());
        issue_one(u, pair_c, std::integral_constant<int, 4>()); issue_one(u, pair_c, std::integral_constant<int, 5>());
        issue_one(u, pair_c, std::integral_constant<int, 6>()); issue_one(u, pair_c, std::integral_constant<int, 7>());
    };

    // The main loop's form of issue_one: the stage's LDS offset and the operands' K byte offsets are LOOP-CARRIED scalars
    // (advanced once per step below) instead of being recomputed per piece from the tile index (u % ST, a multiply, two
    // vector adds and a v_readfirstlane per piece).
    unsigned la_stage = 0;          // LDS offset of the stage of tile u + LA
    int la_ka = 0, la_kb = 0;       // byte offsets along K of tile u + LA in the A / B operand
    auto issue_one_at = [&](auto pair_c, auto idx_c) {
        constexpr int P = decltype(pair_c)::value;
        constexpr int IDX = decltype(idx_c)::value;
        (void)P; (void)IDX; (void)la_ka; (void)la_kb;             // (only the device pass uses them)
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (IDX < AG)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dma_ra[P], (lptr_t)(lds + la_stage + IDX * (NW * 1024) + wave * 1024), 16,
                                                     dma_avo[IDX], la_ka, 0, 0);
        else if constexpr (IDX < G)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dma_rb[P], (lptr_t)(lds + la_stage + (A_BYTES + (IDX - AG) * (NW * 1024)) + wave * 1024),
                                                     16, dma_bvo[IDX - AG], la_kb, 0, 0);
#endif
    };

    // ---- fragment read offsets (bytes inside a stage). Lane reads row r = base + (l & 15), chunk
    // c = 4s + (l >> 4) of k-half s; swizzled chunk = c ^ ((r >> 1) & 7). Row bases are multiples of 16,
    // so (r >> 1) & 7 == ((l & 15) >> 1) for every fragment of the lane.
    const int rsw = (lane & 15) >> 1;
    const int q = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int csw = ((4 * s + q) ^ rsw) * 16;
        a_off[s] = (wm * 64 + (lane & 15)) * 128 + csw;
        b_off[s] = A_BYTES + (wn * 64 + (lane & 15)) * 128 + csw;
    }

    // K-major fragments (KM): see gemm_v3.h. Lane j = lane & 15 of its 16-lane group addresses k-row (j >> 2), columns 4 (j & 3) ..
    typedef __attribute__((address_space(3))) unsigned char* ldsb_t;
    const int trq = (lane & 15) >> 2, trp = lane & 3;
    const int thx = (trq | ((q & 1) << 2)) << 1;
    int a_tr[4], b_tr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_tr[i] = (8 * q + trq) * 512 + (((wm * 8 + 2 * i + (trp >> 1)) ^ thx) * 16) + (trp & 1) * 8;
        b_tr[i] = A_BYTES + (8 * q + trq) * 256 + (((wn * 8 + 2 * i + (trp >> 1)) ^ thx) * 16) + (trp & 1) * 8;
    }
    auto tr_load4 = [&](unsigned a0, unsigned a1, unsigned a2, unsigned a3, auto lo_c, auto hi_c, bf16x8 (&f)[4]) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        bf16x4 l0, h0, l1, h1, l2, h2, l3, h3;
        asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%12\n\tds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
                     "ds_read_b64_tr_b16 %2, %9 offset:%12\n\tds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
                     "ds_read_b64_tr_b16 %4, %10 offset:%12\n\tds_read_b64_tr_b16 %5, %10 offset:%13\n\t"
                     "ds_read_b64_tr_b16 %6, %11 offset:%12\n\tds_read_b64_tr_b16 %7, %11 offset:%13\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2), "=&v"(l3), "=&v"(h3)
                     : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(LO), "n"(HI)
                     : "memory");
        f[0] = __builtin_shufflevector(l0, h0, 0, 1, 2, 3, 4, 5, 6, 7);
        f[1] = __builtin_shufflevector(l1, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        f[2] = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7);
        f[3] = __builtin_shufflevector(l3, h3, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // the same with six fragments KEPT out of the destination registers (inputs + early-clobber outputs): the sources of the
    // wave's last eight MFMAs -- see tr_issue2_keep in gemm_v3.h for the hazard hipcc cannot see through inline asm
    auto tr_load4_keep = [&](unsigned a0, unsigned a1, unsigned a2, unsigned a3, auto lo_c, auto hi_c, bf16x8 (&f)[4],
                             const bf16x8 (&ka)[4], const bf16x8 (&kb)[4]) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        bf16x4 l0, h0, l1, h1, l2, h2, l3, h3;
        asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%12\n\tds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
                     "ds_read_b64_tr_b16 %2, %9 offset:%12\n\tds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
                     "ds_read_b64_tr_b16 %4, %10 offset:%12\n\tds_read_b64_tr_b16 %5, %10 offset:%13\n\t"
                     "ds_read_b64_tr_b16 %6, %11 offset:%12\n\tds_read_b64_tr_b16 %7, %11 offset:%13\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2), "=&v"(l3), "=&v"(h3)
                     : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(LO), "n"(HI), "v"(ka[2]), "v"(ka[3]), "v"(kb[0]), "v"(kb[1]), "v"(kb[2]), "v"(kb[3])
                     : "memory");
        f[0] = __builtin_shufflevector(l0, h0, 0, 1, 2, 3, 4, 5, 6, 7);
        f[1] = __builtin_shufflevector(l1, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        f[2] = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7);
        f[3] = __builtin_shufflevector(l3, h3, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    f32x4 acc1[4][4], acc2[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    std::integral_constant<int, 0> c0;
    std::integral_constant<int, 1> c1;
    // SCHED 4 (r04; the 8-wave tiles only): the tile step as ALTERNATING clusters, the workgroup's halves one cluster apart -- gemm_v3.h's
    // kstep_pp on this kernel's ring: M(u) = tile u's 16 fragment reads with tile u + 2's G pieces between the read groups, closed by
    // s_waitcnt vmcnt(G) lgkmcnt(0) + s_barrier; C(u) = 32 MFMAs, closed by s_barrier. Waves 4-7 (the SIMD partners of waves 0-3) enter
    // the stream one barrier late and waves 0-3 leave it one barrier late. Same pieces in the same order, same MFMA order: same bits.
    constexpr bool PP = SCHED == 4 && WM == 4 && ST == 3;
    static_assert(SCHED != 4 || PP, "the alternating schedule exists for the 8-wave tile with the three-stage ring");

    // ---- prologue: ST - 1 tiles in flight
    issue(0, c0);
    if (ST == 3 && U > 1) { if (DUAL) issue(1, c1); else issue(1, c0); }

    // tile u + LA (LA = ST - 1 tiles of lookahead) is issued while tile u is computed; its operand pair is the pair
    // of tile u when LA = 2 and the other pair when LA = 1 (DUAL alternates pairs tile by tile)
    constexpr int LA = ST - 1;
    constexpr int LA_ = LA;
    la_stage = (unsigned)(LA % ST) * STAGE;
    {
        const int kidx = DUAL ? (LA >> 1) : LA;
        la_ka = 2 * kidx * V2_BK * (KM ? (int)lda : 1);
        la_kb = 2 * kidx * V2_BK * (KM ? (int)ldb : 1);
    }
    const int la_dka = 2 * V2_BK * (KM ? (int)lda : 1), la_dkb = 2 * V2_BK * (KM ? (int)ldb : 1);
    // ALWAYS: tile u + LA exists (every step but the last LA): its DMAs are unconditional and the waits constant
    // eight transpose reads (four fragments of one k-half), issued only; landed4 hands them to the compiler behind the cluster's wait
    auto tr_issue4 = [&](unsigned a0, unsigned a1, unsigned a2, unsigned a3, auto lo_c, auto hi_c, bf16x4 (&l)[4], bf16x4 (&h)[4]) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%12\n\tds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
                     "ds_read_b64_tr_b16 %2, %9 offset:%12\n\tds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
                     "ds_read_b64_tr_b16 %4, %10 offset:%12\n\tds_read_b64_tr_b16 %5, %10 offset:%13\n\t"
                     "ds_read_b64_tr_b16 %6, %11 offset:%12\n\tds_read_b64_tr_b16 %7, %11 offset:%13"
                     : "=&v"(l[0]), "=&v"(h[0]), "=&v"(l[1]), "=&v"(h[1]), "=&v"(l[2]), "=&v"(h[2]), "=&v"(l[3]), "=&v"(h[3])
                     : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(LO), "n"(HI)
                     : "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto landed4 = [&](bf16x4 (&l)[4], bf16x4 (&h)[4]) {
        asm volatile("" : "+v"(l[0]), "+v"(h[0]), "+v"(l[1]), "+v"(h[1]), "+v"(l[2]), "+v"(h[2]), "+v"(l[3]), "+v"(h[3])::"memory");
    };
    auto step_pp = [&](int u, auto pair_c, f32x4 (&acc)[4][4], auto always_c) {
        constexpr bool ALWAYS = decltype(always_c)::value;
        const bool more = ALWAYS ? true : (u + LA_ < U);
        const unsigned char* stage = lds + (u % ST) * STAGE;
        bf16x8 af[2][4], bf[2][4];
        bf16x4 al[2][4], ah[2][4], bl[2][4], bh[2][4];
        auto piece = [&](auto idx_c) { if (more) issue_one_at(pair_c, idx_c); __builtin_amdgcn_sched_barrier(0); };
        auto rd = [&](auto s_c) {                        // the A, then the B fragments of k-half S
            constexpr int S = decltype(s_c)::value;
            if constexpr (KM) {
                const unsigned sb = (unsigned)(uintptr_t)(ldsb_t)stage;
                tr_issue4(sb + a_tr[0], sb + a_tr[1], sb + a_tr[2], sb + a_tr[3], std::integral_constant<int, S * 16384>(),
                          std::integral_constant<int, S * 16384 + 2048>(), al[S], ah[S]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) af[S][i] = *reinterpret_cast<const bf16x8*>(stage + a_off[S] + i * 16 * 128);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto rdb = [&](auto s_c) {
            constexpr int S = decltype(s_c)::value;
            if constexpr (KM) {
                const unsigned sb = (unsigned)(uintptr_t)(ldsb_t)stage;
                tr_issue4(sb + b_tr[0], sb + b_tr[1], sb + b_tr[2], sb + b_tr[3], std::integral_constant<int, S * 8192>(),
                          std::integral_constant<int, S * 8192 + 1024>(), bl[S], bh[S]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[S][j] = *reinterpret_cast<const bf16x8*>(stage + b_off[S] + j * 16 * 128);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // ---------------- M cluster: a read group, a piece (two after the last two groups when the tile has eight pieces)
        rd(c0); piece(std::integral_constant<int, 0>()); piece(std::integral_constant<int, 1>());
        rdb(c0); piece(std::integral_constant<int, 2>()); piece(std::integral_constant<int, 3>());
        rd(c1); piece(std::integral_constant<int, 4>());
        rdb(c1); piece(std::integral_constant<int, 5>());
        if constexpr (G == 8) { piece(std::integral_constant<int, 6>()); piece(std::integral_constant<int, 7>()); }
        // tile u + 1's pieces have landed (tile u + 2's, issued just now, may fly); this cluster's reads are back
        if constexpr (ALWAYS) v2_wait_barrier<G>();
        else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (more) v2_wait_barrier<G>(); else v2_wait_barrier<0>(); }
        if constexpr (KM) {
            landed4(al[0], ah[0]); landed4(al[1], ah[1]); landed4(bl[0], bh[0]); landed4(bl[1], bh[1]);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[s2][i] = __builtin_shufflevector(al[s2][i], ah[s2][i], 0, 1, 2, 3, 4, 5, 6, 7);
                    bf[s2][i] = __builtin_shufflevector(bl[s2][i], bh[s2][i], 0, 1, 2, 3, 4, 5, 6, 7);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- C cluster
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s2][i], bf[s2][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        la_stage = la_stage == (unsigned)(ST - 1) * STAGE ? 0u : la_stage + STAGE;       // tile u + 1 + LA
        if constexpr (!DUAL || decltype(pair_c)::value == 1) { la_ka += la_dka; la_kb += la_dkb; }
    };
    auto step = [&](int u, auto pair_c, auto next_c, f32x4 (&acc)[4][4], auto always_c) {
        constexpr bool ALWAYS = decltype(always_c)::value;
        auto la_c = std::conditional_t<ST == 3, decltype(pair_c), decltype(next_c)>();
        if (ST == 3 && (ALWAYS || u + 1 < U)) v2_wait_barrier<G>();   // tile u + 1's DMAs may stay in flight
        else                                  v2_wait_barrier<0>();
        const bool more = ALWAYS ? true : (u + LA < U);
        if (SCHED == 0 && more) {
            issue_one_at(la_c, std::integral_constant<int, 0>()); issue_one_at(la_c, std::integral_constant<int, 1>());
            issue_one_at(la_c, std::integral_constant<int, 2>()); issue_one_at(la_c, std::integral_constant<int, 3>());
            issue_one_at(la_c, std::integral_constant<int, 4>()); issue_one_at(la_c, std::integral_constant<int, 5>());
            issue_one_at(la_c, std::integral_constant<int, 6>()); issue_one_at(la_c, std::integral_constant<int, 7>());
        }
        const unsigned char* stage = lds + (u % ST) * STAGE;
        bf16x8 paf[4], pbf[4];                           // (K-major) the first k-half's fragments: kept out of the second's reads
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
            bf16x8 af[4], bf[4];
            if constexpr (KM) {
                const unsigned sb = (unsigned)(uintptr_t)(ldsb_t)stage;
                if (sidx == 0) {
                    tr_load4(sb + a_tr[0], sb + a_tr[1], sb + a_tr[2], sb + a_tr[3], std::integral_constant<int, 0>(),
                             std::integral_constant<int, 2048>(), af);
                    tr_load4(sb + b_tr[0], sb + b_tr[1], sb + b_tr[2], sb + b_tr[3], std::integral_constant<int, 0>(),
                             std::integral_constant<int, 1024>(), bf);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { paf[i] = af[i]; pbf[i] = bf[i]; }
                } else {
                    tr_load4_keep(sb + a_tr[0], sb + a_tr[1], sb + a_tr[2], sb + a_tr[3], std::integral_constant<int, 16384>(),
                                  std::integral_constant<int, 16384 + 2048>(), af, paf, pbf);
                    tr_load4_keep(sb + b_tr[0], sb + b_tr[1], sb + b_tr[2], sb + b_tr[3], std::integral_constant<int, 8192>(),
                                  std::integral_constant<int, 8192 + 1024>(), bf, paf, pbf);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[i] = *reinterpret_cast<const bf16x8*>(stage + a_off[sidx] + i * 16 * 128);
                    bf[i] = *reinterpret_cast<const bf16x8*>(stage + b_off[sidx] + i * 16 * 128);
                }
            }
            auto slot = [&](int s_, int i_) {            // the DMA that belongs to MFMA group (k-half s_, row i_)
                if (s_ == 0 && i_ == 0) issue_one_at(la_c, std::integral_constant<int, 0>());
                if (s_ == 0 && i_ == 1) issue_one_at(la_c, std::integral_constant<int, 1>());
                if (s_ == 0 && i_ == 2) issue_one_at(la_c, std::integral_constant<int, 2>());
                if (s_ == 0 && i_ == 3) issue_one_at(la_c, std::integral_constant<int, 6>());
                if (s_ == 1 && i_ == 0) issue_one_at(la_c, std::integral_constant<int, 3>());
                if (s_ == 1 && i_ == 1) issue_one_at(la_c, std::integral_constant<int, 4>());
                if (s_ == 1 && i_ == 2) issue_one_at(la_c, std::integral_constant<int, 5>());
                if (s_ == 1 && i_ == 3) issue_one_at(la_c, std::integral_constant<int, 7>());
            };
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                if (SCHED == 2) {                        // one DMA behind every four MFMAs
                    if (more) slot(sidx, i);
                    __builtin_amdgcn_sched_barrier(0);   // keep each DMA where it was put
                }
            }
        }
        la_stage = la_stage == (unsigned)(ST - 1) * STAGE ? 0u : la_stage + STAGE;       // tile u + 1 + LA
        if constexpr (!DUAL || decltype(la_c)::value == 1) { la_ka += la_dka; la_kb += la_dkb; }
    };
    if constexpr (PP) {
        // every wave: tile 0 has landed, for everybody behind the barrier; then the late half's extra barrier, answered by the early half's at
        // the end of the stream
        if (U > 1) v2_wait_barrier<G>(); else v2_wait_barrier<0>();
        if (wave >= 4) asm volatile("s_barrier" ::: "memory");
        int u = 0;
        if (DUAL) {
            for (; u + 1 + LA < U; u += 2) { step_pp(u, c0, acc1, std::true_type()); step_pp(u + 1, c1, acc2, std::true_type()); }
            for (; u < U; u += 2) { step_pp(u, c0, acc1, std::false_type()); step_pp(u + 1, c1, acc2, std::false_type()); }
        } else {
            for (; u + LA < U; ++u) step_pp(u, c0, acc1, std::true_type());
            for (; u < U; ++u) step_pp(u, c0, acc1, std::false_type());
        }
        if (wave < 4) asm volatile("s_barrier" ::: "memory");
    } else if (DUAL) {
        int u = 0;
        for (; u + 1 + LA < U; u += 2) { step(u, c0, c1, acc1, std::true_type()); step(u + 1, c1, c0, acc2, std::true_type()); }
        for (; u < U; u += 2) { step(u, c0, c1, acc1, std::false_type()); step(u + 1, c1, c0, acc2, std::false_type()); }
    } else {
        int u = 0;
        for (; u + LA < U; ++u) step(u, c0, c0, acc1, std::true_type());
        for (; u < U; ++u) step(u, c0, c0, acc1, std::false_type());
    }

    // ---- epilogue. The ring is idle now: each wave re-lays its 64 x 64 accumulator tile through a private
    // 18 KiB LDS region from the MFMA layout (n = lane & 15, 4 m per lane-group) into rows of n, so that one
    // wave-instruction touches 4 rows x 64 CONTIGUOUS m (256 B of fp32 / 128 B of bf16 per row) instead of 16
    // rows x 16 m; the transposed outputs take a second trip (written [m][n], read back as rows of n).
    typedef typename Epi::elem_t ET;
    constexpr int SP = 68;                                    // fp32 staging pitch (floats): conflict-free b128 rows
    constexpr int TP = 72;                                    // transposed staging pitch (elements), 16-byte multiple
    static_assert(NW * 18432 <= v2_lds(WM, ST), "epilogue staging must fit the allocation");
    __syncthreads();                                          // every wave is done reading the last tile
    float* st = reinterpret_cast<float*>(lds + wave * 18432);
    const int c16 = lane & 15, q4 = lane >> 4;
    f32x4 r1[16], r2[16];
    auto relayout = [&](f32x4 (&acc)[4][4], f32x4 (&rows)[16]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(st + (j * 16 + c16) * SP + i * 16 + q4 * 4) = acc[i][j];
        // same-wave LDS operations complete in order: the reads below see the writes above
#pragma unroll
        for (int p = 0; p < 16; ++p) rows[p] = *reinterpret_cast<const f32x4*>(st + (4 * p + q4) * SP + 4 * c16);
    };
    relayout(acc1, r1);
    if (DUAL) relayout(acc2, r2);
    else {
#pragma unroll
        for (int p = 0; p < 16; ++p) r2[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int wm0 = m0 + wm * 64, wn0 = n0 + wn * 64;
    ET* tp1 = epi.t1_ptr();
    ET* tp2 = epi.t2_ptr();
    const bool any_t = (tp1 != nullptr) || (tp2 != nullptr);
    ET* tl1 = reinterpret_cast<ET*>(st);                      // [64 m][TP] each, reusing the staging region
    ET* tl2 = tl1 + 64 * TP;
    // 8-element chunks XOR-swizzled by (row >> 2) & 7: without it the 16 lanes of a quarter-wave hit two
    // banks (rows 4 apart are 576 B apart: 8-way conflict, 8.1 M conflict cycles per 4096^2 launch measured)
    auto stage_t = [&](int p, const float (&t1)[4], const float (&t2)[4]) {
        const int col = (((4 * p + q4) >> 3) ^ (c16 & 7)) * 8 + ((4 * p + q4) & 7);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tl1[(4 * c16 + j) * TP + col] = Elt<ET>::to(t1[j]);
            tl2[(4 * c16 + j) * TP + col] = Elt<ET>::to(t2[j]);
        }
    };
    // A wave tile that lies fully inside the matrix takes the functor's FAST protocol when it applies (epilogues.h):
    // the epilogue's global loads go out a batch at a time instead of one guarded, waited-for load per position.
    const bool fast_full = epi.fast_ok() && (wm0 + 64 <= epi.m_dim()) && (wn0 + 64 <= epi.n_dim());     // wave-uniform
    // ... and so does one that straddles the last output row when the functor allows it (EDGE_FAST): the lanes whose
    // quad lies past the edge are switched off, the lane on row m_dim hands its first value to edge_row
    const bool fast_edge = Epi::EDGE_FAST && !fast_full && epi.fast_ok() && !any_t && (wm0 < epi.m_dim()) &&
                           (epi.m_dim() % 4 == 0) && (wn0 + 64 <= epi.n_dim());
    if (fast_full || fast_edge) {
        const int um = __builtin_amdgcn_readfirstlane(wm0), un = __builtin_amdgcn_readfirstlane(wn0);
        const typename Epi::Lane eln = epi.lane_init(q4, 4 * c16);
        const bool lane_in = fast_full || (wm0 + 4 * c16 + 4 <= epi.m_dim());
        const bool lane_edge = fast_edge && (wm0 + 4 * c16 == epi.m_dim());
        constexpr int FB = Epi::FAST_BATCH_V2;
#pragma unroll
        for (int p0 = 0; p0 < 16; p0 += FB) {
            typename Epi::Pre pre[FB];
            if (lane_in) {
#pragma unroll
                for (int b = 0; b < FB; ++b) pre[b] = epi.load_fast(um, un + 4 * (p0 + b), eln);
#pragma unroll
                for (int b = 0; b < FB; ++b) {
                    float t1[4], t2[4];
                    epi.apply_fast(um, un + 4 * (p0 + b), eln, r1[p0 + b], r2[p0 + b], pre[b], t1, t2);
                    if (any_t) stage_t(p0 + b, t1, t2);
                }
            }
            if (lane_edge) {
#pragma unroll
                for (int b = 0; b < FB; ++b) epi.edge_row(wn0 + 4 * (p0 + b) + q4, r1[p0 + b][0]);
            }
        }
    } else {
        vbnn_static_for<0, 16>([&](auto P) __attribute__((always_inline)) {     // (not `#pragma unroll`: see vbnn_static_for)
            constexpr int p = decltype(P)::value;
            float t1[4], t2[4];
            epi.template apply<false>(wm0 + 4 * c16, wn0 + 4 * p + q4, r1[p], r2[p], t1, t2);
            if (any_t) stage_t(p, t1, t2);
        });
    }
    if (any_t) {
        // rows of the transposed outputs: lane (row = lane >> 3 (+8 per pass), 8 consecutive n = 8 (lane & 7) ..)
        const int64_t ldt = epi.t_ld();
        const int Md = epi.m_dim(), Nd = epi.n_dim();
        const int nn = wn0 + 8 * (lane & 7);
        const bool vec_ok = ((ldt & 7) == 0) && (nn + 8 <= Nd);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int ml = 8 * p + (lane >> 3);
            const int mm = wm0 + ml;
            if (mm < Md && nn < Nd) {
                if (tp1) {
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(tl1 + ml * TP + 8 * ((lane & 7) ^ ((ml >> 2) & 7)));
                    if (vec_ok) *reinterpret_cast<bf16x8*>(tp1 + (int64_t)mm * ldt + nn) = v;
                    else for (int e = 0; e < 8; ++e) if (nn + e < Nd) tp1[(int64_t)mm * ldt + nn + e] = v[e];
                }
                if (tp2) {
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(tl2 + ml * TP + 8 * ((lane & 7) ^ ((ml >> 2) & 7)));
                    if (vec_ok) *reinterpret_cast<bf16x8*>(tp2 + (int64_t)mm * ldt + nn) = v;
                    else for (int e = 0; e < 8; ++e) if (nn + e < Nd) tp2[(int64_t)mm * ldt + nn + e] = v[e];
                }
            }
        }
    }
}

template <typename T>
static inline bool gemm_v2_possible(int64_t lda, int64_t ldb) {
    return sizeof(T) == 2 && (lda % 64) == 0 && (ldb % 64) == 0;
}
// blocks of the 256 x 128 tiling / of the 128 x 128 tiling
static inline int64_t v2_tiles(int64_t M, int64_t N, int bm) { return ((M + bm - 1) / bm) * ((N + V2_BN - 1) / V2_BN); }
template <typename T>
static inline bool gemm_v2_eligible(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb) {
    (void)K;
    if (!gemm_v2_possible<T>(lda, ldb)) return false;
    // worth it only when a tiling gives the chip something like one block per CU
    return v2_tiles(M, N, 128) >= 96;
}

// would launch_gemm_v2 take the pair split for a DUAL launch of this shape? (shared with the K-major query)
static inline bool gemm_v2_psplit_by_shape(int64_t M, int64_t N, int64_t K) {
    const int nk = (int)((K + V2_BK - 1) / V2_BK);
    return g_v2_psplit == 1 || (g_v2_psplit == -1 && g_v2_tile == 0 && v2_tiles(M, N, 256) <= 128 && nk >= 16);
}

// kmajor: A / A2 and B / B2 are K-major (lda / ldb = pitch of a K row). Only the pair-split launch has that form:
// when the shape does not take it the call returns VBNN_ERR_UNSUPPORTED without launching and the caller falls back.
template <typename T, bool DUAL, class Epi>
static int launch_gemm_v2(vbnn_ctx* ctx, const T* A, const T* A2, int64_t lda, const T* B, const T* B2, int64_t ldb,
                          int M, int N, int K, const Epi& epi, bool kmajor = false) {
    hipStream_t stream = ctx->stream;
    if constexpr (sizeof(T) != 2) {
        vbnn_set_error("gemm_v2 is bf16 only");
        return VBNN_ERR_UNSUPPORTED;
    } else {
        if ((((uintptr_t)A | (uintptr_t)B | (uintptr_t)A2 | (uintptr_t)B2) & 15u) != 0) {
            vbnn_set_error("gemm_v2 operands must be 16-byte aligned");
            return VBNN_ERR_INVALID;
        }
        const int nk = (K + V2_BK - 1) / V2_BK;
        if (!kmajor && (lda < (int64_t)nk * V2_BK || ldb < (int64_t)nk * V2_BK)) {
            vbnn_set_error("packed leading dimension too small for K=%d", K);
            return VBNN_ERR_INVALID;
        }
        if (kmajor && !(K % V2_BK == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= 8 && ldb >= 8 && lda >= M && ldb >= N &&
                        (int64_t)K * lda < (1ll << 31) && (int64_t)K * ldb < (1ll << 31)))
            return VBNN_ERR_UNSUPPORTED;
        // 256 x 128 tiles unless they would leave more than a quarter of the 256 CUs without a block while the
        // 128 x 128 tiling fills more of them (the 784 x 4096 gradient: 128 blocks vs 224)
        const int64_t t256 = v2_tiles(M, N, 256), t128 = v2_tiles(M, N, 128);
        // Variant kept behind the debug key only: 128 x 128 tiles with a 2-stage ring use 72 KiB of LDS, so TWO
        // workgroups share a CU and one's epilogue runs beside the other's main loop. Meant for the K-short forward
        // of layer 1 (26 tile steps, then an epilogue as long as the main loop); measured 113 us against 110 us for
        // the 256 x 128 tile there (and 291 vs 257 us at K = 4096), so it is never picked automatically.
        const bool pairs = g_v2_tile == 64;
        // Pair split: see the kernel. By shape when the functor allows it, the dual 256 x 128 tiling gives at most half
        // the CUs a block (the 784 x 4096 gradient: 128 tiles) and K is long; g_v2_psplit: -1 by shape, 0 never, 1 always.
        bool psplit = false;
        if constexpr (DUAL && Epi::SPLITTABLE)
            psplit = !pairs && g_v2_tile != 128 && gemm_v2_psplit_by_shape(M, N, K);
        if (kmajor && !psplit) return VBNN_ERR_UNSUPPORTED;
        const bool small = !psplit && (pairs || g_v2_tile == 128 || (g_v2_tile == 0 && t256 < 192 && t128 > t256));
        // (4, the alternating clusters, needs two waves per SIMD: the four-wave tiles keep 0 / 2 whatever the key says)
        int sched = (g_v2_sched == 0 || g_v2_sched == 2 || g_v2_sched == 4) ? g_v2_sched : (small && !pairs ? 0 : 4);
        if ((small || pairs) && sched == 4) sched = 2;
        const int vi0 = pairs ? 2 : small ? 1 : 0;       // variant: 256 x 128 / 128 x 128 / 128 x 128 co-resident / pair split
        const int si = sched >> 1;                       // 0, 1, 2
        const void* kern;
        int threads, lds_bytes, bm;
        if (pairs) {
            kern = sched == 0 ? (const void*)gemm_nt_v2<DUAL, 0, 2, 2, Epi> : (const void*)gemm_nt_v2<DUAL, 2, 2, 2, Epi>;
            threads = 256; lds_bytes = v2_lds(2, 2); bm = 128;
        } else if (small) {                              // four waves, one workgroup per CU: no SIMD partner
            kern = sched == 0 ? (const void*)gemm_nt_v2<DUAL, 0, 2, 3, Epi> : (const void*)gemm_nt_v2<DUAL, 2, 2, 3, Epi>;
            threads = 256; lds_bytes = v2_lds(2, 3); bm = 128;
        } else if (psplit) {                             // the single-accumulator instantiations, two blocks per tile
            if (kmajor)
                kern = sched == 4 ? (const void*)gemm_nt_v2<false, 4, 4, 3, Epi, true> : (const void*)gemm_nt_v2<false, 2, 4, 3, Epi, true>;
            else
                kern = sched == 0 ? (const void*)gemm_nt_v2<false, 0, 4, 3, Epi>
                     : sched == 2 ? (const void*)gemm_nt_v2<false, 2, 4, 3, Epi> : (const void*)gemm_nt_v2<false, 4, 4, 3, Epi>;
            threads = 512; lds_bytes = v2_lds(4, 3); bm = 256;
        } else {
            kern = sched == 0 ? (const void*)gemm_nt_v2<DUAL, 0, 4, 3, Epi>
                 : sched == 2 ? (const void*)gemm_nt_v2<DUAL, 2, 4, 3, Epi> : (const void*)gemm_nt_v2<DUAL, 4, 4, 3, Epi>;
            threads = 512; lds_bytes = v2_lds(4, 3); bm = 256;
        }
        static vbnn_per_device_flag configured[5][3];      // per instantiation and device
        const int vi = psplit ? (kmajor ? 4 : 3) : vi0;
        if (!configured[vi][si][ctx->device]) {
            hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            if (e != hipSuccess) { vbnn_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return VBNN_ERR_HIP; }
            configured[vi][si][ctx->device] = true;
        }
        const int tiles_m = (M + bm - 1) / bm, tiles_n = (N + V2_BN - 1) / V2_BN;
        const bf16_t* a = (const bf16_t*)A; const bf16_t* a2 = (const bf16_t*)A2;
        const bf16_t* b = (const bf16_t*)B; const bf16_t* b2 = (const bf16_t*)B2;
        int nk_ = nk, M_ = M, N_ = N, tm_ = tiles_m, tn_ = tiles_n;
        int64_t lda_ = lda, ldb_ = ldb;
        Epi epi_ = epi;
        int psplit_ = psplit ? 1 : 0;
        void* args[] = {&a, &a2, &lda_, &b, &b2, &ldb_, &M_, &N_, &nk_, &tm_, &tn_, &psplit_, &epi_};
        hipError_t e = hipLaunchKernel(kern, dim3(tiles_m * tiles_n * (psplit ? 2 : 1)), dim3(threads), args, lds_bytes, stream);
        if (e != hipSuccess) { vbnn_set_error("launch of gemm_nt_v2 failed: %s", hipGetErrorString(e)); return VBNN_ERR_HIP; }
        return vbnn_check_launch("gemm_nt_v2");
    }
}
