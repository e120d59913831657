// gemm_v3.h -- the 256 x 256 two-pass variant of the pipelined bf16 GEMM (gfx950 only), for outputs with at least
// ~one such tile per CU (the 4096^2 layers of the wide configuration).
//
//   pass 1   C2[m][n] = sum_k A2[m][k] Bt2[n][k]      (the variance GEMM of the pair; DUAL only)
//   fold     acc := f(C2)  in registers              (FWD: b + sqrt(v) z; DX: 2 x . C2; DW: stores d/dlvars, 0)
//   pass 2   acc += sum_k A [m][k] Bt [n][k]         (the mean GEMM, accumulated ON TOP of the folded term)
//
// Why: gemm_v2's dual 256 x 128 tile needs 47.7 B/clk/CU of LDS-DMA at full MFMA rate (two accumulators per output
// cap its tile). One accumulator per output allows 256 x 256: 32 B/clk/CU, a third fewer DMAs and a quarter fewer
// ds_reads per MFMA -- what the vendor library's 256x256x64 macro-tile runs on. Every output of the three epilogues
// is LINEAR in the mean GEMM once the variance GEMM is known (y = m + [b + sqrt(v) z], gx = g mu + [2 x . gv s2],
// d/dmeans and d/dlvars depend on one GEMM each), so the pair runs as two passes of the SAME workgroup over the same
// accumulator registers and nothing is parked in between. (FWD's term needs a Philox block per quad; 32 of them beside
// 128 live accumulators spilled until the folds were serialised -- FOLD_SERIAL in epilogues.h -- so that one quad's
// draw is live at a time. r01's form, which parked the variance tile in a per-workgroup scratch tile and drew the noise in
// the final epilogue, cost +128 MB of cache traffic per launch, 10-15 us: tools/lab/ keeps it.)
//
//   workgroup   8 waves as 2 (M) x 4 (N); wave tile 128 x 64 = acc[8][4] (128 accumulator registers)
//   K step      64 bf16 (128-B rows, the swizzle of gemm_v2.h), two PHASES of 32 MFMAs per wave:
//               phase P uses the wave's A rows 64 P .. 64 P + 63 and all of its B rows (B fragments stay in registers)
//   LDS         160 KiB = A parts [2 K steps][2 phases] x 16 KiB + B tiles [3 K steps] x 32 KiB.
//               A part P holds the 2 x 64 rows the two wave rows use in phase P, so it is free for refill after that
//               phase: A is double-buffered at PHASE granularity, B triple-buffered at K-step granularity.
//   per phase   two clusters, the workgroup's halves (waves 0-3 | 4-7: the two waves of every SIMD) one cluster apart (kstep_pp):
//               M: 8 (16 in phase 0) fragment reads with the phase's 4 DMAs between them -> s_waitcnt vmcnt(8 | 10) lgkmcnt(0), s_barrier
//               C: 32 MFMAs -> s_barrier
//               DMAs:  phase 0 of step t: A part 1 of t+1 (x2), B of t+2 (x2);  phase 1 of step t: A part 0 of t+2 (x2), B of t+2 (x2)
//               every DMA is in flight for at least four clusters (two of them 32 MFMAs of its own wave) before its data is needed.
#pragma once
#include "gemm_v2.h"

constexpr int V3_BM = 256, V3_BN = 256;
constexpr int V3_APART = 128 * 128;                  // bytes: 128 rows x 128 B
constexpr int V3_BTILE = 256 * 128;
constexpr int V3_LDS = 4 * V3_APART + 3 * V3_BTILE;  // 163840: all of the CU's LDS

inline int g_v3_min_k = 704;        // shortest K the shape selection gives to this kernel (vbnn_debug_set key 4)

// Stamp hook: a build that wants in-kernel time stamps defines V3_ST(k) before including this file (tools/lab/pst.h: the K step's
// clusters; tools/lab/ also has r03's); the library does not.
#ifndef V3_ST
#define V3_ST(k) do { } while (0)
#endif


// AK / BK: the operand is stored K-MAJOR -- element (row, k) at X[k * ld + row], i.e. the untransposed activation /
// gradient / weight matrix -- instead of K-contiguous. Its tile then lies in LDS as [64 k][rows] and the MFMA fragments
// (8 consecutive k of one row per lane) are read with ds_read_b64_tr_b16, the hardware 4 x 16 transpose read, two per
// fragment. This is what lets accGradParameters consume x and g as the forward / gradInput GEMMs already hold them
// (no transposed copies written by any epilogue) and gradInput consume mu, sigma^2 as stored (no transposed shadows).
//   LDS image   A part: 64 k-rows x 256 B (the 64 m of wave row 0, then of wave row 1); B tile: 64 k-rows x 512 B
//   swizzle     16-byte chunk index ^= 2 h(k), h(k) = (k & 3) | ((k >> 3) & 1) << 2, on the DMA source and on the read: the
//               8 k-rows x 32 B a half-wave's tr read touches fall on 8 distinct 32-byte bank slots
//   fragment    lane 4 r + p of a 16-lane group addresses (k-row r, columns 4 p .. 4 p + 3) of the 4 x 16 block and
//               receives column (lane & 15) of its four rows: k = 32 s + 8 q + 0..3, then + 4..7 with the second read
// SPLIT + HM (accGradParameters of a layer whose output has too few 256 x 256 tiles to fill the CUs: the 784 x 4096 gradient of the
// input layer): the PAIR is split over workgroups as in gemm_v2.h -- each computes ONE GEMM of the pair with the single-pass loop and
// writes only the outputs that depend on it (EpiDw::part) -- over HALF-HEIGHT tiles, 128 x 256 (wave tile 64 x 64 = acc[0..3][], every
// K step one phase of 32 MFMAs per wave), each workgroup walking ALL of K: 785 x 4096 -> 7 x 16 x 2 = 224 workgroups, ONE round, no
// partial tiles to hand over. The output may be ragged in M (rows past m_dim() are computed on zero padding and never stored; the row
// AT m_dim() is the ones row: edge_row). LDS: A parts as a ring of four (A of step t in part t & 3), B triple-buffered as before.
// (r02's form of this launch -- full-height tiles, each GEMM split again over two K halves that met through write-through slabs and
// tickets -- lost to this one, 84 against 68 us in the step, and lives on in tools/lab/ only.)
template <bool DUAL, bool AK, bool BK, class Epi, bool SPLIT = false, bool HM = false>
__global__ __launch_bounds__(512, 2) void gemm_nt_v3(const bf16_t* __restrict__ A, const bf16_t* __restrict__ A2, int64_t lda,
                                                     const bf16_t* __restrict__ B, const bf16_t* __restrict__ B2, int64_t ldb,
                                                     int M, int N, int nk, int tiles_m, int tiles_n, Epi epi_in) {
    static_assert(!(SPLIT && DUAL), "the split form computes one GEMM of the pair per workgroup");
    static_assert(SPLIT == HM && (!HM || (AK && BK)), "the pair-split launch is the half-height K-major form");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    V3_ST(0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform on the scalar side
    // Static priority for the second-dispatched half of the workgroup (MI355X_MICROARCH.md, two waves per SIMD, item 4: the
    // younger wave of a SIMD loses every issue arbitration at equal priority). Measured on the wide step, two rounds of
    // three builds on one box: 0.824 / 0.827 ms against 0.828 / 0.830 without and 0.827 / 0.832 with the OTHER half raised.
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    const int wr = wave >> 2, wc = wave & 3;
    Epi epi = epi_in;

    // ---- block -> tile mapping: as gemm_v2.h (blocks that share an XCD get a compact 4 x 8 group of tiles)
    const int nblk = tiles_m * tiles_n * (SPLIT ? 2 : 1);
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    if (SPLIT) {
        // part-major: the blocks that share an XCD (a contiguous range of remapped ids) all compute the SAME GEMM of the pair for a
        // compact group of tiles, so they share operand panels in its L2 exactly as the unsplit launch does
        const int ntile = tiles_m * tiles_n;
        const int part = bid / ntile;
        bid -= part * ntile;
        epi.set_part(1 + part);
        if (part) { A = A2; B = B2; }
    }
    int tm, tn;
    {
        // (HM: an A panel is half the size of a B panel and a round is 28 workgroups per XCD -- 8 x 4 groups: for the 7 x 16 tiles of
        // the 784 x 4096 gradient every XCD holds ALL of x^T and four panels of g^T, 15 MB per GEMM instead of 20)
        constexpr int GM = HM ? 8 : 4, GN = HM ? 4 : 8;
        const int per_band = GM * tiles_n;
        const int band = bid / per_band;
        const int in_band = bid - band * per_band;
        const int band_rows = min(GM, tiles_m - band * GM);
        const int full = band_rows * GN;
        const int grp = in_band / full;
        const int in_grp = in_band - grp * full;
        const int grp_cols = min(GN, tiles_n - grp * GN);
        tm = min(band * GM + in_grp / grp_cols, tiles_m - 1);
        tn = grp * GN + in_grp % grp_cols;
    }
    const int m0 = tm * (HM ? V3_BM / 2 : V3_BM), n0 = tn * V3_BN;

    // ---- LDS-DMA source offsets (elements), shared by both passes. A part P, DMA d: 8-row group g = wave + 8 d of
    // the part's 128 rows; part row s belongs to wave row s >> 6 and is tile row (s >> 6) * 128 + 64 P + (s & 63).
    int a_off[2][2], b_off[4];                 // 32-bit: the launcher checks rows x ld < 2^31
    auto hk = [](int kr) { return (kr & 3) | (((kr >> 3) & 1) << 2); };
#pragma unroll
    for (int P = 0; P < 2; ++P)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            if (AK) {                          // instruction g = wave + 8 d: k-rows 4 g .. 4 g + 3, 16 chunks each
                const int kr = 4 * (wave + 8 * d) + (lane >> 4);
                const int cs = (lane & 15) ^ (hk(kr) << 1);
                a_off[P][d] = kr * (int)lda + m0 + (cs >> 3) * (HM ? 64 : 128) + 64 * P + (cs & 7) * 8;      // (HM: P = 0 only)
            } else {
                const int s = 8 * (wave + 8 * d) + (lane >> 3);
                const int row = (s >> 6) * 128 + 64 * P + (s & 63);
                const int chunk = (lane & 7) ^ ((s >> 1) & 7);
                a_off[P][d] = min(m0 + row, M - 1) * (int)lda + chunk * 8;
            }
        }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (BK) {                              // instruction g = wave + 8 d: k-rows 2 g, 2 g + 1, 32 chunks each
            const int kr = 2 * (wave + 8 * d) + (lane >> 5);
            const int cs = (lane & 31) ^ (hk(kr) << 1);
            b_off[d] = kr * (int)ldb + n0 + cs * 8;
        } else {
            const int row = 8 * (wave + 8 * d) + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            b_off[d] = min(n0 + row, N - 1) * (int)ldb + chunk * 8;
        }
    }
    const int a_kstep = AK ? V2_BK * (int)lda : V2_BK, b_kstep = BK ? V2_BK * (int)ldb : V2_BK;
    const bf16_t* Ap = A;
    const bf16_t* Bp = B;
#define V3_KT(t) (t)
    // The LDS-DMA pieces go out in their BUFFER form: resource in SGPRs, the lane's byte offset in one VGPR, the K step in
    // an SGPR -- no 64-bit address arithmetic per piece, and the piece issues a few cycles sooner (lab: -2.5 ... -5 %
    // on the pipelined loops, which are bound by exactly this issue). The host pass never runs the body.
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, 0x7fffffff, 0x00020000);
#define V3_SET_OPERANDS(a_, b_) do { Ap = (a_); Bp = (b_); ra = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, 0x7fffffff, 0x00020000); \
                                     rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, 0x7fffffff, 0x00020000); } while (0)
#else
#define V3_SET_OPERANDS(a_, b_) do { Ap = (a_); Bp = (b_); } while (0)
#endif
    auto dma_a = [&](int t, auto P_c, auto d_c) {
        constexpr int P = decltype(P_c)::value, D = decltype(d_c)::value;
        unsigned char* dst = lds + ((t & 1) * 2 + P) * V3_APART + (wave + 8 * D) * 1024;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)dst, 16, (int)(2u * (unsigned)a_off[P][D]),
                                                 (int)(2u * (unsigned)(V3_KT(t) * a_kstep)), 0, 0);
#else
        __builtin_amdgcn_global_load_lds((gptr_t)(Ap + (a_off[P][D] + V3_KT(t) * a_kstep)), (lptr_t)dst, 16, 0, 0);
#endif
    };
    auto dma_b = [&](int t, int slot, auto d_c) {
        constexpr int D = decltype(d_c)::value;
        unsigned char* dst = lds + 4 * V3_APART + slot * V3_BTILE + (wave + 8 * D) * 1024;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)dst, 16, (int)(2u * (unsigned)b_off[D]),
                                                 (int)(2u * (unsigned)(V3_KT(t) * b_kstep)), 0, 0);
#else
        __builtin_amdgcn_global_load_lds((gptr_t)(Bp + (b_off[D] + V3_KT(t) * b_kstep)), (lptr_t)dst, 16, 0, 0);
#endif
    };
    // the main loop's form: LDS destination and K offset are LOOP-CARRIED scalars (run_pass), so a piece costs one
    // scalar add for M0 instead of six scalar operations recomputing both from the step index (lab: the address work
    // was 8-10 % of the loop, which is bound by the issue of these pieces)
    auto dma_a_at = [&](unsigned lds_off, int so_bytes, auto P_c, auto d_c) {
        constexpr int P = decltype(P_c)::value, D = decltype(d_c)::value;
        unsigned char* dst = lds + lds_off + (P * V3_APART + D * 8192) + wave * 1024;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)dst, 16, (int)(2u * (unsigned)a_off[P][D]), so_bytes, 0, 0);
#else
        __builtin_amdgcn_global_load_lds((gptr_t)(Ap + a_off[P][D] + so_bytes / 2), (lptr_t)dst, 16, 0, 0);
#endif
    };
    auto dma_b_at = [&](unsigned lds_off, int so_bytes, auto d_c) {
        constexpr int D = decltype(d_c)::value;
        unsigned char* dst = lds + lds_off + (4 * V3_APART + D * 8192) + wave * 1024;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)dst, 16, (int)(2u * (unsigned)b_off[D]), so_bytes, 0, 0);
#else
        __builtin_amdgcn_global_load_lds((gptr_t)(Bp + b_off[D] + so_bytes / 2), (lptr_t)dst, 16, 0, 0);
#endif
    };
    std::integral_constant<int, 0> c0;
    std::integral_constant<int, 1> c1;
    std::integral_constant<int, 2> c2;
    std::integral_constant<int, 3> c3;

    // ---- fragment read offsets (bytes inside an A part / a B tile); swizzle as in gemm_v2.h
    const int rsw = (lane & 15) >> 1;
    const int q = lane >> 4;
    int a_rd[2], b_rd[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int csw = ((4 * s + q) ^ rsw) * 16;
        a_rd[s] = (wr * 64 + (lane & 15)) * 128 + csw;
        b_rd[s] = (wc * 64 + (lane & 15)) * 128 + csw;
    }

    // K-major operands: transpose reads. Lane j = lane & 15 of its group addresses k-row (j >> 2), columns 4 (j & 3) ..
    const int trq = (lane & 15) >> 2, trp = lane & 3;
    const int thx = (trq | ((q & 1) << 2)) << 1;              // 2 h(k) of this lane's k-rows (same for both reads)
    int a_tr[4], b_tr[4];                                     // byte offset of the lane's 8 bytes, per 16-row block
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_tr[i] = (8 * q + trq) * 256 + (((wr * 8 + 2 * i + (trp >> 1)) ^ thx) * 16) + (trp & 1) * 8;
        b_tr[i] = (8 * q + trq) * 512 + (((wc * 8 + 2 * i + (trp >> 1)) ^ thx) * 16) + (trp & 1) * 8;
    }
    // The fragments. K-major operands: transpose reads through INLINE ASM -- through the builtin hipcc cannot tell that the read
    // does not alias the LDS-DMA writes in flight and drains vmcnt(0) in front of every group (the whole pipeline, every half phase:
    // 245 vs 191 us at 4096^3); inline asm is outside that bookkeeping, and the DMA -> read ordering is this kernel's own counted
    // vmcnt + barrier, as for the plain reads. The reads are ISSUED by one asm statement and handed to the compiler by a later,
    // empty one that names their registers ("+v": landed4 in the K steps), behind the s_waitcnt lgkmcnt(0) that closes the cluster;
    // tools/audit_tr_reads.py checks on the generated code that nothing touches a register in between.
    // (r01-r03's loops issued such reads BETWEEN MFMAs and had to keep them out of the registers of the wave's last eight MFMAs -- a
    // lab build that did not computed wrong products, differently from launch to launch; tools/lab/gemm_v3_r04.h keeps that form. In
    // the alternating K steps every read sits behind a barrier that follows the cluster's last MFMA: tools/check_lds_war.py.)
    typedef __attribute__((address_space(3))) unsigned char* ldsb_t;
    auto load_a = [&](const unsigned char* sa, auto s_c, bf16x8 (&f)[4]) {          // K-contiguous operands: four ds_read_b128
        constexpr int S = decltype(s_c)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const bf16x8*>(sa + a_rd[S] + i * 2048);
    };
    auto load_b = [&](const unsigned char* sb, auto s_c, bf16x8 (&f)[4]) {
        constexpr int S = decltype(s_c)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = *reinterpret_cast<const bf16x8*>(sb + b_rd[S] + j * 2048);
    };

    f32x4 acc[8][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // ---- one pass: acc = sum over the nk K steps of (Ap tile) x (Bp tile)^T
    // pipeline fill, in the steady state's issue order: B(0) half, A part 0 (0), B(0) half, A part 1 (0), then step 1
    auto prologue = [&]() {
        dma_b(0, 0, c0); dma_b(0, 0, c1); dma_a(0, c0, c0); dma_a(0, c0, c1);
        dma_b(0, 0, c2); dma_b(0, 0, c3); dma_a(0, c1, c0); dma_a(0, c1, c1);
        if (nk > 1) {
            dma_b(1, 1, c0); dma_b(1, 1, c1); dma_a(1, c0, c0); dma_a(1, c0, c1);
            dma_b(1, 1, c2); dma_b(1, 1, c3);
        }
    };
    // the half-height loop's fill, in the order its step issues the pieces -- A three steps ahead, B two: A(0), A(1),
    // B(0), A(2), B(1) -- so that its counted waits hold from the first step on
    auto prologue_hm_pp = [&]() {
        dma_a_at(0u, 0, c0, c0); dma_a_at(0u, 0, c0, c1);
        if (nk > 1) { dma_a_at((unsigned)V3_APART, 2 * a_kstep, c0, c0); dma_a_at((unsigned)V3_APART, 2 * a_kstep, c0, c1); }
        dma_b(0, 0, c0); dma_b(0, 0, c1); dma_b(0, 0, c2); dma_b(0, 0, c3);
        if (nk > 2) { dma_a_at(2u * V3_APART, 4 * a_kstep, c0, c0); dma_a_at(2u * V3_APART, 4 * a_kstep, c0, c1); }
        if (nk > 1) { dma_b(1, 1, c0); dma_b(1, 1, c1); dma_b(1, 1, c2); dma_b(1, 1, c3); }
    };
    auto run_pass = [&]() {
        // loop-carried scalars of step t: LDS offset of A part (t & 1) * 2, of B slots t % 3 and (t + 2) % 3, and the
        // operands' byte offsets of step t (see dma_a_at)
        unsigned oa = 0, ob = 0, ob2 = 2 * V3_BTILE;
        int ka = 0, kb = 0;
        const int da = 2 * a_kstep, db = 2 * b_kstep;
        // ---- the full-height K step: per phase two ALTERNATING clusters, the workgroup's two halves one cluster apart (r04)
        //   M cluster  every LDS read of the phase (A part: 8 fragments; phase 0 also the step's 8 B fragments) with the phase's four
        //              pieces BETWEEN the read groups, then s_waitcnt vmcnt(10 | 8) lgkmcnt(0) + s_barrier
        //   C cluster  the phase's 32 MFMAs, nothing else, then s_barrier
        // Waves 4-7 (the SIMD partners of waves 0-3) enter the pass one barrier late and waves 0-3 leave it one barrier late, so
        // between any two barriers one wave of every SIMD is in an M cluster and the other in a C cluster: the matrix pipe gets an
        // uninterrupted stream from one wave while the other pays the LDS round trips and the pieces' issue stalls. (r01-r03's step had
        // both waves of a SIMD in the same phase behind every barrier, reads and pieces interleaved with the MFMAs: they stalled
        // together, 56 % MFMA-busy; this form 65 % at 12 % fewer cycles -- tools/lab/README.md has that form and the other placements of
        // the pieces that were priced: behind the reads, in the C cluster, two and two, one burst per wave, k-half-0 reads under the
        // MFMAs.) Same pieces in the same order as r03's step, so the same counted waits; same MFMA order per accumulator: the same bits.
        // RAW: a wave waits for its pieces at the END of an M cluster; the first read of that data is by the other half, behind
        //      that barrier, in the next slot. WAR: a part is refilled from the M cluster that follows, by at least one barrier,
        //      the slot in which the late half read it (its reads retired by the lgkmcnt(0) in front of that slot's barrier).
        // Stamp hooks (lab builds only): V3_ST(10) M cluster starts, (11) C cluster starts, (12) C cluster's MFMAs issued.
        auto kstep_pp = [&](const int t, auto tail_c) {
            constexpr bool TAIL = decltype(tail_c)::value;
            const bool n1 = TAIL ? (t + 1 < nk) : true, n2 = TAIL ? (t + 2 < nk) : true;
            const int so_a1 = ka + da, so_a2 = ka + 2 * da, so_b2 = kb + 2 * db;
            const unsigned oa_next = oa ^ (2u * V3_APART);        // A part pair of step t + 1
            const unsigned char* sb = lds + 4 * V3_APART + ob;
            bf16x8 bf[2][4];
            bf16x4 bl[2][4], bh[2][4];
            // eight transpose reads (four fragments of one k-half), issued only: the cluster's closing lgkmcnt(0) retires them and
            // landed4 hands the registers to the compiler behind it
            auto tr_issue4 = [&](unsigned a0, unsigned a1, unsigned a2, unsigned a3, auto lo_c, auto hi_c, bf16x4 (&l)[4], bf16x4 (&h)[4]) {
                constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
                asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%12\n\tds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
                             "ds_read_b64_tr_b16 %2, %9 offset:%12\n\tds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
                             "ds_read_b64_tr_b16 %4, %10 offset:%12\n\tds_read_b64_tr_b16 %5, %10 offset:%13\n\t"
                             "ds_read_b64_tr_b16 %6, %11 offset:%12\n\tds_read_b64_tr_b16 %7, %11 offset:%13"
                             : "=&v"(l[0]), "=&v"(h[0]), "=&v"(l[1]), "=&v"(h[1]), "=&v"(l[2]), "=&v"(h[2]), "=&v"(l[3]), "=&v"(h[3])
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(LO), "n"(HI)
                             : "memory");
            };
            auto landed4 = [&](bf16x4 (&l)[4], bf16x4 (&h)[4]) {
                asm volatile("" : "+v"(l[0]), "+v"(h[0]), "+v"(l[1]), "+v"(h[1]), "+v"(l[2]), "+v"(h[2]), "+v"(l[3]), "+v"(h[3])::"memory");
            };
            auto phase = [&](auto P_c) {
                constexpr int P = decltype(P_c)::value;
                const unsigned char* sa = lds + oa + P * V3_APART;
                bf16x8 af[2][4];
                bf16x4 al[2][4], ah[2][4];
                // ---------------- M cluster
                V3_ST(10);
                auto rd_b = [&](auto s_c) {
                    constexpr int S = decltype(s_c)::value;
                    if constexpr (BK) {
                        const unsigned bb = (unsigned)(uintptr_t)(ldsb_t)sb;
                        tr_issue4(bb + b_tr[0], bb + b_tr[1], bb + b_tr[2], bb + b_tr[3], std::integral_constant<int, S * 16384>(),
                                  std::integral_constant<int, S * 16384 + 2048>(), bl[S], bh[S]);
                    } else {
                        load_b(sb, s_c, bf[S]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                auto rd_a = [&](auto s_c) {
                    constexpr int S = decltype(s_c)::value;
                    if constexpr (AK) {
                        const unsigned ba = (unsigned)(uintptr_t)(ldsb_t)sa;
                        tr_issue4(ba + a_tr[0], ba + a_tr[1], ba + a_tr[2], ba + a_tr[3], std::integral_constant<int, S * 8192>(),
                                  std::integral_constant<int, S * 8192 + 1024>(), al[S], ah[S]);
                    } else {
                        load_a(sa, s_c, af[S]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                // the phase's four pieces, in issue order k = 0..3 (r03's pieces in r03's order):
                //   phase 0 of step t: A part 1 of t + 1 (x2), B of t + 2 (x2);  phase 1: A part 0 of t + 2 (x2), B of t + 2 (x2)
                auto piece = [&](int k) {
                    if constexpr (P == 0) {
                        if (k == 0 && n1) dma_a_at(oa_next, so_a1, c1, c0);
                        if (k == 1 && n1) dma_a_at(oa_next, so_a1, c1, c1);
                        if (k == 2 && n2) dma_b_at(ob2, so_b2, c0);
                        if (k == 3 && n2) dma_b_at(ob2, so_b2, c1);
                    } else {
                        if (k == 0 && n2) dma_a_at(oa, so_a2, c0, c0);      // (t + 2) & 1 == t & 1: this step's part 0
                        if (k == 1 && n2) dma_a_at(oa, so_a2, c0, c1);
                        if (k == 2 && n2) dma_b_at(ob2, so_b2, c2);
                        if (k == 3 && n2) dma_b_at(ob2, so_b2, c3);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                // (the last two steps choose their counted wait at run time: their reads are retired by a wait of its own IN FRONT of that
                // choice, so that no path through it -- tools/check_barrier_lgkm.py walks them all -- reaches a barrier with a read in flight)
                auto tail_reads_back = [&]() { if constexpr (TAIL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
                // a piece BETWEEN the read groups: the wave that issues one gets its next issue slot 30-70 cycles later (the half's four
                // waves offer the texture path their pieces at the same moment); read groups in between overlap part of that
                if constexpr (P == 0) {
                    rd_b(c0); piece(0);
                    rd_b(c1); piece(1);
                    rd_a(c0); piece(2);
                    rd_a(c1); piece(3);
                    // every piece but the youngest ten landed: A part 1 of this step (read in the NEXT M cluster; the other half reads it
                    // one slot after this barrier at the earliest)
                    tail_reads_back();
                    if (TAIL && t == nk - 1) v2_wait_barrier<0>(); else if (TAIL && t == nk - 2) v2_wait_barrier<8>(); else v2_wait_barrier<10>();
                } else {
                    rd_a(c0); piece(0); piece(1);
                    rd_a(c1); piece(2); piece(3);
                    // ... but the youngest eight: A part 0 and B of step t + 1
                    tail_reads_back();
                    if (TAIL && t >= nk - 2) { if (t == nk - 1) v2_wait_barrier<0>(); else v2_wait_barrier<2>(); } else v2_wait_barrier<8>();
                }
                if constexpr (P == 0 && BK) {
                    landed4(bl[0], bh[0]); landed4(bl[1], bh[1]);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int j = 0; j < 4; ++j) bf[s2][j] = __builtin_shufflevector(bl[s2][j], bh[s2][j], 0, 1, 2, 3, 4, 5, 6, 7);
                }
                if constexpr (AK) {
                    landed4(al[0], ah[0]); landed4(al[1], ah[1]);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int i = 0; i < 4; ++i) af[s2][i] = __builtin_shufflevector(al[s2][i], ah[s2][i], 0, 1, 2, 3, 4, 5, 6, 7);
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---------------- C cluster
                V3_ST(11);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[4 * P + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s2][i], bf[s2][j], acc[4 * P + i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                V3_ST(12);
                asm volatile("s_barrier" ::: "memory");
            };
            phase(c0);
            phase(c1);
            oa = oa_next;
            ob = ob == 2 * V3_BTILE ? 0 : ob + V3_BTILE;
            ob2 = ob2 == 2 * V3_BTILE ? 0 : ob2 + V3_BTILE;
            ka += da; kb += db;
        };
        if constexpr (HM) {
            // ---- the half-height form as alternating clusters (r04), one phase per K step on the wave's 64 x 64 tile (acc[0..3][]):
            //   M(t)  the step's 16 fragments (32 transpose reads) with its six pieces -- A(t + 3) into part (t + 3) & 3, B(t + 2) into slot
            //         (t + 2) % 3, both last read during step t - 1 -- between them, then s_waitcnt vmcnt(6) lgkmcnt(0) + s_barrier
            //   C(t)  32 MFMAs, then s_barrier
            // waves 4-7 one barrier behind waves 0-3, exactly as kstep_pp. RAW: the wait that closes M(t) leaves only M(t)'s own six pieces
            // in flight, i.e. A(t + 1) and B(t + 1) have landed, one barrier before the other half reads them. WAR: a buffer read in M(t)
            // (by the late half one slot after the early half) is refilled from M(t + 1) on. MFMA order per accumulator as before.
            bf16x4 al[2][4], ah[2][4], bl[2][4], bh[2][4];
            unsigned oa4 = 0;                                       // A part of step t: (t & 3) * V3_APART
            const unsigned lbase = (unsigned)(uintptr_t)(ldsb_t)lds;
            auto tr2 = [&](unsigned a0, unsigned a1, auto lo_c, auto hi_c, bf16x4& l0, bf16x4& h0, bf16x4& l1, bf16x4& h1) {
                constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
                asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%6\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\t"
                             "ds_read_b64_tr_b16 %2, %5 offset:%6\n\tds_read_b64_tr_b16 %3, %5 offset:%7"
                             : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1) : "v"(a0), "v"(a1), "n"(LO), "n"(HI) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            };
            auto landed4 = [&](bf16x4 (&l)[4], bf16x4 (&h)[4]) {
                asm volatile("" : "+v"(l[0]), "+v"(h[0]), "+v"(l[1]), "+v"(h[1]), "+v"(l[2]), "+v"(h[2]), "+v"(l[3]), "+v"(h[3])::"memory");
            };
            auto step_pp = [&](const int t, auto tail_c) {
                constexpr bool TAIL = decltype(tail_c)::value;
                const bool n2 = TAIL ? (t + 2 < nk) : true, n3 = TAIL ? (t + 3 < nk) : true;
                const unsigned oa3 = (oa4 + 3u * V3_APART) & (4u * V3_APART - 1u);
                const int so_a3 = ka + 3 * da, so_b2 = kb + 2 * db;
                const unsigned ba = lbase + oa4, bb = lbase + 4 * V3_APART + ob;
                auto piece = [&](int k) {
                    if (k == 0 && n3) dma_a_at(oa3, so_a3, c0, c0);
                    if (k == 1 && n3) dma_a_at(oa3, so_a3, c0, c1);
                    if (k == 2 && n2) dma_b_at(ob2, so_b2, c0);
                    if (k == 3 && n2) dma_b_at(ob2, so_b2, c1);
                    if (k == 4 && n2) dma_b_at(ob2, so_b2, c2);
                    if (k == 5 && n2) dma_b_at(ob2, so_b2, c3);
                    __builtin_amdgcn_sched_barrier(0);
                };
                std::integral_constant<int, 1024> a1k; std::integral_constant<int, 8192> a2k; std::integral_constant<int, 8192 + 1024> a3k;
                std::integral_constant<int, 2048> b1k; std::integral_constant<int, 16384> b2k; std::integral_constant<int, 16384 + 2048> b3k;
                // ---------------- M cluster: two fragments (four reads), a piece, ...
                V3_ST(10);
                tr2(bb + b_tr[0], bb + b_tr[1], c0, b1k, bl[0][0], bh[0][0], bl[0][1], bh[0][1]); piece(0);
                tr2(bb + b_tr[2], bb + b_tr[3], c0, b1k, bl[0][2], bh[0][2], bl[0][3], bh[0][3]); piece(1);
                tr2(ba + a_tr[0], ba + a_tr[1], c0, a1k, al[0][0], ah[0][0], al[0][1], ah[0][1]); piece(2);
                tr2(ba + a_tr[2], ba + a_tr[3], c0, a1k, al[0][2], ah[0][2], al[0][3], ah[0][3]); piece(3);
                tr2(bb + b_tr[0], bb + b_tr[1], b2k, b3k, bl[1][0], bh[1][0], bl[1][1], bh[1][1]); piece(4);
                tr2(bb + b_tr[2], bb + b_tr[3], b2k, b3k, bl[1][2], bh[1][2], bl[1][3], bh[1][3]); piece(5);
                tr2(ba + a_tr[0], ba + a_tr[1], a2k, a3k, al[1][0], ah[1][0], al[1][1], ah[1][1]);
                tr2(ba + a_tr[2], ba + a_tr[3], a2k, a3k, al[1][2], ah[1][2], al[1][3], ah[1][3]);
                if constexpr (TAIL) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // (reads retired in front of the run-time choice: kstep_pp)
                    if (n3) v2_wait_barrier<6>(); else if (n2) v2_wait_barrier<4>(); else v2_wait_barrier<0>();
                } else {
                    v2_wait_barrier<6>();
                }
                landed4(bl[0], bh[0]); landed4(bl[1], bh[1]); landed4(al[0], ah[0]); landed4(al[1], ah[1]);
                __builtin_amdgcn_sched_barrier(0);
                // ---------------- C cluster
                V3_ST(11);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_shufflevector(al[s2][i], ah[s2][i], 0, 1, 2, 3, 4, 5, 6, 7),
                                                                               __builtin_shufflevector(bl[s2][j], bh[s2][j], 0, 1, 2, 3, 4, 5, 6, 7), acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                V3_ST(12);
                asm volatile("s_barrier" ::: "memory");
                oa4 = (oa4 + V3_APART) & (4u * V3_APART - 1u);
                ob = ob == 2 * V3_BTILE ? 0 : ob + V3_BTILE;
                ob2 = ob2 == 2 * V3_BTILE ? 0 : ob2 + V3_BTILE;
                ka += da; kb += db;
            };
            // every wave: A(0), B(0) have landed (what the fill issued behind them may fly), for everybody behind the barrier
            if (nk > 2) v2_wait_barrier<6>(); else if (nk > 1) v2_wait_barrier<4>(); else v2_wait_barrier<0>();
            V3_ST(13);
            if (wave >= 4) asm volatile("s_barrier" ::: "memory");
            int t = 0;
            for (; t + 3 < nk; ++t) step_pp(t, std::false_type());
            for (; t < nk; ++t) step_pp(t, std::true_type());
            if (wave < 4) asm volatile("s_barrier" ::: "memory");
            V3_ST(14);
            return;
        }
        // every wave: the pieces of (0, 0) have landed, for everybody behind the barrier; then the late half's extra barrier, which
        // the early half answers with one more at the end (every wave passes 2 + 4 nk barriers)
        int t = 0;
        if (nk > 1) v2_wait_barrier<8>(); else v2_wait_barrier<2>();
        V3_ST(13);
        if (wave >= 4) asm volatile("s_barrier" ::: "memory");
        for (; t + 2 < nk; ++t) kstep_pp(t, std::false_type());
        for (; t < nk; ++t) kstep_pp(t, std::true_type());
        if (wave < 4) asm volatile("s_barrier" ::: "memory");
        V3_ST(14);
    };

    // ---- epilogue: the LDS re-layout of gemm_v2.h, one QUARTER of the wave tile (64 m x 32 n) at a time -- a whole
    // 128 x 64 tile's rows would not fit the register file beside the accumulators.
    // Quarter qq = (hh, jj): m-blocks 4 hh .. 4 hh + 3, n-blocks 2 jj, 2 jj + 1. Per wave 18944 B of LDS:
    // fp32 rows [32 n][SP] + two transposed staging tiles [64 m][TQ] in the operand type.
    typedef typename Epi::elem_t ET;
    constexpr int SP = 68;                                    // fp32 staging pitch (floats)
    constexpr int TQ = 40;                                    // transposed staging pitch (elements): 32 n + pad, 16-B rows
    constexpr int WAVE_LDS = 32 * SP * 4 + 2 * 64 * TQ * 2;   // 18944
    static_assert(8 * WAVE_LDS <= V3_LDS, "epilogue staging must fit the allocation");
    float* st = reinterpret_cast<float*>(lds + wave * WAVE_LDS);
    ET* tl1 = reinterpret_cast<ET*>(lds + wave * WAVE_LDS + 32 * SP * 4);
    ET* tl2 = tl1 + 64 * TQ;
    const int c16 = lane & 15, q4 = lane >> 4;
    // a quarter in the MFMA layout, v[2 i + jl] = block (m-block i, n-block jl) -> rows of n, 4 consecutive m per lane
    auto relayout = [&](const f32x4 (&v)[8], f32x4 (&rows)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jl = 0; jl < 2; ++jl)
                *reinterpret_cast<f32x4*>(st + (jl * 16 + c16) * SP + i * 16 + q4 * 4) = v[2 * i + jl];
        // same-wave LDS operations complete in order: the reads below see the writes above
#pragma unroll
        for (int p = 0; p < 8; ++p) rows[p] = *reinterpret_cast<const f32x4*>(st + (4 * p + q4) * SP + 4 * c16);
    };
    const typename Epi::Lane eln = epi.lane_init(q4, 4 * c16);

    auto for_quarters = [&](auto&& fn) {
        fn(c0, c0, 0); fn(c0, c1, 1);
        if constexpr (!HM) { fn(c1, c0, 2); fn(c1, c1, 3); }
    };

    if (DUAL) {
        // The pair as two passes over ONE accumulator (the functor's FOLD protocol, epilogues.h): the pair's second GEMM
        // first; then every accumulator quad, as it stands in the MFMA layout (lane (q4, c16): 4 consecutive m at one n),
        // is turned into the term the first GEMM accumulates on top of -- in registers, no LDS, nothing parked.
        V3_SET_OPERANDS(A2, B2);
        zero_acc();
        prologue();
        run_pass();
        V3_ST(5);
        // Pass 2's pipeline fill is issued BEFORE the fold, so its DMAs land while the fold computes. (The fold's own
        // loads and stores are younger than those DMAs on the in-order vmcnt counter: the counted waits of run_pass
        // only get stricter.)
        v2_lds_barrier();                                     // every wave is done reading pass 1's last K step (reads retired)
        if constexpr (Epi::FOLD_STAGE == 1) {
            // FOLD_STAGE functors (see the fold below): the per-m addend (the bias) of the wave's 128 m is parked in the
            // wave's LDS staging area NOW, before pass 2's fill is issued -- ONE global load per lane whose wait covers
            // nothing else. (Beside LDS-DMAs in flight hipcc waits vmcnt(0) for any plain load's result, i.e. also for
            // the fold's own stores: a bias load per quad drained the store queue 32 times per fold.)
            constexpr int PITCH0 = 64 * (int)sizeof(typename Epi::fold_st_t) + 16, WSTG0 = 16 * PITCH0 + 512;
            const unsigned stg0 = (unsigned)(uintptr_t)(ldsb_t)(wave < 7 ? lds + 4 * V3_APART + 2 * V3_BTILE + wave * WSTG0
                                                                        : lds + 3 * V3_APART);
            int sl = lane;
            asm volatile("" : "+v"(sl));
            const float* bp = epi.fold_bias_ptr();
            f32x4 bq = f32x4{0.f, 0.f, 0.f, 0.f};
            if (bp) bq = *reinterpret_cast<const f32x4*>(bp + (m0 + wr * 128) + 4 * (sl & 31));
            asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(stg0 + 16 * (sl & 31)), "v"(bq), "n"(16 * PITCH0) : "memory");
        }
        V3_SET_OPERANDS(A, B);
        prologue();
        // opaque to the optimiser from here on: otherwise the fold's address / counter arithmetic is hoisted above
        // pass 1 and held live through its loop, which is already at the register limit (it spilled)
        int fc16 = c16, fq4 = q4, fm0 = m0 + wr * 128, fn0 = n0 + wc * 64;
        asm volatile("" : "+v"(fc16), "+v"(fq4), "+s"(fm0), "+s"(fn0));
        const typename Epi::Lane fln = epi.lane_init(fc16, 4 * fq4);
        // Quads are folded in batches of Epi::FOLD_BATCH m-blocks (4 quads each): a batch's loads fly together, ahead of
        // their first use. (Indices are compile-time constants, not `#pragma unroll` loops over the m-blocks: a loop
        // the optimiser declines to unroll would index the accumulators dynamically and send them all to scratch.)
        auto fold_batch = [&](auto i0_c) {
            constexpr int i0 = decltype(i0_c)::value;
            constexpr int NB = Epi::FOLD_BATCH;
            typename Epi::FPre fp[NB][4];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) fp[b][j] = epi.fold_load(fm0 + 16 * (i0 + b), fn0 + 16 * j, fln);
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i0 + b][j] = epi.fold(fm0 + 16 * (i0 + b), fn0 + 16 * j, fln, acc[i0 + b][j], fp[b][j]);
                    // FOLD_SERIAL quads at a time, also for the optimiser: the next group's coordinates "depend" on this
                    // result (32 interleaved Philox blocks beside 128 live accumulators spill; two at a time fit and
                    // give the VALU two independent dependency chains)
                    if (Epi::FOLD_SERIAL > 0 && (j % (Epi::FOLD_SERIAL > 0 ? Epi::FOLD_SERIAL : 1)) == Epi::FOLD_SERIAL - 1)
                        asm volatile("" : "+s"(fn0), "+s"(fm0)
                                     : "v"(acc[i0 + b][j][0]), "v"(acc[i0 + b][j][1]), "v"(acc[i0 + b][j][2]), "v"(acc[i0 + b][j][3]),
                                       "v"(acc[i0 + b][j ? j - 1 : 0][0]));
                }
        };
        static_assert(8 % Epi::FOLD_BATCH == 0, "FOLD_BATCH divides the 8 m-blocks of a wave tile");
        if constexpr (Epi::FOLD_STAGE == 2) {
            // fp32 form of the staged fold store below (EpiDw's d/dlvars): tile [16 n][64 m] f32, 256-byte rows + 16 B pad;
            // every quad's operand load is issued up front (one batch: the only vmcnt wait of the fold), then per
            // (n-block, four m-blocks): four ds_write_b128, four ds_read_b128, four 16-byte stores of 4 rows x 256 B.
            constexpr int PITCH = 64 * 4 + 16, WSTG = 16 * PITCH;
            static_assert(7 * WSTG <= V3_BTILE && WSTG <= V3_APART, "staging tiles must fit the free LDS");
            const unsigned stg = (unsigned)(uintptr_t)(ldsb_t)(wave < 7 ? lds + 4 * V3_APART + 2 * V3_BTILE + wave * WSTG
                                                                       : lds + 3 * V3_APART);
            int flane = lane;
            asm volatile("" : "+v"(flane));
            const unsigned wr_a = stg + fc16 * PITCH + 16 * fq4;               // this lane's 4 m of m-block b: + 64 b
            const unsigned rd_a = stg + (flane >> 4) * PITCH + (flane & 15) * 16;   // row (lane >> 4) (+ 4 i), 16-byte chunk lane & 15
            float* const outp = epi.fold_st_ptr();
            const int64_t old = epi.fold_st_ld();
            const unsigned olane = (unsigned)((flane >> 4) * (int)old + (flane & 15) * 4);
            typename Epi::FPre fp[8][4];
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) fp[b][j] = epi.fold_load(fm0 + 16 * b, fn0 + 16 * j, fln);
            const typename Epi::FoldK fk = epi.fold_k();
            auto fold_group = [&](auto g_c) {
                constexpr int G = decltype(g_c)::value;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        f32x4 sv;
                        acc[4 * G + b][j] = epi.fold_s(fk, acc[4 * G + b][j], fp[4 * G + b][j], sv);
                        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(wr_a), "v"(sv), "n"(64 * b) : "memory");
                    }
                    f32x4 r0, r1, r2, r3;
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:%5\n\tds_read_b128 %2, %4 offset:%6\n\t"
                                 "ds_read_b128 %3, %4 offset:%7\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                                 : "v"(rd_a), "n"(4 * PITCH), "n"(8 * PITCH), "n"(12 * PITCH) : "memory");
                    float* const row0p = outp + ((int64_t)(fn0 + 16 * j) * old + fm0 + 64 * G);
                    vbnn_store_grad(row0p, olane, r0);
                    vbnn_store_grad(row0p + 4 * old, olane, r1);
                    vbnn_store_grad(row0p + 8 * old, olane, r2);
                    vbnn_store_grad(row0p + 12 * old, olane, r3);
                }
            };
            fold_group(std::integral_constant<int, 0>());
            fold_group(std::integral_constant<int, 1>());
        } else if constexpr (Epi::FOLD_STAGE != 0) {
            // The fold's own output tensor ([n][m], element type ST) leaves through a per-wave LDS tile [16 n][64 m]: the
            // four quads of an (n-block, four m-blocks) group are written as they stand (lane (n = c16, 4 m)), read back
            // as rows and stored as whole 128-byte (2-byte ST) row segments. Free LDS while pass 2's fill is in flight:
            // B slot 2 and A part 3 (the first K step refills them only behind its barrier, which every wave reaches after
            // its fold). LDS accesses go through inline asm: beside LDS-DMAs in flight hipcc would drain vmcnt(0) -- and
            // with it this fold's own stores -- in front of every plain LDS read (see the fragment reads above).
            typedef typename Epi::fold_st_t ST;
            static_assert(sizeof(ST) == 2, "staging tile geometry: 64 m x 2 bytes = one 128-byte row segment");
            constexpr int PITCH = 64 * (int)sizeof(ST) + 16;                   // bytes; 16 rows x 144 B = 2304 B per wave
            constexpr int WSTG = 16 * PITCH + 512;                             // + the wave's 128 per-m addends (fp32)
            static_assert(7 * WSTG <= V3_BTILE && WSTG <= V3_APART, "staging tiles must fit the free LDS");
            const unsigned stg = (unsigned)(uintptr_t)(ldsb_t)(wave < 7 ? lds + 4 * V3_APART + 2 * V3_BTILE + wave * WSTG
                                                                       : lds + 3 * V3_APART);
            const unsigned wr_a = stg + fc16 * PITCH + 8 * fq4;                // this lane's 4 m of m-block b: + 32 b
            int flane = lane;
            asm volatile("" : "+v"(flane));
            const unsigned bias_a = stg + 16 * fq4;                            // m-block i: + 16 PITCH + 64 i
            const unsigned rd_a = stg + (flane >> 3) * PITCH + (flane & 7) * 16;   // row (lane >> 3) (+ 8), m chunk lane & 7
            ST* const outp = epi.fold_st_ptr();
            const int64_t old = epi.fold_st_ld();
            const bool st_stream = epi.fold_st_stream();
            const unsigned olane = (unsigned)((flane >> 3) * (int)old + (flane & 7) * 8);
            auto fold_group = [&](auto g_c) {
                constexpr int G = decltype(g_c)::value;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        f32x4 b4;
                        asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b4) : "v"(bias_a), "n"(16 * PITCH + 64 * (4 * G + b)) : "memory");
                        float sv[4];
                        acc[4 * G + b][j] = epi.fold_s(fm0 + 16 * (4 * G + b), fn0 + 16 * j, fln, acc[4 * G + b][j], b4, sv);
                        const bf16x4 pk = bf16x4{Elt<ST>::to(sv[0]), Elt<ST>::to(sv[1]), Elt<ST>::to(sv[2]), Elt<ST>::to(sv[3])};
                        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(wr_a), "v"(pk), "n"(32 * b) : "memory");
                        if (Epi::FOLD_SERIAL > 0)
                            asm volatile("" : "+s"(fn0), "+s"(fm0)
                                         : "v"(acc[4 * G + b][j][0]), "v"(acc[4 * G + b][j][1]), "v"(acc[4 * G + b][j][2]), "v"(acc[4 * G + b][j][3]));
                    }
                    bf16x8 r0, r1;
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(r0), "=&v"(r1) : "v"(rd_a), "n"(8 * PITCH) : "memory");
                    ST* const row0p = outp + ((int64_t)(fn0 + 16 * j) * old + fm0 + 64 * G);
                    vbnn_store_stream(row0p, olane, r0, st_stream);
                    vbnn_store_stream(row0p + 8 * old, olane, r1, st_stream);
                }
            };
            fold_group(std::integral_constant<int, 0>());
            fold_group(std::integral_constant<int, 1>());
        } else {
        fold_batch(std::integral_constant<int, 0>());
        if constexpr (Epi::FOLD_BATCH < 8) fold_batch(std::integral_constant<int, Epi::FOLD_BATCH>());
        if constexpr (Epi::FOLD_BATCH < 4) {
            fold_batch(std::integral_constant<int, 2 * Epi::FOLD_BATCH>()); fold_batch(std::integral_constant<int, 3 * Epi::FOLD_BATCH>());
        }
        if constexpr (Epi::FOLD_BATCH < 2) {
            fold_batch(std::integral_constant<int, 4>()); fold_batch(std::integral_constant<int, 5>());
            fold_batch(std::integral_constant<int, 6>()); fold_batch(std::integral_constant<int, 7>());
        }
        }
        V3_ST(6);
        run_pass();
    } else {
        zero_acc();
        if constexpr (HM) prologue_hm_pp(); else prologue();
        run_pass();
    }
    __syncthreads();
    V3_ST(1);

    // ---- the classifier head's logits from this tile (EpiFwd::head_slots; see epilogues.h). The accumulators ARE y (the fold
    // put b + sqrt(v) z under pass 2): lane (q4, c16) holds m = 16 i + 4 q4 + 0..3 of row n = 16 j + c16 in acc[i][j]. An MFMA
    // contracts over its k slots whatever they stand for, so slot (q4, e) of the B operand is taken to mean m = 16 i0 + 4 q4 + e
    // (e < 4) and 16 i1 + 4 q4 + e - 4 (e >= 4) for a PAIR of m-blocks (i0, i1): the B fragment is then the two accumulator
    // quads as they stand (ReLU, rounded to bf16 -- the values `h` gets), no LDS trip, and the A fragment is the final weight's
    // row `class = c16` at the same m: two 8-byte loads per pair. 16 MFMAs per wave.
    if constexpr (!SPLIT && Epi::HEAD) {
        if (epi.head_slots) {
            const int hq4 = lane >> 4, hc16 = lane & 15;
            const bf16_t* wrow = epi.head_w3 + (int64_t)min(hc16, epi.head_C - 1) * epi.head_ld_w + (m0 + wr * 128) + 4 * hq4;
            bf16x8 wf[4];
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(wrow + 32 * pr);
                const bf16x4 hi = *reinterpret_cast<const bf16x4*>(wrow + 32 * pr + 16);
                wf[pr] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            const bool relu = epi.relu != 0;
            float* const sl = epi.head_slots + ((int64_t)(tm * 2 + wr) * epi.N + (n0 + wc * 64)) * 16 + hc16 * 16 + 4 * hq4;
            // The SINGLE-GEMM launch (MAP / weight-noise forward: zero_acc, then one pass) holds W x WITHOUT the bias -- there
            // apply_fast adds it only when it stores h (the logits came from relu(W x), wrong as soon as a VB bias
            // is non-zero, i.e. after the first SGD step). The lane's 32 bias values, added exactly as apply_fast adds them.
            f32x4 hbias[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) hbias[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (!DUAL) {
                const float* bp = epi.fold_bias_ptr();
                if (bp) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) hbias[i] = *reinterpret_cast<const f32x4*>(bp + (m0 + wr * 128) + 16 * i + 4 * hq4);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    f32x4 a0 = acc[2 * pr][j], a1 = acc[2 * pr + 1][j];
                    if constexpr (!DUAL) { a0 += hbias[2 * pr]; a1 += hbias[2 * pr + 1]; }
                    bf16x8 hb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hb[e] = (bf16_t)(relu ? fmaxf(a0[e], 0.f) : a0[e]);
                        hb[4 + e] = (bf16_t)(relu ? fmaxf(a1[e], 0.f) : a1[e]);
                    }
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[pr], hb, d, 0, 0, 0);
                }
                // classes 4 q4 .. + 3 of row 16 j + c16: 16 rows x 64 contiguous bytes per wave-instruction
                *reinterpret_cast<f32x4*>(sl + j * 256) = d;
            }
        }
    }

    ET* tp1 = epi.t1_ptr();
    ET* tp2 = epi.t2_ptr();
    const bool any_t = (tp1 != nullptr) || (tp2 != nullptr);
    for_quarters([&](auto hh_c, auto jj_c, int qq) {
        constexpr int HH = decltype(hh_c)::value, JJ = decltype(jj_c)::value;
        const int wm0 = m0 + wr * (HM ? 64 : 128) + HH * 64, wn0 = n0 + wc * 64 + JJ * 32;
        (void)qq;
        if constexpr (SPLIT) { if (wm0 > epi.m_dim()) return; }   // a quarter wholly past the ones row: nothing to store
        f32x4 r1[8], av[8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jl = 0; jl < 2; ++jl) av[2 * i + jl] = acc[4 * HH + i][2 * JJ + jl];
        f32x4 r2[8];
        relayout(av, r1);
#pragma unroll
        for (int p = 0; p < 8; ++p) r2[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the functor's FAST protocol (epilogues.h): every tile of this kernel is interior and vector-aligned (the
        // launcher checks), so the epilogue's global loads are issued a batch at a time, ahead of their first use
        constexpr int FB = Epi::FAST_BATCH;
        if constexpr (SPLIT) {
            // ragged in M (gemm_v2.h's EDGE_FAST form): lanes whose quad lies past the last row are off, the lane ON row
            // m_dim() hands its first value (the ones row's sum) to edge_row. m_dim() % 4 == 0: quads never straddle.
            const int lm = wm0 + 4 * c16;
            const bool lane_in = lm + 4 <= epi.m_dim(), lane_edge = lm == epi.m_dim();
#pragma unroll
            for (int p0 = 0; p0 < 8; p0 += FB) {
                if (lane_in) {
                    typename Epi::Pre pre[FB];
                    if (epi.fast_pre_needed()) {              // (the d/dmeans half with kl_scale = 0 reads nothing: workgroup-uniform)
#pragma unroll
                        for (int b = 0; b < FB; ++b) pre[b] = epi.load_fast(wm0, wn0 + 4 * (p0 + b), eln);
                    } else {
#pragma unroll
                        for (int b = 0; b < FB; ++b) pre[b] = typename Epi::Pre{};
                    }
#pragma unroll
                    for (int b = 0; b < FB; ++b) {
                        float t1[4], t2[4];
                        epi.apply_fast(wm0, wn0 + 4 * (p0 + b), eln, r1[p0 + b], f32x4{0.f, 0.f, 0.f, 0.f}, pre[b], t1, t2);
                    }
                }
                if (lane_edge) {
#pragma unroll
                    for (int b = 0; b < FB; ++b) epi.edge_row(wn0 + 4 * (p0 + b) + q4, r1[p0 + b][0]);
                }
            }
            return;
        }
#pragma unroll
        for (int p0 = 0; p0 < 8; p0 += FB) {
            typename Epi::Pre pre[FB];
            // (a functor whose folded epilogue reads nothing -- accGradParameters with the KL gradient left to the update sweep,
            // kl_scale = 0: the default of the bf16 configuration -- skips the batch as a whole: one wave-uniform branch around
            // all of its loads, not one around each)
            if (!DUAL || epi.folded_pre_needed()) {
#pragma unroll
                for (int b = 0; b < FB; ++b)
                    pre[b] = DUAL ? epi.load_folded(wm0, wn0 + 4 * (p0 + b), eln) : epi.load_fast(wm0, wn0 + 4 * (p0 + b), eln);
            } else {
#pragma unroll
                for (int b = 0; b < FB; ++b) pre[b] = typename Epi::Pre{};
            }
#pragma unroll
            for (int b = 0; b < FB; ++b) {
                const int p = p0 + b;
                float t1[4], t2[4];
                if (DUAL) epi.apply_folded(wm0, wn0 + 4 * p, eln, r1[p], r2[p], pre[b], t1, t2);
                else epi.apply_fast(wm0, wn0 + 4 * p, eln, r1[p], f32x4{0.f, 0.f, 0.f, 0.f}, pre[b], t1, t2);
                if (any_t) {
                    // 8-element chunks XOR-swizzled by the row's c16 & 3 (rows 4 apart would otherwise share banks)
                    const int nl = 4 * p + q4;
                    const int col = ((nl >> 3) ^ (c16 & 3)) * 8 + (nl & 7);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        tl1[(4 * c16 + j) * TQ + col] = Elt<ET>::to(t1[j]);
                        tl2[(4 * c16 + j) * TQ + col] = Elt<ET>::to(t2[j]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);        // one position at a time: interleaving them all spills
            }
        }
        if (any_t) {
            // rows of the transposed outputs: lane (row = lane >> 2 (+16 per pass), 8 consecutive n = 8 (lane & 3) ..)
            const int64_t ldt = epi.t_ld();               // a multiple of 8, 16-byte aligned rows (the launcher checks)
            const unsigned lo = (unsigned)((lane >> 2) * (int)ldt + 8 * (lane & 3));      // scalar base + lane offset
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const int ml = 16 * pp + (lane >> 2);
                const int64_t ub = (int64_t)(wm0 + 16 * pp) * ldt + wn0;
                if (tp1)
                    *reinterpret_cast<bf16x8*>(tp1 + ub + lo) =
                        *reinterpret_cast<const bf16x8*>(tl1 + ml * TQ + 8 * ((lane & 3) ^ ((ml >> 2) & 3)));
                if (tp2)
                    *reinterpret_cast<bf16x8*>(tp2 + ub + lo) =
                        *reinterpret_cast<const bf16x8*>(tl2 + ml * TQ + 8 * ((lane & 3) ^ ((ml >> 2) & 3)));
            }
        }
    });
    V3_ST(4);
}

// The kernel has no ragged-edge or general-output path: whole 256 x 256 tiles, 32-bit element offsets, the functor's
// fast protocol and 16-byte rows of the transposed outputs. Everything else stays with gemm_v2.h.
// (K-major operands: lda / ldb are the pitches of the K rows; K must be whole 64-row steps -- there is no padding row to
// clamp to -- and the row pitch a multiple of 8 elements for the 16-byte DMA chunks.)
template <class Epi>
static inline bool gemm_v3_possible(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, bool ak, bool bk, const Epi& epi) {
    Epi e = epi;
    const bool t_ok = (!e.t1_ptr() && !e.t2_ptr()) ||
                      ((e.t_ld() % 8 == 0) && ((((uintptr_t)e.t1_ptr() | (uintptr_t)e.t2_ptr()) & 15u) == 0));
    // operand extents below 2^30 elements: the LDS-DMA goes out in buffer form with 32-bit BYTE offsets under a 2 GiB range
    const bool a_ok = ak ? (K % V2_BK == 0 && lda % 8 == 0 && lda >= M && K * lda < (1ll << 30)) : M * lda < (1ll << 30);
    const bool b_ok = bk ? (K % V2_BK == 0 && ldb % 8 == 0 && ldb >= N && K * ldb < (1ll << 30)) : N * ldb < (1ll << 30);
    return (M % V3_BM == 0) && (N % V3_BN == 0) && a_ok && b_ok && epi.fast_ok() && epi.v3_ok() && t_ok && M * e.t_ld() < (1ll << 31);
}
// Worth it when K is long enough to amortise the second pipeline fill and the fold between the passes (measured against
// gemm_v2's dual tile on 4096 x 4096 outputs: K = 4096 forward 212 vs 252 us, K = 784 forward 93 vs 99 us) and the
// tile count fills whole rounds of CUs better than the half-size dual tiles would: a 256 x 256 two-pass tile costs about
// 1.85x a 256 x 128 dual tile (235 vs 127 us at K = 4096), so compare rounds x cost on the device's CUs.
// shape part of the choice (also answers vbnn_kmajor_supported: the host decides from it whether to keep transposed copies)
static inline bool gemm_v3_shape_ok(int64_t M, int64_t N, int64_t K) {
    if (M % V3_BM || N % V3_BN || K < g_v3_min_k) return false;
    const int64_t cus = vbnn_cu_count();
    const int64_t t3 = (M / V3_BM) * (N / V3_BN), t2 = 2 * t3;
    if (4 * t3 < 3 * cus) return false;
    const int64_t r3 = (t3 + cus - 1) / cus, r2 = (t2 + cus - 1) / cus;
    return 185 * r3 <= 100 * r2;
}
template <class Epi>
static inline bool gemm_v3_eligible(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, bool ak, bool bk, const Epi& epi) {
    if (!gemm_v3_possible(M, N, K, lda, ldb, ak, bk, epi) || K < g_v3_min_k) return false;
    const int64_t cus = vbnn_cu_count();
    const int64_t t3 = (M / V3_BM) * (N / V3_BN), t2 = 2 * t3;
    if (4 * t3 < 3 * cus) return false;                       // fewer than 3/4 of the CUs busy: the half-size tiles win
    const int64_t r3 = (t3 + cus - 1) / cus, r2 = (t2 + cus - 1) / cus;
    return 185 * r3 <= 100 * r2;
}

template <typename T, bool DUAL, bool AK, bool BK, class Epi>
static int launch_gemm_v3(vbnn_ctx* ctx, const T* A, const T* A2, int64_t lda, const T* B, const T* B2, int64_t ldb, int M, int N,
                          int K, const Epi& epi) {
    if constexpr (sizeof(T) != 2) {
        vbnn_set_error("gemm_v3 is bf16 only");
        return VBNN_ERR_UNSUPPORTED;
    } else {
        if ((((uintptr_t)A | (uintptr_t)B | (uintptr_t)A2 | (uintptr_t)B2) & 15u) != 0) {
            vbnn_set_error("gemm_v3 operands must be 16-byte aligned");
            return VBNN_ERR_INVALID;
        }
        const int nk = (K + V2_BK - 1) / V2_BK;
        if ((!AK && lda < (int64_t)nk * V2_BK) || (!BK && ldb < (int64_t)nk * V2_BK)) {
            vbnn_set_error("packed leading dimension too small for K=%d", K);
            return VBNN_ERR_INVALID;
        }
        if (!gemm_v3_possible(M, N, K, lda, ldb, AK, BK, epi)) { vbnn_set_error("gemm_v3: shape / outputs outside its fast path"); return VBNN_ERR_UNSUPPORTED; }
        const int tiles_m = (M + V3_BM - 1) / V3_BM, tiles_n = (N + V3_BN - 1) / V3_BN;
        const void* kern = (const void*)gemm_nt_v3<DUAL, AK, BK, Epi>;
        static vbnn_per_device_flag configured_on;           // per instantiation and device
        bool& configured = configured_on[ctx->device];
        if (!configured) {
            hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS);
            if (e != hipSuccess) { vbnn_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return VBNN_ERR_HIP; }
            configured = true;
        }
        const bf16_t* a = (const bf16_t*)A; const bf16_t* a2 = (const bf16_t*)A2;
        const bf16_t* b = (const bf16_t*)B; const bf16_t* b2 = (const bf16_t*)B2;
        int nk_ = nk, M_ = M, N_ = N, tm_ = tiles_m, tn_ = tiles_n;
        int64_t lda_ = lda, ldb_ = ldb;
        Epi epi_ = epi;
        void* args[] = {&a, &a2, &lda_, &b, &b2, &ldb_, &M_, &N_, &nk_, &tm_, &tn_, &epi_};
        hipError_t e = hipLaunchKernel(kern, dim3(tiles_m * tiles_n), dim3(512), args, V3_LDS, ctx->stream);
        if (e != hipSuccess) { vbnn_set_error("launch of gemm_nt_v3 failed: %s", hipGetErrorString(e)); return VBNN_ERR_HIP; }
        return vbnn_check_launch("gemm_nt_v3");
    }
}

// ---- the pair-split HALF-HEIGHT launch (K-major operands only: accGradParameters reading x and g as the other GEMMs hold them)
inline int g_v3_split = -1;         // -1 by shape, 0 never, 1 whenever possible (vbnn_debug_set key 8)
// shape part: M rows of output (the ones row included) on N columns, K deep. Wanted when the pair split over 256 x 128 tiles
// (gemm_v2.h) would leave half the CUs idle and this form's 128-row tiles x 2 make ONE round that fills at least 5/8 of them.
static inline bool gemm_v3_split_shape_ok(int64_t M, int64_t N, int64_t K) {
    if (g_v3_split == 0 || N % V3_BN || K % V2_BK) return false;       // (whole K steps; r02-r03 asked for an even number of them: two K halves)
    if (g_v3_split == 1) return true;
    const int64_t tiles = ((M + V3_BM - 1) / V3_BM) * (N / V3_BN);
    const int64_t blocks = ((M + V3_BM / 2 - 1) / (V3_BM / 2)) * (N / V3_BN) * 2;
    const int64_t cus = vbnn_cu_count();
    return tiles * 16 <= 5 * cus && tiles * 16 >= 3 * cus && K >= 2048 && blocks <= cus && blocks * 8 >= 5 * cus;
}
// the pitch of the K rows of A this launch needs: whole 256-column tiles (the columns past M are zero padding)
static inline int64_t gemm_v3_split_lda(int64_t M) { return (M + V3_BM - 1) / V3_BM * V3_BM; }

template <typename T, class Epi>
static int launch_gemm_v3_split(vbnn_ctx* ctx, const T* A, const T* A2, int64_t lda, const T* B, const T* B2, int64_t ldb, int M, int N,
                                int K, const Epi& epi) {
    if constexpr (sizeof(T) != 2 || !Epi::SPLITTABLE || !Epi::EDGE_FAST) {
        return VBNN_ERR_UNSUPPORTED;
    } else {
        Epi e0 = epi;
        if (!A || !A2 || !B || !B2 || !gemm_v3_split_shape_ok(M, N, K) || !epi.fast_ok() || e0.t1_ptr() || e0.t2_ptr() ||
            (((uintptr_t)A | (uintptr_t)B | (uintptr_t)A2 | (uintptr_t)B2) & 15u) != 0 || lda % 8 || ldb % 8 ||
            lda < gemm_v3_split_lda(M) || ldb < N || (int64_t)K * lda >= (1ll << 30) || (int64_t)K * ldb >= (1ll << 30))
            return VBNN_ERR_UNSUPPORTED;
        const int tiles_n = N / V3_BN;
        const int tiles_m = (M + V3_BM / 2 - 1) / (V3_BM / 2);
        const void* kern = (const void*)gemm_nt_v3<false, true, true, Epi, true, true>;
        static vbnn_per_device_flag configured_on;           // per instantiation and device
        bool& configured = configured_on[ctx->device];
        if (!configured) {
            hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS);
            if (e != hipSuccess) { vbnn_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return VBNN_ERR_HIP; }
            configured = true;
        }
        const bf16_t* a = (const bf16_t*)A; const bf16_t* a2 = (const bf16_t*)A2;
        const bf16_t* b = (const bf16_t*)B; const bf16_t* b2 = (const bf16_t*)B2;
        int nk_ = K / V2_BK, M_ = M, N_ = N, tm_ = tiles_m, tn_ = tiles_n;
        int64_t lda_ = lda, ldb_ = ldb;
        Epi epi_ = epi;
        void* args[] = {&a, &a2, &lda_, &b, &b2, &ldb_, &M_, &N_, &nk_, &tm_, &tn_, &epi_};
        hipError_t e = hipLaunchKernel(kern, dim3(tiles_m * tiles_n * 2), dim3(512), args, V3_LDS, ctx->stream);
        if (e != hipSuccess) { vbnn_set_error("launch of gemm_nt_v3 (pair split) failed: %s", hipGetErrorString(e)); return VBNN_ERR_HIP; }
        return vbnn_check_launch("gemm_nt_v3 pair split");
    }
}
