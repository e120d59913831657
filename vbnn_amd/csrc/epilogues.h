// epilogues.h -- fused epilogues of the three VBLinear GEMM families.
//
// Every GEMM kernel in this library computes C[m][n] = sum_k A[m][k] * Bt[n][k] (and, for
// the local-reparameterisation pair, a second accumulator from A2/Bt2). The "weights / feature"
// side is always operand A, so M is the CONTIGUOUS direction of every primary output tensor
// (FWD: output units of a minibatch row; DX: input units of a row; DW: input units i of output
// unit o). An epilogue functor is handed four consecutive m of one n:
//
//     epi.apply<STORE_T>(m, n, acc1, acc2, t1, t2)
//
// and (a) computes and stores the primary outputs at [n][m .. m+3] (one 8- or 16-byte store),
// (b) produces the four values of each TRANSPOSED output (t1, t2) -- stored by the functor itself
// (STORE_T = true: 2-byte stores, the general kernel) or handed back to the kernel, which transposes
// them through LDS and writes whole rows (STORE_T = false: the pipelined kernel).
// Four consecutive m are also exactly one Philox block (4 normals) of the RNG contract.
//
// The FAST protocol (gemm_v3.h): `apply` guards every access (ragged edges, optional outputs), and its loads sit in
// lane-dependent control flow, so a kernel that calls it 8 or 16 times in a row pays the load latency 8 or 16
// times in sequence. When fast_ok() holds (the fused bf16 configuration: vector-aligned operands, only the packed
// outputs) and a tile lies fully inside the matrix, the kernel instead calls
//     Lane ln = epi.lane_init(nl, ml)                once: the lane's position (n = un + nl, m = um + ml) as 32-bit
//                                                    element offsets into each tensor it touches,
//     Pre pre[k] = epi.load_fast(um, un_k, ln)       for a batch of FAST_BATCH positions -- unconditional vector loads,
//     epi.apply_fast(um, un_k, ln, a1, a2, pre[k], t1, t2)                  all in flight before the first use.
// (um, un) are WAVE-UNIFORM: every address is (scalar base) + (one per-lane 32-bit offset), so the global accesses
// take the SGPR-base form and a whole unrolled tile needs a handful of address registers instead of a 64-bit pair per
// access (which spilled). Same arithmetic, same stores.
//
// The FOLD protocol (gemm_v3.h, the GEMM pair as two passes over ONE accumulator): the pair's second GEMM (acc2) runs
// first; between the passes each accumulator quad, still in the MFMA register layout, goes through
//     FPre fp = epi.fold_load(um, un, ln);   acc = epi.fold(um, un, ln, acc, fp);
// which turns acc2 into the term the first GEMM is then accumulated ON TOP of (FWD: b + sqrt(v) z, storing r;
// DX: 2 x . acc2; DW: stores the finished d/dlvars and returns 0), and the final epilogue calls
//     Pre pre = epi.load_folded(um, un, ln);   epi.apply_folded(um, un, ln, acc, pre, t1, t2)
// on the single accumulator (`side` = 0). Nothing is parked between the passes. The sums are the same up to fp32
// association (the term enters the accumulation chain first instead of last).
#pragma once
#include "common.h"

template <typename T> struct V4;                 // four consecutive elements as one register-resident vector
template <> struct V4<float> { typedef f32x4 type; };
template <> struct V4<bf16_t> { typedef bf16x4 type; };
// four values to [base + off .. + 3] as ONE vector store (base wave-uniform, off the lane's element offset)
template <typename T>
__device__ __forceinline__ void store4_out(T* base, unsigned off, float a, float b, float c, float d) {
    vbnn_store_out(base, off, typename V4<T>::type{Elt<T>::to(a), Elt<T>::to(b), Elt<T>::to(c), Elt<T>::to(d)});
}

// ---- FWD: M = output units o, N = minibatch rows n ---------------------------------------------
// WN/MAP: y = acc1 + b                      (inherited nn.Linear:updateOutput, VBLinear.lua:7)
// LRT   : y = acc1 + b + sqrt(acc2) * z,  r = z / (2 sqrt(acc2))
template <typename T>
struct EpiFwd {
    typedef T elem_t;
    const float* bias;
    int noise;
    uint64_t seed; uint32_t layer, draw; int64_t row0;
    const uint32_t* draw_dev = nullptr;              // optional device-resident draw counter (vbnn_fwd_args.draw_dev): added to `draw`
    int rpd;                                         // > 0: stacked draws, vbnn_fwd_args.rows_per_draw
    float* y; int64_t ld_y; int y_vec;
    float* r; T* r_t; int64_t ld_r; int r_vec;       // r as f32 (module path) or as T (fused path), never both
    int relu;
    T* h; T* h2; int64_t ld_h;
    T* hT; T* h2T; int64_t ld_hT;
    int O, N;
    // optional (gemm_v3.h only, the LAST VB layer below the fused classifier head, mlp.lua:29): the final nn.Linear's logits are
    // formed HERE, from the output tile as it stands in the accumulators -- each wave's 128 m x 64 n of relu(y), rounded to the
    // operand type exactly as `h` is stored, times the matching 128 columns of the packed final weight (C <= 16 rows): one
    // [16 classes] x [64 n] partial per wave into head_slots[2 tile_m + wave row][N][16] (fp32, a FIXED slot: no atomics, the
    // head sums the slots in order). The head's forward then never re-reads h (33 MB at 4096 x 4096).
    const T* head_w3 = nullptr; int64_t head_ld_w = 0; int head_C = 0; float* head_slots = nullptr;
    static constexpr bool HEAD = sizeof(T) == 2;

    static constexpr bool SPLITTABLE = false;     // every output needs both GEMMs of the pair
    static constexpr bool EDGE_FAST = false;      // ragged wave tiles take the guarded form
    // kernels that support the device-resident counter call this once at entry (a wave-uniform scalar load)
    __device__ __forceinline__ void bind_draw() { if (draw_dev) draw += *draw_dev; }
    __host__ __device__ __forceinline__ bool has_draw_dev() const { return draw_dev != nullptr; }
    __device__ __forceinline__ void edge_row(int, float) const {}
    __device__ __forceinline__ void set_part(int) {}
    __host__ __device__ __forceinline__ T* t1_ptr() const { return hT; }
    __host__ __device__ __forceinline__ T* t2_ptr() const { return h2T; }
    __host__ __device__ __forceinline__ int64_t t_ld() const { return ld_hT; }
    __device__ __forceinline__ int m_dim() const { return O; }
    __device__ __forceinline__ int n_dim() const { return N; }
    // the two-pass kernel's fold addresses the noise by launch-wide (draw, row0): stacked draws stay with the other kernels
    __host__ __device__ __forceinline__ bool v3_ok() const { return rpd == 0; }
    // the four normals of a quad: the contract's bit-exact form on the fp32 path, its hardware-transcendental form on the bf16
    // path (common.h, vbnn_normal4_hw; -DVBNN_BF16_EXACT_NORMALS: the exact form there too, A/B)
    static __device__ __forceinline__ vbnn_f32x4 normal4(uint64_t seed_, uint32_t layer_, uint32_t draw_, uint32_t row, uint32_t quad) {
#ifndef VBNN_BF16_EXACT_NORMALS
        if constexpr (sizeof(T) == 2) return vbnn_normal4_hw(seed_, VBNN_STREAM_ZETA, layer_, draw_, row, quad);
#endif
        return vbnn_normal4(seed_, VBNN_STREAM_ZETA, layer_, draw_, row, quad);
    }
    // the four normals of output units 4 q .. 4 q + 3 of operand row n: (draw, minibatch row) of that row
    __device__ __forceinline__ vbnn_f32x4 zeta4(int n, uint32_t quad) const {
        uint32_t nn = (uint32_t)n, d = draw;
        if (rpd > 0) { const uint32_t k = nn / (uint32_t)rpd; d += k; nn -= k * (uint32_t)rpd; }
        return normal4(seed, layer, d, (uint32_t)(row0 + nn), quad);
    }

    template <bool STORE_T>
    __device__ __forceinline__ void apply(int m, int n, f32x4 a1, f32x4 a2, float (&t1)[4], float (&t2)[4]) const {
        const int valid = min(4, O - m);
#pragma unroll
        for (int j = 0; j < 4; ++j) { t1[j] = 0.f; t2[j] = 0.f; }
        if (valid <= 0 || n >= N) return;
        float yv[4], rv[4], bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) load4<float>(bias + m, bv, valid, (m & 3) == 0 && (((uintptr_t)bias & 15u) == 0));
        vbnn_f32x4 z;
        if (noise) z = zeta4(n, (uint32_t)(m >> 2));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float mb = a1[j] + bv[j];
            if (noise) {
                // v_rsq_f32 (1 ulp) instead of a correctly rounded sqrt + divide: this is outside the RNG
                // contract, and 2 ulp on sqrt(v) is far inside the fp32 parity tolerance. v == 0 (an all-zero
                // input row) must give sd = 0 and r = 0, not inf * 0.
                const bool pos = a2[j] > 0.f;
                const float rs = __builtin_amdgcn_rsqf(a2[j]);
                const float sd = pos ? a2[j] * rs : 0.f;
                yv[j] = fmaf(sd, z.v[j], mb);
                rv[j] = pos ? 0.5f * z.v[j] * rs : 0.f;
            } else {
                yv[j] = mb;
                rv[j] = 0.f;
            }
        }
        if (y) store4<float>(y + (int64_t)n * ld_y + m, yv[0], yv[1], yv[2], yv[3], valid, y_vec);
        if (r) store4<float>(r + (int64_t)n * ld_r + m, rv[0], rv[1], rv[2], rv[3], valid, r_vec);
        if (r_t) store4<T>(r_t + (int64_t)n * ld_r + m, rv[0], rv[1], rv[2], rv[3], valid, r_vec);
        if (h || hT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t1[j] = relu ? fmaxf(yv[j], 0.f) : yv[j];
                // square the value the consumer will actually read (the rounded one)
                const float hr = Elt<T>::from(Elt<T>::to(t1[j]));
                t2[j] = hr * hr;
            }
            if (h) {
                store4<T>(h + (int64_t)n * ld_h + m, t1[0], t1[1], t1[2], t1[3], valid, true);
                if (h2) store4<T>(h2 + (int64_t)n * ld_h + m, t2[0], t2[1], t2[2], t2[3], valid, true);
            }
            if (STORE_T && hT) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < valid) {
                        hT[(int64_t)(m + j) * ld_hT + n] = Elt<T>::to(t1[j]);
                        if (h2T) h2T[(int64_t)(m + j) * ld_hT + n] = Elt<T>::to(t2[j]);
                    }
            }
        }
    }
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a1, f32x4 a2) const {
        float t1[4], t2[4];
        apply<true>(m, n, a1, a2, t1, t2);
    }

    // ---- fast protocol
    static constexpr int FAST_BATCH = 8;
    static constexpr int FAST_BATCH_V2 = 4;
    struct Pre { f32x4 b; };
    struct Lane { int nl, ml; unsigned oh, orr; };
    __host__ __device__ __forceinline__ bool fast_ok() const {
        return !y && !r && h && (ld_h % 4 == 0) && (!r_t || (r_vec && ld_r % 8 == 0 && (((uintptr_t)r_t & 15u) == 0))) &&
               (!bias || (((uintptr_t)bias & 15u) == 0)) && (int64_t)N * ld_h < (1ll << 31) && (int64_t)N * ld_r < (1ll << 31);
    }
    __device__ __forceinline__ Lane lane_init(int nl, int ml) const {
        return Lane{nl, ml, (unsigned)(nl * (int)ld_h + ml), (unsigned)(nl * (int)ld_r + ml)};
    }
    __device__ __forceinline__ Pre load_fast(int um, int un, const Lane& ln) const {
        (void)un;
        Pre p;
        p.b = bias ? *reinterpret_cast<const f32x4*>(bias + um + ln.ml) : f32x4{0.f, 0.f, 0.f, 0.f};
        return p;
    }
    // the position's four normals, separately: they depend on indices alone, so a kernel with nothing to issue while its
    // first operand loads are in flight (gemm_v0.h) draws them THEN and passes them to apply_fast_z
    struct Noise { vbnn_f32x4 z; };
    __device__ __forceinline__ Noise draw_fast(int um, int un, const Lane& ln) const {
        Noise q = {};
        if (noise) q.z = zeta4(un + ln.nl, (uint32_t)((um + ln.ml) >> 2));
        return q;
    }
    __device__ __forceinline__ void apply_fast(int um, int un, const Lane& ln, f32x4 a1, f32x4 a2, const Pre& pre, float (&t1)[4],
                                               float (&t2)[4]) const {
        apply_fast_z(um, un, ln, a1, a2, pre, draw_fast(um, un, ln), t1, t2);
    }
    __device__ __forceinline__ void apply_fast_z(int um, int un, const Lane& ln, f32x4 a1, f32x4 a2, const Pre& pre, const Noise& q,
                                                 float (&t1)[4], float (&t2)[4]) const {
        float yv[4], rv[4];
        const vbnn_f32x4 z = q.z;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float mb = a1[j] + pre.b[j];
            if (noise) {
                const bool pos = a2[j] > 0.f;
                const float rs = __builtin_amdgcn_rsqf(a2[j]);
                const float sd = pos ? a2[j] * rs : 0.f;
                yv[j] = fmaf(sd, z.v[j], mb);
                rv[j] = pos ? 0.5f * z.v[j] * rs : 0.f;
            } else {
                yv[j] = mb;
                rv[j] = 0.f;
            }
        }
        if (r_t) store4_out<T>(r_t + ((int64_t)un * ld_r + um), ln.orr, rv[0], rv[1], rv[2], rv[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t1[j] = relu ? fmaxf(yv[j], 0.f) : yv[j];
            const float hr = Elt<T>::from(Elt<T>::to(t1[j]));
            t2[j] = hr * hr;
        }
        const int64_t ub = (int64_t)un * ld_h + um;
        store4_out<T>(h + ub, ln.oh, t1[0], t1[1], t1[2], t1[3]);
        if (h2) store4_out<T>(h2 + ub, ln.oh, t2[0], t2[1], t2[2], t2[3]);
    }

    // ---- fold protocol (LRT only: acc2 = v)
    // the noise term b + sqrt(v) z (and r) is formed between the passes
    static constexpr int FOLD_BATCH = 1;          // m-blocks per batch of fold loads (only the bias here)
    static constexpr int FOLD_SERIAL = 1;         // a Philox block per quad: at most this many in flight (0 = no limit; 2 spills 24 registers and is no faster)
    // FOLD_STAGE: the fold's own output (r, N x O in the operand type) leaves through a per-wave LDS tile as whole
    // 128-byte row segments (gemm_v3.h) instead of one 8-byte store per lane straight from the MFMA layout -- 16 rows x
    // 32 bytes per wave-instruction, which ran at 1.9 TB/s (lab: 18 of the fold's 37 us at 4096 x 4096 outputs).
    static constexpr int FOLD_STAGE = 1;
    typedef T fold_st_t;
    __host__ __device__ __forceinline__ T* fold_st_ptr() const { return r_t; }
    __host__ __device__ __forceinline__ int64_t fold_st_ld() const { return ld_r; }
    __host__ __device__ __forceinline__ const float* fold_bias_ptr() const { return bias; }   // per-m addend of the fold, or NULL
    __device__ __forceinline__ bool fold_st_stream() const { return h2 != nullptr; }          // r of a layer with a VB layer above it: read half a step later
    __device__ __forceinline__ f32x4 fold_s(int um, int un, const Lane& ln, f32x4 v, const f32x4& b4, float (&rv)[4]) const {
        const vbnn_f32x4 z = normal4(seed, layer, draw, (uint32_t)(row0 + un + ln.nl), (uint32_t)((um + ln.ml) >> 2));
        f32x4 out;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool pos = v[j] > 0.f;
            const float rs = __builtin_amdgcn_rsqf(v[j]);
            const float sd = pos ? v[j] * rs : 0.f;
            out[j] = fmaf(sd, z.v[j], b4[j]);
            rv[j] = pos ? 0.5f * z.v[j] * rs : 0.f;
        }
        return out;
    }
    struct FPre { f32x4 b; };
    __device__ __forceinline__ FPre fold_load(int um, int un, const Lane& ln) const {
        (void)un;
        FPre p;
        p.b = bias ? *reinterpret_cast<const f32x4*>(bias + um + ln.ml) : f32x4{0.f, 0.f, 0.f, 0.f};
        return p;
    }
    __device__ __forceinline__ f32x4 fold(int um, int un, const Lane& ln, f32x4 v, const FPre& fp) const {
        const vbnn_f32x4 z = normal4(seed, layer, draw, (uint32_t)(row0 + un + ln.nl), (uint32_t)((um + ln.ml) >> 2));
        f32x4 out;
        float rv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool pos = v[j] > 0.f;
            const float rs = __builtin_amdgcn_rsqf(v[j]);
            const float sd = pos ? v[j] * rs : 0.f;
            out[j] = fmaf(sd, z.v[j], fp.b[j]);
            rv[j] = pos ? 0.5f * z.v[j] * rs : 0.f;
        }
        if (r_t) store4_out<T>(r_t + ((int64_t)un * ld_r + um), ln.orr, rv[0], rv[1], rv[2], rv[3]);
        return out;
    }
    __device__ __forceinline__ bool folded_pre_needed() const { return false; }
    __device__ __forceinline__ Pre load_folded(int, int, const Lane&) const { return Pre{f32x4{0.f, 0.f, 0.f, 0.f}}; }
    __device__ __forceinline__ void apply_folded(int um, int un, const Lane& ln, f32x4 a, f32x4, const Pre&, float (&t1)[4],
                                                 float (&t2)[4]) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t1[j] = relu ? fmaxf(a[j], 0.f) : a[j];
            const float hr = Elt<T>::from(Elt<T>::to(t1[j]));
            t2[j] = hr * hr;
        }
        const int64_t ub = (int64_t)un * ld_h + um;
        store4_out<T>(h + ub, ln.oh, t1[0], t1[1], t1[2], t1[3]);
        if (h2) store4_out<T>(h2 + ub, ln.oh, t2[0], t2[1], t2[2], t2[3]);
    }
};

// ---- DX: M = input units i, N = minibatch rows n ------------------------------------------------
// WN : gx = acc1 = g w                         (inherited nn.Linear:updateGradInput)
// LRT: gx = acc1 + 2 x . acc2 = g mu + 2 x . (gv sigma^2)
// optional hand-off to the previous VB layer through the ReLU between them (mlp.lua:19,27).
template <typename T>
struct EpiDx {
    typedef T elem_t;
    int dual;
    const T* x; int64_t ld_x;
    float* gx; int64_t ld_gx; int gx_vec;
    int relu_mask;
    const float* r_prev; const T* r_prev_t; int64_t ld_r_prev; int r_vec;
    T* g_prev; T* gv_prev; int64_t ld_gp;
    T* gT_prev; T* gvT_prev; int64_t ld_gpT;
    int I, N;

    static constexpr bool SPLITTABLE = false;
    static constexpr bool EDGE_FAST = false;
    static constexpr bool HEAD = false;
    __device__ __forceinline__ void bind_draw() {}                 // (no noise in gradInput: r comes from the forward)
    __host__ __device__ __forceinline__ bool has_draw_dev() const { return false; }
    __device__ __forceinline__ void edge_row(int, float) const {}
    __device__ __forceinline__ void set_part(int) {}
    __host__ __device__ __forceinline__ T* t1_ptr() const { return gT_prev; }
    __host__ __device__ __forceinline__ T* t2_ptr() const { return gvT_prev; }
    __host__ __device__ __forceinline__ int64_t t_ld() const { return ld_gpT; }
    __device__ __forceinline__ int m_dim() const { return I; }
    __device__ __forceinline__ int n_dim() const { return N; }
    __host__ __device__ __forceinline__ bool v3_ok() const { return true; }

    template <bool STORE_T>
    __device__ __forceinline__ void apply(int m, int n, f32x4 a1, f32x4 a2, float (&t1)[4], float (&t2)[4]) const {
        const int valid = min(4, I - m);
#pragma unroll
        for (int j = 0; j < 4; ++j) { t1[j] = 0.f; t2[j] = 0.f; }
        if (valid <= 0 || n >= N) return;
        float xv[4] = {0.f, 0.f, 0.f, 0.f};
        if (x) load4<T>(x + (int64_t)n * ld_x + m, xv, valid, (ld_x & 3) == 0);
        float gv4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) gv4[j] = dual ? fmaf(2.0f * xv[j], a2[j], a1[j]) : a1[j];
        if (gx) store4<float>(gx + (int64_t)n * ld_gx + m, gv4[0], gv4[1], gv4[2], gv4[3], valid, gx_vec);
        if (g_prev || gT_prev) {
            float rp[4] = {0.f, 0.f, 0.f, 0.f};
            if (r_prev) load4<float>(r_prev + (int64_t)n * ld_r_prev + m, rp, valid, r_vec);
            if (r_prev_t) load4<T>(r_prev_t + (int64_t)n * ld_r_prev + m, rp, valid, r_vec);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t1[j] = (relu_mask && !(xv[j] > 0.f)) ? 0.f : gv4[j];
                t2[j] = t1[j] * rp[j];
            }
            if (g_prev) store4<T>(g_prev + (int64_t)n * ld_gp + m, t1[0], t1[1], t1[2], t1[3], valid, true);
            if (gv_prev) store4<T>(gv_prev + (int64_t)n * ld_gp + m, t2[0], t2[1], t2[2], t2[3], valid, true);
            if (STORE_T) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < valid) {
                        if (gT_prev) gT_prev[(int64_t)(m + j) * ld_gpT + n] = Elt<T>::to(t1[j]);
                        if (gvT_prev) gvT_prev[(int64_t)(m + j) * ld_gpT + n] = Elt<T>::to(t2[j]);
                    }
            }
        }
    }
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a1, f32x4 a2) const {
        float t1[4], t2[4];
        apply<true>(m, n, a1, a2, t1, t2);
    }

    // ---- fast protocol: the hand-off form (packed g_prev / gv_prev, packed r) of the fused engine
    static constexpr int FAST_BATCH = 8;
    static constexpr int FAST_BATCH_V2 = 4;
    struct Pre { typename V4<T>::type x, r; };
    struct Lane { unsigned ox, orp, ogp; };
    __host__ __device__ __forceinline__ bool fast_ok() const {
        return x && !gx && g_prev && !r_prev && (ld_x % 4 == 0) && (ld_gp % 4 == 0) &&
               (!r_prev_t || (r_vec && ld_r_prev % 4 == 0)) && (!gv_prev || r_prev_t) &&
               ((((uintptr_t)x | (uintptr_t)g_prev | (uintptr_t)gv_prev | (uintptr_t)r_prev_t) & 7u) == 0) &&
               (int64_t)N * ld_x < (1ll << 31) && (int64_t)N * ld_gp < (1ll << 31) && (int64_t)N * ld_r_prev < (1ll << 31);
    }
    __device__ __forceinline__ Lane lane_init(int nl, int ml) const {
        return Lane{(unsigned)(nl * (int)ld_x + ml), (unsigned)(nl * (int)ld_r_prev + ml), (unsigned)(nl * (int)ld_gp + ml)};
    }
    __device__ __forceinline__ Pre load_fast(int um, int un, const Lane& ln) const {
        Pre p;
        p.x = *reinterpret_cast<const typename V4<T>::type*>(x + ((int64_t)un * ld_x + um) + ln.ox);
        if (r_prev_t) p.r = vbnn_load_last_use(reinterpret_cast<const typename V4<T>::type*>(r_prev_t + ((int64_t)un * ld_r_prev + um) + ln.orp));   // r's only reader
        else p.r = typename V4<T>::type{};
        return p;
    }
    __device__ __forceinline__ void apply_fast(int um, int un, const Lane& ln, f32x4 a1, f32x4 a2, const Pre& pre, float (&t1)[4],
                                               float (&t2)[4]) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xv = Elt<T>::from(pre.x[j]);
            const float gv = dual ? fmaf(2.0f * xv, a2[j], a1[j]) : a1[j];
            t1[j] = (relu_mask && !(xv > 0.f)) ? 0.f : gv;
            t2[j] = t1[j] * Elt<T>::from(pre.r[j]);
        }
        const int64_t ub = (int64_t)un * ld_gp + um;
        store4_out<T>(g_prev + ub, ln.ogp, t1[0], t1[1], t1[2], t1[3]);
        if (gv_prev) store4_out<T>(gv_prev + ub, ln.ogp, t2[0], t2[1], t2[2], t2[3]);
    }

    // ---- fold protocol (LRT only: acc2 = gv sigma^2): the accumulator continues from 2 x . acc2
    static constexpr int FOLD_BATCH = 8;          // all 32 quads' x loads (64 registers) in flight together
    static constexpr int FOLD_SERIAL = 0;
    static constexpr int FOLD_STAGE = 0;
    struct FPre { typename V4<T>::type x; };
    __device__ __forceinline__ FPre fold_load(int um, int un, const Lane& ln) const {
        FPre p;
        p.x = *reinterpret_cast<const typename V4<T>::type*>(x + ((int64_t)un * ld_x + um) + ln.ox);
        return p;
    }
    // The ReLU mask of the module in between rides THROUGH pass 2 in the accumulator itself (r04): where the layer input is not
    // positive the fold leaves a quiet NaN instead of 2 x . acc2 (= 0 there anyway), the mean GEMM accumulates on top of it -- NaN + c
    // stays NaN, element by element -- and the final epilogue reads the mask back as (a != a): it no longer re-reads x (33.5 MB of an
    // HBM-bound store burst at 4096 x 4096; handing the mask over as 128 bits per lane had cost more than the read, r03). The values
    // are bitwise what they were. (A genuine NaN of an unmasked element -- a diverged run -- is zeroed in g_prev with it; the loss and
    // every other tensor still show it.)
    __device__ __forceinline__ f32x4 fold(int, int, const Lane&, f32x4 a2, const FPre& fp) const {
        f32x4 out;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xv = Elt<T>::from(fp.x[j]);
            out[j] = (relu_mask && !(xv > 0.f)) ? __builtin_nanf("") : 2.0f * xv * a2[j];
        }
        return out;
    }
    __device__ __forceinline__ bool folded_pre_needed() const { return r_prev_t != nullptr; }
    __device__ __forceinline__ Pre load_folded(int um, int un, const Lane& ln) const {       // (r alone: the mask is in the accumulator)
        Pre p;
        p.x = typename V4<T>::type{};
        if (r_prev_t) p.r = vbnn_load_last_use(reinterpret_cast<const typename V4<T>::type*>(r_prev_t + ((int64_t)un * ld_r_prev + um) + ln.orp));
        else p.r = typename V4<T>::type{};
        return p;
    }
    __device__ __forceinline__ void apply_folded(int um, int un, const Lane& ln, f32x4 a, f32x4, const Pre& pre, float (&t1)[4],
                                                 float (&t2)[4]) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t1[j] = (relu_mask && a[j] != a[j]) ? 0.f : a[j];
            t2[j] = t1[j] * Elt<T>::from(pre.r[j]);
        }
        const int64_t ub = (int64_t)un * ld_gp + um;
        store4_out<T>(g_prev + ub, ln.ogp, t1[0], t1[1], t1[2], t1[3]);
        if (gv_prev) store4_out<T>(gv_prev + ub, ln.ogp, t2[0], t2[1], t2[2], t2[3]);
    }
};

// ---- DW: M = input units i, N = output units o --------------------------------------------------
// acc1 = (g^T x)[o][i], acc2 = (gv^T x.x)[o][i]          (VBLinear.lua:112-118, one GEMM not two)
struct EpiDw {
    typedef bf16_t elem_t;   // no transposed outputs; only the type is needed
    int lrt;                 // 1: LRT (acc2 valid), 0: WN (e regenerated)
    float scale; int accumulate;
    float* gradWeight; float* gradSum; int vec;
    uint64_t seed; uint32_t layer, draw;
    const uint32_t* draw_dev = nullptr;   // optional device-resident draw counter (vbnn_dw_args.draw_dev; weight-noise e only)
    const float* lvars;
    float* grad_mu; float* grad_lv;
    const float* means; const double* stats; float B, S, kl_scale;
    float* gradBias;         // optional: the GEMM has one more A row (all ones) whose output row is the bias gradient
    const bf16_t* mu_s; const bf16_t* var_s; int ld_w;    // optional bf16 shadows of means / exp(lvars): read instead of them
    int I, O;
    // d/dmeans depends on the first GEMM of the pair only and d/dlvars on the second only, so a kernel may compute the
    // two GEMMs in DIFFERENT workgroups (gemm_v2.h, pair split): part 1 = this workgroup holds the first GEMM (in a1)
    // and writes only what depends on it; part 2 = it holds the SECOND GEMM (also handed over in a1) and writes only
    // what depends on that; 0 = both accumulators, as ever.
    int part = 0;
    static constexpr bool SPLITTABLE = true;
    static constexpr bool HEAD = false;
    __device__ __forceinline__ void set_part(int p) { part = p; }
    __device__ __forceinline__ void bind_draw() { if (draw_dev) draw += *draw_dev; }
    __host__ __device__ __forceinline__ bool has_draw_dev() const { return draw_dev != nullptr; }
    // A wave tile that straddles the last row I of the output (I % 4 == 0, so every lane's quad is wholly inside or
    // wholly outside) may still take the FAST protocol with the outside lanes switched off: its batched loads then
    // cost one latency per batch instead of one per position, on the wave that otherwise finishes the launch last.
    // The row right after the output, m = I, is the GEMM's row of ones (gradBias): edge_row takes its sums.
    static constexpr bool EDGE_FAST = true;
    __device__ __forceinline__ void edge_row(int n, float v) const {
        if (part != 2 && gradBias && n < O) gradBias[n] = scale * v;      // fast_ok() implies accumulate == 0
    }

    __host__ __device__ __forceinline__ bf16_t* t1_ptr() const { return nullptr; }
    __host__ __device__ __forceinline__ bf16_t* t2_ptr() const { return nullptr; }
    __host__ __device__ __forceinline__ int64_t t_ld() const { return 0; }
    __device__ __forceinline__ int m_dim() const { return I; }
    __device__ __forceinline__ int n_dim() const { return O; }
    __host__ __device__ __forceinline__ bool v3_ok() const { return true; }

    template <bool STORE_T>
    __device__ __forceinline__ void apply(int m, int n, f32x4 a1, f32x4 a2, float (&t1)[4], float (&t2)[4]) const {
        (void)t1; (void)t2;
        (*this)(m, n, a1, a2);
    }

    __device__ __forceinline__ void operator()(int m, int n, f32x4 a1, f32x4 a2) const {
        if (part == 2) { a2 = a1; a1 = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const bool do_mu = part != 2, do_lv = part != 1;
        const int valid = min(4, I - m);
        if (do_mu && gradBias && n < O && m <= I && I < m + 4) {      // the quad that holds the ones row m = I
            const float s = scale * a1[I - m];
            gradBias[n] = accumulate ? gradBias[n] + s : s;
        }
        if (valid <= 0 || n >= O) return;
        const int64_t base = (int64_t)n * I + m;
        float e[4] = {0.f, 0.f, 0.f, 0.f}, sd[4] = {0.f, 0.f, 0.f, 0.f}, var[4] = {0.f, 0.f, 0.f, 0.f};
        const bool want_lv = do_lv && (gradSum || grad_lv);
        if (!lrt && want_lv) {
            const vbnn_f32x4 z = vbnn_normal4(seed, VBNN_STREAM_EPS, layer, draw, (uint32_t)n, (uint32_t)(m >> 2));
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = z.v[j];
        }
        if (var_s && want_lv && (grad_mu || grad_lv)) {
            load4<bf16_t>(var_s + (int64_t)n * ld_w + m, var, valid, true);
        } else if (lvars && want_lv) {
            float lv4[4];
            load4<float>(lvars + base, lv4, valid, vec);
#pragma unroll
            for (int j = 0; j < 4; ++j) if (j < valid) var[j] = expf(lv4[j]);
            if (gradSum || !lrt) {             // stdv is only needed by gradSum and by the weight-noise form
#pragma unroll
                for (int j = 0; j < 4; ++j) sd[j] = sqrtf(var[j]);
            }
        }
        if (gradWeight && do_mu) {
            float o4[4], old4[4] = {0.f, 0.f, 0.f, 0.f};
            if (accumulate) load4<float>(gradWeight + base, old4, valid, vec);
#pragma unroll
            for (int j = 0; j < 4; ++j) o4[j] = fmaf(scale, a1[j], old4[j]);
            store4<float>(gradWeight + base, o4[0], o4[1], o4[2], o4[3], valid, vec);
        }
        if (gradSum && do_lv) {
            float o4[4], old4[4] = {0.f, 0.f, 0.f, 0.f};
            if (accumulate) load4<float>(gradSum + base, old4, valid, vec);
#pragma unroll
            for (int j = 0; j < 4; ++j) o4[j] = lrt ? fmaf(2.0f * a2[j], sd[j], old4[j]) : old4[j] + a1[j] * e[j];
            store4<float>(gradSum + base, o4[0], o4[1], o4[2], o4[3], valid, vec);
        }
        if (grad_mu || grad_lv) {
            // VBLinear.lua:90-98 folded in: likelihood/S (+ kl_scale * KL gradient on the first draw).
            // Reciprocals are formed once per call, so the per-element work is multiply-adds only.
            const float var_hat = (float)stats[2];
            const float invS = 1.0f / S;
            const float k_mu = kl_scale / (B * var_hat);            // d(KL/B)/dmeans  = means / (B var_hat)
            const float k_lv = kl_scale / (2.0f * B);               // d(KL/B)/dlvars  = (vars / var_hat - 1) / (2B)
            const float inv_vh = 1.0f / var_hat;
            float gm[4], gl[4], mu4[4] = {0.f, 0.f, 0.f, 0.f}, om4[4] = {0.f, 0.f, 0.f, 0.f}, ol4[4] = {0.f, 0.f, 0.f, 0.f};
            if (mu_s && !accumulate && do_mu) load4<bf16_t>(mu_s + (int64_t)n * ld_w + m, mu4, valid, true);
            else if (means && !accumulate && do_mu) load4<float>(means + base, mu4, valid, vec);
            if (accumulate && grad_mu && do_mu) load4<float>(grad_mu + base, om4, valid, vec);
            if (accumulate && grad_lv && do_lv) load4<float>(grad_lv + base, ol4, valid, vec);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float lm = scale * a1[j] * invS;
                float ll = lrt ? a2[j] * var[j] * invS : a1[j] * e[j] * sd[j] * (0.5f * invS);
                if (accumulate) {
                    lm += om4[j];
                    ll += ol4[j];
                } else {
                    lm = fmaf(k_mu, mu4[j], lm);
                    ll = fmaf(k_lv, fmaf(var[j], inv_vh, -1.0f), ll);
                }
                gm[j] = lm; gl[j] = ll;
            }
            if (grad_mu && do_mu) store4<float>(grad_mu + base, gm[0], gm[1], gm[2], gm[3], valid, vec);
            if (grad_lv && do_lv) store4<float>(grad_lv + base, gl[0], gl[1], gl[2], gl[3], valid, vec);
        }
    }

    // ---- fast protocol: the fused total gradients of an LRT layer, first draw of the minibatch (the S = 1 step)
    static constexpr int FAST_BATCH = 8;
    static constexpr int FAST_BATCH_V2 = 4;
    struct Pre { bf16x4 lv, mu; };        // lv: sigma^2 itself (from the shadow); widened at use
    struct Lane { unsigned o, os; };      // element offsets of the lane's quad in the O x I tensors / in the O x ld_w shadows
    // The fast protocol reads mu and sigma^2 from the bf16 shadows ONLY (no runtime choice: a branch around a load makes
    // hipcc wait for each load on its own, and the batches are the point of the protocol). Without shadows the guarded
    // form above runs, on the fp32 parameters.
    __host__ __device__ __forceinline__ bool fast_ok() const {
        return lrt && vec && !accumulate && grad_mu && grad_lv && means && lvars && !gradWeight && !gradSum &&
               (int64_t)O * I < (1ll << 31) && mu_s && var_s && ld_w % 4 == 0 && (int64_t)O * ld_w < (1ll << 31) &&
               ((((uintptr_t)mu_s | (uintptr_t)var_s) & 7u) == 0);
    }
    __device__ __forceinline__ Lane lane_init(int nl, int ml) const { return Lane{(unsigned)(nl * I + ml), (unsigned)(nl * ld_w + ml)}; }
    __device__ __forceinline__ bf16x4 shadow4(const bf16_t* p, int um, int un, const Lane& ln) const {
        return *reinterpret_cast<const bf16x4*>(p + ((int64_t)un * ld_w + um) + ln.os);
    }
    __device__ __forceinline__ bool fast_pre_needed() const { return !(part == 1 && kl_scale == 0.f); }
    __device__ __forceinline__ Pre load_fast(int um, int un, const Lane& ln) const {
        // two UNCONDITIONAL loads whatever the part (a branch around a load costs one load latency per position): a
        // one-GEMM launch reads its one tensor twice from the same address instead
        const bf16_t* pa = part == 1 ? mu_s : var_s;
        const bf16_t* pb = part == 2 ? var_s : mu_s;
        Pre p;
        p.lv = shadow4(pa, um, un, ln);
        p.mu = shadow4(pb, um, un, ln);
        return p;
    }
    __device__ __forceinline__ void apply_fast(int um, int un, const Lane& ln, f32x4 a1, f32x4 a2, const Pre& pre, float (&t1)[4],
                                               float (&t2)[4]) const {
        (void)t1; (void)t2;
        if (part == 2) a2 = a1;
        const int64_t ub = (int64_t)un * I + um;
        const float var_hat = (float)stats[2];
        const float invS = 1.0f / S;
        const float k_mu = kl_scale / (B * var_hat);
        const float k_lv = kl_scale / (2.0f * B);
        const float inv_vh = 1.0f / var_hat;
        f32x4 gm, gl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float var = (float)pre.lv[j];
            const float lm = scale * a1[j] * invS;
            const float ll = a2[j] * var * invS;
            gm[j] = fmaf(k_mu, (float)pre.mu[j], lm);
            gl[j] = fmaf(k_lv, fmaf(var, inv_vh, -1.0f), ll);
        }
        if (part != 2) vbnn_store_grad(grad_mu + ub, ln.o, gm);
        if (part != 1) vbnn_store_grad(grad_lv + ub, ln.o, gl);
    }

    // ---- fold protocol: the two outputs depend on one accumulator each, so d/dlvars is FINISHED between the passes
    // (its stores drain under the second pass) and the second pass starts from zero
    static constexpr int FOLD_BATCH = 8;          // all 32 quads' sigma^2 loads (64 registers) in flight together
    static constexpr int FOLD_SERIAL = 0;
    // FOLD_STAGE 2: the fold's output (d/dlvars, fp32 O x I) leaves through a per-wave LDS tile as whole 256-byte row
    // segments (gemm_v3.h) instead of 16 rows x 64 bytes per wave-instruction straight from the MFMA layout
    static constexpr int FOLD_STAGE = 2;
    typedef float fold_st_t;
    __host__ __device__ __forceinline__ float* fold_st_ptr() const { return grad_lv; }
    __host__ __device__ __forceinline__ int64_t fold_st_ld() const { return I; }
    struct FPre { bf16x4 lv; };
    // the fold's scalars, formed ONCE by the kernel (the staged fold's LDS accesses are asm statements with a memory
    // clobber: left inside fold_s, the load of stats[2] and the divide were redone -- and waited for -- per quad)
    struct FoldK { float k_lv, inv_vh, invS; };
    __device__ __forceinline__ FoldK fold_k() const {
        const float var_hat = (float)stats[2];
        return FoldK{kl_scale / (2.0f * B), 1.0f / var_hat, 1.0f / S};
    }
    __device__ __forceinline__ f32x4 fold_s(const FoldK& k, f32x4 a2, const FPre& fp, f32x4& gl) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float var = (float)fp.lv[j];
            gl[j] = fmaf(k.k_lv, fmaf(var, k.inv_vh, -1.0f), a2[j] * var * k.invS);
        }
        return f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ FPre fold_load(int um, int un, const Lane& ln) const {
        FPre p;
        p.lv = shadow4(var_s, um, un, ln);
        return p;
    }
    __device__ __forceinline__ f32x4 fold(int um, int un, const Lane& ln, f32x4 a2, const FPre& fp) const {
        const float var_hat = (float)stats[2];
        const float invS = 1.0f / S;
        const float k_lv = kl_scale / (2.0f * B);
        const float inv_vh = 1.0f / var_hat;
        f32x4 gl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float var = (float)fp.lv[j];
            gl[j] = fmaf(k_lv, fmaf(var, inv_vh, -1.0f), a2[j] * var * invS);
        }
        *reinterpret_cast<f32x4*>(grad_lv + ((int64_t)un * I + um) + ln.o) = gl;
        return f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ bool folded_pre_needed() const { return kl_scale != 0.f; }     // (mu enters through the KL term only)
    __device__ __forceinline__ Pre load_folded(int um, int un, const Lane& ln) const {
        Pre p;
        p.mu = shadow4(mu_s, um, un, ln);
        p.lv = bf16x4{};
        return p;
    }
    __device__ __forceinline__ void apply_folded(int um, int un, const Lane& ln, f32x4 a, f32x4, const Pre& pre, float (&t1)[4],
                                                 float (&t2)[4]) const {
        (void)t1; (void)t2;
        const float var_hat = (float)stats[2];
        const float invS = 1.0f / S;
        const float k_mu = kl_scale / (B * var_hat);
        f32x4 gm;
#pragma unroll
        for (int j = 0; j < 4; ++j) gm[j] = fmaf(k_mu, (float)pre.mu[j], scale * a[j] * invS);
        vbnn_store_grad(grad_mu + ((int64_t)un * I + um), ln.o, gm);
    }
};
