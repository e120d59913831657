// head.hip -- the classifier head of mlp.lua:29-32 for a small class count (C <= 16):
// final nn.Linear (H -> C) + nn.LogSoftMax + nn.ClassNLLCriterion, forward and backward.
//
// With C = 10 these are skinny, HBM-bound passes over the N x H activation (33 MB at the wide
// configuration), not GEMM-shaped work for 256 x 128 tiles:
//   forward   16 rows per workgroup, K split over its 8 waves, one 16x16 MFMA column of logits per wave,
//             partials folded through LDS, log-softmax / loss / arg-max / d(loss)/d(logits) in registers
//   backward  ONE pass over h and r: 64 x 64 tiles, 16-byte loads/stores; the gradInput (through the ReLU, times r)
//             and its transposed copies via an LDS tile; gradWeight and the bias gradient of the layer below as
//             16x16 MFMA tiles contracted over the minibatch rows of that same LDS tile; row-chunk partials summed
//             in a fixed order by a finish kernel (no float atomics)
#include "common.h"
#include "gemm_v1.h"      // Frag<T>, mfma_step<T>

constexpr int HEAD_CMAX = 16;
constexpr int HEAD_FW = 8;         // waves of a forward workgroup: the K range of its 16 rows is split over them

template <typename T> struct Vec8;             // eight consecutive packed elements
template <> struct Vec8<bf16_t> {
    typedef bf16x8 raw_t;                                   // 16 bytes kept unconverted while loads are in flight
    static __device__ __forceinline__ raw_t load_raw(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
    static __device__ __forceinline__ void cvt(const raw_t& t, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
    }
    static __device__ __forceinline__ raw_t load_guarded(const bf16_t* p, int valid) {       // leading `valid` elements, 0 after
        raw_t t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = e < valid ? p[e] : (bf16_t)0.f;
        return t;
    }
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        bf16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (bf16_t)v[e];
        *reinterpret_cast<bf16x8*>(p) = t;
    }
};
template <> struct Vec8<float> {
    struct raw_t { f32x4 a, b; };
    static __device__ __forceinline__ raw_t load_raw(const float* p) {
        return raw_t{*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4)};
    }
    static __device__ __forceinline__ void cvt(const raw_t& t, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = t.a[e]; v[4 + e] = t.b[e]; }
    }
    static __device__ __forceinline__ raw_t load_guarded(const float* p, int valid) {
        raw_t t;
#pragma unroll
        for (int e = 0; e < 4; ++e) { t.a[e] = e < valid ? p[e] : 0.f; t.b[e] = e + 4 < valid ? p[e + 4] : 0.f; }
        return t;
    }
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
        *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
};

// R partials `stride` apart, added in order r = 0, 1, ...: eight write-through-visible loads in flight at a time (a loop of
// one load per iteration pays the L2 round trip R times: 16 partials = 10 us at the tail of a 10 us kernel)
__device__ __forceinline__ float head_sum_partials(const float* src, int64_t stride, int R) {
    float tot = 0.f;
    int r = 0;
    for (; r + 8 <= R; r += 8) {
        float p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = vbnn_load_wt(src + (int64_t)(r + u) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) tot += p[u];
    }
    if (r < R) {
        float p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = (r + u < R) ? vbnn_load_wt(src + (int64_t)(r + u) * stride) : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (r + u < R) tot += p[u];
    }
    return tot;
}

// ------------------------------------------------------------------------------------------ forward
// The rows of one 16-row block from their logits (without the bias): lane (q = lane >> 4, c16 = lane & 15) of ONE wave holds classes
// 4q .. 4q + 3 of row n0 + c16 in `s`. Bias, log-softmax, arg-max (first maximum wins: Tensor:max), d(loss)/d(logits), and the
// block's loss / hit partials (write-through: the last arriver's payload).
__device__ __forceinline__ void head_rows_from_logits(f32x4 s, int lane, int64_t n0, int64_t N, int C, const float* __restrict__ bias,
                                                      const int32_t* __restrict__ target, int64_t rpd, float inv_n, float* out, float* g_logits,
                                                      float* logits, double* part_loss, int32_t* part_corr) {
    const int q = lane >> 4, c16 = lane & 15;
    const int64_t n = n0 + c16;
    const bool row_ok = n < N;
    float lg[4];
    float mx = -INFINITY; int arg = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * q + j;
        lg[j] = (c < C) ? s[j] + (bias ? bias[c] : 0.f) : -INFINITY;
        if (lg[j] > mx) { mx = lg[j]; arg = c; }
    }
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {           // combine the four lanes (q = 0..3) of a row
        const float om = __shfl_xor(mx, off, 64);
        const int oa = __shfl_xor(arg, off, 64);
        if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }     // first maximum wins (Tensor:max)
    }
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) if (4 * q + j < C) se += expf(lg[j] - mx);
    se += __shfl_xor(se, 16, 64);
    se += __shfl_xor(se, 32, 64);
    const float lse = mx + logf(se);
    const int t = row_ok ? min(max(target[rpd > 0 ? n % rpd : n], 0), C - 1) : 0;     // stacked draws share the minibatch's targets
    double loss_acc = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * q + j;
        if (row_ok && c < C) {
            const float o = lg[j] - lse;
            if (logits) logits[n * C + c] = lg[j];
            if (out) out[n * C + c] = o;
            g_logits[n * C + c] = (expf(o) - (c == t ? 1.0f : 0.0f)) * inv_n;
            if (c == t) loss_acc -= (double)o * (double)inv_n;
        }
    }
    int corr = (row_ok && q == 0 && arg == t) ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { loss_acc += __shfl_xor(loss_acc, off, 64); corr += __shfl_xor(corr, off, 64); }
    if (lane == 0) { vbnn_store_wt(&part_loss[blockIdx.x], loss_acc); vbnn_store_wt(&part_corr[blockIdx.x], (int32_t)corr); }
}

// the last workgroup to arrive adds the per-workgroup loss / hit partials in a fixed order (NT threads per workgroup)
template <int NT>
__device__ __forceinline__ void head_sum_loss_partials(unsigned* counter, const double* part_loss, const int32_t* part_corr, double* loss_sum,
                                                       int32_t* correct, int accumulate, double* red_l, int* red_c, int* last) {
    if (!vbnn_last_arriver(counter, gridDim.x, last)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double ls = 0.0;
    int cs = 0;
    for (int b = (int)threadIdx.x; b < (int)gridDim.x; b += NT) { ls += vbnn_load_wt(&part_loss[b]); cs += vbnn_load_wt(&part_corr[b]); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { ls += __shfl_xor(ls, off, 64); cs += __shfl_xor(cs, off, 64); }
    if (lane == 0) { red_l[wave] = ls; red_c[wave] = cs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double lt = 0.0;
        int ct = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) { lt += red_l[w]; ct += red_c[w]; }
        if (loss_sum) loss_sum[0] = (accumulate ? loss_sum[0] : 0.0) + lt;
        if (correct) correct[0] = (accumulate ? correct[0] : 0) + ct;
    }
}

// MFMA orientation: M = class (A operand = w3 rows, rows >= C clamped and ignored), N = minibatch row.
// Accumulator layout: lane (q = l >> 4, c = l & 15) holds classes 4q .. 4q+3 of row n0 + c.
template <typename T>
__global__ __launch_bounds__(64 * HEAD_FW) void k_head_forward(const T* __restrict__ h, int64_t ld_h, const T* __restrict__ w3, int64_t ld_w,
                                                      const float* __restrict__ bias, const int32_t* __restrict__ target, int64_t N,
                                                      int64_t Hp /* H padded to the K step */, int C, float inv_n, float* out,
                                                      float* g_logits, float* logits, double* loss_sum, int32_t* correct,
                                                      int accumulate, double* part_loss, int32_t* part_corr, unsigned* counter, int64_t rpd) {
    constexpr int KE = 64 / (int)sizeof(T);      // K elements per MFMA step (16 bytes per lane x 4 lane groups)
    constexpr int CE = 16 / (int)sizeof(T);
    typedef typename Frag<T>::type frag_t;
    __shared__ f32x4 part[HEAD_FW][64];
    __shared__ double red_l[HEAD_FW];
    __shared__ int red_c[HEAD_FW];
    __shared__ int last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n0 = (int64_t)blockIdx.x * 16;
    const int q = lane >> 4, c16 = lane & 15;
    const T* hp = h + min(n0 + c16, N - 1) * ld_h + q * CE;
    const T* wp = w3 + (int64_t)min(c16, C - 1) * ld_w + q * CE;
    const int64_t ksteps = Hp / KE;
    const int64_t k_lo = ksteps * wave / HEAD_FW, k_hi = ksteps * (wave + 1) / HEAD_FW;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int64_t ks = k_lo;
    for (; ks + 16 <= k_hi; ks += 16) {               // sixteen K steps of loads in flight: at H = 4096 a wave's whole K range in
        frag_t a[16], b[16];                          // ONE load round (two rounds of eight paid the load latency twice)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            a[u] = *reinterpret_cast<const frag_t*>(wp + (ks + u) * KE);
            b[u] = *reinterpret_cast<const frag_t*>(hp + (ks + u) * KE);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = mfma_step<T>(a[u], b[u], acc);
    }
    for (; ks + 8 <= k_hi; ks += 8) {                 // eight K steps of loads in flight (latency-bound otherwise)
        frag_t a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a[u] = *reinterpret_cast<const frag_t*>(wp + (ks + u) * KE);
            b[u] = *reinterpret_cast<const frag_t*>(hp + (ks + u) * KE);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = mfma_step<T>(a[u], b[u], acc);
    }
    for (; ks + 2 <= k_hi; ks += 2) {
        frag_t a[2], b[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            a[u] = *reinterpret_cast<const frag_t*>(wp + (ks + u) * KE);
            b[u] = *reinterpret_cast<const frag_t*>(hp + (ks + u) * KE);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) acc = mfma_step<T>(a[u], b[u], acc);
    }
    for (; ks < k_hi; ++ks) {
        const frag_t a = *reinterpret_cast<const frag_t*>(wp + ks * KE);
        const frag_t b = *reinterpret_cast<const frag_t*>(hp + ks * KE);
        acc = mfma_step<T>(a, b, acc);
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {
        f32x4 s = part[0][lane];
#pragma unroll
        for (int w = 1; w < HEAD_FW; ++w) s += part[w][lane];
        head_rows_from_logits(s, lane, n0, N, C, bias, target, rpd, inv_n, out, g_logits, logits, part_loss, part_corr);
    }
    // ---- second stage: the last workgroup to arrive adds the per-workgroup loss / hit partials in a fixed order
    // (bitwise reproducible loss; and with accumulate = 0 the caller needs no memset of the two accumulators)
    head_sum_loss_partials<64 * HEAD_FW>(counter, part_loss, part_corr, loss_sum, correct, accumulate, red_l, red_c, &last);
}

// The same forward from partial logits left by the last VB layer's forward launch (EpiFwd::head_slots, gemm_v3.h): one wave per
// 16 rows, lane (q, c16) adds classes 4q .. 4q + 3 of row n0 + c16 over the slots IN ORDER (all loads in flight together),
// then exactly k_head_forward's row arithmetic and its in-launch loss sum. h is not read: 8 MB of slots instead of 33 MB.
__global__ __launch_bounds__(64) void k_head_from_slots(const float* __restrict__ slots, int n_slots, const float* __restrict__ bias,
                                                        const int32_t* __restrict__ target, int64_t N, int C, float inv_n, float* out,
                                                        float* g_logits, float* logits, double* loss_sum, int32_t* correct, int accumulate,
                                                        double* part_loss, int32_t* part_corr, unsigned* counter, int64_t rpd) {
    __shared__ double red_l[1];
    __shared__ int red_c[1];
    __shared__ int last;
    const int lane = threadIdx.x;
    const int64_t n0 = (int64_t)blockIdx.x * 16;
    const int q = lane >> 4, c16 = lane & 15;
    const float* p = slots + (min(n0 + c16, N - 1) * 16 + 4 * q);
    const int64_t stride = N * 16;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 16 <= n_slots; k += 16) {
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (int64_t)(k + u) * stride);
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; k + 2 <= n_slots; k += 2) {
        f32x4 v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (int64_t)(k + u) * stride);
        s += v[0]; s += v[1];
    }
    for (; k < n_slots; ++k) s += *reinterpret_cast<const f32x4*>(p + (int64_t)k * stride);
    head_rows_from_logits(s, lane, n0, N, C, bias, target, rpd, inv_n, out, g_logits, logits, part_loss, part_corr);
    head_sum_loss_partials<64>(counter, part_loss, part_corr, loss_sum, correct, accumulate, red_l, red_c, &last);
}

// ------------------------------------------------------------------------------------------ backward
// ONE pass over h and r for everything the head's backward produces:
//   gx[n][i]         = sum_c g[n][c] w3[c][i];  g_prev = gx . [h > 0];  gv_prev = g_prev . r   (+ transposed copies)
//   gradWeight[c][i] = sum_n g[n][c] h[n][i]          MFMA: M = class, N = hidden unit, K = minibatch row
//   gradBias_prev[i] = sum_n g_prev[n][i]             the same MFMA shape with a row of ones as the A operand
//   gradBias[c]      = sum_n g[n][c]
// grid (H / 64, R): a block owns 64 hidden units and walks the 64-row tiles of its row chunk; thread
// (row = tid >> 3 (+32), chunk = tid & 7) handles 8 consecutive hidden units of a row (16-byte loads / stores, the
// next tile's loads in flight while this one is worked on). The tile of h, g_prev and gv_prev is kept in LDS in the
// operand type: the transposed copies are read back from it as rows of n, the MFMA B fragments as columns. The sums
// over rows leave as R partials per output; k_head_backward_finish adds them in chunk order (no float atomics).
template <typename T> struct HeadRaw {          // one thread's 8 hidden units of h and of r, as loaded
    typename Vec8<T>::raw_t h, rt;
    typename Vec8<float>::raw_t rf;
};
template <typename T, bool VEC>
__device__ __forceinline__ void head_load(HeadRaw<T>& o, const T* __restrict__ h, int64_t ld_h, const float* __restrict__ r_prev,
                                          const T* __restrict__ r_prev_t, int64_t ld_r, int64_t n, int64_t n_hi, int64_t i, int64_t H) {
    const int valid = (n < n_hi && i < H) ? (int)min((int64_t)8, H - i) : 0;
    if (VEC && valid == 8) {
        o.h = Vec8<T>::load_raw(h + n * ld_h + i);
        if (r_prev_t) o.rt = Vec8<T>::load_raw(r_prev_t + n * ld_r + i);
        if (r_prev) o.rf = Vec8<float>::load_raw(r_prev + n * ld_r + i);
    } else {
        o.h = Vec8<T>::load_guarded(h + n * ld_h + i, valid);
        if (r_prev_t) o.rt = Vec8<T>::load_guarded(r_prev_t + n * ld_r + i, valid);
        if (r_prev) o.rf = Vec8<float>::load_guarded(r_prev + n * ld_r + i, valid);
    }
}

// FULL: every tile is whole and vector-accessible (H % 64 == 0, N % 64 == 0, aligned operands): no bounds checks at
// all, and every global access is (block-uniform base) + (one 32-bit lane offset) -- the guarded general form spent
// most of its VALU issue slots on 64-bit address arithmetic and exec-mask bookkeeping (it was issue-bound, not HBM-bound).
// CP > 0 (FULL only, C <= CP, r in the operand type or absent): the block's 64 columns of the final weight live in
// REGISTERS (CP x 8 per thread) instead of being re-read from LDS for every row of every tile -- the loop was bound
// by LDS round trips at two waves per SIMD, not by HBM (g_prev alone: 30 us for 67 MB).
template <typename T, bool VEC, bool FULL, int CP = 0>
__global__ __launch_bounds__(256) void k_head_backward(const T* __restrict__ h, int64_t ld_h, const T* __restrict__ w3, int64_t ld_w,
                                                       const float* __restrict__ g, int64_t N, int64_t H, int C, int relu_mask,
                                                       const float* __restrict__ r_prev, const T* __restrict__ r_prev_t, int64_t ld_r,
                                                       T* g_prev, T* gv_prev, int64_t ld_gp, T* gT_prev, T* gvT_prev, int64_t ld_gpT,
                                                       int rows_per_chunk, float* __restrict__ partial_w /* [R][C][H] */,
                                                       float* __restrict__ partial_b /* [R][C] */,
                                                       float* __restrict__ partial_bp /* [R][H] */,
                                                       unsigned* fin_tickets /* NULL: a finish kernel follows */, int fin_accumulate,
                                                       float* fin_gw, float* fin_gb, float* fin_gbp) {
    constexpr int CE = 16 / (int)sizeof(T);      // K elements per lane of an MFMA fragment
    constexpr int KE = 4 * CE;                   // rows contracted per mfma_step
    constexpr int TP = sizeof(T) == 2 ? 66 : 65;  // tile pitch, 33 / 65 dwords: rows 8 apart fall on different banks
    typedef typename Frag<T>::type frag_t;
    __shared__ __attribute__((aligned(16))) T th[64][TP];      // h tile
    __shared__ __attribute__((aligned(16))) T tg[64][TP];      // g_prev tile, as stored
    __shared__ __attribute__((aligned(16))) T tv[64][TP];      // gv_prev tile
    __shared__ __attribute__((aligned(16))) float gs[2][64][HEAD_CMAX + 4];   // d(loss)/d(logits) rows of the tile, operand-rounded; 80-B rows: 16-byte reads, 8 rows on 8 bank groups
    __shared__ __attribute__((aligned(16))) float ws[HEAD_CMAX][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = tid & 7, c16 = lane & 15, q = lane >> 4;
    const int64_t c0 = (int64_t)blockIdx.x * 64;
    const int64_t n_lo = (int64_t)blockIdx.y * rows_per_chunk, n_hi = min(N, n_lo + rows_per_chunk);
    const int64_t i8 = c0 + ch * 8;
    for (int k = tid; k < HEAD_CMAX * 64; k += 256) {
        const int wc = k >> 6, wi = k & 63;
        ws[wc][wi] = (wc < C && c0 + wi < H) ? Elt<T>::from(w3[(int64_t)wc * ld_w + c0 + wi]) : 0.f;
    }
    float wreg[CP > 0 ? CP : 1][8];
    if constexpr (CP > 0) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CP; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) wreg[c][e] = ws[c][ch * 8 + e];
    }
    const bool has_r = r_prev || r_prev_t;
    f32x4 accw = {0.f, 0.f, 0.f, 0.f}, accp = {0.f, 0.f, 0.f, 0.f};
    float accb = 0.f;
    frag_t ones;
#pragma unroll
    for (int e = 0; e < CE; ++e) ones[e] = Elt<T>::to(c16 == 0 ? 1.f : 0.f);
    HeadRaw<T> cur[2] = {}, nxt[2] = {};
    // FULL addressing: lane offsets (elements) of the thread's row (tid >> 3) and 8-column chunk inside a tile
    const unsigned lo_h = (unsigned)((tid >> 3) * (int)ld_h + ch * 8), lo_r = (unsigned)((tid >> 3) * (int)ld_r + ch * 8);
    const unsigned lo_g = (unsigned)((tid >> 3) * (int)ld_gp + ch * 8), lo_t = (unsigned)((tid >> 3) * (int)ld_gpT + ch * 8);
    auto load_tile = [&](HeadRaw<T> (&dst)[2], int64_t r0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if constexpr (FULL) {
                dst[p].h = Vec8<T>::load_raw(h + ((r0 + 32 * p) * ld_h + c0) + lo_h);
                if (r_prev_t) dst[p].rt = Vec8<T>::load_raw(r_prev_t + ((r0 + 32 * p) * ld_r + c0) + lo_r);
                if constexpr (CP == 0) { if (r_prev) dst[p].rf = Vec8<float>::load_raw(r_prev + ((r0 + 32 * p) * ld_r + c0) + lo_r); }
            } else {
                head_load<T, VEC>(dst[p], h, ld_h, r_prev, r_prev_t, ld_r, r0 + (tid >> 3) + 32 * p, n_hi, i8, H);
            }
        }
    };
    load_tile(cur, n_lo);
    // the tile's rows of g are staged through registers one tile ahead as well: a load + barrier at the top of every
    // tile put ~2 us of latency on each block's critical path (the kernel ran at 2 TB/s)
    float gcur[4], gnxt[4] = {0.f, 0.f, 0.f, 0.f};
    auto load_g = [&](int64_t r0, float (&q)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + 256 * k, rr = idx >> 4, cc = idx & 15;
            q[k] = (cc < C && r0 + rr < n_hi) ? g[(r0 + rr) * C + cc] : 0.f;
        }
    };
    load_g(n_lo, gcur);
    int buf = 0;
    for (int64_t r0 = n_lo; r0 < n_hi; r0 += 64, buf ^= 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + 256 * k;
            gs[buf][idx >> 4][idx & 15] = Elt<T>::from(Elt<T>::to(gcur[k]));
        }
        const bool more = r0 + 64 < n_hi;                   // block-uniform
        if (more) {
            load_g(r0 + 64, gnxt);
            load_tile(nxt, r0 + 64);
        }
        __syncthreads();         // gs[buf] (and ws) visible; every read of the previous tile's th / tg / tv is done
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int rr = (tid >> 3) + 32 * p;
            const int64_t n = r0 + rr;
            float hv[8], rv[8], gx[8], gp[8], gvp[8];
            Vec8<T>::cvt(cur[p].h, hv);                      // out-of-range elements were loaded as 0
#pragma unroll
            for (int e = 0; e < 8; ++e) { gx[e] = 0.f; rv[e] = 0.f; }
            if (r_prev_t) Vec8<T>::cvt(cur[p].rt, rv);
            if constexpr (CP == 0) { if (r_prev) Vec8<float>::cvt(cur[p].rf, rv); }
            if constexpr (CP > 0) {
#pragma unroll
                for (int c4 = 0; c4 < CP; c4 += 4) {
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(&gs[buf][rr][c4]);      // classes past C: weights are 0
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int e = 0; e < 8; ++e) gx[e] = fmaf(g4[k], wreg[c4 + k][e], gx[e]);
                }
            } else {
#pragma unroll
                for (int c = 0; c < HEAD_CMAX; ++c)
                    if (c < C) {
                        const float gv = gs[buf][rr][c];        // 0 for rows past the chunk
#pragma unroll
                        for (int e = 0; e < 8; ++e) gx[e] = fmaf(gv, ws[c][ch * 8 + e], gx[e]);     // ws is 0 past H
                    }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                gp[e] = (relu_mask && !(hv[e] > 0.f)) ? 0.f : gx[e];
                gvp[e] = has_r ? gp[e] * rv[e] : 0.f;
            }
            if constexpr (FULL) {
                if (g_prev) Vec8<T>::store(g_prev + ((r0 + 32 * p) * ld_gp + c0) + lo_g, gp);
                if (gv_prev) Vec8<T>::store(gv_prev + ((r0 + 32 * p) * ld_gp + c0) + lo_g, gvp);
            } else if (n < n_hi && i8 < H) {
                if (VEC && i8 + 8 <= H) {
                    if (g_prev) Vec8<T>::store(g_prev + n * ld_gp + i8, gp);
                    if (gv_prev) Vec8<T>::store(gv_prev + n * ld_gp + i8, gvp);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (i8 + e < H) {
                            if (g_prev) g_prev[n * ld_gp + i8 + e] = Elt<T>::to(gp[e]);
                            if (gv_prev) gv_prev[n * ld_gp + i8 + e] = Elt<T>::to(gvp[e]);
                        }
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                th[rr][ch * 8 + e] = Elt<T>::to(hv[e]);
                tg[rr][ch * 8 + e] = Elt<T>::to(gp[e]);
                tv[rr][ch * 8 + e] = Elt<T>::to(gvp[e]);
            }
        }
        __syncthreads();
        if (gT_prev || gvT_prev) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cc = (tid >> 3) + 32 * p;          // hidden unit inside the tile
                const int64_t i = c0 + cc, n = r0 + ch * 8;
                if (!FULL && (i >= H || n >= n_hi)) continue;
                float a[8], b[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { a[e] = Elt<T>::from(tg[ch * 8 + e][cc]); b[e] = Elt<T>::from(tv[ch * 8 + e][cc]); }
                if constexpr (FULL) {
                    if (gT_prev) Vec8<T>::store(gT_prev + ((c0 + 32 * p) * ld_gpT + r0) + lo_t, a);
                    if (gvT_prev) Vec8<T>::store(gvT_prev + ((c0 + 32 * p) * ld_gpT + r0) + lo_t, b);
                } else if (VEC && n + 8 <= n_hi) {
                    if (gT_prev) Vec8<T>::store(gT_prev + i * ld_gpT + n, a);
                    if (gvT_prev) Vec8<T>::store(gvT_prev + i * ld_gpT + n, b);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (n + e < n_hi) {
                            if (gT_prev) gT_prev[i * ld_gpT + n + e] = Elt<T>::to(a[e]);
                            if (gvT_prev) gvT_prev[i * ld_gpT + n + e] = Elt<T>::to(b[e]);
                        }
                }
            }
        }
        if (partial_w) {                                     // wave w: hidden units 16 w .. 16 w + 15 of the tile
#pragma unroll
            for (int s = 0; s < 64 / KE; ++s) {
                frag_t a, b, b1;
#pragma unroll
                for (int e = 0; e < CE; ++e) {
                    const int k = s * KE + q * CE + e;
                    a[e] = Elt<T>::to(gs[buf][k][c16]);
                    b[e] = th[k][16 * wave + c16];
                    b1[e] = tg[k][16 * wave + c16];
                }
                accw = mfma_step<T>(a, b, accw);
                if (partial_bp) accp = mfma_step<T>(ones, b1, accp);
            }
        }
        if (partial_b && blockIdx.x == 0 && c16 < C) {        // 16 row lanes per class, UNROUNDED g
            const int rows = (int)min((int64_t)64, n_hi - r0);
            for (int rr = tid >> 4; rr < rows; rr += 16) accb += g[(r0 + rr) * C + c16];
        }
        if (more) {
            cur[0] = nxt[0]; cur[1] = nxt[1];
#pragma unroll
            for (int k = 0; k < 4; ++k) gcur[k] = gnxt[k];
        }
    }
    // (with an in-launch finish the partials are the last arriver's payload: write-through stores, vbnn_last_arriver)
    auto put = [&](float* p, float v) { if (fin_tickets) vbnn_store_wt(p, v); else *p = v; };
    if (partial_w) {                                         // lane (q, c16): classes 4q .. 4q+3 of hidden unit 16 w + c16
        const int64_t i = c0 + 16 * wave + c16;
        if (i < H) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * q + j < C) put(&partial_w[((int64_t)blockIdx.y * C + 4 * q + j) * H + i], accw[j]);
            if (partial_bp && q == 0) put(&partial_bp[(int64_t)blockIdx.y * H + i], accp[0]);
        }
    }
    if (partial_b && blockIdx.x == 0) {
        __syncthreads();                                     // the tiles are free
        float* red = reinterpret_cast<float*>(&th[0][0]);    // [16 row lanes][16 classes]
        red[tid] = accb;
        __syncthreads();
        if (tid < C) {
            float tot = 0.f;
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) tot += red[rg * 16 + tid];
            put(&partial_b[(int64_t)blockIdx.y * C + tid], tot);
        }
    }
    // ---- in-launch finish (few workgroups: the launch-bound configurations, where a separate finish kernel is ~4.6 us of
    // an 85 us step): the LAST of the R row-chunk workgroups of this column tile to arrive adds the R partials of the
    // tile's outputs in chunk order -- the order and the arithmetic of k_head_backward_finish, so the same bits.
    if (!fin_tickets) return;
    __shared__ int fin_last;
    if (!vbnn_last_arriver(fin_tickets + blockIdx.x, gridDim.y, &fin_last)) return;
    const int R = (int)gridDim.y;
    const int64_t nw = (int64_t)C * H;
    for (int k = tid; k < C * 64 + 64; k += 256) {           // the tile's C x 64 gradWeight entries, then its 64 gradBias_prev entries
        const bool is_w = k < C * 64;
        const int64_t i = c0 + (is_w ? k & 63 : k - C * 64);
        if (i >= H) continue;
        float* dst = is_w ? (fin_gw ? fin_gw + (int64_t)(k >> 6) * H + i : nullptr) : (fin_gbp ? fin_gbp + i : nullptr);
        if (!dst) continue;
        const float* src = is_w ? partial_w + (int64_t)(k >> 6) * H + i : partial_bp + i;
        const int64_t stride = is_w ? nw : H;
        const float tot = head_sum_partials(src, stride, R);
        *dst = (fin_accumulate ? *dst : 0.f) + tot;
    }
    if (blockIdx.x == 0 && tid < C && fin_gb) {
        const float tot = head_sum_partials(partial_b + tid, C, R);
        fin_gb[tid] = (fin_accumulate ? fin_gb[tid] : 0.f) + tot;
    }
}

// ---- the STREAMING form of the same backward (r05): bf16, whole tiles, C <= 12, r in the operand type, no transposed outputs -- the wide
// configurations. k_head_backward above is latency-bound at TWO waves per SIMD (248 registers: the weight columns as 96 floats;
// two barriers and three LDS tiles per 64 rows; r04's counters: VALU 27 % busy, LDS 26 %, nothing saturated). This form has no LDS
// in its loop and no barrier, and runs FOUR waves per SIMD (1024-thread workgroups, <= 128 registers):
//   - a lane owns 4 consecutive hidden units of a 256-wide column block, a wave a set of row PAIRS of the workgroup's row chunk;
//   - d(loss)/d(logits) of a row is wave-uniform: 24 lanes load the pair's 2 x 12 values, round them to the operand type and hand
//     them round by v_readlane -- they are SGPR operands of the dot products below, packed by the scalar unit as pairs over
//     CLASSES (c, c + 1) of one row for the gradInput and as pairs over ROWS (n, n + 1) of one class for gradWeight;
//   - gx[e] = sum over class pairs of v_dot2c_f32_bf16(g pair, weight pair): the weight columns as 24 registers of packed pairs;
//   - gradWeight[c][i] += v_dot2c_f32_bf16(g row pair of class c, h row pair of unit i): 48 accumulators per lane, per row PAIR;
//   - the bias gradient of the layer below: the rounded g_prev added up per lane; the final bias gradient from the unrounded g.
// The sixteen waves' accumulators meet ONCE, at the end, through LDS in wave order (deterministic); the row-chunk partials go to
// the same scratch layout and the same finish kernel. bf16 products are exact in fp32; the sums differ from the MFMA form's in
// association (tolerances of the parity tests; every host runs the same kernel).
typedef __bf16 hb_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float hb_dot2(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hb_bf16x2, a), __builtin_bit_cast(hb_bf16x2, b), c, false);
}
__device__ __forceinline__ unsigned hb_bf16_bits(float v) {           // RNE, NaN-safe: the operand type's rounding (v_cvt_pk_bf16_f32)
    return (unsigned)__builtin_bit_cast(unsigned short, (bf16_t)v);
}
// U: hidden units per lane (a wave covers 64 U of them). 4 needs 48 + 24 registers of accumulators and weight pairs alone and spilled
// at the 128 a 1024-thread workgroup allows; 2 (4-byte loads, 128 units per wave) leaves room for the next row pair in flight.
template <int U, int HEAD_SW /* waves of a workgroup */>
__global__ __launch_bounds__(64 * HEAD_SW) void k_head_backward_stream(
    const bf16_t* __restrict__ h, int64_t ld_h, const bf16_t* __restrict__ w3, int64_t ld_w, const float* __restrict__ g, int64_t N,
    int64_t H, int C, int relu_mask, const bf16_t* __restrict__ r_prev, int64_t ld_r, bf16_t* __restrict__ g_prev,
    bf16_t* __restrict__ gv_prev, int64_t ld_gp, int rows_per_chunk, float* __restrict__ partial_w /* [R][C][H] */,
    float* __restrict__ partial_b /* [R][C] */, float* __restrict__ partial_bp /* [R][H] */,
    unsigned* fin_tickets /* NULL: a finish kernel follows */, int fin_accumulate, float* fin_gw, float* fin_gb, float* fin_gbp) {
    static_assert(U == 2 || U == 4, "2 or 4 hidden units per lane");
    constexpr int CP = 12, NP = CP / 2, UW = U / 2;                    // UW: 32-bit words of packed bf16 per lane and row
    typedef float accv __attribute__((ext_vector_type(U)));
    typedef unsigned rawv __attribute__((ext_vector_type(UW)));
    __shared__ accv red[4][HEAD_SW][64];                               // four accumulator groups of every wave at a time
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t cb0 = (int64_t)blockIdx.x * (64 * U);                // the workgroup's column block
    const int64_t c0 = cb0 + U * lane;                                 // the lane's U hidden units
    const int64_t n_lo = (int64_t)blockIdx.y * rows_per_chunk, n_hi = min(N, n_lo + rows_per_chunk);
    // ---- the lane's weight columns as pairs over classes: wp[p][e] = (w3[2p][c0 + e], w3[2p + 1][c0 + e])
    unsigned wp[NP][U];
    {
        unsigned wr[CP][UW];
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            rawv t = 0u;
            if (c < C) t = *reinterpret_cast<const rawv*>(w3 + (int64_t)c * ld_w + c0);
#pragma unroll
            for (int k = 0; k < UW; ++k) wr[c][k] = ((const unsigned*)&t)[k];
        }
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int e = 0; e < U; ++e) {
                const unsigned lo = (wr[2 * p][e >> 1] >> (16 * (e & 1))) & 0xffffu, hi = (wr[2 * p + 1][e >> 1] >> (16 * (e & 1))) & 0xffffu;
                wp[p][e] = lo | (hi << 16);
            }
    }
    float accw[CP][U], accb[U], accg = 0.f;
#pragma unroll
    for (int e = 0; e < U; ++e) accb[e] = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c)
#pragma unroll
        for (int e = 0; e < U; ++e) accw[c][e] = 0.f;
    // lanes 0 .. 11: class `lane` of the pair's first row; lanes 12 .. 23: class `lane - 12` of its second row
    const int gcls = lane < CP ? lane : lane - CP;
    const bool gl_on = lane < 2 * CP && gcls < C;
    const bool has_r = r_prev != nullptr;
    const unsigned lo_h = (unsigned)(U * lane);
    auto ldrow = [&](int64_t n, rawv& hv, rawv& rv) {
        hv = *reinterpret_cast<const rawv*>(h + (n * ld_h + cb0) + lo_h);
        rv = 0u;
        if (has_r) rv = *reinterpret_cast<const rawv*>(r_prev + (n * ld_r + cb0) + lo_h);
    };
    auto ldg = [&](int64_t n) { return gl_on ? g[(n + (lane >= CP ? 1 : 0)) * C + gcls] : 0.f; };
    auto word = [](const rawv& v, int k) { return ((const unsigned*)&v)[k]; };
    int64_t n = n_lo + 2 * wave;
    rawv h0 = 0u, h1 = 0u, r0 = 0u, r1 = 0u;                            // one row pair ahead in flight
    float gv = 0.f;
    if (n < n_hi) { ldrow(n, h0, r0); ldrow(n + 1, h1, r1); gv = ldg(n); }
    for (; n < n_hi; n += 2 * HEAD_SW) {
        const rawv hh[2] = {h0, h1}, rr[2] = {r0, r1};
        const float cg = gv;
        const int64_t nn = n + 2 * HEAD_SW;
        if (nn < n_hi) { ldrow(nn, h0, r0); ldrow(nn + 1, h1, r1); gv = ldg(nn); }
        accg += cg;                                                    // UNROUNDED: the final Linear's bias gradient (column block 0 stores it)
        const unsigned gb = hb_bf16_bits(cg);
        // the pair's 2 x 12 rounded values as wave-uniform scalars
        unsigned g0[CP], g1[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) { g0[c] = __builtin_amdgcn_readlane(gb, c); g1[c] = __builtin_amdgcn_readlane(gb, CP + c); }
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            const unsigned* gs = row ? g1 : g0;
            float gx[U];
#pragma unroll
            for (int e = 0; e < U; ++e) gx[e] = 0.f;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const unsigned gp2 = gs[2 * p] | (gs[2 * p + 1] << 16);     // classes (2p, 2p + 1) of this row: scalar
#pragma unroll
                for (int e = 0; e < U; ++e) gx[e] = hb_dot2(gp2, wp[p][e], gx[e]);
            }
            unsigned gbits[U], vbits[U];
#pragma unroll
            for (int e = 0; e < U; ++e) {
                const unsigned hw = word(hh[row], e >> 1), rw = word(rr[row], e >> 1);
                const float hv = __builtin_bit_cast(float, (e & 1) ? (hw & 0xffff0000u) : (hw << 16));
                const float rv = __builtin_bit_cast(float, (e & 1) ? (rw & 0xffff0000u) : (rw << 16));
                const float gp = (relu_mask && !(hv > 0.f)) ? 0.f : gx[e];
                gbits[e] = hb_bf16_bits(gp);
                vbits[e] = hb_bf16_bits(has_r ? gp * rv : 0.f);
                accb[e] += __builtin_bit_cast(float, gbits[e] << 16);  // the layer below's bias gradient adds the ROUNDED g_prev (what its GEMMs read)
            }
            const int64_t ro = (n + row) * ld_gp + cb0;
            rawv og, ov;
#pragma unroll
            for (int k = 0; k < UW; ++k) { ((unsigned*)&og)[k] = gbits[2 * k] | (gbits[2 * k + 1] << 16); ((unsigned*)&ov)[k] = vbits[2 * k] | (vbits[2 * k + 1] << 16); }
            *reinterpret_cast<rawv*>(g_prev + ro + lo_h) = og;
            if (gv_prev) *reinterpret_cast<rawv*>(gv_prev + ro + lo_h) = ov;
        }
        // gradWeight: pairs over the two ROWS -- (h[n][i], h[n + 1][i]) per unit, (g[n][c], g[n + 1][c]) per class (scalar)
        unsigned hp[U];
#pragma unroll
        for (int e = 0; e < U; ++e) {
            const unsigned a = word(hh[0], e >> 1), b = word(hh[1], e >> 1);
            hp[e] = (e & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
        }
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            const unsigned gr2 = g0[c] | (g1[c] << 16);
#pragma unroll
            for (int e = 0; e < U; ++e) accw[c][e] = hb_dot2(gr2, hp[e], accw[c][e]);
        }
    }
    // ---- the sixteen waves' accumulators meet, four groups at a time, summed in wave order by waves 0 .. 3
    const int64_t chunk = blockIdx.y;
    // (with an in-launch finish the partials are the last arriver's payload: write-through stores, vbnn_last_arriver)
    auto put = [&](float* p, const accv& t) {
        if (!fin_tickets) { *reinterpret_cast<accv*>(p) = t; return; }
#pragma unroll
        for (int e = 0; e < U; e += 2) vbnn_store_wt2(p + e, t[e], t[e + 1]);
    };
    auto meet = [&](const accv (&q)[4], int nq, auto&& emit) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < nq) red[k][wave][lane] = q[k];
        __syncthreads();
        if (wave < nq) {
            accv t = red[wave][0][lane];
#pragma unroll
            for (int w = 1; w < HEAD_SW; ++w) t += red[wave][w][lane];
            emit(wave, t);
        }
    };
#pragma unroll
    for (int c4 = 0; c4 < CP; c4 += 4) {
        accv q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < U; ++e) q[k][e] = accw[c4 + k][e];
        meet(q, 4, [&](int k, accv t) {
            if (partial_w && c4 + k < C) put(partial_w + ((chunk * C + c4 + k) * H + c0), t);
        });
    }
    {
        accv q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < U; ++e) q[k][e] = (k == 0) ? accb[e] : ((k == 1 && e == 0) ? accg : 0.f);
        meet(q, 2, [&](int k, accv t) {
            // (k is wave-uniform) lane c holds the even rows' sum of class c, lane 12 + c the odd rows': even + odd
            const float odd = __shfl(t[0], (lane + CP) & 63, 64);
            if (k == 0) { if (partial_bp) put(partial_bp + (chunk * H + c0), t); }
            else if (partial_b && blockIdx.x == 0 && lane < CP && lane < C) {
                if (fin_tickets) vbnn_store_wt(&partial_b[chunk * C + lane], t[0] + odd); else partial_b[chunk * C + lane] = t[0] + odd;
            }
        });
    }
    // ---- in-launch finish (as k_head_backward's): the LAST of the column block's row-chunk workgroups to arrive adds the chunks' partials
    // in chunk order -- head_sum_partials, the finish kernel's order and arithmetic: the same bits -- instead of a ~5 us launch behind
    // 256 workgroups (the head's forward made the same trade at the same grid size)
    if (!fin_tickets) return;
    __shared__ int fin_last;
    if (!vbnn_last_arriver(fin_tickets + blockIdx.x, gridDim.y, &fin_last)) return;
    const int R = (int)gridDim.y;
    const int64_t nw = (int64_t)C * H;
    constexpr int CB = 64 * U;
    for (int k = tid; k < C * CB + CB; k += 64 * HEAD_SW) {         // the block's C x CB gradWeight entries, then its CB gradBias_prev entries
        const bool is_w = k < C * CB;
        const int64_t i = cb0 + (is_w ? k % CB : k - C * CB);
        float* dst = is_w ? (fin_gw ? fin_gw + (int64_t)(k / CB) * H + i : nullptr) : (fin_gbp ? fin_gbp + i : nullptr);
        if (!dst) continue;
        const float* src = is_w ? partial_w + (int64_t)(k / CB) * H + i : partial_bp + i;
        const float tot = head_sum_partials(src, is_w ? nw : H, R);
        *dst = (fin_accumulate ? *dst : 0.f) + tot;
    }
    if (blockIdx.x == 0 && tid < C && fin_gb) {
        const float tot = head_sum_partials(partial_b + tid, C, R);
        fin_gb[tid] = (fin_accumulate ? fin_gb[tid] : 0.f) + tot;
    }
}

// stage 2: k over C * H (gradWeight) followed by H (gradBias_prev); chunk partials added in chunk order
__global__ __launch_bounds__(256) void k_head_backward_finish(const float* __restrict__ partial_w, const float* __restrict__ partial_b,
                                                              const float* __restrict__ partial_bp, int R, int64_t H, int C,
                                                              int accumulate, float* gradWeight, float* gradBias,
                                                              float* gradBias_prev) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nw = (int64_t)C * H;
    const bool is_w = k < nw;
    if ((is_w && gradWeight) || (!is_w && gradBias_prev && k < nw + H)) {
        const float* src = is_w ? partial_w + k : partial_bp + (k - nw);
        const int64_t stride = is_w ? nw : H;
        float tot = 0.f;
        int r = 0;
        for (; r + 8 <= R; r += 8) {                  // eight independent loads in flight, summed in chunk order
            float p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = src[(int64_t)(r + u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) tot += p[u];
        }
        for (; r < R; ++r) tot += src[(int64_t)r * stride];
        float* dst = is_w ? gradWeight + k : gradBias_prev + (k - nw);
        *dst = (accumulate ? *dst : 0.f) + tot;
    }
    if (blockIdx.x == 0 && threadIdx.x < C && gradBias) {
        float tot = 0.f;
        for (int r = 0; r < R; ++r) tot += partial_b[r * C + threadIdx.x];
        gradBias[threadIdx.x] = (accumulate ? gradBias[threadIdx.x] : 0.f) + tot;
    }
}

// ------------------------------------------------------------------------------------------ forward + backward, one launch
// The fp32 launch-bound configurations (784-400-400-10 at batch 256: 256 x 400 activations) spend 7 + 13.5 us in the two
// head kernels above -- two launches, two drains, 16 resp. 28 workgroups, g_logits out and back -- of a 57 us step. Here
// workgroup (cb, rb) owns rows [16 rb, 16 rb + 16) x hidden units [64 cb, 64 cb + 64):
//   1. it computes the logits of its 16 rows over ALL of H exactly as k_head_forward does (same K split over the 8 waves,
//      same MFMA order: the same bits) -- redundantly in every cb, 1 MFLOP in all, instead of waiting for another
//      workgroup's; log-softmax, loss, arg-max and d(loss)/d(logits) in wave 0's registers; only cb = 0 stores them and
//      takes part in the loss's last-arriver sum;
//   2. with d(loss)/d(logits) of its rows in LDS it forms its 16 x 64 tile of the gradInput (through the ReLU, times r) from
//      h, r (loaded before step 1) and the 64 columns of w3, stores g_prev / gv_prev, and adds its rows' contributions to
//      gradWeight (C x 64), gradBias_prev (64) and gradBias (C; cb = 0) in row order;
//   3. the LAST of a column block's row-block workgroups to arrive adds the partials in row-block order (no float atomics).
template <int DUMMY = 0>
__global__ __launch_bounds__(64 * HEAD_FW) void k_head_step_f32(
    const float* __restrict__ h, int64_t ld_h, const float* __restrict__ w3, int64_t ld_w, const float* __restrict__ bias,
    const int32_t* __restrict__ target, int64_t N, int64_t H, int64_t Hp, int C, float inv_n, int64_t rpd, float* out, float* g_logits,
    float* logits, double* loss_sum, int32_t* correct, int accumulate, double* part_loss, int32_t* part_corr,
    int relu_mask, const float* __restrict__ r_prev, int64_t ld_r, float* g_prev, float* gv_prev, int64_t ld_gp,
    float* __restrict__ partial_w /* [R][C][H] */, float* __restrict__ partial_b /* [R][C] */, float* __restrict__ partial_bp /* [R][H] */,
    unsigned* fin_tickets, float* gradWeight, float* gradBias, float* gradBias_prev) {
    constexpr int NT = 64 * HEAD_FW;
    __shared__ f32x4 part[HEAD_FW][64];
    __shared__ double red_l[HEAD_FW];
    __shared__ int red_c[HEAD_FW];
    __shared__ int last;
    __shared__ float gs[16][HEAD_CMAX + 1];                  // d(loss)/d(logits) of the block's rows (0 for rows >= N, classes >= C)
    __shared__ float ws[HEAD_CMAX][64];                      // the block's 64 columns of w3
    __shared__ float hs[16][65], ps[16][65];                 // h and g_prev tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n0 = (int64_t)blockIdx.y * 16, c0 = (int64_t)blockIdx.x * 64;
    const bool first_cb = blockIdx.x == 0;
    const int R = (int)gridDim.y;

    // ---- step 2's operands, requested before step 1 needs anything: thread (row = tid >> 5, columns 2 (tid & 31), + 1)
    const int prow = tid >> 5, pcol = (tid & 31) * 2;
    const int64_t pn = n0 + prow, pi = c0 + pcol;
    const bool p_ok = pn < N && pi < H;                      // (H is even: the pair is in or out together)
    float2 hv = {0.f, 0.f}, rv = {0.f, 0.f};
    if (p_ok) {
        hv = *reinterpret_cast<const float2*>(h + pn * ld_h + pi);
        if (r_prev) rv = *reinterpret_cast<const float2*>(r_prev + pn * ld_r + pi);
    }
    float wv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int k = tid + NT * u, wc = k >> 6, wi = k & 63;
        wv[u] = (wc < C && c0 + wi < H) ? w3[(int64_t)wc * ld_w + c0 + wi] : 0.f;
    }

    // ---- step 1: k_head_forward's logits of rows n0 .. n0 + 15 (M = class, N = minibatch row; lane (q, c16): classes 4q .. 4q + 3 of row n0 + c16)
    constexpr int KE = 16, CE = 4;
    const int q = lane >> 4, c16 = lane & 15;
    const float* hp = h + min(n0 + c16, N - 1) * ld_h + q * CE;
    const float* wp = w3 + (int64_t)min(c16, C - 1) * ld_w + q * CE;
    const int64_t ksteps = Hp / KE;
    const int64_t k_lo = ksteps * wave / HEAD_FW, k_hi = ksteps * (wave + 1) / HEAD_FW;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int64_t ks = k_lo;
    for (; ks + 8 <= k_hi; ks += 8) {
        f32x4 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a[u] = *reinterpret_cast<const f32x4*>(wp + (ks + u) * KE);
            b[u] = *reinterpret_cast<const f32x4*>(hp + (ks + u) * KE);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = mfma_step<float>(a[u], b[u], acc);
    }
    for (; ks + 2 <= k_hi; ks += 2) {
        f32x4 a[2], b[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            a[u] = *reinterpret_cast<const f32x4*>(wp + (ks + u) * KE);
            b[u] = *reinterpret_cast<const f32x4*>(hp + (ks + u) * KE);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) acc = mfma_step<float>(a[u], b[u], acc);
    }
    for (; ks < k_hi; ++ks) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(wp + ks * KE);
        const f32x4 b = *reinterpret_cast<const f32x4*>(hp + ks * KE);
        acc = mfma_step<float>(a, b, acc);
    }
    part[wave][lane] = acc;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int k = tid + NT * u;
        ws[k >> 6][k & 63] = wv[u];
    }
    __syncthreads();
    if (wave == 0) {
        f32x4 s = part[0][lane];
#pragma unroll
        for (int w = 1; w < HEAD_FW; ++w) s += part[w][lane];
        const int64_t n = n0 + c16;
        const bool row_ok = n < N;
        float lg[4];
        float mx = -INFINITY; int arg = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * q + j;
            lg[j] = (c < C) ? s[j] + (bias ? bias[c] : 0.f) : -INFINITY;
            if (lg[j] > mx) { mx = lg[j]; arg = c; }
        }
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float om = __shfl_xor(mx, off, 64);
            const int oa = __shfl_xor(arg, off, 64);
            if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
        }
        float se = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * q + j < C) se += expf(lg[j] - mx);
        se += __shfl_xor(se, 16, 64);
        se += __shfl_xor(se, 32, 64);
        const float lse = mx + logf(se);
        const int t = row_ok ? min(max(target[rpd > 0 ? n % rpd : n], 0), C - 1) : 0;
        double loss_acc = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * q + j;
            float gl = 0.f;
            if (row_ok && c < C) {
                const float o = lg[j] - lse;
                gl = (expf(o) - (c == t ? 1.0f : 0.0f)) * inv_n;
                if (first_cb) {
                    if (logits) logits[n * C + c] = lg[j];
                    if (out) out[n * C + c] = o;
                    if (g_logits) g_logits[n * C + c] = gl;
                }
                if (c == t) loss_acc -= (double)o * (double)inv_n;
            }
            gs[c16][c] = gl;
        }
        if (first_cb) {
            int corr = (row_ok && q == 0 && arg == t) ? 1 : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { loss_acc += __shfl_xor(loss_acc, off, 64); corr += __shfl_xor(corr, off, 64); }
            if (lane == 0) { vbnn_store_wt(&part_loss[blockIdx.y], loss_acc); vbnn_store_wt(&part_corr[blockIdx.y], (int32_t)corr); }
        }
    }
    __syncthreads();

    // ---- step 2: the block's tile of the gradInput, and its rows' terms of the three sums
    {
        float gx0 = 0.f, gx1 = 0.f;
#pragma unroll
        for (int c = 0; c < HEAD_CMAX; ++c) {
            if (c < C) { gx0 += gs[prow][c] * ws[c][pcol]; gx1 += gs[prow][c] * ws[c][pcol + 1]; }
        }
        const float gp0 = (relu_mask && !(hv.x > 0.f)) ? 0.f : gx0, gp1 = (relu_mask && !(hv.y > 0.f)) ? 0.f : gx1;
        if (p_ok) {
            if (g_prev) *reinterpret_cast<float2*>(g_prev + pn * ld_gp + pi) = float2{gp0, gp1};
            if (gv_prev) *reinterpret_cast<float2*>(gv_prev + pn * ld_gp + pi) = float2{gp0 * rv.x, gp1 * rv.y};
        }
        hs[prow][pcol] = hv.x; hs[prow][pcol + 1] = hv.y;            // (0 outside the matrix)
        ps[prow][pcol] = p_ok ? gp0 : 0.f; ps[prow][pcol + 1] = p_ok ? gp1 : 0.f;
    }
    __syncthreads();
    for (int k = tid; k < (C + 1) * 64; k += NT) {                   // C rows of gradWeight, then gradBias_prev
        const int c = k >> 6, col = k & 63;
        const int64_t i = c0 + col;
        float tot = 0.f;
        if (c < C) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) tot += gs[rr][c] * hs[rr][col];
        } else {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) tot += ps[rr][col];
        }
        if (i < H) {
            if (c < C) { if (partial_w) vbnn_store_wt(&partial_w[((int64_t)blockIdx.y * C + c) * H + i], tot); }
            else if (partial_bp) vbnn_store_wt(&partial_bp[(int64_t)blockIdx.y * H + i], tot);
        }
    }
    if (first_cb && partial_b && tid < C) {
        float tot = 0.f;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) tot += gs[rr][tid];
        vbnn_store_wt(&partial_b[(int64_t)blockIdx.y * C + tid], tot);
    }

    // ---- step 3: the column block's last arriver adds the R partials in row-block order; column block 0's also settles the
    // loss and the hit count (its R workgroups are the ones that stored those partials: one ticket round for everything)
    if (!vbnn_last_arriver(fin_tickets + blockIdx.x, (unsigned)R, &last)) return;
    const int64_t nw = (int64_t)C * H;
    for (int k = tid; k < (C + 1) * 64; k += NT) {
        const bool is_w = k < C * 64;
        const int64_t i = c0 + (k & 63);
        if (i >= H) continue;
        float* dst = is_w ? (gradWeight ? gradWeight + (int64_t)(k >> 6) * H + i : nullptr) : (gradBias_prev ? gradBias_prev + i : nullptr);
        if (!dst) continue;
        const float* src = is_w ? partial_w + (int64_t)(k >> 6) * H + i : partial_bp + i;
        const float tot = head_sum_partials(src, is_w ? nw : H, R);
        *dst = (accumulate ? *dst : 0.f) + tot;
    }
    if (!first_cb) return;
    if (tid < C && gradBias) {
        const float tot = head_sum_partials(partial_b + tid, C, R);
        gradBias[tid] = (accumulate ? gradBias[tid] : 0.f) + tot;
    }
    double ls = 0.0;
    int cs = 0;
    for (int b = tid; b < R; b += NT) { ls += vbnn_load_wt(&part_loss[b]); cs += vbnn_load_wt(&part_corr[b]); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { ls += __shfl_xor(ls, off, 64); cs += __shfl_xor(cs, off, 64); }
    if (lane == 0) { red_l[wave] = ls; red_c[wave] = cs; }
    __syncthreads();
    if (tid == 0) {
        double lt = 0.0;
        int ct = 0;
#pragma unroll
        for (int w = 0; w < HEAD_FW; ++w) { lt += red_l[w]; ct += red_c[w]; }
        if (loss_sum) loss_sum[0] = (accumulate ? loss_sum[0] : 0.0) + lt;
        if (correct) correct[0] = (accumulate ? correct[0] : 0) + ct;
    }
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" int vbnn_head_forward(vbnn_ctx* ctx, int dtype, const void* h, int64_t ld_h, const void* w3, int64_t ld_w,
                                 const float* bias, const int32_t* target, int64_t N, int64_t H, int64_t C, float inv_n,
                                 float* logits, float* out, float* g_logits, int accumulate, double* loss_sum_dev,
                                 int32_t* correct_dev, int64_t rows_per_draw) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(rows_per_draw >= 0, "rows_per_draw");
    VBNN_REQUIRE(ctx && h && w3 && target && g_logits, "null argument");
    VBNN_REQUIRE(N > 0 && H > 0 && C > 0 && C <= HEAD_CMAX, "shape (C <= 16)");
    VBNN_REQUIRE(ld_h % VBNN_KPAD == 0 && ld_w % VBNN_KPAD == 0 && ld_h >= H && ld_w >= H, "h and w3 must be packed operands");
    VBNN_REQUIRE((((uintptr_t)h | (uintptr_t)w3) & 15u) == 0, "operands must be 16-byte aligned");
    const unsigned nb = (unsigned)((N + 15) / 16);
    VBNN_REQUIRE((size_t)nb * 2 <= ctx->scratch_doubles, "minibatch too large for the reduction scratch");
    double* part_loss = ctx->scratch;                                       // [nb] doubles, then [nb] ints
    int32_t* part_corr = reinterpret_cast<int32_t*>(ctx->scratch + nb);
    unsigned* ticket = ctx->counters + VBNN_CNT_HEAD_FWD;
    if (dtype == VBNN_F32) {
        const int64_t Hp = (H + 15) / 16 * 16;
        hipLaunchKernelGGL(k_head_forward<float>, dim3(nb), dim3(64 * HEAD_FW), 0, ctx->stream, (const float*)h, ld_h, (const float*)w3,
                           ld_w, bias, target, N, Hp, (int)C, inv_n, out, g_logits, logits, loss_sum_dev, correct_dev, accumulate, part_loss,
                           part_corr, ticket, rows_per_draw);
    } else if (dtype == VBNN_BF16) {
        const int64_t Hp = (H + 31) / 32 * 32;
        hipLaunchKernelGGL(k_head_forward<bf16_t>, dim3(nb), dim3(64 * HEAD_FW), 0, ctx->stream, (const bf16_t*)h, ld_h,
                           (const bf16_t*)w3, ld_w, bias, target, N, Hp, (int)C, inv_n, out, g_logits, logits, loss_sum_dev,
                           correct_dev, accumulate, part_loss, part_corr, ticket, rows_per_draw);
    } else { vbnn_set_error("unsupported dtype %d", dtype); return VBNN_ERR_UNSUPPORTED; }
    return vbnn_check_launch("k_head_forward");
    VBNN_API_END
}

extern "C" int vbnn_head_forward_slots(vbnn_ctx* ctx, const float* slots, int64_t n_slots, const float* bias, const int32_t* target,
                                       int64_t N, int64_t C, float inv_n, float* logits, float* out, float* g_logits, int accumulate,
                                       double* loss_sum_dev, int32_t* correct_dev, int64_t rows_per_draw) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && slots && target && g_logits, "null argument");
    VBNN_REQUIRE(rows_per_draw >= 0, "rows_per_draw");
    VBNN_REQUIRE(N > 0 && C > 0 && C <= HEAD_CMAX && n_slots > 0 && n_slots < (1 << 20), "shape (C <= 16)");
    VBNN_REQUIRE(((uintptr_t)slots & 15u) == 0, "slots must be 16-byte aligned");
    const unsigned nb = (unsigned)((N + 15) / 16);
    VBNN_REQUIRE((size_t)nb * 2 <= ctx->scratch_doubles, "minibatch too large for the reduction scratch");
    double* part_loss = ctx->scratch;                                       // [nb] doubles, then [nb] ints
    int32_t* part_corr = reinterpret_cast<int32_t*>(ctx->scratch + nb);
    hipLaunchKernelGGL(k_head_from_slots, dim3(nb), dim3(64), 0, ctx->stream, slots, (int)n_slots, bias, target, N, (int)C, inv_n, out,
                       g_logits, logits, loss_sum_dev, correct_dev, accumulate, part_loss, part_corr, ctx->counters + VBNN_CNT_HEAD_FWD,
                       rows_per_draw);
    return vbnn_check_launch("k_head_from_slots");
    VBNN_API_END
}

static int g_head_inline_finish = 1;      // A/B: VBNN_HEAD_INLINE_FINISH=0 in the environment keeps the separate finish kernel
template <typename T>
static int head_backward_t(vbnn_ctx* ctx, const T* h, int64_t ld_h, const T* w3, int64_t ld_w, const float* g_logits, int64_t N,
                           int64_t H, int64_t C, int accumulate, float* gradWeight, float* gradBias, float* gradBias_prev,
                           int relu_mask, const void* r_prev_any, int64_t ld_r_prev, int r_prev_packed, T* g_prev, T* gv_prev,
                           int64_t ld_gp, T* gT_prev, T* gvT_prev, int64_t ld_gpT) {
    const float* r_prev = r_prev_packed ? nullptr : (const float*)r_prev_any;
    const T* r_prev_t = r_prev_packed ? (const T*)r_prev_any : nullptr;
    const bool sums = gradWeight || gradBias || gradBias_prev;
    if (!sums && !g_prev && !gT_prev) return VBNN_OK;
    static const bool env_read = [] { const char* e = getenv("VBNN_HEAD_INLINE_FINISH"); if (e && e[0] == '0') g_head_inline_finish = 0; return true; }();
    (void)env_read;
    // row chunks: about 1024 blocks (several per CU, so one's loads sit beside another's LDS work), bounded by the
    // reduction scratch: per chunk C x H + C + H partial sums
    const int64_t tiles_c = (H + 63) / 64, tiles_r = (N + 63) / 64;
    const int64_t per_chunk = C * H + C + H;
    const int64_t cap = (int64_t)(ctx->scratch_doubles * 2) / per_chunk;
    VBNN_REQUIRE(!sums || cap >= 1, "hidden size too large for the reduction scratch");
    // (r04: two workgroups per CU -- ONE round of the grid: 512 workgroups on 256 CUs. A launch of 1024 was two rounds, each workgroup
    // paying its prologue (final-weight columns, first tiles) twice over: 37.2 -> 34.1 us with the finish kernel at 4096 x 4096,
    // tools/time_head.py; 256: 47, 2048: 39. A rewrite of the loop around transpose reads and vector LDS stores -- a quarter of the
    // LDS instructions -- measured NO faster (36.8 us): the launch is bound by its 134 MB of mixed reads and writes, not by LDS.)
    int64_t R = (2 * (int64_t)vbnn_cu_count() + tiles_c - 1) / tiles_c;
    static const int env_blocks = [] { const char* e = getenv("VBNN_HEAD_BLOCKS"); return e ? atoi(e) : 0; }();      // A/B: target workgroup count
    if (env_blocks > 0) R = (env_blocks + tiles_c - 1) / tiles_c;
    if (R > tiles_r) R = tiles_r;
    if (sums && R > cap) R = cap;
    if (R < 1) R = 1;
    VBNN_REQUIRE(R <= 65535, "too many row chunks");
    const int rows_per_chunk = (int)((tiles_r + R - 1) / R) * 64;
    R = (N + rows_per_chunk - 1) / rows_per_chunk;
    float* partial_w = reinterpret_cast<float*>(ctx->scratch);
    float* partial_b = partial_w + R * C * H;
    float* partial_bp = partial_b + R * C;
    const bool vec = (ld_h % 8 == 0) && (!r_prev_any || ld_r_prev % 8 == 0) && (!g_prev || ld_gp % 8 == 0) &&
                     (!(gT_prev || gvT_prev) || ld_gpT % 8 == 0) &&
                     ((((uintptr_t)h | (uintptr_t)r_prev_any | (uintptr_t)g_prev | (uintptr_t)gv_prev | (uintptr_t)gT_prev |
                        (uintptr_t)gvT_prev) & 15u) == 0);
    const dim3 grid((unsigned)tiles_c, (unsigned)R);
    float* pw = (gradWeight || gradBias_prev) ? partial_w : nullptr;
    float* pb = gradBias ? partial_b : nullptr;
    float* pbp = gradBias_prev ? partial_bp : nullptr;
    const bool full = vec && (H % 64 == 0) && (N % 64 == 0) && N * ld_h < (1ll << 31) && N * ld_gp < (1ll << 31) &&
                      H * ld_gpT < (1ll << 31) && (!r_prev_any || N * ld_r_prev < (1ll << 31));
    // few workgroups (one partial wave of the chip): the row-chunk partials are finished INSIDE the launch by each column
    // tile's last arriver (same order, same bits as the finish kernel); otherwise a finish kernel follows (a last
    // arriver's serial tail behind thousands of streaming workgroups costs more than the launch it saves, r01)
    const bool inline_fin = sums && tiles_c * R <= vbnn_cu_count() && tiles_c <= VBNN_CNT_TILES_MAX && g_head_inline_finish;
    unsigned* ft = inline_fin ? ctx->counters + VBNN_CNT_TILES : nullptr;
    // the streaming form (k_head_backward_stream): bf16, whole tiles of 256 hidden units x 32 rows, C <= 12, r packed or absent, no
    // transposed outputs -- the wide configurations. vbnn_debug_set(VBNN_DEBUG_HEAD_BACKWARD, 0 / 1) or VBNN_HEAD_STREAM=0: the tile form / this one (A/B, tests).
    static const bool env_stream = [] { const char* e = getenv("VBNN_HEAD_STREAM"); return !(e && e[0] == '0'); }();
    if constexpr (sizeof(T) == 2) {
        const bool want = g_head_stream == 1 || (g_head_stream == -1 && env_stream && !inline_fin);
        if (want && full && C <= 12 && !r_prev && g_prev && !gT_prev && !gvT_prev && H % 128 == 0 && N % 32 == 0) {
            // 2 hidden units per lane, 16 waves per workgroup, one workgroup per CU. (Lab, same box, head backward + finish per call: the tile
            // form 36.0 / 35.7 us, this one 32.1 / 32.0; 4 units x 8 waves 34.4 / 30.3 -- 142 registers, three waves per SIMD --; 2 x 8
            // with two workgroups per CU 33.1 / 34.3.)
            constexpr int SU = 2, SWV = 16;
            if (H % (64 * SU) == 0 && N % (2 * SWV) == 0) {
            const int64_t cb = H / (64 * SU);
            int64_t Rs = ((int64_t)vbnn_cu_count() + cb - 1) / cb;
            if (Rs < 1) Rs = 1;
            if (sums && Rs > cap) Rs = cap;
            const int rpc = (int)(((N + Rs - 1) / Rs + 2 * SWV - 1) / (2 * SWV) * (2 * SWV));
            Rs = (N + rpc - 1) / rpc;
            float* spw = reinterpret_cast<float*>(ctx->scratch);
            float* spb = spw + Rs * C * H;
            float* spbp = spb + Rs * C;
#define VBNN_HS_LAUNCH(UU, WW) hipLaunchKernelGGL((k_head_backward_stream<UU, WW>), dim3((unsigned)cb, (unsigned)Rs), dim3(64 * WW), 0, ctx->stream, (const bf16_t*)h, ld_h, \
                               (const bf16_t*)w3, ld_w, g_logits, N, H, (int)C, relu_mask, (const bf16_t*)r_prev_t, ld_r_prev, (bf16_t*)g_prev,  \
                               (bf16_t*)gv_prev, ld_gp, rpc, (gradWeight || gradBias_prev) ? spw : nullptr, gradBias ? spb : nullptr,         \
                               gradBias_prev ? spbp : nullptr, sft, accumulate, gradWeight, gradBias, gradBias_prev)
            // the in-launch finish where the tickets reach (one per column block); VBNN_HEAD_INLINE_FINISH=0: the finish kernel (A/B)
            unsigned* sft = (sums && g_head_inline_finish && cb <= VBNN_CNT_TILES_MAX) ? ctx->counters + VBNN_CNT_TILES : nullptr;
            VBNN_HS_LAUNCH(SU, SWV);
#undef VBNN_HS_LAUNCH
            if (sums && !sft) {
                const int64_t outs = C * H + (gradBias_prev ? H : 0);
                hipLaunchKernelGGL(k_head_backward_finish, dim3((unsigned)((outs + 255) / 256)), dim3(256), 0, ctx->stream, spw, spb, spbp,
                                   (int)Rs, H, (int)C, accumulate, gradWeight, gradBias, gradBias_prev);
            }
            return vbnn_check_launch("k_head_backward_stream");
            }
        }
    }
    if (full && C <= 12 && !r_prev)
        hipLaunchKernelGGL((k_head_backward<T, true, true, 12>), grid, dim3(256), 0, ctx->stream, h, ld_h, w3, ld_w, g_logits, N, H, (int)C,
                           relu_mask, r_prev, r_prev_t, ld_r_prev, g_prev, gv_prev, ld_gp, gT_prev, gvT_prev, ld_gpT, rows_per_chunk,
                           pw, pb, pbp, ft, accumulate, gradWeight, gradBias, gradBias_prev);
    else if (full)
        hipLaunchKernelGGL((k_head_backward<T, true, true>), grid, dim3(256), 0, ctx->stream, h, ld_h, w3, ld_w, g_logits, N, H, (int)C,
                           relu_mask, r_prev, r_prev_t, ld_r_prev, g_prev, gv_prev, ld_gp, gT_prev, gvT_prev, ld_gpT, rows_per_chunk,
                           pw, pb, pbp, ft, accumulate, gradWeight, gradBias, gradBias_prev);
    else if (vec)
        hipLaunchKernelGGL((k_head_backward<T, true, false>), grid, dim3(256), 0, ctx->stream, h, ld_h, w3, ld_w, g_logits, N, H, (int)C,
                           relu_mask, r_prev, r_prev_t, ld_r_prev, g_prev, gv_prev, ld_gp, gT_prev, gvT_prev, ld_gpT, rows_per_chunk,
                           pw, pb, pbp, ft, accumulate, gradWeight, gradBias, gradBias_prev);
    else
        hipLaunchKernelGGL((k_head_backward<T, false, false>), grid, dim3(256), 0, ctx->stream, h, ld_h, w3, ld_w, g_logits, N, H, (int)C,
                           relu_mask, r_prev, r_prev_t, ld_r_prev, g_prev, gv_prev, ld_gp, gT_prev, gvT_prev, ld_gpT, rows_per_chunk,
                           pw, pb, pbp, ft, accumulate, gradWeight, gradBias, gradBias_prev);
    if (sums && !inline_fin) {
        const int64_t outs = C * H + (gradBias_prev ? H : 0);
        hipLaunchKernelGGL(k_head_backward_finish, dim3((unsigned)((outs + 255) / 256)), dim3(256), 0, ctx->stream, partial_w,
                           partial_b, partial_bp, (int)R, H, (int)C, accumulate, gradWeight, gradBias, gradBias_prev);
    }
    return vbnn_check_launch("k_head_backward");
}

extern "C" int vbnn_head_backward(vbnn_ctx* ctx, int dtype, const void* h, int64_t ld_h, const void* w3, int64_t ld_w,
                                  const float* g_logits, int64_t N, int64_t H, int64_t C, int accumulate, float* gradWeight,
                                  float* gradBias, float* gradBias_prev, int relu_mask, const void* r_prev, int64_t ld_r_prev,
                                  int r_prev_packed, void* g_prev, void* gv_prev, int64_t ld_gp, void* gT_prev, void* gvT_prev,
                                  int64_t ld_gpT) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && h && w3 && g_logits, "null argument");
    vbnn_cu_scope plan(ctx);
    VBNN_REQUIRE(N > 0 && H > 0 && C > 0 && C <= HEAD_CMAX, "shape (C <= 16)");
    VBNN_REQUIRE(ld_h % VBNN_KPAD == 0 && ld_h >= H && ld_w >= H, "h and w3 must be packed operands");
    VBNN_REQUIRE(!gv_prev || (g_prev && r_prev), "gv_prev needs g_prev and r_prev");
    VBNN_REQUIRE(!(g_prev || gv_prev) || ld_gp >= H, "ld_gp");
    VBNN_REQUIRE(!(gT_prev || gvT_prev) || ld_gpT >= N, "ld_gpT");
    if (dtype == VBNN_F32)
        return head_backward_t<float>(ctx, (const float*)h, ld_h, (const float*)w3, ld_w, g_logits, N, H, C, accumulate, gradWeight,
                                      gradBias, gradBias_prev, relu_mask, r_prev, ld_r_prev, r_prev_packed, (float*)g_prev, (float*)gv_prev,
                                      ld_gp, (float*)gT_prev, (float*)gvT_prev, ld_gpT);
    if (dtype == VBNN_BF16)
        return head_backward_t<bf16_t>(ctx, (const bf16_t*)h, ld_h, (const bf16_t*)w3, ld_w, g_logits, N, H, C, accumulate,
                                       gradWeight, gradBias, gradBias_prev, relu_mask, r_prev, ld_r_prev, r_prev_packed, (bf16_t*)g_prev,
                                       (bf16_t*)gv_prev, ld_gp, (bf16_t*)gT_prev, (bf16_t*)gvT_prev, ld_gpT);
    vbnn_set_error("unsupported dtype %d", dtype);
    return VBNN_ERR_UNSUPPORTED;
    VBNN_API_END
}

static int g_head_step = 1;               // A/B: VBNN_HEAD_STEP=0 in the environment keeps the two launches

extern "C" int vbnn_head_forward_backward(vbnn_ctx* ctx, int dtype, const vbnn_head_args* a) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && a, "null argument");
    static const bool env_read = [] { const char* e = getenv("VBNN_HEAD_STEP"); if (e && e[0] == '0') g_head_step = 0; return true; }();
    (void)env_read;
    const int64_t N = a->N, H = a->H, C = a->C;
    const int64_t R = (N + 15) / 16, tiles_c = (H + 63) / 64;
    const bool sums = a->gradWeight || a->gradBias || a->gradBias_prev;
    // scratch: [R] loss partials (double) + [R] hit partials (int), then the row-block partials of the three sums
    const int64_t head_floats = 4 * R, per_block = C * H + C + H;
    const bool one_launch =
        g_head_step && dtype == VBNN_F32 && a->h && a->w3 && a->target && a->g_logits && N > 0 && H > 0 && C > 0 && C <= HEAD_CMAX &&
        a->rows_per_draw >= 0 && !a->gT_prev && !a->gvT_prev && N * H <= (1ll << 20) && H % 2 == 0 && tiles_c <= VBNN_CNT_TILES_MAX &&
        R <= 65535 && a->ld_h % VBNN_KPAD == 0 && a->ld_w % VBNN_KPAD == 0 && a->ld_h >= H && a->ld_w >= H &&
        (((uintptr_t)a->h | (uintptr_t)a->w3) & 15u) == 0 && (sums || a->g_prev) && (!a->gv_prev || (a->g_prev && a->r_prev)) &&
        (!a->r_prev || (a->ld_r_prev % 2 == 0 && a->ld_r_prev >= H && ((uintptr_t)a->r_prev & 7u) == 0)) &&
        (!a->g_prev || (a->ld_gp % 2 == 0 && a->ld_gp >= H && (((uintptr_t)a->g_prev | (uintptr_t)a->gv_prev) & 7u) == 0)) &&
        (int64_t)(ctx->scratch_doubles * 2) >= head_floats + R * per_block;
    if (!one_launch || a->logit_slots) {
        // the forward half: from the partial logits the last VB layer's forward left (vbnn_fwd_args.head_slots), or from h
        const int st = a->logit_slots
                           ? vbnn_head_forward_slots(ctx, a->logit_slots, a->n_slots, a->bias, a->target, N, C, a->inv_n, a->logits, a->out,
                                                     a->g_logits, a->accumulate, a->loss_sum_dev, a->correct_dev, a->rows_per_draw)
                           : vbnn_head_forward(ctx, dtype, a->h, a->ld_h, a->w3, a->ld_w, a->bias, a->target, N, H, C, a->inv_n, a->logits, a->out,
                                         a->g_logits, a->accumulate, a->loss_sum_dev, a->correct_dev, a->rows_per_draw);
        if (st != VBNN_OK) return st;
        return vbnn_head_backward(ctx, dtype, a->h, a->ld_h, a->w3, a->ld_w, a->g_logits, N, H, C, a->accumulate, a->gradWeight,
                                  a->gradBias, a->gradBias_prev, a->relu_mask, a->r_prev, a->ld_r_prev, a->r_prev_packed, a->g_prev,
                                  a->gv_prev, a->ld_gp, a->gT_prev, a->gvT_prev, a->ld_gpT);
    }
    double* part_loss = ctx->scratch;
    int32_t* part_corr = reinterpret_cast<int32_t*>(ctx->scratch + R);
    float* partial_w = reinterpret_cast<float*>(ctx->scratch) + head_floats;
    float* partial_b = partial_w + R * C * H;
    float* partial_bp = partial_b + R * C;
    const int64_t Hp = (H + 15) / 16 * 16;
    hipLaunchKernelGGL(k_head_step_f32<0>, dim3((unsigned)tiles_c, (unsigned)R), dim3(64 * HEAD_FW), 0, ctx->stream, (const float*)a->h,
                       a->ld_h, (const float*)a->w3, a->ld_w, a->bias, a->target, N, H, Hp, (int)C, a->inv_n, a->rows_per_draw, a->out,
                       a->g_logits, a->logits, a->loss_sum_dev, a->correct_dev, a->accumulate, part_loss, part_corr, a->relu_mask, (const float*)a->r_prev, a->ld_r_prev, (float*)a->g_prev,
                       (float*)a->gv_prev, a->ld_gp, (a->gradWeight ? partial_w : nullptr), (a->gradBias ? partial_b : nullptr),
                       (a->gradBias_prev ? partial_bp : nullptr), ctx->counters + VBNN_CNT_TILES, a->gradWeight, a->gradBias,
                       a->gradBias_prev);
    return vbnn_check_launch("k_head_step_f32");
    VBNN_API_END
}
