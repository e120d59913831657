// head.hip -- the classifier head of mlp.lua:29-32 for a small class count (C <= 16):
// final nn.Linear (H -> C) + nn.LogSoftMax + nn.ClassNLLCriterion, forward and backward.
//
// With C = 10 these are skinny, HBM-bound passes over the N x H activation (33 MB at the wide
// configuration), not GEMM-shaped work for 256 x 128 tiles:
//   forward   16 rows per workgroup, K split over its 4 waves, one 16x16 MFMA column of logits per wave,
//             partials folded through LDS, log-softmax / loss / arg-max / d(loss)/d(logits) in registers
//   dW        each thread owns 8 consecutive hidden units and C accumulators each, g rows broadcast
//             from LDS; deterministic two-stage reduction over row chunks (no float atomics)
//   dX        64 x 64 tiles, 16-byte loads/stores, the two transposed operand copies written through an
//             LDS transpose so every global access is a full 16-byte-per-lane segment
#include "common.h"
#include "gemm_v1.h"      // Frag<T>, mfma_step<T>

constexpr int HEAD_CMAX = 16;

template <typename T> struct Vec8;             // eight consecutive packed elements
template <> struct Vec8<bf16_t> {
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

// ------------------------------------------------------------------------------------------ forward
// MFMA orientation: M = class (A operand = w3 rows, rows >= C clamped and ignored), N = minibatch row.
// Accumulator layout: lane (q = l >> 4, c = l & 15) holds classes 4q .. 4q+3 of row n0 + c.
template <typename T>
__global__ __launch_bounds__(256) void k_head_forward(const T* __restrict__ h, int64_t ld_h, const T* __restrict__ w3, int64_t ld_w,
                                                      const float* __restrict__ bias, const int32_t* __restrict__ target, int64_t N,
                                                      int64_t Hp /* H padded to the K step */, int C, float inv_n, float* out,
                                                      float* g_logits, float* logits, double* loss_sum, int32_t* correct) {
    constexpr int KE = 64 / (int)sizeof(T);      // K elements per MFMA step (16 bytes per lane x 4 lane groups)
    constexpr int CE = 16 / (int)sizeof(T);
    typedef typename Frag<T>::type frag_t;
    __shared__ f32x4 part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n0 = (int64_t)blockIdx.x * 16;
    const int q = lane >> 4, c16 = lane & 15;
    const T* hp = h + min(n0 + c16, N - 1) * ld_h + q * CE;
    const T* wp = w3 + (int64_t)min(c16, C - 1) * ld_w + q * CE;
    const int64_t ksteps = Hp / KE;
    const int64_t k_lo = ksteps * wave / 4, k_hi = ksteps * (wave + 1) / 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int64_t ks = k_lo;
    for (; ks + 4 <= k_hi; ks += 4) {                 // four K steps of loads in flight (latency-bound otherwise)
        frag_t a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const frag_t*>(wp + (ks + u) * KE);
            b[u] = *reinterpret_cast<const frag_t*>(hp + (ks + u) * KE);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = mfma_step<T>(a[u], b[u], acc);
    }
    for (; ks < k_hi; ++ks) {
        const frag_t a = *reinterpret_cast<const frag_t*>(wp + ks * KE);
        const frag_t b = *reinterpret_cast<const frag_t*>(hp + ks * KE);
        acc = mfma_step<T>(a, b, acc);
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave != 0) return;
    const f32x4 s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
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
    const int t = row_ok ? min(max(target[n], 0), C - 1) : 0;
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
    if (lane == 0) {
        if (loss_sum) atomicAdd(loss_sum, loss_acc);
        if (correct && corr) atomicAdd(correct, corr);
    }
}

// ------------------------------------------------------------------------------------------ dW / dbias
// stage 1: grid (H / 512, R). Thread (cg = tid & 63, sub = tid >> 6) owns hidden units 8 cg .. 8 cg + 7 of the
// block's 512 and the rows n = chunk0 + sub, sub + 4, ... ; the four row lanes are folded through LDS.
template <typename T>
__global__ __launch_bounds__(256) void k_head_dw_partial(const T* __restrict__ h, int64_t ld_h, const float* __restrict__ g,
                                                         int64_t N, int64_t H, int C, int rows_per_chunk,
                                                         float* __restrict__ partial /* [R][C][H] */,
                                                         float* __restrict__ partial_b /* [R][C] */) {
    extern __shared__ __attribute__((aligned(16))) float red[];          // [4][64][C * 8] reduction, then g staging in front
    __shared__ float gs[128][HEAD_CMAX];
    const int cg = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * 512 + cg * 8;
    const int64_t n0 = (int64_t)blockIdx.y * rows_per_chunk, n1 = min(N, n0 + rows_per_chunk);
    float acc[HEAD_CMAX][8];
#pragma unroll
    for (int c = 0; c < HEAD_CMAX; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[c][e] = 0.f;
    float accb = 0.f;
    const bool col_ok = i0 < H;                  // ld_h is padded: a chunk that starts in range is readable in full
    for (int64_t nb = n0; nb < n1; nb += 128) {
        const int rows = (int)min((int64_t)128, n1 - nb);
        __syncthreads();
        for (int k = threadIdx.x; k < rows * C; k += 256) {
            const int rr = k / C, cc = k - rr * C;
            gs[rr][cc] = Elt<T>::from(Elt<T>::to(g[(nb + rr) * C + cc]));    // the operand rounding of the packed GEMM path
        }
        __syncthreads();
        if (col_ok) {
            // four rows in flight per thread: the loop is latency-bound otherwise (4 waves per CU, one 16-byte
            // load per 80 FMAs)
            for (int rr = sub; rr < rows; rr += 16) {
                float hv[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r_ = min(rr + 4 * u, rows - 1);
                    Vec8<T>::load(h + (nb + r_) * ld_h + i0, hv[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (rr + 4 * u < rows) {
#pragma unroll
                        for (int c = 0; c < HEAD_CMAX; ++c)
                            if (c < C) {
                                const float gv = gs[rr + 4 * u][c];
#pragma unroll
                                for (int e = 0; e < 8; ++e) acc[c][e] = fmaf(gv, hv[u][e], acc[c][e]);
                            }
                    }
                }
            }
        }
        // bias gradient (column sums of the UNROUNDED g): 16 row lanes per class, folded at the end in a fixed order.
        // (A single thread per class walking the rows is a 128-deep chain of dependent-latency loads.)
        if (blockIdx.x == 0 && (threadIdx.x & 15) < C)
            for (int rr = threadIdx.x >> 4; rr < rows; rr += 16) accb += g[(nb + rr) * C + (threadIdx.x & 15)];
    }
    // fold the four row lanes: red[sub][cg][c * 8 + e]
    const int stride = C * 8;
#pragma unroll
    for (int c = 0; c < HEAD_CMAX; ++c)
        if (c < C) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(sub * 64 + cg) * stride + c * 8 + e] = acc[c][e];
        }
    __syncthreads();
    for (int k = threadIdx.x; k < 64 * stride; k += 256) {
        const int g_ = k / stride, rem = k - g_ * stride;
        const int c = rem >> 3, e = rem & 7;
        const int64_t i = (int64_t)blockIdx.x * 512 + g_ * 8 + e;
        if (i < H) {
            const float tot = (red[(0 * 64 + g_) * stride + rem] + red[(1 * 64 + g_) * stride + rem]) +
                              (red[(2 * 64 + g_) * stride + rem] + red[(3 * 64 + g_) * stride + rem]);
            partial[((int64_t)blockIdx.y * C + c) * H + i] = tot;
        }
    }
    if (blockIdx.x == 0) {
        __syncthreads();                                   // `red` is free again
        red[threadIdx.x] = accb;                           // [16 row lanes][16 classes]
        __syncthreads();
        if (threadIdx.x < C) {
            float tot = 0.f;
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) tot += red[rg * 16 + threadIdx.x];
            partial_b[(int64_t)blockIdx.y * C + threadIdx.x] = tot;
        }
    }
}
__global__ __launch_bounds__(256) void k_head_dw_finish(const float* __restrict__ partial, const float* __restrict__ partial_b, int R,
                                                        int64_t H, int C, int accumulate, float* gradWeight, float* gradBias) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;       // over C * H
    if (k < (int64_t)C * H) {
        float tot = 0.f;
        int r = 0;
        for (; r + 8 <= R; r += 8) {                  // eight independent loads in flight, summed in chunk order
            float p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = partial[(int64_t)(r + u) * C * H + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) tot += p[u];
        }
        for (; r < R; ++r) tot += partial[(int64_t)r * C * H + k];
        gradWeight[k] = (accumulate ? gradWeight[k] : 0.f) + tot;
    }
    if (blockIdx.x == 0 && threadIdx.x < C && gradBias) {
        float tot = 0.f;
        for (int r = 0; r < R; ++r) tot += partial_b[r * C + threadIdx.x];
        gradBias[threadIdx.x] = (accumulate ? gradBias[threadIdx.x] : 0.f) + tot;
    }
}

// ------------------------------------------------------------------------------------------ dX
//   gx[n][i] = sum_c g[n][c] w3[c][i];  g_prev = gx . [h > 0];  gv_prev = g_prev . r
// 64 x 64 tile per block; thread (row = tid >> 3 (+32), chunk = tid & 7) handles 8 consecutive hidden units.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void k_head_dx(const T* __restrict__ h, int64_t ld_h, const T* __restrict__ w3, int64_t ld_w,
                                                 const float* __restrict__ g, int64_t N, int64_t H, int C, int relu_mask,
                                                 const float* __restrict__ r_prev, const T* __restrict__ r_prev_t, int64_t ld_r,
                                                 T* g_prev, T* gv_prev, int64_t ld_gp, T* gT_prev, T* gvT_prev, int64_t ld_gpT) {
    __shared__ float tg[64][65];
    __shared__ float tv[64][65];
    __shared__ float gs[64][HEAD_CMAX];
    __shared__ __attribute__((aligned(16))) float ws[HEAD_CMAX][64];
    const int64_t tiles_c = (H + 63) / 64;
    const int64_t r0 = (blockIdx.x / tiles_c) * 64, c0 = (blockIdx.x % tiles_c) * 64;
    for (int k = threadIdx.x; k < 64 * C; k += 256) {
        const int rr = k / C, cc = k - rr * C;
        gs[rr][cc] = (r0 + rr < N) ? Elt<T>::from(Elt<T>::to(g[(r0 + rr) * C + cc])) : 0.f;
        const int wc = k / 64, wi = k - wc * 64;
        ws[wc][wi] = (c0 + wi < H) ? Elt<T>::from(w3[(int64_t)wc * ld_w + c0 + wi]) : 0.f;
    }
    __syncthreads();
    const int ch = threadIdx.x & 7;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int rr = (threadIdx.x >> 3) + 32 * p;
        const int64_t n = r0 + rr, i = c0 + ch * 8;
        float gp[8], gvp[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { gp[e] = 0.f; gvp[e] = 0.f; }
        if (n < N && i < H) {
            float gx[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) gx[e] = 0.f;
#pragma unroll
            for (int c = 0; c < HEAD_CMAX; ++c)
                if (c < C) {
                    const float gv = gs[rr][c];
#pragma unroll
                    for (int e = 0; e < 8; ++e) gx[e] = fmaf(gv, ws[c][ch * 8 + e], gx[e]);
                }
            float hv[8], rv[8];
            const bool full = VEC && (i + 8 <= H);
            const bool has_r = r_prev || r_prev_t;
            if (full) {
                Vec8<T>::load(h + n * ld_h + i, hv);
                if (r_prev) Vec8<float>::load(r_prev + n * ld_r + i, rv);
                if (r_prev_t) Vec8<T>::load(r_prev_t + n * ld_r + i, rv);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    hv[e] = (i + e < H) ? Elt<T>::from(h[n * ld_h + i + e]) : 0.f;
                    rv[e] = 0.f;
                    if (r_prev && i + e < H) rv[e] = r_prev[n * ld_r + i + e];
                    if (r_prev_t && i + e < H) rv[e] = Elt<T>::from(r_prev_t[n * ld_r + i + e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                gp[e] = (relu_mask && !(hv[e] > 0.f)) ? 0.f : gx[e];
                gvp[e] = has_r ? gp[e] * rv[e] : 0.f;
            }
            if (full) {
                if (g_prev) Vec8<T>::store(g_prev + n * ld_gp + i, gp);
                if (gv_prev) Vec8<T>::store(gv_prev + n * ld_gp + i, gvp);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (i + e < H) {
                        if (g_prev) g_prev[n * ld_gp + i + e] = Elt<T>::to(gp[e]);
                        if (gv_prev) gv_prev[n * ld_gp + i + e] = Elt<T>::to(gvp[e]);
                    }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { tg[rr][ch * 8 + e] = gp[e]; tv[rr][ch * 8 + e] = gvp[e]; }
    }
    if (!gT_prev && !gvT_prev) return;
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int cc = (threadIdx.x >> 3) + 32 * p;              // hidden unit inside the tile
        const int64_t i = c0 + cc, n = r0 + ch * 8;
        if (i >= H || n >= N) continue;
        float a[8], b[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = tg[ch * 8 + e][cc]; b[e] = tv[ch * 8 + e][cc]; }
        if (VEC && n + 8 <= N) {
            if (gT_prev) Vec8<T>::store(gT_prev + i * ld_gpT + n, a);
            if (gvT_prev) Vec8<T>::store(gvT_prev + i * ld_gpT + n, b);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (n + e < N) {
                    if (gT_prev) gT_prev[i * ld_gpT + n + e] = Elt<T>::to(a[e]);
                    if (gvT_prev) gvT_prev[i * ld_gpT + n + e] = Elt<T>::to(b[e]);
                }
        }
    }
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" int vbnn_head_forward(vbnn_ctx* ctx, int dtype, const void* h, int64_t ld_h, const void* w3, int64_t ld_w,
                                 const float* bias, const int32_t* target, int64_t N, int64_t H, int64_t C, float inv_n,
                                 float* logits, float* out, float* g_logits, double* loss_sum_dev, int32_t* correct_dev) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && h && w3 && target && g_logits, "null argument");
    VBNN_REQUIRE(N > 0 && H > 0 && C > 0 && C <= HEAD_CMAX, "shape (C <= 16)");
    VBNN_REQUIRE(ld_h % VBNN_KPAD == 0 && ld_w % VBNN_KPAD == 0 && ld_h >= H && ld_w >= H, "h and w3 must be packed operands");
    VBNN_REQUIRE((((uintptr_t)h | (uintptr_t)w3) & 15u) == 0, "operands must be 16-byte aligned");
    const unsigned nb = (unsigned)((N + 15) / 16);
    if (dtype == VBNN_F32) {
        const int64_t Hp = (H + 15) / 16 * 16;
        hipLaunchKernelGGL(k_head_forward<float>, dim3(nb), dim3(256), 0, ctx->stream, (const float*)h, ld_h, (const float*)w3,
                           ld_w, bias, target, N, Hp, (int)C, inv_n, out, g_logits, logits, loss_sum_dev, correct_dev);
    } else if (dtype == VBNN_BF16) {
        const int64_t Hp = (H + 31) / 32 * 32;
        hipLaunchKernelGGL(k_head_forward<bf16_t>, dim3(nb), dim3(256), 0, ctx->stream, (const bf16_t*)h, ld_h,
                           (const bf16_t*)w3, ld_w, bias, target, N, Hp, (int)C, inv_n, out, g_logits, logits, loss_sum_dev,
                           correct_dev);
    } else { vbnn_set_error("unsupported dtype %d", dtype); return VBNN_ERR_UNSUPPORTED; }
    return vbnn_check_launch("k_head_forward");
    VBNN_API_END
}

template <typename T>
static int head_backward_t(vbnn_ctx* ctx, const T* h, int64_t ld_h, const T* w3, int64_t ld_w, const float* g_logits, int64_t N,
                           int64_t H, int64_t C, int accumulate, float* gradWeight, float* gradBias, int relu_mask,
                           const void* r_prev_any, int64_t ld_r_prev, int r_prev_packed, T* g_prev, T* gv_prev, int64_t ld_gp,
                           T* gT_prev, T* gvT_prev, int64_t ld_gpT) {
    const float* r_prev = r_prev_packed ? nullptr : (const float*)r_prev_any;
    const T* r_prev_t = r_prev_packed ? (const T*)r_prev_any : nullptr;
    if (gradWeight) {
        const int64_t cap = (int64_t)(ctx->scratch_doubles * 2) / (C * H + C);
        VBNN_REQUIRE(cap >= 1, "hidden size too large for the reduction scratch");
        int64_t R = (N + 127) / 128;
        if (R > 32) R = 32;
        if (R > cap) R = cap;
        const int rows_per_chunk = (int)((N + R - 1) / R);
        R = (N + rows_per_chunk - 1) / rows_per_chunk;
        float* partial = reinterpret_cast<float*>(ctx->scratch);
        float* partial_b = partial + R * C * H;
        const dim3 grid((unsigned)((H + 511) / 512), (unsigned)R);
        const size_t red_bytes = (size_t)4 * 64 * C * 8 * sizeof(float);
        auto kern = k_head_dw_partial<T>;
        static bool configured = false;
        if (!configured) {
            VBNN_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * HEAD_CMAX * 8 * 4));
            configured = true;
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), red_bytes, ctx->stream, h, ld_h, g_logits, N, H, (int)C, rows_per_chunk, partial,
                           partial_b);
        hipLaunchKernelGGL(k_head_dw_finish, dim3((unsigned)((C * H + 255) / 256)), dim3(256), 0, ctx->stream, partial, partial_b,
                           (int)R, H, (int)C, accumulate, gradWeight, gradBias);
    }
    if (g_prev || gT_prev) {
        const unsigned nb = (unsigned)(((N + 63) / 64) * ((H + 63) / 64));
        const bool vec = (ld_h % 8 == 0) && (!r_prev_any || ld_r_prev % 8 == 0) && (!g_prev || ld_gp % 8 == 0) &&
                         (!(gT_prev || gvT_prev) || ld_gpT % 8 == 0) &&
                         ((((uintptr_t)h | (uintptr_t)r_prev_any | (uintptr_t)g_prev | (uintptr_t)gv_prev | (uintptr_t)gT_prev |
                            (uintptr_t)gvT_prev) & 15u) == 0);
        if (vec)
            hipLaunchKernelGGL((k_head_dx<T, true>), dim3(nb), dim3(256), 0, ctx->stream, h, ld_h, w3, ld_w, g_logits, N, H, (int)C,
                               relu_mask, r_prev, r_prev_t, ld_r_prev, g_prev, gv_prev, ld_gp, gT_prev, gvT_prev, ld_gpT);
        else
            hipLaunchKernelGGL((k_head_dx<T, false>), dim3(nb), dim3(256), 0, ctx->stream, h, ld_h, w3, ld_w, g_logits, N, H, (int)C,
                               relu_mask, r_prev, r_prev_t, ld_r_prev, g_prev, gv_prev, ld_gp, gT_prev, gvT_prev, ld_gpT);
    }
    return vbnn_check_launch("k_head_backward");
}

extern "C" int vbnn_head_backward(vbnn_ctx* ctx, int dtype, const void* h, int64_t ld_h, const void* w3, int64_t ld_w,
                                  const float* g_logits, int64_t N, int64_t H, int64_t C, int accumulate, float* gradWeight,
                                  float* gradBias, int relu_mask, const void* r_prev, int64_t ld_r_prev, int r_prev_packed,
                                  void* g_prev, void* gv_prev, int64_t ld_gp, void* gT_prev, void* gvT_prev, int64_t ld_gpT) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && h && w3 && g_logits, "null argument");
    VBNN_REQUIRE(N > 0 && H > 0 && C > 0 && C <= HEAD_CMAX, "shape (C <= 16)");
    VBNN_REQUIRE(ld_h % VBNN_KPAD == 0 && ld_h >= H && ld_w >= H, "h and w3 must be packed operands");
    VBNN_REQUIRE(!gv_prev || (g_prev && r_prev), "gv_prev needs g_prev and r_prev");
    VBNN_REQUIRE(!(g_prev || gv_prev) || ld_gp >= H, "ld_gp");
    VBNN_REQUIRE(!(gT_prev || gvT_prev) || ld_gpT >= N, "ld_gpT");
    if (dtype == VBNN_F32)
        return head_backward_t<float>(ctx, (const float*)h, ld_h, (const float*)w3, ld_w, g_logits, N, H, C, accumulate, gradWeight,
                                      gradBias, relu_mask, r_prev, ld_r_prev, r_prev_packed, (float*)g_prev, (float*)gv_prev,
                                      ld_gp, (float*)gT_prev, (float*)gvT_prev, ld_gpT);
    if (dtype == VBNN_BF16)
        return head_backward_t<bf16_t>(ctx, (const bf16_t*)h, ld_h, (const bf16_t*)w3, ld_w, g_logits, N, H, C, accumulate,
                                       gradWeight, gradBias, relu_mask, r_prev, ld_r_prev, r_prev_packed, (bf16_t*)g_prev,
                                       (bf16_t*)gv_prev, ld_gp, (bf16_t*)gT_prev, (bf16_t*)gvT_prev, ld_gpT);
    vbnn_set_error("unsupported dtype %d", dtype);
    return VBNN_ERR_UNSUPPORTED;
    VBNN_API_END
}
