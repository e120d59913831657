// elementwise.hip -- the HBM-bound pieces of the VBLinear path: noise fills, weight sampling,
// operand packing (with transpose), the prior / KL sweeps, KL gradients and the glue modules.
// All of them are streaming kernels: coalesced 16-byte accesses, grid capped at 2048 blocks
// with grid-stride loops (cdna_hip_programming.md Guideline 11), reductions as per-block
// partials in double + a deterministic single-block finish (no float atomics).
#include "common.h"

static inline int grid_for(int64_t work_items, int per_block) {
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

// ---------------------------------------------------------------------------------- fill_normal
template <bool HW>
__global__ __launch_bounds__(256) void k_fill_normal(float* out, int64_t rows, int64_t cols, int64_t ld, uint64_t seed,
                                                     uint32_t stream, uint32_t layer, uint32_t draw, int64_t row0, float scale) {
    const int64_t quads = (cols + 3) >> 2;
    const int64_t total = rows * quads;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t r = t / quads, q = t - r * quads;
        const vbnn_f32x4 z = HW ? vbnn_normal4_hw(seed, stream, layer, draw, (uint32_t)(row0 + r), (uint32_t)q)
                                : vbnn_normal4(seed, stream, layer, draw, (uint32_t)(row0 + r), (uint32_t)q);
        float* p = out + r * ld + q * 4;
        const int valid = (int)min((int64_t)4, cols - q * 4);
        const bool vec = ((ld & 3) == 0) && (((uintptr_t)out & 15u) == 0);
        store4<float>(p, z.v[0] * scale, z.v[1] * scale, z.v[2] * scale, z.v[3] * scale, valid, vec);
    }
}

extern "C" int vbnn_fill_normal(vbnn_ctx* ctx, float* out, int64_t rows, int64_t cols, int64_t ld, uint64_t seed,
                                uint32_t stream, uint32_t layer, uint32_t draw, int64_t row0, float scale) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && out, "null ctx/out");
    VBNN_REQUIRE(rows > 0 && cols > 0 && ld >= cols, "shape");
    VBNN_REQUIRE(stream < 256 && layer < (1u << 24), "stream/layer id range");
    hipLaunchKernelGGL(k_fill_normal<false>, dim3(grid_for(rows * ((cols + 3) / 4), 256)), dim3(256), 0, ctx->stream, out, rows,
                       cols, ld, seed, stream, layer, draw, row0, scale);
    return vbnn_check_launch("k_fill_normal");
    VBNN_API_END
}
// the same normals in the bf16 path's hardware-transcendental form (common.h, vbnn_normal4_hw): what a bf16 forward draws
extern "C" int vbnn_fill_normal_hw(vbnn_ctx* ctx, float* out, int64_t rows, int64_t cols, int64_t ld, uint64_t seed,
                                   uint32_t stream, uint32_t layer, uint32_t draw, int64_t row0, float scale) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && out, "null ctx/out");
    VBNN_REQUIRE(rows > 0 && cols > 0 && ld >= cols, "shape");
    VBNN_REQUIRE(stream < 256 && layer < (1u << 24), "stream/layer id range");
    hipLaunchKernelGGL(k_fill_normal<true>, dim3(grid_for(rows * ((cols + 3) / 4), 256)), dim3(256), 0, ctx->stream, out, rows,
                       cols, ld, seed, stream, layer, draw, row0, scale);
    return vbnn_check_launch("k_fill_normal");
    VBNN_API_END
}

// Both forms of the contract's Box-Muller on GIVEN Philox words (include/vbnn_hip.h, vbnn_box_muller_forms): what lets a host pin the
// hardware form's error against the bit-exact form exhaustively over either word (all 2^24 radii, all 2^24 angles) instead of on
// whatever words a window of counters happens to hold.
__global__ __launch_bounds__(256) void k_box_muller_forms(const uint32_t* __restrict__ x0, const uint32_t* __restrict__ x1, float* z_exact,
                                                          float* z_hw, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float a, b, c, d;
        vbnn_box_muller(x0[i], x1[i], &a, &b);
        vbnn_box_muller_hw(x0[i], x1[i], &c, &d);
        z_exact[2 * i] = a; z_exact[2 * i + 1] = b;
        z_hw[2 * i] = c; z_hw[2 * i + 1] = d;
    }
}
extern "C" int vbnn_box_muller_forms(vbnn_ctx* ctx, const uint32_t* x0, const uint32_t* x1, float* z_exact, float* z_hw, int64_t n) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && x0 && x1 && z_exact && z_hw && n > 0, "argument");
    hipLaunchKernelGGL(k_box_muller_forms, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, x0, x1, z_exact, z_hw, n);
    return vbnn_check_launch("k_box_muller_forms");
    VBNN_API_END
}

// ---------------------------------------------------------------------------------- block reduce
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
// sums `v` over the 256 threads of a block; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* sh /* >= 4 doubles */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---------------------------------------------------------------------------------- compute_prior
// VBLinear.lua:77-88, one sweep: 8 B read per weight (+ 12 B written when the caches are asked for).
__global__ __launch_bounds__(256) void k_prior_partial(const float* __restrict__ means, const float* __restrict__ lvars,
                                                       int64_t W, float* vars, float* stdv, float* mu_sqe,
                                                       double* partial /* [gridDim.x][2] */) {
    __shared__ double sh[4];
    double s1 = 0.0, s2 = 0.0;
    const int64_t W4 = W >> 2;
    const bool vec = ((((uintptr_t)means | (uintptr_t)lvars | (uintptr_t)vars | (uintptr_t)stdv | (uintptr_t)mu_sqe) & 15u) == 0);
    if (vec) {
        for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < W4; t += (int64_t)gridDim.x * 256) {
            const f32x4 m = reinterpret_cast<const f32x4*>(means)[t];
            const f32x4 l = reinterpret_cast<const f32x4*>(lvars)[t];
            f32x4 v, sd, q;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = expf(l[j]); sd[j] = sqrtf(v[j]); q[j] = m[j] * m[j];
                s1 += (double)(v[j] + q[j]); s2 += (double)l[j];
            }
            if (vars) reinterpret_cast<f32x4*>(vars)[t] = v;
            if (stdv) reinterpret_cast<f32x4*>(stdv)[t] = sd;
            if (mu_sqe) reinterpret_cast<f32x4*>(mu_sqe)[t] = q;
        }
    }
    const int64_t tail0 = vec ? (W4 << 2) : 0;
    for (int64_t t = tail0 + (int64_t)blockIdx.x * 256 + threadIdx.x; t < W; t += (int64_t)gridDim.x * 256) {
        const float v = expf(lvars[t]), sd = sqrtf(v), q = means[t] * means[t];
        s1 += (double)(v + q); s2 += (double)lvars[t];
        if (vars) vars[t] = v;
        if (stdv) stdv[t] = sd;
        if (mu_sqe) mu_sqe[t] = q;
    }
    const double r1 = block_sum(s1, sh);
    const double r2 = block_sum(s2, sh);
    if (threadIdx.x == 0) { partial[blockIdx.x * 2] = r1; partial[blockIdx.x * 2 + 1] = r2; }
}
// the deterministic finish of the prior sums: one block adds the block partials in a fixed order
__device__ __forceinline__ void prior_finish(const double* partial, int nblocks, int64_t W, double* stats, double* sh) {
    double s1 = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) { s1 += partial[b * 2]; s2 += partial[b * 2 + 1]; }
    const double r1 = block_sum(s1, sh);
    const double r2 = block_sum(s2, sh);
    if (threadIdx.x == 0) { stats[0] = r1; stats[1] = r2; stats[2] = (1.0 / (double)W) * r1; stats[3] = (double)W; }
}
__global__ __launch_bounds__(256) void k_prior_finish(const double* partial, int nblocks, int64_t W, double* stats) {
    __shared__ double sh[4];
    prior_finish(partial, nblocks, W, stats, sh);
}

extern "C" int vbnn_compute_prior(vbnn_ctx* ctx, const float* means, const float* lvars, int64_t W, float* vars,
                                  float* stdv, float* mu_sqe, double* stats) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && means && lvars && stats, "null argument");
    VBNN_REQUIRE(W > 0, "W");
    const int nb = grid_for(W / 4 + 1, 256);
    VBNN_REQUIRE((size_t)nb * 2 <= ctx->scratch_doubles, "scratch");
    hipLaunchKernelGGL(k_prior_partial, dim3(nb), dim3(256), 0, ctx->stream, means, lvars, W, vars, stdv, mu_sqe, ctx->scratch);
    hipLaunchKernelGGL(k_prior_finish, dim3(1), dim3(256), 0, ctx->stream, ctx->scratch, nb, W, stats);
    return vbnn_check_launch("k_prior");
    VBNN_API_END
}

// ---------------------------------------------------------------------------------- wn_sample
// VBLinear.lua:49-64: w = means + stdv (.) e
__global__ __launch_bounds__(256) void k_wn_sample(const float* __restrict__ means, const float* __restrict__ stdv,
                                                   const float* __restrict__ lvars, float* weight, float* e_out,
                                                   int64_t O, int64_t I, uint64_t seed, uint32_t layer, uint32_t draw) {
    const int64_t quads = (I + 3) >> 2;
    const int64_t total = O * quads;
    const bool vec = ((I & 3) == 0) &&
                     ((((uintptr_t)means | (uintptr_t)stdv | (uintptr_t)lvars | (uintptr_t)weight | (uintptr_t)e_out) & 15u) == 0);
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t o = t / quads, q = t - o * quads;
        const vbnn_f32x4 z = vbnn_normal4(seed, VBNN_STREAM_EPS, layer, draw, (uint32_t)o, (uint32_t)q);
        const int64_t base = o * I + q * 4;
        const int valid = (int)min((int64_t)4, I - q * 4);
        float w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < valid) {
                const float sd = stdv ? stdv[base + j] : sqrtf(expf(lvars[base + j]));
                const float tmp = sd * z.v[j];            // torch.cmul(stdv, e): a rounded temporary (:59)
                w[j] = means[base + j] + tmp;
            } else w[j] = 0.f;
        }
        store4<float>(weight + base, w[0], w[1], w[2], w[3], valid, vec);
        if (e_out) store4<float>(e_out + base, z.v[0], z.v[1], z.v[2], z.v[3], valid, vec);
    }
}

extern "C" int vbnn_wn_sample(vbnn_ctx* ctx, const float* means, const float* stdv, const float* lvars, float* weight,
                              float* e_out, int64_t O, int64_t I, uint64_t seed, uint32_t layer, uint32_t draw) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && means && weight, "null argument");
    VBNN_REQUIRE(stdv || lvars, "need stdv or lvars");
    VBNN_REQUIRE(O > 0 && I > 0, "shape");
    hipLaunchKernelGGL(k_wn_sample, dim3(grid_for(O * ((I + 3) / 4), 256)), dim3(256), 0, ctx->stream, means, stdv, lvars,
                       weight, e_out, O, I, seed, layer, draw);
    return vbnn_check_launch("k_wn_sample");
    VBNN_API_END
}

// ---------------------------------------------------------------------------------- pack (+ transpose)
__device__ __forceinline__ float pack_func(int func, float a, float b) {
    switch (func) {
        case VBNN_PACK_EXP: return expf(a);
        case VBNN_PACK_SQUARE: return a * a;
        case VBNN_PACK_MUL: return a * b;
        case VBNN_PACK_RELU: return fmaxf(a, 0.f);
        case VBNN_PACK_RELU_SQUARE: { const float h = fmaxf(a, 0.f); return h * h; }
        default: return a;
    }
}
// 64 x 64 tile per block; src read row-wise (coalesced), dst written row-wise, dstT written
// row-wise after a transpose through LDS (65-float pitch: conflict-free column reads).
// SQUARE-type functions square the value AFTER rounding to T, so that x2 == T(x)^2 exactly as
// the GEMM epilogues produce it.
template <typename T>
__global__ __launch_bounds__(256) void k_pack(int func, const float* __restrict__ src, const float* __restrict__ src2,
                                              int64_t ld_src, int64_t rows, int64_t cols, T* dst, int64_t ld_dst, T* dstT,
                                              int64_t ld_dstT) {
    __shared__ float tile[64][65];
    const int64_t tiles_c = (cols + 63) / 64;
    const int64_t tiles_r = (rows + 63) / 64;
    const int64_t ntiles = tiles_c * tiles_r;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // ty in 0..3
    for (int64_t tI = blockIdx.x; tI < ntiles; tI += gridDim.x) {
        const int64_t r0 = (tI / tiles_c) * 64, c0 = (tI % tiles_c) * 64;
#pragma unroll 4
        for (int rr = ty; rr < 64; rr += 4) {
            const int64_t r = r0 + rr, c = c0 + tx;
            float v = 0.f;
            if (r < rows && c < cols) {
                const float a = src[r * ld_src + c];
                const float b = src2 ? src2[r * ld_src + c] : 0.f;
                if (func == VBNN_PACK_SQUARE || func == VBNN_PACK_RELU_SQUARE) {
                    const float h = (func == VBNN_PACK_RELU_SQUARE) ? fmaxf(a, 0.f) : a;
                    const float hr = Elt<T>::from(Elt<T>::to(h));
                    v = hr * hr;
                } else {
                    v = pack_func(func, a, b);
                }
                if (dst) dst[r * ld_dst + c] = Elt<T>::to(v);
            }
            tile[rr][tx] = v;
        }
        if (dstT) {
            __syncthreads();
#pragma unroll 4
            for (int cc = ty; cc < 64; cc += 4) {
                const int64_t c = c0 + cc, r = r0 + tx;
                if (c < cols && r < rows) dstT[c * ld_dstT + r] = Elt<T>::to(tile[tx][cc]);
            }
            __syncthreads();
        }
    }
}

extern "C" int vbnn_pack(vbnn_ctx* ctx, int dtype, int func, const float* src, const float* src2, int64_t ld_src,
                         int64_t rows, int64_t cols, void* dst, int64_t ld_dst, void* dstT, int64_t ld_dstT) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && src, "null argument");
    VBNN_REQUIRE(dst || dstT, "need dst or dstT");
    VBNN_REQUIRE(rows > 0 && cols > 0 && ld_src >= cols, "shape");
    VBNN_REQUIRE(!dst || ld_dst >= cols, "ld_dst");
    VBNN_REQUIRE(!dstT || ld_dstT >= rows, "ld_dstT");
    VBNN_REQUIRE(func >= 0 && func <= VBNN_PACK_RELU_SQUARE, "func");
    VBNN_REQUIRE(func != VBNN_PACK_MUL || src2, "MUL needs src2");
    const int64_t ntiles = ((rows + 63) / 64) * ((cols + 63) / 64);
    const int nb = (int)(ntiles < 4096 ? ntiles : 4096);
    if (dtype == VBNN_F32)
        hipLaunchKernelGGL(k_pack<float>, dim3(nb), dim3(256), 0, ctx->stream, func, src, src2, ld_src, rows, cols,
                           (float*)dst, ld_dst, (float*)dstT, ld_dstT);
    else if (dtype == VBNN_BF16)
        hipLaunchKernelGGL(k_pack<bf16_t>, dim3(nb), dim3(256), 0, ctx->stream, func, src, src2, ld_src, rows, cols,
                           (bf16_t*)dst, ld_dst, (bf16_t*)dstT, ld_dstT);
    else { vbnn_set_error("unsupported dtype %d", dtype); return VBNN_ERR_UNSUPPORTED; }
    return vbnn_check_launch("k_pack");
    VBNN_API_END
}

// ---------------------------------------------------------------------------------- KL gradients
// VBLinear.lua:90-93
__global__ __launch_bounds__(256) void k_mugrads(const float* __restrict__ means, const double* __restrict__ stats, float B,
                                                 float S, float* gradWeight, float* lcg, int64_t W) {
    const float den = (float)((double)B * stats[2]);
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < W; t += (int64_t)gridDim.x * 256) {
        if (lcg) lcg[t] = (means[t] - 0.0f) / den;
        if (gradWeight) gradWeight[t] = gradWeight[t] / S;
    }
}
// VBLinear.lua:95-98
__global__ __launch_bounds__(256) void k_vargrads(const float* __restrict__ lvars, const float* __restrict__ vars,
                                                  const float* __restrict__ stdv, const double* __restrict__ stats, float B,
                                                  float S, float* gradSum, float* lcg, int64_t W) {
    const float inv_vh = (float)(1.0 / stats[2]);
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < W; t += (int64_t)gridDim.x * 256) {
        const float v = vars ? vars[t] : expf(lvars[t]);
        const float sd = stdv ? stdv[t] : sqrtf(v);
        if (lcg) { const float a = -(1.0f / v) + inv_vh; lcg[t] = (a / (2.0f * B)) * v; }
        if (gradSum) gradSum[t] = (gradSum[t] / (2.0f * S)) * sd;
    }
}

extern "C" int vbnn_compute_mugrads(vbnn_ctx* ctx, const float* means, const double* stats, float B, float S,
                                    float* gradWeight, float* lcg, int64_t W) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && means && stats, "null argument");
    VBNN_REQUIRE(W > 0 && B > 0 && S > 0, "W, B, S");
    hipLaunchKernelGGL(k_mugrads, dim3(grid_for(W, 256)), dim3(256), 0, ctx->stream, means, stats, B, S, gradWeight, lcg, W);
    return vbnn_check_launch("k_mugrads");
    VBNN_API_END
}
extern "C" int vbnn_compute_vargrads(vbnn_ctx* ctx, const float* lvars, const float* vars, const float* stdv,
                                     const double* stats, float B, float S, float* gradSum, float* lcg, int64_t W) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && stats, "null argument");
    VBNN_REQUIRE(lvars || (vars && stdv), "need lvars or vars+stdv");
    VBNN_REQUIRE(W > 0 && B > 0 && S > 0, "W, B, S");
    hipLaunchKernelGGL(k_vargrads, dim3(grid_for(W, 256)), dim3(256), 0, ctx->stream, lvars, vars, stdv, stats, B, S, gradSum,
                       lcg, W);
    return vbnn_check_launch("k_vargrads");
    VBNN_API_END
}

// ---------------------------------------------------------------------------------- calc_lc
// VBLinear.lua:99-103 + mlp.lua:112 (:sum()).
__global__ __launch_bounds__(256) void k_lc_partial(const float* __restrict__ means, const float* __restrict__ lvars,
                                                    const float* __restrict__ vars, const float* __restrict__ mu_sqe,
                                                    const double* __restrict__ stats, float B, float* lc_elem, int64_t W,
                                                    double* partial) {
    __shared__ double sh[4];
    const double var_hat = stats[2];
    const float lvh = (float)log(sqrt(var_hat));
    const float vh = (float)var_hat;
    const float invB = 1.0f / B;
    double s = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < W; t += (int64_t)gridDim.x * 256) {
        const float v = vars ? vars[t] : expf(lvars[t]);
        const float q = mu_sqe ? mu_sqe[t] : means[t] * means[t];
        const float first = -logf(sqrtf(v)) + lvh;
        const float second = (q + (v - vh)) / (2.0f * vh);
        const float lc = (first + second) * invB;
        if (lc_elem) lc_elem[t] = lc;
        s += (double)lc;
    }
    const double r = block_sum(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
__global__ __launch_bounds__(256) void k_sum_finish(const double* partial, int nblocks, double* out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[b];
    const double r = block_sum(s, sh);
    if (threadIdx.x == 0) out[0] = r;
}

extern "C" int vbnn_calc_lc(vbnn_ctx* ctx, const float* means, const float* lvars, const float* vars, const float* mu_sqe,
                            const double* stats, float B, float* lc_elem, double* lc_sum_dev, int64_t W) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && stats && lc_sum_dev, "null argument");
    VBNN_REQUIRE((means && lvars) || (vars && mu_sqe), "need means+lvars or the cached vars+mu_sqe");
    VBNN_REQUIRE(W > 0 && B > 0, "W, B");
    const int nb = grid_for(W, 1024);
    VBNN_REQUIRE((size_t)nb <= ctx->scratch_doubles, "scratch");
    hipLaunchKernelGGL(k_lc_partial, dim3(nb), dim3(256), 0, ctx->stream, means, lvars, vars, mu_sqe, stats, B, lc_elem, W, ctx->scratch);
    hipLaunchKernelGGL(k_sum_finish, dim3(1), dim3(256), 0, ctx->stream, ctx->scratch, nb, lc_sum_dev);
    return vbnn_check_launch("k_lc");
    VBNN_API_END
}

// ---------------------------------------------------------------------------------- gradBias
// gradBias[o] (+)= scale * sum_n g[n][o], two deterministic stages (no float atomics):
//   stage 1  grid (O / 64, R): a block sums its row chunk for 64 columns (rows read as contiguous
//            64-element segments), 4 row lanes per column folded through LDS -> partial[chunk][o]
//   stage 2  one thread per column adds the R partials in chunk order.
template <typename T>
__global__ __launch_bounds__(256) void k_col_sum_partial(const T* __restrict__ g, int64_t ld, int64_t N, int64_t O,
                                                         int rows_per_chunk, float* __restrict__ partial) {
    __shared__ float sh[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t o = (int64_t)blockIdx.x * 64 + tx;
    const int64_t n0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t n1 = min(N, n0 + rows_per_chunk);
    float s = 0.f;
    if (o < O) {
        const T* p = g + o;
#pragma unroll 4
        for (int64_t n = n0 + ty; n < n1; n += 4) s += Elt<T>::from(p[n * ld]);
    }
    sh[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && o < O) partial[(int64_t)blockIdx.y * O + o] = (sh[0][tx] + sh[1][tx]) + (sh[2][tx] + sh[3][tx]);
}
__global__ __launch_bounds__(256) void k_col_sum_finish(const float* __restrict__ partial, int R, int64_t O, float scale,
                                                        int accumulate, float* gradBias) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= O) return;
    float tot = 0.f;
    int r = 0;
    for (; r + 8 <= R; r += 8) {                      // eight independent loads in flight, summed in chunk order
        float p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = partial[(int64_t)(r + u) * O + o];
#pragma unroll
        for (int u = 0; u < 8; ++u) tot += p[u];
    }
    for (; r < R; ++r) tot += partial[(int64_t)r * O + o];
    const float old = accumulate ? gradBias[o] : 0.f;
    gradBias[o] = fmaf(scale, tot, old);
}
extern "C" int vbnn_acc_grad_bias(vbnn_ctx* ctx, int dtype, const void* g, int64_t ld_g, int64_t N, int64_t O, float scale,
                                  int accumulate, float* gradBias) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && g && gradBias, "null argument");
    VBNN_REQUIRE(N > 0 && O > 0 && ld_g >= O, "shape");
    const int64_t cap = (int64_t)(ctx->scratch_doubles * 2) / O;      // partial rows that fit the scratch (floats)
    VBNN_REQUIRE(cap >= 1, "outputSize too large for the reduction scratch");
    int64_t R = (N + 31) / 32;
    if (R > 32) R = 32;
    if (R > cap) R = cap;
    const int rows_per_chunk = (int)((N + R - 1) / R);
    R = (N + rows_per_chunk - 1) / rows_per_chunk;
    float* partial = reinterpret_cast<float*>(ctx->scratch);
    const dim3 grid((unsigned)((O + 63) / 64), (unsigned)R);
    if (dtype == VBNN_F32)
        hipLaunchKernelGGL(k_col_sum_partial<float>, grid, dim3(256), 0, ctx->stream, (const float*)g, ld_g, N, O,
                           rows_per_chunk, partial);
    else if (dtype == VBNN_BF16)
        hipLaunchKernelGGL(k_col_sum_partial<bf16_t>, grid, dim3(256), 0, ctx->stream, (const bf16_t*)g, ld_g, N, O,
                           rows_per_chunk, partial);
    else { vbnn_set_error("unsupported dtype %d", dtype); return VBNN_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(k_col_sum_finish, dim3((unsigned)((O + 255) / 256)), dim3(256), 0, ctx->stream, partial, (int)R, O, scale,
                       accumulate, gradBias);
    return vbnn_check_launch("k_col_sum");
    VBNN_API_END
}

// ---------------------------------------------------------------------------------- ReLU
__global__ __launch_bounds__(256) void k_relu_fwd(const float* __restrict__ x, float* y, int64_t n) {
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) y[t] = x[t] > 0.f ? x[t] : 0.f;
}
__global__ __launch_bounds__(256) void k_relu_bwd(const float* __restrict__ x, const float* __restrict__ g, float* gx, int64_t n) {
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) gx[t] = x[t] > 0.f ? g[t] : 0.f;
}
extern "C" int vbnn_relu_forward(vbnn_ctx* ctx, const float* x, float* y, int64_t n) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && x && y && n > 0, "argument");
    hipLaunchKernelGGL(k_relu_fwd, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, x, y, n);
    return vbnn_check_launch("k_relu_fwd");
    VBNN_API_END
}
extern "C" int vbnn_relu_backward(vbnn_ctx* ctx, const float* x, const float* g, float* gx, int64_t n) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && x && g && gx && n > 0, "argument");
    hipLaunchKernelGGL(k_relu_bwd, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, x, g, gx, n);
    return vbnn_check_launch("k_relu_bwd");
    VBNN_API_END
}

// ---------------------------------------------------------------------------------- LogSoftMax + ClassNLL
// one wave per row; C <= 64 * 16.
__global__ __launch_bounds__(256) void k_logsoftmax_nll(const float* __restrict__ logits, int64_t ld,
                                                        const int32_t* __restrict__ target, int64_t N, int64_t C, float inv_n,
                                                        float* out, float* g_logits, double* loss_sum, int32_t* correct) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    double loss_acc = 0.0;
    int corr_acc = 0;
    for (int64_t n = wave_global; n < N; n += nwaves) {
        const float* row = logits + n * ld;
        float mx = -INFINITY; int arg = 0;
        for (int64_t c = lane; c < C; c += 64) { const float v = row[c]; if (v > mx) { mx = v; arg = (int)c; } }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float om = __shfl_xor(mx, off, 64);
            const int oa = __shfl_xor(arg, off, 64);
            if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }   // first maximum wins (Tensor:max)
        }
        double s = 0.0;
        for (int64_t c = lane; c < C; c += 64) s += (double)expf(row[c] - mx);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        const float lse = mx + logf((float)s);
        const int32_t t = min(max(target[n], 0), (int32_t)C - 1);   // clamp: never index out of the row
        for (int64_t c = lane; c < C; c += 64) {
            const float o = row[c] - lse;
            if (out) out[n * C + c] = o;
            if (g_logits) g_logits[n * C + c] = (expf(o) - (c == t ? 1.0f : 0.0f)) * inv_n;
        }
        if (lane == 0) {
            loss_acc -= (double)(row[t] - lse) * (double)inv_n;
            corr_acc += (arg == t) ? 1 : 0;
        }
    }
    if (lane == 0) {
        if (loss_sum && loss_acc != 0.0) atomicAdd(loss_sum, loss_acc);
        if (correct && corr_acc) atomicAdd(correct, corr_acc);
    }
}
extern "C" int vbnn_logsoftmax_nll(vbnn_ctx* ctx, const float* logits, int64_t ld, const int32_t* target, int64_t N,
                                   int64_t C, float inv_n, float* out, float* g_logits, double* loss_sum_dev,
                                   int32_t* correct_dev) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && logits && target, "null argument");
    VBNN_REQUIRE(N > 0 && C > 0 && ld >= C, "shape");
    hipLaunchKernelGGL(k_logsoftmax_nll, dim3(grid_for(N, 4)), dim3(256), 0, ctx->stream, logits, ld, target, N, C, inv_n, out,
                       g_logits, loss_sum_dev, correct_dev);
    return vbnn_check_launch("k_logsoftmax_nll");
    VBNN_API_END
}

// ---------------------------------------------------------------------------------- MSE criterion (configs[4])
// One pass over y and target (8 B read + 4 B written per output element, HBM-bound): g = 2 inv_nd (y - t) and the block
// partials of sum (y - t)^2; the last launch of the pair adds the partials in a fixed order.
__global__ __launch_bounds__(256) void k_mse(const float* __restrict__ y, int64_t ld_y, const float* __restrict__ t, int64_t ld_t,
                                             int64_t N, int64_t D, float inv_nd, float* g, int64_t ld_g, double* partial) {
    __shared__ double sh[4];
    double acc = 0.0;
    const bool vec = ((D & 3) == 0) && ((ld_y & 3) == 0) && ((ld_t & 3) == 0) && ((ld_g & 3) == 0) &&
                     ((((uintptr_t)y | (uintptr_t)t | (uintptr_t)g) & 15u) == 0);
    const int64_t D4 = (D + 3) >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N * D4; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / D4, d = (i - n * D4) * 4;
        const int valid = (int)min((int64_t)4, D - d);
        float yv[4], tv[4], gv[4];
        load4<float>(y + n * ld_y + d, yv, valid, vec);
        load4<float>(t + n * ld_t + d, tv, valid, vec);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float df = (j < valid) ? yv[j] - tv[j] : 0.f;
            acc += (double)df * (double)df;
            gv[j] = 2.0f * inv_nd * df;
        }
        if (g) store4<float>(g + n * ld_g + d, gv[0], gv[1], gv[2], gv[3], valid, vec);
    }
    if (partial) {
        const double r = block_sum(acc, sh);
        if (threadIdx.x == 0) partial[blockIdx.x] = r;
    }
}
__global__ __launch_bounds__(256) void k_mse_finish(const double* partial, int nb, double scale, int accumulate, double* loss) {
    __shared__ double sh[4];
    double a = 0.0;
    for (int b = threadIdx.x; b < nb; b += 256) a += partial[b];
    const double r = block_sum(a, sh);
    if (threadIdx.x == 0) loss[0] = (accumulate ? loss[0] : 0.0) + scale * r;
}
static int mse_launch(vbnn_ctx* ctx, const float* y, int64_t ld_y, const float* t, int64_t ld_t, int64_t N, int64_t D, float inv_nd,
                      float* g, int64_t ld_g, int accumulate, double* loss) {
    const int nb = grid_for(N * ((D + 3) / 4), 256 * 4);
    if ((size_t)nb > ctx->scratch_doubles) { vbnn_set_error("scratch"); return VBNN_ERR_INVALID; }
    hipLaunchKernelGGL(k_mse, dim3(nb), dim3(256), 0, ctx->stream, y, ld_y, t, ld_t, N, D, inv_nd, g, ld_g, loss ? ctx->scratch : nullptr);
    if (loss) hipLaunchKernelGGL(k_mse_finish, dim3(1), dim3(256), 0, ctx->stream, ctx->scratch, nb, (double)inv_nd, accumulate, loss);
    return vbnn_check_launch("k_mse");
}
extern "C" int vbnn_mse_forward(vbnn_ctx* ctx, const float* y, int64_t ld_y, const float* target, int64_t ld_t, int64_t N, int64_t D,
                                float inv_nd, float* g, int64_t ld_g, int accumulate, double* loss_sum_dev) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && y && target && loss_sum_dev, "null argument");
    VBNN_REQUIRE(N > 0 && D > 0 && ld_y >= D && ld_t >= D && (!g || ld_g >= D), "shape");
    return mse_launch(ctx, y, ld_y, target, ld_t, N, D, inv_nd, g, ld_g, accumulate, loss_sum_dev);
    VBNN_API_END
}
extern "C" int vbnn_mse_backward(vbnn_ctx* ctx, const float* y, int64_t ld_y, const float* target, int64_t ld_t, int64_t N, int64_t D,
                                 float inv_nd, float* g, int64_t ld_g) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && y && target && g, "null argument");
    VBNN_REQUIRE(N > 0 && D > 0 && ld_y >= D && ld_t >= D && ld_g >= D, "shape");
    return mse_launch(ctx, y, ld_y, target, ld_t, N, D, inv_nd, g, ld_g, 0, nullptr);
    VBNN_API_END
}

// ---------------------------------------------------------------------------------- separate criterion modules
// nn.ClassNLLCriterion (sizeAverage): forward value, backward gradient; nn.LogSoftMax:updateGradInput.
// Used by the module-level path (mlp.lua:78-80 call order); the fused kernel above is the fast path.
__global__ __launch_bounds__(256) void k_nll_forward(const float* __restrict__ out, int64_t ld, const int32_t* __restrict__ target,
                                                     int64_t N, int64_t C, float inv_n, double* loss_sum, int32_t* correct) {
    __shared__ double sh[4];
    double acc = 0.0;
    int corr = 0;
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256) {
        const float* row = out + n * ld;
        const int32_t t = min(max(target[n], 0), (int32_t)C - 1);
        acc -= (double)row[t] * (double)inv_n;
        if (correct) {
            int best = 0;
            for (int c = 1; c < (int)C; ++c) if (row[c] > row[best]) best = c;
            corr += (best == t) ? 1 : 0;
        }
    }
    const double r = block_sum(acc, sh);
    if (threadIdx.x == 0 && loss_sum) atomicAdd(loss_sum, r);
    if (correct && corr) atomicAdd(correct, corr);
}
__global__ __launch_bounds__(256) void k_nll_backward(const int32_t* __restrict__ target, int64_t N, int64_t C, float inv_n, float* g) {
    const int64_t total = N * C;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t n = t / C, c = t - n * C;
        g[t] = (c == (int64_t)target[n]) ? -inv_n : 0.f;
    }
}
__global__ __launch_bounds__(256) void k_logsoftmax_backward(const float* __restrict__ out, const float* __restrict__ g, float* gx,
                                                             int64_t N, int64_t C) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int64_t n = wave_global; n < N; n += (int64_t)gridDim.x * 4) {
        double s = 0.0;
        for (int64_t c = lane; c < C; c += 64) s += (double)g[n * C + c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        for (int64_t c = lane; c < C; c += 64) gx[n * C + c] = g[n * C + c] - expf(out[n * C + c]) * (float)s;
    }
}
extern "C" int vbnn_nll_forward(vbnn_ctx* ctx, const float* out, int64_t ld, const int32_t* target, int64_t N, int64_t C,
                                float inv_n, double* loss_sum_dev, int32_t* correct_dev) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && out && target, "null argument");
    VBNN_REQUIRE(N > 0 && C > 0 && ld >= C, "shape");
    hipLaunchKernelGGL(k_nll_forward, dim3(grid_for(N, 256)), dim3(256), 0, ctx->stream, out, ld, target, N, C, inv_n,
                       loss_sum_dev, correct_dev);
    return vbnn_check_launch("k_nll_forward");
    VBNN_API_END
}
extern "C" int vbnn_nll_backward(vbnn_ctx* ctx, const int32_t* target, int64_t N, int64_t C, float inv_n, float* g) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && target && g, "null argument");
    VBNN_REQUIRE(N > 0 && C > 0, "shape");
    hipLaunchKernelGGL(k_nll_backward, dim3(grid_for(N * C, 256)), dim3(256), 0, ctx->stream, target, N, C, inv_n, g);
    return vbnn_check_launch("k_nll_backward");
    VBNN_API_END
}
extern "C" int vbnn_logsoftmax_backward(vbnn_ctx* ctx, const float* out, const float* g, float* gx, int64_t N, int64_t C) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && out && g && gx, "null argument");
    VBNN_REQUIRE(N > 0 && C > 0, "shape");
    hipLaunchKernelGGL(k_logsoftmax_backward, dim3(grid_for(N, 4)), dim3(256), 0, ctx->stream, out, g, gx, N, C);
    return vbnn_check_launch("k_logsoftmax_backward");
    VBNN_API_END
}


// ---------------------------------------------------------------------------------- prep_layer
// The per-step parameter sweep of the fused path: ONE read of means/lvars (8 B per weight) produces
//   - the GEMM shadows mu, sigma^2 = exp(lvars) (and their transposes for the gradInput GEMM),
//   - the prior statistics of VBLinear:compute_prior (VBLinear.lua:77-88) as block partials.
template <typename T>
__global__ __launch_bounds__(256) void k_prep_layer(const float* __restrict__ means, const float* __restrict__ lvars,
                                                    int64_t O, int64_t I, T* mu_s, T* var_s, int64_t ld_w, T* muT_s, T* varT_s,
                                                    int64_t ld_wT, double* partial) {
    __shared__ float tm[64][65];
    __shared__ float tv[64][65];
    __shared__ double sh[4];
    const int64_t tiles_c = (I + 63) / 64, tiles_r = (O + 63) / 64;
    const int64_t ntiles = tiles_c * tiles_r;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    double s1 = 0.0, s2 = 0.0;
    (void)tx; (void)ty;
    // every thread moves 4 consecutive elements at a time: 16-byte loads of means / lvars, 8-byte (bf16) stores of
    // the shadows, and -- after the LDS transpose -- 4 consecutive rows of one column for the transposed shadows
    const bool vec_in = ((I & 3) == 0) && ((((uintptr_t)means | (uintptr_t)lvars) & 15u) == 0);
    const bool vec_t = muT_s && ((ld_wT & 3) == 0);
    if (!muT_s && vec_in && (int64_t)gridDim.x * 1024 <= O * I) {
        // FLAT form (no transposed shadows to build), as k_vb_update's: a wave-instruction is one contiguous KiB of means / lvars
        const int64_t total = O * I;
        for (int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; base < total; base += (int64_t)gridDim.x * 1024) {
            const int64_t r = base / I, c = base - r * I;
            float m[4], l[4], v[4];
            load4<float>(means + base, m, 4, true);
            load4<float>(lvars + base, l, 4, true);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = expf(l[e]); s1 += (double)__fadd_rn(v[e], __fmul_rn(m[e], m[e])); s2 += (double)l[e]; }
            store4<T>(mu_s + r * ld_w + c, m[0], m[1], m[2], m[3], 4, true);
            store4<T>(var_s + r * ld_w + c, v[0], v[1], v[2], v[3], 4, true);
        }
        const double r1 = block_sum(s1, sh);
        const double r2 = block_sum(s2, sh);
        if (threadIdx.x == 0) { partial[blockIdx.x * 2] = r1; partial[blockIdx.x * 2 + 1] = r2; }
        return;
    }
    for (int64_t tI = blockIdx.x; tI < ntiles; tI += gridDim.x) {
        const int64_t r0 = (tI / tiles_c) * 64, c0 = (tI % tiles_c) * 64;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = threadIdx.x + 256 * k;
            const int rr = idx >> 4, c4 = (idx & 15) * 4;
            const int64_t r = r0 + rr, c = c0 + c4;
            float m[4] = {0.f, 0.f, 0.f, 0.f}, l[4] = {0.f, 0.f, 0.f, 0.f}, v[4] = {0.f, 0.f, 0.f, 0.f};
            const int valid = (r < O) ? (int)min((int64_t)4, I - c) : 0;
            if (valid > 0) {
                load4<float>(means + r * I + c, m, valid, vec_in);
                load4<float>(lvars + r * I + c, l, valid, vec_in);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    // (vars + mu_sqe as the fp32 tensors of VBLinear.lua:82-86: the square and the sum rounded separately, whatever
                    // the compiler would contract -- the update sweep forms the same terms in another order of elements)
                    if (e < valid) { v[e] = expf(l[e]); s1 += (double)__fadd_rn(v[e], __fmul_rn(m[e], m[e])); s2 += (double)l[e]; }
                store4<T>(mu_s + r * ld_w + c, m[0], m[1], m[2], m[3], valid, true);
                store4<T>(var_s + r * ld_w + c, v[0], v[1], v[2], v[3], valid, true);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { tm[rr][c4 + e] = m[e]; tv[rr][c4 + e] = v[e]; }
        }
        if (muT_s) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = threadIdx.x + 256 * k;
                const int cc = idx >> 4, r4 = (idx & 15) * 4;
                const int64_t c = c0 + cc, r = r0 + r4;
                const int valid = (c < I) ? (int)min((int64_t)4, O - r) : 0;
                if (valid > 0) {
                    store4<T>(muT_s + c * ld_wT + r, tm[r4][cc], tm[r4 + 1][cc], tm[r4 + 2][cc], tm[r4 + 3][cc], valid, vec_t);
                    store4<T>(varT_s + c * ld_wT + r, tv[r4][cc], tv[r4 + 1][cc], tv[r4 + 2][cc], tv[r4 + 3][cc], valid, vec_t);
                }
            }
            __syncthreads();
        }
    }
    const double r1 = block_sum(s1, sh);
    const double r2 = block_sum(s2, sh);
    if (threadIdx.x == 0) { partial[blockIdx.x * 2] = r1; partial[blockIdx.x * 2 + 1] = r2; }
}

extern "C" int vbnn_prep_layer(vbnn_ctx* ctx, int dtype, const float* means, const float* lvars, int64_t O, int64_t I,
                               void* mu_s, void* var_s, int64_t ld_w, void* muT_s, void* varT_s, int64_t ld_wT,
                               double* stats) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && means && lvars && mu_s && var_s && stats, "null argument");
    VBNN_REQUIRE((muT_s == nullptr) == (varT_s == nullptr), "muT_s and varT_s go together");
    VBNN_REQUIRE(O > 0 && I > 0 && ld_w >= I && (!muT_s || ld_wT >= O), "shape");
    const int64_t ntiles = ((O + 63) / 64) * ((I + 63) / 64);
    const int nb = (int)(ntiles < 2048 ? ntiles : 2048);
    VBNN_REQUIRE((size_t)nb * 2 <= ctx->scratch_doubles, "scratch");
    if (dtype == VBNN_F32)
        hipLaunchKernelGGL(k_prep_layer<float>, dim3(nb), dim3(256), 0, ctx->stream, means, lvars, O, I, (float*)mu_s,
                           (float*)var_s, ld_w, (float*)muT_s, (float*)varT_s, ld_wT, ctx->scratch);
    else if (dtype == VBNN_BF16)
        hipLaunchKernelGGL(k_prep_layer<bf16_t>, dim3(nb), dim3(256), 0, ctx->stream, means, lvars, O, I, (bf16_t*)mu_s,
                           (bf16_t*)var_s, ld_w, (bf16_t*)muT_s, (bf16_t*)varT_s, ld_wT, ctx->scratch);
    else { vbnn_set_error("unsupported dtype %d", dtype); return VBNN_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(k_prior_finish, dim3(1), dim3(256), 0, ctx->stream, ctx->scratch, nb, O * I, stats);
    return vbnn_check_launch("k_prep_layer");
    VBNN_API_END
}

// vbnn_prepare: every layer's sweep, then ONE finish kernel: block l < n finishes layer l's statistics, the blocks
// after them copy-pack the extra matrix (and its transpose).
struct PrepFinishArgs {
    const double* partial[8]; int nb[8]; int64_t W[8]; double* stats[8]; int n;
    const float* src; int64_t rows, cols, ld_src; void* dst; int64_t ld_dst; void* dstT; int64_t ld_dstT;
};
template <typename T>
__global__ __launch_bounds__(256) void k_prepare_finish(PrepFinishArgs a) {
    __shared__ double sh[4];
    if ((int)blockIdx.x < a.n) {
        prior_finish(a.partial[blockIdx.x], a.nb[blockIdx.x], a.W[blockIdx.x], a.stats[blockIdx.x], sh);
        return;
    }
    T* dst = (T*)a.dst;
    T* dstT = (T*)a.dstT;
    const int64_t total = a.rows * a.cols;
    for (int64_t t = (int64_t)(blockIdx.x - a.n) * 256 + threadIdx.x; t < total; t += (int64_t)(gridDim.x - a.n) * 256) {
        const int64_t r = t / a.cols, c = t - r * a.cols;
        const T v = Elt<T>::to(a.src[r * a.ld_src + c]);
        if (dst) dst[r * a.ld_dst + c] = v;
        if (dstT) dstT[c * a.ld_dstT + r] = v;
    }
}

extern "C" int vbnn_prepare(vbnn_ctx* ctx, int dtype, int n_layers, const vbnn_prep_desc* layers, const vbnn_pack_desc* extra) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && (layers || n_layers == 0), "null argument");
    VBNN_REQUIRE(n_layers >= 0 && n_layers <= 8, "n_layers (0..8)");
    VBNN_REQUIRE(dtype == VBNN_F32 || dtype == VBNN_BF16, "dtype");
    VBNN_REQUIRE((size_t)n_layers * 4096 <= ctx->scratch_doubles, "scratch");
    PrepFinishArgs fa{};
    fa.n = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        const vbnn_prep_desc& d = layers[l];
        VBNN_REQUIRE(d.means && d.lvars && d.mu_s && d.var_s && d.stats, "null layer argument");
        VBNN_REQUIRE((d.muT_s == nullptr) == (d.varT_s == nullptr), "muT_s and varT_s go together");
        VBNN_REQUIRE(d.O > 0 && d.I > 0 && d.ld_w >= d.I && (!d.muT_s || d.ld_wT >= d.O), "layer shape");
        const int64_t ntiles = ((d.O + 63) / 64) * ((d.I + 63) / 64);
        const int nb = (int)(ntiles < 2048 ? ntiles : 2048);
        double* partial = ctx->scratch + (size_t)l * 4096;
        if (dtype == VBNN_F32)
            hipLaunchKernelGGL(k_prep_layer<float>, dim3(nb), dim3(256), 0, ctx->stream, d.means, d.lvars, d.O, d.I, (float*)d.mu_s,
                               (float*)d.var_s, d.ld_w, (float*)d.muT_s, (float*)d.varT_s, d.ld_wT, partial);
        else
            hipLaunchKernelGGL(k_prep_layer<bf16_t>, dim3(nb), dim3(256), 0, ctx->stream, d.means, d.lvars, d.O, d.I,
                               (bf16_t*)d.mu_s, (bf16_t*)d.var_s, d.ld_w, (bf16_t*)d.muT_s, (bf16_t*)d.varT_s, d.ld_wT, partial);
        fa.partial[l] = partial; fa.nb[l] = nb; fa.W[l] = d.O * d.I; fa.stats[l] = d.stats;
    }
    int pack_blocks = 0;
    if (extra) {
        VBNN_REQUIRE(extra->src && extra->rows > 0 && extra->cols > 0 && extra->ld_src >= extra->cols, "extra matrix");
        VBNN_REQUIRE(!extra->dst || extra->ld_dst >= extra->cols, "extra ld_dst");
        VBNN_REQUIRE(!extra->dstT || extra->ld_dstT >= extra->rows, "extra ld_dstT");
        fa.src = extra->src; fa.rows = extra->rows; fa.cols = extra->cols; fa.ld_src = extra->ld_src;
        fa.dst = extra->dst; fa.ld_dst = extra->ld_dst; fa.dstT = extra->dstT; fa.ld_dstT = extra->ld_dstT;
        pack_blocks = grid_for(extra->rows * extra->cols, 1024);
    }
    if (n_layers + pack_blocks > 0) {
        if (dtype == VBNN_F32) hipLaunchKernelGGL(k_prepare_finish<float>, dim3(n_layers + pack_blocks), dim3(256), 0, ctx->stream, fa);
        else hipLaunchKernelGGL(k_prepare_finish<bf16_t>, dim3(n_layers + pack_blocks), dim3(256), 0, ctx->stream, fa);
    }
    return vbnn_check_launch("vbnn_prepare");
    VBNN_API_END
}

// ---------------------------------------------------------------------------------- update (+ next step's sweep)
// VBLinear:update for one layer as ONE sweep (vbnn_update): Adam on means and lvars from the total gradients, the new
// parameters' GEMM shadows (+ transposes) and prior sums, and the sums behind the 14 logged series -- 32 B read +
// 24 B (+ 4..8 B of shadows) written per weight, HBM-bound. Tiling of k_prep_layer.
#ifndef VBNN_NT_UPDATE
#define VBNN_NT_UPDATE 1
#endif
constexpr int UPD_NSUM = 16;      // per-block partials: 12 sums, then min / max of the new variances and of the new means
struct UpdLayer {
    float* means; float* lvars; int64_t O, I;
    void* mu_s; void* var_s; int64_t ld_w; void* muT_s; void* varT_s; int64_t ld_wT;
    const double* stats;
    const float* g_mu; const float* g_lv; float* m_mu; float* v_mu; float* m_lv; float* v_lv;
    float b1_mu, b2_mu, eps_mu, step_mu, b1_lv, b2_lv, eps_lv, step_lv, B;
    float kl_add;          // weight of the KL gradient added HERE, from the fp32 parameters (vbnn_update_desc.kl_add); 0: the gradients are totals
    double* partial;
};
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}
// reduces the 16 per-thread values over the block's 256 threads (sums, 2 x (min, max)); thread 0 writes them
__device__ __forceinline__ void upd_block_reduce(double (&a)[UPD_NSUM], double* out, double (*sh)[4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < UPD_NSUM; ++k) {
        const double r = (k < 12) ? wave_sum(a[k]) : ((k & 1) ? wave_max(a[k]) : wave_min(a[k]));
        if (lane == 0) sh[k][wave] = r;
    }
    __syncthreads();
    if (threadIdx.x < UPD_NSUM) {
        const int k = threadIdx.x;
        const double* v = sh[k];
        out[k] = (k < 12) ? v[0] + v[1] + v[2] + v[3] : ((k & 1) ? fmax(fmax(v[0], v[1]), fmax(v[2], v[3])) : fmin(fmin(v[0], v[1]), fmin(v[2], v[3])));
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_vb_update(UpdLayer a) {
    __shared__ float tm[64][65];
    __shared__ float tv[64][65];
    __shared__ double sh[UPD_NSUM][4];
    const int64_t O = a.O, I = a.I;
    const int64_t tiles_c = (I + 63) / 64, tiles_r = (O + 63) / 64;
    const int64_t ntiles = tiles_c * tiles_r;
    T* mu_s = (T*)a.mu_s; T* var_s = (T*)a.var_s; T* muT_s = (T*)a.muT_s; T* varT_s = (T*)a.varT_s;
    const bool vec_in = ((I & 3) == 0) && ((((uintptr_t)a.means | (uintptr_t)a.lvars | (uintptr_t)a.g_mu | (uintptr_t)a.g_lv |
                                             (uintptr_t)a.m_mu | (uintptr_t)a.v_mu | (uintptr_t)a.m_lv | (uintptr_t)a.v_lv) & 15u) == 0);
    const bool vec_t = muT_s && ((a.ld_wT & 3) == 0);
    // the KL parts of the logged norms, from the pre-update parameters and statistics (VBLinear.lua:91,96)
    const float var_hat = (float)a.stats[2];
    const float k_mu = 1.0f / (a.B * var_hat), k_lv = 1.0f / (2.0f * a.B), inv_vh = 1.0f / var_hat;
    double acc[UPD_NSUM];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.0;
    acc[12] = acc[14] = 1e300; acc[13] = acc[15] = -1e300;
    // four consecutive weights of row r (columns c .. c + valid - 1, flat index base): everything the sweep does with them.
    // (VBNN_NT_UPDATE: parameters, gradients and Adam moments are this sweep's alone -- nontemporal both ways -- so that what it
    // leaves in the caches is the operand shadows the next forward reads.)
    auto ld4 = [&](const float* p, float (&o)[4], int valid) {
#if VBNN_NT_UPDATE
        if (vec_in && valid == 4) { const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3]; return; }
#endif
        load4<float>(p, o, valid, vec_in);
    };
    auto st4 = [&](float* p, const float (&o)[4], int valid) {
#if VBNN_NT_UPDATE
        if (vec_in && valid == 4) { __builtin_nontemporal_store(f32x4{o[0], o[1], o[2], o[3]}, reinterpret_cast<f32x4*>(p)); return; }
#endif
        store4<float>(p, o[0], o[1], o[2], o[3], valid, vec_in);
    };
    auto elem4 = [&](const int64_t base, const int64_t r, const int64_t c, const int valid, float (&m)[4], float (&v)[4]) {
                float l[4], gm[4], gl[4], mm[4], vm[4], ml[4], vl[4];
                ld4(a.means + base, m, valid);
                ld4(a.lvars + base, l, valid);
                ld4(a.g_mu + base, gm, valid);
                ld4(a.g_lv + base, gl, valid);
                ld4(a.m_mu + base, mm, valid);
                ld4(a.v_mu + base, vm, valid);
                ld4(a.m_lv + base, ml, valid);
                ld4(a.v_lv + base, vl, valid);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < valid) {
                        // the two parts of each total gradient (logging only)
                        const float mlc = k_mu * m[e], vlc = k_lv * fmaf(expf(l[e]), inv_vh, -1.0f);
                        if (a.kl_add != 0.f) {            // likelihood-only gradients in: the KL part joins them here, in fp32
                            gm[e] = fmaf(a.kl_add, mlc, gm[e]);
                            gl[e] = fmaf(a.kl_add, vlc, gl[e]);
                        }
                        const float mle = gm[e] - mlc, vle = gl[e] - vlc;
                        acc[6] += (double)mlc * mlc; acc[7] += (double)mle * mle;
                        acc[8] += (double)vlc * vlc; acc[9] += (double)vle * vle;
                        // optim.adam, the arithmetic of k_adam (optim.hip)
                        mm[e] = a.b1_mu * mm[e] + (1.0f - a.b1_mu) * gm[e];
                        vm[e] = a.b2_mu * vm[e] + (1.0f - a.b2_mu) * gm[e] * gm[e];
                        const float um = a.step_mu * mm[e] / (sqrtf(vm[e]) + a.eps_mu);
                        m[e] -= um;
                        ml[e] = a.b1_lv * ml[e] + (1.0f - a.b1_lv) * gl[e];
                        vl[e] = a.b2_lv * vl[e] + (1.0f - a.b2_lv) * gl[e] * gl[e];
                        const float ul = a.step_lv * ml[e] / (sqrtf(vl[e]) + a.eps_lv);
                        l[e] -= ul;
                        v[e] = expf(l[e]);
                        acc[0] += (double)__fadd_rn(v[e], __fmul_rn(m[e], m[e])); acc[1] += (double)l[e];      // as k_prep_layer
                        acc[2] += (double)um * um; acc[3] += (double)m[e] * m[e];
                        acc[4] += (double)ul * ul; acc[5] += (double)l[e] * l[e];
                        acc[10] += (double)v[e]; acc[11] += (double)m[e];
                        acc[12] = fmin(acc[12], (double)v[e]); acc[13] = fmax(acc[13], (double)v[e]);
                        acc[14] = fmin(acc[14], (double)m[e]); acc[15] = fmax(acc[15], (double)m[e]);
                    }
                st4(a.means + base, m, valid);
                st4(a.lvars + base, l, valid);
                st4(a.m_mu + base, mm, valid);
                st4(a.v_mu + base, vm, valid);
                st4(a.m_lv + base, ml, valid);
                st4(a.v_lv + base, vl, valid);
                store4<T>(mu_s + r * a.ld_w + c, m[0], m[1], m[2], m[3], valid, true);
                store4<T>(var_s + r * a.ld_w + c, v[0], v[1], v[2], v[3], valid, true);
    };
    if (!muT_s && vec_in && (int64_t)gridDim.x * 1024 <= O * I) {
        // FLAT form (no transposed shadows to build: the K-major configurations): the eight fp32 streams are walked as flat
        // arrays, 4 KB per workgroup and stream at a time -- a wave-instruction is ONE contiguous KiB instead of four rows
        // x 256 B of a 64 x 64 tile (16 KB apart in a 4096-wide layer: four DRAM pages per instruction and stream)
        const int64_t total = O * I;
        for (int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; base < total; base += (int64_t)gridDim.x * 1024) {
            const int64_t r = base / I, c = base - r * I;          // I % 4 == 0 (vec_in): the four weights share a row
            float m[4], v[4];
            elem4(base, r, c, 4, m, v);
        }
        upd_block_reduce(acc, a.partial + (size_t)blockIdx.x * UPD_NSUM, sh);
        return;
    }
    for (int64_t tI = blockIdx.x; tI < ntiles; tI += gridDim.x) {
        const int64_t r0 = (tI / tiles_c) * 64, c0 = (tI % tiles_c) * 64;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = threadIdx.x + 256 * k;
            const int rr = idx >> 4, c4 = (idx & 15) * 4;
            const int64_t r = r0 + rr, c = c0 + c4;
            float m[4] = {0.f, 0.f, 0.f, 0.f}, v[4] = {0.f, 0.f, 0.f, 0.f};
            const int valid = (r < O) ? (int)min((int64_t)4, I - c) : 0;
            if (valid > 0) elem4(r * I + c, r, c, valid, m, v);
            if (muT_s) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { tm[rr][c4 + e] = m[e]; tv[rr][c4 + e] = v[e]; }
            }
        }
        if (muT_s) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = threadIdx.x + 256 * k;
                const int cc = idx >> 4, r4 = (idx & 15) * 4;
                const int64_t c = c0 + cc, r = r0 + r4;
                const int valid = (c < I) ? (int)min((int64_t)4, O - r) : 0;
                if (valid > 0) {
                    store4<T>(muT_s + c * a.ld_wT + r, tm[r4][cc], tm[r4 + 1][cc], tm[r4 + 2][cc], tm[r4 + 3][cc], valid, vec_t);
                    store4<T>(varT_s + c * a.ld_wT + r, tv[r4][cc], tv[r4 + 1][cc], tv[r4 + 2][cc], tv[r4 + 3][cc], valid, vec_t);
                }
            }
            __syncthreads();
        }
    }
    upd_block_reduce(acc, a.partial + (size_t)blockIdx.x * UPD_NSUM, sh);
}

struct UpdFinishArgs {
    const double* partial[8]; int nb[8]; int64_t W[8]; double* stats[8]; double* log14[8];
    float* bias[8]; const float* grad_bias[8]; int64_t O[8]; float lr_bias[8]; int n;
    const float* src; int64_t rows, cols, ld_src; void* dst; int64_t ld_dst; void* dstT; int64_t ld_dstT;
};
template <typename T>
__global__ __launch_bounds__(256) void k_update_finish(UpdFinishArgs a) {
    __shared__ double sh[UPD_NSUM][4];
    if ((int)blockIdx.x < a.n) {
        const int l = blockIdx.x;
        double acc[UPD_NSUM];
#pragma unroll
        for (int k = 0; k < 12; ++k) acc[k] = 0.0;
        acc[12] = acc[14] = 1e300; acc[13] = acc[15] = -1e300;
        // two partial rows' loads in flight per thread (r05: this launch was ~22 us of the training step, most of it dependent
        // round trips: eight rows per thread one after the other, then sixteen bias elements one after the other); the rows are
        // still ADDED in order: the same bits
        const int nb = a.nb[l];
        const double* pl = a.partial[l];
        for (int b = threadIdx.x; b < nb; b += 512) {
            const bool two = b + 256 < nb;
            double r0[UPD_NSUM], r1[UPD_NSUM];
#pragma unroll
            for (int k = 0; k < UPD_NSUM; ++k) r0[k] = pl[(size_t)b * UPD_NSUM + k];
#pragma unroll
            for (int k = 0; k < UPD_NSUM; ++k) r1[k] = two ? pl[(size_t)(b + 256) * UPD_NSUM + k] : ((k < 12) ? 0.0 : ((k & 1) ? -1e300 : 1e300));
#pragma unroll
            for (int k = 0; k < 12; ++k) acc[k] += r0[k];
            acc[12] = fmin(acc[12], r0[12]); acc[13] = fmax(acc[13], r0[13]);
            acc[14] = fmin(acc[14], r0[14]); acc[15] = fmax(acc[15], r0[15]);
            if (two) {
#pragma unroll
                for (int k = 0; k < 12; ++k) acc[k] += r1[k];
                acc[12] = fmin(acc[12], r1[12]); acc[13] = fmax(acc[13], r1[13]);
                acc[14] = fmin(acc[14], r1[14]); acc[15] = fmax(acc[15], r1[15]);
            }
        }
        __shared__ double tot[UPD_NSUM];
        upd_block_reduce(acc, tot, sh);
        __syncthreads();
        if (threadIdx.x == 0) {
            const double W = (double)a.W[l];
            const double var_hat_old = a.stats[l][2];
            if (a.log14[l]) {                                   // VBLinear.lua:149-164, in the order of the Log:add calls
                double* g = a.log14[l];
                const double nl = sqrt(tot[5]), nm = sqrt(tot[3]);
                g[0] = sqrt(tot[8]) / nl; g[1] = sqrt(tot[9]) / nl; g[2] = sqrt(tot[6]) / nm; g[3] = sqrt(tot[7]) / nm;
                g[4] = tot[12]; g[5] = tot[13]; g[6] = tot[10] / W; g[7] = var_hat_old;
                const double mean = tot[11] / W;
                g[8] = mean; g[9] = W > 1.0 ? sqrt(fmax(0.0, (tot[3] - W * mean * mean) / (W - 1.0))) : 0.0;
                g[10] = tot[14]; g[11] = tot[15]; g[12] = sqrt(tot[2]) / nm; g[13] = sqrt(tot[4]) / nl;
            }
            double* st = a.stats[l];                            // as prior_finish, of the NEW parameters
            st[0] = tot[0]; st[1] = tot[1]; st[2] = (1.0 / W) * tot[0]; st[3] = W;
        }
        if (a.bias[l] && (int)gridDim.x == a.n)                 // optim.sgd on the bias (VBLinear.lua:125-128): here only when there are no
            for (int64_t o = threadIdx.x; o < a.O[l]; o += 256) a.bias[l][o] = fmaf(-a.lr_bias[l], a.grad_bias[l][o], a.bias[l][o]);   // packing blocks to share it
        return;
    }
    // the layers' bias steps, spread over the packing blocks (one element per thread and stride: independent elements)
    for (int l = 0; l < a.n; ++l)
        if (a.bias[l])
            for (int64_t o = (int64_t)(blockIdx.x - a.n) * 256 + threadIdx.x; o < a.O[l]; o += (int64_t)(gridDim.x - a.n) * 256)
                a.bias[l][o] = fmaf(-a.lr_bias[l], a.grad_bias[l][o], a.bias[l][o]);
    T* dst = (T*)a.dst;
    T* dstT = (T*)a.dstT;
    const int64_t total = a.rows * a.cols;
    for (int64_t t = (int64_t)(blockIdx.x - a.n) * 256 + threadIdx.x; t < total; t += (int64_t)(gridDim.x - a.n) * 256) {
        const int64_t r = t / a.cols, c = t - r * a.cols;
        const T v = Elt<T>::to(a.src[r * a.ld_src + c]);
        if (dst) dst[r * a.ld_dst + c] = v;
        if (dstT) dstT[c * a.ld_dstT + r] = v;
    }
}

static inline float adam_step_size(const vbnn_adam_cfg& c, float* b1_out) {
    const double b1t = (double)c.beta1 * pow((double)c.lambda, (double)(c.t - 1));       // as vbnn_adam_step
    const double bc1 = 1.0 - pow((double)c.beta1, (double)c.t), bc2 = 1.0 - pow((double)c.beta2, (double)c.t);
    *b1_out = (float)b1t;
    return (float)((double)c.lr * sqrt(bc2) / bc1);
}

extern "C" int vbnn_update(vbnn_ctx* ctx, int dtype, int n_layers, const vbnn_update_desc* layers, const vbnn_pack_desc* extra) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && (layers || n_layers == 0), "null argument");
    VBNN_REQUIRE(n_layers >= 0 && n_layers <= 8, "n_layers (0..8)");
    VBNN_REQUIRE(dtype == VBNN_F32 || dtype == VBNN_BF16, "dtype");
    constexpr int MAXB = 2048;
    VBNN_REQUIRE((size_t)n_layers * MAXB * UPD_NSUM <= ctx->scratch_doubles, "scratch");
    UpdFinishArgs fa{};
    fa.n = n_layers;
    // (layer order. r05 A/B, two rounds on one box: the layer whose gradients were written LAST first -- they might still sit in the
    // Infinity Cache -- 1.025 / 1.029 ms per training step against 1.008 / 1.003; with accGradParameters' gradient stores plain instead
    // of nontemporal as well, 1.033 / 1.024, and the step without update 0.765 against 0.750: neither is kept.)
    for (int l = 0; l < n_layers; ++l) {
        const vbnn_update_desc& d = layers[l];
        VBNN_REQUIRE(d.means && d.lvars && d.mu_s && d.var_s && d.stats, "null layer argument");
        VBNN_REQUIRE(d.grad_mu && d.grad_lv && d.m_mu && d.v_mu && d.m_lv && d.v_lv, "null gradient / Adam state");
        VBNN_REQUIRE((d.muT_s == nullptr) == (d.varT_s == nullptr), "muT_s and varT_s go together");
        VBNN_REQUIRE(d.O > 0 && d.I > 0 && d.ld_w >= d.I && (!d.muT_s || d.ld_wT >= d.O), "layer shape");
        VBNN_REQUIRE((d.bias == nullptr) == (d.grad_bias == nullptr), "bias and grad_bias go together");
        VBNN_REQUIRE(d.B > 0, "B");
        for (const vbnn_adam_cfg* c : {&d.mu, &d.lv})
            VBNN_REQUIRE(c->t >= 1 && c->beta1 >= 0 && c->beta1 < 1 && c->beta2 >= 0 && c->beta2 < 1 && c->lr >= 0 && c->eps >= 0 &&
                         c->lambda > 0 && c->lambda <= 1, "Adam hyper-parameters (t counts from 1)");
        const int64_t ntiles = ((d.O + 63) / 64) * ((d.I + 63) / 64);
        const int nb = (int)(ntiles < MAXB ? ntiles : MAXB);
        UpdLayer a{};
        a.means = d.means; a.lvars = d.lvars; a.O = d.O; a.I = d.I;
        a.mu_s = d.mu_s; a.var_s = d.var_s; a.ld_w = d.ld_w; a.muT_s = d.muT_s; a.varT_s = d.varT_s; a.ld_wT = d.ld_wT;
        a.stats = d.stats; a.g_mu = d.grad_mu; a.g_lv = d.grad_lv;
        a.m_mu = d.m_mu; a.v_mu = d.v_mu; a.m_lv = d.m_lv; a.v_lv = d.v_lv;
        a.step_mu = adam_step_size(d.mu, &a.b1_mu); a.b2_mu = d.mu.beta2; a.eps_mu = d.mu.eps;
        a.step_lv = adam_step_size(d.lv, &a.b1_lv); a.b2_lv = d.lv.beta2; a.eps_lv = d.lv.eps;
        a.B = d.B;
        a.kl_add = d.kl_add;
        a.partial = ctx->scratch + (size_t)l * MAXB * UPD_NSUM;
        if (dtype == VBNN_F32) hipLaunchKernelGGL(k_vb_update<float>, dim3(nb), dim3(256), 0, ctx->stream, a);
        else hipLaunchKernelGGL(k_vb_update<bf16_t>, dim3(nb), dim3(256), 0, ctx->stream, a);
        fa.partial[l] = a.partial; fa.nb[l] = nb; fa.W[l] = d.O * d.I; fa.stats[l] = d.stats; fa.log14[l] = d.log14;
        fa.bias[l] = d.bias; fa.grad_bias[l] = d.grad_bias; fa.O[l] = d.O; fa.lr_bias[l] = d.lr_bias;
    }
    int pack_blocks = 0;
    if (extra) {
        VBNN_REQUIRE(extra->src && extra->rows > 0 && extra->cols > 0 && extra->ld_src >= extra->cols, "extra matrix");
        VBNN_REQUIRE(!extra->dst || extra->ld_dst >= extra->cols, "extra ld_dst");
        VBNN_REQUIRE(!extra->dstT || extra->ld_dstT >= extra->rows, "extra ld_dstT");
        fa.src = extra->src; fa.rows = extra->rows; fa.cols = extra->cols; fa.ld_src = extra->ld_src;
        fa.dst = extra->dst; fa.ld_dst = extra->ld_dst; fa.dstT = extra->dstT; fa.ld_dstT = extra->ld_dstT;
        pack_blocks = grid_for(extra->rows * extra->cols, 1024);
    }
    if (n_layers + pack_blocks > 0) {
        if (dtype == VBNN_F32) hipLaunchKernelGGL(k_update_finish<float>, dim3(n_layers + pack_blocks), dim3(256), 0, ctx->stream, fa);
        else hipLaunchKernelGGL(k_update_finish<bf16_t>, dim3(n_layers + pack_blocks), dim3(256), 0, ctx->stream, fa);
    }
    return vbnn_check_launch("vbnn_update");
    VBNN_API_END
}

// ---------------------------------------------------------------------------------- pack_input
// The minibatch as GEMM operands in one pass: x, x.x (of the ROUNDED x, as every epilogue produces it) and both
// transposes. Same tiling as k_prep_layer: 4 elements per thread, LDS transpose.
template <typename T>
__global__ __launch_bounds__(256) void k_pack_input(const float* __restrict__ src, int64_t ld_src, int64_t N, int64_t I, T* x_s,
                                                    T* x2_s, int64_t ld_x, T* xT_s, T* x2T_s, int64_t ld_xT, int64_t rpd) {
    __shared__ float ta[64][65];
    __shared__ float tb[64][65];
    const int64_t tiles_c = (I + 63) / 64, tiles_r = (N + 63) / 64;
    const int64_t ntiles = tiles_c * tiles_r;
    const bool vec_in = ((ld_src & 3) == 0) && (((uintptr_t)src & 15u) == 0);
    const bool vec_t = (ld_xT & 3) == 0;
    for (int64_t tI = blockIdx.x; tI < ntiles; tI += gridDim.x) {
        const int64_t r0 = (tI / tiles_c) * 64, c0 = (tI % tiles_c) * 64;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = threadIdx.x + 256 * k;
            const int rr = idx >> 4, c4 = (idx & 15) * 4;
            const int64_t r = r0 + rr, c = c0 + c4;
            float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
            const int valid = (r < N) ? (int)min((int64_t)4, I - c) : 0;
            if (valid > 0) {
                load4<float>(src + (rpd > 0 ? r % rpd : r) * ld_src + c, a, valid, vec_in);     // stacked draws: row n is minibatch row n % rpd
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float ar = Elt<T>::from(Elt<T>::to(a[e])); b[e] = ar * ar; }
                store4<T>(x_s + r * ld_x + c, a[0], a[1], a[2], a[3], valid, true);
                if (x2_s) store4<T>(x2_s + r * ld_x + c, b[0], b[1], b[2], b[3], valid, true);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { ta[rr][c4 + e] = a[e]; tb[rr][c4 + e] = b[e]; }
        }
        if (xT_s) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = threadIdx.x + 256 * k;
                const int cc = idx >> 4, r4 = (idx & 15) * 4;
                const int64_t c = c0 + cc, r = r0 + r4;
                const int valid = (c < I) ? (int)min((int64_t)4, N - r) : 0;
                if (valid > 0) {
                    store4<T>(xT_s + c * ld_xT + r, ta[r4][cc], ta[r4 + 1][cc], ta[r4 + 2][cc], ta[r4 + 3][cc], valid, vec_t);
                    if (x2T_s)
                        store4<T>(x2T_s + c * ld_xT + r, tb[r4][cc], tb[r4 + 1][cc], tb[r4 + 2][cc], tb[r4 + 3][cc], valid, vec_t);
                }
            }
            __syncthreads();
        }
    }
}

extern "C" int vbnn_pack_input(vbnn_ctx* ctx, int dtype, const float* src, int64_t ld_src, int64_t N, int64_t I, void* x_s,
                               void* x2_s, int64_t ld_x, void* xT_s, void* x2T_s, int64_t ld_xT, int64_t rows_per_draw) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && src && x_s, "null argument");
    VBNN_REQUIRE(rows_per_draw >= 0, "rows_per_draw");
    VBNN_REQUIRE(N > 0 && I > 0 && ld_src >= I && ld_x >= I && ld_x % 4 == 0, "shape");
    VBNN_REQUIRE(!x2T_s || xT_s, "x2T_s needs xT_s");
    VBNN_REQUIRE(!xT_s || ld_xT >= N, "ld_xT");
    const int64_t ntiles = ((N + 63) / 64) * ((I + 63) / 64);
    const int nb = (int)(ntiles < 4096 ? ntiles : 4096);
    if (dtype == VBNN_F32)
        hipLaunchKernelGGL(k_pack_input<float>, dim3(nb), dim3(256), 0, ctx->stream, src, ld_src, N, I, (float*)x_s, (float*)x2_s,
                           ld_x, (float*)xT_s, (float*)x2T_s, ld_xT, rows_per_draw);
    else if (dtype == VBNN_BF16)
        hipLaunchKernelGGL(k_pack_input<bf16_t>, dim3(nb), dim3(256), 0, ctx->stream, src, ld_src, N, I, (bf16_t*)x_s,
                           (bf16_t*)x2_s, ld_x, (bf16_t*)xT_s, (bf16_t*)x2T_s, ld_xT, rows_per_draw);
    else { vbnn_set_error("unsupported dtype %d", dtype); return VBNN_ERR_UNSUPPORTED; }
    return vbnn_check_launch("k_pack_input");
    VBNN_API_END
}

// ---- the optional bf16 exchange's casts (include/vbnn_hip.h, vbnn_cast_grads): 8 elements per thread, 16-byte accesses
template <bool TO_BF16>
__global__ __launch_bounds__(256) void k_cast_grads(const void* __restrict__ src, void* __restrict__ dst, int64_t n, int vec) {
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i0 >= n) return;
    if (vec && i0 + 8 <= n) {
        if constexpr (TO_BF16) {
            const f32x4 a = reinterpret_cast<const f32x4*>(static_cast<const float*>(src) + i0)[0];
            const f32x4 b = reinterpret_cast<const f32x4*>(static_cast<const float*>(src) + i0)[1];
            reinterpret_cast<bf16x8*>(static_cast<bf16_t*>(dst) + i0)[0] =
                bf16x8{(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
        } else {
            const bf16x8 v = reinterpret_cast<const bf16x8*>(static_cast<const bf16_t*>(src) + i0)[0];
            reinterpret_cast<f32x4*>(static_cast<float*>(dst) + i0)[0] = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
            reinterpret_cast<f32x4*>(static_cast<float*>(dst) + i0)[1] = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
        }
    } else {
        for (int64_t i = i0; i < n && i < i0 + 8; ++i) {
            if constexpr (TO_BF16) static_cast<bf16_t*>(dst)[i] = (bf16_t)static_cast<const float*>(src)[i];
            else static_cast<float*>(dst)[i] = (float)static_cast<const bf16_t*>(src)[i];
        }
    }
}

extern "C" int vbnn_cast_grads(vbnn_ctx* ctx, int to_bf16, const void* src, void* dst, int64_t n) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && src && dst && n > 0, "argument");
    const int vec = (((uintptr_t)src | (uintptr_t)dst) & 15u) == 0;      // (odd bucket offsets of test-sized nets: scalar path)
    const unsigned nb = (unsigned)((n + 2047) / 2048);
    if (to_bf16) hipLaunchKernelGGL(k_cast_grads<true>, dim3(nb), dim3(256), 0, ctx->stream, src, dst, n, vec);
    else hipLaunchKernelGGL(k_cast_grads<false>, dim3(nb), dim3(256), 0, ctx->stream, src, dst, n, vec);
    return vbnn_check_launch("k_cast_grads");
    VBNN_API_END
}

// ---- the sharded-update exchange's two small device-side steps (include/vbnn_hip.h) -----------------------------------
// (1) a layer's prior statistics from the per-rank statistics of its row slices: sums in RANK order, var_hat by the very
// expression k_update_finish uses -- every rank computes the same four doubles from the same gathered parts.
struct StatsCombineArgs { double* stats[8]; int n_layers, world; const double* parts; };
__global__ void k_stats_combine(StatsCombineArgs a) {
    const int l = threadIdx.x;
    if (l >= a.n_layers) return;
    double s0 = 0.0, s1 = 0.0, w = 0.0;
    for (int r = 0; r < a.world; ++r) {
        const double* p = a.parts + ((size_t)r * a.n_layers + l) * 4;
        s0 += p[0]; s1 += p[1]; w += p[3];
    }
    double* st = a.stats[l];
    st[0] = s0; st[1] = s1; st[2] = (1.0 / w) * s0; st[3] = w;
}
extern "C" int vbnn_stats_combine(vbnn_ctx* ctx, int n_layers, int world, const double* parts, double* const* stats) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && parts && stats && n_layers >= 1 && n_layers <= 8 && world >= 1, "argument (1..8 layers)");
    StatsCombineArgs a{};
    a.n_layers = n_layers; a.world = world; a.parts = parts;
    for (int l = 0; l < n_layers; ++l) { VBNN_REQUIRE(stats[l], "null stats"); a.stats[l] = stats[l]; }
    hipLaunchKernelGGL(k_stats_combine, dim3(1), dim3(64), 0, ctx->stream, a);
    return vbnn_check_launch("k_stats_combine");
    VBNN_API_END
}

// (2) the transposed copy of a packed operand (rows x ld_src -> cols x ld_dst, element type dtype): layers whose gradInput GEMM
// wants transposed shadows rebuild them locally from the gathered mu_s / var_s (a row slice of a shadow is a COLUMN range of its
// transpose: not something an all-gather can deliver). 64 x 64 tiles through LDS.
template <typename T>
__global__ __launch_bounds__(256) void k_transpose_packed(const T* __restrict__ src, int64_t ld_src, int64_t rows, int64_t cols, T* dst, int64_t ld_dst) {
    __shared__ T tile[64][65];
    const int64_t tiles_c = (cols + 63) / 64;
    const int64_t r0 = (blockIdx.x / tiles_c) * 64, c0 = (blockIdx.x % tiles_c) * 64;
    for (int k = threadIdx.x; k < 4096; k += 256) {
        const int rr = k >> 6, cc = k & 63;
        tile[rr][cc] = (r0 + rr < rows && c0 + cc < cols) ? src[(r0 + rr) * ld_src + c0 + cc] : Elt<T>::to(0.f);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 4096; k += 256) {
        const int cc = k >> 6, rr = k & 63;
        if (c0 + cc < cols && r0 + rr < rows) dst[(c0 + cc) * ld_dst + r0 + rr] = tile[rr][cc];
    }
}
extern "C" int vbnn_transpose_packed(vbnn_ctx* ctx, int dtype, const void* src, int64_t ld_src, int64_t rows, int64_t cols, void* dst,
                                     int64_t ld_dst) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, "argument");
    const int64_t tiles = ((rows + 63) / 64) * ((cols + 63) / 64);
    VBNN_REQUIRE(tiles < (1ll << 31), "matrix too large");
    if (dtype == VBNN_F32)
        hipLaunchKernelGGL(k_transpose_packed<float>, dim3((unsigned)tiles), dim3(256), 0, ctx->stream, (const float*)src, ld_src, rows, cols, (float*)dst, ld_dst);
    else if (dtype == VBNN_BF16)
        hipLaunchKernelGGL(k_transpose_packed<bf16_t>, dim3((unsigned)tiles), dim3(256), 0, ctx->stream, (const bf16_t*)src, ld_src, rows, cols, (bf16_t*)dst, ld_dst);
    else { vbnn_set_error("unsupported dtype %d", dtype); return VBNN_ERR_UNSUPPORTED; }
    return vbnn_check_launch("k_transpose_packed");
    VBNN_API_END
}
