// optim.hip -- the parameter update that follows the hot path: VBLinear:update (VBLinear.lua:124-166) and
// mlp:update (mlp.lua:117-142) on the device, so gradients never leave HBM.
//
// optim.adam / optim.sgd are NOT vendored in the reference and carry no version pin; the author also ran a
// locally patched optim that returns the applied update as a third value (VBLinear.lua:135-144). What is
// implemented is the published torch/optim adam [recalled]:
//     m = b1 m + (1 - b1) g ;  v = b2 v + (1 - b2) g.g ;  x -= lr sqrt(1 - b2^t) / (1 - b1^t) . m / (sqrt(v) + eps)
// with the early variant's `lambda` decay of b1 (config.lua:52,56,61 hint at it) available as b1_t = b1 lambda^(t-1);
// and plain optim.sgd:  x -= lr g  (no momentum / decay: config.lua:51-54 sets only learningRate).
// One streaming pass per parameter tensor: 16 B read + 12 B written per element (HBM-bound).
#include "common.h"

__device__ __forceinline__ double opt_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// grad = g1 (+ g2): the reference adds the likelihood and KL parts with torch.add right before optim.adam
// (VBLinear.lua:131-134). norms (optional): partial[block][2] = { |update|^2, |x_new|^2 } for the norm ratios.
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ x, const float* __restrict__ g1, const float* __restrict__ g2,
                                              float* __restrict__ m, float* __restrict__ v, int64_t n, float b1, float b2,
                                              float eps, float step, double* partial) {
    __shared__ double sh[2][4];
    double su = 0.0, sx = 0.0;
    const bool vec = ((((uintptr_t)x | (uintptr_t)g1 | (uintptr_t)g2 | (uintptr_t)m | (uintptr_t)v) & 15u) == 0);
    const int64_t n4 = vec ? (n >> 2) : 0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n4; t += (int64_t)gridDim.x * 256) {
        f32x4 xv = reinterpret_cast<f32x4*>(x)[t];
        f32x4 gv = reinterpret_cast<const f32x4*>(g1)[t];
        if (g2) gv += reinterpret_cast<const f32x4*>(g2)[t];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[t], vv = reinterpret_cast<f32x4*>(v)[t];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            mv[j] = b1 * mv[j] + (1.0f - b1) * gv[j];
            vv[j] = b2 * vv[j] + (1.0f - b2) * gv[j] * gv[j];
            const float up = step * mv[j] / (sqrtf(vv[j]) + eps);
            xv[j] -= up;
            su += (double)up * up; sx += (double)xv[j] * xv[j];
        }
        reinterpret_cast<f32x4*>(x)[t] = xv;
        reinterpret_cast<f32x4*>(m)[t] = mv;
        reinterpret_cast<f32x4*>(v)[t] = vv;
    }
    for (int64_t t = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        const float g = g1[t] + (g2 ? g2[t] : 0.f);
        const float mv = b1 * m[t] + (1.0f - b1) * g;
        const float vv = b2 * v[t] + (1.0f - b2) * g * g;
        const float up = step * mv / (sqrtf(vv) + eps);
        const float xn = x[t] - up;
        m[t] = mv; v[t] = vv; x[t] = xn;
        su += (double)up * up; sx += (double)xn * xn;
    }
    if (partial) {
        su = opt_wave_sum(su); sx = opt_wave_sum(sx);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) { sh[0][wave] = su; sh[1][wave] = sx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            partial[blockIdx.x * 2] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
            partial[blockIdx.x * 2 + 1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
        }
    }
}
__global__ __launch_bounds__(256) void k_norm_finish(const double* partial, int nblocks, double* norms) {
    __shared__ double sh[2][4];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { a += partial[i * 2]; b += partial[i * 2 + 1]; }
    a = opt_wave_sum(a); b = opt_wave_sum(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[0][wave] = a; sh[1][wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        norms[0] = sqrt((sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]));      // torch.norm(update)
        norms[1] = sqrt((sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]));      // torch.norm(x)
    }
}
__global__ __launch_bounds__(256) void k_sgd(float* __restrict__ x, const float* __restrict__ g, int64_t n, float lr) {
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) x[t] = fmaf(-lr, g[t], x[t]);
}

extern "C" int vbnn_adam_step(vbnn_ctx* ctx, float* x, const float* grad, const float* grad2, float* m, float* v, int64_t n,
                              float lr, float beta1, float beta2, float eps, float lambda, int64_t t, double* norms_dev) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && x && grad && m && v, "null argument");
    VBNN_REQUIRE(n > 0 && t >= 1, "n > 0 and t >= 1 (t counts from 1, as state.t after its increment)");
    VBNN_REQUIRE(beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && lr >= 0 && eps >= 0 && lambda > 0 && lambda <= 1, "hyper-parameters");
    const double b1t = (double)beta1 * pow((double)lambda, (double)(t - 1));          // the decayed beta1 of the early variant
    const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
    const float step = (float)((double)lr * sqrt(bc2) / bc1);
    int64_t nb = (n / 4 + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 2048) nb = 2048;
    VBNN_REQUIRE(!norms_dev || (size_t)nb * 2 <= ctx->scratch_doubles, "scratch");
    hipLaunchKernelGGL(k_adam, dim3((unsigned)nb), dim3(256), 0, ctx->stream, x, grad, grad2, m, v, n, (float)b1t, beta2, eps, step,
                       norms_dev ? ctx->scratch : nullptr);
    if (norms_dev) hipLaunchKernelGGL(k_norm_finish, dim3(1), dim3(256), 0, ctx->stream, ctx->scratch, (int)nb, norms_dev);
    return vbnn_check_launch("k_adam");
    VBNN_API_END
}

extern "C" int vbnn_sgd_step(vbnn_ctx* ctx, float* x, const float* grad, int64_t n, float lr) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && x && grad && n > 0, "argument");
    int64_t nb = (n + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_sgd, dim3((unsigned)nb), dim3(256), 0, ctx->stream, x, grad, n, lr);
    return vbnn_check_launch("k_sgd");
    VBNN_API_END
}
