/*
 * vbnn_oracle.c -- CPU restatement of the reference's VBLinear hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product path (vbnn_amd/) may import, link
 * or execute this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do. PARITY UNPINNED: the reference (louissmit/VBNN) ships no
 * tests, golden vectors or fixtures, and no Lua/Torch7
 * interpreter exists in the build image, so this restatement is pinned only by
 * (a) the Random123 Philox known-answer vectors, (b) analytic known answers derived
 * from VBLinear.lua's formulas and (c) finite differences / PyTorch-CPU autograd
 * (tests/test_oracle.py). Each function cites the reference lines it follows.
 * "[recalled]" marks behaviour of un-vendored Torch7 `nn`/`torch` code.
 *
 * All tensors are dense row-major fp32 (main.lua:10 sets FloatTensor as default).
 * O = outputSize, I = inputSize, N = minibatch rows.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mfma).
 */
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include "../include/vbnn_philox.h"

#define VBO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ RNG fills */

/* Raw Philox4x32-10 block, for the Random123 known-answer test. */
VBO_API void vbo_philox_raw(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    vbnn_u32x4 r = vbnn_philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    memcpy(out, r.v, 16);
}

/* rows x cols standard normals, element (r, c) = lane c&3 of
 * vbnn_normal4(seed, stream, layer, draw, row0 + r, c >> 2). Stands in for
 * randomkit.normal(self.e, zeros, ones) at VBLinear.lua:55 (stream EPS, rows = O,
 * cols = I) and generates the LRT activation noise (stream ZETA, rows = N, cols = O). */
VBO_API void vbo_fill_normal(float* out, int64_t rows, int64_t cols, int64_t ld,
                             uint64_t seed, uint32_t stream, uint32_t layer, uint32_t draw,
                             int64_t row0) {
    for (int64_t r = 0; r < rows; ++r) {
        for (int64_t q = 0; q * 4 < cols; ++q) {
            vbnn_f32x4 z = vbnn_normal4(seed, stream, layer, draw, (uint32_t)(row0 + r), (uint32_t)q);
            for (int j = 0; j < 4 && q * 4 + j < cols; ++j) out[r * ld + q * 4 + j] = z.v[j];
        }
    }
}

VBO_API float vbo_det_logf(float x) { return vbnn_det_logf(x); }
VBO_API void vbo_det_sincos2pi(uint32_t k, float* c, float* s) { vbnn_det_sincos2pi(k, c, s); }

/* ------------------------------------------------------------------ GEMM cores
 * k-ordered fp32 fma chains: bitwise what v_mfma_f32_16x16x4_f32 produces when the
 * K loop runs in order (cdna_hip_programming.md, "FP32-input MFMA: numerics"), and a
 * legitimate evaluation order for the reference's sgemm (BLAS leaves it unspecified). */

/* C[m][n] = beta*C + alpha * sum_k A[m][k] * B[n][k]      (A: MxK, B: NxK, "NT") */
static void gemm_nt(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K,
                    float alpha, float beta) {
    for (int64_t m = 0; m < M; ++m)
        for (int64_t n = 0; n < N; ++n) {
            float acc = 0.f;
            const float* a = A + m * K; const float* b = B + n * K;
            for (int64_t k = 0; k < K; ++k) acc = fmaf(a[k], b[k], acc);
            const float prev = (beta == 0.f) ? 0.f : beta * C[m * N + n];
            C[m * N + n] = fmaf(alpha, acc, prev);
        }
}
/* C[m][n] = beta*C + alpha * sum_k A[k][m] * B[k][n]      (A: KxM, B: KxN, "TN") */
static void gemm_tn(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K,
                    float alpha, float beta) {
    float* acc = (float*)calloc((size_t)N, sizeof(float));
    for (int64_t m = 0; m < M; ++m) {
        memset(acc, 0, (size_t)N * sizeof(float));
        for (int64_t k = 0; k < K; ++k) {
            const float a = A[k * M + m]; const float* b = B + k * N;
            for (int64_t n = 0; n < N; ++n) acc[n] = fmaf(a, b[n], acc[n]);
        }
        for (int64_t n = 0; n < N; ++n) {
            const float prev = (beta == 0.f) ? 0.f : beta * C[m * N + n];
            C[m * N + n] = fmaf(alpha, acc[n], prev);
        }
    }
    free(acc);
}
/* C[m][n] = sum_k A[m][k] * B[k][n]                        (A: MxK, B: KxN, "NN") */
static void gemm_nn(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K) {
    for (int64_t m = 0; m < M; ++m) {
        float* c = C + m * N;
        for (int64_t n = 0; n < N; ++n) c[n] = 0.f;
        for (int64_t k = 0; k < K; ++k) {
            const float a = A[m * K + k]; const float* b = B + k * N;
            for (int64_t n = 0; n < N; ++n) c[n] = fmaf(a, b[n], c[n]);
        }
    }
}

/* ------------------------------------------------------------------ VBLinear (reference form) */

/* VBLinear:compute_prior -- VBLinear.lua:77-88.
 *   vars = exp(lvars) (:78); stdv = sqrt(vars) (:79); mu_hat = 0 (:81);
 *   mu_sqe = (means - mu_hat)^2 (:82); var_hat = (1/W) * sum(vars + mu_sqe) (:86).
 * torch.sum on a FloatTensor accumulates in double [recalled: TH accreal]. */
VBO_API void vbo_compute_prior(const float* means, const float* lvars, int64_t W,
                               float* vars, float* stdv, float* mu_sqe, double* var_hat) {
    double s = 0.0;
    for (int64_t i = 0; i < W; ++i) {
        const float v = expf(lvars[i]);
        const float sd = sqrtf(v);
        const float d = means[i] - 0.0f;
        const float q = d * d;
        if (vars) vars[i] = v;
        if (stdv) stdv[i] = sd;
        if (mu_sqe) mu_sqe[i] = q;
        s += (double)(v + q);                 /* torch.add(vars, mu_sqe) is an fp32 tensor */
    }
    *var_hat = (1.0 / (double)W) * s;
}

/* VBLinear:sample -- VBLinear.lua:49-64: w = means + stdv (.) e (:59); weight:copy(w) (:63).
 * `stdv` is the cache of the last compute_prior, exactly as in the reference. */
VBO_API void vbo_sample(const float* means, const float* stdv, const float* e, float* weight, int64_t W) {
    for (int64_t i = 0; i < W; ++i) {
        const float t = stdv[i] * e[i];       /* torch.cmul(stdv, e): a rounded temporary */
        weight[i] = means[i] + t;             /* torch.add(means, .) */
    }
}

/* VBLinear:clamp_to_map -- VBLinear.lua:105-107. */
VBO_API void vbo_clamp_to_map(const float* means, float* weight, int64_t W) {
    memcpy(weight, means, (size_t)W * sizeof(float));
}

/* inherited nn.Linear:updateOutput (VBLinear.lua:7) [recalled]:
 *   output = input * weight^T ; output += ones_N (x) bias. */
VBO_API void vbo_linear_forward(const float* x, const float* weight, const float* bias, float* y,
                                int64_t N, int64_t I, int64_t O) {
    gemm_nt(x, weight, y, N, O, I, 1.f, 0.f);
    if (bias)
        for (int64_t n = 0; n < N; ++n)
            for (int64_t o = 0; o < O; ++o) y[n * O + o] += bias[o];
}

/* inherited nn.Linear:updateGradInput (stub commented out at VBLinear.lua:109-110) [recalled]:
 *   gradInput = gradOutput * weight. */
VBO_API void vbo_linear_grad_input(const float* g, const float* weight, float* gx,
                                   int64_t N, int64_t I, int64_t O) {
    gemm_nn(g, weight, gx, N, I, O);
}

/* VBLinear:accGradParameters -- VBLinear.lua:112-118.
 *   parent (:113) [recalled]: gradWeight += scale * g^T x ; gradBias += scale * g^T 1.
 *   grad = torch.mm(g^T, x) (:114, the same GEMM again); gradSum += grad (.) e (:115).
 * `scale` is NOT applied to gradSum -- that is the reference's behaviour.
 * With e == NULL this is plain nn.Linear:accGradParameters (the final layer, mlp.lua:29). */
VBO_API void vbo_acc_grad_parameters(const float* x, const float* g, const float* e, float scale,
                                     float* gradWeight, float* gradBias, float* gradSum,
                                     int64_t N, int64_t I, int64_t O) {
    float* grad = (float*)malloc((size_t)(O * I) * sizeof(float));
    gemm_tn(g, x, grad, O, I, N, 1.f, 0.f);
    for (int64_t k = 0; k < O * I; ++k) gradWeight[k] = fmaf(scale, grad[k], gradWeight[k]);
    if (gradBias)
        for (int64_t o = 0; o < O; ++o) {
            float s = 0.f;
            for (int64_t n = 0; n < N; ++n) s += g[n * O + o];
            gradBias[o] = fmaf(scale, s, gradBias[o]);
        }
    if (e && gradSum)
        for (int64_t k = 0; k < O * I; ++k) gradSum[k] += grad[k] * e[k];
    free(grad);
}

/* VBLinear:compute_mugrads -- VBLinear.lua:90-93.
 *   lcg = (means - mu_hat) / (B * var_hat) (:91); returns gradWeight:div(S) IN PLACE (:92), lcg. */
VBO_API void vbo_compute_mugrads(const float* means, double var_hat, float B, float S,
                                 float* gradWeight /* in/out */, float* lcg, int64_t W) {
    const float den = (float)((double)B * var_hat);
    for (int64_t i = 0; i < W; ++i) {
        lcg[i] = (means[i] - 0.0f) / den;
        gradWeight[i] = gradWeight[i] / S;
    }
}

/* VBLinear:compute_vargrads -- VBLinear.lua:95-98.
 *   lcg = (-vars^-1 + 1/var_hat) / (2B) (:96), then lcg (.) vars (:97);
 *   likelihood term gradSum:div(2S):cmul(stdv) IN PLACE (:97). */
VBO_API void vbo_compute_vargrads(const float* vars, const float* stdv, double var_hat, float B, float S,
                                  float* gradSum /* in/out */, float* lcg, int64_t W) {
    const float inv_vh = (float)(1.0 / var_hat);
    for (int64_t i = 0; i < W; ++i) {
        const float a = -(1.0f / vars[i]) + inv_vh;
        lcg[i] = (a / (2.0f * B)) * vars[i];
        gradSum[i] = (gradSum[i] / (2.0f * S)) * stdv[i];
    }
}

/* VBLinear:calc_lc -- VBLinear.lua:99-103 (+ the :sum() of mlp.lua:112).
 *   LCfirst  = -log(sqrt(vars)) + log(sqrt(var_hat))            (:100)
 *   LCsecond = (mu_sqe + (vars - var_hat)) / (2 var_hat)        (:101)
 *   return (LCfirst + LCsecond) * (1/B)                         (:102)
 * lc_elem may be NULL; the sum is accumulated in double [recalled: TH accreal]. */
VBO_API double vbo_calc_lc(const float* vars, const float* mu_sqe, double var_hat, float B,
                           float* lc_elem, int64_t W) {
    const float lvh = (float)log(sqrt(var_hat));
    const float vh = (float)var_hat;
    const float invB = 1.0f / B;
    double s = 0.0;
    for (int64_t i = 0; i < W; ++i) {
        const float first = -logf(sqrtf(vars[i])) + lvh;
        const float second = (mu_sqe[i] + (vars[i] - vh)) / (2.0f * vh);
        const float lc = (first + second) * invB;
        if (lc_elem) lc_elem[i] = lc;
        s += (double)lc;
    }
    return s;
}

/* ------------------------------------------------------------------ VBLinear, local reparameterisation
 * (the north_star's form of the same layer; SURVEY.md section 8a "What the build computes").
 *   m = x mu^T + b ; v = (x.x)(sigma^2)^T ; y = m + sqrt(v) . z ,  z ~ N(0,1) per activation
 *   r = z / (2 sqrt(v))  (0 where v == 0), saved for backward.                               */
VBO_API void vbo_lrt_forward(const float* x, const float* means, const float* lvars, const float* bias,
                             const float* zeta, float* y, float* r, float* v_out,
                             int64_t N, int64_t I, int64_t O) {
    float* x2 = (float*)malloc((size_t)(N * I) * sizeof(float));
    float* var = (float*)malloc((size_t)(O * I) * sizeof(float));
    float* v = (float*)malloc((size_t)(N * O) * sizeof(float));
    for (int64_t k = 0; k < N * I; ++k) x2[k] = x[k] * x[k];
    for (int64_t k = 0; k < O * I; ++k) var[k] = expf(lvars[k]);
    gemm_nt(x, means, y, N, O, I, 1.f, 0.f);
    gemm_nt(x2, var, v, N, O, I, 1.f, 0.f);
    for (int64_t n = 0; n < N; ++n)
        for (int64_t o = 0; o < O; ++o) {
            const int64_t k = n * O + o;
            const float sd = sqrtf(v[k]);
            const float mb = y[k] + (bias ? bias[o] : 0.f);
            y[k] = fmaf(sd, zeta[k], mb);
            if (r) r[k] = (v[k] > 0.f) ? zeta[k] / (2.0f * sd) : 0.f;
            if (v_out) v_out[k] = v[k];
        }
    free(x2); free(var); free(v);
}

/* LRT backward for one layer, given g = dL/dy and the saved r:
 *   gv = g . r                                   (dL/dv)
 *   gradWeight += scale * g^T x                  (identical to VBLinear.lua:113)
 *   gradBias   += scale * g^T 1
 *   gradSum    += 2 * (gv^T (x.x)) . stdv        (so that VBLinear.lua:97's
 *                  gradSum/(2S) . stdv equals (dL/dsigma^2) . sigma^2 = dL/dlvars; like the
 *                  reference's gradSum it is not multiplied by `scale`)
 *   gx = g mu + 2 x . (gv sigma^2)               (may be NULL: first layer)                 */
VBO_API void vbo_lrt_backward(const float* x, const float* g, const float* r,
                              const float* means, const float* lvars, float scale,
                              float* gradWeight, float* gradBias, float* gradSum, float* gx,
                              int64_t N, int64_t I, int64_t O) {
    float* x2 = (float*)malloc((size_t)(N * I) * sizeof(float));
    float* var = (float*)malloc((size_t)(O * I) * sizeof(float));
    float* gv = (float*)malloc((size_t)(N * O) * sizeof(float));
    float* t = (float*)malloc((size_t)(O * I) * sizeof(float));
    for (int64_t k = 0; k < N * I; ++k) x2[k] = x[k] * x[k];
    for (int64_t k = 0; k < O * I; ++k) var[k] = expf(lvars[k]);
    for (int64_t k = 0; k < N * O; ++k) gv[k] = g[k] * r[k];
    gemm_tn(g, x, t, O, I, N, 1.f, 0.f);
    for (int64_t k = 0; k < O * I; ++k) gradWeight[k] = fmaf(scale, t[k], gradWeight[k]);
    if (gradBias)
        for (int64_t o = 0; o < O; ++o) {
            float s = 0.f;
            for (int64_t n = 0; n < N; ++n) s += g[n * O + o];
            gradBias[o] = fmaf(scale, s, gradBias[o]);
        }
    gemm_tn(gv, x2, t, O, I, N, 1.f, 0.f);
    for (int64_t k = 0; k < O * I; ++k) {
        const float sd = sqrtf(var[k]);
        gradSum[k] = fmaf(2.0f * t[k], sd, gradSum[k]);
    }
    if (gx) {
        float* a = (float*)malloc((size_t)(N * I) * sizeof(float));
        gemm_nn(g, means, gx, N, I, O);
        gemm_nn(gv, var, a, N, I, O);
        for (int64_t k = 0; k < N * I; ++k) gx[k] = fmaf(2.0f * x[k], a[k], gx[k]);
        free(a);
    }
    free(x2); free(var); free(gv); free(t);
}

/* ------------------------------------------------------------------ glue modules on the measured path
 * (mlp.lua:12-32: Reshape, ReLU, final Linear, LogSoftMax, ClassNLLCriterion) [all recalled]. */

VBO_API void vbo_relu_forward(const float* x, float* y, int64_t n) {
    for (int64_t i = 0; i < n; ++i) y[i] = x[i] > 0.f ? x[i] : 0.f;
}
/* nn.ReLU:updateGradInput: gradInput = gradOutput where input > 0 */
VBO_API void vbo_relu_backward(const float* x, const float* g, float* gx, int64_t n) {
    for (int64_t i = 0; i < n; ++i) gx[i] = x[i] > 0.f ? g[i] : 0.f;
}
/* nn.LogSoftMax: out = x - max - log(sum exp(x - max)), per row; sum in double [recalled]. */
VBO_API void vbo_logsoftmax_forward(const float* x, float* y, int64_t N, int64_t C) {
    for (int64_t n = 0; n < N; ++n) {
        float mx = x[n * C];
        for (int64_t c = 1; c < C; ++c) mx = x[n * C + c] > mx ? x[n * C + c] : mx;
        double s = 0.0;
        for (int64_t c = 0; c < C; ++c) s += exp((double)(x[n * C + c] - mx));
        const float lse = mx + (float)log(s);
        for (int64_t c = 0; c < C; ++c) y[n * C + c] = x[n * C + c] - lse;
    }
}
/* nn.LogSoftMax:updateGradInput: gx = g - exp(out) * sum_c g. */
VBO_API void vbo_logsoftmax_backward(const float* out, const float* g, float* gx, int64_t N, int64_t C) {
    for (int64_t n = 0; n < N; ++n) {
        double s = 0.0;
        for (int64_t c = 0; c < C; ++c) s += (double)g[n * C + c];
        for (int64_t c = 0; c < C; ++c)
            gx[n * C + c] = g[n * C + c] - expf(out[n * C + c]) * (float)s;
    }
}
/* nn.ClassNLLCriterion, sizeAverage = true: loss = -(1/N) sum_n out[n][t_n];
 * gradInput[n][t_n] = -1/N. Targets are 0-based here (the Lua side is 1-based, data.lua:16). */
VBO_API double vbo_nll_forward(const float* out, const int32_t* target, int64_t N, int64_t C) {
    double s = 0.0;
    for (int64_t n = 0; n < N; ++n) s -= (double)out[n * C + target[n]];
    return s / (double)N;
}
VBO_API void vbo_nll_backward(const int32_t* target, float* g, int64_t N, int64_t C) {
    memset(g, 0, (size_t)(N * C) * sizeof(float));
    for (int64_t n = 0; n < N; ++n) g[n * C + target[n]] = -1.0f / (float)N;
}
/* nn.MSECriterion, sizeAverage = true [recalled, torch/nn; not used by the reference itself: it is the criterion of
 * BASELINE.json configs[4], "synthetic 4096-dim regression", a build-side configuration]:
 * loss = (1 / (N D)) sum (y - t)^2;  gradInput = 2 (y - t) / (N D). */
VBO_API double vbo_mse_forward(const float* y, const float* target, int64_t N, int64_t D) {
    double s = 0.0;
    for (int64_t i = 0; i < N * D; ++i) { const double d = (double)y[i] - (double)target[i]; s += d * d; }
    return s / (double)(N * D);
}
VBO_API void vbo_mse_backward(const float* y, const float* target, float* g, int64_t N, int64_t D) {
    const float k = 2.0f / (float)(N * D);
    for (int64_t i = 0; i < N * D; ++i) g[i] = k * (y[i] - target[i]);
}
/* utils.get_accuracy -- utils.lua:11-27: percentage of rows whose arg-max equals the target
 * (first maximum wins, as Tensor:max does [recalled]). */
VBO_API double vbo_get_accuracy(const float* out, const int32_t* target, int64_t N, int64_t C) {
    int64_t correct = 0;
    for (int64_t n = 0; n < N; ++n) {
        int64_t best = 0;
        for (int64_t c = 1; c < C; ++c) if (out[n * C + c] > out[n * C + best]) best = c;
        if (best == target[n]) ++correct;
    }
    return 100.0 * (double)correct / (double)N;
}

/* ------------------------------------------------------------------ the update after the hot path
 * (VBLinear.lua:124-166, mlp.lua:117-142). optim.adam / optim.sgd are un-vendored and unpinned; the
 * reference's author even ran a patched optim (third return value, VBLinear.lua:135-144). Restated here is
 * the published torch/optim adam [recalled]:
 *   state.t += 1; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g.g; denom = sqrt(v) + eps;
 *   stepSize = lr sqrt(1 - b2^t) / (1 - b1^t); x -= stepSize m / denom.
 * `update_out` (optional) receives stepSize m / denom, the quantity whose norm VBLinear.lua:139,144 logs. */
VBO_API void vbo_adam_step(float* x, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                           float eps, int64_t t, float* update_out) {
    const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
    const float step = (float)((double)lr * sqrt(bc2) / bc1);
    for (int64_t i = 0; i < n; ++i) {
        m[i] = b1 * m[i] + (1.0f - b1) * g[i];
        v[i] = b2 * v[i] + (1.0f - b2) * g[i] * g[i];
        const float up = step * m[i] / (sqrtf(v[i]) + eps);
        x[i] -= up;
        if (update_out) update_out[i] = up;
    }
}
/* optim.sgd with only learningRate set (config.lua:51-54) [recalled]: x -= lr g. */
VBO_API void vbo_sgd_step(float* x, const float* g, int64_t n, float lr) {
    for (int64_t i = 0; i < n; ++i) x[i] = fmaf(-lr, g[i], x[i]);
}
