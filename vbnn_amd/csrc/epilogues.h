// epilogues.h -- fused epilogues of the three VBLinear GEMM families.
//
// Every GEMM kernel in this library computes C[m][n] = sum_k A[m][k] * Bt[n][k] (and, for
// the local-reparameterisation pair, a second accumulator from A2/Bt2) on 16x16 MFMA tiles,
// whose accumulator layout is   n = lane & 15,  m = 4 * (lane >> 4) + reg   (reg = 0..3).
// The "weights / feature" side is always operand A, so a lane's four registers are four
// CONSECUTIVE feature indices of one minibatch row (FWD: 4 output units of row n; DX: 4
// input units of row n; DW: 4 input units i of output unit o). That is the contiguous
// direction of every primary output tensor and exactly one Philox block (4 normals).
//
// An epilogue is called once per lane per 16x16 tile:  epi(m, n, acc1, acc2).
#pragma once
#include "common.h"

// ---- FWD: M = output units o, N = minibatch rows n ---------------------------------------------
// WN/MAP: y = acc1 + b                      (inherited nn.Linear:updateOutput, VBLinear.lua:7)
// LRT   : y = acc1 + b + sqrt(acc2) * z,  r = z / (2 sqrt(acc2))
template <typename T>
struct EpiFwd {
    const float* bias;
    int noise;
    uint64_t seed; uint32_t layer, draw; int64_t row0;
    float* y; int64_t ld_y; int y_vec;
    float* r; int64_t ld_r; int r_vec;
    int relu;
    T* h; T* h2; int64_t ld_h;
    T* hT; T* h2T; int64_t ld_hT;
    int O, N;

    __device__ __forceinline__ void operator()(int m, int n, f32x4 a1, f32x4 a2) const {
        const int valid = min(4, O - m);
        if (valid <= 0 || n >= N) return;
        float yv[4], rv[4];
        vbnn_f32x4 z;
        if (noise) z = vbnn_normal4(seed, VBNN_STREAM_ZETA, layer, draw, (uint32_t)(row0 + n), (uint32_t)(m >> 2));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float b = (bias && j < valid) ? bias[m + j] : 0.f;
            const float mb = a1[j] + b;
            if (noise) {
                const float sd = sqrtf(a2[j]);
                yv[j] = fmaf(sd, z.v[j], mb);
                rv[j] = (a2[j] > 0.f) ? z.v[j] / (2.0f * sd) : 0.f;
            } else {
                yv[j] = mb;
                rv[j] = 0.f;
            }
        }
        if (y) store4<float>(y + (int64_t)n * ld_y + m, yv[0], yv[1], yv[2], yv[3], valid, y_vec);
        if (r) store4<float>(r + (int64_t)n * ld_r + m, rv[0], rv[1], rv[2], rv[3], valid, r_vec);
        if (h || hT) {
            float hv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) hv[j] = relu ? fmaxf(yv[j], 0.f) : yv[j];
            if (h) {
                store4<T>(h + (int64_t)n * ld_h + m, hv[0], hv[1], hv[2], hv[3], valid, true);
                if (h2) {
                    // square the value the consumer will actually read (the rounded one)
                    float q[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float hr = Elt<T>::from(Elt<T>::to(hv[j])); q[j] = hr * hr; }
                    store4<T>(h2 + (int64_t)n * ld_h + m, q[0], q[1], q[2], q[3], valid, true);
                }
            }
            if (hT) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < valid) {
                        hT[(int64_t)(m + j) * ld_hT + n] = Elt<T>::to(hv[j]);
                        if (h2T) { const float hr = Elt<T>::from(Elt<T>::to(hv[j])); h2T[(int64_t)(m + j) * ld_hT + n] = Elt<T>::to(hr * hr); }
                    }
            }
        }
    }
};

// ---- DX: M = input units i, N = minibatch rows n ------------------------------------------------
// WN : gx = acc1 = g w                         (inherited nn.Linear:updateGradInput)
// LRT: gx = acc1 + 2 x . acc2 = g mu + 2 x . (gv sigma^2)
// optional hand-off to the previous VB layer through the ReLU between them (mlp.lua:19,27).
template <typename T>
struct EpiDx {
    int dual;
    const T* x; int64_t ld_x;
    float* gx; int64_t ld_gx; int gx_vec;
    int relu_mask;
    const float* r_prev; int64_t ld_r_prev; int r_vec;
    T* g_prev; T* gv_prev; int64_t ld_gp;
    T* gT_prev; T* gvT_prev; int64_t ld_gpT;
    int I, N;

    __device__ __forceinline__ void operator()(int m, int n, f32x4 a1, f32x4 a2) const {
        const int valid = min(4, I - m);
        if (valid <= 0 || n >= N) return;
        float xv[4] = {0.f, 0.f, 0.f, 0.f};
        if (x) load4<T>(x + (int64_t)n * ld_x + m, xv, valid, (ld_x & 3) == 0);
        float gv4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) gv4[j] = dual ? fmaf(2.0f * xv[j], a2[j], a1[j]) : a1[j];
        if (gx) store4<float>(gx + (int64_t)n * ld_gx + m, gv4[0], gv4[1], gv4[2], gv4[3], valid, gx_vec);
        if (g_prev || gT_prev) {
            float gp[4], gvp[4], rp[4] = {0.f, 0.f, 0.f, 0.f};
            if (r_prev) load4<float>(r_prev + (int64_t)n * ld_r_prev + m, rp, valid, r_vec);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gp[j] = (relu_mask && !(xv[j] > 0.f)) ? 0.f : gv4[j];
                gvp[j] = gp[j] * rp[j];
            }
            if (g_prev) store4<T>(g_prev + (int64_t)n * ld_gp + m, gp[0], gp[1], gp[2], gp[3], valid, true);
            if (gv_prev) store4<T>(gv_prev + (int64_t)n * ld_gp + m, gvp[0], gvp[1], gvp[2], gvp[3], valid, true);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < valid) {
                    if (gT_prev) gT_prev[(int64_t)(m + j) * ld_gpT + n] = Elt<T>::to(gp[j]);
                    if (gvT_prev) gvT_prev[(int64_t)(m + j) * ld_gpT + n] = Elt<T>::to(gvp[j]);
                }
        }
    }
};

// ---- DW: M = input units i, N = output units o --------------------------------------------------
// acc1 = (g^T x)[o][i], acc2 = (gv^T x.x)[o][i]          (VBLinear.lua:112-118, one GEMM not two)
struct EpiDw {
    int lrt;                 // 1: LRT (acc2 valid), 0: WN (e regenerated)
    float scale; int accumulate;
    float* gradWeight; float* gradSum; int vec;
    uint64_t seed; uint32_t layer, draw;
    const float* lvars;
    float* grad_mu; float* grad_lv;
    const float* means; const double* stats; float B, S, kl_scale;
    int I, O;

    __device__ __forceinline__ void operator()(int m, int n, f32x4 a1, f32x4 a2) const {
        const int valid = min(4, I - m);
        if (valid <= 0 || n >= O) return;
        const int64_t base = (int64_t)n * I + m;
        float e[4] = {0.f, 0.f, 0.f, 0.f}, sd[4] = {0.f, 0.f, 0.f, 0.f}, var[4] = {0.f, 0.f, 0.f, 0.f};
        if (!lrt && (gradSum || grad_lv)) {
            const vbnn_f32x4 z = vbnn_normal4(seed, VBNN_STREAM_EPS, layer, draw, (uint32_t)n, (uint32_t)(m >> 2));
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = z.v[j];
        }
        if (lvars && (gradSum || grad_lv)) {
            float lv4[4];
            load4<float>(lvars + base, lv4, valid, vec);
#pragma unroll
            for (int j = 0; j < 4; ++j) if (j < valid) { var[j] = expf(lv4[j]); sd[j] = sqrtf(var[j]); }
        }
        if (gradWeight) {
            float o4[4], old4[4] = {0.f, 0.f, 0.f, 0.f};
            if (accumulate) load4<float>(gradWeight + base, old4, valid, vec);
#pragma unroll
            for (int j = 0; j < 4; ++j) o4[j] = fmaf(scale, a1[j], old4[j]);
            store4<float>(gradWeight + base, o4[0], o4[1], o4[2], o4[3], valid, vec);
        }
        if (gradSum) {
            float o4[4], old4[4] = {0.f, 0.f, 0.f, 0.f};
            if (accumulate) load4<float>(gradSum + base, old4, valid, vec);
#pragma unroll
            for (int j = 0; j < 4; ++j) o4[j] = lrt ? fmaf(2.0f * a2[j], sd[j], old4[j]) : old4[j] + a1[j] * e[j];
            store4<float>(gradSum + base, o4[0], o4[1], o4[2], o4[3], valid, vec);
        }
        if (grad_mu || grad_lv) {
            // VBLinear.lua:90-98 folded in: likelihood/S (+ kl_scale * KL gradient on the first draw)
            const float var_hat = (float)stats[2];
            const float invS = 1.0f / S;
            float gm[4], gl[4], mu4[4] = {0.f, 0.f, 0.f, 0.f}, om4[4] = {0.f, 0.f, 0.f, 0.f}, ol4[4] = {0.f, 0.f, 0.f, 0.f};
            if (means && !accumulate) load4<float>(means + base, mu4, valid, vec);
            if (accumulate && grad_mu) load4<float>(grad_mu + base, om4, valid, vec);
            if (accumulate && grad_lv) load4<float>(grad_lv + base, ol4, valid, vec);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float mu = mu4[j];
                float lm = scale * a1[j] * invS;
                float ll = lrt ? a2[j] * var[j] * invS : a1[j] * e[j] * sd[j] * (0.5f * invS);
                if (accumulate) {
                    lm += om4[j];
                    ll += ol4[j];
                } else {
                    lm += kl_scale * mu / (B * var_hat);
                    ll += kl_scale * (var[j] / var_hat - 1.0f) / (2.0f * B);
                }
                gm[j] = lm; gl[j] = ll;
            }
            if (grad_mu) store4<float>(grad_mu + base, gm[0], gm[1], gm[2], gm[3], valid, vec);
            if (grad_lv) store4<float>(grad_lv + base, gl[0], gl[1], gl[2], gl[3], valid, vec);
        }
    }
};
