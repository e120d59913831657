/*
 * vbnn_philox.h -- the RNG contract of the VBLinear hot path.
 *
 * One header, compiled by gcc (oracle), g++ (host side of the library) and hipcc
 * (device code), so that the CPU oracle and the gfx950 kernels generate
 * BIT-IDENTICAL standard normals for the same (seed, stream, layer, draw, index).
 *
 * What it replaces in the reference: `randomkit.normal(self.e, zeros, ones)`
 * (VBLinear.lua:55) -- a host-side, single-threaded MT19937 + polar Box-Muller
 * fill of an O x I tensor [recalled, randomkit is not vendored in the reference].
 * A sequential generator cannot be reproduced across 256 CUs, so the build
 * replaces the *stream* (not the distribution) by a counter-based one:
 * Philox4x32-10 (Salmon et al., SC'11; the same generator rocRAND/cuRAND ship)
 * + Box-Muller, addressed by element index. The Philox core is pinned by the
 * Random123 known-answer vectors in tests/golden/philox_kat.json.
 *
 * Determinism rules for this file (they are what makes gcc == hipcc bitwise):
 *   - only IEEE-exact primitives: +, -, *, correctly rounded /, sqrtf, fmaf;
 *   - every multiply-add is an explicit fmaf(); no expression mixes * and +,
 *     so -ffp-contract cannot change a result (gcc builds pass
 *     -ffp-contract=off anyway);
 *   - no libm log/sin/cos: vbnn_det_logf / vbnn_det_sincos2pi below.
 *
 * Who evaluates it how: the oracle, the host side and every fp32 kernel evaluate this header as written (bitwise equal
 * normals, tested). The bf16 forward kernels use the SAME Philox words and addressing but run Box-Muller on the GPU's
 * log2 / sqrt / sin / cos instructions (vbnn_amd/csrc/common.h, vbnn_normal4_hw): within 2e-6 absolute of the values
 * defined here (tested on 2^22 normals), a third of the instructions; vbnn_fill_normal_hw returns exactly those values.
 */
#ifndef VBNN_PHILOX_H
#define VBNN_PHILOX_H

#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define VBNN_HD __host__ __device__ __forceinline__
#else
#define VBNN_HD static inline
#endif

/* stream ids: the low byte of counter word 3 (the high 24 bits carry the layer id) */
#define VBNN_STREAM_EPS   1u /* weight noise   e[o][i]   (VBLinear.lua:55)            */
#define VBNN_STREAM_ZETA  2u /* activation noise z[n][o] (local reparameterisation)   */
#define VBNN_STREAM_INIT  3u /* means ~ N(0, sqrt(var_init)) (VBLinear.lua:26-28)     */
#define VBNN_STREAM_DATA  4u /* synthetic inputs for bench/tests                      */
#define VBNN_STREAM_HEINIT 5u /* He init of `weight` (mlp.lua:52-53)                  */

typedef struct { uint32_t v[4]; } vbnn_u32x4;
typedef struct { float v[4]; } vbnn_f32x4;

#define VBNN_PHILOX_M0 0xD2511F53u
#define VBNN_PHILOX_M1 0xCD9E8D57u
#define VBNN_PHILOX_W0 0x9E3779B9u
#define VBNN_PHILOX_W1 0xBB67AE85u

/* Philox4x32-10: 10 rounds, key bumped by the Weyl constants between rounds. */
VBNN_HD vbnn_u32x4 vbnn_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                      uint32_t k0, uint32_t k1) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)VBNN_PHILOX_M0 * (uint64_t)c0;
        const uint64_t p1 = (uint64_t)VBNN_PHILOX_M1 * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
#if defined(__HIP_DEVICE_COMPILE__) && defined(__gfx950__)
        /* one three-input bitwise op (truth table 0x96 = a ^ b ^ c) instead of two v_xor: 20 fewer VALU slots per block */
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
#else
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
#endif
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += VBNN_PHILOX_W0; k1 += VBNN_PHILOX_W1;
    }
    vbnn_u32x4 out;
    out.v[0] = c0; out.v[1] = c1; out.v[2] = c2; out.v[3] = c3;
    return out;
}

VBNN_HD float vbnn_u32_as_f32(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}
VBNN_HD uint32_t vbnn_f32_as_u32(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}

/* natural log for x in [2^-24, 1]; relative error ~2e-8 before the final rounding.
 * x = m * 2^e, m in [sqrt(.5), sqrt(2)); ln m = 2 atanh(s), s = (m-1)/(m+1). */
VBNN_HD float vbnn_det_logf(float x) {
    uint32_t b = vbnn_f32_as_u32(x);
    int e = (int)(b >> 23) - 127;
    uint32_t mb = (b & 0x007FFFFFu) | 0x3F800000u;   /* m in [1,2) */
    const uint32_t big = (mb > 0x3FB504F3u) ? 1u : 0u; /* m > sqrt(2): halve it (exact); branch-free on purpose -- */
    mb -= big << 23;                                   /* on the GPU a lane-dependent branch here costs more than   */
    e += (int)big;                                     /* both of its sides                                         */
    const float m = vbnn_u32_as_f32(mb);
    const float f = m - 1.0f;                         /* exact */
    const float den = 2.0f + f;
    const float s = f / den;                          /* correctly rounded */
    const float z = s * s;
    float p = fmaf(z, 0.111111111f, 0.142857143f);    /* 1/9, 1/7 */
    p = fmaf(z, p, 0.2f);
    p = fmaf(z, p, 0.333333333f);
    p = z * p;
    const float s2 = s + s;
    const float lm = fmaf(s2, p, s2);                 /* 2s(1 + z/3 + z^2/5 + ...) */
    return fmaf((float)e, 0.693147181f, lm);
}

/* (cos, sin) of 2*pi*k/2^24 for a 24-bit k. Quadrant reduction is exact integer work;
 * the polynomial runs on a in [-pi/4, pi/4). Absolute error ~1e-7. */
VBNN_HD void vbnn_det_sincos2pi(uint32_t k24, float* c_out, float* s_out) {
    const uint32_t kk = (k24 + 0x00200000u) & 0x00FFFFFFu;  /* round to nearest quadrant */
    const uint32_t q = kk >> 22;
    const int fr = (int)(kk & 0x003FFFFFu) - 0x00200000;    /* [-2^21, 2^21) */
    const float t = (float)fr * 2.38418579e-7f;             /* * 2^-22, exact */
    const float a = t * 1.57079633f;                        /* pi/2 */
    const float a2 = a * a;
    float sp = fmaf(a2, 2.75573192e-6f, -1.98412698e-4f);   /* 1/9!, -1/7! */
    sp = fmaf(a2, sp, 8.33333333e-3f);                      /* 1/5! */
    sp = fmaf(a2, sp, -1.66666667e-1f);                     /* -1/3! */
    sp = a2 * sp;
    const float sn = fmaf(a, sp, a);
    float cp = fmaf(a2, -2.75573192e-7f, 2.48015873e-5f);   /* -1/10!, 1/8! */
    cp = fmaf(a2, cp, -1.38888889e-3f);                     /* -1/6! */
    cp = fmaf(a2, cp, 4.16666667e-2f);                      /* 1/4! */
    cp = fmaf(a2, cp, -0.5f);
    const float cs = fmaf(a2, cp, 1.0f);
    /* quadrant q: (cos, sin)(a + q pi/2) = (cs, sn), (-sn, cs), (-cs, -sn), (sn, -cs). Branch-free: odd quadrants swap
     * the two polynomial values, then a sign bit is flipped (exactly what unary minus does): cos negative for q = 1, 2,
     * sin negative for q = 2, 3. Bit for bit the four-way if it replaces. */
    const uint32_t swap = q & 1u;
    const float bc = swap ? sn : cs;
    const float bs = swap ? cs : sn;
    const uint32_t sign_c = ((q + 1u) & 2u) << 30;
    const uint32_t sign_s = (q & 2u) << 30;
    *c_out = vbnn_u32_as_f32(vbnn_f32_as_u32(bc) ^ sign_c);
    *s_out = vbnn_u32_as_f32(vbnn_f32_as_u32(bs) ^ sign_s);
}

/* Box-Muller on two 32-bit words: u1 in (0,1] from the top 24 bits (+1), u2 in [0,1). */
VBNN_HD void vbnn_box_muller(uint32_t x0, uint32_t x1, float* z0, float* z1) {
    const float u1 = (float)((x0 >> 8) + 1u) * 5.96046448e-8f;   /* * 2^-24, exact */
    const float l = vbnn_det_logf(u1);                          /* <= 0 */
    const float r = sqrtf(-2.0f * l);
    float c, s;
    vbnn_det_sincos2pi(x1 >> 8, &c, &s);
    *z0 = r * c;
    *z1 = r * s;
}

/*
 * The addressing convention (the whole contract):
 *   counter = ( quad index , row index , draw , (layer << 8) | stream ),  key = seed.
 * One Philox call yields the four normals of elements 4*quad .. 4*quad+3 of `row`:
 *   EPS : row = o (output unit),      quad = i >> 2      -> e[o][4q..4q+3]
 *   ZETA: row = global minibatch row, quad = o >> 2      -> z[n][4q..4q+3]
 * Data-parallel ranks pass their global row offset, so z does not depend on the
 * number of GPUs; every rank uses the same (seed, layer, draw) for EPS.
 */
VBNN_HD vbnn_f32x4 vbnn_normal4(uint64_t seed, uint32_t stream, uint32_t layer, uint32_t draw,
                                uint32_t row, uint32_t quad) {
    const vbnn_u32x4 u = vbnn_philox4x32_10(quad, row, draw, (layer << 8) | stream,
                                            (uint32_t)seed, (uint32_t)(seed >> 32));
    vbnn_f32x4 z;
    vbnn_box_muller(u.v[0], u.v[1], &z.v[0], &z.v[1]);
    vbnn_box_muller(u.v[2], u.v[3], &z.v[2], &z.v[3]);
    return z;
}

#endif /* VBNN_PHILOX_H */
