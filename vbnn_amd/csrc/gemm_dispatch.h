// gemm_dispatch.h -- kernel selection shared by the entry points of the three VBLinear GEMM families. The families are compiled as
// separate translation units (gemm_fwd.hip, gemm_dx.hip, gemm_dw.hip: each instantiates its own functor's kernels; one file took
// five minutes to compile), so the selection state below is C++17 `inline` -- one instance in the library, set by vbnn_debug_set.
#pragma once
#include "common.h"
#include "epilogues.h"
#include "gemm_v1.h"
#include "gemm_v2.h"
#include "gemm_v3.h"

static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

// 0 = choose by shape, 1 = always the general kernel (gemm_v1.h), 2 = the pipelined bf16 kernel
// (gemm_v2.h) whenever the operands allow it, 3 = its 256 x 256 two-pass variant (gemm_v3.h) whenever they do.
// Test / A-B hook: vbnn_debug_set(VBNN_DEBUG_GEMM_KERNEL, ..).
inline int g_force_kernel = 0;
inline int g_kmajor = 1;              // K-major operands when the shape allows (vbnn_debug_set key 6)

// would a GEMM of this shape run on gemm_v3 in its K-major form right now? (shape and debug keys only; the functor's
// fast-path conditions are checked at launch)
static inline bool kmajor_selected(int64_t M, int64_t N, int64_t K) {
    if (!g_kmajor || K % V2_BK != 0) return false;
    if (K * (M + 64) >= (1ll << 30) || K * (N + 64) >= (1ll << 30)) return false;      // 32-bit byte offsets of the buffer-form DMA
    if (g_force_kernel == 3) return M % V3_BM == 0 && N % V3_BN == 0;
    return g_force_kernel == 0 && g_v2_tile == 0 && gemm_v3_shape_ok(M, N, K);
}
// accGradParameters only: the pair-split launch of gemm_v2 also has a K-major form (outputs too few for gemm_v3)
static inline bool kmajor_dw_v2_selected(int64_t M, int64_t N, int64_t K) {
    if (K * (M + 64) >= (1ll << 30) || K * (N + 64) >= (1ll << 30)) return false;
    return g_kmajor == 1 && K % V2_BK == 0 && (g_force_kernel == 0 || g_force_kernel == 2) && g_v2_tile != 128 && g_v2_tile != 64 &&
           gemm_v2_eligible<bf16_t>(M, N, K, 64, 64) && gemm_v2_psplit_by_shape(M, N, K);
}
// 0: transposed copies needed; 1: K-major operands as the other GEMMs hold them; 2: K-major, and x / x.x must be allocated
// with their row pitch padded to whole 256-column tiles (zero fill): the split launch of gemm_v3.h
static inline bool kmajor_dw_v3_split_selected(int64_t M, int64_t N, int64_t K) {
    return g_kmajor == 1 && g_force_kernel == 0 && g_v2_tile == 0 && gemm_v3_split_shape_ok(M, N, K) &&
           K * gemm_v3_split_lda(M) < (1ll << 30) && K * (N + 64) < (1ll << 30);
}

// does the forward of an I -> O layer on N rows (M = O, N = N, K = I, K-contiguous operands) run on gemm_v3 right now -- the kernel
// that can carry the classifier head's logits (EpiFwd::head_slots)? Shape and debug keys only, as kmajor_selected; the functor's
// fast-path conditions are checked at launch, where a forward that was given head_slots and does not take that kernel fails loudly.
static inline bool head_slots_selected(int64_t N, int64_t I, int64_t O, int64_t C) {
    if (C < 1 || C > 16 || O % V3_BM || N % V3_BN) return false;
    const int64_t ld = (I + V2_BK - 1) / V2_BK * V2_BK;
    if (O * ld >= (1ll << 30) || N * ld >= (1ll << 30)) return false;
    if (g_force_kernel == 3) return true;
    return g_force_kernel == 0 && g_v2_tile == 0 && gemm_v3_shape_ok(O, N, I);
}

// the K-major launch (A and / or B stored [K][rows]); false = this shape / configuration does not take it
template <typename T, bool DUAL, bool AK, bool BK, class Epi>
static bool try_kmajor(vbnn_ctx* ctx, const void* A, const void* A2, int64_t lda, const void* B, const void* B2, int64_t ldb,
                       int64_t M, int64_t N, int64_t K, const Epi& epi, int* status) {
    if constexpr (sizeof(T) != 2) {
        return false;
    } else {
        if (!A || !B || (DUAL && (!A2 || !B2)) || epi.has_draw_dev() || !kmajor_selected(M, N, K) ||
            !gemm_v3_possible(M, N, K, lda, ldb, AK, BK, epi))
            return false;
        *status = launch_gemm_v3<T, DUAL, AK, BK, Epi>(ctx, (const T*)A, (const T*)A2, lda, (const T*)B, (const T*)B2, ldb, (int)M,
                                                       (int)N, (int)K, epi);
        return true;
    }
}

template <typename T, bool DUAL, class Epi>
static int launch_gemm(vbnn_ctx* ctx, const void* A, const void* A2, int64_t lda, const void* B, const void* B2,
                       int64_t ldb, int64_t M, int64_t N, int64_t K, const Epi& epi, const V1Form& form = V1Form()) {
    if (form.ta || form.tb || form.sq) {
        // fp32 operand forms of the general kernel (gemm_v1.h): K-major sides, the squared partner formed in registers
        if (!A || !B || (DUAL && ((form.sq != 2 && !A2) || (form.sq != 1 && !B2)))) {
            vbnn_set_error("operand missing for the fp32 K-major / squared form");
            return VBNN_ERR_INVALID;
        }
        return launch_gemm_v1<T, DUAL, Epi>(ctx->stream, (const T*)A, (const T*)A2, lda, (const T*)B, (const T*)B2, ldb,
                                            (int)M, (int)N, (int)K, epi, form);
    }
    if (!A || !B || (DUAL && (!A2 || !B2))) {
        vbnn_set_error("the K-contiguous operands are required for this shape (vbnn_kmajor_supported says no)");
        return VBNN_ERR_INVALID;
    }
    // the device-resident draw counter is read by the general kernel only (the launch-bound configurations that get
    // captured into a graph run on it; the pipelined kernels take the counter as a launch argument)
    const bool v2_ok = gemm_v2_possible<T>(lda, ldb) && M * lda < (1ll << 30) && N * ldb < (1ll << 30) && !epi.has_draw_dev();
    if (v2_ok && sizeof(T) == 2 &&
        ((g_force_kernel == 3 && gemm_v3_possible(M, N, K, lda, ldb, false, false, epi)) ||
         (g_force_kernel == 0 && g_v2_tile == 0 && gemm_v3_eligible(M, N, K, lda, ldb, false, false, epi))))
        return launch_gemm_v3<T, DUAL, false, false, Epi>(ctx, (const T*)A, (const T*)A2, lda, (const T*)B, (const T*)B2, ldb, (int)M,
                                                          (int)N, (int)K, epi);
    if constexpr (Epi::HEAD) {
        if (epi.head_slots) {
            vbnn_set_error("head_slots given, but this forward does not run on the two-pass 256 x 256 kernel (functor fast path or operands)");
            return VBNN_ERR_INVALID;
        }
    }
    if (v2_ok && g_force_kernel != 1 && (g_force_kernel == 2 || gemm_v2_eligible<T>(M, N, K, lda, ldb)))
        return launch_gemm_v2<T, DUAL, Epi>(ctx, (const T*)A, (const T*)A2, lda, (const T*)B, (const T*)B2, ldb,
                                            (int)M, (int)N, (int)K, epi);
    return launch_gemm_v1<T, DUAL, Epi>(ctx->stream, (const T*)A, (const T*)A2, lda, (const T*)B, (const T*)B2, ldb,
                                        (int)M, (int)N, (int)K, epi);
}

template <typename T>
static EpiDx<T> make_dx_epi(const vbnn_dx_args* a) {
    EpiDx<T> e;
    e.dual = a->gv != nullptr;
    e.x = (const T*)a->x; e.ld_x = a->ld_x;
    e.gx = a->gx; e.ld_gx = a->ld_gx; e.gx_vec = a->gx && aligned16(a->gx) && (a->ld_gx % 4 == 0);
    e.relu_mask = a->relu_mask;
    e.r_prev = a->r_prev_packed ? nullptr : (const float*)a->r_prev;
    e.r_prev_t = a->r_prev_packed ? (const T*)a->r_prev : nullptr;
    e.ld_r_prev = a->ld_r_prev; e.r_vec = a->r_prev && aligned16(a->r_prev) && (a->ld_r_prev % 4 == 0);
    e.g_prev = (T*)a->g_prev; e.gv_prev = (T*)a->gv_prev; e.ld_gp = a->ld_gp;
    e.gT_prev = (T*)a->gT_prev; e.gvT_prev = (T*)a->gvT_prev; e.ld_gpT = a->ld_gpT;
    e.I = (int)a->I; e.N = (int)a->N;
    return e;
}

template <typename T>
static EpiDw make_dw_epi(const vbnn_dw_args* a) {
    EpiDw e;
    e.lrt = (a->x2T != nullptr) || (a->x2 != nullptr) || (a->gvT != nullptr) || (a->gv != nullptr);
    e.scale = a->scale; e.accumulate = a->accumulate;
    e.gradWeight = a->gradWeight; e.gradSum = a->gradSum;
    e.vec = (a->I % 4 == 0) && (!a->lvars || aligned16(a->lvars)) && (!a->means || aligned16(a->means)) &&
            (!a->gradWeight || aligned16(a->gradWeight)) && (!a->gradSum || aligned16(a->gradSum)) &&
            (!a->grad_mu || aligned16(a->grad_mu)) && (!a->grad_lv || aligned16(a->grad_lv));
    e.seed = a->seed; e.layer = a->layer; e.draw = a->draw; e.draw_dev = a->draw_dev;
    e.lvars = a->lvars;
    e.grad_mu = a->grad_mu; e.grad_lv = a->grad_lv;
    e.means = a->means; e.stats = a->stats; e.B = a->B; e.S = a->S; e.kl_scale = a->kl_scale;
    e.gradBias = a->gradBias;
    const bool shadows = sizeof(T) == 2 && a->mu_s && a->var_s && (a->grad_mu || a->grad_lv) && a->ld_w >= a->I &&
                         a->ld_w < (1ll << 31) / (a->O > 0 ? a->O : 1);
    e.mu_s = shadows ? (const bf16_t*)a->mu_s : nullptr; e.var_s = shadows ? (const bf16_t*)a->var_s : nullptr; e.ld_w = shadows ? (int)a->ld_w : 0;
    e.I = (int)a->I; e.O = (int)a->O;
    return e;
}

static inline int check_dx_args(vbnn_ctx* ctx, const vbnn_dx_args* a) {
    VBNN_REQUIRE(ctx && a, "null ctx/args");
    VBNN_REQUIRE((a->wT || a->w) && a->g, "wT (or the K-major w) and g are required");
    VBNN_REQUIRE(!a->wT || ((a->w2T == nullptr) == (a->gv == nullptr)), "w2T and gv go together (LRT pair)");
    VBNN_REQUIRE(!a->w || ((a->w2 == nullptr) == (a->gv == nullptr)), "w2 and gv go together (LRT pair)");
    VBNN_REQUIRE(!a->gv || a->x, "LRT gradInput needs the layer input x");
    VBNN_REQUIRE(!a->relu_mask || a->x, "relu_mask needs the layer input x");
    VBNN_REQUIRE(a->N > 0 && a->I > 0 && a->O > 0, "N, I, O must be positive");
    VBNN_REQUIRE(a->N < (1ll << 31) && a->I < (1ll << 31) && a->O < (1ll << 31), "dimension too large");
    VBNN_REQUIRE(!a->x || a->ld_x >= a->I, "ld_x");
    VBNN_REQUIRE(!a->gx || a->ld_gx >= a->I, "ld_gx");
    VBNN_REQUIRE(!a->gv_prev || a->g_prev, "gv_prev needs g_prev");
    VBNN_REQUIRE(!a->g_prev || (a->ld_gp >= a->I && a->ld_gp % 4 == 0), "ld_gp");
    VBNN_REQUIRE(!a->gT_prev || a->ld_gpT >= a->N, "ld_gpT");
    return VBNN_OK;
}

static inline int check_dw_args(vbnn_ctx* ctx, int dtype, const vbnn_dw_args* a) {
    VBNN_REQUIRE(ctx && a, "null ctx/args");
    VBNN_REQUIRE((a->xT && a->gT) || (a->x && a->g) || (a->x && a->gT), "xT and gT (or the K-major x and g, or x with gT) are required");
    const bool f32_km = dtype == VBNN_F32 && !a->xT && a->x && a->g;       // fp32 K-major form: x.x may be left to the kernel
    VBNN_REQUIRE(f32_km || (a->x2T != nullptr || a->x2 != nullptr) == (a->gvT != nullptr || a->gv != nullptr), "x.x and gv operands go together (LRT pair)");
    VBNN_REQUIRE(!a->xT || ((a->x2T == nullptr) == (a->gvT == nullptr)), "x2T and gvT go together (LRT pair)");
    VBNN_REQUIRE(f32_km || !a->g || ((a->x2 == nullptr) == (a->gv == nullptr)), "x2 and gv go together (LRT pair)");
    VBNN_REQUIRE(!a->x2 || a->gv || a->gvT, "x2 without gv");
    VBNN_REQUIRE(a->N > 0 && a->I > 0 && a->O > 0, "N, I, O must be positive");
    VBNN_REQUIRE(a->N < (1ll << 31) && a->I < (1ll << 31) && a->O < (1ll << 31), "dimension too large");
    VBNN_REQUIRE(!((a->x2T || a->x2 || a->gv || a->gvT) && (a->gradSum || a->grad_lv)) || a->lvars, "LRT gradSum/grad_lv need lvars");
    VBNN_REQUIRE(!(a->grad_mu || a->grad_lv) || (a->means && a->lvars && a->stats && a->B > 0 && a->S > 0),
                 "fused total gradients need means, lvars, stats, B, S");
    VBNN_REQUIRE(a->part >= 0 && a->part <= 2 && (a->part == 0 || a->x2T || a->x2), "part: 0, or 1 / 2 of an LRT pair (with x.x given)");
    return VBNN_OK;
}

