// gemm.hip -- C-ABI entry points of the three VBLinear GEMM families.
#include "common.h"
#include "epilogues.h"
#include "gemm_v1.h"
#include "gemm_v2.h"
#include "gemm_v3.h"

static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

// 0 = choose by shape, 1 = always the general kernel (gemm_v1.h), 2 = the pipelined bf16 kernel
// (gemm_v2.h) whenever the operands allow it, 3 = its 256 x 256 two-pass variant (gemm_v3.h) whenever they do.
// Test / A-B hook: vbnn_debug_set(VBNN_DEBUG_GEMM_KERNEL, ..).
static int g_force_kernel = 0;
static int g_kmajor = 1;              // K-major operands when the shape allows (vbnn_debug_set key 6)
static int g_fake_noise = 0;          // A/B only (key 7): the forward fold skips Philox + Box-Muller (wrong results, timing only)

extern "C" int vbnn_debug_set(int key, int value) {
    if (key == VBNN_DEBUG_GEMM_KERNEL && value >= 0 && value <= 3) { g_force_kernel = value; return VBNN_OK; }
    if (key == VBNN_DEBUG_V2_SCHEDULE && (value == -1 || value == 0 || value == 2 || value == 4)) { g_v2_sched = value; return VBNN_OK; }
    if (key == VBNN_DEBUG_V2_TILE && (value == 0 || value == 64 || value == 128 || value == 256)) { g_v2_tile = value; return VBNN_OK; }
    if (key == VBNN_DEBUG_V2_SPLITK && value >= -1 && value <= 1) { g_v2_split = value; return VBNN_OK; }
    if (key == VBNN_DEBUG_V3_MIN_K && value >= 64) { g_v3_min_k = value; return VBNN_OK; }
    if (key == VBNN_DEBUG_V2_PSPLIT && value >= -1 && value <= 1) { g_v2_psplit = value; return VBNN_OK; }
    if (key == VBNN_DEBUG_KMAJOR && value >= 0 && value <= 2) { g_kmajor = value; return VBNN_OK; }   // 2: gemm_v3 only
    if (key == VBNN_DEBUG_V3_SPLIT && value >= -1 && value <= 1) { g_v3_split = value; return VBNN_OK; }
    if (key == VBNN_DEBUG_V3_SPLIT && value >= 2 && value <= 4) { g_v3_hm = value - 2; return VBNN_OK; }   // its half-height form: off / by shape / whenever the split launch is taken
    if (key == VBNN_DEBUG_FAKE_NOISE && (value == 0 || value == 1)) { g_fake_noise = value; return VBNN_OK; }
    if (key == VBNN_DEBUG_V0 && (value == 0 || value == 1)) { g_v0 = value; return VBNN_OK; }
    vbnn_set_error("vbnn_debug_set: unknown key %d / value %d", key, value);
    return VBNN_ERR_INVALID;
}

// would a GEMM of this shape run on gemm_v3 in its K-major form right now? (shape and debug keys only; the functor's
// fast-path conditions are checked at launch)
static bool kmajor_selected(int64_t M, int64_t N, int64_t K) {
    if (!g_kmajor || K % V2_BK != 0) return false;
    if (K * (M + 64) >= (1ll << 30) || K * (N + 64) >= (1ll << 30)) return false;      // 32-bit byte offsets of the buffer-form DMA
    if (g_force_kernel == 3) return M % V3_BM == 0 && N % V3_BN == 0;
    return g_force_kernel == 0 && g_v2_tile == 0 && gemm_v3_shape_ok(M, N, K);
}
extern "C" int vbnn_kmajor_supported(int64_t M, int64_t N, int64_t K) { return kmajor_selected(M, N, K) ? 1 : 0; }
extern "C" int vbnn_ctx_kmajor_supported(vbnn_ctx* ctx, int64_t M, int64_t N, int64_t K) {
    vbnn_cu_scope plan(ctx);                                 // the answer for THIS context's launches (its CU budget, if any)
    return kmajor_selected(M, N, K) ? 1 : 0;
}
// accGradParameters only: the pair-split launch of gemm_v2 also has a K-major form (outputs too few for gemm_v3)
static bool kmajor_dw_v2_selected(int64_t M, int64_t N, int64_t K) {
    if (K * (M + 64) >= (1ll << 30) || K * (N + 64) >= (1ll << 30)) return false;
    return g_kmajor == 1 && K % V2_BK == 0 && (g_force_kernel == 0 || g_force_kernel == 2) && g_v2_tile != 128 && g_v2_tile != 64 &&
           g_v2_split != 1 && gemm_v2_eligible<bf16_t>(M, N, K, 64, 64) && gemm_v2_psplit_by_shape(M, N, K);
}
// 0: transposed copies needed; 1: K-major operands as the other GEMMs hold them; 2: K-major, and x / x.x must be allocated
// with their row pitch padded to whole 256-column tiles (zero fill): the split launch of gemm_v3.h
static bool kmajor_dw_v3_split_selected(int64_t M, int64_t N, int64_t K) {
    return g_kmajor == 1 && g_force_kernel == 0 && g_v2_tile == 0 && gemm_v3_split_shape_ok(M, N, K) &&
           K * gemm_v3_split_lda(M) < (1ll << 30) && K * (N + 64) < (1ll << 30);
}
extern "C" int vbnn_kmajor_supported_dw(int64_t I, int64_t O, int64_t N, int bias_row) {
    if (!bias_row && kmajor_selected(I, O, N)) return 1;
    if (I % 4 == 0 && kmajor_dw_v3_split_selected(I + (bias_row ? 1 : 0), O, N)) return 2;
    return kmajor_dw_v2_selected(I + (bias_row ? 1 : 0), O, N) ? 1 : 0;
}
extern "C" int vbnn_ctx_kmajor_supported_dw(vbnn_ctx* ctx, int64_t I, int64_t O, int64_t N, int bias_row) {
    vbnn_cu_scope plan(ctx);
    return vbnn_kmajor_supported_dw(I, O, N, bias_row);
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
    if (v2_ok && g_force_kernel != 1 && (g_force_kernel == 2 || gemm_v2_eligible<T>(M, N, K, lda, ldb)))
        return launch_gemm_v2<T, DUAL, Epi>(ctx, (const T*)A, (const T*)A2, lda, (const T*)B, (const T*)B2, ldb,
                                            (int)M, (int)N, (int)K, epi);
    return launch_gemm_v1<T, DUAL, Epi>(ctx->stream, (const T*)A, (const T*)A2, lda, (const T*)B, (const T*)B2, ldb,
                                        (int)M, (int)N, (int)K, epi);
}

template <typename T>
static int forward_t(vbnn_ctx* ctx, const vbnn_fwd_args* a) {
    EpiFwd<T> e;
    e.bias = a->bias;
    e.noise = a->w2 != nullptr ? (g_fake_noise ? 2 : 1) : 0;
    e.seed = a->seed; e.layer = a->layer; e.draw = a->draw; e.row0 = a->row0; e.draw_dev = a->draw_dev;
    e.rpd = (int)a->rows_per_draw;
    e.y = a->y; e.ld_y = a->ld_y; e.y_vec = a->y && aligned16(a->y) && (a->ld_y % 4 == 0);
    e.r = a->r_packed ? nullptr : (float*)a->r;
    e.r_t = a->r_packed ? (T*)a->r : nullptr;
    e.ld_r = a->ld_r; e.r_vec = a->r && aligned16(a->r) && (a->ld_r % 4 == 0);
    e.relu = a->relu;
    e.h = (T*)a->h; e.h2 = (T*)a->h2; e.ld_h = a->ld_h;
    e.hT = (T*)a->hT; e.h2T = (T*)a->h2T; e.ld_hT = a->ld_hT;
    e.O = (int)a->O; e.N = (int)a->N;
    if (a->w2) {
        V1Form f;
        if (!a->x2) f.sq = 1;                                    // fp32: x.x is formed while staging x (vbnn_fwd_args.x2 == NULL)
        return launch_gemm<T, true>(ctx, a->w, a->w2, a->ld_w, a->x, a->x2, a->ld_x, a->O, a->N, a->I, e, f);
    }
    return launch_gemm<T, false>(ctx, a->w, nullptr, a->ld_w, a->x, nullptr, a->ld_x, a->O, a->N, a->I, e);
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
static int grad_input_t(vbnn_ctx* ctx, const vbnn_dx_args* a) {
    const EpiDx<T> e = make_dx_epi<T>(a);
    const bool dual = a->gv != nullptr;
    int st = VBNN_OK;                                        // K-major weights first (no transposed shadows needed)
    if (dual ? try_kmajor<T, true, true, false>(ctx, a->w, a->w2, a->ld_w, a->g, a->gv, a->ld_g, a->I, a->N, a->O, e, &st)
             : try_kmajor<T, false, true, false>(ctx, a->w, nullptr, a->ld_w, a->g, nullptr, a->ld_g, a->I, a->N, a->O, e, &st))
        return st;
    if constexpr (sizeof(T) == 4) {
        if (!a->wT && a->w) {                                    // fp32: the weights K-major as the forward holds them (gemm_v1.h, TA)
            V1Form f;
            f.ta = true;
            if (dual) return launch_gemm<T, true>(ctx, a->w, a->w2, a->ld_w, a->g, a->gv, a->ld_g, a->I, a->N, a->O, e, f);
            return launch_gemm<T, false>(ctx, a->w, nullptr, a->ld_w, a->g, nullptr, a->ld_g, a->I, a->N, a->O, e, f);
        }
    }
    if (dual) return launch_gemm<T, true>(ctx, a->wT, a->w2T, a->ld_wT, a->g, a->gv, a->ld_g, a->I, a->N, a->O, e);
    return launch_gemm<T, false>(ctx, a->wT, nullptr, a->ld_wT, a->g, nullptr, a->ld_g, a->I, a->N, a->O, e);
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

template <typename T>
static int acc_grad_t(vbnn_ctx* ctx, const vbnn_dw_args* a) {
    EpiDw e = make_dw_epi<T>(a);
    const int64_t M = a->I + (a->gradBias ? 1 : 0);          // the ones row of xT (K-major: column I of x) rides along as one more output row
    int st = VBNN_OK;
    if (a->part == 1 || a->part == 2) {
        // ONE GEMM of the pair and the outputs that depend on it (vbnn_dw_args.part): single-accumulator launches with the
        // functor told which accumulator it holds (EpiDw::part, as the pair-split launches do inside one grid)
        e.part = a->part;
        const bool second = a->part == 2;
        const void* xk = second ? a->x2 : a->x;   const void* gk = second ? a->gv : a->g;
        const void* xt = second ? a->x2T : a->xT; const void* gt = second ? a->gvT : a->gT;
        if (second) e.gradBias = nullptr;
        const int64_t Mp = second ? a->I : M;
        if (!(a->gradBias && !second) &&
            try_kmajor<T, false, true, true>(ctx, xk, nullptr, a->ld_x, gk, nullptr, a->ld_g, a->I, a->O, a->N, e, &st))
            return st;
        return launch_gemm<T, false>(ctx, xt, nullptr, a->ld_n, gt, nullptr, a->ld_n, Mp, a->O, a->N, e);
    }
    const bool dual = e.lrt != 0;
    if constexpr (sizeof(T) == 4) {
        if (!a->xT && a->x && a->g) {
            // fp32: x and g (gv) K-major as the forward / gradInput GEMMs hold them, x.x formed in registers when not given,
            // the bias gradient from a synthetic row of ones (gemm_v1.h: TA, TB, SQ = 2, ones_row)
            V1Form f;
            f.ta = f.tb = true;
            f.sq = (dual && !a->x2) ? 2 : 0;
            f.ones_row = a->gradBias ? (int)a->I : -1;
            if (dual) return launch_gemm<T, true>(ctx, a->x, a->x2, a->ld_x, a->g, a->gv, a->ld_g, M, a->O, a->N, e, f);
            return launch_gemm<T, false>(ctx, a->x, nullptr, a->ld_x, a->g, nullptr, a->ld_g, M, a->O, a->N, e, f);
        }
    }
    // MIXED operands: x, x.x K-major as the forward holds them, g, gv TRANSPOSED (gT, gvT: O x ld_n, K-contiguous) as
    // their producer's epilogue can write them -- the transpose read of the 256-column B tile is the slower of the two
    // (lab: 104 vs 93 us per pass at 4096^3), a transposed g costs its producer one more pair of stores
    if (dual && !a->gradBias && a->x && a->x2 && a->gT && a->gvT && g_kmajor &&
        try_kmajor<T, true, true, false>(ctx, a->x, a->x2, a->ld_x, a->gT, a->gvT, a->ld_n, a->I, a->O, a->N, e, &st))
        return st;
    // K-major x, g (no transposed copies needed)
    if (!a->gradBias &&
        (dual ? try_kmajor<T, true, true, true>(ctx, a->x, a->x2, a->ld_x, a->g, a->gv, a->ld_g, a->I, a->O, a->N, e, &st)
              : try_kmajor<T, false, true, true>(ctx, a->x, nullptr, a->ld_x, a->g, nullptr, a->ld_g, a->I, a->O, a->N, e, &st)))
        return st;
    if constexpr (sizeof(T) == 2) {                          // ... or, for outputs with few tiles, pair split + split-K on gemm_v3
        if (dual && !a->draw_dev && a->x && a->x2 && a->g && a->gv && a->I % 4 == 0 && kmajor_dw_v3_split_selected(M, a->O, a->N)) {
            st = launch_gemm_v3_split<T, EpiDw>(ctx, (const T*)a->x, (const T*)a->x2, a->ld_x, (const T*)a->g, (const T*)a->gv, a->ld_g,
                                                (int)M, (int)a->O, (int)a->N, e);
            if (st != VBNN_ERR_UNSUPPORTED) return st;
        }
    }
    if constexpr (sizeof(T) == 2) {                          // ... or the pair-split form of the pipelined kernel
        if (dual && !a->draw_dev && a->x && a->x2 && a->g && a->gv && kmajor_dw_v2_selected(M, a->O, a->N)) {
            st = launch_gemm_v2<T, true, EpiDw>(ctx, (const T*)a->x, (const T*)a->x2, a->ld_x, (const T*)a->g, (const T*)a->gv, a->ld_g,
                                                (int)M, (int)a->O, (int)a->N, e, true);
            if (st != VBNN_ERR_UNSUPPORTED) return st;
        }
    }
    if (dual) return launch_gemm<T, true>(ctx, a->xT, a->x2T, a->ld_n, a->gT, a->gvT, a->ld_n, M, a->O, a->N, e);
    return launch_gemm<T, false>(ctx, a->xT, nullptr, a->ld_n, a->gT, nullptr, a->ld_n, M, a->O, a->N, e);
}

extern "C" int vbnn_forward(vbnn_ctx* ctx, int dtype, const vbnn_fwd_args* a) {
    VBNN_API_BEGIN
    vbnn_cu_scope plan(ctx);                                 // shape heuristics: this context's compute units
    VBNN_REQUIRE(ctx && a, "null ctx/args");
    VBNN_REQUIRE(a->w && a->x, "w and x are required");
    VBNN_REQUIRE(!a->x2 || a->w2, "x2 needs w2 (LRT pair)");
    VBNN_REQUIRE(!a->w2 || a->x2 || dtype == VBNN_F32, "w2 and x2 go together (LRT pair); only the fp32 kernel squares x itself");
    VBNN_REQUIRE(a->N > 0 && a->I > 0 && a->O > 0, "N, I, O must be positive");
    VBNN_REQUIRE(a->N < (1ll << 31) && a->I < (1ll << 31) && a->O < (1ll << 31), "dimension too large");
    VBNN_REQUIRE(a->rows_per_draw >= 0 && a->rows_per_draw < (1ll << 31), "rows_per_draw");
    VBNN_REQUIRE(!a->h2 || a->h, "h2 needs h");
    VBNN_REQUIRE(!a->h2T || a->hT, "h2T needs hT");
    VBNN_REQUIRE(!a->h || (a->ld_h >= a->O && a->ld_h % 4 == 0), "ld_h");
    VBNN_REQUIRE(!a->hT || a->ld_hT >= a->N, "ld_hT");
    VBNN_REQUIRE(!a->y || a->ld_y >= a->O, "ld_y");
    VBNN_REQUIRE(!a->r || a->ld_r >= a->O, "ld_r");
    if (dtype == VBNN_F32) return forward_t<float>(ctx, a);
    if (dtype == VBNN_BF16) return forward_t<bf16_t>(ctx, a);
    vbnn_set_error("unsupported dtype %d", dtype);
    return VBNN_ERR_UNSUPPORTED;
    VBNN_API_END
}

static int check_dx_args(vbnn_ctx* ctx, const vbnn_dx_args* a) {
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

extern "C" int vbnn_grad_input(vbnn_ctx* ctx, int dtype, const vbnn_dx_args* a) {
    VBNN_API_BEGIN
    vbnn_cu_scope plan(ctx);                                 // shape heuristics: this context's compute units
    const int chk = check_dx_args(ctx, a);
    if (chk != VBNN_OK) return chk;
    if (dtype == VBNN_F32) return grad_input_t<float>(ctx, a);
    if (dtype == VBNN_BF16) return grad_input_t<bf16_t>(ctx, a);
    vbnn_set_error("unsupported dtype %d", dtype);
    return VBNN_ERR_UNSUPPORTED;
    VBNN_API_END
}

static int check_dw_args(vbnn_ctx* ctx, int dtype, const vbnn_dw_args* a) {
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

extern "C" int vbnn_acc_grad_parameters(vbnn_ctx* ctx, int dtype, const vbnn_dw_args* a) {
    VBNN_API_BEGIN
    vbnn_cu_scope plan(ctx);                                 // shape heuristics: this context's compute units
    const int chk = check_dw_args(ctx, dtype, a);
    if (chk != VBNN_OK) return chk;
    if (dtype == VBNN_F32) return acc_grad_t<float>(ctx, a);
    if (dtype == VBNN_BF16) return acc_grad_t<bf16_t>(ctx, a);
    vbnn_set_error("unsupported dtype %d", dtype);
    return VBNN_ERR_UNSUPPORTED;
    VBNN_API_END
}

// updateGradInput and accGradParameters of ONE layer as one call (include/vbnn_hip.h): one launch for the fp32 K-major forms at
// the launch-bound geometry, otherwise exactly the two calls in the order given
extern "C" int vbnn_backward_pair(vbnn_ctx* ctx, int dtype, const vbnn_dx_args* dx, const vbnn_dw_args* dw) {
    VBNN_API_BEGIN
    vbnn_cu_scope plan(ctx);                                 // shape heuristics: this context's compute units
    int chk = check_dx_args(ctx, dx);
    if (chk != VBNN_OK) return chk;
    chk = check_dw_args(ctx, dtype, dw);
    if (chk != VBNN_OK) return chk;
    VBNN_REQUIRE(dx->N == dw->N && dx->I == dw->I && dx->O == dw->O, "the two argument blocks describe one layer");
    if (dtype == VBNN_F32 && !dx->wT && dx->w && !dw->xT && dw->x && dw->g && dw->part == 0 && g_force_kernel == 0) {
        const bool dual = dx->gv != nullptr;
        const bool dw_dual = dw->gv != nullptr;
        if (dual == dw_dual && (!dual || (dx->w2 && !dw->x2))) {
            const EpiDx<float> ea = make_dx_epi<float>(dx);
            const EpiDw eb = make_dw_epi<float>(dw);
            const int M2 = (int)dw->I + (dw->gradBias ? 1 : 0);
            int st;
            if (dual)
                st = launch_gemm_v1_pair<true>(ctx->stream, (const float*)dx->w, (const float*)dx->w2, dx->ld_w, (const float*)dx->g,
                                               (const float*)dx->gv, dx->ld_g, (int)dx->I, (int)dx->N, (int)dx->O, ea, (const float*)dw->x,
                                               dw->ld_x, (const float*)dw->g, (const float*)dw->gv, dw->ld_g, M2, (int)dw->O, (int)dw->N,
                                               dw->gradBias ? (int)dw->I : -1, eb);
            else
                st = launch_gemm_v1_pair<false>(ctx->stream, (const float*)dx->w, nullptr, dx->ld_w, (const float*)dx->g, nullptr, dx->ld_g,
                                                (int)dx->I, (int)dx->N, (int)dx->O, ea, (const float*)dw->x, dw->ld_x, (const float*)dw->g,
                                                nullptr, dw->ld_g, M2, (int)dw->O, (int)dw->N, dw->gradBias ? (int)dw->I : -1, eb);
            if (st != VBNN_ERR_UNSUPPORTED) return st;
        }
    }
    int st = vbnn_acc_grad_parameters(ctx, dtype, dw);
    if (st != VBNN_OK) return st;
    return vbnn_grad_input(ctx, dtype, dx);
    VBNN_API_END
}
