// gemm_dw.hip -- vbnn_acc_grad_parameters (accGradParameters): the C-ABI entry point of the parameter-gradient GEMM family.
#include "gemm_dispatch.h"

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

