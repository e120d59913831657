// gemm_dx.hip -- vbnn_grad_input (updateGradInput): the C-ABI entry point of the gradInput GEMM family.
#include "gemm_dispatch.h"

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

