// gemm.hip -- selection hooks and shape queries of the GEMM entry points, and the one-launch backward pair of the fp32 path.
// (vbnn_forward: gemm_fwd.hip; vbnn_grad_input: gemm_dx.hip; vbnn_acc_grad_parameters: gemm_dw.hip.)
#include "gemm_dispatch.h"

extern "C" int vbnn_debug_set(int key, int value) {
    if (key == VBNN_DEBUG_GEMM_KERNEL && value >= 0 && value <= 3) { g_force_kernel = value; return VBNN_OK; }
    if (key == VBNN_DEBUG_V2_SCHEDULE && (value == -1 || value == 0 || value == 2 || value == 4)) { g_v2_sched = value; return VBNN_OK; }
    if (key == VBNN_DEBUG_V2_TILE && (value == 0 || value == 64 || value == 128 || value == 256)) { g_v2_tile = value; return VBNN_OK; }
    if (key == VBNN_DEBUG_V3_MIN_K && value >= 64) { g_v3_min_k = value; return VBNN_OK; }
    if (key == VBNN_DEBUG_V2_PSPLIT && value >= -1 && value <= 1) { g_v2_psplit = value; return VBNN_OK; }
    if (key == VBNN_DEBUG_KMAJOR && value >= 0 && value <= 2) { g_kmajor = value; return VBNN_OK; }   // 2: gemm_v3 only
    if (key == VBNN_DEBUG_V3_SPLIT && value >= -1 && value <= 1) { g_v3_split = value; return VBNN_OK; }
    if (key == VBNN_DEBUG_V0 && (value == 0 || value == 1)) { g_v0 = value; return VBNN_OK; }
    if (key == VBNN_DEBUG_HEAD_BACKWARD && value >= -1 && value <= 1) { g_head_stream = value; return VBNN_OK; }
    vbnn_set_error("vbnn_debug_set: unknown key %d / value %d", key, value);
    return VBNN_ERR_INVALID;
}

extern "C" int vbnn_kmajor_supported(int64_t M, int64_t N, int64_t K) { return kmajor_selected(M, N, K) ? 1 : 0; }
extern "C" int vbnn_ctx_kmajor_supported(vbnn_ctx* ctx, int64_t M, int64_t N, int64_t K) {
    vbnn_cu_scope plan(ctx);                                 // the answer for THIS context's launches (its CU budget, if any)
    return kmajor_selected(M, N, K) ? 1 : 0;
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
