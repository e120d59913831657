// gemm_fwd.hip -- vbnn_forward (updateOutput): the C-ABI entry point of the forward GEMM family.
#include "gemm_dispatch.h"

template <typename T>
static int forward_t(vbnn_ctx* ctx, const vbnn_fwd_args* a) {
    EpiFwd<T> e;
    e.bias = a->bias;
    e.noise = a->w2 != nullptr ? 1 : 0;
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
    if (a->head_slots) {
        // the classifier head's logits from the forward's own tiles (vbnn_fwd_args.head_slots): the two-pass 256 x 256 kernel only
        if constexpr (sizeof(T) == 2) {
            if (!head_slots_selected(a->N, a->I, a->O, a->head_C) || a->rows_per_draw != 0 || a->draw_dev || !a->head_w3 || a->hT ||
                a->head_ld_w < a->O || a->head_ld_w % 4 != 0 || !aligned16(a->head_w3) || !aligned16(a->head_slots)) {
                vbnn_set_error("head_slots: this forward does not take the fused form (ask vbnn_forward_head_slots; no stacked draws, "
                               "no device draw counter, no transposed output, packed 16-byte-aligned head_w3)");
                return VBNN_ERR_INVALID;
            }
            e.head_w3 = (const T*)a->head_w3; e.head_ld_w = a->head_ld_w; e.head_C = (int)a->head_C; e.head_slots = a->head_slots;
        } else {
            vbnn_set_error("head_slots: bf16 only");
            return VBNN_ERR_INVALID;
        }
    }
    if (a->w2) {
        V1Form f;
        if (!a->x2) f.sq = 1;                                    // fp32: x.x is formed while staging x (vbnn_fwd_args.x2 == NULL)
        return launch_gemm<T, true>(ctx, a->w, a->w2, a->ld_w, a->x, a->x2, a->ld_x, a->O, a->N, a->I, e, f);
    }
    return launch_gemm<T, false>(ctx, a->w, nullptr, a->ld_w, a->x, nullptr, a->ld_x, a->O, a->N, a->I, e);
}

// slots the forward of an I -> O layer on N rows would write with vbnn_fwd_args.head_slots (0: that launch does not carry the
// head's logits -- use vbnn_head_forward on h)
extern "C" int vbnn_forward_head_slots(vbnn_ctx* ctx, int dtype, int64_t N, int64_t I, int64_t O, int64_t C) {
    vbnn_cu_scope plan(ctx);
    if (dtype != VBNN_BF16 || !head_slots_selected(N, I, O, C)) return 0;
    return (int)(2 * (O / V3_BM));
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

