-- vbnn_ffi.lua -- LuaJIT FFI binding of libvbnn_hip.so (include/vbnn_hip.h).
--
-- GENERATED from the header by lua/gen_ffi.py: do not edit the cdef by hand. tests/test_abi.py checks
-- that every function of the header appears here with the same parameter count.
-- No LuaJIT / Torch7 exists in the build image, so this file has never been executed there; it is the
-- reference-side binding a maintainer adds (INTEGRATION.md).
local ffi = require 'ffi'

ffi.cdef[[
enum { VBNN_OK = 0, VBNN_ERR_INVALID = 1, VBNN_ERR_HIP = 2, VBNN_ERR_NOMEM = 3, VBNN_ERR_UNSUPPORTED = 4 };
enum { VBNN_F32 = 0, VBNN_BF16 = 1 };
enum { VBNN_PACK_COPY = 0, VBNN_PACK_EXP = 1, VBNN_PACK_SQUARE = 2, VBNN_PACK_MUL = 3,
       VBNN_PACK_RELU = 4, VBNN_PACK_RELU_SQUARE = 5 };
typedef struct vbnn_ctx vbnn_ctx;
int vbnn_abi_version(void);
const char* vbnn_last_error(void);
int vbnn_debug_set(int key, int value);
int vbnn_kmajor_supported(int64_t M, int64_t N, int64_t K);
int vbnn_kmajor_supported_dw(int64_t I, int64_t O, int64_t N, int bias_row);
int vbnn_ctx_kmajor_supported(vbnn_ctx* ctx, int64_t M, int64_t N, int64_t K);
int vbnn_ctx_kmajor_supported_dw(vbnn_ctx* ctx, int64_t I, int64_t O, int64_t N, int bias_row);
int vbnn_ctx_create(int device, void* hip_stream, vbnn_ctx** out);
int vbnn_ctx_create_cu_budget(int device, int n_cus, vbnn_ctx** out);
int vbnn_ctx_stream(vbnn_ctx* ctx, void** hip_stream_out, int* n_cus_out);
int vbnn_ctx_destroy(vbnn_ctx* ctx);
int vbnn_ctx_set_stream(vbnn_ctx* ctx, void* hip_stream);
int vbnn_sync(vbnn_ctx* ctx);
int vbnn_buf_alloc(vbnn_ctx* ctx, size_t bytes, void** dptr);
int vbnn_buf_free(vbnn_ctx* ctx, void* dptr);
int vbnn_buf_zero(vbnn_ctx* ctx, void* dptr, size_t bytes);
int vbnn_buf_upload(vbnn_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int vbnn_buf_download(vbnn_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int vbnn_fill_normal(vbnn_ctx* ctx, float* out, int64_t rows, int64_t cols, int64_t ld,
                     uint64_t seed, uint32_t stream, uint32_t layer, uint32_t draw, int64_t row0,
                     float scale);
int vbnn_fill_normal_hw(vbnn_ctx* ctx, float* out, int64_t rows, int64_t cols, int64_t ld,
                        uint64_t seed, uint32_t stream, uint32_t layer, uint32_t draw, int64_t row0,
                        float scale);
int vbnn_box_muller_forms(vbnn_ctx* ctx, const uint32_t* x0, const uint32_t* x1, float* z_exact, float* z_hw, int64_t n);
int vbnn_compute_prior(vbnn_ctx* ctx, const float* means, const float* lvars, int64_t W,
                       float* vars, float* stdv, float* mu_sqe, double* stats);
int vbnn_wn_sample(vbnn_ctx* ctx, const float* means, const float* stdv, const float* lvars,
                   float* weight, float* e_out, int64_t O, int64_t I,
                   uint64_t seed, uint32_t layer, uint32_t draw);
int vbnn_pack(vbnn_ctx* ctx, int dtype, int func, const float* src, const float* src2, int64_t ld_src,
              int64_t rows, int64_t cols, void* dst, int64_t ld_dst, void* dstT, int64_t ld_dstT);
typedef struct vbnn_fwd_args {
    const void* w;
    const void* w2;
    const void* x;
    const void* x2;
    int64_t ld_w, ld_x;
    int64_t N, I, O;
    const float* bias;
    uint64_t seed; uint32_t layer; uint32_t draw; int64_t row0;
    float* y;  int64_t ld_y;
    void* r;  int64_t ld_r;
    int r_packed;
    int relu;
    void* h;  void* h2;  int64_t ld_h;
    void* hT; void* h2T; int64_t ld_hT;
    int64_t rows_per_draw;
    const uint32_t* draw_dev;
    const void* head_w3; int64_t head_ld_w; int64_t head_C; float* head_slots;
} vbnn_fwd_args;
int vbnn_forward(vbnn_ctx* ctx, int dtype, const vbnn_fwd_args* a);
int vbnn_forward_head_slots(vbnn_ctx* ctx, int dtype, int64_t N, int64_t I, int64_t O, int64_t C);
typedef struct vbnn_dx_args {
    const void* wT;
    const void* w2T;
    const void* g;
    const void* gv;
    int64_t ld_wT, ld_g;
    int64_t N, I, O;
    const void* x; int64_t ld_x;
    float* gx; int64_t ld_gx;
    int relu_mask;
    const void* r_prev; int64_t ld_r_prev; int r_prev_packed;
    void* g_prev; void* gv_prev; int64_t ld_gp;
    void* gT_prev; void* gvT_prev; int64_t ld_gpT;
    const void* w; const void* w2; int64_t ld_w;
} vbnn_dx_args;
int vbnn_grad_input(vbnn_ctx* ctx, int dtype, const vbnn_dx_args* a);
typedef struct vbnn_dw_args {
    const void* xT;
    const void* x2T;
    const void* gT;
    const void* gvT;
    int64_t ld_n;
    int64_t N, I, O;
    float scale;
    int accumulate;
    float* gradWeight;
    float* gradSum;
    uint64_t seed; uint32_t layer; uint32_t draw;
    const float* lvars;
    float* grad_mu; float* grad_lv;
    const float* means; const double* stats; float B; float S; float kl_scale;
    float* gradBias;
    const void* x; const void* x2; const void* g; const void* gv; int64_t ld_x; int64_t ld_g;
    const void* mu_s; const void* var_s; int64_t ld_w;
    int part;
    const uint32_t* draw_dev;
} vbnn_dw_args;
int vbnn_acc_grad_parameters(vbnn_ctx* ctx, int dtype, const vbnn_dw_args* a);
int vbnn_backward_pair(vbnn_ctx* ctx, int dtype, const vbnn_dx_args* dx, const vbnn_dw_args* dw);
int vbnn_acc_grad_bias(vbnn_ctx* ctx, int dtype, const void* g, int64_t ld_g, int64_t N, int64_t O,
                       float scale, int accumulate, float* gradBias);
int vbnn_prep_layer(vbnn_ctx* ctx, int dtype, const float* means, const float* lvars, int64_t O, int64_t I,
                    void* mu_s, void* var_s, int64_t ld_w, void* muT_s, void* varT_s, int64_t ld_wT,
                    double* stats);
typedef struct vbnn_prep_desc {
    const float* means; const float* lvars; int64_t O, I;
    void* mu_s; void* var_s; int64_t ld_w;
    void* muT_s; void* varT_s; int64_t ld_wT;
    double* stats;
} vbnn_prep_desc;
typedef struct vbnn_pack_desc {
    const float* src; int64_t rows, cols, ld_src;
    void* dst; int64_t ld_dst; void* dstT; int64_t ld_dstT;
} vbnn_pack_desc;
int vbnn_prepare(vbnn_ctx* ctx, int dtype, int n_layers, const vbnn_prep_desc* layers, const vbnn_pack_desc* extra);
int vbnn_compute_mugrads(vbnn_ctx* ctx, const float* means, const double* stats, float B, float S,
                         float* gradWeight, float* lcg, int64_t W);
int vbnn_compute_vargrads(vbnn_ctx* ctx, const float* lvars, const float* vars, const float* stdv,
                          const double* stats, float B, float S, float* gradSum, float* lcg, int64_t W);
int vbnn_calc_lc(vbnn_ctx* ctx, const float* means, const float* lvars, const float* vars, const float* mu_sqe,
                 const double* stats, float B, float* lc_elem, double* lc_sum_dev, int64_t W);
int vbnn_pack_input(vbnn_ctx* ctx, int dtype, const float* src, int64_t ld_src, int64_t N, int64_t I, void* x_s,
                    void* x2_s, int64_t ld_x, void* xT_s, void* x2T_s, int64_t ld_xT, int64_t rows_per_draw);
int vbnn_adam_step(vbnn_ctx* ctx, float* x, const float* grad, const float* grad2, float* m, float* v, int64_t n,
                   float lr, float beta1, float beta2, float eps, float lambda, int64_t t, double* norms_dev);
int vbnn_sgd_step(vbnn_ctx* ctx, float* x, const float* grad, int64_t n, float lr);
typedef struct vbnn_adam_cfg { float lr, beta1, beta2, eps, lambda; int64_t t; } vbnn_adam_cfg;
typedef struct vbnn_update_desc {
    float* means; float* lvars; int64_t O, I;
    void* mu_s; void* var_s; int64_t ld_w;
    void* muT_s; void* varT_s; int64_t ld_wT;
    double* stats;
    const float* grad_mu; const float* grad_lv;
    float* m_mu; float* v_mu; float* m_lv; float* v_lv;
    vbnn_adam_cfg mu, lv;
    float* bias; const float* grad_bias; float lr_bias;
    float B;
    double* log14;
    float kl_add;
} vbnn_update_desc;
int vbnn_update(vbnn_ctx* ctx, int dtype, int n_layers, const vbnn_update_desc* layers, const vbnn_pack_desc* extra);
typedef struct vbnn_comm vbnn_comm;
int vbnn_comm_unique_id(void* id_out );
int vbnn_comm_create(vbnn_ctx* ctx, int rank, int world, const void* id, vbnn_comm** out);
int vbnn_comm_destroy(vbnn_comm* comm);
int vbnn_comm_info(vbnn_comm* comm, int* rank, int* world, int* ranks_in_comm );
int vbnn_allreduce_grads(vbnn_comm* comm, float* buf, int64_t n);
int vbnn_comm_finish(vbnn_comm* comm);
int vbnn_allreduce_grads_bf16(vbnn_comm* comm, void* buf_bf16, int64_t n);
int vbnn_cast_grads(vbnn_ctx* ctx, int to_bf16, const void* src, void* dst, int64_t n);
int vbnn_comm_allgather_u64(vbnn_comm* comm, const uint64_t* mine_dev, uint64_t* all_dev);
int vbnn_comm_reduce_scatter(vbnn_comm* comm, float* buf, int64_t n_per_rank);
int vbnn_comm_all_gather(vbnn_comm* comm, void* buf, int64_t bytes_per_rank);
int vbnn_stats_combine(vbnn_ctx* ctx, int n_layers, int world, const double* parts, double* const* stats);
int vbnn_transpose_packed(vbnn_ctx* ctx, int dtype, const void* src, int64_t ld_src, int64_t rows, int64_t cols, void* dst, int64_t ld_dst);
typedef struct vbnn_p2p vbnn_p2p;
int vbnn_p2p_create(vbnn_ctx* ctx, int rank, int world, size_t arena_floats, vbnn_p2p** out, void** arena_out, void* handle_out);
int vbnn_p2p_connect(vbnn_p2p* p, const void* all_handles);
int vbnn_p2p_allreduce(vbnn_p2p* p, size_t offset_floats, int64_t n);
int vbnn_p2p_finish(vbnn_p2p* p);
int vbnn_p2p_reduce_scatter(vbnn_p2p* p, size_t offset_floats, int64_t n_per_rank);
int vbnn_p2p_all_gather(vbnn_p2p* p, size_t offset_floats, int64_t n_per_rank);
int vbnn_p2p_status(vbnn_p2p* p, int* rank, int* world, unsigned* gave_up);
int vbnn_p2p_set_timeout(vbnn_p2p* p, double seconds);
int vbnn_p2p_clear_status(vbnn_p2p* p);
int vbnn_p2p_destroy(vbnn_p2p* p);
int vbnn_p2p_set_grid(vbnn_p2p* p, int rs_blocks, int ag_blocks_per_peer);
int vbnn_p2p_standin(vbnn_p2p* p, int sim_world, double inbound_GBps);
typedef struct vbnn_box_info {
    double mfma_clock_ghz, mfma_tflops, mfma_ms;
    double hbm_TBps, hbm_ms;
    int64_t hbm_bytes;
    int cus, reserved;
} vbnn_box_info;
int vbnn_box_calibrate(vbnn_ctx* ctx, vbnn_box_info* out);
int vbnn_sample(vbnn_ctx* ctx, uint32_t* draw_dev, uint32_t by);
typedef struct vbnn_graph vbnn_graph;
int vbnn_capture_begin(vbnn_ctx* ctx);
int vbnn_capture_end(vbnn_ctx* ctx, vbnn_graph** out);
int vbnn_graph_launch(vbnn_graph* g);
int vbnn_graph_info(vbnn_graph* g, int* kernel_nodes, int* nodes);
int vbnn_graph_destroy(vbnn_graph* g);
int vbnn_relu_forward(vbnn_ctx* ctx, const float* x, float* y, int64_t n);
int vbnn_relu_backward(vbnn_ctx* ctx, const float* x, const float* g, float* gx, int64_t n);
int vbnn_logsoftmax_nll(vbnn_ctx* ctx, const float* logits, int64_t ld, const int32_t* target,
                        int64_t N, int64_t C, float inv_n, float* out, float* g_logits,
                        double* loss_sum_dev, int32_t* correct_dev);
int vbnn_mse_forward(vbnn_ctx* ctx, const float* y, int64_t ld_y, const float* target, int64_t ld_t, int64_t N, int64_t D,
                     float inv_nd, float* g, int64_t ld_g, int accumulate, double* loss_sum_dev);
int vbnn_mse_backward(vbnn_ctx* ctx, const float* y, int64_t ld_y, const float* target, int64_t ld_t, int64_t N, int64_t D,
                      float inv_nd, float* g, int64_t ld_g);
int vbnn_head_forward(vbnn_ctx* ctx, int dtype, const void* h, int64_t ld_h, const void* w3, int64_t ld_w,
                      const float* bias, const int32_t* target, int64_t N, int64_t H, int64_t C, float inv_n,
                      float* logits, float* out, float* g_logits, int accumulate, double* loss_sum_dev,
                      int32_t* correct_dev, int64_t rows_per_draw);
int vbnn_head_forward_slots(vbnn_ctx* ctx, const float* slots, int64_t n_slots, const float* bias, const int32_t* target,
                            int64_t N, int64_t C, float inv_n, float* logits, float* out, float* g_logits, int accumulate,
                            double* loss_sum_dev, int32_t* correct_dev, int64_t rows_per_draw);
int vbnn_head_backward(vbnn_ctx* ctx, int dtype, const void* h, int64_t ld_h, const void* w3, int64_t ld_w,
                       const float* g_logits, int64_t N, int64_t H, int64_t C, int accumulate, float* gradWeight,
                       float* gradBias, float* gradBias_prev, int relu_mask, const void* r_prev, int64_t ld_r_prev,
                       int r_prev_packed, void* g_prev, void* gv_prev, int64_t ld_gp, void* gT_prev, void* gvT_prev,
                       int64_t ld_gpT);
typedef struct vbnn_head_args {
    const void* h; int64_t ld_h;
    const void* w3; int64_t ld_w;
    const float* bias;
    const int32_t* target;
    int64_t N, H, C;
    int64_t rows_per_draw;
    float inv_n;
    int32_t accumulate;
    float* logits; float* out; float* g_logits;
    double* loss_sum_dev; int32_t* correct_dev;
    float* gradWeight; float* gradBias; float* gradBias_prev;
    int32_t relu_mask; int32_t r_prev_packed;
    const void* r_prev; int64_t ld_r_prev;
    void* g_prev; void* gv_prev; int64_t ld_gp;
    void* gT_prev; void* gvT_prev; int64_t ld_gpT;
    const float* logit_slots; int64_t n_slots;
} vbnn_head_args;
int vbnn_head_forward_backward(vbnn_ctx* ctx, int dtype, const vbnn_head_args* a);
int vbnn_nll_forward(vbnn_ctx* ctx, const float* out, int64_t ld, const int32_t* target, int64_t N, int64_t C,
                     float inv_n, double* loss_sum_dev, int32_t* correct_dev);
int vbnn_nll_backward(vbnn_ctx* ctx, const int32_t* target, int64_t N, int64_t C, float inv_n, float* g);
int vbnn_logsoftmax_backward(vbnn_ctx* ctx, const float* out, const float* g, float* gx, int64_t N, int64_t C);
]]

local M = {}
M.C = ffi.load(os.getenv('VBNN_HIP_LIB') or 'vbnn_hip')
M.ffi = ffi

-- status -> Lua error (the error convention of SURVEY.md 8b: no C++ exception crosses the ABI)
function M.check(status)
    if status ~= 0 then
        error('libvbnn_hip status ' .. tostring(status) .. ': ' .. ffi.string(M.C.vbnn_last_error()), 2)
    end
end

-- one context per process (device from VBNN_DEVICE, default stream)
local ctx_box = ffi.new('vbnn_ctx*[1]')
M.check(M.C.vbnn_ctx_create(tonumber(os.getenv('VBNN_DEVICE') or '0'), nil, ctx_box))
M.ctx = ffi.gc(ctx_box[0], M.C.vbnn_ctx_destroy)

-- device buffer with finaliser
function M.alloc(bytes)
    local p = ffi.new('void*[1]')
    M.check(M.C.vbnn_buf_alloc(M.ctx, bytes, p))
    return ffi.gc(p[0], function(d) M.C.vbnn_buf_free(M.ctx, d) end)
end

function M.pad_ld(k) return math.floor((k + 63) / 64) * 64 end

return M
