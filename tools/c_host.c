/*
 * c_host.c -- a host of the VBLinear hot path written against include/vbnn_hip.h ONLY: no Python, no PyTorch, no HIP
 * headers. It is the executable stand-in for the LuaJIT-FFI host (lua/FusedMLP.lua), which cannot run in the build
 * image (no LuaJIT / Torch7 there): the same protocol as the reference's mlp.lua --
 *     mlp:resetGradients() (mlp.lua:62-67)   mlp:sample() (:69-74)   mlp:run(inputs, targets) (:76-84)
 *     mlp:update(opt) (:117-142 + VBLinear.lua:124-166)              the S-draw loop of main.lua:28-40
 * -- issued call for call as lua/FusedMLP.lua issues it (tests/test_abi.py lints the two against each other), with
 *   - a context on the NULL stream (vbnn_ctx_create(dev, NULL)),
 *   - ALL device memory from vbnn_buf_alloc / vbnn_buf_upload / vbnn_buf_download (the Lua host's only memory path),
 *   - with --comm, the data-parallel exchange (vbnn_comm_*: RCCL bound by dlopen from the system's librccl.so.1, not a
 *     PyTorch copy) with a world of one: every bucket of the step goes through ncclAllReduce on the communicator's stream.
 * tests/test_c_host_gpu.py builds it with gcc, runs it as a child process and compares its gradient arena BITWISE with
 * vbnn_amd/engine.py:FusedMLP on the same configuration.
 *
 *   c_host --dtype f32|bf16 --input 784 --hidden 400,400 --classes 10 --batch 256 [--S 1] [--steps 2] [--update]
 *          [--comm [--sharded]] [--graph] [--kl-shadows] [--seed 3] --out arena.bin
 *   --graph: the context gets a stream of its own (vbnn_ctx_create_cu_budget), the draw counter lives on the device
 *   (vbnn_fwd_args.draw_dev, vbnn_sample), step 2 is CAPTURED (vbnn_capture_begin / _end) and steps 2.. are replays of it.
 *   arena.bin: int64 n_grads, double loss, int32 correct, int32 flags, then n_grads floats (the arena after the last
 *   step), then per VB layer the O x I means after the last update (only with --update).
 *
 * Scope: LRT, total gradients from the accGradParameters epilogue, fused classifier head (n_classes <= 16) -- the
 * configuration bench.py times. Shapes whose GEMMs do not take K-major operands get the transposed copies, as engine.py
 * gives them (vbnn_kmajor_supported*).
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>

#include "vbnn_hip.h"

#define MAX_LAYERS 8
#define STREAM_DATA 4
#define STREAM_HEINIT 5

static vbnn_ctx* g_ctx;

static void check(int st, const char* what) {
    if (st != VBNN_OK) {
        fprintf(stderr, "c_host: %s failed with status %d: %s\n", what, st, vbnn_last_error());
        exit(2);
    }
}
#define CHECK(call) check((call), #call)

static void* dev_alloc(size_t bytes) {                       /* zero-initialised (the pads of packed operands stay zero) */
    void* p = NULL;
    CHECK(vbnn_buf_alloc(g_ctx, bytes ? bytes : 16, &p));
    return p;
}

typedef struct { void* p; int64_t ld; } packed_t;            /* rows x pad64(cols) of the operand type */
static int64_t pad_ld(int64_t k) { return (k + VBNN_KPAD - 1) / VBNN_KPAD * VBNN_KPAD; }
static packed_t packed(int64_t rows, int64_t cols, int esize) {
    packed_t t;
    t.ld = pad_ld(cols);
    t.p = dev_alloc((size_t)rows * t.ld * esize);
    return t;
}

typedef struct {
    int64_t I, O;
    uint32_t layer_id;
    float *means, *lvars, *bias;
    float *m_mu, *v_mu, *m_lv, *v_lv;                        /* Adam state (VBLinear.lua:31-33) */
    float *grad_lv, *grad_mu, *gradBias;                     /* views of the arena */
    int64_t bucket_off, bucket_n;
    double* stats;
    packed_t mu_s, var_s, muT_s, varT_s;
    packed_t x_s, x2_s, xT_s, x2T_s, g_s, gv_s, gT_s, gvT_s;
    void* r;
    int bias_from_dw, dw_km, dx_km, use_muT, early_ok, has_t;
    const void* x_in; int64_t ld_in;                       /* this run's input operand of the layer (x_s, or the raw minibatch) */
    int64_t t;
} layer_t;

typedef struct {
    int dtype, esize, n_layers, n_classes, world, rank, dx_first;
    int sharded;              /* --sharded: the sharded-update exchange (reduce-scatter by layer rows, slice update, shadow all-gather) */
    double* stat_parts;       /* [world][layers][4] doubles: every rank's statistics of its row slices */
    int n_head_slots; float* head_slots;   /* vbnn_forward_head_slots x N x 16 floats, or 0 / NULL */
    int kl_in_update;         /* 1 (default for bf16): the arena holds the likelihood parts, vbnn_update adds the exact fp32 KL gradient (kl_add) */
    int direct;               /* fp32: operands as their producers left them (no packing launch, no squares, no transposes) */
    uint64_t seed;
    float B, S;
    double var_init;
    int64_t sizes[MAX_LAYERS + 1];
    int64_t n_grads;
    float* grads;
    layer_t vb[MAX_LAYERS];
    float *weight3, *bias3, *gradWeight3, *gradBias3;
    packed_t w3_s, h_s;
    float *logits, *out, *g_logits;
    double* acc;
    int32_t* corr;
    uint32_t draw;
    uint32_t* draw_dev;       /* --graph: the draw counter in device memory (vbnn_fwd_args.draw_dev), advanced by vbnn_sample */
    int first;
    int64_t N;
    vbnn_comm* comm;
} fused_mlp;

/* ---- FusedMLP.new (lua/FusedMLP.lua; engine.py:FusedMLP.__init__ + init_parameters) */
static void fm_prepare(fused_mlp* m);
static void fm_new(fused_mlp* m, int dtype, const int64_t* sizes, int n_layers, int n_classes, uint64_t seed, double var_init,
                   float B, float S, int with_comm) {
    memset(m, 0, sizeof *m);
    m->dtype = dtype; m->esize = dtype == VBNN_BF16 ? 2 : 4;
    m->n_layers = n_layers; m->n_classes = n_classes; m->seed = seed; m->B = B; m->S = S; m->var_init = var_init;
    m->world = 1; m->rank = 0;
    m->kl_in_update = dtype == VBNN_BF16;                          /* engine.py: opt.kl_in_update's default (main: --kl-shadows turns it off) */
    memcpy(m->sizes, sizes, (size_t)(n_layers + 1) * sizeof(int64_t));
    /* gradient arena: [d/dlvars | d/dmeans | d/dbias] per VB layer, then the final Linear (vbnn_amd/partition.py) */
    int64_t total = 0;
    for (int li = 0; li < n_layers; ++li) total += 2 * sizes[li] * sizes[li + 1] + sizes[li + 1];
    const int64_t H = sizes[n_layers];
    total += H * n_classes + n_classes;
    m->n_grads = total;
    m->grads = (float*)dev_alloc((size_t)total * 4);
    int64_t off = 0;
    for (int li = 0; li < n_layers; ++li) {
        layer_t* v = &m->vb[li];
        const int64_t I = sizes[li], O = sizes[li + 1];
        v->I = I; v->O = O; v->layer_id = (uint32_t)li; v->bucket_off = off;
        v->means = (float*)dev_alloc((size_t)O * I * 4); v->lvars = (float*)dev_alloc((size_t)O * I * 4);
        v->bias = (float*)dev_alloc((size_t)O * 4);                                          /* VBLinear.lua:13 */
        v->m_mu = (float*)dev_alloc((size_t)O * I * 4); v->v_mu = (float*)dev_alloc((size_t)O * I * 4);
        v->m_lv = (float*)dev_alloc((size_t)O * I * 4); v->v_lv = (float*)dev_alloc((size_t)O * I * 4);
        v->grad_lv = m->grads + off; off += O * I;
        v->grad_mu = m->grads + off; off += O * I;
        v->gradBias = m->grads + off; off += O;
        v->bucket_n = off - v->bucket_off;
        v->stats = (double*)dev_alloc(32);
        v->mu_s = packed(O, I, m->esize); v->var_s = packed(O, I, m->esize);
        if (li > 0) { v->muT_s = packed(I, O, m->esize); v->varT_s = packed(I, O, m->esize); }
        v->use_muT = li > 0;
        /* the He rule of mlp.lua:47-55 for the means (bench / parity initialisation), lvars = log(var_init) (VBLinear.lua:18) */
        CHECK(vbnn_fill_normal(g_ctx, v->means, O, I, I, seed, STREAM_HEINIT, v->layer_id, 0, 0, (float)sqrt(2.0 / (double)I)));
        float* lv0 = (float*)malloc((size_t)O * I * 4);
        const float lv = (float)log(var_init);                   /* (the double 1e-3, as the Python host computes it) */
        for (int64_t k = 0; k < O * I; ++k) lv0[k] = lv;
        CHECK(vbnn_buf_upload(g_ctx, v->lvars, lv0, (size_t)O * I * 4));
        free(lv0);
    }
    m->weight3 = (float*)dev_alloc((size_t)n_classes * H * 4); m->bias3 = (float*)dev_alloc((size_t)n_classes * 4);
    m->gradWeight3 = m->grads + off; off += n_classes * H;
    m->gradBias3 = m->grads + off; off += n_classes;
    CHECK(vbnn_fill_normal(g_ctx, m->weight3, n_classes, H, H, seed, STREAM_HEINIT, (uint32_t)n_layers, 0, 0, (float)sqrt(2.0 / (double)H)));
    m->w3_s = packed(n_classes, H, m->esize);
    m->acc = (double*)dev_alloc(16); m->corr = (int32_t*)dev_alloc(4);
    m->draw = 0; m->first = 1;
    if (with_comm) {                                              /* the exchange (include/vbnn_hip.h: vbnn_comm_*) */
        unsigned char id[VBNN_COMM_ID_BYTES];
        CHECK(vbnn_comm_unique_id(id));
        CHECK(vbnn_comm_create(g_ctx, m->rank, m->world, id, &m->comm));
        int r = -1, w = -1, n = -1;
        CHECK(vbnn_comm_info(m->comm, &r, &w, &n));
        if (r != 0 || w != 1 || n != 1) { fprintf(stderr, "c_host: communicator reports rank %d of %d (%d in comm)\n", r, w, n); exit(2); }
    }
    /* single GPU, layers of different sizes: every updateGradInput first, then the accGradParameters from the first layer
       up (engine.py: dx_first) */
    double wmin = 1e300, wmax = 0;
    for (int k = 0; k < n_layers; ++k) {
        const double w = (double)sizes[k] * (double)sizes[k + 1];
        if (w < wmin) wmin = w;
        if (w > wmax) wmax = w;
    }
    m->dx_first = !m->comm && (2 * wmin <= wmax);
    fm_prepare(m);
}

/* ---- buffers that depend on the local batch size (engine.py:_alloc_batch) */
static void fm_alloc_batch(fused_mlp* m, int64_t N) {
    if (m->N == N) return;
    if (m->N) { fprintf(stderr, "c_host: one batch size per process\n"); exit(2); }
    m->N = N;
    const int km_ok = m->dtype == VBNN_BF16 && (int)m->S == 1;
    /* fp32 (the general kernel): the minibatch raw, x.x formed in registers, x / g / mu / sigma^2 K-major, the bias gradient
       from a synthetic row of ones (engine.py: f32_direct). Not with an exchange (its two-launch accGradParameters wants x.x) */
    m->direct = m->dtype == VBNN_F32 && !m->comm;
    int need_prepare = 0;
    float* ones_host = (float*)malloc((size_t)N * 4);
    for (int64_t k = 0; k < N; ++k) ones_host[k] = 1.0f;
    float* ones_dev = (float*)dev_alloc((size_t)N * 4);
    CHECK(vbnn_buf_upload(g_ctx, ones_dev, ones_host, (size_t)N * 4));
    free(ones_host);
    for (int li = 0; li < m->n_layers; ++li) {
        layer_t* v = &m->vb[li];
        const int last = li == m->n_layers - 1;
        v->bias_from_dw = (v->I % 256 != 0) && !last;             /* the ones column / row of x: bias gradient from the GEMM */
        if (m->direct) {
            v->dw_km = 1; v->early_ok = 0; v->dx_km = li > 0; v->use_muT = 0; v->has_t = 0;
            v->x_s = packed(N, v->I, m->esize);                   /* (layer 1: only when the raw minibatch cannot be read in place) */
            v->g_s = packed(N, v->O, m->esize); v->gv_s = packed(N, v->O, m->esize);
            v->r = dev_alloc((size_t)N * v->O * m->esize);
            continue;
        }
        const int km = km_ok ? vbnn_kmajor_supported_dw(v->I, v->O, N, v->bias_from_dw) : 0;
        v->dw_km = km > 0;
        v->early_ok = !v->dw_km || (!v->bias_from_dw && vbnn_kmajor_supported(v->I, v->O, N));
        v->dx_km = km_ok && li > 0 && vbnn_kmajor_supported(v->I, N, v->O);
        const int use_muT = li > 0 && !v->dx_km;
        if (use_muT && !v->use_muT) need_prepare = 1;
        v->use_muT = use_muT;
        const int extra = v->bias_from_dw ? 1 : 0;
        int64_t xcols = v->I + (v->dw_km ? extra : 0);
        if (km == 2) xcols = (xcols + 255) / 256 * 256;
        v->x_s = packed(N, xcols, m->esize); v->x2_s = packed(N, xcols, m->esize);
        v->has_t = !v->dw_km;
        if (v->dw_km && v->bias_from_dw)                          /* column I of x is all ones, written once */
            CHECK(vbnn_pack(g_ctx, m->dtype, VBNN_PACK_COPY, ones_dev, NULL, 1, N, 1, (char*)v->x_s.p + v->I * m->esize, v->x_s.ld, NULL, 0));
        if (v->has_t) {
            v->xT_s = packed(v->I + extra, N, m->esize); v->x2T_s = packed(v->I + extra, N, m->esize);
            v->gT_s = packed(v->O, N, m->esize); v->gvT_s = packed(v->O, N, m->esize);
            if (v->bias_from_dw)                                  /* row I of x^T is all ones */
                CHECK(vbnn_pack(g_ctx, m->dtype, VBNN_PACK_COPY, ones_dev, NULL, N, 1, N, (char*)v->xT_s.p + v->I * v->xT_s.ld * m->esize,
                                v->xT_s.ld, NULL, 0));
        }
        v->g_s = packed(N, v->O, m->esize); v->gv_s = packed(N, v->O, m->esize);
        v->r = dev_alloc((size_t)N * v->O * m->esize);
    }
    m->h_s = packed(N, m->sizes[m->n_layers], m->esize);
    m->logits = (float*)dev_alloc((size_t)N * m->n_classes * 4);
    m->out = (float*)dev_alloc((size_t)N * m->n_classes * 4);
    m->g_logits = (float*)dev_alloc((size_t)N * m->n_classes * 4);
    /* the head's logits from the last VB layer's forward tiles, where that launch can carry them (vbnn_fwd_args.head_slots) */
    {
        layer_t* vl = &m->vb[m->n_layers - 1];
        m->n_head_slots = m->draw_dev ? 0 : vbnn_forward_head_slots(g_ctx, m->dtype, N, vl->I, vl->O, m->n_classes);
        m->head_slots = m->n_head_slots > 0 ? (float*)dev_alloc((size_t)m->n_head_slots * N * 16 * 4) : NULL;
    }
    if (need_prepare) fm_prepare(m);
}

static void fm_reset_gradients(fused_mlp* m) { m->first = 1; }          /* mlp.lua:62-67: the first draw overwrites */

/* VBLinear:compute_prior (VBLinear.lua:77-88) + the operand shadows, once; afterwards vbnn_update maintains both */
static void fm_prepare(fused_mlp* m) {
    vbnn_prep_desc d[MAX_LAYERS];
    memset(d, 0, sizeof d);
    for (int k = 0; k < m->n_layers; ++k) {
        layer_t* v = &m->vb[k];
        d[k].means = v->means; d[k].lvars = v->lvars; d[k].O = v->O; d[k].I = v->I;
        d[k].mu_s = v->mu_s.p; d[k].var_s = v->var_s.p; d[k].ld_w = v->mu_s.ld;
        d[k].muT_s = v->use_muT ? v->muT_s.p : NULL; d[k].varT_s = v->use_muT ? v->varT_s.p : NULL; d[k].ld_wT = v->muT_s.ld;
        d[k].stats = v->stats;
    }
    vbnn_pack_desc w3;
    memset(&w3, 0, sizeof w3);
    const int64_t H = m->sizes[m->n_layers];
    w3.src = m->weight3; w3.rows = m->n_classes; w3.cols = H; w3.ld_src = H;
    w3.dst = m->w3_s.p; w3.ld_dst = m->w3_s.ld; w3.dstT = NULL; w3.ld_dstT = 0;
    CHECK(vbnn_prepare(g_ctx, m->dtype, m->n_layers, d, &w3));
}

static void fm_sample(fused_mlp* m) {                          /* mlp.lua:69-74: LRT draws its noise in the forward epilogue */
    m->draw += 1;
    if (m->draw_dev) CHECK(vbnn_sample(g_ctx, m->draw_dev, 1));   /* capturable: the counter lives on the device */
}

static void dw_block(fused_mlp* m, int li, int64_t N, int accumulate, vbnn_dw_args* d) {
    layer_t* v = &m->vb[li];
    memset(d, 0, sizeof *d);
    if (v->has_t) { d->xT = v->xT_s.p; d->x2T = v->x2T_s.p; d->gT = v->gT_s.p; d->gvT = v->gvT_s.p; d->ld_n = v->gT_s.ld; }
    d->N = N; d->I = v->I; d->O = v->O; d->scale = 1.0f; d->accumulate = accumulate;
    d->seed = m->seed; d->layer = v->layer_id; d->draw = m->draw; d->lvars = v->lvars;
    d->grad_mu = v->grad_mu; d->grad_lv = v->grad_lv; d->means = v->means; d->stats = v->stats;
    /* the KL gradient: exact, from the fp32 parameters in the update sweep (default where the epilogue would read the bf16 shadows:
       (bf16(s2) / var_hat - 1) cancels, VBLinear.lua:96-97 uses the fp32 vars) -- or fused here (--kl-shadows, the A/B form) */
    d->B = m->B; d->S = m->S; d->kl_scale = m->kl_in_update ? 0.0f : 1.0f / (float)m->world;
    d->gradBias = v->bias_from_dw ? v->gradBias : NULL;
    d->x = v->x_in; d->x2 = m->direct ? NULL : v->x2_s.p; d->g = v->g_s.p; d->gv = v->gv_s.p; d->ld_x = v->ld_in; d->ld_g = v->g_s.ld;
    if (m->dtype == VBNN_BF16) { d->mu_s = v->mu_s.p; d->var_s = v->var_s.p; d->ld_w = v->mu_s.ld; }   /* KL terms from the shadows */
}

static void dx_block(fused_mlp* m, int li, int64_t N, vbnn_dx_args* a) {
    layer_t *v = &m->vb[li], *p = &m->vb[li - 1];
    memset(a, 0, sizeof *a);
    if (v->use_muT) { a->wT = v->muT_s.p; a->w2T = v->varT_s.p; }
    a->ld_wT = v->muT_s.ld;
    a->g = v->g_s.p; a->gv = v->gv_s.p; a->ld_g = v->g_s.ld; a->N = N; a->I = v->I; a->O = v->O;
    a->x = v->x_s.p; a->ld_x = v->x_s.ld; a->relu_mask = 1;
    a->r_prev = p->r; a->ld_r_prev = p->O; a->r_prev_packed = 1;
    a->g_prev = p->g_s.p; a->gv_prev = p->gv_s.p; a->ld_gp = p->g_s.ld;
    if (p->has_t) { a->gT_prev = p->gT_s.p; a->gvT_prev = p->gvT_s.p; a->ld_gpT = p->gT_s.ld; }
    a->w = v->mu_s.p; a->w2 = v->var_s.p; a->ld_w = v->mu_s.ld;
}

/* sharded-update exchange (engine.py: _scatter): layer li's messages after its accGradParameters -- `lv` / `mu`: reduce-scatter of the
   d/dlvars / d/dmeans region by layer rows (rank r keeps the sums of rows [r O / G, (r + 1) O / G)); `small`: the bias gradient (and,
   behind the last layer's, the final Linear's) as a plain all-reduce */
static void fm_scatter(fused_mlp* m, int li, int lv, int mu, int small) {
    layer_t* v = &m->vb[li];
    const int64_t per = v->O * v->I / m->world;
    if (lv) CHECK(vbnn_comm_reduce_scatter(m->comm, m->grads + v->bucket_off, per));
    if (mu) CHECK(vbnn_comm_reduce_scatter(m->comm, m->grads + v->bucket_off + v->O * v->I, per));
    if (small) {
        const int64_t off = v->bucket_off + 2 * v->O * v->I;
        const int64_t end = (li == m->n_layers - 1) ? m->n_grads : (v->bucket_off + v->bucket_n);
        CHECK(vbnn_allreduce_grads(m->comm, m->grads + off, end - off));
    }
}

/* ---- mlp.lua:76-84, fused. inputs: DEVICE pointer to N x input_size floats (row pitch ld), targets: device int32[N], 0-based */
static void fm_run(fused_mlp* m, const float* inputs, int64_t ld, const int32_t* targets, int64_t N) {
    fm_alloc_batch(m, N);
    const int accumulate = m->first ? 0 : 1;
    const float inv_n = (float)(1.0 / (double)(N * m->world));
    const int64_t row0 = (int64_t)m->rank * N;
    const int nl = m->n_layers;
    layer_t* v0 = &m->vb[0];
    for (int li = 0; li < nl; ++li) { m->vb[li].x_in = m->vb[li].x_s.p; m->vb[li].ld_in = m->vb[li].x_s.ld; }
    if (m->direct && ld % 4 == 0 && ((uintptr_t)inputs & 15u) == 0) {
        v0->x_in = inputs; v0->ld_in = ld;                        /* the GEMMs read the minibatch where it lies: no packing launch */
    } else {
        CHECK(vbnn_pack_input(g_ctx, m->dtype, inputs, ld, N, v0->I, v0->x_s.p, m->direct ? NULL : v0->x2_s.p, v0->x_s.ld,
                              v0->has_t ? v0->xT_s.p : NULL, v0->has_t ? v0->x2T_s.p : NULL, v0->has_t ? v0->xT_s.ld : 0, 0));
    }
    /* forward: dual GEMM + noise / ReLU / operand packing in the epilogue */
    for (int li = 0; li < nl; ++li) {
        layer_t* v = &m->vb[li];
        layer_t* nxt = li + 1 < nl ? &m->vb[li + 1] : NULL;
        vbnn_fwd_args fa;
        memset(&fa, 0, sizeof fa);
        fa.w = v->mu_s.p; fa.w2 = v->var_s.p; fa.x = v->x_in; fa.x2 = m->direct ? NULL : v->x2_s.p; fa.ld_w = v->mu_s.ld; fa.ld_x = v->ld_in;
        fa.N = N; fa.I = v->I; fa.O = v->O; fa.bias = v->bias;
        fa.seed = m->seed; fa.layer = v->layer_id; fa.draw = m->draw_dev ? 0 : m->draw; fa.draw_dev = m->draw_dev; fa.row0 = row0;
        fa.r = v->r; fa.ld_r = v->O; fa.r_packed = 1; fa.relu = 1;
        fa.h = nxt ? nxt->x_s.p : m->h_s.p;
        fa.h2 = (nxt && !m->direct) ? nxt->x2_s.p : NULL;
        fa.ld_h = nxt ? nxt->x_s.ld : m->h_s.ld;
        if (nxt && nxt->has_t) { fa.hT = nxt->xT_s.p; fa.h2T = nxt->x2T_s.p; fa.ld_hT = nxt->xT_s.ld; }
        if (!nxt && m->n_head_slots > 0) { fa.head_w3 = m->w3_s.p; fa.head_ld_w = m->w3_s.ld; fa.head_C = m->n_classes; fa.head_slots = m->head_slots; }
        CHECK(vbnn_forward(g_ctx, m->dtype, &fa));
    }
    /* final Linear + LogSoftMax + ClassNLL (mlp.lua:29-32), forward and backward */
    layer_t* vl = &m->vb[nl - 1];
    const int64_t H = m->sizes[nl];
    vbnn_head_args ha;
    memset(&ha, 0, sizeof ha);
    ha.h = m->h_s.p; ha.ld_h = m->h_s.ld; ha.w3 = m->w3_s.p; ha.ld_w = m->w3_s.ld; ha.bias = m->bias3; ha.target = targets;
    ha.N = N; ha.H = H; ha.C = m->n_classes; ha.rows_per_draw = 0; ha.inv_n = inv_n; ha.accumulate = accumulate;
    ha.logits = m->logits; ha.out = m->out; ha.g_logits = m->g_logits; ha.loss_sum_dev = m->acc; ha.correct_dev = m->corr;
    ha.gradWeight = m->gradWeight3; ha.gradBias = m->gradBias3; ha.gradBias_prev = vl->gradBias;
    ha.relu_mask = 1; ha.r_prev_packed = 1; ha.r_prev = vl->r; ha.ld_r_prev = vl->O;
    ha.g_prev = vl->g_s.p; ha.gv_prev = vl->gv_s.p; ha.ld_gp = vl->g_s.ld;
    if (vl->has_t) { ha.gT_prev = vl->gT_s.p; ha.gvT_prev = vl->gvT_s.p; ha.ld_gpT = vl->gT_s.ld; }
    if (m->n_head_slots > 0) { ha.logit_slots = m->head_slots; ha.n_slots = m->n_head_slots; }
    CHECK(vbnn_head_forward_backward(g_ctx, m->dtype, &ha));
    vbnn_dw_args dd;
    vbnn_dx_args xa;
    if (m->direct) {
        /* fp32: accGradParameters and updateGradInput of a layer are independent and go out as ONE launch where the library
           can carry both (vbnn_backward_pair); each tile bitwise what its own launch computes */
        for (int li = nl - 1; li >= 0; --li) {
            layer_t* v = &m->vb[li];
            dw_block(m, li, N, accumulate, &dd);
            if (li > 0) {
                dx_block(m, li, N, &xa);
                CHECK(vbnn_backward_pair(g_ctx, m->dtype, &xa, &dd));
            } else {
                CHECK(vbnn_acc_grad_parameters(g_ctx, m->dtype, &dd));
            }
            if (li < nl - 1 && !v->bias_from_dw)
                CHECK(vbnn_acc_grad_bias(g_ctx, m->dtype, v->g_s.p, v->g_s.ld, N, v->O, 1.0f, accumulate, v->gradBias));
        }
    } else if (m->dx_first) {
        for (int li = nl - 1; li >= 1; --li) {
            dx_block(m, li, N, &xa);
            CHECK(vbnn_grad_input(g_ctx, m->dtype, &xa));
        }
        for (int li = 0; li < nl; ++li) {
            layer_t* v = &m->vb[li];
            dw_block(m, li, N, accumulate, &dd);
            CHECK(vbnn_acc_grad_parameters(g_ctx, m->dtype, &dd));
            if (li < nl - 1 && !v->bias_from_dw)
                CHECK(vbnn_acc_grad_bias(g_ctx, m->dtype, v->g_s.p, v->g_s.ld, N, v->O, 1.0f, accumulate, v->gradBias));
        }
    } else {
        /* last VB layer first: accGradParameters (+ its bucket's all-reduce), then updateGradInput */
        for (int li = nl - 1; li >= 0; --li) {
            layer_t* v = &m->vb[li];
            dw_block(m, li, N, accumulate, &dd);
            int64_t msg_off = v->bucket_off;
            const int early = m->comm && v->O * v->I >= (1 << 22) && v->early_ok;
            if (early) {
                /* two launches (vbnn_dw_args.part): the sigma^2 GEMM and d/dlvars first, whose exchange then starts while
                   the mu GEMM still runs (d/dlvars is the first block of the layer's bucket) */
                dd.part = 2;
                CHECK(vbnn_acc_grad_parameters(g_ctx, m->dtype, &dd));
                if (m->sharded) fm_scatter(m, li, 1, 0, 0);
                else CHECK(vbnn_allreduce_grads(m->comm, m->grads + v->bucket_off, v->O * v->I));
                dd.part = 1;
                msg_off = v->bucket_off + v->O * v->I;
            }
            CHECK(vbnn_acc_grad_parameters(g_ctx, m->dtype, &dd));
            if (li < nl - 1 && !v->bias_from_dw)
                CHECK(vbnn_acc_grad_bias(g_ctx, m->dtype, v->g_s.p, v->g_s.ld, N, v->O, 1.0f, accumulate, v->gradBias));
            if (m->comm && m->sharded) {
                fm_scatter(m, li, !early, 1, 1);
            } else if (m->comm) {                                 /* the final Linear's gradients ride in the last layer's message */
                const int64_t n = ((li == nl - 1) ? m->n_grads : (v->bucket_off + v->bucket_n)) - msg_off;
                CHECK(vbnn_allreduce_grads(m->comm, m->grads + msg_off, n));
            }
            if (li > 0) {
                dx_block(m, li, N, &xa);
                CHECK(vbnn_grad_input(g_ctx, m->dtype, &xa));
            }
        }
    }
    m->first = 0;
}

static void fm_finish(fused_mlp* m) {                             /* end of the minibatch: gradients complete on the stream */
    if (m->comm) CHECK(vbnn_comm_finish(m->comm));
}

/* mlp:update + VBLinear:update (mlp.lua:117-142, VBLinear.lua:124-166) in one call, which also leaves the operand
   shadows and prior statistics of the next minibatch */
/* the same update with the parameters SHARDED by layer rows (engine.py: _update_sharded): vbnn_update on this rank's rows, then the
   all-gather of the operand shadows and of the slices' statistics, vbnn_stats_combine, transposed shadows rebuilt locally */
static void fm_update_sharded(fused_mlp* m, float lr, float lr_mu, float lr_lv) {
    const int64_t H = m->sizes[m->n_layers];
    const int G = m->world, R = m->rank, n = m->n_layers;
    CHECK(vbnn_sgd_step(g_ctx, m->weight3, m->gradWeight3, m->n_classes * H, lr));
    CHECK(vbnn_sgd_step(g_ctx, m->bias3, m->gradBias3, m->n_classes, lr));
    vbnn_update_desc d[MAX_LAYERS];
    memset(d, 0, sizeof d);
    double* mine = m->stat_parts + (size_t)R * n * 4;
    for (int k = 0; k < n; ++k) {
        layer_t* v = &m->vb[k];
        CHECK(vbnn_sgd_step(g_ctx, v->bias, v->gradBias, v->O, lr));
        v->t += 1;
        const int64_t nr = v->O / G, r0 = (int64_t)R * nr, o = r0 * v->I;
        double st[4];                                             /* in: the WHOLE layer's pre-update statistics */
        CHECK(vbnn_buf_download(g_ctx, st, v->stats, sizeof st));
        CHECK(vbnn_buf_upload(g_ctx, mine + 4 * k, st, sizeof st));
        vbnn_update_desc* e = &d[k];
        e->means = v->means + o; e->lvars = v->lvars + o; e->O = nr; e->I = v->I;
        e->mu_s = (char*)v->mu_s.p + r0 * v->mu_s.ld * m->esize; e->var_s = (char*)v->var_s.p + r0 * v->var_s.ld * m->esize; e->ld_w = v->mu_s.ld;
        e->stats = mine + 4 * k; e->grad_mu = v->grad_mu + o; e->grad_lv = v->grad_lv + o;
        e->m_mu = v->m_mu + o; e->v_mu = v->v_mu + o; e->m_lv = v->m_lv + o; e->v_lv = v->v_lv + o;     /* (allocated whole here; a rank touches its rows) */
        e->mu.lr = lr_mu; e->mu.beta1 = 0.9f; e->mu.beta2 = 0.999f; e->mu.eps = 1e-8f; e->mu.lambda = 1.0f; e->mu.t = v->t;
        e->lv.lr = lr_lv; e->lv.beta1 = 0.9f; e->lv.beta2 = 0.999f; e->lv.eps = 1e-8f; e->lv.lambda = 1.0f; e->lv.t = v->t;
        e->lr_bias = lr; e->B = m->B; e->kl_add = 1.0f;
    }
    vbnn_pack_desc w3;
    memset(&w3, 0, sizeof w3);
    w3.src = m->weight3; w3.rows = m->n_classes; w3.cols = H; w3.ld_src = H;
    w3.dst = m->w3_s.p; w3.ld_dst = m->w3_s.ld; w3.dstT = NULL; w3.ld_dstT = 0;
    CHECK(vbnn_update(g_ctx, m->dtype, n, d, &w3));
    double* stats[MAX_LAYERS];
    for (int k = 0; k < n; ++k) {
        layer_t* v = &m->vb[k];
        CHECK(vbnn_comm_all_gather(m->comm, v->mu_s.p, v->O / G * v->mu_s.ld * m->esize));
        CHECK(vbnn_comm_all_gather(m->comm, v->var_s.p, v->O / G * v->var_s.ld * m->esize));
        stats[k] = v->stats;
    }
    CHECK(vbnn_comm_all_gather(m->comm, m->stat_parts, (int64_t)n * 4 * 8));
    CHECK(vbnn_comm_finish(m->comm));
    CHECK(vbnn_stats_combine(g_ctx, n, G, m->stat_parts, stats));
    for (int k = 0; k < n; ++k) {
        layer_t* v = &m->vb[k];
        if (!v->use_muT) continue;
        CHECK(vbnn_transpose_packed(g_ctx, m->dtype, v->mu_s.p, v->mu_s.ld, v->O, v->I, v->muT_s.p, v->muT_s.ld));
        CHECK(vbnn_transpose_packed(g_ctx, m->dtype, v->var_s.p, v->var_s.ld, v->O, v->I, v->varT_s.p, v->varT_s.ld));
    }
}

static void fm_update(fused_mlp* m, float lr, float lr_mu, float lr_lv) {
    fm_finish(m);
    if (m->sharded) { fm_update_sharded(m, lr, lr_mu, lr_lv); return; }
    const int64_t H = m->sizes[m->n_layers];
    CHECK(vbnn_sgd_step(g_ctx, m->weight3, m->gradWeight3, m->n_classes * H, lr));
    CHECK(vbnn_sgd_step(g_ctx, m->bias3, m->gradBias3, m->n_classes, lr));
    vbnn_update_desc d[MAX_LAYERS];
    memset(d, 0, sizeof d);
    for (int k = 0; k < m->n_layers; ++k) {
        layer_t* v = &m->vb[k];
        v->t += 1;
        vbnn_update_desc* e = &d[k];
        e->means = v->means; e->lvars = v->lvars; e->O = v->O; e->I = v->I;
        e->mu_s = v->mu_s.p; e->var_s = v->var_s.p; e->ld_w = v->mu_s.ld;
        e->muT_s = v->use_muT ? v->muT_s.p : NULL; e->varT_s = v->use_muT ? v->varT_s.p : NULL; e->ld_wT = v->muT_s.ld;
        e->stats = v->stats; e->grad_mu = v->grad_mu; e->grad_lv = v->grad_lv;
        e->m_mu = v->m_mu; e->v_mu = v->v_mu; e->m_lv = v->m_lv; e->v_lv = v->v_lv;
        e->mu.lr = lr_mu; e->mu.beta1 = 0.9f; e->mu.beta2 = 0.999f; e->mu.eps = 1e-8f; e->mu.lambda = 1.0f; e->mu.t = v->t;
        e->lv.lr = lr_lv; e->lv.beta1 = 0.9f; e->lv.beta2 = 0.999f; e->lv.eps = 1e-8f; e->lv.lambda = 1.0f; e->lv.t = v->t;
        e->bias = v->bias; e->grad_bias = v->gradBias; e->lr_bias = lr; e->B = m->B;
        e->log14 = NULL;
        e->kl_add = m->kl_in_update ? 1.0f : 0.0f;
    }
    vbnn_pack_desc w3;
    memset(&w3, 0, sizeof w3);
    w3.src = m->weight3; w3.rows = m->n_classes; w3.cols = H; w3.ld_src = H;
    w3.dst = m->w3_s.p; w3.ld_dst = m->w3_s.ld; w3.dstT = NULL; w3.ld_dstT = 0;
    CHECK(vbnn_update(g_ctx, m->dtype, m->n_layers, d, &w3));
}

/* error (mean NLL over the GLOBAL batch, this rank's share) and hit count of the last run(s); synchronises */
static void fm_loss_and_accuracy(fused_mlp* m, double* loss, int32_t* correct) {
    fm_finish(m);
    double a[2];
    CHECK(vbnn_buf_download(g_ctx, a, m->acc, 16));
    CHECK(vbnn_buf_download(g_ctx, correct, m->corr, 4));
    *loss = a[0];
}

static const char* arg_value(int argc, char** argv, const char* name, const char* dflt) {
    for (int i = 1; i + 1 < argc; ++i)
        if (!strcmp(argv[i], name)) return argv[i + 1];
    return dflt;
}
static int arg_flag(int argc, char** argv, const char* name) {
    for (int i = 1; i < argc; ++i)
        if (!strcmp(argv[i], name)) return 1;
    return 0;
}

int main(int argc, char** argv) {
    const char* out_path = arg_value(argc, argv, "--out", NULL);
    if (!out_path) {
        fprintf(stderr, "usage: c_host --dtype f32|bf16 --input I --hidden h1,h2 --classes C --batch N [--S s] [--steps k] [--update] [--comm] --out file\n");
        return 1;
    }
    if (vbnn_abi_version() != VBNN_ABI_VERSION) { fprintf(stderr, "c_host: header / library ABI mismatch\n"); return 2; }
    const int dtype = !strcmp(arg_value(argc, argv, "--dtype", "bf16"), "f32") ? VBNN_F32 : VBNN_BF16;
    int64_t sizes[MAX_LAYERS + 1];
    int n_layers = 0;
    sizes[0] = atoll(arg_value(argc, argv, "--input", "784"));
    char hidden[256];
    strncpy(hidden, arg_value(argc, argv, "--hidden", "400,400"), sizeof hidden - 1);
    hidden[sizeof hidden - 1] = 0;
    for (char* tok = strtok(hidden, ","); tok && n_layers < MAX_LAYERS; tok = strtok(NULL, ",")) sizes[++n_layers] = atoll(tok);
    const int n_classes = atoi(arg_value(argc, argv, "--classes", "10"));
    const int64_t N = atoll(arg_value(argc, argv, "--batch", "256"));
    const int S = atoi(arg_value(argc, argv, "--S", "1"));
    const int steps = atoi(arg_value(argc, argv, "--steps", "2"));
    const uint64_t seed = (uint64_t)atoll(arg_value(argc, argv, "--seed", "3"));
    const int with_update = arg_flag(argc, argv, "--update"), with_comm = arg_flag(argc, argv, "--comm");
    const int with_graph = arg_flag(argc, argv, "--graph");      /* steps 2.. as replays of ONE captured graph (vbnn_capture_*) */
    if (n_layers < 1 || n_classes < 1 || n_classes > 16 || N < 1 || S < 1) { fprintf(stderr, "c_host: bad configuration\n"); return 1; }

    if (with_graph) {
        /* the NULL stream cannot be captured: a context with a stream of the library's own (no CU mask) */
        if (with_comm || with_update) { fprintf(stderr, "c_host: --graph captures the plain step (no --comm / --update)\n"); return 1; }
        CHECK(vbnn_ctx_create_cu_budget(atoi(arg_value(argc, argv, "--device", "0")), 0, &g_ctx));
    } else {
        CHECK(vbnn_ctx_create(atoi(arg_value(argc, argv, "--device", "0")), NULL, &g_ctx));
    }
    fused_mlp net;
    fm_new(&net, dtype, sizes, n_layers, n_classes, seed, 1e-3, 1e6f, (float)S, with_comm);
    if (arg_flag(argc, argv, "--kl-shadows")) net.kl_in_update = 0;
    if (arg_flag(argc, argv, "--sharded")) {                      /* the sharded-update exchange instead of the all-reduce (a world of one here) */
        if (!with_comm || dtype != VBNN_BF16) { fprintf(stderr, "c_host: --sharded needs --comm and --dtype bf16\n"); return 1; }
        net.sharded = 1; net.kl_in_update = 1;
        net.stat_parts = (double*)dev_alloc((size_t)net.world * n_layers * 4 * 8);
    }    /* A/B: the KL gradient fused into the accGradParameters epilogue, from the bf16 shadows */
    if (with_graph) net.draw_dev = (uint32_t*)dev_alloc(4);

    /* the synthetic minibatch of the parity tests: x ~ N(0,1) from the Philox contract (stream DATA), targets by row */
    float* x = (float*)dev_alloc((size_t)N * sizes[0] * 4);
    CHECK(vbnn_fill_normal(g_ctx, x, N, sizes[0], sizes[0], seed, STREAM_DATA, 0, 0, 0, 1.0f));
    int32_t* t_host = (int32_t*)malloc((size_t)N * 4);
    for (int64_t n = 0; n < N; ++n) t_host[n] = (int32_t)((n * 7) % n_classes);
    int32_t* t = (int32_t*)dev_alloc((size_t)N * 4);
    CHECK(vbnn_buf_upload(g_ctx, t, t_host, (size_t)N * 4));
    free(t_host);

    vbnn_graph* graph = NULL;
    int graph_nodes = 0;
    for (int step = 0; step < steps; ++step) {                   /* main.lua:28-40 */
        if (graph) { CHECK(vbnn_graph_launch(graph)); continue; }     /* every replay advances the device counter: its own noise */
        const int capture = with_graph && step == 1;             /* step 1 ran launch by launch (allocations, first launches) */
        if (capture) { CHECK(vbnn_sync(g_ctx)); CHECK(vbnn_capture_begin(g_ctx)); }
        fm_reset_gradients(&net);
        for (int s = 0; s < S; ++s) {
            fm_sample(&net);
            fm_run(&net, x, sizes[0], t, N);
        }
        fm_finish(&net);
        if (capture) {
            CHECK(vbnn_capture_end(g_ctx, &graph));
            CHECK(vbnn_graph_info(graph, &graph_nodes, NULL));
            CHECK(vbnn_graph_launch(graph));                      /* recorded, not run: this is step 2 */
        }
        if (with_update && step + 1 < steps) fm_update(&net, 1e-3f, 1e-4f, 5e-2f);
    }
    double loss = 0;
    int32_t correct = 0;
    fm_loss_and_accuracy(&net, &loss, &correct);
    CHECK(vbnn_sync(g_ctx));

    float* arena = (float*)malloc((size_t)net.n_grads * 4);
    CHECK(vbnn_buf_download(g_ctx, arena, net.grads, (size_t)net.n_grads * 4));
    FILE* f = fopen(out_path, "wb");
    if (!f) { perror(out_path); return 2; }
    const int32_t flags = (with_update ? 1 : 0) | (with_comm ? 2 : 0) | (net.dx_first ? 4 : 0);
    fwrite(&net.n_grads, 8, 1, f); fwrite(&loss, 8, 1, f); fwrite(&correct, 4, 1, f); fwrite(&flags, 4, 1, f);
    fwrite(arena, 4, (size_t)net.n_grads, f);
    if (with_update)
        for (int li = 0; li < n_layers; ++li) {
            const size_t n = (size_t)sizes[li] * sizes[li + 1];
            float* mu = (float*)malloc(n * 4);
            CHECK(vbnn_buf_download(g_ctx, mu, net.vb[li].means, n * 4));
            fwrite(mu, 4, n, f);
            free(mu);
        }
    fclose(f);
    free(arena);
    printf("c_host: %s %lld", dtype == VBNN_F32 ? "f32" : "bf16", (long long)sizes[0]);
    for (int li = 1; li <= n_layers; ++li) printf("-%lld", (long long)sizes[li]);
    printf("-%d batch %lld S %d steps %d%s%s: loss %.9g, %d correct, order %s", n_classes, (long long)N, S, steps, with_update ? " +update" : "",
           with_comm ? " +rccl(world 1)" : "", loss, correct, net.dx_first ? "dx-first" : "layerwise");
    for (int li = 0; li < n_layers; ++li) printf(" | L%d dw_km %d dx_km %d bias_from_dw %d", li, net.vb[li].dw_km, net.vb[li].dx_km, net.vb[li].bias_from_dw);
    printf("\n");
    const int timed = atoi(arg_value(argc, argv, "--time", "0"));
    if (timed > 0) {           /* --time K: K more steps (after the results above were taken), wall clock around issue + vbnn_sync */
        struct timespec t0, t1;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(vbnn_sync(g_ctx));
            clock_gettime(CLOCK_MONOTONIC, &t0);
            for (int step = 0; step < timed; ++step) {
                if (graph) { CHECK(vbnn_graph_launch(graph)); continue; }
                fm_reset_gradients(&net);
                for (int s = 0; s < S; ++s) { fm_sample(&net); fm_run(&net, x, sizes[0], t, N); }
                fm_finish(&net);
            }
            CHECK(vbnn_sync(g_ctx));
            clock_gettime(CLOCK_MONOTONIC, &t1);
            printf("c_host: %d steps%s: %.2f us per step (issue + sync, wall clock)\n", timed, graph ? " (graph replays)" : "",
                   ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / 1e3 / timed);
        }
    }
    if (graph) { printf("c_host: steps 2..%d were replays of one captured graph of %d kernel nodes\n", steps, graph_nodes); CHECK(vbnn_graph_destroy(graph)); }
    if (net.comm) CHECK(vbnn_comm_destroy(net.comm));
    CHECK(vbnn_ctx_destroy(g_ctx));                               /* (device buffers are released with the process) */
    return 0;
}
