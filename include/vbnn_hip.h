/*
 * vbnn_hip.h -- C ABI of the MI355X-native VBLinear hot path (libvbnn_hip.so).
 *
 * The reference (louissmit/VBNN) has no FFI of its own: its hot path is Lua calling
 * Torch7 tensor ops. Each entry point below therefore cites the Lua call site it
 * replaces (file:line under the reference tree). The host side that binds this ABI is
 *   - vbnn_amd/nn.py      (ctypes; same class/method names as VBLinear.lua / mlp.lua)
 *   - lua/VBLinear.lua    (LuaJIT FFI shim)
 *
 * Conventions
 *   - every function returns an int status (0 = VBNN_OK); the message of the last
 *     failure on the calling thread is vbnn_last_error(). No C++ exception crosses.
 *   - all tensor pointers are DEVICE pointers (hipMalloc'ed by anyone: this library's
 *     vbnn_buf_alloc, PyTorch-ROCm, ...). The library never retains them.
 *   - calls are asynchronous on the context's HIP stream; vbnn_sync / vbnn_buf_download
 *     are the only blocking calls.
 *   - matrices are dense row-major. "packed operands" are the GEMM-ready copies made by
 *     vbnn_pack: element type f32 or bf16 (dtype), leading dimension padded to a
 *     multiple of VBNN_KPAD elements with ZERO fill (allocate them zeroed). A packed bf16
 *     operand of 2^30 elements (rows x ld) or more is still computed, by the general kernel:
 *     the pipelined kernels address their operands with 32-bit byte offsets.
 *   - O = outputSize, I = inputSize, N = minibatch rows (local rows on this rank).
 */
#ifndef VBNN_HIP_H
#define VBNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden */
#endif

#define VBNN_ABI_VERSION 6
#define VBNN_KPAD 64            /* packed leading dimensions are multiples of this */

enum { VBNN_OK = 0, VBNN_ERR_INVALID = 1, VBNN_ERR_HIP = 2, VBNN_ERR_NOMEM = 3, VBNN_ERR_UNSUPPORTED = 4 };
enum { VBNN_F32 = 0, VBNN_BF16 = 1 };                       /* dtype of packed operands */
enum { VBNN_PACK_COPY = 0, VBNN_PACK_EXP = 1, VBNN_PACK_SQUARE = 2, VBNN_PACK_MUL = 3,
       VBNN_PACK_RELU = 4, VBNN_PACK_RELU_SQUARE = 5 };

typedef struct vbnn_ctx vbnn_ctx;

int vbnn_abi_version(void);
const char* vbnn_last_error(void);

/* test / A-B hook, PROCESS-wide: WHICH of the library's kernels computes a GEMM -- every setting computes the same results to
 * rounding (the parity tests force each kernel through the same checks); nothing here changes what is computed. (r04: the keys
 * that switched timing-only or never-selected forms -- fake noise, the K-split launches -- left the library with those forms:
 * tools/lab/.) Kernel selection is a property of the build under test, not of a context: set it
 * before the contexts start launching, from one thread). VBNN_DEBUG_GEMM_KERNEL: 0 = pick by shape (default), 1 = always the general MFMA
 * kernel, 2 = the pipelined bf16 kernel whenever the operands allow it, 3 = its two-pass 256 x 256 variant
 * whenever the shape and outputs allow it (whole tiles, the fused configuration). */
#define VBNN_DEBUG_GEMM_KERNEL 0
#define VBNN_DEBUG_V2_SCHEDULE 1   /* pipelined kernel schedule: -1 = by tile (default), 0 = DMAs in a burst after the barrier,
                                      2 = interleaved with the MFMAs, 4 = alternating read / MFMA clusters, the workgroup's
                                      halves one cluster apart (8-wave tiles only; the same results whichever: gemm_v2.h) */
#define VBNN_DEBUG_V2_TILE 2       /* pipelined kernel block tile: 0 = by shape (default), 256 = 256 x 128, 128 = 128 x 128,
                                      64 = 128 x 128 with a 2-stage ring, two workgroups per CU */
#define VBNN_DEBUG_V3_MIN_K 4      /* shortest K for which shape selection picks the two-pass 256 x 256 kernel (default 704) */
#define VBNN_DEBUG_V2_PSPLIT 5     /* pipelined kernel pair split (the pair's two GEMMs in different workgroups, parameter
                                      gradients only): -1 = by shape (default), 0 = never, 1 = whenever possible */
#define VBNN_DEBUG_V3_SPLIT 8      /* two-pass kernel's pair-split half-height launch for few-tile parameter gradients: -1 = by shape (default), 0 = never, 1 = whenever possible */
#define VBNN_DEBUG_V0 9            /* fp32 shapes of the launch-bound geometry: 1 = the latency kernel (gemm_v0.h; default), 0 = gemm_v1's 32 x 32 tile */
#define VBNN_DEBUG_HEAD_BACKWARD 10   /* the fused head's backward (vbnn_head_backward, bf16): -1 = by shape (default: the streaming form for whole
                                         128-unit x 32-row tiles with a separate finish kernel), 0 = the tile form, 1 = the streaming form
                                         whenever the operands allow it (also at sizes whose partial sums the tile form finishes in-launch) */
#define VBNN_DEBUG_KMAJOR 6        /* K-major operands (gemm_v3.h AK / BK): 1 = use when the shape allows (default), 0 = never, 2 = gemm_v3 only */
int vbnn_debug_set(int key, int value);

/* 1 when a GEMM with an M x N output and contraction length K would take the K-major form of the two-pass kernel
 * under the current selection (whole 256 x 256 tiles filling the CUs, K a multiple of 64, bf16): the host asks once
 * per layer whether it needs the transposed copies at all (accGradParameters: M = I, N = O, K = minibatch rows;
 * gradInput: M = I, N = minibatch rows, K = O). For accGradParameters the answer presumes the fused total-gradient form
 * with the operand shadows (vbnn_dw_args.grad_mu / grad_lv + mu_s / var_s): that is the only K-major epilogue. */
int vbnn_kmajor_supported(int64_t M, int64_t N, int64_t K);
/* The same question for accGradParameters of an I -> O layer on N rows, which has a second K-major form (the pair-split
 * launch for outputs with few tiles). bias_row = 1: the bias gradient is to come from the GEMM (vbnn_dw_args.gradBias),
 * K-major that means column I of x is all ones (and ld_x > I). Returns 0 (no: pass xT / gT), 1 (yes) or 2: yes, on the
 * two-pass kernel's pair-split + split-K launch, which reads x and x.x in whole 256-column tiles -- allocate them with
 * ld_x >= the row length (ones column included) rounded up to a multiple of 256, zero filled past the data; with a
 * smaller ld_x the call still works, on the slower launch. */
int vbnn_kmajor_supported_dw(int64_t I, int64_t O, int64_t N, int bias_row);
/* The same two questions for the launches of ONE context: a context made by vbnn_ctx_create_cu_budget tiles for its budget, every
 * other context (and the two calls above) for the whole device -- the plan is a property of the context, not of the process. */
int vbnn_ctx_kmajor_supported(vbnn_ctx* ctx, int64_t M, int64_t N, int64_t K);
int vbnn_ctx_kmajor_supported_dw(vbnn_ctx* ctx, int64_t I, int64_t O, int64_t N, int bias_row);

/* context = (device, stream). hip_stream is a hipStream_t; NULL = the device's default stream
 * (which is also what PyTorch-ROCm's default stream is, so the two stay ordered). */
int vbnn_ctx_create(int device, void* hip_stream, vbnn_ctx** out);
/* The same with a stream of the library's own that may only use `n_cus` of the device's compute units
 * (hipExtStreamCreateWithCUMask; the enabled units are spread evenly over the XCDs): the CU budget of the compute stream in
 * a data-parallel run, where RCCL's channels hold some units for the whole step -- a 256-tile GEMM launched on all 256
 * units then finishes the displaced tiles as a straggling second round, while a launch planned for the units it really
 * has (vbnn_ctx_kmajor_supported* and the shape heuristics of THIS context's calls read the budget) tiles for them. n_cus <= 0 or >= the device's
 * count: no mask, just an own stream. vbnn_ctx_stream returns the hipStream_t for hosts that enqueue their own work
 * (PyTorch: torch.cuda.ExternalStream). */
int vbnn_ctx_create_cu_budget(int device, int n_cus, vbnn_ctx** out);
int vbnn_ctx_stream(vbnn_ctx* ctx, void** hip_stream_out, int* n_cus_out);
int vbnn_ctx_destroy(vbnn_ctx* ctx);
int vbnn_ctx_set_stream(vbnn_ctx* ctx, void* hip_stream);
int vbnn_sync(vbnn_ctx* ctx);

/* device-buffer helpers for hosts without a tensor library on the device (the Lua shim;
 * replaces the cutorch :cuda() copies at VBLinear.lua:40-44,56-58 and main.lua:22-25). */
int vbnn_buf_alloc(vbnn_ctx* ctx, size_t bytes, void** dptr);   /* zero-initialised */
int vbnn_buf_free(vbnn_ctx* ctx, void* dptr);
int vbnn_buf_zero(vbnn_ctx* ctx, void* dptr, size_t bytes);
int vbnn_buf_upload(vbnn_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int vbnn_buf_download(vbnn_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);  /* blocks */

/* rows x cols standard normals (include/vbnn_philox.h contract):
 * out[r][c] = lane c&3 of vbnn_normal4(seed, stream, layer, draw, row0 + r, c >> 2), times `scale`.
 * Replaces randomkit.normal at VBLinear.lua:26-28 (means init) and mlp.lua:53 (He init);
 * also the synthetic-input generator of bench.py. */
int vbnn_fill_normal(vbnn_ctx* ctx, float* out, int64_t rows, int64_t cols, int64_t ld,
                     uint64_t seed, uint32_t stream, uint32_t layer, uint32_t draw, int64_t row0,
                     float scale);
/* The same normals in the form the bf16 forward draws them: the same Philox words, Box-Muller on them by the hardware's
 * log2 / sqrt / sin / cos instead of the contract's bit-exact polynomial forms (csrc/common.h, vbnn_normal4_hw): equal to
 * vbnn_fill_normal's to a few 1e-7 absolute, a quarter of the instructions. The fp32 path (held sample for sample against the
 * oracle) draws the exact form; this entry exists so that a host or a test can see exactly what a bf16 forward used. */
int vbnn_fill_normal_hw(vbnn_ctx* ctx, float* out, int64_t rows, int64_t cols, int64_t ld,
                        uint64_t seed, uint32_t stream, uint32_t layer, uint32_t draw, int64_t row0,
                        float scale);

/* Both forms of the contract's Box-Muller (include/vbnn_philox.h, vbnn_box_muller) on GIVEN Philox words x0[i], x1[i] (device
 * arrays of n words): z_exact[2i], z_exact[2i + 1] = the bit-exact form (what the fp32 path and the oracle draw), z_hw[..] = the
 * hardware-transcendental form the bf16 forward draws. A diagnostic for hosts and tests: it is how the hardware form's distance
 * from the contract is pinned over ALL 2^24 radii and ALL 2^24 angles (tests/test_parity_gpu.py, tests/golden/normals_hw_error.json). */
int vbnn_box_muller_forms(vbnn_ctx* ctx, const uint32_t* x0, const uint32_t* x1, float* z_exact, float* z_hw, int64_t n);

/* VBLinear:compute_prior (VBLinear.lua:77-88). One fused sweep over means/lvars:
 *   stats[0] = sum(exp(lvars) + means^2)   (so var_hat = stats[0] / W, :86)
 *   stats[1] = sum(lvars)
 *   stats[2] = var_hat (as double)  -- read by later kernels on the device, no host sync
 * Optional caches (NULL to skip), exactly the reference's: vars (:78), stdv (:79), mu_sqe (:82).
 * `stats` is a device array of 4 doubles. */
int vbnn_compute_prior(vbnn_ctx* ctx, const float* means, const float* lvars, int64_t W,
                       float* vars, float* stdv, float* mu_sqe, double* stats);

/* VBLinear:sample (VBLinear.lua:49-64): weight = means + stdv (.) e, e ~ N(0,1) of shape O x I
 * regenerated from the Philox counter (stream EPS). e_out may be NULL (the reference keeps
 * self.e; the kernels never need it stored). stdv == NULL: use exp(lvars/2) of `lvars`. */
int vbnn_wn_sample(vbnn_ctx* ctx, const float* means, const float* stdv, const float* lvars,
                   float* weight, float* e_out, int64_t O, int64_t I,
                   uint64_t seed, uint32_t layer, uint32_t draw);

/* Generic packer: dst[r][c] = T(f(src[r][c])) for r < rows, c < cols, with
 *   f = COPY | EXP | SQUARE | MUL (src * src2) | RELU | RELU_SQUARE;
 * dst is rows x ld_dst, dstT (optional) is the transpose, cols x ld_dstT. Pads are not
 * written (keep them zero). Produces every GEMM operand of the module-level path:
 * mu/sigma^2 shadows, x and x.x, g and g.r. */
int vbnn_pack(vbnn_ctx* ctx, int dtype, int func, const float* src, const float* src2, int64_t ld_src,
              int64_t rows, int64_t cols, void* dst, int64_t ld_dst, void* dstT, int64_t ld_dstT);

/* ---- the three GEMM families of the layer (MFMA kernels, fused epilogues) ------------------- */

typedef struct vbnn_fwd_args {
    /* operands (packed, dtype): A = weights-side O x ld_w, B = input-side N x ld_x */
    const void* w;      /* mu shadow (LRT) or sampled-weight shadow (WN / MAP)            */
    const void* w2;     /* sigma^2 shadow, LRT only (NULL: single GEMM)                    */
    const void* x;      /* input                                                           */
    const void* x2;     /* input squared, LRT only. dtype F32: may be NULL -- the kernel forms x.x in  */
                        /* registers while staging x (bit for bit what the packer would store)  */
    int64_t ld_w, ld_x;
    int64_t N, I, O;
    const float* bias;  /* O, may be NULL                                                  */
    /* LRT noise: z[n][o] from (seed, ZETA, layer, draw, row0 + n, o >> 2); ignored if w2 == NULL */
    uint64_t seed; uint32_t layer; uint32_t draw; int64_t row0;
    /* outputs, each optional */
    float* y;  int64_t ld_y;        /* pre-activation output (the module's `output`), N x O     */
    void* r;  int64_t ld_r;         /* z / (2 sqrt(v)), saved for backward (LRT), N x ld_r:      */
    int r_packed;                   /* 0: float (the module path); 1: element type = dtype       */
    int relu;                       /* apply ReLU before writing the packed outputs below        */
    void* h;  void* h2;  int64_t ld_h;     /* next layer's packed input and its square, N x ld_h  */
    void* hT; void* h2T; int64_t ld_hT;    /* their transposes, O x ld_hT                         */
    /* > 0: the N rows are SEVERAL Monte-Carlo draws of one minibatch stacked (main.lua:32-37's S loop as rows): row n is
     * minibatch row n % rows_per_draw of draw `draw + n / rows_per_draw`, and its noise is addressed accordingly -- bit
     * for bit the z of that draw's own launch. The backward GEMMs then sum over all draws in one pass (K = N). 0: off. */
    int64_t rows_per_draw;
    /* optional DEVICE-RESIDENT draw counter (NULL: off): the noise is addressed by draw + *draw_dev, read when the kernel
     * runs -- so that a step whose launches were captured into a graph (vbnn_capture_*) draws fresh noise at every replay:
     * the host's `sample()` is then vbnn_sample (a device-side increment, itself a node of the graph) and `draw` stays a
     * constant base. General kernel only: with it set the pipelined bf16 kernels are not selected (the launch-bound
     * configurations are the ones worth capturing; the others take the counter as the launch argument above). */
    const uint32_t* draw_dev;
    /* optional (NULL: off; ABI 5): the logits of the classifier head that follows the LAST VB layer (mlp.lua:29: the final
     * nn.Linear, C <= 16 outputs) formed by THIS launch, from its output tiles while they are still in registers: every
     * 256 x 256 tile adds nothing to memory traffic but 2 x 256 rows x 16 fp32 partial logits, written to the fixed slot
     * head_slots[2 tile + half][n][16] (n_slots = vbnn_forward_head_slots(...) slots of N x 16 floats; classes >= head_C are
     * undefined), and the head (vbnn_head_forward_slots / vbnn_head_args.logit_slots) adds the slots in order instead of
     * re-reading h: bitwise reproducible, no atomics. head_w3: the final weight PACKED (head_C x head_ld_w, dtype), as
     * vbnn_prepare / vbnn_update leave it. Only where vbnn_forward_head_slots says so (> 0); elsewhere the call fails. */
    const void* head_w3; int64_t head_ld_w; int64_t head_C; float* head_slots;
} vbnn_fwd_args;

/* updateOutput. WN/MAP: y = x w^T + b  (inherited nn.Linear:updateOutput, VBLinear.lua:7).
 * LRT: m = x mu^T + b, v = (x.x)(sigma^2)^T, y = m + sqrt(v) . z   (north_star).
 * dtype F32 (the general kernel): `x` may be the RAW minibatch -- any 16-byte-aligned row-major matrix with ld_x >= I,
 * ld_x % 4 == 0, columns [I, ld_x) finite; the K walk is masked at the row end, no zero padding is needed -- so an fp32
 * host needs no vbnn_pack_input at all. */
int vbnn_forward(vbnn_ctx* ctx, int dtype, const vbnn_fwd_args* a);
/* How many slots the forward of an I -> O layer on N rows writes when given vbnn_fwd_args.head_slots for a C-class head, under
 * this context's kernel selection: 0 = that launch cannot carry the head's logits (use vbnn_head_forward on h), otherwise
 * allocate n x N x 16 floats. (bf16 layers whose forward runs on the two-pass 256 x 256 kernel: 2 slots per 256 output units.) */
int vbnn_forward_head_slots(vbnn_ctx* ctx, int dtype, int64_t N, int64_t I, int64_t O, int64_t C);

typedef struct vbnn_dx_args {
    /* A = transposed weights-side I x ld_wT, B = gradient-side N x ld_g */
    const void* wT;     /* mu^T shadow (LRT) or sampled-weight^T shadow (WN)               */
    const void* w2T;    /* (sigma^2)^T shadow, LRT only                                     */
    const void* g;      /* dL/dy packed                                                     */
    const void* gv;     /* dL/dv = g . r packed, LRT only                                   */
    int64_t ld_wT, ld_g;
    int64_t N, I, O;
    const void* x; int64_t ld_x;    /* this layer's packed input (LRT term 2 x . (gv sigma^2);  */
                                    /* also the ReLU mask of the previous module)               */
    float* gx; int64_t ld_gx;       /* gradInput N x I f32, optional                            */
    /* optional fused hand-off to the previous VB layer: g_prev = gx . [x > 0], gv_prev = g_prev . r_prev */
    int relu_mask;
    const void* r_prev; int64_t ld_r_prev; int r_prev_packed;   /* the previous layer's `r`, same typing rule */
    void* g_prev; void* gv_prev; int64_t ld_gp;      /* N x ld_gp   */
    void* gT_prev; void* gvT_prev; int64_t ld_gpT;   /* I x ld_gpT  */
    /* optional K-MAJOR weights: mu, sigma^2 as the forward holds them (O x ld_w). When vbnn_kmajor_supported(I, N, O)
     * says so the GEMM reads these and wT / w2T may be NULL: the parameter sweep writes no transposed shadows.
     * dtype F32: always (any shape) -- with wT == NULL the general kernel reads w / w2 K-major. */
    const void* w; const void* w2; int64_t ld_w;
} vbnn_dx_args;

/* updateGradInput. WN: gradInput = g w (inherited, VBLinear.lua:109-110).
 * LRT: gradInput = g mu + 2 x . (gv sigma^2). */
int vbnn_grad_input(vbnn_ctx* ctx, int dtype, const vbnn_dx_args* a);

typedef struct vbnn_dw_args {
    /* A = input-side transposed I x ld_n, B = gradient-side transposed O x ld_n (K = N rows) */
    const void* xT;     /* x^T                                                              */
    const void* x2T;    /* (x.x)^T, LRT only                                                */
    const void* gT;     /* g^T                                                              */
    const void* gvT;    /* gv^T, LRT only                                                   */
    int64_t ld_n;
    int64_t N, I, O;
    float scale;        /* accGradParameters' scale (applied to gradWeight only, as the reference) */
    int accumulate;     /* 1: += (Torch semantics); 0: overwrite (first draw after resetAcc)  */
    float* gradWeight;  /* O x I, += scale * g^T x                      (VBLinear.lua:113)     */
    float* gradSum;     /* O x I. WN:  += (g^T x) . e                    (VBLinear.lua:114-115) */
                        /*        LRT: += 2 (gv^T x.x) . stdv   (same units)    */
    /* WN: e regenerated from (seed, EPS, layer, draw). LRT: stdv = exp(lvars / 2). */
    uint64_t seed; uint32_t layer; uint32_t draw;
    const float* lvars;
    /* optional fused total-gradient outputs (likelihood/S + kl_scale * KL gradient),
     * VBLinear.lua:90-98 folded into the epilogue; NULL to skip. Needs means, lvars, stats. */
    float* grad_mu; float* grad_lv;
    const float* means; const double* stats; float B; float S; float kl_scale;
    /* optional: the bias gradient from the same GEMM. xT (and x2T) then have I + 1 rows, row I of xT all ones (row I
     * of x2T anything finite): output row I of the first GEMM is sum_n g[n][o], written as
     * gradBias[o] (+)= scale * that -- what vbnn_acc_grad_bias(g) gives, without another pass over g. NULL: off. */
    float* gradBias;
    /* optional K-MAJOR operands: the untransposed x, x.x (N x ld_x) and g, gv (N x ld_g) exactly as the forward and
     * gradInput GEMMs hold them. When vbnn_kmajor_supported(I, O, N) says so the GEMM reads these (transpose reads in
     * LDS) and xT / x2T / gT / gvT may be NULL: no epilogue has to write transposed copies. Otherwise xT.. are used.
     * dtype F32: always (any shape, part = 0) -- with xT == NULL the general kernel reads x and g (gv) K-major; x2 may
     * then be NULL (x.x is formed in registers), x may be the raw minibatch (ld_x % 4 == 0), and gradBias needs no row or
     * column of ones anywhere: the kernel synthesises it. */
    const void* x; const void* x2; const void* g; const void* gv; int64_t ld_x; int64_t ld_g;
    /* optional (fused total gradients only, dtype BF16): the packed operand shadows mu_s, var_s = exp(lvars) (O x ld_w,
     * as vbnn_prepare / vbnn_update leave them). When given, the epilogue reads mu and sigma^2 FROM THEM -- 4 B per weight
     * instead of 8, and no exp -- i.e. d/dlvars = (gv^T x.x) . s2 + KL'(s2), d/dmeans = g^T x + KL'(mu) with s2, mu the
     * values the forward GEMMs multiplied by (bf16-rounded); means / lvars are then not read. NULL: fp32 means / lvars.
     * REQUIRED by the K-major launches (x / g given, xT / gT NULL): their epilogue reads the shadows only, so a caller that
     * passes K-major operands without mu_s / var_s gets VBNN_ERR_INVALID -- keep the transposed operands for that case. */
    const void* mu_s; const void* var_s; int64_t ld_w;
    /* 0 (default): both GEMMs of the LRT pair, every output. 1: only the first GEMM (g^T x) and what depends on it
     * (gradWeight / grad_mu, gradBias); 2: only the second (gv^T x.x) and what depends on it (gradSum / grad_lv). Calling
     * with part = 2 and then part = 1 gives bit for bit the outputs of part = 0 (each output depends on ONE accumulator)
     * as two launches -- so that a data-parallel host can start the exchange of the finished d/dlvars while the d/dmeans
     * GEMM still runs (vbnn_allreduce_grads between the two calls). LRT pairs only. */
    int part;
    const uint32_t* draw_dev;   /* as vbnn_fwd_args.draw_dev (the weight-noise form regenerates e from the counter) */
} vbnn_dw_args;

/* accGradParameters (VBLinear.lua:112-118), one GEMM instead of the reference's two. */
int vbnn_acc_grad_parameters(vbnn_ctx* ctx, int dtype, const vbnn_dw_args* a);

/* accGradParameters AND updateGradInput of one layer in one call: both consume that layer's gradOutput and neither reads
 * what the other writes (nn.Sequential:backward runs them back to back, mlp.lua:79). Where one launch can carry both -- dtype
 * F32 with the K-major operand forms, at the launch-bound tile geometry (BASELINE configs[1]: a launch there is 10 us of
 * latency chain with the chip a fraction busy) -- it does, each tile computed exactly as by its own launch (bitwise the same
 * outputs); everywhere else this IS the two calls, accGradParameters first. */
int vbnn_backward_pair(vbnn_ctx* ctx, int dtype, const vbnn_dx_args* dx, const vbnn_dw_args* dw);

/* gradBias += scale * sum_n g[n][o]  (the parent call at VBLinear.lua:113). g is N x ld_g of
 * `dtype` (F32: the module's gradOutput; BF16: a packed operand of the fused path). */
int vbnn_acc_grad_bias(vbnn_ctx* ctx, int dtype, const void* g, int64_t ld_g, int64_t N, int64_t O,
                       float scale, int accumulate, float* gradBias);

/* Per-step parameter sweep of the fused path: one read of means/lvars writes the packed GEMM
 * shadows mu_s, var_s = exp(lvars) (O x ld_w) and, if asked, their transposes (I x ld_wT), and the
 * statistics of VBLinear:compute_prior (VBLinear.lua:77-88) into stats[0..3] as vbnn_compute_prior. */
int vbnn_prep_layer(vbnn_ctx* ctx, int dtype, const float* means, const float* lvars, int64_t O, int64_t I,
                    void* mu_s, void* var_s, int64_t ld_w, void* muT_s, void* varT_s, int64_t ld_wT,
                    double* stats);

/* The same sweep for every VB layer of a model in ONE call (n_layers <= 8), plus optionally the plain packing of one
 * more small f32 matrix (the final nn.Linear's weight, mlp.lua:29: dst rows x ld_dst and its transpose cols x
 * ld_dstT, either may be NULL): one sweep kernel per layer and a single finish kernel for all the statistics and the
 * extra matrix -- two launches fewer per step than n_layers x vbnn_prep_layer + vbnn_pack. */
typedef struct vbnn_prep_desc {
    const float* means; const float* lvars; int64_t O, I;
    void* mu_s; void* var_s; int64_t ld_w;
    void* muT_s; void* varT_s; int64_t ld_wT;      /* NULL / 0: no transposed shadows */
    double* stats;
} vbnn_prep_desc;
typedef struct vbnn_pack_desc {
    const float* src; int64_t rows, cols, ld_src;
    void* dst; int64_t ld_dst; void* dstT; int64_t ld_dstT;
} vbnn_pack_desc;
int vbnn_prepare(vbnn_ctx* ctx, int dtype, int n_layers, const vbnn_prep_desc* layers, const vbnn_pack_desc* extra);

/* ---- KL ("LC") terms ------------------------------------------------------------------------ */

/* VBLinear:compute_mugrads (VBLinear.lua:90-93): gradWeight /= S in place; lcg = means / (B var_hat).
 * VBLinear:compute_vargrads (:95-98): gradSum = gradSum / (2S) . stdv in place;
 *                                      lcg = ((-1/vars + 1/var_hat) / (2B)) . vars.
 * var_hat is read from stats[2] on the device. vars/stdv NULL: derived from lvars. */
int vbnn_compute_mugrads(vbnn_ctx* ctx, const float* means, const double* stats, float B, float S,
                         float* gradWeight, float* lcg, int64_t W);
int vbnn_compute_vargrads(vbnn_ctx* ctx, const float* lvars, const float* vars, const float* stdv,
                          const double* stats, float B, float S, float* gradSum, float* lcg, int64_t W);

/* VBLinear:calc_lc (VBLinear.lua:99-103) with the :sum() of mlp.lua:112 fused:
 * lc_sum_dev[0] = sum_w [log sqrt(var_hat) - log sqrt(vars) + (mu_sqe + vars - var_hat) / (2 var_hat)] / B.
 * lc_elem (optional) receives the per-weight tensor the Lua method returns.
 * vars / mu_sqe: the caches of the last compute_prior (the reference reads exactly those, so its
 * LC is one optimiser step stale, main.lua:174-177); NULL: derive from means / lvars. */
int vbnn_calc_lc(vbnn_ctx* ctx, const float* means, const float* lvars, const float* vars, const float* mu_sqe,
                 const double* stats, float B, float* lc_elem, double* lc_sum_dev, int64_t W);

/* The minibatch (N x I f32, row pitch ld_src) as GEMM operands in one pass: x_s, x2_s = (rounded x)^2 (N x ld_x)
 * and their transposes (I x ld_xT); x2_s / xT_s / x2T_s optional. Replaces the host->device copies of
 * main.lua:22-25 plus two vbnn_pack calls. rows_per_draw > 0 (vbnn_fwd_args.rows_per_draw): the N operand rows are several
 * Monte-Carlo draws of ONE minibatch of rows_per_draw rows -- operand row n is src row n % rows_per_draw, so the host never
 * materialises the S-fold input. 0: off. */
int vbnn_pack_input(vbnn_ctx* ctx, int dtype, const float* src, int64_t ld_src, int64_t N, int64_t I, void* x_s,
                    void* x2_s, int64_t ld_x, void* xT_s, void* x2T_s, int64_t ld_xT, int64_t rows_per_draw);

/* ---- the update that follows the hot path -------------------------------- */

/* optim.adam as VBLinear:update calls it on means / lvars (VBLinear.lua:135-143), one streaming pass:
 *   g = grad (+ grad2);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g.g;
 *   x -= lr sqrt(1 - b2^t) / (1 - b1^t) . m / (sqrt(v) + eps)
 * grad2 (optional) is added to grad first (the torch.add of likelihood and KL parts, VBLinear.lua:131-134);
 * lambda < 1 selects the early torch/optim variant's decay b1_t = b1 lambda^(t-1) (config.lua:52,56,61), 1 = off;
 * t is state.t after its increment (>= 1). norms_dev (optional, 2 doubles): { |update|, |x_new| }, the two norms of
 * the reference's `mu_normratio = torch.norm(update) / torch.norm(x)` (VBLinear.lua:139,144).
 * optim is not vendored by the reference and its version is unpinned: this is the published algorithm. */
int vbnn_adam_step(vbnn_ctx* ctx, float* x, const float* grad, const float* grad2, float* m, float* v, int64_t n,
                   float lr, float beta1, float beta2, float eps, float lambda, int64_t t, double* norms_dev);
/* optim.sgd with only learningRate set (config.lua:51-54): x -= lr * grad (VBLinear.lua:125-128, mlp.lua:120-128). */
int vbnn_sgd_step(vbnn_ctx* ctx, float* x, const float* grad, int64_t n, float lr);

/* VBLinear:update (VBLinear.lua:124-166) for every VB layer of a model in ONE call, fused with what the NEXT minibatch
 * needs: per layer one sweep applies optim.sgd to the bias (:125-128) and optim.adam to means and lvars (:135-143, the
 * arithmetic of vbnn_adam_step) from the TOTAL gradients (likelihood / S + KL: what the fused accGradParameters
 * epilogue leaves in grad_mu / grad_lv, i.e. `mugrad` / `vgrad` of :132,134) and, from the NEW parameters in the same
 * pass, writes the packed GEMM shadows and the prior statistics exactly as vbnn_prepare would (the reference runs
 * compute_prior inside update as well, :130) -- so a training step needs no separate parameter sweep. `extra` as in
 * vbnn_prepare (the final nn.Linear's weight, packed after its own vbnn_sgd_step). One finish kernel for all layers.
 * log14 (optional, 14 doubles per layer) receives the series VBLinear.lua:149-164 logs, in that order:
 *   vlc grad, vle grad, mlc grad, mle grad (norm of the KL / likelihood part over norm of the updated parameter; the
 *   KL parts are re-derived from the pre-update parameters and stats[2], the likelihood parts are total - KL),
 *   min / max / mean variance (updated), var hat (pre-update, as self.var_hat is at :156), mean / std (unbiased) / min /
 *   max of the updated means, mu normratio, var normratio (:139,144).
 * stats must hold the statistics of the PRE-update parameters on entry (vbnn_prepare / the previous vbnn_update). */
typedef struct vbnn_adam_cfg { float lr, beta1, beta2, eps, lambda; int64_t t; } vbnn_adam_cfg;
typedef struct vbnn_update_desc {
    float* means; float* lvars; int64_t O, I;          /* updated in place */
    void* mu_s; void* var_s; int64_t ld_w;             /* shadows of the NEW parameters (as vbnn_prep_desc) */
    void* muT_s; void* varT_s; int64_t ld_wT;          /* NULL / 0: no transposed shadows */
    double* stats;                                     /* in: pre-update statistics; out: statistics of the new parameters */
    const float* grad_mu; const float* grad_lv;        /* total gradients, O x I */
    float* m_mu; float* v_mu; float* m_lv; float* v_lv;    /* Adam state, O x I each */
    vbnn_adam_cfg mu, lv;                              /* opt.meanState / opt.varState (config.lua:55-64) */
    float* bias; const float* grad_bias; float lr_bias;    /* optional (NULL): optim.sgd on the bias */
    float B;                                           /* opt.B: scale of the KL parts in the logged norms */
    double* log14;                                     /* optional */
    /* 0 (default): grad_mu / grad_lv are the TOTAL gradients (the accGradParameters epilogue added the KL part, from the
     * bf16 operand shadows in the fused bf16 configuration). > 0: they are the likelihood parts alone (the host passed
     * vbnn_dw_args.kl_scale = 0) and the sweep adds kl_add x the KL gradient itself, from the fp32 means / lvars and stats[2]
     * (VBLinear.lua:91,96 exactly as the reference computes them) -- the exact form: the shadow form's (bf16(s2) / var_hat - 1)
     * cancels to ~2^-9 absolute. Data-parallel: 1 on every rank (the all-reduced likelihood sum + the KL gradient once). */
    float kl_add;
} vbnn_update_desc;
int vbnn_update(vbnn_ctx* ctx, int dtype, int n_layers, const vbnn_update_desc* layers, const vbnn_pack_desc* extra);

/* ---- data-parallel exchange (north_star: "RCCL all-reduce over xGMI on the (mu, log sigma^2) gradients after
 * accGradParameters"; the reference itself is single-device, main.lua:142 sets BLAS threads only) ------------------
 * One process per GPU, one communicator per process. Rank 0 calls vbnn_comm_unique_id and the HOST side hands the
 * VBNN_COMM_ID_BYTES to every rank by its own means (a file, a socket, torch.distributed's store, MPI ...); every
 * rank then calls vbnn_comm_create (collective: it returns when all `world` ranks have joined).
 * vbnn_allreduce_grads(buf, n): in-place SUM over ranks of n floats -- a layer's contiguous gradient bucket
 * [d/dmeans | d/dlvars | d/dbias]. It is ordered after everything enqueued so far on the context's stream (the
 * accGradParameters that fills the bucket) and runs on the communicator's own high-priority stream, i.e. beside
 * the launches that follow (the rest of backward). Every rank must issue the same sequence of calls.
 * vbnn_comm_finish: orders the context's stream behind every exchange issued so far (no host wait; follow with
 * vbnn_sync to block). The criterion already divides by the GLOBAL batch and the fused KL gradient carries
 * 1 / world (vbnn_dw_args.kl_scale), so the sum IS the gradient: no division afterwards.
 * librccl is loaded on first use (dlopen: VBNN_RCCL_PATH, then librccl.so.1); without it these calls return
 * VBNN_ERR_UNSUPPORTED and nothing else in the library is affected. (r05: the ordering behind the context's stream is a TRIGGER word --
 * a one-thread kernel on the context's stream, a one-wave kernel polling it on the exchange stream in front of the collective -- not
 * an event: a marker packet costs the context's stream ~8 us of bubble per bucket; vbnn_comm_finish hands BACK the same way, the roles
 * swapped, as vbnn_p2p_finish does; VBNN_COMM_FLAG_TRIGGER=0 / VBNN_P2P_FLAG_TRIGGER=0 at create keep the events.) */
#define VBNN_COMM_ID_BYTES 128
typedef struct vbnn_comm vbnn_comm;
int vbnn_comm_unique_id(void* id_out /* VBNN_COMM_ID_BYTES, host memory */);
int vbnn_comm_create(vbnn_ctx* ctx, int rank, int world, const void* id, vbnn_comm** out);
int vbnn_comm_destroy(vbnn_comm* comm);
int vbnn_comm_info(vbnn_comm* comm, int* rank, int* world, int* ranks_in_comm /* as RCCL counts them */);
int vbnn_allreduce_grads(vbnn_comm* comm, float* buf, int64_t n);
int vbnn_comm_finish(vbnn_comm* comm);
/* OPTIONAL half-size exchange (160.1 MB fp32 / 80.1 MB bf16 per step of the wide configuration): the
 * same call on a bf16 copy of the bucket, with vbnn_cast_grads to make it and to widen the sum back. The default exchange
 * is fp32; this one rounds every rank's contribution to bf16 (RNE) and lets RCCL sum in bf16 -- a different gradient
 * (relative 2^-9 per element and hop), for hosts that want the exchange hidden more than the last bits.
 * vbnn_cast_grads(to_bf16 = 1): dst[i] = bf16(src[i]), src fp32; (0): dst[i] = float(src[i]), src bf16; on the context's stream. */
int vbnn_allreduce_grads_bf16(vbnn_comm* comm, void* buf_bf16, int64_t n);
int vbnn_cast_grads(vbnn_ctx* ctx, int to_bf16, const void* src, void* dst, int64_t n);
/* all_dev[r] = rank r's *mine_dev (device memory, 8 bytes per rank), gathered over the same communicator and stream:
 * lets the host verify that `world` distinct processes / devices take part in the exchange. */
int vbnn_comm_allgather_u64(vbnn_comm* comm, const uint64_t* mine_dev, uint64_t* all_dev);
/* The two halves of the all-reduce as calls of their own, for the SHARDED-UPDATE exchange (an option beside north_star's
 * all-reduce; engine.py: opt.exchange_mode = "sharded"): after accGradParameters the gradient regions are
 * reduce-SCATTERED (rank r ends with the fp32 sums of its 1 / world of every layer's rows), rank r runs vbnn_update on that slice
 * alone (Adam state and fp32 master parameters sharded), and what is all-GATHERED afterwards is the packed operand shadows the
 * next forward reads -- bf16 mu and sigma^2, 4 bytes per weight instead of the 8 bytes of fp32 gradients an all-reduce's second
 * half moves -- plus four doubles of prior statistics per layer: 0.75x the bytes, 1 / world of the update sweep, fp32 sums.
 * Both are in place, ordered behind the context's stream, run on the exchange stream (vbnn_comm_finish orders the context's
 * stream behind them). reduce_scatter: buf holds world x n_per_rank floats; afterwards floats [rank n_per_rank, (rank + 1)
 * n_per_rank) are the sums over ranks (the rest is undefined). all_gather: buf holds world x bytes_per_rank bytes, rank r
 * contributes bytes [r bytes_per_rank, (r + 1) bytes_per_rank). */
int vbnn_comm_reduce_scatter(vbnn_comm* comm, float* buf, int64_t n_per_rank);
int vbnn_comm_all_gather(vbnn_comm* comm, void* buf, int64_t bytes_per_rank);
/* Device-side glue of that exchange. vbnn_stats_combine: parts = [world][n_layers][4] doubles, rank r's block being the `stats`
 * its slice's vbnn_update left (sum(exp(lvars) + means^2), sum(lvars), -, slice weights), gathered over the ranks; writes each
 * layer's statistics of the WHOLE layer (sums in rank order; var_hat as vbnn_update forms it) into stats[l]. Every rank computes
 * the same doubles. vbnn_transpose_packed: dst[c][r] = src[r][c] for a packed operand -- layers that keep transposed shadows
 * rebuild them from the gathered mu_s / var_s. */
int vbnn_stats_combine(vbnn_ctx* ctx, int n_layers, int world, const double* parts, double* const* stats);
int vbnn_transpose_packed(vbnn_ctx* ctx, int dtype, const void* src, int64_t ld_src, int64_t rows, int64_t cols, void* dst, int64_t ld_dst);

/* ---- the same exchange WITHOUT a collective library: direct reduce-scatter + all-gather over peer-mapped arenas
 * (vbnn_amd/csrc/p2p.hip; SURVEY.md section 5's fallback should RCCL put the 160 MB all-reduce on a single ring: every GPU of an
 * 8-GPU node has seven point-to-point xGMI links, and this form moves an eighth of the bucket over each of them at once).
 * vbnn_p2p_create allocates THE gradient arena of this rank (arena_floats fp32, zeroed: the host keeps its gradients there, as
 * engine.py's flat arena) and returns its device pointer and this rank's VBNN_P2P_HANDLE_BYTES handle; the host passes the
 * handles round by its own means (as the RCCL unique id), every rank calls vbnn_p2p_connect with all of them in rank order
 * (world x VBNN_P2P_HANDLE_BYTES), and from then on vbnn_p2p_allreduce(offset, n) sums arena[offset, offset + n) over the ranks
 * in place: ordered behind the context's stream, run on a high-priority stream of its own, the sum formed in rank order (bitwise
 * the same arena on every rank). vbnn_p2p_finish orders the context's stream behind it -- and launches the ONE exit barrier of the
 * exchanges issued since the last finish ("every rank has finished gathering": only then may the arena be overwritten), so a host
 * calls it before it lets anything write the arena again. Every rank issues the same sequence. (r05: a bucket is handed from the
 * context's stream to the exchange stream through a TRIGGER word of the rank's flag page -- a one-thread kernel behind the launch
 * that fills the bucket, polled by the exchange's first barrier -- not through an event: a marker packet costs the compute stream
 * ~8 us of bubble per bucket and the exchange stream ~12 us to wake up; VBNN_P2P_FLAG_TRIGGER=0 at create keeps the events.)
 * A barrier whose peers never arrive gives up after a bounded WALL time -- 20 s by default, VBNN_P2P_TIMEOUT_S in the environment
 * at create, or vbnn_p2p_set_timeout -- instead of hanging the device, and raises the exchange's status word: from then on every
 * data kernel of the exchange is a no-op (the arena keeps this rank's OWN gradients; no partial sum is ever written over them)
 * and later barriers signal without polling; the rank also marks itself dead in every PEER's flag page, so each peer's next barrier
 * fails at once and its data kernels stop too -- within one barrier no rank believes a sum that a stopped rank took no part in.
 * vbnn_p2p_status (blocking on the exchange stream) reports the epoch of the barrier
 * that failed: the host checks it before it lets an update read the arena (engine.py: FusedMLP.update / check_exchange) and,
 * having re-synchronised the ranks by its own means, may re-arm the exchange with vbnn_p2p_clear_status. The host runs one
 * barrier of its own between vbnn_p2p_connect and the first exchange (comm.P2PExchange). At most 8 ranks (one node); peers on
 * other devices need peer access (xGMI / PCIe P2P). */
#define VBNN_P2P_HANDLE_BYTES 128
typedef struct vbnn_p2p vbnn_p2p;
int vbnn_p2p_create(vbnn_ctx* ctx, int rank, int world, size_t arena_floats, vbnn_p2p** out, void** arena_out, void* handle_out);
int vbnn_p2p_connect(vbnn_p2p* p, const void* all_handles);
int vbnn_p2p_allreduce(vbnn_p2p* p, size_t offset_floats, int64_t n);
int vbnn_p2p_finish(vbnn_p2p* p);
/* the same two halves over the peer-mapped arena (regions of the arena, in floats; equal chunks, rank r's at offset + r x
 * n_per_rank): whatever the sharded-update exchange gathers -- the operand shadows, the statistics -- lives IN the arena then
 * (the host allocates it large enough and places them there) */
int vbnn_p2p_reduce_scatter(vbnn_p2p* p, size_t offset_floats, int64_t n_per_rank);
int vbnn_p2p_all_gather(vbnn_p2p* p, size_t offset_floats, int64_t n_per_rank);
int vbnn_p2p_status(vbnn_p2p* p, int* rank, int* world, unsigned* gave_up);
int vbnn_p2p_set_timeout(vbnn_p2p* p, double seconds);
int vbnn_p2p_clear_status(vbnn_p2p* p);
int vbnn_p2p_destroy(vbnn_p2p* p);
/* (ABI 6) The data kernels are built to CO-RESIDE with the GEMM launches they overlap -- no LDS, at most 48 VGPRs (what two
 * 226-register GEMM waves leave of a SIMD's file), `world` loads in flight per lane -- so their grids are small:
 * vbnn_p2p_set_grid(reduce-scatter workgroups, all-gather workgroups PER PEER; 0 = keep; defaults 256 and 32, or
 * VBNN_P2P_RS_BLOCKS / VBNN_P2P_AG_BLOCKS at create). vbnn_p2p_standin (LAB, world == 1 only): vbnn_p2p_allreduce then runs what one
 * rank of a sim_world-rank exchange runs -- same barriers, kernels, grids, stream, event -- against this arena standing in for its
 * peers', each phase paced to the wall time `inbound_GBps` of link bandwidth would need (0: unpaced): what the exchange costs the
 * launches beside it, priced on ONE GPU (tools/overlap_standin.py). The arena holds nothing meaningful afterwards. 0 switches off. */
int vbnn_p2p_set_grid(vbnn_p2p* p, int rs_blocks, int ag_blocks_per_peer);
int vbnn_p2p_standin(vbnn_p2p* p, int sim_world, double inbound_GBps);

/* ---- (ABI 6) what THIS device holds, measured in the run that reports a throughput (bench.py's `box` block) ------------------
 * The boxes of a pool differ by 5-7 % on one binary (the clock a power-bound chip holds); the reference's only timing hook is a
 * commented-out sys.clock() (main.lua:20). vbnn_box_calibrate runs two fixed probes on the context's stream and BLOCKS until
 * they are done (~40 ms; 1 GiB of scratch, freed again): a register-only MFMA loop on every SIMD (2 waves x 2^15
 * v_mfma_f32_16x16x32_bf16, live operands) -- mfma_clock_ghz = median over workgroups of d(s_memtime) / d(s_memrealtime) x 100 MHz,
 * mfma_tflops = its flops / its time -- and a 512 MiB -> 512 MiB copy: hbm_TBps (bytes read + written per second). */
typedef struct vbnn_box_info {
    double mfma_clock_ghz, mfma_tflops, mfma_ms;
    double hbm_TBps, hbm_ms;
    int64_t hbm_bytes;
    int cus, reserved;
} vbnn_box_info;
int vbnn_box_calibrate(vbnn_ctx* ctx, vbnn_box_info* out);

/* ---- a whole step as ONE graph launch (launch-bound configurations: BASELINE configs[1], the reference's own batch-1
 * S = 30 operating point, config.lua:11,32) ----------------------------------------------------------------------------
 * mlp:sample() (mlp.lua:69-74) on the device: *draw_dev += by, ordered on the context's stream like any launch. With
 * vbnn_fwd_args.draw_dev pointing at the same word, a captured step is replayable: every replay advances the counter
 * and draws the noise of ITS draw, exactly what the same calls issued one by one compute (tested bitwise). */
int vbnn_sample(vbnn_ctx* ctx, uint32_t* draw_dev, uint32_t by);
/* vbnn_capture_begin: from here on the context's launches are RECORDED, not run (hipStreamBeginCapture on its stream --
 * which must not be the NULL stream). Issue one step's calls as usual (every buffer already allocated, every kernel
 * launched at least once before: allocation and first-launch configuration cannot be captured), then vbnn_capture_end
 * returns the executable graph. vbnn_graph_launch replays it on that stream (asynchronous, like a launch); the arguments
 * -- pointers, sizes, the accumulate flags -- are the recorded ones, the data behind the pointers is read at replay time.
 * vbnn_graph_info: kernel nodes / all nodes of the graph. */
typedef struct vbnn_graph vbnn_graph;
int vbnn_capture_begin(vbnn_ctx* ctx);
int vbnn_capture_end(vbnn_ctx* ctx, vbnn_graph** out);
int vbnn_graph_launch(vbnn_graph* g);
int vbnn_graph_info(vbnn_graph* g, int* kernel_nodes, int* nodes);
int vbnn_graph_destroy(vbnn_graph* g);

/* ---- glue modules on the measured path (mlp.lua:12-32) --------------------------------------- */
int vbnn_relu_forward(vbnn_ctx* ctx, const float* x, float* y, int64_t n);
int vbnn_relu_backward(vbnn_ctx* ctx, const float* x, const float* g, float* gx, int64_t n);
/* nn.LogSoftMax + nn.ClassNLLCriterion (sizeAverage) fused, forward and backward in one pass:
 *   out = logsoftmax(logits); loss_sum_dev[0] += -sum_n out[n][t_n] * inv_n;
 *   correct_dev[0] += #rows whose arg-max equals the target (utils.lua:11-27);
 *   g_logits = (exp(out) - onehot) * inv_n     (= LogSoftMax:backward(ClassNLL:backward)).
 * inv_n = 1 / (global minibatch rows). Targets are 0-based int32. */
int vbnn_logsoftmax_nll(vbnn_ctx* ctx, const float* logits, int64_t ld, const int32_t* target,
                        int64_t N, int64_t C, float inv_n, float* out, float* g_logits,
                        double* loss_sum_dev, int32_t* correct_dev);

/* nn.MSECriterion (sizeAverage) on an N x D output, for BASELINE.json configs[4] ("synthetic 4096-dim regression"; the
 * reference itself only ever uses ClassNLL, mlp.lua:32): criterion:forward + :backward in ONE pass over y and target,
 *   loss_sum_dev[0] (+)= inv_nd * sum (y - t)^2;   g[n][d] = 2 inv_nd (y - t)   (g optional),
 * inv_nd = 1 / (GLOBAL rows x D). The sum is formed in a fixed order (block partials + one finish block: bitwise
 * reproducible); accumulate = 0 stores it, 1 adds to it (the S draws of a minibatch). vbnn_mse_backward: g alone. */
int vbnn_mse_forward(vbnn_ctx* ctx, const float* y, int64_t ld_y, const float* target, int64_t ld_t, int64_t N, int64_t D,
                     float inv_nd, float* g, int64_t ld_g, int accumulate, double* loss_sum_dev);
int vbnn_mse_backward(vbnn_ctx* ctx, const float* y, int64_t ld_y, const float* target, int64_t ld_t, int64_t N, int64_t D,
                      float inv_nd, float* g, int64_t ld_g);

/* mlp.lua:29-32 fused for a small class count (C <= 16): final nn.Linear + nn.LogSoftMax +
 * nn.ClassNLLCriterion as streaming kernels over the packed N x H activation `h` (dtype) and the packed
 * final weight `w3` (C x ld_w, dtype). Forward: logits = h w3^T + bias, out = logsoftmax,
 * g_logits = d(loss)/d(logits) (N x C f32); `logits`, `out` optional. The loss (sum of -out[n][target] * inv_n) and
 * the hit count are summed in a fixed order inside the launch (bitwise reproducible) and stored to loss_sum_dev[0]
 * / correct_dev[0] when accumulate = 0, added to them when accumulate = 1 (the S draws of a minibatch,
 * mlp.lua:76-84): no memset of the two accumulators is needed. rows_per_draw > 0: stacked draws, row n's target is
 * target[n % rows_per_draw] (`target` holds one minibatch's rows_per_draw entries). */
int vbnn_head_forward(vbnn_ctx* ctx, int dtype, const void* h, int64_t ld_h, const void* w3, int64_t ld_w,
                      const float* bias, const int32_t* target, int64_t N, int64_t H, int64_t C, float inv_n,
                      float* logits, float* out, float* g_logits, int accumulate, double* loss_sum_dev,
                      int32_t* correct_dev, int64_t rows_per_draw);
/* The same forward from partial logits a vbnn_forward launch left in `slots` (vbnn_fwd_args.head_slots: n_slots x N x 16
 * floats): logits[n][c] = bias[c] + slots[0][n][c] + slots[1][n][c] + ... in that order, then exactly vbnn_head_forward's
 * log-softmax, loss, hit count and g_logits. h is not read. */
int vbnn_head_forward_slots(vbnn_ctx* ctx, const float* slots, int64_t n_slots, const float* bias, const int32_t* target,
                            int64_t N, int64_t C, float inv_n, float* logits, float* out, float* g_logits, int accumulate,
                            double* loss_sum_dev, int32_t* correct_dev, int64_t rows_per_draw);
/* Backward of the same head in one pass over h: gradWeight (C x H) / gradBias (C) of the final Linear (accumulate
 * as in vbnn_acc_grad_parameters; NULL to skip), its gradInput pushed through the ReLU straight into the last VB
 * layer's packed gradient operands (same meaning as the hand-off fields of vbnn_dx_args), and optionally
 * gradBias_prev (H, f32): the column sums of g_prev as stored, i.e. what vbnn_acc_grad_bias(g_prev, scale 1) would
 * give -- the bias gradient of that VB layer (VBLinear.lua:112-113, parent.accGradParameters); NULL to skip. */
int vbnn_head_backward(vbnn_ctx* ctx, int dtype, const void* h, int64_t ld_h, const void* w3, int64_t ld_w,
                       const float* g_logits, int64_t N, int64_t H, int64_t C, int accumulate, float* gradWeight,
                       float* gradBias, float* gradBias_prev, int relu_mask, const void* r_prev, int64_t ld_r_prev,
                       int r_prev_packed, void* g_prev, void* gv_prev, int64_t ld_gp, void* gT_prev, void* gvT_prev,
                       int64_t ld_gpT);

/* mlp.lua:77-83 for the fused head in ONE call: model:forward's last three modules, criterion:forward / :backward and
 * model:backward's first three -- exactly vbnn_head_forward followed by vbnn_head_backward on the same arguments (the
 * fields below carry the names and meanings of those two calls' parameters). For fp32 at launch-bound sizes
 * (N x H <= 2^20, H even, no transposed copies asked for) it is ONE launch: every workgroup recomputes the logits of its 16
 * rows (same bits as vbnn_head_forward's) instead of waiting for a first kernel's g_logits, forms its 16 x 64 tile of the
 * gradInput and its rows' terms of gradWeight / gradBias / gradBias_prev, and the last workgroup of a column block adds the
 * partials in row order (bitwise reproducible; the sums are formed in a different order from vbnn_head_backward's, so
 * the two paths agree to rounding). Everywhere else the call IS the two launches. */
typedef struct vbnn_head_args {
    const void* h; int64_t ld_h;            /* packed N x ld_h input of the final Linear */
    const void* w3; int64_t ld_w;           /* packed C x ld_w weight */
    const float* bias;                      /* C, or NULL */
    const int32_t* target;                  /* rows_per_draw (stacked draws) or N entries */
    int64_t N, H, C;
    int64_t rows_per_draw;
    float inv_n;
    int32_t accumulate;                     /* of loss / hits and of the three gradients alike (the S draws of a minibatch) */
    float* logits; float* out; float* g_logits;         /* N x C each; logits, out optional */
    double* loss_sum_dev; int32_t* correct_dev;
    float* gradWeight; float* gradBias; float* gradBias_prev;   /* C x H, C, H; NULL to skip */
    int32_t relu_mask; int32_t r_prev_packed;
    const void* r_prev; int64_t ld_r_prev;
    void* g_prev; void* gv_prev; int64_t ld_gp;
    void* gT_prev; void* gvT_prev; int64_t ld_gpT;
    /* optional (ABI 5): the forward half from partial logits left by the last VB layer's forward (vbnn_fwd_args.head_slots):
     * as vbnn_head_forward_slots, then vbnn_head_backward. NULL / 0: the logits are computed from h. */
    const float* logit_slots; int64_t n_slots;
} vbnn_head_args;
int vbnn_head_forward_backward(vbnn_ctx* ctx, int dtype, const vbnn_head_args* a);

/* The same criterion as separate modules, for the module-level call order of mlp.lua:77-80:
 * nn.LogSoftMax:updateOutput is vbnn_logsoftmax_nll with g_logits = loss = correct = NULL. */
int vbnn_nll_forward(vbnn_ctx* ctx, const float* out, int64_t ld, const int32_t* target, int64_t N, int64_t C,
                     float inv_n, double* loss_sum_dev, int32_t* correct_dev);          /* criterion:forward  */
int vbnn_nll_backward(vbnn_ctx* ctx, const int32_t* target, int64_t N, int64_t C, float inv_n, float* g); /* :backward */
int vbnn_logsoftmax_backward(vbnn_ctx* ctx, const float* out, const float* g, float* gx, int64_t N, int64_t C);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* VBNN_HIP_H */
