-- lua/FusedMLP.lua -- the DEVICE-RESIDENT form of mlp.lua's step for MI355X: what replaces `mlp:run` when the
-- minibatch is to stay in HBM from the input packer to the gradients (the module-level lua/VBLinear.lua keeps
-- mlp.lua unchanged but crosses PCIe at every module). Same protocol as the reference's mlp.lua --
--     mlp:resetGradients()  (mlp.lua:62-67)      mlp:sample()   (:69-74)      mlp:run(inputs, targets)  (:76-84)
--     mlp:calc_lc(opt)      (:109-115)           mlp:update(opt) (:117-142)
-- -- over the C ABI of include/vbnn_hip.h, call for call the sequence of vbnn_amd/engine.py:FusedMLP (run: lines
-- "forward" / "fused classifier head" / "backward"), which is what the GPU parity tests execute. LRT mode, total
-- gradients from the accGradParameters epilogue (VBLinear.lua:90-98 folded in), S draws accumulate in place.
-- Data-parallel: one process per GPU; `opt.world` / `opt.rank` / `opt.comm_id` (the 128 bytes of
-- vbnn_comm_unique_id, handed to every rank by the launcher) switch on the RCCL exchange of the gradient buckets.
--
-- NOT EXECUTED in the build image (no LuaJIT / Torch7 there). Its executable stand-in is tools/c_host.c: the same
-- functions (new / _alloc_batch / prepare / run / update), the same library calls in the same order, in plain C over the
-- same header, run on the GPU by tests/test_c_host.py and bitwise equal to engine.py there. tests/test_abi.py and
-- tests/test_c_host.py lint this file structurally: every C.vbnn_* it calls is declared in the header with that many
-- parameters, every struct field it sets exists, and the call ORDER of each function is the C host's and engine.py's.
local vb = require('vbnn_ffi')
local ffi, C, check = vb.ffi, vb.C, vb.check

local FusedMLP = {}
FusedMLP.__index = FusedMLP

local function f32(p) return ffi.cast('float*', p) end
local function packed(rows, cols, esize)
    local ld = vb.pad_ld(cols)
    return { p = vb.alloc(rows * ld * esize), ld = ld }
end

-- opt: input_size, hidden = {..}, n_classes (<= 16: the fused classifier head), var_init, B, S, seed, dtype ('bf16')
function FusedMLP.new(opt)
    local self = setmetatable({}, FusedMLP)
    self.opt = opt
    self.dtype = (opt.dtype == 'f32') and C.VBNN_F32 or C.VBNN_BF16
    self.esize = (self.dtype == C.VBNN_BF16) and 2 or 4
    self.seed, self.B, self.S = opt.seed or 3, opt.B, opt.S or 1
    self.world, self.rank = opt.world or 1, opt.rank or 0
    -- opt.kl_in_update (default: true for bf16, as engine.py): the gradient arena holds the LIKELIHOOD parts and update() adds the
    -- exact fp32 KL gradient (vbnn_update_desc.kl_add); false = the KL part fused into the accGradParameters epilogue (A/B)
    if opt.kl_in_update == nil then self.kl_in_update = (self.dtype == C.VBNN_BF16) else self.kl_in_update = opt.kl_in_update end
    -- opt.exchange_mode = 'sharded' (data-parallel, bf16): the sharded-update exchange instead of the all-reduce -- reduce-scatter of the
    -- gradients by layer rows, update() on this rank's rows, all-gather of the operand shadows + statistics (INTEGRATION.md section 4)
    self.sharded = opt.exchange_mode == 'sharded'
    if self.sharded then
        assert(opt.comm_id and self.dtype == C.VBNN_BF16, "exchange_mode = 'sharded': data-parallel (comm_id), bf16")
        self.kl_in_update = true
    end
    self.n_classes = opt.n_classes
    assert(self.n_classes <= 16, 'FusedMLP.lua drives the fused classifier head (mlp.lua:29-32); see engine.py for wider ones')
    local sizes = { opt.input_size }
    for _, h in ipairs(opt.hidden) do sizes[#sizes + 1] = h end
    self.sizes = sizes
    -- gradient arena: [d/dlvars | d/dmeans | d/dbias] per VB layer, then the final Linear (vbnn_amd/partition.py)
    local total = 0
    for li = 1, #sizes - 1 do total = total + 2 * sizes[li] * sizes[li + 1] + sizes[li + 1] end
    total = total + sizes[#sizes] * self.n_classes + self.n_classes
    self.n_grads = total
    self.grads = vb.alloc(total * 4)
    local off = 0
    local function take(n) local p = f32(self.grads) + off; off = off + n; return p end
    self.vb = {}
    for li = 1, #sizes - 1 do
        local I, O = sizes[li], sizes[li + 1]
        local v = { I = I, O = O, layer_id = li - 1, bucket_off = off }
        v.means, v.lvars, v.bias = vb.alloc(O * I * 4), vb.alloc(O * I * 4), vb.alloc(O * 4)
        v.m_mu, v.v_mu, v.m_lv, v.v_lv = vb.alloc(O * I * 4), vb.alloc(O * I * 4), vb.alloc(O * I * 4), vb.alloc(O * I * 4)
        v.grad_lv, v.grad_mu, v.gradBias = take(O * I), take(O * I), take(O)
        v.bucket_n = off - v.bucket_off
        v.stats = vb.alloc(32)
        v.mu_s, v.var_s = packed(O, I, self.esize), packed(O, I, self.esize)
        if li > 1 then v.muT_s, v.varT_s = packed(I, O, self.esize), packed(I, O, self.esize) end
        v.use_muT = li > 1                                        -- until _alloc_batch has asked the library (K-major weights)
        v.t = 0
        -- VBLinear.lua:18,22-28 + the He rule of mlp.lua:47-55 for the means (stream 5 = VBNN_STREAM_HEINIT)
        check(C.vbnn_fill_normal(vb.ctx, f32(v.means), O, I, I, self.seed, 5, v.layer_id, 0, 0, math.sqrt(2 / I)))
        local lv0 = torch.FloatTensor(O * I):fill(math.log(opt.msr_init and 2 / I or opt.var_init))
        check(C.vbnn_buf_upload(vb.ctx, v.lvars, lv0:data(), O * I * 4))
        self.vb[li] = v
    end
    local H = sizes[#sizes]
    self.weight3, self.bias3 = vb.alloc(self.n_classes * H * 4), vb.alloc(self.n_classes * 4)
    self.gradWeight3, self.gradBias3 = take(self.n_classes * H), take(self.n_classes)
    check(C.vbnn_fill_normal(vb.ctx, f32(self.weight3), self.n_classes, H, H, self.seed, 5, #self.vb, 0, 0, math.sqrt(2 / H)))
    self.w3_s = packed(self.n_classes, H, self.esize)
    self.acc, self.corr = vb.alloc(16), vb.alloc(4)
    self.draw, self.first = 0, true
    if opt.device_draw then self.draw_dev = ffi.cast('uint32_t*', vb.alloc(4)) end   -- the draw counter on the device
    if self.world > 1 then                                        -- the exchange (include/vbnn_hip.h: vbnn_comm_*)
        local box = ffi.new('vbnn_comm*[1]')
        check(C.vbnn_comm_create(vb.ctx, self.rank, self.world, opt.comm_id, box))
        self.comm = ffi.gc(box[0], C.vbnn_comm_destroy)
    end
    -- single GPU, layers of different sizes: every updateGradInput first, then the accGradParameters from the first layer
    -- up, so that the heavy launches alternate with lighter ones (engine.py: dx_first; the chip is power-bound in them)
    local wmin, wmax = math.huge, 0
    for k = 1, #sizes - 1 do
        wmin, wmax = math.min(wmin, sizes[k] * sizes[k + 1]), math.max(wmax, sizes[k] * sizes[k + 1])
    end
    self.dx_first = (not self.comm) and (2 * wmin <= wmax)
    self:prepare()
    return self
end

-- buffers that depend on the local batch size (engine.py:_alloc_batch; tools/c_host.c:fm_alloc_batch)
function FusedMLP:_alloc_batch(N)
    if self.N == N then return end
    self.N = N
    local km_ok = self.dtype == C.VBNN_BF16 and self.S == 1
    -- fp32 (the general kernel): the minibatch raw, x.x formed in registers, x / g / mu / sigma^2 K-major, the bias gradient
    -- from a synthetic row of ones (engine.py: f32_direct). Not with an exchange (its two-launch accGradParameters wants x.x)
    self.direct = self.dtype == C.VBNN_F32 and not self.comm
    local need_prepare = false
    local ones, ones_dev = torch.FloatTensor(N):fill(1), vb.alloc(N * 4)
    check(C.vbnn_buf_upload(vb.ctx, ones_dev, ones:data(), N * 4))
    for li, v in ipairs(self.vb) do
        local last = li == #self.vb
        v.bias_from_dw = (v.I % 256 ~= 0) and not last          -- the ones column / row of x: bias gradient from the GEMM
        local km = (km_ok and not self.direct) and C.vbnn_kmajor_supported_dw(v.I, v.O, N, v.bias_from_dw and 1 or 0) or 0
        v.dw_km = km > 0 or self.direct
        -- the two-launch form of accGradParameters (early d/dlvars message) needs either the transposed operands or the
        -- plain K-major launch of the two-pass kernel; the few-tile K-major launches compute both GEMMs in one grid
        v.early_ok = (not self.direct) and ((not v.dw_km) or ((not v.bias_from_dw) and C.vbnn_kmajor_supported(v.I, v.O, N) ~= 0))
        v.dx_km = li > 1 and (self.direct or (km_ok and C.vbnn_kmajor_supported(v.I, N, v.O) ~= 0))
        local use_muT = li > 1 and not v.dx_km
        if use_muT and not v.use_muT then need_prepare = true end
        v.use_muT = use_muT
        local extra = v.bias_from_dw and 1 or 0
        local xcols = v.I + (v.dw_km and extra or 0)
        if km == 2 then xcols = math.floor((xcols + 255) / 256) * 256 end
        if self.direct then xcols = v.I end                       -- (the row of ones is synthesised by the kernel)
        v.x_s = packed(N, xcols, self.esize)
        v.x2_s = (not self.direct) and packed(N, xcols, self.esize) or nil
        v.has_t = not v.dw_km                                     -- shapes without a K-major form get the transposed copies
        if v.dw_km and v.bias_from_dw and not self.direct then    -- column I of x is all ones, written once
            check(C.vbnn_pack(vb.ctx, self.dtype, C.VBNN_PACK_COPY, f32(ones_dev), nil, 1, N, 1,
                              ffi.cast('char*', v.x_s.p) + v.I * self.esize, v.x_s.ld, nil, 0))
        end
        if v.has_t then
            v.xT_s, v.x2T_s = packed(v.I + extra, N, self.esize), packed(v.I + extra, N, self.esize)
            v.gT_s, v.gvT_s = packed(v.O, N, self.esize), packed(v.O, N, self.esize)
            if v.bias_from_dw then                                -- row I of x^T is all ones
                check(C.vbnn_pack(vb.ctx, self.dtype, C.VBNN_PACK_COPY, f32(ones_dev), nil, N, 1, N,
                                  ffi.cast('char*', v.xT_s.p) + v.I * v.xT_s.ld * self.esize, v.xT_s.ld, nil, 0))
            end
        end
        v.g_s, v.gv_s = packed(N, v.O, self.esize), packed(N, v.O, self.esize)
        v.r = vb.alloc(N * v.O * self.esize)
    end
    self.h_s = packed(N, self.sizes[#self.sizes], self.esize)
    self.logits, self.out, self.g_logits = vb.alloc(N * self.n_classes * 4), vb.alloc(N * self.n_classes * 4), vb.alloc(N * self.n_classes * 4)
    -- the head's logits from the last VB layer's forward tiles, where that launch can carry them (vbnn_fwd_args.head_slots)
    local vl = self.vb[#self.vb]
    self.n_head_slots = self.draw_dev and 0 or C.vbnn_forward_head_slots(vb.ctx, self.dtype, N, vl.I, vl.O, self.n_classes)
    self.head_slots = self.n_head_slots > 0 and vb.alloc(self.n_head_slots * N * 16 * 4) or nil
    if need_prepare then self:prepare() end
end

function FusedMLP:resetGradients() self.first = true end          -- mlp.lua:62-67: the first draw overwrites

-- VBLinear:compute_prior (VBLinear.lua:77-88) + the operand shadows, once; afterwards vbnn_update maintains both
function FusedMLP:prepare()
    local n = #self.vb
    local d = ffi.new('vbnn_prep_desc[?]', n)
    for k, v in ipairs(self.vb) do
        local e = d[k - 1]
        e.means, e.lvars, e.O, e.I = f32(v.means), f32(v.lvars), v.O, v.I
        e.mu_s, e.var_s, e.ld_w = v.mu_s.p, v.var_s.p, v.mu_s.ld
        e.muT_s, e.varT_s = v.use_muT and v.muT_s.p or nil, v.use_muT and v.varT_s.p or nil
        e.ld_wT = v.muT_s and v.muT_s.ld or 0
        e.stats = ffi.cast('double*', v.stats)
    end
    local w3 = ffi.new('vbnn_pack_desc[1]')
    w3[0].src, w3[0].rows, w3[0].cols, w3[0].ld_src = f32(self.weight3), self.n_classes, self.sizes[#self.sizes], self.sizes[#self.sizes]
    w3[0].dst, w3[0].ld_dst, w3[0].dstT, w3[0].ld_dstT = self.w3_s.p, self.w3_s.ld, nil, 0
    check(C.vbnn_prepare(vb.ctx, self.dtype, n, d, w3))
end

function FusedMLP:sample()                                        -- mlp.lua:69-74: LRT draws its noise in the forward epilogue
    self.draw = self.draw + 1
    if self.draw_dev then check(C.vbnn_sample(vb.ctx, self.draw_dev, 1)) end   -- opt.device_draw: capturable (vbnn_capture_*)
end

-- mlp.lua:76-84, fused. inputs: DEVICE pointer to N x input_size floats (row pitch ld), targets: device int32[N], 0-based
-- sharded-update exchange: layer li's messages after its accGradParameters (tools/c_host.c: fm_scatter)
function FusedMLP:_scatter(li, lv, mu, small)
    local v = self.vb[li]
    local per = v.O * v.I / self.world
    if lv then check(C.vbnn_comm_reduce_scatter(self.comm, f32(self.grads) + v.bucket_off, per)) end
    if mu then check(C.vbnn_comm_reduce_scatter(self.comm, f32(self.grads) + v.bucket_off + v.O * v.I, per)) end
    if small then
        local off = v.bucket_off + 2 * v.O * v.I
        local stop = (li == #self.vb) and self.n_grads or (v.bucket_off + v.bucket_n)
        check(C.vbnn_allreduce_grads(self.comm, f32(self.grads) + off, stop - off))
    end
end

-- the update with the parameters SHARDED by layer rows (tools/c_host.c: fm_update_sharded; engine.py: _update_sharded)
function FusedMLP:_update_sharded(opt)
    local lr = opt.state.learningRate
    local H, G, R, n = self.sizes[#self.sizes], self.world, self.rank, #self.vb
    check(C.vbnn_sgd_step(vb.ctx, f32(self.weight3), self.gradWeight3, self.n_classes * H, lr))
    check(C.vbnn_sgd_step(vb.ctx, f32(self.bias3), self.gradBias3, self.n_classes, lr))
    self.stat_parts = self.stat_parts or vb.alloc(G * n * 4 * 8)
    local parts = ffi.cast('double*', self.stat_parts)
    local mine = parts + R * n * 4
    local d = ffi.new('vbnn_update_desc[?]', n)
    local st = ffi.new('double[4]')
    for k, v in ipairs(self.vb) do
        check(C.vbnn_sgd_step(vb.ctx, f32(v.bias), v.gradBias, v.O, lr))
        v.t = v.t + 1
        local nr = v.O / G
        local r0 = R * nr
        local o = r0 * v.I
        check(C.vbnn_buf_download(vb.ctx, st, v.stats, 32))          -- in: the WHOLE layer's pre-update statistics
        check(C.vbnn_buf_upload(vb.ctx, mine + 4 * (k - 1), st, 32))
        local e = d[k - 1]
        e.means, e.lvars, e.O, e.I = f32(v.means) + o, f32(v.lvars) + o, nr, v.I
        e.mu_s = ffi.cast('char*', v.mu_s.p) + r0 * v.mu_s.ld * self.esize
        e.var_s = ffi.cast('char*', v.var_s.p) + r0 * v.var_s.ld * self.esize
        e.ld_w = v.mu_s.ld
        e.stats, e.grad_mu, e.grad_lv = mine + 4 * (k - 1), v.grad_mu + o, v.grad_lv + o
        e.m_mu, e.v_mu, e.m_lv, e.v_lv = f32(v.m_mu) + o, f32(v.v_mu) + o, f32(v.m_lv) + o, f32(v.v_lv) + o
        for key, cfg in pairs({ mu = opt.meanState, lv = opt.varState }) do
            e[key].lr, e[key].beta1, e[key].beta2 = cfg.learningRate, cfg.beta1 or 0.9, cfg.beta2 or 0.999
            e[key].eps, e[key].lambda, e[key].t = cfg.epsilon or 1e-8, cfg.lambda or 1, v.t
        end
        e.lr_bias, e.B, e.kl_add = lr, self.B, 1
    end
    local w3 = ffi.new('vbnn_pack_desc[1]')
    w3[0].src, w3[0].rows, w3[0].cols, w3[0].ld_src = f32(self.weight3), self.n_classes, H, H
    w3[0].dst, w3[0].ld_dst, w3[0].dstT, w3[0].ld_dstT = self.w3_s.p, self.w3_s.ld, nil, 0
    check(C.vbnn_update(vb.ctx, self.dtype, n, d, w3))
    local stats = ffi.new('double*[?]', n)
    for k, v in ipairs(self.vb) do
        check(C.vbnn_comm_all_gather(self.comm, v.mu_s.p, v.O / G * v.mu_s.ld * self.esize))
        check(C.vbnn_comm_all_gather(self.comm, v.var_s.p, v.O / G * v.var_s.ld * self.esize))
        stats[k - 1] = ffi.cast('double*', v.stats)
    end
    check(C.vbnn_comm_all_gather(self.comm, self.stat_parts, n * 4 * 8))
    check(C.vbnn_comm_finish(self.comm))
    check(C.vbnn_stats_combine(vb.ctx, n, G, parts, stats))
    for _, v in ipairs(self.vb) do
        if v.use_muT then
            check(C.vbnn_transpose_packed(vb.ctx, self.dtype, v.mu_s.p, v.mu_s.ld, v.O, v.I, v.muT_s.p, v.muT_s.ld))
            check(C.vbnn_transpose_packed(vb.ctx, self.dtype, v.var_s.p, v.var_s.ld, v.O, v.I, v.varT_s.p, v.varT_s.ld))
        end
    end
end

function FusedMLP:run(inputs, ld, targets, N)
    self:_alloc_batch(N)
    local accumulate = self.first and 0 or 1
    local inv_n = 1 / (N * self.world)
    local row0 = self.rank * N
    local v0 = self.vb[1]
    for _, v in ipairs(self.vb) do v.x_in, v.ld_in = v.x_s.p, v.x_s.ld end
    if self.direct and ld % 4 == 0 and tonumber(ffi.cast('uintptr_t', inputs)) % 16 == 0 then
        v0.x_in, v0.ld_in = inputs, ld                            -- the GEMMs read the minibatch where it lies: no packing launch
    else
        check(C.vbnn_pack_input(vb.ctx, self.dtype, f32(inputs), ld, N, v0.I, v0.x_s.p, v0.x2_s and v0.x2_s.p or nil, v0.x_s.ld,
                                v0.has_t and v0.xT_s.p or nil, v0.has_t and v0.x2T_s.p or nil, v0.has_t and v0.xT_s.ld or 0, 0))
    end
    -- forward: dual GEMM + noise / ReLU / operand packing in the epilogue
    for li, v in ipairs(self.vb) do
        local nxt = self.vb[li + 1]
        local fa = ffi.new('vbnn_fwd_args')
        fa.w, fa.w2, fa.x, fa.x2, fa.ld_w, fa.ld_x = v.mu_s.p, v.var_s.p, v.x_in, v.x2_s and v.x2_s.p or nil, v.mu_s.ld, v.ld_in
        fa.N, fa.I, fa.O, fa.bias = N, v.I, v.O, f32(v.bias)
        fa.seed, fa.layer, fa.draw, fa.row0 = self.seed, v.layer_id, self.draw_dev and 0 or self.draw, row0
        fa.draw_dev = self.draw_dev
        fa.r, fa.ld_r, fa.r_packed, fa.relu = v.r, v.O, 1, 1
        fa.h = nxt and nxt.x_s.p or self.h_s.p
        fa.h2 = (nxt and nxt.x2_s) and nxt.x2_s.p or nil
        fa.ld_h = nxt and nxt.x_s.ld or self.h_s.ld
        if nxt and nxt.has_t then fa.hT, fa.h2T, fa.ld_hT = nxt.xT_s.p, nxt.x2T_s.p, nxt.xT_s.ld end
        if not nxt and self.n_head_slots > 0 then
            fa.head_w3, fa.head_ld_w, fa.head_C, fa.head_slots = self.w3_s.p, self.w3_s.ld, self.n_classes, ffi.cast('float*', self.head_slots)
        end
        check(C.vbnn_forward(vb.ctx, self.dtype, fa))
    end
    -- final Linear + LogSoftMax + ClassNLL (mlp.lua:29-32), forward and backward
    local vl, H = self.vb[#self.vb], self.sizes[#self.sizes]
    local ha = ffi.new('vbnn_head_args')
    ha.h, ha.ld_h, ha.w3, ha.ld_w, ha.bias = self.h_s.p, self.h_s.ld, self.w3_s.p, self.w3_s.ld, f32(self.bias3)
    ha.target = ffi.cast('const int32_t*', targets)
    ha.N, ha.H, ha.C, ha.rows_per_draw, ha.inv_n, ha.accumulate = N, H, self.n_classes, 0, inv_n, accumulate
    ha.logits, ha.out, ha.g_logits = f32(self.logits), f32(self.out), f32(self.g_logits)
    ha.loss_sum_dev, ha.correct_dev = ffi.cast('double*', self.acc), ffi.cast('int32_t*', self.corr)
    ha.gradWeight, ha.gradBias, ha.gradBias_prev = self.gradWeight3, self.gradBias3, vl.gradBias
    ha.relu_mask, ha.r_prev_packed, ha.r_prev, ha.ld_r_prev = 1, 1, vl.r, vl.O
    ha.g_prev, ha.gv_prev, ha.ld_gp = vl.g_s.p, vl.gv_s.p, vl.g_s.ld
    if vl.has_t then ha.gT_prev, ha.gvT_prev, ha.ld_gpT = vl.gT_s.p, vl.gvT_s.p, vl.gT_s.ld end
    if self.n_head_slots > 0 then ha.logit_slots, ha.n_slots = ffi.cast('const float*', self.head_slots), self.n_head_slots end
    check(C.vbnn_head_forward_backward(vb.ctx, self.dtype, ha))
    -- backward. The argument blocks of layer li (no library call in these two):
    local function dw_block(li)
        local v = self.vb[li]
        local d = ffi.new('vbnn_dw_args')
        if v.has_t then d.xT, d.x2T, d.gT, d.gvT, d.ld_n = v.xT_s.p, v.x2T_s.p, v.gT_s.p, v.gvT_s.p, v.gT_s.ld end
        d.N, d.I, d.O, d.scale, d.accumulate = N, v.I, v.O, 1, accumulate
        d.seed, d.layer, d.draw, d.lvars = self.seed, v.layer_id, self.draw, f32(v.lvars)
        d.grad_mu, d.grad_lv, d.means, d.stats = v.grad_mu, v.grad_lv, f32(v.means), ffi.cast('double*', v.stats)
        -- the KL gradient: exact, from the fp32 parameters in the update sweep (kl_in_update: the default where the epilogue would
        -- read the bf16 shadows -- (bf16(s2) / var_hat - 1) cancels; VBLinear.lua:96-97 uses the fp32 vars) or fused here (A/B)
        d.B, d.S, d.kl_scale = self.B, self.S, self.kl_in_update and 0 or 1 / self.world
        d.gradBias = v.bias_from_dw and v.gradBias or nil
        d.x, d.x2, d.g, d.gv, d.ld_x, d.ld_g = v.x_in, v.x2_s and v.x2_s.p or nil, v.g_s.p, v.gv_s.p, v.ld_in, v.g_s.ld
        if self.dtype == C.VBNN_BF16 then d.mu_s, d.var_s, d.ld_w = v.mu_s.p, v.var_s.p, v.mu_s.ld end   -- KL terms from the shadows
        return d
    end
    local function dx_block(li)
        local v, p = self.vb[li], self.vb[li - 1]
        local xa = ffi.new('vbnn_dx_args')
        if v.use_muT then xa.wT, xa.w2T = v.muT_s.p, v.varT_s.p end
        xa.ld_wT = v.muT_s.ld
        xa.g, xa.gv, xa.ld_g, xa.N, xa.I, xa.O = v.g_s.p, v.gv_s.p, v.g_s.ld, N, v.I, v.O
        xa.x, xa.ld_x, xa.relu_mask = v.x_s.p, v.x_s.ld, 1
        xa.r_prev, xa.ld_r_prev, xa.r_prev_packed = p.r, p.O, 1
        xa.g_prev, xa.gv_prev, xa.ld_gp = p.g_s.p, p.gv_s.p, p.g_s.ld
        if p.has_t then xa.gT_prev, xa.gvT_prev, xa.ld_gpT = p.gT_s.p, p.gvT_s.p, p.gT_s.ld end
        xa.w, xa.w2, xa.ld_w = v.mu_s.p, v.var_s.p, v.mu_s.ld
        return xa
    end
    if self.direct then
        -- fp32: accGradParameters and updateGradInput of a layer are independent and go out as ONE launch where the library
        -- can carry both (vbnn_backward_pair); each tile bitwise what its own launch computes
        for li = #self.vb, 1, -1 do
            local v = self.vb[li]
            local dd = dw_block(li)
            if li > 1 then
                check(C.vbnn_backward_pair(vb.ctx, self.dtype, dx_block(li), dd))
            else
                check(C.vbnn_acc_grad_parameters(vb.ctx, self.dtype, dd))
            end
            if li < #self.vb and not v.bias_from_dw then
                check(C.vbnn_acc_grad_bias(vb.ctx, self.dtype, v.g_s.p, v.g_s.ld, N, v.O, 1, accumulate, v.gradBias))
            end
        end
    elseif self.dx_first then
        for li = #self.vb, 2, -1 do
            check(C.vbnn_grad_input(vb.ctx, self.dtype, dx_block(li)))
        end
        for li = 1, #self.vb do
            local v = self.vb[li]
            check(C.vbnn_acc_grad_parameters(vb.ctx, self.dtype, dw_block(li)))
            if li < #self.vb and not v.bias_from_dw then
                check(C.vbnn_acc_grad_bias(vb.ctx, self.dtype, v.g_s.p, v.g_s.ld, N, v.O, 1, accumulate, v.gradBias))
            end
        end
    else
        -- last VB layer first: accGradParameters (+ its bucket's all-reduce), then updateGradInput
        for li = #self.vb, 1, -1 do
            local v = self.vb[li]
            local dd = dw_block(li)
            local msg_off = v.bucket_off
            local early = self.comm and v.O * v.I >= 2 ^ 22 and v.early_ok
            if early then
                -- two launches (vbnn_dw_args.part): the sigma^2 GEMM and d/dlvars first, whose exchange then starts while the
                -- mu GEMM still runs (d/dlvars is the first block of the layer's bucket)
                dd.part = 2
                check(C.vbnn_acc_grad_parameters(vb.ctx, self.dtype, dd))
                if self.sharded then self:_scatter(li, true, false, false)
                else check(C.vbnn_allreduce_grads(self.comm, f32(self.grads) + v.bucket_off, v.O * v.I)) end
                dd.part = 1
                msg_off = v.bucket_off + v.O * v.I
            end
            check(C.vbnn_acc_grad_parameters(vb.ctx, self.dtype, dd))
            if li < #self.vb and not v.bias_from_dw then
                check(C.vbnn_acc_grad_bias(vb.ctx, self.dtype, v.g_s.p, v.g_s.ld, N, v.O, 1, accumulate, v.gradBias))
            end
            if self.comm and self.sharded then
                self:_scatter(li, not early, true, true)
            elseif self.comm then                                 -- the final Linear's gradients ride in the last layer's message
                local n = ((li == #self.vb) and self.n_grads or (v.bucket_off + v.bucket_n)) - msg_off
                check(C.vbnn_allreduce_grads(self.comm, f32(self.grads) + msg_off, n))
            end
            if li > 1 then
                check(C.vbnn_grad_input(vb.ctx, self.dtype, dx_block(li)))
            end
        end
    end
    self.first = false
end

function FusedMLP:finish()                                        -- end of the minibatch: gradients complete on the stream
    if self.comm then check(C.vbnn_comm_finish(self.comm)) end
end

-- mlp:update + VBLinear:update (mlp.lua:117-142, VBLinear.lua:124-166) in one call, which also leaves the operand
-- shadows and prior statistics of the next minibatch; `log` (a FloatTensor-free double[14 * layers] on the device)
function FusedMLP:update(opt, log14)
    self:finish()
    if self.sharded then return self:_update_sharded(opt) end
    local lr = opt.state.learningRate
    local H = self.sizes[#self.sizes]
    check(C.vbnn_sgd_step(vb.ctx, f32(self.weight3), self.gradWeight3, self.n_classes * H, lr))
    check(C.vbnn_sgd_step(vb.ctx, f32(self.bias3), self.gradBias3, self.n_classes, lr))
    local n = #self.vb
    local d = ffi.new('vbnn_update_desc[?]', n)
    for k, v in ipairs(self.vb) do
        v.t = v.t + 1
        local e = d[k - 1]
        e.means, e.lvars, e.O, e.I = f32(v.means), f32(v.lvars), v.O, v.I
        e.mu_s, e.var_s, e.ld_w = v.mu_s.p, v.var_s.p, v.mu_s.ld
        e.muT_s, e.varT_s = v.use_muT and v.muT_s.p or nil, v.use_muT and v.varT_s.p or nil
        e.ld_wT = v.muT_s and v.muT_s.ld or 0
        e.stats, e.grad_mu, e.grad_lv = ffi.cast('double*', v.stats), v.grad_mu, v.grad_lv
        e.m_mu, e.v_mu, e.m_lv, e.v_lv = f32(v.m_mu), f32(v.v_mu), f32(v.m_lv), f32(v.v_lv)
        for key, st in pairs({ mu = opt.meanState, lv = opt.varState }) do
            e[key].lr, e[key].beta1, e[key].beta2 = st.learningRate, st.beta1 or 0.9, st.beta2 or 0.999
            e[key].eps, e[key].lambda, e[key].t = st.epsilon or 1e-8, st.lambda or 1, v.t
        end
        e.bias, e.grad_bias, e.lr_bias, e.B = f32(v.bias), v.gradBias, lr, self.B
        e.log14 = log14 and (ffi.cast('double*', log14) + 14 * (k - 1)) or nil
        e.kl_add = self.kl_in_update and 1 or 0
    end
    local w3 = ffi.new('vbnn_pack_desc[1]')
    w3[0].src, w3[0].rows, w3[0].cols, w3[0].ld_src = f32(self.weight3), self.n_classes, H, H
    w3[0].dst, w3[0].ld_dst, w3[0].dstT, w3[0].ld_dstT = self.w3_s.p, self.w3_s.ld, nil, 0
    check(C.vbnn_update(vb.ctx, self.dtype, n, d, w3))
end

-- mlp:calc_lc (mlp.lua:109-115): sum over the VB layers of VBLinear:calc_lc (VBLinear.lua:99-103), fresh statistics
function FusedMLP:calc_lc(opt)
    local lc, box, dev = 0, ffi.new('double[1]'), vb.alloc(8)
    for _, v in ipairs(self.vb) do
        check(C.vbnn_calc_lc(vb.ctx, f32(v.means), f32(v.lvars), nil, nil, ffi.cast('double*', v.stats), (opt or self.opt).B, nil,
                             ffi.cast('double*', dev), v.O * v.I))
        check(C.vbnn_buf_download(vb.ctx, box, dev, 8))
        lc = lc + box[0]
    end
    return lc
end

-- error (mean NLL over the GLOBAL batch, this rank's share) and hit count of the last run(s); synchronises
function FusedMLP:loss_and_accuracy()
    self:finish()
    local a, c = ffi.new('double[2]'), ffi.new('int32_t[1]')
    check(C.vbnn_buf_download(vb.ctx, a, self.acc, 16))
    check(C.vbnn_buf_download(vb.ctx, c, self.corr, 4))
    return a[0], c[0]
end

return FusedMLP
