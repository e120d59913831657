-- lua/VBLinear.lua -- drop-in replacement of the reference's VBLinear.lua for MI355X.
--
-- Same class (`nn.VBLinear`, parent `nn.Linear`), constructor, fields and methods as the reference
-- (VBLinear.lua:7-166), so mlp.lua / convnet.lua run unchanged with `opt.cuda = false` and the new
-- `opt.hip = true`: host FloatTensors stay the visible `weight bias gradWeight gradBias means lvars
-- gradSum output gradInput` (getParameters at mlp.lua:37 keeps working); the arithmetic runs in
-- libvbnn_hip.so through lua/vbnn_ffi.lua. `opt.mode = 'wn'` (default) is the reference's weight-noise
-- sampling; `opt.mode = 'lrt'` is local reparameterisation (the throughput path). `opt.dtype`: 'f32'|'bf16'.
--
-- This is the MODULE-LEVEL path: each call uploads its input and downloads its output (PCIe), which is what
-- "unchanged mlp.lua" implies because nn.ReLU / nn.LogSoftMax between the layers are stock host modules. The
-- device-resident path is the call sequence of vbnn_amd/engine.py:FusedMLP.run over the same FFI
-- (INTEGRATION.md section 4): one sequence per minibatch, activations never leave HBM.
--
-- NOT EXECUTED in the build image (no LuaJIT / Torch7 there): kept in step with vbnn_amd/nn.py, which is the
-- same logic in Python and is what the GPU parity tests run.
require 'nn'
require 'optim'          -- mlp.lua uses `optim` and `randomkit` as globals the reference's VBLinear.lua provided
pcall(require, 'randomkit')
local u = require('utils')
local vb = require('vbnn_ffi')
local ffi, C, check = vb.ffi, vb.C, vb.check

local VBLinear, parent = torch.class('nn.VBLinear', 'nn.Linear')

local DT = { f32 = C.VBNN_F32, bf16 = C.VBNN_BF16 }
local function fptr(t) return t:contiguous():data() end

-- device mirror of a host FloatTensor
local function dev_like(t) return vb.alloc(t:nElement() * 4) end
local function upload(d, t) check(C.vbnn_buf_upload(vb.ctx, d, fptr(t), t:nElement() * 4)) end
local function download(t, d) check(C.vbnn_buf_download(vb.ctx, t:data(), d, t:nElement() * 4)) end
local function packed(rows, cols, esize) return vb.alloc(rows * vb.pad_ld(cols) * esize), vb.pad_ld(cols) end

function VBLinear:__init(inputSize, outputSize, opt)
    parent.__init(self, inputSize, outputSize)
    self.opt = opt
    self.var_init = self.opt.var_init                               -- VBLinear.lua:12
    self.bias:zero()                                                -- :13
    if opt.msr_init then self.var_init = 2 / self.weight:size(2) end -- :14-16
    print("var: ", self.var_init)                                   -- :17
    self.lvars = torch.Tensor(outputSize, inputSize):fill(torch.log(self.var_init))   -- :18
    self.gradSum = torch.Tensor(outputSize, inputSize):zero()       -- :20
    self.W = outputSize * inputSize                                 -- :21
    self.seed = opt.seed or 3                                       -- config.lua:40
    self.layer_id = opt._next_layer_id or 0
    opt._next_layer_id = self.layer_id + 1
    self.draw = 0
    self.mode = opt.mode or 'wn'
    self.dtype = DT[opt.dtype or 'f32']
    self.esize = (self.dtype == C.VBNN_BF16) and 2 or 4
    self.means = torch.Tensor(outputSize, inputSize):zero()         -- :22-23
    self.biasState = u.shallow_copy(opt.state)                      -- :31-33
    self.meanState = u.shallow_copy(opt.meanState)
    self.varState = u.shallow_copy(opt.varState)
    -- device state
    local O, I = outputSize, inputSize
    self.d = { means = dev_like(self.means), lvars = dev_like(self.lvars), weight = dev_like(self.weight),
               bias = dev_like(self.bias), gradWeight = dev_like(self.gradWeight), gradSum = dev_like(self.gradSum),
               gradBias = dev_like(self.bias), vars = dev_like(self.means), stdv = dev_like(self.means),
               mu_sqe = dev_like(self.means), stats = vb.alloc(32), lc = vb.alloc(8), lcg = dev_like(self.means) }
    self.d.w, self.ld_w = packed(O, I, self.esize)
    self.d.w2 = packed(O, I, self.esize)
    self.d.wT, self.ld_wT = packed(I, O, self.esize)
    self.d.w2T = packed(I, O, self.esize)
    if opt.mu_init ~= 0 then                                        -- :24-28: means ~ N(0, sqrt(var_init))
        check(C.vbnn_fill_normal(vb.ctx, ffi.cast('float*', self.d.means), O, I, I, self.seed, 3, self.layer_id, 0, 0,
                                 math.sqrt(self.var_init)))
        download(self.means, self.d.means)
    end
    self.vars = torch.Tensor(O, I); self.stdv = torch.Tensor(O, I); self.mu_sqe = torch.Tensor(O, I)
    self.mu, self.lv = self.means, self.lvars                       -- aliases used by the north_star text
    self:compute_prior()                                            -- :46
end

-- host parameters -> device (call after the optimiser touched means / lvars / bias on the host)
function VBLinear:syncToDevice()
    upload(self.d.means, self.means); upload(self.d.lvars, self.lvars); upload(self.d.bias, self.bias)
end

function VBLinear:compute_prior()                                   -- VBLinear.lua:77-88
    self:syncToDevice()
    local f = function(p) return ffi.cast('float*', p) end
    check(C.vbnn_compute_prior(vb.ctx, f(self.d.means), f(self.d.lvars), self.W, f(self.d.vars), f(self.d.stdv),
                               f(self.d.mu_sqe), ffi.cast('double*', self.d.stats)))
    local stats = ffi.new('double[4]')
    check(C.vbnn_buf_download(vb.ctx, stats, self.d.stats, 32))
    self.mu_hat = 0                                                 -- :81
    self.var_hat = stats[2]                                         -- :86
    download(self.vars, self.d.vars); download(self.stdv, self.d.stdv); download(self.mu_sqe, self.d.mu_sqe)
    return self.mu_hat, self.var_hat
end

function VBLinear:sample(opt)                                       -- VBLinear.lua:49-64
    self.draw = self.draw + 1
    self._map = false
    if self.mode == 'wn' then
        local f = function(p) return ffi.cast('float*', p) end
        check(C.vbnn_wn_sample(vb.ctx, f(self.d.means), f(self.d.stdv), nil, f(self.d.weight), nil,
                               self.weight:size(1), self.weight:size(2), self.seed, self.layer_id, self.draw))
        download(self.weight, self.d.weight)                        -- keep the visible `weight` in step (:63)
    end
end

function VBLinear:clamp_to_map()                                    -- VBLinear.lua:105-107
    self.weight:copy(self.means)
    self._map = true
end

function VBLinear:resetAcc()                                        -- VBLinear.lua:120-122
    self.gradSum:zero()
    check(C.vbnn_buf_zero(vb.ctx, self.d.gradSum, self.W * 4))
end

local function lrt(self) return self.mode == 'lrt' and not self._map end

-- pack the layer's weight-side operands for the current mode
function VBLinear:_pack_weights()
    local O, I = self.weight:size(1), self.weight:size(2)
    local f = function(p) return ffi.cast('float*', p) end
    if lrt(self) then
        check(C.vbnn_pack(vb.ctx, self.dtype, C.VBNN_PACK_COPY, f(self.d.means), nil, I, O, I, self.d.w, self.ld_w, self.d.wT, self.ld_wT))
        check(C.vbnn_pack(vb.ctx, self.dtype, C.VBNN_PACK_EXP, f(self.d.lvars), nil, I, O, I, self.d.w2, self.ld_w, self.d.w2T, self.ld_wT))
    else
        upload(self.d.weight, self.weight)
        check(C.vbnn_pack(vb.ctx, self.dtype, C.VBNN_PACK_COPY, f(self.d.weight), nil, I, O, I, self.d.w, self.ld_w, self.d.wT, self.ld_wT))
    end
end

function VBLinear:_batch(N)
    if self._N == N then return end
    self._N = N
    local O, I = self.weight:size(1), self.weight:size(2)
    local b = {}
    b.xin = vb.alloc(N * I * 4); b.y = vb.alloc(N * O * 4); b.r = vb.alloc(N * O * 4)
    b.gin = vb.alloc(N * O * 4); b.gx = vb.alloc(N * I * 4)
    b.x, self.ld_x = packed(N, I, self.esize);  b.x2 = packed(N, I, self.esize)
    b.xT, self.ld_n = packed(I, N, self.esize); b.x2T = packed(I, N, self.esize)
    b.g, self.ld_g = packed(N, O, self.esize);  b.gv = packed(N, O, self.esize)
    b.gT = packed(O, N, self.esize);            b.gvT = packed(O, N, self.esize)
    self.b = b
end

function VBLinear:updateOutput(input)                               -- inherited nn.Linear:updateOutput (VBLinear.lua:7)
    assert(input:dim() == 2, 'nn.VBLinear needs a 2-D batch x inputSize tensor (VBLinear.lua:114)')
    local N, O, I = input:size(1), self.weight:size(1), self.weight:size(2)
    self:_batch(N); self:_pack_weights()
    local b, f = self.b, function(p) return ffi.cast('float*', p) end
    upload(b.xin, input); upload(self.d.bias, self.bias)
    check(C.vbnn_pack(vb.ctx, self.dtype, C.VBNN_PACK_COPY, f(b.xin), nil, I, N, I, b.x, self.ld_x, b.xT, self.ld_n))
    local a = ffi.new('vbnn_fwd_args')
    a.w = self.d.w; a.x = b.x; a.ld_w = self.ld_w; a.ld_x = self.ld_x; a.N = N; a.I = I; a.O = O
    a.bias = f(self.d.bias); a.y = f(b.y); a.ld_y = O
    if lrt(self) then
        check(C.vbnn_pack(vb.ctx, self.dtype, C.VBNN_PACK_SQUARE, f(b.xin), nil, I, N, I, b.x2, self.ld_x, b.x2T, self.ld_n))
        a.w2 = self.d.w2; a.x2 = b.x2; a.seed = self.seed; a.layer = self.layer_id; a.draw = self.draw
        a.row0 = self.row0 or 0; a.r = f(b.r); a.ld_r = O
    end
    check(C.vbnn_forward(vb.ctx, self.dtype, a))
    self.output:resize(N, O)
    download(self.output, b.y)
    return self.output
end

function VBLinear:_pack_grad(gradOutput)
    local N, O = gradOutput:size(1), self.weight:size(1)
    local b, f = self.b, function(p) return ffi.cast('float*', p) end
    upload(b.gin, gradOutput)
    check(C.vbnn_pack(vb.ctx, self.dtype, C.VBNN_PACK_COPY, f(b.gin), nil, O, N, O, b.g, self.ld_g, b.gT, self.ld_n))
    if lrt(self) then
        check(C.vbnn_pack(vb.ctx, self.dtype, C.VBNN_PACK_MUL, f(b.gin), f(b.r), O, N, O, b.gv, self.ld_g, b.gvT, self.ld_n))
    end
    self._g_fresh = true
end

function VBLinear:updateGradInput(input, gradOutput)                -- inherited (stub at VBLinear.lua:109-110)
    local N, O, I = input:size(1), self.weight:size(1), self.weight:size(2)
    self:_pack_grad(gradOutput)
    local b, f = self.b, function(p) return ffi.cast('float*', p) end
    local a = ffi.new('vbnn_dx_args')
    a.wT = self.d.wT; a.g = b.g; a.ld_wT = self.ld_wT; a.ld_g = self.ld_g; a.N = N; a.I = I; a.O = O
    a.gx = f(b.gx); a.ld_gx = I
    if lrt(self) then a.w2T = self.d.w2T; a.gv = b.gv; a.x = b.x; a.ld_x = self.ld_x end
    check(C.vbnn_grad_input(vb.ctx, self.dtype, a))
    self.gradInput:resize(N, I)
    download(self.gradInput, b.gx)
    return self.gradInput
end

function VBLinear:accGradParameters(input, gradOutput, scale)       -- VBLinear.lua:112-118
    scale = scale or 1
    local N, O, I = input:size(1), self.weight:size(1), self.weight:size(2)
    if not self._g_fresh then self:_pack_grad(gradOutput) end
    self._g_fresh = false
    local b, f = self.b, function(p) return ffi.cast('float*', p) end
    upload(self.d.gradWeight, self.gradWeight); upload(self.d.gradBias, self.gradBias); upload(self.d.gradSum, self.gradSum)
    local a = ffi.new('vbnn_dw_args')
    a.xT = b.xT; a.gT = b.gT; a.ld_n = self.ld_n; a.N = N; a.I = I; a.O = O; a.scale = scale; a.accumulate = 1
    a.gradWeight = f(self.d.gradWeight); a.seed = self.seed; a.layer = self.layer_id; a.draw = self.draw
    a.lvars = f(self.d.lvars)
    if not self._map then a.gradSum = f(self.d.gradSum) end
    if lrt(self) then a.x2T = b.x2T; a.gvT = b.gvT end
    check(C.vbnn_acc_grad_parameters(vb.ctx, self.dtype, a))
    check(C.vbnn_acc_grad_bias(vb.ctx, C.VBNN_F32, b.gin, O, N, O, scale, 1, f(self.d.gradBias)))
    download(self.gradWeight, self.d.gradWeight); download(self.gradBias, self.d.gradBias); download(self.gradSum, self.d.gradSum)
end

function VBLinear:compute_mugrads(opt)                              -- VBLinear.lua:90-93
    local f = function(p) return ffi.cast('float*', p) end
    upload(self.d.gradWeight, self.gradWeight)
    check(C.vbnn_compute_mugrads(vb.ctx, f(self.d.means), ffi.cast('double*', self.d.stats), opt.B, opt.S,
                                 f(self.d.gradWeight), f(self.d.lcg), self.W))
    local lcg = torch.Tensor(self.means:size())
    download(self.gradWeight, self.d.gradWeight); download(lcg, self.d.lcg)
    return self.gradWeight, lcg
end

function VBLinear:compute_vargrads(opt)                             -- VBLinear.lua:95-98
    local f = function(p) return ffi.cast('float*', p) end
    upload(self.d.gradSum, self.gradSum)
    check(C.vbnn_compute_vargrads(vb.ctx, nil, f(self.d.vars), f(self.d.stdv), ffi.cast('double*', self.d.stats), opt.B,
                                  opt.S, f(self.d.gradSum), f(self.d.lcg), self.W))
    local lcg = torch.Tensor(self.means:size())
    download(self.gradSum, self.d.gradSum); download(lcg, self.d.lcg)
    return self.gradSum, lcg
end

function VBLinear:calc_lc(opt)                                      -- VBLinear.lua:99-103; returns a 1-element tensor (:sum() works)
    local f = function(p) return ffi.cast('float*', p) end
    check(C.vbnn_calc_lc(vb.ctx, nil, nil, f(self.d.vars), f(self.d.mu_sqe), ffi.cast('double*', self.d.stats), opt.B, nil,
                         ffi.cast('double*', self.d.lc), self.W))
    local s = ffi.new('double[1]')
    check(C.vbnn_buf_download(vb.ctx, s, self.d.lc, 8))
    return torch.Tensor(1):fill(s[0])
end

-- VBLinear:update (VBLinear.lua:124-166): SGD on the bias, compute_prior, likelihood + KL gradients, Adam on means
-- (opt.meanState) and on lvars (opt.varState) with per-layer moment state -- on the device (vbnn_sgd_step /
-- vbnn_adam_step: one streaming pass per tensor, the two gradient parts added inside it); the host tensors are
-- refreshed afterwards because mlp.lua and the logger read them. Same fourteen series as the reference (:149-164).
local function adam(self, key, x_host, d_x, d_g1, d_g2, cfg)
    local f = function(p) return ffi.cast('float*', p) end
    local st = self._adam[key]
    if not st then
        st = { t = 0, m = dev_like(x_host), v = dev_like(x_host), norms = vb.alloc(16) }
        check(C.vbnn_buf_zero(vb.ctx, st.m, self.W * 4)); check(C.vbnn_buf_zero(vb.ctx, st.v, self.W * 4))
        self._adam[key] = st
    end
    st.t = st.t + 1
    check(C.vbnn_adam_step(vb.ctx, f(d_x), f(d_g1), f(d_g2), f(st.m), f(st.v), self.W, cfg.learningRate,
                           cfg.beta1 or 0.9, cfg.beta2 or 0.999, cfg.epsilon or 1e-8, cfg.lambda or 1, st.t,
                           ffi.cast('double*', st.norms)))
    local n = ffi.new('double[2]')
    check(C.vbnn_buf_download(vb.ctx, n, st.norms, 16))
    return n[0] / n[1]                                              -- torch.norm(update) / torch.norm(x), :139,144
end

function VBLinear:update(opt)
    local f = function(p) return ffi.cast('float*', p) end
    self._adam = self._adam or {}
    upload(self.d.bias, self.bias); upload(self.d.gradBias, self.gradBias)
    check(C.vbnn_sgd_step(vb.ctx, f(self.d.bias), f(self.d.gradBias), self.bias:nElement(), self.biasState.learningRate))  -- :125-128
    download(self.bias, self.d.bias)
    self:compute_prior()                                            -- :130
    local mleg, mlcg = self:compute_mugrads(opt)                    -- :131  (d.gradWeight, d.lcg hold them on the device)
    local d_mlcg = self.d.lcg
    self.d.lcg = self.d.lcg2 or dev_like(self.means)                -- keep the first KL gradient while the second is formed
    local vleg, vlcg = self:compute_vargrads(opt)                   -- :133
    local d_vlcg = self.d.lcg
    local mu_normratio = adam(self, 'mean', self.means, self.d.means, self.d.gradWeight, d_mlcg, self.meanState)   -- :135-139
    local var_normratio = adam(self, 'var', self.lvars, self.d.lvars, self.d.gradSum, d_vlcg, self.varState)       -- :140-144
    self.d.lcg, self.d.lcg2 = d_mlcg, d_vlcg
    download(self.means, self.d.means); download(self.lvars, self.d.lvars)
    if opt.log then                                                 -- :148-164
        local vars = torch.exp(self.lvars)
        Log:add('vlc grad', vlcg:norm() / self.lvars:norm())
        Log:add('vle grad', vleg:norm() / self.lvars:norm())
        Log:add('mlc grad', mlcg:norm() / self.means:norm())
        Log:add('mle grad', mleg:norm() / self.means:norm())
        Log:add('min variance', vars:min())
        Log:add('max variance', vars:max())
        Log:add('mean variance', vars:mean())
        Log:add('var hat', self.var_hat)
        Log:add('mean means', self.means:mean())
        Log:add('std means', self.means:std())
        Log:add('min. means', self.means:min())
        Log:add('max. means', self.means:max())
        Log:add('mu normratio', mu_normratio)
        Log:add('var normratio', var_normratio)
    end
end
