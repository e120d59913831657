"""Host-side mirror of the Torch7 `nn` surface the reference's hot path sits behind.

The reference is Lua (VBLinear.lua, mlp.lua); no Lua/LuaJIT/Torch7 exists in the build image,
so the host side above the C ABI is Python with the reference's own names and argument
meaning: `nn.VBLinear(inputSize, outputSize, opt)` with `updateOutput / updateGradInput /
accGradParameters / sample / compute_prior / compute_mugrads / compute_vargrads / calc_lc /
clamp_to_map / resetAcc`, fields `weight bias gradWeight gradBias means lvars gradSum`
(aliases `mu`, `lv`), and the glue modules of mlp.lua:11-32. The LuaJIT-FFI twin of this file
is lua/VBLinear.lua.

Tensors are torch CUDA(ROCm) tensors used as device memory only: every computation is a call
into libvbnn_hip.so through include/vbnn_hip.h on raw device pointers. Differences from the
Lua surface, all forced by the host language: class targets are 0-based; `opt` is a dict.
"""
import ctypes as C
import math

import torch

from . import _lib as L

_DT = {"f32": (L.F32, torch.float32), "bf16": (L.BF16, torch.bfloat16)}


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Context:
    """(device, stream) handle of the library; one per device, bound to torch's current stream."""
    _cache = {}

    def __init__(self, device, stream=None):
        """stream: a torch.cuda.Stream to bind to (default: torch's current stream on `device`). Each context
        owns its reduction scratch, so kernels launched through different contexts may run concurrently."""
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.torch_stream = stream
        stream = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        h = C.c_void_p()
        L.check(L.lib().vbnn_ctx_create(device, C.c_void_p(stream), C.byref(h)))
        self.h = h

    @classmethod
    def with_cu_budget(cls, device, n_cus):
        """A context on a stream of the library's own that may use only `n_cus` compute units (vbnn_ctx_create_cu_budget):
        the compute stream of a data-parallel rank, beside RCCL's channels. torch work for it goes on `torch_stream`
        (a torch.cuda.ExternalStream over the same hipStream_t)."""
        self = cls.__new__(cls)
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        L.check(L.lib().vbnn_ctx_create_cu_budget(device, int(n_cus), C.byref(h)))
        self.h = h
        sp, n = C.c_void_p(), C.c_int()
        L.check(L.lib().vbnn_ctx_stream(h, C.byref(sp), C.byref(n)))
        self.cu_budget = n.value
        self.torch_stream = torch.cuda.ExternalStream(sp.value, device=self.device)
        return self

    @classmethod
    def get(cls, device=None):
        if device is None:
            device = torch.cuda.current_device()
        if isinstance(device, torch.device):
            device = device.index or 0
        if device not in cls._cache:
            cls._cache[device] = Context(device)
        return cls._cache[device]

    def sync(self):
        L.check(L.lib().vbnn_sync(self.h))


def _opt_get(opt, key, default=None):
    if opt is None:
        return default
    if isinstance(opt, dict):
        return opt.get(key, default)
    return getattr(opt, key, default)


class _Packed:
    """A GEMM-ready operand: rows x pad(cols) of dtype T, zero-initialised (pads stay zero)."""

    def __init__(self, rows, cols, tdtype, device, storage=None):
        """storage: a flat fp32 device tensor to place the operand IN (the peer-mapped arena of the p2p exchange: what the
        sharded-update exchange all-gathers must live there) instead of an allocation of its own; zeroed either way."""
        self.rows, self.cols = rows, cols
        self.ld = L.pad_ld(cols)
        if storage is None:
            self.t = torch.zeros((rows, self.ld), dtype=tdtype, device=device)
        else:
            n = rows * self.ld * torch.empty(0, dtype=tdtype).element_size()
            assert n % 4 == 0 and storage.dtype == torch.float32 and storage.numel() * 4 >= n and storage.is_contiguous()
            self.t = storage[:n // 4].view(tdtype).view(rows, self.ld)
            self.t.zero_()

    @staticmethod
    def floats(rows, cols, tdtype):
        """fp32 words an operand of this shape occupies (rounded up to 4: 16-byte aligned regions)."""
        n = rows * L.pad_ld(cols) * torch.empty(0, dtype=tdtype).element_size()
        return (n // 4 + 3) // 4 * 4

    @property
    def ptr(self):
        return C.c_void_p(self.t.data_ptr())


def pack(ctx, dtype, func, src, src2, want, want_T):
    """vbnn_pack front-end: returns (dst, dstT) packed buffers for a 2-D fp32 device tensor."""
    code, tdt = _DT[dtype]
    rows, cols = src.shape
    assert src.dtype == torch.float32 and src.stride(1) == 1
    dst = _Packed(rows, cols, tdt, src.device) if want is True else want
    dstT = _Packed(cols, rows, tdt, src.device) if want_T is True else want_T
    L.check(L.lib().vbnn_pack(ctx.h, code, func, _p(src), _p(src2), src.stride(0), rows, cols,
                              dst.ptr if dst else None, dst.ld if dst else 0,
                              dstT.ptr if dstT else None, dstT.ld if dstT else 0))
    return dst, dstT


# ------------------------------------------------------------------------------------------------
class Module:
    def __init__(self):
        self.output = None
        self.gradInput = None

    def forward(self, input):
        return self.updateOutput(input)

    def backward(self, input, gradOutput, scale=1.0):
        """nn.Module:backward [recalled]: updateGradInput, then accGradParameters."""
        self.updateGradInput(input, gradOutput)
        self.accGradParameters(input, gradOutput, scale)
        return self.gradInput

    def updateOutput(self, input):
        raise NotImplementedError

    def updateGradInput(self, input, gradOutput):
        raise NotImplementedError

    def accGradParameters(self, input, gradOutput, scale=1.0):
        pass

    def parameters(self):
        return [], []

    def zeroGradParameters(self):
        for g in self.parameters()[1]:
            g.zero_()


class Reshape(Module):
    """nn.Reshape(size) (mlp.lua:12): batch x ... -> batch x size."""

    def __init__(self, *size):
        super().__init__()
        self.size = int(math.prod(size))

    def updateOutput(self, input):
        self.output = input.reshape(-1, self.size)
        return self.output

    def updateGradInput(self, input, gradOutput):
        self.gradInput = gradOutput.reshape(input.shape)
        return self.gradInput


class ReLU(Module):
    """nn.ReLU (mlp.lua:19,27)."""

    def updateOutput(self, input):
        ctx = Context.get(input.device)
        if self.output is None or self.output.shape != input.shape:
            self.output = torch.empty_like(input)
        L.check(L.lib().vbnn_relu_forward(ctx.h, _p(input), _p(self.output), input.numel()))
        return self.output

    def updateGradInput(self, input, gradOutput):
        ctx = Context.get(input.device)
        if self.gradInput is None or self.gradInput.shape != input.shape:
            self.gradInput = torch.empty_like(input)
        L.check(L.lib().vbnn_relu_backward(ctx.h, _p(input), _p(gradOutput), _p(self.gradInput), input.numel()))
        return self.gradInput


class LogSoftMax(Module):
    """nn.LogSoftMax (mlp.lua:30)."""

    def updateOutput(self, input):
        ctx = Context.get(input.device)
        N, Cn = input.shape
        if self.output is None or self.output.shape != input.shape:
            self.output = torch.empty_like(input)
        dummy = getattr(self, "_dummy_t", None)
        if dummy is None or dummy.numel() < N:
            dummy = self._dummy_t = torch.zeros(N, dtype=torch.int32, device=input.device)
        L.check(L.lib().vbnn_logsoftmax_nll(ctx.h, _p(input), input.stride(0), _p(dummy), N, Cn, 1.0,
                                            _p(self.output), None, None, None))
        return self.output

    def updateGradInput(self, input, gradOutput):
        ctx = Context.get(input.device)
        N, Cn = input.shape
        if self.gradInput is None or self.gradInput.shape != input.shape:
            self.gradInput = torch.empty_like(input)
        L.check(L.lib().vbnn_logsoftmax_backward(ctx.h, _p(self.output), _p(gradOutput.contiguous()),
                                                 _p(self.gradInput), N, Cn))
        return self.gradInput


class ClassNLLCriterion:
    """nn.ClassNLLCriterion, sizeAverage = true (mlp.lua:32). Targets: int32, 0-based."""

    def __init__(self):
        self.sizeAverage = True
        self.gradInput = None
        self.output = None
        self._acc = None

    def forward(self, input, target):
        ctx = Context.get(input.device)
        N, Cn = input.shape
        if self._acc is None:
            self._acc = torch.zeros(2, dtype=torch.float64, device=input.device)
            self._corr = torch.zeros(1, dtype=torch.int32, device=input.device)
        self._acc.zero_()
        self._corr.zero_()
        inv_n = 1.0 / N if self.sizeAverage else 1.0
        L.check(L.lib().vbnn_nll_forward(ctx.h, _p(input), input.stride(0), _p(target), N, Cn, inv_n,
                                         _p(self._acc), _p(self._corr)))
        self.output = float(self._acc[0].item())
        self.correct = int(self._corr[0].item())
        return self.output

    def backward(self, input, target):
        ctx = Context.get(input.device)
        N, Cn = input.shape
        if self.gradInput is None or self.gradInput.shape != input.shape:
            self.gradInput = torch.empty_like(input)
        inv_n = 1.0 / N if self.sizeAverage else 1.0
        L.check(L.lib().vbnn_nll_backward(ctx.h, _p(target), N, Cn, inv_n, _p(self.gradInput)))
        return self.gradInput


class Sequential(Module):
    """nn.Sequential [recalled]: forward chains updateOutput; backward walks the modules in
    reverse, each with the previous module's output as its input."""

    def __init__(self):
        super().__init__()
        self.modules = []

    def add(self, m):
        self.modules.append(m)
        return self

    def get(self, i):
        return self.modules[i - 1]          # 1-based, as the Lua call sites (mlp.lua:48,65)

    def updateOutput(self, input):
        cur = input
        for m in self.modules:
            cur = m.updateOutput(cur)
        self.output = cur
        return cur

    def backward(self, input, gradOutput, scale=1.0):
        g = gradOutput
        for k in range(len(self.modules) - 1, -1, -1):
            inp = input if k == 0 else self.modules[k - 1].output
            g = self.modules[k].backward(inp, g, scale)
        self.gradInput = g
        return g

    def parameters(self):
        ps, gs = [], []
        for m in self.modules:
            p, g = m.parameters()
            ps += p
            gs += g
        return ps, gs


# ------------------------------------------------------------------------------------------------
class Linear(Module):
    """nn.Linear(inputSize, outputSize) on the library's GEMM kernels (the final layer of
    mlp.lua:29, and the parent class of VBLinear, VBLinear.lua:7)."""

    def __init__(self, inputSize, outputSize, opt=None, device=None):
        super().__init__()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.ctx = Context.get(self.device)
        self.inputSize, self.outputSize = inputSize, outputSize
        self.dtype = _opt_get(opt, "dtype", "f32")
        self.seed = int(_opt_get(opt, "seed", 3))          # config.lua:40
        self.layer_id = 0
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)
        self.weight, self.bias = z(outputSize, inputSize), z(outputSize)
        self.gradWeight, self.gradBias = z(outputSize, inputSize), z(outputSize)
        self._pk = {}
        self._g_fresh = False
        self.reset()

    # nn.Linear:reset draws U(-1/sqrt(I), 1/sqrt(I)) [recalled]; here N(0, 1/(3I)) (same variance)
    # from the Philox HEINIT stream. mlp.lua:47-55 overwrites it with the He rule anyway.
    def reset(self, stdv=None):
        stdv = stdv if stdv is not None else 1.0 / math.sqrt(3.0 * self.inputSize)
        fill_normal(self.weight, self.seed, L.STREAM_HEINIT, self.layer_id, 0, scale=stdv)
        self.bias.zero_()

    def parameters(self):
        return [self.weight, self.bias], [self.gradWeight, self.gradBias]

    # -- packed-operand bookkeeping
    def _buf(self, key, rows, cols):
        b = self._pk.get(key)
        if b is None or b.rows != rows or b.cols != cols:
            b = self._pk[key] = _Packed(rows, cols, _DT[self.dtype][1], self.device)
        return b

    def _pack_input(self, x, square):
        N, I = x.shape
        pack(self.ctx, self.dtype, L.PACK_COPY, x, None, self._buf("x", N, I), self._buf("xT", I, N))
        if square:
            pack(self.ctx, self.dtype, L.PACK_SQUARE, x, None, self._buf("x2", N, I), self._buf("x2T", I, N))

    def _pack_grad(self, g, r):
        N, O = g.shape
        pack(self.ctx, self.dtype, L.PACK_COPY, g, None, self._buf("g", N, O), self._buf("gT", O, N))
        if r is not None:
            pack(self.ctx, self.dtype, L.PACK_MUL, g, r, self._buf("gv", N, O), self._buf("gvT", O, N))
        self._g_fresh = True

    def _out(self, name, shape):
        t = getattr(self, name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.empty(shape, dtype=torch.float32, device=self.device)
            setattr(self, name, t)
        return t

    def _check_input(self, x):
        if x.dim() != 2 or x.shape[1] != self.inputSize:
            # VBLinear.lua:114 (torch.mm, gradOutput:t()) only works on 2-D input
            raise ValueError(f"input must be 2-D batch x {self.inputSize}, got {tuple(x.shape)}")
        if x.dtype != torch.float32 or not x.is_cuda:
            raise ValueError("input must be a float32 device tensor")
        return x if x.stride(1) == 1 else x.contiguous()

    # -- plain (weight-based) path: y = x W^T + b
    def _forward_plain(self, x):
        N = x.shape[0]
        O, I = self.outputSize, self.inputSize
        self._pack_input(x, False)
        W, WT = self._buf("w", O, I), self._buf("wT", I, O)
        pack(self.ctx, self.dtype, L.PACK_COPY, self.weight, None, W, WT)
        y = self._out("output", (N, O))
        a = L.FwdArgs(w=W.ptr, w2=None, x=self._pk["x"].ptr, x2=None, ld_w=W.ld, ld_x=self._pk["x"].ld,
                      N=N, I=I, O=O, bias=_p(self.bias), y=_p(y), ld_y=O)
        L.check(L.lib().vbnn_forward(self.ctx.h, _DT[self.dtype][0], C.byref(a)))
        return y

    def _grad_input_plain(self, x, g):
        N = x.shape[0]
        O, I = self.outputSize, self.inputSize
        self._pack_grad(g, None)
        gx = self._out("gradInput", (N, I))
        WT = self._pk["wT"]
        a = L.DxArgs(wT=WT.ptr, w2T=None, g=self._pk["g"].ptr, gv=None, ld_wT=WT.ld, ld_g=self._pk["g"].ld,
                     N=N, I=I, O=O, x=None, ld_x=0, gx=_p(gx), ld_gx=I)
        L.check(L.lib().vbnn_grad_input(self.ctx.h, _DT[self.dtype][0], C.byref(a)))
        return gx

    def _acc_bias(self, g, scale):
        N, O = g.shape
        L.check(L.lib().vbnn_acc_grad_bias(self.ctx.h, L.F32, _p(g), g.stride(0), N, O, float(scale), 1, _p(self.gradBias)))

    def updateOutput(self, input):
        return self._forward_plain(self._check_input(input))

    def updateGradInput(self, input, gradOutput):
        return self._grad_input_plain(self._check_input(input), gradOutput.contiguous())

    def accGradParameters(self, input, gradOutput, scale=1.0):
        g = gradOutput.contiguous()
        N = g.shape[0]
        if not self._g_fresh:
            self._pack_grad(g, None)
        self._g_fresh = False
        a = L.DwArgs(xT=self._pk["xT"].ptr, x2T=None, gT=self._pk["gT"].ptr, gvT=None, ld_n=self._pk["xT"].ld,
                     N=N, I=self.inputSize, O=self.outputSize, scale=float(scale), accumulate=1,
                     gradWeight=_p(self.gradWeight), gradSum=None)
        L.check(L.lib().vbnn_acc_grad_parameters(self.ctx.h, _DT[self.dtype][0], C.byref(a)))
        self._acc_bias(g, scale)


def fill_normal(t, seed, stream, layer, draw, row0=0, scale=1.0, hw=False, ctx=None):
    """t[r][c] <- scale * N(0,1) from the Philox contract (include/vbnn_philox.h); hw=True: the form the bf16 forward
    draws (the same Philox words, Box-Muller by the hardware's log2 / sqrt / sin / cos: vbnn_fill_normal_hw).
    ctx: the library context (hence the stream) to launch on; default the device's shared context."""
    ctx = ctx or Context.get(t.device)
    rows, cols = (t.shape if t.dim() == 2 else (1, t.numel()))
    ld = t.stride(0) if t.dim() == 2 else cols
    fn = L.lib().vbnn_fill_normal_hw if hw else L.lib().vbnn_fill_normal
    L.check(fn(ctx.h, _p(t), rows, cols, ld, int(seed), stream, layer, draw, row0, float(scale)))
    return t


class VBLinear(Linear):
    """nn.VBLinear(inputSize, outputSize, opt) -- VBLinear.lua, on MI355X.

    opt fields read (same as the reference, VBLinear.lua:12-33,91-102): var_init, msr_init,
    mu_init, B, S. New fields: `mode` = 'wn' (the reference's weight-noise sampling, sample-exact
    against the oracle) | 'lrt' (local reparameterisation: mu-GEMM + sigma^2-GEMM on squared
    inputs + per-activation Philox noise; the throughput path); `dtype` = 'f32' | 'bf16';
    `seed`.
    """

    def __init__(self, inputSize, outputSize, opt, device=None):
        super().__init__(inputSize, outputSize, opt, device)
        self.opt = opt
        self.mode = _opt_get(opt, "mode", "wn")
        if self.mode not in ("wn", "lrt"):
            raise ValueError("opt.mode must be 'wn' or 'lrt'")
        self.var_init = _opt_get(opt, "var_init")                       # VBLinear.lua:12
        self.bias.zero_()                                               # :13
        if _opt_get(opt, "msr_init"):
            self.var_init = 2.0 / inputSize                             # :14-16
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)
        self.lvars = torch.full((outputSize, inputSize), math.log(self.var_init), dtype=torch.float32,
                                device=self.device)                     # :18
        self.gradSum = z(outputSize, inputSize)                         # :20
        self.W = outputSize * inputSize                                 # :21
        self.means = z(outputSize, inputSize)                           # :22-23
        if _opt_get(opt, "mu_init", 0) != 0:
            self._init_means()
        self.e = None                       # materialised only on request (keep_e); kernels regenerate it
        self.keep_e = bool(_opt_get(opt, "keep_e", False))
        self.stats = torch.zeros(4, dtype=torch.float64, device=self.device)
        self._lc = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.draw = 0
        self.row0 = 0                       # global index of this rank's first minibatch row
        self._map = False
        self.r = None
        self.vars = z(outputSize, inputSize)
        self.stdv = z(outputSize, inputSize)
        self.mu_sqe = z(outputSize, inputSize)
        self.compute_prior()                                            # :46

    def _init_means(self):
        # VBLinear.lua:25-28: means ~ N(0, sqrt(var_init))
        fill_normal(self.means, self.seed, L.STREAM_INIT, self.layer_id, 0, scale=math.sqrt(self.var_init))

    # aliases the north_star uses
    @property
    def mu(self):
        return self.means

    @property
    def lv(self):
        return self.lvars

    def set_layer_id(self, layer_id):
        self.layer_id = layer_id
        if _opt_get(self.opt, "mu_init", 0) != 0:
            self._init_means()
            self.compute_prior()

    # -- VBLinear.lua:77-88
    def compute_prior(self):
        L.check(L.lib().vbnn_compute_prior(self.ctx.h, _p(self.means), _p(self.lvars), self.W, _p(self.vars),
                                           _p(self.stdv), _p(self.mu_sqe), _p(self.stats)))
        self.mu_hat = 0
        self.var_hat = float(self.stats[2].item())
        # the cached statistics calc_lc / compute_*grads use until the next compute_prior
        self._stats_cached = self.stats.clone()
        return self.mu_hat, self.var_hat

    # -- VBLinear.lua:49-64
    def sample(self, opt=None):
        self.draw += 1
        self._map = False
        if self.mode == "wn":
            if self.keep_e and self.e is None:
                self.e = torch.empty_like(self.weight)
            L.check(L.lib().vbnn_wn_sample(self.ctx.h, _p(self.means), _p(self.stdv), None, _p(self.weight),
                                           _p(self.e) if self.keep_e else None, self.outputSize, self.inputSize,
                                           self.seed, self.layer_id, self.draw))

    # -- VBLinear.lua:105-107
    def clamp_to_map(self):
        self.weight.copy_(self.means)
        self._map = True

    # -- VBLinear.lua:120-122
    def resetAcc(self, opt=None):
        self.gradSum.zero_()

    def _lrt_active(self):
        return self.mode == "lrt" and not self._map

    # -- inherited nn.Linear:updateOutput (VBLinear.lua:7) / LRT forward
    def updateOutput(self, input):
        x = self._check_input(input)
        if not self._lrt_active():
            return self._forward_plain(x)
        N = x.shape[0]
        O, I = self.outputSize, self.inputSize
        self._pack_input(x, True)
        MU, MUT = self._buf("mu", O, I), self._buf("muT", I, O)
        VA, VAT = self._buf("var", O, I), self._buf("varT", I, O)
        pack(self.ctx, self.dtype, L.PACK_COPY, self.means, None, MU, MUT)
        pack(self.ctx, self.dtype, L.PACK_EXP, self.lvars, None, VA, VAT)
        y = self._out("output", (N, O))
        r = self._out("r", (N, O))
        a = L.FwdArgs(w=MU.ptr, w2=VA.ptr, x=self._pk["x"].ptr, x2=self._pk["x2"].ptr, ld_w=MU.ld,
                      ld_x=self._pk["x"].ld, N=N, I=I, O=O, bias=_p(self.bias), seed=self.seed,
                      layer=self.layer_id, draw=self.draw, row0=self.row0, y=_p(y), ld_y=O, r=_p(r), ld_r=O)
        L.check(L.lib().vbnn_forward(self.ctx.h, _DT[self.dtype][0], C.byref(a)))
        return y

    # -- inherited nn.Linear:updateGradInput (stub at VBLinear.lua:109-110) / LRT backward
    def updateGradInput(self, input, gradOutput):
        x = self._check_input(input)
        g = gradOutput.contiguous()
        if not self._lrt_active():
            return self._grad_input_plain(x, g)
        N = x.shape[0]
        O, I = self.outputSize, self.inputSize
        self._pack_grad(g, self.r)
        gx = self._out("gradInput", (N, I))
        MUT, VAT = self._pk["muT"], self._pk["varT"]
        a = L.DxArgs(wT=MUT.ptr, w2T=VAT.ptr, g=self._pk["g"].ptr, gv=self._pk["gv"].ptr, ld_wT=MUT.ld,
                     ld_g=self._pk["g"].ld, N=N, I=I, O=O, x=self._pk["x"].ptr, ld_x=self._pk["x"].ld,
                     gx=_p(gx), ld_gx=I)
        L.check(L.lib().vbnn_grad_input(self.ctx.h, _DT[self.dtype][0], C.byref(a)))
        return gx

    # -- VBLinear.lua:112-118
    def accGradParameters(self, input, gradOutput, scale=1.0):
        g = gradOutput.contiguous()
        N = g.shape[0]
        lrt = self._lrt_active()
        if not self._g_fresh:
            self._pack_grad(g, self.r if lrt else None)
        self._g_fresh = False
        pk = self._pk
        a = L.DwArgs(xT=pk["xT"].ptr, x2T=pk["x2T"].ptr if lrt else None, gT=pk["gT"].ptr,
                     gvT=pk["gvT"].ptr if lrt else None, ld_n=pk["xT"].ld, N=N, I=self.inputSize,
                     O=self.outputSize, scale=float(scale), accumulate=1, gradWeight=_p(self.gradWeight),
                     gradSum=None if self._map else _p(self.gradSum), seed=self.seed, layer=self.layer_id,
                     draw=self.draw, lvars=_p(self.lvars))
        L.check(L.lib().vbnn_acc_grad_parameters(self.ctx.h, _DT[self.dtype][0], C.byref(a)))
        self._acc_bias(g, scale)

    # -- VBLinear.lua:90-93: returns (gradWeight / S  [in place], lcg)
    def compute_mugrads(self, opt):
        lcg = torch.empty_like(self.means)
        L.check(L.lib().vbnn_compute_mugrads(self.ctx.h, _p(self.means), _p(self._stats_cached), float(opt["B"]),
                                             float(opt["S"]), _p(self.gradWeight), _p(lcg), self.W))
        return self.gradWeight, lcg

    # -- VBLinear.lua:95-98: returns (gradSum / (2S) . stdv  [in place], lcg)
    def compute_vargrads(self, opt):
        lcg = torch.empty_like(self.means)
        L.check(L.lib().vbnn_compute_vargrads(self.ctx.h, None, _p(self.vars), _p(self.stdv), _p(self._stats_cached),
                                              float(opt["B"]), float(opt["S"]), _p(self.gradSum), _p(lcg), self.W))
        return self.gradSum, lcg

    # -- VBLinear.lua:99-103. Returns a 1-element tensor holding the sum (satisfies the `:sum()` of
    # mlp.lua:112); `elementwise=True` also materialises the O x I tensor the Lua method returns.
    # Uses the statistics cached by the LAST compute_prior, as the reference does (staleness):
    # vars / mu_sqe are recomputed from the current means/lvars only if `fresh=True`.
    def calc_lc(self, opt, elementwise=False):
        lc_elem = torch.empty_like(self.means) if elementwise else None
        L.check(L.lib().vbnn_calc_lc(self.ctx.h, None, None, _p(self.vars), _p(self.mu_sqe), _p(self._stats_cached),
                                     float(opt["B"]), _p(lc_elem), _p(self._lc), self.W))
        return (self._lc, lc_elem) if elementwise else self._lc

    # -- VBLinear.lua:124-166: SGD on bias, compute_prior, likelihood + KL gradients, optim.adam on means (meanState)
    # and lvars (varState). Returns (mu_normratio, var_normratio), the two ratios the reference logs (:139,144);
    # its other twelve Log:add statistics (:149-164) are host-side logging and are not reproduced.
    def update(self, opt):
        lib, h = L.lib(), self.ctx.h
        st = self.__dict__.setdefault("_opt_state", {})
        L.check(lib.vbnn_sgd_step(h, _p(self.bias), _p(self.gradBias), self.bias.numel(),
                                  float(opt["state"]["learningRate"])))                               # :125-128
        self.compute_prior()                                                                           # :130
        mleg, mlcg = self.compute_mugrads(opt)                                                         # :131
        vleg, vlcg = self.compute_vargrads(opt)                                                        # :133
        ratios = []
        for key, x, g1, g2, cfg in (("mean", self.means, mleg, mlcg, opt["meanState"]),
                                    ("var", self.lvars, vleg, vlcg, opt["varState"])):
            s = st.setdefault(key, {"t": 0})
            if "m" not in s:
                s["m"], s["v"] = torch.zeros_like(x), torch.zeros_like(x)
                s["norms"] = torch.zeros(2, dtype=torch.float64, device=self.device)
            s["t"] += 1
            L.check(lib.vbnn_adam_step(h, _p(x), _p(g1), _p(g2), _p(s["m"]), _p(s["v"]), x.numel(),
                                       float(cfg["learningRate"]), float(cfg.get("beta1", 0.9)),
                                       float(cfg.get("beta2", 0.999)), float(cfg.get("epsilon", 1e-8)),
                                       float(cfg.get("lambda", 1.0)), s["t"], _p(s["norms"])))         # :135-143
            n = s["norms"].tolist()
            ratios.append(n[0] / n[1] if n[1] > 0 else float("inf"))
        return tuple(ratios)
