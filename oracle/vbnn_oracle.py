"""ctypes front-end of the CPU oracle (oracle/vbnn_oracle.c) + reference-shaped classes.

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by vbnn_amd/. PARITY UNPINNED (see vbnn_oracle.c header).

`OracleVBLinear` and `OracleMLP` keep the reference's names and call order
(VBLinear.lua, mlp.lua) so a parity test reads like the Lua it restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvbnn_oracle.so")

STREAM_EPS, STREAM_ZETA, STREAM_INIT, STREAM_DATA, STREAM_HEINIT = 1, 2, 3, 4, 5


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "vbnn_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "vbnn_philox.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libvbnn_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if os.path.exists(os.path.join(_HERE, "vbnn_oracle.c")):
            build()                 # no-op when the library is newer than its sources
        _lib = C.CDLL(_LIB_PATH)
        _lib.vbo_calc_lc.restype = C.c_double
        _lib.vbo_nll_forward.restype = C.c_double
        _lib.vbo_mse_forward.restype = C.c_double
        _lib.vbo_get_accuracy.restype = C.c_double
        _lib.vbo_det_logf.restype = C.c_float
        _lib.vbo_det_logf.argtypes = [C.c_float]
    return _lib


def _f(a):
    assert a is None or (a.dtype == np.float32 and a.flags["C_CONTIGUOUS"])
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i32(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


_i64 = C.c_int64


# ---------------------------------------------------------------- RNG
def philox_raw(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().vbo_philox_raw(c, k, o)
    return [int(v) for v in o]


def fill_normal(rows, cols, seed, stream, layer, draw, row0=0):
    out = np.empty((rows, cols), np.float32)
    lib().vbo_fill_normal(_f(out), _i64(rows), _i64(cols), _i64(cols), C.c_uint64(seed),
                          C.c_uint32(stream), C.c_uint32(layer), C.c_uint32(draw), _i64(row0))
    return out


def det_logf(x):
    return float(lib().vbo_det_logf(C.c_float(x)))


def det_sincos2pi(k):
    c, s = C.c_float(), C.c_float()
    lib().vbo_det_sincos2pi(C.c_uint32(k), C.byref(c), C.byref(s))
    return c.value, s.value


# ---------------------------------------------------------------- functional layer ops
def compute_prior(means, lvars):
    W = means.size
    vars_, stdv, mu_sqe = (np.empty_like(means) for _ in range(3))
    vh = C.c_double()
    lib().vbo_compute_prior(_f(means), _f(lvars), _i64(W), _f(vars_), _f(stdv), _f(mu_sqe), C.byref(vh))
    return vars_, stdv, mu_sqe, vh.value


def sample(means, stdv, e):
    w = np.empty_like(means)
    lib().vbo_sample(_f(means), _f(stdv), _f(e), _f(w), _i64(means.size))
    return w


def linear_forward(x, weight, bias):
    N, I = x.shape
    O = weight.shape[0]
    y = np.empty((N, O), np.float32)
    lib().vbo_linear_forward(_f(x), _f(weight), _f(bias), _f(y), _i64(N), _i64(I), _i64(O))
    return y


def linear_grad_input(g, weight):
    N, O = g.shape
    I = weight.shape[1]
    gx = np.empty((N, I), np.float32)
    lib().vbo_linear_grad_input(_f(g), _f(weight), _f(gx), _i64(N), _i64(I), _i64(O))
    return gx


def acc_grad_parameters(x, g, e, scale, gradWeight, gradBias, gradSum):
    N, I = x.shape
    O = g.shape[1]
    lib().vbo_acc_grad_parameters(_f(x), _f(g), _f(e), C.c_float(scale), _f(gradWeight), _f(gradBias),
                                  _f(gradSum), _i64(N), _i64(I), _i64(O))


def compute_mugrads(means, var_hat, B, S, gradWeight):
    lcg = np.empty_like(means)
    lib().vbo_compute_mugrads(_f(means), C.c_double(var_hat), C.c_float(B), C.c_float(S), _f(gradWeight),
                              _f(lcg), _i64(means.size))
    return gradWeight, lcg


def compute_vargrads(vars_, stdv, var_hat, B, S, gradSum):
    lcg = np.empty_like(vars_)
    lib().vbo_compute_vargrads(_f(vars_), _f(stdv), C.c_double(var_hat), C.c_float(B), C.c_float(S),
                               _f(gradSum), _f(lcg), _i64(vars_.size))
    return gradSum, lcg


def calc_lc(vars_, mu_sqe, var_hat, B, want_elem=False):
    lc = np.empty_like(vars_) if want_elem else None
    s = lib().vbo_calc_lc(_f(vars_), _f(mu_sqe), C.c_double(var_hat), C.c_float(B), _f(lc), _i64(vars_.size))
    return (float(s), lc) if want_elem else float(s)


def lrt_forward(x, means, lvars, bias, zeta):
    N, I = x.shape
    O = means.shape[0]
    y, r, v = (np.empty((N, O), np.float32) for _ in range(3))
    lib().vbo_lrt_forward(_f(x), _f(means), _f(lvars), _f(bias), _f(zeta), _f(y), _f(r), _f(v),
                          _i64(N), _i64(I), _i64(O))
    return y, r, v


def lrt_backward(x, g, r, means, lvars, scale, gradWeight, gradBias, gradSum, want_gx=True):
    N, I = x.shape
    O = means.shape[0]
    gx = np.empty((N, I), np.float32) if want_gx else None
    lib().vbo_lrt_backward(_f(x), _f(g), _f(r), _f(means), _f(lvars), C.c_float(scale), _f(gradWeight),
                           _f(gradBias), _f(gradSum), _f(gx), _i64(N), _i64(I), _i64(O))
    return gx


def adam_step(x, g, m, v, lr, b1, b2, eps, t):
    """In-place optim.adam step; returns the applied update (see vbo_adam_step)."""
    upd = np.empty_like(x)
    lib().vbo_adam_step(_f(x), _f(np.ascontiguousarray(g, np.float32)), _f(m), _f(v), _i64(x.size), C.c_float(lr),
                        C.c_float(b1), C.c_float(b2), C.c_float(eps), _i64(t), _f(upd))
    return upd


def sgd_step(x, g, lr):
    lib().vbo_sgd_step(_f(x), _f(np.ascontiguousarray(g, np.float32)), _i64(x.size), C.c_float(lr))


def relu_forward(x):
    y = np.empty_like(x)
    lib().vbo_relu_forward(_f(x), _f(y), _i64(x.size))
    return y


def relu_backward(x, g):
    gx = np.empty_like(x)
    lib().vbo_relu_backward(_f(x), _f(g), _f(gx), _i64(x.size))
    return gx


def logsoftmax_forward(x):
    y = np.empty_like(x)
    lib().vbo_logsoftmax_forward(_f(x), _f(y), _i64(x.shape[0]), _i64(x.shape[1]))
    return y


def logsoftmax_backward(out, g):
    gx = np.empty_like(out)
    lib().vbo_logsoftmax_backward(_f(out), _f(g), _f(gx), _i64(out.shape[0]), _i64(out.shape[1]))
    return gx


def nll_forward(out, target):
    return float(lib().vbo_nll_forward(_f(out), _i32(target), _i64(out.shape[0]), _i64(out.shape[1])))


def nll_backward(target, N, Cn):
    g = np.empty((N, Cn), np.float32)
    lib().vbo_nll_backward(_i32(target), _f(g), _i64(N), _i64(Cn))
    return g


def mse_forward(y, target):
    return float(lib().vbo_mse_forward(_f(y), _f(target), _i64(y.shape[0]), _i64(y.shape[1])))


def mse_backward(y, target):
    g = np.empty_like(y)
    lib().vbo_mse_backward(_f(y), _f(target), _f(g), _i64(y.shape[0]), _i64(y.shape[1]))
    return g


def get_accuracy(out, target):
    return float(lib().vbo_get_accuracy(_f(out), _i32(target), _i64(out.shape[0]), _i64(out.shape[1])))


# ---------------------------------------------------------------- reference-shaped classes
class OracleVBLinear:
    """nn.VBLinear restated (VBLinear.lua). mode 'wn' is the reference's weight-noise
    form; mode 'lrt' is the local-reparameterisation form of the same layer."""

    def __init__(self, inputSize, outputSize, opt, layer_id=0):
        self.opt = opt
        self.I, self.O = inputSize, outputSize
        self.layer_id = layer_id
        self.mode = opt.get("mode", "wn")
        self.seed = opt.get("seed", 3)                                   # config.lua:40
        self.var_init = opt["var_init"]                                  # VBLinear.lua:12
        if opt.get("msr_init"):
            self.var_init = 2.0 / inputSize                              # :14-16
        self.weight = np.zeros((outputSize, inputSize), np.float32)
        self.bias = np.zeros(outputSize, np.float32)                     # :13
        self.gradWeight = np.zeros_like(self.weight)
        self.gradBias = np.zeros_like(self.bias)
        self.lvars = np.full((outputSize, inputSize), np.float32(np.log(self.var_init)), np.float32)  # :18
        self.gradSum = np.zeros_like(self.weight)                        # :20
        self.W = outputSize * inputSize                                  # :21
        if opt.get("mu_init", 0) == 0:
            self.means = np.zeros_like(self.weight)                      # :22-23
        else:
            std_init = np.float32(np.sqrt(self.var_init))                # :25
            self.means = fill_normal(outputSize, inputSize, self.seed, STREAM_INIT, layer_id, 0) * std_init
        self.e = np.zeros_like(self.weight)                              # :37
        self.draw = 0
        self.row0 = 0
        self.compute_prior()                                             # :46

    def compute_prior(self):
        self.vars, self.stdv, self.mu_sqe, self.var_hat = compute_prior(self.means, self.lvars)
        self.mu_hat = 0
        return self.mu_hat, self.var_hat

    def sample(self, opt=None):
        self.draw += 1
        if self.mode == "wn":
            self.e = fill_normal(self.O, self.I, self.seed, STREAM_EPS, self.layer_id, self.draw)
            self.weight = sample(self.means, self.stdv, self.e)

    def clamp_to_map(self):
        self.weight = self.means.copy()
        self._map = True

    def updateOutput(self, x):
        if self.mode == "wn" or getattr(self, "_map", False):
            self.output = linear_forward(x, self.weight, self.bias)
        else:
            zeta = fill_normal(x.shape[0], self.O, self.seed, STREAM_ZETA, self.layer_id, self.draw, self.row0)
            self.output, self.r, self.v = lrt_forward(x, self.means, self.lvars, self.bias, zeta)
        return self.output

    def backward(self, x, g, scale=1.0, want_gx=True):
        """updateGradInput + accGradParameters (nn.Module:backward order [recalled])."""
        if self.mode == "wn" or getattr(self, "_map", False):
            self.gradInput = linear_grad_input(g, self.weight) if want_gx else None
            acc_grad_parameters(x, g, self.e, scale, self.gradWeight, self.gradBias, self.gradSum)
        else:
            self.gradInput = lrt_backward(x, g, self.r, self.means, self.lvars, scale, self.gradWeight,
                                          self.gradBias, self.gradSum, want_gx)
        return self.gradInput

    def resetAcc(self, opt=None):
        self.gradSum[:] = 0

    def compute_mugrads(self, opt):
        return compute_mugrads(self.means, self.var_hat, opt["B"], opt["S"], self.gradWeight)

    def compute_vargrads(self, opt):
        return compute_vargrads(self.vars, self.stdv, self.var_hat, opt["B"], opt["S"], self.gradSum)

    def calc_lc(self, opt):
        return calc_lc(self.vars, self.mu_sqe, self.var_hat, opt["B"])

    def update(self, opt):
        """VBLinear:update -- VBLinear.lua:124-166 (logging omitted). Returns (mu_normratio, var_normratio)."""
        st = self.__dict__.setdefault("_opt_state", {"mean": {"t": 0}, "var": {"t": 0}})
        sgd_step(self.bias, self.gradBias, opt["state"]["learningRate"])                     # :125-128
        self.compute_prior()                                                                  # :130
        mleg, mlcg = self.compute_mugrads(opt)                                                # :131
        mugrad = mleg + mlcg                                                                  # :132
        vleg, vlcg = self.compute_vargrads(opt)                                               # :133
        vgrad = vleg + vlcg                                                                   # :134
        ratios = []
        for x, g, key, cfg in ((self.means, mugrad, "mean", opt["meanState"]), (self.lvars, vgrad, "var", opt["varState"])):
            s = st[key]
            if "m" not in s:
                s["m"], s["v"] = np.zeros_like(x), np.zeros_like(x)
            s["t"] += 1
            upd = adam_step(x, g, s["m"], s["v"], cfg["learningRate"], cfg.get("beta1", 0.9), cfg.get("beta2", 0.999),
                            cfg.get("epsilon", 1e-8), s["t"])
            ratios.append(float(np.linalg.norm(upd.astype(np.float64)) / np.linalg.norm(x.astype(np.float64))))   # :139,144
        return tuple(ratios)


class OracleLinear:
    """plain nn.Linear (the final layer, mlp.lua:29)."""

    def __init__(self, inputSize, outputSize):
        self.weight = np.zeros((outputSize, inputSize), np.float32)
        self.bias = np.zeros(outputSize, np.float32)
        self.gradWeight = np.zeros_like(self.weight)
        self.gradBias = np.zeros_like(self.bias)

    def updateOutput(self, x):
        self.output = linear_forward(x, self.weight, self.bias)
        return self.output

    def backward(self, x, g, scale=1.0, want_gx=True):
        self.gradInput = linear_grad_input(g, self.weight) if want_gx else None
        acc_grad_parameters(x, g, None, scale, self.gradWeight, self.gradBias, None)
        return self.gradInput


class OracleMLP:
    """mlp.lua restated: Reshape -> [VBLinear -> ReLU]* -> Linear -> LogSoftMax + ClassNLL."""

    def __init__(self, opt):
        self.opt = opt
        sizes = [opt["input_size"]] + list(opt["hidden"])
        self.vb = [OracleVBLinear(sizes[i], sizes[i + 1], opt, layer_id=i) for i in range(len(opt["hidden"]))]
        self.last = OracleLinear(sizes[-1], opt["n_classes"])
        seed = opt.get("seed", 3)
        # mlp.lua:47-55: He init of every `weight`, bias zero. VB layers overwrite weight on sample().
        for li, m in enumerate(self.vb + [self.last]):
            O, I = m.weight.shape
            m.weight = fill_normal(O, I, seed, STREAM_HEINIT, li, 0) * np.float32(np.sqrt(2.0 / I))
            m.bias[:] = 0

    def resetGradients(self):                                            # mlp.lua:62-67
        for m in self.vb + [self.last]:
            m.gradWeight[:] = 0
            m.gradBias[:] = 0
        for m in self.vb:
            m.resetAcc(self.opt)

    def sample(self):                                                    # mlp.lua:69-74
        for m in self.vb:
            m._map = False
            m.sample(self.opt)

    def run(self, inputs, targets, masks=None):                          # mlp.lua:76-84
        """masks (tests only): per VB layer a boolean N x O array that REPLACES the ReLU's own `y > 0` in both
        directions -- the on/off pattern another implementation took -- so that a comparison against reduced-precision
        arithmetic measures rounding, not the discontinuity of the few units whose sign the rounding flipped."""
        acts = [inputs.reshape(inputs.shape[0], -1)]                     # nn.Reshape, mlp.lua:12
        pre = []
        for k, m in enumerate(self.vb):
            y = m.updateOutput(acts[-1])
            if masks is not None:
                pre.append(np.where(masks[k], np.float32(1), np.float32(-1)))     # relu_backward keys on pre > 0
                acts.append(np.where(masks[k], y, np.float32(0)).astype(np.float32))
                continue
            pre.append(y)
            acts.append(relu_forward(y))
        logits = self.last.updateOutput(acts[-1])
        if self.opt.get("criterion", "nll") == "mse":
            # BASELINE.json configs[4] (regression; not in the reference): no LogSoftMax, nn.MSECriterion on the outputs
            outputs = logits
            g = mse_backward(outputs, targets)
            g = self.last.backward(acts[-1], g)
            for k in range(len(self.vb) - 1, -1, -1):
                g = relu_backward(pre[k], g)
                g = self.vb[k].backward(acts[k], g, 1.0, want_gx=True)
            self.outputs = outputs
            return mse_forward(outputs, targets), 0.0
        outputs = logsoftmax_forward(logits)
        df_do = nll_backward(targets, outputs.shape[0], outputs.shape[1])  # criterion:backward, :78
        g = logsoftmax_backward(outputs, df_do)                            # model:backward, :79
        g = self.last.backward(acts[-1], g)
        for k in range(len(self.vb) - 1, -1, -1):
            g = relu_backward(pre[k], g)
            # nn.Sequential calls updateGradInput on every module, the first one included
            g = self.vb[k].backward(acts[k], g, 1.0, want_gx=True)
        self.outputs = outputs
        error = nll_forward(outputs, targets)                              # :80
        accuracy = get_accuracy(outputs, targets)                          # :82
        return error, accuracy

    def calc_lc(self, opt=None):                                         # mlp.lua:109-115
        opt = opt or self.opt
        return sum(m.calc_lc(opt) for m in self.vb)

    def update(self, opt=None):                                          # mlp.lua:117-142 (final-layer slice fixed, see vbnn_amd/mlp.py)
        opt = opt or self.opt
        lr = opt["state"]["learningRate"]
        sgd_step(self.last.weight, self.last.gradWeight, lr)
        sgd_step(self.last.bias, self.last.gradBias, lr)
        return [m.update(opt) for m in self.vb]
