"""NumPy/BLAS twin of the oracle (TEST INFRASTRUCTURE ONLY; parity unpinned, see vbnn_oracle.c).

Two uses:
  1. `ReferenceCpuMLP`: the reference's operation sequence executed LITERALLY on the host --
     per draw and per VB layer a host Gaussian fill of O x I values (VBLinear.lua:55), the
     `means + stdv . e` temporaries (:59-63), sgemm forward, sgemm gradInput, sgemm gradWeight, the
     second identical sgemm (:114), `gradSum += grad . e` (:115) -- with multi-threaded BLAS sgemm
     (OpenBLAS through NumPy), i.e. what `th main.lua` does with `opt.threads` BLAS threads
     (main.lua:142). bench.py times it as `cpu_baseline` (kind "port": Torch7 cannot run here).
     The Gaussian stream is NumPy's legacy RandomState (MT19937 + polar Box-Muller), the same
     algorithm family as randomkit [recalled].
  2. bf16 emulation + BLAS-speed LRT/WN math for parity checks at sizes where the scalar C
     oracle would take minutes.
"""
import os

import numpy as np


def bf16_round(a):
    """Round-to-nearest-even to bfloat16, returned as float32 (what v_cvt_pk_bf16_f32 does)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def blas_threads():
    try:
        from threadpoolctl import threadpool_info
        for info in threadpool_info():
            if info.get("user_api") == "blas":
                return int(info["num_threads"])
    except Exception:
        pass
    return os.cpu_count() or 1


def log_softmax(x):
    mx = x.max(axis=1, keepdims=True)
    lse = mx + np.log(np.exp((x - mx).astype(np.float64)).sum(axis=1, keepdims=True)).astype(np.float32)
    return (x - lse).astype(np.float32)


class ReferenceCpuMLP:
    """mlp.lua + VBLinear.lua (weight-noise form), op for op, on NumPy/BLAS."""

    def __init__(self, input_size, hidden, n_classes, var_init=1e-3, seed=3):
        self.rs = np.random.RandomState(seed)                         # config.lua:40 torch.manualSeed(3)
        sizes = [input_size] + list(hidden)
        self.layers = []
        for I, O in zip(sizes[:-1], sizes[1:]):
            lay = dict(I=I, O=O,
                       means=(self.rs.standard_normal((O, I)) * np.sqrt(2.0 / I)).astype(np.float32),
                       lvars=np.full((O, I), np.log(var_init), np.float32),
                       bias=np.zeros(O, np.float32), weight=np.zeros((O, I), np.float32),
                       gradWeight=np.zeros((O, I), np.float32), gradBias=np.zeros(O, np.float32),
                       gradSum=np.zeros((O, I), np.float32), e=np.zeros((O, I), np.float32))
            self.layers.append(lay)
            self.compute_prior(lay)
        H = sizes[-1]
        self.w3 = (self.rs.standard_normal((n_classes, H)) * np.sqrt(2.0 / H)).astype(np.float32)
        self.b3 = np.zeros(n_classes, np.float32)
        self.gw3 = np.zeros_like(self.w3)
        self.gb3 = np.zeros_like(self.b3)

    @staticmethod
    def compute_prior(lay):                                           # VBLinear.lua:77-88
        lay["vars"] = np.exp(lay["lvars"])
        lay["stdv"] = np.sqrt(lay["vars"])
        lay["mu_sqe"] = np.power(lay["means"] - 0, 2)
        lay["var_hat"] = (1.0 / lay["means"].size) * np.sum(lay["vars"] + lay["mu_sqe"], dtype=np.float64)

    def resetGradients(self):                                         # mlp.lua:62-67
        for lay in self.layers:
            lay["gradWeight"][:] = 0
            lay["gradBias"][:] = 0
            lay["gradSum"][:] = 0
        self.gw3[:] = 0
        self.gb3[:] = 0

    def sample(self):                                                 # mlp.lua:69-74 -> VBLinear.lua:49-64
        for lay in self.layers:
            lay["e"] = self.rs.standard_normal(lay["e"].shape).astype(np.float32)   # :55 (host, one thread)
            w = lay["means"] + lay["stdv"] * lay["e"]                                # :59 (two temporaries)
            lay["weight"][:] = w                                                     # :63

    def run(self, x, t):                                              # mlp.lua:76-84
        N = x.shape[0]
        acts, pre = [x], []
        for lay in self.layers:                                       # model:forward (:77)
            y = acts[-1] @ lay["weight"].T                            # nn.Linear:updateOutput (addmm)
            y += lay["bias"]                                          #                         (addr)
            pre.append(y)
            acts.append(np.maximum(y, 0))                             # nn.ReLU
        logits = acts[-1] @ self.w3.T + self.b3
        out = log_softmax(logits)
        df_do = np.zeros_like(out)                                    # criterion:backward (:78)
        df_do[np.arange(N), t] = -1.0 / N
        g = df_do - np.exp(out) * df_do.sum(axis=1, keepdims=True)    # LogSoftMax:updateGradInput (:79)
        gx = g @ self.w3                                              # final Linear
        self.gw3 += g.T @ acts[-1]
        self.gb3 += g.sum(axis=0)
        g = gx
        for k in range(len(self.layers) - 1, -1, -1):
            lay = self.layers[k]
            g = np.where(pre[k] > 0, g, 0).astype(np.float32)         # nn.ReLU:updateGradInput
            gx = g @ lay["weight"]                                    # updateGradInput (every layer, also the first)
            lay["gradWeight"] += g.T @ acts[k]                        # parent.accGradParameters (VBLinear.lua:113)
            lay["gradBias"] += g.sum(axis=0)
            grad = g.T @ acts[k]                                      # the same GEMM again (:114)
            lay["gradSum"] += grad * lay["e"]                         # (:115)
            g = gx
        err = -out[np.arange(N), t].mean()                            # criterion:forward (:80)
        acc = 100.0 * (out.argmax(axis=1) == t).mean()                # utils.get_accuracy (:82)
        return float(err), float(acc)

    def step(self, x, t):
        """One benchmark step (S = 1): resetGradients + sample + run (main.lua:28-37)."""
        self.resetGradients()
        self.sample()
        return self.run(x, t)


# ---- the fused LRT step with explicit rounding points -------------------------------------------------
def emulate_lrt_step(layers, w3, b3, x, t, zetas, rnd=lambda a: a, S=1.0, B=1e6, inv_n=None, kl_shadows=False, criterion="nll", kl_scale=1.0):
    """One LRT draw of the whole MLP in float64 with the engine's rounding points made explicit:
    `rnd` is applied wherever the HIP path stores a GEMM operand (x, x.x, mu, sigma^2, relu(y), its
    square, g, g.r and the final Linear's operands). rnd = identity gives the exact fp32-free reference
    (cross-checks the C oracle); rnd = bf16_round gives what the bf16 MFMA path computes up to fp32
    accumulation order. layers: list of dicts(means, lvars, bias); zetas: per-layer N x O normals.
    kl_shadows: the fused total gradients read mu and sigma^2 from the ROUNDED operand shadows (vbnn_dw_args.mu_s / var_s).
    kl_scale: vbnn_dw_args.kl_scale -- 0 for an engine whose arena holds the likelihood parts alone (opt.kl_in_update).
    Returns loss, per-layer dicts(gradWeight, gradSum, gradBias, grad_mu, grad_lv), gw3, gb3."""
    f8 = np.float64
    N = x.shape[0]
    inv_n = 1.0 / N if inv_n is None else inv_n
    xs, x2s, rs = [rnd(x).astype(f8)], [rnd(rnd(x) * rnd(x)).astype(f8)], []
    for lay, z in zip(layers, zetas):
        mu, var = rnd(lay["means"]).astype(f8), rnd(np.exp(lay["lvars"])).astype(f8)
        m = xs[-1] @ mu.T + lay["bias"].astype(f8)
        v = x2s[-1] @ var.T
        y = (m + np.sqrt(v) * z).astype(np.float32)                 # the accumulators are fp32
        with np.errstate(divide="ignore", invalid="ignore"):
            r = np.where(v > 0, z / (2 * np.sqrt(v)), 0.0).astype(np.float32)
        h = rnd(np.maximum(y, 0))
        rs.append(rnd(r))                                           # the engine stores r in the operand type
        xs.append(h.astype(f8))
        x2s.append(rnd(h * h).astype(f8))
    logits = (xs[-1] @ rnd(w3).astype(f8).T + b3).astype(np.float32)
    if criterion == "mse":                      # nn.MSECriterion over N x D outputs; inv_n = 1 / (global rows)
        D = logits.shape[1]
        diff = logits.astype(f8) - t.astype(f8)
        loss = float((diff * diff).sum() * inv_n / D)
        g3 = (2.0 * (logits - t.astype(np.float32)) * np.float32(inv_n / D)).astype(np.float32)
    else:
        out = log_softmax(logits)
        loss = float(-out[np.arange(N), t].sum() * inv_n)
        onehot = np.zeros_like(out)
        onehot[np.arange(N), t] = 1.0
        g3 = ((np.exp(out) - onehot) * np.float32(inv_n)).astype(np.float32)
    g3r = rnd(g3).astype(f8)
    gw3 = g3r.T @ xs[-1]
    gb3 = g3.astype(f8).sum(axis=0)
    gx = g3r @ rnd(w3).astype(f8)
    res = [None] * len(layers)
    for k in range(len(layers) - 1, -1, -1):
        lay = layers[k]
        gp = np.where(xs[k + 1] > 0, gx, 0.0).astype(np.float32)      # ReLU mask on the stored activation
        g = rnd(gp).astype(f8)
        gv = rnd(gp * rs[k]).astype(f8)
        var32 = np.exp(lay["lvars"])
        gw = g.T @ xs[k]
        gs2 = gv.T @ x2s[k]
        stdv = np.sqrt(var32.astype(f8))
        vh = float(np.sum(var32.astype(f8) + lay["means"].astype(f8) ** 2) / lay["means"].size)
        mu_e = rnd(lay["means"]).astype(f8) if kl_shadows else lay["means"].astype(f8)
        var_e = rnd(var32).astype(f8) if kl_shadows else var32.astype(f8)
        res[k] = dict(gradWeight=gw, gradSum=2 * gs2 * stdv, gradBias=g.sum(axis=0),
                      grad_mu=gw / S + kl_scale * mu_e / (B * vh),
                      grad_lv=gs2 * var_e / S + kl_scale * (var_e / vh - 1.0) / (2 * B))
        if k > 0:
            mu, var = rnd(lay["means"]).astype(f8), rnd(var32).astype(f8)
            gx = g @ mu + 2 * xs[k] * (gv @ var)
    return loss, res, gw3, gb3


# ---- BLAS-speed single-layer math for big-shape parity (operands optionally bf16-rounded) ---------
def lrt_forward(x, means, lvars, bias, zeta, rnd=lambda a: a):
    xr = rnd(x)
    m = xr.astype(np.float64) @ rnd(means).astype(np.float64).T
    v = rnd(xr * xr).astype(np.float64) @ rnd(np.exp(lvars)).astype(np.float64).T
    y = m + bias + np.sqrt(v) * zeta
    return y, v
