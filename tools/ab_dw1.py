#!/usr/bin/env python3
"""A/B of the launch choices for the 784 x 4096 parameter-gradient GEMM pair (N = 4096) in isolation:
(tile, split-K, pair split, K-major, schedule) through vbnn_debug_set. Interleaved rounds, medians."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vbnn_amd import _lib as L
from vbnn_amd.engine import FusedMLP
from vbnn_amd.nn import fill_normal

opt = dict(var_init=1e-3, B=1e6, S=1, mode="lrt", dtype="bf16", seed=3, input_size=784, hidden=[4096, 4096], n_classes=10, keep_transposes=True)
eng = FusedMLP(opt)
N = 4096
x = torch.empty(N, 784, dtype=torch.float32, device="cuda"); fill_normal(x, 3, 4, 0, 0)
t = torch.zeros(N, dtype=torch.int32, device="cuda")
eng.resetGradients(); eng.prepare(); eng.sample(); eng.run(x, t); torch.cuda.synchronize()
lib, ctx, code = L.lib(), eng.ctx.h, eng.code
fn = lambda: L.check(lib.vbnn_acc_grad_parameters(ctx, code, C.byref(eng._dw_args(0, N, 0))))
def run(reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
# (kernel, sched, tile, splitk, psplit, kmajor)
variants = {
    "cur (auto)":            (0, -1, 0, -1, -1, 1),
    "256 psplit NT":         (2, -1, 256, 0, 1, 0),
    "256 psplit KM":         (2, -1, 256, 0, 1, 1),
    "256 dual NT":           (2, -1, 256, 0, 0, 0),
    "256 dual splitK NT":    (2, -1, 256, 1, 0, 0),
    "128 dual NT":           (2, -1, 128, 0, 0, 0),
    "128 psplit NT":         (2, -1, 128, 0, 1, 0),
    "256 psplit NT sched0":  (2, 0, 256, 0, 1, 0),
    "256 psplit NT sched2":  (2, 2, 256, 0, 1, 0),
    "256 psplit NT sched4":  (2, 4, 256, 0, 1, 0),
    "v3 NT":                 (3, -1, 0, -1, 0, 0),
}
res = {}
for rnd in range(5):
    for name, (k, s, tile, sk, ps, km) in variants.items():
        for key, val in ((0, k), (1, s), (2, tile), (3, sk), (5, ps), (6, km)):
            L.check(lib.vbnn_debug_set(key, val))
        try:
            res.setdefault(name, []).append(run())
        except Exception as e:
            res[name] = [float("nan")]
            print(name, "failed:", e)
for name, v in res.items():
    v = sorted(v); med = v[len(v) // 2]
    print(f"{name:24s} {med:7.1f} us  ({4.0 * N * 784 * 4096 / med / 1e6:6.0f} TF)")
