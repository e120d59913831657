#!/usr/bin/env python3
"""Bare main loop: the L2-sized forward as a single GEMM without noise and with no stores, per schedule."""
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
def mk(single, stores):
    a = eng._fwd_args(1, N, 0)
    if single: a.w2 = None; a.x2 = None
    if not stores: a.r = None; a.h = None; a.hT = None; a.h2 = None; a.h2T = None
    if single: a.r = None
    return a
cases = {"single, no stores": mk(True, False), "single, h store": mk(True, True), "dual, no stores": mk(False, False), "dual, full": mk(False, True)}
def run(a, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): L.check(lib.vbnn_forward(ctx, code, C.byref(a)))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
res = {}
for rnd in range(5):
    for s in (0, 1):
        L.check(lib.vbnn_debug_set(1, s))
        for k, a in cases.items(): res.setdefault((k, s), []).append(run(a))
for k in cases:
    fl = 2.0 * N * 4096 * 4096 * (1 if k.startswith("single") else 2)
    line = f"{k:20s}"
    for s in (0, 1):
        v = sorted(res[(k, s)]); med = v[len(v) // 2]
        line += f"  sched{s} {med:7.1f} us ({fl / med / 1e6:6.0f} TF)"
    print(line)
