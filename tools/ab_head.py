#!/usr/bin/env python3
"""Time the classifier head's kernels of the wide step in isolation. python tools/ab_head.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vbnn_amd import _lib as L
from vbnn_amd.engine import FusedMLP, _p
from vbnn_amd.nn import fill_normal
opt = dict(var_init=1e-3, B=1e6, S=1, mode="lrt", dtype="bf16", seed=3, input_size=784, hidden=[4096, 4096], n_classes=10, keep_transposes=True)
eng = FusedMLP(opt)
N = 4096
x = torch.empty(N, 784, dtype=torch.float32, device="cuda"); fill_normal(x, 3, 4, 0, 0)
t = torch.zeros(N, dtype=torch.int32, device="cuda")
eng.resetGradients(); eng.prepare(); eng.sample(); eng.run(x, t); torch.cuda.synchronize()
lib, ctx, code = L.lib(), eng.ctx.h, eng.code
vl = eng.vb[-1]; H, Cn = eng.sizes[-1], eng.n_classes
def fwd():
    L.check(lib.vbnn_head_forward(ctx, code, eng.h_s.ptr, eng.h_s.ld, eng.w3_s.ptr, eng.w3_s.ld, _p(eng.bias3), _p(t), N, H, Cn,
                                  1.0 / N, _p(eng.logits), _p(eng.out), _p(eng.g_logits), 0, _p(eng._acc), _p(eng._corr)))
def bwd(dw=True, tr=True, gv=True, bp=True):
    L.check(lib.vbnn_head_backward(ctx, code, eng.h_s.ptr, eng.h_s.ld, eng.w3_s.ptr, eng.w3_s.ld, _p(eng.g_logits), N, H, Cn, 0,
                                   _p(eng.gradWeight3) if dw else None, _p(eng.gradBias3) if dw else None,
                                   _p(vl.gradBias) if (dw and bp) else None, 1, _p(vl.r) if gv else None, vl.O, 1, vl.g_s.ptr,
                                   vl.gv_s.ptr if gv else None, vl.g_s.ld, vl.gT_s.ptr if tr else None,
                                   vl.gvT_s.ptr if (tr and gv) else None, vl.gT_s.ld))
def run(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print(f"head forward  {run(fwd):7.1f} us   head backward (+finish) {run(bwd):7.1f} us")
print(f"  backward without gradWeight/bias sums      {run(lambda: bwd(dw=False)):7.1f} us")
print(f"  backward without the transposed copies     {run(lambda: bwd(tr=False)):7.1f} us")
print(f"  backward without gv (no r, no gv outputs)  {run(lambda: bwd(gv=False)):7.1f} us")
print(f"  backward: g_prev only                      {run(lambda: bwd(dw=False, tr=False, gv=False)):7.1f} us")
print(f"  no transposes, no sums                     {run(lambda: bwd(dw=False, tr=False)):7.1f} us")
print(f"  no transposes, sums without bias_prev      {run(lambda: bwd(tr=False, bp=False)):7.1f} us")
