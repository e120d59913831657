#!/usr/bin/env python3
"""Ablation of the forward dual-GEMM epilogue on MI355X: which outputs cost what (interleaved rounds, one process)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vbnn_amd import _lib as L
from vbnn_amd.engine import FusedMLP
from vbnn_amd.nn import fill_normal, _p

I = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
opt = dict(var_init=1e-3, B=1e6, S=1, mode="lrt", dtype="bf16", seed=3, input_size=I, hidden=[4096, 4096], n_classes=10, keep_transposes=True)
eng = FusedMLP(opt)
N = 4096
x = torch.empty(N, I, dtype=torch.float32, device="cuda"); fill_normal(x, 3, 4, 0, 0)
t = torch.zeros(N, dtype=torch.int32, device="cuda")
eng.resetGradients(); eng.prepare(); eng.sample(); eng.run(x, t); torch.cuda.synchronize()
lib, ctx, code = L.lib(), eng.ctx.h, eng.code

def variant(**over):
    a = eng._fwd_args(0, N, 0)
    for k, v in over.items(): setattr(a, k, v)
    return a
full = dict()
variants = {
    "full (r,h,h2,hT,h2T)": {},
    "no transposes": dict(hT=None, h2T=None),
    "no h2/h2T": dict(h2=None, h2T=None),
    "only r": dict(h=None, h2=None, hT=None, h2T=None),
    "only h": dict(r=None, h2=None, hT=None, h2T=None),
    "only hT": dict(r=None, h=None, h2=None, h2T=None),
    "nothing stored": dict(r=None, h=None, h2=None, hT=None, h2T=None),
}
args = {k: variant(**v) for k, v in variants.items()}
def run(a, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): L.check(lib.vbnn_forward(ctx, code, C.byref(a)))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
res = {k: [] for k in args}
for rnd in range(5):
    for k, a in args.items(): res[k].append(run(a))
flops = 2 * 2.0 * N * I * 4096
for k, v in res.items():
    med = sorted(v)[len(v) // 2]
    print(f"{k:28s} median {med:8.1f} us  min {min(v):8.1f} us   {flops / med / 1e6:7.1f} TF")
