#!/usr/bin/env python3
"""A/B of the pipelined GEMM's schedules on the three L2-sized launches of the wide step (interleaved rounds)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vbnn_amd import _lib as L
from vbnn_amd.engine import FusedMLP
from vbnn_amd.nn import fill_normal

if os.environ.get("VBNN_AB_KERNEL"):                     # force a GEMM kernel family BEFORE the engine asks the library what it offers
    L.check(L.lib().vbnn_debug_set(0, int(os.environ["VBNN_AB_KERNEL"])))     # (vbnn_debug_set key 0: 2 = gemm_nt_v2, 3 = gemm_nt_v3)
opt = dict(var_init=1e-3, B=1e6, S=1, mode="lrt", dtype="bf16", seed=3, input_size=784, hidden=[4096, 4096], n_classes=10, keep_transposes=True)
eng = FusedMLP(opt)
N = 4096
x = torch.empty(N, 784, dtype=torch.float32, device="cuda"); fill_normal(x, 3, 4, 0, 0)
t = torch.zeros(N, dtype=torch.int32, device="cuda")
eng.resetGradients(); eng.prepare(); eng.sample(); eng.run(x, t); torch.cuda.synchronize()
lib, ctx, code = L.lib(), eng.ctx.h, eng.code
calls = {
    "fwd L2": lambda: L.check(lib.vbnn_forward(ctx, code, C.byref(eng._fwd_args(1, N, 0)))),
    "fwd L1": lambda: L.check(lib.vbnn_forward(ctx, code, C.byref(eng._fwd_args(0, N, 0)))),
    "dw  L2": lambda: L.check(lib.vbnn_acc_grad_parameters(ctx, code, C.byref(eng._dw_args(1, N, 0)))),
    "dw  L1": lambda: L.check(lib.vbnn_acc_grad_parameters(ctx, code, C.byref(eng._dw_args(0, N, 0)))),
    "dx  L2": lambda: L.check(lib.vbnn_grad_input(ctx, code, C.byref(eng._dx_args(1, N)))),
}
flops = {"fwd L2": 4.0 * N * 4096 * 4096, "fwd L1": 4.0 * N * 784 * 4096, "dw  L2": 4.0 * N * 4096 * 4096,
         "dw  L1": 4.0 * N * 784 * 4096, "dx  L2": 4.0 * N * 4096 * 4096}
def run(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
scheds = [int(a) for a in sys.argv[1:]] or [0, 2]

res = {}
for rnd in range(5):
    for s in scheds:
        L.check(lib.vbnn_debug_set(1, s))
        for k, fn in calls.items():
            res.setdefault((k, s), []).append(run(fn))
for k in calls:
    line = f"{k}: "
    for s in scheds:
        v = sorted(res[(k, s)]); med = v[len(v) // 2]
        line += f" sched{s} {med:7.1f} us ({flops[k] / med / 1e6:6.0f} TF)"
    print(line)
