#!/usr/bin/env python3
"""Launch the L2-sized forward dual GEMM a few times (for rocprofv3 --pmc passes). argv: sched [single]"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vbnn_amd import _lib as L
from vbnn_amd.engine import FusedMLP
from vbnn_amd.nn import fill_normal
sched = int(sys.argv[1]) if len(sys.argv) > 1 else 1
opt = dict(var_init=1e-3, B=1e6, S=1, mode="lrt", dtype="bf16", seed=3, input_size=784, hidden=[4096, 4096], n_classes=10, keep_transposes=True)
eng = FusedMLP(opt)
N = 4096
x = torch.empty(N, 784, dtype=torch.float32, device="cuda"); fill_normal(x, 3, 4, 0, 0)
t = torch.zeros(N, dtype=torch.int32, device="cuda")
eng.resetGradients(); eng.prepare(); eng.sample(); eng.run(x, t); torch.cuda.synchronize()
lib, ctx, code = L.lib(), eng.ctx.h, eng.code
L.check(lib.vbnn_debug_set(1, sched))
for _ in range(4):
    a = eng._fwd_args(1, N, 0)
    if len(sys.argv) > 2:          # single GEMM, no noise: the bare main loop
        a.w2 = None; a.x2 = None; a.r = None
    L.check(lib.vbnn_forward(ctx, code, C.byref(a)))
torch.cuda.synchronize()
