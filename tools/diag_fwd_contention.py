#!/usr/bin/env python3
"""Diagnostic (not product): does the pipelined dual GEMM (gemm_nt_v2) ever compute a different forward under memory
contention INSIDE one process? The layer-2 forward of the two-rank rehearsal (N = 512, I = O = 4096) is launched over and
over on the engine's stream while a second stream streams copies through HBM; every output is compared bitwise with the first.
python3 tools/diag_fwd_contention.py [launches]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vbnn_amd import _lib as L
from vbnn_amd.engine import FusedMLP
from vbnn_amd.nn import fill_normal
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
N, I0 = 512, 784
opt = dict(var_init=1e-3, mu_init=1, B=1e6, S=1, mode="lrt", dtype="bf16", seed=3, input_size=I0, hidden=[4096, 4096], n_classes=10, fuse_kl=True)
eng = FusedMLP(opt)
x = torch.empty(N, I0, dtype=torch.float32, device="cuda"); fill_normal(x, 3, 4, 0, 0)
t = (torch.arange(N, device="cuda", dtype=torch.int64) * 7 % 10).to(torch.int32)
eng.resetGradients(); eng.prepare(); eng.sample(); eng.run(x, t); torch.cuda.synchronize()
ref_h, ref_g = eng.h_s.t.clone(), eng.grads.clone()
side = torch.cuda.Stream()
a = torch.empty(256 << 20, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
bad = 0
for r in range(reps):
    with torch.cuda.stream(side):
        for _ in range(4): b.copy_(a, non_blocking=True)          # ~2 GB of HBM traffic beside the step
    eng.resetGradients(); eng.run(x, t)
    torch.cuda.synchronize()
    same_h, same_g = torch.equal(eng.h_s.t, ref_h), torch.equal(eng.grads, ref_g)
    if not (same_h and same_g):
        bad += 1
        print(f"step {r}: activations equal {same_h}, gradients equal {same_g}", flush=True)
print(f"contention stress: {bad} of {reps} steps differ bitwise from the first", flush=True)
