#!/usr/bin/env python3
"""How long the host takes to ISSUE a step vs how long the GPU takes to run it (small and wide configs)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vbnn_amd.engine import FusedMLP
from vbnn_amd.nn import fill_normal
for name, hidden, N, dt in (("small", [400, 400], 256, "f32"), ("wide", [4096, 4096], 4096, "bf16")):
    opt = dict(var_init=1e-3, B=1e6, S=1, mode="lrt", dtype=dt, seed=3, input_size=784, hidden=hidden, n_classes=10, fuse_kl=True)
    eng = FusedMLP(opt)
    x = torch.empty(N, 784, dtype=torch.float32, device="cuda"); fill_normal(x, 3, 4, 0, 0)
    t = torch.zeros(N, dtype=torch.int32, device="cuda")
    def step():
        eng.resetGradients(); eng.prepare(); eng.sample(); eng.run(x, t); eng.finish()
    for _ in range(20): step()
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host issue {1e6 * (t1 - t0) / K:7.1f} us/step, total {1e6 * (t2 - t0) / K:7.1f} us/step")
