#!/usr/bin/env python3
"""A/B of an environment switch of the p2p exchange in the one-GPU stand-in (lab): the step without exchange calls, the stand-in paced
at 770 GB/s inbound and unpaced, three blocks of 20 steps each; run once per setting of the switch, alternately."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vbnn_amd import _lib as L
from vbnn_amd.engine import FusedMLP
from vbnn_amd.nn import fill_normal
opt = dict(var_init=1e-3, B=1e6, S=1, mode="lrt", dtype="bf16", seed=3, input_size=784, hidden=[4096, 4096], n_classes=10, fuse_kl=True, exchange="p2p")
eng = FusedMLP(opt, world_size=1, rank=0, force_reduce=True)
x = torch.empty(4096, 784, dtype=torch.float32, device="cuda"); fill_normal(x, 3, L.STREAM_DATA, 0, 0)
t = eng.synthetic_targets(x, 0)
eng.prepare()
def step():
    eng.resetGradients(); eng.sample(); eng.run(x, t); eng.finish()
def blocks(n=3, k=20):
    for _ in range(5): step()
    out = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k): step()
        torch.cuda.synchronize(); out.append(round((time.perf_counter() - t0) / k * 1e3, 4))
    return out
eng.skip_exchange = True; base = blocks(); eng.skip_exchange = False
res = {"no exchange": base}
for gbps in (770.0, 0.0):
    eng.exchange().standin(8, gbps)
    res[f"stand-in {gbps:.0f}"] = blocks()
print({k: v for k, v in os.environ.items() if k.startswith("VBNN_P2P")}, res, flush=True)
