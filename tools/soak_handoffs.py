#!/usr/bin/env python3
"""Soak of the polled stream hand-offs (lab): N steps of the wide step (a) with the one-GPU stand-in of an 8-rank direct exchange
(trigger word, done word, barrier launches; its "peers" are the rank's own arena, so its sums mean nothing -- a hang, which shows as
this script's timeout, is what is looked for), (b) with the one-rank RCCL exchange (trigger and done words of comm.hip; one rank's sum
is its own gradients, so the arena must stay bitwise the first step's: same minibatch, same draw).
    timeout -k 10 400 python3 tools/soak_handoffs.py [steps, default 1000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vbnn_amd import _lib as L
from vbnn_amd.engine import FusedMLP
from vbnn_amd.nn import fill_normal
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
base = dict(var_init=1e-3, B=1e6, S=1, mode="lrt", dtype="bf16", seed=3, input_size=784, hidden=[4096, 4096], n_classes=10, fuse_kl=True)
x = torch.empty(4096, 784, dtype=torch.float32, device="cuda"); fill_normal(x, 3, L.STREAM_DATA, 0, 0)
def soak(eng, name, check):
    t = eng.synthetic_targets(x, 0)
    def step():
        eng.draw = 0
        eng.resetGradients(); eng.sample(); eng.run(x, t); eng.finish()
    step(); step(); torch.cuda.synchronize()
    ref = eng.grads.clone()
    t0 = time.perf_counter(); bad = 0
    for k in range(steps):
        step()
        if k % 50 == 49:
            torch.cuda.synchronize()
            if check and not torch.equal(eng.grads.view(torch.int32), ref.view(torch.int32)): bad += 1
    torch.cuda.synchronize()
    print(f"{name}: {steps} steps, {(time.perf_counter() - t0) / steps * 1e3:.4f} ms per step" +
          (f", arenas that differed from the first step's: {bad} of {steps // 50} checks" if check else ""), flush=True)
    assert bad == 0
eng = FusedMLP(dict(base, exchange="p2p"), world_size=1, rank=0, force_reduce=True)
eng.prepare()
for name, sim in (("stand-in of 8 ranks, unpaced", (8, 0.0)), ("stand-in of 8 ranks, 770 GB/s", (8, 770.0))):
    eng.exchange().standin(*sim)
    soak(eng, name, False)
del eng
eng = FusedMLP(dict(base, exchange="rccl"), world_size=1, rank=0, force_reduce=True)
eng.prepare()
soak(eng, "one-rank RCCL exchange", True)
print("soak ok")
