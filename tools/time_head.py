#!/usr/bin/env python3
"""The fused head's backward launch (vbnn_head_backward) in isolation at the wide configuration's size, microseconds per launch
(HIP events around 20 back-to-back launches): python tools/time_head.py [label]. Variants by environment (one process each):
VBNN_HEAD_FAST=0 (the general kernel), VBNN_HEAD_BLOCKS=n (target workgroup count), VBNN_HEAD_STREAM=0 (the tile form instead of the streaming one)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                   # noqa: E402
from vbnn_amd import _lib as L                 # noqa: E402
from vbnn_amd.engine import FusedMLP           # noqa: E402
from vbnn_amd.nn import _p, fill_normal        # noqa: E402

N = 4096
opt = dict(var_init=1e-3, B=1e6, S=1, mode="lrt", dtype="bf16", seed=3, input_size=784, hidden=[4096, 4096], n_classes=10, fuse_kl=True)
eng = FusedMLP(opt)
x = torch.empty(N, 784, dtype=torch.float32, device="cuda")
fill_normal(x, 3, L.STREAM_DATA, 0, 0)
t = eng.synthetic_targets(x, 0)
eng.prepare()
for _ in range(3):
    eng.resetGradients(); eng.sample(); eng.run(x, t)
lib, ctx, vl = L.lib(), eng.ctx.h, eng.vb[-1]


def head_bwd():
    L.check(lib.vbnn_head_backward(ctx, eng.code, eng.h_s.ptr, eng.h_s.ld, eng.w3_s.ptr, eng.w3_s.ld, _p(eng.g_logits), N, 4096, 10, 0,
                                   _p(eng.gradWeight3), _p(eng.gradBias3), _p(vl.gradBias), 1, _p(vl.r), vl.O, 1, vl.g_s.ptr, vl.gv_s.ptr,
                                   vl.g_s.ld, None, None, 0))


for _ in range(5):
    head_bwd()
torch.cuda.synchronize()
res = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        head_bwd()
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 20 * 1e3)
print(f"{sys.argv[1] if len(sys.argv) > 1 else ''}: head backward + finish {sorted(res)[2]:.1f} us per call (5 blocks of 20: {' '.join(f'{r:.1f}' for r in res)})")
