#!/usr/bin/env python3
"""LAB: read the cluster stamps of a tools/lab/pst.h build (see that header):
    VBNN_HIP_LIB=$PWD/vbnn_amd/lib/pst/libvbnn_hip.so python tools/pst_read.py
One wide training step, then per translation unit (gemm_fwd / gemm_dx / gemm_dw: the LAST gemm_nt_v3 launch of each) and per wave of
workgroup 8: cycles per PHASE in the M cluster (reads + pieces + its closing wait and barrier) and in the C cluster's MFMAs, the whole
pass per phase (the rest is the barrier behind the C cluster), and one K step's stamps as they are."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vbnn_amd import _lib as L
from vbnn_amd.engine import FusedMLP
from vbnn_amd.nn import fill_normal

opt = dict(var_init=1e-3, B=1e6, S=1, mode="lrt", dtype="bf16", seed=3, fuse_kl=True, input_size=784, hidden=[4096, 4096], n_classes=10,
           state=dict(learningRate=1e-3), meanState=dict(learningRate=1e-4), varState=dict(learningRate=5e-2))
eng = FusedMLP(opt)
N = 4096
x = torch.empty(N, 784, dtype=torch.float32, device="cuda"); fill_normal(x, 3, 4, 0, 0)
t = eng.synthetic_targets(x, 0)
eng.prepare()
for _ in range(5):
    eng.resetGradients(); eng.sample(); eng.run(x, t)
torch.cuda.synchronize()
lib = L.lib()
for tu in ("gemm_fwd", "gemm_dx", "gemm_dw"):
    fn = getattr(lib, "vbnn_lab_pst_" + tu)
    buf = (C.c_ulonglong * 192)()
    rc = fn(buf)
    print(f"{tu}: rc {rc}   (cycles per phase = one M cluster + one C cluster of a wave; a K step is two phases = four slots)")
    for w in range(8):
        am, ac, ap, n = buf[w * 8:w * 8 + 4]
        if n < 2:
            continue
        print(f"  wave {w}: phases {n:4d}  M cluster (start -> C start, with its closing wait and barrier) {am / (n - 1):7.1f}   C cluster (32 MFMAs issued) {ac / (n - 1):7.1f}"
              f"   passes / phases {ap / n:7.1f}")
    raw = [[buf[64 + w * 16 + k] for k in range(6)] for w in range(8)]
    t00 = min(r[0] for r in raw if r[0])
    print("  one K step, cycles from the first wave's M start: phase 0: M start, C start, C end | phase 1: M start, C start, C end")
    for w in range(8):
        print(f"  wave {w}: " + "  ".join(f"{(v - t00):6d}" for v in raw[w][:3]) + "   |   " + "  ".join(f"{(v - t00):6d}" for v in raw[w][3:]))
