#!/usr/bin/env python3
"""LAB: read the cluster stamps of a tools/lab/pst.h build (see that header):
    VBNN_HIP_LIB=$PWD/vbnn_amd/lib/pst/libvbnn_hip.so python tools/pst_read.py
One wide training step, then per translation unit (gemm_fwd / gemm_dx / gemm_dw: the LAST gemm_nt_v3 launch of each) and per wave of
workgroup 8: cycles per PHASE spent issuing the M cluster (reads + pieces), between that and the start of the C cluster (counted wait
+ barrier), in the C cluster's MFMAs -- and the whole pass per phase (the rest is the barrier behind the C cluster)."""
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
    buf = (C.c_ulonglong * 64)()
    rc = fn(buf)
    print(f"{tu}: rc {rc}")
    for w in range(8):
        a0, a1, a2, tot, n = buf[w * 8:w * 8 + 5]
        if n == 0:
            continue
        # s_memtime counts at a constant 100 MHz?  no: it is the shader clock here -- report raw counts per phase
        print(f"  wave {w}: phases {n:4d}  M issue {a0 / n:7.1f}  wait+barrier {a1 / n:7.1f}  C {a2 / n:7.1f}  C barrier+rest {(tot - a0 - a1 - a2) / n:7.1f}  "
              f"pass/phase {tot / n:7.1f}")
