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
    buf = (C.c_ulonglong * 256)()
    rc = fn(buf)
    print(f"{tu}: rc {rc}")
    for w in range(8):
        a0, a1, a2, tot, n, a4, a5 = buf[w * 8:w * 8 + 7]
        if n == 0:
            continue
        # s_memtime counts at a constant 100 MHz?  no: it is the shader clock here -- report raw counts per phase
        print(f"  wave {w}: phases {n:4d}  M start -> C start {a1 / n:7.1f} (M start -> LDS reads back {a4 / n:6.1f}, then pieces landed {a5 / n:6.1f})  C {a2 / n:7.1f}  C barrier+rest {(tot - a0 - a1 - a2) / n:7.1f}  "
              f"pass/phase {tot / n:7.1f}")
    raw = [[buf[64 + w * 16 + k] for k in range(12)] for w in range(8)]
    t00 = min(r[0] for r in raw if r[0])
    print("  one K step (phase 0 then phase 1), cycles from the first wave's M start: M start, (M start), reads back, pieces landed, C start, C end")
    for w in range(8):
        print(f"  wave {w}: " + "  ".join(f"{(v - t00):6d}" for v in raw[w][:6]) + "   |   " + "  ".join(f"{(v - t00):6d}" for v in raw[w][6:]))
    print("  phase 0's M cluster, cycles from its start, after: B k0 reads, piece 0, B k1 reads, piece 1, A k0 reads, piece 2, A k1 reads, piece 3")
    for w in range(8):
        print(f"  wave {w}: " + "  ".join(f"{buf[192 + w * 8 + k]:5d}" for k in range(8)))
