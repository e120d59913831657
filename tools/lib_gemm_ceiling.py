#!/usr/bin/env python3
"""Calibration only (never part of the product path): what the vendor library reaches on one plain bf16 NT GEMM of the
wide layer's shape, as a practical ceiling to read gemm_nt_v2's numbers against. python tools/lib_gemm_ceiling.py"""
import torch, time
torch.manual_seed(0)
for n in (4096, 8192):
    a = torch.randn(n, n, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(n, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(5): c = a @ b.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): c = a @ b.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"n={n} NT bf16: {ms*1e3:.1f} us  {2*n**3/ms/1e9:.0f} TFLOP/s", flush=True)
