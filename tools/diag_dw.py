#!/usr/bin/env python3
"""Diagnostic (not product): one K-major accGradParameters launch at I = O = 4096 against float64, with a map of WHERE the
errors are (which 16-wide m / n blocks of the 256 x 256 tile, which launch of two).  python3 tools/diag_dw.py [N]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vbnn_amd import _lib as L
from vbnn_amd.engine import FusedMLP
from oracle.ref_numpy import bf16_round
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
I = O = 4096
opt = dict(var_init=1e-3, mu_init=1, B=1e6, S=1, mode="lrt", dtype="bf16", seed=3, input_size=I, hidden=[I, O], n_classes=10, fuse_kl=True)
eng = FusedMLP(opt)
rng = np.random.default_rng(11)
v = eng.vb[1]
eng._alloc_batch(N); eng.prepare(); torch.cuda.synchronize()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x = bf16_round(np.maximum(rng.normal(0, 1, (N, I)), 0).astype(np.float32))
g = bf16_round(rng.normal(0, 1e-3, (N, O)).astype(np.float32))
gv = bf16_round((g * rng.normal(0, 3.0, (N, O))).astype(np.float32))
v.x_s.t[:, :I].copy_(dev(x)); v.x2_s.t[:, :I].copy_(dev(bf16_round(x * x)))
v.g_s.t[:, :O].copy_(dev(g)); v.gv_s.t[:, :O].copy_(dev(gv))
mu = v.mu_s.t[:, :I].float().cpu().numpy().astype(np.float64)
means = v.means.cpu().numpy().astype(np.float64); var32 = np.exp(v.lvars.cpu().numpy().astype(np.float64))
vh = float(np.sum(var32 + means ** 2) / means.size)
want = g.astype(np.float64).T @ x.astype(np.float64) + mu / (opt["B"] * vh)
tol = 4e-6 * (np.abs(g).astype(np.float64).T @ np.abs(x).astype(np.float64)) + 1e-12
lib, ctx = L.lib(), eng.ctx.h
print("library:", L.LIB_PATH, " dw_km:", v.dw_km, " N:", N)
prev = None
for rep in range(3):
    v.gradWeight.zero_()
    L.check(lib.vbnn_acc_grad_parameters(ctx, eng.code, C.byref(eng._dw_args(1, N, 0))))
    torch.cuda.synchronize()
    got = v.gradWeight.cpu().numpy().astype(np.float64)
    bad = np.abs(got - want) > tol                         # [o][i]: kernel M = i (A = x), N = o (B = g)
    print(f"launch {rep}: {bad.sum()} of {bad.size} off, max err {np.abs(got - want).max():.3e}, same as previous launch: {prev is not None and np.array_equal(prev, got)}")
    if bad.any():
        o_idx, i_idx = np.nonzero(bad)
        print("   tiles (o // 256, i // 256) with errors:", len(set(zip(o_idx // 256, i_idx // 256))), "of 256")
        print("   o % 256 // 16 histogram (n-blocks 0..15):", np.bincount((o_idx % 256) // 16, minlength=16).tolist())
        print("   i % 256 // 16 histogram (m-blocks 0..15):", np.bincount((i_idx % 256) // 16, minlength=16).tolist())
        print("   o % 16 histogram:", np.bincount(o_idx % 16, minlength=16).tolist())
        print("   i % 16 histogram:", np.bincount(i_idx % 16, minlength=16).tolist())
    prev = got

# ---- stress: many launches, every result compared bitwise with the first (on the device); VBNN_DIAG_REPS launches
reps = int(os.environ.get("VBNN_DIAG_REPS", "0"))
if reps:
    v.gradWeight.zero_(); v.gradSum.zero_()
    L.check(lib.vbnn_acc_grad_parameters(ctx, eng.code, C.byref(eng._dw_args(1, N, 0))))
    ref_w, ref_s = v.gradWeight.clone(), v.gradSum.clone()
    bad = 0
    for r in range(reps):
        L.check(lib.vbnn_acc_grad_parameters(ctx, eng.code, C.byref(eng._dw_args(1, N, 0))))
        if not (torch.equal(v.gradWeight, ref_w) and torch.equal(v.gradSum, ref_s)):
            bad += 1
            d = (v.gradWeight != ref_w)
            idx = d.nonzero()[:3].tolist()
            print(f"   launch {r}: {int(d.sum())} elements of d/dmeans differ from launch 0, e.g. {idx}", flush=True)
    print(f"stress: {bad} of {reps} launches differ bitwise from the first", flush=True)
