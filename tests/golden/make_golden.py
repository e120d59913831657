#!/usr/bin/env python3
"""Generates tests/golden/*.json|npz. The reference (louissmit/VBNN) has no golden vectors, tests or
fixtures of its own and cannot be executed here (Lua/Torch7), so these are:
  philox_kat.json   the published Random123 known-answer vectors for Philox4x32-10 (kat_vectors of
                    Random123 1.09 [recalled]; they pin the RNG core independently of this repo)
  vblinear_small.npz  inputs + expected outputs of the CPU oracle (oracle/vbnn_oracle.c) for two small
                    layers in both modes -- a regression pin of the oracle and a portable known-answer set
                    for the HIP path. Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import vbnn_oracle as vo  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

KAT = [
    {"ctr": [0, 0, 0, 0], "key": [0, 0], "out": [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]},
    {"ctr": [0xffffffff] * 4, "key": [0xffffffff] * 2, "out": [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]},
    {"ctr": [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], "key": [0xa4093822, 0x299f31d0],
     "out": [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]},
]


def layer_case(mode, N, I, O, seed=3):
    opt = dict(var_init=1e-3, mu_init=1, B=1e6, S=2, mode=mode, seed=seed)
    om = vo.OracleVBLinear(I, O, opt, layer_id=1)
    rng = np.random.default_rng(100 + N + I + O)
    om.means[:] = rng.normal(0, np.sqrt(2.0 / I), (O, I)).astype(np.float32)
    om.lvars[:] = rng.normal(np.log(1e-3), 0.3, (O, I)).astype(np.float32)
    om.bias[:] = rng.normal(0, 0.1, O).astype(np.float32)
    means, lvars, bias = om.means.copy(), om.lvars.copy(), om.bias.copy()
    om.compute_prior()
    x = rng.normal(0, 1, (N, I)).astype(np.float32)
    g = rng.normal(0, 1.0 / N, (N, O)).astype(np.float32)
    outs = {}
    for d in range(2):
        om.sample()
        outs[f"output{d}"] = om.updateOutput(x).copy()
        outs[f"gradInput{d}"] = om.backward(x, g, 0.5).copy()
    outs.update(gradWeight=om.gradWeight.copy(), gradSum=om.gradSum.copy(), gradBias=om.gradBias.copy(),
                var_hat=np.float64(om.var_hat), lc=np.float64(om.calc_lc(opt)))
    mle, mlc = om.compute_mugrads(opt)
    vle, vlc = om.compute_vargrads(opt)
    outs.update(mugrad_le=mle.copy(), mugrad_lc=mlc.copy(), vargrad_le=vle.copy(), vargrad_lc=vlc.copy())
    ins = dict(means=means, lvars=lvars, bias=bias, x=x, g=g)
    return {f"{mode}_{N}x{I}x{O}_in_{k}": v for k, v in ins.items()} | \
           {f"{mode}_{N}x{I}x{O}_out_{k}": v for k, v in outs.items()}


def main():
    with open(os.path.join(HERE, "philox_kat.json"), "w") as f:
        json.dump(KAT, f, indent=1)
    arrays = {}
    for mode in ("wn", "lrt"):
        for shape in ((3, 7, 5), (32, 64, 48)):
            arrays.update(layer_case(mode, *shape))
    arrays["normal_eps_5x7"] = vo.fill_normal(5, 7, 3, vo.STREAM_EPS, 1, 1)
    arrays["normal_zeta_3x5_row1000"] = vo.fill_normal(3, 5, 3, vo.STREAM_ZETA, 2, 7, 1000)
    np.savez_compressed(os.path.join(HERE, "vblinear_small.npz"), **arrays)
    print("wrote", len(arrays), "arrays")


if __name__ == "__main__":
    main()
