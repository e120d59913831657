#!/usr/bin/env python3
"""Hashes of what a training step produces, for comparing two BUILDS of the library bit for bit:
    VBNN_HIP_LIB=.../lib/<name>/libvbnn_hip.so python tools/step_bits.py [config] [steps] [repeats]
prints one line per (repeat, step): sha256 of the gradient arena, of the loss sum, and after the update of the operand shadows.
The same seed, inputs and draws every repeat: the lines of one build must repeat, and two builds that claim the same arithmetic
(same MFMA order per accumulator) must print the same lines."""
import hashlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vbnn_amd.engine import FusedMLP
from vbnn_amd.nn import fill_normal

CONFIGS = {
    "wide": dict(input_size=784, hidden=[4096, 4096], n_classes=10, batch=4096),
    "deep": dict(input_size=784, hidden=[4096] * 4, n_classes=10, batch=4096),
    "ragged": dict(input_size=784, hidden=[4096 + 256, 4096], n_classes=10, batch=4096 + 256),
}


def h(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


def run(name="wide", steps=3, repeats=2, out=print):
    """-> per repetition, the list of lines (hashes per step); prints them as they come"""
    cfg = dict(CONFIGS[name])
    N = cfg.pop("batch")
    reps = []
    for rep in range(repeats):
        opt = dict(var_init=1e-3, B=1e6, S=1, mode="lrt", dtype="bf16", seed=3, fuse_kl=True, state=dict(learningRate=1e-3),
                   meanState=dict(learningRate=1e-4), varState=dict(learningRate=5e-2), **cfg)
        eng = FusedMLP(opt)
        x = torch.empty(N, cfg["input_size"], dtype=torch.float32, device="cuda")
        fill_normal(x, 3, 4, 0, 0)
        t = eng.synthetic_targets(x, 0)
        eng.prepare()
        lines = []
        for s in range(steps):
            eng.resetGradients()
            eng.sample()
            eng.run(x, t)
            torch.cuda.synchronize()
            line = f"step {s} grads {h(eng.grads)}"
            eng.update()
            torch.cuda.synchronize()
            line += " shadows " + " ".join(h(v.mu_s.t.view(torch.int16)) + h(v.var_s.t.view(torch.int16)) for v in eng.vb)
            loss, acc = eng.loss_and_accuracy()
            line += f" loss {loss!r}"
            lines.append(line)
            out(f"rep {rep} {line}", flush=True)
        reps.append(lines)
        del eng
    return reps


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "wide"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    reps = run(name, steps, repeats)
    same = all(r == reps[0] for r in reps)
    print(f"{name}: {repeats} repetitions of {steps} training steps, identical: {same}")
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
