#!/usr/bin/env python3
"""What does the data-parallel exchange cost the launches it is meant to hide behind? Priced on ONE GPU.

The wide bench step in the order an exchange needs (layer by layer, the two-launch accGradParameters with its early d/dlvars
message), a world of ONE, and vbnn_p2p's STAND-IN of an 8-rank exchange (vbnn_p2p_standin: the same barriers, data kernels, grids,
register footprint, high-priority stream and events as a real rank's; the "peers" are this arena shifted by whole chunks; every
phase paced to the wall time the links would need for the bytes a rank receives). For each arm: the step (median of 3 blocks of 20),
and in one more block HIP events around every overlapped launch. Arms: no exchange calls at all; the world-of-one exchange (events
only); the stand-in at several grids, paced at 7 x 110 GB/s of inbound link bandwidth and unpaced.

    python tools/overlap_standin.py [out.json]          (on the GPU box; ~1 minute)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                              # noqa: E402
from vbnn_amd import _lib as L                            # noqa: E402
from vbnn_amd.engine import FusedMLP                      # noqa: E402
from vbnn_amd.nn import fill_normal                       # noqa: E402

STEPS, BLOCKS = 20, 3


def one_arm():
    """The default arm alone, 40 steps (for a kernel trace: tools/standin_timeline.py)."""
    opt = dict(var_init=1e-3, B=1e6, S=1, mode="lrt", dtype="bf16", seed=3, input_size=784, hidden=[4096, 4096], n_classes=10,
               fuse_kl=True, exchange="p2p")
    eng = FusedMLP(opt, world_size=1, rank=0, force_reduce=True)
    x = torch.empty(4096, 784, dtype=torch.float32, device="cuda")
    fill_normal(x, 3, L.STREAM_DATA, 0, 0)
    t = eng.synthetic_targets(x, 0)
    eng.prepare()
    eng.exchange().standin(8, 770.0)
    for _ in range(40):
        eng.resetGradients(); eng.sample(); eng.run(x, t); eng.finish()
    torch.cuda.synchronize()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--one-arm":
        return one_arm()
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05_overlap_standin.json")
    N = 4096
    opt = dict(var_init=1e-3, B=1e6, S=1, mode="lrt", dtype="bf16", seed=3, input_size=784, hidden=[4096, 4096], n_classes=10,
               fuse_kl=True, exchange="p2p")
    eng = FusedMLP(opt, world_size=1, rank=0, force_reduce=True)
    x = torch.empty(N, 784, dtype=torch.float32, device="cuda")
    fill_normal(x, 3, L.STREAM_DATA, 0, 0)
    t = eng.synthetic_targets(x, 0)
    ex = eng.exchange()

    def step():
        eng.resetGradients(); eng.sample(); eng.run(x, t); eng.finish()

    def measure():
        for _ in range(5):
            step()
        wall = []
        for _ in range(BLOCKS):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(STEPS):
                step()
            torch.cuda.synchronize()
            wall.append((time.perf_counter() - t0) / STEPS * 1e3)
        eng.probe = ("all", {})
        for _ in range(STEPS):
            step()
        torch.cuda.synchronize()
        probe, eng.probe = eng.probe, None
        per = {k: round(sum(a.elapsed_time(b) for a, b in ev) / len(ev) * 1e3, 1) for k, ev in sorted(probe[1].items())}
        return {"ms_per_step": round(sorted(wall)[len(wall) // 2], 4), "blocks_ms": [round(w, 4) for w in wall], "launch_us": per}

    eng.prepare()
    res = {"workload": "784-4096-4096-10, batch 4096, bf16, LRT, S = 1; layer-by-layer backward with the early d/dlvars message",
           "messages_bytes": [int(b.numel() * 4) for b in eng.buckets()],
           "stand_in": "vbnn_p2p_standin(8, inbound GB/s): csrc/p2p.hip -- what ONE rank of an 8-rank direct exchange runs, against its own arena",
           "arms": []}

    def arm(name, **kw):
        r = measure()
        r.update(name=name, **kw)
        res["arms"].append(r)
        print(name, r["ms_per_step"], r["launch_us"], flush=True)

    eng.skip_exchange = True
    arm("no exchange calls")
    eng.skip_exchange = False
    ex.standin(0)
    arm("world of one (events and stream hand-offs only)")
    for gbps in (770.0, 0.0):
        for rs, ag in ((1024, 256), (256, 37), (128, 16), (64, 8), (32, 4), (16, 2), (8, 1)) if gbps else ((1024, 256), (256, 37), (64, 8), (16, 2)):
            ex.set_grid(rs, ag)
            ex.standin(8, gbps)
            arm(f"stand-in 8 ranks, grid rs {rs} / ag {ag} x 7 peers, " + (f"paced {gbps:.0f} GB/s inbound" if gbps else "unpaced (local-memory speed)"),
                rs_blocks=rs, ag_blocks_per_peer=ag, inbound_GBps=gbps)
    # the library's default grids at three link rates: what the prediction is worth if the links deliver less (or more) than 110 GB/s each
    ex.set_grid(256, 32)
    for gbps in (490.0, 770.0, 1050.0):
        ex.standin(8, gbps)
        arm(f"stand-in 8 ranks, default grid rs 256 / ag 32 x 7 peers, paced {gbps:.0f} GB/s inbound ({gbps / 7:.0f} GB/s per link)",
            rs_blocks=256, ag_blocks_per_peer=32, inbound_GBps=gbps)
    ex.standin(0)
    base = res["arms"][0]["ms_per_step"]
    for a in res["arms"]:
        a["ms_over_no_exchange"] = round(a["ms_per_step"] - base, 4)
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
