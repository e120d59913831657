"""Worker of tests/test_dist_gpu.py::test_sharded_update_*: one rank of `world` ranks sharing the box's one GPU. Runs `steps`
TRAINING steps (run + update) of the fused engine in the given exchange mode -- "allreduce" (north_star's: every rank updates
every parameter) or "sharded" (reduce-scatter, update of this rank's rows, all-gather of the operand shadows) -- over the exchange
VBNN_EXCHANGE selects (p2p: vbnn_p2p_* on peer-mapped arenas; otherwise torch.distributed / gloo), then brings the fp32 parameters
together and saves, from rank 0, everything the next step would read: means, lvars, biases, the final Linear, the shadows, the
statistics. Every rank also checks that its shadows and statistics are bitwise rank 0's."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vbnn_amd.engine import FusedMLP            # noqa: E402
from vbnn_amd.nn import fill_normal             # noqa: E402


def main():
    out, mode, dtype = sys.argv[1], sys.argv[2], sys.argv[3]
    hidden, I0, n_loc, steps = [int(h) for h in sys.argv[4].split(",")], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    opt = dict(var_init=1e-3, mu_init=1, B=50.0, S=1, mode="lrt", dtype=dtype, seed=3, input_size=I0, hidden=hidden, n_classes=10,
               fuse_kl=True, kl_in_update=True, exchange_mode=mode, state=dict(learningRate=1e-3), meanState=dict(learningRate=1e-4),
               varState=dict(learningRate=5e-2))
    eng = FusedMLP(opt, world_size=world, rank=rank, force_reduce=True)       # (a world of one runs the exchange code as well)
    for v in eng.vb:                        # posterior variances away from the prior's: a KL gradient that matters (B = 50)
        v.lvars.add_(0.7 * torch.sin(torch.arange(v.lvars.numel(), device="cuda", dtype=torch.float32)).view_as(v.lvars))
    x = torch.empty(n_loc, I0, dtype=torch.float32, device="cuda")
    fill_normal(x, 3, 4, 0, 0, row0=rank * n_loc)
    t = ((torch.arange(n_loc, device="cuda", dtype=torch.int64) + rank * n_loc) * 7 % 10).to(torch.int32)
    eng.prepare()
    losses = []
    for _ in range(steps):
        eng.resetGradients(); eng.sample(); eng.run(x, t); eng.finish()
        loss, _ = eng.loss_and_accuracy()
        losses.append(loss)
        eng.update(opt)
    torch.cuda.synchronize()
    eng.check_exchange()
    eng.gather_parameters()
    # what every rank holds for the next forward must be the same bits everywhere
    mine = torch.cat([torch.cat([v.mu_s.t.view(torch.int16).reshape(-1).to(torch.int64) if dtype == "bf16" else v.mu_s.t.view(torch.int32).reshape(-1).to(torch.int64),
                                 v.var_s.t.view(torch.int16).reshape(-1).to(torch.int64) if dtype == "bf16" else v.var_s.t.view(torch.int32).reshape(-1).to(torch.int64),
                                 v.stats.view(torch.int64)]) for v in eng.vb])
    digest = torch.stack([mine.sum(), (mine * torch.arange(1, mine.numel() + 1, device="cuda")).sum()]).cpu()
    seen = [torch.empty_like(digest) for _ in range(world)]
    dist.all_gather(seen, digest)
    same = all(torch.equal(s, seen[0]) for s in seen)
    print(f"rank {rank}: mode {mode}, exchange {eng.comm_backend()}, losses {losses!r}, shadows + statistics equal across ranks: {same}", flush=True)
    lt = torch.tensor(losses, dtype=torch.float64)
    dist.all_reduce(lt)
    if rank == 0:
        np.savez(out, losses=lt.numpy(), w3=eng.weight3.cpu().numpy(), b3=eng.bias3.cpu().numpy(),
                 **{f"means{k}": v.means.cpu().numpy() for k, v in enumerate(eng.vb)},
                 **{f"lvars{k}": v.lvars.cpu().numpy() for k, v in enumerate(eng.vb)},
                 **{f"bias{k}": v.bias.cpu().numpy() for k, v in enumerate(eng.vb)},
                 **{f"mu_s{k}": v.mu_s.t.float().cpu().numpy() for k, v in enumerate(eng.vb)},
                 **{f"var_s{k}": v.var_s.t.float().cpu().numpy() for k, v in enumerate(eng.vb)},
                 **{f"stats{k}": v.stats.cpu().numpy() for k, v in enumerate(eng.vb)})
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
