"""Worker of tests/test_dist_gpu.py: one rank of a world of 2 on ONE GPU (gloo carries the collectives, so two ranks may
share the device -- RCCL refuses that). Runs the engine's real multi-rank path: rows [rank N/2, (rank+1) N/2), async
all-reduces issued from inside the step, finish(); rank 0 saves the reduced gradient arena and the summed loss."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vbnn_amd.engine import FusedMLP            # noqa: E402
from vbnn_amd.nn import fill_normal             # noqa: E402


def main():
    out, dtype = sys.argv[1], sys.argv[2]
    hidden, I0, N = [int(h) for h in sys.argv[3].split(",")], int(sys.argv[4]), int(sys.argv[5])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    opt = dict(var_init=1e-3, mu_init=1, B=1e6, S=1, mode="lrt", dtype=dtype, seed=3, input_size=I0, hidden=hidden,
               n_classes=10, fuse_kl=True)
    eng = FusedMLP(opt, world_size=world, rank=rank)
    n_loc = N // world
    x = torch.empty(n_loc, I0, dtype=torch.float32, device="cuda")
    fill_normal(x, 3, 4, 0, 0, row0=rank * n_loc)                      # the global minibatch's rows of this rank
    t = ((torch.arange(n_loc, device="cuda", dtype=torch.int64) + rank * n_loc) * 7 % 10).to(torch.int32)
    for _ in range(2):                                                  # twice: the second step reuses every buffer
        eng.resetGradients(); eng.prepare(); eng.sample()
        eng.run(x, t)
        eng.finish()
    loss, _ = eng.loss_and_accuracy()
    print(f"rank {rank}: local loss {loss!r} (exchange: {eng.comm_backend()})", flush=True)
    lt = torch.tensor([loss], dtype=torch.float64)
    dist.all_reduce(lt)
    if rank == 0:
        np.save(out + ".grads.npy", eng.grads.cpu().numpy())
        np.save(out + ".loss.npy", lt.numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
