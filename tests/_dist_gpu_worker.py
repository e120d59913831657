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
    S = int(sys.argv[7]) if len(sys.argv) > 7 else 1                  # sequential Monte-Carlo draws per minibatch (main.lua:32-37)
    opt = dict(var_init=1e-3, mu_init=1, B=1e6, S=S, mode="lrt", dtype=dtype, seed=3, input_size=I0, hidden=hidden,
               n_classes=10, fuse_kl=True, exchange_dtype=sys.argv[6] if len(sys.argv) > 6 else "f32")
    eng = FusedMLP(opt, world_size=world, rank=rank)
    n_loc = N // world
    x = torch.empty(n_loc, I0, dtype=torch.float32, device="cuda")
    fill_normal(x, 3, 4, 0, 0, row0=rank * n_loc)                      # the global minibatch's rows of this rank
    t = ((torch.arange(n_loc, device="cuda", dtype=torch.int64) + rank * n_loc) * 7 % 10).to(torch.int32)
    # The ranks OVERLAP on the GPU (two processes' kernels interleaved on one device). In r02 about one such run in twenty
    # computed one 8-row A piece of one K step of a pipelined GEMM wrong, and the ranks were made to take turns. The cause
    # was in the kernel (gemm_v2.h, v2_wait_barrier): in the DUAL SCHED-0 instantiations -- the ones this 512-row shard
    # takes -- hipcc had sunk the wait for a step's last two LDS reads below the next barrier, so they were in flight
    # while the refill of their stage was issued. Fixed there and held by tests/test_kernel_hazards.py on the generated
    # code; the ranks run concurrently again, and tests/test_parity_gpu.py has the in-process form (two engines, two streams).
    for _ in range(2):                                                  # twice: the second step reuses every buffer
        eng.resetGradients(); eng.prepare()
        for _ in range(S):                                              # the draws accumulate; the buckets leave with the last one
            eng.sample()
            eng.run(x, t)
        eng.finish()
    loss, _ = eng.loss_and_accuracy()
    print(f"rank {rank}: local loss {loss!r} (exchange: {eng.comm_backend()})", flush=True)
    if os.environ.get("VBNN_TEST_DIAG") == "1":     # where does a deviating loss come from: the activations or the reduction?
        out = eng.out.double().cpu().numpy()                             # log-probabilities of the last step
        tt = t.cpu().numpy()
        inv_n = 1.0 / (n_loc * world)
        from_out = float(-(out[np.arange(n_loc), tt]).sum() * inv_n)
        h_sum = [float(eng.h_s.t.float().double().sum().item())] + [float(v.x_s.t.float().double().sum().item()) for v in eng.vb]
        print(f"rank {rank}: diag loss-from-out {from_out!r} logits-sum {float(eng.logits.double().sum().item())!r} "
              f"out-sum {float(out.sum())!r} h-sums {h_sum}", flush=True)
        dump = os.environ.get("VBNN_TEST_DUMP")     # directory: keep the last layer's output of one good and of every deviating run
        if dump and hidden == [4096, 4096] and N == 1024:
            good = {0: 2.271684411796741, 1: 2.4192868926911615}[rank]
            kind = "good" if loss == good else f"bad{os.getpid()}"
            path = os.path.join(dump, f"h_rank{rank}_{kind}.npy")
            if kind != "good" or not os.path.exists(path):
                np.save(path, eng.h_s.t[:, :4096].float().cpu().numpy())
    lt = torch.tensor([loss], dtype=torch.float64)
    dist.all_reduce(lt)
    if rank == 0:
        np.save(out + ".grads.npy", eng.grads.cpu().numpy())
        np.save(out + ".loss.npy", lt.numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
