"""Worker of tests/test_dist_gpu.py::test_p2p_exchange_*: one rank of `world` ranks that SHARE the one GPU of the box (gloo carries
the 128-byte IPC handles and the check; the exchange itself is vbnn_p2p_*: peer-mapped arenas, reduce-scatter + all-gather
kernels, flag-page barriers). Each rank fills its arena with its own values, all-reduces several regions (aligned and odd
offsets, lengths that do not divide by the world), and the result must be, bitwise, the rank-ordered fp32 sum on every rank."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vbnn_amd.comm import P2PExchange            # noqa: E402
from vbnn_amd.nn import Context                  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    ctx = Context.get()
    n = 3_000_011
    ex = P2PExchange(ctx, rank, world, n)
    regions = [(0, 1_000_000), (1_000_000, 999_999), (1_999_999, 7), (2_000_006, 1_000_005)]
    ok = True
    for rep in range(3):
        g = torch.Generator(device="cuda").manual_seed(1000 * rep + rank)
        ex.arena.copy_(torch.randn(n, generator=g, device="cuda"))        # producer on the compute stream
        torch.cuda.synchronize()
        mine = ex.arena.cpu()
        everyone = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        dist.barrier()                                                      # nobody starts summing before everybody has read its own arena
        want = mine.clone()
        for off, cnt in regions:
            acc = everyone[0][off:off + cnt].clone()
            for p in range(1, world):
                acc += everyone[p][off:off + cnt]                          # rank order, fp32: the kernel's order
            want[off:off + cnt] = acc
        for off, cnt in regions:
            ex.allreduce(ex.arena[off:off + cnt])
        ex.finish()
        got = (ex.arena * 1.0).cpu()                                        # consumer on the compute stream, behind finish()
        gave_up = ex.gave_up()
        same = torch.equal(got.view(torch.int32), want.view(torch.int32))
        print(f"rank {rank} rep {rep}: bitwise {'equal' if same else 'DIFFERENT'} (barrier gave up at epoch {gave_up})", flush=True)
        ok = ok and same and gave_up == 0
        dist.barrier()
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    ex.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
