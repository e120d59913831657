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


def delayed_rank():
    """A rank that arrives late: with the barrier's bound set to 0.5 s, rank 1 issues its all-reduce 3 s after rank 0.
    Rank 0's first barrier gives up: its data kernels become no-ops (its arena is bitwise what it was), check() raises
    BEFORE anything consumes the arena, and rank 1 -- whose barriers all find rank 0's epochs already signalled -- learns of
    the failure from the dead word rank 0 left in its flag page: its arena is untouched as well. After a host barrier and
    clear_status on both ranks the same exchange sums again."""
    import time
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2
    torch.cuda.set_device(0)
    ctx = Context.get()
    n = 1_000_003
    ex = P2PExchange(ctx, rank, world, n)
    ex.set_timeout(0.5)
    g = torch.Generator(device="cuda").manual_seed(77 + rank)
    ex.arena.copy_(torch.randn(n, generator=g, device="cuda"))
    torch.cuda.synchronize()
    mine = ex.arena.cpu()
    dist.barrier()
    if rank == 1:
        time.sleep(3.0)
    ex.allreduce(ex.arena[:n])
    ex.finish()
    raised = False
    try:
        ex.check()
    except RuntimeError as e:
        raised = "gave up" in str(e)
    untouched = torch.equal((ex.arena * 1.0).cpu().view(torch.int32), mine.view(torch.int32))
    print(f"rank {rank}: check raised {raised}, arena untouched {untouched} (epoch {ex.gave_up()})", flush=True)
    ok = raised and untouched
    # re-arm: every exchange stream has drained (gave_up synchronised), a host barrier, clear on every rank, a host barrier
    dist.barrier()
    ex.clear_status()
    ex.set_timeout(20.0)
    dist.barrier()
    everyone = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine)
    ex.allreduce(ex.arena[:n])
    ex.finish()
    ex.check()
    got = (ex.arena * 1.0).cpu()
    same = torch.equal(got.view(torch.int32), (everyone[0] + everyone[1]).view(torch.int32))
    print(f"rank {rank}: after clear_status the sum is bitwise {'equal' if same else 'DIFFERENT'}", flush=True)
    flag = torch.tensor([1 if (ok and same) else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    ex.close()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "delayed":
        return delayed_rank()
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
        # r05: the data kernels' grids are small by design (they co-reside with the GEMMs): a tiny grid walks every region in many
        # grid-stride rounds, a large one in a single round -- the same bits either way
        ex.set_grid(*[(0, 0), (3, 2), (1024, 256)][rep])
        for off, cnt in regions:
            ex.allreduce(ex.arena[off:off + cnt])
        if rep == 2:
            # the SAME region again with no finish() in between (a timing loop does this): the exchange must put its exit barrier in
            # front -- a peer may still be gathering from the chunk this rank is about to reduce into. Every rank holds the sum S:
            # the second all-reduce leaves (S + S) + ... in rank order
            off, cnt = regions[0]
            ex.allreduce(ex.arena[off:off + cnt])
            acc = want[off:off + cnt].clone()
            for p in range(1, world):
                acc += want[off:off + cnt]
            want[off:off + cnt] = acc
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
