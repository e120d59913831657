"""The data-parallel exchange of the fused engine: a SUM all-reduce of each layer's gradient bucket.

Two transports behind one small interface (allreduce / finish / sync_finish):

  RcclExchange   the product path: RCCL driven through the C ABI (include/vbnn_hip.h: vbnn_comm_create /
                 vbnn_allreduce_grads / vbnn_comm_finish) -- the calls a LuaJIT host makes too (lua/FusedMLP.lua).
                 torch.distributed is used ONLY as the out-of-band channel that hands rank 0's RCCL unique id to the
                 other ranks (a file or a socket would do as well).
  TorchExchange  torch.distributed.all_reduce on the default (or a given) process group: the rehearsal transport
                 (gloo: several ranks sharing one GPU, CPU tests) and the fall-back if librccl cannot be bound --
                 which the bench line reports, it is never silent.
"""
import ctypes as C
import os

import torch

from . import _lib as L


class TorchExchange:
    def __init__(self, process_group=None, why=""):
        import torch.distributed as dist
        self.dist, self.pg, self.works = dist, process_group, []
        self.backend = f"torch.distributed/{dist.get_backend(process_group)}" + (f" ({why})" if why else "")

    def allreduce(self, bucket):
        self.works.append(self.dist.all_reduce(bucket, op=self.dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self):
        for w in self.works:
            w.wait()
        self.works = []
        for piece in getattr(self, "_poison", []):             # (reduce_scatter: what this rank was not given)
            piece.fill_(float("nan"))
        self._poison = []
        for dst, src in getattr(self, "_copyback", []):        # (all_gather: pieces gathered into temporaries)
            dst.copy_(src)
        self._copyback = []

    # the two halves as calls of their own (sharded-update exchange), with the C ABI's in-place slicing
    # (vbnn_comm_reduce_scatter: recvbuff = buf + rank * n_per_rank; the other slices are undefined afterwards): slice r is
    # REDUCED to rank r alone -- one dist.reduce per slice, the offset arithmetic of the product path -- and every slice this
    # rank does not own is then POISONED (NaN), so that host logic which reads a sum it was not given fails a rehearsal or a
    # CPU test instead of passing on an all-reduce's leftovers.
    def reduce_scatter(self, buf, n_per_rank):
        world, rank = self.dist.get_world_size(self.pg), self.dist.get_rank(self.pg)
        flat = buf.reshape(-1)
        assert flat.data_ptr() == buf.data_ptr() and flat.numel() == n_per_rank * world
        for r in range(world):
            self.works.append(self.dist.reduce(flat[r * n_per_rank:(r + 1) * n_per_rank], dst=self.dist.get_global_rank(self.pg, r) if self.pg is not None else r,
                                               op=self.dist.ReduceOp.SUM, group=self.pg, async_op=True))
        if not hasattr(self, "_poison"):
            self._poison = []
        self._poison += [flat[r * n_per_rank:(r + 1) * n_per_rank] for r in range(world) if r != rank]

    def all_gather(self, buf):
        world, rank = self.dist.get_world_size(self.pg), self.dist.get_rank(self.pg)
        flat = buf.reshape(-1)
        assert flat.data_ptr() == buf.data_ptr() and flat.numel() % world == 0
        n = flat.numel() // world
        mine = flat[rank * n:(rank + 1) * n].clone()
        pieces = [torch.empty_like(mine) for _ in range(world)]
        self.works.append(self.dist.all_gather(pieces, mine, group=self.pg, async_op=True))
        if not hasattr(self, "_copyback"):
            self._copyback = []
        for r in range(world):
            if r != rank:
                self._copyback.append((flat[r * n:(r + 1) * n], pieces[r]))


class RcclExchange:
    """vbnn_comm over RCCL. `process_group` (any backend) only carries the 128-byte unique id."""

    def __init__(self, ctx, rank, world, process_group=None):
        lib = L.lib()
        self.ctx, self.rank, self.world = ctx, rank, world
        uid = (C.c_ubyte * 128)()
        if rank == 0:
            L.check(lib.vbnn_comm_unique_id(uid))
        if world > 1:
            import torch.distributed as dist
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=0, group=process_group)
            uid = (C.c_ubyte * 128).from_buffer_copy(box[0])
        h = C.c_void_p()
        L.check(lib.vbnn_comm_create(ctx.h, rank, world, uid, C.byref(h)))
        self.h = h
        n = C.c_int()
        L.check(lib.vbnn_comm_info(h, None, None, C.byref(n)))
        assert n.value == world, (n.value, world)
        self.backend = "vbnn_comm/rccl"

    def allreduce(self, bucket):
        assert bucket.dtype in (torch.float32, torch.bfloat16) and bucket.is_contiguous()
        fn = L.lib().vbnn_allreduce_grads if bucket.dtype == torch.float32 else L.lib().vbnn_allreduce_grads_bf16
        L.check(fn(self.h, C.c_void_p(bucket.data_ptr()), bucket.numel()))

    def finish(self):
        L.check(L.lib().vbnn_comm_finish(self.h))

    def reduce_scatter(self, buf, n_per_rank):
        assert buf.dtype == torch.float32 and buf.is_contiguous() and buf.numel() == n_per_rank * self.world
        L.check(L.lib().vbnn_comm_reduce_scatter(self.h, C.c_void_p(buf.data_ptr()), n_per_rank))

    def all_gather(self, buf):
        nbytes = buf.numel() * buf.element_size()
        assert buf.is_contiguous() and nbytes % self.world == 0
        L.check(L.lib().vbnn_comm_all_gather(self.h, C.c_void_p(buf.data_ptr()), nbytes // self.world))

    def gather_u64(self, value):
        """Every rank's 64-bit word, exchanged over the communicator itself (bench.py: ranks_seen)."""
        dev = self.ctx.device
        mine = torch.tensor([value], dtype=torch.int64, device=dev)
        out = torch.zeros(self.world, dtype=torch.int64, device=dev)
        L.check(L.lib().vbnn_comm_allgather_u64(self.h, C.c_void_p(mine.data_ptr()), C.c_void_p(out.data_ptr())))
        self.finish()
        torch.cuda.synchronize(dev)
        return [int(v) for v in out.tolist()]

    def close(self):
        if self.h:
            L.check(L.lib().vbnn_comm_destroy(self.h))
            self.h = None


class _DevArray:
    """A device allocation of the library's as something torch.as_tensor can wrap without copying."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class P2PExchange:
    """vbnn_p2p_*: direct reduce-scatter + all-gather over peer-mapped arenas (xGMI), no collective library. It OWNS the
    gradient arena (one exportable hipMalloc allocation): the engine takes `self.arena` as its flat gradient tensor.
    `process_group` (any backend) only carries the 128-byte IPC handles, once."""

    def __init__(self, ctx, rank, world, arena_floats, process_group=None):
        """Collective over `process_group` when world > 1 -- and collective in FAILURE too: a rank that cannot allocate / export /
        map says so in the handle exchange and in the agreement after the mapping, so that every rank raises together instead of
        one rank leaving the others in an all-gather or a barrier (as comm.make_exchange does for RCCL)."""
        lib = L.lib()
        self.ctx, self.rank, self.world = ctx, rank, world
        h, arena = C.c_void_p(), C.c_void_p()
        mine = (C.c_ubyte * 128)()
        st = lib.vbnn_p2p_create(ctx.h, rank, world, arena_floats, C.byref(h), C.byref(arena), mine)
        err = None if st == L.OK else (lib.vbnn_last_error() or b"?").decode()
        self.h, self._ptr = (h, arena.value) if err is None else (None, None)
        if world > 1:
            import torch.distributed as dist
            box = [None] * world
            dist.all_gather_object(box, (err, bytes(mine)), group=process_group)
            bad = [(r, e) for r, (e, _) in enumerate(box) if e is not None]
            if bad:
                self.close()
                raise L.VbnnError(f"vbnn_p2p_create failed on rank(s) {[r for r, _ in bad]}: {bad[0][1]}")
            allh = (C.c_ubyte * (128 * world)).from_buffer_copy(b"".join(hb for _, hb in box))
            st = lib.vbnn_p2p_connect(h, allh)
            err = None if st == L.OK else (lib.vbnn_last_error() or b"?").decode()
            box = [None] * world
            dist.all_gather_object(box, err, group=process_group)        # (also the host barrier before the first exchange: every rank has
            bad = [(r, e) for r, e in enumerate(box) if e is not None]   # mapped every arena and flag page when this returns)
            if bad:
                self.close()
                raise L.VbnnError(f"vbnn_p2p_connect failed on rank(s) {[r for r, _ in bad]}: {bad[0][1]}")
        else:
            if err is not None:
                raise L.VbnnError(f"vbnn_p2p_create: {err}")
            L.check(lib.vbnn_p2p_connect(h, None))
        self._keep = _DevArray(self._ptr, arena_floats)
        self.arena = torch.as_tensor(self._keep, device=ctx.device)
        assert self.arena.data_ptr() == self._ptr and self.arena.dtype == torch.float32
        self.backend = "vbnn_p2p/ipc"

    def allreduce(self, bucket):
        assert bucket.dtype == torch.float32 and bucket.is_contiguous()
        off = bucket.data_ptr() - self._ptr
        assert off >= 0 and off % 4 == 0 and off // 4 + bucket.numel() <= self.arena.numel(), "the bucket must lie in the exchange's arena"
        L.check(L.lib().vbnn_p2p_allreduce(self.h, off // 4, bucket.numel()))

    def finish(self):
        L.check(L.lib().vbnn_p2p_finish(self.h))

    def _region(self, buf):
        off = buf.data_ptr() - self._ptr
        nbytes = buf.numel() * buf.element_size()
        assert buf.is_contiguous() and off >= 0 and off % 4 == 0 and nbytes % 4 == 0 and off + nbytes <= self.arena.numel() * 4, \
            "the region must lie in the exchange's arena (whole floats)"
        return off // 4, nbytes // 4

    def reduce_scatter(self, buf, n_per_rank):
        off, n = self._region(buf)
        assert buf.dtype == torch.float32 and n == n_per_rank * self.world
        L.check(L.lib().vbnn_p2p_reduce_scatter(self.h, off, n_per_rank))

    def all_gather(self, buf):
        off, n = self._region(buf)
        assert n % self.world == 0
        L.check(L.lib().vbnn_p2p_all_gather(self.h, off, n // self.world))

    def set_grid(self, rs_blocks=0, ag_blocks_per_peer=0):
        """Workgroups of the reduce-scatter / of the all-gather per peer (0: keep). The data kernels co-reside with the GEMMs they
        overlap (no LDS, <= 48 VGPRs): small grids, csrc/p2p.hip."""
        L.check(L.lib().vbnn_p2p_set_grid(self.h, int(rs_blocks), int(ag_blocks_per_peer)))

    def standin(self, sim_world, inbound_GBps=0.0):
        """LAB, one rank: allreduce() runs what a rank of a sim_world-rank exchange runs, paced to `inbound_GBps` of link bandwidth
        (tools/overlap_standin.py). The arena holds nothing meaningful afterwards; 0 switches it off."""
        L.check(L.lib().vbnn_p2p_standin(self.h, int(sim_world), float(inbound_GBps)))

    def gave_up(self):
        """Blocks until the exchange stream is idle; the epoch of a barrier that timed out waiting for a peer, or 0."""
        g = C.c_uint()
        L.check(L.lib().vbnn_p2p_status(self.h, None, None, C.byref(g)))
        return g.value

    def check(self):
        """Raise if a barrier of this rank gave up (the exchange's data kernels are no-ops from that point on: the arena
        holds this rank's own gradients, not the sums). Blocks on the exchange stream: call it where the sums are about to
        be consumed -- before an update, at the end of a timed block -- not inside the step."""
        epoch = self.gave_up()
        if epoch:
            raise RuntimeError(f"vbnn_p2p: barrier {epoch} of rank {self.rank} gave up waiting for a peer rank -- the gradient "
                               "arena holds this rank's own gradients, not the sums (vbnn_p2p_clear_status re-arms the exchange "
                               "once the ranks are in step again)")

    def set_timeout(self, seconds):
        L.check(L.lib().vbnn_p2p_set_timeout(self.h, float(seconds)))

    def clear_status(self):
        L.check(L.lib().vbnn_p2p_clear_status(self.h))

    def close(self):
        if self.h:
            self.arena = None
            L.check(L.lib().vbnn_p2p_destroy(self.h))
            self.h = None


def make_exchange(ctx, rank, world, process_group=None):
    """RCCL through the C ABI unless VBNN_EXCHANGE=torch (or the torch backend is not nccl: a gloo rehearsal has no
    device per rank for RCCL); a failure to bind librccl falls back to torch.distributed and says so.
    The decision is COLLECTIVE: every rank probes the binding (vbnn_comm_unique_id loads librccl and asks it for an id; only
    rank 0's is used), the ranks agree on MIN(ok) over the out-of-band process group, and either all of them build the RCCL
    communicator or all of them fall back -- a rank that cannot bind never leaves the others waiting in a broadcast or
    inside ncclCommInitRank. The same agreement follows vbnn_comm_create."""
    import torch.distributed as dist
    want = os.environ.get("VBNN_EXCHANGE", "")
    if want == "torch":
        return TorchExchange(process_group, "VBNN_EXCHANGE=torch")
    have_pg = dist.is_available() and dist.is_initialized()
    if have_pg and dist.get_backend(process_group) != "nccl" and want != "rccl":
        return TorchExchange(process_group)
    if world == 1 or not have_pg:
        try:
            return RcclExchange(ctx, rank, world, process_group)
        except (L.VbnnError, OSError, AssertionError) as e:
            if not have_pg:
                raise
            return TorchExchange(process_group, f"vbnn_comm unavailable: {e}")

    def agree(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32,
                            device=ctx.device if dist.get_backend(process_group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=process_group)
        return bool(flag.item())

    why = ""
    try:
        L.check(L.lib().vbnn_comm_unique_id((C.c_ubyte * 128)()))          # probe: is librccl bound on THIS rank?
        ok = True
    except (L.VbnnError, OSError) as e:
        ok, why = False, str(e)
    if not agree(ok):
        return TorchExchange(process_group, f"vbnn_comm unavailable on some rank{': ' + why if why else ''}")
    ex = None
    try:
        ex = RcclExchange(ctx, rank, world, process_group)
    except (L.VbnnError, OSError, AssertionError) as e:
        why = str(e)
    if not agree(ex is not None):
        if ex is not None:
            ex.close()
        return TorchExchange(process_group, f"vbnn_comm_create failed on some rank{': ' + why if why else ''}")
    return ex
