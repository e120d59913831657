"""world_size-2 gloo test (CPU) of the data-parallel recipe the engine uses on N GPUs:

  - rank r takes rows [r N/G, (r+1) N/G) of the global minibatch, with z addressed by GLOBAL row;
  - the criterion divides by the GLOBAL batch (inv_n = 1 / (N_local * world));
  - the fused KL gradient carries kl_scale = 1 / world;
  - a SUM all-reduce over [grad_mu | grad_lv | gradBias] per layer + the final Linear's gradients.

The per-rank compute here is the oracle (tests may use it); what is under test is the ENGINE's host logic, called,
not restated: vbnn_amd/partition.py (row shards, criterion / KL scales, arena layout, message order -- the functions
FusedMLP computes with) and vbnn_amd/comm.py's TorchExchange (the transport FusedMLP uses under gloo). The reduced
arena must equal the single-process result on the whole batch.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    from oracle import vbnn_oracle as vo
    opt = dict(var_init=1e-3, mu_init=1, B=50.0, S=1, seed=3, mode="lrt", input_size=12, hidden=[10, 8], n_classes=5)
    onet = vo.OracleMLP(opt)
    layers = []
    for om in onet.vb:
        om.means[:] = om.weight
        layers.append(dict(means=om.means.copy(), lvars=om.lvars.copy(), bias=om.bias.copy()))
    N = 16
    x = vo.fill_normal(N, 12, 3, vo.STREAM_DATA, 0, 0)
    t = (np.arange(N) * 3 % 5).astype(np.int32)
    return opt, layers, onet.last.weight.copy(), onet.last.bias.copy(), x, t, [om.O for om in onet.vb]


def _rank_step(rank, world, layers, w3, b3, x, t, outs, B):
    """One engine rank: its rows (partition.shard_rows), noise by GLOBAL row, the criterion and KL scales of
    partition.scales, its gradients written into an arena laid out by partition.arena_layout."""
    from oracle import vbnn_oracle as vo
    from oracle.ref_numpy import emulate_lrt_step
    from vbnn_amd import partition
    N = x.shape[0]
    r0, n_loc = partition.shard_rows(N, world, rank)
    sc = partition.scales(n_loc, world)
    xs, ts = x[r0:r0 + n_loc], t[r0:r0 + n_loc]
    zetas = [vo.fill_normal(n_loc, O, 3, vo.STREAM_ZETA, k, 1, r0).astype(np.float64) for k, O in enumerate(outs)]
    loss, res, gw3, gb3 = emulate_lrt_step(layers, w3, b3, xs, ts, zetas, S=1.0, B=B, inv_n=sc["inv_n"])
    sizes = [x.shape[1]] + list(outs)
    lay, fin, total, buckets = partition.arena_layout(sizes, w3.shape[0], early_lv=[False] + [True] * (len(sizes) - 2))
    arena = np.zeros(total, np.float64)

    def put(span, a):
        assert a.size == span[1]
        arena[span[0]:span[0] + span[1]] = a.ravel()

    for d, layer, r in zip(lay, layers, res):
        vh = float(np.sum(np.exp(layer["lvars"].astype(np.float64)) + layer["means"].astype(np.float64) ** 2) / layer["means"].size)
        var = np.exp(layer["lvars"].astype(np.float64))
        kl_mu = layer["means"] / (B * vh)
        kl_lv = (var / vh - 1.0) / (2 * B)
        # the fused epilogue: likelihood / S + kl_scale * KL
        put(d["mu"], r["grad_mu"] - kl_mu + sc["kl_scale"] * kl_mu)
        put(d["lv"], r["grad_lv"] - kl_lv + sc["kl_scale"] * kl_lv)
        put(d["bias"], r["gradBias"])
    put(fin["weight"], gw3)
    put(fin["bias"], gb3)
    return loss, arena, buckets


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from vbnn_amd import partition
    from vbnn_amd.comm import TorchExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    opt, layers, w3, b3, x, t, outs = _problem()
    loss, flat, buckets = _rank_step(rank, world, layers, w3, b3, x, t, outs, opt["B"])
    buf = torch.from_numpy(flat.copy())
    lbuf = torch.tensor([loss], dtype=torch.float64)
    ex = TorchExchange()
    cov = sorted(buckets)
    assert cov[0][0] == 0 and cov[-1][1] == flat.size and all(a[1] == b[0] for a, b in zip(cov, cov[1:]))   # the whole arena, once
    partition.exchange_step(buf, buckets, ex)                 # the messages of FusedMLP.run + finish, in issue order
    ex.allreduce(lbuf); ex.finish()
    if rank == 0:
        q.put((float(lbuf[0]), buf.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_allreduce_equals_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    loss2, flat2 = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    opt, layers, w3, b3, x, t, outs = _problem()
    loss1, flat1, _ = _rank_step(0, 1, layers, w3, b3, x, t, outs, opt["B"])
    assert abs(loss1 - loss2) <= 1e-6 * abs(loss1)          # the criterion sums fp32 log-probabilities
    # not bitwise: the per-rank dgemm sees a different row count, its last-bit differences move a few fp32
    # roundings of the activations (1e-7 relative), exactly as on the GPUs
    np.testing.assert_allclose(flat2, flat1, rtol=5e-6, atol=1e-8)


def _adam(x, g, m, v, t, lr):
    """optim.adam in fp32 NumPy (the arithmetic of vbnn_adam_step, one array at a time): same function for both modes below."""
    f = np.float32
    b1, b2, eps = f(0.9), f(0.999), f(1e-8)
    m[:] = b1 * m + (f(1) - b1) * g
    v[:] = b2 * v + (f(1) - b2) * g * g
    step = f(lr * np.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t))
    x -= step * m / (np.sqrt(v) + eps)


def _sharded_worker(rank, world, port, q, mode):
    """One rank of `world` gloo ranks on the CPU running TWO training steps with the engine's host logic around oracle compute:
    mode "allreduce" -- partition.exchange_step over the arena's messages, then every rank updates every row; mode "sharded" --
    partition.sharded_plan / sharded_exchange_grads (reduce-scatter by layer rows + the small all-reduce) over comm.TorchExchange,
    the update on THIS rank's rows (partition.layer_row_shard) with its slice of the Adam state, the all-gather of what the next
    forward reads (here: the new means / lvars rows themselves and the slices' statistics). Rank 0 reports the parameters."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import vbnn_oracle as vo
    from oracle.ref_numpy import emulate_lrt_step
    from vbnn_amd import partition
    from vbnn_amd.comm import TorchExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    opt = dict(var_init=1e-3, mu_init=1, B=50.0, S=1, seed=3, mode="lrt", input_size=12, hidden=[12, 6], n_classes=5)
    onet = vo.OracleMLP(opt)
    layers = []
    for om in onet.vb:
        om.means[:] = om.weight
        layers.append(dict(means=om.means.copy(), lvars=om.lvars.copy(), bias=om.bias.copy()))
    w3, b3 = onet.last.weight.copy(), onet.last.bias.copy()
    outs = [om.O for om in onet.vb]
    N = 12 * world
    x = vo.fill_normal(N, 12, 3, vo.STREAM_DATA, 0, 0)
    t = (np.arange(N) * 3 % 5).astype(np.int32)
    sizes = [12] + outs
    lay, fin, total, buckets = partition.arena_layout(sizes, 5)
    plan = partition.sharded_plan(lay, world)
    ex = TorchExchange()
    r0, n_loc = partition.shard_rows(N, world, rank)
    sc = partition.scales(n_loc, world)
    adam = [{k: np.zeros_like(l["means"]) for k in ("mm", "vm", "ml", "vl")} for l in layers]
    B = opt["B"]
    for step in range(1, 3):
        zetas = [vo.fill_normal(n_loc, O, 3, vo.STREAM_ZETA, k, step, r0).astype(np.float64) for k, O in enumerate(outs)]
        # the arena holds the LIKELIHOOD parts (kl_scale = 0: the update adds the KL gradient, opt.kl_in_update)
        loss, res, gw3, gb3 = emulate_lrt_step(layers, w3, b3, x[r0:r0 + n_loc], t[r0:r0 + n_loc], zetas, S=1.0, B=B, inv_n=sc["inv_n"], kl_scale=0.0)
        arena = np.zeros(total, np.float32)
        for d, r in zip(lay, res):
            arena[d["mu"][0]:d["mu"][0] + d["mu"][1]] = r["grad_mu"].astype(np.float32).ravel()
            arena[d["lv"][0]:d["lv"][0] + d["lv"][1]] = r["grad_lv"].astype(np.float32).ravel()
            arena[d["bias"][0]:d["bias"][0] + d["bias"][1]] = r["gradBias"].astype(np.float32)
        arena[fin["weight"][0]:fin["weight"][0] + fin["weight"][1]] = gw3.astype(np.float32).ravel()
        arena[fin["bias"][0]:fin["bias"][0] + fin["bias"][1]] = gb3.astype(np.float32)
        buf = torch.from_numpy(arena)
        if mode == "allreduce":
            partition.exchange_step(buf, buckets, ex)
        else:
            partition.sharded_exchange_grads(buf, plan, world, ex)
        w3 -= np.float32(1e-3) * arena[fin["weight"][0]:fin["weight"][0] + fin["weight"][1]].reshape(w3.shape)
        b3 -= np.float32(1e-3) * arena[fin["bias"][0]:fin["bias"][0] + fin["bias"][1]]
        for k, (d, l) in enumerate(zip(lay, layers)):
            O, I = d["O"], d["I"]
            l["bias"] -= np.float32(1e-3) * arena[d["bias"][0]:d["bias"][0] + O]
            rows = slice(0, O) if mode == "allreduce" else slice(*(lambda a: (a[0], a[0] + a[1]))(partition.layer_row_shard(O, world, rank)))
            vh = np.float32(np.sum(np.exp(l["lvars"].astype(np.float64)) + l["means"].astype(np.float64) ** 2) / l["means"].size)
            gmu = arena[d["mu"][0]:d["mu"][0] + O * I].reshape(O, I)[rows] + l["means"][rows] / (np.float32(B) * vh)
            glv = arena[d["lv"][0]:d["lv"][0] + O * I].reshape(O, I)[rows] + (np.exp(l["lvars"][rows]) / vh - np.float32(1)) / np.float32(2 * B)
            _adam(l["means"][rows], gmu.astype(np.float32), adam[k]["mm"][rows], adam[k]["vm"][rows], step, 1e-4)
            _adam(l["lvars"][rows], glv.astype(np.float32), adam[k]["ml"][rows], adam[k]["vl"][rows], step, 5e-2)
            if mode == "sharded":                              # every rank's new rows to every rank (in place, by rows)
                for name in ("means", "lvars"):
                    tt = torch.from_numpy(l[name])
                    ex.all_gather(tt)
                ex.finish()
    if rank == 0:
        q.put([l["means"].copy() for l in layers] + [l["lvars"].copy() for l in layers] + [l["bias"].copy() for l in layers] + [w3, b3])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [1, 2, 3])
def test_sharded_update_host_logic_over_gloo_matches_the_all_reduce(world):
    """VERDICT r03 item 3 on the CPU: the sharded-update exchange's host logic (partition.sharded_plan, sharded_exchange_grads,
    layer_row_shard) and comm.TorchExchange's reduce_scatter / all_gather with `world` gloo ranks around oracle compute: after two
    training steps the parameters equal the all-reduce mode's -- bitwise for one and two ranks (gloo's ring adds three ranks'
    terms in an order that depends on the message cut: to rounding there)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    got = {}
    for mode in ("allreduce", "sharded"):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q, mode)) for r in range(world)]
        for p in procs:
            p.start()
        got[mode] = q.get(timeout=240)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    for a, b in zip(got["allreduce"], got["sharded"]):
        if world <= 2:
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{int((a != b).sum())} of {a.size} differ"
        else:
            np.testing.assert_allclose(b, a, rtol=0, atol=2e-6 * np.abs(a).max())
    from vbnn_amd import partition
    lay, fin, total, _ = partition.arena_layout([784, 4096, 4096], 10)
    ar, sh = partition.exchange_bytes(lay, fin, 8)
    assert abs(sh / ar - 0.75) < 0.01 and abs(ar - 0.875 * 2 * 4 * total) < 1            # 0.75 x the bytes at 8 ranks, bf16 shadows


def _agree_worker(rank, world, port, q, break_probe_on):
    """One rank of comm.make_exchange's collective decision on the CPU: there is no HIP device here, so building the RCCL
    communicator cannot succeed -- what must hold is that every rank reaches the SAME fallback without waiting for a peer
    that took another path (rank 0 raising before the id broadcast left the others in it)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from vbnn_amd import _lib as L
    from vbnn_amd import comm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["VBNN_EXCHANGE"] = "rccl"                      # take the RCCL attempt although the out-of-band group is gloo
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Ctx:                                                # no device: vbnn_comm_create must fail on every rank
        h = None
        device = torch.device("cpu")
    if rank == break_probe_on:                                # this rank cannot even bind librccl
        lib = L.lib()
        real = lib.vbnn_comm_unique_id

        class Broken:
            def __getattr__(self, name):
                if name == "vbnn_comm_unique_id":
                    return lambda buf: 4
                return getattr(lib, name)
        L._lib = Broken()
        try:
            ex = comm.make_exchange(Ctx(), rank, world)
        finally:
            L._lib = lib
        assert real is lib.vbnn_comm_unique_id
    else:
        ex = comm.make_exchange(Ctx(), rank, world)
    t = torch.tensor([float(rank + 1)])
    ex.allreduce(t); ex.finish()                              # the chosen transport works, on every rank
    q.put((rank, ex.backend, float(t[0])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("break_probe_on", [-1, 1])
def test_exchange_backend_decision_is_collective(break_probe_on):
    """Two gloo ranks call comm.make_exchange with the RCCL path forced. -1: every rank can bind librccl (or none can) and the
    communicator cannot be built (no device): all fall back after the SECOND agreement. 1: rank 1 alone cannot bind it: all
    fall back at the FIRST agreement, nobody is left in the unique-id broadcast. Either way both ranks return the same
    transport within the timeout and an all-reduce through it gives 1 + 2."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, 2, port, q, break_probe_on)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=200) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[0] for g in got] == [0, 1] and all(g[2] == 3.0 for g in got), got
    assert all(g[1].startswith("torch.distributed/gloo") for g in got), got
    assert ("unavailable on some rank" in got[0][1]) == ("unavailable on some rank" in got[1][1])
    assert ("vbnn_comm_create failed" in got[0][1]) == ("vbnn_comm_create failed" in got[1][1])
    if break_probe_on >= 0:
        assert all("unavailable on some rank" in g[1] for g in got), got


def test_partition_host_logic():
    """Row shards tile the global batch; the scales make a plain SUM the global-batch mean; the arena is covered once."""
    from vbnn_amd import partition
    for world in (1, 2, 4, 8):
        rows = [partition.shard_rows(32768, world, r) for r in range(world)]
        assert rows[0][0] == 0 and all(rows[i][0] + rows[i][1] == rows[i + 1][0] for i in range(world - 1))
        assert rows[-1][0] + rows[-1][1] == 32768
        sc = partition.scales(rows[0][1], world)
        assert abs(sc["inv_n"] * 32768 - 1.0) < 1e-12 and abs(sc["kl_scale"] * world - 1.0) < 1e-12
    with pytest.raises(ValueError):
        partition.shard_rows(10, 4, 0)
    lay, fin, total, buckets = partition.arena_layout([784, 4096, 4096], 10)
    assert total == 40026122                                    # the all-reduced gradient count
    covered = sorted(buckets)
    assert covered[0][0] == 0 and covered[-1][1] == total and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    assert buckets[0] == (lay[1]["bucket"][0], total)           # last VB layer + final Linear first
    deep = partition.arena_layout([4096] * 9, 10)
    assert len(deep[3]) == 8 and deep[2] == 8 * (2 * 4096 * 4096 + 4096) + 10 * 4096 + 10
    # early d/dlvars messages: the last layer's d/dlvars leaves first, then [d/dmeans | d/dbias | final Linear]
    lay, fin, total, buckets = partition.arena_layout([784, 4096, 4096], 10, early_lv=[False, True])
    W = 4096 * 4096
    assert buckets[0] == (lay[1]["lv"][0], lay[1]["lv"][0] + W) and buckets[1] == (lay[1]["mu"][0], total)
    assert buckets[2] == lay[0]["bucket"] and len(buckets) == 3
    covered = sorted(buckets)
    assert covered[0][0] == 0 and covered[-1][1] == total and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
