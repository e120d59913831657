"""world_size-2 gloo test (CPU) of the data-parallel recipe the engine uses on N GPUs:

  - rank r takes rows [r N/G, (r+1) N/G) of the global minibatch, with z addressed by GLOBAL row;
  - the criterion divides by the GLOBAL batch (inv_n = 1 / (N_local * world));
  - the fused KL gradient carries kl_scale = 1 / world;
  - a SUM all-reduce over [grad_mu | grad_lv | gradBias] per layer + the final Linear's gradients.

The per-rank compute here is the oracle (tests may use it); what is under test is the partition /
scaling / reduction logic of vbnn_amd/engine.py: the reduced result must equal the single-process result
on the whole batch.
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
    """What one engine rank computes: local rows, global-row noise, 1/(N_global) criterion, KL / world."""
    from oracle import vbnn_oracle as vo
    from oracle.ref_numpy import emulate_lrt_step
    N = x.shape[0]
    n_loc = N // world
    r0 = rank * n_loc
    xs, ts = x[r0:r0 + n_loc], t[r0:r0 + n_loc]
    zetas = [vo.fill_normal(n_loc, O, 3, vo.STREAM_ZETA, k, 1, r0).astype(np.float64) for k, O in enumerate(outs)]
    loss, res, gw3, gb3 = emulate_lrt_step(layers, w3, b3, xs, ts, zetas, S=1.0, B=B, inv_n=1.0 / N)
    flat = []
    for lay, r in zip(layers, res):
        vh = float(np.sum(np.exp(lay["lvars"].astype(np.float64)) + lay["means"].astype(np.float64) ** 2) / lay["means"].size)
        var = np.exp(lay["lvars"].astype(np.float64))
        kl_mu = lay["means"] / (B * vh)
        kl_lv = (var / vh - 1.0) / (2 * B)
        # engine epilogue: likelihood / S + kl_scale * KL with kl_scale = 1 / world
        flat += [(r["grad_mu"] - kl_mu + kl_mu / world).ravel(), (r["grad_lv"] - kl_lv + kl_lv / world).ravel(),
                 r["gradBias"].ravel()]
    flat += [gw3.ravel(), gb3.ravel()]
    return loss, np.concatenate(flat)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    opt, layers, w3, b3, x, t, outs = _problem()
    loss, flat = _rank_step(rank, world, layers, w3, b3, x, t, outs, opt["B"])
    buf = torch.from_numpy(flat.copy())
    lbuf = torch.tensor([loss], dtype=torch.float64)
    works = [dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True),       # as FusedMLP._reduce
             dist.all_reduce(lbuf, op=dist.ReduceOp.SUM, async_op=True)]
    for w in works:
        w.wait()
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
    loss1, flat1 = _rank_step(0, 1, layers, w3, b3, x, t, outs, opt["B"])
    assert abs(loss1 - loss2) <= 1e-6 * abs(loss1)          # the criterion sums fp32 log-probabilities
    # not bitwise: the per-rank dgemm sees a different row count, its last-bit differences move a few fp32
    # roundings of the activations (1e-7 relative), exactly as on the GPUs
    np.testing.assert_allclose(flat2, flat1, rtol=5e-6, atol=1e-8)
