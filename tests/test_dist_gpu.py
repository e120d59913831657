"""The engine's REAL multi-rank path on the MI355X: two processes, one GPU, gloo collectives (RCCL needs a GPU per rank;
the builder's box has one). Everything but the transport is what `bench.py --gpus N` runs: the row partition, z by global
row, the criterion over the global batch, KL / world, the async all-reduces issued inside the step (the final Linear's
gradients riding in the last VB layer's), finish(). The reduced arena must equal a single process on the whole batch."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import _children

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("dtype,hidden,I0,N,tol,xdt,S", [("f32", "50,34", 70, 64, 2e-5, "f32", 1), ("bf16", "512,256", 256, 512, 2e-3, "f32", 1),
                                                          ("bf16", "4096,4096", 784, 1024, 2e-3, "f32", 1),
                                                          ("bf16", "512,256", 256, 512, 6e-3, "bf16", 1),
                                                          # S sequential draws: the buckets are exchanged ONCE, with the last draw
                                                          ("f32", "50,34", 70, 64, 2e-5, "f32", 3), ("bf16", "512,256", 256, 512, 2e-3, "f32", 2)])
def test_two_ranks_on_one_gpu_match_the_single_process_step(tmp_path, dtype, hidden, I0, N, tol, xdt, S):
    import torch
    from vbnn_amd.engine import FusedMLP
    from vbnn_amd.nn import fill_normal
    out = str(tmp_path / "r")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    if os.environ.get("VBNN_EXCHANGE_FOR_WORKER"):
        env["VBNN_EXCHANGE"] = os.environ["VBNN_EXCHANGE_FOR_WORKER"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_gpu_worker.py"), out, dtype, hidden, str(I0), str(N), xdt, str(S)]
    res = _children.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    if os.environ.get("VBNN_EXCHANGE_FOR_WORKER") == "p2p":
        assert res.stdout.count("exchange: vbnn_p2p/ipc") == 2, res.stdout[-1500:]
    got = np.load(out + ".grads.npy")
    loss2 = float(np.load(out + ".loss.npy")[0])
    opt = dict(var_init=1e-3, mu_init=1, B=1e6, S=S, mode="lrt", dtype=dtype, seed=3, input_size=I0,
               hidden=[int(h) for h in hidden.split(",")], n_classes=10, fuse_kl=True)
    eng = FusedMLP(opt)
    x = torch.empty(N, I0, dtype=torch.float32, device="cuda")
    fill_normal(x, 3, 4, 0, 0)
    t = (torch.arange(N, device="cuda", dtype=torch.int64) * 7 % 10).to(torch.int32)
    for _ in range(2):
        eng.resetGradients(); eng.prepare()
        for _ in range(S):
            eng.sample()
            eng.run(x, t)
    loss1, _ = eng.loss_and_accuracy()
    want = eng.grads.cpu().numpy()
    print(f"single process loss {loss1!r}; ranks: {[l for l in res.stdout.splitlines() if 'local loss' in l]}")
    assert abs(loss2 - loss1) <= 1e-5 * abs(loss1), f"summed loss of the two ranks {loss2!r} vs single process {loss1!r}; worker output: {res.stdout[-600:]}"
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    print(f"two ranks vs one process: relative Frobenius {rel:.3e} (bound {tol})")
    if rel > tol:                                   # say WHICH gradient is off: per arena region, relative to that region
        from vbnn_amd.partition import arena_layout
        layers, final, _, _ = arena_layout([I0] + opt["hidden"], 10)
        regions = [(f"layer {k} {name}", d[name]) for k, d in enumerate(layers) for name in ("lv", "mu", "bias")]
        regions += [("final weight", final["weight"]), ("final bias", final["bias"])]
        detail = "; ".join(f"{n}: {np.linalg.norm(got[o:o + c] - want[o:o + c]) / max(np.linalg.norm(want[o:o + c]), 1e-30):.2e}"
                           f" (max |d| {np.abs(got[o:o + c] - want[o:o + c]).max():.2e} at {int(np.abs(got[o:o + c] - want[o:o + c]).argmax())})"
                           for n, (o, c) in regions)
        raise AssertionError(f"relative Frobenius {rel:.3e} > {tol}: {detail}; worker output: {res.stdout[-600:]}")


def test_bench_launches_its_own_ranks_from_a_bare_shell():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts torch.distributed.run as a child before
    touching HIP, the ranks rendezvous (gloo here: two ranks share the box's one GPU), rank 0's JSON line comes back
    through the parent and proves that two processes took part."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VBNN_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "small", "--steps", "3", "--warmup", "1",
           "--repeats", "2", "--no-cpu-baseline"]
    res = _children.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2 * 256 and out["value"] > 0
    seen = out["comm"]["ranks_seen"]
    assert sorted(r[0] for r in seen) == [0, 1] and len({r[3] for r in seen}) == 2          # two ranks, two processes
    assert len(out["comm"]["allreduce"]) == 2 and all(a["ms"] > 0 for a in out["comm"]["allreduce"])
    assert len(out["config"]["repeats_wall_ms"]) == 2
    # the ONE line a multi-GPU run has to answer everything in: the step without its collective calls and
    # the exposed exchange time, every bucket through the OTHER backend as well (here: the step on torch.distributed / gloo,
    # vbnn_p2p timed beside it over IPC-mapped arenas), no p2p barrier given up, the overlapped GEMMs' own times, and the
    # training step
    # the exchange checked itself before anything was timed (r05): one minibatch and draw without and with the collective calls
    val = out["comm"]["validation"]
    assert val["ok"] and val["identical_across_ranks"] and val["finite"] and val["max_abs_err_over_max_abs_vs_process_group_sum"] <= 1e-5, val
    noex = out["comm"]["step_without_exchange"]
    assert noex["ms_per_step"] > 0 and abs(noex["exposed_exchange_ms"] - (out["ms_per_step"] - noex["ms_per_step"])) < 1e-3
    # (--exchange auto: both backends were timed on a 64 MB message at start-up and the faster carries the step; the other one's
    # buckets are priced beside it -- here gloo against vbnn_p2p over IPC-mapped arenas, RCCL being impossible with ranks that share a GPU)
    probe = out["comm"]["probe"]
    assert probe["choice"] in ("library", "p2p") and probe["p2p"]["backend"] == "vbnn_p2p/ipc" and probe["p2p"]["ms"] > 0
    assert probe["library"]["backend"].startswith("torch.distributed/gloo") and probe["library"]["ms"] > 0
    assert out["comm"]["backend"] == ("vbnn_p2p/ipc" if probe["choice"] == "p2p" else probe["library"]["backend"])
    other = out["comm"]["other_backend"]
    if probe["choice"] == "library":
        assert other.get("backend") == "vbnn_p2p/ipc" and "error" not in other, other
        assert len(other["buckets"]) == 2 and all(b["ms"] > 0 and b["bus_GBps"] > 0 for b in other["buckets"])
        assert other["p2p_barrier_gave_up_epoch"] == 0
    else:
        assert "error" in other                              # RCCL is not timed with ranks sharing a device, and says so
        assert out["comm"]["p2p_barrier_gave_up_epoch"] == 0
    assert "distinct device(s)" in out["comm"]["scaling_note"] and out["comm"]["distinct_devices"] == 1
    assert out["roofline"]["timed_region_kernels_ms"]
    assert out["train_step"]["ms_per_train_step"] > 0 and out["train_step"]["samples_per_s"] > 0


def test_bench_rehearsal_with_the_p2p_exchange_reports_its_status():
    """The same line with VBNN_EXCHANGE=p2p: the step's buckets go through vbnn_p2p_allreduce, the line carries the exchange's
    status word (0: no barrier gave up) -- a non-zero status would have been rc 3 and no value."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VBNN_DIST_BACKEND="gloo", VBNN_EXCHANGE="p2p")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "small", "--steps", "3", "--warmup", "1",
           "--repeats", "2", "--no-cpu-baseline"]
    res = _children.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert out["comm"]["backend"] == "vbnn_p2p/ipc" and out["comm"]["p2p_barrier_gave_up_epoch"] == 0
    assert out["comm"]["validation"]["ok"] and out["comm"]["validation"]["backend"] == "vbnn_p2p/ipc", out["comm"]["validation"]
    assert "error" in out["comm"]["other_backend"]                  # RCCL is not timed with ranks sharing a device, and says so
    assert out["comm"]["step_without_exchange"]["ms_per_step"] > 0
    # a rank count the launcher did not provide is an error, not a silent single-rank run
    bad = _children.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "small"],
                         env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0


@pytest.mark.parametrize("world", [2, 3, 4])
def test_p2p_exchange_sums_peer_mapped_arenas_bitwise(world):
    """vbnn_p2p_* (csrc/p2p.hip): the exchange without a collective library -- IPC-mapped arenas, a reduce-scatter and an
    all-gather kernel, flag-page barriers -- with `world` processes sharing the box's one GPU: every rank ends with, bitwise,
    the rank-ordered fp32 sum of every region (aligned and odd offsets, lengths that do not divide by the world), three
    rounds (default, tiny and large grids; a region exchanged twice without a finish), no barrier ever giving up. (Four ranks is
    what the box's process limit allows beside the test runner; the reduce-scatter's second batch of peer loads -- worlds above
    four -- is held bitwise by the one-process stand-in test, test_parity_gpu.py::test_p2p_stand_in_...)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_p2p_worker.py")]
    res = _children.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    print("\n".join(l for l in res.stdout.splitlines() if "rank" in l))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count("bitwise equal") == 3 * world


def test_p2p_exchange_with_a_late_rank_fails_loudly_and_leaves_the_arenas_alone():
    """A rank later than the barrier's bound (set to 0.5 s here; 20 s by default): the early rank's barrier gives up, BOTH ranks'
    data kernels become no-ops (the failure reaches the late rank through its flag page), check() raises on both before
    anything reads the arena, and after clear_status on both the exchange sums bitwise again (tests/_p2p_worker.py)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_p2p_worker.py"), "delayed"]
    res = _children.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    print("\n".join(l for l in res.stdout.splitlines() if "rank" in l))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count("check raised True, arena untouched True") == 2, res.stdout[-1500:]
    assert res.stdout.count("after clear_status the sum is bitwise equal") == 2, res.stdout[-1500:]


def test_two_ranks_through_the_p2p_exchange_match_the_single_process_step(tmp_path):
    """The engine's two-rank step with opt.exchange = "p2p" (VBNN_EXCHANGE=p2p): the gradient arena is the exchange's own
    IPC-exported allocation, the buckets are all-reduced by vbnn_p2p_allreduce inside the step."""
    os.environ["VBNN_EXCHANGE_FOR_WORKER"] = "p2p"
    try:
        test_two_ranks_on_one_gpu_match_the_single_process_step(tmp_path, "bf16", "512,256", 256, 512, 2e-3, "f32", 1)
    finally:
        os.environ.pop("VBNN_EXCHANGE_FOR_WORKER", None)



@pytest.mark.parametrize("world,exchange,dtype,hidden,I0,n_loc", [
    (1, "torch", "bf16", "48,36", 70, 64),
    (1, "rccl", "bf16", "48,36", 70, 64),          # vbnn_comm_reduce_scatter / _all_gather: RCCL itself, a world of one (it wants a device per rank)
    (2, "torch", "bf16", "48,36", 70, 64), (3, "torch", "bf16", "48,36", 70, 64),
    (2, "p2p", "bf16", "48,36", 70, 64), (3, "p2p", "bf16", "48,36", 70, 64),
    # the wide configuration itself: K-major operands (no transposed shadows), the two-launch accGradParameters with its early
    # d/dlvars message, the head's logits from the forward tiles
    (2, "p2p", "bf16", "4096,4096", 784, 4096)])
def test_sharded_update_exchange_leaves_bitwise_the_all_reduce_parameters(tmp_path, world, exchange, dtype, hidden, I0, n_loc):
    """opt.exchange_mode = "sharded" (VERDICT r03 item 3): reduce-scatter of the likelihood gradients by layer rows, vbnn_update on
    this rank's rows alone (Adam state sharded), all-gather of the bf16 operand shadows + the slices' prior statistics -- 0.75 x
    the bytes of the all-reduce, 1 / world of the update sweep, the sums still fp32. After TWO training steps (the second Adam
    step depends on the first's moments and on the gathered statistics) the parameters, biases, final Linear and operand shadows
    are BITWISE those of the all-reduce mode with the same ranks over the same exchange (the reduce-scatter forms the very sums
    the all-reduce forms, the slice update is the whole update restricted to rows); the statistics agree to 1e-14 (a sum over row
    slices in rank order against one sum over the layer)."""
    res = {}
    for mode in ("allreduce", "sharded"):
        out = str(tmp_path / f"{mode}.npz")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", VBNN_EXCHANGE=exchange)
        env["VBNN_FORCE_DIST"] = "1"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_sharded_worker.py"), out, mode, dtype, hidden, str(I0),
               str(n_loc), "2"]
        r = _children.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert r.stdout.count("equal across ranks: True") == world, r.stdout[-1500:]
        if exchange == "p2p" and world > 1:
            assert r.stdout.count("exchange vbnn_p2p/ipc") == world, r.stdout[-1500:]
        if exchange == "rccl":
            assert r.stdout.count("exchange vbnn_comm/rccl") == world, r.stdout[-1500:]
        res[mode] = dict(np.load(out))
    a, b = res["allreduce"], res["sharded"]
    # (gloo's ring all-reduce adds three or more ranks' terms in an order that depends on where an element lies in the MESSAGE, and
    # the two modes cut the arena into different messages: bitwise there only for one or two ranks -- a + b has one order. The
    # library's own exchange sums in rank order whatever the message.)
    bitwise = exchange == "p2p" or world <= 2
    if bitwise:
        assert np.array_equal(a["losses"], b["losses"]), (a["losses"], b["losses"])
    else:
        np.testing.assert_allclose(b["losses"], a["losses"], rtol=1e-6)
    for k in a:
        if k.startswith("stats"):
            np.testing.assert_allclose(b[k][:3], a[k][:3], rtol=1e-14 if bitwise else 1e-6, atol=0)
        elif k != "losses" and bitwise:
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), f"{k}: {int((a[k] != b[k]).sum())} of {a[k].size} elements differ (max {np.abs(a[k] - b[k]).max():.3e})"
        elif k != "losses":
            np.testing.assert_allclose(b[k], a[k], rtol=0, atol=2e-6 * np.abs(a[k]).max(), err_msg=k)


def test_bench_line_of_a_multi_rank_run_prices_the_sharded_update_exchange():
    """`bench.py --gpus 2` on the headline configuration (two gloo ranks sharing the GPU, the step's buckets over vbnn_p2p): beside
    the all-reduce step and its train_step the line carries train_step_sharded -- the same training step with the reduce-scatter /
    sliced update / shadow all-gather exchange -- with the bytes each mode sends per rank and step."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VBNN_DIST_BACKEND="gloo", VBNN_EXCHANGE="p2p")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--repeats", "1",
           "--no-cpu-baseline", "--no-deep-config"]
    res = _children.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8192 and out["comm"]["backend"] == "vbnn_p2p/ipc"
    sh = out["train_step_sharded"]
    assert "error" not in sh, sh
    assert sh["backend"] == "vbnn_p2p/ipc" and sh["ms_per_train_step"] > 0 and sh["update_rows_per_rank"] == [2048, 2048]
    assert abs(sh["bytes_sent_per_rank_per_step"]["ratio"] - 0.75) < 0.01
    assert out["train_step"]["ms_per_train_step"] > 0 and out["comm"]["p2p_barrier_gave_up_epoch"] == 0
