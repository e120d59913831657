#!/usr/bin/env python3
"""bench.py -- VBLinear fwd+bwd samples/sec on MI355X (BASELINE.json's metric).

A step = one pass of the hot path over one synthetic minibatch, exactly main.lua:28-37 with S = 1:
    resetGradients -> sample -> run (input packing, forward of every VB layer, final Linear, LogSoftMax + ClassNLL,
    backward with accGradParameters incl. the KL-gradient epilogue) -> all-reduce of the gradient arena when N > 1.
The optimiser update is excluded -- and with it the parameter sweep: the reference runs
`compute_prior` inside `VBLinear:update` (VBLinear.lua:130), and here the update kernel (vbnn_update) is what
leaves the bf16 mu / sigma^2 operand shadows and the prior statistics for the next minibatch; `prepare()` runs once,
before the first step (`--prepare-each-step` puts it back into the step: round 1's protocol). `--with-update`
times the whole training step (step + update) beside it. Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config wide|small|deep]
N > 1: launched by the driver through torch.distributed.run (one rank per GPU, RCCL) -- or, from a bare shell, by
bench.py itself: a parent that has not imported torch or touched HIP starts the same torch.distributed.run line as a
CHILD process, relays its output and exits with its code. The timed region is `--repeats` (7) blocks of exactly K
steps, each bracketed by barrier + synchronize (wall clock, MAX over ranks) and by HIP events; `ms_per_step` is the
median block. Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[2]/[3]: the config the metric is quoted on
    "wide": dict(input_size=784, hidden=[4096, 4096], n_classes=10, batch=4096, dtype="bf16",
                 name="784-4096-4096-10 VBLinear MLP, batch 4096 per GPU, LRT, S=1"),
    # BASELINE.json configs[1]: the fp32 numerics configuration (launch-bound)
    "small": dict(input_size=784, hidden=[400, 400], n_classes=10, batch=256, dtype="f32",
                  name="784-400-400-10 VBLinear MLP, batch 256 per GPU, LRT, S=1"),
    # BASELINE.json configs[4]: 8 x 4096-wide VBLinear stack on a synthetic 4096-dimensional regression (final Linear
    # 4096 -> 4096 + nn.MSECriterion, target x R / sqrt(4096): BASELINE.md section 2 config 5): deep-stack KL / bandwidth stress
    "deep": dict(input_size=4096, hidden=[4096] * 8, n_classes=4096, batch=4096, dtype="bf16", criterion="mse",
                 name="4096-(4096 x 8)-4096 VBLinear stack, MSE regression, batch 4096 per GPU, LRT, S=1"),
    # the same stack under the reference's own criterion (10 classes, LogSoftMax + ClassNLL): round 1's "deep" row
    "deep-nll": dict(input_size=4096, hidden=[4096] * 8, n_classes=10, batch=4096, dtype="bf16",
                     name="4096-(4096 x 8)-10 VBLinear stack, batch 4096 per GPU, LRT, S=1"),
}
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}     # MI355X_MICROARCH.md: dense MFMA peaks
LIVE_NAMES = {"forward": "forward(dual GEMM + LRT epilogue)",
              "accGradParameters": "accGradParameters(dual GEMM + KL epilogue)",
              "updateGradInput": "updateGradInput(dual GEMM + ReLU/dv epilogue)"}


def algorithmic_flops_per_sample(sizes, n_classes):
    """per VB layer 4IO fwd + 4IO param grads + 4IO input grads (none for layer 1);
    final Linear 6IO. Elementwise / RNG / KL flops are not counted."""
    f = 0
    for li in range(len(sizes) - 1):
        io = sizes[li] * sizes[li + 1]
        f += 8 * io if li == 0 else 12 * io
    f += 6 * sizes[-1] * n_classes
    return f


def time_kernel(fn, reps, torch):
    """Average duration (ms) of `fn`'s launches, HIP events on the stream the kernels run on."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def dominant_kernel_roofline(eng, N, torch):
    """Time the three dual-GEMM families of the widest VB layer in isolation and report the slowest
    one against the MFMA roofline. Algorithmic flops per launch: 2 GEMMs x 2*N*I*O."""
    import ctypes as C
    from vbnn_amd import _lib as L
    lib, ctx, code = L.lib(), eng.ctx.h, eng.code
    li = max(range(len(eng.vb)), key=lambda k: eng.vb[k].I * eng.vb[k].O)
    v = eng.vb[li]
    flops = 2 * 2.0 * N * v.I * v.O

    def fwd():
        a = eng._fwd_args(li, N, 0)
        L.check(lib.vbnn_forward(ctx, code, C.byref(a)))

    def dw():
        d = eng._dw_args(li, N, 0)
        L.check(lib.vbnn_acc_grad_parameters(ctx, code, C.byref(d)))

    kernels = {"forward(dual GEMM + LRT epilogue)": fwd, "accGradParameters(dual GEMM + KL epilogue)": dw}
    if li > 0:
        def dx():
            a = eng._dx_args(li, N)
            L.check(lib.vbnn_grad_input(ctx, code, C.byref(a)))
        kernels["updateGradInput(dual GEMM + ReLU/dv epilogue)"] = dx
    res = {k: time_kernel(fn, 10, torch) for k, fn in kernels.items()}
    name = max(res, key=res.get)
    return name, res[name], flops, {k: round(v_, 4) for k, v_ in res.items()}


def measured_traffic(kernel_name, cfg_name):
    """HBM bytes per launch of the dominant kernel from the PMC passes (FETCH_SIZE / WRITE_SIZE are collected
    in their own rocprofv3 runs, never inside this timed process): profiles/*_traffic.json, newest round."""
    if cfg_name != "wide":
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None
    try:
        return json.load(open(files[-1])).get(kernel_name, {}).get("hbm_bytes")
    except Exception:
        return None


def box_calibration(L, ctx_handle):
    """Two fixed probes of THIS box, run in this process (vbnn_box_calibrate, csrc/calib.hip; ~40 ms): a register-only MFMA loop
    on every SIMD -> the clock the chip holds under a pure matrix load and the MFMA rate it reaches there; a 512 MiB -> 512 MiB
    copy -> the HBM rate. The boxes of a pool differ by 5-7 % on one binary: with these a round-to-round delta of the line
    can be read as code or as box."""
    import ctypes as C
    info = L.BoxInfo()
    L.check(L.lib().vbnn_box_calibrate(ctx_handle, C.byref(info)))
    return {"mfma_clock_ghz": round(info.mfma_clock_ghz, 4), "mfma_tflops": round(info.mfma_tflops, 1),
            "mfma_frac_of_dense_peak": round(info.mfma_tflops / PEAK_TFLOPS["bf16"], 4), "mfma_probe_ms": round(info.mfma_ms, 4),
            "hbm_TBps": round(info.hbm_TBps, 3), "hbm_probe_ms": round(info.hbm_ms, 4), "hbm_probe_bytes": int(info.hbm_bytes), "cus": int(info.cus)}


def cpu_baseline(cfg, budget_s=25.0):
    """The reference's op sequence on the host cores (oracle/ref_numpy.py), bounded sample: once with every core the
    BLAS pool has (`value`, `cores`) and once at the reference's own default of 8 threads (config.lua:5 `threads = 8`,
    main.lua:142 torch.setnumthreads) -- `at_reference_threads`."""
    import numpy as np
    from oracle.ref_numpy import ReferenceCpuMLP, blas_threads
    N = cfg["batch"]
    net = ReferenceCpuMLP(cfg["input_size"], cfg["hidden"], cfg["n_classes"])
    rs = np.random.RandomState(3)
    x = rs.standard_normal((N, cfg["input_size"])).astype(np.float32)
    t = rs.randint(0, cfg["n_classes"], N)

    def sample(budget):
        net.step(x, t)                               # warm-up (BLAS thread pool, page faults)
        steps, t0 = 0, time.perf_counter()
        while True:
            net.step(x, t)
            steps += 1
            el = time.perf_counter() - t0
            if el > budget or steps >= 50:
                break
        return steps, el

    steps, el = sample(budget_s * 0.4)
    out = dict(value=round(N * steps / el, 1), unit="samples/s", cores=blas_threads(), kind="port",
               sample=f"{steps} full steps (batch {N}) of the same MLP, weight-noise op sequence of VBLinear.lua "
                      f"on NumPy/OpenBLAS sgemm + MT19937 Gaussian fill, {el:.1f} s; Torch7 itself cannot run here")
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=8, user_api="blas"):
            used = blas_threads()
            s8, e8 = sample(budget_s * 0.3)
        out["at_reference_threads"] = dict(value=round(N * s8 / e8, 1), unit="samples/s", cores=used,
                                           sample=f"{s8} full steps in {e8:.1f} s with the BLAS pool limited to the "
                                                  "reference's default, opt.threads = 8 (config.lua:5, main.lua:142)")
    except Exception as e:                            # threadpoolctl missing: say so, keep the all-cores figure
        out["at_reference_threads"] = dict(value=None, note=f"not measured: {e}")
    # `value` is the FASTER of the two (the baseline is not to be strawmanned -- on the 64-core boxes of
    # this pool OpenBLAS runs this op sequence faster on 8 threads than on all cores); both samples stay in the line
    ref = out["at_reference_threads"]
    out["all_cores"] = dict(value=out["value"], cores=out["cores"], sample=out["sample"])
    if ref.get("value") and ref["value"] > out["value"]:
        out["value"], out["cores"], out["sample"] = ref["value"], ref["cores"], ref["sample"] + "; " + out["sample"].split(", ", 1)[1].rsplit(",", 1)[0]
    return out


def reporting_config_row(torch, L, FusedMLP, fill_normal, with_cpu):
    """north_star's own reporting configuration beside the headline, in the same run: the synthetic 784-400-400-10 MLP at
    batch 256, fp32, LRT, S = 1 (BASELINE.json configs[1]) -- forward + backward samples/s, the fraction of the fp32 MFMA
    peak over the step, the CPU baseline on the same host cores -- and the reference's shipped operating point, batch 1 with
    S = 30 draws (config.lua:11,32), the draws stacked as rows. Launch-bound (LAB_NOTES.md section 3): reported, not the metric."""
    cfg = CONFIGS["small"]
    out = {"workload": cfg["name"], "dtype": "f32", "launch": "one stream launch per kernel (5 per step: forward x 2, the one-launch head, the layer-2 backward pair, accGradParameters of layer 1)"}
    for key, N, S, stack in (("batch256", cfg["batch"], 1, False), ("batch1_S30_stacked", 1, 30, True)):
        opt = dict(var_init=1e-3, B=1e6, S=S, mode="lrt", dtype="f32", seed=3, input_size=cfg["input_size"], hidden=cfg["hidden"],
                   n_classes=cfg["n_classes"], fuse_kl=True)
        eng = FusedMLP(opt)
        x = torch.empty(N, cfg["input_size"], dtype=torch.float32, device="cuda")
        fill_normal(x, 3, L.STREAM_DATA, 0, 0)
        t = eng.synthetic_targets(x, 0)

        def step():
            eng.resetGradients()
            if stack:
                eng.run_draws(x, t, S)
            else:
                eng.sample()
                eng.run(x, t)
            eng.finish()
        eng.prepare()
        for _ in range(10):
            step()
        blocks = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                step()
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / 50 * 1e3)
        ms = sorted(blocks)[2]
        fps = algorithmic_flops_per_sample([cfg["input_size"]] + cfg["hidden"], cfg["n_classes"]) * S
        out[key] = {"ms_per_step": round(ms, 4), "samples_per_s": round(N * S / (ms * 1e-3), 1), "blocks_ms": [round(b, 4) for b in blocks],
                    "step_frac_of_fp32_mfma_peak": round(fps * N / (ms * 1e-3) / 1e12 / PEAK_TFLOPS["f32"], 4)}
        del eng
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline(dict(cfg), budget_s=4.0)
    return out


def side_config_row(name, torch, dist, L, FusedMLP, fill_normal, steps, warmup, world, rank, use_dist, blocks=3, exchange=None):
    """Another BASELINE.json configuration measured in the SAME run, by the same protocol (warm-up, `blocks` blocks of `steps`
    steps between barrier + synchronize, MAX over ranks, the median block): configs[4], the 8 x 4096 VBLinear stack with its
    regression head, so that the driver's line carries it. Its own engine, its own exchange when N > 1."""
    cfg = CONFIGS[name]
    N = cfg["batch"]
    opt = dict(var_init=1e-3, B=1e6, S=1, mode="lrt", dtype=cfg["dtype"], seed=3, input_size=cfg["input_size"], hidden=cfg["hidden"],
               n_classes=cfg["n_classes"], fuse_kl=True, criterion=cfg.get("criterion", "nll"))
    if exchange:
        opt["exchange"] = exchange
    eng = FusedMLP(opt, world_size=world, rank=rank, force_reduce=use_dist)
    x = torch.empty(N, cfg["input_size"], dtype=torch.float32, device="cuda")
    fill_normal(x, 3, L.STREAM_DATA, 0, 0, row0=rank * N)
    t = eng.synthetic_targets(x, rank * N)

    def step():
        eng.resetGradients(); eng.sample(); eng.run(x, t); eng.finish()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
    eng.prepare()
    for _ in range(warmup):
        step()
    wall = []
    for _ in range(blocks):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        el = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        wall.append(el / steps * 1e3)
    eng.check_exchange()
    loss, _ = eng.loss_and_accuracy()
    ms = sorted(wall)[len(wall) // 2]
    fps = algorithmic_flops_per_sample([cfg["input_size"]] + cfg["hidden"], cfg["n_classes"])
    peak = PEAK_TFLOPS[cfg["dtype"]]
    out = {"workload": cfg["name"], "dtype": cfg["dtype"], "n_gpus": world, "global_batch": N * world, "steps": steps, "warmup": warmup,
           "ms_per_step": round(ms, 4), "samples_per_s": round(N * world / (ms * 1e-3), 1), "blocks_ms": [round(b, 4) for b in wall],
           "flop_per_sample": fps, "step_tflops_per_gpu": round(fps * N / (ms * 1e-3) / 1e12, 2),
           "step_frac_of_mfma_peak": round(fps * N / (ms * 1e-3) / 1e12 / peak, 4), "loss": round(loss, 5),
           "gradient_bytes_exchanged": int(eng.grads.numel() * 4) if use_dist else 0}
    del eng
    torch.cuda.empty_cache()
    return out


def sharded_train_row(cfg, torch, dist, L, FusedMLP, fill_normal, steps, warmup, world, rank, blocks=3, exchange=None):
    """The TRAINING step with the sharded-update exchange (opt.exchange_mode = "sharded": reduce-scatter of the gradients by layer
    rows, vbnn_update on this rank's rows, all-gather of the bf16 operand shadows + statistics -- 0.75 x the all-reduce's bytes,
    1 / world of the update sweep, fp32 sums) beside the all-reduce line, same protocol. An OPTION beside north_star's all-reduce:
    its step cannot leave the update out (the gather follows it), so it is comparable with `train_step`, not with `value`."""
    from vbnn_amd import partition
    N = cfg["batch"]
    opt = dict(var_init=1e-3, B=1e6, S=1, mode="lrt", dtype=cfg["dtype"], seed=3, input_size=cfg["input_size"], hidden=cfg["hidden"],
               n_classes=cfg["n_classes"], fuse_kl=True, criterion=cfg.get("criterion", "nll"), exchange_mode="sharded",
               state=dict(learningRate=1e-3), meanState=dict(learningRate=1e-4), varState=dict(learningRate=5e-2))
    if exchange:
        opt["exchange"] = exchange
    eng = FusedMLP(opt, world_size=world, rank=rank, force_reduce=True)
    x = torch.empty(N, cfg["input_size"], dtype=torch.float32, device="cuda")
    fill_normal(x, 3, L.STREAM_DATA, 0, 0, row0=rank * N)
    t = eng.synthetic_targets(x, rank * N)

    def train_step():
        eng.resetGradients(); eng.sample(); eng.run(x, t); eng.update()

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
    eng.prepare()
    for _ in range(max(3, warmup)):
        train_step()
    wall = []
    for _ in range(blocks):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            train_step()
        barrier()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall.append(float(tt.item()) / steps * 1e3)
    eng.check_exchange()
    ms = sorted(wall)[len(wall) // 2]
    ar, sh = partition.exchange_bytes(eng._lay, {"weight": (0, eng.gradWeight3.numel()), "bias": (0, eng.gradBias3.numel())}, world)
    out = {"exchange_mode": "sharded", "backend": eng.comm_backend(), "ms_per_train_step": round(ms, 4),
           "samples_per_s": round(N * world / (ms * 1e-3), 1), "repeats_wall_ms": [round(v, 4) for v in wall],
           "bytes_sent_per_rank_per_step": {"all_reduce": int(ar), "sharded": int(sh), "ratio": round(sh / ar, 4) if ar else None},
           "update_rows_per_rank": [p["rows"] for p in eng._plan],
           "note": "reduce-scatter (fp32 sums, by layer rows) -> vbnn_update on this rank's rows (Adam state + fp32 master rows sharded, the exact "
                   "KL gradient added there) -> all-gather of the bf16 operand shadows + the slices' prior statistics; compare with train_step"}
    del eng
    torch.cuda.empty_cache()
    return out


def probe_exchange_backends(torch, dist, rank, world, local_rank, n_floats=1 << 24, reps=5):
    """Which exchange moves a 64 MB all-reduce faster on THIS node -- the collective library (vbnn_comm over RCCL; torch.distributed in a
    gloo rehearsal) or the direct reduce-scatter + all-gather over peer-mapped arenas (vbnn_p2p)? SURVEY.md section 5's plan ("measure
    RCCL's achieved bus bandwidth; if RCCL serialises onto one ring, fall back to the peer-to-peer kernels") as a measurement at the
    start of every multi-rank run: both are timed back to back on the idle GPUs, MAX over ranks, and the step uses the faster. Every
    step of it is collective in failure as well (comm.make_exchange / comm.P2PExchange agree over the process group)."""
    from vbnn_amd import comm
    from vbnn_amd.nn import Context
    ctx = Context.get(torch.device("cuda", local_rank))
    out = {"bytes": n_floats * 4}

    def timed(ex, buf):
        for _ in range(2):
            ex.allreduce(buf); ex.finish()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            ex.allreduce(buf)
        ex.finish()
        torch.cuda.synchronize()
        tt = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())
    # crash guard for the one path that has never seen a second device: rank 0 leaves a marker while the p2p probe runs; a run that
    # finds the marker of an earlier run on this checkout (that run died inside its probe: a fault on a peer mapping kills the process,
    # it cannot be caught) does not probe p2p again and says so
    marker = os.path.join(ROOT, "gpurun_out", ".p2p_probe_inflight")
    skip = torch.tensor([1 if os.path.exists(marker) else 0], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(skip, op=dist.ReduceOp.MAX)
    for name in ("library", "p2p"):
        ex = None
        if name == "p2p" and int(skip.item()):
            out[name] = {"skipped": "a previous run on this checkout did not come back from its p2p probe (marker gpurun_out/.p2p_probe_inflight)"}
            continue
        if name == "p2p" and rank == 0:
            try:
                os.makedirs(os.path.dirname(marker), exist_ok=True)
                with open(marker, "w") as f:
                    f.write(f"world {world}\n")
                    f.flush()
                    os.fsync(f.fileno())
            except OSError:
                pass
        try:
            if name == "library":
                ex = comm.make_exchange(ctx, rank, world, None)
                buf = torch.zeros(n_floats, dtype=torch.float32, device="cuda")
            else:
                ex = comm.P2PExchange(ctx, rank, world, n_floats, None)
                ex.set_timeout(5.0)                 # a probe that cannot see its peers gives up in seconds, not minutes
                buf = ex.arena
            ms = timed(ex, buf)
            bad = int(ex.gave_up()) if hasattr(ex, "gave_up") else 0
            out[name] = {"backend": ex.backend, "ms": round(ms, 4), "bus_GBps": round(n_floats * 4 / (ms * 1e-3) / 1e9 * 2 * (world - 1) / world, 1)}
            if bad:
                out[name]["barrier_gave_up_epoch"] = bad
                out[name]["ms"] = None
        except Exception as e:                  # noqa: BLE001 -- (raised on EVERY rank: see the docstring)
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            try:
                if ex is not None and hasattr(ex, "close"):
                    ex.close()
            except Exception:                   # noqa: BLE001
                pass
            del ex
            torch.cuda.empty_cache()
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0 and os.path.exists(marker) and not int(skip.item()):
        try:
            os.remove(marker)
        except OSError:
            pass
    lib_ms, p2p_ms = out["library"].get("ms"), out["p2p"].get("ms")
    out["choice"] = "p2p" if (p2p_ms is not None and (lib_ms is None or p2p_ms < lib_ms)) else "library"
    return out


def validate_exchange(eng, x, t, torch, dist, world):
    """Before anything is timed on N > 1 ranks: does the exchange the step will use SUM? The same minibatch and the same draw twice --
    once with the collective calls left out (this rank's own gradients, saved), once with them -- then (1) every rank must hold the
    same arena (two 64-bit checksums of its bits, gathered over the process group: the direct exchange sums in rank order, so its
    arenas are bitwise equal; RCCL's are equal on every rank as well) and (2) the arena must equal the process group's own all-reduce
    of the saved gradients to fp32 summation-order rounding. The direct exchange has never run across two DEVICES (its visibility
    argument -- data crosses devices at kernel boundaries -- is an argument): the first multi-GPU run checks itself, and a
    throughput measured on sums that are not sums is not reported. Collective: every rank calls it."""
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    d0 = eng.draw
    eng.skip_exchange = True
    eng.resetGradients(); eng.sample(); eng.run(x, t); eng.finish()
    torch.cuda.synchronize()
    local = eng.grads.clone()
    eng.skip_exchange = False
    eng.draw = d0                                           # the same draw again: the same noise, the same local gradients
    eng.resetGradients(); eng.sample(); eng.run(x, t); eng.finish()
    torch.cuda.synchronize()
    gave_up = None
    try:
        eng.check_exchange()
    except RuntimeError as e:                               # (a p2p barrier that gave up: the verdict below is collective all the same)
        gave_up = str(e)[:300]
    bits = eng.grads.view(torch.int32).to(torch.int64)
    mine = torch.stack([bits.sum(), (bits * (torch.arange(bits.numel(), device=bits.device, dtype=torch.int64) % 8191 + 1)).sum()]).to(dev)
    seen = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(seen, mine)
    identical = all(bool(torch.equal(sv, seen[0])) for sv in seen)
    ref = local                                             # (in place: the saved copy becomes the library's sum)
    dist.all_reduce(ref, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    scale = float(ref.abs().max().item()) or 1.0
    err = float((eng.grads - ref).abs().max().item()) / scale
    finite = bool(torch.isfinite(eng.grads).all().item())
    tol = 1e-5 if eng.exchange_dtype == "f32" else 2e-2
    flag = torch.tensor([1 if (identical and finite and err <= tol and gave_up is None) else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)             # one verdict for every rank
    return {"ok": bool(flag.item()), "identical_across_ranks": identical, "finite": finite,
            "max_abs_err_over_max_abs_vs_process_group_sum": err, "tolerance": tol, "backend": eng.comm_backend(), "gave_up": gave_up,
            "note": "one minibatch and draw twice, without and with the exchange: arena checksums equal on every rank; arena against "
                    "torch.distributed's all-reduce of the saved local gradients (fp32 summation order)"}


def time_other_backend(eng, torch, reps=5):
    """The buckets of the step through the exchange backend the step did NOT use, back to back on the idle GPUs: vbnn_p2p (the
    direct reduce-scatter + all-gather over peer-mapped arenas) when the step ran on RCCL, RCCL when it ran on vbnn_p2p -- so
    that ONE multi-GPU run prices both. Collective: every rank calls it. Never fatal: an error is reported."""
    from vbnn_amd import comm
    base = eng.grads.data_ptr()
    spans = [((b.data_ptr() - base) // 4, b.numel()) for b in eng.buckets()]
    out = {"buckets": []}
    ex = None
    try:
        if eng.exchange_kind == "p2p":
            import torch.distributed as dist
            if dist.get_backend(eng.pg) != "nccl":
                raise RuntimeError("RCCL wants one device per rank: not timed in a gloo rehearsal (ranks share devices)")
            ex = comm.RcclExchange(eng.ctx, eng.rank, eng.world, eng.pg)
            scratch = torch.zeros(eng.grads.numel(), dtype=torch.float32, device=eng.device)
        else:
            ex = comm.P2PExchange(eng.ctx, eng.rank, eng.world, eng.grads.numel(), eng.pg)
            scratch = ex.arena
        out["backend"] = ex.backend
        for off, n in spans:
            buf = scratch[off:off + n]
            for _ in range(2):
                ex.allreduce(buf); ex.finish()
            torch.cuda.synchronize(eng.device)
            t0 = time.perf_counter()
            for _ in range(reps):
                ex.allreduce(buf)
            ex.finish()
            torch.cuda.synchronize(eng.device)
            ms = (time.perf_counter() - t0) / reps * 1e3
            alg = n * 4 / (ms * 1e-3) / 1e9
            out["buckets"].append({"bytes": n * 4, "ms": round(ms, 4), "alg_GBps": round(alg, 1),
                                   "bus_GBps": round(alg * 2 * (eng.world - 1) / max(eng.world, 1), 1)})
        if hasattr(ex, "gave_up"):
            out["p2p_barrier_gave_up_epoch"] = int(ex.gave_up())
    except Exception as e:                      # noqa: BLE001 -- the headline must survive a backend this node cannot run
        out["error"] = f"{type(e).__name__}: {e}"[:300]
    finally:
        try:
            if ex is not None:
                ex.close()
        except Exception:                       # noqa: BLE001
            pass
    return out


def self_launch(n, argv):
    """`python bench.py --gpus N` from a bare shell: start the ranks as a CHILD process (never exec), before this
    process has imported torch or made any HIP call, and hand back the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def device_identity(torch, idx):
    p = torch.cuda.get_device_properties(idx)
    uuid = getattr(p, "uuid", None)
    return str(uuid) if uuid is not None else f"{p.name}#{getattr(p, 'pci_bus_id', idx)}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=7, help="timed blocks of --steps steps; ms_per_step is the median block (seven: "
                    "behind a short warm-up the first two blocks still run on the chip's clock ramp, and every block is reported)")
    ap.add_argument("--config", default="wide", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reporting-config", action="store_true", help="skip the 784-400-400-10 / batch 256 / fp32 row measured beside the headline")
    ap.add_argument("--overlap", action="store_true", help="run accGradParameters on a second stream (A/B; slower)")
    ap.add_argument("--mode", default="lrt", choices=["lrt", "wn"])
    ap.add_argument("--batch", type=int, default=0, help="exploration only: rows per GPU instead of the configuration's")
    ap.add_argument("--S", type=int, default=1, help="Monte-Carlo draws per minibatch (main.lua:32-37); the metric is quoted at 1")
    ap.add_argument("--stack-draws", action="store_true", help="with --S > 1: the draws as rows of ONE pass (FusedMLP.run_draws) instead of S passes")
    ap.add_argument("--host-input", action="store_true", help="the minibatch starts in (pinned) HOST memory every step: the PCIe-inclusive "
                    "rate, reported beside the metric in config.pcie_inclusive -- never `value`")
    ap.add_argument("--prepare-each-step", action="store_true", help="round 1's protocol: the parameter sweep inside every step")
    ap.add_argument("--with-update", action="store_true", help="(default since r04: on) also time step + optimiser update, reported beside the metric as train_step")
    ap.add_argument("--no-train-step", action="store_true", help="skip the train_step entry (step + FusedMLP.update)")
    ap.add_argument("--no-box", action="store_true", help="skip the box block (vbnn_box_calibrate: the MFMA clock / rate and the HBM rate this device holds)")
    ap.add_argument("--no-deep-config", action="store_true", help="skip the deep_config entry (BASELINE configs[4] measured beside the headline)")
    ap.add_argument("--debug-set", default="", help="A/B only: comma-separated key=value pairs for vbnn_debug_set")
    ap.add_argument("--backward-order", default="auto", choices=["auto", "dx-first", "layerwise"], help="A/B: every updateGradInput first "
                    "then the accGradParameters (auto: when the layers differ in size and there is no exchange), or layer by layer")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"], help="the step as ONE hipGraph launch (vbnn_capture_*, device-resident "
                    "draw counter); auto = off (measured slower than stream launches on this ROCm)")
    ap.add_argument("--cu-budget", type=int, default=0, help="data-parallel A/B: the engine's launches on a CU-masked stream of that many "
                    "compute units (vbnn_ctx_create_cu_budget), the rest left to RCCL's channels; 0 = no mask")
    ap.add_argument("--exchange", default="auto", choices=["auto", "rccl", "p2p"], help="N > 1: the gradient exchange -- the collective library "
                    "(vbnn_comm over RCCL), the direct peer-to-peer reduce-scatter + all-gather (vbnn_p2p), or auto: both are timed on a 64 MB "
                    "message at start-up and the faster one carries the step (comm.probe in the line); VBNN_EXCHANGE in the environment wins")
    ap.add_argument("--exchange-dtype", default="f32", choices=["f32", "bf16"], help="data-parallel exchange payload: fp32 (default, the "
                    "metric) or the optional bf16 copy (half the bytes, a DIFFERENT gradient: reported as an option, config.exchange_dtype)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    # multi-process GPU work on this driver needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); the boxes
    # export it already, this keeps a hand-built environment from dropping it
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from vbnn_amd import _lib as L
    from vbnn_amd.engine import FusedMLP
    from vbnn_amd.nn import fill_normal

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: the launcher's rank count must match",
              file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X: no HIP device is visible and there is no CPU fallback", file=sys.stderr)
        sys.exit(2)
    backend = os.environ.get("VBNN_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if world > ndev and backend == "nccl":
        print(f"bench.py --gpus {world}: {ndev} HIP device(s) visible and RCCL wants one per rank "
              "(a rehearsal on fewer devices: VBNN_DIST_BACKEND=gloo, ranks share devices)", file=sys.stderr)
        sys.exit(2)
    local_rank %= ndev
    torch.cuda.set_device(local_rank)
    # VBNN_FORCE_DIST=1: run the RCCL code path (process group, bucketed all-reduce, barriers) even with one
    # rank -- the only way to exercise it on a one-GPU box
    use_dist = world > 1 or os.environ.get("VBNN_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL on ROCm
        else:
            dist.init_process_group(backend)
    L.lib()                                               # fail loudly if the HIP extension is missing

    cfg = dict(CONFIGS[args.config])
    if args.batch > 0:
        cfg["batch"] = args.batch
        cfg["name"] += f" [batch overridden: {args.batch}]"
    if args.S != 1:
        cfg["name"] = cfg["name"].replace("S=1", f"S={args.S}" + (" (draws stacked as rows)" if args.stack_draws else ""))
    N = cfg["batch"]
    opt = dict(var_init=1e-3, B=1e6, S=args.S, mode=args.mode, dtype=cfg["dtype"], seed=3, input_size=cfg["input_size"],
               hidden=cfg["hidden"], n_classes=cfg["n_classes"], fuse_kl=True, overlap=args.overlap,
               criterion=cfg.get("criterion", "nll"), exchange_dtype=args.exchange_dtype, dx_first={"auto": None, "dx-first": True, "layerwise": False}[args.backward_order],
               state=dict(learningRate=1e-3), meanState=dict(learningRate=1e-4), varState=dict(learningRate=5e-2))
    # (auto = off everywhere: on this ROCm a captured step is bitwise the launched one but ~1.4 us per kernel node SLOWER than
    # the same kernels launched back to back on a stream -- 95 against 82 us for the small configuration, LAB_NOTES.md section 3)
    use_graph = args.mode == "lrt" and not use_dist and not args.overlap and args.graph == "on"
    if args.cu_budget > 0:
        opt["cu_budget"] = args.cu_budget
    stream = None
    if use_graph:                                         # a stream of the engine's own: the NULL stream cannot be captured
        stream = torch.cuda.Stream()
        torch.cuda.set_stream(stream)                     # (everything of this process from here on: fills, events, the step)
        opt["device_draw"] = True
    xprobe = None
    if use_dist and not os.environ.get("VBNN_EXCHANGE"):
        if args.exchange == "p2p":
            opt["exchange"] = "p2p"
        elif args.exchange == "auto" and world > 1:
            xprobe = probe_exchange_backends(torch, dist, rank, world, local_rank)
            if xprobe["choice"] == "p2p":
                opt["exchange"] = "p2p"
    eng = FusedMLP(opt, world_size=world, rank=rank, force_reduce=use_dist, stream=stream)
    if args.cu_budget > 0:
        torch.cuda.set_stream(eng.ctx.torch_stream)      # the masked stream is the process's stream from here on
    for kv in filter(None, args.debug_set.split(",")):
        k, v = kv.split("=")
        L.check(L.lib().vbnn_debug_set(int(k), int(v)))
    # synthetic minibatch, resident in HBM: x ~ N(0,1) addressed by GLOBAL row, targets uniform in 0..9
    x = torch.empty(N, cfg["input_size"], dtype=torch.float32, device="cuda")
    fill_normal(x, 3, L.STREAM_DATA, 0, 0, row0=rank * N)
    t = eng.synthetic_targets(x, rank * N)

    def step():
        eng.resetGradients()
        if args.prepare_each_step:
            eng.prepare()
        if args.stack_draws and args.S > 1:
            eng.run_draws(x, t, args.S)
        else:
            for _ in range(args.S):
                eng.sample()
                eng.run(x, t)
        eng.finish()

    issue = step                                          # the step launch by launch (warm-up, capture, the probed block)
    graph = None

    def train_step():
        step()
        eng.update()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_blocks(fn, repeats):
        """`repeats` blocks of exactly --steps calls of fn: wall clock between barrier + synchronize on both sides (MAX
        over ranks) and HIP events on the launch stream around the same block."""
        wall, evms = [], []
        for _ in range(repeats):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            barrier()
            t0 = time.perf_counter()
            e0.record()
            for _ in range(args.steps):
                fn()
            e1.record()
            barrier()
            el = time.perf_counter() - t0
            if use_dist:
                tt = torch.tensor([el], dtype=torch.float64, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            wall.append(el / args.steps * 1e3)
            evms.append(e0.elapsed_time(e1) / args.steps)
        return wall, evms

    eng.prepare()                                        # once: afterwards the update kernel maintains shadows + statistics
    xval = None
    if use_dist and world > 1 and args.S == 1 and not args.stack_draws:
        # the exchange checks itself before anything is timed; the direct exchange failing that check falls back to the library once
        xval = validate_exchange(eng, x, t, torch, dist, world)
        if not xval["ok"] and eng.exchange_kind == "p2p" and backend == "nccl":
            first = xval
            del eng
            torch.cuda.empty_cache()
            opt.pop("exchange", None)
            eng = FusedMLP(opt, world_size=world, rank=rank, force_reduce=use_dist, stream=stream)
            eng.prepare()
            xval = validate_exchange(eng, x, t, torch, dist, world)
            xval["fell_back_from"] = first
        if not xval["ok"]:
            if rank == 0:
                print(json.dumps({"metric": "VBLinear fwd+bwd samples/sec", "value": None, "n_gpus": world, "exchange_failed": "the exchange did not sum",
                                  "exchange_validation": xval}), flush=True)
            sys.exit(3)
    for _ in range(args.warmup):
        step()
    # the box's own speed, measured here and now (rank 0's device; every rank runs it so the ranks stay in step): after the
    # warm-up -- the clock has ramped -- and before the timed blocks; once more behind them (`after_timed_blocks`)
    box = None
    if not args.no_box:
        try:
            torch.cuda.synchronize()
            box = box_calibration(L, eng.ctx.h)
            for _ in range(3):
                step()                                   # (the probes' last launches are a different load: back into the step's regime)
        except Exception as e:                           # noqa: BLE001 -- never at the price of the headline
            box = {"error": f"{type(e).__name__}: {e}"[:300]}
    if use_graph:
        torch.cuda.synchronize()
        graph = eng.capture_step(issue)                   # the same calls, recorded once; every replay draws its own noise

        def step():                                       # noqa: F811 -- from here on a step is ONE graph launch
            graph.launch()
        for _ in range(3):
            step()
    wall, evms = timed_blocks(step, max(1, args.repeats))
    # (a p2p barrier that gave up: the sums are not sums and the timing means nothing -- say so and fail, on every rank)
    try:
        eng.check_exchange()
    except RuntimeError as e:
        if rank == 0:
            print(json.dumps({"metric": "VBLinear fwd+bwd samples/sec", "value": None, "n_gpus": world, "exchange_failed": str(e)}), flush=True)
        sys.exit(3)
    loss, correct = eng.loss_and_accuracy()
    ms = sorted(wall)[len(wall) // 2]
    if box is not None and "error" not in box:
        try:
            box["after_timed_blocks"] = {k: v for k, v in box_calibration(L, eng.ctx.h).items() if k in ("mfma_clock_ghz", "mfma_tflops", "hbm_TBps")}
        except Exception as e:                           # noqa: BLE001
            box["after_timed_blocks"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    # the SAME step (same launch order, same two-launch accGradParameters) with the collective calls left out: what the
    # exchange costs beyond the compute it overlaps with = ms - this
    noex = None
    if use_dist:
        eng.skip_exchange = True
        for _ in range(3):
            step()
        nw, _ = timed_blocks(step, min(3, max(1, args.repeats)))
        eng.skip_exchange = False
        step(); barrier()                                 # the arena holds exchanged sums again
        noex = {"ms_per_step": round(sorted(nw)[len(nw) // 2], 4), "repeats_wall_ms": [round(v, 4) for v in nw]}
        noex["exposed_exchange_ms"] = round(ms - noex["ms_per_step"], 4)

    # HIP events around every launch of the widest layer's three dual GEMMs, on the stream they are launched on, in one
    # more block of --steps steps of the same step (rank 0; kept out of the blocks above: an event record is a marker
    # packet between two kernels, six per step cost ~3 % of the step)
    wide_li = max(range(len(eng.vb)), key=lambda k: eng.vb[k].I * eng.vb[k].O)
    eng.probe = (wide_li, {}) if rank == 0 else None
    barrier()
    tp0 = time.perf_counter()
    for _ in range(args.steps):
        issue()                                           # launch by launch: events cannot be placed inside a graph replay
    barrier()
    probed_ms = (time.perf_counter() - tp0) / args.steps * 1e3
    probe, eng.probe = eng.probe, None
    # what a kernel boundary costs inside a graph on this box, measured here on the device: a captured graph of 64 dependent
    # one-thread kernels (vbnn_sample with an increment of 0), replayed 20 times
    floor_us = None
    if rank == 0 and eng.device_draw and graph is not None:
        import ctypes as C
        lib = L.lib()
        dd = C.c_void_p(eng._draw_dev.data_ptr())
        L.check(lib.vbnn_sample(eng.ctx.h, dd, 0))
        torch.cuda.synchronize()
        L.check(lib.vbnn_capture_begin(eng.ctx.h))
        for _ in range(64):
            L.check(lib.vbnn_sample(eng.ctx.h, dd, 0))
        gh = C.c_void_p()
        L.check(lib.vbnn_capture_end(eng.ctx.h, C.byref(gh)))
        for _ in range(3):
            L.check(lib.vbnn_graph_launch(gh))
        fe0, fe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        fe0.record()
        for _ in range(20):
            L.check(lib.vbnn_graph_launch(gh))
        fe1.record()
        torch.cuda.synchronize()
        floor_us = fe0.elapsed_time(fe1) * 1e3 / (20 * 64)
        L.check(lib.vbnn_graph_destroy(gh))

    pcie = None
    if args.host_input:
        xh, th = x.cpu().pin_memory(), t.cpu().pin_memory()
        xd, td = torch.empty_like(x), torch.empty_like(t)

        def host_step():
            xd.copy_(xh, non_blocking=True); td.copy_(th, non_blocking=True)      # H2D on the launch stream, then the step
            eng.resetGradients()
            eng.sample()
            eng.run(xd, td)
            eng.finish()
        for _ in range(3):
            host_step()
        hw, _ = timed_blocks(host_step, max(1, args.repeats))
        hms = sorted(hw)[len(hw) // 2]
        pcie = {"ms_per_step": round(hms, 4), "samples_per_s": round(N * world / (hms * 1e-3), 1),
                "h2d_bytes_per_step": int(x.numel() * 4 + t.numel() * t.element_size()),
                "note": "minibatch copied from pinned host memory at the head of every step (no overlap with the previous step)"}

    train = None
    if (args.with_update or not args.no_train_step) and args.mode == "lrt" and not use_graph:
        for _ in range(3):
            train_step()
        tw, te = timed_blocks(train_step, max(1, args.repeats) if args.with_update else min(3, max(1, args.repeats)))
        tms = sorted(tw)[len(tw) // 2]
        fps_t = algorithmic_flops_per_sample([cfg["input_size"]] + cfg["hidden"], cfg["n_classes"]) * args.S
        train = {"ms_per_train_step": round(tms, 4), "samples_per_s": round(N * world * args.S / (tms * 1e-3), 1),
                 "repeats_wall_ms": [round(v, 4) for v in tw],
                 "step_frac_of_mfma_peak": round(fps_t * N / (tms * 1e-3) / 1e12 / PEAK_TFLOPS[cfg["dtype"]], 4),
                 "kl_gradient": ("exact: added by the update sweep from the fp32 means / lvars (vbnn_update_desc.kl_add, VBLinear.lua:91,96)"
                                 if eng.kl_in_update else "fused into the accGradParameters epilogue"),
                 "note": "the whole training step of main.lua:28-40: the metric's step + mlp:update / VBLinear:update (VBLinear.lua:124-166) "
                         "as ONE sweep per layer (Adam on means / lvars, SGD on biases and the final Linear, the next minibatch's operand "
                         "shadows and prior statistics); the update's flops are not counted in the fraction (same numerator as the metric)"}

    comm = None
    if use_dist:
        ident = [None] * world
        dist.all_gather_object(ident, (rank, local_rank, device_identity(torch, local_rank), os.getpid()))
        comm = {"backend": eng.comm_backend(), "payload": eng.exchange_dtype, "ranks_seen": [list(i) for i in ident],
                "distinct_devices": len({i[2] for i in ident}), "allreduce": eng.time_buckets(5)}
        if xprobe is not None:
            comm["probe"] = xprobe
        if xval is not None:
            comm["validation"] = xval
        if noex is not None:
            comm["step_without_exchange"] = noex
        if world > 1 and os.environ.get("VBNN_BENCH_OTHER_BACKEND", "1") != "0":
            comm["other_backend"] = time_other_backend(eng, torch)
        st = 0
        if hasattr(eng.exchange(), "gave_up"):
            st = int(eng.exchange().gave_up())
        comm["p2p_barrier_gave_up_epoch"] = st if eng.exchange_kind == "p2p" else None
        comm["scaling_note"] = ("no scaling number exists until this line has been produced on N > 1 DEVICES: " +
                                (f"{len({i[2] for i in ident})} distinct device(s) took part in this run"))
        ex = eng.exchange()
        if hasattr(ex, "gather_u64"):
            # the same census over the EXCHANGE's own communicator (vbnn_comm_allgather_u64: RCCL's all-gather on the stream
            # the gradient buckets travel on), so the line proves that N ranks on N devices took part in THAT communicator --
            # not just in torch.distributed's process group. One word per rank: rank | local rank | pid | hash of the device uuid.
            import zlib
            dev_hash = zlib.crc32(device_identity(torch, local_rank).encode()) & 0xFFFF
            word = (rank & 0xFF) << 56 | (local_rank & 0xFF) << 48 | (os.getpid() & 0xFFFFFFFF) << 16 | dev_hash
            words = ex.gather_u64(word)
            seen = [{"rank": w >> 56 & 0xFF, "local_rank": w >> 48 & 0xFF, "pid": w >> 16 & 0xFFFFFFFF, "device_uuid_crc16": w & 0xFFFF} for w in words]
            comm["rccl_ranks_seen"] = seen
            comm["rccl_distinct_devices"] = len({(r["device_uuid_crc16"], r["local_rank"]) for r in seen})
            comm["rccl_census_matches_process_group"] = sorted((r["rank"], r["pid"] & 0xFFFFFFFF) for r in seen) == sorted((i[0], i[3] & 0xFFFFFFFF) for i in ident)

    if rank == 0:
        sizes = [cfg["input_size"]] + cfg["hidden"]
        fps = algorithmic_flops_per_sample(sizes, cfg["n_classes"]) * args.S
        peak = PEAK_TFLOPS[cfg["dtype"]]
        live = {LIVE_NAMES[k]: sum(a.elapsed_time(b) for a, b in ev) / len(ev) for k, ev in probe[1].items() if ev}
        _, _, kflops, kall = dominant_kernel_roofline(eng, N, torch)      # the same launches in isolation (cross-check)
        kname = max(live, key=live.get)
        kms = live[kname]
        achieved = kflops / (kms * 1e-3) / 1e12
        out = {
            "metric": "VBLinear fwd+bwd samples/sec", "value": round(N * world * args.S / (ms * 1e-3), 1),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": cfg["dtype"], "data": "synthetic",
            "config": {"workload": cfg["name"], "mode": args.mode, "global_batch": N * world,
                       "parallelism": f"dp{world}", "flop_per_sample": fps,
                       "step_tflops": round(fps * N / (ms * 1e-3) / 1e12, 2),
                       "step_frac_of_mfma_peak": round(fps * N / (ms * 1e-3) / 1e12 / peak, 4),
                       "loss": round(loss, 5),
                       "step": ("resetGradients, prepare, S x (sample, run), all-reduce" if args.prepare_each_step else
                                "resetGradients, S x (sample, run), all-reduce; the parameter sweep (compute_prior + operand "
                                "shadows) belongs to the excluded update, as VBLinear.lua:130 has it: prepare() ran once"),
                       "timing": f"median of {len(wall)} blocks of {args.steps} steps (all in repeats_wall_ms; behind a short warm-up the first blocks run on the clock ramp), wall clock between barrier+synchronize, MAX over ranks",
                       "repeats_wall_ms": [round(v, 4) for v in wall], "repeats_event_ms": [round(v, 4) for v in evms],
                       "probed_block_ms": round(probed_ms, 4),
                       "backward_order": ("every updateGradInput, then the accGradParameters from the first layer up" if eng.dx_first
                                          and not eng.reduce else "layer by layer, accGradParameters (+ its all-reduce) first"),
                       "cu_budget": (getattr(eng.ctx, "cu_budget", None) if args.cu_budget > 0 else None),
                       "launch": (f"ONE hipGraph launch per step: {graph.kernel_nodes} kernel nodes captured once (vbnn_capture_*), the draw "
                                  "counter in device memory (vbnn_sample is a node), bitwise the launch-by-launch step" if graph else
                                  "one stream launch per kernel")},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": measured_traffic(kname, args.config),
                         "traffic_source": "profiles/r*_traffic.json: separate rocprofv3 --pmc passes of the same launch "
                                           "(tools/collect_traffic.py), NOT measured in this run", "kernel": kname,
                         "kernel_ms": round(kms, 4), "flop_per_launch": kflops,
                         "timed_region_kernels_ms": {k: round(v_, 4) for k, v_ in live.items()},
                         "isolated_kernels_ms": kall},
        }
        if box is not None:
            out["box"] = box
            ghz = box.get("mfma_clock_ghz")
            if ghz:
                # 2.5 PFLOP/s (157.3 TFLOP/s fp32) is the dense peak at 2.4 GHz; at the clock this box holds under a pure matrix load the
                # ceiling is peak x ghz / 2.4 -- the same kernel time against THAT ceiling
                out["roofline"]["held_clock_ghz"] = ghz
                out["roofline"]["peak_at_held_clock"] = round(peak * ghz / 2.4, 1)
                out["roofline"]["frac_at_held_clock"] = round(achieved / (peak * ghz / 2.4), 4)
                out["config"]["step_frac_of_mfma_peak_at_held_clock"] = round(fps * N / (ms * 1e-3) / 1e12 / (peak * ghz / 2.4), 4)
        if floor_us is not None and graph is not None:
            out["roofline"]["launch_floor"] = {"us_per_graph_node": round(floor_us, 3), "kernel_nodes": graph.kernel_nodes,
                                               "ms_per_step": round(floor_us * graph.kernel_nodes * 1e-3, 4),
                                               "note": "a captured graph of 64 dependent one-thread kernels replayed 20 times on the step's stream, in this "
                                                       "run: what the step's kernel boundaries alone cost; the step's MFMA time at the peak is "
                                                       f"{fps * N / (peak * 1e12) * 1e3:.4f} ms"}
        if train:
            out["config"]["train"] = train
            out["train_step"] = train
        if pcie:
            out["config"]["pcie_inclusive"] = pcie
        if comm:
            out["comm"] = comm
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        if world == 1 and args.config == "wide" and not use_graph and args.cu_budget == 0 and not args.no_reporting_config:
            out["reporting_config"] = reporting_config_row(torch, L, FusedMLP, fill_normal, not args.no_cpu_baseline)
    # BASELINE.json configs[4] beside the headline, every rank (its step has its own exchange when N > 1)
    deep = None
    if args.config == "wide" and not args.no_deep_config and not use_graph and args.cu_budget == 0 and args.mode == "lrt" and args.S == 1:
        try:
            del eng
            torch.cuda.empty_cache()
            deep = side_config_row("deep", torch, dist, L, FusedMLP, fill_normal, min(args.steps, 20), min(args.warmup, 5), world, rank, use_dist,
                                   exchange=opt.get("exchange"))
        except Exception as e:                  # noqa: BLE001 -- never at the price of the headline
            deep = {"error": f"{type(e).__name__}: {e}"[:300]}
    sharded = None
    if use_dist and cfg["dtype"] == "bf16" and args.mode == "lrt" and args.S == 1 and not args.no_train_step and cfg.get("criterion", "nll") == "nll":
        try:
            sharded = sharded_train_row(cfg, torch, dist, L, FusedMLP, fill_normal, min(args.steps, 20), min(args.warmup, 5), world, rank,
                                        exchange=opt.get("exchange"))
        except Exception as e:                  # noqa: BLE001
            sharded = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0:
        if sharded is not None:
            out["train_step_sharded"] = sharded
        if deep is not None:
            out["deep_config"] = deep
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
