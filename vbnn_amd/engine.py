"""Fused whole-step engine: mlp.lua's protocol (resetGradients / sample / run / calc_lc) with the
elementwise modules folded into the GEMM epilogues, activations and packed operands resident in
HBM, one process per GPU and an RCCL all-reduce of the gradient arena after accGradParameters.

Per minibatch (main.lua:28-40):   resetGradients(); prepare(); S x { sample(); run(x, t) }
One `run` = forward of every VB layer (dual MFMA GEMM, Philox noise + ReLU + operand packing in the
epilogue), final Linear + LogSoftMax + ClassNLL, backward (gradInput GEMMs with the ReLU mask and the
dL/dv hand-off in the epilogue, accGradParameters GEMMs with the gradSum / KL-gradient epilogue).

Everything here is argument plumbing around include/vbnn_hip.h; torch is device memory, the
stream and torch.distributed.
"""
import contextlib
import ctypes as C
import functools
import math

import torch

from . import _lib as L
from . import partition
from .nn import Context, _DT, _Packed, _p, fill_normal


class _VB:
    pass


_NULL_CM = contextlib.nullcontext()


def _ordered(fn):
    """Engine entry points that launch or allocate: the body runs with the engine's stream as torch's current stream, that
    stream ordered behind whatever the caller queued on ITS current stream before the call (FusedMLP._on_stream)."""
    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        with self._on_stream():
            return fn(self, *args, **kwargs)
    return wrapped


class _View:
    """A caller's 2-D fp32 tensor handed to a GEMM as it lies (ptr, row pitch): the raw minibatch of the fp32 path."""

    def __init__(self, t):
        self.t, self.ld = t, t.stride(0)

    @property
    def ptr(self):
        return C.c_void_p(self.t.data_ptr())


class _StepGraph:
    """A captured step (FusedMLP.capture_step): launch() replays it on the engine's stream."""

    def __init__(self, eng, handle, draws):
        self.eng, self.h, self.draws = eng, handle, draws
        k, n = C.c_int(), C.c_int()
        L.check(L.lib().vbnn_graph_info(self.h, C.byref(k), C.byref(n)))
        self.kernel_nodes, self.nodes = k.value, n.value

    def launch(self):
        # the replay reads the captured input buffers: it goes behind the copies the caller queued on its own stream
        with self.eng._on_stream():
            L.check(L.lib().vbnn_graph_launch(self.h))
        self.eng.draw += self.draws                      # the host's mirror of the device counter
        self.eng._first = False

    def close(self):
        if self.h:
            L.check(L.lib().vbnn_graph_destroy(self.h))
            self.h = None


class FusedMLP:
    def __init__(self, opt, device=None, world_size=1, rank=0, process_group=None, force_reduce=False, stream=None):
        """stream: a torch.cuda.Stream for this engine's launches (a library context of its own, hence its own reduction
        scratch and tickets): two engines on two streams may compute on one device at the same time. Default: the
        device's shared context on torch's current stream."""
        self.opt = opt
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        # opt.cu_budget (data-parallel runs): the engine's launches go on a stream of the library's own limited to that many
        # compute units, leaving the rest to RCCL's channels for the whole step (Context.with_cu_budget)
        cu_budget = int(opt.get("cu_budget", 0) or 0)
        if cu_budget > 0:
            assert stream is None, "cu_budget creates the engine's stream itself"
            self.ctx = Context.with_cu_budget(self.device.index or 0, cu_budget)
        else:
            self.ctx = Context.get(self.device) if stream is None else Context(self.device.index or 0, stream=stream)
        with self._on_stream():                # every allocation / fill below is queued on the engine's own stream
            self._build(opt, world_size, rank, process_group, force_reduce)

    def _on_stream(self):
        """With a stream of the engine's own (stream= / cu_budget) the library launches there while torch's fills, copies
        and allocations go to torch's CURRENT stream: make the ordering the engine's job. The engine stream
        first waits for the caller's current stream (a minibatch copied there is complete before a kernel reads it), then
        is torch's current stream for the body (the engine's own zero fills / repeats queue in launch order). Results are
        ordered for the caller by join() or any device synchronisation (loss_and_accuracy does one)."""
        ts = self.ctx.torch_stream
        if ts is None:
            return _NULL_CM
        cur = torch.cuda.current_stream(self.device)
        if cur != ts:
            ts.wait_stream(cur)
        return torch.cuda.stream(ts)

    def join(self):
        """The caller's current stream waits for everything this engine has queued (no host block)."""
        ts = self.ctx.torch_stream
        if ts is not None:
            cur = torch.cuda.current_stream(self.device)
            if cur != ts:
                cur.wait_stream(ts)

    def _build(self, opt, world_size, rank, process_group, force_reduce):
        self.dtype = opt.get("dtype", "bf16")
        self.code, self.tdt = _DT[self.dtype]
        self.mode = opt.get("mode", "lrt")
        assert self.mode in ("wn", "lrt")
        self.seed = int(opt.get("seed", 3))
        self.world, self.rank, self.pg = world_size, rank, process_group
        self.reduce = world_size > 1 or force_reduce
        self._exchange = None                 # made on first use (vbnn_amd/comm.py): RCCL through the C ABI
        self.exchange_dtype = str(opt.get("exchange_dtype", "f32"))      # "bf16": the optional half-size exchange (_reduce)
        assert self.exchange_dtype in ("f32", "bf16")
        self._stage, self._staged = None, []
        # opt.exchange_mode = "sharded" (or VBNN_EXCHANGE_MODE=sharded): the SHARDED-UPDATE exchange instead of north_star's
        # all-reduce (partition.py): reduce-scatter of the likelihood gradients by layer rows, update() on this
        # rank's rows only (Adam state + fp32 master rows sharded), all-gather of the operand shadows + prior statistics. An option
        # beside the all-reduce, never its replacement: update() is then part of every step.
        import os as _os
        self.exchange_mode = str(opt.get("exchange_mode", _os.environ.get("VBNN_EXCHANGE_MODE", "")) or "allreduce")
        assert self.exchange_mode in ("allreduce", "sharded")
        self.sharded = self.reduce and self.exchange_mode == "sharded"
        self._params_stale = False            # sharded: the fp32 means / lvars of OTHER ranks' rows are stale after an update
        self._dx_first_opt = opt.get("dx_first", None)      # None: decided from the layer sizes once they are known (below)
        self.fuse_kl = bool(opt.get("fuse_kl", True))
        # "nll": LogSoftMax + ClassNLLCriterion (mlp.lua:30-32); "mse": nn.MSECriterion on the final Linear's outputs
        # (BASELINE.json configs[4], a regression target of n_classes dimensions -- not in the reference)
        self.criterion = opt.get("criterion", "nll")
        assert self.criterion in ("nll", "mse")
        self.kl_from_shadows = self.dtype == "bf16" and bool(opt.get("kl_from_shadows", True))
        # opt.kl_in_update: the gradient arena holds the LIKELIHOOD parts only (vbnn_dw_args.kl_scale = 0) and update() adds the
        # KL gradient from the fp32 means / lvars (vbnn_update_desc.kl_add) -- exact, where the epilogue's shadow form carries
        # bf16(sigma^2) / var_hat - 1; in a data-parallel run the exchanged sum is then a pure sum of likelihoods.
        # DEFAULT (r04): on wherever the fused epilogue would otherwise take the KL terms from the bf16 shadows -- the exact form is
        # what every host gets unless it asks for the A/B form (kl_in_update = False); fp32 epilogues read the fp32 parameters
        # and are exact either way, so they keep the total gradients in the arena.
        self.kl_in_update = bool(opt.get("kl_in_update", self.kl_from_shadows and self.fuse_kl and self.mode == "lrt"))
        if self.sharded:
            assert self.fuse_kl and self.mode == "lrt" and self.exchange_dtype == "f32" and not opt.get("overlap", False), \
                "exchange_mode = sharded: the fused LRT engine, fp32 sums, one compute stream"
            # bf16 only: there the step reads the parameters through the operand shadows alone (GEMM operands and the
            # accGradParameters epilogue's sigma^2: vbnn_dw_args.var_s) -- which is what is gathered, at half the bytes of the
            # gradients. The fp32 configuration's epilogue forms exp(lvars) from the fp32 master rows, which other ranks hold stale,
            # and its "shadows" are as large as the gradients: nothing to gain there.
            assert self.dtype == "bf16" and self.kl_from_shadows, "exchange_mode = sharded: the bf16 configuration (operand shadows)"
            self.kl_in_update = True          # what is reduce-scattered is a pure sum of likelihood parts; the slice update adds the KL
        assert not self.kl_in_update or (self.fuse_kl and self.mode == "lrt"), "kl_in_update: the fused LRT engine"
        # optional second HIP stream (+ its own context, hence its own reduction scratch) for the accGradParameters
        # GEMMs. Measured on MI355X (wide config): 1.32 ms with, 1.29 ms without -- two 512-block GEMMs sharing the
        # CUs thrash each other's L2 panels more than the staggered epilogues save -- so it is off by default.
        self.overlap = bool(opt.get("overlap", False))
        self.head_step = bool(opt.get("head_step", True))     # the fused head's forward + backward as one call (A/B, tests: False = two calls)
        if self.overlap:
            self._side = torch.cuda.Stream(device=self.device)
            self.ctx2 = Context(self.device.index or 0, stream=self._side)
        self.B, self.S = float(opt.get("B", 1e6)), float(opt.get("S", 1))
        hidden = list(opt["hidden"])
        sizes = [opt["input_size"]] + hidden
        self.n_classes = opt["n_classes"]
        self.sizes = sizes
        weights = [sizes[i] * sizes[i + 1] for i in range(len(sizes) - 1)]
        self.dx_first = bool(self._dx_first_opt) if self._dx_first_opt is not None else (2 * min(weights) <= max(weights))
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        # ---- gradient arena: [gradWeight | gradSum | gradBias] per VB layer, then the final Linear (partition.arena_layout)
        # early_lv: layers whose accGradParameters runs as two launches (d/dlvars first) so that its exchange starts one
        # GEMM earlier; default: the big layers of a data-parallel run
        el = opt.get("early_lv", None)
        if el is None:
            el = [(world_size > 1 or force_reduce) and self.mode == "lrt" and bool(opt.get("fuse_kl", True)) and
                  sizes[i] * sizes[i + 1] >= (1 << 22) for i in range(len(hidden))]
        elif isinstance(el, bool):
            el = [el and self.mode == "lrt"] * len(hidden)
        self.early_lv = list(el)
        lay, fin, n_g, self._bucket_ranges = partition.arena_layout(sizes, self.n_classes, self.early_lv)
        self._lay = lay
        # opt.exchange = "p2p" (or VBNN_EXCHANGE=p2p): the direct reduce-scatter + all-gather over peer-mapped arenas
        # (vbnn_p2p_*, csrc/p2p.hip) instead of RCCL; that exchange owns the arena (an IPC-exportable allocation of the library's)
        import os
        self.exchange_kind = str(opt.get("exchange", os.environ.get("VBNN_EXCHANGE", "")) or "")
        n_layers = len(hidden)
        self._plan = partition.sharded_plan(lay, world_size) if self.sharded else None
        arena_extra = None
        if self.reduce and self.exchange_kind == "p2p":
            assert self.exchange_dtype == "f32", "the p2p exchange sums fp32 arenas"
            from .comm import P2PExchange
            n_arena = (n_g + 3) // 4 * 4
            if self.sharded:            # what the sharded exchange all-gathers lives in the peer-mapped arena too
                n_arena += sum(2 * _Packed.floats(sizes[i + 1], sizes[i], self.tdt) for i in range(n_layers)) + 8 * world_size * n_layers
            self._exchange = P2PExchange(self.ctx, rank, world_size, n_arena, process_group)
            self.grads = self._exchange.arena[:n_g]
            arena_extra = self._exchange.arena[(n_g + 3) // 4 * 4:]
        else:
            self.grads = torch.zeros(n_g, **f32)

        def carve(n_floats):
            nonlocal arena_extra
            if arena_extra is None:
                return None
            piece, arena_extra = arena_extra[:n_floats], arena_extra[n_floats:]
            return piece

        def take(span, shape):
            return self.grads[span[0]:span[0] + span[1]].view(*shape)

        var_init = opt["var_init"] if not opt.get("msr_init") else None
        self.vb = []
        for li in range(len(hidden)):
            I, O = sizes[li], sizes[li + 1]
            v = _VB()
            v.I, v.O, v.layer_id = I, O, li
            v.var_init = var_init if var_init is not None else 2.0 / I           # VBLinear.lua:12-16
            v.means = torch.zeros(O, I, **f32)                                     # :22-23
            v.lvars = torch.full((O, I), math.log(v.var_init), **f32)              # :18
            v.bias = torch.zeros(O, **f32)                                         # :13
            v.weight = torch.zeros(O, I, **f32) if self.mode == "wn" else None
            # with fuse_kl the first two blocks hold the TOTAL gradients d/dmeans, d/dlvars
            # (likelihood / S + KL, VBLinear.lua:90-98); otherwise the reference's gradWeight / gradSum
            v.gradWeight = take(lay[li]["mu"], (O, I))
            v.gradSum = take(lay[li]["lv"], (O, I))
            v.gradBias = take(lay[li]["bias"], (O,))
            v.bucket = self.grads[lay[li]["bucket"][0]:lay[li]["bucket"][1]]
            v.stats = torch.zeros(4, dtype=torch.float64, device=dev)
            nf = _Packed.floats(O, I, self.tdt)
            v.mu_s = _Packed(O, I, self.tdt, dev, storage=carve(nf) if self.sharded else None)
            v.var_s = _Packed(O, I, self.tdt, dev, storage=carve(nf) if self.sharded else None)
            if li > 0:
                v.muT_s, v.varT_s = _Packed(I, O, self.tdt, dev), _Packed(I, O, self.tdt, dev)
            else:
                v.muT_s = v.varT_s = None
            self.vb.append(v)
        H = sizes[-1]
        self.weight3 = torch.zeros(self.n_classes, H, **f32)
        self.bias3 = torch.zeros(self.n_classes, **f32)
        self.gradWeight3 = take(fin["weight"], (self.n_classes, H))
        self.gradBias3 = take(fin["bias"], (self.n_classes,))
        self.bucket3 = self.grads[fin["bucket"][0]:fin["bucket"][1]]
        # the final Linear's gradients sit right behind the last VB layer's in the arena: ONE all-reduce covers both (the
        # head's backward finishes before that layer's accGradParameters is even launched), a collective fewer per step
        self.bucket_tail = self.grads[lay[-1]["bucket"][0]:fin["bucket"][1]]
        for li, v in enumerate(self.vb):                         # the layer's message(s): [d/dlvars] early (optional), the rest late
            v.msg_early = self.grads[lay[li]["early"][0]:lay[li]["early"][1]] if lay[li]["early"] else None
            v.msg_late = self.grads[lay[li]["late"][0]:lay[li]["late"][1]]
        self.w3_s = _Packed(self.n_classes, H, self.tdt, dev)
        self.w3T_s = _Packed(H, self.n_classes, self.tdt, dev)
        if self.sharded:
            # every rank's statistics of its row slices, [world][layers][4] doubles, gathered after the slice update; and this
            # rank's block of it as the in / out `stats` of the slice update (in: the whole layer's pre-update statistics)
            st = carve(8 * world_size * n_layers)
            self._stat_parts = (st.view(torch.float64) if st is not None else
                                torch.zeros(4 * world_size * n_layers, dtype=torch.float64, device=dev)).view(world_size, n_layers, 4)
            self._stat_parts.zero_()
        self._acc = torch.zeros(2, dtype=torch.float64, device=dev)
        self._corr = torch.zeros(1, dtype=torch.int32, device=dev)
        self._lc = torch.zeros(1, dtype=torch.float64, device=dev)
        self.draw = 0
        # opt.device_draw: the draw counter lives in device memory (vbnn_fwd_args.draw_dev), sample() is a device-side
        # increment (vbnn_sample) -- what makes a step CAPTURABLE (capture_step): every replay of the graph advances the
        # counter and draws its own noise. `self.draw` stays the host's mirror of the count.
        self.device_draw = bool(opt.get("device_draw", False))
        assert not self.device_draw or self.mode == "lrt", "device_draw: LRT (weight noise draws in vbnn_wn_sample, a host-addressed launch)"
        self._draw_dev = torch.zeros(1, dtype=torch.int32, device=dev) if self.device_draw else None
        self._map = False
        self._first = True
        self._draws_run = 0
        self._exchange_now = True
        self._exchange_done = False
        self._N = None
        self._argcache = {}
        self._rpd = 0            # > 0 inside run_draws: rows per Monte-Carlo draw of the stacked minibatch
        self._x_in = None        # fp32: the raw minibatch of the current run, read in place by layer 1's GEMMs
        self._draws = None
        self.init_parameters()

    # mlp.lua:47-55 (He rule for every weight, bias zero) + the bench's non-degenerate means
    # (means ~ N(0, sqrt(2/I)); the shipped mu_init = 0 gives an all-zero net).
    @_ordered
    def init_parameters(self):
        for v in self.vb:
            fill_normal(v.means, self.seed, L.STREAM_HEINIT, v.layer_id, 0, scale=math.sqrt(2.0 / v.I), ctx=self.ctx)
            v.bias.zero_()
        fill_normal(self.weight3, self.seed, L.STREAM_HEINIT, len(self.vb), 0, scale=math.sqrt(2.0 / self.sizes[-1]), ctx=self.ctx)
        self.bias3.zero_()

    # ---- buffers that depend on the local batch size
    def _alloc_batch(self, N):
        if self._N == N:
            return
        self._N = N
        self._argcache = {}                     # the argument blocks hold pointers into the buffers allocated below
        dev, tdt = self.device, self.tdt
        lib = L.lib()
        # K-major operands (vbnn_kmajor_supported): when a layer's parameter-gradient GEMM reads x and g as the forward
        # / gradInput GEMMs hold them, nobody has to write x^T, (x.x)^T, g^T, gv^T for it; when its gradInput GEMM reads
        # mu, sigma^2 as stored, the parameter sweep writes no transposed shadows. The library decides by shape; the
        # fused configuration that takes that kernel is bf16, LRT, total gradients from the epilogue, one draw.
        km_ok = (self.opt.get("dtype", "f32") == "bf16") and not self.opt.get("keep_transposes", False)
        # fp32 (the general kernel, gemm_v1.h): every GEMM takes its operands as the producer left them -- the minibatch raw,
        # x.x formed in registers, x / g / mu / sigma^2 K-major, the bias gradient from a synthetic row of ones -- so the fp32
        # step has no packing launch, no squared and no transposed copy of anything. (Not with an exchange: the two-launch
        # accGradParameters of a data-parallel run wants x.x as an operand.)
        self.f32_direct = (self.dtype == "f32" and self.mode == "lrt" and not self.reduce and
                           not self.opt.get("keep_transposes", False))
        need_prepare = False
        for li, v in enumerate(self.vb):
            last_fused = (v is self.vb[-1]) and self.n_classes <= 16 and self.criterion == "nll"
            v.bias_from_dw = (v.I % 256 != 0) and not last_fused
            if self.f32_direct:
                v.dw_km, v.x_pad256, v.early_ok, v.dx_km = True, False, False, li > 0
                v.use_muT = False
                continue
            # (the K-major launches take the epilogue's fast protocol, which reads mu / sigma^2 from the operand shadows
            # only: vbnn_dw_args.mu_s / var_s are REQUIRED for them, include/vbnn_hip.h)
            km = lib.vbnn_ctx_kmajor_supported_dw(self.ctx.h, v.I, v.O, N, 1 if v.bias_from_dw else 0) if (
                km_ok and self.mode == "lrt" and self.fuse_kl and int(self.S) == 1 and self.kl_from_shadows) else 0
            v.dw_km = km > 0
            v.x_pad256 = km == 2         # the split launch of gemm_v3 reads x / x.x in whole 256-column tiles
            # the two-launch form of accGradParameters (early d/dlvars message) needs either the transposed operands or the
            # plain K-major launch of the two-pass kernel; the few-tile K-major launches compute both GEMMs in one grid
            v.early_ok = (not v.dw_km) or bool(not v.bias_from_dw and lib.vbnn_ctx_kmajor_supported(self.ctx.h, v.I, v.O, N))
            v.dx_km = bool(km_ok and li > 0 and lib.vbnn_ctx_kmajor_supported(self.ctx.h, v.I, N, v.O))
            use_muT = (li > 0) and not v.dx_km
            if use_muT and not getattr(v, "use_muT", True):
                need_prepare = True                            # the shadows' transposes were being skipped: refresh them
            v.use_muT = use_muT
        for v in self.vb:
            # Layers whose parameter-gradient GEMM is not the whole-tile 256 x 256 kernel anyway (I % 256 != 0: the
            # 784-wide input layer) get their bias gradient from that GEMM: x^T carries one more row, all ones, whose
            # output row is sum_n g[n][o] (vbnn_dw_args.gradBias). Written once here: the packers only touch rows < I.
            # K-major the row of ones is COLUMN I of x (in the K padding of the forward operand, where the packed
            # weights are zero, so the forward does not see it).
            extra = 1 if v.bias_from_dw else 0
            if self.f32_direct:
                v.x_s = _Packed(N, v.I, tdt, dev)        # (layer 1: used only when the raw minibatch cannot be read in place)
                v.x2_s = None
                v.xT_s = v.x2T_s = v.gT_s = v.gvT_s = None
                v.mixed_g = False
                v.g_s, v.gv_s = _Packed(N, v.O, tdt, dev), _Packed(N, v.O, tdt, dev)
                v.r = torch.zeros(N, v.O, dtype=tdt, device=dev)
                continue
            xcols = v.I + (extra if v.dw_km else 0)
            if v.x_pad256:
                xcols = (xcols + 255) // 256 * 256
            v.x_s = _Packed(N, xcols, tdt, dev)
            v.x2_s = _Packed(N, xcols, tdt, dev)
            if v.dw_km and v.bias_from_dw:
                v.x_s.t[:, v.I] = 1.0
            # opt.mixed_g (A/B, off): the last VB layer below the fused head gets g, gv from vbnn_head_backward TRANSPOSED as
            # well, and accGradParameters reads x K-major with g K-contiguous. Measured in the step (r02): the GEMM 220 ->
            # 216 us, the head's backward 32 -> 40 us for the extra stores: a net loss, so it stays off.
            v.mixed_g = bool(v.dw_km and v is self.vb[-1] and self.n_classes <= 16 and self.criterion == "nll" and
                             self.opt.get("mixed_g", False) and lib.vbnn_ctx_kmajor_supported(self.ctx.h, v.I, v.O, N) and not v.bias_from_dw)
            if v.dw_km:
                v.xT_s = v.x2T_s = v.gT_s = v.gvT_s = None
                if v.mixed_g:
                    v.gT_s, v.gvT_s = _Packed(v.O, N, tdt, dev), _Packed(v.O, N, tdt, dev)
            else:
                v.xT_s, v.x2T_s = _Packed(v.I + extra, N, tdt, dev), _Packed(v.I + extra, N, tdt, dev)
                v.gT_s, v.gvT_s = _Packed(v.O, N, tdt, dev), _Packed(v.O, N, tdt, dev)
                if v.bias_from_dw:
                    v.xT_s.t[v.I, :N] = 1.0
            v.g_s, v.gv_s = _Packed(N, v.O, tdt, dev), _Packed(N, v.O, tdt, dev)
            v.r = torch.zeros(N, v.O, dtype=tdt, device=dev)          # z / (2 sqrt(v)), stored in the operand type
        H = self.sizes[-1]
        self.h_s = _Packed(N, H, tdt, dev)            # input of the final Linear
        self.hT_s = _Packed(H, N, tdt, dev)
        self.logits = torch.zeros(N, self.n_classes, dtype=torch.float32, device=dev)
        self.g_logits = torch.zeros(N, self.n_classes, dtype=torch.float32, device=dev)
        self.out = torch.zeros(N, self.n_classes, dtype=torch.float32, device=dev)
        self.g3_s = _Packed(N, self.n_classes, tdt, dev)
        self.g3T_s = _Packed(self.n_classes, N, tdt, dev)
        # the fused head's logits from the last VB layer's forward tiles (vbnn_fwd_args.head_slots): where the library's forward
        # of that layer can carry them, the head never re-reads h. opt.head_from_h = True keeps the head's own pass over h (A/B).
        vl = self.vb[-1]
        self.n_head_slots, self.head_slots = 0, None
        if self.n_classes <= 16 and self.criterion == "nll" and not self.device_draw and not self.opt.get("head_from_h", False):
            ns = lib.vbnn_forward_head_slots(self.ctx.h, self.code, N, vl.I, vl.O, self.n_classes)
            if ns > 0:
                self.n_head_slots, self.head_slots = ns, torch.zeros(ns * N * 16, dtype=torch.float32, device=dev)
        if need_prepare:
            self.prepare()

    # ---- mlp.lua:62-67. Nothing is zeroed: the first accGradParameters of the minibatch overwrites the gradients
    # and the first criterion pass overwrites the loss / hit accumulators.
    def resetGradients(self):
        self._first = True
        self._draws_run = 0          # sequential draws since the reset: the exchange leaves with the LAST one (run)
        self._exchange_done = False

    # ---- once per minibatch, after the parameters changed: VBLinear:compute_prior (VBLinear.lua:77-88)
    # fused with the packing of the GEMM shadows.
    @_ordered
    def prepare(self):
        lib = L.lib()
        if self._params_stale:
            # after a sharded update: every rank already holds what this sweep would produce -- the gathered operand shadows, the
            # combined prior statistics, the packed final weight (_update_sharded; "update leaves what prepare would" is a test)
            # -- while the fp32 rows of other ranks are stale. No sweep, and NO collective hidden in here: only the
            # transposed shadows of a layer that did not keep them so far are rebuilt, locally, from the gathered ones.
            for v in self.vb:
                if v.muT_s is not None and getattr(v, "use_muT", True):
                    L.check(lib.vbnn_transpose_packed(self.ctx.h, self.code, v.mu_s.ptr, v.mu_s.ld, v.O, v.I, v.muT_s.ptr, v.muT_s.ld))
                    L.check(lib.vbnn_transpose_packed(self.ctx.h, self.code, v.var_s.ptr, v.var_s.ld, v.O, v.I, v.varT_s.ptr, v.varT_s.ld))
            return
        if self.mode == "lrt":                                # one call: a sweep per layer + ONE finish kernel
            descs = (L.PrepDesc * len(self.vb))()
            for k, v in enumerate(self.vb):
                descs[k] = L.PrepDesc(means=_p(v.means), lvars=_p(v.lvars), O=v.O, I=v.I, mu_s=v.mu_s.ptr, var_s=v.var_s.ptr,
                                      ld_w=v.mu_s.ld, muT_s=v.muT_s.ptr if (v.muT_s and getattr(v, "use_muT", True)) else None,
                                      varT_s=v.varT_s.ptr if (v.varT_s and getattr(v, "use_muT", True)) else None,
                                      ld_wT=v.muT_s.ld if v.muT_s else 0,
                                      stats=_p(v.stats))
            w3 = L.PackDesc(src=_p(self.weight3), rows=self.n_classes, cols=self.sizes[-1], ld_src=self.sizes[-1],
                            dst=self.w3_s.ptr, ld_dst=self.w3_s.ld, dstT=self.w3T_s.ptr, ld_dstT=self.w3T_s.ld)
            L.check(lib.vbnn_prepare(self.ctx.h, self.code, len(self.vb), descs, C.byref(w3)))
            return
        for v in self.vb:
            L.check(lib.vbnn_compute_prior(self.ctx.h, _p(v.means), _p(v.lvars), v.O * v.I, None, None, None, _p(v.stats)))
        L.check(lib.vbnn_pack(self.ctx.h, self.code, L.PACK_COPY, _p(self.weight3), None, self.sizes[-1],
                              self.n_classes, self.sizes[-1], self.w3_s.ptr, self.w3_s.ld, self.w3T_s.ptr, self.w3T_s.ld))

    def _lrt(self):
        return self.mode == "lrt" and not self._map

    # ---- measurement hook (bench.py): when `self.probe` is (layer index, dict), every launch of that layer's three
    # GEMM families is bracketed by HIP events on the launch stream; bench.py reads the elapsed times after the
    # timed region. Off (None) it costs nothing.
    probe = None

    def _probed(self, name, li):
        import contextlib
        if self.probe is None or (self.probe[0] != li and self.probe[0] != "all"):
            return contextlib.nullcontext()
        if self.probe[0] == "all":                       # lab (tools/overlap_standin.py): every layer's launches, keyed by layer
            name = f"{name}[{li}]"
        eng = self

        class _Bracket:
            def __enter__(self_inner):
                self_inner.e0 = torch.cuda.Event(enable_timing=True)
                self_inner.e1 = torch.cuda.Event(enable_timing=True)
                self_inner.e0.record()

            def __exit__(self_inner, *exc):
                self_inner.e1.record()
                eng.probe[1].setdefault(name, []).append((self_inner.e0, self_inner.e1))
                return False
        return _Bracket()

    def _probed_part(self, name, li):
        """The two launches of an early-message accGradParameters separately: only in the lab's all-layers probe."""
        if self.probe is None or self.probe[0] != "all" or not self._early(self.vb[li], self._lrt()):
            return _NULL_CM
        return self._probed(name, li)

    # ---- VBLinear:clamp_to_map on every VB layer (mlp.lua:88-91): the forward uses the means as weights.
    @_ordered
    def clamp_to_map(self):
        self._map = True
        if self.mode == "wn":                 # the weight shadows currently hold a sampled draw: repack the means
            lib = L.lib()
            for v in self.vb:
                L.check(lib.vbnn_pack(self.ctx.h, self.code, L.PACK_COPY, _p(v.means), None, v.I, v.O, v.I,
                                      v.mu_s.ptr, v.mu_s.ld, v.muT_s.ptr if v.muT_s else None,
                                      v.muT_s.ld if v.muT_s else 0))

    # ---- mlp:test (mlp.lua:86-107): `quicktest` = one MAP pass, otherwise the average over testSamples draws.
    # The reference's `run` also back-propagates at test time and throws the gradients away; here the test pass
    # is forward + criterion only. Returns (mean error, accuracy in percent) of THIS rank's rows.
    def test(self, inputs, targets):
        N = inputs.shape[0]
        self.prepare()
        draws = 1 if self.opt.get("quicktest") else int(self.opt["testSamples"])
        err = acc = 0.0
        for _ in range(draws):
            self._first = True                      # this draw's criterion pass overwrites the accumulators
            if self.opt.get("quicktest"):
                self.clamp_to_map()
            else:
                self.sample()
            self.run(inputs, targets, backward=False)
            loss, correct = self.loss_and_accuracy()
            err += loss * self.world            # the criterion divides by the global batch; this is the local mean
            acc += 100.0 * correct / N
        return err / draws, acc / draws

    # ---- mlp.lua:69-74
    @_ordered
    def sample(self, by=1):
        self.draw += by
        self._map = False
        if self.device_draw:
            L.check(L.lib().vbnn_sample(self.ctx.h, _p(self._draw_dev), by))
        if self.mode == "wn":
            lib = L.lib()
            for v in self.vb:
                L.check(lib.vbnn_wn_sample(self.ctx.h, _p(v.means), None, _p(v.lvars), _p(v.weight), None, v.O, v.I,
                                           self.seed, v.layer_id, self.draw))
                L.check(lib.vbnn_pack(self.ctx.h, self.code, L.PACK_COPY, _p(v.weight), None, v.I, v.O, v.I,
                                      v.mu_s.ptr, v.mu_s.ld, v.muT_s.ptr if v.muT_s else None,
                                      v.muT_s.ld if v.muT_s else 0))

    # ---- final Linear + criterion through the generic GEMM kernels (class counts above 16)
    def _generic_head(self, N, targets, inv_n, accumulate, backward=True):
        lib, ctx, code = L.lib(), self.ctx.h, self.code
        lrt = self._lrt()
        H, Cn = self.sizes[-1], self.n_classes
        a = L.FwdArgs(w=self.w3_s.ptr, w2=None, x=self.h_s.ptr, x2=None, ld_w=self.w3_s.ld, ld_x=self.h_s.ld,
                      N=N, I=H, O=Cn, bias=_p(self.bias3), y=_p(self.logits), ld_y=Cn)
        L.check(lib.vbnn_forward(ctx, code, C.byref(a)))
        if self.criterion == "mse":
            assert targets.dtype == torch.float32 and tuple(targets.shape) == (N, Cn) and targets.is_contiguous()
            L.check(lib.vbnn_mse_forward(ctx, _p(self.logits), Cn, _p(targets), Cn, N, Cn, inv_n / Cn, _p(self.g_logits), Cn,
                                         accumulate, _p(self._acc)))
            if not accumulate:
                L.check(lib.vbnn_buf_zero(ctx, _p(self._corr), 4))
        else:
            if not accumulate:                          # vbnn_logsoftmax_nll adds into its accumulators
                L.check(lib.vbnn_buf_zero(ctx, _p(self._acc), 16))
                L.check(lib.vbnn_buf_zero(ctx, _p(self._corr), 4))
            L.check(lib.vbnn_logsoftmax_nll(ctx, _p(self.logits), Cn, _p(targets), N, Cn, inv_n, _p(self.out),
                                            _p(self.g_logits), _p(self._acc), _p(self._corr)))
        if not backward:
            return
        L.check(lib.vbnn_pack(ctx, code, L.PACK_COPY, _p(self.g_logits), None, Cn, N, Cn, self.g3_s.ptr, self.g3_s.ld,
                              self.g3T_s.ptr, self.g3T_s.ld))
        d = L.DwArgs(xT=self.hT_s.ptr, x2T=None, gT=self.g3T_s.ptr, gvT=None, ld_n=self.hT_s.ld, N=N, I=H, O=Cn,
                     scale=1.0, accumulate=accumulate, gradWeight=_p(self.gradWeight3), gradSum=None)
        L.check(lib.vbnn_acc_grad_parameters(ctx, code, C.byref(d)))
        L.check(lib.vbnn_acc_grad_bias(ctx, L.F32, _p(self.g_logits), Cn, N, Cn, 1.0, accumulate, _p(self.gradBias3)))
        vl = self.vb[-1]
        dx = L.DxArgs(wT=self.w3T_s.ptr, w2T=None, g=self.g3_s.ptr, gv=None, ld_wT=self.w3T_s.ld, ld_g=self.g3_s.ld,
                      N=N, I=H, O=Cn, x=self.h_s.ptr, ld_x=self.h_s.ld, gx=None, ld_gx=0, relu_mask=1,
                      r_prev=_p(vl.r) if lrt else None, ld_r_prev=vl.O, r_prev_packed=1, g_prev=vl.g_s.ptr,
                      gv_prev=vl.gv_s.ptr if lrt else None, ld_gp=vl.g_s.ld, gT_prev=vl.gT_s.ptr if vl.gT_s else None,
                      gvT_prev=vl.gvT_s.ptr if (lrt and vl.gvT_s) else None, ld_gpT=vl.gT_s.ld if vl.gT_s else 0)
        L.check(lib.vbnn_grad_input(ctx, code, C.byref(dx)))

    # ---- argument blocks of the three GEMM families for VB layer `li` (also used by bench.py to time
    # exactly the launches of the step in isolation)
    # The blocks are built once per (layer, batch shape, ...) and kept (the launch-bound step is ~50 us of GPU time: rebuilding
    # five ctypes structures per step cost the Python host ~20 us of it); what varies from call to call -- the draw counter, the
    # raw minibatch's address -- is patched in.
    def _fwd_args(self, li, N, row0):
        direct_x = li == 0 and self._x_in is not None
        key = ("fwd", li, N, row0, self._rpd, direct_x, self._lrt())     # (the block bakes in the LRT / MAP operand set)
        a = self._argcache.get(key)
        if a is None:
            a = self._argcache[key] = self._build_fwd_args(li, N, row0)
        if direct_x:
            a.x, a.ld_x = self._x_in.ptr, self._x_in.ld
        a.draw = 0 if self.device_draw else self.draw
        return a

    def _build_fwd_args(self, li, N, row0):
        v, lrt = self.vb[li], self._lrt()
        last = li == len(self.vb) - 1
        nxt = None if last else self.vb[li + 1]
        xin = self._x_in if (li == 0 and self._x_in is not None) else v.x_s       # the raw minibatch, read in place (fp32)
        sq = lrt and not self.f32_direct                                          # squares as operands (else formed by the kernel)
        return L.FwdArgs(w=v.mu_s.ptr, w2=v.var_s.ptr if lrt else None, x=xin.ptr, x2=v.x2_s.ptr if sq else None,
                         ld_w=v.mu_s.ld, ld_x=xin.ld, N=N, I=v.I, O=v.O, bias=_p(v.bias), seed=self.seed,
                         layer=v.layer_id, draw=0 if self.device_draw else self.draw,
                         draw_dev=_p(self._draw_dev) if self.device_draw else None, row0=row0, y=None, ld_y=0,
                         r=_p(v.r) if lrt else None, ld_r=v.O, r_packed=1, relu=1,
                         h=self.h_s.ptr if last else nxt.x_s.ptr,
                         h2=None if (last or not sq) else nxt.x2_s.ptr,
                         ld_h=self.h_s.ld if last else nxt.x_s.ld,
                         hT=(None if (self.n_classes <= 16 and self.criterion == "nll") else self.hT_s.ptr) if last else (nxt.xT_s.ptr if nxt.xT_s else None),
                         h2T=None if (last or not lrt or not nxt.x2T_s) else nxt.x2T_s.ptr,
                         ld_hT=self.hT_s.ld if last else (nxt.xT_s.ld if nxt.xT_s else 0), rows_per_draw=self._rpd,
                         **(dict(head_w3=self.w3_s.ptr, head_ld_w=self.w3_s.ld, head_C=self.n_classes, head_slots=_p(self.head_slots))
                            if (last and self._use_head_slots()) else {}))

    def _use_head_slots(self):
        return self.n_head_slots > 0 and not self._rpd

    def _dw_args(self, li, N, accumulate):
        direct_x = li == 0 and self._x_in is not None
        key = ("dw", li, N, accumulate, float(self._draws or self.S), direct_x, self._lrt())
        d = self._argcache.get(key)
        if d is None:
            d = self._argcache[key] = self._build_dw_args(li, N, accumulate)
        if direct_x:
            d.x, d.ld_x = self._x_in.ptr, self._x_in.ld
        d.draw, d.part = self.draw, 0
        return d

    def _build_dw_args(self, li, N, accumulate):
        v, lrt = self.vb[li], self._lrt()
        has_t = v.xT_s is not None
        has_gt = v.gT_s is not None
        xin = self._x_in if (li == 0 and self._x_in is not None) else v.x_s
        d = L.DwArgs(xT=v.xT_s.ptr if has_t else None, x2T=v.x2T_s.ptr if (lrt and has_t) else None,
                     gT=v.gT_s.ptr if has_gt else None, gvT=v.gvT_s.ptr if (lrt and has_gt) else None,
                     ld_n=v.gT_s.ld if has_gt else 0, N=N, I=v.I, O=v.O, scale=1.0,
                     accumulate=accumulate, seed=self.seed, layer=v.layer_id, draw=self.draw, lvars=_p(v.lvars),
                     x=xin.ptr, x2=v.x2_s.ptr if (lrt and v.x2_s is not None) else None, g=v.g_s.ptr, gv=v.gv_s.ptr if lrt else None,
                     ld_x=xin.ld, ld_g=v.g_s.ld)
        if self.fuse_kl:
            d.gradWeight, d.gradSum = None, None
            d.grad_mu, d.grad_lv = _p(v.gradWeight), _p(v.gradSum)
            d.means, d.stats = _p(v.means), _p(v.stats)
            d.B, d.S, d.kl_scale = self.B, float(self._draws or self.S), partition.scales(1, self.world)["kl_scale"]
            if self.kl_in_update:
                d.kl_scale = 0.0                      # likelihood parts only: update() adds the KL gradient from the fp32 parameters
            if self.kl_from_shadows and lrt:
                # the epilogue reads mu, sigma^2 from the bf16 operand shadows (4 B per weight instead of 8, no exp): the
                # values the forward GEMMs multiplied by
                d.mu_s, d.var_s, d.ld_w = v.mu_s.ptr, v.var_s.ptr, v.mu_s.ld
        else:
            d.gradWeight, d.gradSum = _p(v.gradWeight), _p(v.gradSum)
        d.gradBias = _p(v.gradBias) if v.bias_from_dw else None
        return d

    def _dx_args(self, li, N):
        key = ("dx", li, N, self._lrt())
        a = self._argcache.get(key)
        if a is None:
            a = self._argcache[key] = self._build_dx_args(li, N)
        return a

    def _build_dx_args(self, li, N):
        v, p, lrt = self.vb[li], self.vb[li - 1], self._lrt()
        use_t = v.use_muT
        return L.DxArgs(wT=v.muT_s.ptr if use_t else None, w2T=v.varT_s.ptr if (lrt and use_t) else None, g=v.g_s.ptr,
                        gv=v.gv_s.ptr if lrt else None, ld_wT=v.muT_s.ld, ld_g=v.g_s.ld, N=N, I=v.I, O=v.O,
                        x=v.x_s.ptr, ld_x=v.x_s.ld, gx=None, ld_gx=0, relu_mask=1,
                        r_prev=_p(p.r) if lrt else None, ld_r_prev=p.O, r_prev_packed=1, g_prev=p.g_s.ptr,
                        gv_prev=p.gv_s.ptr if lrt else None, ld_gp=p.g_s.ld,
                        gT_prev=p.gT_s.ptr if p.gT_s else None, gvT_prev=p.gvT_s.ptr if (lrt and p.gvT_s) else None,
                        ld_gpT=p.gT_s.ld if p.gT_s else 0,
                        w=v.mu_s.ptr, w2=v.var_s.ptr if lrt else None, ld_w=v.mu_s.ld)

    # ---- mlp.lua:76-84, fused
    @_ordered
    def run(self, inputs, targets, row0=None, backward=True, last_draw=None):
        """last_draw (data-parallel only): whether this run's gradients are the minibatch's final ones, i.e. whether the
        buckets' all-reduces are issued. Default: the opt.S-th sequential run since resetGradients() (always, inside
        run_draws). The draws of a minibatch ACCUMULATE in the arena (main.lua:32-37): exchanging after every draw would
        all-reduce sums that already contain other ranks' contributions, and the exchange stream's in-place write would
        race with the next draw's accumulate epilogue -- so earlier draws only accumulate."""
        lib, ctx, code = L.lib(), self.ctx.h, self.code
        if backward:
            self._draws_run += 1
            if last_draw is None:
                last_draw = bool(self._rpd) or self._draws_run >= int(self.S)
            if self.reduce and not self._rpd and self._draws_run > int(self.S) and self._exchange_done:
                raise RuntimeError(f"run() called {self._draws_run} times since resetGradients() with opt.S = {int(self.S)}: the "
                                   "gradient buckets were already exchanged after draw S")
            self._exchange_now = bool(last_draw)
            self._exchange_done = self._exchange_now
        x = inputs.reshape(inputs.shape[0], -1)                         # nn.Reshape (mlp.lua:12)
        # stacked draws (run_draws): the operands have S x rows_per_draw rows, the input and the targets one minibatch's --
        # the packer and the head address row n % rows_per_draw (no S-fold copy of the input is ever made)
        N = x.shape[0] * (self._draws if self._rpd else 1)
        assert x.shape[1] == self.sizes[0] and x.dtype == torch.float32 and x.is_cuda
        assert not self._rpd or x.shape[0] == self._rpd
        self._alloc_batch(N)
        if row0 is None:
            row0 = self.rank * (self._rpd or N)
        lrt = self._lrt()
        accumulate = 0 if self._first else 1
        inv_n = partition.scales(self._rpd or N, self.world)["inv_n"]       # per DRAW: stacked draws sum their criteria
        v0 = self.vb[0]
        self._x_in = None
        if self.f32_direct and not self._rpd and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0:
            self._x_in = _View(x)                        # the GEMMs read the minibatch where it lies: no packing launch
        else:
            L.check(lib.vbnn_pack_input(ctx, code, _p(x), x.stride(0), N, v0.I, v0.x_s.ptr,
                                        v0.x2_s.ptr if (lrt and v0.x2_s is not None) else None,
                                        v0.x_s.ld, v0.xT_s.ptr if v0.xT_s else None,
                                        v0.x2T_s.ptr if (lrt and v0.x2T_s) else None, v0.xT_s.ld if v0.xT_s else 0, self._rpd))
        nl = len(self.vb)
        # ---------------- forward
        for li in range(nl):
            a = self._fwd_args(li, N, row0)
            with self._probed("forward", li):
                L.check(lib.vbnn_forward(ctx, code, C.byref(a)))
        H, Cn = self.sizes[-1], self.n_classes
        vl = self.vb[-1]
        fused_head = Cn <= 16 and self.criterion == "nll"
        if fused_head:
            # ---------------- fused classifier head (mlp.lua:29-32): streaming kernels, no 10-wide MFMA tiles
            if backward and self.head_step:
                # forward + backward of the head as ONE call (vbnn_head_forward_backward: one launch for fp32 at launch-bound
                # sizes -- the logits recomputed per workgroup instead of a second kernel waiting for g_logits --, the two
                # launches below everywhere else)
                key = ("head", N, self._rpd, accumulate, lrt)
                a = self._argcache.get(key)
                if a is None:
                    a = self._argcache[key] = L.HeadArgs()
                    a.h, a.ld_h, a.w3, a.ld_w, a.bias = self.h_s.ptr, self.h_s.ld, self.w3_s.ptr, self.w3_s.ld, _p(self.bias3)
                    a.N, a.H, a.C, a.rows_per_draw, a.inv_n, a.accumulate = N, H, Cn, self._rpd, inv_n, accumulate
                    a.logits, a.out, a.g_logits = _p(self.logits), _p(self.out), _p(self.g_logits)
                    a.loss_sum_dev, a.correct_dev = _p(self._acc), _p(self._corr)
                    a.gradWeight, a.gradBias, a.gradBias_prev = _p(self.gradWeight3), _p(self.gradBias3), _p(vl.gradBias)
                    a.relu_mask, a.r_prev_packed, a.r_prev, a.ld_r_prev = 1, 1, (_p(vl.r) if lrt else None), vl.O
                    a.g_prev, a.gv_prev, a.ld_gp = vl.g_s.ptr, (vl.gv_s.ptr if lrt else None), vl.g_s.ld
                    a.gT_prev = vl.gT_s.ptr if vl.gT_s else None
                    a.gvT_prev = vl.gvT_s.ptr if (lrt and vl.gvT_s) else None
                    a.ld_gpT = vl.gT_s.ld if vl.gT_s else 0
                    if self._use_head_slots():            # the logits' partial sums are in the slots the forward just wrote
                        a.logit_slots, a.n_slots = _p(self.head_slots), self.n_head_slots
                a.target = _p(targets)
                L.check(lib.vbnn_head_forward_backward(ctx, code, C.byref(a)))
            else:
                if self._use_head_slots():
                    L.check(lib.vbnn_head_forward_slots(ctx, _p(self.head_slots), self.n_head_slots, _p(self.bias3), _p(targets), N, Cn,
                                                        inv_n, _p(self.logits), _p(self.out), _p(self.g_logits), accumulate,
                                                        _p(self._acc), _p(self._corr), self._rpd))
                else:
                    L.check(lib.vbnn_head_forward(ctx, code, self.h_s.ptr, self.h_s.ld, self.w3_s.ptr, self.w3_s.ld,
                                                  _p(self.bias3), _p(targets), N, H, Cn, inv_n, _p(self.logits), _p(self.out),
                                                  _p(self.g_logits), accumulate, _p(self._acc), _p(self._corr), self._rpd))
                if not backward:
                    self._first = False
                    return
                L.check(lib.vbnn_head_backward(ctx, code, self.h_s.ptr, self.h_s.ld, self.w3_s.ptr, self.w3_s.ld,
                                               _p(self.g_logits), N, H, Cn, accumulate, _p(self.gradWeight3),
                                               _p(self.gradBias3), _p(vl.gradBias), 1, _p(vl.r) if lrt else None, vl.O, 1,
                                               vl.g_s.ptr,
                                               vl.gv_s.ptr if lrt else None, vl.g_s.ld, vl.gT_s.ptr if vl.gT_s else None,
                                               vl.gvT_s.ptr if (lrt and vl.gvT_s) else None, vl.gT_s.ld if vl.gT_s else 0))
        else:
            if self._rpd:                                 # (the generic criteria take one target per operand row)
                targets = targets.repeat(self._draws) if targets.dim() == 1 else targets.repeat(self._draws, 1)
            self._generic_head(N, targets, inv_n, accumulate, backward)
            if not backward:
                self._first = False
                return
        # ---------------- backward: VB layers, last to first. accGradParameters (+ bias gradient + the bucket's
        # all-reduce) of layer li and updateGradInput of layer li are independent of each other (both consume
        # g_li): with `overlap` they run on two HIP streams, so the HBM-bound epilogue of one GEMM sits beside the
        # MFMA main loop of the other instead of every CU hitting its epilogue at the same moment.
        if self.f32_direct and not self.overlap:
            # fp32, launch-bound sizes: accGradParameters and updateGradInput of a layer are independent (both consume g_li) and
            # go out as ONE launch where the library can carry both (vbnn_backward_pair: the general kernel's K-major forms at
            # its 32 x 32 geometry; elsewhere the call is the two launches) -- each tile bitwise what its own launch computes
            for li in range(nl - 1, -1, -1):
                v = self.vb[li]
                d = self._dw_args(li, N, accumulate)
                with self._probed("accGradParameters", li):
                    if li > 0:
                        L.check(lib.vbnn_backward_pair(ctx, code, C.byref(self._dx_args(li, N)), C.byref(d)))
                    else:
                        L.check(lib.vbnn_acc_grad_parameters(ctx, code, C.byref(d)))
                if not (fused_head and li == nl - 1) and not v.bias_from_dw:
                    L.check(lib.vbnn_acc_grad_bias(ctx, code, v.g_s.ptr, v.g_s.ld, N, v.O, 1.0, accumulate, _p(v.gradBias)))
        elif self.dx_first and not self.reduce and not self.overlap:
            # every updateGradInput first, then the accGradParameters from the first layer up: in the wide net the two 4096^3
            # launches are then separated by the lighter 784-wide gradient. The chip is power-bound in these launches
            # (LAB_NOTES.md section 3, r02): alternating heavy and light ones lets it hold a higher clock -- measured -25 / -7 / +3 us on a
            # slow / middle / fast box of the pool; gradients are the same launches on the same operands, bit for bit.
            # Not with an exchange (the widest layer's buckets must leave first) and not for stacks of equal layers (no
            # lighter launch to put between: +0.5 % on the deep stack).
            for li in range(nl - 1, 0, -1):
                with self._probed("updateGradInput", li):
                    L.check(lib.vbnn_grad_input(ctx, code, C.byref(self._dx_args(li, N))))
            for li in range(nl):
                v = self.vb[li]
                d = self._dw_args(li, N, accumulate)
                with self._probed("accGradParameters", li):
                    L.check(lib.vbnn_acc_grad_parameters(ctx, code, C.byref(d)))
                if not (fused_head and li == nl - 1) and not v.bias_from_dw:
                    L.check(lib.vbnn_acc_grad_bias(ctx, code, v.g_s.ptr, v.g_s.ld, N, v.O, 1.0, accumulate, _p(v.gradBias)))
        elif not self.overlap:
            for li in range(nl - 1, -1, -1):
                v = self.vb[li]
                d = self._dw_args(li, N, accumulate)
                with self._probed("accGradParameters", li):
                    if self._early(v, lrt):
                        # two launches (vbnn_dw_args.part): the sigma^2 GEMM and d/dlvars first -- its exchange starts while the
                        # mu GEMM still runs
                        d.part = 2
                        with self._probed_part("accGradParameters.lv", li):
                            L.check(lib.vbnn_acc_grad_parameters(ctx, code, C.byref(d)))
                        if self.sharded:
                            self._scatter(li, ("lv",))
                        else:
                            self._reduce(v.msg_early)
                        d.part = 1
                    with self._probed_part("accGradParameters.mu", li):
                        L.check(lib.vbnn_acc_grad_parameters(ctx, code, C.byref(d)))
                # the fused head already summed the last layer's g columns; ones-row layers got theirs from the GEMM
                if not (fused_head and li == nl - 1) and not v.bias_from_dw:
                    L.check(lib.vbnn_acc_grad_bias(ctx, code, v.g_s.ptr, v.g_s.ld, N, v.O, 1.0, accumulate, _p(v.gradBias)))
                if self.sharded:
                    self._scatter(li, ("mu", "small") if self._early(v, lrt) else ("lv", "mu", "small"))
                else:
                    self._reduce(v.msg_late if self._early(v, lrt) else (self.bucket_tail if li == nl - 1 else v.bucket))
                if li > 0:
                    dx = self._dx_args(li, N)
                    with self._probed("updateGradInput", li):
                        L.check(lib.vbnn_grad_input(ctx, code, C.byref(dx)))
        else:
            main, side, ctx2 = torch.cuda.current_stream(self.device), self._side, self.ctx2.h
            for li in range(nl - 1, -1, -1):
                v = self.vb[li]
                ready = torch.cuda.Event()
                ready.record(main)                           # g_li is complete on the main stream
                side.wait_event(ready)
                d = self._dw_args(li, N, accumulate)
                L.check(lib.vbnn_acc_grad_parameters(ctx2, code, C.byref(d)))
                if not (fused_head and li == nl - 1) and not v.bias_from_dw:
                    L.check(lib.vbnn_acc_grad_bias(ctx2, code, v.g_s.ptr, v.g_s.ld, N, v.O, 1.0, accumulate, _p(v.gradBias)))
                if self.reduce:                              # the exchange orders itself behind the MAIN stream: bring the
                    filled = torch.cuda.Event()              # side stream's accGradParameters in front of it first
                    filled.record(side)
                    main.wait_event(filled)
                    self._reduce(self.bucket_tail if li == nl - 1 else v.bucket)
                if li > 0:
                    dx = self._dx_args(li, N)
                    L.check(lib.vbnn_grad_input(ctx, code, C.byref(dx)))
            done = torch.cuda.Event()
            done.record(side)
            main.wait_event(done)                            # the next launch on the main stream sees every gradient
        self._first = False

    # ---- main.lua:32-37's S draws as ONE pass: the minibatch stacked S times as rows (draw s = rows [s N, (s + 1) N)), each
    # row's noise addressed by (its draw, its minibatch row) -- vbnn_fwd_args.rows_per_draw -- so every draw sees bit for bit
    # the z of its own sequential launch; the backward GEMMs sum over all draws at once (K = S N), the criterion over all
    # stacked rows with the per-draw 1 / N. One launch per GEMM family instead of S; the gradients equal the S sequential
    # draws' up to fp32 summation order. LRT only (weight noise draws a different weight matrix per draw). The gain is for the
    # reference's own operating point -- a handful of rows, S = 30 (config.lua:11,32) -- which is launch-bound.
    def run_draws(self, inputs, targets, S):
        assert self.mode == "lrt" and not self._map and self._first, "run_draws: LRT, right after resetGradients()"
        x = inputs.reshape(inputs.shape[0], -1)
        N = x.shape[0]
        self.sample()                                    # the first draw's counter; row n uses draw + n // N
        self._rpd, self._draws = N, S
        try:
            self.run(x, targets)
        finally:
            self._rpd, self._draws = 0, None
        if S > 1:
            self.sample(S - 1)

    # ---- one step as ONE graph launch (include/vbnn_hip.h: vbnn_capture_*). `issue` is a callable that issues the step's
    # calls on this engine (resetGradients / sample / run or run_draws / finish) exactly as it would to run them; between
    # begin and end they are recorded instead. Needs opt.device_draw (the noise counter must not be a launch argument), a
    # stream of the engine's own (FusedMLP(..., stream=...): the NULL stream cannot be captured) and one un-captured call
    # of `issue` before (allocations and first-launch configuration cannot be recorded). The inputs are read from the
    # same device buffers at every replay: copy the next minibatch into them.
    def capture_step(self, issue):
        assert self.device_draw, "capture_step needs opt.device_draw = True"
        assert self.ctx.torch_stream is not None, "capture_step needs an engine stream: FusedMLP(opt, stream=torch.cuda.Stream())"
        assert not self.reduce, "capture_step: single-GPU steps (the exchange runs on the communicator's own stream)"
        lib = L.lib()
        draws0 = self.draw
        with torch.cuda.stream(self.ctx.torch_stream):
            L.check(lib.vbnn_capture_begin(self.ctx.h))
            h = C.c_void_p()
            try:
                issue()
            finally:
                st = lib.vbnn_capture_end(self.ctx.h, C.byref(h))
            L.check(st)
        draws = self.draw - draws0
        self.draw = draws0                               # nothing ran: the recorded vbnn_sample nodes advance the counter at replay
        return _StepGraph(self, h, draws)

    # ---- data-parallel exchange: sum all-reduce of one layer's gradient bucket over RCCL/xGMI, issued
    # right after that layer's accGradParameters so it overlaps the rest of backward. The criterion
    # already divides by the GLOBAL batch and the KL gradient carries 1/world, so the sum is the result.
    # opt.exchange_dtype = "bf16" (default "f32"): the bucket is rounded to bf16 into a half-size staging arena, THAT is
    # all-reduced (80 instead of 160 MB per step of the wide configuration), and finish() widens the sums
    # back into the fp32 arena. A different gradient (every rank's contribution rounded, RCCL sums in bf16): an option for
    # hosts whose step is exchange-bound, never the default, and the bench line says which one ran.
    skip_exchange = False      # bench.py: the step with its collective calls left out (what the exchange costs beyond the overlap)

    def _reduce(self, bucket):
        if not self.reduce or not self._exchange_now or self.skip_exchange:
            return
        if self.exchange_dtype != "bf16":
            self.exchange().allreduce(bucket)
            return
        if self._stage is None:
            self._stage = torch.zeros(self.grads.numel(), dtype=torch.bfloat16, device=self.device)
        off = bucket.storage_offset() - self.grads.storage_offset()
        stage = self._stage[off:off + bucket.numel()]
        L.check(L.lib().vbnn_cast_grads(self.ctx.h, 1, _p(bucket), _p(stage), bucket.numel()))
        self.exchange().allreduce(stage)
        self._staged.append((bucket, stage))

    def _scatter(self, li, what):
        """Sharded mode: layer li's messages (partition.sharded_plan): reduce-scatter of d/dlvars, of d/dmeans; the small all-reduce."""
        if not self._exchange_now or self.skip_exchange:
            return
        p, ex, W = self._plan[li], self.exchange(), self.world
        for key in what:
            if key == "small":
                s0, e0 = p["small"]
                ex.allreduce(self.grads[s0:e0])
            else:
                off, per = p[key]
                ex.reduce_scatter(self.grads[off:off + per * W], per)

    def gather_parameters(self):
        """Sharded mode: bring every rank's fp32 means / lvars rows to every rank (checkpoints, calc_lc, tests). A COLLECTIVE --
        every rank calls it at the same point of its program -- and never a hidden part of another call: prepare() / test() do
        not need it (the sharded update leaves the gathered shadows and combined statistics they would produce), calc_lc() and
        anything else that reads other ranks' fp32 rows RAISES while they are stale instead of gathering on one rank and
        deadlocking the others. On the device wherever a device transport exists: the RCCL exchange's own in-place
        all-gather (rank r's rows are its contribution), else the out-of-band group -- nccl: all_gather_into_tensor on the
        device; gloo (rehearsals, CPU tests): through the host."""
        if not self.sharded or self.world == 1:
            self._params_stale = False
            return
        import torch.distributed as dist
        from .comm import RcclExchange
        ex = self.exchange()
        self.finish()
        on_exchange = isinstance(ex, RcclExchange)
        backend = dist.get_backend(self.pg)
        if not on_exchange:
            torch.cuda.synchronize(self.device)
        for v in self.vb:
            r0, nr = partition.layer_row_shard(v.O, self.world, self.rank)
            for t in (v.means, v.lvars):
                assert t.is_contiguous()
                if on_exchange:
                    ex.all_gather(t)
                elif backend == "nccl":
                    mine = t[r0:r0 + nr].clone()
                    dist.all_gather_into_tensor(t.view(-1), mine.view(-1), group=self.pg)
                else:
                    mine = t[r0:r0 + nr].cpu()
                    pieces = [torch.empty_like(mine) for _ in range(self.world)]
                    dist.all_gather(pieces, mine, group=self.pg)
                    for r, pc in enumerate(pieces):
                        if r != self.rank:
                            t[r * nr:(r + 1) * nr].copy_(pc)
        if on_exchange:
            ex.finish()
        self._params_stale = False

    def _need_gathered_parameters(self, what):
        if self._params_stale:
            raise RuntimeError(f"{what}: the fp32 means / lvars of other ranks' rows are stale after a sharded update -- call "
                               "gather_parameters() on EVERY rank first (a collective; it is never issued implicitly)")

    def _update_sharded(self, opt):
        """mlp:update / VBLinear:update with the parameters SHARDED by layer rows: the reduce-scatter left this rank the summed
        likelihood gradients of its rows; vbnn_update runs on those rows alone (+ the exact KL gradient, kl_add), then the new
        operand shadows and the slices' prior statistics are all-gathered and every rank forms the layers' statistics from the
        same gathered parts (vbnn_stats_combine). Biases and the final Linear: all-reduced gradients, replicated SGD."""
        lib, h, W, R = L.lib(), self.ctx.h, self.world, self.rank
        ex = self.exchange()
        lr = float(opt["state"]["learningRate"])
        st = self.__dict__.setdefault("_opt_state", {})
        L.check(lib.vbnn_sgd_step(h, _p(self.weight3), _p(self.gradWeight3), self.weight3.numel(), lr))
        L.check(lib.vbnn_sgd_step(h, _p(self.bias3), _p(self.gradBias3), self.bias3.numel(), lr))
        n = len(self.vb)
        descs = (L.UpdateDesc * n)()
        mine = self._stat_parts[R]
        for k, v in enumerate(self.vb):
            L.check(lib.vbnn_sgd_step(h, _p(v.bias), _p(v.gradBias), v.O, lr))
            r0, nr = partition.layer_row_shard(v.O, W, R)
            mine[k].copy_(v.stats)                           # in: the WHOLE layer's pre-update statistics (var_hat of the KL terms)
            cfgs = []
            for key, cfg in (("mean", opt["meanState"]), ("var", opt["varState"])):
                s = st.setdefault((v.layer_id, key), {"t": 0})
                if "m" not in s:
                    s["m"] = torch.zeros(nr, v.I, dtype=torch.float32, device=self.device)
                    s["v"] = torch.zeros(nr, v.I, dtype=torch.float32, device=self.device)
                s["t"] += 1
                cfgs.append((s, L.AdamCfg(lr=float(cfg["learningRate"]), beta1=float(cfg.get("beta1", 0.9)),
                                          beta2=float(cfg.get("beta2", 0.999)), eps=float(cfg.get("epsilon", 1e-8)),
                                          lambda_=float(cfg.get("lambda", 1.0)), t=s["t"])))
            (sm, cm), (sv, cv) = cfgs
            descs[k] = L.UpdateDesc(means=_p(v.means[r0:r0 + nr]), lvars=_p(v.lvars[r0:r0 + nr]), O=nr, I=v.I,
                                    mu_s=_p(v.mu_s.t[r0:r0 + nr]), var_s=_p(v.var_s.t[r0:r0 + nr]), ld_w=v.mu_s.ld,
                                    muT_s=None, varT_s=None, ld_wT=0, stats=_p(mine[k]), grad_mu=_p(v.gradWeight[r0:r0 + nr]),
                                    grad_lv=_p(v.gradSum[r0:r0 + nr]), m_mu=_p(sm["m"]), v_mu=_p(sm["v"]), m_lv=_p(sv["m"]),
                                    v_lv=_p(sv["v"]), mu=cm, lv=cv, bias=None, grad_bias=None, lr_bias=lr, B=self.B,
                                    log14=None, kl_add=1.0)
        w3 = L.PackDesc(src=_p(self.weight3), rows=self.n_classes, cols=self.sizes[-1], ld_src=self.sizes[-1],
                        dst=self.w3_s.ptr, ld_dst=self.w3_s.ld, dstT=self.w3T_s.ptr, ld_dstT=self.w3T_s.ld)
        L.check(lib.vbnn_update(h, self.code, n, descs, C.byref(w3)))
        # what the next forward reads, from every rank: the shadows (in place: rank r's rows are its contribution), the parts
        for v in self.vb:
            ex.all_gather(v.mu_s.t)
            ex.all_gather(v.var_s.t)
        ex.all_gather(self._stat_parts)
        ex.finish()
        ptrs = (C.c_void_p * n)(*[v.stats.data_ptr() for v in self.vb])
        L.check(lib.vbnn_stats_combine(h, n, W, _p(self._stat_parts), ptrs))
        for v in self.vb:
            if v.muT_s is not None and getattr(v, "use_muT", True):       # transposed shadows: rebuilt locally from the gathered ones
                L.check(lib.vbnn_transpose_packed(h, self.code, v.mu_s.ptr, v.mu_s.ld, v.O, v.I, v.muT_s.ptr, v.muT_s.ld))
                L.check(lib.vbnn_transpose_packed(h, self.code, v.var_s.ptr, v.var_s.ld, v.O, v.I, v.varT_s.ptr, v.varT_s.ld))
        self._params_stale = W > 1

    def exchange(self):
        if self._exchange is None:
            from .comm import make_exchange
            self._exchange = make_exchange(self.ctx, self.rank, self.world, self.pg)
        return self._exchange

    def comm_backend(self):
        return self.exchange().backend if self.reduce else "none"

    @_ordered
    def finish(self):
        """Order the compute stream behind the outstanding all-reduces (end of the step)."""
        if self._exchange is not None:
            self._exchange.finish()
        for bucket, stage in self._staged:            # (bf16 exchange) the sums, widened back behind the exchange
            L.check(L.lib().vbnn_cast_grads(self.ctx.h, 0, _p(stage), _p(bucket), bucket.numel()))
        self._staged = []

    def check_exchange(self):
        """Raise if the exchange cannot vouch for the sums in the arena (the p2p exchange: a barrier that gave up -- ADVICE r03).
        Blocks on the exchange stream; update() and loss_and_accuracy() call it, a loop that does neither (a benchmark, a
        custom host) calls it before it believes a gradient or a timing."""
        if self._exchange is not None and hasattr(self._exchange, "check"):
            self._exchange.check()

    def _early(self, v, lrt=True):
        return v.msg_early is not None and lrt and self.fuse_kl and getattr(v, "early_ok", True)

    def buckets(self):
        """The all-reduce messages of one step, in issue order (last VB layer + final Linear first)."""
        out, nl = [], len(self.vb)
        for li in range(nl - 1, -1, -1):
            v = self.vb[li]
            if self._early(v):
                out += [v.msg_early, v.msg_late]
            else:
                out.append(self.bucket_tail if li == nl - 1 else v.bucket)
        return out

    @_ordered
    def time_buckets(self, reps=5):
        """Each bucket's all-reduce alone on an otherwise idle GPU: milliseconds (host clock around reps exchanges,
        device-synchronised), algorithm GB/s and bus GB/s (x 2 (world - 1) / world). Collective: every rank calls it."""
        import time
        out = []
        for b in self.buckets():
            if self.exchange_kind == "p2p":
                scratch = b                                # the p2p exchange sums regions of ITS arena: time it on the bucket itself
            else:
                scratch = torch.zeros_like(b) if self.exchange_dtype == "f32" else torch.zeros(b.numel(), dtype=torch.bfloat16, device=b.device)
            for _ in range(2):
                self.exchange().allreduce(scratch); self.finish()
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for _ in range(reps):
                self.exchange().allreduce(scratch)
            self.finish()
            torch.cuda.synchronize(self.device)
            ms = (time.perf_counter() - t0) / reps * 1e3
            nbytes = b.numel() * scratch.element_size()
            alg = nbytes / (ms * 1e-3) / 1e9
            out.append({"bytes": nbytes, "ms": round(ms, 4), "alg_GBps": round(alg, 1),
                        "bus_GBps": round(alg * 2 * (self.world - 1) / max(self.world, 1), 1)})
        return out

    def synthetic_targets(self, x, row0=0):
        """bench / tests: class targets uniform in 0..n_classes-1 by GLOBAL row (data.lua:16 convention, 0-based); for the
        regression criterion y* = x R / sqrt(I) with a fixed Philox-drawn R -- data
        preparation outside the timed step, so a plain torch matmul."""
        N = x.shape[0]
        if self.criterion == "mse":
            R = torch.empty(x.shape[1], self.n_classes, dtype=torch.float32, device=x.device)
            fill_normal(R, self.seed, L.STREAM_DATA, 1, 0)
            return (x @ R / math.sqrt(x.shape[1])).contiguous()
        return ((torch.arange(N, device=x.device, dtype=torch.int64) + row0) * 2654435761 % self.n_classes).to(torch.int32)

    # ---- mlp:update + VBLinear:update (mlp.lua:117-142, VBLinear.lua:124-166) on the device: the dW epilogue
    # already produced the TOTAL gradients (likelihood / S + KL), so the update is ONE sweep per layer (vbnn_update):
    # SGD on the bias, Adam on means and lvars, and -- from the new parameters, in the same pass -- the GEMM shadows
    # and prior statistics of the next minibatch (the reference's update calls compute_prior too, :130). Nothing is
    # downloaded. Needs fuse_kl (otherwise use vbnn_amd/mlp.py's module-level update). `log` = True also produces the
    # 14 series of VBLinear.lua:149-164 per layer (self.update_log, a device tensor [layers][14]).
    @_ordered
    def update(self, opt=None, log=False):
        if not self.fuse_kl:
            raise RuntimeError("FusedMLP.update needs opt.fuse_kl = True (total gradients from the dW epilogue)")
        opt = opt or self.opt
        self.finish()
        self.check_exchange()                  # BEFORE a parameter is touched: an incomplete exchange must not reach them
        if self.sharded:
            assert not log, "the 14 logged series need whole-layer norms: not with the sharded update"
            return self._update_sharded(opt)
        lib, h = L.lib(), self.ctx.h
        lr = float(opt["state"]["learningRate"])
        st = self.__dict__.setdefault("_opt_state", {})
        if log and getattr(self, "update_log", None) is None:
            self.update_log = torch.zeros(len(self.vb), 14, dtype=torch.float64, device=self.device)
        L.check(lib.vbnn_sgd_step(h, _p(self.weight3), _p(self.gradWeight3), self.weight3.numel(), lr))
        L.check(lib.vbnn_sgd_step(h, _p(self.bias3), _p(self.gradBias3), self.bias3.numel(), lr))
        if self.mode != "lrt":
            # weight-noise mode keeps no parameter shadows between minibatches (sample() packs the drawn weights):
            # per-tensor Adam, then the statistics
            for v in self.vb:
                L.check(lib.vbnn_sgd_step(h, _p(v.bias), _p(v.gradBias), v.O, lr))
                for key, x, g, cfg in (("mean", v.means, v.gradWeight, opt["meanState"]), ("var", v.lvars, v.gradSum, opt["varState"])):
                    s = st.setdefault((v.layer_id, key), {"t": 0})
                    if "m" not in s:
                        s["m"], s["v"] = torch.zeros_like(x), torch.zeros_like(x)
                    s["t"] += 1
                    L.check(lib.vbnn_adam_step(h, _p(x), _p(g), None, _p(s["m"]), _p(s["v"]), x.numel(),
                                               float(cfg["learningRate"]), float(cfg.get("beta1", 0.9)),
                                               float(cfg.get("beta2", 0.999)), float(cfg.get("epsilon", 1e-8)),
                                               float(cfg.get("lambda", 1.0)), s["t"], None))
            self.prepare()
            return
        descs = (L.UpdateDesc * len(self.vb))()
        for k, v in enumerate(self.vb):
            cfgs = []
            for key, x, cfg in (("mean", v.means, opt["meanState"]), ("var", v.lvars, opt["varState"])):
                s = st.setdefault((v.layer_id, key), {"t": 0})
                if "m" not in s:
                    s["m"], s["v"] = torch.zeros_like(x), torch.zeros_like(x)
                s["t"] += 1
                cfgs.append((s, L.AdamCfg(lr=float(cfg["learningRate"]), beta1=float(cfg.get("beta1", 0.9)),
                                          beta2=float(cfg.get("beta2", 0.999)), eps=float(cfg.get("epsilon", 1e-8)),
                                          lambda_=float(cfg.get("lambda", 1.0)), t=s["t"])))
            (sm, cm), (sv, cv) = cfgs
            use_t = v.muT_s is not None and getattr(v, "use_muT", True)
            descs[k] = L.UpdateDesc(means=_p(v.means), lvars=_p(v.lvars), O=v.O, I=v.I, mu_s=v.mu_s.ptr, var_s=v.var_s.ptr,
                                    ld_w=v.mu_s.ld, muT_s=v.muT_s.ptr if use_t else None, varT_s=v.varT_s.ptr if use_t else None,
                                    ld_wT=v.muT_s.ld if v.muT_s else 0, stats=_p(v.stats), grad_mu=_p(v.gradWeight),
                                    grad_lv=_p(v.gradSum), m_mu=_p(sm["m"]), v_mu=_p(sm["v"]), m_lv=_p(sv["m"]), v_lv=_p(sv["v"]),
                                    mu=cm, lv=cv, bias=_p(v.bias), grad_bias=_p(v.gradBias), lr_bias=lr, B=self.B,
                                    log14=C.c_void_p(self.update_log[k].data_ptr()) if log else None,
                                    kl_add=1.0 if self.kl_in_update else 0.0)
        w3 = L.PackDesc(src=_p(self.weight3), rows=self.n_classes, cols=self.sizes[-1], ld_src=self.sizes[-1],
                        dst=self.w3_s.ptr, ld_dst=self.w3_s.ld, dstT=self.w3T_s.ptr, ld_dstT=self.w3T_s.ld)
        L.check(lib.vbnn_update(h, self.code, len(self.vb), descs, C.byref(w3)))

    # ---- reporting (each of these synchronises)
    def loss_and_accuracy(self):
        self.finish()
        torch.cuda.synchronize(self.device)
        self.check_exchange()
        loss = float(self._acc[0].item())
        correct = int(self._corr[0].item())
        return loss, correct

    @_ordered
    def calc_lc(self, opt=None):                                         # mlp.lua:109-115, fresh statistics
        self._need_gathered_parameters("calc_lc")
        lc = 0.0
        B = float((opt or self.opt).get("B", self.B))
        for v in self.vb:
            L.check(L.lib().vbnn_calc_lc(self.ctx.h, _p(v.means), _p(v.lvars), None, None, _p(v.stats), B, None,
                                         _p(self._lc), v.O * v.I))
            lc += float(self._lc[0].item())
        return lc
