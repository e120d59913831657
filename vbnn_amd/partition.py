"""Host logic of the data-parallel recipe (no GPU, no torch needed): which rows a rank owns, how the criterion and the
KL gradient are scaled so that a plain SUM over ranks IS the global gradient, and how the gradient arena is laid out
and cut into all-reduce messages. vbnn_amd/engine.py computes with exactly these functions; tests/test_dist_cpu.py
drives them with two gloo ranks on the CPU.

The reference is single-device (main.lua:142 only sets BLAS threads); the rule it fixes is the objective:
    (1 / N_global) sum_n NLL_n   (nn.ClassNLLCriterion, sizeAverage, mlp.lua:32)   +   KL / B   (VBLinear.lua:90-103)
so with G ranks of N_local rows each: every rank's criterion divides by N_local * G, and every rank adds KL-gradient / G.
"""


def shard_rows(n_global, world, rank):
    """Rank r owns global rows [r * N/G, (r + 1) * N/G): returns (row0, n_local). N must divide evenly (the bench's
    weak scaling gives every rank the same row count by construction)."""
    if n_global % world:
        raise ValueError(f"global batch {n_global} does not divide over {world} ranks")
    n_local = n_global // world
    return rank * n_local, n_local


def scales(n_local, world):
    """inv_n: what the criterion multiplies each row's loss / gradient by; kl_scale: the weight of the KL gradient in
    each rank's fused accGradParameters epilogue (vbnn_dw_args.kl_scale)."""
    return {"inv_n": 1.0 / (n_local * world), "kl_scale": 1.0 / world}


def arena_layout(sizes, n_classes, early_lv=None):
    """The flat fp32 gradient arena: per VB layer [d/dlvars (O x I) | d/dmeans (O x I) | d/dbias (O)], then the final
    Linear [gradWeight (C x H) | gradBias (C)].
    early_lv: per VB layer, True where accGradParameters runs as TWO launches (vbnn_dw_args.part = 2, then 1) so that the
    finished d/dlvars can leave while the d/dmeans GEMM still runs: that layer then has two messages, [d/dlvars] and
    [d/dmeans | d/dbias] (d/dlvars comes first in the arena so that both are contiguous).
    Returns (layers, final, total, buckets):
      layers[k] = dict(I, O, lv=(off, n), mu=(off, n), bias=(off, n), bucket=(start, end),
                       early=(start, end) or None, late=(start, end))   -- the layer's message(s)
      final     = dict(weight=(off, n), bias=(off, n), bucket=(start, end))
      buckets   = every all-reduce message of a step as (start, end), in ISSUE order: backward runs last layer first, and
                  the final Linear's gradients (adjacent in the arena, finished before the last VB layer's
                  accGradParameters is launched) ride in that layer's LAST message."""
    n_layers = len(sizes) - 1
    early_lv = list(early_lv) if early_lv is not None else [False] * n_layers
    assert len(early_lv) == n_layers
    off, layers = 0, []
    for i in range(n_layers):
        I, O = sizes[i], sizes[i + 1]
        start = off
        d = {"I": I, "O": O, "lv": (off, O * I)}
        off += O * I
        d["mu"] = (off, O * I)
        off += O * I
        d["bias"] = (off, O)
        off += O
        d["bucket"] = (start, off)
        d["early"] = (start, start + O * I) if early_lv[i] else None
        d["late"] = (start + O * I, off) if early_lv[i] else (start, off)
        layers.append(d)
    H = sizes[-1]
    start = off
    final = {"weight": (off, n_classes * H)}
    off += n_classes * H
    final["bias"] = (off, n_classes)
    off += n_classes
    final["bucket"] = (start, off)
    layers[-1]["late"] = (layers[-1]["late"][0], off)          # the final Linear rides in the last VB layer's last message
    buckets = []
    for k in range(n_layers - 1, -1, -1):
        if layers[k]["early"]:
            buckets.append(layers[k]["early"])
        buckets.append(layers[k]["late"])
    return layers, final, off, buckets


def exchange_step(arena, buckets, exchange):
    """Issue one step's all-reduces in order and complete them (what FusedMLP.run + finish do around the kernels)."""
    for s, e in buckets:
        exchange.allreduce(arena[s:e])
    exchange.finish()


# ---- the SHARDED-UPDATE exchange (an option beside the all-reduce) -------------------------------
# reduce-scatter the likelihood gradients by ROWS of each layer (rank r ends with the sums for output units
# [r O / G, (r + 1) O / G): a contiguous range of the O x I tensors), rank r runs the update on those rows alone (Adam state and
# fp32 master parameters sharded), all-gather the packed operand shadows (bf16 mu, sigma^2: 4 B per weight instead of the 8 B of
# fp32 gradients an all-reduce's second half moves) + four doubles of prior statistics per layer. Biases and the final Linear
# (a few KB) keep a plain all-reduce and a replicated update.
def layer_row_shard(O, world, rank):
    """Rows of an O x I parameter tensor owned by `rank`: (row0, n_rows). O must divide evenly."""
    if O % world:
        raise ValueError(f"{O} output units do not divide over {world} ranks (sharded-update exchange)")
    n = O // world
    return rank * n, n


def sharded_plan(layers, world):
    """Per VB layer of arena_layout's `layers`: the two reduce-scatter regions (offset, floats PER RANK) -- d/dlvars and d/dmeans,
    each world x per_rank floats long -- and the small all-reduce message (start, end): the bias gradient, plus the final Linear's
    gradients behind the last layer's. Issue order is the all-reduce's: last layer first, d/dlvars before d/dmeans."""
    plan = []
    for d in layers:
        layer_row_shard(d["O"], world, 0)
        per = d["O"] * d["I"] // world
        plan.append({"lv": (d["lv"][0], per), "mu": (d["mu"][0], per), "small": (d["bias"][0], d["late"][1]), "rows": d["O"] // world})
    return plan


def sharded_exchange_grads(arena, plan, world, exchange):
    """One step's gradient exchange in sharded mode, in issue order (what FusedMLP.run issues around the kernels)."""
    for p in reversed(plan):
        for key in ("lv", "mu"):
            off, per = p[key]
            exchange.reduce_scatter(arena[off:off + per * world], per)
        s, e = p["small"]
        exchange.allreduce(arena[s:e])
    exchange.finish()


def exchange_bytes(layers, final, world, shadow_bytes=2):
    """Bytes each rank SENDS per step: (all-reduce of the whole arena, sharded-update exchange) -- ring / direct accounting alike
    move (G - 1) / G of a buffer per phase. All-reduce: two phases over every fp32 gradient. Sharded: one phase over the fp32 O x I
    gradients, one over the packed shadows (2 x shadow_bytes per weight), the small messages twice."""
    f = (world - 1) / world
    big = sum(2 * d["O"] * d["I"] for d in layers)
    small = sum(d["O"] for d in layers) + final["weight"][1] + final["bias"][1]
    return f * 2 * 4 * (big + small), f * (4 * big + 2 * shadow_bytes * big // 2 + 2 * 4 * small)
