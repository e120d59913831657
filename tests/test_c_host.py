"""tools/c_host.c -- the executable stand-in for the LuaJIT-FFI host (lua/FusedMLP.lua cannot run here: no LuaJIT): a
plain C program against include/vbnn_hip.h, no Python, no PyTorch, all device memory through vbnn_buf_*, the NULL stream,
RCCL from the system's librccl.so.1. CPU part: it builds warning-free with gcc and fails loudly without a GPU, and
lua/FusedMLP.lua's `run` issues the library calls in the order the C program does. GPU part (-m gpu): it runs as a child
process and its gradient arena is BITWISE vbnn_amd/engine.py's on the same configuration (mlp.lua:76-84, main.lua:28-40)."""
import os
import re
import struct
import subprocess
import sys

import numpy as np
import pytest

from tests import _children

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "c_host.c")


def build(tmp_path):
    from vbnn_amd import _lib as L
    exe = str(tmp_path / "c_host")
    libdir = os.path.dirname(L.LIB_PATH)
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
           "-L", libdir, "-lvbnn_hip", "-lm", f"-Wl,-rpath,{libdir}"]
    res = _children.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    return exe


def test_c_host_builds_against_the_header_alone_and_needs_the_gpu(tmp_path):
    src = open(SRC).read()
    includes = re.findall(r'#include\s+[<"]([^>"]+)[>"]', src)
    assert "vbnn_hip.h" in includes and not [i for i in includes if "hip/" in i or "torch" in i or "Python" in i], includes
    exe = build(tmp_path)
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU test runs it")
    res = _children.run([exe, "--out", str(tmp_path / "a.bin")], capture_output=True, text=True, timeout=120)
    assert res.returncode == 2 and "vbnn_ctx_create" in res.stderr, (res.returncode, res.stderr[-500:])   # loud, no fallback


def _c_calls(body):
    out = []
    for m in re.finditer(r"\b(vbnn_[a-z0-9_]+)\s*\(", body):
        if not out or out[-1] != m.group(1):
            out.append(m.group(1))
    return out


def test_lua_fused_mlp_issues_the_calls_of_the_c_host_in_the_same_order():
    """The Lua host cannot be executed; the C host can. Their `run`, `prepare`, `update` and constructor issue the same
    library calls in the same order (consecutive repeats folded), so what the GPU test proves about the C program's call
    sequence holds for the Lua file's."""
    c = open(SRC).read()
    c = re.sub(r"/\*.*?\*/", " ", c, flags=re.S)
    lua = open(os.path.join(ROOT, "lua", "FusedMLP.lua")).read()
    lua = re.sub(r"--[^\n]*", " ", lua)

    def c_fn(name):
        i = re.search(r"static void %s\([^;{]*\)\s*\{" % name, c).end() - 1          # the definition, not a forward declaration
        depth, j = 1, i + 1
        while depth:
            depth += {"{": 1, "}": -1}.get(c[j], 0)
            j += 1
        return c[i:j]

    def lua_fn(name, until):
        i = lua.index("function FusedMLP%s" % name)
        return lua[i:lua.index("function FusedMLP%s" % until, i + 1)]

    def lua_calls(body):
        out = []
        for m in re.finditer(r"\bC\.(vbnn_[a-z0-9_]+)\s*\(", body):
            if not out or out[-1] != m.group(1):
                out.append(m.group(1))
        return out

    pairs = [("fm_run", lua_fn(":run", ":finish")), ("fm_prepare", lua_fn(":prepare", ":sample")),
             ("fm_update", lua_fn(":update", ":calc_lc")), ("fm_alloc_batch", lua_fn(":_alloc_batch", ":resetGradients")),
             ("fm_scatter", lua_fn(":_scatter", ":_update_sharded")), ("fm_update_sharded", lua_fn(":_update_sharded", ":run"))]
    for cname, lbody in pairs:
        cc = [n for n in _c_calls(c_fn(cname)) if n not in ("vbnn_last_error",)]
        ll = lua_calls(lbody)
        assert cc == ll, (cname, cc, ll)


# ------------------------------------------------------------------------------------------------ GPU
def _read_arena(path, n_layers_sizes=None):
    raw = open(path, "rb").read()
    n, loss, correct, flags = struct.unpack_from("<qdii", raw, 0)
    arena = np.frombuffer(raw, dtype=np.float32, count=n, offset=24)
    rest = np.frombuffer(raw, dtype=np.float32, offset=24 + 4 * n)
    return n, loss, correct, flags, arena, rest


CASES = [
    # dtype, input, hidden, batch, S, steps, update, comm (comm = "graph": steps 2.. replayed from ONE captured graph)
    ("f32", 784, [400, 400], 256, 1, 2, False, False),          # BASELINE configs[1]: the numerics configuration
    ("f32", 784, [400, 400], 100, 3, 2, True, False),           # configs[0]'s batch, S draws accumulating, an update between
    ("bf16", 256, [512, 256], 512, 1, 2, False, False),         # transposed operands (no K-major form at this size)
    ("bf16", 256, [512, 256], 512, 1, 3, True, True),           # + update + the RCCL exchange (layerwise order)
    ("bf16", 784, [4096, 4096], 4096, 1, 2, True, False),       # the bench's configuration: K-major, split launch, dx-first
    ("bf16", 784, [4096, 4096], 4096, 1, 2, False, True),       # + RCCL: early d/dlvars messages, three all-reduces per step
    ("bf16", 256, [512, 256], 512, 1, 3, True, "sharded"),      # the sharded-update exchange driven from C (a world of one): reduce-scatter,
    ("bf16", 784, [4096, 4096], 4096, 1, 3, True, "sharded"),   # slice update, shadow + statistics all-gather, vbnn_stats_combine, transposes rebuilt
    ("f32", 784, [400, 400], 256, 1, 4, False, "graph"),        # the C host captures its step and replays it: own stream, device draw counter
    ("f32", 784, [400, 400], 100, 3, 3, False, "graph"),        # ... with S = 3 accumulating draws inside the graph
]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,I0,hidden,N,S,steps,update,comm", CASES)
def test_c_host_gradient_arena_is_bitwise_the_python_engines(tmp_path, dtype, I0, hidden, N, S, steps, update, comm):
    import torch
    from vbnn_amd.engine import FusedMLP
    from vbnn_amd.nn import fill_normal
    exe = build(tmp_path)
    out = str(tmp_path / "arena.bin")
    cmd = [exe, "--dtype", dtype, "--input", str(I0), "--hidden", ",".join(str(h) for h in hidden), "--classes", "10",
           "--batch", str(N), "--S", str(S), "--steps", str(steps), "--out", out]
    graph = comm == "graph"
    sharded = comm == "sharded"
    comm = bool(comm) and not graph
    cmd += ["--update"] if update else []
    cmd += ["--comm"] if comm else []
    cmd += ["--sharded"] if sharded else []
    cmd += ["--graph"] if graph else []
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VBNN_RCCL_PATH="/opt/rocm/lib/librccl.so.1")
    res = _children.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-1000:] + res.stderr[-3000:]
    print(res.stdout.strip())
    assert not graph or "replays of one captured graph" in res.stdout
    n, loss_c, correct_c, flags, arena_c, rest = _read_arena(out)

    opt = dict(var_init=1e-3, mu_init=1, B=1e6, S=S, mode="lrt", dtype=dtype, seed=3, input_size=I0, hidden=hidden,
               n_classes=10, fuse_kl=True, state=dict(learningRate=1e-3), meanState=dict(learningRate=1e-4),
               varState=dict(learningRate=5e-2))
    if sharded:
        opt["exchange_mode"] = "sharded"
    eng = FusedMLP(opt, force_reduce=comm)
    x = torch.empty(N, I0, dtype=torch.float32, device="cuda")
    fill_normal(x, 3, 4, 0, 0)
    t = (torch.arange(N, device="cuda", dtype=torch.int64) * 7 % 10).to(torch.int32)
    eng.prepare()
    for step in range(steps):
        eng.resetGradients()
        for _ in range(S):
            eng.sample(); eng.run(x, t)
        eng.finish()
        if update and step + 1 < steps:
            eng.update(opt)
    loss_p, correct_p = eng.loss_and_accuracy()
    arena_p = eng.grads.cpu().numpy()
    assert n == arena_p.size
    assert bool(flags & 4) == bool(eng.dx_first and not eng.reduce), (flags, eng.dx_first)
    if comm:
        assert eng.comm_backend() == "vbnn_comm/rccl"
    assert loss_c == loss_p and correct_c == correct_p, (loss_c, loss_p, correct_c, correct_p)
    same = arena_c.view(np.uint32) == arena_p.view(np.uint32)
    assert same.all(), f"{(~same).sum()} of {same.size} arena words differ; first at {int(np.argmax(~same))}"
    assert np.isfinite(arena_c).all() and np.abs(arena_c).max() > 0
    if update:
        mu_p = np.concatenate([v.means.cpu().numpy().ravel() for v in eng.vb])
        assert np.array_equal(rest.view(np.uint32), mu_p.view(np.uint32)), "means after the update differ"
