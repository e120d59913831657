"""CPU tests of the drop-in boundary: the shared library loads, exports every symbol the header
declares, the ctypes mirror agrees with the C struct layouts, and the Lua FFI cdef is in step with the
header. No compute call is made (there is no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vbnn_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|const char\*)\s+(vbnn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from vbnn_amd import _lib as L
    assert os.path.exists(L.LIB_PATH), "libvbnn_hip.so must be built in-tree (python -c 'import __graft_entry__ as g; g.build()')"
    lib = C.CDLL(L.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/vbnn_hip.h but not exported"
    # and the ctypes table covers exactly the header
    assert sorted(L.exported_symbols()) == names
    L.lib()
    assert L.lib().vbnn_abi_version() == 6


def test_no_gpu_is_an_error_not_a_fallback():
    """On a box without a HIP device the product path must fail loudly (no CPU / oracle fallback)."""
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from vbnn_amd import _lib as L
    h = C.c_void_p()
    st = L.lib().vbnn_ctx_create(0, None, C.byref(h))
    assert st != 0
    assert len(L.lib().vbnn_last_error()) > 0
    with pytest.raises(L.VbnnError):
        L.check(st)


def test_product_path_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vbnn_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                bad = re.findall(r"^\s*(?:from\s+\S*oracle\S*\s+import|import\s+\S*oracle|#include\s+\S*oracle)|libvbnn_oracle|vbo_",
                                 txt, flags=re.M)
                assert not bad, f"{f} reaches into oracle/: {bad}"


def test_ctypes_structs_match_the_header():
    """Compile a C program against the header that prints sizeof/offsetof of every argument block."""
    from vbnn_amd import _lib as L
    structs = {"vbnn_fwd_args": L.FwdArgs, "vbnn_dx_args": L.DxArgs, "vbnn_dw_args": L.DwArgs,
               "vbnn_prep_desc": L.PrepDesc, "vbnn_pack_desc": L.PackDesc, "vbnn_adam_cfg": L.AdamCfg,
               "vbnn_update_desc": L.UpdateDesc, "vbnn_head_args": L.HeadArgs, "vbnn_box_info": L.BoxInfo}
    cfield = lambda f: f.rstrip("_")                       # `lambda` is a Python keyword: the mirror calls it lambda_
    lines = ["#include <stdio.h>", "#include <stddef.h>", f'#include "{HEADER}"', "int main(void){"]
    for cname, st in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in st._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {cfield(fname)}));')
    lines.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.c"), os.path.join(d, "t")
        open(src, "w").write("\n".join(lines))
        subprocess.check_call(["gcc", "-std=c11", "-o", exe, src])
        out = subprocess.check_output([exe]).decode().split("\n")
    got = dict(l.split() for l in out if l)
    for cname, st in structs.items():
        assert int(got[cname]) == C.sizeof(st)
        for fname, _ in st._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(st, fname).offset, f"{cname}.{fname}"


def test_lua_cdef_matches_header():
    """lua/VBLinear.lua cannot be executed here (no LuaJIT); keep its ffi.cdef in step with the header:
    every function the header declares must appear in the cdef with the same parameter count."""
    lua = os.path.join(ROOT, "lua", "vbnn_ffi.lua")
    if not os.path.exists(lua):
        pytest.skip("Lua shim not written yet")
    txt = open(lua).read()
    cdef = txt[txt.index("ffi.cdef[["):txt.index("]]")]
    hdr = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)

    def protos(s):
        return {m.group(1): len([p for p in m.group(2).split(",") if p.strip() and p.strip() != "void"])
                for m in re.finditer(r"(vbnn_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", s, flags=re.S)}

    ph, pl = protos(hdr), protos(cdef)
    for name in declared_functions():
        assert name in pl, f"{name} missing from the Lua cdef"
        assert pl[name] == ph[name], f"{name}: {pl[name]} parameters in Lua, {ph[name]} in the header"


def test_pad_ld():
    from vbnn_amd import _lib as L
    assert [L.pad_ld(k) for k in (1, 10, 64, 65, 784, 4096)] == [64, 64, 64, 128, 832, 4096]


if __name__ == "__main__":
    sys.exit(pytest.main([__file__, "-q"]))


def _lua_tokens(path):
    """Identifier / keyword tokens of a Lua file with comments and string literals removed."""
    txt = open(path).read()
    txt = re.sub(r"--\[\[.*?\]\]", " ", txt, flags=re.S)
    txt = re.sub(r"--[^\n]*", " ", txt)
    txt = re.sub(r"'(?:\\.|[^'\\\n])*'|\"(?:\\.|[^\"\\\n])*\"", " S ", txt)
    return txt, re.findall(r"[A-Za-z_][A-Za-z_0-9]*", txt)


def test_lua_shim_structure():
    """No Lua interpreter exists here, so lua/VBLinear.lua is linted structurally: block keywords balance, every
    C.vbnn_* it calls is declared in include/vbnn_hip.h, and it defines every method the reference's callers use
    (mlp.lua:14-142, main.lua:127-128: the §8b method set)."""
    path = os.path.join(ROOT, "lua", "VBLinear.lua")
    txt, toks = _lua_tokens(path)
    opens = sum(toks.count(k) for k in ("function", "if", "for", "while")) + toks.count("repeat") * 0
    # `for ... do` / `while ... do` open one block each (counted by for / while); a bare `do ... end` would add one
    bare_do = toks.count("do") - toks.count("for") - toks.count("while")
    assert bare_do >= 0
    assert opens + bare_do == toks.count("end"), (opens, bare_do, toks.count("end"))
    assert txt.count("(") == txt.count(")") and txt.count("{") == txt.count("}") and txt.count("[") == txt.count("]")
    header = open(os.path.join(ROOT, "include", "vbnn_hip.h")).read()
    declared = set(re.findall(r"\b(vbnn_[a-z0-9_]+)\s*\(", header))
    used = set(re.findall(r"\bC\.(vbnn_[a-z0-9_]+)", txt))
    assert used and used <= declared, used - declared
    methods = set(re.findall(r"function\s+VBLinear:([A-Za-z_]+)", txt))
    need = {"__init", "sample", "clamp_to_map", "resetAcc", "updateOutput", "updateGradInput", "accGradParameters",
            "compute_prior", "compute_mugrads", "compute_vargrads", "calc_lc", "update"}
    assert need <= methods, need - methods


def _split_args(argtxt):
    """Top-level comma split of a Lua argument list."""
    out, depth, cur = [], 0, ""
    for ch in argtxt:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def _calls(txt):
    """(name, n_args) of every C.vbnn_*(...) call, in source order."""
    res = []
    for m in re.finditer(r"\bC\.(vbnn_[a-z0-9_]+)\s*\(", txt):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(txt[i], 0)
            i += 1
        res.append((m.group(1), len(_split_args(txt[m.end():i - 1]))))
    return res


def test_lua_fused_mlp_structure():
    """lua/FusedMLP.lua is the device-resident Lua caller; no interpreter exists here, so it is
    linted: blocks balance; every C.vbnn_* call is declared in the header and passes that many arguments; every field it
    sets on a vbnn_* argument block exists in that struct; and `run` issues the library calls in the ORDER
    vbnn_amd/engine.py:FusedMLP.run does (the sequence the GPU parity tests execute)."""
    path = os.path.join(ROOT, "lua", "FusedMLP.lua")
    txt, toks = _lua_tokens(path)
    opens = sum(toks.count(k) for k in ("function", "if", "for", "while"))
    bare_do = toks.count("do") - toks.count("for") - toks.count("while")
    assert bare_do >= 0 and opens + bare_do == toks.count("end"), (opens, bare_do, toks.count("end"))
    assert txt.count("(") == txt.count(")") and txt.count("{") == txt.count("}") and txt.count("[") == txt.count("]")
    hdr = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    protos = {m.group(1): len([p for p in m.group(2).split(",") if p.strip() and p.strip() != "void"])
              for m in re.finditer(r"(vbnn_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S)}
    calls = _calls(txt)
    assert len(calls) >= 20
    for name, n in calls:
        assert name in protos, f"{name} is not declared in include/vbnn_hip.h"
        assert n == protos[name], f"{name}: {n} arguments in FusedMLP.lua, {protos[name]} parameters in the header"
    # struct fields
    structs = {m.group(2): set(re.findall(r"(\w+)\s*(?:,|;)", re.sub(r"\b(?:const|void|float|double|int|int64_t|uint64_t|uint32_t|vbnn_adam_cfg)\b|\*", " ", m.group(1))))
               for m in re.finditer(r"typedef struct \w+ \{(.*?)\}\s*(vbnn_\w+);", hdr, flags=re.S)}
    raw = open(path).read()
    checked = 0
    for chunk in re.split(r"\n(?=function |local function )", raw):          # variable names are per function
        types = {}
        for m in re.finditer(r"local\s+(\w+)\s*=\s*ffi\.new\('(vbnn_\w+?)(?:\[[^']*\])?'", chunk):
            types[m.group(1)] = m.group(2)
        for m in re.finditer(r"local\s+(\w+)\s*=\s*(\w+)\[[^\]]*\]\s*\n", chunk):
            if m.group(2) in types:
                types[m.group(1)] = types[m.group(2)]
        for var, st in types.items():
            if st not in structs:
                continue
            for m in re.finditer(r"(?<![\w.])%s(?:\[0\])?\.(\w+)" % re.escape(var), chunk):
                assert m.group(1) in structs[st], f"{var}.{m.group(1)}: no such field in {st}"
                checked += 1
    assert checked >= 60, checked
    # call order of run == engine.py's
    body = raw[raw.index("function FusedMLP:run"):raw.index("function FusedMLP:finish")]
    lua_order = []
    for name, _ in _calls(body):
        if not lua_order or lua_order[-1] != name:
            lua_order.append(name)
    eng = open(os.path.join(ROOT, "vbnn_amd", "engine.py")).read()
    run = eng[eng.index("    def run(self, inputs, targets"):eng.index("            main, side, ctx2 =")]
    # (the head: the one-call form is what the Lua host issues; engine.py's two-call branch -- test path, opt.head_step = False -- is cut)
    run = run[:run.index("                if self._use_head_slots():\n                    L.check(lib.vbnn_head_forward_slots(")] + run[run.index("        # ---------------- backward: VB layers"):]
    py_order = []
    for m in re.finditer(r"lib\.(vbnn_[a-z0-9_]+)\(|self\.(_reduce)\(", run):
        name = m.group(1) or "vbnn_allreduce_grads"
        if not py_order or py_order[-1] != name:
            py_order.append(name)
    assert lua_order == py_order, (lua_order, py_order)
