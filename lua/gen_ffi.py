#!/usr/bin/env python3
"""Regenerates the ffi.cdef block of lua/vbnn_ffi.lua from include/vbnn_hip.h (comments and preprocessor
lines stripped; LuaJIT's cdef parser takes plain C declarations)."""
import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hdr = open(os.path.join(ROOT, "include", "vbnn_hip.h")).read()
body = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
keep = [l.rstrip() for l in body.split("\n") if l.strip() and not l.strip().startswith("#")
        and not l.strip().startswith('extern "C"') and l.strip() != "}"]
path = os.path.join(ROOT, "lua", "vbnn_ffi.lua")
txt = open(path).read()
a, b = txt.index("ffi.cdef[[") + len("ffi.cdef[[\n"), txt.index("]]")
open(path, "w").write(txt[:a] + "\n".join(keep) + "\n" + txt[b:])
print("regenerated", path)
