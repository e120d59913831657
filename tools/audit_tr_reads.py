#!/usr/bin/env python3
"""Audit of the split issue / wait inline-asm transpose reads in gemm_v3.h and gemm_v2.h (K-major operands).

A `ds_read_b64_tr_b16` issued by one asm statement is waited for by a LATER one; hipcc knows nothing about the
load in flight, so nothing in between may read or write its destination registers (a compiler-inserted copy or spill
there would move garbage). This script compiles csrc/gemm_fwd.hip, gemm_dx.hip and gemm_dw.hip to assembly and checks exactly that for every
gemm_nt_v3 / gemm_nt_v2 instantiation with a K-major operand. LDS operations return in order, so a counted
`s_waitcnt lgkmcnt(N)` retires all but the N youngest of them: the reads are tracked as a FIFO. Run it after any change to gemm_v3.h or to the compiler:
    python3 tools/audit_tr_reads.py                  (CPU only; compiles three translation units to assembly: ~6 min)
    python3 tools/audit_tr_reads.py --dis file.dis   (the same over `llvm-objdump -d` text of built code objects)
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs_of(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def audit_function(name, body_lines):
    """-> (transpose reads, violations) over one kernel's instruction lines (a `hipcc -S` listing or `llvm-objdump -d` text)"""
    fifo, bad, nread = [], 0, 0                     # LDS operations in flight, oldest first: sets of destination VGPRs
    for line in body_lines:
        line = line.split(";")[0].split("//")[0].strip()
        if not line or line.endswith(":") or line.startswith("."):
            continue
        if line.startswith("ds_read_b64_tr_b16"):
            fifo.append(regs_of(line.split()[1].rstrip(",")))
            nread += 1
            continue
        m = re.search(r"lgkmcnt\((\d+)\)", line) if line.startswith("s_waitcnt") else None
        if m:
            keep = int(m.group(1))
            fifo = fifo[len(fifo) - keep:] if keep else []
            continue
        inflight = set().union(*fifo) if fifo else set()
        if regs_of(line) & inflight:
            bad += 1
            print("  touches a register in flight:", line)
        if line.startswith("ds_") and not line.startswith("ds_read_b64_tr_b16"):
            fifo.append(set())                      # any other LDS operation takes a slot in the in-order queue
    print(f"{name[:72]:72s} transpose reads {nread:4d}  violations {bad}")
    return nread, bad


WANTED = re.compile(r"^_Z10gemm_nt_v3ILb[01]E(?:Lb1ELb[01]|Lb0ELb1)\S+$|^_Z10gemm_nt_v2I\S+?Lb1EEv\S+$")     # the K-major instantiations


def audit_disassembly(path):
    """The same check on `llvm-objdump -d` text of the SHIPPED code objects (tests/test_kernel_hazards.py: seconds instead of minutes)."""
    funcs, cur = {}, None
    for raw in open(path, errors="replace"):
        m = re.match(r"^[0-9a-fA-F]+ <([^>]+)>:", raw)
        if m:
            cur = funcs.setdefault(m.group(1), []) if WANTED.match(m.group(1)) else None
            continue
        if cur is not None and raw.strip():
            cur.append(raw)
    total = 0
    for name, body in funcs.items():
        total += audit_function(name, body)[1]
    if not funcs:
        print("no K-major instantiation found")
        return 1
    return 1 if total else 0


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--dis":
        return audit_disassembly(sys.argv[2])
    txt = ""
    with tempfile.TemporaryDirectory() as d:
        for src in ("gemm_fwd.hip", "gemm_dx.hip", "gemm_dw.hip"):          # the translation units that instantiate the pipelined kernels
            asm = os.path.join(d, src + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include",
                            f"-I{ROOT}/vbnn_amd/csrc", "-S", "--cuda-device-only", "-o", asm, f"{ROOT}/vbnn_amd/csrc/{src}"],
                           check=True, stderr=subprocess.DEVNULL)
            txt += open(asm).read() + "\n"
    total_bad, kernels = 0, 0
    names = [n for n in re.findall(r"^(_Z10gemm_nt_v[23]I\S+):", txt, re.M) if WANTED.match(n)]
    for name in names:
        a = txt.index(name + ":")
        b = txt.index(".Lfunc_end", a)
        total_bad += audit_function(name, txt[a:b].splitlines())[1]
        kernels += 1
    if kernels == 0:
        print("no K-major instantiation found")
        return 1
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
