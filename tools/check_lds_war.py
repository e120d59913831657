#!/usr/bin/env python3
"""Static check of a hipcc -S listing: does any ds_read_b64_tr_b16 (issued from inline asm, invisible to hipcc's hazard
logic) write a register that one of the wave's last DIST matrix instructions reads as its A / B operand, with no
s_barrier in between?  python3 tools/check_lds_war.py file.s [DIST=4]   (see gemm_v3.h, tr_issue2_keep)"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
dist = int(sys.argv[2]) if len(sys.argv) > 2 else 4
def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()
recent, viol, kernel = [], 0, "?"
for i, l in enumerate(lines):
    t = l.strip()
    if l.startswith("_Z") and ":" in l: kernel, recent = l.split(":")[0][:70], []
    if t.startswith("s_barrier"): recent = []
    if t.startswith("v_mfma"):
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
        recent = (recent + [(i, regs(ops[1]) | regs(ops[2]))])[-dist:]
    if t.startswith("ds_read_b64_tr_b16"):
        dst = regs(t.split()[1].rstrip(","))
        for k, (ln, src) in enumerate(reversed(recent)):
            if dst & src:
                viol += 1
                print(f"{kernel} line {i + 1}: {t.split()[1]} is an operand of the MFMA {k} back (line {ln + 1})")
                break
print(f"{viol} read(s) land in operands of the last {dist} MFMAs")
sys.exit(1 if viol else 0)
