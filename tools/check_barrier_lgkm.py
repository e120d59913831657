#!/usr/bin/env python3
"""Static check of generated gfx950 code: how many LDS reads (ds_read* / ds_load*) can a wave have OUTSTANDING when it
executes an s_barrier?

Why it matters (gemm_v2.h, v2_wait_barrier): the pipelined GEMM loops refill an LDS stage by LDS-DMA ONE phase after its
last read. That is only legal when every reader's ds_reads have RETURNED (s_waitcnt lgkmcnt) before the reader passes the
barrier that releases the refill -- a read that is merely issued sits in the LDS queue while another wave's DMA piece
heads for the same bytes. hipcc places the lgkmcnt wait in front of the consuming MFMA, and an MFMA may be scheduled
across s_barrier; in r02 it sank eight MFMAs and their wait below the barrier in the DUAL SCHED-0 instantiations of
gemm_nt_v2 (two reads outstanding at the barrier) -- the "late piece" of the two-rank rehearsal.

Input: `llvm-objdump -d` of a code object, or a `hipcc -S` listing. Dataflow over the instruction list of each function
(forward, max over predecessors, to a fixpoint): count += 1 per LDS read, count = min(count, n) at `s_waitcnt
lgkmcnt(n)` (LDS operations of one wave return in order). A barrier whose incoming count is > 0 is reported.
SMEM / GWS also use lgkmcnt; they only make a real wait look weaker, never hide a read: the check is conservative.

  python3 tools/check_barrier_lgkm.py file.dis|file.s [substring the kernel name must contain]   exit 1 if any"""
import re
import sys

LG = re.compile(r"lgkmcnt\((\d+)\)")
CAP = 255


def parse(path):
    """-> {function: [(op, text, label_here or None, branch_target or None)]}; labels / targets are function-local keys"""
    funcs, cur, pending = {}, None, None
    objdump_fn = re.compile(r"^[0-9a-fA-F]+ <([^>]+)>:")
    asm_fn = re.compile(r"^([A-Za-z_][\w$.]*):")
    for raw in open(path, errors="replace"):
        line = raw.rstrip("\n")
        m = objdump_fn.match(line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            pending = None
            continue
        s = line.strip()
        if not s or s.startswith(";") or s.startswith("//"):
            continue
        if not line[0].isspace():
            m = asm_fn.match(line)
            if m and not m.group(1).startswith(".L"):
                cur = funcs.setdefault(m.group(1), [])
                pending = None
                continue
            m = re.match(r"^(\.L[\w$.]+):", line)
            if m:
                pending = m.group(1)
                continue
            if line.startswith("."):
                continue
        if cur is None or s.startswith("."):
            continue
        op = s.split()[0]
        label, target = pending, None
        pending = None
        am = re.search(r"//\s*([0-9A-Fa-f]{8,}):", s)              # objdump: the instruction's own address
        if am:
            label = int(am.group(1), 16)
        if op.startswith("s_cbranch") or op == "s_branch":
            tm = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", s)
            if tm:
                target = ("off", int(tm.group(1), 16))
            else:
                parts = s.split()
                if len(parts) > 1 and parts[1].startswith(".L"):
                    target = parts[1]
        cur.append((op, s, label, target))
    return funcs


def analyse(insts):
    n = len(insts)
    idx_of = {}
    base = None
    for i, (_, _, label, _) in enumerate(insts):
        if isinstance(label, int):
            base = label if base is None else base
            idx_of[("off", label - base)] = i
        elif label is not None:
            idx_of[label] = i
    state_in = [None] * n
    if n:
        state_in[0] = 0
    work = [0] if n else []
    while work:
        i = work.pop()
        c = state_in[i]
        while i < n:
            op, s, _, target = insts[i]
            out = c
            if op.startswith("ds_read") or op.startswith("ds_load"):
                out = min(CAP, c + 1)
            elif op == "s_waitcnt":
                m = LG.search(s)
                if m:
                    out = min(c, int(m.group(1)))
            if target is not None and target in idx_of:
                j = idx_of[target]
                if state_in[j] is None or state_in[j] < out:
                    state_in[j] = out
                    work.append(j)
            if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                break
            i += 1
            if i < n:
                if state_in[i] is not None and state_in[i] >= out:
                    break
                state_in[i] = out
                c = out
    return [(i, state_in[i]) for i in range(n) if insts[i][0] == "s_barrier" and state_in[i]]


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    funcs = parse(path)
    bad = barriers = kernels = 0
    for name, insts in funcs.items():
        if want and want not in name:
            continue
        nb = sum(1 for x in insts if x[0] == "s_barrier")
        if not nb:
            continue
        kernels += 1
        barriers += nb
        for i, c in analyse(insts):
            bad += 1
            print(f"{name[:110]}: s_barrier (instruction {i}) with up to {c} LDS read(s) outstanding")
    print(f"{kernels} kernel(s), {barriers} barrier(s): {bad} with LDS reads possibly outstanding")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
