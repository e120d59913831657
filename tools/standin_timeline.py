#!/usr/bin/env python3
"""Do the stand-in exchange's kernels RUN BESIDE the backward's GEMM launches, or before / after them? One step of
tools/overlap_standin.py's default arm out of a rocprofv3 --kernel-trace database: every dispatch between two k_pack_input
launches with its start and end relative to the step's start, ordered by start, and for every exchange kernel (k_p2p_*) the GEMM
launch(es) whose interval it overlaps and by how much.
    python tools/standin_timeline.py <results.db> [step index from the end, default 12]
The database:  cd /tmp && rocprofv3 --kernel-trace -d <dir> -o st -- python3 <repo>/tools/overlap_standin.py --one-arm"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute(f"select k.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {disp} d join {sym} k on d.kernel_id=k.id order by d.start"))
idx = [i for i, r in enumerate(rows) if "k_pack_input" in r[0]]
i0, i1 = idx[-back - 1], idx[-back]
t0, t1 = rows[i0][1], rows[i1][1]
step = [r for r in rows if t0 <= r[1] < t1]
gemms = [r for r in step if "gemm_nt" in r[0]]
print(f"one step, {(t1 - t0) / 1e3:.1f} us; start / end in us from the step's first launch")
for r in step:
    name = r[0].replace("void ", "")[:70]
    line = f"{(r[1] - t0) / 1e3:8.1f} -> {(r[2] - t0) / 1e3:8.1f}  ({(r[2] - r[1]) / 1e3:6.1f} us)  grid {r[3] // max(r[4], 1):5d} x {r[4]:4d}  {name}"
    if "k_p2p" in r[0]:
        ov = [(g, min(r[2], g[2]) - max(r[1], g[1])) for g in gemms]
        ov = [(g, o) for g, o in ov if o > 0]
        line += "   || " + (", ".join(f"{o / 1e3:.1f} us beside {('EpiFwd' if 'EpiFwd' in g[0] else 'EpiDx' if 'EpiDx' in g[0] else 'EpiDw')} @{(g[1] - t0) / 1e3:.0f}" for g, o in ov) if ov else "beside no GEMM launch")
    print(line)
ex = [r for r in step if "k_p2p_reduce_scatter" in r[0] or "k_p2p_all_gather" in r[0]]
tot = sum(r[2] - r[1] for r in ex)
beside = sum(max(0, min(r[2], g[2]) - max(r[1], g[1])) for r in ex for g in gemms)
print(f"exchange data kernels: {tot / 1e3:.1f} us in all, {beside / 1e3:.1f} us of it beside a GEMM launch")
