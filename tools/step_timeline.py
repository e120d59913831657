#!/usr/bin/env python3
"""One steady-state step of bench.py out of a rocprofv3 --kernel-trace database (rocpd): every dispatch between two
k_pack_input launches, with duration and the gap to its predecessor; plus the median span over 20 steps around it.
bench.py's LAST block of --steps steps is the probed one (HIP events around three launches: ~6 us of gap at each event),
so the default looks 30 steps back from the end, into the last timed block of a `--steps 20` run.
python tools/step_timeline.py <results.db> [step index from the end, default 30] [marker substring, default k_pack_input] [every n-th marker]
(the fp32 step has no packing launch: mark it by its first forward, `EpiFwd 2`, or -- captured -- by `k_sample_advance`)"""
import sqlite3, statistics, sys
db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute(f"select k.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {disp} d join {sym} k on d.kernel_id=k.id order by d.start"))
marker = sys.argv[3] if len(sys.argv) > 3 else "k_pack_input"
nth = int(sys.argv[4]) if len(sys.argv) > 4 else 1
idx = [i for i, r in enumerate(rows) if marker in r[0]][::nth]
if len(idx) < back + 2:
    print("too few steps in the trace"); sys.exit(1)
i0, i1 = idx[-back - 1], idx[-back]
prev = None
for r in rows[i0:i1]:
    gap = (r[1] - prev) / 1e3 if prev else 0.0
    print(f"{(r[2] - r[1]) / 1e3:8.1f} us  gap {gap:5.1f}  grid {r[3] // max(r[4], 1):5d} x {r[4]:4d}  {r[0][:96]}")
    prev = r[2]
print(f"step span {(rows[i1][1] - rows[i0][1]) / 1e3:.1f} us")
lo = max(0, len(idx) - back - 11)
spans = [(rows[idx[k + 1]][1] - rows[idx[k]][1]) / 1e3 for k in range(lo, min(lo + 20, len(idx) - 1))]
print(f"median span of {len(spans)} steps around it: {statistics.median(spans):.1f} us (min {min(spans):.1f}, max {max(spans):.1f})")
