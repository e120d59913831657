#!/usr/bin/env python3
"""LAB: gaps between consecutive dispatches of tools/gap_probe.py's trace: python3 tools/gap_probe_read.py <results.db>"""
import sqlite3, sys, statistics as st, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute(f"select k.kernel_name, d.start, d.end from {disp} d join {sym} k on d.kernel_id = k.id order by d.start"))
g = collections.defaultdict(list)
for (n0, s0, e0), (n1, s1, e1) in zip(rows[:-1], rows[1:]):
    if (s1 - e0) < 50000:
        g[(n0[:48], round((e0 - s0) / 1e3, -1), n1[:24])].append((s1 - e0) / 1e3)
for k, v in sorted(g.items(), key=lambda kv: -len(kv[1])):
    if len(v) >= 6:
        print(f"{k[0]:50s} ~{k[1]:7.0f} us  -> {k[2]:26s} gap median {st.median(v):5.1f} us (n={len(v)})")
