#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run stored as a rocpd database:
python profiles/summarize_db.py <results.db> [steps]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
unit = 1e3 if rows and rows[0][3] > 1e4 else 1.0          # ns or us, depending on the rocprofv3 build
print(f"{'total ms':>10} {'calls':>6} {'avg us':>9} {'%':>6}  kernel")
tot = 0.0
for n, calls, total, avg, pct in rows[:24]:
    tot += total / unit
    print(f"{total/unit/1e3:10.3f} {calls:6d} {avg/unit:9.1f} {pct:6.2f}  {n[:110]}")
tot = sum(r[2] for r in rows) / unit
print(f"sum of kernel time: {tot/1e3:.3f} ms" + (f" = {tot/1e3/steps:.3f} ms per step over {steps:g} steps" if steps else ""))
