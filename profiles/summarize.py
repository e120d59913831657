#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run: python profiles/summarize.py <dir-or-kernel_stats.csv> [steps]"""
import csv, glob, os, re, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
rows = list(csv.DictReader(open(p)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'total ms':>10} {'calls':>6} {'avg us':>9} {'%':>6}  kernel")
for r in rows[:22]:
    n = r["Name"]
    m = re.match(r"(?:void )?(?:_Z\d+)?([A-Za-z_0-9:]+)", n)
    short = n if len(n) < 100 else n[:100]
    print(f"{float(r['TotalDurationNs'])/1e6:10.3f} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}  {short}")
print(f"sum of kernel time: {tot/1e6:.3f} ms" + (f" = {tot/1e6/steps:.3f} ms per step over {steps:g} steps" if steps else ""))
