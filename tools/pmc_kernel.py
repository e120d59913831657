#!/usr/bin/env python3
"""One rocprofv3 PMC pass over a command, per-kernel means of the counters (ON the GPU box):
python3 tools/pmc_kernel.py <kernel-name-substring[+more][+!excluded]> CTR1,CTR2,... -- python3 /abs/path/script.py args"""
import csv, glob, os, subprocess, sys
sub, ctrs = sys.argv[1], sys.argv[2].split(",")
cmd = sys.argv[sys.argv.index("--") + 1:]
d = "/tmp/pmc_one"
subprocess.run(["rm", "-rf", d])
subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *ctrs, "--output-format", "csv", "-d", d, "-o", "p", "--", *cmd],
               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=open("/tmp/pmc_one.err", "w"))
if not glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    print(open("/tmp/pmc_one.err").read()[-1500:]); sys.exit(1)
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc, n = {}, {}
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if all((t[1:] not in name) if t.startswith("!") else (t in name) for t in sub.split("+")):        # "a+b+!c": contains a and b, not c
        k = r["Counter_Name"]
        acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"]); n[k] = n.get(k, 0) + 1
for k in ctrs:
    if k in acc: print(f"{k:32s} {acc[k] / n[k]:16.1f}   ({n[k]} dispatches)")
