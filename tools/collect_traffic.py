#!/usr/bin/env python3
"""Collect the PMC figures of the wide step's dominant GEMM kernels into profiles/rNN_traffic.json.

Run ON the GPU box from the repository root:   python3 tools/collect_traffic.py r01
One rocprofv3 pass per counter group (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE cannot share a pass; counters are
collected with --kernel-trace only), each pass over `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`.
Per GEMM family the launches of the widest layer are the kernel's LONGEST dispatches; their counter values are averaged.
FETCH_SIZE is doubled (gfx950 tallies a 128-B request as 64 B); hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) KB.
"""
import csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"],
    ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"],
    ["TCC_HIT_sum", "TCC_MISS_sum"],
]
# family name as bench.py prints it -> (substring that identifies the epilogue functor in the kernel name, which of the
# family's dispatches: "wide" = its longest (the 4096 x 4096 layer), "short" = the others (the 784-wide input layer))
FAMILIES = {
    "forward(dual GEMM + LRT epilogue)": ("EpiFwd", "wide"),
    "accGradParameters(dual GEMM + KL epilogue)": ("EpiDw", "wide"),
    "updateGradInput(dual GEMM + ReLU/dv epilogue)": ("EpiDx", "wide"),
    "layer 1 forward (K = 784, two-pass kernel)": ("EpiFwd", "short"),
    "layer 1 accGradParameters (784 x 4096 gradient)": ("EpiDw", "short"),
}
EXTRA = os.environ.get("VBNN_PMC_EXTRA", "")           # e.g. "TCC_EA0_RDREQ_sum,TCC_EA0_RDREQ_32B_sum;TCC_REQ_sum,TCC_READ_sum"


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    out_dir = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}")
    os.makedirs(out_dir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    res = {k: {} for k in FAMILIES}
    groups = GROUPS + [g.split(",") for g in EXTRA.split(";") if g]
    bench_args = os.environ.get("VBNN_PMC_BENCH_ARGS", "").split()
    for gi, ctrs in enumerate(groups):
        d = os.path.join(out_dir, f"g{gi}")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", *ctrs, "--output-format", "csv", "-d", d, "-o", f"g{gi}", "--",
               "python3", os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline", "--no-box",
               "--no-reporting-config", "--no-deep-config", "--no-train-step", *bench_args]
        print("pass", gi, " ".join(ctrs), flush=True)
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            print("  no counter_collection.csv for this pass", flush=True)
            continue
        rows = list(csv.DictReader(open(files[0])))
        # dispatch id -> (kernel name, duration, {counter: value})
        disp = {}
        for r in rows:
            k = r.get("Dispatch_Id") or r.get("Dispatch_ID")
            e = disp.setdefault(k, {"name": r["Kernel_Name"], "c": {},
                                    "dur": (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3 if "End_Timestamp" in r else 0.0})
            e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for fam, (key, which) in FAMILIES.items():
            ds = [e for e in disp.values() if "gemm_nt_v" in e["name"] and key in e["name"]]
            if not ds:
                continue
            # the widest layer's launches: the longest dispatches of the family (duration, else the first counter)
            metric = (lambda e: e["dur"]) if any(e["dur"] > 0 for e in ds) else (lambda e: e["c"].get(ctrs[0], 0.0))
            top = max(metric(e) for e in ds)
            sel = [e for e in ds if (metric(e) >= 0.8 * top) == (which == "wide")]
            if not sel:
                continue
            for c in ctrs:
                res[fam][c] = sum(e["c"].get(c, 0.0) for e in sel) / len(sel)
            res[fam].setdefault("kernel", sel[0]["name"].split("(")[0][:80])
            if any(e["dur"] > 0 for e in sel):
                res[fam]["duration_us_profiled"] = sum(e["dur"] for e in sel) / len(sel)
    for fam, v in res.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["hbm_bytes"] = int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
        if "TCC_HIT_sum" in v:
            v["l2_hit"] = round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4)
        if "GRBM_GUI_ACTIVE" in v and v.get("duration_us_profiled"):
            v["clock_ghz"] = round(v["GRBM_GUI_ACTIVE"] / 8 / v["duration_us_profiled"] / 1e3, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
            v["mfma_busy"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * v["GRBM_GUI_ACTIVE"] / 8), 4)
        if "SQ_WAIT_ANY" in v and v.get("SQ_WAVE_CYCLES"):
            v["wave_wait_frac"] = round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4)
    res["_note"] = ("tools/collect_traffic.py: rocprofv3 --kernel-trace --pmc <one counter group per pass> on `python3 bench.py "
                    "--steps 2 --warmup 1`; means over the launches of the 4096 x 4096 (N = 4096) dual GEMM of each family "
                    "(that family's longest dispatches); FETCH_SIZE / WRITE_SIZE in KB as reported; hbm_bytes = (2 * FETCH_SIZE + "
                    "WRITE_SIZE) * 1024 -- FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM; mfma_busy = "
                    "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8); l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS)")
    path = os.path.join(ROOT, "gpurun_out", f"{tag}_traffic.json")
    json.dump(res, open(path, "w"), indent=1)
    print("wrote", path)
    for fam, v in res.items():
        if fam != "_note":
            print(fam, {k: v.get(k) for k in ("hbm_bytes", "l2_hit", "mfma_busy", "clock_ghz", "wave_wait_frac", "duration_us_profiled")})


if __name__ == "__main__":
    main()
