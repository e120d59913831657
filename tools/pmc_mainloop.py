#!/usr/bin/env python3
"""Where do the cycles of the bare pipelined main loops go? PMC passes (rocprofv3 --pmc, one group per pass, --kernel-trace
only) over tools/bin/gemm_lab_nd -- the gemm_nt_v2 / gemm_nt_v3 main loops (tools/lab/: the headers as of round 3; the main loops are the shipped ones) with a trivial epilogue on random
bf16 operands at 4096^3 -- summarised per kernel: every counter as its mean per dispatch, and the derived shares the
round-3 question needs (is 1.5 PFLOP/s the ceiling of a dual-GEMM tile on this CU, and why?).
ON the GPU box, from the repository root:   python3 tools/pmc_mainloop.py r03   -> gpurun_out/<tag>_mainloop_pmc.json"""
import csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"],
    ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_SALU"],
    ["SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_VALU"],
    ["SQ_INST_CYCLES_VMEM_RD", "SQ_INSTS_VMEM_RD", "SQ_INST_LEVEL_VMEM", "SQ_ACTIVE_INST_MISC"],
    ["SQ_WAIT_INST_LDS", "SQ_INSTS_LDS", "SQ_INST_LEVEL_LDS", "SQ_LDS_IDX_ACTIVE"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL"],
    ["SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VMEM_TA_CMD_FIFO_FULL", "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_SALU"],
    ["SQ_IFETCH", "SQ_IFETCH_LEVEL", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"],
    ["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_CYCLES", "SQ_WAVES"],
]
KERNELS = {  # label -> substrings that must all be in the (demangled) kernel name
    "gemm_nt_v3 dual (two passes, 256x256)": ["gemm_nt_v3<true, false, false"],
    "gemm_nt_v3 single (one pass, 256x256)": ["gemm_nt_v3<false, false, false"],
    "gemm_nt_v2 dual sched 2 (256x128, two accumulators)": ["gemm_nt_v2<true, 2, 4, 3"],
}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r03"
    exe = os.path.join(ROOT, "tools", "bin", "gemm_lab_nd")
    out_dir = os.path.join(ROOT, "gpurun_out", f"pmc_loop_{tag}")
    os.makedirs(out_dir, exist_ok=True)
    res = {k: {} for k in KERNELS}
    for gi, ctrs in enumerate(GROUPS):
        d = os.path.join(out_dir, f"g{gi}")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", *ctrs, "--output-format", "csv", "-d", d, "-o", f"g{gi}", "--", exe, "4096", "4096", "4096"]
        print("pass", gi, " ".join(ctrs), flush=True)
        if "--summarise" not in sys.argv:                     # (re-read the CSVs of an earlier collection)
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            print("  no counter_collection.csv for this pass", flush=True)
            continue
        acc = {}
        for r in csv.DictReader(open(files[0])):
            for label, subs in KERNELS.items():
                if all(s in r["Kernel_Name"] for s in subs):
                    a = acc.setdefault((label, r["Counter_Name"]), [0.0, set()])
                    a[0] += float(r["Counter_Value"])
                    a[1].add(r.get("Dispatch_Id") or r.get("Dispatch_ID"))
        for (label, c), (tot, disp) in acc.items():
            res[label][c] = tot / max(len(disp), 1)
    for label, c in res.items():
        d = {}
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                      "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_SALU", "SQ_LDS_DATA_FIFO_FULL",
                      "SQ_LDS_CMD_FIFO_FULL", "SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VMEM_TA_CMD_FIFO_FULL"):
                if k in c:
                    d[k + " / SQ_WAVE_CYCLES"] = round(c[k] / wc, 4)
        if c.get("SQ_BUSY_CU_CYCLES") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            d["MFMA busy share of CU-busy cycles (x4 SIMDs)"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CU_CYCLES"], 4)
        if c.get("SQ_INSTS_VMEM_RD") and c.get("SQ_INST_CYCLES_VMEM_RD"):
            d["issue cycles per VMEM read (LDS-DMA piece)"] = round(c["SQ_INST_CYCLES_VMEM_RD"] / c["SQ_INSTS_VMEM_RD"], 1)
        if c.get("SQ_INSTS_LDS") and c.get("SQ_ACTIVE_INST_LDS"):
            d["active cycles per LDS instruction"] = round(c["SQ_ACTIVE_INST_LDS"] / c["SQ_INSTS_LDS"], 1)
        if c.get("SQ_INSTS_MFMA") and c.get("SQ_INSTS_VMEM_RD"):
            d["MFMA per LDS-DMA piece"] = round(c["SQ_INSTS_MFMA"] / c["SQ_INSTS_VMEM_RD"], 2)
            d["MFMA per LDS instruction"] = round(c["SQ_INSTS_MFMA"] / max(c.get("SQ_INSTS_LDS", 0), 1), 2)
        c["derived"] = d
    path = os.path.join(ROOT, "gpurun_out", f"{tag}_mainloop_pmc.json")
    json.dump(res, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
