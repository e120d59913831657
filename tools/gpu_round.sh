#!/bin/bash
# One gpurun call's worth of checks ON the GPU box (from the repository root): bash tools/gpu_round.sh <tag>
set -o pipefail
tag=${1:-r02}
root=$(pwd)
mkdir -p gpurun_out
echo "== tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${tag}_tests.log
echo "== bench"; timeout -k 10 300 python3 bench.py --with-update > gpurun_out/${tag}_bench_wide.json 2> gpurun_out/${tag}_bench.err; echo "rc=$?"
timeout -k 10 300 python3 bench.py --no-cpu-baseline --prepare-each-step > gpurun_out/${tag}_bench_wide_prep.json 2>> gpurun_out/${tag}_bench.err; echo "rc=$?"
timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/${tag}_bench_wide_driver.json 2>> gpurun_out/${tag}_bench.err; echo "rc=$?"
VBNN_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/${tag}_bench_dist1.json 2>> gpurun_out/${tag}_bench.err; echo "dist1 rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_${tag} -o ${tag} -- python3 $root/bench.py --no-cpu-baseline --steps 20 --warmup 10 --repeats 1 > /dev/null 2>&1; echo "prof rc=$?"
cd $root
python3 profiles/summarize_db.py gpurun_out/prof_${tag}/${tag}_results.db 50 > gpurun_out/${tag}_wide_kernel_stats.txt 2>&1
head -16 gpurun_out/${tag}_wide_kernel_stats.txt
python3 - <<PY
import json
for f in ("bench_wide", "bench_wide_prep", "bench_wide_driver", "bench_dist1"):
    try:
        d = json.loads(open("gpurun_out/${tag}_%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["config"]["repeats_wall_ms"], d["roofline"]["timed_region_kernels_ms"], d["config"].get("train"), d.get("comm"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
tail -5 gpurun_out/${tag}_bench.err
