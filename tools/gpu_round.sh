#!/bin/bash
# One gpurun call's worth of the round's measurements ON the GPU box (from the repository root): bash tools/gpu_round.sh <tag>
# tests, the bench lines (wide with --with-update, the driver's protocol, small, deep, one-rank RCCL), the rocprofv3 kernel
# trace of the wide bench with one step's timeline, and the PMC passes (tools/collect_traffic.py). Copy what is to be kept
# from gpurun_out/ into profiles/.
set -o pipefail
tag=${1:-r02}
root=$(pwd)
mkdir -p gpurun_out
echo "== tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${tag}_tests.log
cp gpurun_out/test_durations.txt gpurun_out/${tag}_test_durations.txt 2>/dev/null   # (tests/conftest.py: torch import time, tests > 1 s, every child process waited for)
echo "== bit-for-bit repeatability of the training step (tools/step_bits.py: the race screen of the pipelined kernels at bench size)"
timeout -k 10 300 python3 tools/step_bits.py wide 10 2 > gpurun_out/${tag}_step_bits_wide.txt 2>&1; tail -1 gpurun_out/${tag}_step_bits_wide.txt
echo "== bench"
timeout -k 10 400 python3 bench.py > gpurun_out/${tag}_bench_wide.json 2> gpurun_out/${tag}_bench.err; echo "wide rc=$?"
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_wide_driver.json 2>> gpurun_out/${tag}_bench.err; echo "driver-protocol rc=$?"
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-reporting-config --no-deep-config --no-train-step --prepare-each-step > gpurun_out/${tag}_bench_wide_prep.json 2>> gpurun_out/${tag}_bench.err; echo "prepare-each-step rc=$?"
timeout -k 10 300 python3 bench.py --config small > gpurun_out/${tag}_bench_small.json 2>> gpurun_out/${tag}_bench.err; echo "small rc=$?"
timeout -k 10 300 python3 bench.py --config small --S 30 --batch 1 --stack-draws --no-cpu-baseline > gpurun_out/${tag}_bench_small_S30.json 2>> gpurun_out/${tag}_bench.err; echo "small S30 rc=$?"
timeout -k 10 300 python3 bench.py --config small --graph on --no-cpu-baseline > gpurun_out/${tag}_bench_small_graph.json 2>> gpurun_out/${tag}_bench.err; echo "small graph rc=$?"
timeout -k 10 300 python3 bench.py --config small --S 30 --batch 1 --stack-draws --graph on --no-cpu-baseline > gpurun_out/${tag}_bench_small_S30_graph.json 2>> gpurun_out/${tag}_bench.err; echo "small S30 graph rc=$?"
timeout -k 10 400 python3 bench.py --config deep --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/${tag}_bench_deep.json 2>> gpurun_out/${tag}_bench.err; echo "deep rc=$?"
VBNN_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-deep-config --steps 20 --warmup 5 > gpurun_out/${tag}_bench_dist1.json 2>> gpurun_out/${tag}_bench.err; echo "dist1 rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_${tag} -o ${tag} -- python3 $root/bench.py --no-cpu-baseline --no-reporting-config --no-deep-config --no-train-step --no-box --steps 20 --warmup 10 --repeats 2 > /dev/null 2>&1; echo "prof rc=$?"
cd $root
python3 profiles/summarize_db.py gpurun_out/prof_${tag}/${tag}_results.db 70 > gpurun_out/${tag}_wide_kernel_stats.txt 2>&1
python3 tools/step_timeline.py gpurun_out/prof_${tag}/${tag}_results.db > gpurun_out/${tag}_wide_step_timeline.txt 2>&1
head -14 gpurun_out/${tag}_wide_kernel_stats.txt; cat gpurun_out/${tag}_wide_step_timeline.txt
# the update sweep (VBLinear:update, excluded from the metric): its kernel rows
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_${tag}u -o ${tag}u -- python3 $root/bench.py --no-cpu-baseline --no-reporting-config --no-deep-config --no-box --steps 10 --warmup 5 --repeats 1 --with-update > /dev/null 2>&1; echo "prof update rc=$?"
cd $root
python3 profiles/summarize_db.py gpurun_out/prof_${tag}u/${tag}u_results.db 70 | grep -i "update\|total ms" > gpurun_out/${tag}_update_kernel_stats.txt 2>&1; cat gpurun_out/${tag}_update_kernel_stats.txt
# the fp32 configuration: kernel stats and one step's timeline
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_${tag}s -o ${tag}s -- python3 $root/bench.py --config small --no-cpu-baseline --steps 50 --warmup 10 --repeats 2 > /dev/null 2>&1; echo "prof small rc=$?"
cd $root
python3 profiles/summarize_db.py gpurun_out/prof_${tag}s/${tag}s_results.db 30 > gpurun_out/${tag}_small_kernel_stats.txt 2>&1
python3 tools/step_timeline.py gpurun_out/prof_${tag}s/${tag}s_results.db 70 EpiFwd 2 > gpurun_out/${tag}_small_step_timeline.txt 2>&1; cat gpurun_out/${tag}_small_step_timeline.txt
VBNN_PMC_EXTRA="TCC_EA0_RDREQ_sum,TCC_EA0_RDREQ_32B_sum,TCC_EA0_RDREQ_DRAM_sum;TCC_REQ_sum,TCC_READ_sum,TCC_WRITE_sum" timeout -k 10 900 python3 tools/collect_traffic.py ${tag} > gpurun_out/${tag}_traffic.log 2>&1; tail -6 gpurun_out/${tag}_traffic.log
python3 - <<PY
import json
for f in ("bench_wide", "bench_wide_driver", "bench_wide_prep", "bench_small", "bench_small_S30", "bench_small_graph", "bench_small_S30_graph", "bench_deep", "bench_dist1"):
    try:
        d = json.loads(open("gpurun_out/${tag}_%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d["config"]["repeats_wall_ms"], d["roofline"]["frac"], d["roofline"]["timed_region_kernels_ms"], d["config"].get("train", {}).get("ms_per_train_step"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
tail -3 gpurun_out/${tag}_bench.err
# keep what is merged back small: the databases and counter CSVs stay on the box
rm -rf gpurun_out/prof_${tag} gpurun_out/prof_${tag}u gpurun_out/prof_${tag}s gpurun_out/pmc_${tag}
