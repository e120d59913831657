#!/bin/bash
# A/B of bench.py variants in one box:  bash tools/ab_bench.sh <tag> "<args of variant 1>" "<args of variant 2>" ...
tag=$1; shift
mkdir -p gpurun_out
i=0
for v in "$@"; do
  i=$((i+1))
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --repeats 3 $v > gpurun_out/${tag}_ab$i.json 2> gpurun_out/${tag}_ab$i.err
  python3 - "$v" gpurun_out/${tag}_ab$i.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"[{sys.argv[1]}] ms/step {d['ms_per_step']} repeats {d['config']['repeats_wall_ms']} kernels {d['roofline']['timed_region_kernels_ms']}")
except Exception as e:
    print(f"[{sys.argv[1]}] unreadable: {e}")
PY
done
