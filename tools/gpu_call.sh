#!/bin/bash
# One short gpurun call (from the repository root ON the GPU box): the GPU suite, the overlap stand-in, the driver-protocol bench line.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/gpu_call.sh <tag>'      (tools/gpu_round.sh is the full profile set)
set -o pipefail
tag=${1:-call}
mkdir -p gpurun_out
echo "== tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${tag}_tests.log
echo "== stand-in"; timeout -k 10 400 python3 tools/overlap_standin.py gpurun_out/${tag}_overlap_standin.json > gpurun_out/${tag}_standin.log 2>&1; echo "standin rc=$?"; tail -3 gpurun_out/${tag}_standin.log
echo "== bench"; timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_wide_driver.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
python3 - "$tag" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/{sys.argv[1]}_bench_wide_driver.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d.get("box"), d["roofline"].get("frac"), d["roofline"].get("frac_at_held_clock"), d["train_step"]["ms_per_train_step"], d["deep_config"]["ms_per_step"])
PY
