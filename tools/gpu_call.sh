set -o pipefail
mkdir -p gpurun_out
echo "== tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r05h_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05h_tests.log
echo "== bench"; timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05h_bench_wide_driver.json 2> gpurun_out/r05h_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r05h_bench_wide_driver.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["config"]["repeats_wall_ms"], d["box"]["mfma_clock_ghz"], d["roofline"].get("frac"), d["roofline"].get("frac_at_held_clock"), d["train_step"]["ms_per_train_step"], d["deep_config"]["ms_per_step"])
PY
