set -o pipefail
mkdir -p gpurun_out
echo "== tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/r05a_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r05a_tests.log
echo "== standin"; timeout -k 10 400 python3 tools/overlap_standin.py gpurun_out/r05_overlap_standin.json > gpurun_out/r05a_standin.log 2>&1; echo "standin rc=$?"; tail -3 gpurun_out/r05a_standin.log
echo "== bench"; timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05a_bench_wide_driver.json 2> gpurun_out/r05a_bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r05a_bench.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r05a_bench_wide_driver.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d.get("box"), d["roofline"].get("frac"), d["roofline"].get("frac_at_held_clock"), d["train_step"]["ms_per_train_step"])
PY
