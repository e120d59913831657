set -o pipefail
mkdir -p gpurun_out
echo "== tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r05d_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r05d_tests.log
echo "== standin"; timeout -k 10 400 python3 tools/overlap_standin.py gpurun_out/r05_overlap_standin.json > gpurun_out/r05d_standin.log 2>&1; echo "standin rc=$?"; tail -4 gpurun_out/r05d_standin.log
echo "== bench"; timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05d_bench_wide_driver.json 2> gpurun_out/r05d_bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r05d_bench.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r05d_bench_wide_driver.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d.get("box"), d["roofline"].get("frac"), d["roofline"].get("frac_at_held_clock"), d["train_step"]["ms_per_train_step"], d["deep_config"]["ms_per_step"], d["reporting_config"]["batch256"]["ms_per_step"])
PY
echo "== nostride A/B"; bash tools/ab_lib.sh r05ns 2 "--steps 20 --warmup 5 --no-reporting-config --no-deep-config --no-train-step --no-box" . nostride 2>&1 | tail -6
echo "== dist1"; VBNN_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-deep-config --no-reporting-config --steps 20 --warmup 5 > gpurun_out/r05d_bench_dist1.json 2>> gpurun_out/r05d_bench.err; echo "dist1 rc=$?"
VBNN_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --exchange p2p --no-cpu-baseline --no-deep-config --no-reporting-config --steps 20 --warmup 5 > gpurun_out/r05d_bench_dist1_p2p.json 2>> gpurun_out/r05d_bench.err; echo "dist1 p2p rc=$?"
python3 - <<'PY'
import json
for f in ("r05d_bench_dist1", "r05d_bench_dist1_p2p"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["comm"]["backend"], d["comm"].get("step_without_exchange"))
    except Exception as e:
        print(f, "unreadable", e)
PY
