set -o pipefail
mkdir -p gpurun_out
echo "== tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r05e_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r05e_tests.log
echo "== A/B sweep order / gradient stores"; bash tools/ab_lib.sh r05ntg 2 "--steps 20 --warmup 5 --no-reporting-config --no-deep-config --no-box" . rev ntg0 ntg0r 2>&1 | tail -10
