set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_parity_gpu.py tests/test_dist_gpu.py -m gpu -x -q -k "box_calibration or stand_in or p2p" > gpurun_out/r05j_tests.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/r05j_tests.log
