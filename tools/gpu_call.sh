set -o pipefail
mkdir -p gpurun_out
echo "== head tests"; timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py tests/test_c_host.py -m gpu -x -q -k "streaming_head or head_logits or full_size_wide_step or wide_training_steps or c_host" > gpurun_out/r05i_tests.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r05i_tests.log
echo "== head timing"; for r in 1 2; do VBNN_HEAD_STREAM=0 timeout -k 10 120 python3 tools/time_head.py tile; VBNN_HEAD_INLINE_FINISH=0 timeout -k 10 120 python3 tools/time_head.py stream+finish-kernel; timeout -k 10 120 python3 tools/time_head.py stream+inline-finish; done 2>&1 | grep -v amdgpu.ids
