#!/bin/bash
# How often do two ranks SHARING one GPU compute a different forward? (diagnostic for tests/test_dist_gpu.py's rare failure)
#   bash tools/flake_probe.sh <runs> [env assignments for the workers: VBNN_TEST_CONCURRENT=1 (the ranks overlap on the GPU), VBNN_TEST_DIAG=1, VBNN_TEST_DUMP=<dir>]
runs=${1:-20}; shift
mkdir -p gpurun_out/flake
for i in $(seq 1 $runs); do
  env "$@" HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 timeout -k 10 120 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port $((29600 + i)) \
    tests/_dist_gpu_worker.py /tmp/flake_r bf16 4096,4096 784 1024 2>/dev/null | grep "local loss\|diag" | sort | sed 's/ (exchange.*//' | tr '\n' ' '
  echo
done | sort | uniq -c | sort -rn
