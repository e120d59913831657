#!/bin/bash
# Regenerate the round's measurement artifacts ON the GPU box (from the repository root):
#   bash tools/refresh_profiles.sh r01        -> gpurun_out/<tag>_*  (copy the ones to keep into profiles/)
set -e -o pipefail
tag=${1:-r01}
root=$(pwd)
mkdir -p gpurun_out
python3 bench.py > gpurun_out/${tag}_bench_wide.json 2> gpurun_out/${tag}_bench.err
python3 bench.py --config small > gpurun_out/${tag}_bench_small.json 2>> gpurun_out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_${tag} -o ${tag} -- python3 $root/bench.py --no-cpu-baseline --steps 20 --warmup 10 > /dev/null 2>&1 || true
cd $root
python3 profiles/summarize_db.py gpurun_out/prof_${tag}/${tag}_results.db 30 > gpurun_out/${tag}_wide_kernel_stats.txt
python3 tools/collect_traffic.py ${tag} > gpurun_out/${tag}_traffic.log 2>&1
tail -4 gpurun_out/${tag}_traffic.log
