#!/bin/bash
# kernel trace of the small configuration's step, once per argument set:  [MARKER=k_sample_advance NTH=1] bash tools/small_prof.sh <tag> "<bench args>" ...
set -o pipefail
root=$(pwd); tag=$1; shift
i=0
for v in "$@"; do
  i=$((i+1))
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_${tag}$i -o p -- python3 $root/bench.py --config small --no-cpu-baseline --steps 50 --warmup 10 --repeats 2 $v > /dev/null 2>&1; echo "prof rc=$?"
  cd $root
  echo "== $v"
  python3 tools/step_timeline.py gpurun_out/prof_${tag}$i/p_results.db 70 ${MARKER:-EpiFwd} ${NTH:-2} | tee gpurun_out/${tag}${i}_timeline.txt
done
