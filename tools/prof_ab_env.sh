#!/bin/bash
# kernel-trace timeline of one wide step per setting of an environment switch (lab): tools/prof_ab_env.sh VAR v1 v2 ...
root=$(pwd); var=$1; shift
for v in "$@"; do
  export $var=$v
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_ab_$v -o ab -- python3 $root/bench.py --no-cpu-baseline --no-reporting-config --no-deep-config --no-train-step --no-box --steps 20 --warmup 10 --repeats 2 > /dev/null 2>&1; echo "prof $var=$v rc=$?")
  python3 tools/step_timeline.py gpurun_out/prof_ab_$v/ab_results.db > gpurun_out/ab_${var}_$v.txt 2>&1; cat gpurun_out/ab_${var}_$v.txt
done
