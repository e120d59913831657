#!/bin/bash
# rocprofv3 kernel trace of bench.py and one steady-state step's timeline: bash tools/prof_step.sh <tag> [bench args]
tag=$1; shift
root=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_${tag} -o ${tag} -- python3 $root/bench.py --no-cpu-baseline --steps 20 --warmup 10 --repeats 2 "$@" > $root/gpurun_out/prof_${tag}.json 2> $root/gpurun_out/prof_${tag}.err
echo "prof rc=$?"
cd $root
python3 tools/step_timeline.py gpurun_out/prof_${tag}/${tag}_results.db | tee gpurun_out/${tag}_timeline.txt
