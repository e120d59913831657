#!/bin/bash
# A/B of library BUILDS in one box (make LIBDIR=../lib/<name> EXTRA=...):  bash tools/ab_lib.sh <tag> <rounds> "<bench args>" <name> <name> ...
# ("." = the default build); the builds are visited round-robin <rounds> times so that drift of the box shows.
tag=$1; rounds=$2; args=$3; shift 3
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for name in "$@"; do
    if [ "$name" = "." ]; then unset VBNN_HIP_LIB; else export VBNN_HIP_LIB=$(pwd)/vbnn_amd/lib/$name/libvbnn_hip.so; fi
    out=gpurun_out/${tag}_${name//./default}_$r.json
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --repeats 3 $args > $out 2> gpurun_out/${tag}_${name//./default}_$r.err || { echo "[$name] failed"; tail -3 gpurun_out/${tag}_${name//./default}_$r.err; exit 1; }
    python3 - "$name" $out <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"[{sys.argv[1]:8s}] ms/step {d['ms_per_step']} repeats {d['config']['repeats_wall_ms']} kernels {d['roofline']['timed_region_kernels_ms']}" + (f" train {d['train_step']['ms_per_train_step']} {d['train_step']['repeats_wall_ms']}" if d.get('train_step') else ""))
PY
  done
done
