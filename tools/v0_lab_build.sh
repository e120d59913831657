#!/bin/bash
# builds the lab's variants into tools/bin:  bash tools/v0_lab_build.sh name "-Dflags" [name "-Dflags" ...]
mkdir -p tools/bin
while [ $# -gt 0 ]; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 $2 tools/v0_lab.hip -o tools/bin/v0_lab_$1 2>&1 | grep -v "argument unused" &
  shift 2
done
wait
ls -la tools/bin
