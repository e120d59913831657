#!/bin/bash
# What an LDS-DMA piece costs against other forms of the same transfer (timing only; profiles/r03_piece_lab.txt).
#   build (anywhere hipcc is):  bash tools/piece_lab.sh build      -> tools/bin/gemm_lab_{base,piece1,piece2,piece3}
#   run (on the GPU box):       bash tools/piece_lab.sh
# -DV3_LAB_PIECE=1: no piece at all; 2: a plain 16-byte buffer load into registers, no LDS write; 3: that load + a
# ds_write_b128 of the registers loaded a phase earlier (gemm_v3.h, dma_a_at / dma_b_at).
if [ "$1" = "build" ]; then
  mkdir -p tools/bin
  for v in base piece1 piece2 piece3; do
    f=""; [ $v != base ] && f="-DV3_LAB_PIECE=${v#piece}"
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DLAB_NO_DIAG $f tools/gemm_lab.hip -o tools/bin/gemm_lab_$v 2>&1 | grep -E "error" &
  done
  wait; ls -la tools/bin | grep gemm_lab; exit 0
fi
for r in 1 2; do
  for km in 0 2; do
    for v in base piece1 piece2 piece3; do
      echo "== round $r LAB_KMAJOR=$km $v"
      LAB_KMAJOR=$km timeout -k 10 120 tools/bin/gemm_lab_$v 4096 4096 4096 | grep "v3"
    done
  done
done
