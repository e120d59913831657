#!/bin/bash
# what an LDS-DMA piece costs against other forms of the same transfer (tools/gemm_lab.hip built with -DV3_LAB_PIECE=1|2|3; timing only)
for r in 1 2; do
  for km in 0 2; do
    for v in base piece1 piece2 piece3; do
      echo "== round $r LAB_KMAJOR=$km $v"
      LAB_KMAJOR=$km timeout -k 10 120 tools/bin/gemm_lab_$v 4096 4096 4096 | grep "v3"
    done
  done
done
