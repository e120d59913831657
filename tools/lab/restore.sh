#!/bin/bash
# tools/lab/restore.sh -- puts the FROZEN kernel headers the lab programs include back into tools/lab/ from this repository's
# own history (they are byte-for-byte files of earlier commits, so the tree does not carry 4.4 k lines of copies):
#   common.h epilogues.h gemm_v2.h gemm_v3.h   = vbnn_amd/csrc/<name> of commit becd29e (end of round 3, lab hooks still inside)
#   gemm_v3_r04.h                              = tools/lab/gemm_v3_r04.h of commit 9087b77 (middle of round 4: every K-step form priced)
# Run it HERE (it needs .git) before building tools/gemm_lab.hip / split_lab.hip / piece_lab.sh; the restored files are git-ignored
# and travel to the GPU box with the snapshot like any other built file.
set -e
cd "$(dirname "$0")/../.."
for f in common.h epilogues.h gemm_v2.h gemm_v3.h; do git show becd29e:vbnn_amd/csrc/$f > tools/lab/$f; done
git show 9087b77:tools/lab/gemm_v3_r04.h > tools/lab/gemm_v3_r04.h
ls -la tools/lab/*.h
