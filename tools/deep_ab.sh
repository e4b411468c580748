#!/bin/bash
# tools/deep_ab.sh <launches> — runs ON the GPU box: the "deep:" rows of x2bench (16-bit sources / 10-bit destinations) on the lines form
# (scale_yuvl_h16_kernel) and on the tiled kernel, 32 frames and one frame a launch: the A/B behind the deep clause of gsws.cpp yuvl_eligible()
L=${1:-12}
for NF in 32 1; do
  echo "== lines forced (GMAT_LINES=2), $NF frames a launch";  env GMAT_LINES=2 timeout 300 tools/bin/x2bench $NF $L "deep:" 2>&1
  echo "== tiled forced (GMAT_LINES=0), $NF frames a launch";  env GMAT_LINES=0 timeout 300 tools/bin/x2bench $NF $(( L / 3 + 1 )) "deep:" 2>&1
  echo "== default rule, $NF frames a launch";                  timeout 300 tools/bin/x2bench $NF $L "deep:" 2>&1
done
