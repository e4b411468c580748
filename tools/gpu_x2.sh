#!/bin/bash
# Runs on the GPU box: instruction-rate microbench + strip-kernel sweep (tuning aid).  Usage: tools/gpu_x2.sh <tag>
TAG=${1:-r02a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx" | head -4 > $OUT/rocminfo.txt
timeout 120 tools/bin/ubench > $OUT/ubench.txt 2>&1
echo "== strip kernel, default segment rows" | tee $OUT/x2bench.txt
timeout 120 tools/bin/x2bench 32 40 2>&1 | tee -a $OUT/x2bench.txt
echo "== tiled kernel (GMAT_SCALE_NO_STRIP=1)" | tee -a $OUT/x2bench.txt
GMAT_SCALE_NO_STRIP=1 X2BENCH_VERIFY=0 timeout 120 tools/bin/x2bench 32 40 "rgb24 bicubic" 2>&1 | tee -a $OUT/x2bench.txt
for rows in 12 16 20 27 36 45 54 72 135; do
  echo "== GMAT_STRIP_ROWS=$rows" | tee -a $OUT/x2bench.txt
  GMAT_STRIP_ROWS=$rows X2BENCH_VERIFY=0 timeout 120 tools/bin/x2bench 32 40 "nv12 4K->1080p rgb24 bicubic" 2>&1 | tee -a $OUT/x2bench.txt
done
echo "== one frame per launch: tiled vs strip" | tee -a $OUT/x2bench.txt
X2BENCH_VERIFY=0 timeout 120 tools/bin/x2bench 1 400 "nv12 4K->1080p rgb24 bicubic" 2>&1 | tee -a $OUT/x2bench.txt
for rows in 4 8 12 16; do
GMAT_STRIP_SINGLE=1 GMAT_STRIP_ROWS=$rows X2BENCH_VERIFY=0 timeout 120 tools/bin/x2bench 1 400 "nv12 4K->1080p rgb24 bicubic" 2>&1 | tee -a $OUT/x2bench.txt
done
echo "== 8 frames per launch" | tee -a $OUT/x2bench.txt
X2BENCH_VERIFY=0 timeout 120 tools/bin/x2bench 8 100 "nv12 4K->1080p rgb24 bicubic" 2>&1 | tee -a $OUT/x2bench.txt
