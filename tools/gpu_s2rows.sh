#!/bin/bash
# rows-per-segment sweep of scale_yuv2s_kernel (the headline) per batch size; "default" = the launcher's own rule
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-s2rows}; mkdir -p $OUT; cd $R
export X2BENCH_VERIFY=0
for rep in 1 2; do for nf in 1 2 4 8 16 32; do for r in default 3 4 6 8 12 16 20 24 32 45 64; do
  if [ $r = default ]; then unset GMAT_STRIP_ROWS; else export GMAT_STRIP_ROWS=$r; fi
  echo -n "rep=$rep nf=$nf rows=$r us_per_launch " | tee -a $OUT/rows.txt; timeout 60 tools/bin/x2bench $nf 40 "nv12 4K->1080p rgb24 bicubic" | awk '{print $6}' | tee -a $OUT/rows.txt
done; done; done
