#!/bin/bash
# rows-per-segment sweep of scale_yuv2p_kernel per batch size (verify off: parity is tools/gpu_p2.sh's job)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-p2rows}; mkdir -p $OUT; cd $R
export X2BENCH_VERIFY=0
for nf in 1 2 4 8 16 32; do for r in 2 3 4 6 8 12 16 24 32; do
  echo -n "nf=$nf rows=$r  " | tee -a $OUT/rows.txt; GMAT_STRIP_ROWS=$r timeout 60 tools/bin/x2bench $nf 40 "nv12 4K->1080p nv12" | awk '{print "us_per_frame", $7, "GBps", $9}' | tee -a $OUT/rows.txt
done; done
