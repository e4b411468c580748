#!/bin/bash
# rows-per-segment sweep of scale_yuv2p_kernel per batch size; "-" = the launcher's own rule
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-p2rows}; mkdir -p $OUT; cd $R
python3 tools/sweep.py "nv12 4K->1080p nv12" --nf 1,2,4,8,16,32 --env GMAT_STRIP_ROWS=-,2,3,4,6,8,12,16,24,32 --out $OUT/rows.txt
