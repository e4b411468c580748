#!/bin/bash
# rows-per-segment sweep of every batched strip kernel (32 frames per launch) on ONE box: does the headline's finding — short
# segments in raster order beat long column walks — carry over?  Output: one line per (case, rows).  usage: tools/sweep_rows_all.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for c in "nv12 4K->1080p nv12 bicubic" "yuv420p 4K->1080p yuv420p bicubic" "nv12 4K->1080p yuv420p bicubic" "rgb24 4K->1080p rgb24 bicubic" \
         "nv12 4K->720p nv12 bicubic (3:1)" "nv12 4K->720p rgb24 bicubic (3:1)" "nv12 4K->540p nv12 bicubic (4:1)" "nv12 4K->540p rgb24 bicubic (4:1)" \
         "nv12 1080p->720p nv12 bicubic (3:2)" "nv12 1080p->720p rgb24 bicubic (3:2)" "rgb24 4K->1080p nv12 bicubic (scale)" "rgb24 4K->4K nv12 (convert)" \
         "nv12 1080p->4K nv12 bicubic (up)" "p010 4K->1080p p010 bicubic" "nv12 1080p->540p rgb24 bicubic" "any: nv12 4K->1600x900 rgb24 bicubic" "any: nv12 4K->1600x900 nv12 bicubic"; do
  echo "== $c"
  python3 tools/sweep.py "$c" --nf 32 --env GMAT_STRIP_ROWS=-,6,8,9,10,11,12,14,16,24 --reps ${REPS:-2} --launches 20 2>&1 | sed 's/^/   /'
done
