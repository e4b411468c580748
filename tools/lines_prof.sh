#!/bin/bash
# tools/lines_prof.sh <nf> — runs ON the GPU box: rocprofv3 kernel stats of the lines form's two passes over the thumb: rows (per-kernel durations)
NF=${1:-32}
cd /tmp && export TMPDIR=/tmp
for CASE in "4K->480x270 rgb24" "4K->320x180 rgb24" "4K->160x90" "1080p->240x136"; do
  rm -rf /tmp/lp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -o t -- $GRAFT_REPO_ROOT/tools/bin/x2bench $NF 10 "$CASE" > /tmp/lp.out 2>&1
  grep "thumb" /tmp/lp.out
  F=$(find /tmp/lp -name "*kernel_stats*" | head -1); [ -n "$F" ] && grep yuvl $F | cut -d, -f1-6
done
