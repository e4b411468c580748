#!/bin/bash
# tools/trace_gaps.sh <nf> <launches> <case> — runs ON the GPU box: rocprofv3 --kernel-trace of ONE x2bench case; prints per kernel the median duration and
# the median gap from the previous dispatch's end to this one's start (where a call's time goes between its launches)
NF=${1:-1}; L=${2:-50}; CASE=$3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tg; rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o t -- $GRAFT_REPO_ROOT/tools/bin/x2bench $NF $L "$CASE" > /tmp/tg.out 2>&1
grep -E "us/frame" /tmp/tg.out | cut -c1-160
F=$(find /tmp/tg -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, statistics, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
prev = None
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-40:]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n].append(e - s)
    if prev is not None: gap[n].append(s - prev)
    prev = e
for n in dur:
    if len(dur[n]) < 8: continue
    d = sorted(dur[n]); g = sorted(gap[n]) or [0]
    print("%-42s n %5d  duration p50 %7.2f us (p10 %.2f)   gap before p50 %6.2f us (p10 %.2f)" % (n, len(d), d[len(d)//2]/1e3, d[len(d)//10]/1e3, g[len(g)//2]/1e3, g[len(g)//10]/1e3))
PY
