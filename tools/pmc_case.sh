#!/bin/bash
# counters of ONE x2bench case: tools/pmc_case.sh <tag> "<case filter>" <frames per launch> COUNTERS_PASS_1 [COUNTERS_PASS_2 ...]
# (a pass = a space-separated counter list in quotes; kernel-trace only beside --pmc, MI355X_MICROARCH.md).  Prints per-kernel averages.
TAG=$1; CASE=$2; NF=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export X2BENCH_VERIFY=0
i=0
for C in "$@"; do
  i=$((i + 1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$i -o p -- $R/tools/bin/x2bench $NF 6 "$CASE" > $OUT/pass$i.log 2>&1 || tail -3 $OUT/pass$i.log
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "gmat" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void gmat::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %16.1f  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
