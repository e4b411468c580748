#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE) of the headline kernel, plus a calibration copy of known size.
# Separate --pmc passes, kernel-trace only (no other tracing domains).
TAG=${1:-traffic}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  KBENCH_ONLY="direct nv12 (1" timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/k_$C -o p -- python $R/tools/kbench.py 48 > $OUT/k_$C.log 2>&1
  KBENCH_ONLY="convert 4K" timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/c_$C -o p -- python $R/tools/kbench.py 48 > $OUT/c_$C.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/x_$C -o p -- $R/tools/exp/build/k1bench > $OUT/x_$C.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "at::" in k or "rocclr" in k: continue
        agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in agg.items():
    res[k] = {c: sum(v) / len(v) for c, v in d.items()}
    print(k, {c: round(x, 1) for c, x in res[k].items()})
json.dump(res, open("$OUT/traffic_raw.json", "w"), indent=1)
PY
