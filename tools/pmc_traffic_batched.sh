#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, kernel-trace only) of the batched headline launch
# (32 frames per launch) and, as calibration, of the batched same-size converter whose traffic is known
# (12 441 600 B read + 24 883 200 B written per 4K frame).
# usage: tools/pmc_traffic_batched.sh <tag>
TAG=${1:-traffic_b}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export KBENCH_NF=32
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/s_$C -o p -- python $R/tools/kbench_ops.py 2 "1080p rgb24, 16 frames/launch" > $OUT/s_$C.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/c_$C -o p -- python $R/tools/kbench_ops.py 2 "rgb24 (convert), 16 frames/launch" > $OUT/c_$C.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gmat" not in k: continue
        agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in agg.items():
    res[k] = {c: sum(v) / len(v) for c, v in d.items()}
    res[k]["launches"] = {c: len(v) for c, v in d.items()}
    print(k, {c: (round(x, 1) if not isinstance(x, dict) else x) for c, x in res[k].items()})
json.dump(res, open("$OUT/traffic_raw.json", "w"), indent=1)
PY
