#!/bin/bash
# HBM traffic of the batched launches (32 frames per launch) from PMC counters through tools/bin/x2bench: FETCH_SIZE and
# WRITE_SIZE in separate passes, kernel-trace only beside --pmc (MI355X_MICROARCH.md).  FETCH_SIZE is doubled (gfx950
# reports half the bytes of wide coalesced reads); the same-size converter, whose traffic is known exactly, rides along
# as calibration.  Writes gpurun_out/<tag>/traffic.json in the format bench.py reads from profiles/r02_traffic.json.
TAG=${1:-r02_traffic}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export X2BENCH_VERIFY=0
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o p -- $R/tools/bin/x2bench 32 6 "4K->" > $OUT/$C.log 2>&1
  # the filter kernels run one frame per launch: their own pass, so that their counters are never divided by 32
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/ops_$C -o p -- $R/tools/bin/x2bench 1 6 "op: " > $OUT/ops_$C.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
ops = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True)):
    into = ops if "/ops_" in f else agg
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gmat" not in k: continue
        into[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"frames_per_launch": 32, "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) over tools/bin/x2bench 32 6; "
       "bytes = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB (MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts half of wide coalesced reads)", "kernels": {}}
for k, d in agg.items():
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d: continue
    f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    short = k.split("(")[0].replace("void gmat::", "")
    res["kernels"][short] = {"fetch_size_kib_per_launch": round(f, 1), "write_size_kib_per_launch": round(w, 1),
                             "read_bytes_per_frame": round(2 * f * 1024 / 32), "written_bytes_per_frame": round(w * 1024 / 32),
                             "traffic_bytes_per_frame": round((2 * f + w) * 1024 / 32), "launches": len(d["FETCH_SIZE"])}
    print(short, res["kernels"][short])
res["filter_kernels_one_frame_per_launch"] = {}
for k, d in ops.items():
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d: continue
    f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    short = k.split("(")[0].replace("void gmat::", "")
    res["filter_kernels_one_frame_per_launch"][short] = {"read_bytes_per_launch": round(2 * f * 1024), "written_bytes_per_launch": round(w * 1024),
                                                         "traffic_bytes_per_launch": round((2 * f + w) * 1024), "launches": len(d["FETCH_SIZE"])}
    print(short, res["filter_kernels_one_frame_per_launch"][short])
json.dump(res, open("$OUT/traffic.json", "w"), indent=1)
PY
