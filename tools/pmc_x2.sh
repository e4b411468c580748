#!/bin/bash
# PMC passes over the batched headline launch through tools/bin/x2bench (no Python): SQ counters in three groups, then
# FETCH_SIZE and WRITE_SIZE in their own passes (MI355X_MICROARCH.md: TCC slots; kernel-trace only beside --pmc).
# usage: tools/pmc_x2.sh <tag> [case substring]   (environment variables such as GMAT_STRIP_ROWS pass through)
TAG=$1; CASE=${2:-"nv12 4K->1080p rgb24 bicubic"}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export X2BENCH_VERIFY=0
i=0
for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pmc$i -o p -- $R/tools/bin/x2bench 32 6 "$CASE" > $OUT/pmc$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gmat" not in k: continue
        agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/pmc_summary.txt", "w") as o:
    for k, d in agg.items():
        print(k); o.write(k + "\n")
        for c, v in sorted(d.items()):
            line = f"   {c:30s} mean {sum(v)/len(v):16.1f}  n={len(v)}"
            print(line); o.write(line + "\n")
PY
