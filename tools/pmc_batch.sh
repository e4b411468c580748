#!/bin/bash
# PMC passes over the batched headline launch (kbench_ops "rgb24, 16 frames/launch"); separate passes per group.
# usage: tools/pmc_batch.sh <tag> [lib]
TAG=$1; LIBP=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "$LIBP" ] && export KBENCH_LIB=$LIBP
export KBENCH_NF=16
i=0
for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pmc$i -o p -- python $R/tools/kbench_ops.py 2 "rgb24, 16 frames/launch" > $OUT/pmc$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gmat" not in k: continue
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
