#!/bin/bash
# tools/ab_variants.sh <rounds> <variant> [<variant> ...]: alternating runs of bench.py's headline legs with each library of
# tools/variants/<variant>/ swapped in (GPU box only: overwrites the box's scratch copy of gmat_amd/lib/libgmat_hip.so and
# restores the shipped one at the end).  Prints value, roofline.frac, avg launch us, frac_overlapped per run.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
N=$1; shift
cp gmat_amd/lib/libgmat_hip.so /tmp/libgmat_hip_shipped.so
B="python bench.py --no-detail --no-pipeline --no-cpu --no-pmc --steps 30 --warmup 5 $BENCH_EXTRA"
for i in $(seq 1 $N); do
  for v in "$@"; do
    cp tools/variants/$v/libgmat_hip.so gmat_amd/lib/libgmat_hip.so
    echo -n "$v #$i: "; $B 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['frac'], r['avg_launch_us'], r['frac_overlapped'])"
  done
done
cp /tmp/libgmat_hip_shipped.so gmat_amd/lib/libgmat_hip.so
