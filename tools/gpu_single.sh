#!/bin/bash
# one frame per call: the late walkers against the generic kernel they replace (a per-call user must not lose)
mkdir -p gpurun_out/single
for c in "nv12 4K->720p rgb24" "nv12 1080p->720p rgb24" "nv12 4K->540p rgb24" "nv12 4K->540p nv12" "rgb24 4K->1080p nv12" "land: nv12 4K->1440p rgb24"; do
  for ns in 0 1; do
    echo "== $c  GMAT_SCALE_NO_STRIP=$ns"; GMAT_SCALE_NO_STRIP=$ns X2BENCH_VERIFY=0 tools/bin/x2bench 1 30 "$c" | grep -v "^#"
  done
done 2>&1 | tee gpurun_out/single/x2.txt
