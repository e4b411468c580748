#!/bin/bash
# HBM-traffic counters (FETCH_SIZE, WRITE_SIZE in separate passes) of the late round-2 walkers, 32 frames per launch
mkdir -p gpurun_out/pmc_late
for c in "nv12 4K->720p rgb24" "nv12 1080p->720p rgb24" "nv12 4K->540p rgb24" "nv12 4K->540p nv12" "nv12 4K->720p nv12"; do
  tag=$(echo "$c" | tr ' >' '__' | tr -d '-')
  tools/pmc_case.sh pmc_late/$tag "$c" 32 "FETCH_SIZE" "WRITE_SIZE" 2>&1 | tail -4
done | tee gpurun_out/pmc_late/summary.txt
