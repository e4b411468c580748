#!/bin/bash
# the alternating walk direction in the 4:1 walkers: time and HBM traffic, on and off
mkdir -p gpurun_out/d4ud
timeout 600 python -m pytest tests/test_parity_down4rgb.py tests/test_parity_down4.py -q -m gpu 2>&1 | tail -1 | tee gpurun_out/d4ud/tests.txt
{
for ud in 1 0; do
  for c in "nv12 4K->540p rgb24" "nv12 4K->540p nv12"; do
    echo "== GMAT_STRIP_UPDOWN=$ud  $c"; GMAT_STRIP_UPDOWN=$ud X2BENCH_VERIFY=0 tools/bin/x2bench 32 20 "$c" | grep -v "^#"
    tag=ud${ud}_$(echo "$c" | tr ' >' '__' | tr -d '-')
    GMAT_STRIP_UPDOWN=$ud tools/pmc_case.sh d4ud/$tag "$c" 32 "FETCH_SIZE" "WRITE_SIZE" 2>&1 | tail -3
  done
done
} 2>&1 | tee gpurun_out/d4ud/summary.txt
