#!/bin/bash
mkdir -p gpurun_out/clamp
timeout 600 python -m pytest tests/test_parity_down3rgb.py tests/test_parity_down32rgb.py -q -m gpu 2>&1 | tail -1 | tee gpurun_out/clamp/tests.txt
{
for c in "nv12 4K->720p rgb24" "nv12 1080p->720p rgb24"; do
  echo "== $c"; X2BENCH_VERIFY=0 tools/bin/x2bench 32 20 "$c" | grep -v "^#"
  tag=$(echo "$c" | tr ' >' '__' | tr -d '-')
  tools/pmc_case.sh clamp/$tag "$c" 32 "FETCH_SIZE" 2>&1 | tail -2
done
} 2>&1 | tee gpurun_out/clamp/summary.txt
