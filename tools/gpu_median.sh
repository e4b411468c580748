#!/bin/bash
mkdir -p gpurun_out/median
{
echo "== strip"; tools/bin/x2bench 1 20 "op: median" | grep -v "^#"
for r in 8 12 16 24 32 48 64 96; do echo "== strip, GMAT_STRIP_ROWS=$r"; GMAT_STRIP_ROWS=$r tools/bin/x2bench 1 20 "op: median" | grep -v "^#"; done
echo "== bytes (GMAT_NO_MEDIAN_STRIP=1)"; GMAT_NO_MEDIAN_STRIP=1 tools/bin/x2bench 1 20 "op: median" | grep -v "^#"
} 2>&1 | tee gpurun_out/median/x2.txt
timeout 900 python -m pytest tests/test_parity_filters.py -q -m gpu -x -k median 2>&1 | tail -2 | tee gpurun_out/median/tests.txt
