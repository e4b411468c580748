#!/bin/bash
# plane transposes: tile-size A/B of the 1- and 2-byte sample paths
mkdir -p gpurun_out/tr
for t in 128 64; do
  echo "== GMAT_TRANSPOSE_TILE=$t"; GMAT_TRANSPOSE_TILE=$t tools/bin/x2bench 1 20 "op: transpose" 2>&1 | grep -v "^#"
done | tee gpurun_out/tr/tiles.txt
echo "== default"; tools/bin/x2bench 1 20 "op:" 2>&1 | tee gpurun_out/tr/ops.txt
timeout 900 python -m pytest tests/test_parity_filters.py -q -m gpu -x 2>&1 | tail -2 | tee gpurun_out/tr/tests.txt
