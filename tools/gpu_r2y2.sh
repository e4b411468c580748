#!/bin/bash
# scale_rgb2y_kernel: parity on the GPU, then timing against the generic kernel and a segment-length sweep
mkdir -p gpurun_out/r2y2
timeout 900 python -m pytest tests/test_parity_rgb2y.py -q -m gpu -x 2>&1 | tail -3 | tee gpurun_out/r2y2/tests.txt
C="rgb24 4K->1080p nv12"
{
for nf in 32 8 1; do
  echo "== strip, $nf frames per launch"; tools/bin/x2bench $nf 20 "$C" | grep -v "^#"
done
echo "== generic, 32 frames per launch"; GMAT_SCALE_NO_STRIP=1 tools/bin/x2bench 32 5 "$C" | grep -v "^#"
for r in 8 12 16 20 24 27 32 45 54 68; do
  echo "== GMAT_STRIP_ROWS=$r, 32 frames"; GMAT_STRIP_ROWS=$r tools/bin/x2bench 32 20 "$C" | grep -v "^#\|verify"
done
for r in 8 16 27 34 45; do
  echo "== GMAT_STRIP_ROWS=$r, 1 frame"; GMAT_STRIP_ROWS=$r tools/bin/x2bench 1 20 "$C" | grep -v "^#\|verify"
done
} 2>&1 | tee gpurun_out/r2y2/x2.txt
