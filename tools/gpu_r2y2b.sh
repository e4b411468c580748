#!/bin/bash
mkdir -p gpurun_out/r2y2
C="rgb24 4K->1080p nv12"
{
for r in 1 2 3 4 5 6; do
  echo "== GMAT_STRIP_ROWS=$r, 1 frame"; GMAT_STRIP_ROWS=$r tools/bin/x2bench 1 20 "$C" | grep -v "^#\|verify"
done
for r in 4 6 8 10 12 16 20 27; do
  echo "== GMAT_STRIP_ROWS=$r, 8 frames"; GMAT_STRIP_ROWS=$r tools/bin/x2bench 8 20 "$C" | grep -v "^#\|verify"
done
for r in 3 4 6 8 10; do
  echo "== GMAT_STRIP_ROWS=$r, 4 frames"; GMAT_STRIP_ROWS=$r tools/bin/x2bench 4 20 "$C" | grep -v "^#\|verify"
done
} 2>&1 | tee gpurun_out/r2y2/x2b.txt
