#!/bin/bash
# scale_yuv4x1_kernel: parity on the GPU, timing against the generic kernel, segment sweep
mkdir -p gpurun_out/d4
timeout 900 python -m pytest tests/test_parity_down4.py -q -m gpu -x 2>&1 | tail -3 | tee gpurun_out/d4/tests.txt
C="nv12 4K->540p nv12"
{
for nf in 32 8 1; do echo "== strip, $nf frames per launch"; tools/bin/x2bench $nf 20 "$C" | grep -v "^#"; done
echo "== generic, 32 frames per launch"; GMAT_SCALE_NO_STRIP=1 tools/bin/x2bench 32 5 "$C" | grep -v "^#"
for r in 5 9 13 17 21 29 45; do echo "== GMAT_STRIP_ROWS=$r, 32 frames"; GMAT_STRIP_ROWS=$r tools/bin/x2bench 32 20 "$C" | grep -v "^#\|verify"; done
for r in 2 3 5; do echo "== GMAT_STRIP_ROWS=$r, 1 frame"; GMAT_STRIP_ROWS=$r tools/bin/x2bench 1 20 "$C" | grep -v "^#\|verify"; done
} 2>&1 | tee gpurun_out/d4/x2.txt
