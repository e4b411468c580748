#!/bin/bash
# scale_yuv32r_kernel: parity on the GPU, timing against the generic kernel, segment sweep, and the unbounded (171 VGPR) build
mkdir -p gpurun_out/d32r
timeout 900 python -m pytest tests/test_parity_down32rgb.py -q -m gpu -x 2>&1 | tail -3 | tee gpurun_out/d32r/tests.txt
C="nv12 1080p->720p rgb24"
{
for nf in 32 8 1; do echo "== strip, $nf frames per launch"; tools/bin/x2bench $nf 20 "$C" | grep -v "^#"; done
echo "== generic, 32 frames per launch"; GMAT_SCALE_NO_STRIP=1 tools/bin/x2bench 32 5 "$C" | grep -v "^#"
for r in 8 12 16 20 24 32 48; do echo "== GMAT_STRIP_ROWS=$r, 32 frames"; GMAT_STRIP_ROWS=$r tools/bin/x2bench 32 20 "$C" | grep -v "^#\|verify"; done
for r in 4 8 12; do echo "== GMAT_STRIP_ROWS=$r, 1 frame"; GMAT_STRIP_ROWS=$r tools/bin/x2bench 1 20 "$C" | grep -v "^#\|verify"; done
echo "== 4K -> 1440p rgb24, 32 frames"; tools/bin/x2bench 32 10 "land: nv12 4K->1440p rgb24" | grep -v "^#"
echo "== 4K -> 1440p rgb24, generic"; GMAT_SCALE_NO_STRIP=1 tools/bin/x2bench 32 5 "land: nv12 4K->1440p rgb24" | grep -v "^#"
if [ -f gpurun_exp/unb/libgmat_hip.so ]; then
  cp gmat_amd/lib/libgmat_hip.so /tmp/lib_default.so; cp gpurun_exp/unb/libgmat_hip.so gmat_amd/lib/libgmat_hip.so
  for nf in 32 1; do echo "== unbounded build (171 VGPRs, 2 waves), $nf frames per launch"; tools/bin/x2bench $nf 20 "$C" | grep -v "^#"; done
  cp /tmp/lib_default.so gmat_amd/lib/libgmat_hip.so
fi
} 2>&1 | tee gpurun_out/d32r/x2.txt
