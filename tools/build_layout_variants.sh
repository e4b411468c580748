#!/bin/bash
# tools/build_layout_variants.sh — the product library with every kernel file that issues a memory instruction from INLINE ASSEMBLY rebuilt under
# -O2 and under -Os: different schedules, register allocations and code sizes around the asm strings, whose hazards the compiler cannot check
# (FINDINGS.md R3-walker-bands).  -> tools/variants/layout_o2/libgmat_hip.so, tools/variants/layout_os/libgmat_hip.so (git-ignored; they
# travel to the GPU box, where tests/test_layout_variants.py runs the walkers' parity tests against each; tests/test_isa_guard.py lints them here).
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/gmat_amd/csrc
make -C $C -j8 > /dev/null
FILES="k_scale_yuvg k_scale_yuvu k_scale_yuv3x1 k_scale_yuv3x2 k_scale_yuv2x"
for V in o2:-O2 os:-Os; do
  N=layout_${V%%:*}; F=${V##*:}; O=$R/tools/variants/$N; mkdir -p $O
  OBJS=$(ls $C/build/*.o)
  for B in $FILES; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -ffp-contract=off --offload-arch=gfx950 \
      -fhip-fp32-correctly-rounded-divide-sqrt -I$R/include $F -c $C/$B.hip -o $O/$B.o &
    OBJS=$(echo "$OBJS" | grep -v "/$B.o"); OBJS="$OBJS $O/$B.o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libgmat_hip.so $OBJS
  rm -f $O/*.o
  echo "built $O/libgmat_hip.so ($F: $FILES)"
done
