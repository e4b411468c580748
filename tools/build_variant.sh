#!/bin/bash
# tools/build_variant.sh <name> <file.hip | all> [-DFLAG=..]...: the product library with ONE object (or every kernel file) rebuilt under extra flags
# -> tools/variants/<name>/libgmat_hip.so (git-ignored, travels with gpurun).  On the GPU box an A/B swaps it in:
#   cp tools/variants/<name>/libgmat_hip.so gmat_amd/lib/libgmat_hip.so   (the box's copy is scratch)
set -e
NAME=$1; SRC=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd); C=$R/gmat_amd/csrc; O=$R/tools/variants/$NAME; mkdir -p $O
make -C $C -j8 > /dev/null
if [ "$SRC" = all ]; then                 # every kernel file under the flags (e.g. -DGMAT_NT_STORES=0)
  OBJS=$(ls $C/build/*.o)
  for S in $C/*.hip; do
    B=$(basename $S .hip)
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -ffp-contract=off --offload-arch=gfx950 \
      -fhip-fp32-correctly-rounded-divide-sqrt -I$R/include "$@" -c $S -o $O/$B.o &
    OBJS=$(echo "$OBJS" | grep -v "/$B.o"); OBJS="$OBJS $O/$B.o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libgmat_hip.so $OBJS
  echo "built $O/libgmat_hip.so (all: $*)"; exit 0
fi
B=$(basename $SRC .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -ffp-contract=off --offload-arch=gfx950 \
  -fhip-fp32-correctly-rounded-divide-sqrt -I$R/include "$@" -c $C/$B.hip -o $O/$B.o
OBJS=$(ls $C/build/*.o | grep -v "/$B.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libgmat_hip.so $OBJS $O/$B.o
echo "built $O/libgmat_hip.so ($*)"
