#!/bin/bash
# Tuning aid: builds variants of the product library that differ only in k_scale_yuv2s.hip's compile-time knobs, into
# tools/bin/variants/<name>/libgmat_hip.so (git-ignored, shipped to the GPU box).  x2bench picks one up through
# LD_LIBRARY_PATH.   usage: tools/build_variants.sh name1:"-DFOO=1 -DBAR=2" name2:"..."
set -e
R=$(cd $(dirname $0)/.. && pwd)
make -C $R/gmat_amd/csrc -j8 > /dev/null
FLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -ffp-contract=off --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -I$R/include"
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  d=$R/tools/bin/variants/$name; mkdir -p $d
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c $R/gmat_amd/csrc/k_scale_yuv2s.hip -o $d/k_scale_yuv2s.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A6 "kernelILb1ELi0E" | grep -E "VGPRs:|Occupancy" | sed "s/^.*remark: [^ ]* */  $name: /" | tr '\n' ' '; echo
    objs=$(ls $R/gmat_amd/csrc/build/*.o | grep -v k_scale_yuv2s.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libgmat_hip.so $objs $d/k_scale_yuv2s.o ) &
done
wait
