#!/bin/bash
# tools/ab_x2.sh <nf> <launches> <case filter> <variant> [<variant> ...]: tools/bin/x2bench with each library of tools/variants/<variant>/
# swapped in ("shipped" = the library as built), on the GPU box (its copy of the repo is scratch)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
NF=$1; L=$2; C=$3; shift 3
cp gmat_amd/lib/libgmat_hip.so /tmp/libgmat_hip_shipped.so
for v in "$@"; do
  if [ $v = shipped ]; then cp /tmp/libgmat_hip_shipped.so gmat_amd/lib/libgmat_hip.so; else cp tools/variants/$v/libgmat_hip.so gmat_amd/lib/libgmat_hip.so; fi
  echo "== $v"; X2BENCH_VERIFY=0 timeout 600 tools/bin/x2bench $NF $L "$C"
done
cp /tmp/libgmat_hip_shipped.so gmat_amd/lib/libgmat_hip.so
