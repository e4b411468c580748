#!/bin/bash
# Runs on the GPU box: x2bench over the library variants of tools/build_variants.sh.  usage: tools/gpu_variants.sh <tag> "<rows list>" variant...
TAG=$1; ROWS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
export X2BENCH_VERIFY=0
for v in "$@"; do
  for rows in $ROWS; do
    printf "%-14s rows %-3s " $v $rows | tee -a $OUT/variants.txt
    LD_LIBRARY_PATH=$R/tools/bin/variants/$v GMAT_STRIP_ROWS=$rows timeout 120 tools/bin/x2bench 32 40 "nv12 4K->1080p rgb24 bicubic" 2>&1 | tail -1 | tee -a $OUT/variants.txt
  done
done
