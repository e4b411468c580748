#!/bin/bash
# alternating walk direction (GMAT_STRIP_UPDOWN, default on) against all-down, on one x2bench case: time and HBM fetch per launch
# usage: tools/gpu_updown.sh <tag> "<case filter>"
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-updown}; CASE=${2:-"nv12 4K->720p nv12 bicubic (3:1)"}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for ud in 1 0; do for rep in 1 2; do
  echo "== GMAT_STRIP_UPDOWN=$ud" | tee -a $OUT/x2.txt; GMAT_STRIP_UPDOWN=$ud timeout 200 tools/bin/x2bench 32 30 "$CASE" | tee -a $OUT/x2.txt | grep -v "verify.*identical"
done; done
for ud in 1 0; do
  echo "== counters, GMAT_STRIP_UPDOWN=$ud"; GMAT_STRIP_UPDOWN=$ud tools/pmc_case.sh $TAG/pmc$ud "$CASE" 32 "FETCH_SIZE" "WRITE_SIZE" | grep -E "kernel|FETCH|WRITE"
done
