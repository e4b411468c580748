#!/bin/bash
# several seeds x all fuzzers on the GPU; one summary line per run, failures (if any) with their reproduction lines
# usage: tools/gpu_fuzz_campaign.sh <tag> <cases> <seed> [<seed> ...]
TAG=$1; N=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for SEED in "$@"; do for f in fuzz_strip fuzz_parity fuzz_yuvopts fuzz_transforms fuzz_filters; do
  timeout 1500 python tests/fuzz/$f.py $N $SEED --hip > $OUT/${f}_$SEED.log 2>&1; rc=$?
  echo "$f seed $SEED n $N: rc=$rc $(tail -1 $OUT/${f}_$SEED.log)"
  [ $rc -ne 0 ] && grep -E "MISMATCH|ERROR|Traceback" -A3 $OUT/${f}_$SEED.log | head -20
done; done | tee $OUT/summary.txt
echo "total cases: $(grep -o 'cases [0-9]*' $OUT/summary.txt | awk '{s+=$2} END {print s}')  failures: $(grep -o 'failures [0-9]*' $OUT/summary.txt | awk '{s+=$2} END {print s}')  non-zero exits: $(grep -c 'rc=[1-9]' $OUT/summary.txt)" | tee -a $OUT/summary.txt
