#!/bin/bash
# randomised differential testing on the GPU: the four fuzzers against the oracle, fresh seeds given on the command line
# usage: tools/gpu_fuzz.sh <tag> <seed> [cases per fuzzer]
TAG=${1:-fuzz}; SEED=${2:-101}; N=${3:-3000}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for f in fuzz_strip fuzz_parity fuzz_yuvopts fuzz_transforms fuzz_filters; do
  timeout 1500 python tests/fuzz/$f.py $N $SEED --hip > $OUT/$f.log 2>&1; echo "$f seed $SEED n $N: rc=$? $(tail -1 $OUT/$f.log)"; [ $f = fuzz_strip ] && tail -14 $OUT/$f.log | head -13
done | tee $OUT/summary.txt
