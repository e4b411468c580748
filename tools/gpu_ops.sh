#!/bin/bash
# filter kernels on the GPU: parity tests of the filters, then the op benches of tools/bin/x2bench with both 3x3 kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-ops}; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_parity_filters.py -x -q -m gpu > $OUT/pytest_filters.log 2>&1; tail -3 $OUT/pytest_filters.log
export X2BENCH_VERIFY=0
for i in 1 2; do
  echo "== separable"; timeout 120 tools/bin/x2bench 1 50 "op:" | tee -a $OUT/ops_new.txt
done
echo "== general";   GMAT_NO_SMOOTH121=1 timeout 120 tools/bin/x2bench 1 50 "op: " | tee -a $OUT/ops_old.txt
