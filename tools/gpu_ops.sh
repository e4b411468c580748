#!/bin/bash
# filter kernels on the GPU: parity tests of the filters, then the one-frame-per-launch benches of tools/bin/x2bench with the
# separable 1-2-1 kernel (default) and the general 3x3 kernel (GMAT_NO_SMOOTH121=1).  usage: tools/gpu_ops.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-ops}; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_parity_filters.py -x -q -m gpu > $OUT/pytest_filters.log 2>&1; tail -3 $OUT/pytest_filters.log
export X2BENCH_VERIFY=0
echo "== separable 1-2-1 kernel" | tee $OUT/ops.txt; timeout 120 tools/bin/x2bench 1 50 "op: " | tee -a $OUT/ops.txt
echo "== general 3x3 kernel" | tee -a $OUT/ops.txt;  GMAT_NO_SMOOTH121=1 timeout 120 tools/bin/x2bench 1 50 "op: " | tee -a $OUT/ops.txt
