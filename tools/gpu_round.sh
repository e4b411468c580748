#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, smoke, bench, rocprofv3 kernel trace.
# Usage: tools/gpu_round.sh [tag]
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx" | head -8 > $OUT/rocminfo.txt
nproc > $OUT/nproc.txt
echo "== pytest gpu" ; timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.txt
echo "== bench" ; timeout 900 python bench.py 2>&1 | tail -3 | tee $OUT/bench.txt
echo "== bench driver form" ; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu 2>&1 | tail -1 | tee $OUT/bench_driver.txt
echo "== bench 1 stream" ; timeout 600 python bench.py --steps 50 --warmup 10 --branches 1 --no-cpu --no-chained --no-pipeline 2>&1 | tail -1 | tee $OUT/bench_1stream.txt
echo "== rocprof"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu > $OUT/rocprof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-chained --no-pipeline --branches 1 > $OUT/rocprof_serial.log 2>&1
find $OUT/prof_serial -name "*kernel_stats*" | head -1 | xargs -r head -8
