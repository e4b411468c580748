TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
nproc > $OUT/nproc.txt
echo "== bench" ; timeout 900 python bench.py 2>&1 | tail -3 | tee $OUT/bench.txt
echo "== bench driver form" ; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu 2>&1 | tail -1 | tee $OUT/bench_driver.txt
echo "== bench 1 stream" ; timeout 600 python bench.py --steps 50 --warmup 10 --branches 1 --no-cpu --no-chained --no-pipeline 2>&1 | tail -1 | tee $OUT/bench_1stream.txt
