cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02h
timeout 900 python bench.py --steps 50 --warmup 10 --no-cpu --no-pipeline 2>&1 | tail -1 > gpurun_out/r02h/bench.txt
