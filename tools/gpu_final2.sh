#!/bin/bash
# tools/gpu_final.sh + the HBM-traffic counters of scale_rgb2y_kernel and every x2bench case at 32 frames per launch
TAG=${1:-final}
tools/gpu_final.sh $TAG > gpurun_out/${TAG}_final_stdout.txt 2>&1
tools/pmc_case.sh $TAG/pmc_rgb2y "rgb24 4K->1080p nv12" 32 "FETCH_SIZE" "WRITE_SIZE" > gpurun_out/$TAG/pmc_rgb2y.txt 2>&1
X2BENCH_VERIFY=0 timeout 300 tools/bin/x2bench 32 20 > gpurun_out/$TAG/x2bench_32.txt 2>&1
X2BENCH_VERIFY=0 timeout 300 tools/bin/x2bench 32 10 land > gpurun_out/$TAG/x2bench_land.txt 2>&1
cat gpurun_out/${TAG}_final_stdout.txt | tail -40; cat gpurun_out/$TAG/pmc_rgb2y.txt | tail -8
