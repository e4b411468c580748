#!/bin/bash
# the last pass of the round: tools/gpu_final.sh, every x2bench case at 32 frames per launch, GPU fuzz with every strip family
TAG=${1:-final}
tools/gpu_final.sh $TAG > gpurun_out/${TAG}_final_stdout.txt 2>&1
X2BENCH_VERIFY=0 timeout 300 tools/bin/x2bench 32 20 > gpurun_out/$TAG/x2bench_32.txt 2>&1
{
for s in 1 2; do timeout 300 python tests/fuzz/fuzz_strip.py 3000 $((9100+s)) --hip 2>&1 | tail -45 | grep -E "yuv3r|yuv32r|yuv4r|rgb2y|cases|MISMATCH"; done
GMAT_STRIP_ROWS=2 timeout 300 python tests/fuzz/fuzz_strip.py 2000 9111 --hip 2>&1 | tail -45 | grep -E "yuv3r|yuv32r|yuv4r|rgb2y|cases|MISMATCH"
timeout 300 python tests/fuzz/fuzz_transforms.py 2000 9121 --hip 2>&1 | tail -2
} > gpurun_out/$TAG/fuzz.txt 2>&1
tail -32 gpurun_out/${TAG}_final_stdout.txt; cat gpurun_out/$TAG/fuzz.txt
