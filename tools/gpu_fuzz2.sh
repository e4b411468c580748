#!/bin/bash
# GPU fuzz of the kernels added late in round 2 (scale_rgb2y_kernel, median3x3s_kernel, the 64 x 64 / 2-byte transposes) with the rest
mkdir -p gpurun_out/fuzz2
{
for s in 1 2 3; do timeout 300 python tests/fuzz/fuzz_strip.py 3000 $((7100+s)) --hip 2>&1 | tail -40 | grep -E "rgb2y|cases|MISMATCH"; done
for r in 1 2 5; do GMAT_STRIP_ROWS=$r timeout 300 python tests/fuzz/fuzz_strip.py 2000 $((7200+r)) --hip 2>&1 | tail -40 | grep -E "rgb2y|cases|MISMATCH"; done
for s in 1 2; do timeout 300 python tests/fuzz/fuzz_transforms.py 3000 $((7300+s)) --hip 2>&1 | tail -3; done
GMAT_STRIP_ROWS=3 timeout 300 python tests/fuzz/fuzz_transforms.py 2000 7311 --hip 2>&1 | tail -2
GMAT_TRANSPOSE_TILE=128 timeout 300 python tests/fuzz/fuzz_transforms.py 1500 7312 --hip 2>&1 | tail -2
} 2>&1 | tee gpurun_out/fuzz2/fuzz.txt
