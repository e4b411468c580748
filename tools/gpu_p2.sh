#!/bin/bash
# the plane-walking 4:2:0 -> 4:2:0 kernel on the GPU: its parity tests (both kernels), then x2bench on the transcode cases
# with the strip kernel and with the tiled one.  usage: tools/gpu_p2.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-p2}; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1
for nf in 32 1 4; do
  for c in "nv12 4K->1080p nv12" "yuv420p 4K->1080p yuv420p"; do
    echo "== strip, $nf frames per launch" | tee -a $OUT/x2.txt; timeout 120 tools/bin/x2bench $nf 40 "$c" | tee -a $OUT/x2.txt
    echo "== tiled, $nf frames per launch" | tee -a $OUT/x2.txt; GMAT_SCALE_NO_STRIP=1 timeout 120 tools/bin/x2bench $nf 40 "$c" | tee -a $OUT/x2.txt
  done
done
echo "== pytest -m gpu (whole suite) — read this before any number above"; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -20; tail -1 $OUT/pytest.log
grep -c MISMATCH $OUT/x2.txt | sed 's/^/x2bench batched-vs-single MISMATCH lines: /'
