#!/bin/bash
# mixed chroma layouts on the plane-walking kernels: whole suite, strip fuzzer, timings against the tiled kernel, same-layout control
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-px}; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1
timeout 900 python tests/fuzz/fuzz_strip.py 5000 ${2:-8642} --hip > $OUT/fuzz_strip.log 2>&1
for nf in 32 1; do for c in "nv12 4K->1080p yuv420p" "yuv420p 4K->1080p nv12"; do
  echo "== strip, $nf frames per launch" >> $OUT/x2.txt; timeout 200 tools/bin/x2bench $nf 30 "$c" | tee -a $OUT/x2.txt | grep -v "verify.*identical"
  echo "== tiled, $nf frames per launch" >> $OUT/x2.txt; GMAT_SCALE_NO_STRIP=1 timeout 200 tools/bin/x2bench $nf 30 "$c" | tee -a $OUT/x2.txt | grep -v "verify.*identical"
done; done
echo "== same-layout control (4.03 / 3.96 / 6.39)"; for c in "nv12 4K->1080p nv12 bicubic" "yuv420p 4K->1080p yuv420p" "p010 4K->1080p p010"; do timeout 100 tools/bin/x2bench 32 30 "$c" | grep -v verify; done
echo "== fuzz_strip"; tail -22 $OUT/fuzz_strip.log
echo "== pytest -m gpu (whole suite) — read first"; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head; tail -1 $OUT/pytest.log
grep -c MISMATCH $OUT/x2.txt | sed 's/^/x2bench batched-vs-single MISMATCH lines: /'
