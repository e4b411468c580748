#!/bin/bash
# the plane-walking 4:2:0 -> 4:2:0 kernels (8 / 10 bits on either side) on the GPU: whole suite first, then the strip fuzzer, then
# timings of all four depth pairs against the kernel they replace
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-p16}; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1
timeout 900 python tests/fuzz/fuzz_strip.py 4000 ${2:-4242} --hip > $OUT/fuzz_strip.log 2>&1
for nf in 32 1; do
  for c in "nv12 4K->1080p nv12" "yuv420p 4K->1080p yuv420p" "p010 4K->1080p p010" "yuv420p10le 4K" "nv12 4K->1080p p010" "p010 4K->1080p nv12"; do
    echo "== strip, $nf frames per launch" >> $OUT/x2.txt; timeout 200 tools/bin/x2bench $nf 30 "$c" | tee -a $OUT/x2.txt | grep -v "^ *verify.*identical"
  done
  for c in "nv12 4K->1080p p010" "p010 4K->1080p nv12"; do
    echo "== replaced kernel (GMAT_SCALE_NO_STRIP=1), $nf frames per launch" | tee -a $OUT/x2.txt; GMAT_SCALE_NO_STRIP=1 timeout 200 tools/bin/x2bench $nf 30 "$c" | tee -a $OUT/x2.txt | grep -v "^ *verify.*identical"
  done
done
echo "== fuzz_strip"; tail -16 $OUT/fuzz_strip.log
echo "== pytest -m gpu (whole suite) — read first"; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head; tail -1 $OUT/pytest.log
grep -c MISMATCH $OUT/x2.txt | sed 's/^/x2bench batched-vs-single MISMATCH lines: /'
