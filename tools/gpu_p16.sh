#!/bin/bash
# the 10-bit plane-walking kernel on the GPU: whole suite first, then the strip fuzzer, then timings against the generic kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-p16}; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1
timeout 900 python tests/fuzz/fuzz_strip.py 3000 31337 --hip > $OUT/fuzz_strip.log 2>&1
for nf in 32 1; do
  echo "== strip, $nf frames per launch" | tee -a $OUT/x2.txt; timeout 200 tools/bin/x2bench $nf 30 "p010 4K->1080p p010" | tee -a $OUT/x2.txt; timeout 200 tools/bin/x2bench $nf 30 "yuv420p10le 4K" | tee -a $OUT/x2.txt
  echo "== generic (GMAT_SCALE_NO_STRIP=1), $nf frames per launch" | tee -a $OUT/x2.txt; GMAT_SCALE_NO_STRIP=1 timeout 200 tools/bin/x2bench $nf 30 "p010 4K->1080p p010" | tee -a $OUT/x2.txt
done
python3 tools/sweep.py "p010 4K->1080p p010" --nf 1,4,32 --env GMAT_STRIP_ROWS=-,3,4,6,8,12,16,24,32 --out $OUT/rows.txt
echo "== fuzz_strip"; tail -14 $OUT/fuzz_strip.log
echo "== pytest -m gpu (whole suite) — read first"; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head; tail -1 $OUT/pytest.log
grep -c MISMATCH $OUT/x2.txt | sed 's/^/x2bench batched-vs-single MISMATCH lines: /'
