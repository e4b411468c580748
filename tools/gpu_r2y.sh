#!/bin/bash
# the strip-walking RGB -> 4:2:0 converter (rgb2yuv420s_kernel): whole suite, strip fuzzer, timings against the tiled kernel, rows sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r2y}; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1
timeout 900 python tests/fuzz/fuzz_strip.py 5000 ${2:-2222} --hip > $OUT/fuzz_strip.log 2>&1
for c in "rgb24 4K->4K nv12 (convert)" "rgb24 1080p->1080p yuv420p (convert)"; do for nf in 1 8; do
  echo "== strip, $nf frames per step" >> $OUT/x2.txt; timeout 200 tools/bin/x2bench $nf 30 "$c" | tee -a $OUT/x2.txt | grep -v "verify.*identical"
  echo "== tiled, $nf frames per step" >> $OUT/x2.txt; GMAT_SCALE_NO_STRIP=1 timeout 300 tools/bin/x2bench $nf 10 "$c" | tee -a $OUT/x2.txt | grep -v "verify.*identical"
done; done
python3 tools/sweep.py "rgb24 4K->4K nv12 (convert)" --nf 1,8 --env GMAT_STRIP_ROWS=-,2,3,4,6,8,12,16,32 --out $OUT/rows.txt | sed 's/ kernel=.*//'
python3 tools/sweep.py "rgb24 1080p->1080p yuv420p (convert)" --nf 1 --env GMAT_STRIP_ROWS=-,2,3,4,6,8 --out $OUT/rows1080.txt | sed 's/ kernel=.*//'
echo "== fuzz_strip"; grep -E "rgb2yuv420|cases" $OUT/fuzz_strip.log
echo "== pytest -m gpu (whole suite) — read first"; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head; tail -1 $OUT/pytest.log
grep -c MISMATCH $OUT/x2.txt | sed 's/^/x2bench batched-vs-single MISMATCH lines: /'
