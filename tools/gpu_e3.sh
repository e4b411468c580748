#!/bin/bash
# the 3:2 down-scale strip kernel: whole suite, strip fuzzer, then timings against the generic kernel and a rows-per-segment sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-e3}; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1
timeout 900 python tests/fuzz/fuzz_strip.py 5000 ${2:-2121} --hip > $OUT/fuzz_strip.log 2>&1
for nf in 32 1; do for c in "nv12 1080p->720p nv12 bicubic (3:2)" "land: yuv420p 1080p->720p" "land: nv12 4K->1440p"; do
  echo "== strip, $nf frames per launch" >> $OUT/x2.txt; timeout 200 tools/bin/x2bench $nf 30 "$c" | tee -a $OUT/x2.txt | grep -v "verify.*identical"
  echo "== generic, $nf frames per launch" >> $OUT/x2.txt; GMAT_SCALE_NO_STRIP=1 timeout 300 tools/bin/x2bench $nf 10 "$c" | tee -a $OUT/x2.txt | grep -v "verify.*identical"
done; done
python3 tools/sweep.py "nv12 1080p->720p nv12 bicubic (3:2)" --nf 1,4,32 --env GMAT_STRIP_ROWS=-,4,8,12,16,24,32,48,64 --out $OUT/rows.txt | sed 's/ kernel=.*//'
echo "== fuzz_strip"; grep -E "scale_yuv3x2|cases" $OUT/fuzz_strip.log
echo "== pytest -m gpu (whole suite) — read first"; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head; tail -1 $OUT/pytest.log
grep -c MISMATCH $OUT/x2.txt | sed 's/^/x2bench batched-vs-single MISMATCH lines: /'
