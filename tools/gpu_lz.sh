#!/bin/bash
# Lanczos on the plane-walking kernel (6 coefficient pairs): whole suite, strip fuzzer, then timings against the tiled / generic kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-lz}; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1
timeout 900 python tests/fuzz/fuzz_strip.py 4000 ${2:-777} --hip > $OUT/fuzz_strip.log 2>&1
for nf in 32 1; do for c in "land: nv12 4K->1080p nv12 lanczos" "land: p010 4K->1080p p010 lanczos"; do
  echo "== strip (6 pairs), $nf frames per launch" >> $OUT/x2.txt; timeout 200 tools/bin/x2bench $nf 30 "$c" | tee -a $OUT/x2.txt | grep -v "verify.*identical"
  echo "== replaced kernel, $nf frames per launch" >> $OUT/x2.txt; GMAT_SCALE_NO_STRIP=1 timeout 200 tools/bin/x2bench $nf 30 "$c" | tee -a $OUT/x2.txt | grep -v "verify.*identical"
done; done
python3 tools/sweep.py "land: nv12 4K->1080p nv12 lanczos" --nf 1,4,32 --env GMAT_STRIP_ROWS=-,4,6,8,12,16,24,32,48 --out $OUT/rows.txt | sed 's/ kernel=.*//'
echo "== bicubic unchanged?"; timeout 100 tools/bin/x2bench 32 30 "nv12 4K->1080p nv12 bicubic" | grep -v verify; timeout 100 tools/bin/x2bench 32 30 "p010 4K->1080p p010" | grep -v verify
echo "== fuzz_strip"; tail -16 $OUT/fuzz_strip.log
echo "== pytest -m gpu (whole suite) — read first"; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head; tail -1 $OUT/pytest.log
grep -c MISMATCH $OUT/x2.txt | sed 's/^/x2bench batched-vs-single MISMATCH lines: /'
