#!/bin/bash
# Lanczos on the strip kernels (6 coefficient pairs): whole suite, strip fuzzer, then timings against the tiled / generic kernel,
# and the 4-pair (bicubic) cases that must not have moved
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-lz}; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1
timeout 900 python tests/fuzz/fuzz_strip.py 4000 ${2:-777} --hip > $OUT/fuzz_strip.log 2>&1
for nf in 32 1; do for c in "land: nv12 4K->1080p rgb24 lanczos" "land: nv12 4K->1080p nv12 lanczos"; do
  echo "== strip (6 pairs), $nf frames per launch" >> $OUT/x2.txt; timeout 200 tools/bin/x2bench $nf 30 "$c" | tee -a $OUT/x2.txt | grep -v "verify.*identical"
  echo "== replaced kernel, $nf frames per launch" >> $OUT/x2.txt; GMAT_SCALE_NO_STRIP=1 timeout 200 tools/bin/x2bench $nf 30 "$c" | tee -a $OUT/x2.txt | grep -v "verify.*identical"
done; done
python3 tools/sweep.py "land: nv12 4K->1080p rgb24 lanczos" --nf 1,4,32 --env GMAT_STRIP_ROWS=-,6,8,12,16,24,32,48,64 --out $OUT/rows.txt | sed 's/ kernel=.*//'
echo "== the 4-pair cases (must not have moved: headline 4.10, rgba 4.36, yuv420p 4.17, 1080p 1.25, single 7.95)"
for nf in 32 1; do timeout 100 tools/bin/x2bench $nf 40 "4K->1080p rgb" | grep -v verify; done; timeout 100 tools/bin/x2bench 32 40 "1080p->540p" | grep -v verify
echo "== fuzz_strip"; tail -16 $OUT/fuzz_strip.log
echo "== pytest -m gpu (whole suite) — read first"; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head; tail -1 $OUT/pytest.log
grep -c MISMATCH $OUT/x2.txt | sed 's/^/x2bench batched-vs-single MISMATCH lines: /'
