#!/bin/bash
# scale_rgb2h_kernel (converted samples shared between lanes) against scale_rgb2s_kernel: whole suite, strip fuzzer, timings, the
# chained (convert-then-scale) form of bench.py with either kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-rgb2h}; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1
timeout 900 python tests/fuzz/fuzz_strip.py 5000 ${2:-1919} --hip > $OUT/fuzz_strip.log 2>&1
for sh in 1 0; do for nf in 32 1; do for c in "rgb24 4K->1080p rgb24 bicubic" "rgb24 4K->1080p bgra bilinear"; do
  echo "== GMAT_RGB2_SHARED=$sh, $nf frames per launch" | tee -a $OUT/x2.txt; GMAT_RGB2_SHARED=$sh timeout 200 tools/bin/x2bench $nf 30 "$c" | tee -a $OUT/x2.txt | grep -v "verify.*identical"
done; done; done
python3 tools/sweep.py "rgb24 4K->1080p rgb24 bicubic" --nf 1,4,32 --env GMAT_STRIP_ROWS=-,3,6,12,16,24,32,48,64 --out $OUT/rows.txt | sed 's/ kernel=.*//'
for sh in 1 0; do
  echo "== bench.py chained, GMAT_RGB2_SHARED=$sh"
  GMAT_RGB2_SHARED=$sh timeout 250 python bench.py --steps 30 --warmup 5 --no-cpu --no-pipeline 2>/dev/null | tee $OUT/bench_shared$sh.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['roofline']['frac'], 'two_kernels', d['chained']['two_kernels']['value'], d['chained']['two_kernels']['frac'], 'fused', d['chained']['fused_kernel']['value'])"
done
echo "== fuzz_strip"; grep -E "scale_rgb2|cases" $OUT/fuzz_strip.log
echo "== pytest -m gpu (whole suite) — read first"; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head; tail -1 $OUT/pytest.log
grep -c MISMATCH $OUT/x2.txt | sed 's/^/x2bench batched-vs-single MISMATCH lines: /'
