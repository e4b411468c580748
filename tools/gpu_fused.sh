#!/bin/bash
# the fused convert-then-scale form on the strip kernel (scale_rgb2h_kernel<yuv>): whole suite, strip fuzzer, bench.py's chained block
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-fused}; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1
timeout 900 python tests/fuzz/fuzz_strip.py 5000 ${2:-2020} --hip > $OUT/fuzz_strip.log 2>&1
for sh in 1 0; do
  echo "== bench.py chained, GMAT_RGB2_SHARED=$sh"
  GMAT_RGB2_SHARED=$sh timeout 250 python bench.py --steps 30 --warmup 5 --no-cpu --no-pipeline 2>/dev/null | tee $OUT/bench_shared$sh.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['roofline']['frac'], 'two_kernels', d['chained']['two_kernels']['value'], d['chained']['two_kernels']['frac'], 'fused', d['chained']['fused_kernel'])"
done
for rows in 6 12 16 24 32 48; do echo "GMAT_STRIP_ROWS=$rows"; GMAT_STRIP_ROWS=$rows timeout 250 python bench.py --steps 30 --warmup 5 --no-cpu --no-pipeline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   fused', d['chained']['fused_kernel']['value'], 'two_kernels', d['chained']['two_kernels']['value'])"; done
echo "== fuzz_strip"; grep -E "scale_rgb2|cases" $OUT/fuzz_strip.log
echo "== pytest -m gpu (whole suite) — read first"; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head; tail -1 $OUT/pytest.log
