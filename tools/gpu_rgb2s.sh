cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02i; export X2BENCH_VERIFY=0
( echo "== strip"; tools/bin/x2bench 8 40 "rgb24 4K"; 
for rows in 3 4 6 8 12 16 24; do echo "rows $rows"; GMAT_STRIP_ROWS=$rows tools/bin/x2bench 8 40 "rgb24 4K->1080p rgb24"; done
echo "== tiled"; GMAT_SCALE_NO_STRIP=1 tools/bin/x2bench 8 40 "rgb24 4K"
echo "== verify strip vs tiled: run with verify (batch = frame-by-frame either way; compares CRCs of both env settings)"
) 2>&1 | tee gpurun_out/r02i/rgb2s.txt
timeout 600 python -m pytest tests/test_parity_scale.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r02i/pytest.txt
