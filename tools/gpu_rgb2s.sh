cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02j; export X2BENCH_VERIFY=1
( for nf in 1 4 8 32; do echo "== $nf frames per launch"; tools/bin/x2bench $nf 40 "rgb24 4K"; done
for rows in 16 32 48 64; do echo "rows $rows, 32 frames"; X2BENCH_VERIFY=0 GMAT_STRIP_ROWS=$rows tools/bin/x2bench 32 20 "rgb24 4K->1080p rgb24"; done
) 2>&1 | tee gpurun_out/r02j/rgb2s.txt
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r02j/pytest.txt
