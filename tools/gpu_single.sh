cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02g; export X2BENCH_VERIFY=0
for c in "nv12 4K->1080p rgb24 bicubic" "nv12 1080p->540p rgb24 bicubic"; do
echo "tiled: " ; tools/bin/x2bench 1 600 "$c" | tail -1
for rows in 2 3 4 5 6 8; do printf "strip rows $rows: "; GMAT_STRIP_SINGLE=1 GMAT_STRIP_ROWS=$rows tools/bin/x2bench 1 600 "$c" | tail -1; done
for nf in 2 4 8 16; do printf "batch $nf default rows: "; tools/bin/x2bench $nf 200 "$c" | tail -1; done
done 2>&1 | tee gpurun_out/r02g/single.txt
