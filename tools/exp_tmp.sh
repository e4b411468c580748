export X2BENCH_VERIFY=0 X2BENCH_SETS=4
python -m pytest tests/test_parity_walker16.py tests/test_parity_generic_walker.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3
echo "== deep, 32 frames"; tools/bin/x2bench 32 12 "deep:" | cut -c1-175
echo "== deep, 1 frame"; tools/bin/x2bench 1 40 "deep:" | cut -c1-175
echo "== deep, 32 frames, walker16 off"; GMAT_SCALE_NO_WALKER16=1 tools/bin/x2bench 32 12 "deep:" | cut -c1-175
