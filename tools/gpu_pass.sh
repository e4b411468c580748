#!/bin/bash
# ONE parameterised GPU pass (replaces round 2's 35 one-off gpu_*.sh scripts).  Runs ON the GPU box:
#     tools/gpurun.sh --timeout 1500 -- 'tools/gpu_pass.sh <tag> <step> [<step> ...]'
# Everything is written under gpurun_out/<tag>/; copy what is to be judged into profiles/.  Steps, in the order given:
#   suite                 whole GPU suite (pytest -m gpu); its verdict is printed LAST, after every number
#   tests:<expr>          pytest -m gpu -k <expr>
#   smoke                 __graft_entry__.smoke()
#   bench[:args]          python bench.py <args> (':' separated), last line -> bench.json, BENCH_DETAIL lines -> bench_detail.txt
#   driver                python bench.py --gpus 1 --steps 20 --warmup 5   (the driver's form) -> bench_driver_form.json
#   rocprof               rocprofv3 --kernel-trace --stats over the driver-form bench (one stream) -> rocprof_kernel_stats.csv
#   traffic               FETCH_SIZE / WRITE_SIZE passes over every 32-frame x2bench case + the filter ops -> traffic.json
#   pmc:<case>:<nf>:<ctrs>[:<ctrs>]   counter passes of ONE x2bench case (ctrs: comma separated list per pass)
#   x2[:nf[:launches[:case]]]         tools/bin/x2bench table (default 32 20, every case)
#   x2env:<VAR=val,...>:<nf>:<launches>:<case>   the same with environment knobs (A/B of a switch on one box)
#   layout                the GPU suite against every library under tools/variants/ (differently compiled builds of all kernels)
#   ops                   filter kernels, one 4K frame per launch
#   fuzz[:n[:seed]]       the five differential fuzzers against the oracle on the GPU
#   sh:<command>          anything else (quoted), output -> sh_<n>.txt
TAG=${1:-pass}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export X2BENCH_VERIFY=${X2BENCH_VERIFY:-0}
SUITE=0; NSH=0
summ() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s: %s %s  ms/step %s  roofline %s (%s us per %s-frame launch)  traffic %s (%s)  cpu %s" % (
        sys.argv[1].split("/")[-1], d["value"], d["unit"], d["ms_per_step"], r["frac"], r["avg_launch_us"], r["frames_per_launch"],
        r["traffic"], (r.get("traffic_source") or "")[:9], (d.get("cpu_baseline") or {}).get("value")), " last line", len(json.dumps(d)), "bytes")
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
}
for STEP in "$@"; do
  IFS=':' read -r -a A <<< "$STEP"
  case ${A[0]} in
  suite) SUITE=1; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1 ;;
  tests) timeout 1200 python -m pytest tests -q -m gpu -k "${A[1]}" -p no:cacheprovider > $OUT/pytest_k.log 2>&1; tail -3 $OUT/pytest_k.log ;;
  smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt ;;
  bench) ARGS="${A[@]:1}"; N=$(echo $ARGS | tr -c 'a-zA-Z0-9' '_'); timeout 1200 python bench.py $ARGS > $OUT/bench${N:+_$N}.stdout 2> $OUT/bench${N:+_$N}.stderr
         tail -1 $OUT/bench${N:+_$N}.stdout > $OUT/bench${N:+_$N}.json; grep '^BENCH_DETAIL' $OUT/bench${N:+_$N}.stdout > $OUT/bench${N:+_$N}_detail.txt
         summ $OUT/bench${N:+_$N}.json ;;
  driver) timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.stdout 2>/dev/null
          tail -1 $OUT/bench_driver_form.stdout > $OUT/bench_driver_form.json; summ $OUT/bench_driver_form.json
          echo "driver tail check: last 8192 bytes parse ->" $(tail -c 8192 $OUT/bench_driver_form.stdout | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ok', 'roofline' in d, 'cpu_baseline' in d)" 2>&1) ;;
  rocprof) ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- \
             python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-detail --no-pipeline --no-pmc --branches 1 > $OUT/rocprof_bench.stdout 2> $OUT/rocprof.log )
           tail -1 $OUT/rocprof_bench.stdout > $OUT/bench_under_rocprofv3.json; summ $OUT/bench_under_rocprofv3.json
           F=$(find $OUT/prof -name "*kernel_stats*" | head -1); [ -n "$F" ] && cp $F $OUT/rocprof_kernel_stats.csv && head -5 $OUT/rocprof_kernel_stats.csv ;;
  traffic) tools/pmc_traffic_x2.sh $TAG/traffic > $OUT/traffic.log 2>&1; cp $OUT/traffic/traffic.json $OUT/traffic.json 2>/dev/null; tail -14 $OUT/traffic.log ;;
  pmc) CASE=${A[1]}; NF=${A[2]}; P=(); for c in "${A[@]:3}"; do P+=("$(echo $c | tr ',' ' ')"); done
       tools/pmc_case.sh $TAG/pmc "$CASE" $NF "${P[@]}" | tee $OUT/pmc_$(echo $CASE | tr -c 'a-zA-Z0-9' '_').txt ;;
  x2) SFX=$(echo "${A[3]:-}" | tr -c 'a-zA-Z0-9\n' '_'); timeout 600 tools/bin/x2bench ${A[1]:-32} ${A[2]:-20} "${A[3]:-}" 2>&1 | tee $OUT/x2bench_${A[1]:-32}${SFX:+_$SFX}.txt ;;
  x2env) ( for kv in $(echo ${A[1]} | tr ',' ' '); do export $kv; done; echo "== ${A[1]}"; timeout 600 tools/bin/x2bench ${A[2]:-32} ${A[3]:-20} "${A[4]:-}" 2>&1 ) | tee -a $OUT/x2env.txt ;;
  layout) # the GPU suite against differently compiled builds of every kernel (built beforehand in the build container:
          #   tools/build_variant.sh o2 all -O2; tools/build_variant.sh os all -Os): layout-dependent faults (FINDINGS.md R3-walker-bands)
          cp gmat_amd/lib/libgmat_hip.so /tmp/libgmat_hip_shipped.so
          for v in $(ls tools/variants 2>/dev/null); do
            cp tools/variants/$v/libgmat_hip.so gmat_amd/lib/libgmat_hip.so
            echo "layout $v: $(timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | tail -1)" | tee -a $OUT/layout.txt
          done
          cp /tmp/libgmat_hip_shipped.so gmat_amd/lib/libgmat_hip.so ;;
  ops) echo "== filter ops, one 4K frame per launch" | tee $OUT/ops.txt; timeout 200 tools/bin/x2bench 1 50 "op: " 2>&1 | tee -a $OUT/ops.txt ;;
  fuzz) N=${A[1]:-2000}; SEED=${A[2]:-301}
        for f in fuzz_strip fuzz_walker fuzz_parity fuzz_yuvopts fuzz_unit fuzz_transforms fuzz_filters; do
          timeout 1500 python tests/fuzz/$f.py $N $SEED --hip > $OUT/$f.log 2>&1; echo "$f seed $SEED n $N: rc=$? $(tail -1 $OUT/$f.log)"
          { [ $f = fuzz_strip ] || [ $f = fuzz_walker ]; } && tail -24 $OUT/$f.log | head -23
        done | tee $OUT/fuzz.txt ;;
  sh) NSH=$((NSH + 1)); bash -c "${STEP#sh:}" 2>&1 | tee $OUT/sh_$NSH.txt ;;
  *) echo "gpu_pass.sh: unknown step $STEP" ;;
  esac
done
if [ $SUITE = 1 ]; then echo "== pytest -m gpu"; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head; tail -1 $OUT/pytest_gpu.log; fi
