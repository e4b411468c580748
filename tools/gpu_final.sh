#!/bin/bash
# end-of-milestone GPU pass: whole GPU suite, smoke, bench (default and driver form), rocprofv3 kernel stats of the bench,
# HBM-traffic counters, filter benches.  The suite's verdict is printed LAST.  usage: tools/gpu_final.sh <tag>
TAG=${1:-final}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
timeout 900 python bench.py 2>/dev/null | tail -1 > $OUT/bench.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_driver_form.json
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-chained --no-pipeline --branches 1 > $OUT/rocprof.log 2>&1 )
tools/pmc_traffic_x2.sh $TAG/traffic > $OUT/traffic.log 2>&1
tools/gpu_ops.sh $TAG/ops > $OUT/ops.log 2>&1
find $OUT/prof -name "*kernel_stats*" | head -1 | xargs -r head -6
python - <<PY
import json
for f in ("bench.json", "bench_driver_form.json"):
    try:
        d = json.loads(open("$OUT/" + f).read())
        print(f, d["value"], d["unit"], "roofline", d["roofline"]["frac"], "avg_launch_us", d["roofline"]["avg_launch_us"], "traffic", d["roofline"]["traffic"])
        for k, v in d.get("other_configs", {}).items(): print("   ", k, v.get("frac"), v.get("avg_launch_us") or v.get("us_per_frame"))
    except Exception as e: print(f, "unreadable:", e)
PY
cat $OUT/smoke.txt | tail -2; tail -12 $OUT/traffic.log
echo "== pytest -m gpu"; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head; tail -1 $OUT/pytest_gpu.log
