"""bench.py's output contract: the driver keeps an 8 KB tail of stdout and parses the LAST line, so
that line must be one compact JSON object carrying the headline, `roofline` and `cpu_baseline`; everything secondary goes to
earlier `BENCH_DETAIL` lines.  Run on the CPU-emulated library (--dry): plumbing only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HEAD_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "config", "roofline", "cpu_baseline"}
ROOF_KEYS = {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frames_per_launch", "algorithmic_bytes_per_launch",
             "avg_launch_us", "frac_p50", "frac_min", "frac_max"}


def test_last_line_is_compact_json_with_roofline_and_cpu_baseline():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    last = lines[-1]
    assert len(last) < 1800, f"last line has {len(last)} bytes; the budget is 2 KB for the real run (longer numbers, a traffic source)"
    d = json.loads(last)
    assert HEAD_KEYS <= set(d), HEAD_KEYS - set(d)
    assert ROOF_KEYS <= set(d["roofline"]), ROOF_KEYS - set(d["roofline"])
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"])
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "2 streams" in d["config"]["workload"]                      # `value` is a two-stream figure and the workload says so
    assert d["config"]["passes_per_step"] >= 1 and d["config"]["frames_per_step"] == d["config"]["passes_per_step"] * d["config"]["distinct_frame_pairs"]
    assert d["cpu_baseline"]["kind"] == "port" and "NOT stock libswscale" in d["cpu_baseline"]["note"]
    assert d["dtype"] == "u8" and d["vs_baseline"] is None and d["n_gpus"] == 1
    # no secondary block rides in the last line
    for k in ("chained", "other_configs", "c_harness", "host_pipeline"):
        assert k not in d
    # the detail lines are JSON too, and everything before the last line is a detail line or free text
    for l in lines[:-1]:
        if l.startswith("BENCH_DETAIL "):
            json.loads(l[len("BENCH_DETAIL "):])


def test_real_run_last_line_budget():
    """the non-dry line differs from the dry one by a few longer numbers and the traffic source string: keep 2x headroom"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "print(json.dumps(out), flush=True)" in src and src.rstrip().endswith("main()")
