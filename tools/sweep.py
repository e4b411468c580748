#!/usr/bin/env python3
"""Sweeps environment knobs of a kernel over tools/bin/x2bench (JSON mode) and prints one tidy line per point.
usage: tools/sweep.py "<case substring>" --nf 1,2,4 --env GMAT_STRIP_PF=1,3 --env GMAT_STRIP_ROWS=2,3,4 [--reps 2] [--out file]
A value `-` leaves the variable unset (the launcher's own rule)."""
import argparse, itertools, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case")
    ap.add_argument("--nf", default="32")
    ap.add_argument("--env", action="append", default=[])
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--launches", type=int, default=40)
    ap.add_argument("--out")
    a = ap.parse_args()
    knobs = [(e.split("=")[0], e.split("=")[1].split(",")) for e in a.env]
    out = open(a.out, "a") if a.out else None
    for nf in [int(x) for x in a.nf.split(",")]:
        for combo in itertools.product(*[v for _, v in knobs]):
            env = dict(os.environ, X2BENCH_JSON="1", X2BENCH_VERIFY="0")
            for (k, _), v in zip(knobs, combo):
                env.pop(k, None)
                if v != "-":
                    env[k] = v
            best = None
            for _ in range(a.reps):
                r = subprocess.run([os.path.join(ROOT, "tools", "bin", "x2bench"), str(nf), str(a.launches), a.case],
                                   env=env, capture_output=True, text=True, timeout=120)
                rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
                if not rows:
                    print("no output:", r.stderr[-300:], file=sys.stderr)
                    continue
                d = rows[0]
                if best is None or d["us_per_launch"] < best["us_per_launch"]:
                    best = d
            if best:
                line = (f"nf={nf:<3d} " + " ".join(f"{k}={v}" for (k, _), v in zip(knobs, combo)) +
                        f"  us_per_launch={best['us_per_launch']:.2f} us_per_frame={best['us_per_frame']:.3f} frac={best['frac']:.3f} kernel={best['kernel']}")
                print(line, flush=True)
                if out:
                    out.write(line + "\n"); out.flush()


if __name__ == "__main__":
    main()
