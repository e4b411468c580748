#!/usr/bin/env python3
"""Randomised differential test aimed at the quad-lane walker of up-scales (k_scale_yuvu.hip, scale_yuvu_kernel: ratios down to 1 : 6.5, GMAT_QUAD_WALKER
0 / 1 / 2) and at the any-ratio band walker (k_scale_yuvg.hip) in BOTH forms — the register-window walker
(scale_yuvg_kernel) and the block-cooperative form of small launches (scale_yuvg_blk_kernel, round 4) — against the oracle: random
ratios between 1 : 1.9 and 6 : 1 on each axis independently (anamorphic included), every SWS algorithm, both source layouts, packed RGB
and 4:2:0 destinations, random band heights (GMAT_STRIP_ROWS), random launch sizes through gmat_sws_scale_batch (1 .. 5 frames), widths
on and off the 64-column strips, pitches of 4 .. 256 bytes.  The other fuzzers draw geometries that mostly land on the tiled kernels.
Every case is compared with the oracle whatever kernel serves it; the histogram printed at the end shows what was reached.
usage: tests/fuzz/fuzz_walker.py [ncases] [seed] [--hip]"""
import os, sys, random, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness
from harness import SWS
from gmat_amd.lib import load
from test_batch_api import _run_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hip = "--hip" in sys.argv
rng = random.Random(seed)
orc = harness.load_oracle(os.path.join(ROOT, "oracle", "liborc.so"))
lib = load() if hip else load(os.environ.get("GMAT_TEST_EMU_LIBRARY") or os.path.join(ROOT, "tests", "hipemu", "build", "libgmat_hip_emu.so"))
dev = harness.Dev(lib, "hip" if hip else "emu")
hist = collections.Counter()
fails = 0
maxw, maxh = (2600, 400) if hip else (900, 160)
ALGOS = ["bicubic", "bicubic", "bicubic", "bilinear", "lanczos", "point", "area", "gauss", "fast_bilinear", "sinc", "spline", "bicublin", "x"]

for case in range(n):
    for k in ("GMAT_STRIP_ROWS", "GMAT_STRIP_BLOCK", "GMAT_SCALE_NO_STRIP", "GMAT_BLOCK_FIRST", "GMAT_QUAD_WALKER", "GMAT_SCALE_NO_WALKER16", "GMAT_RGBSRC_WALKER", "GMAT_RGBSRC_BLOCK", "GMAT_RGBSRC_FUSED", "GMAT_RGBSRC_NO_PX4"):
        os.environ.pop(k, None)
    q = rng.random()
    if q < 0.15:   os.environ["GMAT_QUAD_WALKER"] = "0"          # up-scales on the band walker / the tiled kernel
    elif q < 0.45: os.environ["GMAT_QUAD_WALKER"] = "2"          # the quad-lane walker wherever it is eligible (short-filter down-scales too)
    if rng.random() < 0.5:
        os.environ["GMAT_STRIP_ROWS"] = str(rng.choice([1, 3, 4, 5, 8, 12, 13, 16, 20, 24, 31, 32, 64]))
    r = rng.random()
    if r < 0.35:   os.environ["GMAT_STRIP_BLOCK"] = "0"          # the walker at every launch size
    elif r < 0.6:  os.environ["GMAT_STRIP_BLOCK"] = "32"         # the block form at every launch size
    if rng.random() < 0.7:
        os.environ["GMAT_SCALE_NO_STRIP"] = "1"                  # exact ratios reach the walker too
    sf = rng.choice(["nv12", "yuv420p"])
    df = rng.choice(["rgb24", "bgr24", "rgba", "bgra", sf, sf, "yuv420p" if sf == "nv12" else "nv12"])      # (the other chroma layout: round 4's cascade, GMAT_NO_CROSS_CASCADE)
    if rng.random() < 0.3:
        # round 5: 16-bit samples in (the band walker of k_scale_yuvg16.hip), 10-bit samples out (its output stage, the 8-bit walker's too)
        semi = rng.random() < 0.5
        if rng.random() < 0.75:
            sf = rng.choice(["p010le", "p010le", "p016le"] if semi else ["yuv420p10le", "yuv420p10le", "yuv420p16le"])
        else:
            sf = "nv12" if semi else "yuv420p"
        df = rng.choice(["rgb24", "bgr24", "rgba", "bgra"] + 3 * ["nv12" if semi else "yuv420p"] + 3 * ["p010le" if semi else "yuv420p10le"])
        if rng.random() < 0.15:
            os.environ["GMAT_SCALE_NO_WALKER16"] = "1"
    elif rng.random() < (0.8 if "--rgbsrc" in sys.argv else 0.12):      # (--rgbsrc: mostly packed RGB sources — the block-cooperative kernels of round 5's second half)
        # round 5: a packed RGB source into a 4:2:0 frame (the 16-bit walker's converter; up-scales and odd widths: whatever serves them)
        sf = rng.choice(["rgb24", "bgr24", "rgb24", "bgr24", "rgba", "bgra"])      # (RGBA / BGRA: read as they are by the block-cooperative kernels, alpha -> alpha as a fourth line)
        df = rng.choice(["nv12", "yuv420p", "nv12", "yuv420p", "p010le", "rgb24", "bgr24", "rgba", "bgra"])     # (RGB -> RGB: scale_yuvg_rgbsrc_kernel from four frames a launch on)
        if rng.random() < 0.5:
            os.environ["GMAT_RGBSRC_WALKER"] = "2"                # ... or at every launch size
        if rng.random() < 0.6:
            os.environ["GMAT_RGBSRC_FUSED"] = rng.choice(["0", "1", "1"])    # an RGB source into a 4:2:0 frame: never / always on the fused block form (the rule: from four frames a launch on, up-scales always)
        if rng.random() < 0.15:
            os.environ["GMAT_RGBSRC_NO_PX4"] = "1"                # (RGBA sources through the 32 -> 24-bit pass, as before round 5)
        if rng.random() < 0.4:
            os.environ["GMAT_RGBSRC_BLOCK"] = "0"                 # (the block-cooperative form, scale_yuvg_rgbsrc_blk_kernel, is the rule wherever it has an instance: without it)
    os.environ.pop("GMAT_NO_CROSS_CASCADE", None)
    if rng.random() < 0.2:
        os.environ["GMAT_NO_CROSS_CASCADE"] = "1"
    dw = 2 * rng.randint(8, maxw // 8)
    dh = 2 * rng.randint(4, maxh // 6)
    if rng.random() < 0.3:
        dw = 64 * rng.randint(1, 8) + rng.choice([0, 0, 2, 62])   # on / just past / just short of the strips
    rx = rng.choice([rng.uniform(0.55, 1.0), rng.uniform(0.15, 1.0), rng.uniform(1.0, 2.0), rng.uniform(1.0, 3.0), rng.uniform(3.0, 6.0)])
    ry = rx * rng.uniform(0.8, 1.25) if rng.random() < 0.7 else rng.choice([rng.uniform(0.55, 1.0), rng.uniform(0.15, 1.0), rng.uniform(1.0, 6.0)])
    sw = max(16, min(maxw if sf in ("nv12", "yuv420p") else maxw // 2, 4 * int(dw * rx / 4)))       # (16-bit samples, packed pixels: half the width for the same oracle time)
    sh = max(8, min(maxh, 2 * int(dh * ry / 2)))
    if (sw, sh) == (dw, dh):
        hist["(same size: the converter's semantics, tests/test_parity_yuv2rgb.py)"] += 1    # nearest-chroma yuv2rgb.c by design, not orc.sws's generic lines
        continue
    algo = rng.choice(ALGOS)
    if algo not in SWS:
        algo = "bicubic"
    nframes = rng.choice([1, 1, 1, 2, 3, 4, 5])
    align = rng.choice([4, 8, 16, 64, 256])
    ctx = lib.gmat_sws_getContext(sw, sh, harness.PIX_FMT[sf], dw, dh, harness.PIX_FMT[df], SWS[algo] | SWS["hwaccel"], None)
    if not ctx:
        hist["(declined: -ENOSYS)"] += 1              # filters beyond what any kernel takes (sinc / spline at high ratios): not a parity failure
        continue
    lib.gmat_sws_freeContext(ctx)
    try:
        k = _run_batch(dev, orc, sf, df, sw, sh, dw, dh, nframes=nframes, nstreams=rng.choice([1, 2]), align=align, flags=SWS[algo])
        hist[k] += 1
    except AssertionError as e:
        fails += 1
        print("MISMATCH case", case, sf, "->", df, (sw, sh, dw, dh), algo, "frames", nframes, "align", align,
              {k: os.environ.get(k) for k in ("GMAT_STRIP_ROWS", "GMAT_STRIP_BLOCK", "GMAT_SCALE_NO_STRIP", "GMAT_QUAD_WALKER")}, "->", str(e)[:300])
for k, v in hist.most_common():
    print("%6d  %s" % (v, k))
print("cases", n, "failures", fails)
sys.exit(1 if fails else 0)
