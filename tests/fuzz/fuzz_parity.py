#!/usr/bin/env python3
"""Randomised differential test: the kernel sources (on the CPU emulation of HIP, or on the GPU with --hip) against
the oracle over random geometries, formats, flags, strides.  usage: tests/fuzz/fuzz_parity.py [ncases] [seed] [--hip]"""
import os, sys, random, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import harness
from harness import PIX_FMT, SWS, synth_planes
from gmat_amd.lib import load

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hip = "--hip" in sys.argv
rng = random.Random(seed)
orc = harness.load_oracle(os.path.join(ROOT, "oracle", "liborc.so"))
lib = load() if hip else load(os.environ.get("GMAT_TEST_EMU_LIBRARY") or os.path.join(ROOT, "tests", "hipemu", "build", "libgmat_hip_emu.so"))
dev = harness.Dev(lib, "hip" if hip else "emu")

SRC = ["nv12", "yuv420p", "rgb24", "bgr24", "yuv444p", "rgba", "bgra", "rgba64le", "bgra64le"]   # the four with alpha: their alpha plane is scaled into rgba / bgra / rgba64le / bgra64le
ALGOS = ["bicubic", "bilinear", "lanczos", "point", "area", "fast_bilinear", "bicublin", "x", "gauss"]
fails = 0
for case in range(n):
    sf = rng.choice(SRC)
    dsts = ["rgb24", "bgr24", "rgba", "bgra", "nv12", "yuv420p", "yuv444p"]
    if sf in ("rgba64le", "bgra64le"): dsts += ["rgba64le", "bgra64le", "p016le", "yuv444p16le", "p010le"]
    if sf in ("rgb24", "bgr24", "rgba", "bgra"): dsts += ["rgba64le", "bgra64le", "p016le", "yuv444p16le", "yuv420p16le"]
    df = rng.choice(dsts)
    sw, sh = rng.randint(2, 300), rng.randint(2, 120)
    same = rng.random() < 0.25
    dw, dh = (sw, sh) if same else (rng.randint(2, 300), rng.randint(2, 120))
    same = same or (dw, dh) == (sw, sh)           # (the draw can hit the source's size by itself — seed 9300 case 1606: libswscale's unscaled converters apply, not orc.sws's generic lines)
    algo = rng.choice(ALGOS)
    flags = SWS[algo]
    if rng.random() < 0.2: flags |= SWS["full_chr_h_int"]
    if rng.random() < 0.2: flags |= SWS["accurate_rnd"]
    align, extra = rng.choice([(256, 0), (64, 0), (16, 0), (4, 0), (1, 1), (1, 3), (2, 2)])
    if ("64le" in sf or "16le" in df or "10le" in df or "64le" in df) and (align, extra) in ((1, 1), (1, 3)):
        align, extra = 2, 2                       # 16-bit samples: rows on even addresses (anything else is -EINVAL)
    src = synth_planes(orc, sf, sw, sh, seed=1000 + case)
    c = orc.L.orc_sws_create(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], flags, None)
    if not c:
        continue
    orc.L.orc_sws_free(c)
    ctx = lib.gmat_sws_getContext(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], flags | SWS["hwaccel"], None)
    if not ctx:
        continue                                  # geometry the kernels decline (ENOSYS): not a parity failure
    lib.gmat_sws_freeContext(ctx)
    # which oracle composition applies
    fused = None
    if same and sf in ("nv12", "yuv420p") and df in ("rgb24", "bgr24", "rgba", "bgra") and not (flags & SWS["accurate_rnd"]):
        want = [orc.yuv2rgb(src, sw, sh, sf, df)]
    elif same and sf in ("nv12", "yuv420p") and df in ("nv12", "yuv420p"):
        continue                                  # lossless re-layout, covered elsewhere
    elif same and sf in ("rgb24", "bgr24", "rgba", "bgra") and df in ("rgb24", "bgr24", "rgba", "bgra"):
        continue                                  # copies / swaps, covered elsewhere
    elif same and sf == df:
        continue                                  # the plain copy
    else:
        want = orc.sws(src, sw, sh, sf, dw, dh, df, flags)
    d = dev.upload_planes(src, align, extra)
    try:
        got, pads, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, flags, dst_align=align, dst_extra=extra)
    except AssertionError as e:
        print("CASE", case, sf, df, (sw, sh, dw, dh), algo, hex(flags), (align, extra), "->", e); fails += 1
        continue
    finally:
        for p in d: p.free()
    ok = all((g == w).all() for g, w in zip(got, want)) and all((p == 0xCD).all() for p in pads)
    if not ok:
        fails += 1
        bad = [int((g != w).sum()) for g, w in zip(got, want)]
        print("MISMATCH case", case, sf, "->", df, (sw, sh, dw, dh), algo, hex(flags), "align", (align, extra), kernel, "bad bytes", bad)
print("cases", n, "failures", fails)
sys.exit(1 if fails else 0)
