#!/usr/bin/env python3
"""Randomised differential test of the YUV plane scaler's options (range conversion, chroma positions, 4:4:4
sources) against the oracle.  usage: tests/fuzz/fuzz_yuvopts.py [ncases] [seed] [--hip]"""
import os, sys, random, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import harness
from harness import PIX_FMT, SWS, synth_planes, planes, ints, alloc_planes
from gmat_amd.lib import load

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hip = "--hip" in sys.argv
rng = random.Random(seed)
orc = harness.load_oracle(os.path.join(ROOT, "oracle", "liborc.so"))
lib = load() if hip else load(os.environ.get("GMAT_TEST_EMU_LIBRARY") or os.path.join(ROOT, "tests", "hipemu", "build", "libgmat_hip_emu.so"))
dev = harness.Dev(lib, "hip" if hip else "emu")
L = orc.L
L.orc_sws_create_ex.restype = C.c_void_p
L.orc_sws_create_ex.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
fails = skipped = 0
for case in range(n):
    sf = rng.choice(["nv12", "yuv420p", "yuv444p", "p010le", "p016le", "yuv444p16le"])
    df = rng.choice(["nv12", "yuv420p", "yuv444p", "p010le", "p016le", "yuv444p16le", "rgba64le", "rgb24", "bgra"])
    sw, sh = rng.randint(2, 300), rng.randint(2, 120)
    dw, dh = (sw, sh) if rng.random() < 0.3 else (rng.randint(2, 300), rng.randint(2, 120))
    flags = SWS[rng.choice(["bicubic", "bilinear", "lanczos", "point", "area"])]
    pos = [rng.choice([-513, -513, 0, 64, 128, 192, 256, rng.randint(-512, 512)]) for _ in range(4)]
    use_pos = rng.random() < 0.5
    if not use_pos: pos = [-513] * 4
    sr, dr = (rng.randint(0, 1), rng.randint(0, 1)) if df in ("nv12", "yuv420p", "yuv444p", "p010le", "p016le", "yuv444p16le") else (0, 0)
    if df in ("p010le", "p016le") and (sw, sh) == (dw, dh) and sf in ("nv12", "yuv420p", df) and sr == dr:
        continue                                   # depth-expansion converter / plane copy, covered elsewhere
    if df in ("p016le", "yuv444p16le", "rgba64le"):
        if df == "rgba64le": sr = dr = 0           # (an RGB end has no range)
        if sf == df and (sw, sh) == (dw, dh) and sr == dr:
            continue                               # plane copy
    if (sw, sh) == (dw, dh) and sr == dr and ((sf, df) in (("yuv444p", "yuv444p16le"), ("yuv420p", "yuv420p16le"), ("yuv420p", "yuv420p10le"))):
        continue                                   # planarCopyWrapper (equal ranges): tests/test_parity_scale.py
    if sf in ("nv12", "yuv420p") and (sw, sh) == (dw, dh) and sr == dr and not use_pos and df in ("nv12", "yuv420p"):
        continue
    if sf in ("nv12", "yuv420p") and (sw, sh) == (dw, dh) and df in ("rgb24", "bgra"):
        continue                                   # the unscaled converter, covered elsewhere
    oc = L.orc_sws_create_ex(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], flags, None, (C.c_int * 4)(*pos), sr, dr)
    if not oc:
        skipped += 1; continue
    src = synth_planes(orc, sf, sw, sh, seed=2000 + case)
    want = alloc_planes(df, dw, dh)
    r = L.orc_sws_scale(oc, planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                        planes([p.ctypes.data for p in want]), ints([p.strides[0] for p in want]))
    L.orc_sws_free(oc)
    assert r == dh
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], flags | SWS["hwaccel"], None)
    if not c:
        skipped += 1; continue
    ok_cfg = True
    if sr != dr or (sr and dr):
        ok_cfg = lib.gmat_sws_setRange(c, sr, dr) == 0
    if ok_cfg and use_pos:
        ok_cfg = lib.gmat_sws_setChromaPos(c, *pos) == 0
    if not ok_cfg:
        lib.gmat_sws_freeContext(c); skipped += 1; continue
    align, extra = rng.choice([(256, 0), (16, 0), (4, 0), (1, 1), (2, 2)])
    if (sf in ("p010le", "p016le", "yuv444p16le") or df in ("p010le", "p016le", "yuv444p16le", "rgba64le")) and align == 1:
        align, extra = 2, 2                        # rows of 16-bit samples are at least 2-byte aligned
    d = dev.upload_planes(src, align, extra)
    dst = dev.planes_like(df, dw, dh, align, extra)
    r = lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, sh,
                           planes([p.ptr for p in dst]), ints([p.stride for p in dst]))
    k = lib.gmat_sws_lastKernel(c).decode()
    if r == -38:
        skipped += 1
    elif r != dh:
        fails += 1; print("ERROR", case, sf, df, (sw, sh, dw, dh), hex(flags), pos, (sr, dr), "->", r)
    else:
        got = [p.download() for p in dst]
        pads = [p.download(True)[:, p.row_bytes:] for p in dst]
        if not (all((g == w).all() for g, w in zip(got, want)) and all((p == 0xCD).all() for p in pads)):
            fails += 1
            print("MISMATCH", case, sf, "->", df, (sw, sh, dw, dh), hex(flags), "pos", pos, "range", (sr, dr), "align", (align, extra), k,
                  [int((g != w).sum()) for g, w in zip(got, want)])
    lib.gmat_sws_freeContext(c)
    for p in d + dst: p.free()
print("cases", n, "skipped", skipped, "failures", fails)
sys.exit(1 if fails else 0)
