#!/usr/bin/env python3
"""Randomised differential test of the SAME-SIZE conversions between the YUV depths and layouts and into RGBA64 / BGRA64 (the tile kernel's unit forms,
k_scale19.hip: scale19_unit_kernel, scale19_unit64_kernel, unit_rgb_kernel — and whatever takes the draws they refuse: differing chroma positions, odd planes, ragged widths)
against the oracle: every format pair (deep 4:2:0 sources into packed 8-bit RGB too), range conversions, chroma positions, plane alignments, filter families.
usage: tests/fuzz/fuzz_unit.py [ncases] [seed] [--hip]"""
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
kernels = {}
YUV = ["nv12", "yuv420p", "p010le", "p016le", "yuv420p10le", "yuv420p16le", "yuv444p", "yuv444p16le"]
DEEP = ("p010le", "p016le", "yuv420p10le", "yuv420p16le", "yuv444p16le", "rgba64le", "bgra64le")
for case in range(n):
    sf = rng.choice(YUV)
    df = rng.choice(YUV + ["rgba64le", "bgra64le", "rgba64le"] + (["rgb24", "bgr24", "rgba", "bgra"] if sf in ("p010le", "p016le", "yuv420p10le", "yuv420p16le") else []))
    sw = rng.choice([rng.randint(2, 400), 8 * rng.randint(1, 50)])
    sh = rng.randint(2, 90)
    dw, dh = sw, sh
    flags = SWS[rng.choice(["bicubic", "bilinear", "lanczos", "point", "area"])]
    pos = [rng.choice([-513, -513, 0, 64, 128, 192, 256, rng.randint(-512, 512)]) for _ in range(4)]
    use_pos = rng.random() < 0.3
    if use_pos and rng.random() < 0.5: pos[2], pos[3] = pos[0], pos[1]          # the same at both ends: identities again
    if not use_pos: pos = [-513] * 4
    sr, dr = (rng.randint(0, 1), rng.randint(0, 1)) if df in YUV else (0, 0)
    if sf == "yuv420p" and df in ("p010le", "p016le") and sr == dr:
        continue                                   # planar8ToP01xleWrapper: its own oracle entry (tests/test_parity_rgb2yuv.py)
    if sf == df and sr == dr:
        continue                                   # plane copy
    if sr == 1 and dr == 1 and (sf, df) in (("yuv444p", "yuv444p16le"), ("yuv420p", "yuv420p16le"), ("yuv420p", "yuv420p10le")):
        continue                                   # planarCopyWrapper's bit-replicated full-range luma: orc_plane_copy_up is its oracle (tests/test_parity_scale.py), not orc_sws_scale
    oc = L.orc_sws_create_ex(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], flags, None, (C.c_int * 4)(*pos), sr, dr)
    if not oc:
        skipped += 1; continue
    src = synth_planes(orc, sf, sw, sh, seed=2000 + case)
    if sf == "yuv420p10le":
        for p in src: p.view("<u2")[...] &= 0x3FF
    if sf == "p010le":
        for p in src: p.view("<u2")[...] &= 0xFFC0
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
    if (sf in DEEP or df in DEEP) and align == 1:
        align, extra = 2, 2                        # rows of 16-bit samples are at least 2-byte aligned
    d = dev.upload_planes(src, align, extra)
    dst = dev.planes_like(df, dw, dh, align, extra)
    r = lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, sh,
                           planes([p.ptr for p in dst]), ints([p.stride for p in dst]))
    k = lib.gmat_sws_lastKernel(c).decode()
    kernels[k] = kernels.get(k, 0) + 1
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
for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]): print("%6d  %s" % (v, k))
print("cases", n, "skipped", skipped, "failures", fails)
sys.exit(1 if fails else 0)
