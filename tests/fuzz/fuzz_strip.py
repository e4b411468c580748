#!/usr/bin/env python3
"""Randomised differential test aimed at the exact-2:1 strip kernels (scale_yuv2s / scale_yuv2p / scale_rgb2s) and the
separable smooth (smooth121_kernel), on and around their eligibility rules: the other fuzzers draw widths and heights
uniformly and essentially never produce an exact 2:1 geometry with an eligible width, so they do not reach these kernels.
Every case is compared with the oracle whatever kernel serves it; the histogram printed at the end shows what was reached.
usage: tests/fuzz/fuzz_strip.py [ncases] [seed] [--hip]"""
import os, sys, random, collections, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import harness
from harness import PIX_FMT, SWS, synth_planes, DevPlane
from gmat_amd.lib import load

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hip = "--hip" in sys.argv
rng = random.Random(seed)
orc = harness.load_oracle(os.path.join(ROOT, "oracle", "liborc.so"))
lib = load() if hip else load(os.environ.get("GMAT_TEST_EMU_LIBRARY") or os.path.join(ROOT, "tests", "hipemu", "build", "libgmat_hip_emu.so"))
dev = harness.Dev(lib, "hip" if hip else "emu")
hist = collections.Counter()
fails = 0
YUV, RGB3, RGBX = ["nv12", "yuv420p"], ["rgb24", "bgr24"], ["rgb24", "bgr24", "rgba", "bgra"]
maxw = 4400 if hip else 1200                       # the emulator is slow on wide frames


def width():
    r = rng.random()
    if r < 0.5:  return 16 * rng.randint(2, maxw // 16)          # multiples of 16
    if r < 0.8:  return 8 * rng.randint(2, maxw // 8)            # multiples of 8 (yuv2s takes them, yuv2p does not)
    return 2 * rng.randint(4, maxw // 2)                          # even only: tiled / generic kernels


for case in range(n):
    os.environ.pop("GMAT_STRIP_ROWS", None)
    if rng.random() < 0.6:
        os.environ["GMAT_STRIP_ROWS"] = str(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 13, 16, 31, 64]))
    # both forms of the 4-pair strip kernel: one frame per call is the block-cooperative form by default (round 4); half the cases
    # keep the walker on it (GMAT_STRIP_BLOCK=0)
    os.environ.pop("GMAT_STRIP_BLOCK", None)
    if rng.random() < 0.5:
        os.environ["GMAT_STRIP_BLOCK"] = "0"
    kind = rng.random()
    if kind < 0.75:
        # ---- 2:1 scale ---------------------------------------------------------------------------------------------
        sw = width()
        sh = 2 * rng.randint(2, 70) if rng.random() < 0.8 else 4 * rng.randint(2, 40)
        if rng.random() < 0.08: sw, sh = sw + 2, sh             # off the exact ratio by construction below
        dw, dh = sw // 2, sh // 2
        if rng.random() < 0.08: dw += rng.choice([-1, 1])       # near 2:1, not exact: generic kernels
        fam = rng.random()
        if fam < 0.45:   sf, df = rng.choice(YUV), rng.choice(RGBX)
        elif fam < 0.8:
            sf = rng.choice(YUV); df = sf if rng.random() < 0.8 else rng.choice(YUV)
            if rng.random() < 0.7:                              # the geometries scale_yuv2p_kernel takes
                sw, sh = 16 * rng.randint(4, maxw // 16), 4 * rng.randint(8, 40); dw, dh = sw // 2, sh // 2
        else:            sf, df = rng.choice(RGB3), rng.choice(RGBX)
        if fam >= 0.45 and fam < 0.8 and rng.random() < 0.12:   # 4:2:0 -> planar 4:4:4 (the luma walker + a chroma re-layout at exactly 2:1)
            df = "yuv444p"
        if fam >= 0.45 and fam < 0.8 and df != "yuv444p" and rng.random() < 0.35:   # the 10-bit twins of the 4:2:0 -> 4:2:0 family
            fam4 = ["nv12", "yuv420p", "p010le", "yuv420p10le"]                 # every pairing of chroma layout and sample depth
            sf, df = rng.choice(fam4), rng.choice(fam4)
        if rng.random() < 0.12:                                 # the exact 1:2 UP-scale family (scale_yuv1x2_kernel) and near misses
            sf = rng.choice(YUV); df = sf if rng.random() < 0.85 else rng.choice(YUV)
            sw = rng.choice([8 * rng.randint(4, maxw // 16), 4 * rng.randint(8, maxw // 8)]); sh = rng.choice([2 * rng.randint(8, 50), rng.randint(16, 99)])
            dw, dh = 2 * sw, 2 * sh
            if rng.random() < 0.06: dh += 1
        if rng.random() < 0.12:                                 # the exact 3:1 down-scale family (scale_yuv3x1_kernel) and near misses
            sf = rng.choice(YUV); df = sf if rng.random() < 0.85 else rng.choice(YUV)
            dw = rng.choice([8 * rng.randint(4, maxw // 24), 4 * rng.randint(8, maxw // 12)]); dh = rng.choice([2 * rng.randint(6, 40), rng.randint(8, 60)])
            sw, sh = 3 * dw, 3 * dh
            if rng.random() < 0.06: sh += 1
        if rng.random() < 0.08:                                 # NV12 at exactly 3:1 into packed RGB (scale_yuv3r_kernel) and near misses
            sf = "nv12" if rng.random() < 0.85 else "yuv420p"; df = rng.choice(RGBX)
            dw = rng.choice([4 * rng.randint(8, maxw // 12), 2 * rng.randint(16, maxw // 6)]); dh = rng.choice([2 * rng.randint(6, 40), rng.randint(8, 60)])
            sw, sh = 3 * dw, 3 * dh
            if rng.random() < 0.06: sh += 1
        if rng.random() < 0.08:                                 # NV12 at exactly 3:2 into packed RGB (scale_yuv32r_kernel) and near misses
            sf = "nv12" if rng.random() < 0.85 else "yuv420p"; df = rng.choice(RGBX)
            dw = rng.choice([8 * rng.randint(8, maxw // 12), 4 * rng.randint(16, maxw // 6)]); dh = rng.choice([4 * rng.randint(4, 30), 2 * rng.randint(8, 60)])
            sw, sh = 3 * dw // 2, 3 * dh // 2
            if rng.random() < 0.06: sh += 2
        if rng.random() < 0.06:                                 # NV12 at exactly 4:1 into packed RGB (scale_yuv4r_kernel) and near misses
            sf = "nv12" if rng.random() < 0.85 else "yuv420p"; df = rng.choice(RGBX)
            dw = rng.choice([4 * rng.randint(8, maxw // 16), 2 * rng.randint(16, maxw // 8)]); dh = rng.randint(6, 40)
            sw, sh = 4 * dw, 4 * dh
            if rng.random() < 0.06: sh += 2
        if rng.random() < 0.06:                                 # the exact 4:1 down-scale family (scale_yuv4x1_kernel) and near misses
            sf = rng.choice(YUV); df = sf if rng.random() < 0.85 else rng.choice(YUV)
            dw = rng.choice([8 * rng.randint(8, maxw // 32), 4 * rng.randint(16, maxw // 16)]); dh = rng.choice([2 * rng.randint(8, 30), rng.randint(12, 50)])
            sw, sh = 4 * dw, 4 * dh
            if rng.random() < 0.06: sh += 2
        if rng.random() < 0.12:                                 # the exact 3:2 down-scale family (scale_yuv3x2_kernel) and near misses
            sf = rng.choice(YUV); df = sf if rng.random() < 0.85 else rng.choice(YUV)
            dw = rng.choice([16 * rng.randint(4, maxw // 24), 8 * rng.randint(8, maxw // 12)]); dh = rng.choice([4 * rng.randint(4, 30), 2 * rng.randint(8, 60)])
            sw, sh = 3 * dw // 2, 3 * dh // 2
            if rng.random() < 0.06: sh += 2
        if rng.random() < 0.08:                                 # same-size packed RGB -> 4:2:0 (rgb2yuv420s_kernel) and near misses
            sf = rng.choice(RGB3); df = rng.choice(["nv12", "yuv420p"])
            sw = rng.choice([8 * rng.randint(8, maxw // 8), 4 * rng.randint(16, maxw // 4)]); sh = rng.choice([2 * rng.randint(8, 60), rng.randint(8, 99)])
            dw, dh = sw, sh
        if rng.random() < 0.1:                                  # packed RGB -> 4:2:0 at exactly 2:1 (scale_rgb2y_kernel) and near misses
            sf = rng.choice(RGB3); df = rng.choice(["nv12", "yuv420p"])
            dw = rng.choice([4 * rng.randint(16, maxw // 8), 2 * rng.randint(32, maxw // 4)]); dh = rng.choice([2 * rng.randint(8, 50), rng.randint(14, 80)])
            sw, sh = 2 * dw, 2 * dh
            if rng.random() < 0.06: sh += 2
        algo = rng.choice(["bicubic", "bicubic", "bilinear", "point", "area", "gauss", "fast_bilinear", "lanczos"])
        cs = rng.choice([None, None, 1, 5, 7]) if df in RGBX and sf in YUV else None
        # the fused convert-then-scale form (setFused(1): scale_rgb2h_kernel<yuv> at exactly 2:1, the tiled kernel otherwise)
        fused = 1 if (df in RGBX and sf in ("nv12", "yuv420p") and cs is None and rng.random() < 0.35) else None
        align, extra = rng.choice([(256, 0), (64, 0), (16, 0), (8, 0), (4, 0), (4, 4), (1, 1), (2, 2)])
        try:
            synth = synth_planes(orc, sf, sw, sh, seed=7000 + case)
            if sf == "yuv420p10le":
                for pl in synth: pl.view("<u2")[...] &= 0x3FF        # valid input: 10 bits in the low end
            if (sf in ("p010le", "yuv420p10le") or df in ("p010le", "yuv420p10le")) and align < 2: align, extra = 2, 2   # 16-bit samples
            want = (orc.chained(synth, sw, sh, sf, dw, dh, df, SWS[algo]) if fused else
                    orc.sws(synth, sw, sh, sf, dw, dh, df, SWS[algo], colorspace=cs))
        except AssertionError:
            continue
        d = dev.upload_planes(synth, align, extra)
        try:
            got, pads, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, SWS[algo], dst_align=align, dst_extra=extra, fused=fused,
                                        colorspace=None if cs is None else (cs, 0))
        except AssertionError as e:
            if "-38" in str(e) or "getContext" in str(e):
                hist["(declined)"] += 1
                for p in d: p.free()
                continue
            raise
        for p in d: p.free()
        hist[kernel if kernel.startswith(("scale_yuv2p", "scale_yuv2s", "scale_yuv1x2", "scale_yuv3x1", "scale_yuv3r", "scale_yuv3x2", "scale_yuv32r", "scale_yuv4r", "scale_yuv4x1", "scale_rgb2h", "scale_rgb2y", "rgb2yuv420")) else kernel.split("<")[0]] += 1
        bad = sum(int((g != w).sum()) for g, w in zip(got, want)) + sum(int((pd != 0xCD).sum()) for pd in pads)
        if bad:
            fails += 1
            print("MISMATCH case", case, sf, "->", df, (sw, sh, dw, dh), algo, "cs", cs, "align", (align, extra), "rows",
                  os.environ.get("GMAT_STRIP_ROWS"), "block", os.environ.get("GMAT_STRIP_BLOCK"), kernel, "bad bytes", bad)
    else:
        # ---- 1 2 1 smooth, plain and fused with rotate + flip -----------------------------------------------------------
        bpp = rng.choice([1, 2, 3, 4]); fused = rng.random() < 0.5 and bpp >= 3
        w = rng.choice([4 * rng.randint(1, 90), rng.randint(1, 360)]); h = rng.randint(1, 150)
        src = orc.lcg((h, w * bpp), 9000 + case)
        m = (C.c_int * 9)(1, 2, 1, 2, 4, 2, 1, 2, 1)
        align, extra = rng.choice([(256, 0), (16, 0), (4, 0), (4, 4), (1, 3)])
        d = dev.upload_planes([src], align, extra)[0]
        if fused:
            a = np.zeros((w, h * bpp), np.uint8); b = np.zeros((w, h * bpp), np.uint8); want = np.zeros((w, h * bpp), np.uint8)
            orc.L.orc_transpose(src.ctypes.data, src.strides[0], a.ctypes.data, a.strides[0], w, h, bpp, 1)
            orc.L.orc_hflip(a.ctypes.data, a.strides[0], b.ctypes.data, b.strides[0], h, w, bpp)
            orc.L.orc_conv3x3(b.ctypes.data, b.strides[0], want.ctypes.data, want.strides[0], h, w, bpp, m, 1 / 16, 0.0)
            o = DevPlane(dev, w, h * bpp, (h * bpp + extra + align - 1) // align * align)
            r = lib.gmat_rotate_flip_smooth(d.ptr, d.stride, o.ptr, o.stride, w, h, bpp, None)
            hist["rotate_flip_smooth"] += 1
        else:
            want = np.zeros((h, w * bpp), np.uint8)
            orc.L.orc_conv3x3(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, bpp, m, 1 / 16, 0.0)
            o = DevPlane(dev, h, w * bpp, (w * bpp + extra + align - 1) // align * align)
            r = lib.gmat_smooth3x3(d.ptr, d.stride, o.ptr, o.stride, w, h, bpp, m, 1 / 16, 0.0, None)
            hist["smooth3x3"] += 1
        ok = r == 0 and (o.download() == want).all() and (o.download(True)[:, o.row_bytes:] == 0xCD).all()
        # which of the two kernels served it is the host rule of smooth121_ok(): whole dwords per row, >= 4 pixels, dword-aligned
        hist["  of which separable (smooth121_kernel)"] += int((w * bpp) % 4 == 0 and w >= 4 and align % 4 == 0 and extra % 4 == 0)
        d.free(); o.free()
        if not ok:
            fails += 1
            print("MISMATCH case", case, "fused" if fused else "smooth", (w, h, bpp), "align", (align, extra), "rc", r)
os.environ.pop("GMAT_STRIP_ROWS", None)
os.environ.pop("GMAT_STRIP_BLOCK", None)
for k, v in sorted(hist.items(), key=lambda kv: -kv[1]):
    print(f"   {v:6d}  {k}")
print("cases", n, "failures", fails)
sys.exit(1 if fails else 0)
