#!/usr/bin/env python3
"""Randomised differential test of the AVFilter-shaped layer (frame pools, upload / download, option parsing, per-
plane filtering of 4:2:0 frames) against the oracle.  usage: tests/fuzz/fuzz_filters.py [ncases] [seed] [--hip]"""
import os, sys, random, math, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import harness
from harness import PIX_FMT, SWS, synth_planes, plane_shapes
from gmat_amd.lib import load, GmatFrame

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hip = "--hip" in sys.argv
rng = random.Random(seed)
orc = harness.load_oracle(os.path.join(ROOT, "oracle", "liborc.so"))
lib = load() if hip else load(os.environ.get("GMAT_TEST_EMU_LIBRARY") or os.path.join(ROOT, "tests", "hipemu", "build", "libgmat_hip_emu.so"))
L = orc.L
NAMES = {v: k for k, v in PIX_FMT.items()}


def run_filter(name, opts, src, w, h, fmt):
    fc = lib.gmat_hwframe_ctx_create(0, PIX_FMT[fmt], w, h, 1)
    f = lib.gmat_filter_alloc(name.encode())
    assert fc and f
    for k, v in opts.items():
        assert lib.gmat_filter_set_option(f, k.encode(), str(v).encode()) == 0, (k, v)
    r = lib.gmat_filter_init(f)
    if r == 0:
        r = lib.gmat_filter_config_props(f, fc, None)
    if r < 0:
        lib.gmat_filter_free(f); lib.gmat_hwframe_ctx_free(fc)
        return None
    host = GmatFrame()
    assert lib.gmat_host_frame_alloc(C.byref(host), PIX_FMT[fmt], w, h) == 0
    for i, pl in enumerate(src):
        hv = np.ctypeslib.as_array(C.cast(host.data[i], C.POINTER(C.c_uint8)), (pl.shape[0], host.linesize[i]))
        hv[:, :pl.shape[1]] = pl
    fin = lib.gmat_frame_alloc()
    assert lib.gmat_hwframe_get_buffer(fc, fin) == 0
    assert lib.gmat_hwframe_transfer_data(fin, C.byref(host), None) == 0
    out = C.POINTER(GmatFrame)()
    r = lib.gmat_filter_frame(f, fin, C.byref(out))
    assert r == 0 and out, r
    o = out.contents
    hout = GmatFrame()
    assert lib.gmat_host_frame_alloc(C.byref(hout), o.sw_format, o.width, o.height) == 0
    assert lib.gmat_hwframe_transfer_data(C.byref(hout), out, None) == 0
    lib.gmat_device_sync()
    ofmt = NAMES[o.sw_format]
    res = []
    for i, (rows, rb) in enumerate(plane_shapes(ofmt, o.width, o.height)):
        res.append(np.ctypeslib.as_array(C.cast(hout.data[i], C.POINTER(C.c_uint8)), (rows, hout.linesize[i]))[:, :rb].copy())
    ow, oh = o.width, o.height
    lib.gmat_frame_free(C.byref(out))
    lib.gmat_host_frame_free(C.byref(host)); lib.gmat_host_frame_free(C.byref(hout))
    lib.gmat_filter_free(f); lib.gmat_hwframe_ctx_free(fc)
    return res, ow, oh, ofmt


def plane_bpp(fmt, i):
    if fmt in ("rgb24", "bgr24"): return 3
    if fmt in ("rgba", "bgra"): return 4
    return 2 if fmt == "nv12" and i == 1 else 1


fails = skipped = 0
for case in range(n):
    fmt = rng.choice(["rgb24", "bgr24", "rgba", "bgra", "nv12", "yuv420p"])
    yuv = fmt in ("nv12", "yuv420p")
    w, h = rng.randint(2, 200), rng.randint(2, 120)
    src = synth_planes(orc, fmt, w, h, seed=3000 + case)
    op = rng.choice(["crop", "flip", "transpose", "rotate", "smooth", "scale", "format"])
    desc = (case, op, fmt, w, h)
    want = None

    def per_plane(fn):
        outs = []
        for i, pl in enumerate(src):
            bpp = plane_bpp(fmt, i)
            outs.append(fn(np.ascontiguousarray(pl), pl.shape[1] // bpp, pl.shape[0], bpp, i))
        return outs

    if op == "crop":
        cw, ch = rng.randint(1, w), rng.randint(1, h)
        cx, cy = rng.randint(0, w - cw), rng.randint(0, h - ch)
        opts = {"w": cw, "h": ch, "x": cx, "y": cy}
        if yuv:
            cw &= ~1; ch &= ~1; cx &= ~1; cy &= ~1
            if cw <= 0 or ch <= 0: continue
            want = [src[0][cy:cy + ch, cx:cx + cw]] + [p[cy // 2:cy // 2 + ch // 2, (cx // 2) * plane_bpp(fmt, 1):(cx // 2 + cw // 2) * plane_bpp(fmt, 1)] for p in src[1:]]
        else:
            b = plane_bpp(fmt, 0)
            want = [src[0][cy:cy + ch, cx * b:(cx + cw) * b]]
        name = "crop_hip"
    elif op == "flip":
        code = rng.choice([0, 1, -1]); opts = {"code": code}; name = "flip_hip"
        def fn(pl, pw, ph, bpp, i):
            t = pl
            if code != 0:
                o = np.zeros_like(pl); L.orc_hflip(t.ctypes.data, t.strides[0], o.ctypes.data, o.strides[0], pw, ph, bpp); t = o
            if code <= 0:
                o = np.zeros_like(pl); L.orc_vflip(t.ctypes.data, t.strides[0], o.ctypes.data, o.strides[0], pw, ph, bpp); t = o
            return t
        want = per_plane(fn)
    elif op == "transpose":
        dr = rng.randint(0, 3); opts = {"dir": dr}; name = "transpose_hip"
        def fn(pl, pw, ph, bpp, i):
            o = np.zeros((pw, ph * bpp), np.uint8)
            L.orc_transpose(pl.ctypes.data, pl.strides[0], o.ctypes.data, o.strides[0], pw, ph, bpp, dr); return o
        want = per_plane(fn)
    elif op == "rotate":
        ang = rng.choice([0, 90, 180, 270, -90, rng.uniform(-360, 360)]); interp = rng.choice(["linear", "nearest"])
        opts = {"angle": repr(ang), "interp": interp}; name = "rotate_hip"
        q = ang / 90.0
        if abs(q - round(q)) < 1e-9:
            k = int(round(q)) % 4
            def fn(pl, pw, ph, bpp, i):
                if k == 0: return pl
                if k == 2: return np.ascontiguousarray(pl.reshape(ph, pw, bpp)[::-1, ::-1].reshape(ph, pw * bpp))
                o = np.zeros((pw, ph * bpp), np.uint8)
                L.orc_transpose(pl.ctypes.data, pl.strides[0], o.ctypes.data, o.strides[0], pw, ph, bpp, 1 if k == 1 else 2); return o
        else:
            def fn(pl, pw, ph, bpp, i):
                fill = np.array([16 if i == 0 else 128, 128, 0, 0] if yuv else [0, 0, 0, 255], np.uint8)
                o = np.zeros_like(pl)
                L.orc_rotate(pl.ctypes.data, pl.strides[0], o.ctypes.data, o.strides[0], pw, ph, pw, ph, bpp,
                             float(ang) * math.pi / 180.0, 1 if interp == "linear" else 0, fill.ctypes.data); return o
        want = per_plane(fn)
    elif op == "smooth":
        opts = {"type": "gaussian"}; name = "smooth_hip"
        def fn(pl, pw, ph, bpp, i):
            o = np.zeros_like(pl); m = (C.c_int * 9)(1, 2, 1, 2, 4, 2, 1, 2, 1)
            L.orc_conv3x3(pl.ctypes.data, pl.strides[0], o.ctypes.data, o.strides[0], pw, ph, bpp, m, 1 / 16, 0.0); return o
        want = per_plane(fn)
    elif op == "scale":
        dw, dh = rng.randint(2, 200), rng.randint(2, 120)
        algo = rng.choice(["nearest", "bilinear", "bicubic", "lanczos"])
        opts = {"w": dw, "h": dh, "interp_algo": algo}; name = "scale_hip"
        fl = {"nearest": SWS["point"], "bilinear": SWS["bilinear"], "bicubic": SWS["bicubic"], "lanczos": SWS["lanczos"]}[algo]
        dfmt = fmt
        if fmt in ("rgba", "bgra"): continue                # 32-bit sources are not offered by the scaler
        if (dw, dh) == (w, h): continue
        oc = L.orc_sws_create(w, h, PIX_FMT[fmt], dw, dh, PIX_FMT[dfmt], fl, None)
        if not oc: continue
        L.orc_sws_free(oc)
        want = orc.sws(src, w, h, fmt, dw, dh, dfmt, fl)
    else:
        if fmt in ("rgba", "bgra"): continue
        dfmt = rng.choice(["rgb24", "bgr24", "rgba", "bgra"] if yuv else ["nv12", "yuv420p", "bgr24" if fmt == "rgb24" else "rgb24"])
        opts = {"pix_fmt": dfmt}; name = "format_hip"
        if yuv: want = [orc.yuv2rgb(src, w, h, fmt, dfmt)]
        elif dfmt in ("nv12", "yuv420p"): want = orc.sws(src, w, h, fmt, w, h, dfmt, SWS["bicubic"])
        else: want = [np.ascontiguousarray(src[0].reshape(h, w, 3)[:, :, ::-1].reshape(h, 3 * w))]
    got = run_filter(name, opts, src, w, h, fmt)
    if got is None:
        skipped += 1; continue
    res = got[0]
    ok = len(res) == len(want) and all(a.shape == b.shape and (a == b).all() for a, b in zip(res, want))
    if not ok:
        fails += 1
        print("MISMATCH", desc, opts, [a.shape for a in res], [b.shape for b in want])
print("cases", n, "skipped", skipped, "failures", fails)
sys.exit(1 if fails else 0)
