"""The HIP path against the reference's NUT-md5 golden values, with NO oracle in between: frames of the reference's
vsynth1 clip go through the C ABI (libgpuscale contexts and the direct filter launchers), are muxed by the test-side
restatement of the NUT writer (tests/nutmux.py) and must hash to the md5 the reference tree ships in
tests/ref/fate/filter-* and filter-pixfmts-* (fixture tests/golden/fate_refs.json `nut_md5`).

  filter-vflip / -crop / -crop_vflip / -vflip_crop / -scale200 / -scale500 / -crop_scale   5 frames of yuv420p
  filter-pixfmts-{null,hflip,vflip,crop,transpose,rotate,scale} x {nv12, rgb24, bgr24, rgba, bgra, yuv444p, p010le, yuv420p}
      = `scale,format=<fmt>,<filter>`: the yuv420p -> <fmt> context (bicubic + accurate_rnd + bitexact, MPEG-2 vertical
        chroma position on the YUV420P end, vf_scale.c:563-573) followed by the filter on every plane
The oracle only supplies the input clip."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import nutmux
from harness import PIX_FMT, SWS, DevPlane, planes, ints, plane_shapes
from test_fate_product import clip, yuv420p_planes, W, H  # noqa: F401

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fate_refs.json")))["nut_md5"]
FLAGS = SWS["bicubic"] | SWS["accurate_rnd"] | SWS["bitexact"]
FOURCC = dict(nutmux.FOURCC, p010le=b"RGB\x0f", p016le=b"RGB\x0f", rgba64le=b"RBA\x40", bgra64le=b"BRA\x40",
              yuv444p16le=b"Y3\x00\x10", yuv420p10le=b"Y3\x0b\x0a", yuv420p16le=b"Y3\x0b\x10")   # see tests/test_oracle_fate_nut.py
# (bytes per pixel of the plane, chroma shift) per plane
LAYOUT = {"yuv420p": [(1, 0), (1, 1), (1, 1)], "yuv444p": [(1, 0)] * 3, "nv12": [(1, 0), (2, 1)], "p010le": [(2, 0), (4, 1)],
          "rgb24": [(3, 0)], "bgr24": [(3, 0)], "rgba": [(4, 0)], "bgra": [(4, 0)],
          "p016le": [(2, 0), (4, 1)], "yuv444p16le": [(2, 0)] * 3, "rgba64le": [(8, 0)], "bgra64le": [(8, 0)],
          "yuv420p10le": [(2, 0), (2, 1), (2, 1)], "yuv420p16le": [(2, 0), (2, 1), (2, 1)]}


def sws(dev, src, sf, sw, sh, df, dw, dh):
    """one libgpuscale context as vf_scale configures it; src / result: lists of host planes"""
    if sf == df and (sw, sh) == (dw, dh):
        return [p.copy() for p in src]
    lib = dev.lib
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], FLAGS, None)
    assert c, (sf, df)
    if "yuv420p" in (sf, df):
        r = lib.gmat_sws_setChromaPos(c, -513, 128 if sf == "yuv420p" else -513, -513, 128 if df == "yuv420p" else -513)
        assert r == 0 or (sw, sh) == (dw, dh)          # the unscaled special converters have no chroma filter to position
    d = dev.upload_planes(src, 64)
    o = dev.planes_like(df, dw, dh, 64)
    assert lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, sh,
                              planes([p.ptr for p in o]), ints([p.stride for p in o])) == dh
    out = [p.download() for p in o]
    lib.gmat_sws_freeContext(c)
    for p in d + o:
        p.free()
    return out


def filt(dev, which, fmt, src, w, h, *args):
    """hflip / vflip / transpose / crop / rotate on every plane through the direct launchers"""
    lib = dev.lib
    out = []
    for p, (bpp, sub) in zip(src, LAYOUT[fmt]):
        rows, pw = p.shape[0], p.shape[1] // bpp
        d = dev.upload_planes([p], 16)[0]
        if which == "transpose":
            o = DevPlane(dev, pw, rows * bpp, (rows * bpp + 15) // 16 * 16)
            assert lib.gmat_transpose(d.ptr, d.stride, o.ptr, o.stride, pw, rows, bpp, 0, None) == 0     # cclock_flip
        elif which == "crop":
            cw, ch, x, y = args
            cpw, cph = ((cw + 1) >> 1, (ch + 1) >> 1) if sub else (cw, ch)
            o = DevPlane(dev, cph, cpw * bpp, (cpw * bpp + 15) // 16 * 16)
            assert lib.gmat_crop(d.ptr, d.stride, o.ptr, o.stride, x >> sub, y >> sub, cpw, cph, bpp, None) == 0
        elif which == "rotate":
            o = DevPlane(dev, rows, pw * bpp, d.stride)
            assert lib.gmat_rotate(d.ptr, d.stride, o.ptr, o.stride, pw, rows, pw, rows, bpp, 0.0, 1, None, None) == 0
        else:
            o = DevPlane(dev, rows, pw * bpp, d.stride)
            # a vertical flip reverses rows whatever the pixel is (vf_vflip.c:108-127 only negates the line sizes): a row of
            # 8-byte pixels goes in as twice as many 4-byte ones
            fw, fb = (2 * pw, 4) if (bpp == 8 and which == "vflip") else (pw, bpp)
            assert lib.gmat_flip(d.ptr, d.stride, o.ptr, o.stride, fw, rows, fb, 1 if which == "hflip" else 0, None) == 0
        out.append(o.download())
        d.free(); o.free()
    return out


def nut(frames, fmt, w, h):
    return nutmux.md5([b"".join(np.ascontiguousarray(p).tobytes() for p in fr) for fr in frames], w, h, FOURCC[fmt])


def frames5(clip):
    return [[np.ascontiguousarray(p) for p in yuv420p_planes(clip[i])] for i in range(5)]


def test_product_fate_vflip_crop_family(dev, clip):
    fr = frames5(clip)
    cw, ch = W - 100, H - 100
    vf = lambda f, w, h: filt(dev, "vflip", "yuv420p", f, w, h)
    cr = lambda f: filt(dev, "crop", "yuv420p", f, W, H, cw, ch, 100, 100)
    assert nut([vf(f, W, H) for f in fr], "yuv420p", W, H) == GOLD["video_filter"]["vflip"]
    assert nut([cr(f) for f in fr], "yuv420p", cw, ch) == GOLD["video_filter"]["crop"]
    assert nut([vf(cr(f), cw, ch) for f in fr], "yuv420p", cw, ch) == GOLD["video_filter"]["crop_vflip"]
    assert nut([cr(vf(f, W, H)) for f in fr], "yuv420p", cw, ch) == GOLD["video_filter"]["vflip_crop"]


@pytest.mark.parametrize("name,size", [("scale200", (200, 200)), ("scale500", (500, 500))])
def test_product_fate_scale(dev, clip, name, size):
    out = [sws(dev, f, "yuv420p", W, H, "yuv420p", *size) for f in frames5(clip)]
    assert nut(out, "yuv420p", *size) == GOLD["video_filter"][name]


def test_product_fate_crop_scale(dev, clip):
    cw, ch = W - 100, H - 100
    oh = (400 * ch + cw // 2) // cw
    out = [sws(dev, filt(dev, "crop", "yuv420p", f, W, H, cw, ch, 100, 100), "yuv420p", cw, ch, "yuv420p", 400, oh) for f in frames5(clip)]
    assert nut(out, "yuv420p", 400, oh) == GOLD["video_filter"]["crop_scale"]


PIX = ["yuv420p", "nv12", "rgb24", "bgr24", "rgba", "bgra", "yuv444p", "p010le",
       "p016le", "yuv444p16le", "rgba64le", "bgra64le",      # these four: destinations of the 19-bit path (k_scale16.hip)
       "yuv420p10le", "yuv420p16le"]                         # planar high-depth 4:2:0, sources and destinations
WIDE = ("rgba64le", "bgra64le")                              # 8-byte pixels: the pixel-permuting launchers take 1..4 bytes


@pytest.fixture(scope="module")
def converted(dev, clip):
    """frame 0 of the clip in every pixel format, converted by the library"""
    f0 = frames5(clip)[0]
    return {fmt: sws(dev, f0, "yuv420p", W, H, fmt, W, H) for fmt in PIX}


@pytest.mark.parametrize("fmt", PIX)
def test_product_fate_pixfmts_conversion(dev, converted, fmt):
    assert nut([converted[fmt]], fmt, W, H) == GOLD["pixfmts"]["null"][fmt]


@pytest.mark.parametrize("fmt", PIX)
@pytest.mark.parametrize("which", ["hflip", "vflip", "transpose", "crop", "rotate"])
def test_product_fate_pixfmts_filters(dev, converted, which, fmt):
    if which == "rotate" and fmt not in GOLD["pixfmts"]["rotate"]:
        pytest.skip("vf_rotate does not take this format")
    if fmt in WIDE and which in ("hflip", "transpose", "rotate"):
        pytest.skip("8-byte pixels: hflip / transpose / rotate launchers take 1..4-byte pixels (vflip and crop move rows and byte runs)")
    f = converted[fmt]
    if which == "crop":
        assert nut([filt(dev, "crop", fmt, f, W, H, 100, 100, 100, 100)], fmt, 100, 100) == GOLD["pixfmts"]["crop"][fmt]
    elif which == "transpose":
        assert nut([filt(dev, "transpose", fmt, f, W, H)], fmt, H, W) == GOLD["pixfmts"]["transpose"][fmt]
    else:
        assert nut([filt(dev, which, fmt, f, W, H)], fmt, W, H) == GOLD["pixfmts"][which][fmt]


@pytest.mark.parametrize("fmt", PIX)
def test_product_fate_pixfmts_scale(dev, converted, fmt):
    assert nut([sws(dev, converted[fmt], fmt, W, H, fmt, 200, 100)], fmt, 200, 100) == GOLD["pixfmts"]["scale"][fmt]
