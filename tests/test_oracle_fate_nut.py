"""More of the reference's OWN golden values for the oracle: the FATE filter tests whose reference is the md5 of a
NUT-muxed raw stream (tests/fate-run.sh:458-497 video_filter / pixfmts; refs tests/ref/fate/filter-* and
filter-pixfmts-*; fixture tests/golden/fate_refs.json `nut_md5`, extracted by tools/gen_golden_fate.py).

tests/nutmux.py restates the muxer (validated here by filter-null / pixfmts-null on yuv420p, whose frames are the clip
itself).  With the container out of the way each md5 pins the oracle stages between the clip and the frame bytes:

  filter-vflip, -crop, -crop_vflip, -vflip_crop, -vflip_vflip      orc_vflip, orc_crop on 4:2:0 planes, 5 frames
  filter-scale200 / -scale500 / -crop_scale                        the generic scaler 4:2:0 -> 4:2:0, down and up
  filter-pixfmts-<f> rows yuv420p nv12 rgb24 bgr24 rgba bgra yuv444p p010le, f in null copy hflip vflip crop transpose
  rotate scale: `scale,format=<fmt>,<f>`                           the yuv420p -> <fmt> conversion (generic path with
      accurate_rnd: yuv2rgb_X_c / 32-bit writers; planarToNv12Wrapper; planar8ToP01xleWrapper; chroma up-scaling) and
      the filter on every plane layout, incl. the interleaved NV12 / P010 chroma planes
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import nutmux
from test_oracle_fate import fate, W, H, NFRAMES, BICUBIC, ACCURATE_RND, BITEXACT  # noqa: F401

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fate_refs.json")))["nut_md5"]
FLAGS = BICUBIC | ACCURATE_RND | BITEXACT           # -sws_flags +accurate_rnd+bitexact on libswscale's default bicubic
FMT = {"yuv420p": 0, "rgb24": 2, "bgr24": 3, "yuv444p": 5, "nv12": 23, "rgba": 26, "bgra": 28, "p010le": 159,
       "p016le": 170, "yuv444p16le": 49, "rgba64le": 105, "bgra64le": 107, "yuv420p10le": 62, "yuv420p16le": 45}
FOURCC = dict(nutmux.FOURCC)
# rawvideo has no fourcc for P010LE (libavcodec/raw.c), so the muxer takes av_codec_get_tag2's first RAWVIDEO entry of
# ff_nut_video_tags (libavformat/nut.c:49) — confirmed by the reference md5s themselves
FOURCC["p010le"] = b"RGB\x0f"
FOURCC["p016le"] = b"RGB\x0f"                        # no tag either
# libavcodec/raw.c:89-90,160: MKTAG('R','B','A',64), MKTAG('B','R','A',64), MKTAG('Y','3',0,16)
FOURCC.update({"rgba64le": b"RBA\x40", "bgra64le": b"BRA\x40", "yuv444p16le": b"Y3\x00\x10",
               "yuv420p10le": b"Y3\x0b\x0a", "yuv420p16le": b"Y3\x0b\x10"})     # raw.c:138,156: MKTAG('Y','3',11,10 | 16)


def shapes(fmt, w, h):
    """[(rows, bytes per row, bytes per pixel of that plane, chroma shift)]"""
    cw, ch = (w + 1) // 2, (h + 1) // 2
    return {"yuv420p": [(h, w, 1, 0), (ch, cw, 1, 1), (ch, cw, 1, 1)], "yuv444p": [(h, w, 1, 0)] * 3,
            "nv12": [(h, w, 1, 0), (ch, 2 * cw, 2, 1)], "p010le": [(h, 2 * w, 2, 0), (ch, 4 * cw, 4, 1)],
            "rgb24": [(h, 3 * w, 3, 0)], "bgr24": [(h, 3 * w, 3, 0)], "rgba": [(h, 4 * w, 4, 0)], "bgra": [(h, 4 * w, 4, 0)],
            "p016le": [(h, 2 * w, 2, 0), (ch, 4 * cw, 4, 1)], "yuv444p16le": [(h, 2 * w, 2, 0)] * 3,
            "rgba64le": [(h, 8 * w, 8, 0)], "bgra64le": [(h, 8 * w, 8, 0)],
            "yuv420p10le": [(h, 2 * w, 2, 0), (ch, 2 * cw, 2, 1), (ch, 2 * cw, 2, 1)],
            "yuv420p16le": [(h, 2 * w, 2, 0), (ch, 2 * cw, 2, 1), (ch, 2 * cw, 2, 1)]}[fmt]


def planes_of(fmt, w, h, data=None):
    out, o = [], 0
    for rows, rb, _, _ in shapes(fmt, w, h):
        out.append(np.zeros((rows, rb), np.uint8) if data is None else np.ascontiguousarray(data[o:o + rows * rb].reshape(rows, rb)))
        o += rows * rb
    return out


def scale(L, src, sf, sw, sh, df, dw, dh):
    """what vf_scale's context computes: MPEG-2 vertical chroma position 128 on every YUV420P end (vf_scale.c:563-573)"""
    # vf_scale.c:563-573 sets the MPEG-2 vertical position for AV_PIX_FMT_YUV420P only, not for its high-depth siblings
    pos = (-513, 128 if sf == "yuv420p" else -513, -513, 128 if df == "yuv420p" else -513)
    if sf == df and (sw, sh) == (dw, dh):
        return [p.copy() for p in src]                   # vf_scale passes equal frames through
    if sf == "yuv420p" and (sw, sh) == (dw, dh) and df == "nv12":       # planarToNv12Wrapper (swscale_unscaled.c:170-185)
        uv = np.zeros((src[1].shape[0], 2 * src[1].shape[1]), np.uint8)
        uv[:, 0::2], uv[:, 1::2] = src[1], src[2]
        return [src[0].copy(), uv]
    dst = planes_of(df, dw, dh)
    P4, I4 = C.c_void_p * 4, C.c_int * 4
    if sf == "yuv420p" and (sw, sh) == (dw, dh) and df in ("p010le", "p016le"):     # planar8ToP01xleWrapper (swscale_unscaled.c:286-324)
        L.orc_yuv420_to_p01x.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int]
        L.orc_yuv420_to_p01x(P4(*[p.ctypes.data for p in src]), I4(*[p.shape[1] for p in src]), P4(*[p.ctypes.data for p in dst]),
                             I4(*[p.shape[1] for p in dst]), sw, sh, 0)
        return dst
    ctx = L.orc_sws_create_ex(sw, sh, FMT[sf], dw, dh, FMT[df], FLAGS, None, (C.c_int * 4)(*pos), 0, 0)
    assert ctx, (sf, df)
    r = L.orc_sws_scale(ctx, P4(*[p.ctypes.data for p in src]), I4(*[p.shape[1] for p in src]),
                        P4(*[p.ctypes.data for p in dst]), I4(*[p.shape[1] for p in dst]))
    L.orc_sws_free(ctx)
    assert r == dh
    return dst


def per_plane(L, fn, fmt, planes, w, h, *args, swap=False):
    out = []
    for p, (rows, rb, bpp, sub) in zip(planes, shapes(fmt, w, h)):
        pw = rb // bpp
        d = np.zeros((pw, rows * bpp) if swap else (rows, rb), np.uint8)
        fn(p.ctypes.data, p.shape[1], d.ctypes.data, d.shape[1], pw, rows, bpp, *args)
        out.append(d)
    return out


def setup_vf(L):
    for n in ("orc_hflip", "orc_vflip"):
        getattr(L, n).argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_transpose.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_crop.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_int] * 5
    L.orc_rotate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]


def crop(L, fmt, planes, w, h, cw, ch, x, y):
    """vf_crop on a planar / packed frame: plane offsets shifted by the chroma subsampling (vf_crop.c:262-304)"""
    out = []
    for p, (rows, rb, bpp, sub) in zip(planes, shapes(fmt, w, h)):
        pw, ph, px, py = (cw + sub) >> sub if sub else cw, (ch + sub) >> sub if sub else ch, x >> sub, y >> sub
        d = np.zeros((ph, pw * bpp), np.uint8)
        L.orc_crop(p.ctypes.data, p.shape[1], d.ctypes.data, d.shape[1], px, py, pw, ph, bpp)
        out.append(d)
    return out


def nut(frames, fmt, w, h):
    return nutmux.md5([b"".join(p.tobytes() for p in fr) for fr in frames], w, h, FOURCC[fmt])


def clip_frames(clip, n):
    return [planes_of("yuv420p", W, H, f) for f in clip.reshape(NFRAMES, -1)[:n]]


# ---- the muxer restatement itself --------------------------------------------------------------------------------
def test_nut_restatement_reproduces_the_untouched_clip(fate):
    L, clip = fate
    assert nut(clip_frames(clip, 5), "yuv420p", W, H) == GOLD["video_filter"]["null"]
    assert nut(clip_frames(clip, 1), "yuv420p", W, H) == GOLD["pixfmts"]["null"]["yuv420p"] == GOLD["pixfmts"]["copy"]["yuv420p"]


# ---- video_filter: 5 frames of the yuv420p clip ------------------------------------------------------------------
def test_fate_filter_vflip_crop_family(fate):
    L, clip = fate
    setup_vf(L)
    fr = clip_frames(clip, 5)
    vf = lambda f, w, h: per_plane(L, L.orc_vflip, "yuv420p", f, w, h)
    cr = lambda f: crop(L, "yuv420p", f, W, H, W - 100, H - 100, 100, 100)        # crop=iw-100:ih-100:100:100
    assert nut([vf(f, W, H) for f in fr], "yuv420p", W, H) == GOLD["video_filter"]["vflip"]
    assert nut([vf(vf(f, W, H), W, H) for f in fr], "yuv420p", W, H) == GOLD["video_filter"]["vflip_vflip"]
    assert nut([cr(f) for f in fr], "yuv420p", W - 100, H - 100) == GOLD["video_filter"]["crop"]
    assert nut([vf(cr(f), W - 100, H - 100) for f in fr], "yuv420p", W - 100, H - 100) == GOLD["video_filter"]["crop_vflip"]
    assert nut([cr(vf(f, W, H)) for f in fr], "yuv420p", W - 100, H - 100) == GOLD["video_filter"]["vflip_crop"]


@pytest.mark.parametrize("name,size", [("scale200", (200, 200)), ("scale500", (500, 500))])
def test_fate_filter_scale(fate, name, size):
    L, clip = fate
    fr = clip_frames(clip, 5)
    assert nut([scale(L, f, "yuv420p", W, H, "yuv420p", *size) for f in fr], "yuv420p", *size) == GOLD["video_filter"][name]


def test_fate_filter_crop_scale(fate):
    """crop=iw-100:ih-100:100:100,scale=w=400:h=-1 -> 400 x round(400 * 188 / 252) = 400 x 298"""
    L, clip = fate
    setup_vf(L)
    fr = clip_frames(clip, 5)
    cw, ch = W - 100, H - 100
    oh = (400 * ch + cw // 2) // cw
    out = [scale(L, crop(L, "yuv420p", f, W, H, cw, ch, 100, 100), "yuv420p", cw, ch, "yuv420p", 400, oh) for f in fr]
    assert nut(out, "yuv420p", 400, oh) == GOLD["video_filter"]["crop_scale"]


# ---- pixfmts: scale,format=<fmt>,<filter> on the first frame ------------------------------------------------------
PIX = ["yuv420p", "nv12", "rgb24", "bgr24", "rgba", "bgra", "yuv444p", "p010le",
       "p016le", "yuv444p16le", "rgba64le", "bgra64le",    # these four: libswscale's 19-bit lines (yuv2planeX_16_c, yuv2rgba64_X_c)
       "yuv420p10le", "yuv420p16le"]                       # planar high-depth 4:2:0: yuv2planeX_10_c / yuv2planeX_16_c per plane
NO_SRC = ("rgba64le", "bgra64le")                          # vf_rotate has no 64-bit packed formats: no rotate row


def converted(fate, fmt):
    L, clip = fate
    setup_vf(L)
    return L, scale(L, clip_frames(clip, 1)[0], "yuv420p", W, H, fmt, W, H)


@pytest.mark.parametrize("fmt", PIX)
def test_fate_pixfmts_null_copy(fate, fmt):
    """the conversion alone: yuv420p -> fmt as vf_scale's auto-inserted context does it"""
    L, f = converted(fate, fmt)
    assert nut([f], fmt, W, H) == GOLD["pixfmts"]["null"][fmt] == GOLD["pixfmts"]["copy"][fmt]


@pytest.mark.parametrize("fmt", PIX)
def test_fate_pixfmts_hflip_vflip(fate, fmt):
    L, f = converted(fate, fmt)
    assert nut([per_plane(L, L.orc_hflip, fmt, f, W, H)], fmt, W, H) == GOLD["pixfmts"]["hflip"][fmt]
    assert nut([per_plane(L, L.orc_vflip, fmt, f, W, H)], fmt, W, H) == GOLD["pixfmts"]["vflip"][fmt]


@pytest.mark.parametrize("fmt", PIX)
def test_fate_pixfmts_crop(fate, fmt):
    L, f = converted(fate, fmt)                                           # crop=100:100:100:100
    assert nut([crop(L, fmt, f, W, H, 100, 100, 100, 100)], fmt, 100, 100) == GOLD["pixfmts"]["crop"][fmt]


@pytest.mark.parametrize("fmt", PIX)
def test_fate_pixfmts_transpose(fate, fmt):
    L, f = converted(fate, fmt)                                           # default dir = cclock_flip
    assert nut([per_plane(L, L.orc_transpose, fmt, f, W, H, 0, swap=True)], fmt, H, W) == GOLD["pixfmts"]["transpose"][fmt]


@pytest.mark.parametrize("fmt", [f for f in PIX if f in GOLD["pixfmts"]["rotate"] and f not in NO_SRC])
def test_fate_pixfmts_rotate(fate, fmt):
    """rotate=2*PI*n/50 on frame n = 0: angle 0 through vf_rotate's fixed-point walk with the default bilinear taps"""
    L, f = converted(fate, fmt)
    out = []
    for p, (rows, rb, bpp, sub) in zip(f, shapes(fmt, W, H)):
        d = np.zeros_like(p)
        L.orc_rotate(p.ctypes.data, p.shape[1], d.ctypes.data, d.shape[1], rb // bpp, rows, rb // bpp, rows, bpp, 0.0, 1, None)
        out.append(d)
    assert nut([out], fmt, W, H) == GOLD["pixfmts"]["rotate"][fmt]


@pytest.mark.parametrize("fmt", PIX)
def test_fate_pixfmts_scale(fate, fmt):
    """scale=200:100 in the converted format: the generic scaler fmt -> fmt (packed RGB through its YUV lines; the formats with an
    alpha channel — rgba, bgra, rgba64le, bgra64le — also scale their alpha plane: rgbaToA_c / rgba64leToA_c, needAlpha)"""
    L, f = converted(fate, fmt)
    assert nut([scale(L, f, fmt, W, H, fmt, 200, 100)], fmt, 200, 100) == GOLD["pixfmts"]["scale"][fmt]
