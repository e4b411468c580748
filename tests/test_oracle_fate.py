"""Pins the oracle against the reference's OWN golden values (FATE), not against itself.

The fixtures in tests/golden/fate_refs.json are the checksums the reference tree ships under
ffmpeg-gpu/tests/ref (extracted by tools/gen_golden_fate.py).  Each of them is computed by the
reference over its synthetic clip "vsynth1" (tests/videogen.c), which oracle/orc_vsynth.c restates;
reproducing a checksum therefore proves that every oracle stage between the clip and the
checksummed bytes is bit-identical with the reference:

  filter-transpose    vsynth1 generator + orc_transpose                       (50 frames)
  sws-yuv-range       1-tap hScale8To15, lum/chrRangeToJpeg, yuv2plane1_8     (1 frame)
  filter-scalechroma  initFilter with shifted chroma positions, multi-tap hScale8To15,
                      yuv2planeX_8 vertical                                   (25 frames)
  filter-colorlevels  yuv420p -> rgb24 through the generic scaler: 2x vertical chroma bicubic,
                      yuv2rgb_X_c + the yuv2rgb.c tables                      (50 frames)
  pixfmt-rgb24/bgr24  the same, followed by rgb24ToY/ToUV + hScale16To15 back to yuv444p (md5)
  pixfmt-yuv420p      yuv420p -> yuv444p chroma up-scaling (md5)
framecrc = Adler-32 started from 0 (libavformat/framecrcenc.c:52) over the tightly packed frame.
"""
import ctypes as C
import hashlib
import json
import os
import zlib

import numpy as np
import pytest

W, H, NFRAMES = 352, 288, 50
YUV420P, RGB24, BGR24, YUV444P = 0, 2, 3, 5
BICUBIC, ACCURATE_RND, BITEXACT = 4, 0x40000, 0x80000

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fate_refs.json")))


@pytest.fixture(scope="module")
def fate(orc):
    L = orc.L
    L.orc_vsynth1.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.orc_sws_create_ex.restype = C.c_void_p
    L.orc_sws_create_ex.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    clip = np.zeros(NFRAMES * W * H * 3 // 2, np.uint8)
    assert L.orc_vsynth1(clip.ctypes.data, W, H, NFRAMES) == NFRAMES
    return L, clip


def adler0(b):
    return "0x%08x" % (zlib.adler32(bytes(b), 0) & 0xFFFFFFFF)


def split(fmt, w, h, data=None):
    shapes = {RGB24: [(h, 3 * w)], BGR24: [(h, 3 * w)], YUV420P: [(h, w), (h // 2, w // 2), (h // 2, w // 2)],
              YUV444P: [(h, w)] * 3}[fmt]
    out, o = [], 0
    for r, c in shapes:
        out.append(np.zeros((r, c), np.uint8) if data is None else np.ascontiguousarray(data[o:o + r * c].reshape(r, c)))
        o += r * c
    return out


def sws(L, data, sf, df, flags, pos=(-513, -513, -513, -513), src_range=0, dst_range=0):
    s, d = split(sf, W, H, data), split(df, W, H)
    ctx = L.orc_sws_create_ex(W, H, sf, W, H, df, flags, None, (C.c_int * 4)(*pos), src_range, dst_range)
    assert ctx
    P4, I4 = C.c_void_p * 4, C.c_int * 4
    r = L.orc_sws_scale(ctx, P4(*[p.ctypes.data for p in s]), I4(*[p.shape[1] for p in s]),
                        P4(*[p.ctypes.data for p in d]), I4(*[p.shape[1] for p in d]))
    L.orc_sws_free(ctx)
    assert r == H
    return np.concatenate([p.ravel() for p in d])


def check_crcs(name, frames):
    ref = GOLD["framecrc"][name]
    assert len(frames) == len(ref)
    for i, (fr, g) in enumerate(zip(frames, ref)):
        assert fr.size == g["size"], (name, i)
        assert adler0(fr) == g["adler32"], (name, i)


# vf_scale gives 4:2:0 inputs/outputs an explicit vertical chroma position of 128 (vf_scale.c:567-573)
POS_420_IN = (-513, 128, -513, -513)


def test_fate_filter_transpose(fate):
    L, clip = fate
    out = []
    for f in clip.reshape(NFRAMES, -1):
        planes = []
        for p in split(YUV420P, W, H, f):
            h, w = p.shape
            d = np.zeros((w, h), np.uint8)
            L.orc_transpose(p.ctypes.data, w, d.ctypes.data, h, w, h, 1, 0)      # default dir = cclock_flip
            planes.append(d.ravel())
        out.append(np.concatenate(planes))
    check_crcs("filter-transpose", out)


def test_fate_sws_yuv_range(fate):
    L, clip = fate
    f0 = clip.reshape(NFRAMES, -1)[0]
    out = sws(L, f0, YUV420P, YUV420P, BICUBIC | ACCURATE_RND | BITEXACT, (-513, 128, -513, 128), 0, 1)
    check_crcs("sws-yuv-range", [out])


def test_fate_sws_yuv_colorspace(fate):
    """scale=in_color_matrix=bt709:in_range=limited:out_color_matrix=bt601:out_range=full on yuv420p at equal size
    (libswscale.mak:20-26).  Differing matrices between two YUV ends make libswscale cascade two contexts through an
    8-bit BGR24 frame (sws_setColorspaceDetails, utils.c:966-1036): yuv420p -> bgr24 with the SOURCE matrix (BT.709
    tables; generic path because of accurate_rnd), then bgr24 -> yuv420p with the DESTINATION matrix (BT.601 literals)
    and the limited -> full conversion of its 15-bit lines (the RGB side's range is forced to 0).  The cascaded
    contexts are fresh ones: default chroma positions."""
    L, clip = fate
    L.orc_sws_set_colorspace.argtypes = [C.c_void_p, C.c_int]
    f0 = clip.reshape(NFRAMES, -1)[0]
    flags = BICUBIC | ACCURATE_RND | BITEXACT
    P4, I4 = C.c_void_p * 4, C.c_int * 4
    s, mid = split(YUV420P, W, H, f0), split(BGR24, W, H)
    c0 = L.orc_sws_create_ex(W, H, YUV420P, W, H, BGR24, flags, None, (C.c_int * 4)(-513, -513, -513, -513), 0, 0)
    assert c0 and L.orc_sws_set_colorspace(c0, 1) == 0                       # SWS_CS_ITU709
    assert L.orc_sws_scale(c0, P4(*[p.ctypes.data for p in s]), I4(*[p.shape[1] for p in s]),
                           P4(*[p.ctypes.data for p in mid]), I4(*[p.shape[1] for p in mid])) == H
    L.orc_sws_free(c0)
    out = sws(L, mid[0].ravel(), BGR24, YUV420P, flags, src_range=0, dst_range=1)
    check_crcs("sws-yuv-colorspace", [out])


def test_fate_filter_scalechroma(fate):
    L, clip = fate
    # the recipe reads vsynth1.yuv as 352x288 yuv444p: 25 frames of 304128 bytes
    out = [sws(L, f, YUV444P, YUV420P, BICUBIC | BITEXACT, (-513, -513, 151, 33)) for f in clip.reshape(25, -1)]
    check_crcs("filter-scalechroma", out)


def test_fate_filter_colorlevels(fate):
    L, clip = fate
    out = [sws(L, f, YUV420P, RGB24, BICUBIC | ACCURATE_RND | BITEXACT, POS_420_IN) for f in clip.reshape(NFRAMES, -1)]
    check_crcs("filter-colorlevels", out)


@pytest.mark.parametrize("fmt,name", [(RGB24, "rgb24"), (BGR24, "bgr24")])
def test_fate_pixfmt_rgb(fate, fmt, name):
    L, clip = fate
    f0 = clip.reshape(NFRAMES, -1)[0]
    mid = sws(L, f0, YUV420P, fmt, BICUBIC | ACCURATE_RND | BITEXACT, POS_420_IN)
    back = sws(L, mid, fmt, YUV444P, BICUBIC | ACCURATE_RND | BITEXACT)
    assert hashlib.md5(bytes(back)).hexdigest() == GOLD["pixfmt_md5"][name]


def test_fate_pixfmt_yuv420p(fate):
    L, clip = fate
    f0 = clip.reshape(NFRAMES, -1)[0]
    out = sws(L, f0, YUV420P, YUV444P, BICUBIC | ACCURATE_RND | BITEXACT, POS_420_IN)
    assert hashlib.md5(bytes(out)).hexdigest() == GOLD["pixfmt_md5"]["yuv420p"]
