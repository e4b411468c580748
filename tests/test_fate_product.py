"""The HIP path checked DIRECTLY against the reference's own golden values (no oracle in between):
the frames of the reference's vsynth1 clip go through the C ABI and the result must carry the
checksum the reference tree ships in tests/ref/fate (fixture tests/golden/fate_refs.json).

  filter-colorlevels  = scale,format=rgb24 of yuv420p with flags bicubic+accurate_rnd+bitexact
                        (tests/fate/filter-video.mak:423-424; colorlevels at defaults is the identity)
  filter-transpose    = transpose (cclock_flip) of each yuv420p plane (filter-video.mak:297-298)
  pixfmt-rgb24/bgr24  = yuv420p -> rgb24|bgr24 -> yuv444p, md5 of the raw frame (tests/fate-run.sh pixfmt_conversion)
  pixfmt-yuv420p      = yuv420p -> yuv444p
  sws-yuv-colorspace  = yuv420p (bt709, limited) -> yuv420p (bt601, full): libswscale's two-context cascade through bgr24
The oracle only supplies the input clip (oracle/orc_vsynth.c restates tests/videogen.c).
"""
import ctypes as C
import json
import os
import zlib

import numpy as np
import pytest

from harness import is_generic, SWS, DevPlane

W, H, NFRAMES = 352, 288, 50
_REFS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fate_refs.json")))
GOLD, GOLD_MD5 = _REFS["framecrc"], _REFS["pixfmt_md5"]


@pytest.fixture(scope="module")
def clip(orc):
    orc.L.orc_vsynth1.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    buf = np.zeros(NFRAMES * W * H * 3 // 2, np.uint8)
    assert orc.L.orc_vsynth1(buf.ctypes.data, W, H, NFRAMES) == NFRAMES
    return buf.reshape(NFRAMES, -1)


def adler0(b):
    return "0x%08x" % (zlib.adler32(bytes(b), 0) & 0xFFFFFFFF)


def yuv420p_planes(f):
    y = f[:W * H].reshape(H, W)
    u = f[W * H:W * H * 5 // 4].reshape(H // 2, W // 2)
    v = f[W * H * 5 // 4:].reshape(H // 2, W // 2)
    return [y, u, v]


def frame_set(dev):
    return range(NFRAMES) if dev.kind == "hip" else (0, 7, 49)


@pytest.mark.parametrize("src_fmt", ["yuv420p", "nv12"])
def test_product_reproduces_fate_filter_colorlevels(dev, clip, src_fmt):
    flags = SWS["bicubic"] | SWS["accurate_rnd"] | SWS["bitexact"]
    for i in frame_set(dev):
        y, u, v = yuv420p_planes(clip[i])
        if src_fmt == "nv12":       # same samples, interleaved chroma: the reference's result is the same frame
            src = [y, np.stack([u, v], axis=2).reshape(H // 2, W)]
        else:
            src = [y, u, v]
        d = dev.upload_planes(src)
        outs, _, kernel = dev.sws(d, W, H, src_fmt, W, H, "rgb24", flags)
        for p in d:
            p.free()
        assert outs[0].size == GOLD["filter-colorlevels"][i]["size"]
        assert adler0(outs[0]) == GOLD["filter-colorlevels"][i]["adler32"], (i, kernel)


def test_product_reproduces_fate_filter_transpose(dev, clip):
    for i in frame_set(dev):
        out = []
        for p in yuv420p_planes(clip[i]):
            h, w = p.shape
            d = dev.upload_planes([p])[0]
            o = DevPlane(dev, w, h)
            assert dev.lib.gmat_transpose(d.ptr, d.stride, o.ptr, o.stride, w, h, 1, 0, None) == 0
            out.append(o.download().ravel())
            d.free(); o.free()
        frame = np.concatenate(out)
        assert frame.size == GOLD["filter-transpose"][i]["size"]
        assert adler0(frame) == GOLD["filter-transpose"][i]["adler32"], i


def test_product_reproduces_fate_sws_yuv_range(dev, clip):
    """fate-sws-yuv-range (tests/fate/libswscale.mak:28-34): yuv420p limited -> full range at the same size, i.e.
    1-tap hScale8To15, lum/chrRangeToJpeg_c, yuv2plane1_8_c — the HIP generic path with gmat_sws_setRange."""
    from harness import PIX_FMT, planes, ints
    lib = dev.lib
    src = yuv420p_planes(clip[0])
    d = dev.upload_planes(src, 64)
    c = lib.gmat_sws_getContext(W, H, PIX_FMT["yuv420p"], W, H, PIX_FMT["yuv420p"],
                                SWS["bicubic"] | SWS["accurate_rnd"] | SWS["bitexact"], None)
    assert c
    assert lib.gmat_sws_setRange(c, 0, 1) == 0
    dst = dev.planes_like("yuv420p", W, H, 64)
    r = lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, H,
                           planes([p.ptr for p in dst]), ints([p.stride for p in dst]))
    assert r == H
    frame = np.concatenate([p.download().ravel() for p in dst])
    assert is_generic(lib.gmat_sws_lastKernel(c).decode())
    g = GOLD["sws-yuv-range"][0]
    assert frame.size == g["size"] and adler0(frame) == g["adler32"]
    # equal ranges again: back to the lossless plane copy
    assert lib.gmat_sws_setRange(c, 1, 1) == 0
    r = lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, H,
                           planes([p.ptr for p in dst]), ints([p.stride for p in dst]))
    assert r == H and all((a.download() == b).all() for a, b in zip(dst, src))
    lib.gmat_sws_freeContext(c)
    for p in d + dst:
        p.free()


def test_product_reproduces_fate_filter_scalechroma(dev, clip):
    """fate-filter-scalechroma (tests/fate/filter-video.mak:416-418): vsynth1.yuv read as 352x288 yuv444p, scaled to
    yuv420p with out_v_chr_pos=33:out_h_chr_pos=151 — shifted chroma positions, 2:1 bicubic chroma in both axes."""
    from harness import PIX_FMT, planes, ints
    lib = dev.lib
    frames444 = clip.reshape(25, -1)
    c = lib.gmat_sws_getContext(W, H, PIX_FMT["yuv444p"], W, H, PIX_FMT["yuv420p"], SWS["bicubic"] | SWS["bitexact"], None)
    assert c
    assert lib.gmat_sws_setChromaPos(c, -513, -513, 151, 33) == 0
    dst = dev.planes_like("yuv420p", W, H, 64)
    for i in (range(25) if dev.kind == "hip" else (0, 12, 24)):
        f = frames444[i]
        src = [np.ascontiguousarray(f[k * W * H:(k + 1) * W * H].reshape(H, W)) for k in range(3)]
        d = dev.upload_planes(src, 64)
        r = lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, H,
                               planes([p.ptr for p in dst]), ints([p.stride for p in dst]))
        assert r == H
        frame = np.concatenate([p.download().ravel() for p in dst])
        g = GOLD["filter-scalechroma"][i]
        assert frame.size == g["size"] and adler0(frame) == g["adler32"], i
        for p in d:
            p.free()
    lib.gmat_sws_freeContext(c)
    for p in dst:
        p.free()


@pytest.mark.parametrize("fmt", ["rgb24", "bgr24", "yuv420p"])
def test_product_reproduces_fate_pixfmt(dev, clip, fmt):
    """fate-pixfmt-<fmt>: scale=...,format=<fmt> then back to yuv444p (the comparison format of the test), all
    with flags bicubic+accurate_rnd+bitexact; the reference file holds the md5 of the one-frame result"""
    import hashlib
    flags = SWS["bicubic"] | SWS["accurate_rnd"] | SWS["bitexact"]
    d = dev.upload_planes(yuv420p_planes(clip[0]))
    if fmt == "yuv420p":
        outs, _, kernel = dev.sws(d, W, H, "yuv420p", W, H, "yuv444p", flags)
    else:
        mid, _, _ = dev.sws(d, W, H, "yuv420p", W, H, fmt, flags)
        dm = dev.upload_planes(mid)
        outs, _, kernel = dev.sws(dm, W, H, fmt, W, H, "yuv444p", flags)
        for p in dm:
            p.free()
    for p in d:
        p.free()
    raw = b"".join(bytes(np.ascontiguousarray(o)) for o in outs)
    assert hashlib.md5(raw).hexdigest() == GOLD_MD5[fmt], kernel


def test_product_reproduces_fate_sws_yuv_colorspace(dev, clip):
    """differing matrices between two YUV ends: what libswscale does internally (sws_setColorspaceDetails, utils.c:966-
    1036) spelled out with two contexts of the C ABI — yuv420p -> bgr24 with the source's BT.709 matrix, then bgr24 ->
    yuv420p with BT.601 and a full-range destination"""
    from harness import planes, ints, PIX_FMT
    lib = dev.lib
    flags = SWS["bicubic"] | SWS["accurate_rnd"] | SWS["bitexact"]
    src = dev.upload_planes(yuv420p_planes(clip[0]))
    mid = dev.planes_like("bgr24", W, H)
    dst = dev.planes_like("yuv420p", W, H)
    c0 = lib.gmat_sws_getContext(W, H, PIX_FMT["yuv420p"], W, H, PIX_FMT["bgr24"], flags, None)
    c1 = lib.gmat_sws_getContext(W, H, PIX_FMT["bgr24"], W, H, PIX_FMT["yuv420p"], flags, None)
    assert c0 and c1
    assert lib.gmat_sws_setColorspace(c0, 1, 0) == 0                       # SWS_CS_ITU709, limited-range source
    assert lib.gmat_sws_setRange(c1, 0, 1) == 0                            # full-range destination
    assert lib.gmat_sws_scale(c0, planes([p.ptr for p in src]), ints([p.stride for p in src]), 0, H,
                              planes([p.ptr for p in mid]), ints([p.stride for p in mid])) == H
    assert lib.gmat_sws_scale(c1, planes([p.ptr for p in mid]), ints([p.stride for p in mid]), 0, H,
                              planes([p.ptr for p in dst]), ints([p.stride for p in dst])) == H
    out = np.concatenate([p.download().ravel() for p in dst])
    assert out.size == GOLD["sws-yuv-colorspace"][0]["size"]
    assert adler0(out) == GOLD["sws-yuv-colorspace"][0]["adler32"], (lib.gmat_sws_lastKernel(c0), lib.gmat_sws_lastKernel(c1))
    lib.gmat_sws_freeContext(c0); lib.gmat_sws_freeContext(c1)
    for p in src + mid + dst:
        p.free()
