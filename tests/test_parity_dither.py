"""8-bit planar output of a source DEEPER than 8 bits is dithered (round 4; found by running the reference's real libswscale, tests/fuzz/fuzz_ref_core.py).

libswscale's vertical output functions take a dither row: the constant 64 for 8-bit sources, row dstY & 7 (luma) / chrDstY & 7 (chroma) of
ff_dither_8x8_128 when the source has 9 .. 16 bits (swscale.c:36-46, 263-264, 349-351, 482-485; yuv2planeX_8_c output.c:400-413 with the V plane three
columns on, vscale.c:98-101; yuv2nv12cX_c :425-458 u: i & 7, v: (i + 3) & 7).  Rounds 1-3 used 64 for every source: P010 / P016 / YUV420P10 / 16 /
YUV444P16 / RGBA64 -> NV12 / YUV420P / YUV444P came out +-1 from libswscale on a quarter of the bytes.  Known answers (constant planes: every filter
sums to one, so a sample v of a 10-bit plane leaves as (32 v + dither) >> 7 whatever the geometry) hold the kernels against the TABLE, written out
here from the reference; the oracle comparison covers real content, every kernel the dispatch can pick and both chroma layouts.
Equal size and equal layout is planarCopyWrapper instead (swscale_unscaled.c:1743-1800): its own tables (dithers[shift - 1]), no V offset."""
import ctypes as C

import numpy as np
import pytest

from harness import SWS, PIX_FMT, synth_planes, planes, ints, alloc_planes

# swscale.c:36-46 (rows 0 .. 7)
T128 = np.array([[36, 68, 60, 92, 34, 66, 58, 90], [100, 4, 124, 28, 98, 2, 122, 26], [52, 84, 44, 76, 50, 82, 42, 74], [116, 20, 108, 12, 114, 18, 106, 10],
                 [32, 64, 56, 88, 38, 70, 62, 94], [96, 0, 120, 24, 102, 6, 126, 30], [48, 80, 40, 72, 54, 86, 46, 78], [112, 16, 104, 8, 118, 22, 110, 14]], np.int64)
# swscale_unscaled.c:40-113, dithers[1] (shift 2); dithers[7] (shift 8) has T128's values
T2 = np.array([[1, 2], [3, 0]], np.int64)


def _const_src(fmt, w, h, vals):
    """planes of one value each: y, u, v given as 10-bit numbers (P010: in the high bits)"""
    cw, ch = (w + 1) // 2, (h + 1) // 2
    hi = fmt == "p010le"
    mk = lambda rows, cols, v: np.full((rows, cols), (v << 6) if hi else v, "<u2").view(np.uint8).reshape(rows, 2 * cols).copy()
    if fmt == "p010le":
        uv = np.empty((ch, 2 * cw), "<u2"); uv[:, 0::2] = vals[1] << 6; uv[:, 1::2] = vals[2] << 6
        return [mk(h, w, vals[0]), uv.view(np.uint8).reshape(ch, 4 * cw).copy()]
    return [mk(h, w, vals[0]), mk(ch, cw, vals[1]), mk(ch, cw, vals[2])]


def _want(dst_fmt, dw, dh, vals):
    cw, ch = (dw + 1) // 2, (dh + 1) // 2
    yy, xx = np.mgrid[0:dh, 0:dw]
    cy, cx = np.mgrid[0:ch, 0:cw]
    y = np.minimum((32 * vals[0] + T128[yy & 7, xx & 7]) >> 7, 255)          # av_clip_uint8
    u = np.minimum((32 * vals[1] + T128[cy & 7, cx & 7]) >> 7, 255)
    v = np.minimum((32 * vals[2] + T128[cy & 7, (cx + 3) & 7]) >> 7, 255)
    if dst_fmt == "nv12":
        uv = np.empty((ch, 2 * cw), np.int64); uv[:, 0::2] = u; uv[:, 1::2] = v
        return [y.astype(np.uint8), uv.astype(np.uint8)]
    return [y.astype(np.uint8), u.astype(np.uint8), v.astype(np.uint8)]


GEOMS = [(512, 96, 256, 48), (640, 72, 320, 36), (384, 216, 160, 90), (160, 90, 240, 136), (320, 48, 320, 48), (258, 50, 128, 24)]


@pytest.mark.parametrize("sf", ["p010le", "yuv420p10le"])
@pytest.mark.parametrize("df", ["nv12", "yuv420p"])
@pytest.mark.parametrize("geom", GEOMS)
def test_known_answer_constant_planes(dev, sf, df, geom):
    """every value class of 32 v mod 128 (0, 32, 64, 96): the output shows which table entries are >= 96, 64, 32 — on every kernel the table can pick
    (2:1 walkers of both layouts, the tiled kernel, the up-scaler's fall-back)"""
    sw, sh, dw, dh = geom
    if (sf, df) == ("yuv420p10le", "yuv420p") and (sw, sh) == (dw, dh):
        pytest.skip("equal size and layout: planarCopyWrapper (test_plane_copy_down)")
    for vals in ((513, 258, 771), (130, 515, 640), (1023, 0, 511)):
        src = _const_src(sf, sw, sh, vals)
        d = dev.upload_planes(src, 64)
        got, pads, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, SWS["bicubic"], dst_align=64)
        for p in d:
            p.free()
        for i, (g, w) in enumerate(zip(got, _want(df, dw, dh, vals))):
            bad = np.argwhere(g != w)
            assert bad.size == 0, f"{kernel} {sf} -> {df} {geom} plane {i} values {vals}: {len(bad)} wrong, first {bad[:4].tolist()}: {g[tuple(bad[0])]} != {w[tuple(bad[0])]}"


@pytest.mark.parametrize("pair", [("p010le", "nv12"), ("p010le", "yuv420p"), ("yuv420p10le", "nv12"), ("yuv420p10le", "yuv420p"), ("p016le", "nv12"),
                                  ("yuv420p16le", "yuv420p"), ("yuv444p16le", "nv12"), ("yuv444p16le", "yuv444p"), ("rgba64le", "nv12"), ("bgra64le", "yuv420p")])
@pytest.mark.parametrize("geom", GEOMS + [(200, 120, 68, 42), (161, 91, 80, 45)])
def test_real_content_against_the_oracle(dev, orc, pair, geom):
    sf, df = pair
    sw, sh, dw, dh = geom
    if (sf in ("p010le", "p016le") or df == "nv12") and ((sw | dw) & 1):
        sw, dw = sw + 1, dw + 1
    src = synth_planes(orc, sf, sw, sh, seed=77)
    if sf in ("p010le", "yuv420p10le"):
        for p in src:
            p.view("<u2")[...] = (p.view("<u2") >> 6) << (6 if sf == "p010le" else 0)
    for algo in ("bicubic", "bilinear"):
        want = orc.sws(src, sw, sh, sf, dw, dh, df, SWS[algo])
        d = dev.upload_planes(src, 64)
        got, pads, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, SWS[algo], dst_align=64)
        for p in d:
            p.free()
        for i, (g, w) in enumerate(zip(got, want)):
            assert (g == w).all(), f"{kernel} {sf} -> {df} {geom} {algo} plane {i}: {int((g != w).sum())} bytes"
            assert (pads[i] == 0xCD).all()


def test_eight_bit_sources_keep_the_constant(dev):
    """should_dither is a property of the SOURCE: an 8-bit source's planar output starts at 64 << 12 on every column"""
    sw, sh, dw, dh = 512, 96, 256, 48
    src = [np.full((sh, sw), 131, np.uint8), np.full((sh // 2, sw), 77, np.uint8)]
    d = dev.upload_planes(src, 64)
    got, _, kernel = dev.sws(d, sw, sh, "nv12", dw, dh, "nv12", SWS["bicubic"], dst_align=64)
    assert (got[0] == 131).all() and (got[1] == 77).all(), kernel


@pytest.mark.parametrize("case", [("yuv420p10le", "yuv420p", 10), ("yuv420p16le", "yuv420p", 16), ("yuv444p16le", "yuv444p", 16)])
@pytest.mark.parametrize("full", [0, 1])
def test_plane_copy_down(dev, orc, case, full):
    """equal size, the 8-bit format of the same layout: planarCopyWrapper's DITHER_COPY — (v + d) >> shift, minus its own bit 8, for chroma and for the luma
    of a limited-range source; (v - (v >> 8) + d) >> shift for the luma of a full-range one; d = dithers[shift - 1][y & 7][x & 7] on EVERY plane.
    Held against the tables written out above and against the oracle; a context whose ranges differ leaves the wrapper for the generic lines."""
    sf, df, depth = case
    lib = dev.lib
    w, h = 162, 50
    src = synth_planes(orc, sf, w, h, seed=91)
    if depth == 10:
        for p in src:
            p.view("<u2")[...] &= 0x3FF
    src[0].view("<u2")[0, :8] = [(1 << depth) - 1, (1 << depth) - 2, (1 << depth) - 3, (1 << depth) - 4, 0, 1, 2, 3]     # the t - (t >> 8) clause
    d = dev.upload_planes(src, 2)
    c = lib.gmat_sws_getContext(w, h, PIX_FMT[sf], w, h, PIX_FMT[df], SWS["bicubic"] | SWS["hwaccel"], None)
    assert c
    if full:
        assert lib.gmat_sws_setRange(c, 1, 1) == 0
    dst = dev.planes_like(df, w, h, 64)
    assert lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, h, planes([p.ptr for p in dst]), ints([p.stride for p in dst])) == h
    assert lib.gmat_sws_lastKernel(c).decode() == "plane_copy_down3_kernel"
    got = [p.download() for p in dst]
    shift = depth - 8
    for i, g in enumerate(got):
        v = src[i].view("<u2").astype(np.int64)
        yy, xx = np.mgrid[0:v.shape[0], 0:v.shape[1]]
        dith = T2[yy & 1, xx & 1] if shift == 2 else T128[yy & 7, xx & 7]
        if i == 0 and full:
            want = (v - (v >> 8) + dith) >> shift
        else:
            t = (v + dith) >> shift
            want = t - (t >> 8)
        assert (g == want.astype(np.uint8)).all(), (i, int((g != want).sum()))
    # the oracle's context entry point takes the same wrapper
    L = orc.L
    L.orc_sws_create_ex.restype = C.c_void_p
    L.orc_sws_create_ex.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    oc = L.orc_sws_create_ex(w, h, PIX_FMT[sf], w, h, PIX_FMT[df], SWS["bicubic"], None, (C.c_int * 4)(-513, -513, -513, -513), full, full)
    want = alloc_planes(df, w, h)
    assert L.orc_sws_scale(C.c_void_p(oc), planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                           planes([p.ctypes.data for p in want]), ints([p.strides[0] for p in want])) == h
    L.orc_sws_free(C.c_void_p(oc))
    for g, wv in zip(got, want):
        assert (g == wv[:, :g.shape[1]]).all()
    # ranges that differ: the generic lines carry the conversion (utils.c:1996-2000), their own dither
    assert lib.gmat_sws_setRange(c, 0, 1) == 0
    assert lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, h, planes([p.ptr for p in dst]), ints([p.stride for p in dst])) == h
    assert lib.gmat_sws_lastKernel(c).decode() != "plane_copy_down3_kernel"
    lib.gmat_sws_freeContext(c)
    for p in d + dst:
        p.free()
