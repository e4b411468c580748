"""scale19_kernel on the 15-bit lines (k_scale19.hip, round 6, second half): the tile kernel — any plane layout and depth in — writing 8-bit and 10-bit YUV frames
(yuv2planeX_8_c / yuv2nv12cX_c, yuv2planeX_10_c / yuv2p010lX_c / cX_c: output.c:400-519) for the format pairs no walker serves: semi-planar <-> planar with a deep
end, 4:4:4 <-> 4:2:0, 16-bit 4:4:4 sources into 8 / 10 bits.  The format sweep (profiles/r06_sweep_*.txt) had them on the lines form's two launches or the tiled
kernel of round 1, 0.11-0.15 of the roofline.  Bit-exact against the oracle; tests/test_parity_lines.py and test_parity_planes2p.py hold the kernels behind it
(GMAT_T15=0)."""
import ctypes as C

import numpy as np
import pytest

from harness import PIX_FMT, SWS, synth_planes, planes, ints, alloc_planes

SRC = ["nv12", "yuv420p", "yuv444p", "p010le", "p016le", "yuv420p10le", "yuv420p16le", "yuv444p16le"]
DST = ["nv12", "yuv420p", "yuv444p", "p010le", "yuv420p10le"]
SEMI = {"nv12", "p010le", "p016le"}
F444 = {"yuv444p", "yuv444p16le"}


def mixed(sf, df):
    """the rule of gsws.cpp t15Mixed: the chroma layouts differ, or 4:4:4 at either end"""
    return ((sf in SEMI) != (df in SEMI)) or sf in F444 or df in F444


def _synth(orc, fmt, w, h, seed):
    src = synth_planes(orc, fmt, w, h, seed=seed)
    if fmt == "yuv420p10le":
        for p in src:
            p.view("<u2")[...] &= 0x3FF
    if fmt == "p010le":
        for p in src:
            p.view("<u2")[...] &= 0xFFC0
    return src


def _check(dev, orc, sf, df, geom, flags="bicubic", align=64, extra=0, seed=23):
    sw, sh, dw, dh = geom
    src = _synth(orc, sf, sw, sh, seed)
    want = orc.sws(src, sw, sh, sf, dw, dh, df, SWS[flags])
    d = dev.upload_planes(src, align, extra)
    got, pads, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, SWS[flags], dst_align=align, dst_extra=extra)
    for p in d:
        p.free()
    for i, (g, wv) in enumerate(zip(got, want)):
        bad = np.argwhere(g != wv)
        assert bad.size == 0, f"{sf}->{df} {geom} {flags} align {align}: plane {i}: {len(bad)} bytes differ, first at {bad[:4].tolist()} ({kernel})"
        assert (pads[i] == 0xCD).all(), (sf, df, geom, i)
    return kernel


@pytest.mark.parametrize("df", DST)
@pytest.mark.parametrize("sf", SRC)
def test_pairs(dev, orc, sf, df):
    """every plane source into every 8- / 10-bit YUV destination, down- and up-scaled and at odd sizes, aligned and odd-aligned planes: the pairs whose layouts
    differ run the tile kernel below 4 : 1 (the others stay with their walkers — whatever serves them, the bytes)"""
    for geom in ((192, 108, 128, 72), (96, 40, 144, 60), (101, 45, 75, 33)):
        for align, extra in ((64, 0), (2, 2)):
            k = _check(dev, orc, sf, df, geom, "bicubic", align, extra)
            cascade = {sf, df} == {"nv12", "yuv420p"}           # (round 4's cascade keeps the 8-bit NV12 <-> YUV420P frames: a walker in the source's layout, then a copy)
            # (an 8-bit 4:4:4 source into 4:2:0: the band walker's plane jobs where it has an instance, else — its luma and chroma jobs need different pair counts — the tiled kernel)
            if mixed(sf, df) and not cascade and sf != "yuv444p":
                assert k == "scale19_kernel", (k, geom, align)


@pytest.mark.parametrize("flags", ["bicubic", "lanczos", "bilinear", "point", "area", "gauss", "spline", "fast_bilinear"])
@pytest.mark.parametrize("pair", [("nv12", "yuv420p10le"), ("p010le", "yuv420p"), ("yuv420p16le", "nv12"), ("nv12", "yuv444p"), ("yuv444p", "nv12"), ("yuv444p16le", "yuv444p"),
                                  ("p016le", "yuv444p"), ("yuv420p", "p010le")])
def test_algorithms(dev, orc, pair, flags):
    sf, df = pair
    for geom in ((320, 180, 128, 72), (128, 72, 320, 180), (258, 66, 129, 33), (66, 34, 131, 67)):
        k = _check(dev, orc, sf, df, geom, flags, seed=31)
        assert k is not None


@pytest.mark.parametrize("pair", [("p010le", "yuv420p"), ("yuv420p16le", "nv12"), ("yuv444p16le", "yuv444p"), ("p016le", "yuv444p"), ("yuv420p10le", "nv12")])
def test_dither_of_a_deep_source(dev, orc, pair):
    """8-bit output of a source deeper than 8 bits: ff_dither_8x8_128 by (x, y), the V plane (the V bytes of an interleaved row) three columns on
    (swscale.c:263-264, vscale.c:98-101) — widths and heights that walk the 8 x 8 pattern through every phase"""
    sf, df = pair
    for geom in ((200, 120, 136, 88), (136, 88, 200, 120), (88, 24, 90, 26)):
        assert _check(dev, orc, sf, df, geom, "bicubic", seed=7) == "scale19_kernel"


def test_range_conversion_and_chroma_positions(dev, orc):
    """lum / chrRange{To,From}Jpeg_c on the 15-bit lines of a tile (swscale.c:157-188), chroma positions (the plan's own banks)"""
    L, lib = orc.L, dev.lib
    L.orc_sws_create_ex.restype = C.c_void_p
    L.orc_sws_create_ex.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    for sf, df in (("nv12", "yuv420p10le"), ("p010le", "yuv420p"), ("yuv444p16le", "nv12"), ("nv12", "yuv444p")):
        for ranges, pos in (((0, 1), (-513,) * 4), ((1, 0), (-513,) * 4), ((0, 0), (128, 0, 256, 64)), ((1, 0), (37, -200, 511, 3))):
            sw, sh, dw, dh = 192, 80, 132, 60
            src = _synth(orc, sf, sw, sh, 61)
            oc = L.orc_sws_create_ex(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], SWS["bicubic"], None, (C.c_int * 4)(*pos), ranges[0], ranges[1])
            assert oc
            want = alloc_planes(df, dw, dh)
            assert L.orc_sws_scale(oc, planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                                   planes([p.ctypes.data for p in want]), ints([p.strides[0] for p in want])) == dh
            L.orc_sws_free(oc)
            c = lib.gmat_sws_getContext(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], SWS["bicubic"], None)
            assert c and lib.gmat_sws_setRange(c, ranges[0], ranges[1]) == 0 and lib.gmat_sws_setChromaPos(c, *pos) == 0
            d = dev.upload_planes(src, 64)
            dst = dev.planes_like(df, dw, dh, 64)
            assert lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, sh,
                                      planes([p.ptr for p in dst]), ints([p.stride for p in dst])) == dh
            assert lib.gmat_sws_lastKernel(c).decode() == "scale19_kernel"
            for i, (p, wv) in enumerate(zip(dst, want)):
                assert (p.download() == wv).all(), (sf, df, ranges, pos, i)
            lib.gmat_sws_freeContext(c)
            for p in d + dst:
                p.free()


@pytest.mark.parametrize("pair", [("nv12", "yuv420p10le"), ("p010le", "yuv420p"), ("yuv444p16le", "nv12"), ("p016le", "yuv444p")])
def test_batch_is_one_launch(dev, orc, pair):
    sf, df = pair
    lib = dev.lib
    sw, sh, dw, dh, nf = 160, 90, 104, 58, 5
    srcs = [_synth(orc, sf, sw, sh, 300 + f) for f in range(nf)]
    wants = [orc.sws(s, sw, sh, sf, dw, dh, df, SWS["bicubic"]) for s in srcs]
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], SWS["bicubic"], None)
    assert c
    dsrc = [dev.upload_planes(s, 64) for s in srcs]
    ddst = [dev.planes_like(df, dw, dh, 64) for _ in srcs]
    sp, dp = (C.c_void_p * (4 * nf))(), (C.c_void_p * (4 * nf))()
    for f in range(nf):
        for i, p in enumerate(dsrc[f]):
            sp[4 * f + i] = p.ptr
        for i, p in enumerate(ddst[f]):
            dp[4 * f + i] = p.ptr
    assert lib.gmat_sws_scale_batch(c, nf, C.cast(sp, C.POINTER(C.c_void_p)), ints([p.stride for p in dsrc[0]]), C.cast(dp, C.POINTER(C.c_void_p)),
                                    ints([p.stride for p in ddst[0]]), C.cast((C.c_void_p * 1)(None), C.POINTER(C.c_void_p)), 1, 3) == nf
    lib.gmat_device_sync()
    assert lib.gmat_sws_lastKernel(c).decode() == "scale19_kernel" and lib.gmat_sws_lastLaunchFrames(c) == nf
    for f in range(nf):
        for i, (a, b) in enumerate(zip(ddst[f], wants[f])):
            assert (a.download() == b).all(), (pair, f, i)
    for fr in dsrc + ddst:
        for p in fr:
            p.free()
    lib.gmat_sws_freeContext(c)


def test_the_rule(dev, orc, monkeypatch):
    """in front of the lines form only where the layouts differ, an end is deep (16-bit samples in, 10 bits or the dithered 8 out) and the ratio is below 4 : 1 — 8-bit ends
    keep the lines form where its rule takes them (launches of more than three frames, 2 : 1 and beyond); in front of the tiled catch-all wherever it has a plan; the 8-bit NV12 <-> YUV420P
    cascade (a walker in the source's layout, then a copy) keeps its frames; GMAT_T15=0: the kernels behind it"""
    assert _check(dev, orc, "nv12", "yuv444p", (384, 216, 256, 144)) == "scale19_kernel"
    assert _check(dev, orc, "nv12", "yuv444p", (640, 360, 128, 72)) != "scale19_kernel"              # 5 : 1: the lines form
    assert _check(dev, orc, "nv12", "yuv444p", (512, 288, 256, 144)) != "scale19_kernel"             # 2 : 1, 8-bit ends: the lines form
    assert _check(dev, orc, "p016le", "yuv444p", (512, 288, 256, 144)) == "scale19_kernel"           # ... a deep end: the tile kernel
    assert _check(dev, orc, "nv12", "yuv420p", (384, 216, 256, 144)) != "scale19_kernel"              # the cascade: a walker in the source's layout, then the re-layout
    assert _check(dev, orc, "p010le", "yuv420p", (384, 216, 256, 144)) == "scale19_kernel"
    monkeypatch.setenv("GMAT_T15", "0")
    assert _check(dev, orc, "nv12", "yuv444p", (384, 216, 256, 144)) != "scale19_kernel"
    assert _check(dev, orc, "p010le", "yuv420p", (384, 216, 256, 144)) != "scale19_kernel"


@pytest.mark.gpu
def test_full_size(dev, orc):
    if dev.kind != "hip":
        pytest.skip("full-size frames run on the real GPU only")
    for sf, df, geom in (("nv12", "yuv420p10le", (1920, 1080, 1280, 720)), ("p010le", "yuv420p", (3840, 2160, 1920, 1088)), ("yuv444p16le", "nv12", (1920, 1080, 1280, 720)),
                         ("nv12", "yuv444p", (1920, 1080, 1920, 1080))):
        assert _check(dev, orc, sf, df, geom, "bicubic", 256, 0, seed=3) == "scale19_kernel"
