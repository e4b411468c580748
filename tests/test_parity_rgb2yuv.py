"""RGB -> YUV 4:2:0 (same size) and NV12 <-> YUV420P vs the oracle, bit-exact."""
import numpy as np
import pytest

from harness import PIX_FMT, SWS, planes, ints, synth_planes

SIZES = [(128, 32), (256, 70), (130, 34), (66, 18), (6, 4), (2, 2), (300, 66)]


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("src_fmt", ["rgb24", "bgr24"])
@pytest.mark.parametrize("dst_fmt", ["nv12", "yuv420p"])
def test_rgb_to_yuv420_bit_exact(dev, orc, w, h, src_fmt, dst_fmt):
    src = synth_planes(orc, src_fmt, w, h, seed=51)
    want = orc.sws(src, w, h, src_fmt, w, h, dst_fmt, SWS["bicubic"])
    for align, extra in [(256, 0), (1, 1)]:
        d_src = dev.upload_planes(src, align, extra)
        got, pads, kernel = dev.sws(d_src, w, h, src_fmt, w, h, dst_fmt, SWS["bicubic"], dst_align=align, dst_extra=extra)
        assert kernel == "rgb2yuv420_kernel"
        for i, (g, wv) in enumerate(zip(got, want)):
            bad = np.argwhere(g != wv)
            assert bad.size == 0, f"plane {i}: {len(bad)} mismatches, first {bad[:4].tolist()} (align {align})"
        for p in pads:
            assert (p == 0xCD).all()


@pytest.mark.parametrize("w,h", SIZES + [(1, 1), (257, 5), (1023, 3)])
@pytest.mark.parametrize("src_fmt", ["rgb24", "bgr24"])
def test_rgb_to_yuv444p_bit_exact(dev, orc, w, h, src_fmt):
    """no subsampling at either end: rgb24ToY_c / rgb24ToUV_c, one-tap filters, yuv2plane1_8_c for the three planes"""
    src = synth_planes(orc, src_fmt, w, h, seed=57)
    for flags in ("bicubic", "point"):
        want = orc.sws(src, w, h, src_fmt, w, h, "yuv444p", SWS[flags])
        for align, extra in [(256, 0), (1, 1)]:
            d_src = dev.upload_planes(src, align, extra)
            got, pads, kernel = dev.sws(d_src, w, h, src_fmt, w, h, "yuv444p", SWS[flags], dst_align=align, dst_extra=extra)
            assert kernel == "rgb2yuv444_kernel"
            for i, (g, wv) in enumerate(zip(got, want)):
                bad = np.argwhere(g != wv)
                assert bad.size == 0, f"plane {i}: {len(bad)} mismatches, first {bad[:4].tolist()} (align {align})"
            for p in pads:
                assert (p == 0xCD).all()


@pytest.mark.parametrize("cs", [1, 4, 5, 7, 9])          # ITU709, FCC, ITU601, SMPTE240M, BT2020 (swscale.h:98-107)
@pytest.mark.parametrize("dst_fmt", ["nv12", "yuv420p", "yuv444p"])
def test_rgb_to_yuv_colorspaces(dev, orc, cs, dst_fmt):
    """the destination's matrix: fill_rgb2yuv_table (utils.c:765-858) inverts the yuv2rgb coefficient row of the
    colourspace (BT.601 keeps its literals); gmat_sws_setColorspace on an RGB -> YUV context selects it"""
    w, h = 130, 34
    src = synth_planes(orc, "rgb24", w, h, seed=59)
    want = orc.sws(src, w, h, "rgb24", w, h, dst_fmt, SWS["bicubic"], colorspace=cs)
    d_src = dev.upload_planes(src, 64)
    got, _, kernel = dev.sws(d_src, w, h, "rgb24", w, h, dst_fmt, SWS["bicubic"], dst_align=64, colorspace=(cs, 0))
    for i, (g, wv) in enumerate(zip(got, want)):
        assert (g == wv).all(), (i, kernel)
    if cs not in (5, 6):
        base = orc.sws(src, w, h, "rgb24", w, h, dst_fmt, SWS["bicubic"])
        assert any((a != b).any() for a, b in zip(want, base))      # the matrix really changed something


@pytest.mark.parametrize("src_fmt", ["rgb24", "bgr24"])
@pytest.mark.parametrize("dst_fmt", ["nv12", "yuv420p"])
def test_rgb_to_full_range_yuv(dev, orc, src_fmt, dst_fmt):
    """a full-range YUV destination of an RGB source: lum/chrRangeToJpeg_c on the 15-bit lines (the RGB end's range is
    forced to 0); gmat_sws_setRange(c, 0, 1)"""
    import ctypes as C
    w, h = 130, 34
    L = orc.L
    L.orc_sws_create_ex.restype = C.c_void_p
    L.orc_sws_create_ex.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    from harness import alloc_planes
    src = synth_planes(orc, src_fmt, w, h, seed=61)
    oc = L.orc_sws_create_ex(w, h, PIX_FMT[src_fmt], w, h, PIX_FMT[dst_fmt], SWS["bicubic"], None, (C.c_int * 4)(-513, -513, -513, -513), 0, 1)
    assert oc
    want = alloc_planes(dst_fmt, w, h)
    assert L.orc_sws_scale(oc, planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                           planes([p.ctypes.data for p in want]), ints([p.strides[0] for p in want])) == h
    L.orc_sws_free(oc)
    c = dev.lib.gmat_sws_getContext(w, h, PIX_FMT[src_fmt], w, h, PIX_FMT[dst_fmt], SWS["bicubic"], None)
    assert c and dev.lib.gmat_sws_setRange(c, 0, 1) == 0 and dev.lib.gmat_sws_setRange(c, 1, 1) < 0
    assert dev.lib.gmat_sws_setRange(c, 0, 1) == 0
    d = dev.upload_planes(src, 64)
    dst = dev.planes_like(dst_fmt, w, h, 64)
    assert dev.lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, h,
                                  planes([p.ptr for p in dst]), ints([p.stride for p in dst])) == h
    for a, b in zip(dst, want):
        assert (a.download() == b).all()
    base = orc.sws(src, w, h, src_fmt, w, h, dst_fmt, SWS["bicubic"])
    assert (base[0] != want[0]).any()
    dev.lib.gmat_sws_freeContext(c)
    for p in d + dst:
        p.free()


@pytest.mark.parametrize("flags", ["bilinear", "point", "lanczos"])
def test_rgb_to_nv12_other_vertical_filters(dev, orc, flags):
    w, h = 192, 40
    src = synth_planes(orc, "rgb24", w, h, seed=53)
    want = orc.sws(src, w, h, "rgb24", w, h, "nv12", SWS[flags])
    d_src = dev.upload_planes(src, 64)
    got, _, _ = dev.sws(d_src, w, h, "rgb24", w, h, "nv12", SWS[flags], dst_align=64)
    assert all((g == wv).all() for g, wv in zip(got, want))


def test_reference_entry_point_rgb2yuv_cuda(dev, orc):
    w, h = 128, 36
    src = synth_planes(orc, "rgb24", w, h, seed=55)
    want = orc.sws(src, w, h, "rgb24", w, h, "nv12")
    d_src = dev.upload_planes(src, 256)
    dst = dev.planes_like("nv12", w, h, 256)
    for _ in range(2):                                   # second call hits the cached tables
        r = dev.lib.rgb2yuv_cuda(planes([p.ptr for p in d_src]), ints([p.stride for p in d_src]),
                                 planes([p.ptr for p in dst]), ints([p.stride for p in dst]), w, h,
                                 PIX_FMT["rgb24"], PIX_FMT["nv12"], None)
        assert r == 0
    assert all((d.download() == wv).all() for d, wv in zip(dst, want))


@pytest.mark.parametrize("w,h", [(64, 16), (130, 34), (5, 3)])
@pytest.mark.parametrize("pair", [("nv12", "yuv420p"), ("yuv420p", "nv12"), ("nv12", "nv12"), ("yuv420p", "yuv420p")])
def test_yuv_relayout_lossless(dev, orc, w, h, pair):
    s, d = pair
    src = synth_planes(orc, s, w, h, seed=57)
    d_src = dev.upload_planes(src, 64)
    got, pads, kernel = dev.sws(d_src, w, h, s, w, h, d, dst_align=64)
    cw, ch = (w + 1) // 2, (h + 1) // 2
    if s == "nv12":
        u, v = src[1][:, 0::2], src[1][:, 1::2]
    else:
        u, v = src[1], src[2]
    assert (got[0] == src[0]).all()
    if d == "nv12":
        assert (got[1][:, 0::2] == u).all() and (got[1][:, 1::2] == v).all()
    else:
        assert (got[1] == u).all() and (got[2] == v).all()
    for p in pads:
        assert (p == 0xCD).all()
    # the reference's symbol
    dst = dev.planes_like(d, w, h, 64)
    r = dev.lib.yuv2yuv_cuda(planes([p.ptr for p in d_src]), ints([p.stride for p in d_src]),
                             planes([p.ptr for p in dst]), ints([p.stride for p in dst]), w, h, PIX_FMT[s], PIX_FMT[d], None)
    assert r == 0 and all((a.download() == b).all() for a, b in zip(dst, got))


@pytest.mark.parametrize("w,h", [(64, 16), (130, 34), (131, 35), (8, 2), (1920, 8)])
@pytest.mark.parametrize("src_fmt", ["yuv420p", "nv12"])
@pytest.mark.parametrize("dst_fmt", ["p010le", "p016le"])
def test_depth_expansion_to_p01x(dev, orc, w, h, src_fmt, dst_fmt):
    """planar8ToP01xleWrapper: t -> t | t << 8 (swscale_unscaled.c:286-324); odd widths leave the last chroma pair
    unwritten exactly like the CPU loop (srcW / 2 pairs)."""
    from harness import alloc_planes
    src = synth_planes(orc, src_fmt, w, h, seed=59)
    want = alloc_planes(dst_fmt, w, h, fill=0xCD)
    orc.L.orc_yuv420_to_p01x(planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                             planes([p.ctypes.data for p in want]), ints([p.strides[0] for p in want]), w, h,
                             1 if src_fmt == "nv12" else 0)
    for align, extra in [(256, 0), (2, 2)]:
        d_src = dev.upload_planes(src, align if align > 2 else 1, extra)
        got, pads, kernel = dev.sws(d_src, w, h, src_fmt, w, h, dst_fmt, dst_align=align, dst_extra=extra)
        assert kernel == "widen8to16_kernel"
        for g, wv, pd in zip(got, want, pads):
            assert (g == wv).all()
            assert (pd == 0xCD).all()
        for p in d_src:
            p.free()
    # the reference's symbol
    d_src = dev.upload_planes(src, 64)
    dst = dev.planes_like(dst_fmt, w, h, 64)
    r = dev.lib.yuv2yuv_cuda(planes([p.ptr for p in d_src]), ints([p.stride for p in d_src]),
                             planes([p.ptr for p in dst]), ints([p.stride for p in dst]), w, h, PIX_FMT[src_fmt],
                             PIX_FMT[dst_fmt], None)
    assert r == 0 and all((a.download() == b).all() for a, b in zip(dst, want))
