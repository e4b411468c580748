"""RGB -> YUV 4:2:0 (same size) and NV12 <-> YUV420P vs the oracle, bit-exact."""
import numpy as np
import pytest

from harness import is_generic, PIX_FMT, SWS, planes, ints, synth_planes

# one partial strip of the strip kernel (512 pixel columns per wave), exactly one, one and a bit, several workgroups of strips, widths
# that are multiples of 8 only; then what it declines: widths off the 8-pixel grid, odd heights, tiny frames
SIZES = [(128, 32), (256, 70), (512, 16), (520, 20), (1032, 16), (2056, 18), (2568, 16), (72, 18), (130, 34), (66, 18), (6, 4), (2, 2), (300, 66),
         (128, 33), (128, 6), (128, 14)]


def strip_takes(w, h, align):
    """the rule of rgb2yuv420_strip_takes restated: whole 8-pixel lanes, at least 64 columns, rows pair up, at least 8 chroma rows
    (the replication check compares every row of the table with its middle row, whose 8-tap window must be interior), every plane
    movable in dwords — and the context's vertical chroma filter must be the replicated 8-tap window (true for bicubic)"""
    return w % 8 == 0 and w >= 64 and h % 2 == 0 and h >= 16 and align % 4 == 0


@pytest.fixture(params=["strip", "tiled"])
def kern_r2y(request, monkeypatch):
    if request.param == "tiled":
        monkeypatch.setenv("GMAT_SCALE_NO_STRIP", "1")
    else:
        monkeypatch.delenv("GMAT_SCALE_NO_STRIP", raising=False)
    return request.param


def test_sizes_cover_both_kernels():
    took = [strip_takes(w, h, 256) for w, h in SIZES]
    assert sum(took) >= 8 and took.count(False) >= 6


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("src_fmt", ["rgb24", "bgr24"])
@pytest.mark.parametrize("dst_fmt", ["nv12", "yuv420p"])
def test_rgb_to_yuv420_bit_exact(dev, orc, kern_r2y, w, h, src_fmt, dst_fmt):
    """both kernels of the same-size RGB -> 4:2:0 conversion against the oracle: the strip walker (rgb2yuv420s_kernel) where its
    rule takes the frame, the tiled kernel everywhere else and everywhere with GMAT_SCALE_NO_STRIP=1"""
    src = synth_planes(orc, src_fmt, w, h, seed=51)
    want = orc.sws(src, w, h, src_fmt, w, h, dst_fmt, SWS["bicubic"])
    for align, extra in [(256, 0), (4, 4), (1, 1)]:
        d_src = dev.upload_planes(src, align, extra)
        got, pads, kernel = dev.sws(d_src, w, h, src_fmt, w, h, dst_fmt, SWS["bicubic"], dst_align=align, dst_extra=extra)
        assert kernel == ("rgb2yuv420s_kernel" if kern_r2y == "strip" and strip_takes(w, h, align) else "rgb2yuv420_kernel"), (kernel, align)
        for i, (g, wv) in enumerate(zip(got, want)):
            bad = np.argwhere(g != wv)
            assert bad.size == 0, f"{kernel} plane {i}: {len(bad)} mismatches, first {bad[:4].tolist()} (align {align})"
        for p in pads:
            assert (p == 0xCD).all()
        for p in d_src:
            p.free()


@pytest.mark.parametrize("sat", ["proven-dead", "kept"])
@pytest.mark.parametrize("rows", [1, 2, 3, 4, 5, 7, 16, 1000])
@pytest.mark.parametrize("dst_fmt", ["nv12", "yuv420p"])
def test_rgb_to_yuv420_strip_segmentation(dev, orc, monkeypatch, dst_fmt, rows, sat):
    """segments of `rows` chroma rows: the 3 warm-up row pairs of every segment re-create the vertical chroma window, their luma
    belongs to the neighbouring segment and is written exactly once.  Both instantiations: the one without the saturations the
    host has proven dead from the coefficients (the default for every standard matrix) and, with GMAT_R2Y_NOSAT=0, the one that
    keeps them"""
    monkeypatch.setenv("GMAT_STRIP_ROWS", str(rows))
    if sat == "kept":
        monkeypatch.setenv("GMAT_R2Y_NOSAT", "0")
    else:
        monkeypatch.delenv("GMAT_R2Y_NOSAT", raising=False)
    w, h = 520, 46
    src = synth_planes(orc, "rgb24", w, h, seed=53)
    want = orc.sws(src, w, h, "rgb24", w, h, dst_fmt, SWS["bicubic"])
    d_src = dev.upload_planes(src, 64)
    got, pads, kernel = dev.sws(d_src, w, h, "rgb24", w, h, dst_fmt, SWS["bicubic"], dst_align=64)
    assert kernel == "rgb2yuv420s_kernel"
    for g, wv, p in zip(got, want, pads):
        assert (g == wv).all() and (p == 0xCD).all()


@pytest.mark.parametrize("flags", ["bicubic", "bilinear", "point", "area", "lanczos", "gauss"])
def test_rgb_to_yuv420_other_filters(dev, orc, kern_r2y, flags):
    """the vertical chroma filter decides: whatever is the replicated 8-tap window takes the strip kernel, the rest stays tiled;
    the bytes are libswscale's either way"""
    w, h = 264, 38
    src = synth_planes(orc, "rgb24", w, h, seed=55)
    want = orc.sws(src, w, h, "rgb24", w, h, "nv12", SWS[flags])
    d_src = dev.upload_planes(src, 64)
    got, _, kernel = dev.sws(d_src, w, h, "rgb24", w, h, "nv12", SWS[flags], dst_align=64)
    assert kernel in ("rgb2yuv420s_kernel", "rgb2yuv420_kernel") or is_generic(kernel), kernel
    if kern_r2y == "tiled":
        assert kernel != "rgb2yuv420s_kernel"
    if flags == "bicubic" and kern_r2y == "strip":
        assert kernel == "rgb2yuv420s_kernel"
    for g, wv in zip(got, want):
        assert (g == wv).all(), (flags, kernel)


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


@pytest.mark.parametrize("geom", [(130, 34), (136, 34)])      # the tiled kernel's / the strip kernel's geometry
@pytest.mark.parametrize("cs", [1, 4, 5, 7, 9])          # ITU709, FCC, ITU601, SMPTE240M, BT2020 (swscale.h:98-107)
@pytest.mark.parametrize("dst_fmt", ["nv12", "yuv420p", "yuv444p"])
def test_rgb_to_yuv_colorspaces(dev, orc, cs, dst_fmt, geom):
    """the destination's matrix: fill_rgb2yuv_table (utils.c:765-858) inverts the yuv2rgb coefficient row of the
    colourspace (BT.601 keeps its literals); gmat_sws_setColorspace on an RGB -> YUV context selects it"""
    w, h = geom
    src = synth_planes(orc, "rgb24", w, h, seed=59)
    want = orc.sws(src, w, h, "rgb24", w, h, dst_fmt, SWS["bicubic"], colorspace=cs)
    d_src = dev.upload_planes(src, 64)
    got, _, kernel = dev.sws(d_src, w, h, "rgb24", w, h, dst_fmt, SWS["bicubic"], dst_align=64, colorspace=(cs, 0))
    for i, (g, wv) in enumerate(zip(got, want)):
        assert (g == wv).all(), (i, kernel)
    if dst_fmt != "yuv444p":
        assert kernel == ("rgb2yuv420s_kernel" if strip_takes(w, h, 64) else "rgb2yuv420_kernel"), kernel
    if cs not in (5, 6):
        base = orc.sws(src, w, h, "rgb24", w, h, dst_fmt, SWS["bicubic"])
        assert any((a != b).any() for a, b in zip(want, base))      # the matrix really changed something


@pytest.mark.parametrize("geom", [(130, 34), (136, 34)])      # the tiled kernel's / the strip kernel's geometry
@pytest.mark.parametrize("src_fmt", ["rgb24", "bgr24"])
@pytest.mark.parametrize("dst_fmt", ["nv12", "yuv420p"])
def test_rgb_to_full_range_yuv(dev, orc, src_fmt, dst_fmt, geom):
    """a full-range YUV destination of an RGB source: lum/chrRangeToJpeg_c on the 15-bit lines (the RGB end's range is
    forced to 0); gmat_sws_setRange(c, 0, 1)"""
    import ctypes as C
    w, h = geom
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
    assert dev.lib.gmat_sws_lastKernel(c) == (b"rgb2yuv420s_kernel" if strip_takes(w, h, 64) else b"rgb2yuv420_kernel")
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
    unwritten exactly like the CPU loop (srcW / 2 pairs).  libswscale selects it for PLANAR 8-bit sources only (:2108-2112): an
    NV12 source runs the generic lines (t << 8), as the reference's real core showed (round 4, tests/test_libswscale_core.py)."""
    from harness import alloc_planes
    src = synth_planes(orc, src_fmt, w, h, seed=59)
    if src_fmt == "nv12":
        want = orc.sws(src, w, h, src_fmt, w, h, dst_fmt)
        d_src = dev.upload_planes(src, 64)
        got, pads, kernel = dev.sws(d_src, w, h, src_fmt, w, h, dst_fmt, dst_align=64)
        assert kernel != "widen8to16_kernel"
        for g, wv in zip(got, want):
            assert (g == wv).all()
        assert (got[0].view("<u2") == src[0].astype(np.uint16) << 8).all()          # known answer: the sample in the high byte
        dst = dev.planes_like(dst_fmt, w, h, 64)
        r = dev.lib.yuv2yuv_cuda(planes([p.ptr for p in d_src]), ints([p.stride for p in d_src]),
                                 planes([p.ptr for p in dst]), ints([p.stride for p in dst]), w, h, PIX_FMT[src_fmt], PIX_FMT[dst_fmt], None)
        assert r == 0 and all((a.download() == b).all() for a, b in zip(dst, want))
        for p in d_src + dst:
            p.free()
        return
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


@pytest.mark.parametrize("w,h", [(128, 32), (520, 20), (1032, 16), (72, 18), (130, 34), (300, 66), (128, 33)])
@pytest.mark.parametrize("src_fmt", ["rgba", "bgra"])
@pytest.mark.parametrize("dst_fmt", ["nv12", "yuv420p"])
def test_rgba_to_yuv420_reads_the_pixels_as_they_are(dev, orc, w, h, src_fmt, dst_fmt, monkeypatch):
    """rgb2yuv_cuda's RGBA / BGRA sources (libswscale/cuda/yuv2rgb_cuda.cu:909-947) at equal size: rgb32ToY / ToUV read the same three channels, so the strip
    kernel's <PX = 4> instances take the frame where its rule does (round 6: a 32 -> 24-bit pass ran in front — two launches a frame, batches frame by frame); the
    pass stays behind GMAT_RGBSRC_NO_PX4=1 and for the frames the rule declines.  Single calls, a batch (one launch), the full-range destination"""
    import ctypes as C
    lib = dev.lib
    src = synth_planes(orc, src_fmt, w, h, seed=57)
    want = orc.sws(src, w, h, src_fmt, w, h, dst_fmt, SWS["bicubic"])
    for knob in ("", "1"):
        monkeypatch.setenv("GMAT_RGBSRC_NO_PX4", knob) if knob else monkeypatch.delenv("GMAT_RGBSRC_NO_PX4", raising=False)
        for align, extra in [(256, 0), (4, 4)]:
            d_src = dev.upload_planes(src, align, extra)
            got, pads, kernel = dev.sws(d_src, w, h, src_fmt, w, h, dst_fmt, SWS["bicubic"], dst_align=align, dst_extra=extra)
            assert kernel == ("rgb2yuv420s_kernel" if strip_takes(w, h, align) else "rgb2yuv420_kernel"), (kernel, align)
            for i, (g, wv) in enumerate(zip(got, want)):
                assert (g == wv).all(), (kernel, knob, align, i, np.argwhere(g != wv)[:3].tolist())
            for p in pads:
                assert (p == 0xCD).all()
            for p in d_src:
                p.free()
    monkeypatch.delenv("GMAT_RGBSRC_NO_PX4", raising=False)
    # a batch of three frames through gmat_sws_scale_batch: one launch where the strip rule takes the frames
    nf = 3
    srcs = [synth_planes(orc, src_fmt, w, h, seed=60 + f) for f in range(nf)]
    wants = [orc.sws(s_, w, h, src_fmt, w, h, dst_fmt, SWS["bicubic"]) for s_ in srcs]
    c = lib.gmat_sws_getContext(w, h, PIX_FMT[src_fmt], w, h, PIX_FMT[dst_fmt], SWS["bicubic"], None)
    assert c
    dsrc = [dev.upload_planes(s_, 256) for s_ in srcs]
    ddst = [dev.planes_like(dst_fmt, w, h, 256) for _ in srcs]
    sp, dp = (C.c_void_p * (4 * nf))(), (C.c_void_p * (4 * nf))()
    for f in range(nf):
        for i, p in enumerate(dsrc[f]):
            sp[4 * f + i] = p.ptr
        for i, p in enumerate(ddst[f]):
            dp[4 * f + i] = p.ptr
    assert lib.gmat_sws_scale_batch(c, nf, C.cast(sp, C.POINTER(C.c_void_p)), ints([p.stride for p in dsrc[0]]), C.cast(dp, C.POINTER(C.c_void_p)),
                                    ints([p.stride for p in ddst[0]]), C.cast((C.c_void_p * 1)(None), C.POINTER(C.c_void_p)), 1, 3) == nf
    lib.gmat_device_sync()
    if strip_takes(w, h, 256):
        assert lib.gmat_sws_lastKernel(c).decode() == "rgb2yuv420s_kernel" and lib.gmat_sws_lastLaunchFrames(c) == nf
    for f in range(nf):
        for a, b in zip(ddst[f], wants[f]):
            assert (a.download() == b).all(), (f, lib.gmat_sws_lastKernel(c).decode())
    lib.gmat_sws_freeContext(c)
    for fr in dsrc + ddst:
        for p in fr:
            p.free()
