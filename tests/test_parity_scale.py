"""The fused LDS-tiled scaler vs the oracle (bit-exact), through gmat_sws_scale."""
import numpy as np
import pytest

from harness import is_generic, PIX_FMT, SWS, synth_planes
from harness import LINES


def _check(dev, orc, src_fmt, sw, sh, dw, dh, dst_fmt, flags, fused=None, align=256, extra=0, seed=21):
    src = synth_planes(orc, src_fmt, sw, sh, seed=seed)
    if src_fmt in ("nv12", "yuv420p"):
        want = orc.chained(src, sw, sh, src_fmt, dw, dh, dst_fmt, flags)[0]
    else:
        want = orc.sws(src, sw, sh, src_fmt, dw, dh, dst_fmt, flags)[0]
    d_src = dev.upload_planes(src, align, extra)
    got, pads, kernel = dev.sws(d_src, sw, sh, src_fmt, dw, dh, dst_fmt, flags, fused=fused, dst_align=align,
                                dst_extra=extra)
    for p in d_src:
        p.free()
    bad = np.argwhere(got[0] != want)
    assert bad.size == 0, f"{len(bad)} mismatching bytes, first at {bad[:4].tolist()} ({kernel})"
    assert (pads[0] == 0xCD).all(), "kernel wrote into the row padding"
    return kernel


# (srcW, srcH, dstW, dstH): exact 2:1, odd sizes, upscale, anisotropic, tiny, > one tile in both axes
GEOMS = [(256, 64, 128, 32), (200, 90, 100, 45), (131, 77, 64, 33), (96, 40, 144, 60), (300, 50, 100, 70),
         (64, 64, 17, 9), (40, 30, 41, 31), (520, 36, 260, 18)]


@pytest.mark.parametrize("geom", GEOMS)
def test_rgb24_bicubic(dev, orc, kern, geom):
    """exact 2:1 geometries take the strip-walking kernel (k_scale_rgb2s.hip) unless it is switched off; everything else,
    and both with GMAT_SCALE_NO_STRIP=1, the block-cooperative form of round 5 where it has an instance, the tiled generic kernel where not"""
    sw, sh, dw, dh = geom
    k = _check(dev, orc, "rgb24", sw, sh, dw, dh, "rgb24", SWS["bicubic"])
    strip = kern.startswith("scale_yuv2s") and sw == 2 * dw and sh == 2 * dh and sw % 8 == 0 and sw >= 32 and dh >= 8
    # (round 5: away from 2 : 1, widths that are multiples of four, filters of three vertical taps or more: the block-cooperative RGB-source form,
    # tests/test_parity_walker16.py; GMAT_SCALE_NO_STRIP leaves it alone)
    # (... and exactly 2 : 1 wherever the strip kernel does not take the frame: switched off, or a width that is not a multiple of eight)
    blk = geom in ((96, 40, 144, 60), (40, 30, 41, 31), (300, 50, 100, 70)) or (not strip and geom in ((256, 64, 128, 32), (200, 90, 100, 45), (520, 36, 260, 18)))
    # (64 -> 17: ten coefficient pairs; the odd widths: pixel pairs)
    assert k.startswith("scale_rgb2h_kernel" if strip else "scale_yuvg_rgbsrc_blk_kernel" if blk else "scale_rgb_kernel"), k


@pytest.fixture(params=["shared", "per-lane"])
def rgb2_kernel(request, monkeypatch):
    """the two strip kernels of k_scale_rgb2s.hip: scale_rgb2h_kernel (default: every lane converts its own 8 pixels, the window's
    other samples come from the neighbouring lanes by DPP) and scale_rgb2s_kernel (GMAT_RGB2_SHARED=0: every lane converts its
    whole 14-pixel window)"""
    if request.param == "per-lane":
        monkeypatch.setenv("GMAT_RGB2_SHARED", "0")
        return "scale_rgb2s_kernel"
    monkeypatch.delenv("GMAT_RGB2_SHARED", raising=False)
    return "scale_rgb2h_kernel"


# (srcW, srcH, alignment): one partial wave, exactly the 248 / 256 output columns of one wave of either kernel and one group more,
# several workgroups of strips, widths that are multiples of 8 only
@pytest.mark.parametrize("src_fmt,dst_fmt", [("rgb24", "rgb24"), ("bgr24", "rgb24"), ("rgb24", "bgr24"), ("bgr24", "bgra"), ("rgb24", "rgba")])
@pytest.mark.parametrize("geom", [(32, 16, 4), (64, 32, 16), (496, 18, 4), (504, 18, 4), (512, 40, 64), (520, 24, 4), (1032, 36, 8), (2056, 20, 4),
                                  (2560, 18, 256)])
def test_rgb_strip_kernel_bit_exact(dev, orc, monkeypatch, rgb2_kernel, src_fmt, dst_fmt, geom):
    """k_scale_rgb2s.hip: partial strips, several strip groups, widths that are multiples of 8 only, frame edges
    (replicated pixels instead of libswscale's folded coefficient rows), both channel orders at both ends — on both kernels"""
    sw, sh, align = geom
    if dst_fmt in ("rgba", "bgra"):
        align = max(align, 16)
    for rows in (None, 1, 5):
        if rows is None:
            monkeypatch.delenv("GMAT_STRIP_ROWS", raising=False)
        else:
            monkeypatch.setenv("GMAT_STRIP_ROWS", str(rows))
        k = _check(dev, orc, src_fmt, sw, sh, sw // 2, sh // 2, dst_fmt, SWS["bicubic"], align=align)
        assert k == rgb2_kernel, k


@pytest.mark.parametrize("flags", ["bilinear", "point", "area", "gauss"])
def test_rgb_strip_kernel_other_filters(dev, orc, flags):
    k = _check(dev, orc, "rgb24", 320, 48, 160, 24, "rgb24", SWS[flags], align=16)
    assert k.startswith("scale_rgb"), k


@pytest.mark.parametrize("flags", ["bilinear", "lanczos", "point", "area", "bicubic"])
def test_rgb24_algorithms(dev, orc, flags):
    _check(dev, orc, "rgb24", 192, 70, 96, 35, "rgb24", SWS[flags])
    _check(dev, orc, "rgb24", 80, 30, 120, 50, "rgb24", SWS[flags])        # upscale: 2-tap special forms


@pytest.mark.parametrize("src_fmt,dst_fmt", [("bgr24", "rgb24"), ("rgb24", "bgr24"), ("rgb24", "rgba"),
                                             ("bgr24", "bgra")])
def test_packed_format_pairs(dev, orc, src_fmt, dst_fmt):
    _check(dev, orc, src_fmt, 160, 48, 80, 24, dst_fmt, SWS["bicubic"])


@pytest.mark.parametrize("align,extra", [(1, 0), (1, 3), (64, 0)])
def test_unaligned_strides(dev, orc, align, extra):
    _check(dev, orc, "rgb24", 136, 40, 68, 20, "rgb24", SWS["bicubic"], align=align, extra=extra)


@pytest.mark.parametrize("src_fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("geom", [(256, 64, 128, 32), (130, 50, 64, 25), (96, 40, 144, 60)])
def test_yuv_to_scaled_rgb_chained_contract(dev, orc, src_fmt, fused, geom):
    """4K nv12 -> rgb24 -> 1080p in miniature: fused and two-kernel forms both equal the chained oracle."""
    sw, sh, dw, dh = geom
    k = _check(dev, orc, src_fmt, sw, sh, dw, dh, "rgb24", SWS["bicubic"], fused=fused)
    assert ("yuv" in k) == bool(fused)
    if fused and fused_strip_takes(sw, sh, dw, dh):
        assert k == "scale_rgb2h_kernel<yuv>", k


def fused_strip_takes(sw, sh, dw, dh):
    """the rule of the fused convert-then-scale form on the strip kernel restated (rgb2s_prepare's geometry part): exactly 2:1,
    source width a multiple of 8 and >= 32, at least 8 output rows"""
    return sw == 2 * dw and sh == 2 * dh and sw % 8 == 0 and sw >= 32 and dh >= 8


@pytest.fixture(params=["strip", "tiled"])
def fused_kernel(request, monkeypatch):
    """the two kernels that serve setFused(1): scale_rgb2h_kernel<yuv> (exactly 2:1) and the tiled scale_rgb_kernel<..,yuv>
    (everything else, and everything with GMAT_RGB2_SHARED=0)"""
    if request.param == "tiled":
        monkeypatch.setenv("GMAT_RGB2_SHARED", "0")
    else:
        monkeypatch.delenv("GMAT_RGB2_SHARED", raising=False)
    return request.param


# (srcW, srcH): one partial wave, exactly the 248 output columns of one wave and one group more, several workgroups, widths that are
# multiples of 8 only, odd chroma row counts at the bottom (srcH = 2 * odd); then geometries the strip form declines
FUSED_GEOMS = [(32, 16), (64, 36), (496, 20), (504, 18), (520, 28), (1032, 20), (2056, 18), (264, 54), (36, 16), (64, 14), (130, 50)]


@pytest.mark.parametrize("src_fmt,dst_fmt", [("nv12", "rgb24"), ("yuv420p", "rgb24"), ("nv12", "bgra"), ("yuv420p", "bgr24"), ("nv12", "rgba")])
@pytest.mark.parametrize("geom", FUSED_GEOMS)
def test_fused_convert_then_scale_on_both_kernels(dev, orc, monkeypatch, fused_kernel, src_fmt, dst_fmt, geom):
    """setFused(1): sws(YUV -> RGB24 at the source size) then sws(RGB24 -> RGB at half the size) in one kernel, against the chained
    oracle: frame edges (a lane outside the frame presents the edge pixel's sample), the chroma row under the clamped luma row at
    the top and bottom, segment boundaries, both chroma layouts, both channel orders and 4-byte pixels at the output"""
    sw, sh = geom
    dw, dh = sw // 2, sh // 2
    align = 16 if dst_fmt in ("rgba", "bgra") else 4
    for rows in (None, 1, 5):
        if rows is None:
            monkeypatch.delenv("GMAT_STRIP_ROWS", raising=False)
        else:
            monkeypatch.setenv("GMAT_STRIP_ROWS", str(rows))
        k = _check(dev, orc, src_fmt, sw, sh, dw, dh, dst_fmt, SWS["bicubic"], fused=1, align=align)
        if fused_kernel == "strip" and fused_strip_takes(sw, sh, dw, dh):
            assert k == "scale_rgb2h_kernel<yuv>", k
        else:
            assert k.startswith("scale_rgb_kernel") and "yuv" in k, k


def test_fused_strip_form_declines_misaligned_planes(dev, orc):
    """dword loads on every plane: a destination or source off the 4-byte grid goes to the tiled kernel, same bytes"""
    k = _check(dev, orc, "nv12", 264, 40, 132, 20, "rgb24", SWS["bicubic"], fused=1, align=1, extra=1)
    assert k.startswith("scale_rgb_kernel") and "yuv" in k, k


def test_product_filter_tables_match_oracle(dev, orc):
    import ctypes as C
    lib = dev.lib
    for (sw, sh, dw, dh, flags) in [(3840, 2160, 1920, 1080, "bicubic"), (1920, 1080, 1280, 720, "lanczos"),
                                    (640, 360, 1920, 1080, "bilinear"), (1001, 777, 333, 555, "bicubic")]:
        ofs = orc.sws_filters(sw, sh, "rgb24", dw, dh, "rgb24", SWS[flags])
        c = lib.gmat_sws_getContext(sw, sh, PIX_FMT["rgb24"], dw, dh, PIX_FMT["rgb24"], SWS[flags], None)
        assert c
        for which, (ocoef, opos) in enumerate(ofs):
            n, taps = ocoef.shape
            coef = np.zeros((n, taps + 4), np.int16).reshape(-1)
            pos = np.zeros(n, np.int32)
            cnt = C.c_int()
            t = lib.gmat_sws_getFilter(c, which, coef.ctypes.data, pos.ctypes.data, n, C.byref(cnt))
            assert t == taps and cnt.value == n
            assert (coef[:n * taps].reshape(n, taps) == ocoef).all() and (pos == opos).all()
        lib.gmat_sws_freeContext(c)


# ---- YUV sources, libswscale single-context semantics (mode 2, the default) -----------------------------
YUV_GEOMS = [(256, 64, 128, 32), (130, 50, 64, 26), (96, 40, 144, 60), (200, 90, 100, 45), (64, 64, 18, 10),
             (128, 48, 128, 24), (128, 48, 64, 48), (66, 34, 33, 17), (40, 30, 41, 31), (520, 36, 260, 18)]


@pytest.mark.parametrize("src_fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("geom", YUV_GEOMS)
def test_yuv_single_context_bicubic(dev, orc, src_fmt, geom):
    """What ONE libswscale context nv12 -> rgb24 (different size) computes: planes scaled separately,
    LUT colour stage with half chroma (even dstW) or full chroma (odd dstW)."""
    sw, sh, dw, dh = geom
    src = synth_planes(orc, src_fmt, sw, sh, seed=31)
    want = orc.sws(src, sw, sh, src_fmt, dw, dh, "rgb24", SWS["bicubic"])[0]
    d_src = dev.upload_planes(src, 256)
    got, pads, kernel = dev.sws(d_src, sw, sh, src_fmt, dw, dh, "rgb24", SWS["bicubic"], dst_align=256)
    assert kernel.startswith("scale_yuv")
    if is_generic(kernel) and kernel != LINES:              # (the lines form's name does not carry its chroma form)
        assert ("full" in kernel) == bool(dw & 1)
    bad = np.argwhere(got[0] != want)
    assert bad.size == 0, f"{len(bad)} mismatching bytes, first at {bad[:4].tolist()} ({kernel})"
    assert (pads[0] == 0xCD).all()


@pytest.mark.parametrize("flags", ["bilinear", "lanczos", "point", "area", "fast_bilinear"])
@pytest.mark.parametrize("geom", [(192, 70, 96, 36), (80, 30, 120, 50), (64, 32, 128, 64)])
def test_yuv_single_context_algorithms(dev, orc, flags, geom):
    sw, sh, dw, dh = geom
    if flags == "fast_bilinear":
        pytest.skip("fast_bilinear uses the x86/C hyscale_fast path, not restated")
    src = synth_planes(orc, "nv12", sw, sh, seed=33)
    want = orc.sws(src, sw, sh, "nv12", dw, dh, "rgb24", SWS[flags])[0]
    d_src = dev.upload_planes(src, 64)
    got, pads, kernel = dev.sws(d_src, sw, sh, "nv12", dw, dh, "rgb24", SWS[flags], dst_align=64)
    bad = np.argwhere(got[0] != want)
    assert bad.size == 0, f"{flags}: {len(bad)} mismatching bytes, first at {bad[:4].tolist()} ({kernel})"


@pytest.mark.parametrize("dst_fmt", ["bgr24", "rgba", "bgra"])
@pytest.mark.parametrize("full", [0, 1])
def test_yuv_single_context_formats_and_full_chroma_flag(dev, orc, dst_fmt, full):
    sw, sh, dw, dh = 160, 48, 80, 24
    flags = SWS["bicubic"] | (SWS["full_chr_h_int"] if full else 0)
    src = synth_planes(orc, "nv12", sw, sh, seed=35)
    want = orc.sws(src, sw, sh, "nv12", dw, dh, dst_fmt, flags)[0]
    d_src = dev.upload_planes(src, 1, 3)                      # misaligned rows
    got, pads, kernel = dev.sws(d_src, sw, sh, "nv12", dw, dh, dst_fmt, flags, dst_align=1, dst_extra=1)
    assert is_generic(kernel) and ("full" in kernel) == bool(full)
    assert (got[0] == want).all() and (pads[0] == 0xCD).all()


# ---- the 2:1 horizontal specialisation (scale_yuv2x_kernel) --------------------------------------------
X2_GEOMS = [(256, 64, 128, 32), (512, 256, 256, 128), (64, 64, 32, 32), (1024, 96, 512, 48), (128, 60, 64, 30),
            (192, 34, 96, 17), (2048, 32, 1024, 16)]


@pytest.mark.parametrize("src_fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("geom", X2_GEOMS)
def test_yuv2x_specialisation_bit_exact(dev, orc, kern, src_fmt, geom):
    sw, sh, dw, dh = geom
    src = synth_planes(orc, src_fmt, sw, sh, seed=41)
    want = orc.sws(src, sw, sh, src_fmt, dw, dh, "rgb24", SWS["bicubic"])[0]
    d_src = dev.upload_planes(src, 256)
    got, pads, kernel = dev.sws(d_src, sw, sh, src_fmt, dw, dh, "rgb24", SWS["bicubic"], dst_align=256)
    assert kernel == kern, kernel
    bad = np.argwhere(got[0] != want)
    assert bad.size == 0, f"{len(bad)} mismatching bytes, first at {bad[:4].tolist()}"
    assert (pads[0] == 0xCD).all()
    # the generic kernel must give the same bytes (misaligned source rows force it)
    d_src2 = dev.upload_planes(src, 1, 2)
    got2, _, kernel2 = dev.sws(d_src2, sw, sh, src_fmt, dw, dh, "rgb24", SWS["bicubic"], dst_align=256)
    assert is_generic(kernel2) and (got2[0] == want).all()


@pytest.mark.parametrize("dst_fmt", ["bgr24", "rgba", "bgra"])
def test_yuv2x_dst_formats_and_bilinear(dev, orc, kern, dst_fmt):
    sw, sh, dw, dh = 256, 48, 128, 24
    src = synth_planes(orc, "nv12", sw, sh, seed=43)
    for flags in ("bicubic", "bilinear"):
        want = orc.sws(src, sw, sh, "nv12", dw, dh, dst_fmt, SWS[flags])[0]
        d_src = dev.upload_planes(src, 256)
        got, pads, kernel = dev.sws(d_src, sw, sh, "nv12", dw, dh, dst_fmt, SWS[flags], dst_align=256)
        assert kernel == kern
        assert (got[0] == want).all() and (pads[0] == 0xCD).all()


@pytest.mark.parametrize("src_fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("w,h", [(128, 48), (66, 34), (258, 20)])
def test_same_size_generic_path_with_accurate_rnd(dev, orc, src_fmt, w, h):
    """SWS_ACCURATE_RND on a same-size YUV->RGB context = libswscale's generic path (bicubic vertical chroma),
    SURVEY.md 8f.2; without the flag the nearest-chroma converter is used."""
    src = synth_planes(orc, src_fmt, w, h, seed=61)
    want = orc.sws(src, w, h, src_fmt, w, h, "rgb24", SWS["bicubic"])[0]
    d_src = dev.upload_planes(src, 64)
    got, pads, kernel = dev.sws(d_src, w, h, src_fmt, w, h, "rgb24", SWS["bicubic"] | SWS["accurate_rnd"], dst_align=64)
    assert kernel.startswith("scale_yuv")
    assert (got[0] == want).all() and (pads[0] == 0xCD).all()
    got2, _, kernel2 = dev.sws(d_src, w, h, src_fmt, w, h, "rgb24", SWS["bicubic"], dst_align=64)
    assert kernel2 == "yuv2rgb_kernel" and (got2[0] != want).any()


# ---- YUV 4:2:0 -> YUV 4:2:0 at another size (scale_cuda's job; planes scaled separately) -----------------
@pytest.mark.parametrize("src_fmt,dst_fmt", [("nv12", "nv12"), ("yuv420p", "yuv420p"), ("nv12", "yuv420p"),
                                             ("yuv420p", "nv12")])
@pytest.mark.parametrize("geom", [(256, 64, 128, 32), (200, 90, 100, 46), (131, 77, 65, 33), (96, 40, 144, 60),
                                  (300, 50, 100, 70), (64, 64, 18, 10), (520, 36, 260, 18)])
def test_yuv_to_yuv_scaled(dev, orc, src_fmt, dst_fmt, geom):
    sw, sh, dw, dh = geom
    src = synth_planes(orc, src_fmt, sw, sh, seed=41)
    want = orc.sws(src, sw, sh, src_fmt, dw, dh, dst_fmt, SWS["bicubic"])
    for align, extra in [(256, 0), (1, 1)]:
        d_src = dev.upload_planes(src, align, extra)
        got, pads, kernel = dev.sws(d_src, sw, sh, src_fmt, dw, dh, dst_fmt, SWS["bicubic"], dst_align=align,
                                    dst_extra=extra)
        for p in d_src:
            p.free()
        # a YUV-output kernel: the generic / tiled ones ("...yuv>"), or the plane-walking 2:1 one for the aligned pass of the
        # one geometry in this list that is exactly 2:1 with the same layout on both sides
        from test_parity_planes2p import strip_takes, strip_name
        if (dw, dh) == (sw // 2, sh // 2) and align % 4 == 0 and strip_takes(sw, sh, src_fmt, dst_fmt):
            assert kernel == strip_name(src_fmt, dst_fmt), kernel
        else:
            assert "yuv>" in kernel or kernel in ("scale_yuvg_kernel", "scale_yuvg_blk_kernel", "scale_yuvu_kernel", LINES, "scale19_kernel"), kernel      # (scale19_kernel: the tile kernel on the 15-bit lines, round 6 — the cross-layout pairs)
        for i, (g, w) in enumerate(zip(got, want)):
            bad = np.argwhere(g != w)
            assert bad.size == 0, f"plane {i}: {len(bad)} mismatching bytes, first at {bad[:4].tolist()} ({kernel})"
            assert (pads[i] == 0xCD).all(), f"plane {i}: kernel wrote into the row padding"


@pytest.mark.parametrize("flags", ["bilinear", "lanczos", "point", "area"])
def test_yuv_to_yuv_algorithms(dev, orc, flags):
    for (sw, sh, dw, dh) in [(192, 70, 96, 36), (80, 30, 120, 50)]:
        src = synth_planes(orc, "nv12", sw, sh, seed=43)
        want = orc.sws(src, sw, sh, "nv12", dw, dh, "nv12", SWS[flags])
        d_src = dev.upload_planes(src, 64)
        got, pads, kernel = dev.sws(d_src, sw, sh, "nv12", dw, dh, "nv12", SWS[flags], dst_align=64)
        for g, w in zip(got, want):
            assert (g == w).all(), (flags, kernel)


@pytest.mark.parametrize("src_fmt,dst_fmt", [("nv12", "nv12"), ("yuv420p", "yuv420p"), ("nv12", "yuv420p"),
                                             ("yuv420p", "nv12")])
@pytest.mark.parametrize("geom", X2_GEOMS)
def test_yuv2x_yuv_output_bit_exact(dev, orc, src_fmt, dst_fmt, geom):
    """The 2:1 specialisation with 4:2:0 output (the transcoding down-scale) vs the oracle, and vs the generic kernel."""
    sw, sh, dw, dh = geom
    src = synth_planes(orc, src_fmt, sw, sh, seed=45)
    want = orc.sws(src, sw, sh, src_fmt, dw, dh, dst_fmt, SWS["bicubic"])
    d_src = dev.upload_planes(src, 256)
    got, pads, kernel = dev.sws(d_src, sw, sh, src_fmt, dw, dh, dst_fmt, SWS["bicubic"], dst_align=256)
    # same-layout pairs the plane-walking kernel takes are ITS cases (tests/test_parity_planes2p.py runs this matrix on both
    # kernels); here the name must be the one the selection rule gives
    from test_parity_planes2p import strip_takes, strip_name
    assert kernel == (strip_name(src_fmt, dst_fmt) if strip_takes(sw, sh, src_fmt, dst_fmt) else "scale_yuv2x_kernel<yuv>"), kernel
    for i, (g, w) in enumerate(zip(got, want)):
        bad = np.argwhere(g != w)
        assert bad.size == 0, f"plane {i}: {len(bad)} mismatching bytes, first at {bad[:4].tolist()}"
        assert (pads[i] == 0xCD).all()
    # unaligned destination rows: byte stores, same bytes
    got2, pads2, kernel2 = dev.sws(d_src, sw, sh, src_fmt, dw, dh, dst_fmt, SWS["bicubic"], dst_align=1, dst_extra=1)
    assert kernel2 == "scale_yuv2x_kernel<yuv>"
    for g, w, pd in zip(got2, want, pads2):
        assert (g == w).all() and (pd == 0xCD).all()
    for p in d_src:
        p.free()
    for flags in ("bilinear", "lanczos"):
        want = orc.sws(src, sw, sh, src_fmt, dw, dh, dst_fmt, SWS[flags])
        d = dev.upload_planes(src, 256)
        got, _, kernel = dev.sws(d, sw, sh, src_fmt, dw, dh, dst_fmt, SWS[flags], dst_align=256)
        for g, w in zip(got, want):
            assert (g == w).all(), (flags, kernel)
        for p in d:
            p.free()


@pytest.mark.parametrize("ranges", [(0, 1), (1, 0)])
@pytest.mark.parametrize("geom", [(256, 64, 128, 32), (96, 40, 144, 60), (130, 50, 130, 50)])
def test_yuv_to_yuv_range_conversion(dev, orc, ranges, geom):
    """limited <-> full range on 4:2:0 -> 4:2:0 contexts (lum/chrRangeToJpeg_c / FromJpeg_c), scaled and same-size."""
    import ctypes as C
    from harness import planes, ints, alloc_planes
    sw, sh, dw, dh = geom
    src = synth_planes(orc, "nv12", sw, sh, seed=47)
    L = orc.L
    L.orc_sws_create_ex.restype = C.c_void_p
    L.orc_sws_create_ex.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    oc = L.orc_sws_create_ex(sw, sh, PIX_FMT["nv12"], dw, dh, PIX_FMT["nv12"], SWS["bicubic"], None,
                             (C.c_int * 4)(-513, -513, -513, -513), ranges[0], ranges[1])
    assert oc
    want = alloc_planes("nv12", dw, dh)
    assert L.orc_sws_scale(oc, planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                           planes([p.ctypes.data for p in want]), ints([p.strides[0] for p in want])) == dh
    L.orc_sws_free(oc)
    lib = dev.lib
    d = dev.upload_planes(src, 256)
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT["nv12"], dw, dh, PIX_FMT["nv12"], SWS["bicubic"], None)
    assert c and lib.gmat_sws_setRange(c, ranges[0], ranges[1]) == 0
    dst = dev.planes_like("nv12", dw, dh, 256)
    assert lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, sh,
                              planes([p.ptr for p in dst]), ints([p.stride for p in dst])) == dh
    for a, b in zip(dst, want):
        assert (a.download() == b).all()
    lib.gmat_sws_freeContext(c)
    for p in d + dst:
        p.free()


@pytest.mark.parametrize("dst_fmt", ["yuv420p", "nv12", "rgb24", "bgra"])
@pytest.mark.parametrize("geom", [(256, 64, 128, 32), (130, 50, 130, 50), (96, 40, 144, 60), (200, 90, 151, 67)])
def test_yuv444p_source(dev, orc, dst_fmt, geom):
    """planar 4:4:4 sources: chroma planes at full size, full chroma interpolation forced for RGB outputs
    (utils.c:1439-1447), chroma filtered 2:1 for 4:2:0 outputs even at the same luma size."""
    sw, sh, dw, dh = geom
    src = synth_planes(orc, "yuv444p", sw, sh, seed=49)
    want = orc.sws(src, sw, sh, "yuv444p", dw, dh, dst_fmt, SWS["bicubic"])
    d = dev.upload_planes(src, 64)
    got, pads, kernel = dev.sws(d, sw, sh, "yuv444p", dw, dh, dst_fmt, SWS["bicubic"], dst_align=64)
    assert is_generic(kernel), kernel
    for i, (g, wv) in enumerate(zip(got, want)):
        bad = np.argwhere(g != wv)
        assert bad.size == 0, f"plane {i}: {len(bad)} mismatching bytes, first at {bad[:4].tolist()} ({kernel})"
        assert (pads[i] == 0xCD).all()
    for p in d:
        p.free()


@pytest.mark.parametrize("src_fmt", ["yuv420p", "nv12", "yuv444p"])
@pytest.mark.parametrize("flags", ["bicubic", "bilinear", "lanczos"])
@pytest.mark.parametrize("geom", [(256, 64, 128, 32), (130, 50, 130, 50), (96, 40, 144, 60), (201, 91, 151, 67)])
def test_yuv444p_output(dev, orc, src_fmt, flags, geom):
    """planar 4:4:4 destination (scale_cuda's format list, vf_scale_cuda.c:45-54): chroma planes at luma size,
    a 4:2:0 source's chroma is up-scaled 1:2 on top of the luma ratio; yuv2planeX_8_c / yuv2plane1_8_c for all
    three planes"""
    sw, sh, dw, dh = geom
    src = synth_planes(orc, src_fmt, sw, sh, seed=61)
    want = orc.sws(src, sw, sh, src_fmt, dw, dh, "yuv444p", SWS[flags])
    d = dev.upload_planes(src, 64)
    got, pads, kernel = dev.sws(d, sw, sh, src_fmt, dw, dh, "yuv444p", SWS[flags], dst_align=64)
    # exactly 2:1 from 8-bit 4:2:0 with a filter on the 8-sample window: the luma walker + a chroma re-layout (test_parity_planes2p.py
    # restates that rule and runs both paths); everything else here is the generic plane scaler
    if src_fmt in ("yuv420p", "nv12") and (sw, sh) == (2 * dw, 2 * dh) and sw % 16 == 0 and sw >= 64 and dh >= 16 and flags != "lanczos":
        assert kernel.startswith("scale_yuv2p_kernel<luma>"), kernel
    else:
        # (round 5: 4:4:4 at BOTH ends is three plane jobs of the band walker where it has an instance, tests/test_parity_walker16.py)
        assert is_generic(kernel) and ("yuv444" in kernel or kernel in (LINES, "scale19_kernel", "scale19_unit_kernel") or (src_fmt == "yuv444p" and kernel.startswith("scale_yuvg_"))), kernel
    assert len(got) == 3
    for i, (g, wv) in enumerate(zip(got, want)):
        bad = np.argwhere(g != wv)
        assert bad.size == 0, f"plane {i}: {len(bad)} mismatching bytes, first at {bad[:4].tolist()} ({kernel})"
        assert (pads[i] == 0xCD).all()
    for p in d:
        p.free()


@pytest.mark.parametrize("align", [1, 64])
def test_yuv444p_output_unaligned_and_range(dev, orc, align):
    """odd sizes with byte-aligned planes, plus the limited -> full range conversion on the way"""
    import ctypes as C
    from harness import planes, ints, alloc_planes
    L = orc.L
    L.orc_sws_create_ex.restype = C.c_void_p
    L.orc_sws_create_ex.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    sw, sh, dw, dh = 77, 35, 93, 41
    src = synth_planes(orc, "yuv420p", sw, sh, seed=62)
    for sr, dr in ((0, 0), (0, 1), (1, 0)):
        oc = L.orc_sws_create_ex(sw, sh, PIX_FMT["yuv420p"], dw, dh, PIX_FMT["yuv444p"], SWS["bicubic"], None,
                                 (C.c_int * 4)(-513, -513, -513, -513), sr, dr)
        want = alloc_planes("yuv444p", dw, dh)
        assert L.orc_sws_scale(oc, planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                               planes([p.ctypes.data for p in want]), ints([p.strides[0] for p in want])) == dh
        L.orc_sws_free(oc)
        c = dev.lib.gmat_sws_getContext(sw, sh, PIX_FMT["yuv420p"], dw, dh, PIX_FMT["yuv444p"], SWS["bicubic"], None)
        assert c and dev.lib.gmat_sws_setRange(c, sr, dr) == 0
        d = dev.upload_planes(src, align)
        dst = dev.planes_like("yuv444p", dw, dh, align)
        assert dev.lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, sh,
                                      planes([p.ptr for p in dst]), ints([p.stride for p in dst])) == dh
        for a, b in zip(dst, want):
            assert (a.download() == b).all(), (sr, dr)
        dev.lib.gmat_sws_freeContext(c)
        for p in d + dst:
            p.free()


@pytest.mark.parametrize("dst_fmt", ["nv12", "yuv420p", "yuv444p", "rgb24", "bgra"])
@pytest.mark.parametrize("src_fmt", ["p010le", "p016le"])
@pytest.mark.parametrize("geom", [(256, 64, 128, 32), (130, 50, 130, 50), (96, 40, 144, 60), (201, 91, 151, 67), (64, 34, 20, 11)])
def test_p01x_sources(dev, orc, src_fmt, dst_fmt, geom):
    """P010LE / P016LE sources (scale_cuda's list, vf_scale_cuda.c:45-54) to the 8-bit destinations, any size incl.
    equal: p010LEToY_c / p010LEToUV_c (>> 6) or the 16-bit samples as they are, hScale16To15_c with sh = depth - 1
    (swscale.c:93-119), then the same vertical / output stage as an 8-bit source.  Random 16-bit words: the low six
    bits of P010 are not zero here, the high bit of P016 is exercised."""
    sw, sh, dw, dh = geom
    src = synth_planes(orc, src_fmt, sw, sh, seed=71)
    for flags in ("bicubic", "bilinear"):
        want = orc.sws(src, sw, sh, src_fmt, dw, dh, dst_fmt, SWS[flags])
        for align, extra in ((64, 0), (2, 2)):
            d = dev.upload_planes(src, align, extra)
            got, pads, kernel = dev.sws(d, sw, sh, src_fmt, dw, dh, dst_fmt, SWS[flags], dst_align=align, dst_extra=extra)
            # the generic plane scaler, except P010 -> NV12 at exactly half size on dword-aligned rows: the plane-walking kernel's
            # 10 -> 8 instantiation (its own matrix: tests/test_parity_planes2p.py)
            from test_parity_planes2p import strip_takes, strip_name
            strip = (src_fmt == "p010le" and dst_fmt in ("nv12", "yuv420p") and (dw, dh) == (sw // 2, sh // 2) and align % 4 == 0 and
                     strip_takes(sw, sh, src_fmt, dst_fmt, flags))
            assert kernel == strip_name(src_fmt, dst_fmt) if strip else is_generic(kernel), kernel
            for i, (g, wv) in enumerate(zip(got, want)):
                bad = np.argwhere(g != wv)
                assert bad.size == 0, f"plane {i}: {len(bad)} mismatching bytes, first at {bad[:4].tolist()} ({kernel}, {flags}, align {align})"
                assert (pads[i] == 0xCD).all()
            for p in d:
                p.free()


@pytest.mark.parametrize("src_fmt", ["nv12", "yuv420p", "yuv444p", "p010le", "p016le"])
@pytest.mark.parametrize("geom", [(256, 64, 128, 32), (96, 40, 144, 60), (201, 91, 151, 67), (130, 50, 130, 50)])
def test_p010_destination(dev, orc, src_fmt, geom):
    """P010LE as a (scaled) destination: dstBpc = 10 keeps the 15-bit lines; yuv2p010l1_c / lX_c / cX_c
    (output.c:459-519): clip_uintp2((1 << 16 + sum) >> 17, 10) << 6, chroma always in the X form"""
    sw, sh, dw, dh = geom
    if src_fmt in ("nv12", "yuv420p") and (sw, sh) == (dw, dh):
        pytest.skip("equal-size 8-bit 4:2:0 -> P010 is the depth-expansion converter (test_parity_rgb2yuv.py)")
    src = synth_planes(orc, src_fmt, sw, sh, seed=73)
    for flags in ("bicubic", "point"):
        want = orc.sws(src, sw, sh, src_fmt, dw, dh, "p010le", SWS[flags])
        for align, extra in ((64, 0), (2, 2)):
            if src_fmt == "p010le" and (sw, sh) == (dw, dh):
                continue
            d = dev.upload_planes(src, align, extra)
            got, pads, kernel = dev.sws(d, sw, sh, src_fmt, dw, dh, "p010le", SWS[flags], dst_align=align, dst_extra=extra)
            # the generic plane scaler, except P010 -> P010 at exactly half size on 8-byte aligned rows: the 10-bit
            # plane-walking kernel (its own matrix: tests/test_parity_planes2p.py)
            from test_parity_planes2p import strip_takes, strip_name
            strip = (dw, dh) == (sw // 2, sh // 2) and align % 8 == 0 and strip_takes(sw, sh, src_fmt, "p010le", flags)
            assert kernel == strip_name(src_fmt, "p010le") if strip else is_generic(kernel), kernel
            for i, (g, wv) in enumerate(zip(got, want)):
                bad = np.argwhere(g != wv)
                assert bad.size == 0, f"plane {i}: {len(bad)} mismatching bytes, first at {bad[:4].tolist()} ({kernel}, {flags}, align {align})"
                assert (pads[i] == 0xCD).all()
            for p in d:
                p.free()


@pytest.mark.parametrize("form", ["tile", "passes"])
@pytest.mark.parametrize("src_fmt", ["nv12", "yuv420p", "yuv444p", "p010le", "p016le"])
@pytest.mark.parametrize("geom", [(256, 64, 128, 32), (96, 40, 144, 60), (201, 91, 151, 67), (130, 50, 130, 50)])
def test_p016_destination(dev, orc, src_fmt, geom, form, monkeypatch):
    """P016LE as a (scaled) destination: dstBpc = 16 switches to 19-bit int32 lines — hScale8To19_c / hScale16To19_c,
    yuv2plane1_16_c / yuv2planeX_16_c / yuv2nv12cX_16_c (32-bit wrap-around accumulation, output.c:143-211); in one launch with a tile's
    lines in LDS (k_scale19.hip, round 6) and as the two passes of k_scale16.hip behind GMAT_S19=0"""
    sw, sh, dw, dh = geom
    if form == "passes":
        monkeypatch.setenv("GMAT_S19", "0")
    if (sw, sh) == (dw, dh) and src_fmt in ("nv12", "yuv420p", "p016le"):
        pytest.skip("equal-size 8-bit 4:2:0 -> P016 is the depth-expansion converter, P016 -> P016 a plane copy")
    src = synth_planes(orc, src_fmt, sw, sh, seed=75)
    for flags in ("bicubic", "lanczos", "point"):
        want = orc.sws(src, sw, sh, src_fmt, dw, dh, "p016le", SWS[flags])
        for align, extra in ((64, 0), (2, 2)):
            d = dev.upload_planes(src, align, extra)
            got, pads, kernel = dev.sws(d, sw, sh, src_fmt, dw, dh, "p016le", SWS[flags], dst_align=align, dst_extra=extra)
            assert kernel == (("scale19_unit_kernel" if (sw, sh) == (dw, dh) and "444" not in src_fmt else "scale19_kernel") if form == "tile" else "hscale19_kernel+vscale16_kernel"), kernel      # (equal size, 4:2:0 at both ends: the tile kernel's unit form)
            for i, (g, wv) in enumerate(zip(got, want)):
                bad = np.argwhere(g != wv)
                assert bad.size == 0, f"plane {i}: {len(bad)} mismatching bytes, first at {bad[:4].tolist()} ({flags}, align {align})"
                assert (pads[i] == 0xCD).all()
            for p in d:
                p.free()


@pytest.mark.parametrize("form", ["tile", "passes"])
@pytest.mark.parametrize("dst_fmt", ["rgba64le", "bgra64le"])
@pytest.mark.parametrize("src_fmt", ["nv12", "yuv420p", "yuv444p", "p010le", "p016le"])
@pytest.mark.parametrize("geom", [(256, 64, 128, 32), (96, 40, 144, 60), (201, 91, 151, 67), (130, 50, 130, 50), (64, 34, 64, 17)])
def test_rgba64_destinations(dev, orc, src_fmt, dst_fmt, geom, form, monkeypatch):
    """RGBA64LE / BGRA64LE (yuv2rgb_cuda's 64-bit outputs) with libswscale's semantics: 19-bit lines, then
    yuv2rgba64_X_c / _2_c / _1_c or their full-chroma twins as packed_vscale picks them (vscale.c:135-167); the oracle
    restates the forms one by one, the HIP path runs the X form with the effective coefficients.  Geometries cover the
    1-tap luma with 1- and 2-tap chroma (equal size / equal height), full chroma (odd width, 4:4:4 source) and the X form.  In one launch behind a
    tile's lines in LDS (scale19_kernel's colour stage, round 6) and as the two passes of k_scale16.hip behind GMAT_S19=0."""
    sw, sh, dw, dh = geom
    if form == "passes":
        monkeypatch.setenv("GMAT_S19", "0")
    src = synth_planes(orc, src_fmt, sw, sh, seed=77)
    for flags in ("bicubic", "bilinear", "point"):
        want = orc.sws(src, sw, sh, src_fmt, dw, dh, dst_fmt, SWS[flags])
        for align, extra in ((64, 0), (2, 2)):
            d = dev.upload_planes(src, align, extra)
            got, pads, kernel = dev.sws(d, sw, sh, src_fmt, dw, dh, dst_fmt, SWS[flags], dst_align=align, dst_extra=extra)
            assert kernel == ("scale19_kernel" if form == "tile" else "hscale19_kernel+vrgba64_kernel"), kernel
            bad = np.argwhere(got[0] != want[0])
            assert bad.size == 0, f"{len(bad)} mismatching bytes, first at {bad[:4].tolist()} ({flags}, align {align})"
            assert (pads[0] == 0xCD).all()
            for p in d:
                p.free()
    if src_fmt == "nv12":
        want = orc.sws(src, sw, sh, src_fmt, dw, dh, dst_fmt, SWS["bicubic"], colorspace=1)
        d = dev.upload_planes(src, 64)
        got, _, _ = dev.sws(d, sw, sh, src_fmt, dw, dh, dst_fmt, SWS["bicubic"], dst_align=64, colorspace=(1, 0))
        assert (got[0] == want[0]).all()


@pytest.mark.parametrize("pair", [("yuv444p16le", "yuv444p16le"), ("yuv444p16le", "p016le"), ("yuv444p16le", "rgba64le"),
                                  ("nv12", "yuv444p16le"), ("p010le", "yuv444p16le"), ("yuv444p", "yuv444p16le")])
@pytest.mark.parametrize("geom", [(128, 48, 64, 24), (96, 40, 144, 60), (101, 45, 75, 33), (70, 22, 70, 22)])
def test_yuv444p16_on_the_19bit_path(dev, orc, pair, geom):
    """scale_cuda's last format: planar 16-bit 4:4:4 as a source and as a destination of the 19-bit path (equal format and
    size: plane copy)"""
    sf, df = pair
    sw, sh, dw, dh = geom
    src = synth_planes(orc, sf, sw, sh, seed=79)
    same = sf == df and (sw, sh) == (dw, dh)
    want = src if same else orc.sws(src, sw, sh, sf, dw, dh, df, SWS["bicubic"])
    for align, extra in ((64, 0), (2, 2)):
        d = dev.upload_planes(src, align, extra)
        got, pads, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, SWS["bicubic"], dst_align=align, dst_extra=extra)
        for i, (g, wv) in enumerate(zip(got, want)):
            assert (g == wv).all(), (i, kernel)
            assert (pads[i] == 0xCD).all()
        for p in d:
            p.free()


@pytest.mark.parametrize("dst_fmt", ["nv12", "yuv420p", "yuv444p", "rgb24", "bgra", "p010le"])
@pytest.mark.parametrize("geom", [(128, 48, 64, 24), (96, 40, 144, 60), (101, 45, 75, 33), (70, 22, 70, 22)])
def test_yuv444p16_source_to_8bit_and_p010(dev, orc, dst_fmt, geom):
    """planar 16-bit 4:4:4 towards the 15-bit-line destinations: hScale16To15_c with sh = 15 per plane (the LDS image
    biased like P016's), full chroma interpolation forced for RGB outputs (non-subsampled source)"""
    sw, sh, dw, dh = geom
    src = synth_planes(orc, "yuv444p16le", sw, sh, seed=80)
    want = orc.sws(src, sw, sh, "yuv444p16le", dw, dh, dst_fmt, SWS["bicubic"])
    for align, extra in ((64, 0), (2, 2)):
        d = dev.upload_planes(src, align, extra)
        got, pads, kernel = dev.sws(d, sw, sh, "yuv444p16le", dw, dh, dst_fmt, SWS["bicubic"], dst_align=align, dst_extra=extra)
        if (sw, sh) == (dw, dh) and dst_fmt == "yuv444p":                    # equal size and layout: planarCopyWrapper (tests/test_parity_dither.py)
            assert kernel == "plane_copy_down3_kernel"
        else:
            assert is_generic(kernel), kernel
        for i, (g, wv) in enumerate(zip(got, want)):
            assert (g == wv).all(), (i, kernel)
            assert (pads[i] == 0xCD).all()
        for p in d:
            p.free()


def _synth_hi420(orc, fmt, w, h, seed):
    src = synth_planes(orc, fmt, w, h, seed=seed)
    if fmt == "yuv420p10le":                              # valid input: 10 significant bits in the low end
        for p in src:
            p.view("<u2")[...] &= 0x3FF
    return src


@pytest.mark.parametrize("src_fmt", ["yuv420p10le", "yuv420p16le"])
@pytest.mark.parametrize("dst_fmt", ["nv12", "yuv420p", "yuv444p", "rgb24", "bgra", "p010le", "p016le", "yuv444p16le", "rgba64le"])
@pytest.mark.parametrize("geom", [(128, 48, 64, 24), (96, 40, 144, 60), (101, 45, 75, 33), (70, 22, 70, 22)])
def test_planar_high_depth_420_sources(dev, orc, src_fmt, dst_fmt, geom):
    """swscale_cuda's planar high-depth 4:2:0 sources (swscale_cuda.c:34-44): no input converter on a little-endian host,
    hScale16To15_c with sh = depth - 1 (15-bit lines) or hScale16To19_c with sh = depth - 5 (19-bit lines) per plane"""
    sw, sh, dw, dh = geom
    src = _synth_hi420(orc, src_fmt, sw, sh, 83)
    want = orc.sws(src, sw, sh, src_fmt, dw, dh, dst_fmt, SWS["bicubic"])
    for align, extra in ((64, 0), (2, 2)):
        d = dev.upload_planes(src, align, extra)
        got, pads, kernel = dev.sws(d, sw, sh, src_fmt, dw, dh, dst_fmt, SWS["bicubic"], dst_align=align, dst_extra=extra)
        for i, (g, wv) in enumerate(zip(got, want)):
            assert (g == wv).all(), (i, kernel)
            assert (pads[i] == 0xCD).all()
        for p in d:
            p.free()


@pytest.mark.parametrize("fmt", ["yuv420p10le", "yuv420p16le"])
def test_planar_high_depth_420_equal_format_and_size_is_a_plane_copy(dev, orc, fmt):
    w, h = 71, 23
    src = _synth_hi420(orc, fmt, w, h, 84)
    d = dev.upload_planes(src, 2, 2)
    got, pads, kernel = dev.sws(d, w, h, fmt, w, h, fmt, SWS["bicubic"], dst_align=2, dst_extra=2)
    assert kernel == "copy2d_kernel"
    for g, s_, pd in zip(got, src, pads):
        assert (g == s_).all() and (pd == 0xCD).all()
    for p in d:
        p.free()


@pytest.mark.parametrize("dst_fmt", ["yuv420p10le", "yuv420p16le"])
@pytest.mark.parametrize("src_fmt", ["nv12", "yuv420p", "yuv444p", "p010le", "p016le", "yuv420p10le", "yuv420p16le", "yuv444p16le"])
@pytest.mark.parametrize("geom", [(128, 48, 64, 24), (96, 40, 144, 60), (101, 45, 75, 33), (70, 22, 70, 22)])
def test_planar_high_depth_420_destinations(dev, orc, src_fmt, dst_fmt, geom):
    """YUV420P10LE on the 15-bit lines (yuv2plane1_10_c / yuv2planeX_10_c per plane: P010's arithmetic, sample in the low
    bits), YUV420P16LE on the 19-bit lines (yuv2planeX_16_c per plane).  Equal format and size is the plane copy; equal size
    from 8-bit planar is planarCopyWrapper's shift in libswscale, which the generic lines reproduce for limited range — the
    reference's filter-pixfmts-null rows of both formats pin that (tests/test_fate_product_nut.py)"""
    sw, sh, dw, dh = geom
    src = _synth_hi420(orc, src_fmt, sw, sh, 85)
    same = src_fmt == dst_fmt and (sw, sh) == (dw, dh)
    want = src if same else orc.sws(src, sw, sh, src_fmt, dw, dh, dst_fmt, SWS["bicubic"])
    for align, extra in ((64, 0), (8, 0), (2, 2)):
        d = dev.upload_planes(src, align, extra)
        got, pads, kernel = dev.sws(d, sw, sh, src_fmt, dw, dh, dst_fmt, SWS["bicubic"], dst_align=align, dst_extra=extra)
        for i, (g, wv) in enumerate(zip(got, want)):
            assert (g == wv).all(), (i, kernel, align)
            assert (pads[i] == 0xCD).all()
        for p in d:
            p.free()


@pytest.mark.parametrize("flags", ["bilinear", "lanczos", "point", "area"])
def test_planar_high_depth_420_destinations_algorithms(dev, orc, flags):
    for sf, df, geom in (("nv12", "yuv420p10le", (192, 70, 96, 36)), ("yuv420p10le", "yuv420p10le", (80, 30, 120, 50)),
                         ("yuv420p", "yuv420p16le", (192, 70, 96, 36)), ("yuv420p16le", "yuv420p16le", (80, 30, 120, 50))):
        sw, sh, dw, dh = geom
        src = _synth_hi420(orc, sf, sw, sh, 86)
        want = orc.sws(src, sw, sh, sf, dw, dh, df, SWS[flags])
        d = dev.upload_planes(src, 64)
        got, _, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, SWS[flags], dst_align=64)
        for g, wv in zip(got, want):
            assert (g == wv).all(), (sf, df, flags, kernel)
        for p in d:
            p.free()


@pytest.mark.parametrize("pair", [("yuv420p", "yuv420p10le"), ("yuv420p", "yuv420p16le"), ("yuv444p", "yuv444p16le")])
def test_full_range_planar_depth_expansion(dev, orc, pair):
    """same size, 8-bit planar -> high-depth planar with both ends full range: planarCopyWrapper (swscale_unscaled.c:1803-1862)
    bit-replicates the luma (v << (d - 8) | v >> (16 - d)) and shifts the chroma; with both ends limited it shifts everything —
    which the generic lines give too (pinned by the reference's filter-pixfmts md5s); with differing ranges libswscale leaves
    the wrapper for the generic path (utils.c:1996-2000) and so does the context"""
    import ctypes as C
    from harness import alloc_planes, planes, ints
    sf, df = pair
    lib, w, h = dev.lib, 70, 34
    depth = 10 if df.endswith("10le") else 16
    orc.L.orc_sws_create_ex.restype = C.c_void_p
    orc.L.orc_sws_create_ex.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    src = synth_planes(orc, sf, w, h, seed=77)
    src[0][0, :16] = [0, 1, 15, 16, 17, 127, 128, 129, 200, 234, 235, 236, 253, 254, 255, 255]
    want = alloc_planes(df, w, h)
    for i, (s_, d_) in enumerate(zip(src, want)):
        orc.L.orc_plane_copy_up(s_.ctypes.data, s_.strides[0], d_.ctypes.data, d_.strides[0], s_.shape[1], s_.shape[0], depth, int(i > 0))
    shifted = [(s_.astype(np.uint16) << (depth - 8)).view(np.uint8).reshape(s_.shape[0], -1) for s_ in src]
    assert not (want[0] == shifted[0]).all() and all((a == b).all() for a, b in zip(want[1:], shifted[1:]))
    d = dev.upload_planes(src, 64)
    c = lib.gmat_sws_getContext(w, h, PIX_FMT[sf], w, h, PIX_FMT[df], SWS["bicubic"], None)
    assert c

    def run():
        dst = dev.planes_like(df, w, h, 64)
        assert lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, h,
                                  planes([p.ptr for p in dst]), ints([p.stride for p in dst])) == h
        out = [p.download() for p in dst]
        assert all((p.download(with_padding=True)[:, p.row_bytes:] == 0xCD).all() for p in dst)
        for p in dst:
            p.free()
        return out, lib.gmat_sws_lastKernel(c).decode()

    got, k0 = run()                                               # limited -> limited: the shift
    assert all((g == wv).all() for g, wv in zip(got, shifted)) and k0 != "plane_copy_up_kernel"
    assert lib.gmat_sws_setRange(c, 1, 1) == 0
    got, k = run()
    assert k == "plane_copy_up_kernel" and all((g == wv).all() for g, wv in zip(got, want))
    if True:                                                      # differing ranges: the generic lines carry the conversion (15-bit
        assert lib.gmat_sws_setRange(c, 1, 0) == 0                # lines for the 10-bit destination, 19-bit lines for the 16-bit ones)
        got, k = run()
        oc = orc.L.orc_sws_create_ex(w, h, PIX_FMT[sf], w, h, PIX_FMT[df], SWS["bicubic"], None, (C.c_int * 4)(-513, -513, -513, -513), 1, 0)
        assert oc
        ow = alloc_planes(df, w, h)
        assert orc.L.orc_sws_scale(oc, planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                                   planes([p.ctypes.data for p in ow]), ints([p.strides[0] for p in ow])) == h
        orc.L.orc_sws_free(oc)
        assert k == k0 and all((g == wv).all() for g, wv in zip(got, ow))
    assert lib.gmat_sws_setRange(c, 0, 0) == 0
    got, k = run()
    assert k == k0 and all((g == wv).all() for g, wv in zip(got, shifted))
    lib.gmat_sws_freeContext(c)
    for p in d:
        p.free()
    c = lib.gmat_sws_getContext(64, 32, PIX_FMT[sf], 32, 16, PIX_FMT[df], SWS["bicubic"], None)     # scaled: the generic path in libswscale too
    assert c and lib.gmat_sws_setRange(c, 1, 1) == 0 and lib.gmat_sws_lastKernel(c).decode() != "plane_copy_up_kernel"
    lib.gmat_sws_freeContext(c)


@pytest.mark.parametrize("fmt", ["p010le", "p016le"])
def test_p01x_equal_format_and_size_is_a_plane_copy(dev, orc, fmt):
    w, h = 70, 22
    src = synth_planes(orc, fmt, w, h, seed=74)
    d = dev.upload_planes(src, 2, 2)
    got, pads, kernel = dev.sws(d, w, h, fmt, w, h, fmt, dst_align=64)
    assert all((g == s).all() for g, s in zip(got, src)) and all((p == 0xCD).all() for p in pads)


def test_p01x_source_extremes_and_errors(dev, orc):
    """all-ones and all-zero samples (the P016 image bias at both ends), Lanczos, range conversion; byte-aligned rows of
    16-bit samples are refused"""
    import ctypes as C
    from harness import planes, ints
    sw, sh, dw, dh = 64, 16, 48, 12
    for fmt in ("p010le", "p016le"):
        for val in (0x00, 0xFF):
            src = [np.full(s, val, np.uint8) for s in [(sh, 2 * sw), (sh // 2, 2 * sw)]]
            want = orc.sws(src, sw, sh, fmt, dw, dh, "nv12", SWS["lanczos"])
            d = dev.upload_planes(src, 64)
            got, _, _ = dev.sws(d, sw, sh, fmt, dw, dh, "nv12", SWS["lanczos"], dst_align=64)
            assert all((g == w).all() for g, w in zip(got, want)), (fmt, val)
            for p in d:
                p.free()
    c = dev.lib.gmat_sws_getContext(sw, sh, PIX_FMT["p010le"], dw, dh, PIX_FMT["nv12"], SWS["bicubic"], None)
    assert c
    s = dev.planes_like("p010le", sw, sh, 1, 1)            # odd strides
    d = dev.planes_like("nv12", dw, dh, 64)
    assert dev.lib.gmat_sws_scale(c, planes([p.ptr for p in s]), ints([p.stride for p in s]), 0, sh,
                                  planes([p.ptr for p in d]), ints([p.stride for p in d])) < 0
    assert dev.lib.gmat_sws_setFused(c, 1) < 0            # no convert-then-scale form for 16-bit sources
    dev.lib.gmat_sws_freeContext(c)
    for p in s + d:
        p.free()


@pytest.mark.parametrize("src_fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("dst_fmt", ["rgb24", "bgra", "nv12", "yuv420p"])
@pytest.mark.parametrize("flags", ["bicubic", "bilinear", "lanczos", "point"])
def test_2to1_kernel_interior_tiles(dev, orc, monkeypatch, src_fmt, dst_fmt, flags):
    """5 x 5 tiles: the 3 x 3 in the middle touch no border and take the uniform-coefficient path (coefficients as
    kernel arguments, closed-form window rows); the ring around them takes the table path; both must agree with the
    oracle, and GMAT_SCALE_NO_UNIFORM=1 (all tiles on the table path) must give the same bytes"""
    monkeypatch.setenv("GMAT_SCALE_NO_STRIP", "1")              # this test is about the TILED kernel's two paths
    sw, sh, dw, dh = 640, 160, 320, 80
    src = synth_planes(orc, src_fmt, sw, sh, seed=81)
    want = orc.sws(src, sw, sh, src_fmt, dw, dh, dst_fmt, SWS[flags])
    d = dev.upload_planes(src, 64)
    got, pads, kernel = dev.sws(d, sw, sh, src_fmt, dw, dh, dst_fmt, SWS[flags], dst_align=64)
    assert kernel.startswith("scale_yuv2x_kernel"), kernel
    for i, (g, wv) in enumerate(zip(got, want)):
        bad = np.argwhere(g != wv)
        assert bad.size == 0, f"plane {i}: {len(bad)} mismatching bytes, first at {bad[:4].tolist()}"
        assert (pads[i] == 0xCD).all()
    for p in d:
        p.free()


@pytest.mark.parametrize("src_fmt", ["rgb24", "bgr24"])
@pytest.mark.parametrize("dst_fmt", ["nv12", "yuv420p", "yuv444p", "p010le"])
@pytest.mark.parametrize("geom", [(256, 64, 128, 32), (96, 40, 144, 60), (201, 91, 151, 67), (130, 51, 63, 25), (64, 34, 100, 34)])
def test_rgb_scaled_into_yuv(dev, orc, src_fmt, dst_fmt, geom):
    """packed RGB scaled into a YUV frame by ONE context, as libswscale does: rgb24ToY_c, rgb24ToUV_c or ToUV_half_c
    (chrSrcHSubSample = 1 when the destination's chroma is at most half the source width, utils.c:1529-1545),
    hScale16To15_c with sh = 13, planar vertical stage; the plane scaler's RGB loader"""
    sw, sh, dw, dh = geom
    src = synth_planes(orc, src_fmt, sw, sh, seed=87)
    for flags in ("bicubic", "bilinear"):
        want = orc.sws(src, sw, sh, src_fmt, dw, dh, dst_fmt, SWS[flags])
        for align, extra in ((64, 0), (2, 2)):
            d = dev.upload_planes(src, align, extra)
            got, pads, kernel = dev.sws(d, sw, sh, src_fmt, dw, dh, dst_fmt, SWS[flags], dst_align=align, dst_extra=extra)
            # exactly 2:1 into 8-bit 4:2:0 on dword-aligned planes is scale_rgb2y_kernel's (tests/test_parity_rgb2y.py)
            strip = (sw, sh) == (2 * dw, 2 * dh) and dst_fmt in ("nv12", "yuv420p") and align % 4 == 0 and dw % 4 == 0 and dw >= 64 and dh >= 16
            assert kernel == "scale_rgb2y_kernel" if strip else is_generic(kernel), kernel
            for i, (g, wv) in enumerate(zip(got, want)):
                bad = np.argwhere(g != wv)
                assert bad.size == 0, f"plane {i}: {len(bad)} mismatching bytes, first at {bad[:4].tolist()} ({flags}, align {align})"
                assert (pads[i] == 0xCD).all()
            for p in d:
                p.free()
    if dst_fmt == "nv12":
        want = orc.sws(src, sw, sh, src_fmt, dw, dh, dst_fmt, SWS["bicubic"], colorspace=1)
        d = dev.upload_planes(src, 64)
        got, _, _ = dev.sws(d, sw, sh, src_fmt, dw, dh, dst_fmt, SWS["bicubic"], dst_align=64, colorspace=(1, 0))
        assert all((g == wv).all() for g, wv in zip(got, want))


@pytest.mark.parametrize("src_fmt", ["rgba", "bgra"])
@pytest.mark.parametrize("case", [("rgb24", 96, 40, 50, 30), ("bgra", 96, 40, 144, 60), ("nv12", 128, 32, 128, 32), ("nv12", 128, 32, 64, 16),
                                  ("yuv420p", 130, 34, 130, 34), ("yuv444p", 70, 22, 70, 22), ("rgba", 131, 35, 64, 17)])
def test_32bit_rgb_sources(dev, orc, src_fmt, case):
    """RGBA / BGRA sources of the scaling and RGB -> YUV paths (swscale_cuda.c:34-44): rgb32ToY / ToUV read the same
    three channels with the same coefficients as the 24-bit readers, so the colour channels must equal the 24-bit
    context's on the same pixels; the source's alpha is dropped unless the destination has an alpha channel too, in
    which case it is scaled as a plane of its own (needAlpha, utils.c:1902 — tests/test_parity_rgb64_src.py)"""
    df, sw, sh, dw, dh = case
    src24 = synth_planes(orc, "rgb24" if src_fmt == "rgba" else "bgr24", sw, sh, seed=85)
    alpha = orc.lcg((sh, sw), 86)
    src32 = [np.ascontiguousarray(np.concatenate([src24[0].reshape(sh, sw, 3), alpha.reshape(sh, sw, 1)], axis=2).reshape(sh, 4 * sw))]
    want24 = orc.sws(src24, sw, sh, "rgb24" if src_fmt == "rgba" else "bgr24", dw, dh, df, SWS["bicubic"])
    want = orc.sws(src32, sw, sh, src_fmt, dw, dh, df, SWS["bicubic"])
    if df in ("rgba", "bgra"):
        assert (want[0].reshape(dh, dw, 4)[:, :, :3] == want24[0].reshape(dh, dw, 4)[:, :, :3]).all()
        assert (want24[0].reshape(dh, dw, 4)[:, :, 3] == 255).all() and not (want[0].reshape(dh, dw, 4)[:, :, 3] == 255).all()
    else:
        assert all((a == b).all() for a, b in zip(want, want24))
    for align in (64, 1):
        d = dev.upload_planes(src32, align)
        got, pads, kernel = dev.sws(d, sw, sh, src_fmt, dw, dh, df, SWS["bicubic"], dst_align=align)
        for i, (g, wv) in enumerate(zip(got, want)):
            assert (g == wv).all(), (i, kernel)
            assert (pads[i] == 0xCD).all()
        for p in d:
            p.free()


@pytest.mark.parametrize("cs", [1, 7, 9])
@pytest.mark.parametrize("case", [("nv12", "rgb24", 256, 64, 128, 32, 0), ("yuv420p", "bgra", 96, 40, 144, 60, 0),
                                  ("nv12", "rgb24", 130, 50, 63, 25, 0), ("yuv444p", "rgb24", 64, 32, 48, 20, 0),
                                  ("nv12", "bgr24", 640, 160, 320, 80, 0)])
def test_scaled_yuv_to_rgb_colorspaces(dev, orc, cs, case):
    """the source's matrix in the scaled (one-context) YUV -> RGB paths: LUT form, full-chroma form (odd width / 4:4:4
    source) and the 2:1 kernel incl. its interior tiles"""
    sf, df, sw, sh, dw, dh, _ = case
    src = synth_planes(orc, sf, sw, sh, seed=83)
    want = orc.sws(src, sw, sh, sf, dw, dh, df, SWS["bicubic"], colorspace=cs)
    d = dev.upload_planes(src, 64)
    got, _, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, SWS["bicubic"], dst_align=64, colorspace=(cs, 0))
    assert (got[0] == want[0]).all(), kernel
    for p in d:
        p.free()


@pytest.mark.parametrize("case", [("bgr24", "bgra", (86, 118, 66, 81), SWS["point"] | SWS["accurate_rnd"]),
                                  ("bgr24", "bgr24", (6, 97, 196, 67), SWS["point"] | SWS["full_chr_h_int"]),
                                  ("rgb24", "bgra", (158, 36, 257, 49), SWS["area"])])
def test_regressions_found_by_the_fuzzer(dev, orc, case):
    """tests/fuzz/fuzz_parity.py finds: a tile whose source window is a single 4-pixel group (ng == 1) divided by a
    32-bit magic that does not exist."""
    sf, df, (sw, sh, dw, dh), flags = case
    _check(dev, orc, sf, sw, sh, dw, dh, df, flags, align=1, extra=1)


@pytest.mark.parametrize("dst_fmt", ["rgb24", "nv12"])
@pytest.mark.parametrize("geom", [(207, 57, 54, 51), (512, 96, 64, 12), (640, 64, 80, 40), (300, 200, 40, 25)])
def test_yuv_single_context_large_downscale_ratios(dev, orc, dst_fmt, geom):
    """ratios beyond ~3.7:1 need more than 16 horizontal taps: the generic plane scaler handles them in its tail
    loops instead of falling back to convert-then-scale semantics"""
    sw, sh, dw, dh = geom
    src = synth_planes(orc, "nv12", sw, sh, seed=51)
    want = orc.sws(src, sw, sh, "nv12", dw, dh, dst_fmt, SWS["bicubic"])
    d = dev.upload_planes(src, 64)
    got, pads, kernel = dev.sws(d, sw, sh, "nv12", dw, dh, dst_fmt, SWS["bicubic"], dst_align=64)
    assert is_generic(kernel), kernel
    for g, wv, pd in zip(got, want, pads):
        assert (g == wv).all() and (pd == 0xCD).all()


@pytest.mark.parametrize("geom", [(207, 57, 54, 51), (512, 96, 64, 12), (300, 200, 40, 25)])
def test_rgb_large_downscale_ratios(dev, orc, geom):
    sw, sh, dw, dh = geom
    _check(dev, orc, "rgb24", sw, sh, dw, dh, "rgb24", SWS["bicubic"])
    _check(dev, orc, "nv12", sw, sh, dw, dh, "bgra", SWS["bicubic"], fused=1)


@pytest.mark.parametrize("src_fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("geom", X2_GEOMS)
def test_yuv2x_lanczos_uses_the_14_sample_window(dev, orc, kern, src_fmt, geom):
    """Lanczos-3 at 2:1 has 12 taps: the tiled 2:1 kernel's P = 7 variant (window of 14 samples, 4 vertical chroma pairs) — and,
    by default on frames at least 128 wide and 24 tall, the strip kernel's 6-pair form (tests/test_parity_strip.py): both via
    the `kern` fixture"""
    sw, sh, dw, dh = geom
    src = synth_planes(orc, src_fmt, sw, sh, seed=53)
    for dst_fmt in ("rgb24", "bgra"):
        want = orc.sws(src, sw, sh, src_fmt, dw, dh, dst_fmt, SWS["lanczos"])[0]
        d_src = dev.upload_planes(src, 256)
        got, pads, kernel = dev.sws(d_src, sw, sh, src_fmt, dw, dh, dst_fmt, SWS["lanczos"], dst_align=256)
        for p in d_src:
            p.free()
        strip = kern.startswith("scale_yuv2s") and sw >= 128 and dh >= 12
        assert kernel == ("scale_yuv2s_np_kernel<6>" if strip else "scale_yuv2x_kernel"), kernel
        bad = np.argwhere(got[0] != want)
        assert bad.size == 0, f"{len(bad)} mismatching bytes, first at {bad[:4].tolist()}"
        assert (pads[0] == 0xCD).all()


def test_one_tap_vertical_forms_ignore_the_coefficient(dev, orc):
    """yuv2packed1 / yuv2plane1 take the line as it is; for degenerate geometries (3 source rows, a shifted vertical
    chroma position) initFilter emits 1-tap rows whose coefficient is 0, not 4096 (fuzzer finding)."""
    import ctypes as C
    from harness import planes, ints, alloc_planes
    L = orc.L
    L.orc_sws_create_ex.restype = C.c_void_p
    L.orc_sws_create_ex.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    sw, sh, dw, dh, pos = 26, 3, 100, 27, [64, 256, 0, 128]
    src = synth_planes(orc, "yuv444p", sw, sh, seed=2130)
    for dst_fmt in ("bgra", "rgb24", "yuv420p", "nv12"):
        oc = L.orc_sws_create_ex(sw, sh, PIX_FMT["yuv444p"], dw, dh, PIX_FMT[dst_fmt], SWS["bilinear"], None,
                                 (C.c_int * 4)(*pos), 0, 0)
        want = alloc_planes(dst_fmt, dw, dh)
        assert L.orc_sws_scale(oc, planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                               planes([p.ctypes.data for p in want]), ints([p.strides[0] for p in want])) == dh
        L.orc_sws_free(oc)
        c = dev.lib.gmat_sws_getContext(sw, sh, PIX_FMT["yuv444p"], dw, dh, PIX_FMT[dst_fmt], SWS["bilinear"], None)
        assert c and dev.lib.gmat_sws_setChromaPos(c, *pos) == 0
        d = dev.upload_planes(src, 64)
        dst = dev.planes_like(dst_fmt, dw, dh, 64)
        assert dev.lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, sh,
                                      planes([p.ptr for p in dst]), ints([p.stride for p in dst])) == dh
        for a, b in zip(dst, want):
            assert (a.download() == b).all(), dst_fmt
        dev.lib.gmat_sws_freeContext(c)
        for p in d + dst:
            p.free()


@pytest.mark.parametrize("case", [("nv12", "p010le"), ("yuv420p", "p010le"), ("p010le", "p010le"), ("nv12", "nv12")])
@pytest.mark.parametrize("ranges", [(0, 1), (1, 0)])
def test_same_size_special_converters_leave_for_the_generic_path_when_ranges_differ(dev, orc, case, ranges):
    """utils.c:1996-2000: the unscaled special converters (plane copy, planar8ToP01xleWrapper) are skipped when
    srcRange != dstRange; the context then runs the generic path with lum/chrRangeTo/FromJpeg_c on its 15-bit lines.
    Setting equal ranges again returns to the special converter."""
    import ctypes as C
    from harness import alloc_planes, planes, ints
    sf, df = case
    w, h = 96, 40
    L = orc.L
    L.orc_sws_create_ex.restype = C.c_void_p
    L.orc_sws_create_ex.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    src = synth_planes(orc, sf, w, h, seed=91)
    if sf == "p010le":
        for p in src:                                           # 10 significant bits in the high end of each 16-bit sample
            v = p.view(np.uint16); v &= 0xFFC0
    lib = dev.lib
    d = dev.upload_planes(src, 64)
    c = lib.gmat_sws_getContext(w, h, PIX_FMT[sf], w, h, PIX_FMT[df], SWS["bicubic"], None)
    assert c
    for sr, dr in (ranges, (0, 0), ranges, (1, 1)):
        want = alloc_planes(df, w, h)
        if sr != dr:
            oc = L.orc_sws_create_ex(w, h, PIX_FMT[sf], w, h, PIX_FMT[df], SWS["bicubic"], None, (C.c_int * 4)(-513, -513, -513, -513), sr, dr)
            assert oc
            assert L.orc_sws_scale(oc, planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                                   planes([p.ctypes.data for p in want]), ints([p.strides[0] for p in want])) == h
            L.orc_sws_free(oc)
        elif sf == df:                                          # equal format, size and range: the planes verbatim
            for a, b in zip(want, src):
                a[...] = b
        elif sf == "nv12":                                      # no special converter for a semi-planar source (:2108-2112): always generic
            want = orc.sws(src, w, h, sf, w, h, df)
        else:                                                   # planar8ToP01xleWrapper (swscale_unscaled.c:286-324)
            L.orc_yuv420_to_p01x(planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                                 planes([p.ctypes.data for p in want]), ints([p.strides[0] for p in want]), w, h, 0)
        assert lib.gmat_sws_setRange(c, sr, dr) == 0
        dst = dev.planes_like(df, w, h, 64)
        assert lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, h,
                                  planes([p.ptr for p in dst]), ints([p.stride for p in dst])) == h
        k = lib.gmat_sws_lastKernel(c).decode()
        assert (k.startswith("scale_yuv") or k in ("scale19_kernel", "scale19_unit_kernel")) == (sr != dr), (k, sr, dr)
        if (sf, df) == ("nv12", "p010le") and sr == dr:        # round 4: the generic lines' t << 8 as the copy it is (k_rgb2yuv.hip nv12_shift8_kernel)
            assert k == "nv12_shift8_kernel", k
        for a, b in zip(dst, want):
            assert (a.download() == b).all(), (sr, dr, k)
        for p in dst:
            p.free()
    lib.gmat_sws_freeContext(c)
    for p in d:
        p.free()


@pytest.mark.parametrize("ranges", [(0, 1), (1, 0)])
@pytest.mark.parametrize("pair", [("nv12", "p016le"), ("p016le", "p016le"), ("yuv444p16le", "yuv444p16le"), ("nv12", "yuv444p16le"),
                                  ("yuv420p", "yuv420p16le"), ("p010le", "p016le"), ("rgba64le", "yuv444p16le")])
def test_range_conversion_on_the_19bit_lines(dev, orc, pair, ranges):
    """16-bit YUV destinations with differing ranges: lum / chrRange{To,From}Jpeg16_c on the 19-bit lines (swscale.c:189-226; the
    chroma ToJpeg product wraps past 2^31 on its way) — refused through round 2; at equal size the special converters step aside
    (utils.c:1996-2000); an RGB destination has no range of its own (swscale.c:536)"""
    import ctypes as C
    from harness import alloc_planes, planes, ints
    sf, df = pair
    if sf == "rgba64le" and ranges[0]:
        pytest.skip("an RGB source has no range")
    L, lib = orc.L, dev.lib
    L.orc_sws_create_ex.restype = C.c_void_p
    L.orc_sws_create_ex.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    for (sw, sh, dw, dh) in [(64, 32, 64, 32), (96, 40, 50, 30), (64, 24, 96, 36)]:
        src = synth_planes(orc, sf, sw, sh, seed=44)
        if sf == "p010le":
            for p in src:
                v = p.view(np.uint16); v &= 0xFFC0
        src[0][0, :8] = [0, 0, 255, 255, 16, 0, 235, 255]            # extremes of either range, whatever the sample width
        oc = L.orc_sws_create_ex(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], SWS["bicubic"], None, (C.c_int * 4)(-513, -513, -513, -513), ranges[0], ranges[1])
        assert oc
        want = alloc_planes(df, dw, dh)
        assert L.orc_sws_scale(oc, planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                               planes([p.ctypes.data for p in want]), ints([p.strides[0] for p in want])) == dh
        L.orc_sws_free(oc)
        c = lib.gmat_sws_getContext(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], SWS["bicubic"], None)
        assert c and lib.gmat_sws_setRange(c, ranges[0], ranges[1]) == 0
        d = dev.upload_planes(src, 64)
        dst = dev.planes_like(df, dw, dh, 64)
        assert lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, sh,
                                  planes([p.ptr for p in dst]), ints([p.stride for p in dst])) == dh
        got = [p.download() for p in dst]
        for i, (g, wv) in enumerate(zip(got, want)):
            assert (g == wv).all(), (pair, ranges, (sw, sh, dw, dh), i, np.argwhere(g != wv)[:3].tolist())
        assert lib.gmat_sws_setRange(c, 0, 0) == 0
        lib.gmat_sws_freeContext(c)
        for p in d + dst:
            p.free()
    c = lib.gmat_sws_getContext(64, 32, PIX_FMT["nv12"], 32, 16, PIX_FMT["rgba64le"], SWS["bicubic"], None)
    assert c and lib.gmat_sws_setRange(c, 0, 1) < 0
    lib.gmat_sws_freeContext(c)


@pytest.mark.parametrize("pair", [("nv12", "p016le"), ("yuv420p", "yuv444p16le"), ("p016le", "yuv420p16le"), ("yuv444p16le", "p016le")])
def test_chroma_positions_on_the_19bit_path(dev, orc, pair):
    """src_h / src_v / dst_h / dst_v_chr_pos (options.c:67-70) on contexts with a 16-bit destination: the 19-bit path's own filter banks"""
    import ctypes as C
    from harness import alloc_planes, planes, ints
    sf, df = pair
    L, lib = orc.L, dev.lib
    L.orc_sws_create_ex.restype = C.c_void_p
    L.orc_sws_create_ex.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    for pos in [(0, 128, 0, 128), (128, 0, 256, 64), (-513, 128, -513, -513), (37, -200, 511, 3)]:
        for (sw, sh, dw, dh) in [(96, 40, 50, 30), (64, 24, 96, 36)]:
            src = synth_planes(orc, sf, sw, sh, seed=52)
            oc = L.orc_sws_create_ex(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], SWS["bicubic"], None, (C.c_int * 4)(*pos), 0, 0)
            assert oc
            want = alloc_planes(df, dw, dh)
            assert L.orc_sws_scale(oc, planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                                   planes([p.ctypes.data for p in want]), ints([p.strides[0] for p in want])) == dh
            L.orc_sws_free(oc)
            c = lib.gmat_sws_getContext(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], SWS["bicubic"], None)
            assert c and lib.gmat_sws_setChromaPos(c, *pos) == 0
            d = dev.upload_planes(src, 64)
            dst = dev.planes_like(df, dw, dh, 64)
            assert lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, sh,
                                      planes([p.ptr for p in dst]), ints([p.stride for p in dst])) == dh
            assert all((p.download() == wv).all() for p, wv in zip(dst, want)), (pair, pos)
            lib.gmat_sws_freeContext(c)
            for p in d + dst:
                p.free()


@pytest.mark.parametrize("df", ["p010le", "p016le"])
@pytest.mark.parametrize("w,h", [(96, 40), (64, 8), (130, 34), (37, 9), (520, 16), (16, 2)])
@pytest.mark.parametrize("align", [64, 16, 2])
def test_nv12_to_p01x_at_equal_size_is_a_shift(dev, orc, df, w, h, align):
    """NV12 -> P010LE / P016LE at equal size: libswscale's generic lines with one-tap filters (no special converter for a semi-planar 8-bit
    source, swscale_unscaled.c:2108-2112) give t << 8 for every sample of both planes; round 4 runs that as a streaming copy
    (nv12_shift8_kernel) instead of the tiled plane scaler (35 -> see profiles/r04_relayouts.txt).  Against the oracle's generic lines, over
    widths on and off the 8-byte groups, odd sizes, pitches the vector path takes and declines; GMAT_NO_SHIFT8=1 keeps the plane scaler."""
    src = synth_planes(orc, "nv12", w, h, seed=17)
    want = orc.sws(src, w, h, "nv12", w, h, df)
    for p, q in zip(want, src):
        assert (p.view(np.uint16) == (q.astype(np.uint16) << 8)).all()              # the statement itself, on the oracle
    d = dev.upload_planes(src, 8 if align != 2 else 1)
    got, pads, k = dev.sws(d, w, h, "nv12", w, h, df, dst_align=align)
    assert k == "nv12_shift8_kernel", k
    for g, wv, pd in zip(got, want, pads):
        assert (g == wv).all()
        assert (pd == 0xCD).all()
    for p in d:
        p.free()


def test_nv12_to_p01x_knob_keeps_the_plane_scaler(dev, orc, monkeypatch):
    monkeypatch.setenv("GMAT_NO_SHIFT8", "1")
    src = synth_planes(orc, "nv12", 96, 40, seed=18)
    d = dev.upload_planes(src, 64)
    for df in ("p010le", "p016le"):
        got, _, k = dev.sws(d, 96, 40, "nv12", 96, 40, df, dst_align=64)
        assert k != "nv12_shift8_kernel"
        for g, wv in zip(got, orc.sws(src, 96, 40, "nv12", 96, 40, df)):
            assert (g == wv).all()
    for p in d:
        p.free()


@pytest.mark.parametrize("df", ["p010le", "p016le"])
@pytest.mark.parametrize("pos", [(0, 0, 256, 256), (0, 128, 0, 128), (128, 128, 128, 128), (-513, 0, -513, 256)])
def test_nv12_to_p01x_with_chroma_positions_that_differ_is_not_a_shift(dev, orc, df, pos):
    """gmat_sws_setChromaPos succeeds on these contexts, and positions that differ between the ends make the chroma banks real filters at
    equal size — the t << 8 shortcut must step aside (it is taken only while all four banks are one-tap identities); one frame and a batch of three"""
    import ctypes as C
    from harness import PIX_FMT, planes, ints, alloc_planes
    lib, L = dev.lib, orc.L
    L.orc_sws_create_ex.restype = C.c_void_p
    L.orc_sws_create_ex.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    w, h, nf = 96, 40, 3
    srcs = [synth_planes(orc, "nv12", w, h, seed=19 + f) for f in range(nf)]
    wants = []
    oc = L.orc_sws_create_ex(w, h, PIX_FMT["nv12"], w, h, PIX_FMT[df], SWS["bicubic"], None, (C.c_int * 4)(*pos), 0, 0)
    assert oc
    for src in srcs:
        want = alloc_planes(df, w, h)
        assert L.orc_sws_scale(oc, planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                               planes([p.ctypes.data for p in want]), ints([p.strides[0] for p in want])) == h
        wants.append(want)
    L.orc_sws_free(oc)
    same = pos[0] == pos[2] and pos[1] == pos[3]
    shifted = (wants[0][1].view(np.uint16) == (srcs[0][1].astype(np.uint16) << 8)).all()
    assert shifted == same                                           # the oracle: a pure shift exactly when the positions agree
    c = lib.gmat_sws_getContext(w, h, PIX_FMT["nv12"], w, h, PIX_FMT[df], SWS["bicubic"], None)
    assert c
    assert lib.gmat_sws_setChromaPos(c, *pos) == 0
    dsrc = [dev.upload_planes(s_, 64) for s_ in srcs]
    ddst = [dev.planes_like(df, w, h, 64) for _ in srcs]
    assert lib.gmat_sws_scale(c, planes([p.ptr for p in dsrc[0]]), ints([p.stride for p in dsrc[0]]), 0, h,
                              planes([p.ptr for p in ddst[0]]), ints([p.stride for p in ddst[0]])) == h
    assert (lib.gmat_sws_lastKernel(c).decode() == "nv12_shift8_kernel") == same
    for a, b in zip(ddst[0], wants[0]):
        assert (a.download() == b).all(), lib.gmat_sws_lastKernel(c).decode()
        lib.gmat_memset(a.ptr, 0xCD, a.stride * a.rows)
    sp, dp = (C.c_void_p * (4 * nf))(), (C.c_void_p * (4 * nf))()
    for f in range(nf):
        for i, p in enumerate(dsrc[f]):
            sp[4 * f + i] = p.ptr
        for i, p in enumerate(ddst[f]):
            dp[4 * f + i] = p.ptr
    assert lib.gmat_sws_scale_batch(c, nf, C.cast(sp, C.POINTER(C.c_void_p)), ints([p.stride for p in dsrc[0]]), C.cast(dp, C.POINTER(C.c_void_p)),
                                    ints([p.stride for p in ddst[0]]), C.cast((C.c_void_p * 1)(None), C.POINTER(C.c_void_p)), 1, 3) == nf
    lib.gmat_device_sync()
    assert (lib.gmat_sws_lastKernel(c).decode() == "nv12_shift8_kernel") == same
    for f in range(nf):
        for a, b in zip(ddst[f], wants[f]):
            assert (a.download() == b).all(), (f, lib.gmat_sws_lastKernel(c).decode())
    for f in dsrc + ddst:
        for p in f:
            p.free()
    lib.gmat_sws_freeContext(c)


@pytest.mark.parametrize("pair", [("nv12", "yuv420p"), ("yuv420p", "nv12")])
@pytest.mark.parametrize("w,h", [(96, 40), (64, 8), (130, 34), (37, 9), (4128, 6), (32, 2)])
@pytest.mark.parametrize("align", [64, 16, 4])
def test_yuv420_relayout_in_one_launch(dev, orc, pair, w, h, align, monkeypatch):
    """NV12 <-> YUV420P (nv12ToPlanarWrapper / planarToNv12Wrapper, swscale_unscaled.c): lossless; round 4 moves the luma plane and (de)interleaves
    the chroma in ONE launch (yuv420_relayout_kernel: 16 bytes a lane, streaming both ways, grid.z = frame) where every plane moves 16 / 8 bytes at a
    time, the 2-D copy + chroma kernel of rounds 1-3 elsewhere — widths on and off the 16-byte groups, odd sizes, a batch of three frames"""
    from test_batch_api import _run_batch
    sf, df = pair
    src = synth_planes(orc, sf, w, h, seed=23)
    want = orc.sws(src, w, h, sf, w, h, df)
    d = dev.upload_planes(src, align)
    got, pads, k = dev.sws(d, w, h, sf, w, h, df, dst_align=align)
    up = lambda v: (v + align - 1) // align * align                  # the pitches the harness gives the planes (their bases are 256-byte aligned)
    cw = (w + 1) // 2
    fused = up(w) % 16 == 0 and up(2 * cw) % 16 == 0 and up(cw) % 8 == 0
    assert (k == "yuv420_relayout_kernel") == fused, (k, align)
    for g, wv, pd in zip(got, want, pads):
        assert (g == wv).all()
        assert (pd == 0xCD).all()
    for p in d:
        p.free()
    k = _run_batch(dev, orc, sf, df, w, h, w, h, nframes=3, nstreams=1, align=align)
    assert (k == "yuv420_relayout_kernel") == fused, (k, align)
    monkeypatch.setenv("GMAT_NO_RELAYOUT_FUSED", "1")
    d = dev.upload_planes(src, align)
    got, _, k = dev.sws(d, w, h, sf, w, h, df, dst_align=align)
    assert k != "yuv420_relayout_kernel"
    for g, wv in zip(got, want):
        assert (g == wv).all()
    for p in d:
        p.free()


@pytest.mark.parametrize("src_fmt", ["rgb24", "bgr24", "rgba", "bgra"])
def test_packed_rgb_to_planar_10bit(dev, orc, src_fmt):
    """packed RGB into YUV420P10LE (swscale_cuda.c:34-44 lists the format; round 6's format sweep — tools/x2bench sweep: — found the pair REFUSED while P010LE, the same
    samples interleaved, was served): the plane scaler with its RGB loader and the planar 10-bit output stage (yuv2planeX_10_c), equal size and scaled"""
    for geom in ((128, 72, 128, 72), (192, 108, 128, 72), (96, 40, 144, 60), (101, 45, 75, 33)):
        sw, sh, dw, dh = geom
        for flags in ("bicubic", "bilinear", "fast_bilinear"):
            src = synth_planes(orc, src_fmt, sw, sh, seed=4)
            want = orc.sws(src, sw, sh, src_fmt, dw, dh, "yuv420p10le", SWS[flags])
            d = dev.upload_planes(src, 64)
            got, pads, kernel = dev.sws(d, sw, sh, src_fmt, dw, dh, "yuv420p10le", SWS[flags], dst_align=64)
            for i, (g, wv) in enumerate(zip(got, want)):
                assert (g == wv).all(), (src_fmt, geom, flags, i, kernel)
                assert (pads[i] == 0xCD).all()
            for p in d:
                p.free()
