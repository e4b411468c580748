"""RGBA64LE / BGRA64LE as SOURCES (swscale_cuda.c:34-44 lists them; rgb64ToY_c / ToUV_c / ToUV_half_c, input.c:36-121) and the
alpha plane of contexts with an alpha channel at both ends (needAlpha, utils.c:1902: rgbaToA_c / rgba64leToA_c through the luma
filters into the packed writers' alpha, output.c) — VERDICT round 2, missing #3.  The product path: k_rgb64.hip in front of the
planar-16 contexts; bit-exact with the oracle, which reproduces the reference's filter-pixfmts-scale md5s for rgba64le, bgra64le,
rgba and bgra (tests/test_oracle_fate_nut.py)."""
import numpy as np
import pytest

from harness import SWS, synth_planes

DST8 = ["rgb24", "bgr24", "rgba", "bgra", "nv12", "yuv420p", "yuv444p"]
DST16 = ["p010le", "yuv420p10le", "p016le", "yuv444p16le", "yuv420p16le", "rgba64le", "bgra64le"]
GEOMS = [(96, 40, 50, 30), (64, 24, 96, 36), (70, 22, 70, 22), (131, 35, 64, 17), (48, 48, 97, 31), (200, 16, 100, 8)]


def _run(dev, orc, sf, df, geom, flags="bicubic", seed=5, align=64, colorspace=None, fill=None):
    sw, sh, dw, dh = geom
    if df in ("nv12", "yuv420p", "p010le", "yuv420p10le", "p016le", "yuv420p16le") and (dw % 2 or dh % 2):
        pytest.skip("4:2:0 destinations of odd size: the tiled kernel's own tests")
    src = synth_planes(orc, sf, sw, sh, seed=seed)
    if fill is not None:
        fill(src[0])
    fl = SWS[flags] if isinstance(flags, str) else flags
    want = orc.sws(src, sw, sh, sf, dw, dh, df, fl, colorspace=colorspace)
    d = dev.upload_planes(src, align)
    got, pads, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, fl, dst_align=align,
                                colorspace=None if colorspace is None else (colorspace, 0))
    for p in d:
        p.free()
    for i, (g, w) in enumerate(zip(got, want)):
        bad = np.argwhere(g != w)
        assert bad.size == 0, f"{sf}->{df} {geom} {kernel} plane {i}: {len(bad)} bytes differ, first {bad[:5].tolist()} got {g[tuple(bad[0])]} want {w[tuple(bad[0])]}"
        assert (pads[i] == 0xCD).all(), f"{kernel} plane {i}: wrote into the row padding"
    return kernel


@pytest.mark.parametrize("sf", ["rgba64le", "bgra64le"])
@pytest.mark.parametrize("df", DST8 + DST16)
@pytest.mark.parametrize("geom", GEOMS)
def test_rgb64_sources_every_destination(dev, orc, sf, df, geom):
    if sf == df and geom[0] == geom[2] and geom[1] == geom[3]:
        pytest.skip("equal format and size is the plain copy (test below)")
    _run(dev, orc, sf, df, geom)


@pytest.mark.parametrize("sf", ["rgba64le", "bgra64le"])
def test_equal_format_and_size_is_a_copy(dev, orc, sf):
    sw, sh = 70, 22
    src = synth_planes(orc, sf, sw, sh, seed=9)
    d = dev.upload_planes(src, 64)
    got, pads, kernel = dev.sws(d, sw, sh, sf, sw, sh, sf, SWS["bicubic"], dst_align=64)
    assert (got[0] == src[0]).all() and (pads[0] == 0xCD).all() and kernel == "copy2d"
    for p in d:
        p.free()


@pytest.mark.parametrize("flags", ["point", "bilinear", "fast_bilinear", "area", "lanczos",
                                   SWS["bicubic"] | SWS["full_chr_h_int"], SWS["bilinear"] | SWS["full_chr_h_int"],
                                   SWS["bicubic"] | SWS["full_chr_h_inp"], SWS["bilinear"] | SWS["accurate_rnd"] | SWS["bitexact"]])
@pytest.mark.parametrize("df", ["rgba", "bgra64le", "rgb24", "nv12", "p016le"])
def test_every_writer_form(dev, orc, flags, df):
    """the packed writers' one-tap, two-tap and X forms each have their own alpha rounding (output.c:1709-1821, :2069-2175; 64-bit
    :1052-1064, :1144-1150, :1196-1202): same height (one tap), bilinear up-scale (two proper taps), down-scales (X); chroma
    halved or not at the source (full_chr_h_inp, fast_bilinear) and at the writer (full_chr_h_int)"""
    for geom in [(64, 24, 64, 24), (64, 24, 128, 24), (64, 24, 96, 48), (64, 24, 128, 60), (96, 40, 50, 30)]:
        _run(dev, orc, "rgba64le", df, geom, flags)


@pytest.mark.parametrize("pattern", ["max", "zero", "checker", "alpha_edges"])
def test_saturating_content(dev, orc, pattern):
    """0xFFFF everywhere (hScale16To19_c's clamp, the 2^31 edge of the writers' 32-bit sums), bicubic overshoot on checkers, alpha
    steps from 0 to 0xFFFF"""
    def fill(p):
        v = p.view(np.uint16).reshape(p.shape[0], -1, 4)
        if pattern == "max":
            v[...] = 0xFFFF
        elif pattern == "zero":
            v[...] = 0
        elif pattern == "checker":
            v[...] = 0xFFFF; v[::2, ::2] = 0; v[1::2, 1::2] = 0
        else:
            v[..., 3] = 0; v[:, ::3, 3] = 0xFFFF; v[1::4, :, 3] = 0xFFFF
    for df in ("bgra64le", "rgba", "yuv444p16le", "nv12"):
        for geom in [(64, 24, 24, 10), (64, 24, 100, 40), (64, 24, 64, 24)]:
            _run(dev, orc, "rgba64le", df, geom, fill=fill)
            _run(dev, orc, "rgba64le", df, geom, "bilinear", fill=fill)


@pytest.mark.parametrize("cs", [1, 7, 9])
def test_destination_matrix(dev, orc, cs):
    """a YUV destination's matrix belongs to the RGB -> YUV stage (fill_rgb2yuv_table, utils.c:765-858)"""
    _run(dev, orc, "rgba64le", "nv12", (96, 40, 48, 20), colorspace=cs)
    _run(dev, orc, "bgra64le", "yuv444p", (70, 22, 70, 22), colorspace=cs)


@pytest.mark.parametrize("sf", ["rgba", "bgra"])
@pytest.mark.parametrize("df", ["rgba", "bgra"])
@pytest.mark.parametrize("flags", ["bicubic", "bilinear", "point", SWS["bilinear"] | SWS["full_chr_h_int"], "area", "lanczos", "fast_bilinear"])
def test_8bit_alpha_is_scaled(dev, orc, sf, df, flags):
    """RGBA / BGRA -> RGBA / BGRA at another size: the alpha channel is a fourth plane through the luma filters (rounds 1 and 2 wrote
    255 — right only for opaque frames, which is all the reference's own vectors hold).  SWS_FAST_BILINEAR keeps the half-chroma
    writer behind an RGB source (utils.c:1439-1447): the plane scaler with its RGB loader, not the RGB scaler."""
    for geom in [(96, 40, 50, 30), (64, 24, 128, 24), (64, 24, 96, 48), (64, 24, 128, 60), (131, 35, 64, 17), (64, 24, 33, 24)]:
        _run(dev, orc, sf, df, geom, flags, seed=21)


def test_8bit_alpha_extremes(dev, orc):
    def fill(p):
        v = p.reshape(p.shape[0], -1, 4)
        v[..., 3] = 0; v[:, ::2, 3] = 255; v[1::3, :, 3] = 255
    for flags in ("bicubic", "bilinear", "lanczos"):
        for geom in [(64, 24, 24, 10), (64, 24, 100, 40)]:
            _run(dev, orc, "rgba", "bgra", geom, flags, fill=fill)


def test_padding_twins_carry_no_alpha(dev, orc):
    """RGB0 / BGR0: the fourth byte is padding on either end (handle_0alpha, utils.c:1121-1144) — an RGB0 source's bytes are not
    an alpha plane, a BGR0 destination gets 255"""
    sw, sh, dw, dh = 64, 24, 40, 16
    src = synth_planes(orc, "rgba", sw, sh, seed=33)
    rgb = [np.ascontiguousarray(src[0].reshape(sh, sw, 4)[:, :, :3].reshape(sh, 3 * sw))]
    want = orc.sws(rgb, sw, sh, "rgb24", dw, dh, "rgba")[0]
    d = dev.upload_planes(src, 64)
    for sf, df in (("rgb0", "rgba"), ("rgba", "rgb0"), ("rgb0", "rgb0")):
        got, _, _ = dev.sws(d, sw, sh, sf, dw, dh, df, SWS["bicubic"], dst_align=64)
        assert (got[0] == want).all(), (sf, df)
    for p in d:
        p.free()


@pytest.mark.parametrize("sf", ["rgb24", "bgr24"])
@pytest.mark.parametrize("df", ["rgb24", "bgr24", "rgba", "bgra"])
def test_rgb_to_rgb_with_the_half_chroma_writer(dev, orc, sf, df):
    """SWS_FAST_BILINEAR on an RGB -> RGB context: chroma from pixel pairs at the source (rgb24ToUV_half_c) AND one chroma sample per
    pixel pair at the writer (yuv2rgb_X_c and its one- / two-tap forms) — refused through round 2; odd widths fall back to the full
    writer as in libswscale (utils.c:1431-1437)"""
    for geom in [(96, 40, 50, 30), (64, 24, 128, 24), (64, 24, 96, 48), (64, 24, 128, 60), (130, 36, 64, 18), (64, 24, 33, 24), (200, 16, 100, 8)]:
        k = _run(dev, orc, sf, df, geom, "fast_bilinear", seed=41)
        # (an odd destination width: the full writer — the ordinary RGB -> RGB context, on the block-cooperative form since round 5)
        assert "scale_yuv_kernel" in k or "scale_rgb" in k or (geom[2] & 1 and k == "scale_yuvg_rgbsrc_blk_kernel"), k


@pytest.mark.parametrize("geom", [(96, 20, 50, 20), (64, 12, 150, 12), (200, 8, 78, 8), (33, 8, 32, 8), (50, 8, 100, 8)])
def test_fast_bilinear_on_8bit_sources_is_hyscale_fast(dev, orc, geom):
    """SWS_FAST_BILINEAR with 8-bit samples: libswscale's horizontal scalers are ff_hyscale_fast_c / ff_hcscale_fast_c
    (hscale_fast_bilinear.c:23-55, selected swscale.c:566-574), not hScale8To15_c over initFilter's bilinear bank — rounds 1-2 ran the
    bank.  Known answer from the functions' own text, written out here in numpy for a context that keeps the height (one vertical tap:
    yuv2plane1_8_c / yuv2nv12cX_c give (line + 64) >> 7), then the oracle and the library against it and against each other."""
    sw, sh, dw, dh = geom
    src = synth_planes(orc, "yuv420p", sw, sh, seed=12)

    def fast(row, n_out, inc, chroma):
        out = np.zeros(n_out, np.int64)
        n = len(row)
        r = row.astype(np.int64)
        for i in range(n_out):
            xpos = (i * inc) & 0xFFFFFFFF
            xx, xa = xpos >> 16, (xpos & 0xFFFF) >> 9
            if ((i * inc) & 0xFFFFFFFF) >> 16 >= n - 1:
                out[i] = r[n - 1] * 128
            elif chroma:
                out[i] = r[xx] * (xa ^ 127) + r[xx + 1] * xa
            else:
                out[i] = (r[xx] << 7) + (r[xx + 1] - r[xx]) * xa
        return np.clip((out + 64) >> 7, 0, 255).astype(np.uint8)

    cw_s, cw_d = (sw + 1) // 2, (dw + 1) // 2
    inc_l = ((sw << 16) + (dw >> 1)) // dw
    inc_c = ((cw_s << 16) + (cw_d >> 1)) // cw_d
    want = [np.stack([fast(r, dw, inc_l, False) for r in src[0]]),
            np.stack([fast(r, cw_d, inc_c, True) for r in src[1]]), np.stack([fast(r, cw_d, inc_c, True) for r in src[2]])]
    got_o = orc.sws(src, sw, sh, "yuv420p", dw, dh, "yuv420p", SWS["fast_bilinear"])
    for a, b in zip(got_o, want):
        assert (a == b).all(), np.argwhere(a != b)[:4]
    for sf, df in (("yuv420p", "yuv420p"), ("yuv420p", "rgb24"), ("nv12", "nv12"), ("nv12", "bgra"), ("yuv420p", "p010le")):
        _run(dev, orc, sf, df, geom, "fast_bilinear")
    _run(dev, orc, "nv12", "rgb24", (sw, sh, dw, 2 * sh), "fast_bilinear")                  # with a real vertical filter behind it
    _run(dev, orc, "yuv420p", "nv12", (sw, sh, dw, sh // 2), "fast_bilinear")


@pytest.mark.parametrize("sf", ["rgb24", "bgr24", "rgba"])
def test_same_size_rgb_to_yuv444p_with_fast_bilinear(dev, orc, sf):
    """SWS_FAST_BILINEAR halves an RGB source's chroma even when nothing is scaled (utils.c:1529-1545): the equal-size RGB -> YUV444P
    context is then NOT a per-pixel conversion (rgb24ToUV_half_c and a 1:2 chroma filter) — found by the fuzzer once it drew the flag"""
    for geom in [(92, 120, 92, 120), (11, 28, 11, 28), (177, 26, 177, 26)]:
        k = _run(dev, orc, sf, "yuv444p", geom, "fast_bilinear", seed=17)
        assert k != "rgb2yuv444_kernel", k
        assert _run(dev, orc, sf, "yuv444p", geom, "bicubic", seed=17) == "rgb2yuv444_kernel"


@pytest.mark.parametrize("sf", ["rgb24", "bgr24", "rgba", "bgra"])
@pytest.mark.parametrize("df", ["p016le", "yuv444p16le", "yuv420p16le", "rgba64le", "bgra64le"])
def test_8bit_rgb_sources_to_16bit_destinations(dev, orc, sf, df):
    """an RGB source's lines are 16 bits wide whatever its depth (utils.c:1561-1570), so a 16-bit destination takes them to the 19-bit
    path with hScale16To19_c's own shift for them (9, swscale.c:74-76) — refused through round 2; RGBA / BGRA into RGBA64 / BGRA64
    scales the alpha plane as well"""
    for geom in [(96, 40, 50, 30), (64, 24, 96, 36), (70, 22, 70, 22), (130, 36, 64, 18)]:
        _run(dev, orc, sf, df, geom)
    _run(dev, orc, sf, df, (64, 24, 128, 48), "bilinear")
    _run(dev, orc, sf, df, (64, 24, 64, 24), "fast_bilinear")


@pytest.mark.parametrize("sf", ["p010le", "p016le", "yuv420p16le", "yuv420p10le", "yuv444p16le", "rgba64le", "rgb24"])
@pytest.mark.parametrize("df", ["nv12", "rgb24", "yuv420p", "p010le", "bgra"])
def test_16bit_lines_with_long_horizontal_filters(dev, orc, sf, df):
    """16-bit samples (and an 8-bit RGB source's 16-bit lines) at down-scale ratios beyond 3.7:1 — horizontal filters of more than 16
    taps (4K P010 -> 480p ...): refused through round 2 ("no 16-bit variant" of the tiled kernel's long-filter loop); the two template
    switches were independent all along"""
    def valid(p):
        if sf == "yuv420p10le":
            v = p.view(np.uint16); v &= 0x3FF                   # a 10-bit format holds 10-bit samples
        if sf == "p010le":
            v = p.view(np.uint16); v &= 0xFFC0
    for geom in [(384, 96, 80, 20), (640, 64, 100, 32), (300, 60, 50, 30), (512, 32, 64, 16)]:
        sw, sh, dw, dh = geom
        src = synth_planes(orc, sf, sw, sh, seed=3)
        for p in src:
            valid(p)
        want = orc.sws(src, sw, sh, sf, dw, dh, df, SWS["bicubic"])
        d = dev.upload_planes(src, 64)
        got, pads, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, SWS["bicubic"], dst_align=64)
        for pl in d:
            pl.free()
        assert all((a == b).all() for a, b in zip(got, want)), (geom, kernel)
        assert all((p == 0xCD).all() for p in pads)
