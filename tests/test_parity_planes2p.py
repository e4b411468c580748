"""The plane-walking 2:1 scaler with 4:2:0 output (k_scale_yuv2p.hip: NV12 -> NV12 and YUV420P -> YUV420P at exactly half
size, the transcode down-scale) and the tiled kernel it supersedes for those cases (scale_yuv2x_kernel<yuv>, still the path
of cross-layout pairs, range conversion, unaligned rows, widths that are not multiples of 16, Lanczos and
GMAT_SCALE_NO_STRIP): BOTH are compared with the oracle on every geometry, and every test names the kernel it expects, so
that a change of the selection rule cannot silently move a case from one kernel's coverage to the other's.

Pinning: no vector the reference holds is a 2:1 4:2:0 -> 4:2:0 scale (tests/ref/fate/filter-scale200, -scale500,
-crop_scale and filter-pixfmts-scale nv12 / yuv420p are 352x288 -> 200x200, 500x500, 400x298, 200x100).  Those pin the
oracle's code for this path (hScale8To15_c, yuv2planeX_8_c, yuv2nv12cX_c, initFilter); at this ratio the kernels are held
to the oracle only."""
import os

import numpy as np
import pytest

from harness import is_generic, SWS, synth_planes, LINES16, WALK16
from test_batch_api import _run_batch
from test_parity_strip import strip_rows  # noqa: F401


@pytest.fixture(autouse=True)
def _no_tile15(monkeypatch):
    """this file is about the 2:1 plane walkers and the tiled kernel behind them: the tile kernel on the 15-bit lines (round 6, k_scale19.hip — tests/test_parity_tile15.py) stands in front of both for the
    pairs whose plane layouts differ and is switched off here"""
    monkeypatch.setenv("GMAT_T15", "0")


STRIP, TILED, GENERIC = "scale_yuv2p_kernel", "scale_yuv2x_kernel<yuv>", "scale_yuv_kernel<64,yuv>"


def strip_takes(sw, sh, src_fmt, dst_fmt, flags="bicubic"):
    """the host rule of yuv2p_prepare restated: same chroma layout on both sides, width a multiple of 16 and >= 64,
    output height >= 16 and even (so that every plane is exactly halved), a filter that fits the 8-sample window"""
    dh = sh // 2
    fam = ("nv12", "yuv420p", "p010le", "yuv420p10le")           # 8 / 10-bit 4:2:0, interleaved or planar chroma, any pairing
    if not (src_fmt in fam and dst_fmt in fam and sw % 16 == 0 and sw >= 64 and sh % 4 == 0 and dh >= 16):
        return False
    if flags == "lanczos":                               # 12 taps: the 6-pair instantiation, on planes wide and tall enough for it
        return sw >= 128 and dh >= 24
    return flags in ("bicubic", "bilinear", "point", "area", "fast_bilinear", "gauss")


@pytest.fixture(params=["strip", "tiled"])
def kern_yuv(request):
    old = os.environ.pop("GMAT_SCALE_NO_STRIP", None)
    if request.param == "tiled":
        os.environ["GMAT_SCALE_NO_STRIP"] = "1"
    yield request.param
    os.environ.pop("GMAT_SCALE_NO_STRIP", None)
    if old is not None:
        os.environ["GMAT_SCALE_NO_STRIP"] = old


def strip_name(sf, df):
    """the name the plane-walking kernels report: by chroma layouts (same: scale_yuv2p, mixed: scale_yuv2px) and sample depths"""
    inter = ("nv12", "p010le")
    s10, d10 = sf in ("p010le", "yuv420p10le"), df in ("p010le", "yuv420p10le")
    if (sf in inter) == (df in inter):
        return {(0, 0): "scale_yuv2p_kernel", (0, 1): "scale_yuv2p_kernel<8to10>", (1, 0): "scale_yuv2p_kernel<10to8>", (1, 1): "scale_yuv2p16_kernel"}[(s10, d10)]
    return {(0, 0): "scale_yuv2px_kernel", (0, 1): "scale_yuv2px_kernel<8to10>", (1, 0): "scale_yuv2px_kernel<10to8>", (1, 1): "scale_yuv2px_kernel<10to10>"}[(s10, d10)]


def expect(kern_yuv, sw, sh, sf, df, flags="bicubic"):
    """three kernels serve a bicubic 2:1 4:2:0 -> 4:2:0 context: the plane-walking one, the tiled 2:1 one (whole 16-byte
    chunks: yuv2x_prepare wants srcW % 16 == 0) and the generic plane scaler for what both decline"""
    if kern_yuv == "strip" and strip_takes(sw, sh, sf, df, flags):
        return strip_name(sf, df)
    if sw % 16 == 0:
        return TILED
    from harness import walker_takes                      # round 3: the polyphase band walker in front of the generic plane scaler
    return "scale_yuvg_blk_kernel" if walker_takes(sw, sh, sf, df, sw // 2, sh // 2) else GENERIC      # (single-frame calls: its block-cooperative form)


# (srcW, srcH): both sides of every clause of the rule — one partial strip, exactly one luma strip (512) and one UV strip
# (256), strips + a partial one, several strip groups, widths that are multiples of 8 only (tiled), odd output heights
# (tiled), the smallest the strip kernel takes
GEOMS = [(64, 32), (256, 64), (512, 256), (528, 52), (1024, 96), (1040, 36), (2048, 32), (2064, 40), (4112, 32),
         (72, 40), (520, 24), (192, 34), (128, 60), (48, 32), (64, 28)]


def test_geometries_cover_both_kernels():
    took = [strip_takes(w, h, "nv12", "nv12") for w, h in GEOMS]
    assert sum(took) >= 8 and took.count(False) >= 4
    names = {expect(k, w, h, "nv12", "nv12") for k in ("strip", "tiled") for w, h in GEOMS}
    assert names == {STRIP, TILED, "scale_yuvg_blk_kernel"}


LANCZOS_GEOMS = [(128, 48), (256, 64), (528, 52), (1040, 96), (2064, 48), (4112, 48), (64, 48), (128, 44), (520, 48)]


def _check(dev, orc, sf, df, sw, sh, flags="bicubic", align=256, extra=0):
    src = synth_planes(orc, sf, sw, sh, seed=45)
    want = orc.sws(src, sw, sh, sf, sw // 2, sh // 2, df, SWS[flags])
    d = dev.upload_planes(src, 256)
    got, pads, kernel = dev.sws(d, sw, sh, sf, sw // 2, sh // 2, df, SWS[flags], dst_align=align, dst_extra=extra)
    for i, (g, w) in enumerate(zip(got, want)):
        bad = np.argwhere(g != w)
        assert bad.size == 0, f"{kernel} plane {i}: {len(bad)} mismatching bytes, first at {bad[:4].tolist()}"
        assert (pads[i] == 0xCD).all(), f"{kernel} plane {i}: wrote into the row padding"
    for p in d:
        p.free()
    return kernel


@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("geom", GEOMS)
def test_same_layout_bit_exact_on_both_kernels(dev, orc, strip_rows, kern_yuv, fmt, geom):
    sw, sh = geom
    strip_rows(0)
    assert _check(dev, orc, fmt, fmt, sw, sh) == expect(kern_yuv, sw, sh, fmt, fmt)


@pytest.mark.parametrize("pair", [("nv12", "yuv420p"), ("yuv420p", "nv12")])
@pytest.mark.parametrize("geom", GEOMS)
def test_mixed_layouts_bit_exact_on_both_kernels(dev, orc, strip_rows, kern_yuv, pair, geom, monkeypatch):
    """NV12 -> YUV420P (a hardware decoder's frames into a software encoder) and the reverse: the chroma stage of
    scale_yuv2px_kernel loads or stores the other layout; luma is the same walker"""
    monkeypatch.setenv("GMAT_NO_CROSS_CASCADE", "1")         # (what the 2:1 cross-layout walker declines: the tiled kernels, not round 4's cascade)
    strip_rows(0)
    assert _check(dev, orc, pair[0], pair[1], *geom) == expect(kern_yuv, geom[0], geom[1], pair[0], pair[1])


@pytest.mark.parametrize("pair", [("nv12", "yuv420p"), ("yuv420p", "nv12")])
@pytest.mark.parametrize("rows", [1, 3, 5, 13, 1000])
def test_mixed_layouts_segmentation(dev, orc, strip_rows, pair, rows):
    strip_rows(rows)
    assert _check(dev, orc, pair[0], pair[1], 528, 52) == "scale_yuv2px_kernel"


@pytest.mark.parametrize("pair", [("nv12", "yuv420p"), ("yuv420p", "nv12")])
@pytest.mark.parametrize("geom", LANCZOS_GEOMS)
def test_mixed_layouts_lanczos(dev, orc, strip_rows, kern_yuv, pair, geom):
    strip_rows(0)
    sw, sh = geom
    k = _check(dev, orc, pair[0], pair[1], sw, sh, "lanczos")
    if kern_yuv == "strip" and strip_takes(sw, sh, pair[0], pair[1], "lanczos"):
        assert k == "scale_yuv2px_kernel", k
    else:
        assert k in (TILED, GENERIC, "scale_yuvg_kernel", "scale_yuvg_blk_kernel"), k


@pytest.mark.parametrize("pair", [("nv12", "yuv420p"), ("yuv420p", "nv12")])
def test_mixed_layouts_batched(dev, orc, strip_rows, kern_yuv, pair):
    strip_rows(0)
    k = _run_batch(dev, orc, pair[0], pair[1], 528, 52, 264, 26, nframes=5, nstreams=2, align=16)
    assert k == expect(kern_yuv, 528, 52, pair[0], pair[1]), k


@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
def test_unaligned_destination_rows_fall_back(dev, orc, fmt):
    """the strip kernel stores dwords: rows that are not 4-byte aligned take the tiled kernel's byte stores"""
    assert strip_takes(528, 52, fmt, fmt)
    assert _check(dev, orc, fmt, fmt, 528, 52, align=1, extra=1) == TILED
    assert _check(dev, orc, fmt, fmt, 528, 52, align=4, extra=4) == STRIP


@pytest.mark.parametrize("updown", ["alternating", "all-down"])
@pytest.mark.parametrize("chroma_seg", ["equal", "half"])
@pytest.mark.parametrize("rows", [1, 2, 3, 4, 5, 8, 13, 64, 1000])
@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
def test_segmentation_does_not_change_the_result(dev, orc, strip_rows, monkeypatch, fmt, rows, chroma_seg, updown):
    """luma segments of `rows` rows; chroma segments of as many rows (the 8-bit rule) or of max(2, (rows + 1) / 2) (the 16-bit
    rule, forced here by GMAT_P2_CHROMA_SEG=0): the 3 warm-up row pairs of every segment re-create the vertical window exactly,
    also where the last segment is short (26 luma / 13 chroma rows).  Odd segments walk UPWARD by default — the walker sees the
    vertically mirrored plane, its row pairs reversed with swapped halves, so that both neighbours of a segment boundary read its
    rows at the same time (HBM traffic 1.155x -> 1.026x of the algorithmic bytes); GMAT_STRIP_UPDOWN=0: every segment downward —
    the bytes must not depend on it"""
    strip_rows(rows)
    if updown == "all-down":
        monkeypatch.setenv("GMAT_STRIP_UPDOWN", "0")
    else:
        monkeypatch.delenv("GMAT_STRIP_UPDOWN", raising=False)
    if chroma_seg == "half":
        monkeypatch.setenv("GMAT_P2_CHROMA_SEG", "0")
    else:
        monkeypatch.delenv("GMAT_P2_CHROMA_SEG", raising=False)
    assert _check(dev, orc, fmt, fmt, 528, 52) == STRIP


@pytest.mark.parametrize("flags", ["bilinear", "bicubic", "point", "area", "fast_bilinear", "gauss", "lanczos", "sinc"])
def test_filters(dev, orc, kern_yuv, flags):
    k = _check(dev, orc, "nv12", "nv12", 528, 52, flags)
    if kern_yuv == "tiled" or flags == "sinc":
        assert k != STRIP, (flags, k)            # filters wider than 12 taps: the tiled 2:1 kernel or the generic plane scaler
    else:
        assert k == STRIP, (flags, k)            # Lanczos-3 (12 taps) on the 6-pair instantiation


@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("geom", LANCZOS_GEOMS)
def test_lanczos_bit_exact_on_both_kernels(dev, orc, strip_rows, kern_yuv, fmt, geom):
    """Lanczos-3 at 2:1: 12 taps on [2x - 5, 2x + 6], border rows = edge replication (the host checks it); in the waves on a
    plane edge two lanes a side overlap the border by different amounts — every dword of their windows comes from its own
    clamped address"""
    sw, sh = geom
    strip_rows(0)
    k = _check(dev, orc, fmt, fmt, sw, sh, "lanczos")
    if kern_yuv == "strip" and strip_takes(sw, sh, fmt, fmt, "lanczos"):
        assert k == STRIP, k
    else:
        assert k in (TILED, GENERIC, "scale_yuvg_kernel", "scale_yuvg_blk_kernel"), k


@pytest.mark.parametrize("rows", [1, 2, 3, 5, 8, 13, 64])
@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
def test_lanczos_segmentation(dev, orc, strip_rows, fmt, rows):
    """5 warm-up row pairs per segment re-create the 6-slot vertical window exactly"""
    strip_rows(rows)
    assert _check(dev, orc, fmt, fmt, 528, 52, "lanczos") == STRIP


def test_lanczos_batched(dev, orc, strip_rows, kern_yuv):
    strip_rows(0)
    k = _run_batch(dev, orc, "nv12", "nv12", 528, 52, 264, 26, nframes=5, nstreams=2, align=16, flags=SWS["lanczos"])
    assert (k == STRIP) == (kern_yuv == "strip"), k


@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
def test_batched_frames(dev, orc, strip_rows, kern_yuv, fmt):
    strip_rows(0)
    k = _run_batch(dev, orc, fmt, fmt, 528, 52, 264, 26, nframes=5, nstreams=2, align=16)
    assert k == expect(kern_yuv, 528, 52, fmt, fmt), k


def test_range_conversion_is_not_the_strip_kernels(dev, orc):
    """limited <-> full range sits between the horizontal and the vertical stage: such contexts keep the generic kernel"""
    import ctypes as C
    from harness import PIX_FMT, planes, ints
    lib = dev.lib
    sw, sh = 256, 64
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT["nv12"], sw // 2, sh // 2, PIX_FMT["nv12"], SWS["bicubic"], None)
    assert c and lib.gmat_sws_setRange(c, 0, 1) == 0
    src = synth_planes(orc, "nv12", sw, sh, seed=3)
    d = dev.upload_planes(src, 256)
    o = dev.planes_like("nv12", sw // 2, sh // 2, 256)
    assert lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, sh,
                              planes([p.ptr for p in o]), ints([p.stride for p in o])) == sh // 2
    assert lib.gmat_sws_lastKernel(c).decode() not in (STRIP,)
    lib.gmat_sws_freeContext(c)
    for p in d + o:
        p.free()


# ---- 10 bits in 16-bit containers: P010LE -> P010LE (HDR transcode) and YUV420P10LE -> YUV420P10LE -------------------------
STRIP16 = "scale_yuv2p16_kernel"


def _synth10(orc, fmt, w, h, seed):
    src = synth_planes(orc, fmt, w, h, seed=seed)         # P010: all 16 bits random — the low six must be ignored (>> 6)
    if fmt == "yuv420p10le":
        for p in src:
            p.view("<u2")[...] &= 0x3FF                   # planar: 10 significant bits in the low end
    return src


def _check10(dev, orc, fmt, sw, sh, flags="bicubic", align=256, extra=0, seed=51):
    src = _synth10(orc, fmt, sw, sh, seed)
    want = orc.sws(src, sw, sh, fmt, sw // 2, sh // 2, fmt, SWS[flags])
    d = dev.upload_planes(src, 256)
    got, pads, kernel = dev.sws(d, sw, sh, fmt, sw // 2, sh // 2, fmt, SWS[flags], dst_align=align, dst_extra=extra)
    for i, (g, w) in enumerate(zip(got, want)):
        bad = np.argwhere(g != w)
        assert bad.size == 0, f"{kernel} plane {i}: {len(bad)} mismatching bytes, first at {bad[:4].tolist()}"
        assert (pads[i] == 0xCD).all(), f"{kernel} plane {i}: wrote into the row padding"
    for p in d:
        p.free()
    return kernel


@pytest.mark.parametrize("fmt", ["p010le", "yuv420p10le"])
@pytest.mark.parametrize("geom", GEOMS)
def test_10bit_same_layout_bit_exact_on_both_kernels(dev, orc, strip_rows, kern_yuv, fmt, geom):
    """the rule is the 8-bit one; what declines goes to the generic plane scaler (the tiled 2:1 kernel takes 8-bit sources only)"""
    sw, sh = geom
    strip_rows(0)
    k = _check10(dev, orc, fmt, sw, sh)
    if kern_yuv == "strip" and strip_takes(sw, sh, fmt, fmt):
        assert k == STRIP16, k
    else:
        assert k in WALK16 + (LINES16,), k      # (a context the strip kernel declines: the band walker over 16-bit samples since round 5, the lines form where it declines too)


@pytest.mark.parametrize("rows", [1, 2, 3, 4, 5, 8, 13, 64, 1000])
@pytest.mark.parametrize("fmt", ["p010le", "yuv420p10le"])
def test_10bit_segmentation_does_not_change_the_result(dev, orc, strip_rows, fmt, rows):
    strip_rows(rows)
    assert _check10(dev, orc, fmt, 528, 52) == STRIP16


@pytest.mark.parametrize("fmt", ["p010le", "yuv420p10le"])
def test_10bit_rows_that_are_not_8_byte_aligned_fall_back(dev, orc, fmt):
    """the 10-bit twin stores 8 bytes per lane; rows of dwords are the band walker's (16-bit samples, round 5), anything less the tiled kernel's"""
    assert _check10(dev, orc, fmt, 528, 52, align=4, extra=4) in WALK16
    assert _check10(dev, orc, fmt, 528, 52, align=2, extra=2) == GENERIC
    assert _check10(dev, orc, fmt, 528, 52, align=8, extra=8) == STRIP16


@pytest.mark.parametrize("flags", ["bilinear", "bicubic", "point", "area", "fast_bilinear", "gauss", "lanczos"])
def test_10bit_filters(dev, orc, kern_yuv, flags):
    k = _check10(dev, orc, "p010le", 528, 52, flags)
    assert (k == STRIP16) == (kern_yuv == "strip"), (flags, k)


@pytest.mark.parametrize("fmt", ["p010le", "yuv420p10le"])
@pytest.mark.parametrize("geom", LANCZOS_GEOMS)
def test_10bit_lanczos(dev, orc, strip_rows, kern_yuv, fmt, geom):
    sw, sh = geom
    strip_rows(0)
    k = _check10(dev, orc, fmt, sw, sh, "lanczos")
    assert (k == STRIP16) == (kern_yuv == "strip" and strip_takes(sw, sh, fmt, fmt, "lanczos")), k


@pytest.mark.parametrize("pair", [("nv12", "p010le"), ("p010le", "nv12"), ("yuv420p", "yuv420p10le"), ("yuv420p10le", "yuv420p")])
def test_cross_depth_lanczos(dev, orc, strip_rows, pair):
    strip_rows(0)
    for sw, sh in ((528, 52), (2064, 48)):
        assert _check_cross(dev, orc, pair[0], pair[1], sw, sh, "lanczos") == CROSS[pair]


def test_10bit_saturating_content(dev, orc, strip_rows):
    """all-maximum and checkerboard samples: hScale16To15_c's min(val >> 9, 32767) and the 10-bit output clip both trigger
    (bicubic overshoot of 1023-valued samples exceeds 15 bits)"""
    strip_rows(0)
    lib = dev.lib
    sw, sh = 528, 52
    for fmt, hi in (("p010le", 0xFFC0), ("yuv420p10le", 0x03FF)):
        for pattern in ("max", "checker"):
            src = _synth10(orc, fmt, sw, sh, 3)
            for p in src:
                v = p.view("<u2")
                v[...] = hi
                if pattern == "checker":
                    v[::2, ::2] = 0; v[1::2, 1::2] = 0
            want = orc.sws(src, sw, sh, fmt, sw // 2, sh // 2, fmt, SWS["bicubic"])
            d = dev.upload_planes(src, 256)
            got, pads, kernel = dev.sws(d, sw, sh, fmt, sw // 2, sh // 2, fmt, SWS["bicubic"], dst_align=256)
            assert kernel == STRIP16
            for g, w in zip(got, want):
                assert (g == w).all(), (fmt, pattern)
            for p in d:
                p.free()


@pytest.mark.parametrize("fmt", ["p010le", "yuv420p10le"])
def test_10bit_batched_frames(dev, orc, strip_rows, kern_yuv, fmt):
    strip_rows(0)
    k = _run_batch(dev, orc, fmt, fmt, 528, 52, 264, 26, nframes=5, nstreams=2, align=16)
    assert k == STRIP16 if kern_yuv == "strip" else k in WALK16 + (LINES16,), k


# ---- across depths: 8 -> 10 bits (an 8-bit source into a 10-bit encode) and 10 -> 8, same chroma layout -----------------------
CROSS = {("nv12", "p010le"): "scale_yuv2p_kernel<8to10>", ("yuv420p", "yuv420p10le"): "scale_yuv2p_kernel<8to10>",
         ("p010le", "nv12"): "scale_yuv2p_kernel<10to8>", ("yuv420p10le", "yuv420p"): "scale_yuv2p_kernel<10to8>"}


def _check_cross(dev, orc, sf, df, sw, sh, flags="bicubic", align=256, extra=0):
    src = _synth10(orc, sf, sw, sh, 57)
    want = orc.sws(src, sw, sh, sf, sw // 2, sh // 2, df, SWS[flags])
    d = dev.upload_planes(src, 256)
    got, pads, kernel = dev.sws(d, sw, sh, sf, sw // 2, sh // 2, df, SWS[flags], dst_align=align, dst_extra=extra)
    for i, (g, w) in enumerate(zip(got, want)):
        bad = np.argwhere(g != w)
        assert bad.size == 0, f"{kernel} plane {i}: {len(bad)} mismatching bytes, first at {bad[:4].tolist()}"
        assert (pads[i] == 0xCD).all(), f"{kernel} plane {i}: wrote into the row padding"
    for p in d:
        p.free()
    return kernel


@pytest.mark.parametrize("pair", list(CROSS))
@pytest.mark.parametrize("geom", GEOMS)
def test_cross_depth_bit_exact_on_both_kernels(dev, orc, strip_rows, kern_yuv, pair, geom):
    sw, sh = geom
    strip_rows(0)
    sf, df = pair
    k = _check_cross(dev, orc, sf, df, sw, sh)
    if kern_yuv == "strip" and strip_takes(sw, sh, "nv12", "nv12"):     # the geometry rule does not depend on the depth
        assert k == CROSS[pair], k
    else:
        assert is_generic(k), k                      # the generic plane scaler (the tiled 2:1 kernel is 8 -> 8 only)


@pytest.mark.parametrize("pair", list(CROSS))
@pytest.mark.parametrize("rows", [1, 3, 5, 13, 1000])
def test_cross_depth_segmentation(dev, orc, strip_rows, pair, rows):
    strip_rows(rows)
    assert _check_cross(dev, orc, pair[0], pair[1], 528, 52) == CROSS[pair]


@pytest.mark.parametrize("pair", list(CROSS))
def test_cross_depth_destination_alignment(dev, orc, pair):
    """8-bit destinations store dwords, 10-bit ones 8 bytes"""
    need = 8 if CROSS[pair].endswith("<8to10>") else 4
    assert _check_cross(dev, orc, pair[0], pair[1], 528, 52, align=need, extra=need) == CROSS[pair]
    assert is_generic(_check_cross(dev, orc, pair[0], pair[1], 528, 52, align=need // 2, extra=need // 2))


MIXED_DEPTH = [("nv12", "yuv420p10le"), ("yuv420p", "p010le"), ("p010le", "yuv420p"), ("yuv420p10le", "nv12"), ("p010le", "yuv420p10le"),
               ("yuv420p10le", "p010le")]


@pytest.mark.parametrize("pair", MIXED_DEPTH)
@pytest.mark.parametrize("geom", [(64, 32), (528, 52), (1040, 36), (2064, 40), (4112, 32), (72, 40), (128, 60)])
def test_mixed_layouts_across_depths(dev, orc, strip_rows, kern_yuv, pair, geom):
    """every pairing of layout and depth: the mixed-layout kernel's four depth instantiations"""
    strip_rows(0)
    sw, sh = geom
    k = _check_cross(dev, orc, pair[0], pair[1], sw, sh)
    if kern_yuv == "strip" and strip_takes(sw, sh, pair[0], pair[1]):
        assert k == strip_name(pair[0], pair[1]), k
    else:
        assert is_generic(k), k


@pytest.mark.parametrize("pair", MIXED_DEPTH[:4])
def test_mixed_layouts_across_depths_lanczos(dev, orc, strip_rows, pair):
    strip_rows(0)
    assert _check_cross(dev, orc, pair[0], pair[1], 2064, 48, "lanczos") == strip_name(pair[0], pair[1])


# ---- 8-bit 4:2:0 -> planar 4:4:4 at exactly 2:1: the luma walker + a chroma re-layout ----------------------------------------------
def to444_takes(sw, sh, src_fmt, flags="bicubic"):
    """the rule of yuv2p_prepare's 4:4:4 branch restated: an 8-bit 4:2:0 source, width a multiple of 16 and >= 64, output height >= 16,
    a luma filter that fits the 8-sample window; the chroma planes keep their size, their filters are one tap (the identity)"""
    return (src_fmt in ("nv12", "yuv420p") and sw % 16 == 0 and sw >= 64 and sh % 2 == 0 and sh // 2 >= 16 and
            flags in ("bicubic", "bilinear", "point", "area", "fast_bilinear", "gauss"))


def _check444(dev, orc, sf, sw, sh, flags="bicubic", align=256, extra=0, seed=83):
    src = synth_planes(orc, sf, sw, sh, seed=seed)
    want = orc.sws(src, sw, sh, sf, sw // 2, sh // 2, "yuv444p", SWS[flags])
    d = dev.upload_planes(src, 256)
    got, pads, kernel = dev.sws(d, sw, sh, sf, sw // 2, sh // 2, "yuv444p", SWS[flags], dst_align=align, dst_extra=extra)
    for i, (g, w) in enumerate(zip(got, want)):
        bad = np.argwhere(g != w)
        assert bad.size == 0, f"{kernel} plane {i}: {len(bad)} mismatching bytes, first at {bad[:6].tolist()}"
        assert (pads[i] == 0xCD).all(), f"{kernel} plane {i}: wrote into the row padding"
    for p in d:
        p.free()
    return kernel


@pytest.mark.parametrize("sf", ["nv12", "yuv420p"])
@pytest.mark.parametrize("geom", [(64, 32), (256, 40), (528, 52), (1040, 36), (2064, 32), (72, 40), (64, 30), (130, 52)])
def test_420_to_444_at_half_size_on_both_paths(dev, orc, strip_rows, kern_yuv, sf, geom):
    """nv12 / yuv420p -> yuv444p at exactly 2:1 (the chroma planes keep their size: libswscale's chroma filters are one tap, the
    identity): scale_yuv2p_kernel's luma walker plus uv_deinterleave_kernel / two plane copies, and the generic plane scaler for what
    the rule declines — both against the oracle"""
    sw, sh = geom
    strip_rows(0)
    k = _check444(dev, orc, sf, sw, sh)
    if kern_yuv == "strip" and to444_takes(sw, sh, sf):
        assert k == ("scale_yuv2p_kernel<luma>+uv_deinterleave_kernel" if sf == "nv12" else "scale_yuv2p_kernel<luma>+copy2d"), k
    else:
        assert is_generic(k), k


@pytest.mark.parametrize("rows", [1, 3, 5, 16, 1000])
def test_420_to_444_segmentation(dev, orc, strip_rows, rows):
    strip_rows(rows)
    assert _check444(dev, orc, "nv12", 528, 52).startswith("scale_yuv2p_kernel<luma>")


@pytest.mark.parametrize("flags", ["bilinear", "point", "area", "lanczos"])
def test_420_to_444_filters(dev, orc, flags):
    k = _check444(dev, orc, "nv12", 528, 52, flags)
    if flags == "lanczos":
        assert is_generic(k), k


def test_420_to_444_alignment_and_batches(dev, orc, strip_rows):
    """luma rows off the 4-byte grid go to the generic kernel; the chroma re-layout handles any alignment itself"""
    strip_rows(0)
    assert _check444(dev, orc, "nv12", 528, 52, align=4, extra=4).startswith("scale_yuv2p_kernel<luma>")
    assert is_generic(_check444(dev, orc, "nv12", 528, 52, align=1, extra=1))
    for sf in ("nv12", "yuv420p"):
        k = _run_batch(dev, orc, sf, "yuv444p", 528, 52, 264, 26, nframes=5, nstreams=2, align=16)
        assert k.startswith("scale_yuv2p_kernel<luma>"), k
