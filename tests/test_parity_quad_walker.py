"""The quad-lane walker (k_scale_yuvu.hip, scale_yuvu_kernel; round 4): 8-bit 4:2:0 -> packed RGB / 4:2:0 of the same chroma layout for
UP-scales of ANY factor (and, behind GMAT_QUAD_WALKER=2, whatever else has horizontal filters of at most 8 taps), against the oracle (one
libswscale context: swscale.c:234-520, initFilter utils.c:367-763) bit for bit.  A lane owns four adjacent outputs; the vertical filter is a
gather over a register ring of horizontally filtered row pairs with coefficient pairs laid out by output row — so every instance of the
ring depths (4:2:0: 3 / 5 pairs; RGB: 4 / 3, 4 / 4 with the luma stream a step behind, 6 / 4, 8 / 5), of the coefficient pairs per
horizontal window (2 / 3 / 4) and of the row-image size (one / two dwords a lane) is reached here by a geometry chosen for it, and the test
says which one through GMAT_DEBUG_WALKER's line where it matters.  Kernel names are asserted on both sides of the rule."""
import numpy as np
import os

import pytest

from harness import SWS, synth_planes, quad_takes, QUAD
from test_batch_api import _run_batch
from test_parity_strip import strip_rows  # noqa: F401

RGB = ["rgb24", "bgr24", "rgba", "bgra"]
# (srcW, srcH, dstW, dstH): 1 : 1.5 (720p -> 1080p), 1 : 1.33 (1080p -> 1440p), 1 : 2 (an RGB destination: the 1:2 plane walker does not take
# it), 1 : 2.5, 1 : 3 (720p -> 4K), 1 : 5, barely an up-scale (two dwords of a row a lane), anamorphic (more up on one axis), down
# horizontally (3 : 2, 8 taps) and up vertically, widths off the 256-column strips and off whole dwords, one strip and a remainder
GEOMS = [(160, 90, 240, 136), (192, 108, 256, 144), (128, 72, 256, 144), (128, 72, 320, 180), (128, 72, 384, 216), (64, 36, 320, 180),
         (296, 60, 300, 64), (160, 90, 400, 120), (480, 72, 320, 180), (176, 100, 262, 150), (172, 98, 258, 146), (96, 54, 522, 300)]


@pytest.fixture(autouse=True)
def behind_the_ratio_walkers(monkeypatch):
    monkeypatch.setenv("GMAT_SCALE_NO_STRIP", "1")           # (the 1 : 2 plane walker out of the way: this file is about the tier behind it)
    monkeypatch.delenv("GMAT_QUAD_WALKER", raising=False)
    monkeypatch.delenv("GMAT_SCALE_NO_QUAD_WALKER", raising=False)


def _check(dev, orc, sf, df, geom, flags="bicubic", align=256, extra=0, seed=93, src_align=256, src_extra=0, fill=None, colorspace=None):
    sw, sh, dw, dh = geom
    src = synth_planes(orc, sf, sw, sh, seed=seed)
    if fill is not None:
        fill(src)
    want = orc.sws(src, sw, sh, sf, dw, dh, df, SWS[flags], colorspace=colorspace)
    d = dev.upload_planes(src, src_align, src_extra)
    got, pads, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, SWS[flags], dst_align=align, dst_extra=extra,
                                colorspace=None if colorspace is None else (colorspace, 0))
    for p in d:
        p.free()
    for i, (g, w) in enumerate(zip(got, want)):
        bad = np.argwhere(g != w)
        assert bad.size == 0, f"{kernel} {geom} {sf} -> {df} {flags} plane {i}: {len(bad)} mismatching bytes, first at {bad[:6].tolist()}"
        assert (pads[i] == 0xCD).all(), f"{kernel} plane {i}: wrote into the row padding"
    return kernel


@pytest.mark.parametrize("sf", ["nv12", "yuv420p"])
@pytest.mark.parametrize("df", ["rgb24", "bgra", "same"])
@pytest.mark.parametrize("geom", GEOMS)
def test_up_scales(dev, orc, strip_rows, sf, df, geom):
    strip_rows(0)
    df = sf if df == "same" else df
    k = _check(dev, orc, sf, df, geom)
    assert (k == QUAD) == quad_takes(*geom[:2], sf, df, *geom[2:]), k
    assert k == QUAD or (sf == "yuv420p" and geom == (172, 98, 258, 146))     # (172 bytes a row are whole dwords, its 86-byte planar chroma rows are not)


@pytest.mark.parametrize("df", ["bgr24", "rgba"])
def test_remaining_rgb_orders(dev, orc, strip_rows, df):
    strip_rows(0)
    for geom in (GEOMS[0], GEOMS[3], GEOMS[9]):
        assert _check(dev, orc, "nv12", df, geom) == QUAD
        assert _check(dev, orc, "yuv420p", df, geom) == QUAD


@pytest.mark.parametrize("flags", ["bicubic", "bilinear", "fast_bilinear", "lanczos", "area", "point", "gauss", "sinc"])
def test_every_algorithm(dev, orc, strip_rows, flags):
    """the tables as initFilter made them: 2, 4 and 6 taps on 2 / 3 coefficient pairs and rings of 3 - 6 row pairs here; the algorithms whose
    windows grow past 8 taps on an up-scale (sinc, gauss, spline) and the 1- / 2-tap vertical forms with their per-row rounding stay where they
    were — libswscale's bytes either way"""
    strip_rows(0)
    for geom in [(160, 90, 240, 136), (128, 72, 384, 216)]:
        for sf, df in (("nv12", "rgb24"), ("yuv420p", "bgra"), ("nv12", "nv12"), ("yuv420p", "yuv420p")):
            k = _check(dev, orc, sf, df, geom, flags)
            if flags in ("bicubic", "bilinear", "lanczos"):
                assert k == QUAD, (flags, k)


@pytest.mark.parametrize("rows", [1, 2, 3, 5, 8, 13, 32, 64])
def test_band_heights(dev, orc, strip_rows, rows):
    """every band height against the rings: a band shorter than a vertical window (its first rows lean on pairs filtered for nothing above
    it), a band that ends inside a step, one band for the whole frame"""
    strip_rows(rows)
    for geom in [(160, 90, 240, 136), (128, 72, 320, 180), (480, 72, 320, 180)]:
        assert _check(dev, orc, "nv12", "rgb24", geom) == QUAD
        assert _check(dev, orc, "yuv420p", "yuv420p", geom) == QUAD
        assert _check(dev, orc, "nv12", "nv12", geom) == QUAD
    assert _check(dev, orc, "yuv420p", "bgra", (128, 72, 384, 216), "lanczos") == QUAD


def test_rule_clauses(dev, orc, strip_rows, monkeypatch):
    """both sides of the host rule (yuvu_prepare / yuvu_eligible): the band walker's pointer and format rule, an up-scale on the vertical axis,
    horizontal filters of at most 8 taps; GMAT_QUAD_WALKER = 0 / 2"""
    strip_rows(0)
    g = (160, 90, 240, 136)
    assert _check(dev, orc, "nv12", "rgb24", g) == QUAD
    assert _check(dev, orc, "nv12", "rgb24", g, align=4, extra=4) == QUAD                   # dword pitch is enough ...
    assert _check(dev, orc, "nv12", "rgb24", g, align=1, extra=1) != QUAD                   # ... a byte pitch is not
    assert _check(dev, orc, "nv12", "rgb24", g, src_align=1, src_extra=1) != QUAD
    assert _check(dev, orc, "nv12", "rgb24", g, src_align=4, src_extra=4) == QUAD
    assert _check(dev, orc, "nv12", "nv12", g, align=2, extra=2) != QUAD
    assert _check(dev, orc, "nv12", "rgb24", (162, 90, 240, 136)) != QUAD                   # 162 bytes: not whole dwords
    assert _check(dev, orc, "yuv420p", "rgb24", (164, 90, 240, 136)) != QUAD                # planar chroma rows of 82 bytes
    assert _check(dev, orc, "nv12", "rgb24", (164, 90, 240, 136)) == QUAD                   # interleaved chroma rows of 164 bytes
    assert _check(dev, orc, "nv12", "yuv420p", g) == QUAD and _check(dev, orc, "yuv420p", "nv12", g) == QUAD      # the kernel in the source's layout + a re-layout (the cascade)
    monkeypatch.setenv("GMAT_NO_CROSS_CASCADE", "1")
    assert _check(dev, orc, "nv12", "yuv420p", g) != QUAD and _check(dev, orc, "yuv420p", "nv12", g) != QUAD      # ... its own rule: one chroma layout
    monkeypatch.delenv("GMAT_NO_CROSS_CASCADE")
    assert _check(dev, orc, "nv12", "rgb24", (160, 90, 241, 136)) != QUAD                   # odd width: libswscale's full-chroma output
    assert _check(dev, orc, "nv12", "rgb24", (12, 8, 24, 16)) != QUAD                       # narrower than 16
    assert _check(dev, orc, "nv12", "rgb24", (16, 8, 32, 16)) == QUAD
    assert _check(dev, orc, "nv12", "rgb24", (160, 90, 240, 90)) != QUAD                    # not an up-scale on the vertical axis
    assert _check(dev, orc, "nv12", "rgb24", (800, 72, 320, 180)) != QUAD                   # 2.5 : 1 down horizontally: 11-tap filters
    monkeypatch.setenv("GMAT_QUAD_WALKER", "0")
    assert _check(dev, orc, "nv12", "rgb24", g) != QUAD
    monkeypatch.setenv("GMAT_QUAD_WALKER", "2")                                             # wherever it is eligible
    assert _check(dev, orc, "nv12", "rgb24", (160, 90, 240, 90)) == QUAD
    monkeypatch.delenv("GMAT_QUAD_WALKER")
    monkeypatch.setenv("GMAT_SCALE_NO_QUAD_WALKER", "1")
    assert _check(dev, orc, "nv12", "rgb24", g) != QUAD


@pytest.mark.parametrize("sf", ["nv12", "yuv420p"])
@pytest.mark.parametrize("geom", [(384, 216, 256, 144), (384, 216, 200, 120), (320, 180, 300, 170), (400, 220, 392, 216), (384, 216, 220, 124)])
def test_short_filter_down_scales(dev, orc, strip_rows, monkeypatch, sf, geom):
    """GMAT_QUAD_WALKER=2: down-scales up to 2 : 1 — 6- and 8-tap filters on 3 / 4 coefficient pairs, rings of 5 row pairs (4:2:0) and
    6 / 4, 8 / 5 (RGB): the instances no up-scale reaches"""
    monkeypatch.setenv("GMAT_QUAD_WALKER", "2")
    strip_rows(0)
    for df in ("rgb24", "bgra", sf):
        assert _check(dev, orc, sf, df, geom) == QUAD
    assert _check(dev, orc, sf, "rgb24", geom, "bilinear") == QUAD


@pytest.mark.parametrize("pattern", ["max", "checker", "stripes3", "edge"])
def test_saturating_content(dev, orc, strip_rows, pattern):
    """bicubic overshoot against hScale8To15_c's min(.., 32767), the tables' index clamp of U / V and the unclipped luma sums"""
    strip_rows(0)

    def fill(src):
        for p in src:
            p[...] = 255
            if pattern == "checker":
                p[::2, ::2] = 0; p[1::2, 1::2] = 0
            if pattern == "stripes3":
                p[:, ::3] = 0; p[1::3, :] = 0
            if pattern == "edge":
                p[:, 2:-2] = 0; p[2:-2, :] = 0
    for df in ("rgb24", "bgra", "nv12"):
        assert _check(dev, orc, "nv12", df, (160, 90, 240, 136), fill=fill) == QUAD
        assert _check(dev, orc, "nv12", df, (128, 72, 384, 216), "lanczos", fill=fill) == QUAD


@pytest.mark.parametrize("cs", [1, 5, 9])
def test_colour_matrices(dev, orc, strip_rows, cs):
    strip_rows(0)
    assert _check(dev, orc, "nv12", "rgb24", (160, 90, 240, 136), colorspace=cs) == QUAD
    assert _check(dev, orc, "yuv420p", "bgra", (160, 90, 240, 136), colorspace=cs) == QUAD


@pytest.mark.parametrize("df", ["rgb24", "bgra", "nv12"])
def test_batched_frames(dev, orc, strip_rows, df):
    """grid.y = frame through gmat_sws_scale_batch, two streams; every launch size on the same kernel"""
    strip_rows(0)
    assert _run_batch(dev, orc, "nv12", df, 160, 90, 240, 136, nframes=9, nstreams=2, align=16) == QUAD
    assert _run_batch(dev, orc, "nv12", df, 160, 90, 240, 136, nframes=2, nstreams=1, align=16) == QUAD
    assert _run_batch(dev, orc, "yuv420p", "yuv420p" if df == "nv12" else df, 128, 72, 320, 180, nframes=3, nstreams=1, align=64) == QUAD


@pytest.mark.parametrize("df", ["rgb24", "nv12"])
def test_short_filter_down_scales_in_launches_of_more_than_three_frames(dev, orc, strip_rows, df):
    """the shipped rule (gsws.cpp yuvu_eligible, measured in profiles/r04v_*): a down-scale with filters of at most 8 taps takes this kernel when
    a launch holds more than three frames (1440p -> 1080p: 5.0 -> 3.2 us a frame against the band walker), the band walker's forms below that"""
    strip_rows(0)
    assert _run_batch(dev, orc, "nv12", df, 384, 216, 288, 162, nframes=9, nstreams=2, align=16) == QUAD         # four and five frames a launch
    assert _run_batch(dev, orc, "nv12", df, 384, 216, 288, 162, nframes=4, nstreams=1, align=16) == QUAD
    assert _run_batch(dev, orc, "nv12", df, 384, 216, 288, 162, nframes=3, nstreams=1, align=16) in ("scale_yuvg_blk_kernel", "scale_yuvg_kernel")
    assert _run_batch(dev, orc, "nv12", df, 384, 216, 160, 90, nframes=5, nstreams=1, align=16) == "scale_yuvg_blk_kernel"  # 2.4 : 1: 11-tap filters — the band walker's (a launch this small: its block form, round 5)
    os.environ["GMAT_STRIP_BLOCK"] = "3"
    try:
        assert _run_batch(dev, orc, "nv12", df, 384, 216, 160, 90, nframes=5, nstreams=1, align=16) == "scale_yuvg_kernel"
    finally:
        os.environ.pop("GMAT_STRIP_BLOCK", None)
