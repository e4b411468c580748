"""The strip-walking 1:2 UP-scale of 8-bit 4:2:0 (k_scale_yuv1x2.hip: NV12 -> NV12 and YUV420P -> YUV420P at exactly twice the
size, e.g. 1080p -> 4K) and the generic plane scaler it supersedes for those cases: both against the oracle on every geometry,
every test naming the kernel the selection rule must pick.

The left border of a bicubic up-scale is NOT the interior filter on an edge-replicated line for outputs 0 and 2 (libswscale's
table reads 17729, -1345 and 3835, 13894, -1345 there); the kernel carries those two rows as extra coefficient sets, per axis.
No vector the reference holds is a 1:2 up-scale: held to the oracle only."""
import os

import numpy as np
import pytest

from harness import is_generic, SWS, synth_planes
from test_batch_api import _run_batch
from test_parity_strip import strip_rows  # noqa: F401

UP = "scale_yuv1x2_kernel"


def up_takes(sw, sh, sf, df, flags="bicubic"):
    """the host rule of yuv1x2_prepare restated: 8-bit, same chroma layout on both sides, source width a multiple of 8 and >= 32,
    source height even and >= 16, filters of at most 4 taps"""
    return (sf == df and sf in ("nv12", "yuv420p") and sw % 8 == 0 and sw >= 32 and sh % 2 == 0 and sh >= 16 and
            flags in ("bicubic", "bilinear", "point", "fast_bilinear", "area", "gauss"))


@pytest.fixture(params=["strip", "generic"])
def kern_up(request, monkeypatch):
    if request.param == "generic":
        monkeypatch.setenv("GMAT_SCALE_NO_STRIP", "1")
    else:
        monkeypatch.delenv("GMAT_SCALE_NO_STRIP", raising=False)
    return request.param


# (srcW, srcH): one partial strip (512 output columns = 256 source), exactly one, strips + a partial one, several strip groups,
# the UV plane's strip boundaries (256 output positions = 128 source = srcW 256), the smallest the kernel takes; then geometries it
# declines: widths that are multiples of 4 only, odd heights, too small
GEOMS = [(32, 16), (64, 18), (256, 32), (264, 20), (512, 24), (520, 36), (1032, 16), (2056, 22), (136, 50),
         (36, 16), (60, 20), (64, 17), (24, 16), (64, 14)]


def test_geometries_cover_both_kernels():
    took = [up_takes(w, h, "nv12", "nv12") for w, h in GEOMS]
    assert sum(took) >= 8 and took.count(False) >= 4


def _check(dev, orc, fmt, sw, sh, flags="bicubic", align=256, extra=0, seed=61):
    src = synth_planes(orc, fmt, sw, sh, seed=seed)
    want = orc.sws(src, sw, sh, fmt, 2 * sw, 2 * sh, fmt, SWS[flags])
    d = dev.upload_planes(src, 256)
    got, pads, kernel = dev.sws(d, sw, sh, fmt, 2 * sw, 2 * sh, fmt, SWS[flags], dst_align=align, dst_extra=extra)
    for i, (g, w) in enumerate(zip(got, want)):
        bad = np.argwhere(g != w)
        assert bad.size == 0, f"{kernel} plane {i}: {len(bad)} mismatching bytes, first at {bad[:6].tolist()}"
        assert (pads[i] == 0xCD).all(), f"{kernel} plane {i}: wrote into the row padding"
    for p in d:
        p.free()
    return kernel


@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("geom", GEOMS)
def test_up2_bit_exact_on_both_kernels(dev, orc, strip_rows, kern_up, fmt, geom):
    sw, sh = geom
    strip_rows(0)
    k = _check(dev, orc, fmt, sw, sh)
    if kern_up == "strip" and up_takes(sw, sh, fmt, fmt):
        assert k == UP, k
    else:
        assert is_generic(k), k


@pytest.mark.parametrize("chroma_seg", ["equal", "half"])
@pytest.mark.parametrize("rows", [2, 4, 6, 8, 10, 16, 26, 64, 1000])
@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
def test_up2_segmentation_does_not_change_the_result(dev, orc, strip_rows, monkeypatch, fmt, rows, chroma_seg):
    """segments of `rows` output rows (chroma: as many, or with GMAT_U2_CHROMA_SEG=0 half, at least 4): the 3 warm-up source rows
    of every segment re-create the four filtered rows its first output rows need; a segment writes exactly its own rows (the steps
    at its ends also produce a neighbour's row, which is dropped)"""
    strip_rows(rows)
    if chroma_seg == "half":
        monkeypatch.setenv("GMAT_U2_CHROMA_SEG", "0")
    else:
        monkeypatch.delenv("GMAT_U2_CHROMA_SEG", raising=False)
    assert _check(dev, orc, fmt, 264, 26) == UP


def filters_fit(orc, sw, sh, fmt, flags):
    """the filter part of the rule, restated on the ORACLE's tables: at most 4 taps per filter, and every output's non-zero taps
    inside its nominal window (x even: [x/2 - 2, x/2 + 1], x odd: [x/2 - 1, x/2 + 2]) — for all four filters"""
    for co, pos in orc.sws_filters(sw, sh, fmt, 2 * sw, 2 * sh, fmt, SWS[flags]):
        n, taps = co.shape
        for x in range(n):
            ws = (x >> 1) - (1 if x & 1 else 2)
            nz = np.nonzero(co[x])[0]
            if len(nz) and (pos[x] + nz.min() < ws or pos[x] + nz.max() > ws + 3):
                return False
    return True


@pytest.mark.parametrize("flags", ["bicubic", "bilinear", "point", "fast_bilinear", "area", "gauss", "lanczos", "sinc"])
def test_up2_filters(dev, orc, kern_up, flags):
    """whatever filter fits the two nominal 4-sample windows takes the strip kernel, the others (Lanczos, sinc: more taps) stay
    on the generic one — the expectation comes from the oracle's own filter tables, the bytes are libswscale's either way"""
    k = _check(dev, orc, "nv12", 264, 26, flags)
    fits = filters_fit(orc, 264, 26, "nv12", flags)
    assert fits == (flags not in ("lanczos", "sinc")), (flags, fits)      # what this list is meant to cover
    if flags == "fast_bilinear":
        # ff_hyscale_fast_c's two-tap bank (chroma weights sum to 127): inside the windows, but not a bank the strip kernel's
        # host rule accepts (rows that sum to 16384) — the generic kernel serves it
        assert is_generic(k) or k == UP, (flags, k)
    elif kern_up == "strip" and fits:
        assert k == UP, (flags, k)
    else:
        assert is_generic(k), (flags, k)


@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
def test_up2_destination_alignment(dev, orc, fmt):
    """the kernel stores 8 bytes per lane on every plane"""
    assert _check(dev, orc, fmt, 264, 26, align=8, extra=8) == UP
    assert is_generic(_check(dev, orc, fmt, 264, 26, align=4, extra=4))
    assert is_generic(_check(dev, orc, fmt, 264, 26, align=1, extra=1))


def test_up2_saturating_content(dev, orc, strip_rows):
    """all-maximum and checkerboard samples: bicubic overshoot drives hScale8To15_c's min(.., 32767) and the 8-bit clip"""
    strip_rows(0)
    sw, sh = 264, 26
    for pattern in ("max", "checker", "edge"):
        src = synth_planes(orc, "nv12", sw, sh, seed=5)
        for p in src:
            p[...] = 255
            if pattern == "checker":
                p[::2, ::2] = 0; p[1::2, 1::2] = 0
            if pattern == "edge":
                p[:, 2:] = 0; p[2:, :] = 0          # energy only in the first two columns / rows: the special border sets
        want = orc.sws(src, sw, sh, "nv12", 2 * sw, 2 * sh, "nv12", SWS["bicubic"])
        d = dev.upload_planes(src, 256)
        got, pads, k = dev.sws(d, sw, sh, "nv12", 2 * sw, 2 * sh, "nv12", SWS["bicubic"], dst_align=256)
        assert k == UP
        for g, w in zip(got, want):
            assert (g == w).all(), pattern
        for p in d:
            p.free()


@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
def test_up2_batched_frames(dev, orc, strip_rows, kern_up, fmt):
    strip_rows(0)
    k = _run_batch(dev, orc, fmt, fmt, 264, 26, 528, 52, nframes=5, nstreams=2, align=16)
    assert (k == UP) == (kern_up == "strip"), k


def test_up2_mixed_layouts_and_depths_stay_generic(dev, orc, monkeypatch):
    monkeypatch.setenv("GMAT_NO_CROSS_CASCADE", "1")          # (round 4: mixed layouts run the same-layout walker + a re-layout, tests/test_parity_cross_layout.py; this test is about the tier behind)
    for sf, df in (("nv12", "yuv420p"), ("yuv420p", "nv12")):
        src = synth_planes(orc, sf, 264, 26, seed=7)
        want = orc.sws(src, 264, 26, sf, 528, 52, df, SWS["bicubic"])
        d = dev.upload_planes(src, 256)
        got, _, k = dev.sws(d, 264, 26, sf, 528, 52, df, SWS["bicubic"], dst_align=256)
        assert is_generic(k) and all((g == w).all() for g, w in zip(got, want))
        for p in d:
            p.free()
