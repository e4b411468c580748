"""The strip-walking 3:1 down-scale of 8-bit 4:2:0 (k_scale_yuv3x1.hip: NV12 -> NV12 and YUV420P -> YUV420P at exactly a third of
the size, e.g. 4K -> 720p, 1080p -> 360p) and the generic plane scaler it supersedes for those cases: both against the oracle on
every geometry, every test naming the kernel the selection rule must pick.

At 3:1 the bicubic filter is one phase of 11 taps on [3x - 4, 3x + 6]; every border row of libswscale's tables is that row on an
edge-replicated line (checked by the host per context, and by test_down3_filters on the oracle's own tables).  No vector the
reference holds is a 3:1 scale: held to the oracle only."""
import numpy as np
import pytest

from harness import is_generic, SWS, synth_planes
from test_batch_api import _run_batch
from test_parity_strip import strip_rows  # noqa: F401

D3 = "scale_yuv3x1_kernel"


def d3_takes(dw, dh, sf, df):
    """the geometry part of yuv3x1_prepare restated: 8-bit, same chroma layout on both sides, destination width a multiple of 8
    and >= 32, destination height even and >= 12 (the window of the middle row of the 6 chroma rows must not touch a border: it
    is the row every other row is compared with)"""
    return sf == df and sf in ("nv12", "yuv420p") and dw % 8 == 0 and dw >= 32 and dh % 2 == 0 and dh >= 12


@pytest.fixture(params=["strip", "generic"])
def kern_d3(request, monkeypatch):
    if request.param == "generic":
        monkeypatch.setenv("GMAT_SCALE_NO_STRIP", "1")
    else:
        monkeypatch.delenv("GMAT_SCALE_NO_STRIP", raising=False)
    return request.param


# (dstW, dstH): one partial strip (256 output columns), exactly one, strips + a partial one, more than one group of four strips,
# the UV plane's strip boundaries (128 output positions = dstW 256), a width whose last interior-looking wave touches the right
# border, the smallest the kernel takes; then geometries it declines: widths that are multiples of 4 only, odd heights, too small
GEOMS = [(32, 12), (64, 14), (256, 12), (264, 16), (512, 12), (520, 18), (1032, 12), (1288, 14), (136, 26), (248, 12), (768, 12),
         (36, 12), (60, 12), (64, 13), (24, 12), (64, 10)]


def test_geometries_cover_both_kernels():
    took = [d3_takes(w, h, "nv12", "nv12") for w, h in GEOMS]
    assert sum(took) >= 9 and took.count(False) >= 4


def _check(dev, orc, fmt, dw, dh, flags="bicubic", align=256, extra=0, seed=67, src_fill=None):
    sw, sh = 3 * dw, 3 * dh
    src = synth_planes(orc, fmt, sw, sh, seed=seed)
    if src_fill is not None:
        src_fill(src)
    want = orc.sws(src, sw, sh, fmt, dw, dh, fmt, SWS[flags])
    d = dev.upload_planes(src, 256)
    got, pads, kernel = dev.sws(d, sw, sh, fmt, dw, dh, fmt, SWS[flags], dst_align=align, dst_extra=extra)
    for i, (g, w) in enumerate(zip(got, want)):
        bad = np.argwhere(g != w)
        assert bad.size == 0, f"{kernel} plane {i}: {len(bad)} mismatching bytes, first at {bad[:6].tolist()}"
        assert (pads[i] == 0xCD).all(), f"{kernel} plane {i}: wrote into the row padding"
    for p in d:
        p.free()
    return kernel


@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("geom", GEOMS)
def test_down3_bit_exact_on_both_kernels(dev, orc, strip_rows, kern_d3, fmt, geom):
    dw, dh = geom
    strip_rows(0)
    k = _check(dev, orc, fmt, dw, dh)
    if kern_d3 == "strip" and d3_takes(dw, dh, fmt, fmt):
        assert k == D3, k
    else:
        assert is_generic(k), k


@pytest.mark.parametrize("updown", ["alternating", "all-down"])
@pytest.mark.parametrize("rows", [1, 2, 3, 4, 5, 7, 8, 13, 64])
@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
def test_down3_segmentation_does_not_change_the_result(dev, orc, strip_rows, monkeypatch, fmt, rows, updown):
    """segments of `rows` output rows (on every plane): the three warm-up steps of every segment re-create the ring of
    row pairs its first output row needs; every residue of the segment start modulo the ring's four steps is exercised.
    Odd segments walk UPWARD by default (reversed row order, coefficient pairs reversed with swapped halves, so that both
    neighbours of a boundary read its rows at the same time); GMAT_STRIP_UPDOWN=0 makes every segment walk downward — the
    bytes must not depend on it"""
    strip_rows(rows)
    if updown == "all-down":
        monkeypatch.setenv("GMAT_STRIP_UPDOWN", "0")
    else:
        monkeypatch.delenv("GMAT_STRIP_UPDOWN", raising=False)
    assert _check(dev, orc, fmt, 264, 26) == D3


def filters_fit(orc, dw, dh, fmt, flags):
    """the filter part of the rule, restated on the ORACLE's tables: every output's non-zero taps inside [3x - 4, 3x + 6], and every
    row equal to the middle row folded onto the clamped samples — for all four filters"""
    for co, pos in orc.sws_filters(3 * dw, 3 * dh, fmt, dw, dh, fmt, SWS[flags]):
        n, taps = co.shape
        srcn = 3 * n
        xm = n // 2
        nominal = np.zeros(12, dtype=np.int64)
        for j in range(taps):
            if co[xm, j]:
                slot = pos[xm] + j - (3 * xm - 4)
                if slot < 0 or slot > 10:
                    return False
                nominal[slot] = co[xm, j]
        for x in range(n):
            eff = {}
            for k in range(12):
                s = min(max(3 * x - 4 + k, 0), srcn - 1)
                eff[s] = eff.get(s, 0) + int(nominal[k])
            tab = {}
            for j in range(taps):
                if co[x, j]:
                    tab[pos[x] + j] = tab.get(pos[x] + j, 0) + int(co[x, j])
            if {k: v for k, v in eff.items() if v} != tab:
                return False
    return True


@pytest.mark.parametrize("flags", ["bicubic", "bilinear", "point", "fast_bilinear", "area", "gauss", "lanczos", "sinc"])
def test_down3_filters(dev, orc, kern_d3, flags):
    """whatever filter fits the 11-sample window with replicated borders takes the strip kernel, the others stay on the generic
    one — the expectation comes from the oracle's own filter tables, the bytes are libswscale's either way"""
    k = _check(dev, orc, "nv12", 264, 14, flags)
    fits = filters_fit(orc, 264, 14, "nv12", flags)
    if flags == "bicubic":
        assert fits
    if flags in ("lanczos", "sinc"):
        assert not fits
    if kern_d3 == "strip" and fits:
        assert k == D3, (flags, k)
    else:
        assert is_generic(k), (flags, k)


@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
def test_down3_destination_alignment(dev, orc, fmt):
    """the kernel stores 4 bytes per lane on every plane"""
    assert _check(dev, orc, fmt, 264, 14, align=4, extra=4) == D3
    assert is_generic(_check(dev, orc, fmt, 264, 14, align=2, extra=2))
    assert is_generic(_check(dev, orc, fmt, 264, 14, align=1, extra=1))


@pytest.mark.parametrize("pattern", ["max", "checker", "stripes3", "edge"])
def test_down3_saturating_content(dev, orc, strip_rows, pattern):
    """all-maximum, checkerboard, period-3 stripes (aliasing straight onto the filter's lobes) and energy in the border columns /
    rows only: bicubic overshoot drives hScale8To15_c's min(.., 32767) and the 8-bit clip; the border pattern reaches nothing but
    the replicated taps"""
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
    for fmt in ("nv12", "yuv420p"):
        assert _check(dev, orc, fmt, 264, 14, src_fill=fill) == D3


@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
def test_down3_batched_frames(dev, orc, strip_rows, kern_d3, fmt):
    strip_rows(0)
    k = _run_batch(dev, orc, fmt, fmt, 792, 78, 264, 26, nframes=5, nstreams=2, align=16)
    assert (k == D3) == (kern_d3 == "strip"), k


def test_down3_mixed_layouts_and_depths_stay_generic(dev, orc, monkeypatch):
    monkeypatch.setenv("GMAT_NO_CROSS_CASCADE", "1")          # (round 4: mixed layouts run the same-layout walker + a re-layout, tests/test_parity_cross_layout.py; this test is about the tier behind)
    for sf, df in (("nv12", "yuv420p"), ("yuv420p", "nv12"), ("p010le", "p010le")):
        src = synth_planes(orc, sf, 792, 42, seed=7)
        want = orc.sws(src, 792, 42, sf, 264, 14, df, SWS["bicubic"])
        d = dev.upload_planes(src, 256)
        got, _, k = dev.sws(d, 792, 42, sf, 264, 14, df, SWS["bicubic"], dst_align=256)
        assert is_generic(k) and all((g == w).all() for g, w in zip(got, want)), k
        for p in d:
            p.free()
