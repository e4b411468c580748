"""The strip-walking 3:2 down-scale of 8-bit 4:2:0 (k_scale_yuv3x2.hip: NV12 -> NV12 and YUV420P -> YUV420P at exactly two thirds
of the size, e.g. 1080p -> 720p, 4K -> 1440p) and the generic plane scaler it supersedes for those cases: both against the oracle
on every geometry, every test naming the kernel the selection rule must pick.

At 3:2 the bicubic filter has 6 taps and two phases (output 2k reads [3k - 2, 3k + 3], output 2k + 1 reads [3k - 1, 3k + 4]); every
border row of libswscale's tables is its phase's row on an edge-replicated line EXCEPT output 1 (second column / second row), whose
table row the kernel carries as an extra coefficient set (test_down32_filters restates the rule on the oracle's own tables).  No
vector the reference holds is a 3:2 scale: held to the oracle only."""
import numpy as np
import pytest

from harness import is_generic, SWS, synth_planes
from test_batch_api import _run_batch
from test_parity_strip import strip_rows  # noqa: F401

E3 = "scale_yuv3x2_kernel"


def e3_takes(dw, dh, sf, df):
    """the geometry part of yuv3x2_prepare restated: 8-bit, same chroma layout on both sides, destination width >= 64 and a multiple
    of 8 (NV12) or 16 (planar chroma: a lane makes 8 samples of a plane), destination height >= 16 and a multiple of 4 (rows come
    in pairs on both planes)"""
    return (sf == df and sf in ("nv12", "yuv420p") and dw % (8 if sf == "nv12" else 16) == 0 and dw >= 64 and dh % 4 == 0 and dh >= 16)


@pytest.fixture(params=["strip", "generic"])
def kern_e3(request, monkeypatch):
    if request.param == "generic":
        monkeypatch.setenv("GMAT_SCALE_NO_STRIP", "1")
    else:
        monkeypatch.delenv("GMAT_SCALE_NO_STRIP", raising=False)
    return request.param


# (dstW, dstH): one partial strip (512 output columns), exactly one, strips + a partial one, more than one workgroup of strips, the UV
# plane's strip boundaries (256 output positions = dstW 512), widths that are multiples of 8 but not of 16 (NV12 only), the smallest
# the kernel takes; then geometries it declines: widths that are multiples of 4 only, heights that are not multiples of 4, too small
GEOMS = [(64, 16), (128, 20), (512, 16), (528, 24), (1024, 16), (1040, 20), (2064, 16), (2576, 16), (136, 28), (72, 16), (520, 16),
         (68, 16), (128, 18), (128, 12), (48, 16), (100, 20)]


def test_geometries_cover_both_kernels():
    took = [e3_takes(w, h, "nv12", "nv12") for w, h in GEOMS]
    assert sum(took) >= 9 and took.count(False) >= 4
    assert any(e3_takes(w, h, "nv12", "nv12") and not e3_takes(w, h, "yuv420p", "yuv420p") for w, h in GEOMS)


def _check(dev, orc, fmt, dw, dh, flags="bicubic", align=256, extra=0, seed=71, src_fill=None):
    sw, sh = 3 * dw // 2, 3 * dh // 2
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
def test_down32_bit_exact_on_both_kernels(dev, orc, strip_rows, kern_e3, fmt, geom):
    dw, dh = geom
    strip_rows(0)
    k = _check(dev, orc, fmt, dw, dh)
    if kern_e3 == "strip" and e3_takes(dw, dh, fmt, fmt):
        assert k == E3, k
    else:
        assert is_generic(k), k


@pytest.mark.parametrize("rows", [2, 4, 6, 8, 10, 14, 16, 26, 64])
@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
def test_down32_segmentation_does_not_change_the_result(dev, orc, strip_rows, fmt, rows):
    """segments of `rows` output rows on every plane (odd values are rounded up: segments start on even rows): the two warm-up steps
    of every segment re-create the row pairs its first two output rows need; both parities of the step count are exercised"""
    strip_rows(rows)
    assert _check(dev, orc, fmt, 272, 28) == E3


def filters_fit(orc, dw, dh, fmt, flags):
    """the filter part of the rule, restated on the ORACLE's tables: every output's non-zero taps inside its 6-sample window
    (x = 2k: [3k - 2, 3k + 3], x = 2k + 1: [3k - 1, 3k + 4]) and every row but output 1 equal to the middle row of its parity folded
    onto the clamped samples — for all four filters"""
    for co, pos in orc.sws_filters(3 * dw // 2, 3 * dh // 2, fmt, dw, dh, fmt, SWS[flags]):
        n, taps = co.shape
        srcn = 3 * n // 2

        def window(x):
            ws = 3 * (x >> 1) - (1 if x & 1 else 2)
            w = np.zeros(6, dtype=np.int64)
            for j in range(taps):
                if co[x, j]:
                    k = pos[x] + j - ws
                    if k < 0 or k > 5:
                        return None
                    w[k] += co[x, j]
            return w
        xm = (n // 2) & ~1
        nom = [window(xm), window(xm + 1)]
        if nom[0] is None or nom[1] is None:
            return False
        for x in range(n):
            w = window(x)
            if w is None:
                return False
            if x == 1:
                continue
            ws = 3 * (x >> 1) - (1 if x & 1 else 2)
            e = np.zeros(6, dtype=np.int64)
            for k in range(6):
                s = min(max(ws + k, 0), srcn - 1)
                e[s - ws] += nom[x & 1][k]
            if (w != e).any():
                return False
    return True


@pytest.mark.parametrize("flags", ["bicubic", "bilinear", "point", "fast_bilinear", "area", "gauss", "lanczos", "sinc"])
def test_down32_filters(dev, orc, kern_e3, flags):
    """whatever filter fits the two 6-sample windows with replicated borders takes the strip kernel, the others stay on the generic
    one — the expectation comes from the oracle's own filter tables, the bytes are libswscale's either way"""
    k = _check(dev, orc, "nv12", 272, 28, flags)
    fits = filters_fit(orc, 272, 28, "nv12", flags)
    if flags == "bicubic":
        assert fits
    if flags in ("lanczos", "sinc"):
        assert not fits
    if kern_e3 == "strip" and fits:
        assert k == E3, (flags, k)
    else:
        assert is_generic(k), (flags, k)


@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
def test_down32_destination_alignment(dev, orc, fmt):
    """the kernel stores 8 bytes per lane on every plane"""
    assert _check(dev, orc, fmt, 272, 28, align=8, extra=8) == E3
    assert is_generic(_check(dev, orc, fmt, 272, 28, align=4, extra=4))
    assert is_generic(_check(dev, orc, fmt, 272, 28, align=1, extra=1))


@pytest.mark.parametrize("pattern", ["max", "checker", "stripes3", "edge", "second"])
def test_down32_saturating_content(dev, orc, strip_rows, pattern):
    """all-maximum, checkerboard, period-3 stripes, energy in the border columns / rows only, and energy in the samples that only
    output 1's own coefficient row weighs differently (columns / rows 0 .. 4): bicubic overshoot drives hScale8To15_c's
    min(.., 32767) and the 8-bit clip"""
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
            if pattern == "second":
                p[:, 5:] = 0; p[5:, :] = 0
                p[::2, 0] = 7; p[0, 1::2] = 200
    for fmt in ("nv12", "yuv420p"):
        assert _check(dev, orc, fmt, 272, 28, src_fill=fill) == E3


@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
def test_down32_batched_frames(dev, orc, strip_rows, kern_e3, fmt):
    strip_rows(0)
    k = _run_batch(dev, orc, fmt, fmt, 408, 42, 272, 28, nframes=5, nstreams=2, align=16)
    assert (k == E3) == (kern_e3 == "strip"), k


def test_down32_mixed_layouts_and_depths_stay_generic(dev, orc, monkeypatch):
    monkeypatch.setenv("GMAT_NO_CROSS_CASCADE", "1")          # (round 4: mixed layouts run the same-layout walker + a re-layout, tests/test_parity_cross_layout.py; this test is about the tier behind)
    for sf, df in (("nv12", "yuv420p"), ("yuv420p", "nv12"), ("p010le", "p010le")):
        src = synth_planes(orc, sf, 408, 42, seed=7)
        want = orc.sws(src, 408, 42, sf, 272, 28, df, SWS["bicubic"])
        d = dev.upload_planes(src, 256)
        got, _, k = dev.sws(d, 408, 42, sf, 272, 28, df, SWS["bicubic"], dst_align=256)
        assert is_generic(k) and all((g == w).all() for g, w in zip(got, want)), k
        for p in d:
            p.free()
