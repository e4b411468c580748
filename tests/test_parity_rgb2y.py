"""The strip-walking 2:1 scale of packed RGB into 8-bit 4:2:0 (k_scale_rgb2s.hip scale_rgb2y_kernel: RGB24 / BGR24 -> NV12 /
YUV420P at exactly half the size, e.g. a 4K RGB frame into a 1080p NV12 one for an encoder) and the generic plane scaler with its
RGB loader, which it supersedes for those cases: both against the oracle on every geometry, every test naming the kernel the
selection rule must pick.

One libswscale context: rgb24ToY_c / rgb24ToUV_half_c (input.c:795-866), hScale16To15_c with sh = 13 (8 taps on [2x - 3, 2x + 4]
for luma, the same on the pixel-PAIR samples for chroma), yuv2planeX_8_c with 8 taps on [2y - 3, 2y + 4] for luma and with 16 taps
on [4c - 6, 4c + 9] for chroma (full-height chroma source, quarter-height destination).  No vector the reference holds is this
scale: held to the oracle only."""
import numpy as np
import pytest

from harness import is_generic, SWS, synth_planes
from test_batch_api import _run_batch
from test_parity_strip import strip_rows  # noqa: F401

R2Y = "scale_rgb2y_kernel"


def r2y_takes(dw, dh):
    """the geometry part of rgb2y_prepare restated: destination width a multiple of 4 (a lane owns 4 luma and 2 chroma columns)
    and >= 64, destination height even and >= 16"""
    return dw % 4 == 0 and dw >= 64 and dh % 2 == 0 and dh >= 16


@pytest.fixture(params=["strip", "generic"])
def kern_r2y(request, monkeypatch):
    if request.param == "generic":
        monkeypatch.setenv("GMAT_SCALE_NO_STRIP", "1")
    else:
        monkeypatch.delenv("GMAT_SCALE_NO_STRIP", raising=False)
    return request.param


# (dstW, dstH): the smallest, one partial strip (248 output columns per wave), exactly one, one + a partial one (a last strip of 4 and
# of 8 columns: its only producing lanes sit beside the right border), more than four strips, heights of every residue of the
# chroma rows modulo the loop's period; then geometries it declines: widths not a multiple of 4, odd heights, too small
GEOMS = [(64, 16), (128, 18), (248, 16), (252, 20), (256, 22), (496, 16), (500, 24), (1000, 16), (1240, 18), (132, 26), (68, 30),
         (66, 16), (250, 16), (128, 17), (60, 16), (128, 14)]


def test_geometries_cover_both_kernels():
    took = [r2y_takes(w, h) for w, h in GEOMS]
    assert sum(took) >= 9 and took.count(False) >= 4


def _check(dev, orc, sf, df, dw, dh, flags="bicubic", align=256, extra=0, seed=71, src_fill=None, src_align=256, src_extra=0):
    sw, sh = 2 * dw, 2 * dh
    src = synth_planes(orc, sf, sw, sh, seed=seed)
    if src_fill is not None:
        src_fill(src)
    want = orc.sws(src, sw, sh, sf, dw, dh, df, SWS[flags])
    d = dev.upload_planes(src, src_align, src_extra)
    got, pads, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, SWS[flags], dst_align=align, dst_extra=extra)
    for i, (g, w) in enumerate(zip(got, want)):
        bad = np.argwhere(g != w)
        assert bad.size == 0, f"{kernel} plane {i}: {len(bad)} mismatching bytes, first at {bad[:6].tolist()}"
        assert (pads[i] == 0xCD).all(), f"{kernel} plane {i}: wrote into the row padding"
    for p in d:
        p.free()
    return kernel


@pytest.mark.parametrize("fmts", [("rgb24", "nv12"), ("bgr24", "yuv420p")])
@pytest.mark.parametrize("geom", GEOMS)
def test_rgb2y_bit_exact_on_both_kernels(dev, orc, strip_rows, kern_r2y, fmts, geom):
    dw, dh = geom
    strip_rows(0)
    k = _check(dev, orc, fmts[0], fmts[1], dw, dh)
    if kern_r2y == "strip" and r2y_takes(dw, dh):
        assert k == R2Y, k
    else:
        assert is_generic(k), k


@pytest.mark.parametrize("fmts", [("rgb24", "yuv420p"), ("bgr24", "nv12")])
def test_rgb2y_other_format_pairs(dev, orc, fmts):
    assert _check(dev, orc, fmts[0], fmts[1], 252, 20) == R2Y


@pytest.mark.parametrize("rows", [1, 2, 3, 4, 5, 7, 8, 13, 64])
@pytest.mark.parametrize("df", ["nv12", "yuv420p"])
def test_rgb2y_segmentation_does_not_change_the_result(dev, orc, strip_rows, df, rows):
    """segments of `rows` chroma rows (2 * rows luma rows): the seven warm-up row pairs of every segment re-create the luma ring
    and the five open chroma sums its first rows need"""
    strip_rows(rows)
    assert _check(dev, orc, "rgb24", df, 252, 26) == R2Y


def filters_fit(orc, dw, dh, flags):
    """the filter part of the rule, restated on the ORACLE's tables: horizontal luma / chroma and vertical luma on the window
    [2x - 3, 2x + 4], vertical chroma on [4c - 6, 4c + 9], every row equal to the middle row folded onto the clamped samples"""
    tabs = orc.sws_filters(2 * dw, 2 * dh, "rgb24", dw, dh, "nv12", SWS[flags])
    for (co, pos), (R, L, W) in zip(tabs, [(2, 3, 8), (2, 3, 8), (2, 3, 8), (4, 6, 16)]):
        n, taps = co.shape
        srcn = R * n
        xm = n // 2
        nominal = np.zeros(W, dtype=np.int64)
        for j in range(taps):
            if co[xm, j]:
                slot = pos[xm] + j - (R * xm - L)
                if slot < 0 or slot >= W:
                    return False
                nominal[slot] = co[xm, j]
        for x in range(n):
            eff = {}
            for k in range(W):
                s = min(max(R * x - L + k, 0), srcn - 1)
                eff[s] = eff.get(s, 0) + int(nominal[k])
            tab = {}
            for j in range(taps):
                if co[x, j]:
                    tab[pos[x] + j] = tab.get(pos[x] + j, 0) + int(co[x, j])
            if {k: v for k, v in eff.items() if v} != {k: v for k, v in tab.items() if v}:
                return False
    return True


@pytest.mark.parametrize("flags", ["bicubic", "bilinear", "point", "fast_bilinear", "area", "gauss", "lanczos"])
def test_rgb2y_filters(dev, orc, kern_r2y, flags):
    """(sinc is not in the list: its 4:1 vertical chroma window is beyond the generic kernel's LDS budget and the context is refused,
    with or without this kernel.)  Whatever filter fits the windows with replicated borders takes the strip kernel, the others stay on the generic one — the
    expectation comes from the oracle's own filter tables, the bytes are libswscale's either way"""
    k = _check(dev, orc, "rgb24", "nv12", 252, 20, flags)
    fits = filters_fit(orc, 252, 20, flags)
    if flags == "bicubic":
        assert fits
    if kern_r2y == "strip" and fits:
        assert k == R2Y, (flags, k)
    else:
        assert is_generic(k), (flags, k)


@pytest.mark.parametrize("df", ["nv12", "yuv420p"])
def test_rgb2y_alignment(dev, orc, df):
    """dword stores on luma and NV12 chroma, dword loads of the pixels"""
    assert _check(dev, orc, "rgb24", df, 252, 20, align=4, extra=4) == R2Y
    assert is_generic(_check(dev, orc, "rgb24", df, 252, 20, align=2, extra=2))
    assert is_generic(_check(dev, orc, "rgb24", df, 252, 20, align=1, extra=1))
    assert _check(dev, orc, "rgb24", df, 252, 20, src_align=4, src_extra=4) == R2Y
    assert is_generic(_check(dev, orc, "rgb24", df, 252, 20, src_align=1, src_extra=1))


@pytest.mark.parametrize("pattern", ["max", "checker", "primaries", "edge"])
def test_rgb2y_saturating_content(dev, orc, strip_rows, pattern):
    """all-maximum, a pixel checkerboard, saturated primaries in column stripes (the largest chroma swings the matrix can produce,
    driving the filters' overshoot into hScale16To15_c's min(.., 32767) and the 8-bit clips) and energy in the border columns /
    rows only, which reaches nothing but the replicated taps"""
    strip_rows(0)

    def fill(src):
        p = src[0]
        h, wb = p.shape
        px = p.reshape(h, wb // 3, 3)
        px[...] = 255
        if pattern == "checker":
            px[::2, ::2] = 0; px[1::2, 1::2] = 0
        if pattern == "primaries":
            px[:, 0::4] = (255, 0, 0); px[:, 1::4] = (0, 0, 255); px[:, 2::4] = (0, 255, 0); px[:, 3::4] = (255, 0, 255)
            px[1::3] = px[1::3, ::-1]
        if pattern == "edge":
            px[:, 2:-2] = 0; px[2:-2, :] = 0
    for df in ("nv12", "yuv420p"):
        assert _check(dev, orc, "rgb24", df, 252, 20, src_fill=fill) == R2Y


@pytest.mark.parametrize("fmts", [("rgb24", "nv12"), ("bgr24", "yuv420p")])
def test_rgb2y_batched_frames(dev, orc, strip_rows, kern_r2y, fmts):
    strip_rows(0)
    k = _run_batch(dev, orc, fmts[0], fmts[1], 504, 52, 252, 26, nframes=5, nstreams=2, align=16)
    assert (k == R2Y) == (kern_r2y == "strip"), k


def test_rgb2y_other_ratios_and_depths_stay_generic(dev, orc):
    for sw, sh, dw, dh, df in ((504, 40, 168, 20, "nv12"), (504, 40, 252, 10, "nv12"), (504, 40, 252, 20, "p010le")):
        src = synth_planes(orc, "rgb24", sw, sh, seed=7)
        want = orc.sws(src, sw, sh, "rgb24", dw, dh, df, SWS["bicubic"])
        d = dev.upload_planes(src, 256)
        got, _, k = dev.sws(d, sw, sh, "rgb24", dw, dh, df, SWS["bicubic"], dst_align=256)
        assert is_generic(k) and all((g == w).all() for g, w in zip(got, want)), k
        for p in d:
            p.free()
