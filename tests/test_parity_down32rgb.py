"""The strip-walking 3:2 down-scale of NV12 into packed RGB (k_scale_yuv3x2.hip scale_yuv32r_kernel: 1080p -> 720p, 4K -> 1440p) and
the generic plane scaler it supersedes for those cases: both against the oracle on every geometry, every test naming the kernel the
selection rule must pick.

One libswscale context: hScale8To15_c with the two 6-tap phases of 3:2 for luma and — an RGB destination keeps half-width chroma —
for chroma; vertically the same pair for luma (output row 1 with its own table row), and for chroma a 3:4 UP-scale (the source's chroma
has half the source's rows, the destination's all of the destination's): 4 taps, four phases, output rows 0 and 1 with their own table
rows; yuv2rgb_X_c's sums and tables.  No vector the reference holds is a 3:2 scale: held to the oracle only."""
import numpy as np
import pytest

from harness import is_generic, SWS, synth_planes
from test_batch_api import _run_batch
from test_parity_strip import strip_rows  # noqa: F401
from harness import ratio_kernels_keep_single_frames  # noqa: F401  (autouse: this file is about the exact-ratio kernel at every launch size)

D32R = "scale_yuv32r_kernel"


def d32r_takes(dw, dh, sf):
    """the geometry part of yuv32r_prepare restated: NV12 source, destination width a multiple of 8 and >= 64, height a multiple of 4 and >= 16"""
    return sf == "nv12" and dw % 8 == 0 and dw >= 64 and dh % 4 == 0 and dh >= 16


@pytest.fixture(params=["strip", "generic"])
def kern_d32r(request, monkeypatch):
    if request.param == "generic":
        monkeypatch.setenv("GMAT_SCALE_NO_STRIP", "1")
    else:
        monkeypatch.delenv("GMAT_SCALE_NO_STRIP", raising=False)
    return request.param


# (dstW, dstH): the smallest, one partial strip (512 output columns per wave), exactly one, one + a partial one of 8 / 16 columns,
# several strips, heights over every residue of the rows modulo the unrolled pair of steps; then geometries it declines
GEOMS = [(64, 16), (128, 20), (512, 16), (520, 24), (528, 28), (1024, 16), (1040, 32), (136, 36), (72, 44), (256, 40),
         (68, 16), (500, 16), (128, 18), (56, 16), (128, 12)]


def test_geometries_cover_both_kernels():
    took = [d32r_takes(w, h, "nv12") for w, h in GEOMS]
    assert sum(took) >= 9 and took.count(False) >= 4


def _check(dev, orc, sf, df, dw, dh, flags="bicubic", align=256, extra=0, seed=73, src_fill=None, colorspace=None):
    sw, sh = 3 * dw // 2, 3 * dh // 2
    src = synth_planes(orc, sf, sw, sh, seed=seed)
    if src_fill is not None:
        src_fill(src)
    want = orc.sws(src, sw, sh, sf, dw, dh, df, SWS[flags], colorspace=colorspace)
    d = dev.upload_planes(src, 256)
    got, pads, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, SWS[flags], dst_align=align, dst_extra=extra,
                                colorspace=None if colorspace is None else (colorspace, 0))
    for i, (g, w) in enumerate(zip(got, want)):
        bad = np.argwhere(g != w)
        assert bad.size == 0, f"{kernel} plane {i}: {len(bad)} mismatching bytes, first at {bad[:6].tolist()}"
        assert (pads[i] == 0xCD).all(), f"{kernel} plane {i}: wrote into the row padding"
    for p in d:
        p.free()
    return kernel


@pytest.mark.parametrize("df", ["rgb24", "bgr24", "rgba", "bgra"])
@pytest.mark.parametrize("geom", GEOMS)
def test_down32rgb_bit_exact_on_both_kernels(dev, orc, strip_rows, kern_d32r, df, geom):
    dw, dh = geom
    strip_rows(0)
    k = _check(dev, orc, "nv12", df, dw, dh)
    if kern_d32r == "strip" and d32r_takes(dw, dh, "nv12"):
        assert k == D32R, k
    else:
        assert is_generic(k), k


def test_down32rgb_planar_source_stays_generic(dev, orc):
    assert is_generic(_check(dev, orc, "yuv420p", "rgb24", 264, 16))


@pytest.mark.parametrize("rows", [1, 4, 5, 8, 12, 13, 64])
@pytest.mark.parametrize("df", ["rgb24", "bgra"])
def test_down32rgb_segmentation_does_not_change_the_result(dev, orc, strip_rows, df, rows):
    """segments of `rows` output rows (rounded up to a multiple of 4): the warm-up steps of every segment rebuild the open luma sums
    and the chroma window of its first rows; only the first segment meets the own taps of output rows 0 and 1"""
    strip_rows(rows)
    assert _check(dev, orc, "nv12", df, 264, 44) == D32R


@pytest.mark.parametrize("cs", [1, 5, 7])
def test_down32rgb_colourspaces(dev, orc, cs):
    assert _check(dev, orc, "nv12", "rgb24", 264, 16, colorspace=cs) == D32R


@pytest.mark.parametrize("flags", ["bicubic", "bilinear", "point", "fast_bilinear", "area", "gauss", "lanczos", "sinc"])
def test_down32rgb_filters(dev, orc, kern_d32r, flags):
    """bicubic takes the strip kernel; whatever the host's checks decline stays on the generic one — the bytes are libswscale's either way"""
    k = _check(dev, orc, "nv12", "rgb24", 264, 16, flags)
    if flags == "bicubic" and kern_d32r == "strip":
        assert k == D32R, k
    if kern_d32r == "generic" or flags in ("lanczos", "sinc"):
        assert is_generic(k), (flags, k)


@pytest.mark.parametrize("df", ["rgb24", "rgba"])
def test_down32rgb_destination_alignment(dev, orc, df):
    """the kernel stores 12 / 16 bytes per lane: the tiled kernels' rule (4-byte aligned rgb24 rows, 16-byte aligned rgba rows)"""
    assert _check(dev, orc, "nv12", df, 264, 16, align=16, extra=0) == D32R
    assert is_generic(_check(dev, orc, "nv12", df, 264, 16, align=1, extra=1))


@pytest.mark.parametrize("pattern", ["max", "checker", "stripes3", "edge"])
def test_down32rgb_saturating_content(dev, orc, strip_rows, pattern):
    """all-maximum, checkerboard, period-3 stripes and energy in the border columns / rows
    only: bicubic overshoot drives hScale8To15_c's min(.., 32767), the table headroom clamp of U / V and the unclipped luma sums"""
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
    for df in ("rgb24", "bgra"):
        assert _check(dev, orc, "nv12", df, 264, 16, src_fill=fill) == D32R


@pytest.mark.parametrize("df", ["rgb24", "bgra"])
def test_down32rgb_batched_frames(dev, orc, strip_rows, kern_d32r, df):
    strip_rows(0)
    k = _run_batch(dev, orc, "nv12", df, 396, 66, 264, 44, nframes=5, nstreams=2, align=16)
    assert (k == D32R) == (kern_d32r == "strip"), k
