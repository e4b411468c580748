"""The strip-walking 2:1 scaler (k_scale_yuv2s.hip): 4:2:0 -> packed RGB at exactly half size, the headline kernel.
Every case goes through gmat_sws_scale_batch (the entry point that selects it) and must equal, byte for byte, what ONE
libswscale context computes (the oracle: swscale.c:234-520 with yuv2rgb_X_c output) — including the frame borders, where
the kernel replicates edge samples instead of reading libswscale's folded coefficient rows (initFilter, utils.c:601-640)."""
import ctypes as C
import os

import numpy as np
import pytest

from harness import PIX_FMT, SWS, ints, synth_planes
from test_batch_api import _run_batch


@pytest.fixture
def strip_rows():
    """GMAT_STRIP_ROWS (rows per strip segment) is read at every launch: vary the segmentation per test"""
    old = os.environ.get("GMAT_STRIP_ROWS")

    def set_rows(n):
        if n:
            os.environ["GMAT_STRIP_ROWS"] = str(n)
        else:
            os.environ.pop("GMAT_STRIP_ROWS", None)
    yield set_rows
    if old is None:
        os.environ.pop("GMAT_STRIP_ROWS", None)
    else:
        os.environ["GMAT_STRIP_ROWS"] = old


@pytest.fixture(params=["walk", "blk"])
def strip_form(request):
    """the two forms of the 4-pair strip kernel: the register-window WALKER (scale_yuv2s_kernel: every launch of more than two
    frames) and the BLOCK-cooperative form of small launches (scale_yuv2s_blk_kernel, round 4: what one sws_scale() call gets).
    GMAT_STRIP_BLOCK = n sends launches of up to n frames to the block form: 0 / 32 force either at every launch size.  Yields the
    kernel name to expect."""
    old = os.environ.get("GMAT_STRIP_BLOCK")
    os.environ["GMAT_STRIP_BLOCK"] = "0" if request.param == "walk" else "32"
    yield "scale_yuv2s_kernel" if request.param == "walk" else "scale_yuv2s_blk_kernel"
    if old is None:
        os.environ.pop("GMAT_STRIP_BLOCK", None)
    else:
        os.environ["GMAT_STRIP_BLOCK"] = old


# (srcW, srcH, row alignment): one partial strip, exactly one strip, strips + a partial one, two strip groups, widths that
# are multiples of 8 only, heights that leave a short last segment
GEOMS = [(32, 16, 4), (64, 32, 16), (512, 40, 64), (520, 24, 4), (1032, 36, 8), (2056, 20, 4), (2560, 18, 256), (4104, 16, 4)]


@pytest.mark.parametrize("dst_fmt", ["rgb24", "bgr24", "rgba", "bgra"])
@pytest.mark.parametrize("src_fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("geom", GEOMS)
def test_strip_kernel_bit_exact(dev, orc, strip_rows, strip_form, src_fmt, dst_fmt, geom):
    sw, sh, align = geom
    if dst_fmt in ("rgba", "bgra"):
        align = max(align, 16)                  # 16-byte pixel-group stores
    strip_rows(0)
    k = _run_batch(dev, orc, src_fmt, dst_fmt, sw, sh, sw // 2, sh // 2, nframes=2, nstreams=1, align=align)
    assert k == strip_form, k


@pytest.mark.parametrize("rows", [1, 2, 3, 5, 8, 13, 64, 1000])
def test_strip_segmentation_does_not_change_the_result(dev, orc, strip_rows, strip_form, rows):
    """segments of any height: the 3 warm-up row pairs of every segment re-create the vertical window exactly (the block form
    has three band heights: 8, 12 and 16 rows)"""
    strip_rows(rows)
    k = _run_batch(dev, orc, "nv12", "rgb24", 528, 52, 264, 26, nframes=4, nstreams=2, align=16)
    assert k == strip_form, k
    for sh in (16, 18, 34, 46, 70):                 # last bands of 1 .. 11 rows, frames shorter than a band
        k = _run_batch(dev, orc, "yuv420p", "bgra", 264, sh, 132, sh // 2, nframes=1, nstreams=1, align=16)
        assert k == strip_form, k


def test_strip_form_follows_the_launch_size(dev, orc, strip_rows):
    """the shipped rule (no knob, round 5): the block-cooperative kernel while the launch has at most 17 wave-rows a wave slot — every launch of frames
    this small — the walker beyond (4K -> 1080p: from 13 frames on, tests/test_fullsize_gpu.py); GMAT_STRIP_BLOCK = n draws the line at n FRAMES"""
    strip_rows(0)
    old = os.environ.pop("GMAT_STRIP_BLOCK", None)
    try:
        for n in (1, 3, 4, 7, 33):
            assert _run_batch(dev, orc, "nv12", "rgb24", 528, 52, 264, 26, nframes=n, nstreams=1, align=16) == "scale_yuv2s_blk_kernel"
        os.environ["GMAT_STRIP_BLOCK"] = "3"
        for n, want in ((3, "scale_yuv2s_blk_kernel"), (4, "scale_yuv2s_kernel"), (7, "scale_yuv2s_kernel")):
            assert _run_batch(dev, orc, "nv12", "rgb24", 528, 52, 264, 26, nframes=n, nstreams=1, align=16) == want
    finally:
        os.environ.pop("GMAT_STRIP_BLOCK", None)
        if old is not None:
            os.environ["GMAT_STRIP_BLOCK"] = old


@pytest.mark.parametrize("flags", ["bilinear", "bicubic", "point", "area", "fast_bilinear", "gauss"])
def test_strip_kernel_filters_that_fit_the_window(dev, orc, strip_rows, flags):
    """every filter whose taps fit [2x - 3, 2x + 4] and whose border rows are edge replication takes the strip kernel;
    the others (lanczos: 12 taps) stay on the tiled kernel — either way the bytes are libswscale's"""
    strip_rows(0)
    k = _run_batch(dev, orc, "nv12", "rgb24", 320, 48, 160, 24, nframes=2, nstreams=1, align=16, flags=SWS[flags])
    assert k.startswith("scale_yuv2"), k


def test_filters_wider_than_12_taps_fall_back(dev, orc, strip_rows):
    strip_rows(0)
    k = _run_batch(dev, orc, "nv12", "rgb24", 320, 48, 160, 24, nframes=2, nstreams=1, align=16, flags=SWS["sinc"])
    assert not k.startswith("scale_yuv2s"), k


# Lanczos-3 at 2:1: 12 taps on [2x - 5, 2x + 6] = the 6-pair instantiation, on frames at least 128 wide and 24 tall (12 output
# rows); smaller ones stay on the tiled kernel.  (srcW, srcH, alignment): partial strips, two strip groups, widths that are
# multiples of 8 only, both edge lanes of a side inside one wave (the narrow ones)
LANCZOS_GEOMS = [(128, 24, 4), (136, 32, 8), (320, 48, 16), (520, 36, 4), (1032, 28, 8), (2056, 40, 4), (4104, 24, 4), (64, 32, 16), (128, 20, 4)]


LZ = "scale_yuv2s_np_kernel<6>"                      # the 6-pair kernel (the 4-pair headline kernel is a separate, untouched one)


def lanczos_takes(sw, sh):
    return sw % 8 == 0 and sw >= 128 and sh // 2 >= 12


@pytest.mark.parametrize("dst_fmt", ["rgb24", "bgra"])
@pytest.mark.parametrize("src_fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("geom", LANCZOS_GEOMS)
def test_lanczos_on_the_strip_kernel(dev, orc, strip_rows, kern, src_fmt, dst_fmt, geom):
    """both kernels (the `kern` fixture: strip / tiled) against the oracle on every geometry"""
    sw, sh, align = geom
    if dst_fmt == "bgra":
        align = max(align, 16)
    strip_rows(0)
    k = _run_batch(dev, orc, src_fmt, dst_fmt, sw, sh, sw // 2, sh // 2, nframes=2, nstreams=1, align=align, flags=SWS["lanczos"])
    if kern.startswith("scale_yuv2s") and lanczos_takes(sw, sh):
        assert k == LZ, k
    else:
        assert not k.startswith("scale_yuv2s"), k


@pytest.mark.parametrize("rows", [1, 2, 3, 5, 8, 13, 64])
def test_lanczos_segmentation(dev, orc, strip_rows, rows):
    """5 warm-up row pairs per segment re-create the 6-slot vertical window; the chroma row of an output row is requested one
    iteration ahead of it, whatever the segment length"""
    strip_rows(rows)
    assert _run_batch(dev, orc, "nv12", "rgb24", 528, 52, 264, 26, nframes=3, nstreams=2, align=16, flags=SWS["lanczos"]) == LZ


@pytest.mark.parametrize("cs", [1, 9])
def test_lanczos_colorspace(dev, orc, strip_rows, cs):
    strip_rows(0)
    lib = dev.lib
    sw, sh = 528, 52
    src = synth_planes(orc, "yuv420p", sw, sh, seed=91)
    want = orc.sws(src, sw, sh, "yuv420p", sw // 2, sh // 2, "rgb24", SWS["lanczos"], colorspace=cs)
    d = dev.upload_planes(src, 64)
    got, pads, k = dev.sws(d, sw, sh, "yuv420p", sw // 2, sh // 2, "rgb24", SWS["lanczos"], dst_align=64, colorspace=(cs, 0))
    assert k == LZ and (got[0] == want[0]).all() and (pads[0] == 0xCD).all()
    for p in d:
        p.free()


@pytest.mark.parametrize("cs", [1, 5, 9])
def test_strip_kernel_colorspace(dev, orc, strip_rows, cs):
    """the LDS colour tables are built from the context's constants: BT.709 / BT.601 / BT.2020 sources"""
    strip_rows(0)
    lib = dev.lib
    sw, sh, n = 256, 32, 2
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT["nv12"], sw // 2, sh // 2, PIX_FMT["rgb24"], SWS["bicubic"], None)
    assert c and lib.gmat_sws_setColorspace(c, cs, 0) == 0
    srcs = [synth_planes(orc, "nv12", sw, sh, seed=900 + f + cs) for f in range(n)]
    dsrc = [dev.upload_planes(s, 16) for s in srcs]
    ddst = [dev.planes_like("rgb24", sw // 2, sh // 2, 16) for _ in range(n)]
    sp = (C.c_void_p * (4 * n))(); dp = (C.c_void_p * (4 * n))()
    for f in range(n):
        for i, p in enumerate(dsrc[f]): sp[4 * f + i] = p.ptr
        dp[4 * f] = ddst[f][0].ptr
    st = C.c_void_p(); assert lib.gmat_stream_create(C.byref(st)) == 0
    streams = (C.c_void_p * 1)(st)
    r = lib.gmat_sws_scale_batch(c, n, C.cast(sp, C.POINTER(C.c_void_p)), ints([p.stride for p in dsrc[0]]),
                                 C.cast(dp, C.POINTER(C.c_void_p)), ints([ddst[0][0].stride]),
                                 C.cast(streams, C.POINTER(C.c_void_p)), 1, 0)
    assert r == n and lib.gmat_sws_lastKernel(c) in (b"scale_yuv2s_kernel", b"scale_yuv2s_blk_kernel")
    lib.gmat_stream_sync(st)
    for f in range(n):
        want = orc.sws(srcs[f], sw, sh, "nv12", sw // 2, sh // 2, "rgb24", SWS["bicubic"], colorspace=cs)
        assert (ddst[f][0].download() == want[0]).all(), (cs, f)
    lib.gmat_stream_destroy(st)
    lib.gmat_sws_freeContext(c)
    for f in range(n):
        for p in dsrc[f] + ddst[f]: p.free()


@pytest.mark.gpu
@pytest.mark.parametrize("src_fmt,dst_fmt", [("nv12", "rgb24"), ("yuv420p", "bgra")])
def test_strip_kernel_4k_full_size(dev, orc, strip_rows, src_fmt, dst_fmt):
    """BASELINE configs[2] at full size, AVHWFramesContext row alignment: 20 frames in one launch (the walker), and 3 (the
    block-cooperative form of launches of up to twelve 4K frames on 256 compute units: round 5's launch-size rule — its boundary, by the device's
    compute units, is tests/test_fullsize_gpu.py's; the counts here sit far from it)"""
    if dev.kind != "hip":
        pytest.skip("full size: GPU only")
    strip_rows(0)
    k = _run_batch(dev, orc, src_fmt, dst_fmt, 3840, 2160, 1920, 1080, nframes=20, nstreams=1, align=256)
    assert k == "scale_yuv2s_kernel", k
    k = _run_batch(dev, orc, src_fmt, dst_fmt, 3840, 2160, 1920, 1080, nframes=3, nstreams=1, align=256)
    assert k == "scale_yuv2s_blk_kernel", k
