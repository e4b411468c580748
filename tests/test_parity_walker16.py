"""The polyphase band walker over 16-BIT SAMPLES (k_scale_yuvg16.hip = k_scale_yuvg.hip compiled with two bytes a sample; round 5): P010LE / P016LE /
YUV420P10LE / YUV420P16LE sources at the walker's ratios into packed RGB, 8-bit 4:2:0 (libswscale's ordered dither of a deeper source) and 10-bit
4:2:0 of the same chroma layout — and the 8-bit walker's new 10-bit destinations (NV12 -> P010LE, YUV420P -> YUV420P10LE).  Against ONE libswscale
context (the oracle: hScale16To15_c swscale.c:93-119, yuv2planeX_8_c with ff_dither_8x8_128 swscale.c:263-264 / 482-485, yuv2p010lX_c / cX_c and
yuv2planeX_10_c output.c:459-519) bit for bit; before it these contexts ran the lines form's two passes or the tiled kernel (0.12 - 0.24 of the roofline)."""
import ctypes as C
import os

import numpy as np
import pytest

from harness import PIX_FMT, SWS, ints, planes, synth_planes
from test_batch_api import _run_batch

W16 = ("scale_yuvg16_kernel", "scale_yuvg16_blk_kernel")
W8 = ("scale_yuvg_kernel", "scale_yuvg_blk_kernel")


def _deep(orc, fmt, sw, sh, seed, valid=True):
    src = synth_planes(orc, fmt, sw, sh, seed=seed)
    if valid and fmt == "yuv420p10le":                     # valid input: 10 significant bits in the low end
        for p in src:
            p.view("<u2")[...] &= 0x3FF
    if valid and fmt == "p010le":                          # ... in the high end
        for p in src:
            p.view("<u2")[...] &= 0xFFC0
    return src


def _check(dev, orc, sf, df, geom, flags="bicubic", align=256, seed=23, valid=True, src_align=256):
    sw, sh, dw, dh = geom
    src = _deep(orc, sf, sw, sh, seed, valid)
    want = orc.sws(src, sw, sh, sf, dw, dh, df, SWS[flags])
    d = dev.upload_planes(src, src_align)
    got, pads, k = dev.sws(d, sw, sh, sf, dw, dh, df, SWS[flags], dst_align=align)
    for p in d:
        p.free()
    for i, (g, w) in enumerate(zip(got, want)):
        bad = np.argwhere(g != w)
        assert bad.size == 0, f"{k} {sf}->{df} {geom} plane {i}: {len(bad)} mismatching bytes, first at {bad[:6].tolist()}"
        assert (pads[i] == 0xCD).all(), f"{k} plane {i}: wrote into the row padding"
    return k


@pytest.fixture(params=["walk", "blk"])
def form(request, monkeypatch):
    """both forms of the walker at every launch size: the band walker proper and the block-cooperative form of short launches"""
    monkeypatch.setenv("GMAT_STRIP_BLOCK", "0" if request.param == "walk" else "32")
    return request.param


PAIRS16 = [("p010le", "p010le"), ("p010le", "nv12"), ("p010le", "rgb24"), ("p010le", "bgra"), ("p016le", "nv12"), ("p016le", "p010le"), ("p016le", "bgr24"),
           ("yuv420p10le", "yuv420p10le"), ("yuv420p10le", "yuv420p"), ("yuv420p10le", "rgb24"), ("yuv420p16le", "yuv420p"), ("yuv420p16le", "yuv420p10le"),
           ("yuv420p16le", "rgba")]
# 2.4 : 1 (11 taps), 3 : 1, 3 : 2, 4 : 1, 1.3 : 1, 5 : 1 (21 taps behind up to three leading zeros: 12 of the 13 pairs), anamorphic, odd destination sizes, several 64-column strips with a partial last one
GEOMS = [(384, 216, 160, 90), (768, 96, 256, 32), (384, 216, 256, 144), (1024, 64, 256, 16), (400, 240, 308, 184), (1280, 96, 256, 16), (640, 96, 200, 64),
         (520, 100, 173, 41), (2048, 40, 700, 16)]


@pytest.mark.parametrize("pair", PAIRS16, ids=lambda p: "%s-%s" % p)
@pytest.mark.parametrize("geom", GEOMS, ids=lambda g: "%dx%d-%dx%d" % g)
def test_walker16_every_layout(dev, orc, form, pair, geom):
    k = _check(dev, orc, pair[0], pair[1], geom)
    if (geom[2] & 1) and pair[1] in ("rgb24", "bgr24", "rgba", "bgra"):
        assert k.startswith("scale_yuv"), k              # an RGB destination of odd width is not the walker's (8 bits or 16): whatever serves it, the bytes
    elif form == "walk":
        assert k == "scale_yuvg16_kernel", k
    else:
        assert k in W16, k                               # (a context whose bands fit no block keeps the walker)


@pytest.mark.parametrize("flags", ["bilinear", "bicubic", "lanczos", "area", "gauss", "spline", "point", "sinc"])
@pytest.mark.parametrize("pair", [("p010le", "p010le"), ("yuv420p10le", "rgb24"), ("p016le", "nv12")], ids=lambda p: "%s-%s" % p)
def test_walker16_every_algorithm(dev, orc, form, pair, flags):
    """every SWS algorithm whose tables fit the walker (the others: whatever serves them — the bytes either way)"""
    for geom in ((384, 216, 160, 90), (640, 128, 420, 84)):
        _check(dev, orc, pair[0], pair[1], geom, flags)


def test_walker16_garbage_in_the_unused_bits(dev, orc, form):
    """libswscale shifts a P010 sample right by 6 whatever its low bits hold: frames that break the format's promise (random low bits) still come out as
    one libswscale context makes them.  (A planar 10-bit sample with high bits set is read as the 16 bits it is and overflows hScale16To15_c's int: no
    kernel here follows that.)"""
    for pair in (("p010le", "p010le"), ("p010le", "rgb24"), ("p010le", "nv12")):
        assert _check(dev, orc, pair[0], pair[1], (384, 216, 160, 90), valid=False) in W16


@pytest.mark.parametrize("pair", [("nv12", "p010le"), ("yuv420p", "yuv420p10le")], ids=lambda p: "%s-%s" % p)
@pytest.mark.parametrize("geom", [(384, 216, 160, 90), (768, 96, 256, 32), (384, 216, 256, 144), (520, 100, 173, 41)], ids=lambda g: "%dx%d-%dx%d" % g)
def test_walker8_writes_ten_bit_destinations(dev, orc, form, pair, geom):
    """the 8-bit walker's output stage in its 10-bit form: NV12 -> P010LE (a hardware decoder's frame for a 10-bit encoder), YUV420P -> YUV420P10LE"""
    assert _check(dev, orc, pair[0], pair[1], geom) in W8


@pytest.mark.parametrize("pair", [("p010le", "p010le"), ("p010le", "nv12"), ("yuv420p10le", "rgb24"), ("p016le", "bgra")], ids=lambda p: "%s-%s" % p)
def test_walker16_batches_and_streams(dev, orc, pair, monkeypatch):
    """n frames = one launch (grid.y = frame), shares of a batch on two streams; both forms by launch size"""
    monkeypatch.setenv("GMAT_STRIP_BLOCK", "3")
    assert _run_batch(dev, orc, pair[0], pair[1], 384, 216, 160, 90, nframes=5, nstreams=1, align=256) == "scale_yuvg16_kernel"
    assert _run_batch.last_frames == 5
    assert _run_batch(dev, orc, pair[0], pair[1], 384, 216, 160, 90, nframes=3, nstreams=1, align=256) == "scale_yuvg16_blk_kernel"
    assert _run_batch(dev, orc, pair[0], pair[1], 384, 216, 160, 90, nframes=9, nstreams=2, align=256) == "scale_yuvg16_kernel"


def test_walker16_pitches_and_band_heights(dev, orc, monkeypatch):
    """destination pitches that are multiples of 4 only, source pitches of 64; bands of 4 ... 40 rows (GMAT_STRIP_ROWS) walking down and up"""
    monkeypatch.setenv("GMAT_STRIP_BLOCK", "0")
    for rows in (4, 7, 16, 40):
        monkeypatch.setenv("GMAT_STRIP_ROWS", str(rows))
        for pair in (("p010le", "p010le"), ("yuv420p10le", "yuv420p"), ("p016le", "rgb24")):
            assert _check(dev, orc, pair[0], pair[1], (520, 200, 172, 66), align=4, src_align=64) == "scale_yuvg16_kernel"
    monkeypatch.setenv("GMAT_STRIP_UPDOWN", "0")
    assert _check(dev, orc, "p010le", "nv12", (520, 200, 172, 66)) == "scale_yuvg16_kernel"


def test_walker16_knob_and_what_it_leaves_alone(dev, orc, monkeypatch):
    """GMAT_SCALE_NO_WALKER16=1: the lines form / tiled kernel as before; 2 : 1 keeps its own kernel; a range conversion and 4:4:4 ends are not the walker's"""
    assert _check(dev, orc, "p010le", "p010le", (384, 216, 192, 108)).startswith("scale_yuv2p")
    monkeypatch.setenv("GMAT_SCALE_NO_WALKER16", "1")
    k = _check(dev, orc, "p010le", "p010le", (384, 216, 160, 90))
    assert k not in W16 and k.startswith("scale_yuv"), k


# ---- YUV444P -> YUV444P on the 8-bit band walker (round 5, last): three plane jobs whose chroma tables are the full-size ones ------------------------------------
@pytest.mark.parametrize("geom", GEOMS + [(384, 216, 192, 108), (256, 144, 384, 216), (200, 120, 260, 150)], ids=lambda g: "%dx%d-%dx%d" % g)
def test_walker8_planar_444_at_both_ends(dev, orc, form, geom):
    """a format of scale_cuda's list (vf_scale_cuda.c:45-54): it ran the lines form's two passes (0.12 of the roofline: 1080p -> 720p 9.5 us a frame) or, alone, the
    tiled kernel (14 us); the walker's plane jobs take sizes and tables as they come, so 4:4:4 is an acceptance rule, not a kernel"""
    up = geom[2] > geom[0]
    for flags in ("bicubic", "bilinear", "lanczos"):
        k = _check(dev, orc, "yuv444p", "yuv444p", geom, flags)
        if not up and (flags == "bicubic" or geom[0] < 4 * geom[2]):      # (from 4 : 1 on the short filters' 64 columns do not fit one dword a lane, Lanczos' taps its pairs: the lines form)
            assert k in W8, (flags, k)
    if not up:
        assert _check(dev, orc, "yuv444p", "yuv444p", geom, align=4, src_align=4) in W8
        assert _run_batch(dev, orc, "yuv444p", "yuv444p", *geom, nframes=5, nstreams=1, align=256) in W8


# ---- YUV444P -> YUV420P on the same plane jobs (round 6): the chroma jobs' source planes are the full-size ones, their destinations the halved ones -----------------------
@pytest.mark.parametrize("geom", [(384, 216, 256, 144), (384, 216, 192, 108), (640, 128, 420, 84), (520, 100, 300, 60), (384, 216, 384, 216)], ids=lambda g: "%dx%d-%dx%d" % g)
def test_walker8_planar_444_into_420(dev, orc, form, geom):
    """swscale_cuda's YUV444P source into YUV420P (swscale_cuda.c:34-44): the format sweep (profiles/r06_sweep_before.txt) found it on the lines form's two launches, 0.126 of
    the roofline; the luma job at the frame's ratio, the chroma jobs at twice it — an acceptance rule.  (Semi-planar ends and 4:2:0 -> 4:4:4, whose chroma is an UP-scale
    beside a luma down-scale, stay where they were.)"""
    for flags in ("bicubic", "bilinear"):
        k = _check(dev, orc, "yuv444p", "yuv420p", geom, flags)
        if flags == "bicubic" or geom[0] < 2 * geom[2]:             # (the chroma jobs run at twice the frame's ratio: from 4 : 1 on the short filters' 64 columns do not fit one dword a lane — the lines form)
            assert k in W8, (flags, k)
    assert _check(dev, orc, "yuv444p", "yuv420p", geom, "lanczos") is not None
    assert _check(dev, orc, "yuv444p", "yuv420p", geom, align=4, src_align=4) in W8
    assert _run_batch(dev, orc, "yuv444p", "yuv420p", *geom, nframes=5, nstreams=1, align=256) in W8
    for other in (("yuv420p", "yuv444p"), ("yuv444p", "nv12"), ("nv12", "yuv444p")):           # bit-exact wherever they land
        assert _check(dev, orc, other[0], other[1], geom) is not None


# ---- packed RGB sources into 4:2:0 frames: the walker's own converter in front of the same 16-bit lines (round 5) -----------------------------------
RGBSRC_GEOMS = [(384, 216, 256, 144), (768, 96, 256, 32), (384, 216, 160, 90), (640, 128, 420, 84), (1024, 64, 256, 16), (520, 100, 172, 40), (2048, 40, 700, 16),
                (384, 216, 380, 212)]


FUSED = "scale_yuvg_rgb2p_blk_kernel"     # luma and chroma of a band behind ONE load of the pixels, a pair's rows converted side by side: the rule for 8-bit 4:2:0 frames


@pytest.fixture(params=["walk", "blk", "fused"])
def rgbp(request, monkeypatch):
    """the three forms an RGB source into a 4:2:0 frame has, each at every launch size: the band walker's plane jobs, their block-cooperative form, and the
    fused block form (8-bit destinations, filters of up to 16 coefficient pairs: the shipped rule below 1.75 : 1, from three frames a launch on, and for every up-scale)"""
    if request.param != "fused":
        monkeypatch.setenv("GMAT_RGBSRC_FUSED", "0")
        monkeypatch.setenv("GMAT_STRIP_BLOCK", "0" if request.param == "walk" else "32")
    else:
        monkeypatch.setenv("GMAT_RGBSRC_FUSED", "1")       # (the shipped rule: from three frames a launch on beyond 1.75 : 1)
    return request.param


def _rgbp_name(rgbp, df, geom):
    if rgbp == "fused" and df != "p010le" and geom != (1024, 64, 256, 16):       # (4 : 1: ten coefficient pairs)
        return (FUSED,)
    return ("scale_yuvg16_kernel",) if rgbp == "walk" else W16


@pytest.mark.parametrize("sf", ["rgb24", "bgr24"])
@pytest.mark.parametrize("df", ["nv12", "yuv420p", "p010le"])
@pytest.mark.parametrize("geom", RGBSRC_GEOMS, ids=lambda g: "%dx%d-%dx%d" % g)
def test_walker16_packed_rgb_sources(dev, orc, rgbp, sf, df, geom):
    """RGB24 / BGR24 -> NV12 / YUV420P (and their 10-bit twins) at the walker's down-scale ratios — the frames a network writes, scaled for an encoder.  One
    libswscale context: rgb24ToY_c on every pixel, rgb24ToUV_half_c on horizontal pixel pairs at full height (input.c:815-866), hScale16To15_c with sh = 13,
    the planar vertical stage.  These contexts ran the tiled kernel at 0.07 - 0.12 of the roofline (rgb24 4K -> 720p nv12: 42 us a frame)."""
    k = _check(dev, orc, sf, df, geom)
    assert k in _rgbp_name(rgbp, df, geom), k


@pytest.mark.parametrize("geom", [(384, 216, 258, 146), (520, 100, 174, 42), (1280, 40, 1000, 30), (644, 60, 322, 40), (388, 216, 130, 70), (640, 128, 422, 85)],
                         ids=lambda g: "%dx%d-%dx%d" % g)
def test_rgb_to_420_fused_edges(dev, orc, geom, monkeypatch):
    """the fused block form at the edges: partial last blocks (luma and chroma), odd chroma sizes, widths that are multiples of four but not of eight, odd
    destination heights (a last band whose chroma rows end first), rows at pitches that are not multiples of 64"""
    monkeypatch.setenv("GMAT_RGBSRC_FUSED", "1")
    for sf, df in (("rgb24", "nv12"), ("bgr24", "yuv420p")):
        assert _check(dev, orc, sf, df, geom) == FUSED
        assert _check(dev, orc, sf, df, geom, align=4, src_align=4) == FUSED


@pytest.mark.parametrize("flags", ["bilinear", "lanczos", "area", "gauss", "point", "spline", "sinc"])
def test_walker16_packed_rgb_sources_algorithms(dev, orc, rgbp, flags):
    for df in ("nv12", "yuv420p"):
        _check(dev, orc, "rgb24", df, (640, 128, 420, 84), flags)
        _check(dev, orc, "bgr24", df, (384, 216, 160, 90), flags)


UP_RGB = [(256, 144, 384, 216), (128, 72, 384, 216), (160, 90, 640, 360), (96, 64, 700, 500), (200, 120, 260, 150), (1280, 40, 1600, 50), (644, 60, 1284, 90),
          (132, 76, 200, 115)]


@pytest.mark.parametrize("geom", UP_RGB, ids=lambda g: "%dx%d-%dx%d" % g)
@pytest.mark.parametrize("flags", ["bicubic", "bilinear", "lanczos"])
def test_rgb_to_420_up_scales(dev, orc, geom, flags):
    """UP-scales of an RGB source into NV12 / YUV420P: the chroma of every pixel (rgb24ToUV_c: chrSrcHSubSample stays 0 once the destination's chroma is wider
    than half the source, utils.c:1529-1545), any factor — the fused block form alone (no walker instance); before: the tiled kernel (720p -> 1080p 9.8 us)"""
    for sf, df in (("rgb24", "nv12"), ("bgr24", "yuv420p")):
        assert _check(dev, orc, sf, df, geom, flags) == FUSED
    assert _check(dev, orc, "rgb24", "yuv420p", geom, flags, align=4, src_align=4) == FUSED


def test_walker16_packed_rgb_what_it_leaves_alone(dev, orc, rgbp):
    """up-scales into 10-bit frames, odd widths, 2 : 1 (its own kernel), RGB destinations: not these forms"""
    ours = W16 + (FUSED,)
    assert _check(dev, orc, "rgb24", "p010le", (256, 144, 384, 216)) not in ours
    assert _check(dev, orc, "rgb24", "nv12", (256, 144, 384, 216)) == FUSED          # (whatever the knobs say: an up-scale has no other form here)
    assert _check(dev, orc, "rgb24", "nv12", (386, 216, 160, 90)) == FUSED               # (a width that is not a multiple of four: the fused block form alone)
    assert _check(dev, orc, "rgb24", "nv12", (385, 216, 160, 90)) not in ours
    assert _check(dev, orc, "rgb24", "nv12", (512, 64, 256, 32)) == "scale_rgb2y_kernel"
    assert _check(dev, orc, "rgb24", "rgb24", (384, 216, 160, 90)) not in ours


def test_walker16_packed_rgb_batches(dev, orc, monkeypatch):
    """the shipped rule: below 1.75 : 1 (four pixels a lane) and for every up-scale the fused block form always; beyond (eight pixels a lane) from three
    frames a launch on, plane jobs (their block form) below"""
    for df in ("nv12", "yuv420p"):
        for n in (1, 2, 3, 5, 35):
            k = _run_batch(dev, orc, "rgb24", df, 384, 216, 160, 90, nframes=n, nstreams=1, align=256)
            assert k == (FUSED if n >= 3 else "scale_yuvg16_blk_kernel"), (n, k)
        assert _run_batch(dev, orc, "rgb24", df, 384, 216, 160, 90, nframes=7, nstreams=2, align=64) == FUSED      # (launches of four and three frames)
        for n in (1, 2, 4):
            assert _run_batch(dev, orc, "bgr24", df, 384, 216, 256, 144, nframes=n, nstreams=1, align=256) == FUSED
        assert _run_batch(dev, orc, "bgr24", df, 160, 90, 384, 216, nframes=2, nstreams=1, align=64) == FUSED
    monkeypatch.setenv("GMAT_RGBSRC_FUSED", "1")
    for rows in (4, 8, 12, 20, 64):
        monkeypatch.setenv("GMAT_STRIP_ROWS", str(rows))
        assert _check(dev, orc, "rgb24", "nv12", (520, 200, 172, 66), align=4, src_align=64) == FUSED
        assert _check(dev, orc, "bgr24", "yuv420p", (520, 200, 300, 134), align=4, src_align=64) == FUSED
    monkeypatch.delenv("GMAT_STRIP_ROWS")
    monkeypatch.setenv("GMAT_RGBSRC_FUSED", "0")
    monkeypatch.setenv("GMAT_STRIP_BLOCK", "3")
    for df in ("nv12", "yuv420p"):
        assert _run_batch(dev, orc, "rgb24", df, 384, 216, 160, 90, nframes=5, nstreams=2, align=256) in W16
        assert _run_batch(dev, orc, "bgr24", df, 384, 216, 256, 144, nframes=6, nstreams=1, align=64) == "scale_yuvg16_kernel"


# ---- packed RGB -> packed RGB away from 2 : 1 (round 5): the BASELINE's literal second stage at any ratio -----------------------------------------------------
# scale_yuvg_rgbsrc_blk_kernel: the block-cooperative form — every launch size, up-scales of any factor, filters up to 16 coefficient pairs (3.3 : 1);
# scale_yuvg_rgbsrc_kernel: the band walker's form behind it (ratios up to 6 : 1 from four frames a launch on)
RGBRGB = "scale_yuvg_rgbsrc_kernel"
RGBBLK = "scale_yuvg_rgbsrc_blk_kernel"
RGB_GEOMS = [(384, 216, 256, 144), (768, 96, 256, 32), (384, 216, 160, 90), (640, 128, 420, 84), (1024, 64, 256, 16), (520, 100, 172, 40),
             (256, 144, 384, 216), (400, 100, 380, 96), (2048, 40, 700, 16), (384, 216, 161, 91)]


@pytest.fixture(params=["walk", "blk"])
def rgbform(request, monkeypatch):
    """both forms at every launch size (the shipped rule: the block form wherever it has an instance)"""
    if request.param == "walk":
        monkeypatch.setenv("GMAT_RGBSRC_WALKER", "2")
        monkeypatch.setenv("GMAT_RGBSRC_BLOCK", "0")
    return request.param


@pytest.mark.parametrize("sf", ["rgb24", "bgr24"])
@pytest.mark.parametrize("df", ["rgb24", "bgr24", "rgba", "bgra"])
@pytest.mark.parametrize("geom", RGB_GEOMS, ids=lambda g: "%dx%d-%dx%d" % g)
def test_rgb_to_rgb_any_ratio(dev, orc, rgbform, sf, df, geom):
    """rgb24ToY_c + rgb24ToUV_c / rgb24ToUV_half_c (from 2 : 1 on), hScale16To15_c (sh = 13) of the three lines, yuv2rgb_full_X_c + yuv2rgb_write_full: one
    libswscale context, bit for bit, down- and up-scales, odd destination sizes; before: the tiled kernel of round 1 (rgb24 1080p -> 720p 14 us a frame, 0.08)."""
    k = _check(dev, orc, sf, df, geom)
    if rgbform == "blk" and geom == (1024, 64, 256, 16):       # (4 : 1 needs ten coefficient pairs: the walker's form only, from four frames a launch on)
        assert k.startswith("scale_rgb_kernel"), k
    else:
        assert k == (RGBRGB if rgbform == "walk" else RGBBLK), k


@pytest.mark.parametrize("geom", [(128, 72, 384, 216), (160, 90, 640, 360), (96, 64, 700, 500), (200, 120, 260, 150), (1280, 40, 1600, 50), (644, 60, 1284, 90)],
                         ids=lambda g: "%dx%d-%dx%d" % g)
def test_rgb_to_rgb_block_form_up_scales(dev, orc, geom):
    """up-scales of any factor: every output row is a gather down the LDS columns of filtered row pairs, so the number of rows open at once (the walker's
    limit: 12) does not matter; several 64-column blocks with a partial last one, bands cut by the plane's end"""
    for sf, df in (("rgb24", "rgb24"), ("bgr24", "rgba")):
        assert _check(dev, orc, sf, df, geom) == RGBBLK
    assert _check(dev, orc, "rgb24", "bgr24", geom, "lanczos") == RGBBLK


@pytest.mark.parametrize("geom", [(256, 144, 384, 216), (160, 90, 640, 360), (640, 640, 1920, 1080), (384, 216, 256, 216), (256, 100, 384, 100), (384, 216, 160, 90)],
                         ids=lambda g: "%dx%d-%dx%d" % g)
@pytest.mark.parametrize("flags", ["bilinear", "point", "area"])
def test_rgb_to_rgb_one_and_two_tap_vertical_filters(dev, orc, geom, flags):
    """packed_vscale's special forms (vscale.c:135-160): ONE vertical tap — equal heights, SWS_POINT: yuv2rgb_full_1_c takes the line as it is; TWO taps that
    are a proper blend — bilinear up-scales: yuv2rgb_full_2_c, no rounding constant (output.c:2118-2120).  They differ from yuv2rgb_full_X_c in the sums'
    start alone; the block form reads it per output row (the walker's form needs three taps or more: running sums from one start)"""
    assert _check(dev, orc, "rgb24", "bgra", geom, flags) == RGBBLK
    assert _check(dev, orc, "bgr24", "rgb24", geom, flags, align=4, src_align=4) == RGBBLK


@pytest.mark.parametrize("geom", [(384, 216, 256, 144), (384, 216, 160, 90), (256, 144, 384, 216), (520, 100, 172, 40), (200, 120, 260, 150), (384, 216, 161, 91),
                                  (256, 100, 384, 100)], ids=lambda g: "%dx%d-%dx%d" % g)
@pytest.mark.parametrize("pair", [("rgba", "rgb24"), ("bgra", "bgra"), ("rgba", "bgra"), ("bgra", "bgr24")], ids=lambda p: "%s-%s" % p)
def test_rgba_sources_are_read_as_they_are(dev, orc, pair, geom, monkeypatch):
    """RGBA / BGRA sources (rgb32ToY / ToUV: the 24-bit readers' coefficients on the same three channels): the block form reads four-byte pixels itself — no
    32 -> 24-bit pass in front — and, with an alpha channel at both ends, scales alpha as a FOURTH line (rgbaToA_c's a << 6 | a >> 2, the luma filters on both
    axes, yuv2rgb_full_X_c's (2^18 + sum) >> 19: needAlpha, utils.c:1902) instead of two more passes behind.  The same bytes with GMAT_RGBSRC_NO_PX4=1"""
    sf, df = pair
    assert _check(dev, orc, sf, df, geom) == RGBBLK
    assert _check(dev, orc, sf, df, geom, "bilinear" if geom[0] >= 2 * geom[2] else "lanczos", align=4, src_align=4) == RGBBLK
    monkeypatch.setenv("GMAT_RGBSRC_NO_PX4", "1")
    _check(dev, orc, sf, df, geom)


@pytest.mark.parametrize("geom", [(384, 216, 256, 144), (384, 216, 160, 90), (256, 144, 384, 216), (520, 100, 172, 40), (388, 216, 130, 70), (640, 128, 422, 85)],
                         ids=lambda g: "%dx%d-%dx%d" % g)
@pytest.mark.parametrize("pair", [("rgba", "nv12"), ("bgra", "yuv420p"), ("bgra", "nv12")], ids=lambda p: "%s-%s" % p)
def test_rgba_sources_into_420_are_read_as_they_are(dev, orc, pair, geom, monkeypatch):
    """... and into NV12 / YUV420P: the fused block form reads the four-byte pixels (at every launch size: no other entry of the plane scaler's table does)"""
    sf, df = pair
    assert _check(dev, orc, sf, df, geom) == FUSED
    assert _check(dev, orc, sf, df, geom, "bilinear", align=4, src_align=4) == FUSED
    monkeypatch.setenv("GMAT_RGBSRC_NO_PX4", "1")
    _check(dev, orc, sf, df, geom)


@pytest.mark.parametrize("geom", [(386, 216, 256, 144), (1366, 76, 640, 36), (854, 48, 426, 24), (426, 60, 854, 120), (390, 100, 172, 40), (202, 120, 260, 150),
                                  (342, 60, 1284, 90), (394, 216, 161, 91), (385, 60, 500, 80)], ids=lambda g: "%dx%d-%dx%d" % g)
@pytest.mark.parametrize("trio", [("rgb24", "bgr24", RGBBLK), ("bgr24", "nv12", FUSED), ("rgb24", "yuv420p", FUSED), ("bgra", "bgra", RGBBLK), ("rgba", "nv12", FUSED)],
                         ids=lambda t: "%s-%s" % t[:2])
def test_rgb_sources_of_any_width(dev, orc, trio, geom):
    """source widths that are not multiples of four (1366, 854, 426 ...; odd ones where the chroma comes from every pixel): the block-cooperative kernels read
    whole dwords of a row through a buffer resource whose size is rounded up to a dword — the last pixels' dword lies in the page of the row's last byte —
    and every sample past the row's end meets a zero coefficient.  The walker's streams (groups of four / eight pixels checked against the exact size) decline"""
    sf, df, k = trio
    assert _check(dev, orc, sf, df, geom) == k
    assert _check(dev, orc, sf, df, geom, "bilinear", align=4, src_align=4) == k


@pytest.mark.parametrize("geom", [(512, 64, 256, 32), (384, 216, 192, 108), (520, 100, 260, 50), (640, 40, 320, 20)], ids=lambda g: "%dx%d-%dx%d" % g)
@pytest.mark.parametrize("trio", [("bgra", "bgra", RGBBLK), ("rgba", "rgb24", RGBBLK), ("bgra", "nv12", FUSED), ("rgba", "yuv420p", FUSED)], ids=lambda t: "%s-%s" % t[:2])
def test_rgba_sources_at_exactly_two_to_one(dev, orc, trio, geom):
    """exactly 2 : 1 belongs to the strip kernels of k_scale_rgb2s.hip — which read three-byte pixels; an RGBA / BGRA source takes the block-cooperative kernels there
    too rather than a 32 -> 24-bit pass in front of them (and a batch of such frames stays a batch); rgb24 / bgr24 keep the strip kernels"""
    sf, df, k = trio
    assert _check(dev, orc, sf, df, geom) == k
    assert _check(dev, orc, sf, df, geom, "bilinear", align=16, src_align=16) == k
    if geom == (512, 64, 256, 32):
        assert _check(dev, orc, "rgb24", "rgb24", geom).startswith("scale_rgb2")
        assert _check(dev, orc, "rgb24", "nv12", geom) == "scale_rgb2y_kernel"
        assert _run_batch(dev, orc, sf, df, *geom, nframes=5, nstreams=1, align=256) == k
        assert _run_batch(dev, orc, "rgb24", "rgb24", *geom, nframes=5, nstreams=1, align=256).startswith("scale_rgb2")


def test_rgba_to_rgba_at_the_largest_block(dev, orc):
    """the block form's largest footprint: four lines (alpha), chroma from every pixel at eight pixels a lane (1.9 : 1), eight row pairs a wave and 16-pair vertical
    windows (7 : 1) — 16 KB of row images + 48 KB of filtered pairs = the 64 KB a workgroup may hold; a taller window is declined"""
    assert _check(dev, orc, "bgra", "bgra", (760, 700, 400, 100)) == RGBBLK
    assert _check(dev, orc, "rgba", "bgra", (760, 350, 400, 50)) == RGBBLK
    assert _check(dev, orc, "bgra", "bgra", (764, 720, 400, 90)).startswith("scale_rgb_kernel")


def test_rgba_sources_batches(dev, orc):
    for n in (2, 5, 34):
        assert _run_batch(dev, orc, "bgra", "nv12", 384, 216, 160, 90, nframes=n, nstreams=1, align=256) == FUSED
    assert _run_batch(dev, orc, "rgba", "yuv420p", 256, 144, 384, 216, nframes=3, nstreams=2, align=64) == FUSED

    for n in (2, 5, 35):
        assert _run_batch(dev, orc, "bgra", "bgra", 384, 216, 256, 144, nframes=n, nstreams=1, align=256) == RGBBLK
    assert _run_batch(dev, orc, "rgba", "rgb24", 384, 216, 160, 90, nframes=5, nstreams=2, align=64) == RGBBLK
    assert _run_batch(dev, orc, "rgba", "bgra", 256, 144, 384, 216, nframes=3, nstreams=1, align=64) == RGBBLK


@pytest.mark.parametrize("flags", ["bilinear", "lanczos", "area", "gauss", "spline", "sinc", "bicublin", "x"])
def test_rgb_to_rgb_algorithms(dev, orc, rgbform, flags):
    for geom in ((640, 128, 420, 84), (384, 216, 160, 90)):
        _check(dev, orc, "rgb24", "rgb24", geom, flags)


def test_rgb_to_rgb_what_it_leaves_alone(dev, orc, rgbform, monkeypatch):
    """exactly 2 : 1 (its own strip walker), odd widths under pixel pairs, SWS_FAST_BILINEAR (an RGB source keeps its halved chroma: the plane
    scaler), two-tap vertical filters on the walker's form, the knobs"""
    both = (RGBRGB, RGBBLK)
    assert _check(dev, orc, "rgb24", "rgb24", (512, 64, 256, 32)).startswith("scale_rgb2")
    assert _check(dev, orc, "rgb24", "rgb24", (386, 216, 160, 90)) == RGBBLK            # (a width that is not a multiple of four: the block form alone)
    assert _check(dev, orc, "rgb24", "rgb24", (385, 216, 160, 90)) not in both           # (pixel pairs of an odd width: the tiled kernel)
    assert _check(dev, orc, "rgb24", "rgb24", (384, 216, 256, 144), "fast_bilinear") not in both
    assert _check(dev, orc, "rgb24", "rgb24", (256, 144, 384, 216), "bilinear") == RGBBLK         # (whatever the knobs say: the walker's form has no instance)
    monkeypatch.setenv("GMAT_SCALE_NO_WALKER16", "1")
    assert _check(dev, orc, "rgb24", "rgb24", (384, 216, 256, 144)).startswith("scale_rgb_kernel")
    monkeypatch.delenv("GMAT_SCALE_NO_WALKER16")
    monkeypatch.setenv("GMAT_RGBSRC_WALKER", "0")
    assert _check(dev, orc, "rgb24", "rgb24", (384, 216, 256, 144)).startswith("scale_rgb_kernel")


def test_rgb_to_rgb_batches_and_bands(dev, orc, monkeypatch):
    """the shipped rule: the block form at every launch size; the walker's form where the block form has no instance (4 : 1: ten coefficient pairs) from
    four frames a launch on, the tiled kernel below"""
    for n in (1, 3, 4, 35):
        assert _run_batch(dev, orc, "rgb24", "rgb24", 384, 216, 256, 144, nframes=n, nstreams=1, align=256) == RGBBLK
    assert _run_batch(dev, orc, "bgr24", "bgra", 384, 216, 160, 90, nframes=5, nstreams=2, align=64) == RGBBLK
    assert _run_batch(dev, orc, "rgb24", "rgb24", 1024, 64, 256, 16, nframes=3, nstreams=1, align=256).startswith("scale_rgb_kernel")
    assert _run_batch(dev, orc, "rgb24", "rgb24", 1024, 64, 256, 16, nframes=4, nstreams=1, align=256) == RGBRGB
    for rows in (4, 7, 12, 40):
        monkeypatch.setenv("GMAT_STRIP_ROWS", str(rows))
        assert _check(dev, orc, "rgb24", "bgr24", (520, 200, 172, 66), align=4, src_align=64) == RGBBLK
        assert _check(dev, orc, "rgb24", "rgba", (200, 120, 520, 310), align=4, src_align=64) == RGBBLK
    monkeypatch.setenv("GMAT_RGBSRC_WALKER", "2")
    monkeypatch.setenv("GMAT_RGBSRC_BLOCK", "0")
    assert _run_batch(dev, orc, "rgb24", "rgb24", 384, 216, 256, 144, nframes=5, nstreams=2, align=256) == RGBRGB
    assert _run_batch(dev, orc, "bgr24", "bgra", 384, 216, 160, 90, nframes=35, nstreams=1, align=64) == RGBRGB
    for rows in (4, 7, 40):
        monkeypatch.setenv("GMAT_STRIP_ROWS", str(rows))
        assert _check(dev, orc, "rgb24", "bgr24", (520, 200, 172, 66), align=4, src_align=64) == RGBRGB


# ---- the quad-lane walker over 16-bit samples (k_scale_yuvu16.hip, round 5): up-scales of any factor, short-filter down-scales ------------------------------
QUAD16 = "scale_yuvu16_kernel"
UP_GEOMS = [(160, 90, 240, 136), (128, 72, 384, 216), (320, 180, 384, 216), (200, 120, 600, 330), (256, 64, 1024, 256), (132, 76, 200, 114)]


@pytest.mark.parametrize("pair", PAIRS16, ids=lambda p: "%s-%s" % p)
@pytest.mark.parametrize("geom", UP_GEOMS, ids=lambda g: "%dx%d-%dx%d" % g)
def test_quad16_up_scales(dev, orc, pair, geom):
    """P010 / P016 / planar 10- and 16-bit 4:2:0 sources up-scaled by any factor: a lane owns four adjacent outputs, the vertical filter is a gather over a register
    ring (k_scale_yuvu.hip's structure), the horizontal stage reads 8-byte aligned windows of 16-bit samples (no v_perm_b32 for a plane)"""
    k = _check(dev, orc, pair[0], pair[1], geom)
    assert k == QUAD16 or ((geom[2] & 1) and pair[1] in ("rgb24", "bgr24", "rgba", "bgra")), k


@pytest.mark.parametrize("flags", ["bilinear", "lanczos", "point", "area", "gauss", "spline"])
def test_quad16_algorithms(dev, orc, flags):
    for pair in (("p010le", "p010le"), ("yuv420p10le", "rgb24"), ("p016le", "nv12")):
        _check(dev, orc, pair[0], pair[1], (160, 90, 240, 136), flags)
        _check(dev, orc, pair[0], pair[1], (128, 72, 384, 216), flags)


@pytest.mark.parametrize("pair", [("nv12", "p010le"), ("yuv420p", "yuv420p10le")], ids=lambda p: "%s-%s" % p)
def test_quad8_writes_ten_bit_destinations(dev, orc, pair):
    for geom in UP_GEOMS[:4]:
        assert _check(dev, orc, pair[0], pair[1], geom) == "scale_yuvu_kernel"


def test_quad16_short_filter_down_scales_and_batches(dev, orc, monkeypatch):
    """the shipped rule: up-scales on the vertical axis (the short-filter down-scales the 8-bit kernel takes in launches of more than three frames stay on the band
    walker here: measured slower, profiles/r05t_quad16.txt); GMAT_QUAD_WALKER=2: wherever it is eligible"""
    assert _run_batch(dev, orc, "p010le", "p010le", 384, 216, 288, 162, nframes=5, nstreams=1, align=256) in W16
    monkeypatch.setenv("GMAT_QUAD_WALKER", "2")
    assert _run_batch(dev, orc, "p010le", "p010le", 384, 216, 288, 162, nframes=5, nstreams=1, align=256) == QUAD16
    monkeypatch.delenv("GMAT_QUAD_WALKER")
    assert _run_batch(dev, orc, "yuv420p10le", "yuv420p", 160, 90, 240, 136, nframes=9, nstreams=2, align=64) == QUAD16
    monkeypatch.setenv("GMAT_QUAD_WALKER", "0")
    assert _check(dev, orc, "p010le", "p010le", (160, 90, 240, 136)) in W16
