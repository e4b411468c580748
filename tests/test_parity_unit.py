"""scale19_unit_kernel (k_scale19.hip, round 6): same-size conversions between the YUV depths and layouts — yuv2yuv_cuda's space
(libswscale/cuda/yuv2yuv_cuda.cu:324-366) — where libswscale has no unscaled special converter and runs its generic scaler with one-tap filters
(swscale_unscaled.c: ff_get_unscaled_swscale answers for a few pairs only; everything else is hScale*To15_c / To19_c with one coefficient of 2^14 and
yuv2plane1_* / yuv2nv12cX_*).  Against the oracle's restatement of that path, bit-exact; the same pairs through the tile form (GMAT_S19_UNIT=0) as a second
witness."""
import ctypes as C

import numpy as np
import pytest

from harness import PIX_FMT, SWS, synth_planes, planes, ints, alloc_planes

pytestmark = []

YUV420 = ["nv12", "yuv420p", "p010le", "p016le", "yuv420p10le", "yuv420p16le"]
YUV444 = ["yuv444p", "yuv444p16le"]


def _synth(orc, fmt, w, h, seed):
    src = synth_planes(orc, fmt, w, h, seed=seed)
    if fmt == "yuv420p10le":
        for p in src:
            p.view("<u2")[...] &= 0x3FF
    if fmt == "p010le":
        for p in src:
            p.view("<u2")[...] &= 0xFFC0
    return src


def _run(dev, orc, sf, df, w, h, align, extra, seed=7, flags="bicubic"):
    src = _synth(orc, sf, w, h, seed)
    want = orc.sws(src, w, h, sf, w, h, df, SWS[flags])
    d = dev.upload_planes(src, align, extra)
    got, pads, kernel = dev.sws(d, w, h, sf, w, h, df, SWS[flags], dst_align=align, dst_extra=extra)
    for p in d:
        p.free()
    for i, (g, wv) in enumerate(zip(got, want)):
        bad = np.argwhere(g != wv)
        assert bad.size == 0, f"{sf}->{df} {w}x{h} align {align}+{extra}: plane {i}: {len(bad)} bytes differ, first at {bad[:4].tolist()} ({kernel})"
        assert (pads[i] == 0xCD).all(), (sf, df, w, h, i)
    return kernel


def _pairs():
    out = []
    for group in (YUV420, YUV444):
        for sf in group:
            for df in group:
                if sf != df and not (sf == "yuv420p" and df in ("p010le", "p016le")):    # (planar8ToP01xleWrapper: its own arithmetic and oracle entry, tests/test_parity_rgb2yuv.py)
                    out.append((sf, df))
    return out


@pytest.mark.parametrize("pair", _pairs(), ids=lambda p: f"{p[0]}-{p[1]}")
def test_same_size_conversions(dev, orc, pair):
    """every pair of one chroma sub-sampling: widths that are and are not multiples of the eight samples a thread takes, odd heights, planes on 64-byte
    and on odd-ish (2 + 2) addresses — the vector and the sample-by-sample forms of the loads and the stores"""
    sf, df = pair
    kernels = set()
    for (w, h) in ((64, 16), (200, 37), (66, 10), (18, 7), (8, 6)):
        for align, extra in ((64, 0), (2, 2)):
            kernels.add(_run(dev, orc, sf, df, w, h, align, extra))
    if "scale19_unit_kernel" in kernels:
        assert kernels == {"scale19_unit_kernel"}, kernels
    else:                                   # (libswscale's unscaled special converters and the copies: entry points of their own)
        assert not any("scale" in k for k in kernels), kernels


@pytest.mark.parametrize("pair", [("p010le", "nv12"), ("yuv420p16le", "p016le"), ("nv12", "yuv420p10le"), ("yuv444p", "yuv444p16le"), ("p016le", "yuv420p10le"),
                                  ("yuv420p10le", "p010le"), ("yuv420p", "yuv420p16le")], ids=lambda p: f"{p[0]}-{p[1]}")
def test_unit_form_is_taken_and_agrees_with_the_tile_form(dev, orc, pair, monkeypatch):
    sf, df = pair
    assert _run(dev, orc, sf, df, 136, 22, 64, 0) == "scale19_unit_kernel"
    monkeypatch.setenv("GMAT_S19_UNIT", "0")
    assert _run(dev, orc, sf, df, 136, 22, 64, 0) != "scale19_unit_kernel"


def _ex(dev, orc, sf, df, w, h, ranges, pos, seed, flags="bicubic"):
    """one call with ranges and chroma positions set at both ends: the library's planes and kernel against the oracle's"""
    L, lib = orc.L, dev.lib
    L.orc_sws_create_ex.restype = C.c_void_p
    L.orc_sws_create_ex.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    src = _synth(orc, sf, w, h, seed)
    oc = L.orc_sws_create_ex(w, h, PIX_FMT[sf], w, h, PIX_FMT[df], SWS[flags], None, (C.c_int * 4)(*pos), ranges[0], ranges[1])
    assert oc
    want = alloc_planes(df, w, h)
    assert L.orc_sws_scale(oc, planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                           planes([p.ctypes.data for p in want]), ints([p.strides[0] for p in want])) == h
    L.orc_sws_free(oc)
    c = lib.gmat_sws_getContext(w, h, PIX_FMT[sf], w, h, PIX_FMT[df], SWS[flags], None)
    assert c and lib.gmat_sws_setRange(c, ranges[0], ranges[1]) == 0 and lib.gmat_sws_setChromaPos(c, *pos) == 0
    d = dev.upload_planes(src, 64)
    dst = dev.planes_like(df, w, h, 64)
    assert lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, h, planes([p.ptr for p in dst]), ints([p.stride for p in dst])) == h
    kernel = lib.gmat_sws_lastKernel(c).decode()
    for i, (p, wv) in enumerate(zip(dst, want)):
        assert (p.download() == wv).all(), (sf, df, ranges, pos, i, kernel)
    lib.gmat_sws_freeContext(c)
    for p in d + dst:
        p.free()
    return kernel


@pytest.mark.parametrize("pair", [("p010le", "nv12"), ("yuv420p16le", "yuv420p10le"), ("nv12", "yuv420p16le"), ("yuv444p", "yuv444p16le"), ("p016le", "yuv420p")],
                         ids=lambda p: f"{p[0]}-{p[1]}")
def test_range_conversion_keeps_the_unit_form(dev, orc, pair):
    """lum / chrRange{To,From}Jpeg on the line value (swscale.c:157-226): a per-sample step, the unit form's too"""
    sf, df = pair
    for ranges in ((0, 1), (1, 0)):
        assert _ex(dev, orc, sf, df, 120, 18, ranges, (-513,) * 4, 31) == "scale19_unit_kernel"


def test_differing_chroma_positions_are_filters(dev, orc):
    """chroma positions that differ between the ends make the chroma banks real filters at equal size: not the unit form (bit-exact on whatever takes it)"""
    for sf, df in (("p010le", "nv12"), ("nv12", "yuv420p16le")):
        assert _ex(dev, orc, sf, df, 96, 20, (0, 0), (0, 128, 128, 128), 3) != "scale19_unit_kernel"
        assert _ex(dev, orc, sf, df, 96, 20, (0, 0), (128, 0, 128, 0), 3) == "scale19_unit_kernel"          # (the same at both ends: identities)


@pytest.mark.parametrize("df", ["yuv420p16le", "yuv444p16le", "p016le", "yuv420p10le"])
def test_one_tap_chroma_bank_whose_coefficient_is_not_the_unit(dev, orc, df):
    """found by tests/fuzz/fuzz_unit.py (seed 72001, case 757): a destination chroma position far above the plane makes initFilter's one-tap vertical chroma bank carry the
    coefficient 0 at the first row; yuv2plane1_* does not read the coefficient (vscale.c:30-105) and writes the line, the X form with the bank as it is writes the sums' floor —
    YUV420P16LE was left out of the planar 16-bit destinations' substitution (init_scale16) since round 4"""
    for sf in ("nv12", "yuv420p", "p010le"):
        if sf == df or (sf == "yuv420p" and df in ("p016le",)):
            continue
        for ranges in ((1, 0), (0, 0)):
            _ex(dev, orc, sf, df, 278, 6, ranges, (192, 256, 192, -263), 2757, flags="bilinear")
            _ex(dev, orc, sf, df, 64, 9, ranges, (-513, -513, 0, -400), 11, flags="point")


@pytest.mark.parametrize("pair", [("p010le", "nv12"), ("nv12", "yuv420p16le"), ("yuv420p10le", "p016le")], ids=lambda p: f"{p[0]}-{p[1]}")
def test_batch_is_one_launch(dev, orc, pair):
    """gmat_sws_scale_batch: the frames of a batch are one launch, every frame bit-exact"""
    sf, df = pair
    lib = dev.lib
    w, h, nf = 72, 14, 5
    srcs = [_synth(orc, sf, w, h, 40 + f) for f in range(nf)]
    wants = [orc.sws(s, w, h, sf, w, h, df, SWS["bicubic"]) for s in srcs]
    c = lib.gmat_sws_getContext(w, h, PIX_FMT[sf], w, h, PIX_FMT[df], SWS["bicubic"], None)
    assert c
    dsrc = [dev.upload_planes(s, 64) for s in srcs]
    ddst = [dev.planes_like(df, w, h, 64) for _ in srcs]
    sp, dp = (C.c_void_p * (4 * nf))(), (C.c_void_p * (4 * nf))()
    for f in range(nf):
        for i, p in enumerate(dsrc[f]):
            sp[4 * f + i] = p.ptr
        for i, p in enumerate(ddst[f]):
            dp[4 * f + i] = p.ptr
    assert lib.gmat_sws_scale_batch(c, nf, C.cast(sp, C.POINTER(C.c_void_p)), ints([p.stride for p in dsrc[0]]), C.cast(dp, C.POINTER(C.c_void_p)),
                                    ints([p.stride for p in ddst[0]]), C.cast((C.c_void_p * 1)(None), C.POINTER(C.c_void_p)), 1, 3) == nf
    lib.gmat_device_sync()
    assert lib.gmat_sws_lastKernel(c).decode() == "scale19_unit_kernel"
    assert lib.gmat_sws_lastLaunchFrames(c) == nf
    for f in range(nf):
        for i, (a, b) in enumerate(zip(ddst[f], wants[f])):
            assert (a.download() == b).all(), (pair, f, i)
    for fr in dsrc + ddst:
        for p in fr:
            p.free()
    lib.gmat_sws_freeContext(c)


@pytest.mark.gpu
@pytest.mark.parametrize("pair", [("p010le", "nv12"), ("nv12", "yuv420p10le"), ("yuv420p16le", "p016le"), ("yuv444p", "yuv444p16le")], ids=lambda p: f"{p[0]}-{p[1]}")
def test_full_size(dev, orc, pair):
    sf, df = pair
    assert _run(dev, orc, sf, df, 1920, 1080, 64, 0) == "scale19_unit_kernel"


SRC64 = ["nv12", "yuv420p", "p010le", "p016le", "yuv420p10le", "yuv420p16le", "yuv444p", "yuv444p16le"]


@pytest.mark.parametrize("df", ["rgba64le", "bgra64le"])
@pytest.mark.parametrize("sf", SRC64)
def test_rgba64_at_equal_size(dev, orc, sf, df):
    """yuv2rgb_cuda's 64-bit outputs (libswscale/cuda/yuv2rgb_cuda.cu:862-907) at equal size: identity horizontal banks, an identity vertical luma bank, the chroma's
    vertical filter as it is (four taps under bicubic, two under bilinear, one under point) — scale19_unit64_kernel where the planes sit on 16-byte addresses and
    the width is a multiple of 8, the tile form elsewhere; yuv2rgba64_X_c / _2_c / _1_c (output.c:1025-1270) in the oracle"""
    for flags in ("bicubic", "bilinear", "point"):
        assert _run(dev, orc, sf, df, 64, 18, 64, 0, flags=flags) == "scale19_unit64_kernel", (sf, df, flags)
        assert _run(dev, orc, sf, df, 136, 11, 16, 0, flags=flags) == "scale19_unit64_kernel", (sf, df, flags)
    assert _run(dev, orc, sf, df, 64, 18, 2, 2) == "scale19_kernel"                  # planes off 16-byte addresses
    assert _run(dev, orc, sf, df, 68, 10, 64, 0) == "scale19_kernel"                 # a width that is not whole units


def test_rgba64_unit_form_colourspaces_and_the_knob(dev, orc, monkeypatch):
    lib = dev.lib
    for sf, cs in (("nv12", 1), ("p010le", 9), ("yuv444p", 5)):
        w, h = 72, 14
        src = _synth(orc, sf, w, h, 77)
        want = orc.sws(src, w, h, sf, w, h, "rgba64le", SWS["bicubic"], colorspace=cs)
        c = lib.gmat_sws_getContext(w, h, PIX_FMT[sf], w, h, PIX_FMT["rgba64le"], SWS["bicubic"], None)
        assert c and lib.gmat_sws_setColorspace(c, cs, 0) == 0
        d = dev.upload_planes(src, 64)
        dst = dev.planes_like("rgba64le", w, h, 64)
        assert lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, h, planes([p.ptr for p in dst]), ints([p.stride for p in dst])) == h
        assert lib.gmat_sws_lastKernel(c).decode() == "scale19_unit64_kernel"
        assert (dst[0].download() == want[0]).all(), (sf, cs)
        lib.gmat_sws_freeContext(c)
        for p in d + dst:
            p.free()
    monkeypatch.setenv("GMAT_S19_UNIT", "0")
    assert _run(dev, orc, "nv12", "rgba64le", 64, 18, 64, 0) == "scale19_kernel"


@pytest.mark.gpu
@pytest.mark.parametrize("pair", [("nv12", "rgba64le"), ("p010le", "bgra64le"), ("yuv444p16le", "rgba64le")], ids=lambda p: f"{p[0]}-{p[1]}")
def test_rgba64_full_size(dev, orc, pair):
    assert _run(dev, orc, pair[0], pair[1], 1920, 1080, 64, 0) == "scale19_unit64_kernel"


@pytest.mark.parametrize("df", ["rgb24", "bgr24", "rgba", "bgra"])
@pytest.mark.parametrize("sf", ["p010le", "p016le", "yuv420p10le", "yuv420p16le"])
def test_deep_sources_into_rgb_at_equal_size(dev, orc, sf, df):
    """P010 / P016 / planar 10- and 16-bit 4:2:0 into packed 8-bit RGB at equal size: no unscaled converter in libswscale — hScale16To15_c with one coefficient, the chroma's
    vertical filter, yuv2rgb_X_c / _2_c / _1_c by the filter family (output.c:1600-1800) — unit_rgb_kernel where the planes sit on 16-byte addresses and the width is a multiple
    of 8, the walkers elsewhere"""
    for flags in ("bicubic", "bilinear", "point"):
        assert _run(dev, orc, sf, df, 64, 18, 64, 0, flags=flags) == "unit_rgb_kernel", (sf, df, flags)
        assert _run(dev, orc, sf, df, 136, 11, 16, 0, flags=flags) == "unit_rgb_kernel", (sf, df, flags)
    assert _run(dev, orc, sf, df, 64, 18, 2, 2) != "unit_rgb_kernel"                  # planes off 16-byte addresses
    assert _run(dev, orc, sf, df, 68, 10, 64, 0) != "unit_rgb_kernel"                 # a width that is not whole units


def test_deep_sources_into_rgb_colourspaces_batches_and_the_knob(dev, orc, monkeypatch):
    lib = dev.lib
    for sf, cs in (("p010le", 1), ("yuv420p16le", 9), ("p016le", 5)):
        w, h, nf = 72, 14, 3
        srcs = [_synth(orc, sf, w, h, 90 + f) for f in range(nf)]
        wants = [orc.sws(s_, w, h, sf, w, h, "bgra", SWS["bicubic"], colorspace=cs) for s_ in srcs]
        c = lib.gmat_sws_getContext(w, h, PIX_FMT[sf], w, h, PIX_FMT["bgra"], SWS["bicubic"], None)
        assert c and lib.gmat_sws_setColorspace(c, cs, 0) == 0
        dsrc = [dev.upload_planes(s_, 64) for s_ in srcs]
        ddst = [dev.planes_like("bgra", w, h, 64) for _ in srcs]
        sp, dp = (C.c_void_p * (4 * nf))(), (C.c_void_p * (4 * nf))()
        for f in range(nf):
            for i, p in enumerate(dsrc[f]):
                sp[4 * f + i] = p.ptr
            for i, p in enumerate(ddst[f]):
                dp[4 * f + i] = p.ptr
        assert lib.gmat_sws_scale_batch(c, nf, C.cast(sp, C.POINTER(C.c_void_p)), ints([p.stride for p in dsrc[0]]), C.cast(dp, C.POINTER(C.c_void_p)),
                                        ints([p.stride for p in ddst[0]]), C.cast((C.c_void_p * 1)(None), C.POINTER(C.c_void_p)), 1, 3) == nf
        lib.gmat_device_sync()
        assert lib.gmat_sws_lastKernel(c).decode() == "unit_rgb_kernel" and lib.gmat_sws_lastLaunchFrames(c) == nf
        for f in range(nf):
            assert (ddst[f][0].download() == wants[f][0]).all(), (sf, cs, f)
        lib.gmat_sws_freeContext(c)
        for fr in dsrc + ddst:
            for p in fr:
                p.free()
    monkeypatch.setenv("GMAT_S19_UNIT", "0")
    assert _run(dev, orc, "p010le", "rgb24", 64, 18, 64, 0) != "unit_rgb_kernel"


@pytest.mark.gpu
@pytest.mark.parametrize("pair", [("p010le", "rgb24"), ("yuv420p10le", "bgra"), ("p016le", "rgba")], ids=lambda p: f"{p[0]}-{p[1]}")
def test_deep_sources_into_rgb_full_size(dev, orc, pair):
    assert _run(dev, orc, pair[0], pair[1], 1920, 1080, 64, 0) == "unit_rgb_kernel"
