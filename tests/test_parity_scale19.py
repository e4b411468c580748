"""scale19_kernel (k_scale19.hip, round 6): 16-bit YUV destinations — P016LE, YUV420P16LE, YUV444P16LE, three of scale_cuda's formats
(libavfilter/vf_scale_cuda.c:45-54) — in ONE launch with a tile's 19-bit lines in LDS, against the oracle's restatement of libswscale's
19-bit path (hScale8To19_c / hScale16To19_c swscale.c:63-153, yuv2planeX_16_c / yuv2nv12cX_16_c output.c:157-211).  Bit-exact; every case
also runs as the two passes through HBM (k_scale16.hip, GMAT_S19=0) in tests/test_parity_scale.py."""
import ctypes as C

import numpy as np
import pytest

from harness import PIX_FMT, SWS, synth_planes, planes, ints

pytestmark = []

SRC = ["nv12", "yuv420p", "yuv444p", "p010le", "p016le", "yuv420p10le", "yuv420p16le", "yuv444p16le"]
DST = ["p016le", "yuv420p16le", "yuv444p16le", "rgba64le", "bgra64le"]


def _synth(orc, fmt, w, h, seed):
    src = synth_planes(orc, fmt, w, h, seed=seed)
    if fmt == "yuv420p10le":                              # valid input: 10 significant bits in the low end
        for p in src:
            p.view("<u2")[...] &= 0x3FF
    if fmt == "p010le":
        for p in src:
            p.view("<u2")[...] &= 0xFFC0
    return src


def _check(dev, orc, sf, df, geom, flags, align, extra, seed=91):
    sw, sh, dw, dh = geom
    src = _synth(orc, sf, sw, sh, seed)
    want = orc.sws(src, sw, sh, sf, dw, dh, df, SWS[flags])
    d = dev.upload_planes(src, align, extra)
    got, pads, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, SWS[flags], dst_align=align, dst_extra=extra)
    for p in d:
        p.free()
    for i, (g, wv) in enumerate(zip(got, want)):
        bad = np.argwhere(g != wv)
        assert bad.size == 0, f"{sf}->{df} {geom} {flags} align {align}: plane {i}: {len(bad)} bytes differ, first at {bad[:4].tolist()} ({kernel})"
        assert (pads[i] == 0xCD).all(), (sf, df, geom, i)
    return kernel


@pytest.mark.parametrize("df", DST)
@pytest.mark.parametrize("sf", SRC)
def test_formats(dev, orc, sf, df):
    """every plane source of the 19-bit path into every 16-bit YUV destination: interleaved / planar chroma at either end, 8- / 10- / 16-bit
    samples, 4:2:0 and 4:4:4; aligned and odd-aligned planes (the dword and the byte form of the stage, dword and 16-bit stores)"""
    for geom in ((192, 108, 128, 72), (96, 40, 144, 60), (101, 45, 75, 33)):
        for align, extra in ((64, 0), (2, 2)):
            k = _check(dev, orc, sf, df, geom, "bicubic", align, extra)
            assert k == "scale19_kernel", k


@pytest.mark.parametrize("flags", ["bicubic", "lanczos", "bilinear", "point", "area", "gauss", "sinc", "spline"])
@pytest.mark.parametrize("pair", [("p016le", "p016le"), ("nv12", "p016le"), ("yuv444p16le", "yuv444p16le"), ("yuv420p", "yuv420p16le"), ("nv12", "rgba64le"), ("yuv444p16le", "bgra64le")])
def test_algorithms_and_ratios(dev, orc, pair, flags):
    """filters of 1 to more than 16 horizontal taps (the 4- and 8-pair instances with their coefficients in registers, the any-length one), up-
    and down-scales, widths that are not multiples of the tile's 64 columns, a down-scale whose windows overlap 64 banks' worth"""
    sf, df = pair
    for geom in ((320, 180, 128, 72), (128, 72, 320, 180), (400, 120, 70, 30), (66, 34, 131, 67), (258, 66, 129, 33)):
        k = _check(dev, orc, sf, df, geom, flags, 64, 0, seed=17)
        assert k == "scale19_kernel" or flags == "sinc", k       # (SWS_SINC's widest banks: the planner may refuse — rows past 128 units, sum |c| past 2^16 — and the two passes serve)


@pytest.mark.parametrize("knobs", [{"GMAT_S19_LDS": "4096"}, {"GMAT_S19_LDS": "8192"}, {"GMAT_S19_ROWS": "1"}, {"GMAT_S19_ROWS": "4"},
                                   {"GMAT_S19_LDS": "65536"}, {"GMAT_S19_LDS": "65536", "GMAT_S19_ROWS": "48"}])
@pytest.mark.parametrize("pair", [("p016le", "p016le"), ("yuv420p", "yuv444p16le"), ("yuv444p16le", "p016le"), ("p010le", "rgba64le"), ("yuv444p", "bgra64le")])
def test_tile_plans(dev, orc, pair, knobs, monkeypatch):
    """the planner's corners: tiles of one output row, rows staged in several groups (a small LDS budget), the tallest tiles"""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    sf, df = pair
    for geom in ((384, 216, 128, 72), (130, 74, 200, 150), (640, 90, 176, 60)):
        for flags in ("bicubic", "lanczos"):
            assert _check(dev, orc, sf, df, geom, flags, 64, 0, seed=5) == "scale19_kernel"


@pytest.mark.parametrize("pair", [("p016le", "p016le"), ("nv12", "rgba64le"), ("yuv444p16le", "yuv420p16le")])
def test_long_windows_take_more_lds(dev, orc, pair, monkeypatch, capfd):
    """a tile whose vertical windows leave it under 85 % of its staged rows at 32 KB of LDS is planned with 40, then 48 KB (s19_prepare: 3 : 1 lanczos
    23.9 -> 19.5 us a 4K frame); the plan is read from the GMAT_S19_DEBUG line, the pixels held to the oracle as everywhere"""
    import re
    monkeypatch.setenv("GMAT_S19_DEBUG", "1")
    sf, df = pair
    seen = []
    for geom, flags in (((72, 900, 66, 300), "lanczos"), ((96, 1024, 40, 256), "bicubic"), ((64, 600, 64, 400), "bicubic")):
        capfd.readouterr()
        assert _check(dev, orc, sf, df, geom, flags, 64, 0, seed=23) == "scale19_kernel"
        m = re.findall(r"s19 job 0: .* lds (\d+) ", capfd.readouterr().err)
        assert m, "no plan line"
        seen.append(int(m[-1]))
    assert seen[0] > 32768 and seen[1] > 32768 and seen[2] <= 32768, seen


@pytest.mark.parametrize("df", ["rgba64le", "bgra64le"])
@pytest.mark.parametrize("sf", ["nv12", "yuv420p", "yuv444p", "p016le", "yuv420p10le"])
def test_rgba64_forms(dev, orc, sf, df):
    """yuv2rgb_cuda's 64-bit outputs (libswscale/cuda/yuv2rgb_cuda.cu:862-907) and swscale_cuda's (swscale_cuda.c:34-44): equal size (one-tap luma
    with the one- / two-tap / bicubic chroma forms of packed_vscale), equal height, odd widths (full chroma: a chroma column a pixel), colourspaces"""
    for geom in ((128, 72, 128, 72), (130, 50, 130, 50), (64, 34, 64, 17), (201, 91, 151, 67), (66, 20, 131, 41)):
        for flags in ("bicubic", "bilinear", "point"):
            unit = geom[:2] == geom[2:] and geom[0] % 8 == 0                  # (equal size, rows of whole units on aligned planes: the unit form, tests/test_parity_unit.py)
            assert _check(dev, orc, sf, df, geom, flags, 64, 0, seed=3) == ("scale19_unit64_kernel" if unit else "scale19_kernel")
    sw, sh = 96, 40
    src = _synth(orc, sf, sw, sh, 5)
    want = orc.sws(src, sw, sh, sf, sw, sh, df, SWS["bicubic"], colorspace=1)
    d = dev.upload_planes(src, 64)
    got, _, k = dev.sws(d, sw, sh, sf, sw, sh, df, SWS["bicubic"], dst_align=64, colorspace=(1, 0))
    assert k == "scale19_unit64_kernel" and (got[0] == want[0]).all()
    for p_ in d:
        p_.free()


def test_beyond_the_tile_kernel(dev, orc):
    """a row of more than 128 staging units (64 columns' windows spanning more than 512 samples: 16 : 1 under Lanczos): the planner refuses,
    the two passes of k_scale16.hip serve the context — bit-exact too"""
    k = _check(dev, orc, "p016le", "p016le", (1280, 90, 80, 60), "lanczos", 64, 0, seed=8)
    assert k == "hscale19_kernel+vscale16_kernel", k
    k = _check(dev, orc, "p016le", "p016le", (640, 90, 96, 60), "lanczos", 64, 0, seed=8)           # two column passes of the stage (a lane: two units a row)
    assert k == "scale19_kernel", k


@pytest.mark.parametrize("pair", [("p016le", "p016le"), ("nv12", "yuv420p16le"), ("yuv444p16le", "yuv444p16le"), ("p010le", "p016le"), ("nv12", "rgba64le"), ("yuv420p", "bgra64le")])
def test_batch_is_one_launch(dev, orc, pair):
    """gmat_sws_scale_batch: the frames of a batch are one launch (grid.y), every frame bit-exact"""
    sf, df = pair
    lib = dev.lib
    sw, sh, dw, dh, nf = 160, 90, 96, 54, 5
    srcs = [_synth(orc, sf, sw, sh, 200 + f) for f in range(nf)]
    wants = [orc.sws(s, sw, sh, sf, dw, dh, df, SWS["bicubic"]) for s in srcs]
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], SWS["bicubic"], None)
    assert c
    dsrc = [dev.upload_planes(s, 64) for s in srcs]
    ddst = [dev.planes_like(df, dw, dh, 64) for _ in srcs]
    sp, dp = (C.c_void_p * (4 * nf))(), (C.c_void_p * (4 * nf))()
    for f in range(nf):
        for i, p in enumerate(dsrc[f]):
            sp[4 * f + i] = p.ptr
        for i, p in enumerate(ddst[f]):
            dp[4 * f + i] = p.ptr
    assert lib.gmat_sws_scale_batch(c, nf, C.cast(sp, C.POINTER(C.c_void_p)), ints([p.stride for p in dsrc[0]]), C.cast(dp, C.POINTER(C.c_void_p)),
                                    ints([p.stride for p in ddst[0]]), C.cast((C.c_void_p * 1)(None), C.POINTER(C.c_void_p)), 1, 3) == nf
    lib.gmat_device_sync()
    assert lib.gmat_sws_lastKernel(c).decode() == "scale19_kernel"
    assert lib.gmat_sws_lastLaunchFrames(c) == nf
    for f in range(nf):
        for i, (a, b) in enumerate(zip(ddst[f], wants[f])):
            assert (a.download() == b).all(), (pair, f, i)
    for fr in dsrc + ddst:
        for p in fr:
            p.free()
    lib.gmat_sws_freeContext(c)


def test_range_conversion_and_chroma_positions(dev, orc):
    """lum / chrRange{To,From}Jpeg16_c on the lines of a tile (swscale.c:189-226) and chroma positions (the 19-bit path's own filter banks)"""
    from harness import alloc_planes
    L, lib = orc.L, dev.lib
    L.orc_sws_create_ex.restype = C.c_void_p
    L.orc_sws_create_ex.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    for sf, df in (("nv12", "p016le"), ("p016le", "yuv420p16le"), ("yuv444p16le", "yuv444p16le")):
        for ranges, pos in (((0, 1), (-513,) * 4), ((1, 0), (-513,) * 4), ((0, 0), (128, 0, 256, 64)), ((1, 0), (37, -200, 511, 3))):
            sw, sh, dw, dh = 192, 80, 100, 60
            src = _synth(orc, sf, sw, sh, 61)
            oc = L.orc_sws_create_ex(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], SWS["bicubic"], None, (C.c_int * 4)(*pos), ranges[0], ranges[1])
            assert oc
            want = alloc_planes(df, dw, dh)
            assert L.orc_sws_scale(oc, planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                                   planes([p.ctypes.data for p in want]), ints([p.strides[0] for p in want])) == dh
            L.orc_sws_free(oc)
            c = lib.gmat_sws_getContext(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], SWS["bicubic"], None)
            assert c and lib.gmat_sws_setRange(c, ranges[0], ranges[1]) == 0 and lib.gmat_sws_setChromaPos(c, *pos) == 0
            d = dev.upload_planes(src, 64)
            dst = dev.planes_like(df, dw, dh, 64)
            assert lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, sh,
                                      planes([p.ptr for p in dst]), ints([p.stride for p in dst])) == dh
            assert lib.gmat_sws_lastKernel(c).decode() == "scale19_kernel"
            for i, (p, wv) in enumerate(zip(dst, want)):
                assert (p.download() == wv).all(), (sf, df, ranges, pos, i)
            lib.gmat_sws_freeContext(c)
            for p in d + dst:
                p.free()


def test_packed_rgb_sources_ride_on_it(dev, orc):
    """packed RGB / RGBA64 sources into 16-bit YUV: their planes of 16-bit lines (k_rgb64.hip) are the inner context's source"""
    for sf in ("rgb24", "bgra", "rgba64le"):
        for df in ("p016le", "yuv444p16le"):
            sw, sh, dw, dh = 128, 72, 80, 44
            src = synth_planes(orc, sf, sw, sh, seed=33)
            want = orc.sws(src, sw, sh, sf, dw, dh, df, SWS["bicubic"])
            d = dev.upload_planes(src, 64)
            got, _, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, SWS["bicubic"], dst_align=64)
            assert kernel == "scale19_kernel", kernel
            for g, wv in zip(got, want):
                assert (g == wv).all(), (sf, df)
            for p in d:
                p.free()


@pytest.mark.gpu
def test_full_size(dev, orc):
    """BASELINE-sized frames: P016LE 4K -> 1080p and 1080p -> 720p, YUV444P16LE 1080p -> 720p, NV12 4K -> P016LE 1080p (the GPU only: the
    emulator's fibers take minutes at this size)"""
    if dev.kind != "hip":
        pytest.skip("full-size frames run on the real GPU only")
    for sf, df, geom in (("p016le", "p016le", (3840, 2160, 1920, 1080)), ("p016le", "p016le", (1920, 1080, 1280, 720)),
                         ("yuv444p16le", "yuv444p16le", (1920, 1080, 1280, 720)), ("nv12", "p016le", (3840, 2160, 1920, 1080))):
        assert _check(dev, orc, sf, df, geom, "bicubic", 256, 0, seed=3) == "scale19_kernel"
