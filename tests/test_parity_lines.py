"""The LINES form (k_scale_yuvl.hip, scale_yuvl_h_kernel + scale_yuvl_v_kernel; round 4): 8-bit YUV sources through a frame of horizontally
filtered 15-bit lines — the down-scales no walker takes (beyond the band walker's 6.1 : 1, range conversion, long filters, 4:4:4 ends), which
sat on the tiled kernel of round 1 at 0.03 - 0.1 of the roofline and were REFUSED beyond ~ 20 : 1 (no tile's window fits the LDS).  Against the
oracle (one libswscale context: swscale.c:234-520, initFilter utils.c:367-763) bit for bit; every test names the kernel the rule must pick."""
import ctypes as C

import numpy as np
import pytest

from harness import PIX_FMT, SWS, alloc_planes, ints, is_generic, planes, synth_planes
from test_batch_api import _run_batch


@pytest.fixture(autouse=True)
def _no_tile15(monkeypatch):
    """this file is about the lines form: the tile kernel on the 15-bit lines (round 6, k_scale19.hip — tests/test_parity_tile15.py) stands in front of both for the
    pairs whose plane layouts differ and is switched off here"""
    monkeypatch.setenv("GMAT_T15", "0")


LINES = "scale_yuvl_h_kernel+scale_yuvl_v_kernel"
RGB = ("rgb24", "bgr24", "rgba", "bgra")


@pytest.fixture
def forced(monkeypatch):
    """every walker out of the way and GMAT_LINES=2: the lines form wherever its own rule (formats, dword-aligned source planes) lets it"""
    monkeypatch.setenv("GMAT_SCALE_NO_STRIP", "1")
    monkeypatch.setenv("GMAT_SCALE_NO_GENERIC_WALKER", "1")
    monkeypatch.setenv("GMAT_SCALE_NO_QUAD_WALKER", "1")
    monkeypatch.setenv("GMAT_LINES", "2")


def _check(dev, orc, sf, df, geom, flags="bicubic", align=256, extra=0, seed=17, src_align=256, src_extra=0, colorspace=None, xflags=0):
    sw, sh, dw, dh = geom
    src = synth_planes(orc, sf, sw, sh, seed=seed)
    want = orc.sws(src, sw, sh, sf, dw, dh, df, SWS[flags] | xflags, colorspace=colorspace)
    d = dev.upload_planes(src, src_align, src_extra)
    got, pads, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, SWS[flags] | xflags, dst_align=align, dst_extra=extra,
                                colorspace=None if colorspace is None else (colorspace, 0))
    for p in d:
        p.free()
    for i, (g, w) in enumerate(zip(got, want)):
        bad = np.argwhere(g != w)
        assert bad.size == 0, f"{kernel} {sf}->{df} {geom} plane {i}: {len(bad)} mismatching bytes, first at {bad[:6].tolist()}"
        assert (pads[i] == 0xCD).all(), f"{kernel} plane {i}: wrote into the row padding"
    return kernel


# ratios 8 / 12 / 24 : 1 (filters of 33 / 49 / 97 taps), anamorphic, odd sizes on both sides, one partial 64-column group / exactly one / one
# and a remainder, moderate ratios (the tier the band walker normally keeps), an up-scaling vertical axis beside a shrinking horizontal one
GEOMS = [(384, 216, 48, 26), (768, 432, 64, 36), (960, 480, 40, 20), (768, 432, 273, 153), (200, 120, 67, 41), (264, 64, 64, 24),
         (1031, 57, 130, 31), (520, 36, 173, 12), (640, 48, 80, 96), (1920, 32, 66, 8)]


@pytest.mark.parametrize("sf", ["nv12", "yuv420p"])
@pytest.mark.parametrize("df", ["rgb24", "bgra", "nv12", "yuv420p", "yuv444p"])
@pytest.mark.parametrize("geom", GEOMS)
def test_lines_every_layout(dev, orc, forced, sf, df, geom):
    if df in ("nv12", "yuv420p") and (geom[2] % 2 or geom[3] % 2):
        pytest.skip("4:2:0 destinations of odd size: test_parity_scale.py")
    assert _check(dev, orc, sf, df, geom) == LINES


@pytest.mark.parametrize("df", ["rgb24", "rgba", "nv12", "yuv420p", "yuv444p"])
@pytest.mark.parametrize("geom", [(384, 216, 48, 26), (200, 120, 67, 41), (264, 64, 130, 30)])
def test_lines_planar444_source(dev, orc, forced, df, geom):
    """chroma planes at full size: the chroma axis of a 4:2:0 destination shrinks twice as far as the luma axis; RGB out = full chroma"""
    if df in ("nv12", "yuv420p") and (geom[2] % 2 or geom[3] % 2):
        pytest.skip("odd 4:2:0 size")
    assert _check(dev, orc, "yuv444p", df, geom) == LINES


@pytest.mark.parametrize("df", ["bgr24", "rgba"])
def test_lines_remaining_rgb_orders_and_pitches(dev, orc, forced, df):
    for geom in (GEOMS[0], GEOMS[4]):
        assert _check(dev, orc, "nv12", df, geom, align=1, extra=3) == LINES          # byte-aligned destination rows: the byte stores
        assert _check(dev, orc, "yuv420p", df, geom, align=4, extra=8, src_align=4, src_extra=12) == LINES


@pytest.mark.parametrize("flags", ["bicubic", "bilinear", "fast_bilinear", "point", "area", "gauss", "lanczos", "sinc", "spline", "bicublin", "x"])
def test_lines_every_algorithm(dev, orc, forced, flags):
    """the tables as initFilter made them: 1 to ~ 100 taps an axis"""
    wide = flags in ("sinc", "spline", "gauss", "x")            # 8 - 20 taps at 1 : 1: beyond 128 at 8 : 1 (declined)
    for geom in ((480, 128, 120, 32) if wide else (480, 128, 60, 16), (300, 90, 100, 30)):
        assert _check(dev, orc, "nv12", "rgb24", geom, flags=flags) == LINES
        assert _check(dev, orc, "yuv420p", "yuv420p", geom, flags=flags) == LINES


def test_lines_full_chroma_and_colorspaces(dev, orc, forced):
    for geom in ((384, 216, 48, 26), (200, 120, 67, 41)):
        assert _check(dev, orc, "nv12", "rgb24", geom, xflags=SWS["full_chr_h_int"]) == LINES
        assert _check(dev, orc, "yuv420p", "bgra", geom, xflags=SWS["full_chr_h_int"]) == LINES
        for cs in (1, 5, 7, 9):
            assert _check(dev, orc, "nv12", "rgb24", geom, colorspace=cs) == LINES


@pytest.mark.parametrize("rp", ["1", "3", "8", "64"])
def test_lines_row_pairs_per_wave(dev, orc, forced, monkeypatch, rp):
    """the run of source row pairs a wave filters (GMAT_LINES_RP): runs that end inside the plane, one run for the whole plane, an odd last row"""
    monkeypatch.setenv("GMAT_LINES_RP", rp)
    for geom in ((384, 217, 48, 27), (264, 62, 64, 24)):
        assert _check(dev, orc, "nv12", "rgb24", geom) == LINES
        assert _check(dev, orc, "yuv420p", "yuv444p", geom) == LINES


# ---- the default rule -------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sf,df", [("nv12", "rgb24"), ("nv12", "nv12"), ("yuv420p", "yuv420p"), ("nv12", "yuv420p"), ("yuv420p", "bgra")])
def test_default_rule_beyond_the_band_walker(dev, orc, sf, df):
    """no knob: 8 : 1 and beyond take the lines form, the band walker keeps 4 : 1 (a 2 : 1 .. 6.1 : 1 ratio none of whose walkers is switched off)"""
    assert _check(dev, orc, sf, df, (768, 432, 96, 54)) == LINES
    assert _check(dev, orc, sf, df, (1024, 128, 64, 8)) == LINES
    k = _check(dev, orc, sf, df, (768, 432, 160, 90))
    assert k != LINES and (is_generic(k) or k.startswith("scale_yuv")), k


def test_default_rule_source_alignment(dev, orc):
    """a source plane that does not start on a dword leaves the frame to the tiled kernel (the windows are read as aligned dwords)"""
    sw, sh, dw, dh = 768, 432, 96, 54
    src = synth_planes(orc, "nv12", sw, sh, seed=3)
    want = orc.sws(src, sw, sh, "nv12", dw, dh, "rgb24", SWS["bicubic"])
    for extra, lines in ((0, True), (2, False)):
        d = dev.upload_planes(src, 1, extra)                # pitch = width + extra
        got, pads, k = dev.sws(d, sw, sh, "nv12", dw, dh, "rgb24", SWS["bicubic"], dst_align=64)
        for p in d:
            p.free()
        assert (k == LINES) == lines, (extra, k)
        assert (got[0] == want[0]).all() and (pads[0] == 0xCD).all()


def test_ratios_the_tiled_kernel_refused(dev, orc):
    """no tile's window fits a workgroup's LDS beyond ~ 20 : 1: gmat_sws_getContext answered NULL before round 4's lines form (vf_scale_cuda.c:348-402
    serves any w:h); filters beyond 128 taps are still declined"""
    for sf, df, geom in (("nv12", "rgb24", (1536, 768, 64, 32)), ("yuv420p", "yuv420p", (1536, 768, 64, 32)), ("nv12", "nv12", (2048, 64, 64, 2)),
                         ("nv12", "bgra", (1920, 1080, 64, 36))):
        assert _check(dev, orc, sf, df, geom) == LINES
    lib = dev.lib
    assert not lib.gmat_sws_getContext(2048, 64, PIX_FMT["nv12"], 32, 2, PIX_FMT["rgb24"], SWS["bicubic"], None)      # 64 : 1 = 257 taps


@pytest.mark.parametrize("ranges", [(0, 1), (1, 0)])
@pytest.mark.parametrize("fmts", [("nv12", "nv12"), ("yuv420p", "yuv420p"), ("nv12", "yuv420p"), ("yuv420p", "yuv444p")])
@pytest.mark.parametrize("geom", [(768, 432, 96, 54), (480, 270, 200, 112)])
def test_range_conversion_on_the_lines(dev, orc, ranges, fmts, geom):
    """lum / chrRangeToJpeg_c / FromJpeg_c act on the horizontally filtered lines (hscale.c:60,193; swscale.c:157-188): pass H applies them.  No
    walker converts ranges, so a shrinking context takes the lines form at any ratio from 2 : 1"""
    sw, sh, dw, dh = geom
    sf, df = fmts
    src = synth_planes(orc, sf, sw, sh, seed=47)
    L = orc.L
    L.orc_sws_create_ex.restype = C.c_void_p
    L.orc_sws_create_ex.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    oc = L.orc_sws_create_ex(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], SWS["bicubic"], None, (C.c_int * 4)(-513, -513, -513, -513), ranges[0], ranges[1])
    assert oc
    want = alloc_planes(df, dw, dh)
    assert L.orc_sws_scale(oc, planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                           planes([p.ctypes.data for p in want]), ints([p.strides[0] for p in want])) == dh
    L.orc_sws_free(oc)
    lib = dev.lib
    d = dev.upload_planes(src, 256)
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], SWS["bicubic"], None)
    assert c and lib.gmat_sws_setRange(c, ranges[0], ranges[1]) == 0
    dst = dev.planes_like(df, dw, dh, 256)
    assert lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, sh,
                              planes([p.ptr for p in dst]), ints([p.stride for p in dst])) == dh
    k = lib.gmat_sws_lastKernel(c).decode()
    for a, b in zip(dst, want):
        assert (a.download() == b).all(), k
    lib.gmat_sws_freeContext(c)
    for p in d + dst:
        p.free()
    assert k == LINES, k


# ---- 16-bit sources, 10-bit destinations (round 4: refused from ~ 10 : 1 before: no tile's window fits the LDS) ---------------------------
def _deep(orc, fmt, sw, sh, seed):
    src = synth_planes(orc, fmt, sw, sh, seed=seed)
    if fmt == "yuv420p10le":                               # valid input: 10 significant bits in the low end
        for p in src:
            p.view("<u2")[...] &= 0x3FF
    if fmt == "p010le":                                    # ... in the high end
        for p in src:
            p.view("<u2")[...] &= 0xFFC0
    return src


@pytest.mark.parametrize("sf,df", [("p010le", "nv12"), ("p010le", "p010le"), ("p010le", "rgb24"), ("p016le", "yuv420p"), ("p016le", "bgra"),
                                   ("yuv420p10le", "rgb24"), ("yuv420p10le", "yuv420p10le"), ("yuv420p10le", "nv12"), ("yuv420p16le", "yuv420p"),
                                   ("yuv444p16le", "rgb24"), ("yuv444p16le", "yuv444p"), ("nv12", "p010le"), ("yuv420p", "yuv420p10le"), ("nv12", "yuv420p10le")])
@pytest.mark.parametrize("geom", [(768, 432, 64, 36), (1536, 96, 64, 4), (520, 100, 66, 12)])
def test_lines_deep_sources_and_destinations(dev, orc, sf, df, geom):
    """hScale16To15_c on P010 (sample >> 6) / P016 and planar 16 bit (biased by 32768) / planar 10 bit samples, interleaved 16-bit chroma; 8-bit planar
    output of a deeper source takes ff_dither_8x8_128 (swscale.c:263-264), 10-bit output yuv2p010lX_c / yuv2planeX_10_c"""
    sw, sh, dw, dh = geom
    src = _deep(orc, sf, sw, sh, 23)
    want = orc.sws(src, sw, sh, sf, dw, dh, df, SWS["bicubic"])
    d = dev.upload_planes(src, 256)
    got, pads, k = dev.sws(d, sw, sh, sf, dw, dh, df, SWS["bicubic"], dst_align=256)
    for p in d:
        p.free()
    for i, (g, w) in enumerate(zip(got, want)):
        assert (g == w).all(), (k, i, int((g != w).sum()))
        assert (pads[i] == 0xCD).all()
    deep_src = sf not in ("nv12", "yuv420p")
    assert k == ("scale_yuvl_h16_kernel+scale_yuvl_v_kernel" if deep_src else LINES), k


def test_lines_deep_source_batch_and_moderate_ratio(dev, orc, forced):
    for sf, df in (("p010le", "nv12"), ("yuv420p10le", "rgb24"), ("p016le", "p010le")):
        assert _run_batch(dev, orc, sf, df, 480, 128, 120, 32, 5, 2, 256) == "scale_yuvl_h16_kernel+scale_yuvl_v_kernel"


# ---- many frames, several streams, graphs ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sf,df", [("nv12", "rgb24"), ("yuv420p", "yuv420p"), ("nv12", "yuv420p")])
def test_lines_batches(dev, orc, sf, df):
    """n frames = ONE launch of each pass (grid.y = frame, n lines frames); the frames of a batch handed to two streams are ordered by the
    context (its lines frames are an intermediate it owns: stream_handoff_*)"""
    assert _run_batch(dev, orc, sf, df, 768, 216, 96, 26, 5, 1, 256) == LINES
    assert _run_batch.last_frames == 5
    assert _run_batch(dev, orc, sf, df, 768, 216, 96, 26, 6, 2, 256) == LINES


def test_lines_rule_by_launch_size(dev, orc):
    """below 2 : 1 one frame a launch stays on the tiled kernel (two launches and a lines frame through memory cost more than they save:
    18.9 against 12.3 us at 1080p -> 720p), four frames or more take the lines form (5.2 against 6.8 us a frame at 32)"""
    geom = (480, 270, 320, 180)
    k = _check(dev, orc, "nv12", "yuv444p", geom)
    assert k.startswith("scale_yuv_kernel"), k
    assert _run_batch(dev, orc, "nv12", "yuv444p", *geom, 5, 1, 256) == LINES
    assert _run_batch(dev, orc, "nv12", "yuv444p", *geom, 3, 1, 256).startswith("scale_yuv_kernel")


def test_lines_more_frames_than_one_launch(dev, orc):
    assert _run_batch(dev, orc, "nv12", "rgb24", 512, 64, 64, 8, 35, 1, 64) == LINES
    assert _run_batch.last_frames == 3


def test_lines_graph_replay(dev, orc):
    if dev.kind != "hip":
        pytest.skip("graph capture needs the HIP runtime")
    assert _run_batch(dev, orc, "nv12", "rgb24", 768, 216, 96, 26, 4, 2, 256, graph=True) == LINES


def test_lines_frame_is_handed_between_streams(dev, orc):
    """one context called on alternating streams: each call waits for the previous one's use of the lines frame (gmat_sws_streamHandoffs counts)"""
    lib = dev.lib
    sw, sh, dw, dh = 768, 216, 96, 26
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT["nv12"], dw, dh, PIX_FMT["rgb24"], SWS["bicubic"], None)
    assert c
    streams = []
    for _ in range(2):
        h = C.c_void_p()
        assert lib.gmat_stream_create(C.byref(h)) == 0
        streams.append(h)
    frames = [synth_planes(orc, "nv12", sw, sh, seed=60 + i) for i in range(6)]
    dsrc = [dev.upload_planes(f, 256) for f in frames]
    ddst = [dev.planes_like("rgb24", dw, dh, 256) for _ in frames]
    for i in range(len(frames)):
        lib.gmat_sws_setStream(c, streams[i & 1])
        assert lib.gmat_sws_scale(c, planes([p.ptr for p in dsrc[i]]), ints([p.stride for p in dsrc[i]]), 0, sh,
                                  planes([p.ptr for p in ddst[i]]), ints([p.stride for p in ddst[i]])) == dh
    assert lib.gmat_sws_lastKernel(c).decode() == LINES
    assert lib.gmat_sws_streamHandoffs(c) == len(frames) - 1
    lib.gmat_device_sync()
    for i, f in enumerate(frames):
        want = orc.sws(f, sw, sh, "nv12", dw, dh, "rgb24", SWS["bicubic"])
        assert (ddst[i][0].download() == want[0]).all(), i
    for s in streams:
        lib.gmat_stream_destroy(s)
    for f in dsrc + ddst:
        for p in f:
            p.free()
    lib.gmat_sws_freeContext(c)


def test_lines_one_stream_needs_no_ordering_and_a_destroyed_stream_is_never_touched(dev, orc, forced):
    """round 4 (FINDINGS R4-gaps): a context that only ever sees one stream records no event per call (the record cost the next call's first launch
    ~ 4 us); the first call on a second stream orders itself behind everything (once), and the FIRST stream may be gone by then"""
    lib = dev.lib
    sw, sh, dw, dh = 512, 128, 64, 16
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT["nv12"], dw, dh, PIX_FMT["rgb24"], SWS["bicubic"], None)
    assert c
    a, b = C.c_void_p(), C.c_void_p()
    assert lib.gmat_stream_create(C.byref(a)) == 0 and lib.gmat_stream_create(C.byref(b)) == 0
    frames = [synth_planes(orc, "nv12", sw, sh, seed=80 + i) for i in range(5)]
    dsrc = [dev.upload_planes(f, 256) for f in frames]
    ddst = [dev.planes_like("rgb24", dw, dh, 256) for _ in frames]

    def call(i, stream):
        lib.gmat_sws_setStream(c, stream)
        assert lib.gmat_sws_scale(c, planes([p.ptr for p in dsrc[i]]), ints([p.stride for p in dsrc[i]]), 0, sh,
                                  planes([p.ptr for p in ddst[i]]), ints([p.stride for p in ddst[i]])) == dh

    for i in range(3):
        call(i, a)
    assert lib.gmat_sws_lastKernel(c).decode() == LINES
    assert lib.gmat_sws_streamHandoffs(c) == 0
    assert lib.gmat_stream_sync(a) == 0
    lib.gmat_stream_destroy(a)                             # the first stream is gone before the second one shows up
    call(3, b)
    assert lib.gmat_sws_streamHandoffs(c) == 1
    call(4, b)
    assert lib.gmat_sws_streamHandoffs(c) == 1
    lib.gmat_device_sync()
    for i, f in enumerate(frames):
        want = orc.sws(f, sw, sh, "nv12", dw, dh, "rgb24", SWS["bicubic"])
        assert (ddst[i][0].download() == want[0]).all(), i
    lib.gmat_stream_destroy(b)
    for f in dsrc + ddst:
        for p in f:
            p.free()
    lib.gmat_sws_freeContext(c)


def test_a_call_that_leaves_the_lines_frame_alone_on_a_second_stream(dev, orc):
    """the first call on a second stream that does NOT touch the intermediates (below 2 : 1 a launch of fewer than four frames is the
    tiled kernel's) records nothing — the context must still own an event afterwards, or the next call that does touch them waits on a null handle
    (hipErrorInvalidHandle from then on).  Touched on a, untouched on b, touched on b, touched on a, untouched on a third stream, touched there."""
    lib = dev.lib
    sw, sh, dw, dh = 480, 270, 320, 180
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT["nv12"], dw, dh, PIX_FMT["yuv444p"], SWS["bicubic"], None)
    assert c
    st = []
    for _ in range(3):
        h = C.c_void_p()
        assert lib.gmat_stream_create(C.byref(h)) == 0
        st.append(h)
    plan = [(5, 0), (1, 1), (5, 1), (4, 0), (2, 2), (6, 2), (1, 0)]          # (frames in the call, stream)
    nf = sum(n for n, _ in plan)
    frames = [synth_planes(orc, "nv12", sw, sh, seed=140 + i) for i in range(nf)]
    dsrc = [dev.upload_planes(f, 256) for f in frames]
    ddst = [dev.planes_like("yuv444p", dw, dh, 256) for _ in frames]
    ss, ds = ints([p.stride for p in dsrc[0]]), ints([p.stride for p in ddst[0]])
    f0, kernels = 0, []
    for n, si in plan:
        sp, dp = (C.c_void_p * (4 * n))(), (C.c_void_p * (4 * n))()
        for f in range(n):
            for i, p in enumerate(dsrc[f0 + f]):
                sp[4 * f + i] = p.ptr
            for i, p in enumerate(ddst[f0 + f]):
                dp[4 * f + i] = p.ptr
        one = (C.c_void_p * 1)(st[si])
        assert lib.gmat_sws_scale_batch(c, n, C.cast(sp, C.POINTER(C.c_void_p)), ss, C.cast(dp, C.POINTER(C.c_void_p)), ds,
                                        C.cast(one, C.POINTER(C.c_void_p)), 1, 3) == n, (n, si, kernels)
        kernels.append(lib.gmat_sws_lastKernel(c).decode())
        f0 += n
    assert [k == LINES for k in kernels] == [True, False, True, True, False, True, False], kernels
    lib.gmat_device_sync()
    for i, f in enumerate(frames):
        want = orc.sws(f, sw, sh, "nv12", dw, dh, "yuv444p", SWS["bicubic"])
        for a, b in zip(ddst[i], want):
            assert (a.download() == b).all(), i
    for s in st:
        lib.gmat_stream_destroy(s)
    for f in dsrc + ddst:
        for p in f:
            p.free()
    lib.gmat_sws_freeContext(c)


@pytest.mark.parametrize("dot4", ["0", "1"])
@pytest.mark.parametrize("sf,df", [("nv12", "rgb24"), ("yuv420p", "yuv420p"), ("yuv420p", "bgra"), ("nv12", "yuv444p")])
@pytest.mark.parametrize("geom", [(768, 432, 64, 36), (1536, 96, 64, 4), (520, 100, 66, 12), (384, 216, 160, 90)])
def test_lines_byte_planes_in_both_table_forms(dev, orc, forced, monkeypatch, dot4, sf, df, geom):
    """pass H's byte planes: int16 coefficient pairs (v_perm_b32 + v_dot2) and the signed-byte form (samples - 128 in the row image, c = 256 ch + cl,
    two v_dot4c_i32_i8 a dword; the shipped rule takes it from 40 pairs on) — the same sums, bit for bit"""
    monkeypatch.setenv("GMAT_LINES_DOT4", dot4)
    dev.lib.gmat_knobs_reload()
    assert _check(dev, orc, sf, df, geom) == LINES
    if geom[0] <= 12 * geom[2]:                            # (Lanczos at 24 : 1 is 144 taps: beyond the form's 128)
        assert _check(dev, orc, sf, df, geom, flags="lanczos", seed=19) == LINES


@pytest.mark.parametrize("sf,df,geom,flags", [("yuv420p", "rgb24", (321, 432, 81, 108), "sinc"), ("nv12", "rgb24", (1538, 98, 64, 4), "bicubic"),
                                              ("yuv420p", "yuv420p", (1001, 120, 40, 6), "bicubic"), ("nv12", "nv12", (1538, 98, 64, 4), "bicubic"),
                                              ("p010le", "nv12", (1538, 98, 64, 4), "bicubic"), ("yuv420p16le", "rgb24", (1610, 96, 66, 4), "bicubic")])
@pytest.mark.parametrize("src_align,src_extra", [(1, 0), (1, 1), (2, 2)])
def test_lines_serve_unaligned_planes_where_no_tile_fits(dev, orc, sf, df, geom, flags, src_align, src_extra):
    """a context the tiled kernel has no tiling for was accepted on the strength of the lines form: it must serve source planes of any alignment
    (fuzz_ref_core: yuv420p 321 x 432 -> 81 x 108 sinc, chroma rows of 161 bytes — gmat_sws_scale returned -ENOSYS frame by frame)"""
    if sf in ("p010le", "yuv420p16le") and src_align == 1:
        src_align, src_extra = 2, 6                        # (rows of 16-bit samples are at least 2-byte aligned)
    k = _check(dev, orc, sf, df, geom, flags=flags, src_align=src_align, src_extra=src_extra)
    assert k == (LINES if sf in ("nv12", "yuv420p") else "scale_yuvl_h16_kernel+scale_yuvl_v_kernel"), k
