"""gmat_sws_scale_batch / gmat_sws_graph_create: many frames of one geometry per call.  Contexts on the 2:1 kernel send
each stream's share of the frames as ONE launch (grid.y = frame, plane pointers in the kernel-argument segment);
every other case goes frame by frame.  Each frame must equal what gmat_sws_scale / the oracle gives for it alone."""
import ctypes as C

import numpy as np
import pytest

from harness import is_generic, PIX_FMT, SWS, ints, synth_planes


def _run_batch(dev, orc, src_fmt, dst_fmt, sw, sh, dw, dh, nframes, nstreams, align, flags=SWS["bicubic"], graph=False):
    lib = dev.lib
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT[src_fmt], dw, dh, PIX_FMT[dst_fmt], flags | SWS["hwaccel"], None)
    assert c
    srcs = [synth_planes(orc, src_fmt, sw, sh, seed=300 + 7 * f) for f in range(nframes)]
    if src_fmt == "yuv420p10le":                           # valid input: 10 significant bits in the low end of each sample
        for fr in srcs:
            for p in fr:
                p.view("<u2")[...] &= 0x3FF
    dsrc = [dev.upload_planes(s, align) for s in srcs]
    ddst = [dev.planes_like(dst_fmt, dw, dh, align) for _ in range(nframes)]
    sp = (C.c_void_p * (4 * nframes))()
    dp = (C.c_void_p * (4 * nframes))()
    for f in range(nframes):
        for i, p in enumerate(dsrc[f]):
            sp[4 * f + i] = p.ptr
        for i, p in enumerate(ddst[f]):
            dp[4 * f + i] = p.ptr
    ss, ds = ints([p.stride for p in dsrc[0]]), ints([p.stride for p in ddst[0]])
    streams = (C.c_void_p * nstreams)()
    for s in range(nstreams):
        h = C.c_void_p()
        assert lib.gmat_stream_create(C.byref(h)) == 0
        streams[s] = h
    if graph:
        ge = C.c_void_p()
        assert lib.gmat_sws_graph_create(c, nframes, C.cast(sp, C.POINTER(C.c_void_p)), ss, C.cast(dp, C.POINTER(C.c_void_p)), ds,
                                         streams[0], nstreams, C.byref(ge)) == 0
        for p in ddst[0]:                                   # the warm launch wrote frame 0: make the replay prove itself
            lib.gmat_memset(p.ptr, 0xCD, p.stride * p.rows)
        assert lib.gmat_graph_launch(ge, streams[0]) == 0
        lib.gmat_stream_sync(streams[0])
        lib.gmat_graph_destroy(ge)
    else:
        r = lib.gmat_sws_scale_batch(c, nframes, C.cast(sp, C.POINTER(C.c_void_p)), ss, C.cast(dp, C.POINTER(C.c_void_p)), ds,
                                     C.cast(streams, C.POINTER(C.c_void_p)), nstreams, 3)
        assert r == nframes
        lib.gmat_stream_sync(streams[0])
    kernel = lib.gmat_sws_lastKernel(c).decode()
    _run_batch.last_frames = lib.gmat_sws_lastLaunchFrames(c)
    lib.gmat_device_sync()
    for f in range(nframes):
        want = orc.sws(srcs[f], sw, sh, src_fmt, dw, dh, dst_fmt, flags)
        for i, (p, w) in enumerate(zip(ddst[f], want)):
            assert (p.download() == w).all(), (f, i, kernel)
            assert (p.download(with_padding=True)[:, p.row_bytes:] == 0xCD).all()
    for s in range(nstreams):
        lib.gmat_stream_destroy(streams[s])
    for f in range(nframes):
        for p in dsrc[f] + ddst[f]:
            p.free()
    lib.gmat_sws_freeContext(c)
    return kernel


@pytest.fixture(params=["strip", "tiled"])
def strip_or_tiled(request, monkeypatch):
    """GMAT_SCALE_NO_STRIP at context creation keeps the tiled 2:1 kernels (the path of the frames the strip kernels
    decline): the batched entry point is checked on both"""
    if request.param == "tiled":
        monkeypatch.setenv("GMAT_SCALE_NO_STRIP", "1")
    else:
        monkeypatch.delenv("GMAT_SCALE_NO_STRIP", raising=False)
    return request.param


def _expected_2to1(which, src_fmt, dst_fmt):
    """256 x 64 and 64 x 32 sources, bicubic: packed RGB -> scale_yuv2s_kernel; 4:2:0 with the same chroma layout on both sides
    -> scale_yuv2p_kernel when the output has >= 16 rows; everything else, and everything under GMAT_SCALE_NO_STRIP, the tiled
    kernel"""
    if which == "strip" and dst_fmt in ("rgb24", "bgra"):
        return "scale_yuv2s_blk_kernel"              # (frames this small: every launch is the block form, k_scale_yuv2s.hip yuv2s_block_form)
    if which == "strip" and dst_fmt in ("nv12", "yuv420p"):
        return "scale_yuv2p_kernel" if src_fmt == dst_fmt else "scale_yuv2px_kernel"     # same / mixed chroma layouts
    return "scale_yuv2x_kernel<yuv>" if dst_fmt in ("nv12", "yuv420p") else "scale_yuv2x_kernel"


@pytest.mark.parametrize("dst_fmt", ["rgb24", "bgra", "nv12", "yuv420p"])
@pytest.mark.parametrize("src_fmt", ["nv12", "yuv420p"])
def test_batch_on_the_2to1_kernel(dev, orc, strip_or_tiled, src_fmt, dst_fmt):
    k = _run_batch(dev, orc, src_fmt, dst_fmt, 256, 64, 128, 32, nframes=5, nstreams=2, align=64)
    want = _expected_2to1(strip_or_tiled, src_fmt, dst_fmt)
    assert k == want, k


@pytest.mark.parametrize("case", [("nv12", "rgb24", 96, 40, 144, 60), ("yuv420p", "nv12", 200, 90, 80, 36),
                                  ("yuv444p", "bgra", 64, 32, 64, 32), ("p010le", "nv12", 128, 48, 80, 24),   # (not 2:1: that is the strip kernel's)
                                  ("nv12", "p010le", 128, 48, 96, 40), ("nv12", "yuv444p", 64, 32, 64, 32),
                                  ("rgb24", "nv12", 128, 48, 80, 24)])               # (at exactly 2:1: scale_rgb2y_kernel, test_parity_rgb2y.py)
def test_batch_on_the_generic_plane_scaler(dev, orc, case):
    """geometries the 2:1 kernel does not take batch too: scale_yuv_kernel with grid.y = frame"""
    sf, df, sw, sh, dw, dh = case
    k = _run_batch(dev, orc, sf, df, sw, sh, dw, dh, nframes=5, nstreams=2, align=64)
    assert is_generic(k), k
    assert _run_batch.last_frames == 2          # the second stream's share of 5 frames


@pytest.mark.parametrize("case", [("rgb24", "nv12", 264, 40), ("bgr24", "yuv420p", 264, 40), ("rgb24", "nv12", 260, 40)])
def test_batch_on_the_rgb_to_yuv_converter(dev, orc, case):
    """the same-size RGB -> 4:2:0 converter batches as well: rgb2yuv420s_kernel with grid.y = frame when every frame passes its rule
    (width a multiple of 8), frame by frame on the tiled kernel otherwise"""
    sf, df, w, h = case
    k = _run_batch(dev, orc, sf, df, w, h, w, h, nframes=5, nstreams=2, align=16)
    if w % 8 == 0:
        assert k == "rgb2yuv420s_kernel" and _run_batch.last_frames == 2, (k, _run_batch.last_frames)
    else:
        assert k == "rgb2yuv420_kernel", k


@pytest.mark.parametrize("case", [("rgb24", "rgb24"), ("bgr24", "bgra")])
def test_batch_on_the_rgb_strip_kernel(dev, orc, case):
    """packed RGB at exactly 2:1 batches too: scale_rgb2h_kernel with grid.y = frame"""
    sf, df = case
    k = _run_batch(dev, orc, sf, df, 264, 40, 132, 20, nframes=5, nstreams=2, align=16)
    assert k == "scale_rgb2h_kernel", k
    assert _run_batch.last_frames == 2


@pytest.mark.parametrize("fused", [0, 1])
@pytest.mark.parametrize("src_fmt", ["nv12", "yuv420p"])
def test_batch_two_kernel_form(dev, orc, src_fmt, fused):
    """setFused(0) — convert at source size, then scale (the reference's structure) — batches as two launches for n frames:
    the converter into n context-owned RGB24 intermediates, then the strip scaler; setFused(1) — the same arithmetic without the
    intermediate — as ONE launch of scale_rgb2h_kernel<yuv>; bytes = the chained oracle's either way"""
    import ctypes as C
    lib = dev.lib
    sw, sh, dw, dh, n = 264, 40, 132, 20, 5
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT[src_fmt], dw, dh, PIX_FMT["rgb24"], SWS["bicubic"], None)
    assert c and lib.gmat_sws_setFused(c, fused) == 0
    srcs = [synth_planes(orc, src_fmt, sw, sh, seed=800 + f) for f in range(n)]
    dsrc = [dev.upload_planes(s, 16) for s in srcs]
    ddst = [dev.planes_like("rgb24", dw, dh, 16) for _ in range(n)]
    sp = (C.c_void_p * (4 * n))(); dp = (C.c_void_p * (4 * n))()
    for f in range(n):
        for i, p in enumerate(dsrc[f]): sp[4 * f + i] = p.ptr
        dp[4 * f] = ddst[f][0].ptr
    st = C.c_void_p(); assert lib.gmat_stream_create(C.byref(st)) == 0
    streams = (C.c_void_p * 1)(st)
    r = lib.gmat_sws_scale_batch(c, n, C.cast(sp, C.POINTER(C.c_void_p)), ints([p.stride for p in dsrc[0]]),
                                 C.cast(dp, C.POINTER(C.c_void_p)), ints([ddst[0][0].stride]),
                                 C.cast(streams, C.POINTER(C.c_void_p)), 1, 0)
    assert r == n and lib.gmat_sws_lastLaunchFrames(c) == n
    assert lib.gmat_sws_lastKernel(c) == (b"scale_rgb2h_kernel<yuv>" if fused else b"scale_rgb2h_kernel")
    lib.gmat_stream_sync(st)
    for f in range(n):
        assert (ddst[f][0].download() == orc.chained(srcs[f], sw, sh, src_fmt, dw, dh, "rgb24")[0]).all(), f
    lib.gmat_stream_destroy(st)
    lib.gmat_sws_freeContext(c)
    for f in range(n):
        for p in dsrc[f] + ddst[f]: p.free()


def test_batch_more_frames_than_one_launch_carries(dev, orc, strip_or_tiled):
    """kYuv2xMaxFrames = 32 per launch: 37 frames on one stream = two launches, on every 2:1 kernel"""
    k = _run_batch(dev, orc, "nv12", "rgb24", 64, 32, 32, 16, nframes=37, nstreams=1, align=16)
    assert k == _expected_2to1(strip_or_tiled, "nv12", "rgb24"), k
    k = _run_batch(dev, orc, "nv12", "nv12", 64, 32, 32, 16, nframes=37, nstreams=1, align=16)
    assert k == _expected_2to1(strip_or_tiled, "nv12", "nv12"), k


@pytest.mark.parametrize("dst_fmt", ["rgb24", "bgra"])
@pytest.mark.parametrize("src_fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("geom", [(64, 16, 64), (130, 34, 1), (36, 6, 4)])
def test_batch_same_size_converter(dev, orc, src_fmt, dst_fmt, geom):
    """the unscaled yuv -> rgb converter batches too (grid.z = frame); nearest-chroma yuv2rgb.c semantics"""
    w, h, align = geom
    lib = dev.lib
    n = 5
    c = lib.gmat_sws_getContext(w, h, PIX_FMT[src_fmt], w, h, PIX_FMT[dst_fmt], SWS["bicubic"] | SWS["hwaccel"], None)
    assert c
    srcs = [synth_planes(orc, src_fmt, w, h, seed=400 + f) for f in range(n)]
    dsrc = [dev.upload_planes(s, align) for s in srcs]
    ddst = [dev.planes_like(dst_fmt, w, h, align) for _ in range(n)]
    sp = (C.c_void_p * (4 * n))(); dp = (C.c_void_p * (4 * n))()
    for f in range(n):
        for i, p in enumerate(dsrc[f]): sp[4 * f + i] = p.ptr
        dp[4 * f] = ddst[f][0].ptr
    st = C.c_void_p(); assert lib.gmat_stream_create(C.byref(st)) == 0
    streams = (C.c_void_p * 1)(st)
    r = lib.gmat_sws_scale_batch(c, n, C.cast(sp, C.POINTER(C.c_void_p)), ints([p.stride for p in dsrc[0]]),
                                 C.cast(dp, C.POINTER(C.c_void_p)), ints([ddst[0][0].stride]),
                                 C.cast(streams, C.POINTER(C.c_void_p)), 1, 0)
    assert r == n and lib.gmat_sws_lastLaunchFrames(c) == n and lib.gmat_sws_lastKernel(c) == b"yuv2rgb_kernel"
    lib.gmat_stream_sync(st)
    for f in range(n):
        assert (ddst[f][0].download() == orc.yuv2rgb(srcs[f], w, h, src_fmt, dst_fmt)).all(), f
        assert (ddst[f][0].download(with_padding=True)[:, ddst[f][0].row_bytes:] == 0xCD).all()
    lib.gmat_stream_destroy(st)
    lib.gmat_sws_freeContext(c)
    for f in range(n):
        for p in dsrc[f] + ddst[f]: p.free()


@pytest.mark.parametrize("case", [("nv12", "rgb24", 260, 64, 130, 32, 1),       # rows not 16-byte aligned: generic kernel
                                  ("nv12", "rgb24", 96, 40, 144, 60, 64),       # not 2:1
                                  ("rgb24", "bgra", 96, 40, 50, 30, 64)])       # RGB source
def test_batch_falls_back_frame_by_frame(dev, orc, case):
    sf, df, sw, sh, dw, dh, align = case
    k = _run_batch(dev, orc, sf, df, sw, sh, dw, dh, nframes=4, nstreams=2, align=align)
    assert not k.startswith("scale_yuv2"), k


def test_batch_single_frame_and_more_streams_than_frames(dev, orc):
    _run_batch(dev, orc, "nv12", "rgb24", 256, 64, 128, 32, nframes=1, nstreams=2, align=64)
    _run_batch(dev, orc, "nv12", "rgb24", 256, 64, 128, 32, nframes=3, nstreams=4, align=64)


@pytest.mark.gpu
@pytest.mark.parametrize("dst_fmt", ["rgb24", "nv12"])
def test_graph_replay_of_a_batch(dev, orc, strip_or_tiled, dst_fmt):
    if dev.kind != "hip":
        pytest.skip("graph capture needs the HIP runtime")
    k = _run_batch(dev, orc, "nv12", dst_fmt, 256, 64, 128, 32, nframes=6, nstreams=2, align=64, graph=True)
    want = _expected_2to1(strip_or_tiled, "nv12", dst_fmt)
    if want == "scale_yuv2s_kernel":                # each branch's share is 3 frames: a launch that small is the block form
        want = "scale_yuv2s_blk_kernel"
    assert k == want, k

