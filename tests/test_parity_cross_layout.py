"""NV12 <-> YUV420P SCALED (round 4): a hardware decoder's NV12 scaled for a planar consumer (scale_cuda=w:h:format=yuv420p) and the reverse.  No walker
but the 2:1 one writes the other chroma layout, so rounds 1-3 ran every such context on the tiled kernel (0.04 - 0.14 of the roofline).  Now the
context keeps a sibling in the SOURCE's layout — every walker applies — whose frame (the destination's size) yuv420_relayout_kernel moves into the
destination: libswscale's bytes (the chroma planes are filtered alike whichever way they are stored), the kernel named is the sibling's."""
import numpy as np
import pytest

from harness import SWS, synth_planes
from test_batch_api import _run_batch
from test_parity_strip import strip_rows  # noqa: F401

PAIRS = [("nv12", "yuv420p"), ("yuv420p", "nv12")]
# (geometry, the kernel of the sibling context for one frame / for a launch of five)
CASES = [((792, 78, 264, 26), "scale_yuv3x1_kernel", "scale_yuv3x1_kernel"), ((768, 72, 512, 48), "scale_yuv3x2_kernel", "scale_yuv3x2_kernel"),
         ((1056, 96, 264, 24), "scale_yuvg_blk_kernel", "scale_yuv4x1_kernel"), ((384, 216, 160, 90), "scale_yuvg_blk_kernel", "scale_yuvg_kernel"),
         ((160, 90, 240, 136), "scale_yuvu_kernel", "scale_yuvu_kernel"), ((128, 72, 384, 216), "scale_yuvu_kernel", "scale_yuvu_kernel"),
         ((384, 216, 288, 162), "scale_yuvg_blk_kernel", "scale_yuvu_kernel"), ((200, 120, 68, 42), "scale_yuvg_blk_kernel", "scale_yuvg_kernel")]


def _check(dev, orc, sf, df, geom, flags="bicubic", align=64, src_align=256):
    sw, sh, dw, dh = geom
    src = synth_planes(orc, sf, sw, sh, seed=57)
    want = orc.sws(src, sw, sh, sf, dw, dh, df, SWS[flags])
    d = dev.upload_planes(src, src_align)
    got, pads, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, SWS[flags], dst_align=align)
    for p in d:
        p.free()
    for i, (g, w) in enumerate(zip(got, want)):
        bad = np.argwhere(g != w)
        assert bad.size == 0, f"{kernel} {geom} {sf} -> {df} plane {i}: {len(bad)} mismatching bytes, first at {bad[:6].tolist()}"
        assert (pads[i] == 0xCD).all(), f"{kernel} plane {i}: wrote into the row padding"
    return kernel


@pytest.mark.parametrize("pair", PAIRS)
@pytest.mark.parametrize("case", CASES)
def test_scaled_between_the_layouts(dev, orc, strip_rows, pair, case, monkeypatch):
    strip_rows(0)
    monkeypatch.setenv("GMAT_STRIP_BLOCK", "3")           # the block forms by FRAMES (frames this small take them at every launch size under the shipped rule, round 5): five frames = the walkers
    geom, one, five = case
    assert _check(dev, orc, pair[0], pair[1], geom) == one
    k = _run_batch(dev, orc, pair[0], pair[1], *geom, nframes=5, nstreams=1, align=64)
    assert k == five, k


@pytest.mark.parametrize("pair", PAIRS)
def test_rule_clauses(dev, orc, strip_rows, pair, monkeypatch):
    """the 2:1 cross-layout walker keeps its frames; destination planes the re-layout kernel cannot move in 16 / 8 bytes, and the knob, leave the
    context on its own table (the tiled kernel); odd destination sizes, every algorithm the sibling's walkers take"""
    strip_rows(0)
    sf, df = pair
    assert _check(dev, orc, sf, df, (528, 52, 264, 26)) == "scale_yuv2px_kernel"
    assert _check(dev, orc, sf, df, (384, 216, 164, 90), align=4) == "scale19_kernel"                      # 164-byte luma rows: not 16-byte lines (round 6: the tile kernel on the 15-bit lines, in front of the tiled one)
    assert _check(dev, orc, sf, df, (384, 216, 162, 90), align=64) == "scale_yuvg_blk_kernel"                # an odd chroma width
    assert _check(dev, orc, sf, df, (384, 216, 161, 91), align=64).startswith(("scale_yuv", "scale19"))     # odd sizes: whatever serves them, the bytes
    for flags in ("bilinear", "lanczos", "area", "point", "gauss"):
        _check(dev, orc, sf, df, (384, 216, 160, 90), flags)
        _check(dev, orc, sf, df, (160, 90, 240, 136), flags)
    monkeypatch.setenv("GMAT_NO_CROSS_CASCADE", "1")
    assert _check(dev, orc, sf, df, (384, 216, 160, 90)) == "scale19_kernel"
    monkeypatch.setenv("GMAT_T15", "0")
    assert _check(dev, orc, sf, df, (384, 216, 160, 90)).startswith("scale_yuv_kernel")


def test_positions_and_ranges_reach_the_sibling(dev, orc):
    """gmat_sws_setRange / gmat_sws_setChromaPos on the context after its sibling exists: the sibling is rebuilt with them"""
    import ctypes as C
    from harness import PIX_FMT, planes, ints, alloc_planes
    lib, L = dev.lib, orc.L
    L.orc_sws_create_ex.restype = C.c_void_p
    L.orc_sws_create_ex.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    sw, sh, dw, dh = 384, 216, 160, 90
    src = synth_planes(orc, "nv12", sw, sh, seed=58)
    d = dev.upload_planes(src, 256)
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT["nv12"], dw, dh, PIX_FMT["yuv420p"], SWS["bicubic"], None)
    assert c
    for sr, dr, pos in ((0, 0, None), (0, 1, None), (1, 0, (0, 128, 0, 128)), (0, 0, (0, 128, 0, 128))):
        cp = pos or (-513, -513, -513, -513)
        oc = L.orc_sws_create_ex(sw, sh, PIX_FMT["nv12"], dw, dh, PIX_FMT["yuv420p"], SWS["bicubic"], None, (C.c_int * 4)(*cp), sr, dr)
        assert oc
        want = alloc_planes("yuv420p", dw, dh)
        assert L.orc_sws_scale(oc, planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                               planes([p.ctypes.data for p in want]), ints([p.strides[0] for p in want])) == dh
        L.orc_sws_free(oc)
        assert lib.gmat_sws_setChromaPos(c, *cp) == 0
        assert lib.gmat_sws_setRange(c, sr, dr) == 0
        dst = dev.planes_like("yuv420p", dw, dh, 64)
        assert lib.gmat_sws_scale(c, planes([p.ptr for p in d]), ints([p.stride for p in d]), 0, sh, planes([p.ptr for p in dst]), ints([p.stride for p in dst])) == dh
        for a, b in zip(dst, want):
            assert (a.download() == b).all(), (sr, dr, pos, lib.gmat_sws_lastKernel(c).decode())
        for p in dst:
            p.free()
    lib.gmat_sws_freeContext(c)
    for p in d:
        p.free()


def _alternating_streams(dev, orc, sf, df, geom, nframes, nstreams, fused=None, flags="bicubic"):
    """one gmat_sws_scale per frame, the caller alternating `nstreams` streams of its own (gmat_sws_setStream before each call) — what
    gmat_sws_scale_batch does with fewer than two frames a stream; returns the kernel and how often the context ordered one stream behind another"""
    import ctypes as C
    from harness import PIX_FMT, planes, ints
    lib = dev.lib
    sw, sh, dw, dh = geom
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], SWS[flags] | SWS["hwaccel"], None)
    assert c
    if fused is not None:
        assert lib.gmat_sws_setFused(c, fused) == 0
    srcs = [synth_planes(orc, sf, sw, sh, seed=900 + 11 * f) for f in range(nframes)]
    dsrc = [dev.upload_planes(s, 64) for s in srcs]
    ddst = [dev.planes_like(df, dw, dh, 64) for _ in range(nframes)]
    streams = []
    for _ in range(nstreams):
        h = C.c_void_p()
        assert lib.gmat_stream_create(C.byref(h)) == 0
        streams.append(h)
    for f in range(nframes):
        lib.gmat_sws_setStream(c, streams[f % nstreams])
        r = lib.gmat_sws_scale(c, planes([p.ptr for p in dsrc[f]]), ints([p.stride for p in dsrc[f]]), 0, sh,
                               planes([p.ptr for p in ddst[f]]), ints([p.stride for p in ddst[f]]))
        assert r == dh
    lib.gmat_device_sync()
    kernel, handoffs = lib.gmat_sws_lastKernel(c).decode(), lib.gmat_sws_streamHandoffs(c)
    for f in range(nframes):
        if (sw, sh) == (dw, dh) and df == "rgb24":
            want = [orc.yuv2rgb(srcs[f], sw, sh, sf, df)]                                                    # the same-size converter's rule
        else:
            want = (orc.chained if fused == 0 else orc.sws)(srcs[f], sw, sh, sf, dw, dh, df, SWS[flags])     # fused = 0: the reference's order of operations
        for i, (p, w) in enumerate(zip(ddst[f], want)):
            assert (p.download() == w).all(), (f, i, kernel)
    for h in streams:
        lib.gmat_stream_destroy(h)
    for f in range(nframes):
        for p in dsrc[f] + ddst[f]:
            p.free()
    lib.gmat_sws_freeContext(c)
    return kernel, handoffs


@pytest.mark.parametrize("pair", PAIRS)
def test_streams_share_the_cascade_frame(dev, orc, strip_rows, pair):
    """The cascade's frame (crossBuf) is ONE per context: calls on alternating streams are ordered behind each other by the context
    (stream_handoff_*, gsws.cpp).  Found by fuzz_walker on the GPU (two frames over two streams: the second frame's scale overwrote the frame
    the first one's re-layout was reading); the emulator runs a stream's work at once, so there it is the count of hand-offs that proves the rule."""
    strip_rows(0)
    geom = (1428, 248, 272, 58) if dev.kind == "hip" else (384, 216, 160, 90)
    for nstreams, nframes in ((2, 6), (3, 7)):
        kernel, handoffs = _alternating_streams(dev, orc, pair[0], pair[1], geom, nframes, nstreams)
        assert handoffs == nframes - 1, kernel                  # (the kernel named is the sibling's, whichever its table picks: the frame is shared either way)
    # through the batch entry point: fewer than two frames a stream (frame by frame over the streams), and shares of several frames
    for nframes, nstreams in ((2, 2), (3, 2), (5, 2), (9, 4)):
        _run_batch(dev, orc, pair[0], pair[1], *geom, nframes=nframes, nstreams=nstreams, align=64)
    # a context whose own table serves the pair (2:1) owns no frame: nothing to order
    kernel, handoffs = _alternating_streams(dev, orc, pair[0], pair[1], (528, 52, 264, 26), 4, 2)
    assert (kernel, handoffs) == ("scale_yuv2px_kernel", 0)


def test_streams_share_the_two_kernel_forms_intermediate(dev, orc, strip_rows):
    """the same rule for every context with an intermediate: the two-kernel form of a scaled yuv -> rgb context (fused = 0), a 16-bit path;
    the fused kernels and the same-size converters own nothing and are never ordered"""
    strip_rows(0)
    kernel, handoffs = _alternating_streams(dev, orc, "nv12", "rgb24", (384, 216, 192, 108), 5, 2, fused=0)
    assert handoffs == 4, kernel
    kernel, handoffs = _alternating_streams(dev, orc, "nv12", "rgb24", (384, 216, 192, 108), 5, 2, fused=2)
    assert handoffs == 0, kernel
    kernel, handoffs = _alternating_streams(dev, orc, "nv12", "rgb24", (384, 216, 384, 216), 5, 2)
    assert (kernel, handoffs) == ("yuv2rgb_kernel", 0)
    kernel, handoffs = _alternating_streams(dev, orc, "rgba", "nv12", (384, 216, 160, 90), 4, 2)
    assert handoffs in (0, 3), kernel
