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
def test_scaled_between_the_layouts(dev, orc, strip_rows, pair, case):
    strip_rows(0)
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
    assert _check(dev, orc, sf, df, (384, 216, 164, 90), align=4).startswith("scale_yuv_kernel")          # 164-byte luma rows: not 16-byte lines
    assert _check(dev, orc, sf, df, (384, 216, 162, 90), align=64) == "scale_yuvg_blk_kernel"                # an odd chroma width
    assert _check(dev, orc, sf, df, (384, 216, 161, 91), align=64).startswith("scale_yuv")                   # odd sizes: whatever serves them, the bytes
    for flags in ("bilinear", "lanczos", "area", "point", "gauss"):
        _check(dev, orc, sf, df, (384, 216, 160, 90), flags)
        _check(dev, orc, sf, df, (160, 90, 240, 136), flags)
    monkeypatch.setenv("GMAT_NO_CROSS_CASCADE", "1")
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
