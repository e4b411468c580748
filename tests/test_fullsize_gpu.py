"""BASELINE.json's full sizes on the GPU: bit-exact against the oracle where it finishes in seconds, plus
size-independent properties (involutions, linearity of re-layouts, padding canaries).  GPU only: the CPU
emulation would need minutes at these sizes."""
import ctypes as C

import numpy as np
import pytest

from harness import is_generic, PIX_FMT, SWS, synth_planes, DevPlane

pytestmark = pytest.mark.gpu


@pytest.fixture()
def gpu(dev):
    if dev.kind != "hip":
        pytest.skip("full-size tests run on the real GPU only")
    return dev


def test_cfg2_1080p_nv12_to_rgb24(gpu, orc):
    w, h = 1920, 1080
    src = synth_planes(orc, "nv12", w, h, seed=7)
    d = gpu.upload_planes(src, 256)
    got, pads, k = gpu.sws(d, w, h, "nv12", w, h, "rgb24", dst_align=256)
    assert k == "yuv2rgb_kernel" and (got[0] == orc.yuv2rgb(src, w, h, "nv12", "rgb24")).all()


@pytest.mark.parametrize("fused,oracle", [(2, "single"), (1, "chained"), (0, "chained")])
def test_cfg3_4k_nv12_to_1080p_rgb24_bicubic(gpu, orc, fused, oracle):
    sw, sh, dw, dh = 3840, 2160, 1920, 1080
    src = synth_planes(orc, "nv12", sw, sh, seed=11)
    want = (orc.sws(src, sw, sh, "nv12", dw, dh, "rgb24") if oracle == "single"
            else orc.chained(src, sw, sh, "nv12", dw, dh, "rgb24"))[0]
    d = gpu.upload_planes(src, 256)
    got, pads, k = gpu.sws(d, sw, sh, "nv12", dw, dh, "rgb24", fused=fused, dst_align=256)
    bad = np.argwhere(got[0] != want)
    assert bad.size == 0, f"{k}: {len(bad)} mismatching bytes, first {bad[:3].tolist()}"
    assert (pads[0] == 0xCD).all()
    assert k == {2: "scale_yuv2s_blk_kernel", 1: "scale_rgb2h_kernel<yuv>", 0: "scale_rgb2h_kernel"}[fused], k


@pytest.mark.parametrize("side,kernel", [(0, "scale_yuv2s_blk_kernel"), (1, "scale_yuv2s_kernel")])
def test_cfg3_launch_size_rule_at_full_size(gpu, orc, side, kernel, monkeypatch):
    """round 5: launches of up to twelve 4K -> 1080p frames (on the MI355X's 256 compute units: 17 wave-rows a wave slot, 24 slots a unit for the
    walker's 74 VGPRs) take the block-cooperative form, larger ones the walker (k_scale_yuv2s.hip yuv2s_block_form; profiles/r05f_blk_frames.txt) —
    two distinct frames repeated through ONE launch, every output against the oracle.  The last launch size of the block form is derived from the
    device's compute units (the literal 12 / 13 held on a 256-unit part only)"""
    monkeypatch.delenv("GMAT_STRIP_BLOCK", raising=False)
    monkeypatch.delenv("GMAT_STRIP_ROWS", raising=False)
    from harness import ints
    lib = gpu.lib
    cus = lib.gmat_device_compute_units(0)
    assert cus > 0
    nframes = min(17 * cus * 24 // (1080 * 8) + side, 32)            # 1080 rows x 8 strips of 256 columns a frame
    if side and nframes == 32:
        pytest.skip("every launch size takes the block form on this device")
    sw, sh, dw, dh = 3840, 2160, 1920, 1080
    srcs = [synth_planes(orc, "nv12", sw, sh, seed=21 + i) for i in range(2)]
    wants = [orc.sws(s, sw, sh, "nv12", dw, dh, "rgb24")[0] for s in srcs]
    dsrc = [gpu.upload_planes(s, 256) for s in srcs]
    ddst = [gpu.planes_like("rgb24", dw, dh, 256) for _ in range(nframes)]
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT["nv12"], dw, dh, PIX_FMT["rgb24"], SWS["bicubic"], None)
    assert c
    sp, dp = (C.c_void_p * (4 * nframes))(), (C.c_void_p * (4 * nframes))()
    for f in range(nframes):
        for i, p in enumerate(dsrc[f & 1]):
            sp[4 * f + i] = p.ptr
        dp[4 * f] = ddst[f][0].ptr
    st = (C.c_void_p * 1)(None)
    assert lib.gmat_sws_scale_batch(c, nframes, C.cast(sp, C.POINTER(C.c_void_p)), ints([p.stride for p in dsrc[0]]), C.cast(dp, C.POINTER(C.c_void_p)),
                                    ints([ddst[0][0].stride]), C.cast(st, C.POINTER(C.c_void_p)), 1, 0) == nframes
    lib.gmat_device_sync()
    assert lib.gmat_sws_lastKernel(c).decode() == kernel and lib.gmat_sws_lastLaunchFrames(c) == nframes
    for f in range(nframes):
        assert (ddst[f][0].download() == wants[f & 1]).all(), f
    lib.gmat_sws_freeContext(c)
    for fr in dsrc + ddst:
        for p in fr:
            p.free()


def test_cfg3_rgb24_4k_to_1080p_lanczos(gpu, orc):
    sw, sh, dw, dh = 3840, 2160, 1920, 1080
    src = synth_planes(orc, "rgb24", sw, sh, seed=13)
    want = orc.sws(src, sw, sh, "rgb24", dw, dh, "rgb24", SWS["lanczos"])[0]
    d = gpu.upload_planes(src, 256)
    got, _, _ = gpu.sws(d, sw, sh, "rgb24", dw, dh, "rgb24", SWS["lanczos"], dst_align=256)
    assert (got[0] == want).all()


def test_cfg4_4k_rotate_flip_smooth(gpu, orc):
    w, h, bpp = 3840, 2160, 3
    src = orc.lcg((h, w * bpp), 17)
    a = np.zeros((w, h * bpp), np.uint8)
    orc.L.orc_transpose(src.ctypes.data, src.strides[0], a.ctypes.data, a.strides[0], w, h, bpp, 1)
    b = np.zeros_like(a)
    orc.L.orc_hflip(a.ctypes.data, a.strides[0], b.ctypes.data, b.strides[0], h, w, bpp)
    want = np.zeros_like(a)
    m = (C.c_int * 9)(1, 2, 1, 2, 4, 2, 1, 2, 1)
    orc.L.orc_conv3x3(b.ctypes.data, b.strides[0], want.ctypes.data, want.strides[0], h, w, bpp, m, 1 / 16, 0.0)
    lib = gpu.lib
    d = gpu.upload_planes([src], 256)[0]
    s1 = DevPlane(gpu, w, h * bpp, (h * bpp + 255) // 256 * 256)
    s2 = DevPlane(gpu, w, h * bpp, s1.stride)
    s3 = DevPlane(gpu, w, h * bpp, s1.stride)
    fused = DevPlane(gpu, w, h * bpp, s1.stride)
    assert lib.gmat_transpose(d.ptr, d.stride, s1.ptr, s1.stride, w, h, bpp, 1, None) == 0
    assert lib.gmat_flip(s1.ptr, s1.stride, s2.ptr, s2.stride, h, w, bpp, 1, None) == 0
    assert lib.gmat_smooth3x3(s2.ptr, s2.stride, s3.ptr, s3.stride, h, w, bpp, m, 1 / 16, 0.0, None) == 0
    assert lib.gmat_rotate_flip_smooth(d.ptr, d.stride, fused.ptr, fused.stride, w, h, bpp, None) == 0
    assert (s3.download() == want).all(), "three-filter chain differs from the CPU filters"
    assert (fused.download() == want).all(), "fused kernel differs from the CPU filters"
    # involutions: transpose(clock) then transpose(cclock) and double flips give the source back
    back = DevPlane(gpu, h, w * bpp, d.stride)
    assert lib.gmat_transpose(s1.ptr, s1.stride, back.ptr, back.stride, h, w, bpp, 2, None) == 0
    assert (back.download() == src).all()
    f1 = DevPlane(gpu, h, w * bpp, d.stride); f2 = DevPlane(gpu, h, w * bpp, d.stride)
    for code in (0, 1, -1):
        assert lib.gmat_flip(d.ptr, d.stride, f1.ptr, f1.stride, w, h, bpp, code, None) == 0
        assert lib.gmat_flip(f1.ptr, f1.stride, f2.ptr, f2.stride, w, h, bpp, code, None) == 0
        assert (f2.download() == src).all()


def test_rgb_to_nv12_4k_and_relayout_round_trip(gpu, orc):
    w, h = 3840, 2160
    src = synth_planes(orc, "rgb24", w, h, seed=19)
    want = orc.sws(src, w, h, "rgb24", w, h, "nv12")
    d = gpu.upload_planes(src, 256)
    got, _, _ = gpu.sws(d, w, h, "rgb24", w, h, "nv12", dst_align=256)
    assert all((g == wv).all() for g, wv in zip(got, want))
    # nv12 -> yuv420p -> nv12 is the identity
    dn = gpu.upload_planes(got, 256)
    p, _, _ = gpu.sws(dn, w, h, "nv12", w, h, "yuv420p", dst_align=256)
    dp = gpu.upload_planes(p, 256)
    n2, _, _ = gpu.sws(dp, w, h, "yuv420p", w, h, "nv12", dst_align=256)
    assert all((a == b).all() for a, b in zip(n2, got))


def _oracle_rows(orc, src, sw, sh, src_fmt, dw, dh, dst_fmt, y0, y1, flags=SWS["bicubic"]):
    """rows [y0, y1) of the oracle's output (orc_sws_scale_rows), full-size planes returned"""
    from harness import alloc_planes, planes, ints
    L = orc.L
    c = L.orc_sws_create(sw, sh, PIX_FMT[src_fmt], dw, dh, PIX_FMT[dst_fmt], flags, None)
    assert c
    outs = alloc_planes(dst_fmt, dw, dh)
    r = L.orc_sws_scale_rows(c, planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                             planes([p.ctypes.data for p in outs]), ints([p.strides[0] for p in outs]), y0, y1)
    L.orc_sws_free(c)
    assert r == y1 - y0
    return outs


def test_8k_nv12_to_4k_rgb24_bands(gpu, orc, kern):
    """maximum size of the path (8K UHD): three 32-row bands of the 2:1 scaler's output against the oracle"""
    sw, sh, dw, dh = 7680, 4320, 3840, 2160
    src = synth_planes(orc, "nv12", sw, sh, seed=19)
    d = gpu.upload_planes(src, 256)
    got, pads, k = gpu.sws(d, sw, sh, "nv12", dw, dh, "rgb24", dst_align=256)
    assert k == kern and (pads[0] == 0xCD).all()
    for y0 in (0, 1072, dh - 32):
        want = _oracle_rows(orc, src, sw, sh, "nv12", dw, dh, "rgb24", y0, y0 + 32)[0]
        assert (got[0][y0:y0 + 32] == want[y0:y0 + 32]).all(), y0


@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
def test_8k_to_4k_transcode_bands(gpu, orc, fmt):
    """maximum size on the plane-walking 4:2:0 -> 4:2:0 kernel (32-bit row offsets: 4320 rows x 7680 bytes): three bands of
    every plane against the oracle"""
    sw, sh, dw, dh = 7680, 4320, 3840, 2160
    src = synth_planes(orc, fmt, sw, sh, seed=29)
    d = gpu.upload_planes(src, 256)
    got, pads, k = gpu.sws(d, sw, sh, fmt, dw, dh, fmt, dst_align=256)
    assert k == "scale_yuv2p_kernel" and all((pd == 0xCD).all() for pd in pads)
    for y0 in (0, 1072, dh - 32):                       # even starts: 4:2:0 rows come in pairs sharing a chroma row
        want = _oracle_rows(orc, src, sw, sh, fmt, dw, dh, fmt, y0, y0 + 32)
        assert (got[0][y0:y0 + 32] == want[0][y0:y0 + 32]).all(), y0
        for pl in range(1, len(got)):
            assert (got[pl][y0 // 2:y0 // 2 + 16] == want[pl][y0 // 2:y0 // 2 + 16]).all(), (pl, y0)


@pytest.mark.parametrize("fmt", ["p010le", "yuv420p10le"])
def test_4k_to_1080p_10bit_transcode(gpu, orc, fmt):
    """the HDR transcode down-scale at full size on the 10-bit plane-walking kernel: every plane against the oracle"""
    sw, sh, dw, dh = 3840, 2160, 1920, 1080
    src = synth_planes(orc, fmt, sw, sh, seed=41)
    if fmt == "yuv420p10le":
        for p in src:
            p.view("<u2")[...] &= 0x3FF
    want = orc.sws(src, sw, sh, fmt, dw, dh, fmt)
    d = gpu.upload_planes(src, 256)
    got, pads, k = gpu.sws(d, sw, sh, fmt, dw, dh, fmt, dst_align=256)
    assert k == "scale_yuv2p16_kernel"
    for g, wv, pd in zip(got, want, pads):
        assert (g == wv).all() and (pd == 0xCD).all()


def test_8k_rgb24_to_4k_bands(gpu, orc):
    """maximum size on the packed-RGB strip kernel (99.5 MB source frame)"""
    sw, sh, dw, dh = 7680, 4320, 3840, 2160
    src = synth_planes(orc, "rgb24", sw, sh, seed=31)
    d = gpu.upload_planes(src, 256)
    got, pads, k = gpu.sws(d, sw, sh, "rgb24", dw, dh, "bgra", dst_align=256)
    assert k == "scale_rgb2h_kernel" and (pads[0] == 0xCD).all()
    for y0 in (0, 1072, dh - 32):
        want = _oracle_rows(orc, src, sw, sh, "rgb24", dw, dh, "bgra", y0, y0 + 32)[0]
        assert (got[0][y0:y0 + 32] == want[y0:y0 + 32]).all(), y0


@pytest.mark.parametrize("bpp", [3, 4])
def test_8k_rotate_flip_smooth(gpu, orc, bpp):
    """maximum size on smooth121_kernel, fused with the transpose (row offsets up to 4320 x 30720 bytes = 132 MB for rgba)"""
    import ctypes as C
    from harness import DevPlane
    w, h = 7680, 4320
    src = orc.lcg((h, w * bpp), 37)
    a = np.zeros((w, h * bpp), np.uint8); b = np.zeros((w, h * bpp), np.uint8); want = np.zeros((w, h * bpp), np.uint8)
    m = (C.c_int * 9)(1, 2, 1, 2, 4, 2, 1, 2, 1)
    orc.L.orc_transpose(src.ctypes.data, src.strides[0], a.ctypes.data, a.strides[0], w, h, bpp, 1)
    orc.L.orc_hflip(a.ctypes.data, a.strides[0], b.ctypes.data, b.strides[0], h, w, bpp)
    orc.L.orc_conv3x3(b.ctypes.data, b.strides[0], want.ctypes.data, want.strides[0], h, w, bpp, m, 1 / 16, 0.0)
    d = gpu.upload_planes([src], 256)[0]
    o = DevPlane(gpu, w, h * bpp, (h * bpp + 255) // 256 * 256)
    assert gpu.lib.gmat_rotate_flip_smooth(d.ptr, d.stride, o.ptr, o.stride, w, h, bpp, None) == 0
    assert (o.download() == want).all()
    assert (o.download(True)[:, o.row_bytes:] == 0xCD).all()
    d.free(); o.free()


@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("which", ["strip", "tiled"])
def test_4k_to_1080p_transcode(gpu, orc, fmt, which, monkeypatch):
    """the transcode down-scale at full size on BOTH kernels that serve it (the plane-walking one by default, the tiled one
    for the frames it declines); yuv420p exercises the two separate chroma planes"""
    if which == "tiled":
        monkeypatch.setenv("GMAT_SCALE_NO_STRIP", "1")
    else:
        monkeypatch.delenv("GMAT_SCALE_NO_STRIP", raising=False)
    sw, sh, dw, dh = 3840, 2160, 1920, 1080
    src = synth_planes(orc, fmt, sw, sh, seed=23)
    want = orc.sws(src, sw, sh, fmt, dw, dh, fmt)
    d = gpu.upload_planes(src, 256)
    got, pads, k = gpu.sws(d, sw, sh, fmt, dw, dh, fmt, dst_align=256)
    assert k == ("scale_yuv2p_kernel" if which == "strip" else "scale_yuv2x_kernel<yuv>")
    for g, wv, pd in zip(got, want, pads):
        assert (g == wv).all() and (pd == 0xCD).all()


@pytest.mark.parametrize("fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("which", ["strip", "generic"])
def test_4k_to_720p_transcode(gpu, orc, fmt, which, monkeypatch):
    """the 3:1 ladder step at full size on BOTH kernels that serve it (the 3:1 strip walker by default, the generic plane
    scaler for the frames it declines): every plane against the oracle"""
    if which == "generic":
        monkeypatch.setenv("GMAT_SCALE_NO_STRIP", "1")
    else:
        monkeypatch.delenv("GMAT_SCALE_NO_STRIP", raising=False)
    sw, sh, dw, dh = 3840, 2160, 1280, 720
    src = synth_planes(orc, fmt, sw, sh, seed=43)
    want = orc.sws(src, sw, sh, fmt, dw, dh, fmt)
    d = gpu.upload_planes(src, 256)
    got, pads, k = gpu.sws(d, sw, sh, fmt, dw, dh, fmt, dst_align=256)
    assert k == "scale_yuv3x1_kernel" if which == "strip" else is_generic(k), k
    for g, wv, pd in zip(got, want, pads):
        assert (g == wv).all() and (pd == 0xCD).all()


@pytest.mark.parametrize("fmt,geom", [("nv12", (1920, 1080, 1280, 720)), ("yuv420p", (1920, 1080, 1280, 720)), ("nv12", (3840, 2160, 2560, 1440))])
@pytest.mark.parametrize("which", ["strip", "generic"])
def test_3_to_2_ladder_steps(gpu, orc, fmt, geom, which, monkeypatch):
    """1080p -> 720p and 4K -> 1440p at full size on BOTH kernels that serve them (the 3:2 strip walker by default, the generic plane
    scaler for the frames it declines): every plane against the oracle"""
    if which == "generic":
        monkeypatch.setenv("GMAT_SCALE_NO_STRIP", "1")
    else:
        monkeypatch.delenv("GMAT_SCALE_NO_STRIP", raising=False)
    sw, sh, dw, dh = geom
    src = synth_planes(orc, fmt, sw, sh, seed=47)
    want = orc.sws(src, sw, sh, fmt, dw, dh, fmt)
    d = gpu.upload_planes(src, 256)
    got, pads, k = gpu.sws(d, sw, sh, fmt, dw, dh, fmt, dst_align=256)
    assert k == "scale_yuv3x2_kernel" if which == "strip" else is_generic(k), k
    for g, wv, pd in zip(got, want, pads):
        assert (g == wv).all() and (pd == 0xCD).all()


@pytest.mark.parametrize("fmt,geom", [("nv12", (3840, 2160, 960, 540)), ("yuv420p", (3840, 2160, 960, 540))])
@pytest.mark.parametrize("first", ["ratio", "blk"])
def test_4k_to_540p(gpu, orc, monkeypatch, first, fmt, geom):
    """the 540p rung of a ladder from a 4K frame, every plane against the oracle"""
    # one frame per call: the exact-ratio walker (GMAT_BLOCK_FIRST=0) and the band walker's block-cooperative form the shipped rule puts in front of it
    monkeypatch.setenv("GMAT_BLOCK_FIRST", "0" if first == "ratio" else "1")
    sw, sh, dw, dh = geom
    src = synth_planes(orc, fmt, sw, sh, seed=71)
    want = orc.sws(src, sw, sh, fmt, dw, dh, fmt)
    d = gpu.upload_planes(src, 256)
    got, pads, k = gpu.sws(d, sw, sh, fmt, dw, dh, fmt, dst_align=256)
    assert k == ("scale_yuv4x1_kernel" if first == "ratio" else "scale_yuvg_blk_kernel"), k
    for g, wv, pd in zip(got, want, pads):
        assert (g == wv).all() and (pd == 0xCD).all()


@pytest.mark.parametrize("geom", [(3840, 2160, 960, 540, "rgb24"), (1920, 1080, 480, 270, "bgra")])
@pytest.mark.parametrize("first", ["ratio", "blk"])
def test_nv12_to_a_quarter_in_rgb(gpu, orc, monkeypatch, first, geom):
    """4K -> 960x540 and 1080p -> 480x270 from NV12 into packed RGB at full size"""
    # one frame per call: the exact-ratio walker (GMAT_BLOCK_FIRST=0) and the band walker's block-cooperative form the shipped rule puts in front of it
    monkeypatch.setenv("GMAT_BLOCK_FIRST", "0" if first == "ratio" else "1")
    sw, sh, dw, dh, df = geom
    src = synth_planes(orc, "nv12", sw, sh, seed=67)
    want = orc.sws(src, sw, sh, "nv12", dw, dh, df)
    d = gpu.upload_planes(src, 256)
    got, pads, k = gpu.sws(d, sw, sh, "nv12", dw, dh, df, dst_align=256)
    assert k == ("scale_yuv4r_kernel" if first == "ratio" else "scale_yuvg_blk_kernel"), k
    for g, wv, pd in zip(got, want, pads):
        assert (g == wv).all() and (pd == 0xCD).all()


@pytest.mark.parametrize("geom", [(1920, 1080, 1280, 720, "rgb24"), (3840, 2160, 2560, 1440, "bgra")])
@pytest.mark.parametrize("first", ["ratio", "blk"])
def test_nv12_to_two_thirds_in_rgb(gpu, orc, monkeypatch, first, geom):
    """1080p -> 720p and 4K -> 1440p from NV12 into packed RGB at full size"""
    # one frame per call: the exact-ratio walker (GMAT_BLOCK_FIRST=0) and the band walker's block-cooperative form the shipped rule puts in front of it
    monkeypatch.setenv("GMAT_BLOCK_FIRST", "0" if first == "ratio" else "1")
    sw, sh, dw, dh, df = geom
    src = synth_planes(orc, "nv12", sw, sh, seed=61)
    want = orc.sws(src, sw, sh, "nv12", dw, dh, df)
    d = gpu.upload_planes(src, 256)
    got, pads, k = gpu.sws(d, sw, sh, "nv12", dw, dh, df, dst_align=256)
    assert k == ("scale_yuv32r_kernel" if first == "ratio" else "scale_yuvg_blk_kernel"), k
    for g, wv, pd in zip(got, want, pads):
        assert (g == wv).all() and (pd == 0xCD).all()


@pytest.mark.parametrize("geom", [(3840, 2160, 1280, 720, "rgb24"), (1920, 1080, 640, 360, "bgra")])
@pytest.mark.parametrize("first", ["ratio", "blk"])
def test_nv12_to_a_third_in_rgb(gpu, orc, monkeypatch, first, geom):
    """4K -> 720p and 1080p -> 360p from NV12 into packed RGB (a decoder's frame into a network's input) at full size"""
    # one frame per call: the exact-ratio walker (GMAT_BLOCK_FIRST=0) and the band walker's block-cooperative form the shipped rule puts in front of it
    monkeypatch.setenv("GMAT_BLOCK_FIRST", "0" if first == "ratio" else "1")
    sw, sh, dw, dh, df = geom
    src = synth_planes(orc, "nv12", sw, sh, seed=59)
    want = orc.sws(src, sw, sh, "nv12", dw, dh, df)
    d = gpu.upload_planes(src, 256)
    got, pads, k = gpu.sws(d, sw, sh, "nv12", dw, dh, df, dst_align=256)
    assert k == ("scale_yuv3r_kernel" if first == "ratio" else "scale_yuvg_blk_kernel"), k
    for g, wv, pd in zip(got, want, pads):
        assert (g == wv).all() and (pd == 0xCD).all()


@pytest.mark.parametrize("fmts", [("rgb24", "nv12"), ("bgr24", "yuv420p")])
def test_4k_rgb_to_1080p_420(gpu, orc, fmts):
    """a 4K packed RGB frame into a 1080p 4:2:0 one (encoder input) at full size, every plane against the oracle"""
    sw, sh, dw, dh = 3840, 2160, 1920, 1080
    src = synth_planes(orc, fmts[0], sw, sh, seed=53)
    want = orc.sws(src, sw, sh, fmts[0], dw, dh, fmts[1])
    d = gpu.upload_planes(src, 256)
    got, pads, k = gpu.sws(d, sw, sh, fmts[0], dw, dh, fmts[1], dst_align=256)
    assert k == "scale_rgb2y_kernel", k
    for g, wv, pd in zip(got, want, pads):
        assert (g == wv).all() and (pd == 0xCD).all()


def test_4k_rotate_17_degrees(gpu, orc):
    import math
    w, h, bpp = 3840, 2160, 3
    src = orc.lcg((h, w * bpp), 29)
    fill = np.array([0, 0, 0, 255], np.uint8)
    want = np.zeros((h, w * bpp), np.uint8)
    orc.L.orc_rotate(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, w, h, bpp,
                     math.radians(17.0), 1, fill.ctypes.data)
    d = gpu.upload_planes([src], 256)[0]
    o = DevPlane(gpu, h, w * bpp, d.stride)
    assert gpu.lib.gmat_rotate(d.ptr, d.stride, o.ptr, o.stride, w, h, w, h, bpp, math.radians(17.0), 1,
                               fill.ctypes.data, None) == 0
    assert (o.download() == want).all()


@pytest.mark.parametrize("dst_fmt", ["rgb24", "nv12"])
def test_4k_batched_launch_equals_single_launches(gpu, orc, dst_fmt):
    """BASELINE configs[2] through gmat_sws_scale_batch: 5 frames on 2 streams (launches of 3 and 2 frames, grid.y =
    frame) must give, frame by frame, the bytes of gmat_sws_scale on the same frame (which the tests above check
    against the oracle)"""
    from harness import ints
    lib = gpu.lib
    sw, sh, dw, dh, n = 3840, 2160, 1920, 1080, 5
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT["nv12"], dw, dh, PIX_FMT[dst_fmt], SWS["bicubic"], None)
    assert c
    srcs = [synth_planes(orc, "nv12", sw, sh, seed=500 + f) for f in range(n)]
    dsrc = [gpu.upload_planes(s, 256) for s in srcs]
    single = []
    for f in range(n):
        got, _, k = gpu.sws(dsrc[f], sw, sh, "nv12", dw, dh, dst_fmt, dst_align=256)
        assert k.startswith("scale_yuv2")
        single.append(got)
    ddst = [gpu.planes_like(dst_fmt, dw, dh, 256) for _ in range(n)]
    sp = (C.c_void_p * (4 * n))(); dp = (C.c_void_p * (4 * n))()
    for f in range(n):
        for i, p in enumerate(dsrc[f]): sp[4 * f + i] = p.ptr
        for i, p in enumerate(ddst[f]): dp[4 * f + i] = p.ptr
    streams = (C.c_void_p * 2)()
    for s in range(2):
        h = C.c_void_p(); assert lib.gmat_stream_create(C.byref(h)) == 0; streams[s] = h
    r = lib.gmat_sws_scale_batch(c, n, C.cast(sp, C.POINTER(C.c_void_p)), ints([p.stride for p in dsrc[0]]),
                                 C.cast(dp, C.POINTER(C.c_void_p)), ints([p.stride for p in ddst[0]]),
                                 C.cast(streams, C.POINTER(C.c_void_p)), 2, 3)
    assert r == n and lib.gmat_sws_lastLaunchFrames(c) == 2
    lib.gmat_stream_sync(streams[0])
    for f in range(n):
        for i, p in enumerate(ddst[f]):
            assert (p.download() == single[f][i]).all(), (f, i)
            assert (p.download(with_padding=True)[:, p.row_bytes:] == 0xCD).all()
    for s in range(2):
        lib.gmat_stream_destroy(streams[s])
    lib.gmat_sws_freeContext(c)
    for f in range(n):
        for p in dsrc[f] + ddst[f]:
            p.free()


@pytest.mark.parametrize("which", ["walker", "tiled"])
@pytest.mark.parametrize("df", ["rgb24", "nv12"])
@pytest.mark.parametrize("geom", [(3840, 2160, 1600, 900), (3840, 2160, 1366, 768), (3840, 2160, 854, 480), (1920, 1080, 768, 432)])
@pytest.mark.parametrize("form", ["walk", "blk"])
def test_any_ratio_full_size(gpu, orc, monkeypatch, which, df, geom, form):
    """4K -> 1600x900 / 1366x768 / 854x480 and 1080p -> 768x432, to nv12 and to rgb24, on the polyphase
    band walker (scale_yuvg_kernel) and on the tiled plane scaler behind it — both bit-exact with one libswscale context"""
    if which == "tiled":
        if form == "blk":
            pytest.skip("one tiled kernel")
        monkeypatch.setenv("GMAT_SCALE_NO_GENERIC_WALKER", "1")
    monkeypatch.setenv("GMAT_STRIP_BLOCK", "0" if form == "walk" else "3")      # a single frame: the walker / its block-cooperative form (round 4)
    sw, sh, dw, dh = geom
    src = synth_planes(orc, "nv12", sw, sh, seed=61)
    want = orc.sws(src, sw, sh, "nv12", dw, dh, df)
    d = gpu.upload_planes(src, 256)
    got, pads, k = gpu.sws(d, sw, sh, "nv12", dw, dh, df, dst_align=256)
    for i, (g, w) in enumerate(zip(got, want)):
        bad = np.argwhere(g != w)
        assert bad.size == 0, f"{k} plane {i}: {len(bad)} mismatching bytes, first {bad[:3].tolist()}"
        assert (pads[i] == 0xCD).all()
    assert (k == ("scale_yuvg_kernel" if form == "walk" else "scale_yuvg_blk_kernel")) == (which == "walker"), k
    for p in d:
        p.free()
