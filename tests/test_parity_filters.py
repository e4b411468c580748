"""crop / flip / transpose / rotate / smooth kernels and the AVFilter-shaped layer vs the oracle."""
import ctypes as C

import os

import numpy as np
import pytest

from harness import PIX_FMT, synth_planes, DevPlane
from gmat_amd.lib import GmatFrame

SIZES = [(64, 64), (200, 70), (67, 129), (5, 3), (1, 1), (130, 66)]


def _orc_out(rows, row_bytes):
    return np.zeros((rows, row_bytes), np.uint8)


@pytest.mark.parametrize("w,h", SIZES + [(256, 128), (300, 140), (128, 192)])
@pytest.mark.parametrize("bpp", [1, 2, 3, 4])
@pytest.mark.parametrize("dir", [0, 1, 2, 3])
@pytest.mark.parametrize("tile", [0, 64, 128])
def test_transpose(dev, orc, monkeypatch, w, h, bpp, dir, tile):
    """tile: the 1- and 2-byte plane paths pick 64 x 64 or 128 x 128 tiles by the plane's size; GMAT_TRANSPOSE_TILE forces either"""
    if tile:
        if bpp > 2:
            pytest.skip("only 1- and 2-byte planes have two tile sizes")
        monkeypatch.setenv("GMAT_TRANSPOSE_TILE", str(tile))
    dev.lib.gmat_knobs_reload()                              # the stateless launchers read the environment on request, not per call
    src = orc.lcg((h, w * bpp), 5)
    want = _orc_out(w, h * bpp)
    orc.L.orc_transpose(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, bpp, dir)
    for align, extra in [(256, 0), (1, 1)]:
        d = dev.upload_planes([src], align, extra)[0]
        o = DevPlane(dev, w, h * bpp, (h * bpp + extra + align - 1) // align * align)
        assert dev.lib.gmat_transpose(d.ptr, d.stride, o.ptr, o.stride, w, h, bpp, dir, None) == 0
        assert (o.download() == want).all()
        assert (o.download(True)[:, o.row_bytes:] == 0xCD).all()
        d.free(); o.free()


@pytest.mark.parametrize("w,h", SIZES + [(300, 5), (513, 4)])
@pytest.mark.parametrize("bpp", [1, 2, 3, 4])
@pytest.mark.parametrize("code", [0, 1, -1])
def test_flip(dev, orc, w, h, bpp, code):
    src = orc.lcg((h, w * bpp), 6)
    tmp = src
    if code != 0:
        t = _orc_out(h, w * bpp)
        orc.L.orc_hflip(tmp.ctypes.data, tmp.strides[0], t.ctypes.data, t.strides[0], w, h, bpp)
        tmp = t
    if code <= 0:
        t = _orc_out(h, w * bpp)
        orc.L.orc_vflip(tmp.ctypes.data, tmp.strides[0], t.ctypes.data, t.strides[0], w, h, bpp)
        tmp = t
    for align, extra in [(256, 0), (1, 2)]:
        d = dev.upload_planes([src], align, extra)[0]
        o = DevPlane(dev, h, w * bpp, (w * bpp + extra + align - 1) // align * align)
        assert dev.lib.gmat_flip(d.ptr, d.stride, o.ptr, o.stride, w, h, bpp, code, None) == 0
        assert (o.download() == tmp).all()
        assert (o.download(True)[:, o.row_bytes:] == 0xCD).all()
        d.free(); o.free()


@pytest.fixture(params=["separable", "general"])
def smooth_kern(request, dev):
    """the 1 2 1 / 2 4 2 / 1 2 1 matrix on dword-aligned frames takes smooth121_kernel; GMAT_NO_SMOOTH121 keeps the general
    conv3x3_kernel (the path of every other matrix and of unaligned frames) so both are compared with the oracle"""
    old = os.environ.pop("GMAT_NO_SMOOTH121", None)
    if request.param == "general":
        os.environ["GMAT_NO_SMOOTH121"] = "1"
    dev.lib.gmat_knobs_reload()
    yield request.param
    os.environ.pop("GMAT_NO_SMOOTH121", None)
    if old is not None:
        os.environ["GMAT_NO_SMOOTH121"] = old


@pytest.mark.parametrize("w,h", SIZES + [(300, 5), (256, 130), (196, 17), (4, 1), (8, 200), (52, 64)])
@pytest.mark.parametrize("bpp", [1, 2, 3, 4])
def test_smooth3x3(dev, orc, w, h, bpp, smooth_kern):
    src = orc.lcg((h, w * bpp), 8)
    m = (C.c_int * 9)(1, 2, 1, 2, 4, 2, 1, 2, 1)
    want = _orc_out(h, w * bpp)
    orc.L.orc_conv3x3(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, bpp, m, 1 / 16, 0.0)
    for align, extra in [(256, 0), (1, 3)]:
        d = dev.upload_planes([src], align, extra)[0]
        o = DevPlane(dev, h, w * bpp, (w * bpp + extra + align - 1) // align * align)
        assert dev.lib.gmat_smooth3x3(d.ptr, d.stride, o.ptr, o.stride, w, h, bpp, m, 1 / 16, 0.0, None) == 0
        assert (o.download() == want).all()
        assert (o.download(True)[:, o.row_bytes:] == 0xCD).all()
        d.free(); o.free()


def test_conv3x3_general_matrix_and_rounding(dev, orc):
    w, h, bpp = 97, 33, 3
    src = orc.lcg((h, w * bpp), 12)
    m = (C.c_int * 9)(-1, -2, 3, 0, 5, -1, 2, 1, -3)
    want = _orc_out(h, w * bpp)
    orc.L.orc_conv3x3(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, bpp, m, 0.3, 7.25)
    d = dev.upload_planes([src], 64)[0]
    o = DevPlane(dev, h, w * bpp, 320)
    assert dev.lib.gmat_smooth3x3(d.ptr, d.stride, o.ptr, o.stride, w, h, bpp, m, 0.3, 7.25, None) == 0
    assert (o.download() == want).all()


@pytest.mark.parametrize("w,h", [(64, 64), (200, 70), (67, 129), (3, 2), (256, 130), (196, 17), (4, 1), (8, 200), (52, 64), (128, 48)])
@pytest.mark.parametrize("bpp", [3, 4])
def test_rotate_flip_smooth_fused_equals_three_filters(dev, orc, w, h, bpp, smooth_kern):
    """cfg4: transpose(clock) -> hflip -> 3x3 smooth as three oracle filters == one fused kernel."""
    src = orc.lcg((h, w * bpp), 9)
    a = _orc_out(w, h * bpp)
    orc.L.orc_transpose(src.ctypes.data, src.strides[0], a.ctypes.data, a.strides[0], w, h, bpp, 1)
    b = _orc_out(w, h * bpp)
    orc.L.orc_hflip(a.ctypes.data, a.strides[0], b.ctypes.data, b.strides[0], h, w, bpp)
    want = _orc_out(w, h * bpp)
    m = (C.c_int * 9)(1, 2, 1, 2, 4, 2, 1, 2, 1)
    orc.L.orc_conv3x3(b.ctypes.data, b.strides[0], want.ctypes.data, want.strides[0], h, w, bpp, m, 1 / 16, 0.0)
    d = dev.upload_planes([src], 256)[0]
    o = DevPlane(dev, w, h * bpp, (h * bpp + 255) // 256 * 256)
    assert dev.lib.gmat_rotate_flip_smooth(d.ptr, d.stride, o.ptr, o.stride, w, h, bpp, None) == 0
    assert (o.download() == want).all()
    assert (o.download(True)[:, o.row_bytes:] == 0xCD).all()


def test_crop(dev, orc):
    w, h, bpp = 120, 50, 3
    src = orc.lcg((h, w * bpp), 4)
    x, y, cw, ch = 13, 7, 77, 31
    d = dev.upload_planes([src], 64)[0]
    o = DevPlane(dev, ch, cw * bpp, 256)
    assert dev.lib.gmat_crop(d.ptr, d.stride, o.ptr, o.stride, x, y, cw, ch, bpp, None) == 0
    assert (o.download() == src[y:y + ch, x * bpp:(x + cw) * bpp]).all()


@pytest.mark.parametrize("bpp", [1, 2, 3, 4])
def test_crop_offsets_of_every_alignment(dev, orc, bpp):
    """crop = a strided copy: copy2d_kernel (16 bytes a lane from a source of any alignment, streaming stores, 16-byte aligned
    destinations with rows of 64 bytes and more) and the runtime's 2-D copy for the rest — x offsets 0..17 (every byte alignment of
    the source), widths that end inside a 16-byte piece, the window touching the frame's last byte, destination pitches aligned and
    not; the padding of the destination rows stays untouched"""
    w, h = 301, 37
    src = orc.lcg((h, w * bpp), 40 + bpp)
    d = dev.upload_planes([src], 4, 4)[0]
    for x in list(range(0, 18)) + [w - 70]:
        for (cw, ch, y) in [(70, 20, 3), (w - x, h, 0), (21, 5, h - 5), (64 // bpp + 1, 9, 1)]:
            cw = min(cw, w - x)
            for stride in ((cw * bpp + 15) // 16 * 16 + 16, cw * bpp + 3):
                o = DevPlane(dev, ch, cw * bpp, stride)
                assert dev.lib.gmat_crop(d.ptr, d.stride, o.ptr, o.stride, x, y, cw, ch, bpp, None) == 0
                assert (o.download() == src[y:y + ch, x * bpp:(x + cw) * bpp]).all(), (bpp, x, cw, ch, stride)
                assert (o.download(True)[:, o.row_bytes:] == 0xCD).all(), (bpp, x, cw, ch, stride)
                o.free()
    d.free()


# ---- the AVFilter-shaped layer ------------------------------------------------------------------------
def _run_filter(dev, name, opts, src, w, h, fmt="rgb24"):
    lib = dev.lib
    fc = lib.gmat_hwframe_ctx_create(0, PIX_FMT[fmt], w, h, 1)
    assert fc
    f = lib.gmat_filter_alloc(name.encode())
    assert f
    for k, v in opts.items():
        assert lib.gmat_filter_set_option(f, k.encode(), str(v).encode()) == 0
    assert lib.gmat_filter_init(f) == 0
    assert lib.gmat_filter_config_props(f, fc, None) == 0
    # hwupload: host frame -> pooled device frame
    host = GmatFrame()
    assert lib.gmat_host_frame_alloc(C.byref(host), PIX_FMT[fmt], w, h) == 0
    rb = src.shape[1]
    hv = np.ctypeslib.as_array(C.cast(host.data[0], C.POINTER(C.c_uint8)), (h, host.linesize[0]))
    hv[:, :rb] = src
    fin = lib.gmat_frame_alloc()
    assert lib.gmat_hwframe_get_buffer(fc, fin) == 0
    assert lib.gmat_hwframe_transfer_data(fin, C.byref(host), None) == 0
    fin.contents.pts = 1234
    out = C.POINTER(GmatFrame)()
    r = lib.gmat_filter_frame(f, fin, C.byref(out))
    assert r == 0 and out
    o = out.contents
    assert o.pts == 1234 and o.format == PIX_FMT["hip"]
    # hwdownload
    hout = GmatFrame()
    assert lib.gmat_host_frame_alloc(C.byref(hout), o.sw_format, o.width, o.height) == 0
    assert lib.gmat_hwframe_transfer_data(C.byref(hout), out, None) == 0
    lib.gmat_device_sync()
    bpp = 4 if o.sw_format in (PIX_FMT["rgba"], PIX_FMT["bgra"]) else 3
    res = np.ctypeslib.as_array(C.cast(hout.data[0], C.POINTER(C.c_uint8)), (o.height, hout.linesize[0]))[:, :o.width * bpp].copy()
    ow, oh = o.width, o.height
    lib.gmat_frame_free(C.byref(out))
    lib.gmat_host_frame_free(C.byref(host)); lib.gmat_host_frame_free(C.byref(hout))
    lib.gmat_filter_free(f)
    lib.gmat_hwframe_ctx_free(fc)
    return res, ow, oh


def test_filter_layer_crop_defaults_to_centre(dev, orc):
    w, h = 100, 60
    src = orc.lcg((h, w * 3), 2)
    res, ow, oh = _run_filter(dev, "crop_hip", {"w": 40, "h": 20}, src, w, h)     # x=y=-1 -> centred
    assert (ow, oh) == (40, 20)
    assert (res == src[20:40, 30 * 3:70 * 3]).all()


def test_filter_layer_rotate_swaps_dimensions(dev, orc):
    w, h = 70, 30
    src = orc.lcg((h, w * 3), 3)
    res, ow, oh = _run_filter(dev, "rotate_hip", {"angle": 90}, src, w, h)
    assert (ow, oh) == (h, w)
    want = np.zeros((w, h * 3), np.uint8)
    orc.L.orc_transpose(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, 3, 1)
    assert (res == want).all()
    res, ow, oh = _run_filter(dev, "rotate_hip", {"angle": -90}, src, w, h)
    orc.L.orc_transpose(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, 3, 2)
    assert (res == want).all()
    res, ow, oh = _run_filter(dev, "rotate_hip", {"angle": 180}, src, w, h)
    assert (res.reshape(h, w, 3) == src.reshape(h, w, 3)[::-1, ::-1]).all()


def test_filter_layer_flip_smooth_scale_format(dev, orc):
    w, h = 96, 40
    src = orc.lcg((h, w * 3), 4)
    res, _, _ = _run_filter(dev, "flip_hip", {"code": 1}, src, w, h)
    assert (res.reshape(h, w, 3) == src.reshape(h, w, 3)[:, ::-1]).all()
    res, _, _ = _run_filter(dev, "smooth_hip", {"type": "gaussian", "kw": 3, "kh": 3}, src, w, h)
    want = np.zeros_like(src)
    m = (C.c_int * 9)(1, 2, 1, 2, 4, 2, 1, 2, 1)
    orc.L.orc_conv3x3(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, 3, m, 1 / 16, 0.0)
    assert (res == want).all()
    res, ow, oh = _run_filter(dev, "scale_hip", {"w": 48, "h": 20, "interp_algo": "bicubic"}, src, w, h)
    assert (ow, oh) == (48, 20)
    assert (res == orc.sws([src], w, h, "rgb24", 48, 20, "rgb24")[0]).all()
    res, _, _ = _run_filter(dev, "format_hip", {"pix_fmt": "bgr24"}, src, w, h)
    assert (res.reshape(h, w, 3) == src.reshape(h, w, 3)[:, :, ::-1]).all()


def test_filter_layer_errors(dev):
    lib = dev.lib
    assert not lib.gmat_filter_alloc(b"no_such_filter")
    f = lib.gmat_filter_alloc(b"crop_hip")
    assert lib.gmat_filter_set_option(f, b"bogus", b"1") < 0
    assert lib.gmat_filter_init(f) < 0                       # w/h == 0 (vf_crop_nvcv.c:116-119)
    lib.gmat_filter_free(f)
    f = lib.gmat_filter_alloc(b"crop_hip")
    lib.gmat_filter_set_option(f, b"w", b"500"); lib.gmat_filter_set_option(f, b"h", b"10")
    assert lib.gmat_filter_init(f) == 0
    fc = lib.gmat_hwframe_ctx_create(0, PIX_FMT["rgb24"], 100, 100, 0)
    assert lib.gmat_filter_config_props(f, fc, None) < 0     # crop area outside the frame
    lib.gmat_filter_free(f)
    f = lib.gmat_filter_alloc(b"rotate_hip")
    lib.gmat_filter_set_option(f, b"angle", b"33")
    assert lib.gmat_filter_init(f) == 0                      # arbitrary angles: vf_rotate.c fixed point
    lib.gmat_filter_set_option(f, b"interp", b"cubic")
    assert lib.gmat_filter_init(f) == 0                      # cubic: Catmull-Rom in integers (rule in oracle/orc_vf.c)
    lib.gmat_filter_set_option(f, b"interp", b"area")
    assert lib.gmat_filter_init(f) == 0                      # area = linear, as cv::warpAffine
    lib.gmat_filter_set_option(f, b"interp", b"lanczos")
    assert lib.gmat_filter_init(f) < 0                       # not one of vf_rotate_nvcv.c:114-135's four
    lib.gmat_filter_free(f)
    f = lib.gmat_filter_alloc(b"flip_hip")
    assert lib.gmat_filter_init(f) == 0
    nv = lib.gmat_hwframe_ctx_create(0, PIX_FMT["rgbpf32le"], 64, 64, 0)
    assert lib.gmat_filter_config_props(f, nv, None) < 0     # packed RGB and 8-bit 4:2:0 only
    lib.gmat_filter_free(f)
    lib.gmat_hwframe_ctx_free(nv); lib.gmat_hwframe_ctx_free(fc)


# ---- planar / semi-planar 4:2:0 frames through the filter layer (per-plane, like the CPU filters) -----
def _run_filter_planes(dev, name, opts, src_planes, w, h, fmt):
    from harness import plane_shapes
    lib = dev.lib
    fc = lib.gmat_hwframe_ctx_create(0, PIX_FMT[fmt], w, h, 1)
    f = lib.gmat_filter_alloc(name.encode())
    assert fc and f
    for k, v in opts.items():
        assert lib.gmat_filter_set_option(f, k.encode(), str(v).encode()) == 0
    assert lib.gmat_filter_init(f) == 0
    assert lib.gmat_filter_config_props(f, fc, None) == 0
    host = GmatFrame()
    assert lib.gmat_host_frame_alloc(C.byref(host), PIX_FMT[fmt], w, h) == 0
    for i, pl in enumerate(src_planes):
        hv = np.ctypeslib.as_array(C.cast(host.data[i], C.POINTER(C.c_uint8)), (pl.shape[0], host.linesize[i]))
        hv[:, :pl.shape[1]] = pl
    fin = lib.gmat_frame_alloc()
    assert lib.gmat_hwframe_get_buffer(fc, fin) == 0
    assert lib.gmat_hwframe_transfer_data(fin, C.byref(host), None) == 0
    out = C.POINTER(GmatFrame)()
    assert lib.gmat_filter_frame(f, fin, C.byref(out)) == 0 and out
    o = out.contents
    hout = GmatFrame()
    assert lib.gmat_host_frame_alloc(C.byref(hout), o.sw_format, o.width, o.height) == 0
    assert lib.gmat_hwframe_transfer_data(C.byref(hout), out, None) == 0
    lib.gmat_device_sync()
    res = []
    ofmt = {v: k for k, v in PIX_FMT.items()}[o.sw_format]
    for i, (rows, rb) in enumerate(plane_shapes(ofmt, o.width, o.height)):
        res.append(np.ctypeslib.as_array(C.cast(hout.data[i], C.POINTER(C.c_uint8)), (rows, hout.linesize[i]))[:, :rb].copy())
    ow, oh = o.width, o.height
    lib.gmat_frame_free(C.byref(out))
    lib.gmat_host_frame_free(C.byref(host)); lib.gmat_host_frame_free(C.byref(hout))
    lib.gmat_filter_free(f)
    lib.gmat_hwframe_ctx_free(fc)
    return res, ow, oh


def _plane_bpp(fmt, i):
    return 2 if fmt == "nv12" and i == 1 else 1


@pytest.mark.parametrize("fmt", ["yuv420p", "nv12"])
@pytest.mark.parametrize("w,h", [(96, 40), (70, 34)])
def test_filter_layer_planar_frames(dev, orc, fmt, w, h):
    src = synth_planes(orc, fmt, w, h, 77)

    def per_plane(fn):
        outs = []
        for i, pl in enumerate(src):
            bpp = _plane_bpp(fmt, i)
            outs.append(fn(np.ascontiguousarray(pl), pl.shape[1] // bpp, pl.shape[0], bpp))
        return outs

    def o_transpose(direction):
        def fn(pl, pw, ph, bpp):
            want = np.zeros((pw, ph * bpp), np.uint8)
            orc.L.orc_transpose(pl.ctypes.data, pl.strides[0], want.ctypes.data, want.strides[0], pw, ph, bpp, direction)
            return want
        return fn

    def o_hflip(pl, pw, ph, bpp):
        want = np.zeros_like(pl)
        orc.L.orc_hflip(pl.ctypes.data, pl.strides[0], want.ctypes.data, want.strides[0], pw, ph, bpp)
        return want

    def o_smooth(pl, pw, ph, bpp):
        want = np.zeros_like(pl)
        m = (C.c_int * 9)(1, 2, 1, 2, 4, 2, 1, 2, 1)
        orc.L.orc_conv3x3(pl.ctypes.data, pl.strides[0], want.ctypes.data, want.strides[0], pw, ph, bpp, m, 1 / 16, 0.0)
        return want

    res, ow, oh = _run_filter_planes(dev, "transpose_hip", {"dir": "clock"}, src, w, h, fmt)
    assert (ow, oh) == (h, w)
    for a, b in zip(res, per_plane(o_transpose(1))):
        assert (a == b).all()
    res, ow, oh = _run_filter_planes(dev, "rotate_hip", {"angle": -90}, src, w, h, fmt)
    for a, b in zip(res, per_plane(o_transpose(2))):
        assert (a == b).all()
    res, _, _ = _run_filter_planes(dev, "flip_hip", {"code": 1}, src, w, h, fmt)
    for a, b in zip(res, per_plane(o_hflip)):
        assert (a == b).all()
    res, _, _ = _run_filter_planes(dev, "smooth_hip", {"type": "gaussian"}, src, w, h, fmt)
    for a, b in zip(res, per_plane(o_smooth)):
        assert (a == b).all()
    # crop: x, y, w, h are aligned down to the chroma grid like vf_crop.c:186-187,:223-224
    res, ow, oh = _run_filter_planes(dev, "crop_hip", {"w": 41, "h": 21, "x": 11, "y": 7}, src, w, h, fmt)
    assert (ow, oh) == (40, 20)
    assert (res[0] == src[0][6:26, 10:50]).all()
    if fmt == "nv12":
        assert (res[1] == src[1][3:13, 10:50]).all()
    else:
        assert (res[1] == src[1][3:13, 5:25]).all() and (res[2] == src[2][3:13, 5:25]).all()


def test_filter_layer_scale_keeps_yuv_format(dev, orc):
    """scale_hip without a format option keeps the input format (vf_scale_cuda.c:594 "same"): nv12 in, nv12 out."""
    from harness import SWS
    w, h = 128, 48
    src = synth_planes(orc, "nv12", w, h, 91)
    res, ow, oh = _run_filter_planes(dev, "scale_hip", {"w": 64, "h": 24, "interp_algo": "bicubic"}, src, w, h, "nv12")
    assert (ow, oh) == (64, 24)
    want = orc.sws(src, w, h, "nv12", 64, 24, "nv12", SWS["bicubic"])
    for a, b in zip(res, want):
        assert (a == b).all()
    # explicit format option converts on the way
    res, ow, oh = _run_filter_planes(dev, "scale_hip", {"w": 64, "h": 24, "format": "yuv420p"}, src, w, h, "nv12")
    want = orc.sws(src, w, h, "nv12", 64, 24, "yuv420p", SWS["bicubic"])
    assert len(res) == 3
    for a, b in zip(res, want):
        assert (a == b).all()


# ---- arbitrary-angle rotation (vf_rotate.c fixed point) ---------------------------------------------------
import math


def test_rotate_fixed_point_sincos_known_values(orc):
    s, c = C.c_int(), C.c_int()
    for deg, es, ec in [(0, 0, 65536), (90, 65536, 0), (180, 0, -65536), (30, 32768, 56756), (-45, -46341, 46341)]:
        orc.L.orc_rotate_sincos(math.radians(deg), C.byref(s), C.byref(c))
        assert abs(s.value - es) <= 4 and abs(c.value - ec) <= 4, (deg, s.value, c.value)   # 5-term Taylor series


@pytest.mark.parametrize("bpp", [1, 2, 3, 4])
@pytest.mark.parametrize("deg", [17.0, -33.5, 45.0, 100.0, 181.0, 359.0])
@pytest.mark.parametrize("bilinear", [1, 0])
def test_rotate_arbitrary_angle(dev, orc, bpp, deg, bilinear):
    w, h = (131, 77) if bpp != 4 else (96, 40)
    src = orc.lcg((h, w * bpp), 9)
    fill = np.array([3, 250, 128, 255], np.uint8)
    want = np.full((h, w * bpp), 0x11, np.uint8)
    orc.L.orc_rotate(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, w, h, bpp,
                     math.radians(deg), bilinear, fill.ctypes.data)
    for align, extra in [(256, 0), (1, 1)]:
        d = dev.upload_planes([src], align, extra)[0]
        o = DevPlane(dev, h, w * bpp, (w * bpp + extra + align - 1) // align * align)
        assert dev.lib.gmat_rotate(d.ptr, d.stride, o.ptr, o.stride, w, h, w, h, bpp, math.radians(deg), bilinear,
                                   fill.ctypes.data, None) == 0
        got = o.download()
        bad = np.argwhere(got != want)
        assert bad.size == 0, f"{len(bad)} mismatching bytes, first at {bad[:4].tolist()}"
        assert (o.download(True)[:, o.row_bytes:] == 0xCD).all()
        d.free(); o.free()


def test_rotate_without_fill_and_other_output_size(dev, orc):
    w, h, ow, oh, bpp = 90, 50, 120, 70, 3
    src = orc.lcg((h, w * bpp), 10)
    want = np.full((oh, ow * bpp), 0xCD, np.uint8)            # DevPlane's initial fill: untouched pixels keep it
    orc.L.orc_rotate(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, ow, oh, bpp,
                     math.radians(25.0), 1, None)
    d = dev.upload_planes([src], 64)[0]
    o = DevPlane(dev, oh, ow * bpp, (ow * bpp + 63) // 64 * 64)
    assert dev.lib.gmat_rotate(d.ptr, d.stride, o.ptr, o.stride, w, h, ow, oh, bpp, math.radians(25.0), 1, None, None) == 0
    assert (o.download() == want).all()
    d.free(); o.free()


@pytest.mark.parametrize("fmt", ["rgb24", "rgba", "yuv420p", "nv12"])
def test_filter_layer_rotate_arbitrary(dev, orc, fmt):
    w, h = 96, 40
    src = synth_planes(orc, fmt, w, h, 93)
    res, ow, oh = _run_filter_planes(dev, "rotate_hip", {"angle": 30, "interp": "linear"}, src, w, h, fmt)
    assert (ow, oh) == (w, h)
    for i, pl in enumerate(src):
        bpp = {"rgb24": 3, "rgba": 4}.get(fmt, _plane_bpp(fmt, i))
        pw, ph = pl.shape[1] // bpp, pl.shape[0]
        if fmt in ("rgb24", "rgba"):
            fill = np.array([0, 0, 0, 255], np.uint8)
        else:
            fill = np.array([16 if i == 0 else 128, 128, 0, 0], np.uint8)
        want = np.zeros_like(pl)
        pl = np.ascontiguousarray(pl)
        orc.L.orc_rotate(pl.ctypes.data, pl.strides[0], want.ctypes.data, want.strides[0], pw, ph, pw, ph, bpp,
                         30 * math.pi / 180.0, 1, fill.ctypes.data)
        assert (res[i] == want).all(), (fmt, i)


def test_filter_layer_format_nv12_rgbpf32_round_trip(dev, orc):
    """format_hip (vf_format_cuda.c:69-79): nv12 -> rgbpf32le and back to nv12 through the frame pool."""
    w, h = 64, 16
    src = synth_planes(orc, "nv12", w, h, 95)
    pf, ow, oh = _run_filter_planes(dev, "format_hip", {"pix_fmt": "rgbpf32le"}, src, w, h, "nv12")
    want = orc.nv12_to_rgbpf32(src, w, h)
    assert len(pf) == 3
    for k in range(3):
        assert (pf[k].view(np.float32) == want[k]).all()
    back, _, _ = _run_filter_planes(dev, "format_hip", {"pix_fmt": "nv12"}, pf, w, h, "rgbpf32le")
    rgb = orc.yuv2rgb(src, w, h, "nv12", "rgb24")
    want_nv12 = orc.sws([rgb], w, h, "rgb24", w, h, "nv12")
    for a, b in zip(back, want_nv12):
        assert (a == b).all()


def test_filter_layer_scale_from_yuv444p(dev, orc):
    from harness import SWS
    w, h = 128, 48
    src = synth_planes(orc, "yuv444p", w, h, 97)
    res, ow, oh = _run_filter_planes(dev, "scale_hip", {"w": 64, "h": 24, "format": "nv12"}, src, w, h, "yuv444p")
    assert (ow, oh) == (64, 24)
    want = orc.sws(src, w, h, "yuv444p", 64, 24, "nv12", SWS["bicubic"])
    for a, b in zip(res, want):
        assert (a == b).all()


@pytest.mark.parametrize("src_fmt", ["p010le", "p016le"])
def test_filter_layer_scale_from_p01x(dev, orc, src_fmt):
    """scale_hip on 16-bit semi-planar frames (scale_cuda's input list, vf_scale_cuda.c:45-54) to an 8-bit format,
    and format_hip to the same; 8-bit frames go the other way through format_hip (planar8ToP01xleWrapper)"""
    from harness import SWS
    w, h = 128, 48
    src = synth_planes(orc, src_fmt, w, h, 99)
    res, ow, oh = _run_filter_planes(dev, "scale_hip", {"w": 64, "h": 24, "format": "nv12"}, src, w, h, src_fmt)
    assert (ow, oh) == (64, 24)
    for a, b in zip(res, orc.sws(src, w, h, src_fmt, 64, 24, "nv12", SWS["bicubic"])):
        assert (a == b).all()
    res, _, _ = _run_filter_planes(dev, "format_hip", {"pix_fmt": "yuv420p"}, src, w, h, src_fmt)
    for a, b in zip(res, orc.sws(src, w, h, src_fmt, w, h, "yuv420p", SWS["bicubic"])):
        assert (a == b).all()
    # scale_cuda's own case: the 16-bit format in, the same format out
    res, ow, oh = _run_filter_planes(dev, "scale_hip", {"w": 64, "h": 24}, src, w, h, src_fmt)
    for a, b in zip(res, orc.sws(src, w, h, src_fmt, 64, 24, src_fmt, SWS["bicubic"])):
        assert (a == b).all()
    res, _, _ = _run_filter_planes(dev, "scale_hip", {"w": 64, "h": 24, "format": "rgba64le"}, src, w, h, src_fmt)
    assert (res[0] == orc.sws(src, w, h, src_fmt, 64, 24, "rgba64le", SWS["bicubic"])[0]).all()
    # planar 8-bit frames: libswscale's special converter t | t << 8; a semi-planar (nv12) source has none and runs the generic lines, t << 8
    pl = synth_planes(orc, "yuv420p", w, h, 100)
    up, _, _ = _run_filter_planes(dev, "format_hip", {"pix_fmt": src_fmt}, pl, w, h, "yuv420p")
    assert (up[0].view(np.uint16) == pl[0].astype(np.uint16) * 257).all()
    assert (up[1].view(np.uint16)[:, 0::2] == pl[1].astype(np.uint16) * 257).all()
    nv = synth_planes(orc, "nv12", w, h, 100)
    up, _, _ = _run_filter_planes(dev, "format_hip", {"pix_fmt": src_fmt}, nv, w, h, "nv12")
    for a, b in zip(up, orc.sws(nv, w, h, "nv12", w, h, src_fmt, SWS["bicubic"])):
        assert (a == b).all()
    assert (up[0].view(np.uint16) == nv[0].astype(np.uint16) << 8).all()


@pytest.mark.parametrize("bpp", [1, 2, 3, 4])
@pytest.mark.parametrize("w,h", [(64, 16), (130, 35), (5, 3), (1, 1), (2, 7), (259, 4), (200, 70), (68, 129), (520, 9), (4, 65), (248, 3)])
@pytest.mark.parametrize("kernel", ["strip", "bytes"])
def test_median3x3(dev, orc, monkeypatch, w, h, bpp, kernel):
    """smooth type=median at 3x3: per channel the 5th smallest of the window, edges clamped (vf_median.c at radius 1).  Both kernels:
    median3x3s_kernel (rows of whole dwords on dword-aligned planes: the first alignment below) and the byte-wise median3x3_kernel
    (everything else, and everything under GMAT_NO_MEDIAN_STRIP); the sizes cross the strip kernel's 62-dword / 64-row tiles, end
    on a single row of a tile, and have rows shorter than a tile"""
    if kernel == "bytes":
        monkeypatch.setenv("GMAT_NO_MEDIAN_STRIP", "1")
    else:
        monkeypatch.delenv("GMAT_NO_MEDIAN_STRIP", raising=False)
    dev.lib.gmat_knobs_reload()
    src = orc.lcg((h, w * bpp), 111 + bpp)
    want = np.zeros_like(src)
    orc.L.orc_median3x3(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, bpp)
    # independent statement: numpy median over the edge-padded 3x3 neighbourhood
    px = np.pad(src.reshape(h, w, bpp), ((1, 1), (1, 1), (0, 0)), mode="edge")
    stack = np.stack([px[j:j + h, i:i + w] for j in range(3) for i in range(3)], axis=0)
    assert (want.reshape(h, w, bpp) == np.sort(stack, axis=0)[4]).all()
    for align, extra in ((64, 0), (1, 1)):
        d = dev.upload_planes([src], align, extra)[0]
        o = dev.planes_like("rgb24", 1, 1)[0] if False else None
        from harness import DevPlane
        out = DevPlane(dev, h, w * bpp, ((w * bpp + extra + align - 1) // align) * align)
        assert dev.lib.gmat_median3x3(d.ptr, d.stride, out.ptr, out.stride, w, h, bpp, None) == 0
        assert (out.download() == want).all()
        assert (out.download(with_padding=True)[:, out.row_bytes:] == 0xCD).all()
        d.free(); out.free()


@pytest.mark.parametrize("rows", [1, 2, 3, 7, 10, 64])
@pytest.mark.parametrize("bpp", [1, 2, 3, 4])
def test_median3x3_segmentation_does_not_change_the_result(dev, orc, monkeypatch, bpp, rows):
    """median3x3s_kernel walks segments of `rows` output rows two at a time with loads four iterations ahead: odd lengths end on
    half a row pair, short ones finish inside the first unrolled group"""
    from harness import DevPlane
    monkeypatch.setenv("GMAT_STRIP_ROWS", str(rows))
    w, h = 252, 37
    src = orc.lcg((h, w * bpp), 131 + bpp)
    want = np.zeros_like(src)
    orc.L.orc_median3x3(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, bpp)
    d = dev.upload_planes([src], 64, 0)[0]
    out = DevPlane(dev, h, w * bpp, ((w * bpp + 63) // 64) * 64)
    assert dev.lib.gmat_median3x3(d.ptr, d.stride, out.ptr, out.stride, w, h, bpp, None) == 0
    assert (out.download() == want).all()
    assert (out.download(with_padding=True)[:, out.row_bytes:] == 0xCD).all()
    d.free(); out.free()


def test_filter_layer_smooth_median(dev, orc):
    w, h = 96, 40
    for fmt in ("rgb24", "nv12"):
        src = synth_planes(orc, fmt, w, h, 113)
        res, _, _ = _run_filter_planes(dev, "smooth_hip", {"type": "median"}, src, w, h, fmt)
        for i, pl in enumerate(src):
            bpp = 3 if fmt == "rgb24" else _plane_bpp(fmt, i)
            pl = np.ascontiguousarray(pl)
            want = np.zeros_like(pl)
            orc.L.orc_median3x3(pl.ctypes.data, pl.strides[0], want.ctypes.data, want.strides[0], pl.shape[1] // bpp, pl.shape[0], bpp)
            assert (res[i] == want).all(), (fmt, i)


def test_filter_layer_0rgb32_frames(dev, orc):
    """the nvcv filters list 0rgb32 / 0bgr32 (vf_crop_nvcv.c:90-98): 4-byte pixels, the padding byte moves with them"""
    w, h = 64, 24
    src = synth_planes(orc, "rgba", w, h, 115)
    res, _, _ = _run_filter_planes(dev, "flip_hip", {"code": 1}, src, w, h, "bgr0")
    assert (res[0].reshape(h, w, 4) == src[0].reshape(h, w, 4)[:, ::-1]).all()
    res, ow, oh = _run_filter_planes(dev, "scale_hip", {"w": 32, "h": 12}, src, w, h, "bgr0")
    want = orc.sws([np.ascontiguousarray(src[0].reshape(h, w, 4)[:, :, :3].reshape(h, 3 * w))], w, h, "bgr24", 32, 12, "bgra")
    assert (res[0] == want[0]).all()


def test_filter_layer_yuv444p_frames(dev, orc):
    """scale_hip / format_hip with a planar 4:4:4 destination, then crop + flip + transpose on those frames"""
    from harness import SWS
    w, h = 128, 48
    src = synth_planes(orc, "nv12", w, h, 98)
    res, ow, oh = _run_filter_planes(dev, "scale_hip", {"w": 96, "h": 40, "format": "yuv444p"}, src, w, h, "nv12")
    assert (ow, oh) == (96, 40) and len(res) == 3
    want = orc.sws(src, w, h, "nv12", 96, 40, "yuv444p", SWS["bicubic"])
    for a, b in zip(res, want):
        assert (a == b).all()
    fmt, _, _ = _run_filter_planes(dev, "format_hip", {"pix_fmt": "yuv444p"}, src, w, h, "nv12")
    for a, b in zip(fmt, orc.sws(src, w, h, "nv12", w, h, "yuv444p", SWS["bicubic"])):
        assert (a == b).all()
    t, tw, th = _run_filter_planes(dev, "transpose_hip", {"dir": "clock"}, want, 96, 40, "yuv444p")
    assert (tw, th) == (40, 96)
    for a, b in zip(t, want):
        assert (a == np.rot90(b, -1)).all()
    f, _, _ = _run_filter_planes(dev, "flip_hip", {"code": 1}, want, 96, 40, "yuv444p")
    for a, b in zip(f, want):
        assert (a == b[:, ::-1]).all()
    c, cw, ch = _run_filter_planes(dev, "crop_hip", {"w": 31, "h": 17, "x": 5, "y": 3}, want, 96, 40, "yuv444p")
    assert (cw, ch) == (31, 17)
    for a, b in zip(c, want):
        assert (a == b[3:20, 5:36]).all()
    r, _, _ = _run_filter_planes(dev, "rotate_hip", {"angle": 23}, want, 96, 40, "yuv444p")
    for i, (a, b) in enumerate(zip(r, want)):
        fill = np.array([16 if i == 0 else 128, 128, 0, 0], np.uint8)
        b = np.ascontiguousarray(b)
        ref = np.zeros_like(b)
        orc.L.orc_rotate(b.ctypes.data, b.strides[0], ref.ctypes.data, ref.strides[0], 96, 40, 96, 40, 1,
                         23 * math.pi / 180.0, 1, fill.ctypes.data)
        assert (a == ref).all(), i


@pytest.mark.parametrize("case", [(277, 128, 4, [5, -1, -1, 7, 0, -3, -3, 0, 3]), (241, 116, 1, [0, 6, -1, 3, 0, 6, -1, 0, -3]),
                                  (264, 196, 3, [1, 2, -2, 2, 9, 5, -3, 5, -2]), (130, 40, 3, [1, 2, 3, 4, 5, 6, 7, 8, 9])])
def test_conv3x3_float_epilogue_is_not_contracted(dev, orc, case):
    """sum * rdiv + bias + 0.5f must round after the multiply and after each add like vf_convolution.c:495-512 on the
    CPU; hipcc's default fp contraction fused them into fma and came out one ulp low on hardware (fuzzer finding)."""
    w, h, bpp, mat = case
    src = orc.lcg((h, w * bpp), 500 + w)
    m = (C.c_int * 9)(*mat)
    want = _orc_out(h, w * bpp)
    orc.L.orc_conv3x3(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, bpp, m, 0.1, 3.5)
    d = dev.upload_planes([src], 16)[0]
    o = DevPlane(dev, h, w * bpp, (w * bpp + 15) // 16 * 16)
    assert dev.lib.gmat_smooth3x3(d.ptr, d.stride, o.ptr, o.stride, w, h, bpp, m, 0.1, 3.5, None) == 0
    assert (o.download() == want).all()
    d.free(); o.free()


# ---- rotate_nvcv's remaining options and smooth_nvcv's median beyond 3 x 3 (missing #2) -----------------------------
@pytest.mark.parametrize("bpp", [1, 3, 4])
@pytest.mark.parametrize("case", [(17.0, 2, 0.0, 0.0), (17.0, 1, 5.0, -3.0), (-133.5, 2, 2.25, 7.5), (90.0, 0, 3.0, 1.0), (0.0, 2, 0.5, 0.5),
                                  (33.0, 1, -40.0, 12.75), (271.0, 2, 0.0, 9.0)])
def test_rotate_interp_and_shift_match_the_stated_rule(dev, orc, case, bpp):
    """interp = cubic | linear | nearest with shift_x / shift_y: the rule of oracle/orc_vf.c (vf_rotate.c's 16.16 walk, translated;
    Catmull-Rom in 14-bit integer weights), bit for bit, outside pixels filled"""
    deg, interp, sx, sy = case
    w, h = 70, 45
    src = orc.lcg((h, w * bpp), 23 + bpp)
    d = dev.upload_planes([src], 4)[0]
    o = DevPlane(dev, h, w * bpp, (w * bpp + 15) // 16 * 16)
    fill = (C.c_uint8 * 4)(9, 8, 7, 255)
    assert dev.lib.gmat_rotate2(d.ptr, d.stride, o.ptr, o.stride, w, h, w, h, bpp, math.radians(deg), interp, sx, sy, fill, None) == 0
    want = np.zeros_like(src)
    orc.L.orc_rotate2(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, w, h, bpp, math.radians(deg), interp, sx, sy, fill)
    got = o.download()
    assert (got == want).all(), np.argwhere(got != want)[:4]
    assert (o.download(with_padding=True)[:, w * bpp:] == 0xCD).all()
    d.free(); o.free()


@pytest.mark.parametrize("lds", ["0", "1", "1w2", "1m0"])
@pytest.mark.parametrize("interp", [0, 1, 2])
def test_rotate_both_kernels_over_tile_edges(dev, orc, monkeypatch, lds, interp):
    """rotate_lds_kernel (source patch in LDS: shipped whenever the source rows are dword-aligned; four waves a tile, and the two-wave
    A/B form) and rotate_kernel (direct gathers: the fallback), each forced on every interpolation, on frames several 32 x 32 tiles large whose rotated image leaves the source on all sides — the walk clamps a
    tap index BEFORE it forms the neighbour's (x1 = -1 reads pixels 0 and 1, vf_rotate.c:463-492), which the patch's bounding box has to
    follow (a 3-in-1000 fuzz find of round 3) — with sizes one off the tile grid and a source pitch that is not the row length"""
    monkeypatch.setenv("GMAT_ROTATE_LDS", lds[0])
    if lds == "1w2":
        monkeypatch.setenv("GMAT_ROTATE_WAVES", "2")
    if lds == "1m0":                                          # the 32 x 32 form behind the macro tiles (nearest / bilinear; cubic always runs it)
        monkeypatch.setenv("GMAT_ROTATE_MT", "0")
    dev.lib.gmat_knobs_reload()
    fill = (C.c_uint8 * 4)(1, 2, 3, 4)
    for (w, h, bpp, deg, sx, sy) in [(113, 179, 4, 143.7, 0.0, 0.0), (283, 167, 2, 17.0, 0.0, 0.0), (258, 175, 1, -61.3, 0.0, 0.0),
                                     (230, 130, 3, 100.9, 0.0, 0.0), (195, 69, 3, 271.25, 0.0, 0.0), (97, 65, 3, 45.0, 40.5, -20.25),
                                     (64, 64, 4, 7.5, -3.0, 2.0), (33, 31, 1, 333.0, 0.0, 0.0)]:
        src = orc.lcg((h, w * bpp), 60 + bpp)
        d = dev.upload_planes([src], 4, 4)[0]
        o = DevPlane(dev, h, w * bpp, (w * bpp + 19) // 4 * 4)
        assert dev.lib.gmat_rotate2(d.ptr, d.stride, o.ptr, o.stride, w, h, w, h, bpp, math.radians(deg), interp, sx, sy, fill, None) == 0
        want = np.zeros_like(src)
        orc.L.orc_rotate2(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, w, h, bpp, math.radians(deg), interp, sx, sy, fill)
        got = o.download()
        assert (got == want).all(), ((w, h, bpp, deg), np.argwhere(got != want)[:4].tolist())
        assert (o.download(with_padding=True)[:, w * bpp:] == 0xCD).all()
        d.free(); o.free()


@pytest.mark.parametrize("lds", ["0", "1"])
@pytest.mark.parametrize("interp", [0, 1, 2])
def test_rotate_frames_one_column_or_row_thick(dev, orc, monkeypatch, lds, interp):
    """a 1 x h or w x 1 frame: its first column / row is also its last one, so x1 = -1 — which keeps its fraction and reads pixels 0 and
    min(1, W - 1) — pairs pixel 0 with itself (a GPU fuzz find of round 3: the LDS form's zero-weight rule for the last column only looked
    at the upper clamp; fuzz_transforms seed 913 case 373: 1 x 24, 2 bytes per pixel)"""
    monkeypatch.setenv("GMAT_ROTATE_LDS", lds)
    dev.lib.gmat_knobs_reload()
    fill = (C.c_uint8 * 4)(200, 100, 50, 25)
    for (w, h, bpp, deg) in [(1, 24, 2, 77.3), (1, 24, 2, -160.0), (40, 1, 3, 12.5), (1, 1, 4, 33.0), (1, 70, 1, 5.0), (2, 33, 3, 91.5), (57, 2, 4, -3.0)]:
        src = orc.lcg((h, w * bpp), 90 + bpp)
        d = dev.upload_planes([src], 4, 4)[0]
        o = DevPlane(dev, h, w * bpp, (w * bpp + 11) // 4 * 4)
        assert dev.lib.gmat_rotate2(d.ptr, d.stride, o.ptr, o.stride, w, h, w, h, bpp, math.radians(deg), interp, 0.0, 0.0, fill, None) == 0
        want = np.zeros_like(src)
        orc.L.orc_rotate2(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, w, h, bpp, math.radians(deg), interp, 0.0, 0.0, fill)
        got = o.download()
        assert (got == want).all(), ((w, h, bpp, deg), np.argwhere(got != want)[:4].tolist())
        d.free(); o.free()


def _shift_translation(angle_deg, sx, sy, w, h):
    """rotate_nvcv's shift as the reference means it — rotation about the ORIGIN, src = M (dst - shift), the shift re-centres
    (vf_rotate_nvcv.c:85-86,276) — as gmat_rotate2's translation of the centre-rotated image: t = shift - C + M^T C with the walk's
    M = [c s; -s c] (vf_rotate.c:538-548) and C = ((w - 1) / 2, (h - 1) / 2).  Restated here independently of the library's helper."""
    a = math.radians(angle_deg)
    c, s = math.cos(a), math.sin(a)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    return sx - cx + (c * cx - s * cy), sy - cy + (s * cx + c * cy)


def test_rotate_filter_honours_interp_and_shift(dev, orc):
    """through the filter: rotate_hip angle=17:interp=cubic:shift_x=..:shift_y=.. on rgb24, and on nv12 (the chroma planes take half
    the shift on their own grid).  A given shift means what the reference's option means: rotation about the origin."""
    w, h = 96, 40
    src = orc.lcg((h, w * 3), 77)
    fill = (C.c_uint8 * 4)(0, 0, 0, 255)
    # the shift that re-centres a 17 degree turn, moved by (6, -2.5): the centre-rotated picture translated by (6, -2.5)
    rcx, rcy = _shift_translation(17, 0, 0, w, h)
    sx, sy = 6 - rcx, -2.5 - rcy
    res, ow, oh = _run_filter(dev, "rotate_hip", {"angle": 17, "interp": "cubic", "shift_x": repr(sx), "shift_y": repr(sy)}, src, w, h)
    tx, ty = _shift_translation(17, sx, sy, w, h)
    assert abs(tx - 6) < 1e-9 and abs(ty + 2.5) < 1e-9
    want = np.zeros_like(src)
    orc.L.orc_rotate2(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, w, h, 3, math.radians(17), 2, tx, ty, fill)
    assert (ow, oh) == (w, h) and (res == want).all()
    # an arbitrary shift: the reference's rule, rotation about the origin
    res, _, _ = _run_filter(dev, "rotate_hip", {"angle": 17, "interp": "linear", "shift_x": 11, "shift_y": -4}, src, w, h)
    tx, ty = _shift_translation(17, 11, -4, w, h)
    want = np.zeros_like(src)
    orc.L.orc_rotate2(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, w, h, 3, math.radians(17), 1, tx, ty, fill)
    assert (res == want).all()
    # ... known answer of that rule: dst = shift is the source's ORIGIN, so the output pixel at (11, 36 - 4 ... ) — use a visible one:
    res, _, _ = _run_filter(dev, "rotate_hip", {"angle": 17, "interp": "nearest", "shift_x": 20, "shift_y": 9}, src, w, h)
    assert (res[9, 3 * 20:3 * 21] == src[0, 0:3]).all()
    # the re-centring shift itself gives EXACTLY the shift-free picture (what the option is documented for)
    res0, _, _ = _run_filter(dev, "rotate_hip", {"angle": 33, "interp": "linear"}, src, w, h)
    rx, ry = _shift_translation(33, 0, 0, w, h)
    res1, _, _ = _run_filter(dev, "rotate_hip", {"angle": 33, "interp": "linear", "shift_x": repr(-rx), "shift_y": repr(-ry)}, src, w, h)
    assert (res0 == res1).all()
    # a shift with no rotation is a translation: pixels move right / down by whole amounts, the uncovered part is the background
    res, _, _ = _run_filter(dev, "rotate_hip", {"angle": 0, "interp": "nearest", "shift_x": 5, "shift_y": 3}, src, w, h)
    # (vf_rotate.c's validity window reaches one sample past the frame, clamped: the row / column just before the image repeats its edge)
    assert (res[3:, 15:] == src[:-3, :-15]).all() and (res[:2] == 0).all() and (res[:, :12] == 0).all()
    src = synth_planes(orc, "nv12", w, h, seed=78)
    res, _, _ = _run_filter_planes(dev, "rotate_hip", {"angle": 33, "interp": "linear", "shift_x": 8, "shift_y": 4}, src, w, h, "nv12")
    for i, (pl, bpp, sub) in enumerate(((src[0], 1, 0), (src[1], 2, 1))):
        pw, ph = pl.shape[1] // bpp, pl.shape[0]
        want = np.zeros_like(pl)
        fill = (C.c_uint8 * 4)(16 if i == 0 else 128, 128, 0, 0)
        tx, ty = _shift_translation(33, 8.0 / (1 << sub), 4.0 / (1 << sub), pw, ph)
        orc.L.orc_rotate2(pl.ctypes.data, pl.strides[0], want.ctypes.data, want.strides[0], pw, ph, pw, ph, bpp, math.radians(33), 1, tx, ty, fill)
        assert (res[i] == want).all()


def test_rotate_filter_quarter_turn_with_a_shift_keeps_the_input_size(dev, orc):
    """ONE predicate decides both the output size and the kernel.  A quarter turn swaps width and height
    only without a shift; with one the arbitrary-angle walk runs at the input's size — on a portrait AND a landscape frame."""
    for w, h in ((40, 96), (96, 40)):
        src = orc.lcg((h, w * 3), 79)
        res, ow, oh = _run_filter(dev, "rotate_hip", {"angle": 90}, src, w, h)
        assert (ow, oh) == (h, w)
        res, ow, oh = _run_filter(dev, "rotate_hip", {"angle": 90, "interp": "nearest", "shift_x": 1, "shift_y": 0}, src, w, h)
        assert (ow, oh) == (w, h) and res.shape == src.shape
        tx, ty = _shift_translation(90, 1, 0, w, h)
        want = np.zeros_like(src)
        fill = (C.c_uint8 * 4)(0, 0, 0, 255)
        orc.L.orc_rotate2(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, w, h, 3, math.radians(90), 0, tx, ty, fill)
        assert (res == want).all()


def test_rotate_shift_out_of_every_range(dev, orc):
    """a translation that pushes the whole source out of the output is an all-background frame, however large
    (the 32-bit walk must not wrap); a non-finite one is refused"""
    w, h = 64, 48
    src = orc.lcg((h, w * 3), 80)
    d = dev.upload_planes([src], 64)[0]
    o = dev.planes_like("rgb24", w, h, 64)[0]
    fill = (C.c_uint8 * 4)(7, 8, 9, 0)
    for sx, sy in ((30000.0, 30000.0), (-1e9, 5.0), (1e300, -1e300), (40000.0, 0.0)):
        assert dev.lib.gmat_rotate2(d.ptr, d.stride, o.ptr, o.stride, w, h, w, h, 3, math.radians(45), 1, sx, sy, fill, None) == 0
        got = o.download().reshape(h, w, 3)
        assert (got == np.array([7, 8, 9], np.uint8)).all(), (sx, sy)
        want = np.zeros_like(src)
        orc.L.orc_rotate2(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, w, h, 3, math.radians(45), 1, sx, sy, fill)
        assert (want.reshape(h, w, 3) == np.array([7, 8, 9], np.uint8)).all()
    for bad in (float("nan"), float("inf")):
        assert dev.lib.gmat_rotate2(d.ptr, d.stride, o.ptr, o.stride, w, h, w, h, 3, 0.3, 1, bad, 0.0, fill, None) < 0
    d.free(); o.free()


@pytest.mark.parametrize("bpp", [1, 3, 4])
@pytest.mark.parametrize("k", [(5, 5), (3, 7), (7, 3), (9, 9), (1, 5), (5, 1), (15, 15), (31, 3)])
def test_median_of_any_odd_window(dev, orc, k, bpp):
    """vf_median.c at radius (kw - 1) / 2, radiusV (kh - 1) / 2, percentile 0.5: the (t + 1)-th smallest of the clamped window;
    frames narrower than the window clip the radius (check_params)"""
    kw, kh = k
    for w, h in ((41, 23), (6, 4)):
        src = orc.lcg((h, w * bpp), 31 + kw + kh)
        d = dev.upload_planes([src], 4)[0]
        o = DevPlane(dev, h, w * bpp, (w * bpp + 15) // 16 * 16)
        assert dev.lib.gmat_median(d.ptr, d.stride, o.ptr, o.stride, w, h, bpp, kw, kh, None) == 0
        want = np.zeros_like(src)
        orc.L.orc_median(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, bpp, kw, kh)
        got = o.download()
        assert (got == want).all(), np.argwhere(got != want)[:4]
        assert (o.download(with_padding=True)[:, w * bpp:] == 0xCD).all()
        d.free(); o.free()
    assert dev.lib.gmat_median(0, 0, 0, 0, 8, 8, 1, 4, 3, None) < 0          # even


def test_median_3x3_rule_is_the_general_rule(dev, orc):
    """the strip / byte-wise 3 x 3 kernels and the bisecting kernel state the same thing: orc_median at 3 x 3 == orc_median3x3"""
    src = orc.lcg((19, 33 * 3), 5)
    a, b = np.zeros_like(src), np.zeros_like(src)
    orc.L.orc_median(src.ctypes.data, src.strides[0], a.ctypes.data, a.strides[0], 33, 19, 3, 3, 3)
    orc.L.orc_median3x3(src.ctypes.data, src.strides[0], b.ctypes.data, b.strides[0], 33, 19, 3)
    assert (a == b).all()


def test_smooth_filter_median_5x5(dev, orc):
    w, h = 64, 24
    src = orc.lcg((h, w * 3), 91)
    res, _, _ = _run_filter(dev, "smooth_hip", {"type": "median", "kw": 5, "kh": 7}, src, w, h)
    want = np.zeros_like(src)
    orc.L.orc_median(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, 3, 5, 7)
    assert (res == want).all()


@pytest.mark.parametrize("bpp", [1, 2, 3, 4])
def test_rotate_bilinear_interior_tiles(dev, orc, bpp):
    """the single-precision form of interpolate_bilinear8 (round 4: whole 64 x 32 macro tiles whose source box met no clamp — one fused
    multiply-add rounded toward zero per sample, k_transform.hip rot_fma2 / rot_put_u8) against the 64-bit integer rule of vf_rotate.c:224-249
    on frames that are mostly such tiles, a flat-ish source (long runs of equal neighbours: differences of 0), a noisy one and the
    extremes 0 / 255 next to each other (the blend's largest differences)"""
    w, h = 352, 256
    rng = np.random.default_rng(40 + bpp)
    srcs = [orc.lcg((h, w * bpp), 70 + bpp), (rng.integers(0, 2, (h, w * bpp)) * 255).astype(np.uint8),
            np.repeat(rng.integers(0, 256, (h, w * bpp // 8), dtype=np.uint8), 8, axis=1)]
    for src, deg, interp in zip(srcs * 3, [17.0, -3.25, 45.0, 91.0, 200.5, 0.01, 17.0, 135.0, -60.0], [1, 1, 1, 1, 1, 1, 0, 0, 0]):
        src = np.ascontiguousarray(src)
        want = np.zeros_like(src)
        orc.L.orc_rotate(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, w, h, bpp, math.radians(deg), interp, None)
        d = dev.upload_planes([src], 256, 0)[0]
        o = DevPlane(dev, h, w * bpp, (w * bpp + 255) // 256 * 256)
        o.upload(np.zeros((h, w * bpp), np.uint8))
        assert dev.lib.gmat_rotate(d.ptr, d.stride, o.ptr, o.stride, w, h, w, h, bpp, math.radians(deg), interp, None, None) == 0
        got = o.download()
        assert (got == want).all(), ((bpp, deg, interp), np.argwhere(got != want)[:4].tolist())
        d.free(); o.free()


@pytest.mark.parametrize("interp", [0, 1, 2])
@pytest.mark.parametrize("bpp", [1, 3, 4])
def test_rotate_tiles_outside_the_source(dev, orc, interp, bpp):
    """whole 64 x 32 tiles none of whose pixels maps into the source (a shift pushes the picture out of most of the frame): the fill
    colour, or the destination as it was — the tile-level short cut of rotate_mt_kernel next to tiles that cross the source's rim"""
    w, h = 384, 224
    src = orc.lcg((h, w * bpp), 55 + bpp)
    for deg, sx, sy, fill in [(17.0, 210.0, 120.0, (9, 8, 7, 6)), (-40.0, -260.0, 90.0, None), (3.0, 0.0, -190.0, (255, 0, 1, 254))]:
        fp = (C.c_uint8 * 4)(*fill) if fill else None
        before = orc.lcg((h, w * bpp), 77)
        want = before.copy()
        orc.L.orc_rotate2(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, w, h, bpp, math.radians(deg), interp, sx, sy, fp)
        d = dev.upload_planes([src], 256, 0)[0]
        o = DevPlane(dev, h, w * bpp, (w * bpp + 255) // 256 * 256)
        o.upload(before)
        assert dev.lib.gmat_rotate2(d.ptr, d.stride, o.ptr, o.stride, w, h, w, h, bpp, math.radians(deg), interp, sx, sy, fp, None) == 0
        got = o.download()
        assert (got == want).all(), ((bpp, deg, interp), np.argwhere(got != want)[:4].tolist())
        assert (o.download(with_padding=True)[:, w * bpp:] == 0xCD).all()
        d.free(); o.free()
