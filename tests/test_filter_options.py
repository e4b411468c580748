"""The option contract of the AVFilter-shaped layer (option names and defaults are part of it): every
option of vf_scale_cuda.c:586-603 and vf_smooth_nvcv.c:88-105 either has its effect or is refused — none is accepted and
ignored."""
import ctypes as C

import numpy as np
import pytest

from harness import PIX_FMT, SWS, DevPlane
from gmat_amd.lib import GmatFrame
from test_parity_filters import _run_filter


def _cfg(dev, name, opts, w, h, fmt="rgb24"):
    """init + config_props only: returns (code of the first failing call or 0, out_w, out_h)"""
    lib = dev.lib
    fc = lib.gmat_hwframe_ctx_create(0, PIX_FMT[fmt], w, h, 0)
    f = lib.gmat_filter_alloc(name.encode())
    try:
        for k, v in opts.items():
            r = lib.gmat_filter_set_option(f, k.encode(), str(v).encode())
            if r < 0:
                return r, 0, 0
        r = lib.gmat_filter_init(f)
        if r < 0:
            return r, 0, 0
        r = lib.gmat_filter_config_props(f, fc, None)
        if r < 0:
            return r, 0, 0
        of = lib.gmat_filter_out_frames(f)
        ow, oh = C.c_int(), C.c_int()
        lib.gmat_hwframe_ctx_info(of, None, None, C.byref(ow), C.byref(oh))
        return 0, ow.value, oh.value
    finally:
        lib.gmat_filter_free(f)
        lib.gmat_hwframe_ctx_free(fc)


# ---- gaussian: kw x kh, sigma, borders ----------------------------------------------------------------------------
@pytest.mark.parametrize("border", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("case", [(3, 3, 0.0, 0.0), (5, 5, 0.0, 0.0), (7, 3, 0.0, 0.0), (9, 9, 0.0, 0.0), (5, 11, 1.7, 0.0),
                                  (3, 3, 2.0, 0.5), (1, 1, 0.0, 0.0), (31, 31, 6.0, 6.0), (13, 1, 0.0, 3.0),
                                  (63, 5, 0.0, 0.0), (3, 101, 20.0, 0.0), (255, 1, 40.0, 0.0)])       # round 4: windows up to 255 taps an axis
@pytest.mark.parametrize("bpp", [1, 3, 4])
def test_gauss_blur_matches_the_stated_rule(dev, orc, case, border, bpp):
    kw, kh, sx, sy = case
    w, h = (37, 19) if kw < 31 else (21, 9)               # frames smaller than the window too (borders fold repeatedly)
    src = orc.lcg((h, w * bpp), 17 + kw + border)
    d = dev.upload_planes([src], 4)[0]
    o = DevPlane(dev, h, w * bpp, (w * bpp + 15) // 16 * 16)
    assert dev.lib.gmat_gauss_blur(d.ptr, d.stride, o.ptr, o.stride, w, h, bpp, kw, kh, sx, sy, border, None) == 0
    want = np.zeros_like(src)
    assert orc.L.orc_gauss_blur(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, bpp, kw, kh, sx, sy, border) == 0
    got = o.download()
    assert (got == want).all(), np.argwhere(got != want)[:4]
    assert (o.download(with_padding=True)[:, w * bpp:] == 0xCD).all()
    d.free(); o.free()


def test_gauss_default_kernel_is_the_integer_one_in_the_interior(dev, orc):
    """sigma <= 0 at 3x3 is OpenCV's fixed 1/4 1/2 1/4 table: the float kernel and the integer 1-2-1 kernel agree away
    from the border (their border rules differ: vf_convolution.c:555-569 mirrors, border_type 0 pads with 0)"""
    w, h = 40, 20
    src = orc.lcg((h, w * 3), 9)
    a, _, _ = _run_filter(dev, "smooth_hip", {}, src, w, h)
    b, _, _ = _run_filter(dev, "smooth_hip", {"border_type": "constant"}, src, w, h)
    assert (a[1:-1, 3:-3] == b[1:-1, 3:-3]).all() and (a != b).any()


def test_smooth_options_have_effect_or_are_refused(dev, orc):
    w, h = 48, 24
    src = orc.lcg((h, w * 3), 5)

    def want(kw, kh, sx, sy, border):
        o = np.zeros_like(src)
        assert orc.L.orc_gauss_blur(src.ctypes.data, src.strides[0], o.ctypes.data, o.strides[0], w, h, 3, kw, kh, sx, sy, border) == 0
        return o
    res, _, _ = _run_filter(dev, "smooth_hip", {"kw": 5, "kh": 7}, src, w, h)
    assert (res == want(5, 7, 0, 0, 0)).all()                              # border_type defaults to constant (:93)
    res, _, _ = _run_filter(dev, "smooth_hip", {"sigmaX": 2}, src, w, h)
    assert (res == want(3, 3, 2.0, 0, 0)).all()
    res, _, _ = _run_filter(dev, "smooth_hip", {"kw": 9, "kh": 9, "sigmaX": 1.5, "sigmaY": 3, "border_type": "reflect101"}, src, w, h)
    assert (res == want(9, 9, 1.5, 3.0, 4)).all()
    res, _, _ = _run_filter(dev, "smooth_hip", {"border_type": "warp"}, src, w, h)          # the reference's spelling of wrap
    assert (res == want(3, 3, 0, 0, 3)).all()
    assert _cfg(dev, "smooth_hip", {"kw": 4}, w, h)[0] < 0                 # even
    assert _cfg(dev, "smooth_hip", {"kw": 33, "kh": 33}, w, h)[0] == 0     # (round 4: up to 255 taps an axis)
    assert _cfg(dev, "smooth_hip", {"kw": 257, "kh": 3}, w, h)[0] < 0      # beyond the implemented size
    assert _cfg(dev, "smooth_hip", {"border_type": "mirror"}, w, h)[0] < 0
    assert _cfg(dev, "smooth_hip", {"sigmaX": -1}, w, h)[0] < 0
    assert _cfg(dev, "smooth_hip", {"type": "median", "kw": 5, "kh": 5}, w, h)[0] == 0    # vf_median.c's rule at radius 2 (round 3)
    assert _cfg(dev, "smooth_hip", {"type": "median", "kw": 4, "kh": 5}, w, h)[0] < 0     # even
    assert _cfg(dev, "smooth_hip", {"type": "median", "kw": 33, "kh": 3}, w, h)[0] == 0
    assert _cfg(dev, "smooth_hip", {"type": "median", "kw": 257, "kh": 3}, w, h)[0] < 0   # vf_median.c's radius ends at 127
    assert _cfg(dev, "smooth_hip", {"type": "median", "border_type": "reflect"}, w, h)[0] < 0   # gaussian-only option
    assert _cfg(dev, "smooth_hip", {"type": "median", "sigmaX": 1}, w, h)[0] < 0
    assert _cfg(dev, "smooth_hip", {"type": "boxcar"}, w, h)[0] < 0
    assert _cfg(dev, "smooth_hip", {"type": "median"}, w, h)[0] == 0


# ---- scale: expressions, -1 / -n, aspect-ratio forcing, passthrough, param ------------------------------------------
@pytest.mark.parametrize("opts,want", [
    ({}, (96, 40)),                                             # defaults "iw" / "ih"
    ({"w": "iw/2", "h": "ih/2"}, (48, 20)),
    ({"w": "iw/2", "h": -1}, (48, 20)),
    ({"w": -1, "h": 30}, (72, 30)),
    ({"w": 50, "h": -2}, (50, 20)),                             # 50 * 40 / 96 = 20.83 -> multiple of 2 nearest: 20
    ({"w": -4, "h": 30}, (72, 30)),
    ({"w": "2*iw", "h": "oh"}, None),                           # self-reference: NaN -> error
    ({"w": "ow", "h": 10}, None),
    ({"w": "oh*2", "h": 16}, (32, 16)),                         # w may refer to the output height (second pass)
    ({"w": "trunc(iw/3)", "h": "max(ih/4, 12)"}, (32, 12)),
    ({"w": "min(iw, 64)", "h": "ih - 8"}, (64, 32)),
    ({"w": "iw*sar/dar*a", "h": "ih/vsub*ovsub"}, (96, 40)),
    ({"w": 0, "h": 0}, (96, 40)),                               # 0 -> input size (scale_eval.c:86,:92)
    ({"w": "(iw", "h": 10}, None),
    ({"w": "iw/2 foo", "h": 10}, None),
    ({"w": "bar", "h": 10}, None),
    ({"w": -1, "h": -1}, (96, 40)),
    ({"w": 200, "h": 20, "force_original_aspect_ratio": "decrease"}, (48, 20)),
    ({"w": 200, "h": 20, "force_original_aspect_ratio": "increase"}, (200, 83)),
    ({"w": 50, "h": 50, "force_original_aspect_ratio": "decrease", "force_divisible_by": 8}, (48, 16)),
    ({"w": 50, "h": 50, "force_original_aspect_ratio": 2, "force_divisible_by": 16}, (128, 64)),
    ({"w": 50, "h": 50, "force_original_aspect_ratio": "sideways"}, None),
    ({"w": 50, "h": 50, "force_divisible_by": 0}, None),
    ({"w": -100, "h": 10}, (0, 0)),                              # rounds to a multiple of 100 -> 0: invalid size
])
def test_scale_size_expressions(dev, opts, want):
    r, ow, oh = _cfg(dev, "scale_hip", opts, 96, 40)
    if want is None or want == (0, 0):
        assert r < 0, (opts, ow, oh)
    else:
        assert r == 0 and (ow, oh) == want, (opts, r, ow, oh)


def test_scale_expression_chroma_variables_are_those_of_a_hardware_link(dev):
    """hsub / vsub / ohsub / ovsub come from the LINK's pixel format in ff_scale_eval_dimensions (scale_eval.c:76-83); for a
    hardware link that is the hardware format, whose descriptor has no chroma shift: all four are 1 under scale_cuda, whatever the
    sw_format — and through the AVFilter glue of integration/vf_gmat_hip.c, which calls that function"""
    r, ow, oh = _cfg(dev, "scale_hip", {"w": "iw/hsub", "h": "ih/vsub", "format": "rgb24"}, 96, 40, fmt="nv12")
    assert r == 0 and (ow, oh) == (96, 40)
    r, ow, oh = _cfg(dev, "scale_hip", {"w": "iw/ohsub", "h": "ih/ovsub", "format": "rgb24"}, 96, 40, fmt="nv12")
    assert r == 0 and (ow, oh) == (96, 40)


def test_scale_passthrough_hands_the_frame_on(dev, orc):
    """vf_scale_cuda.c:254-260,:543: same size and format with passthrough=1 (the default) does not touch the frame"""
    lib = dev.lib
    w, h = 64, 16
    fc = lib.gmat_hwframe_ctx_create(0, PIX_FMT["rgb24"], w, h, 1)
    for pt, same in ((None, True), (1, True), (0, False)):
        f = lib.gmat_filter_alloc(b"scale_hip")
        if pt is not None:
            assert lib.gmat_filter_set_option(f, b"passthrough", str(pt).encode()) == 0
        assert lib.gmat_filter_init(f) == 0 and lib.gmat_filter_config_props(f, fc, None) == 0
        fin = lib.gmat_frame_alloc()
        assert lib.gmat_hwframe_get_buffer(fc, fin) == 0
        src = orc.lcg((h, w * 3), 3)
        host = np.zeros((h, fin.contents.linesize[0]), np.uint8)
        host[:, :w * 3] = src
        assert lib.gmat_memcpy_h2d(fin.contents.data[0], host.ctypes.data, host.size) == 0
        in_ptr = fin.contents.data[0]
        out = C.POINTER(GmatFrame)()
        assert lib.gmat_filter_frame(f, fin, C.byref(out)) == 0
        lib.gmat_device_sync()
        assert (out.contents.data[0] == in_ptr) == same
        back = np.zeros_like(host)
        assert lib.gmat_memcpy_d2h(back.ctypes.data, out.contents.data[0], back.size) == 0
        assert (back[:, :w * 3] == src).all()              # rgb24 -> rgb24 at equal size is a copy either way
        lib.gmat_frame_free(C.byref(out))
        lib.gmat_filter_free(f)
    # a format change at equal size is NOT a passthrough
    f = lib.gmat_filter_alloc(b"scale_hip")
    assert lib.gmat_filter_set_option(f, b"format", b"bgr24") == 0
    assert lib.gmat_filter_init(f) == 0 and lib.gmat_filter_config_props(f, fc, None) == 0
    fin = lib.gmat_frame_alloc()
    assert lib.gmat_hwframe_get_buffer(fc, fin) == 0
    in_ptr = fin.contents.data[0]
    out = C.POINTER(GmatFrame)()
    assert lib.gmat_filter_frame(f, fin, C.byref(out)) == 0
    assert out.contents.data[0] != in_ptr and out.contents.sw_format == PIX_FMT["bgr24"]
    lib.gmat_device_sync()
    lib.gmat_frame_free(C.byref(out))
    lib.gmat_filter_free(f)
    lib.gmat_hwframe_ctx_free(fc)


@pytest.mark.parametrize("algo,param", [("bicubic", 0.5), ("bicubic", 1.0), ("lanczos", 2.0)])
def test_scale_param_is_libswscale_param0(dev, orc, algo, param):
    """param is not ignored: it reaches the filter generator as libswscale's param[0] (bicubic B / lanczos lobes)"""
    w, h, dw, dh = 96, 40, 40, 24
    src = orc.lcg((h, w * 3), 4)
    res, ow, oh = _run_filter(dev, "scale_hip", {"w": dw, "h": dh, "interp_algo": algo, "param": param}, src, w, h)
    pr = (C.c_double * 2)(param, 123456.0)
    c = orc.L.orc_sws_create(w, h, PIX_FMT["rgb24"], dw, dh, PIX_FMT["rgb24"], SWS[algo], pr)
    assert c
    from harness import planes, ints
    want = np.zeros((dh, dw * 3), np.uint8)
    assert orc.L.orc_sws_scale(c, planes([src.ctypes.data]), ints([src.strides[0]]), planes([want.ctypes.data]), ints([want.strides[0]])) == dh
    orc.L.orc_sws_free(c)
    assert (res == want).all()
    dflt, _, _ = _run_filter(dev, "scale_hip", {"w": dw, "h": dh, "interp_algo": algo}, src, w, h)
    assert (dflt != res).any()                              # and it changes the picture


# ---- queued form: send_frame / receive_frame / flush ----------------------------------------------------------------
def _dev_frame(lib, fc, planes_np, pts):
    fr = lib.gmat_frame_alloc()
    assert lib.gmat_hwframe_get_buffer(fc, fr) == 0
    # a pool hands a buffer out again as soon as its frame is released — by the filters right after they ENQUEUE the work that reads
    # it, as libavfilter's hardware filters do: the next writer has to be ordered behind that work.  gmat_hwframe_transfer_data is
    # (it copies on the stream); this helper's plain hipMemcpy of a few KB is not (seen on the GPU: the first frame of a batch
    # overwritten by the upload of the next batch's), so it waits for the device first.
    lib.gmat_device_sync()
    for i, pl in enumerate(planes_np):
        host = np.zeros((pl.shape[0], fr.contents.linesize[i]), np.uint8)
        host[:, :pl.shape[1]] = pl
        assert lib.gmat_memcpy_h2d(fr.contents.data[i], host.ctypes.data, host.size) == 0
    fr.contents.pts = pts
    return fr


@pytest.mark.parametrize("batch,kernel", [(4, b"scale_yuv2s_blk_kernel"), (1, None), (16, b"scale_yuv2s_blk_kernel")])   # (frames this small: the block form at every launch size)
def test_queued_scale_batches_frames_into_one_launch(dev, orc, batch, kernel):
    from harness import synth_planes
    lib = dev.lib
    sw, sh, dw, dh, n = 128, 32, 64, 16, 10
    fc = lib.gmat_hwframe_ctx_create(0, PIX_FMT["nv12"], sw, sh, 2)
    f = lib.gmat_filter_alloc(b"scale_hip")
    for k, v in (("w", "iw/2"), ("h", "ih/2"), ("format", "rgb24"), ("batch", batch)):
        assert lib.gmat_filter_set_option(f, k.encode(), str(v).encode()) == 0
    assert lib.gmat_filter_init(f) == 0 and lib.gmat_filter_config_props(f, fc, None) == 0
    srcs = [synth_planes(orc, "nv12", sw, sh, seed=700 + i) for i in range(n)]
    outs = []
    out = C.POINTER(GmatFrame)()
    for i in range(n):
        assert lib.gmat_filter_send_frame(f, _dev_frame(lib, fc, srcs[i], 100 + i)) == 0
        got_now = 0
        while lib.gmat_filter_receive_frame(f, C.byref(out)) == 0:
            outs.append(out); out = C.POINTER(GmatFrame)(); got_now += 1
        assert got_now == (batch if (i + 1) % batch == 0 else 0)        # nothing before the batch is full, then all of it
    assert lib.gmat_filter_receive_frame(f, C.byref(out)) < 0            # -EAGAIN
    assert lib.gmat_filter_flush(f) == 0                                  # the partial batch at EOF
    while lib.gmat_filter_receive_frame(f, C.byref(out)) == 0:
        outs.append(out); out = C.POINTER(GmatFrame)()
    assert len(outs) == n
    lib.gmat_device_sync()
    for i, o in enumerate(outs):
        assert o.contents.pts == 100 + i and o.contents.width == dw and o.contents.sw_format == PIX_FMT["rgb24"]
        back = np.zeros((dh, o.contents.linesize[0]), np.uint8)
        assert lib.gmat_memcpy_d2h(back.ctypes.data, o.contents.data[0], back.size) == 0
        assert (back[:, :dw * 3] == orc.sws(srcs[i], sw, sh, "nv12", dw, dh, "rgb24")[0]).all(), i
        lib.gmat_frame_free(C.byref(o))
    lib.gmat_filter_free(f)
    lib.gmat_hwframe_ctx_free(fc)


def test_queued_form_of_an_unbatched_filter_is_immediate(dev, orc):
    lib = dev.lib
    w, h = 64, 16
    fc = lib.gmat_hwframe_ctx_create(0, PIX_FMT["rgb24"], w, h, 1)
    f = lib.gmat_filter_alloc(b"flip_hip")
    assert lib.gmat_filter_set_option(f, b"code", b"0") == 0
    assert lib.gmat_filter_set_option(f, b"batch", b"0") == 0 and lib.gmat_filter_init(f) < 0   # out of range, refused at init
    assert lib.gmat_filter_set_option(f, b"batch", b"1") == 0
    fcrop = lib.gmat_filter_alloc(b"crop_hip")
    assert lib.gmat_filter_set_option(fcrop, b"batch", b"4") < 0       # a crop is a copy: nothing to batch, no such option
    lib.gmat_filter_free(fcrop)
    assert lib.gmat_filter_init(f) == 0 and lib.gmat_filter_config_props(f, fc, None) == 0
    src = orc.lcg((h, w * 3), 8)
    assert lib.gmat_filter_send_frame(f, _dev_frame(lib, fc, [src], 5)) == 0
    out = C.POINTER(GmatFrame)()
    assert lib.gmat_filter_receive_frame(f, C.byref(out)) == 0 and out.contents.pts == 5
    lib.gmat_device_sync()
    back = np.zeros((h, out.contents.linesize[0]), np.uint8)
    assert lib.gmat_memcpy_d2h(back.ctypes.data, out.contents.data[0], back.size) == 0
    assert (back[:, :w * 3] == src[::-1]).all()
    lib.gmat_frame_free(C.byref(out))
    # frames still queued or ready when the filter goes are released with it (nothing leaks, nothing dangles)
    f2 = lib.gmat_filter_alloc(b"format_hip")
    assert lib.gmat_filter_set_option(f2, b"pix_fmt", b"bgr24") == 0 and lib.gmat_filter_set_option(f2, b"batch", b"8") == 0
    assert lib.gmat_filter_init(f2) == 0 and lib.gmat_filter_config_props(f2, fc, None) == 0
    for i in range(3):
        assert lib.gmat_filter_send_frame(f2, _dev_frame(lib, fc, [src], i)) == 0
    lib.gmat_filter_free(f2)
    lib.gmat_filter_free(f)
    lib.gmat_hwframe_ctx_free(fc)


@pytest.mark.parametrize("case", [("flip_hip", {"code": "1"}, "rgb24"), ("flip_hip", {"code": "0"}, "nv12"), ("transpose_hip", {"dir": "1"}, "rgb24"),
                                  ("transpose_hip", {"dir": "0"}, "yuv420p"), ("rotate_hip", {"angle": "90"}, "rgba"), ("rotate_hip", {"angle": "180"}, "nv12"),
                                  ("rotate_hip", {"angle": "270"}, "rgb24"), ("smooth_hip", {}, "rgb24"), ("smooth_hip", {"type": "median"}, "yuv444p"),
                                  ("smooth_hip", {"type": "gaussian", "kw": "5", "kh": "5"}, "rgb24"), ("rotate_hip", {"angle": "17"}, "rgb24"),
                                  ("rotate_hip", {"angle": "-33.5", "interp": "cubic", "shift_x": "5", "shift_y": "-3"}, "nv12"),
                                  ("rotate_hip", {"angle": "201", "interp": "nearest"}, "yuv420p")])
@pytest.mark.parametrize("batch", [3, 16, 20])
def test_queued_transform_filters_batch_frames_into_one_launch(dev, orc, case, batch):
    """option batch on the transform filters: flip, transpose, rotate (quarter turns and vf_rotate.c's walk at any angle — background
    and shift per plane), the 3 x 3 smooth and median take the whole queue in ONE launch per plane (gmat_op_batch / gmat_rotate2_batch:
    a grid dimension = frame; more than 16 frames: more launches); a filter form without a frame table (general gaussian) works
    through a full queue frame by frame.  Every frame must equal what filter_frame gives for it, in order, with its pts."""
    from harness import synth_planes
    name, opts, fmt = case
    lib = dev.lib
    w, h, n = 96, 40, 23
    fc = lib.gmat_hwframe_ctx_create(0, PIX_FMT[fmt], w, h, 2)

    def make(extra):
        f = lib.gmat_filter_alloc(name.encode())
        for k, v in list(opts.items()) + list(extra.items()):
            assert lib.gmat_filter_set_option(f, k.encode(), str(v).encode()) == 0, k
        assert lib.gmat_filter_init(f) == 0 and lib.gmat_filter_config_props(f, fc, None) == 0
        return f

    def fetch(fr):
        outs = []
        nplanes = 1 if fmt in ("rgb24", "rgba") else 2 if fmt == "nv12" else 3
        ow, oh = fr.contents.width, fr.contents.height
        for k in range(nplanes):
            rows = oh if k == 0 or fmt == "yuv444p" else (oh + 1) // 2
            rb = {"rgb24": 3 * ow, "rgba": 4 * ow}.get(fmt, ow if k == 0 or fmt == "yuv444p" else 2 * ((ow + 1) // 2) if fmt == "nv12" else (ow + 1) // 2)
            back = np.zeros((rows, fr.contents.linesize[k]), np.uint8)
            assert lib.gmat_memcpy_d2h(back.ctypes.data, fr.contents.data[k], back.size) == 0
            outs.append(back[:, :rb].copy())                     # (the row padding of a pool frame is whatever was there before)
        return outs

    srcs = [synth_planes(orc, fmt, w, h, seed=900 + i) for i in range(n)]
    one, many = make({}), make({"batch": batch})
    want = []
    for i, s_ in enumerate(srcs):
        out = C.POINTER(GmatFrame)()
        assert lib.gmat_filter_frame(one, _dev_frame(lib, fc, s_, i), C.byref(out)) == 0
        lib.gmat_device_sync()
        want.append(fetch(out))
        lib.gmat_frame_free(C.byref(out))
    got = []
    for i, s_ in enumerate(srcs):
        assert lib.gmat_filter_send_frame(many, _dev_frame(lib, fc, s_, i)) == 0
        while True:
            out = C.POINTER(GmatFrame)()
            if lib.gmat_filter_receive_frame(many, C.byref(out)) != 0:
                break
            got.append(out)
    assert len(got) == n // batch * batch
    assert lib.gmat_filter_flush(many) == 0
    while True:
        out = C.POINTER(GmatFrame)()
        if lib.gmat_filter_receive_frame(many, C.byref(out)) != 0:
            break
        got.append(out)
    assert len(got) == n
    lib.gmat_device_sync()
    for i, out in enumerate(got):
        assert out.contents.pts == i
        for a, b in zip(fetch(out), want[i]):
            assert (a == b).all(), (case, batch, i)
        lib.gmat_frame_free(C.byref(out))
    lib.gmat_filter_free(one); lib.gmat_filter_free(many)
    lib.gmat_hwframe_ctx_free(fc)


@pytest.mark.parametrize("op", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("bpp", [1, 3, 4])
def test_op_batch_equals_the_single_frame_calls(dev, orc, op, bpp):
    """gmat_op_batch: n frames, one launch per 16 — byte for byte what the single-frame entry points write, padding untouched;
    unaligned frames (the kernels without a frame table) go one by one inside the call"""
    from harness import DevPlane
    lib = dev.lib
    if op == 0 and bpp == 1:
        pytest.skip("the fused rotate + flip + smooth takes packed RGB")
    for (w, h, align, extra) in [(96, 40, 64, 0), (131, 35, 1, 1), (64, 64, 256, 0)]:
        n = 19
        tr = op in (0, 2)
        ow, oh = (h, w) if tr else (w, h)
        srcs = [orc.lcg((h, w * bpp), 300 + i) for i in range(n)]
        d = [dev.upload_planes([s_], align, extra)[0] for s_ in srcs]
        ostride = (ow * bpp + extra + align - 1) // align * align
        o1 = [DevPlane(dev, oh, ow * bpp, ostride) for _ in range(n)]
        o2 = [DevPlane(dev, oh, ow * bpp, ostride) for _ in range(n)]
        m = (C.c_int * 9)(1, 2, 1, 2, 4, 2, 1, 2, 1)
        for i in range(n):
            if op == 0: r = lib.gmat_rotate_flip_smooth(d[i].ptr, d[i].stride, o1[i].ptr, ostride, w, h, bpp, None)
            elif op == 1: r = lib.gmat_smooth3x3(d[i].ptr, d[i].stride, o1[i].ptr, ostride, w, h, bpp, m, 1.0 / 16, 0.0, None)
            elif op == 2: r = lib.gmat_transpose(d[i].ptr, d[i].stride, o1[i].ptr, ostride, w, h, bpp, 2, None)
            elif op == 3: r = lib.gmat_flip(d[i].ptr, d[i].stride, o1[i].ptr, ostride, w, h, bpp, -1, None)
            else: r = lib.gmat_median3x3(d[i].ptr, d[i].stride, o1[i].ptr, ostride, w, h, bpp, None)
            assert r == 0
        sp = (C.c_void_p * n)(*[p.ptr for p in d]); dp = (C.c_void_p * n)(*[p.ptr for p in o2])
        assert lib.gmat_op_batch(op, n, sp, d[0].stride, dp, ostride, w, h, bpp, 2 if op == 2 else -1 if op == 3 else 0, None) == 0
        for i in range(n):
            assert (o1[i].download(True) == o2[i].download(True)).all(), (op, bpp, (w, h), i)
        for p in d + o1 + o2:
            p.free()
    assert lib.gmat_op_batch(9, 1, sp, 4, dp, 4, 1, 1, 1, 0, None) < 0 and lib.gmat_op_batch(2, 1, sp, 4, dp, 4, 1, 1, 1, 7, None) < 0


@pytest.mark.parametrize("interp", [0, 1, 2])
@pytest.mark.parametrize("bpp", [1, 2, 3, 4])
def test_rotate2_batch_equals_the_single_frame_calls(dev, orc, interp, bpp):
    """gmat_rotate2_batch: n frames of one geometry and angle, one launch per 16 (grid.z = frame) — byte for byte what gmat_rotate2
    writes frame by frame, padding untouched, with and without a background (without: pixels whose source position is out of range
    keep what the destination held); sources that are not dword-aligned take the direct form frame by frame inside the call"""
    import math
    from harness import DevPlane
    lib = dev.lib
    fillc = (C.c_uint8 * 4)(9, 8, 7, 6)
    for (w, h, align, extra, deg, fill) in [(96, 70, 64, 0, 17.0, fillc), (131, 35, 1, 1, -48.5, fillc), (64, 64, 256, 0, 123.0, None)]:
        n = 19
        srcs = [orc.lcg((h, w * bpp), 500 + i) for i in range(n)]
        d = [dev.upload_planes([s_], align, extra)[0] for s_ in srcs]
        ostride = (w * bpp + extra + align - 1) // align * align
        o1 = [DevPlane(dev, h, w * bpp, ostride) for _ in range(n)]
        o2 = [DevPlane(dev, h, w * bpp, ostride) for _ in range(n)]
        for i in range(n):
            assert lib.gmat_rotate2(d[i].ptr, d[i].stride, o1[i].ptr, ostride, w, h, w, h, bpp, math.radians(deg), interp, 2.5, -1.0, fill, None) == 0
        sp = (C.c_void_p * n)(*[p.ptr for p in d]); dp = (C.c_void_p * n)(*[p.ptr for p in o2])
        assert lib.gmat_rotate2_batch(n, sp, d[0].stride, dp, ostride, w, h, w, h, bpp, math.radians(deg), interp, 2.5, -1.0, fill, None) == 0
        for i in range(n):
            assert (o1[i].download(True) == o2[i].download(True)).all(), (interp, bpp, (w, h), i)
        for p_ in d + o1 + o2:
            p_.free()
    assert lib.gmat_rotate2_batch(1, sp, 4, dp, 4, 1, 1, 1, 1, 1, 0.5, 3, 0.0, 0.0, None, None) < 0


# ---- a frame's own colour description drives the conversion (vf_format_cuda.c:184-217, vf_scale.c:793-824) ---------------------------
@pytest.mark.parametrize("filt,opts", [("format_hip", {"pix_fmt": "rgb24"}), ("scale_hip", {"w": "iw/2", "h": "ih/2", "format": "bgra"}),
                                       ("scale_hip", {"w": "iw/2", "h": "ih/2", "format": "rgb24", "batch": 3})])
def test_frame_colourspace_selects_the_matrix(dev, orc, filt, opts):
    """frames tagged BT.709, BT.601, untagged, BT.2020, SMPTE 240M in ONE stream (the queued form's batches hold mixed tags): each comes out in the matrix
    of its own tag, as the reference's format_cuda converts it (in->colorspace per frame) and as libavfilter's `scale` does (in_color_matrix=auto);
    tests/test_libavfilter_core.py holds the same against the reference's CPU filter with tagged frames"""
    from harness import synth_planes
    lib = dev.lib
    sw, sh = 128, 32
    sf = "yuv420p" if filt == "format_hip" else "nv12"
    df = {"rgb24": "rgb24", "bgra": "bgra"}[opts.get("pix_fmt", opts.get("format"))]
    dw, dh = (sw, sh) if filt == "format_hip" else (sw // 2, sh // 2)
    fc = lib.gmat_hwframe_ctx_create(0, PIX_FMT[sf], sw, sh, 2)
    f = lib.gmat_filter_alloc(filt.encode())
    for k, v in opts.items():
        assert lib.gmat_filter_set_option(f, k.encode(), str(v).encode()) == 0
    assert lib.gmat_filter_init(f) == 0 and lib.gmat_filter_config_props(f, fc, None) == 0
    tags = [1, 6, 2, 9, 7, 1, 1, 5]                       # AVCOL_SPC_*: bt709, smpte170m, unspecified, bt2020nc, smpte240m, ..., bt470bg
    row = {1: 1, 9: 9, 7: 7}                              # -> SWS_CS_* (the others: 5 = BT.601)
    srcs = [synth_planes(orc, sf, sw, sh, seed=900 + i) for i in range(len(tags))]
    outs, out = [], C.POINTER(GmatFrame)()
    for i, t in enumerate(tags):
        fr = _dev_frame(lib, fc, srcs[i], i)
        fr.contents.colorspace = t
        assert lib.gmat_filter_send_frame(f, fr) == 0
        while lib.gmat_filter_receive_frame(f, C.byref(out)) == 0:
            outs.append(out); out = C.POINTER(GmatFrame)()
    assert lib.gmat_filter_flush(f) == 0
    while lib.gmat_filter_receive_frame(f, C.byref(out)) == 0:
        outs.append(out); out = C.POINTER(GmatFrame)()
    assert len(outs) == len(tags)
    lib.gmat_device_sync()
    bpp = 3 if df == "rgb24" else 4
    for i, o in enumerate(outs):
        assert o.contents.pts == i and o.contents.colorspace == tags[i]
        back = np.zeros((dh, o.contents.linesize[0]), np.uint8)
        assert lib.gmat_memcpy_d2h(back.ctypes.data, o.contents.data[0], back.size) == 0
        cs = row.get(tags[i], 5)
        want = [orc.yuv2rgb(srcs[i], sw, sh, sf, df, colorspace=cs)] if filt == "format_hip" else orc.sws(srcs[i], sw, sh, sf, dw, dh, df, colorspace=cs)
        assert (back[:, :dw * bpp] == want[0]).all(), (i, tags[i])
        lib.gmat_frame_free(C.byref(o))
    lib.gmat_filter_free(f)
    lib.gmat_hwframe_ctx_free(fc)
