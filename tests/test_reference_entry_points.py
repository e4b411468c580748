"""The plain-pointer back-end symbols libswscale core binds (swscale_unscaled.c:1970-2012), under the reference's names:
every format pair the reference's own dispatch lists is served, with libswscale's CPU arithmetic (the oracle).

  yuv2rgb_cuda   libswscale/cuda/yuv2rgb_cuda.cu:862-907
  rgb2yuv_cuda   libswscale/cuda/yuv2rgb_cuda.cu:909-947
  yuv2yuv_cuda   libswscale/cuda/yuv2yuv_cuda.cu:324-366
"""
import numpy as np
import pytest

from harness import PIX_FMT, SWS, synth_planes, planes, ints

W, H = 128, 36


def _call(dev, name, d_src, s_fmt, dst, d_fmt, w=W, h=H):
    return getattr(dev.lib, name)(planes([p.ptr for p in d_src]), ints([p.stride for p in d_src]),
                                  planes([p.ptr for p in dst]), ints([p.stride for p in dst]), w, h,
                                  PIX_FMT[s_fmt], PIX_FMT[d_fmt], None)


@pytest.mark.parametrize("s_fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("d_fmt", ["rgb24", "bgr24", "rgba", "bgra", "rgba64le", "bgra64le"])
def test_reference_entry_point_yuv2rgb_cuda_every_pair(dev, orc, s_fmt, d_fmt):
    src = synth_planes(orc, s_fmt, W, H, seed=5)
    d_src = dev.upload_planes(src, 256)
    dst = dev.planes_like(d_fmt, W, H, 256)
    for _ in range(2):
        assert _call(dev, "yuv2rgb_cuda", d_src, s_fmt, dst, d_fmt) == 0
    if d_fmt.endswith("64le"):
        want = orc.sws(src, W, H, s_fmt, W, H, d_fmt)[0]      # the context's 19-bit lines (yuv2rgba64_*_c)
    else:
        want = orc.yuv2rgb(src, W, H, s_fmt, d_fmt)           # yuv2rgb.c's nearest-chroma converters
    assert (dst[0].download() == want).all()
    for p in d_src + dst:
        p.free()


def test_reference_entry_point_yuv2rgb_cuda_rgbpf32(dev, orc):
    src = synth_planes(orc, "nv12", W, H, seed=7)
    d_src = dev.upload_planes(src, 256)
    dst = dev.planes_like("rgbpf32le", W, H, 1)
    # the reference's planar float frame: one allocation, planes stacked (yuv2rgb_cuda.cu:381-545)
    from harness import DevBuf
    buf = DevBuf(dev, 3 * 4 * W * H)
    r = dev.lib.yuv2rgb_cuda(planes([p.ptr for p in d_src]), ints([p.stride for p in d_src]),
                             planes([buf.ptr]), ints([4 * W]), W, H, PIX_FMT["nv12"], PIX_FMT["rgbpf32le"], None)
    assert r == 0
    got = np.empty((3, H, W), np.float32)
    dev.lib.gmat_device_sync()
    dev.lib.gmat_memcpy_d2h(got.ctypes.data, buf.ptr, got.nbytes)
    assert (got == orc.nv12_to_rgbpf32(src, W, H)).all()
    buf.free()
    for p in d_src + dst:
        p.free()


@pytest.mark.parametrize("d_fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("s_fmt", ["rgb24", "bgr24", "rgba", "bgra", "rgba64le", "bgra64le"])
def test_reference_entry_point_rgb2yuv_cuda_every_pair(dev, orc, s_fmt, d_fmt):
    src = synth_planes(orc, s_fmt, W, H, seed=55)
    want = orc.sws(src, W, H, s_fmt, W, H, d_fmt)
    d_src = dev.upload_planes(src, 256)
    dst = dev.planes_like(d_fmt, W, H, 256)
    for _ in range(2):                                   # second call hits the cached tables
        assert _call(dev, "rgb2yuv_cuda", d_src, s_fmt, dst, d_fmt) == 0
    assert all((d.download() == wv).all() for d, wv in zip(dst, want))
    for p in d_src + dst:
        p.free()


@pytest.mark.parametrize("s_fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("d_fmt", ["nv12", "yuv420p", "p010le", "p016le", "yuv420p10le", "yuv420p16le"])
def test_reference_entry_point_yuv2yuv_cuda_every_pair(dev, orc, s_fmt, d_fmt):
    """yuv2yuv_cuda.cu:324-366: equal formats copy, NV12 / YUV420P -> the other layout, P010, P016, YUV420P10, YUV420P16"""
    w, h = 130, 34
    src = synth_planes(orc, s_fmt, w, h, seed=57)
    d_src = dev.upload_planes(src, 64)
    dst = dev.planes_like(d_fmt, w, h, 64)
    assert _call(dev, "yuv2yuv_cuda", d_src, s_fmt, dst, d_fmt, w, h) == 0
    if d_fmt in ("p010le", "p016le") and s_fmt == "yuv420p":
        # planar8ToP01xleWrapper (swscale_unscaled.c:286-324): libswscale's converter for planar 8-bit sources (an NV12 source runs
        # the generic lines, the else branch)
        from harness import alloc_planes
        want = alloc_planes(d_fmt, w, h, fill=0xCD)
        orc.L.orc_yuv420_to_p01x(planes([p.ctypes.data for p in src]), ints([p.strides[0] for p in src]),
                                 planes([p.ctypes.data for p in want]), ints([p.strides[0] for p in want]), w, h,
                                 0)
    else:
        want = orc.sws(src, w, h, s_fmt, w, h, d_fmt)
    for d, wv in zip(dst, want):
        assert (d.download() == wv).all()
    for p in d_src + dst:
        p.free()


@pytest.mark.parametrize("fmt", ["p010le", "p016le", "yuv420p10le", "yuv420p16le"])
def test_reference_entry_point_yuv2yuv_cuda_same_format_is_a_copy(dev, orc, fmt):
    w, h = 66, 18
    src = synth_planes(orc, fmt, w, h, seed=59)
    d_src = dev.upload_planes(src, 64)
    dst = dev.planes_like(fmt, w, h, 64)
    assert _call(dev, "yuv2yuv_cuda", d_src, fmt, dst, fmt, w, h) == 0
    for d, s in zip(dst, src):
        assert (d.download() == s).all()
    for p in d_src + dst:
        p.free()


def test_reference_entry_points_refuse_what_the_reference_does_not_list(dev, orc):
    src = synth_planes(orc, "rgb24", W, H, seed=61)
    d_src = dev.upload_planes(src, 256)
    dst = dev.planes_like("rgb24", W, H, 256)
    assert _call(dev, "rgb2yuv_cuda", d_src, "rgb24", dst, "rgb24") < 0
    assert _call(dev, "yuv2yuv_cuda", d_src, "rgb24", dst, "nv12") < 0
    assert _call(dev, "yuv2rgb_cuda", d_src, "rgb24", dst, "rgb24") < 0
    for p in d_src + dst:
        p.free()


def test_sws_scale_ignores_the_slice_like_the_reference_back_end(dev, orc):
    """ff_swscale_cuda never reads srcSliceY / srcSliceH (swscale_cuda.c:273-479); the core checks their range first
    (swscale.c:902-907).  A slice inside the frame converts the WHOLE frame; one beyond it is EINVAL; an empty one returns 0."""
    w, h = 64, 16
    src = synth_planes(orc, "nv12", w, h, seed=63)
    want = orc.yuv2rgb(src, w, h, "nv12", "rgb24")
    d_src = dev.upload_planes(src, 64)
    dst = dev.planes_like("rgb24", w, h, 64)
    lib = dev.lib
    c = lib.gmat_sws_getContext(w, h, PIX_FMT["nv12"], w, h, PIX_FMT["rgb24"], SWS["hwaccel"], None)
    args = (planes([p.ptr for p in d_src]), ints([p.stride for p in d_src]))
    out = (planes([p.ptr for p in dst]), ints([p.stride for p in dst]))
    assert lib.gmat_sws_scale(c, *args, 4, 8, *out) == h
    assert (dst[0].download() == want).all()
    assert lib.gmat_sws_scale(c, *args, 8, 16, *out) < 0
    assert lib.gmat_sws_scale(c, *args, -2, 8, *out) < 0
    assert lib.gmat_sws_scale(c, *args, 16, 0, *out) == 0
    lib.gmat_sws_freeContext(c)
    for p in d_src + dst:
        p.free()
