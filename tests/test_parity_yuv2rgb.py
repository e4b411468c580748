"""HIP colour-conversion kernels vs the oracle, through the C ABI (bit-exact).

Runs twice: `emu` (same kernel sources on the CPU emulation, in the not-gpu suite) and
`hip` (the product library on a real MI355X, -m gpu).
"""
import ctypes as C

import numpy as np
import pytest

from harness import PIX_FMT, SWS, planes, ints, synth_planes

SIZES = [(64, 16), (260, 34), (130, 7), (3, 3), (1, 1), (517, 9)]


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("src_fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("dst_fmt", ["rgb24", "bgr24", "rgba", "bgra"])
def test_yuv2rgb_bit_exact(dev, orc, w, h, src_fmt, dst_fmt):
    src = synth_planes(orc, src_fmt, w, h, seed=7)
    want = orc.yuv2rgb(src, w, h, src_fmt, dst_fmt)
    for align, extra in [(256, 0), (1, 0), (1, 5)]:          # aligned, tight, deliberately misaligned
        d_src = dev.upload_planes(src, align, extra)
        got, pads, kernel = dev.sws(d_src, w, h, src_fmt, w, h, dst_fmt, dst_align=align if align > 1 else 1,
                                    dst_extra=extra)
        assert kernel == "yuv2rgb_kernel"
        assert (got[0] == want).all(), f"mismatch at align={align} extra={extra}"
        assert (pads[0] == 0xCD).all(), "kernel wrote into the row padding"
        for p in d_src:
            p.free()


def test_yuv2rgb_colorspace_and_range(dev, orc):
    w, h = 96, 20
    src = synth_planes(orc, "nv12", w, h, seed=11)
    d_src = dev.upload_planes(src, 64)
    for cs, full in [(1, 0), (9, 0), (5, 1)]:
        want = orc.yuv2rgb(src, w, h, "nv12", "rgb24", cs, full)
        got, _, _ = dev.sws(d_src, w, h, "nv12", w, h, "rgb24", colorspace=(cs, full))
        assert (got[0] == want).all()


def test_reference_entry_point_yuv2rgb_cuda(dev, orc):
    # the plain-pointer symbol libswscale core binds (swscale_unscaled.c:1980)
    w, h = 128, 24
    src = synth_planes(orc, "nv12", w, h, seed=5)
    d_src = dev.upload_planes(src, 256)
    dst = dev.planes_like("rgb24", w, h, 256)
    r = dev.lib.yuv2rgb_cuda(planes([p.ptr for p in d_src]), ints([p.stride for p in d_src]),
                             planes([p.ptr for p in dst]), ints([p.stride for p in dst]), w, h,
                             PIX_FMT["nv12"], PIX_FMT["rgb24"], None)
    assert r == 0
    assert (dst[0].download() == orc.yuv2rgb(src, w, h, "nv12", "rgb24")).all()


@pytest.mark.parametrize("w,h", [(64, 8), (130, 6), (5, 3)])
def test_rgb24_bgr24_swap(dev, orc, w, h):
    src = synth_planes(orc, "rgb24", w, h, seed=3)
    want = np.zeros_like(src[0])
    orc.L.orc_rgb24_swap_rb(src[0].ctypes.data, src[0].strides[0], want.ctypes.data, want.strides[0], w, h)
    for align, extra in [(256, 0), (1, 1)]:
        d_src = dev.upload_planes(src, align, extra)
        got, pads, kernel = dev.sws(d_src, w, h, "rgb24", w, h, "bgr24", dst_align=align, dst_extra=extra)
        assert kernel == "swap_rb24_kernel"
        assert (got[0] == want).all() and (pads[0] == 0xCD).all()


@pytest.mark.parametrize("w,h", [(64, 8), (36, 6)])
def test_nv12_to_rgbpf32_cswscale_shape(dev, orc, w, h):
    # metrans/app/CSwscale.c: tightly packed nv12 in, three stacked float planes out, value = u8/255
    src = synth_planes(orc, "nv12", w, h, seed=9)
    want = orc.nv12_to_rgbpf32(src, w, h)
    packed = np.concatenate([src[0].reshape(-1), src[1].reshape(-1)])
    from harness import DevBuf, ptr
    din = DevBuf(dev, packed.size)
    dev.lib.gmat_memcpy_h2d(din.ptr, ptr(packed), packed.size)
    dout = DevBuf(dev, 12 * w * h)
    ctx = dev.lib.SwscaleCuda_Nv12ToRgbpf32_Init(w, h)
    assert ctx
    r = dev.lib.SwscaleCuda_Nv12ToRgbpf32_Convert(ctx, din.ptr, w, dout.ptr, 4 * w, w, h, None)
    assert r == h
    host = np.empty((3, h, w), np.float32)
    dev.lib.gmat_device_sync()
    dev.lib.gmat_memcpy_d2h(ptr(host), dout.ptr, host.nbytes)
    dev.lib.SwscaleCuda_Nv12ToRgbpf32_Delete(ctx)
    assert (host.view(np.uint32) == want.view(np.uint32)).all()      # bit-identical floats
    din.free(); dout.free()


def test_error_behaviour(dev):
    lib = dev.lib
    assert not lib.gmat_sws_getContext(0, 10, 23, 10, 10, 2, 0, None)             # invalid dimension
    assert not lib.gmat_sws_getContext(16, 16, 999, 16, 16, 2, 0, None)           # unsupported format
    c = lib.gmat_sws_getContext(16, 16, PIX_FMT["nv12"], 16, 16, PIX_FMT["rgb24"], 0, None)
    assert c
    assert lib.gmat_sws_scale(c, None, None, 0, 16, None, None) < 0                # NULL parameters
    src = dev.planes_like("nv12", 16, 16)
    dst = dev.planes_like("rgb24", 16, 16)
    r = lib.gmat_sws_scale(c, planes([p.ptr for p in src]), ints([p.stride for p in src]), 4, 8,
                           planes([p.ptr for p in dst]), ints([p.stride for p in dst]))
    assert r < 0                                                                    # partial slices rejected
    lib.gmat_sws_freeContext(c)
