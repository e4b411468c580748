"""MeTrans front-ends (include/gmat_metrans.h): exported under the reference's C++-mangled names and served by
the same kernels — checked against the oracle through those symbols."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from harness import PIX_FMT, SWS, synth_planes, DevBuf, ptr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MANGLED = {
    "Nv12ToBgra32": "_Z12Nv12ToBgra32PhiS_iiiiP11CUstream_st",
    "Nv12ToRgba32": "_Z12Nv12ToRgba32PhiS_iiiiP11CUstream_st",
    "ScaleNv12": "_Z9ScaleNv12PhiiiS_iii",
    "ScaleNv12_Bicubic": "_Z17ScaleNv12_BicubicPhiiiS_iii",
}


def test_header_declares_exactly_the_exported_front_ends():
    text = open(os.path.join(ROOT, "include", "gmat_metrans.h")).read()
    declared = set(re.findall(r"GMAT_MT_API\s+void\s+(\w+)\s*\(", text))
    assert declared == set(MANGLED)


def _upload_nv12(dev, src):
    packed = np.concatenate([src[0].reshape(-1), src[1].reshape(-1)])
    buf = DevBuf(dev, packed.size)
    dev.lib.gmat_memcpy_h2d(buf.ptr, ptr(packed), packed.size)
    return buf


@pytest.mark.parametrize("name,fmt", [("Nv12ToBgra32", "bgra"), ("Nv12ToRgba32", "rgba")])
@pytest.mark.parametrize("matrix", [1, 6])
def test_nv12_to_32bit_front_ends(dev, orc, name, fmt, matrix):
    w, h = 128, 36
    src = synth_planes(orc, "nv12", w, h, seed=81)
    want = orc.yuv2rgb(src, w, h, "nv12", fmt, colorspace=matrix, full_range=0)
    fn = getattr(dev.lib, MANGLED[name])
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    din = _upload_nv12(dev, src)
    dout = DevBuf(dev, 4 * w * h)
    fn(din.ptr, w, dout.ptr, 4 * w, w, h, matrix, None)
    dev.lib.gmat_device_sync()
    got = np.empty((h, 4 * w), np.uint8)
    dev.lib.gmat_memcpy_d2h(ptr(got), dout.ptr, got.size)
    assert (got == want).all()
    din.free(); dout.free()


@pytest.mark.parametrize("name,flags", [("ScaleNv12", "bilinear"), ("ScaleNv12_Bicubic", "bicubic")])
def test_scale_nv12_front_ends(dev, orc, name, flags):
    sw, sh, dw, dh = 256, 64, 128, 32
    src = synth_planes(orc, "nv12", sw, sh, seed=83)
    want = orc.sws(src, sw, sh, "nv12", dw, dh, "nv12", SWS[flags])
    fn = getattr(dev.lib, MANGLED[name])
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    din = _upload_nv12(dev, src)
    dout = DevBuf(dev, dw * dh * 3 // 2)
    for _ in range(2):                                       # second call uses the cached context
        fn(din.ptr, sw, sw, sh, dout.ptr, dw, dw, dh)
    dev.lib.gmat_device_sync()
    got = np.empty((dh * 3 // 2, dw), np.uint8)
    dev.lib.gmat_memcpy_d2h(ptr(got), dout.ptr, got.size)
    assert (got[:dh] == want[0]).all() and (got[dh:] == want[1]).all()
    din.free(); dout.free()
