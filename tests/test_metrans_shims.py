"""MeTrans front-ends (include/gmat_metrans.h): all 17 entry points of metrans/include/NvCodec/NvCommon.h:232-255, exported under
the reference's C++-mangled names and checked against the oracle through those symbols.

Mangled names: g++ over the reference's own prototypes with cudaStream_t = CUstream_st * (its real definition)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from harness import SWS, synth_planes, DevBuf, ptr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CONV = "PhiS_iiiiP11CUstream_st"          # (uint8_t *, int, uint8_t *, int, int, int, int, cudaStream_t)
_CONVF = "PhiPfiiiiP11CUstream_st"         # ... float * destination
_SCALE = "PhiiiS_iii"
MANGLED = {
    "Nv12ToBgra32": "_Z12Nv12ToBgra32" + _CONV,
    "Nv12ToRgba32": "_Z12Nv12ToRgba32" + _CONV,
    "Nv12ToBgra64": "_Z12Nv12ToBgra64" + _CONV,
    "P016ToBgra32": "_Z12P016ToBgra32" + _CONV,
    "P016ToBgra64": "_Z12P016ToBgra64" + _CONV,
    "Nv12ToBgrPlanar": "_Z15Nv12ToBgrPlanar" + _CONV,
    "Nv12ToRgbPlanar": "_Z15Nv12ToRgbPlanar" + _CONV,
    "P016ToBgrPlanar": "_Z15P016ToBgrPlanar" + _CONV,
    "Nv12ToBgrFloatPlanar": "_Z20Nv12ToBgrFloatPlanar" + _CONVF,
    "Nv12ToRgbFloatPlanar": "_Z20Nv12ToRgbFloatPlanar" + _CONVF,
    "P016ToBgrFloatPlanar": "_Z20P016ToBgrFloatPlanar" + _CONVF,
    "Bgra64ToP016": "_Z12Bgra64ToP016" + _CONV,
    "ConvertUInt8ToUInt16": "_Z20ConvertUInt8ToUInt16PhPti",
    "ConvertUInt16ToUInt8": "_Z20ConvertUInt16ToUInt8PtPhi",
    "ScaleNv12": "_Z9ScaleNv12" + _SCALE,
    "ScaleP016": "_Z9ScaleP016" + _SCALE,
    "ScaleNv12_Bicubic": "_Z17ScaleNv12_Bicubic" + _SCALE,
}
_CONV_ARGS = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
_SCALE_ARGS = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]


def fn(dev, name, args):
    f = getattr(dev.lib, MANGLED[name])
    f.restype = None
    f.argtypes = args
    return f


def test_header_declares_exactly_the_17_reference_entry_points():
    text = open(os.path.join(ROOT, "include", "gmat_metrans.h")).read()
    declared = set(re.findall(r"^GMAT_MT_API\s+void\s+(\w+)\s*\(", text, re.M))
    assert declared == set(MANGLED) and len(declared) == 17


@pytest.mark.skipif(not os.path.exists("/root/reference/metrans/include/NvCodec/NvCommon.h"), reason="reference tree not present")
def test_mangled_names_are_those_of_the_reference_prototypes(tmp_path):
    """the reference's own declarations (NvCommon.h:232-255) through g++: the symbols an application object file asks for"""
    lines = open("/root/reference/metrans/include/NvCodec/NvCommon.h").read().splitlines()[231:255]
    protos = [ln for ln in lines if re.match(r"^void \w+\(", ln)]
    assert len(protos) == 17
    src = tmp_path / "m.cpp"
    src.write_text("#include <stdint.h>\nstruct CUstream_st; typedef CUstream_st *cudaStream_t;\n" +
                   "\n".join(p.rstrip(";") + " {}" for p in protos) + "\n")
    subprocess.run(["g++", "-c", str(src), "-o", str(tmp_path / "m.o")], check=True)
    syms = set(subprocess.run(["nm", str(tmp_path / "m.o")], check=True, capture_output=True, text=True).stdout.split()[2::3])
    assert syms == set(MANGLED.values())


def test_every_entry_point_is_exported(dev):
    for name, sym in MANGLED.items():
        assert getattr(dev.lib, sym) is not None, name
    assert dev.lib.gmat_metrans_bicubic_mode is not None


# ---- helpers: one allocation per semi-planar frame, chroma at base + pitch * height ----------------------------------------
def _upload(dev, arr):
    a = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    buf = DevBuf(dev, a.size)
    assert dev.lib.gmat_memcpy_h2d(buf.ptr, ptr(a), a.size) == 0
    return buf


def _semi(src, pitch):
    """[luma (h, rb), chroma (h / 2, rb)] -> one (3h / 2, pitch) image, padding 0xCD"""
    rows = src[0].shape[0] + src[1].shape[0]
    img = np.full((rows, pitch), 0xCD, np.uint8)
    img[:src[0].shape[0], :src[0].shape[1]] = src[0]
    img[src[0].shape[0]:, :src[1].shape[1]] = src[1]
    return img


def _download(dev, buf, shape, dtype=np.uint8):
    dev.lib.gmat_device_sync()
    out = np.empty(shape, dtype)
    assert dev.lib.gmat_memcpy_d2h(ptr(out), buf.ptr, out.nbytes) == 0
    return out


# ---- colour conversion ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,fmt", [("Nv12ToBgra32", "bgra"), ("Nv12ToRgba32", "rgba")])
@pytest.mark.parametrize("matrix,cs", [(1, 1), (6, 5), (0, 1), (9, 9)])
def test_nv12_to_32bit_front_ends(dev, orc, name, fmt, matrix, cs):
    """iMatrix is a ColorSpaceStandard: 6 (BT.601) is libswscale's row 5, and codes GetConstants does not list (0: what
    app/FrameExtractor.h passes) are BT.709 (ColorSpace.cu:32-64)"""
    w, h = 128, 36
    src = synth_planes(orc, "nv12", w, h, seed=81)
    want = orc.yuv2rgb(src, w, h, "nv12", fmt, colorspace=cs, full_range=0)
    din = _upload(dev, _semi(src, w))
    dout = DevBuf(dev, 4 * w * h)
    fn(dev, name, _CONV_ARGS)(din.ptr, w, dout.ptr, 4 * w, w, h, matrix, None)
    assert (_download(dev, dout, (h, 4 * w)) == want).all()
    din.free(); dout.free()


@pytest.mark.parametrize("matrix,cs", [(1, 1), (6, 5)])
def test_nv12_to_bgra64(dev, orc, matrix, cs):
    w, h = 96, 20
    src = synth_planes(orc, "nv12", w, h, seed=85)
    want = orc.sws(src, w, h, "nv12", w, h, "bgra64le", colorspace=cs)[0]
    pitch = 128
    din = _upload(dev, _semi(src, pitch))
    dout = DevBuf(dev, 8 * w * h)
    fn(dev, "Nv12ToBgra64", _CONV_ARGS)(din.ptr, pitch, dout.ptr, 8 * w, w, h, matrix, None)
    assert (_download(dev, dout, (h, 8 * w)) == want).all()
    din.free(); dout.free()


@pytest.mark.parametrize("name,fmt,bpp", [("P016ToBgra32", "bgra", 4), ("P016ToBgra64", "bgra64le", 8)])
def test_p016_to_packed(dev, orc, name, fmt, bpp):
    w, h = 64, 16
    src = synth_planes(orc, "p016le", w, h, seed=87)
    want = orc.sws(src, w, h, "p016le", w, h, fmt, colorspace=1)[0]
    pitch = 2 * w + 64
    din = _upload(dev, _semi(src, pitch))
    dout = DevBuf(dev, bpp * w * h)
    fn(dev, name, _CONV_ARGS)(din.ptr, pitch, dout.ptr, bpp * w, w, h, 1, None)
    assert (_download(dev, dout, (h, bpp * w)) == want).all()
    din.free(); dout.free()


@pytest.mark.parametrize("w,h", [(128, 36), (70, 10), (6, 4)])
@pytest.mark.parametrize("name,order,f32", [("Nv12ToBgrPlanar", "bgr", 0), ("Nv12ToRgbPlanar", "rgb", 0),
                                            ("Nv12ToBgrFloatPlanar", "bgr", 1), ("Nv12ToRgbFloatPlanar", "rgb", 1)])
def test_nv12_to_planar(dev, orc, name, order, f32, w, h):
    """three stacked planes in the order of the name, the values of the packed converter; float = value / 255 (ToValue,
    ColorSpace.cu:157-163)"""
    src = synth_planes(orc, "nv12", w, h, seed=89)
    packed = orc.yuv2rgb(src, w, h, "nv12", "rgba", colorspace=5).reshape(h, w, 4)
    idx = {"r": 0, "g": 1, "b": 2}
    want = np.stack([packed[:, :, idx[ch]] for ch in order])
    din = _upload(dev, _semi(src, w + (-w) % 2))
    es = 4 if f32 else 1
    pitch = es * w
    dout = DevBuf(dev, 3 * pitch * h)
    args = list(_CONV_ARGS)
    fn(dev, name, args)(din.ptr, w + (-w) % 2, dout.ptr, pitch, w, h, 6, None)
    if f32:
        got = _download(dev, dout, (3, h, w), np.float32)
        assert (got == want.astype(np.float32) / np.float32(255.0)).all()
    else:
        assert (_download(dev, dout, (3, h, w)) == want).all()
    din.free(); dout.free()


@pytest.mark.parametrize("name,f32", [("P016ToBgrPlanar", 0), ("P016ToBgrFloatPlanar", 1)])
def test_p016_to_planar(dev, orc, name, f32):
    w, h = 64, 16
    src = synth_planes(orc, "p016le", w, h, seed=91)
    packed = orc.sws(src, w, h, "p016le", w, h, "bgra", colorspace=1)[0].reshape(h, w, 4)
    want = np.stack([packed[:, :, k] for k in range(3)])          # B, G, R
    din = _upload(dev, _semi(src, 2 * w))
    es = 4 if f32 else 1
    dout = DevBuf(dev, 3 * es * w * h)
    for _ in range(2):                                            # second call: cached context and scratch frame
        fn(dev, name, _CONV_ARGS)(din.ptr, 2 * w, dout.ptr, es * w, w, h, 1, None)
    if f32:
        assert (_download(dev, dout, (3, h, w), np.float32) == want.astype(np.float32) / np.float32(255.0)).all()
    else:
        assert (_download(dev, dout, (3, h, w)) == want).all()
    din.free(); dout.free()


@pytest.mark.parametrize("matrix,cs", [(1, 1), (6, 5)])
def test_bgra64_to_p016(dev, orc, matrix, cs):
    w, h = 64, 16
    src = synth_planes(orc, "bgra64le", w, h, seed=93)
    want = orc.sws(src, w, h, "bgra64le", w, h, "p016le", colorspace=cs)
    din = _upload(dev, src[0])
    dout = DevBuf(dev, 2 * w * h * 3 // 2)
    fn(dev, "Bgra64ToP016", _CONV_ARGS)(din.ptr, 8 * w, dout.ptr, 2 * w, w, h, matrix, None)
    got = _download(dev, dout, (h * 3 // 2, 2 * w))
    assert (got[:h] == want[0]).all() and (got[h:] == want[1]).all()
    din.free(); dout.free()


# ---- bit depth (BitDepth.cu:15-36) ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,off", [(4096 + 37, 0), (100, 0), (5000, 1), (15, 0)])
def test_convert_uint8_uint16(dev, orc, n, off):
    a = orc.lcg((n + 2,), 95)
    want16 = np.zeros(n, np.uint16)
    orc.L.orc_mt_u8_to_u16(ptr(a[off:]), ptr(want16), C.c_long(n))
    assert (want16 == a[off:off + n].astype(np.uint16) << 8).all()
    d8 = _upload(dev, a)
    d16 = DevBuf(dev, 2 * n + 2)
    dev.lib.gmat_memset(d16.ptr, 0xCD, 2 * n + 2)
    f = fn(dev, "ConvertUInt8ToUInt16", [C.c_void_p, C.c_void_p, C.c_int])
    f(d8.ptr + off, d16.ptr, n)
    got = _download(dev, d16, (n + 1,), np.uint16)
    assert (got[:n] == want16).all() and got[n] == 0xCDCD
    # and back: the high byte
    back = DevBuf(dev, n + 1)
    dev.lib.gmat_memset(back.ptr, 0xCD, n + 1)
    fn(dev, "ConvertUInt16ToUInt8", [C.c_void_p, C.c_void_p, C.c_int])(d16.ptr, back.ptr, n)
    got8 = _download(dev, back, (n + 1,))
    want8 = np.zeros(n, np.uint8)
    orc.L.orc_mt_u16_to_u8(ptr(want16), ptr(want8), C.c_long(n))
    assert (got8[:n] == want8).all() and (want8 == a[off:off + n]).all() and got8[n] == 0xCD
    d8.free(); d16.free(); back.free()


def test_convert_uint16_to_uint8_takes_the_high_byte(dev, orc):
    n = 4096 + 5
    a = orc.lcg((2 * n,), 97).view(np.uint16)
    d16 = _upload(dev, a)
    d8 = DevBuf(dev, n)
    fn(dev, "ConvertUInt16ToUInt8", [C.c_void_p, C.c_void_p, C.c_int])(d16.ptr, d8.ptr, n)
    assert (_download(dev, d8, (n,)) == (a >> 8).astype(np.uint8)).all()
    d16.free(); d8.free()


# ---- resize -------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,fmt,es", [("ScaleNv12", "nv12", 1), ("ScaleP016", "p016le", 2)])
def test_scale_bilinear_front_ends(dev, orc, name, fmt, es):
    sw, sh, dw, dh = 256, 64, 128, 32
    src = synth_planes(orc, fmt, sw, sh, seed=83)
    want = orc.sws(src, sw, sh, fmt, dw, dh, fmt, SWS["bilinear"])
    din = _upload(dev, _semi(src, es * sw))
    dout = DevBuf(dev, es * dw * dh * 3 // 2)
    for _ in range(2):                                       # second call uses the cached context
        fn(dev, name, _SCALE_ARGS)(din.ptr, es * sw, sw, sh, dout.ptr, es * dw, dw, dh)
    got = _download(dev, dout, (dh * 3 // 2, es * dw))
    assert (got[:dh] == want[0]).all() and (got[dh:] == want[1]).all()
    din.free(); dout.free()


def _bicubic_case(dev, orc, sw, sh, dw, dh, spitch, dpitch, seed=101):
    src = synth_planes(orc, "nv12", sw, sh, seed=seed)
    simg = _semi(src, spitch)
    want = np.full((dh * 3 // 2 + 1, dpitch), 0xCD, np.uint8)
    assert orc.L.orc_mt_scale_nv12_bicubic(ptr(simg), spitch, sw, sh, ptr(want), dpitch, dw, dh) == 0
    din = _upload(dev, simg)
    dout = DevBuf(dev, want.size)
    dev.lib.gmat_memset(dout.ptr, 0xCD, want.size)
    fn(dev, "ScaleNv12_Bicubic", _SCALE_ARGS)(din.ptr, spitch, sw, sh, dout.ptr, dpitch, dw, dh)
    got = _download(dev, dout, want.shape)
    din.free(); dout.free()
    return got, want


@pytest.mark.parametrize("sw,sh,dw,dh", [(256, 64, 128, 32),      # 2:1
                                         (200, 120, 150, 90),     # 4:3
                                         (64, 48, 160, 100),      # up-scale
                                         (320, 200, 66, 30),      # 4.8 : 1 and 6.7 : 1
                                         (512, 1000, 256, 20),    # 50 : 1 vertically: rows beyond one LDS tile
                                         (64, 64, 63, 61),        # odd destination: last column / row left unwritten
                                         (8, 8, 16, 16), (300, 40, 300, 40)])
def test_scale_nv12_bicubic_is_the_reference_kernel(dev, orc, sw, sh, dw, dh):
    """Resize_bicubic.cu:83-159 restated in oracle/orc_metrans.c (float 4 x 4, a = -0.5, clamp [2, n - 2], truncation).  Tolerance
    +-1 LSB (nvcc contracts a * b + c into fma, the restatement and this build do not); without contraction on either side the
    two agree bit for bit, which is asserted too.  Padding and the rows / columns the reference leaves unwritten stay untouched."""
    dpitch = (dw + 7) // 4 * 4
    got, want = _bicubic_case(dev, orc, sw, sh, dw, dh, sw + 8, dpitch)
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1
    assert (got == want).all()
    assert (got[:, 2 * (dw // 2):] == 0xCD).all() and (got[dh * 3 // 2:] == 0xCD).all()
    if dh % 2:
        assert (got[dh - 1] == 0xCD).all()


def test_scale_nv12_bicubic_known_answers(dev, orc):
    """independent of the C restatement: a constant frame stays constant (the weights sum to 1), and an identity-size call
    returns the source where the clamp is not active (x * 1.0 is an integer: weights 0, 1, 0, 0)"""
    sw = sh = 64
    got, _ = _bicubic_case(dev, orc, sw, sh, sw, sh, sw, sw, seed=103)
    src = synth_planes(orc, "nv12", sw, sh, seed=103)
    assert (got[2:sh - 2, 2:sw - 2] == src[0][2:sh - 2, 2:sw - 2]).all()
    assert (got[sh + 2:sh + sh // 2 - 2, 4:sw - 4] == src[1][2:sh // 2 - 2, 4:sw - 4]).all()
    # numpy restatement of one interior sample at 2:1 (float32, products and sums in the reference's order)
    got2, _ = _bicubic_case(dev, orc, 64, 64, 32, 32, 64, 32, seed=105)
    s = synth_planes(orc, "nv12", 64, 64, seed=105)[0].astype(np.float32)
    x, y = 10, 7
    fx, fy = np.float32(x * 2.0), np.float32(y * 2.0)            # integers: the taps are (0, 1, 0, 0)
    assert got2[y, x] == int(s[int(fy), int(fx)])


def test_scale_nv12_bicubic_mode_switch(dev, orc):
    """gmat_metrans_bicubic_mode(1): libswscale's SWS_BICUBIC of one context, rounds 1-3's meaning of the symbol"""
    sw, sh, dw, dh = 256, 64, 128, 32
    src = synth_planes(orc, "nv12", sw, sh, seed=83)
    want = orc.sws(src, sw, sh, "nv12", dw, dh, "nv12", SWS["bicubic"])
    dev.lib.gmat_metrans_bicubic_mode.restype = C.c_int
    dev.lib.gmat_metrans_bicubic_mode.argtypes = [C.c_int]
    assert dev.lib.gmat_metrans_bicubic_mode(1) == 0
    try:
        din = _upload(dev, _semi(src, sw))
        dout = DevBuf(dev, dw * dh * 3 // 2)
        fn(dev, "ScaleNv12_Bicubic", _SCALE_ARGS)(din.ptr, sw, sw, sh, dout.ptr, dw, dw, dh)
        got = _download(dev, dout, (dh * 3 // 2, dw))
        assert (got[:dh] == want[0]).all() and (got[dh:] == want[1]).all()
        din.free(); dout.free()
    finally:
        assert dev.lib.gmat_metrans_bicubic_mode(0) == 1
    assert dev.lib.gmat_metrans_bicubic_mode(7) < 0


@pytest.mark.gpu
def test_scale_nv12_bicubic_4k_to_1080p(orc):
    """BASELINE's geometry through the reference's symbol on the GPU"""
    import harness
    from gmat_amd.lib import load
    dev = harness.Dev(load(), "hip")
    got, want = _bicubic_case(dev, orc, 3840, 2160, 1920, 1080, 3840, 1920, seed=107)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
    assert (got == want).all()
