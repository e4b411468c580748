"""Shared test plumbing: oracle binding, device-buffer helpers, synthetic frames."""
import os
import ctypes as C

import numpy as np
import pytest

from gmat_amd.lib import planes, ints, PIX_FMT, SWS  # noqa: F401

u8p = C.POINTER(C.c_uint8)


def load_oracle(path):
    L = C.CDLL(path)
    L.orc_sws_create.restype = C.c_void_p
    L.orc_sws_create.argtypes = [C.c_int] * 7 + [C.c_void_p]
    L.orc_sws_scale.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_void_p),
                                C.POINTER(C.c_int)]
    L.orc_sws_scale_rows.argtypes = L.orc_sws_scale.argtypes + [C.c_int, C.c_int]
    L.orc_sws_free.argtypes = [C.c_void_p]
    L.orc_sws_filter.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(C.c_int16)), C.POINTER(C.POINTER(C.c_int32)),
                                 C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_sws_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 5
    L.orc_yuv2rgb_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_yuv2rgb_selfcheck.restype = C.c_long
    L.orc_yuv2rgb_selfcheck.argtypes = [C.c_void_p]
    L.orc_yuv2rgb_frame.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_nv12_to_rgbpf32.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p, C.c_int,
                                      C.c_int, C.c_int]
    L.orc_init_filter.argtypes = [C.POINTER(C.POINTER(C.c_int16)), C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int),
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_transpose.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    for n in ("orc_hflip", "orc_vflip"):
        getattr(L, n).argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_crop.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_int] * 5
    L.orc_conv3x3.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.POINTER(C.c_int), C.c_float, C.c_float]
    L.orc_rotate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_double, C.c_int, C.c_void_p]
    L.orc_rotate2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                              C.c_double, C.c_double, C.c_void_p]
    L.orc_median.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_rotate_sincos.argtypes = [C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_yuv420_to_p01x.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                     C.c_int, C.c_int, C.c_int]
    L.orc_rgb24_swap_rb.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.orc_plane_copy_up.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_plane_copy_up.restype = None
    L.orc_gauss_blur.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_double, C.c_double, C.c_int]
    L.orc_median3x3.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_sws_set_colorspace.argtypes = [C.c_void_p, C.c_int]
    L.orc_rgb_repack.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_fill_lcg.argtypes = [C.c_void_p, C.c_long, C.c_uint32]
    L.orc_mt_scale_nv12_bicubic.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.orc_mt_u8_to_u16.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    L.orc_mt_u8_to_u16.restype = None
    L.orc_mt_u16_to_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    L.orc_mt_u16_to_u8.restype = None
    return Oracle(L)


def is_generic(kernel):
    """a kernel of the any-geometry tier behind the exact-ratio walkers: the polyphase band walker of round 3
    (scale_yuvg_kernel: dword-aligned 8-bit 4:2:0 -> packed RGB / 4:2:0 of the same chroma layout, filters up to 20 x 18 taps) or
    the tiled plane scaler of round 1 behind it (scale_yuv_kernel<...>).  WHICH of the two a context gets is asserted by
    tests/test_parity_generic_walker.py clause by clause; the per-ratio test files only need "not a specialised walker"."""
    return kernel in ("scale_yuvg_kernel", "scale_yuvg_blk_kernel", QUAD, LINES, LINES16, RGB2P, "scale19_kernel", "scale19_unit_kernel", "scale19_unit64_kernel", "unit_rgb_kernel") + WALK16 or kernel.startswith("scale_yuv_kernel")      # (_blk_: the walker's one-frame form, round 4; QUAD: the quad-lane walker of up-scales, round 4)


@pytest.fixture(autouse=True)
def ratio_kernels_keep_single_frames(monkeypatch):
    """round 4: a call of one frame (4:1: up to three) takes the band walker's block-cooperative form IN FRONT of the 4:1 walkers and of the
    3:1 / 3:2 -> RGB walkers (gsws.cpp kPlaneKernels; tests/test_parity_generic_walker.py::test_block_form_in_front_of_the_ratio_walkers holds
    the rule).  The per-ratio parity files import this fixture to keep every launch size on the kernel they are about."""
    monkeypatch.setenv("GMAT_BLOCK_FIRST", "0")


QUAD = "scale_yuvu_kernel"
RGB2P = "scale_yuvg_rgb2p_blk_kernel"                     # a packed RGB source into an 8-bit 4:2:0 frame, block-cooperative and fused (k_scale_yuvg16.hip, round 5): up-scales, launches of four frames or more
WALK16 = ("scale_yuvg16_kernel", "scale_yuvg16_blk_kernel", "scale_yuvu16_kernel")   # the band walker over 16-bit samples (k_scale_yuvg16.hip, round 5): P010 / P016 / planar 10- and 16-bit 4:2:0 sources; the quad-lane walker of their up-scales (k_scale_yuvu16.hip)
LINES16 = "scale_yuvl_h16_kernel+scale_yuvl_v_kernel"  # ... its pass H for 16-bit samples (P010 / P016 / planar 10 / 16 bit sources)
LINES = "scale_yuvl_h_kernel+scale_yuvl_v_kernel"     # the lines form (k_scale_yuvl.hip, round 4): what no walker takes, from 2 : 1 on the horizontal axis


def quad_takes(sw, sh, sf, df, dw, dh, flags="bicubic"):
    """the host rule of yuvu_prepare / yuvu_eligible (k_scale_yuvu.hip, the quad-lane walker) restated for dword-aligned frames: the band
    walker's format rule (8-bit 4:2:0 in; packed 8-bit RGB of even width or 4:2:0 of the SAME chroma layout out; whole dwords in every source
    row; at least 16 x 8 on both sides), an UP-scale on the vertical axis (the default of GMAT_QUAD_WALKER) and horizontal filters of at most
    8 taps: any up-scale, a down-scale up to about 1.7 : 1 with the four-tap algorithms.  The algorithms whose windows grow past that on an
    up-scale (sinc, gauss, spline: 8 - 20 taps) stay where they were.  No limit on the up-scale factor: the vertical filter is a gather."""
    rgb = df in ("rgb24", "bgr24", "rgba", "bgra")
    if os.environ.get("GMAT_SCALE_NO_QUAD_WALKER", "0") not in ("", "0") or os.environ.get("GMAT_QUAD_WALKER", "1") == "0":
        return False
    if sf not in ("nv12", "yuv420p") or not (rgb or df == sf):
        return False
    if sw % 4 or (sf == "yuv420p" and ((sw + 1) // 2) % 4) or (sf == "nv12" and (2 * ((sw + 1) // 2)) % 4):
        return False
    if rgb and dw % 2:
        return False
    if not (dw >= 16 and dh >= 8 and sw >= 16 and sh >= 8):
        return False
    if flags not in ("bicubic", "bilinear", "fast_bilinear", "lanczos", "area", "point"):
        return False
    return dh > sh and sw / dw <= (1.7 if flags != "lanczos" else 1.2)


def walker_takes(sw, sh, sf, df, dw, dh):
    """the host rule of yuvg_prepare / yuvg_eligible restated for dword-aligned frames and bicubic-sized filters: 8-bit 4:2:0 in,
    packed 8-bit RGB (even width: an odd one forces libswscale's full-chroma output) or 4:2:0 of the SAME chroma layout out, whole
    dwords in every source row, at least 16 x 8 on both sides, and a ratio on both axes between about 6.1 : 1 (round 4: filters of at most 26 taps on 13 coefficient pairs; round 3 ended at 4.7 : 1,
    20 taps) and 1 : 1.9 (output rows open at once: up to 15 in the plane jobs of a 4:2:0 destination, up to 22 in the four-pair
    instances of an RGB destination, whose chroma plane is up-scaled twice as far as its luma).  Callers keep away from the ends."""
    rgb = df in ("rgb24", "bgr24", "rgba", "bgra")
    if sf not in ("nv12", "yuv420p") or not (rgb or df == sf):
        return False
    if sw % 4 or (sf == "yuv420p" and ((sw + 1) // 2) % 4) or (sf == "nv12" and (2 * ((sw + 1) // 2)) % 4):
        return False
    if rgb and dw % 2:
        return False
    if not (dw >= 16 and dh >= 8 and sw >= 16 and sh >= 8):
        return False
    return 0.53 <= sw / dw <= 6.1 and 0.53 <= sh / dh <= 6.1


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    YUV2RGB_SIZE = 64 * 1024      # >= sizeof(OrcYuv2Rgb)

    def __init__(self, L):
        self.L = L
        self._y2r = {}

    def y2r(self, colorspace=5, full_range=0):
        key = (colorspace, full_range)
        if key not in self._y2r:
            buf = C.create_string_buffer(self.YUV2RGB_SIZE)
            self.L.orc_yuv2rgb_init(buf, colorspace, full_range, 0, 1 << 16, 1 << 16)
            self._y2r[key] = buf
        return self._y2r[key]

    def lcg(self, shape, seed):
        a = np.empty(shape, np.uint8)
        self.L.orc_fill_lcg(ptr(a), a.size, seed)
        return a

    def yuv2rgb(self, src_planes, w, h, src_fmt, dst_fmt, colorspace=5, full_range=0):
        bpp = 4 if dst_fmt in ("rgba", "bgra") else 3
        out = np.zeros((h, w * bpp), np.uint8)
        r = self.L.orc_yuv2rgb_frame(self.y2r(colorspace, full_range), planes([p.ctypes.data for p in src_planes]),
                                     ints([p.strides[0] for p in src_planes]), ptr(out), out.strides[0], w, h,
                                     PIX_FMT[src_fmt], PIX_FMT[dst_fmt])
        assert r == 0
        return out

    def nv12_to_rgbpf32(self, src_planes, w, h):
        out = np.zeros((3, h, w), np.float32)
        r = self.L.orc_nv12_to_rgbpf32(self.y2r(), planes([p.ctypes.data for p in src_planes]),
                                       ints([p.strides[0] for p in src_planes]), ptr(out), 4 * w, w, h)
        assert r == 0
        return out

    def sws(self, src_planes, sw, sh, src_fmt, dw, dh, dst_fmt, flags=SWS["bicubic"], colorspace=None):
        c = self.L.orc_sws_create(sw, sh, PIX_FMT[src_fmt], dw, dh, PIX_FMT[dst_fmt], flags, None)
        assert c, "oracle refused the conversion"
        try:
            if colorspace is not None:
                assert self.L.orc_sws_set_colorspace(c, colorspace) == 0
            outs = alloc_planes(dst_fmt, dw, dh, tight=True)
            r = self.L.orc_sws_scale(c, planes([p.ctypes.data for p in src_planes]),
                                     ints([p.strides[0] for p in src_planes]),
                                     planes([p.ctypes.data for p in outs]), ints([p.strides[0] for p in outs]))
            assert r == dh
            return outs
        finally:
            self.L.orc_sws_free(c)

    def sws_filters(self, sw, sh, src_fmt, dw, dh, dst_fmt, flags=SWS["bicubic"]):
        c = self.L.orc_sws_create(sw, sh, PIX_FMT[src_fmt], dw, dh, PIX_FMT[dst_fmt], flags, None)
        assert c
        res = []
        for which in range(4):
            coef = C.POINTER(C.c_int16)(); pos = C.POINTER(C.c_int32)(); size = C.c_int(); cnt = C.c_int()
            self.L.orc_sws_filter(c, which, C.byref(coef), C.byref(pos), C.byref(size), C.byref(cnt))
            n, s = cnt.value, size.value
            res.append((np.ctypeslib.as_array(coef, (n * s,)).reshape(n, s).copy(),
                        np.ctypeslib.as_array(pos, (n,)).copy()))
        self.L.orc_sws_free(c)
        return res

    def chained(self, src_planes, sw, sh, src_fmt, dw, dh, dst_fmt, flags=SWS["bicubic"]):
        """The scaled YUV->RGB contract: nearest-chroma convert at source size, then RGB24->dst scale."""
        rgb = self.yuv2rgb(src_planes, sw, sh, src_fmt, "rgb24")
        return self.sws([rgb], sw, sh, "rgb24", dw, dh, dst_fmt, flags)


def plane_shapes(fmt, w, h):
    if fmt in ("rgb24", "bgr24"):
        return [(h, 3 * w)]
    if fmt in ("rgba", "bgra", "rgb0", "bgr0"):
        return [(h, 4 * w)]
    if fmt in ("rgba64le", "bgra64le"):
        return [(h, 8 * w)]
    if fmt == "nv12":
        return [(h, w), ((h + 1) // 2, 2 * ((w + 1) // 2))]
    if fmt == "yuv420p":
        return [(h, w), ((h + 1) // 2, (w + 1) // 2), ((h + 1) // 2, (w + 1) // 2)]
    if fmt == "yuv444p":
        return [(h, w)] * 3
    if fmt == "yuv444p16le":
        return [(h, 2 * w)] * 3
    if fmt in ("yuv420p16le", "yuv420p10le"):
        return [(h, 2 * w)] + [((h + 1) // 2, 2 * ((w + 1) // 2))] * 2
    if fmt == "rgbpf32le":
        return [(h, 4 * w)] * 3
    if fmt in ("p010le", "p016le"):
        return [(h, 2 * w), ((h + 1) // 2, 4 * ((w + 1) // 2))]
    raise ValueError(fmt)


def alloc_planes(fmt, w, h, tight=True, align=64, fill=0):
    out = []
    for (rows, rb) in plane_shapes(fmt, w, h):
        stride = rb if tight else (rb + align - 1) // align * align
        buf = np.full((rows, stride), fill, np.uint8)
        out.append(buf[:, :rb] if not tight else buf)
    return out


def synth_planes(orc, fmt, w, h, seed, tight=True, align=64):
    out = alloc_planes(fmt, w, h, tight, align)
    for i, p in enumerate(out):
        p[...] = orc.lcg(p.shape, seed + 17 * i)
    return out


class DevBuf:
    def __init__(self, dev, nbytes):
        self.dev, self.nbytes = dev, nbytes
        p = C.c_void_p()
        r = dev.lib.gmat_malloc(C.byref(p), nbytes)
        assert r == 0 and p.value
        self.ptr = p.value

    def free(self):
        if self.ptr:
            self.dev.lib.gmat_free(self.ptr)
            self.ptr = None


class DevPlane:
    """A strided 2-D byte plane in device memory."""

    def __init__(self, dev, rows, row_bytes, stride=None, fill=0xCD):
        self.rows, self.row_bytes = rows, row_bytes
        self.stride = stride or row_bytes
        self.buf = DevBuf(dev, self.stride * rows)
        dev.lib.gmat_memset(self.buf.ptr, fill, self.stride * rows)
        self.dev = dev

    @property
    def ptr(self):
        return self.buf.ptr

    def upload(self, arr):
        assert arr.shape == (self.rows, self.row_bytes)
        host = np.full((self.rows, self.stride), 0xCD, np.uint8)
        host[:, :self.row_bytes] = arr
        assert self.dev.lib.gmat_memcpy_h2d(self.ptr, ptr(host), host.size) == 0
        return self

    def download(self, with_padding=False):
        host = np.empty((self.rows, self.stride), np.uint8)
        self.dev.lib.gmat_device_sync()
        assert self.dev.lib.gmat_memcpy_d2h(ptr(host), self.ptr, host.size) == 0
        return host if with_padding else host[:, :self.row_bytes].copy()

    def free(self):
        self.buf.free()


class Dev:
    def __init__(self, lib, kind):
        self.lib, self.kind = lib, kind

    def planes_like(self, fmt, w, h, stride_align=1, extra=0):
        out = []
        for (rows, rb) in plane_shapes(fmt, w, h):
            stride = (rb + extra + stride_align - 1) // stride_align * stride_align
            out.append(DevPlane(self, rows, rb, stride))
        return out

    def upload_planes(self, arrays, stride_align=1, extra=0):
        out = []
        for a in arrays:
            rows, rb = a.shape
            stride = (rb + extra + stride_align - 1) // stride_align * stride_align
            out.append(DevPlane(self, rows, rb, stride).upload(np.ascontiguousarray(a)))
        return out

    def sws(self, src_dev, sw, sh, src_fmt, dw, dh, dst_fmt, flags=SWS["bicubic"], fused=None, dst_align=1, dst_extra=0,
            colorspace=None):
        lib = self.lib
        c = lib.gmat_sws_getContext(sw, sh, PIX_FMT[src_fmt], dw, dh, PIX_FMT[dst_fmt], flags | SWS["hwaccel"], None)
        assert c, "gmat_sws_getContext failed"
        try:
            if fused is not None:
                lib.gmat_sws_setFused(c, int(fused))
            if colorspace is not None:
                lib.gmat_sws_setColorspace(c, colorspace[0], colorspace[1])
            dst = self.planes_like(dst_fmt, dw, dh, dst_align, dst_extra)
            r = lib.gmat_sws_scale(c, planes([p.ptr for p in src_dev]), ints([p.stride for p in src_dev]), 0, sh,
                                   planes([p.ptr for p in dst]), ints([p.stride for p in dst]))
            assert r == dh, f"gmat_sws_scale returned {r}"
            kernel = lib.gmat_sws_lastKernel(c).decode()
            outs = [p.download() for p in dst]
            pads = [p.download(with_padding=True)[:, p.row_bytes:] for p in dst]
            for p in dst:
                p.free()
            return outs, pads, kernel
        finally:
            lib.gmat_sws_freeContext(c)
