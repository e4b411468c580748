"""Python mirror of the reference's libgpuscale binding (metrans/python/swscale.py:11-23).

Same class name, constructor and method signature: ``SwscaleCuda(w, h).nv12_to_rgbpf32(in_nv12,
in_stride, out_rgbp, out_stride, stream=0)`` with raw device addresses, now backed by the HIP
library.  ``SwsContext`` is the general sws_getContext/sws_scale shape over device pointers.
"""
import ctypes as C

from .lib import load, planes, ints, GmatError, PIX_FMT, SWS

_lib = None


def _L():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


class SwscaleCuda:
    def __init__(self, w, h):
        self.ctx = _L().SwscaleCuda_Nv12ToRgbpf32_Init(C.c_int(w), C.c_int(h))
        if not self.ctx:
            raise GmatError("SwscaleCuda_Nv12ToRgbpf32_Init failed")
        self.w = w
        self.h = h

    def __del__(self):
        if getattr(self, "ctx", None):
            _L().SwscaleCuda_Nv12ToRgbpf32_Delete(C.c_void_p(self.ctx))
            self.ctx = None

    # in_nv12 and out_rgbp are device addresses (ints), tightly packed as the reference expects
    def nv12_to_rgbpf32(self, in_nv12, in_stride, out_rgbp, out_stride, stream=0):
        return _L().SwscaleCuda_Nv12ToRgbpf32_Convert(C.c_void_p(self.ctx), C.c_void_p(in_nv12), in_stride,
                                                      C.c_void_p(out_rgbp), C.c_int(out_stride), self.w, self.h,
                                                      C.c_void_p(stream))


class SwsContext:
    """sws_getContext / sws_scale / sws_setCudaStream / sws_freeContext_cuda over device pointers."""

    def __init__(self, src_w, src_h, src_fmt, dst_w, dst_h, dst_fmt, flags=SWS["bicubic"], lib=None):
        self._lib = lib or _L()
        fmt = lambda f: PIX_FMT[f] if isinstance(f, str) else int(f)
        self.src_h = src_h
        self.ctx = self._lib.gmat_sws_getContext(src_w, src_h, fmt(src_fmt), dst_w, dst_h, fmt(dst_fmt),
                                                 flags | SWS["hwaccel"], None)
        if not self.ctx:
            raise GmatError("gmat_sws_getContext failed (unsupported conversion?)")

    def set_stream(self, stream):
        self._lib.gmat_sws_setStream(self.ctx, C.c_void_p(stream))

    def scale(self, src_ptrs, src_strides, dst_ptrs, dst_strides):
        r = self._lib.gmat_sws_scale(self.ctx, planes(src_ptrs), ints(src_strides), 0, self.src_h,
                                     planes(dst_ptrs), ints(dst_strides))
        if r < 0:
            raise GmatError(f"gmat_sws_scale failed: {r}")
        return r

    def close(self):
        if self.ctx:
            self._lib.gmat_sws_freeContext(self.ctx)
            self.ctx = None

    __del__ = close
