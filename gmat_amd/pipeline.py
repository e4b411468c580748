"""Host <-> device frame pipeline: ctypes mirror of the C ABI's gmat_pipeline_* (include/gmat_hip.h
section 5, gmat_amd/csrc/gpipeline.cpp).  The ring, the three streams and the per-slot events live in C — a libavfilter
caller (vf_hwupload_cuda.c:123-150) uses the same entry points; this class only adds numpy views for tests and the bench.
"""
import ctypes as C

from .lib import GmatFrame, GmatError, PIX_FMT, SWS


class FramePipeline:
    def __init__(self, lib, src_w, src_h, src_fmt, dst_w, dst_h, dst_fmt, depth=4, flags=SWS["bicubic"], device=0):
        self.lib, self.depth = lib, depth
        self.src = (src_w, src_h, PIX_FMT[src_fmt])
        self.dst = (dst_w, dst_h, PIX_FMT[dst_fmt])
        self.p = lib.gmat_pipeline_create(device, src_w, src_h, self.src[2], dst_w, dst_h, self.dst[2], flags, depth)
        if not self.p:
            raise GmatError("gmat_pipeline_create failed")
        self.n = 0

    def _ck(self, r):
        if r < 0:
            raise GmatError(f"gmat call failed: {r}")
        return r

    def host_input(self, k):
        """pinned host frame of ring slot k % depth (the decoder writes here)"""
        f = GmatFrame()
        self._ck(self.lib.gmat_pipeline_host_input(self.p, k, C.byref(f)))
        return f

    def host_output(self, k):
        f = GmatFrame()
        self._ck(self.lib.gmat_pipeline_host_output(self.p, k, C.byref(f)))
        return f

    def submit(self):
        """Enqueue upload -> convert/scale -> download for the next slot; returns the frame's sequence number.  Blocks
        only when the slot's previous download has not finished."""
        k = self._ck(self.lib.gmat_pipeline_submit(self.p))
        self.n = k + 1
        return k

    def wait(self, k):
        self._ck(self.lib.gmat_pipeline_wait(self.p, k))

    def drain(self):
        self._ck(self.lib.gmat_pipeline_drain(self.p))

    def close(self):
        if self.p:
            self.lib.gmat_pipeline_free(self.p)
            self.p = None
