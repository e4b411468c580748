"""Host <-> device frame pipeline: what hwupload -> filter -> hwdownload does in the
reference (vf_hwupload_cuda.c:123-150, hwcontext_cuda.c:221-279: one cuMemcpy2DAsync per plane on the NULL
stream from PAGEABLE memory), rebuilt so that copies overlap compute:

  * host frames live in PINNED staging rings (gmat_host_frame_alloc -> hipHostMalloc);
  * three HIP streams: upload, compute, download, chained with events per ring slot;
  * slot k of depth D is reused only after its download of D frames ago has completed.

Software decode/encode stay on the host (VCN is out of scope), so frames arrive as pinned host NV12.
"""
import ctypes as C

from .lib import GmatFrame, GmatError, PIX_FMT, SWS


class FramePipeline:
    def __init__(self, lib, src_w, src_h, src_fmt, dst_w, dst_h, dst_fmt, depth=4, flags=SWS["bicubic"], device=0):
        self.lib, self.depth = lib, depth
        self.src = (src_w, src_h, PIX_FMT[src_fmt])
        self.dst = (dst_w, dst_h, PIX_FMT[dst_fmt])
        self.ctx = lib.gmat_sws_getContext(src_w, src_h, self.src[2], dst_w, dst_h, self.dst[2], flags | SWS["hwaccel"], None)
        if not self.ctx:
            raise GmatError("gmat_sws_getContext failed")
        self.up, self.comp, self.down = (self._stream() for _ in range(3))
        lib.gmat_sws_setStream(self.ctx, self.comp)
        self.in_pool = lib.gmat_hwframe_ctx_create(device, self.src[2], src_w, src_h, depth)
        self.out_pool = lib.gmat_hwframe_ctx_create(device, self.dst[2], dst_w, dst_h, depth)
        self.slots = []
        for _ in range(depth):
            s = {"hin": GmatFrame(), "hout": GmatFrame(), "din": GmatFrame(), "dout": GmatFrame(),
                 "uploaded": self._event(), "computed": self._event(), "downloaded": self._event(), "busy": False}
            self._ck(lib.gmat_host_frame_alloc(C.byref(s["hin"]), self.src[2], src_w, src_h))
            self._ck(lib.gmat_host_frame_alloc(C.byref(s["hout"]), self.dst[2], dst_w, dst_h))
            self._ck(lib.gmat_hwframe_get_buffer(self.in_pool, C.byref(s["din"])))
            self._ck(lib.gmat_hwframe_get_buffer(self.out_pool, C.byref(s["dout"])))
            self.slots.append(s)
        self.n = 0

    def _ck(self, r):
        if r < 0:
            raise GmatError(f"gmat call failed: {r}")

    def _stream(self):
        h = C.c_void_p()
        self._ck(self.lib.gmat_stream_create(C.byref(h)))
        return h

    def _event(self):
        h = C.c_void_p()
        self._ck(self.lib.gmat_event_create(C.byref(h)))
        return h

    def host_input(self, k):
        """pinned host frame of ring slot k (the decoder writes here)"""
        return self.slots[k % self.depth]["hin"]

    def host_output(self, k):
        return self.slots[k % self.depth]["hout"]

    def submit(self):
        """Enqueue upload -> convert/scale -> download for the next slot; returns the slot index.  Blocks only when
        the slot's previous download has not finished."""
        lib = self.lib
        s = self.slots[self.n % self.depth]
        if s["busy"]:
            self._ck(lib.gmat_event_sync(s["downloaded"]))
        self._ck(lib.gmat_hwframe_transfer_data(C.byref(s["din"]), C.byref(s["hin"]), self.up))
        self._ck(lib.gmat_event_record(s["uploaded"], self.up))
        self._ck(lib.gmat_stream_wait_event(self.comp, s["uploaded"]))
        r = lib.gmat_sws_scale(self.ctx, C.cast(s["din"].data, C.POINTER(C.c_void_p)), C.cast(s["din"].linesize, C.POINTER(C.c_int)),
                               0, self.src[1], C.cast(s["dout"].data, C.POINTER(C.c_void_p)),
                               C.cast(s["dout"].linesize, C.POINTER(C.c_int)))
        self._ck(r)
        self._ck(lib.gmat_event_record(s["computed"], self.comp))
        self._ck(lib.gmat_stream_wait_event(self.down, s["computed"]))
        self._ck(lib.gmat_hwframe_transfer_data(C.byref(s["hout"]), C.byref(s["dout"]), self.down))
        self._ck(lib.gmat_event_record(s["downloaded"], self.down))
        s["busy"] = True
        k = self.n
        self.n += 1
        return k

    def wait(self, k):
        self._ck(self.lib.gmat_event_sync(self.slots[k % self.depth]["downloaded"]))

    def drain(self):
        for s in self.slots:
            if s["busy"]:
                self._ck(self.lib.gmat_event_sync(s["downloaded"]))

    def close(self):
        lib = self.lib
        self.drain()
        for s in self.slots:
            lib.gmat_host_frame_free(C.byref(s["hin"])); lib.gmat_host_frame_free(C.byref(s["hout"]))
            for e in ("uploaded", "computed", "downloaded"):
                lib.gmat_event_destroy(s[e])
        lib.gmat_sws_freeContext(self.ctx)
        for st in (self.up, self.comp, self.down):
            lib.gmat_stream_destroy(st)
        # device frames are plain structs here: hand their blocks back by freeing the pools
        lib.gmat_hwframe_ctx_free(self.in_pool); lib.gmat_hwframe_ctx_free(self.out_pool)
