"""Pinned-host upload -> scale -> download pipeline produces the oracle's bytes for every frame."""
import ctypes as C

import numpy as np

from harness import synth_planes
from gmat_amd.pipeline import FramePipeline


def _view(frame, plane, rows, row_bytes):
    return np.ctypeslib.as_array(C.cast(frame.data[plane], C.POINTER(C.c_uint8)), (rows, frame.linesize[plane]))[:, :row_bytes]


def test_overlapped_pipeline_matches_oracle(dev, orc):
    sw, sh, dw, dh = 256, 64, 128, 32
    p = FramePipeline(dev.lib, sw, sh, "nv12", dw, dh, "rgb24", depth=3)
    frames = [synth_planes(orc, "nv12", sw, sh, seed=70 + i) for i in range(7)]
    wants = [orc.sws(f, sw, sh, "nv12", dw, dh, "rgb24")[0] for f in frames]
    pending = []
    for i, f in enumerate(frames):
        if len(pending) == p.depth:                      # the ring is full: collect the oldest result first
            k = pending.pop(0)
            p.wait(k)
            assert (_view(p.host_output(k), 0, dh, dw * 3) == wants[k]).all()
        hin = p.host_input(i)
        _view(hin, 0, sh, sw)[...] = f[0]
        _view(hin, 1, sh // 2, sw)[...] = f[1]
        pending.append(p.submit())
    for k in pending:
        p.wait(k)
        assert (_view(p.host_output(k), 0, dh, dw * 3) == wants[k]).all()
    p.close()


def test_frames_outlive_their_context(dev):
    """AVBufferRef semantics (hwcontext.c:236-258): freeing a frames context while frames are out must not pull the pool
    from under them; the blocks are released when the frames come back, and a freed context hands out nothing."""
    from gmat_amd.lib import GmatFrame, PIX_FMT
    lib = dev.lib
    fc = lib.gmat_hwframe_ctx_create(0, PIX_FMT["rgb24"], 64, 16, 1)
    assert fc
    a, b = GmatFrame(), GmatFrame()
    assert lib.gmat_hwframe_get_buffer(fc, C.byref(a)) == 0
    assert lib.gmat_hwframe_get_buffer(fc, C.byref(b)) == 0
    pa = lib.gmat_frame_alloc()
    assert lib.gmat_hwframe_get_buffer(fc, pa) == 0
    lib.gmat_frame_unref(C.byref(a))                    # back to the (open) pool
    assert not a.data[0] and not a.hw_frames_ctx
    lib.gmat_hwframe_ctx_free(fc)                       # b and *pa are still out
    assert lib.gmat_memset(b.data[0], 7, 64 * 3) == 0   # their memory is still valid
    host = np.zeros(192, np.uint8)
    assert lib.gmat_memcpy_d2h(host.ctypes.data, b.data[0], 192) == 0 and (host == 7).all()
    c = GmatFrame()
    assert lib.gmat_hwframe_get_buffer(fc, C.byref(c)) < 0      # closed
    lib.gmat_frame_unref(C.byref(b))
    ppa = C.pointer(pa)
    lib.gmat_frame_free(ppa)                            # last one out: the context goes with it
    lib.gmat_frame_unref(C.byref(b))                    # a second unref of a cleared struct is a no-op


def test_filter_output_frames_survive_reconfiguration(dev, orc):
    """a downstream holder of an output frame outlives gmat_filter_config_props being re-run and gmat_filter_free"""
    from gmat_amd.lib import GmatFrame, PIX_FMT
    lib = dev.lib
    w, h = 64, 16
    inp = lib.gmat_hwframe_ctx_create(0, PIX_FMT["rgb24"], w, h, 2)
    f = lib.gmat_filter_alloc(b"flip_hip")
    assert f and lib.gmat_filter_set_option(f, b"code", b"1") == 0 and lib.gmat_filter_init(f) == 0
    assert lib.gmat_filter_config_props(f, inp, None) == 0
    src = orc.lcg((h, w * 3), 5)
    fin = lib.gmat_frame_alloc()
    assert lib.gmat_hwframe_get_buffer(inp, fin) == 0
    host = np.zeros((h, fin.contents.linesize[0]), np.uint8)
    host[:, :w * 3] = src
    assert lib.gmat_memcpy_h2d(fin.contents.data[0], host.ctypes.data, host.size) == 0
    out = C.POINTER(GmatFrame)()
    assert lib.gmat_filter_frame(f, fin, C.byref(out)) == 0
    lib.gmat_device_sync()
    assert lib.gmat_filter_config_props(f, inp, None) == 0       # frees and re-creates the output pool
    lib.gmat_filter_free(f)
    back = np.zeros((h, out.contents.linesize[0]), np.uint8)
    assert lib.gmat_memcpy_d2h(back.ctypes.data, out.contents.data[0], back.size) == 0
    want = src.reshape(h, w, 3)[:, ::-1, :].reshape(h, w * 3)
    assert (back[:, :w * 3] == want).all()
    lib.gmat_frame_free(C.byref(out))
    lib.gmat_hwframe_ctx_free(inp)
