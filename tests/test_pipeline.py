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
