"""The AVFilter glue (integration/vf_gmat_hip.c) RUNS: tests/c/filter_caller.c implements the handful of libavfilter / libavutil
functions it calls (over integration/compat's declarations and the library's device memory) and drives one filter instance as
libavfilter does — AVOption defaults and a "k=v:k=v" string, .init, the output pad's .config_props, frames through the input pad's
.filter_frame or through .activate until EOF, .uninit.  Bytes are compared with the oracle's CPU filters.  Rounds 1-3 only type-checked
this file; ADVICE round 3 (high) was a defect only a run could show: rotate_hip's config_props swapped width and height for angle=90
even with a shift, while the launch took the arbitrary-angle walk at the input's size (an out-of-bounds device write on a portrait frame)."""
import ctypes as C
import math
import os
import struct
import subprocess

import numpy as np
import pytest

from harness import PIX_FMT, SWS, plane_shapes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def caller(dev, tmp_path_factory):
    if dev.kind == "hip":
        libdir, libname = os.path.join(ROOT, "gmat_amd", "lib"), "gmat_hip"
    else:
        libdir, libname = os.path.join(ROOT, "tests", "hipemu", "build"), "gmat_hip_emu"
    exe = str(tmp_path_factory.mktemp("glue") / ("filter_caller_" + libname))
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Werror=implicit-function-declaration", "-D_DEFAULT_SOURCE",
           "-I" + os.path.join(ROOT, "integration", "compat"), "-I" + os.path.join(ROOT, "integration", "compat", "libavfilter"),
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "filter_caller.c"),
           os.path.join(ROOT, "integration", "vf_gmat_hip.c"), "-L" + libdir, "-l" + libname, "-Wl,-rpath," + libdir, "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def run(caller, orc, filt, opts, fmt, w, h, n=1, seed=5):
    r = subprocess.run([caller, filt, opts or "-", fmt, str(w), str(h), str(n), str(seed)], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    srcs = [[orc.lcg(shape, seed + 17 * i + 1000 * f) for i, shape in enumerate(plane_shapes(fmt, w, h))] for f in range(n)]
    buf, off, outs = r.stdout, 0, []
    names = {v: k for k, v in PIX_FMT.items()}
    for f in range(n):
        ow, oh, ofmt, pts = struct.unpack_from("<4i", buf, off)
        off += 16
        assert pts == 1000 + f                                       # props copied, frames in order
        planes = []
        for rows, rb in plane_shapes(names[ofmt], ow, oh):
            planes.append(np.frombuffer(buf, np.uint8, rows * rb, off).reshape(rows, rb))
            off += rows * rb
        outs.append((ow, oh, names[ofmt], planes))
    assert off == len(buf)
    return srcs, outs


def test_flip_and_transpose_through_the_glue(caller, orc):
    w, h = 64, 16
    srcs, outs = run(caller, orc, "flip_hip", "code=1", "rgb24", w, h, n=2)
    for s, (ow, oh, fmt, p) in zip(srcs, outs):
        want = np.zeros_like(s[0])
        orc.L.orc_hflip(s[0].ctypes.data, s[0].strides[0], want.ctypes.data, want.strides[0], w, h, 3)
        assert (ow, oh, fmt) == (w, h, "rgb24") and (p[0] == want).all()
    srcs, outs = run(caller, orc, "flip_hip", "code=0:batch=2", "nv12", w, h, n=3)      # three frames through a queue of two + the EOF flush
    for s, (ow, oh, fmt, p) in zip(srcs, outs):
        assert fmt == "nv12" and (p[0] == s[0][::-1]).all() and (p[1] == s[1][::-1]).all()
    srcs, outs = run(caller, orc, "transpose_hip", "dir=1", "rgb24", w, h)
    want = np.zeros((w, h * 3), np.uint8)
    orc.L.orc_transpose(srcs[0][0].ctypes.data, srcs[0][0].strides[0], want.ctypes.data, want.strides[0], w, h, 3, 1)
    assert outs[0][:2] == (h, w) and (outs[0][3][0] == want).all()


@pytest.mark.parametrize("w,h", [(40, 96), (96, 40)])
def test_rotate_quarter_turn_with_and_without_a_shift_through_the_glue(caller, orc, w, h):
    """ONE predicate for config_props and the launch — a quarter turn swaps the output's width and height, the
    same angle with a shift keeps the input's size and takes the arbitrary-angle walk; portrait and landscape"""
    srcs, outs = run(caller, orc, "rotate_hip", "angle=90", "rgb24", w, h)
    want = np.zeros((w, h * 3), np.uint8)
    orc.L.orc_transpose(srcs[0][0].ctypes.data, srcs[0][0].strides[0], want.ctypes.data, want.strides[0], w, h, 3, 1)
    assert outs[0][:2] == (h, w) and (outs[0][3][0] == want).all()
    srcs, outs = run(caller, orc, "rotate_hip", "angle=90:interp=nearest:shift_x=1:shift_y=0", "rgb24", w, h)
    assert outs[0][:2] == (w, h)
    a = math.radians(90)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    tx, ty = 1 - cx + (math.cos(a) * cx - math.sin(a) * cy), 0 - cy + (math.sin(a) * cx + math.cos(a) * cy)
    src = srcs[0][0]
    want = np.zeros_like(src)
    fill = (C.c_uint8 * 4)(0, 0, 0, 255)
    orc.L.orc_rotate2(src.ctypes.data, src.strides[0], want.ctypes.data, want.strides[0], w, h, w, h, 3, a, 0, tx, ty, fill)
    assert (outs[0][3][0] == want).all()


def test_rotate_any_angle_crop_and_smooth_through_the_glue(caller, orc):
    w, h = 96, 40
    srcs, outs = run(caller, orc, "rotate_hip", "angle=17:interp=cubic:batch=2", "rgb24", w, h, n=2)
    fill = (C.c_uint8 * 4)(0, 0, 0, 255)
    for s, o in zip(srcs, outs):
        want = np.zeros_like(s[0])
        orc.L.orc_rotate2(s[0].ctypes.data, s[0].strides[0], want.ctypes.data, want.strides[0], w, h, w, h, 3, math.radians(17), 2, 0.0, 0.0, fill)
        assert (o[3][0] == want).all()
    srcs, outs = run(caller, orc, "crop_hip", "w=32:h=8:x=4:y=2", "nv12", 64, 16)
    assert outs[0][:3] == (32, 8, "nv12")
    assert (outs[0][3][0] == srcs[0][0][2:10, 4:36]).all() and (outs[0][3][1] == srcs[0][1][1:5, 4:36]).all()
    srcs, outs = run(caller, orc, "smooth_hip", None, "rgb24", w, h)
    m = (C.c_int * 9)(1, 2, 1, 2, 4, 2, 1, 2, 1)
    want = np.zeros_like(srcs[0][0])
    orc.L.orc_conv3x3(srcs[0][0].ctypes.data, srcs[0][0].strides[0], want.ctypes.data, want.strides[0], w, h, 3, m, 1 / 16, 0.0)
    assert (outs[0][3][0] == want).all()
    srcs, outs = run(caller, orc, "smooth_hip", "type=median", "rgb24", w, h)
    want = np.zeros_like(srcs[0][0])
    orc.L.orc_median3x3(srcs[0][0].ctypes.data, srcs[0][0].strides[0], want.ctypes.data, want.strides[0], w, h, 3)
    assert (outs[0][3][0] == want).all()


@pytest.mark.parametrize("batch", [1, 2])
def test_scale_and_format_through_the_glue(caller, orc, batch):
    """scale_hip / format_hip over libgpuscale, per frame and through the activate() queue (one launch per `batch` frames + the EOF flush)"""
    sw, sh = 128, 64
    srcs, outs = run(caller, orc, "scale_hip", "w=iw/2:h=ih/2:format=rgb24:batch=%d" % batch, "nv12", sw, sh, n=3)
    for s, (ow, oh, fmt, p) in zip(srcs, outs):
        assert (ow, oh, fmt) == (64, 32, "rgb24")
        assert (p[0] == orc.sws(s, sw, sh, "nv12", 64, 32, "rgb24", SWS["bicubic"])[0]).all()
    srcs, outs = run(caller, orc, "scale_hip", "w=80:h=36:interp_algo=lanczos", "nv12", sw, sh)
    for a, b in zip(outs[0][3], orc.sws(srcs[0], sw, sh, "nv12", 80, 36, "nv12", SWS["lanczos"])):
        assert (a == b).all()
    srcs, outs = run(caller, orc, "format_hip", "pix_fmt=rgba:batch=%d" % batch, "yuv420p", 64, 32, n=2)
    for s, (ow, oh, fmt, p) in zip(srcs, outs):
        assert fmt == "rgba" and (p[0] == orc.yuv2rgb(s, 64, 32, "yuv420p", "rgba")).all()


def test_glue_refuses_what_the_options_refuse(caller, orc):
    for filt, opts in (("rotate_hip", "interp=bogus"), ("crop_hip", "w=0:h=0"), ("smooth_hip", "kw=4"), ("crop_hip", "w=100:h=8")):
        r = subprocess.run([caller, filt, opts, "rgb24", "64", "16", "1", "5"], capture_output=True, timeout=120)
        assert r.returncode != 0, (filt, opts)
