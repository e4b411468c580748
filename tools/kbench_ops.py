#!/usr/bin/env python3
"""Per-op timing table on one GPU (HIP events, frames rotating through a set larger than L2/MALL).
usage: python tools/kbench_ops.py [reps] [filter-substring]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gmat_amd
from gmat_amd.lib import PIX_FMT, SWS, planes, ints

lib = gmat_amd.load(os.environ["KBENCH_LIB"]) if os.environ.get("KBENCH_LIB") else gmat_amd.load()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""
NF = int(os.environ.get('KBENCH_NF', '16'))
stream = C.c_void_p(); lib.gmat_stream_create(C.byref(stream))


def timeit(fn):
    t = C.c_void_p(); lib.gmat_timer_create(C.byref(t))
    for i in range(8): fn(i % NF)
    lib.gmat_stream_sync(stream)
    best = 1e9
    for _ in range(3):
        lib.gmat_timer_begin(t, stream)
        for i in range(REPS): fn(i % NF)
        lib.gmat_timer_end(t, stream)
        ms = C.c_float(); lib.gmat_timer_elapsed_ms(t, C.byref(ms))
        best = min(best, ms.value / REPS * 1e3)
    lib.gmat_timer_destroy(t)
    return best


def frame_bytes(fmt, w, h):
    return {"nv12": w * h * 3 // 2, "yuv420p": w * h * 3 // 2, "rgb24": w * h * 3, "bgr24": w * h * 3,
            "rgba": w * h * 4, "bgra": w * h * 4, "yuv444p": w * h * 3}[fmt]


def frame_ptrs(t, fmt, w, h):
    b = t.data_ptr()
    if fmt == "nv12": return [b, b + w * h], [w, w]
    if fmt == "yuv420p": return [b, b + w * h, b + w * h + (w // 2) * (h // 2)], [w, w // 2, w // 2]
    if fmt == "yuv444p": return [b, b + w * h, b + 2 * w * h], [w, w, w]
    bpp = 4 if fmt in ("rgba", "bgra") else 3
    return [b], [w * bpp]


def sws_case(label, sf, sw, sh, df, dw, dh, flags=SWS["bicubic"], fused=None):
    if ONLY and ONLY not in label: return
    src = [torch.randint(0, 256, (frame_bytes(sf, sw, sh),), dtype=torch.uint8, device="cuda") for _ in range(NF)]
    dst = [torch.empty((frame_bytes(df, dw, dh),), dtype=torch.uint8, device="cuda") for _ in range(NF)]
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], flags, None)
    assert c, label
    lib.gmat_sws_setStream(c, stream)
    if fused is not None: lib.gmat_sws_setFused(c, fused)

    def f(i):
        sp, ss = frame_ptrs(src[i], sf, sw, sh); dp, ds = frame_ptrs(dst[i], df, dw, dh)
        r = lib.gmat_sws_scale(c, planes(sp), ints(ss), 0, sh, planes(dp), ints(ds))
        assert r == dh, r
    us = timeit(f)
    k = lib.gmat_sws_lastKernel(c).decode()
    lib.gmat_sws_freeContext(c)
    alg = frame_bytes(sf, sw, sh) + frame_bytes(df, dw, dh)
    print(f"{label:44s} {k:28s} {us:8.2f} us  {alg / us / 1e3:8.1f} GB/s ({alg / us / 1e3 / 80:4.1f}%)  {sw * sh / us / 1e3:7.1f} Gpix/s", flush=True)


def plane_case(label, fn_name, w, h, bpp, *extra):
    if ONLY and ONLY not in label: return
    src = [torch.randint(0, 256, (h, w * bpp), dtype=torch.uint8, device="cuda") for _ in range(NF)]
    tr = fn_name == "gmat_transpose"
    dst = [torch.empty((w, h * bpp) if tr else (h, w * bpp), dtype=torch.uint8, device="cuda") for _ in range(NF)]
    fn = getattr(lib, fn_name)

    def f(i):
        r = fn(src[i].data_ptr(), w * bpp, dst[i].data_ptr(), (h if tr else w) * bpp, w, h, bpp, *extra, stream)
        assert r == 0, r
    us = timeit(f)
    alg = 2 * w * h * bpp
    print(f"{label:44s} {fn_name:28s} {us:8.2f} us  {alg / us / 1e3:8.1f} GB/s ({alg / us / 1e3 / 80:4.1f}%)", flush=True)


def batch_case(label, sf, sw, sh, df, dw, dh, nstreams, flags=SWS["bicubic"]):
    """gmat_sws_scale_batch: NF frames per call; contexts on the 2:1 kernel put each stream's share into one launch"""
    if ONLY and ONLY not in label: return
    src = [torch.randint(0, 256, (frame_bytes(sf, sw, sh),), dtype=torch.uint8, device="cuda") for _ in range(NF)]
    dst = [torch.empty((frame_bytes(df, dw, dh),), dtype=torch.uint8, device="cuda") for _ in range(NF)]
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], flags, None)
    assert c, label
    sp = (C.c_void_p * (4 * NF))(); dp = (C.c_void_p * (4 * NF))()
    for i in range(NF):
        a, ss = frame_ptrs(src[i], sf, sw, sh); b, ds = frame_ptrs(dst[i], df, dw, dh)
        for k, v in enumerate(a): sp[4 * i + k] = v
        for k, v in enumerate(b): dp[4 * i + k] = v
    streams = (C.c_void_p * nstreams)(); streams[0] = stream
    for k in range(1, nstreams):
        h = C.c_void_p(); lib.gmat_stream_create(C.byref(h)); streams[k] = h

    def f(i):
        r = lib.gmat_sws_scale_batch(c, NF, C.cast(sp, C.POINTER(C.c_void_p)), ints(ss), C.cast(dp, C.POINTER(C.c_void_p)), ints(ds),
                                     C.cast(streams, C.POINTER(C.c_void_p)), nstreams, 3)
        assert r == NF, r
    us = timeit(f) / NF
    k = lib.gmat_sws_lastKernel(c).decode() + f" x{lib.gmat_sws_lastLaunchFrames(c)}"
    lib.gmat_sws_freeContext(c)
    alg = frame_bytes(sf, sw, sh) + frame_bytes(df, dw, dh)
    print(f"{label:44s} {k:28s} {us:8.2f} us  {alg / us / 1e3:8.1f} GB/s ({alg / us / 1e3 / 80:4.1f}%)  {sw * sh / us / 1e3:7.1f} Gpix/s  per frame", flush=True)


batch_case("4K nv12 -> 1080p rgb24, 16 frames/launch", "nv12", 3840, 2160, "rgb24", 1920, 1080, 1)
batch_case("4K nv12 -> 1080p rgb24, 2 x 8 frames", "nv12", 3840, 2160, "rgb24", 1920, 1080, 2)
batch_case("4K nv12 -> 1080p nv12, 16 frames/launch", "nv12", 3840, 2160, "nv12", 1920, 1080, 1)
batch_case("4K nv12 -> rgb24 (convert), 16 frames/launch", "nv12", 3840, 2160, "rgb24", 3840, 2160, 1)
batch_case("1080p nv12 -> rgb24 (convert), 16 fr/launch", "nv12", 1920, 1080, "rgb24", 1920, 1080, 1)
batch_case("4K nv12 -> 720p nv12, 16 frames/launch", "nv12", 3840, 2160, "nv12", 1280, 720, 1)
batch_case("1080p nv12 -> 720p nv12, 16 frames/launch", "nv12", 1920, 1080, "nv12", 1280, 720, 1)
batch_case("1080p nv12 -> 4K nv12 (up), 16 fr/launch", "nv12", 1920, 1080, "nv12", 3840, 2160, 1)
batch_case("4K nv12 -> 1080p nv12 lanczos, 16 fr/launch", "nv12", 3840, 2160, "nv12", 1920, 1080, 1, SWS["lanczos"])
batch_case("4K nv12 -> 1080p rgb24 lanczos, 16 fr/launch", "nv12", 3840, 2160, "rgb24", 1920, 1080, 1, SWS["lanczos"])
sws_case("4K nv12 -> 1080p rgb24 (headline)", "nv12", 3840, 2160, "rgb24", 1920, 1080)
sws_case("4K nv12 -> 1080p rgb24 lanczos", "nv12", 3840, 2160, "rgb24", 1920, 1080, SWS["lanczos"])
sws_case("4K nv12 -> 1080p nv12", "nv12", 3840, 2160, "nv12", 1920, 1080)
sws_case("4K yuv420p -> 1080p yuv420p", "yuv420p", 3840, 2160, "yuv420p", 1920, 1080)
sws_case("4K nv12 -> 720p nv12", "nv12", 3840, 2160, "nv12", 1280, 720)
sws_case("1080p nv12 -> 720p nv12", "nv12", 1920, 1080, "nv12", 1280, 720)
sws_case("1080p nv12 -> 4K nv12 (up)", "nv12", 1920, 1080, "nv12", 3840, 2160)
sws_case("4K nv12 -> 1080p nv12 lanczos", "nv12", 3840, 2160, "nv12", 1920, 1080, SWS["lanczos"])
sws_case("4K nv12 -> 1080p nv12 bilinear", "nv12", 3840, 2160, "nv12", 1920, 1080, SWS["bilinear"])
sws_case("4K nv12 -> rgb24 (convert)", "nv12", 3840, 2160, "rgb24", 3840, 2160)
sws_case("4K rgb24 -> nv12", "rgb24", 3840, 2160, "nv12", 3840, 2160)
sws_case("4K nv12 -> yuv420p (relayout)", "nv12", 3840, 2160, "yuv420p", 3840, 2160)
sws_case("4K nv12 -> 1080p yuv444p", "nv12", 3840, 2160, "yuv444p", 1920, 1080)
sws_case("4K nv12 -> yuv444p (chroma up)", "nv12", 3840, 2160, "yuv444p", 3840, 2160)
sws_case("4K rgb24 -> yuv444p", "rgb24", 3840, 2160, "yuv444p", 3840, 2160)
sws_case("4K rgb24 -> 1080p nv12 (scaled)", "rgb24", 3840, 2160, "nv12", 1920, 1080)
sws_case("4K rgb24 -> bgra (repack)", "rgb24", 3840, 2160, "bgra", 3840, 2160)
sws_case("4K rgba -> bgr24 (repack)", "rgba", 3840, 2160, "bgr24", 3840, 2160)
sws_case("4K rgb24 -> bgr24 (swap)", "rgb24", 3840, 2160, "bgr24", 3840, 2160)
plane_case("4K Y plane transpose", "gmat_transpose", 3840, 2160, 1, 1)
plane_case("4K UV plane transpose (bpp2)", "gmat_transpose", 1920, 1080, 2, 1)
plane_case("4K rgb24 transpose", "gmat_transpose", 3840, 2160, 3, 1)
plane_case("4K Y plane hflip", "gmat_flip", 3840, 2160, 1, 1)
plane_case("4K rgb24 hflip", "gmat_flip", 3840, 2160, 3, 1)
m = (C.c_int * 9)(1, 2, 1, 2, 4, 2, 1, 2, 1)
plane_case("4K Y plane smooth3x3", "gmat_smooth3x3", 3840, 2160, 1, m, C.c_float(1 / 16), C.c_float(0.0))
plane_case("4K rgb24 smooth3x3", "gmat_smooth3x3", 3840, 2160, 3, m, C.c_float(1 / 16), C.c_float(0.0))
plane_case("4K rgb24 median3x3", "gmat_median3x3", 3840, 2160, 3)
plane_case("4K Y plane median3x3", "gmat_median3x3", 3840, 2160, 1)


def rotate_case(label, w, h, bpp, deg, bilinear):
    if ONLY and ONLY not in label: return
    import math
    src = [torch.randint(0, 256, (h, w * bpp), dtype=torch.uint8, device="cuda") for _ in range(NF)]
    dst = [torch.empty((h, w * bpp), dtype=torch.uint8, device="cuda") for _ in range(NF)]
    fill = (C.c_uint8 * 4)(0, 0, 0, 255)

    def f(i):
        r = lib.gmat_rotate(src[i].data_ptr(), w * bpp, dst[i].data_ptr(), w * bpp, w, h, w, h, bpp,
                            C.c_double(math.radians(deg)), bilinear, fill, stream)
        assert r == 0, r
    us = timeit(f)
    alg = 2 * w * h * bpp
    print(f"{label:44s} {'gmat_rotate':28s} {us:8.2f} us  {alg / us / 1e3:8.1f} GB/s ({alg / us / 1e3 / 80:4.1f}%)", flush=True)


rotate_case("4K rgb24 rotate 17deg bilinear", 3840, 2160, 3, 17.0, 1)
rotate_case("4K rgb24 rotate 17deg nearest", 3840, 2160, 3, 17.0, 0)
rotate_case("4K Y plane rotate 17deg bilinear", 3840, 2160, 1, 17.0, 1)
