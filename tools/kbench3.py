#!/usr/bin/env python3
"""Per-kernel timings of the other configs: cfg2 (1080p convert), cfg4 (4K transforms), rgb->yuv."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, gmat_amd
from gmat_amd.lib import PIX_FMT, SWS, planes, ints
lib = gmat_amd.load()
stream = C.c_void_p(); lib.gmat_stream_create(C.byref(stream))
NF = 16
def timeit(fn, reps=64):
    t = C.c_void_p(); lib.gmat_timer_create(C.byref(t))
    for i in range(4): fn(i % NF)
    lib.gmat_stream_sync(stream)
    best = 1e9
    for _ in range(3):
        lib.gmat_timer_begin(t, stream)
        for i in range(reps): fn(i % NF)
        lib.gmat_timer_end(t, stream)
        ms = C.c_float(); lib.gmat_timer_elapsed_ms(t, C.byref(ms)); best = min(best, ms.value / reps * 1e3)
    return best
def report(name, us, alg_bytes, px):
    print(f"{name:44s} {us:8.2f} us  {alg_bytes/us/1e3:8.1f} GB/s ({alg_bytes/us/1e3/80:5.1f}% of 8 TB/s)  {px/us/1e3:8.1f} Gpix/s")
# cfg2: 1080p nv12 -> rgb24
W, H = 1920, 1080
src = [torch.randint(0, 256, (H * 3 // 2, W), dtype=torch.uint8, device="cuda") for _ in range(NF)]
dst = [torch.empty((H, W * 3), dtype=torch.uint8, device="cuda") for _ in range(NF)]
c = lib.gmat_sws_getContext(W, H, PIX_FMT["nv12"], W, H, PIX_FMT["rgb24"], 0, None); lib.gmat_sws_setStream(c, stream)
report("cfg2 1080p nv12->rgb24", timeit(lambda i: lib.gmat_sws_scale(c, planes([src[i].data_ptr(), src[i].data_ptr() + W * H]), ints([W, W]), 0, H, planes([dst[i].data_ptr()]), ints([W * 3]))), W * H * 4.5, W * H)
# 4K frames
W, H = 3840, 2160
rgb = [torch.randint(0, 256, (H, W * 3), dtype=torch.uint8, device="cuda") for _ in range(NF)]
out = [torch.empty((W, H * 3), dtype=torch.uint8, device="cuda") for _ in range(NF)]     # transposed shape (same bytes)
nv = [torch.empty((H * 3 // 2, W), dtype=torch.uint8, device="cuda") for _ in range(NF)]
B = W * H * 3
report("cfg4 transpose(clock) 4K rgb24", timeit(lambda i: lib.gmat_transpose(rgb[i].data_ptr(), W * 3, out[i].data_ptr(), H * 3, W, H, 3, 1, stream)), 2 * B, W * H)
report("cfg4 hflip 4K rgb24", timeit(lambda i: lib.gmat_flip(rgb[i].data_ptr(), W * 3, out[i].data_ptr(), W * 3, W, H, 3, 1, stream)), 2 * B, W * H)
report("cfg4 vflip 4K rgb24", timeit(lambda i: lib.gmat_flip(rgb[i].data_ptr(), W * 3, out[i].data_ptr(), W * 3, W, H, 3, 0, stream)), 2 * B, W * H)
m = (C.c_int * 9)(1, 2, 1, 2, 4, 2, 1, 2, 1)
report("cfg4 smooth3x3 4K rgb24", timeit(lambda i: lib.gmat_smooth3x3(rgb[i].data_ptr(), W * 3, out[i].data_ptr(), W * 3, W, H, 3, m, 1 / 16, 0.0, stream)), 2 * B, W * H)
report("cfg4 fused rotate+flip+smooth 4K", timeit(lambda i: lib.gmat_rotate_flip_smooth(rgb[i].data_ptr(), W * 3, out[i].data_ptr(), H * 3, W, H, 3, stream)), 2 * B, W * H)
report("crop 4K -> 1080p window", timeit(lambda i: lib.gmat_crop(rgb[i].data_ptr(), W * 3, out[i].data_ptr(), 1920 * 3, 960, 540, 1920, 1080, 3, stream)), 2 * 1920 * 1080 * 3, 1920 * 1080)
c2 = lib.gmat_sws_getContext(W, H, PIX_FMT["rgb24"], W, H, PIX_FMT["nv12"], 0, None); lib.gmat_sws_setStream(c2, stream)
report("rgb24 -> nv12 4K", timeit(lambda i: lib.gmat_sws_scale(c2, planes([rgb[i].data_ptr()]), ints([W * 3]), 0, H, planes([nv[i].data_ptr(), nv[i].data_ptr() + W * H]), ints([W, W]))), W * H * 4.5, W * H)
c3 = lib.gmat_sws_getContext(W, H, PIX_FMT["rgb24"], W, H, PIX_FMT["bgr24"], 0, None); lib.gmat_sws_setStream(c3, stream)
report("rgb24 -> bgr24 4K", timeit(lambda i: lib.gmat_sws_scale(c3, planes([rgb[i].data_ptr()]), ints([W * 3]), 0, H, planes([out[i].data_ptr()]), ints([W * 3]))), 2 * B, W * H)
