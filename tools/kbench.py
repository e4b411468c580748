#!/usr/bin/env python3
"""Kernel micro-benchmark used during tuning (not the headline bench): times single kernels over a
rotating 4K frame set with HIP events, for a list of env-var configurations.
usage: python tools/kbench.py [reps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gmat_amd
from gmat_amd.lib import PIX_FMT, SWS, planes, ints

lib = gmat_amd.load()
SW, SH, DW, DH = 3840, 2160, 1920, 1080
NF = 24
stream = C.c_void_p(); lib.gmat_stream_create(C.byref(stream))
nv12 = [torch.randint(0, 256, (SH * 3 // 2, SW), dtype=torch.uint8, device="cuda") for _ in range(NF)]
rgb4k = [torch.randint(0, 256, (SH, SW * 3), dtype=torch.uint8, device="cuda") for _ in range(NF)]
out = [torch.empty((DH, 5888), dtype=torch.uint8, device="cuda") for _ in range(NF)]
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 96


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


def ctx(src, dw, dh, dst="rgb24", flags=SWS["bicubic"], fused=1, sw=SW, sh=SH):
    c = lib.gmat_sws_getContext(sw, sh, PIX_FMT[src], dw, dh, PIX_FMT[dst], flags, None)
    assert c
    lib.gmat_sws_setStream(c, stream); lib.gmat_sws_setFused(c, fused)
    return c


def run(label, env, src, bytes_alg, fused=1):
    for k in list(os.environ):
        if k.startswith("GMAT_SCALE_"): del os.environ[k]
    os.environ.update(env)
    if src == "nv12":
        c = ctx("nv12", DW, DH, fused=fused)
        f = lambda i: lib.gmat_sws_scale(c, planes([nv12[i].data_ptr(), nv12[i].data_ptr() + SW * SH]), ints([SW, SW]), 0, SH,
                                         planes([out[i].data_ptr()]), ints([5888]))
    elif src == "rgb24":
        c = ctx("rgb24", DW, DH)
        f = lambda i: lib.gmat_sws_scale(c, planes([rgb4k[i].data_ptr()]), ints([SW * 3]), 0, SH,
                                         planes([out[i].data_ptr()]), ints([5888]))
    else:   # convert only
        c = ctx("nv12", SW, SH)
        f = lambda i: lib.gmat_sws_scale(c, planes([nv12[i].data_ptr(), nv12[i].data_ptr() + SW * SH]), ints([SW, SW]), 0, SH,
                                         planes([rgb4k[i].data_ptr()]), ints([SW * 3]))
    us = timeit(f)
    k = lib.gmat_sws_lastKernel(c).decode()
    lib.gmat_sws_freeContext(c)
    print(f"{label:34s} {k:28s} {us:8.2f} us  {bytes_alg / us / 1e3:8.1f} GB/s  {bytes_alg / us / 1e3 / 80:5.1f}% of 8TB/s  {SW*SH/us/1e3:7.1f} Gpix/s")


ALG_F, ALG_S, ALG_C = 18662400, 31104000, 37324800
CONFIGS = [
    ("convert 4K nv12->rgb24", {}, "conv", ALG_C),
    ("scale rgb24 default", {}, "rgb24", ALG_S),
    ("direct nv12 (1 sws ctx)", {}, "nv12", ALG_F, 2),
    ("direct TH8", {"GMAT_SCALE_TH": "8"}, "nv12", ALG_F, 2),
    ("direct TH32", {"GMAT_SCALE_TH": "32"}, "nv12", ALG_F, 2),
    ("direct TW32", {"GMAT_SCALE_TW": "32"}, "nv12", ALG_F, 2),
    ("direct noxcd", {"GMAT_SCALE_XCD": "0"}, "nv12", ALG_F, 2),
    ("fused nv12 default", {}, "nv12", ALG_F),
    ("fused noxcd", {"GMAT_SCALE_XCD": "0"}, "nv12", ALG_F),
    ("fused TH8", {"GMAT_SCALE_TH": "8"}, "nv12", ALG_F),
    ("fused TH32", {"GMAT_SCALE_TH": "32"}, "nv12", ALG_F),
    ("fused TW32", {"GMAT_SCALE_TW": "32"}, "nv12", ALG_F),
    ("fused nodirect", {"GMAT_SCALE_NO_DIRECT": "1"}, "nv12", ALG_F),
    ("scale rgb24 TH32", {"GMAT_SCALE_TH": "32"}, "rgb24", ALG_S),
    ("scale rgb24 noxcd", {"GMAT_SCALE_XCD": "0"}, "rgb24", ALG_S),
]
extra = os.environ.get("KBENCH_ONLY")
for c in CONFIGS:
    if extra and extra not in c[0]: continue
    run(*c)
