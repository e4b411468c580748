#!/usr/bin/env python3
"""Throughput of the direct nv12 4K->1080p context with S concurrent streams (tuning aid)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, gmat_amd
from gmat_amd.lib import PIX_FMT, SWS, planes, ints
lib = gmat_amd.load()
SW, SH, DW, DH = 3840, 2160, 1920, 1080
NF = 32
nv12 = [torch.randint(0, 256, (SH * 3 // 2, SW), dtype=torch.uint8, device="cuda") for _ in range(NF)]
out = [torch.empty((DH, 5888), dtype=torch.uint8, device="cuda") for _ in range(NF)]
for S in (1, 2, 3, 4, 8):
    streams, ctxs = [], []
    for s in range(S):
        st = C.c_void_p(); lib.gmat_stream_create(C.byref(st)); streams.append(st)
        c = lib.gmat_sws_getContext(SW, SH, PIX_FMT["nv12"], DW, DH, PIX_FMT["rgb24"], SWS["bicubic"], None)
        lib.gmat_sws_setStream(c, st); ctxs.append(c)
    def go(n):
        for i in range(n):
            s = i % S; f = i % NF
            lib.gmat_sws_scale(ctxs[s], planes([nv12[f].data_ptr(), nv12[f].data_ptr() + SW * SH]), ints([SW, SW]), 0, SH,
                               planes([out[f].data_ptr()]), ints([5888]))
    go(64); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); go(512); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = min(best, dt / 512 * 1e6)
    print(f"streams={S}: {best:7.2f} us/frame  {SW*SH/best/1e3:7.1f} Gpix/s  {18662400/best/1e3:7.1f} GB/s")
    for c in ctxs: lib.gmat_sws_freeContext(c)
