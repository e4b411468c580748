#!/usr/bin/env python3
"""Throughput of the direct nv12 4K->1080p context with S concurrent streams, submitted from C
(gmat_sws_scale_batch) so the host is not the limit (tuning aid)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, gmat_amd
from gmat_amd.lib import PIX_FMT, SWS, ints
lib = gmat_amd.load()
SW, SH, DW, DH = 3840, 2160, 1920, 1080
NF = 32
MODE = int(os.environ.get("KB_MODE", "2"))
nv12 = [torch.randint(0, 256, (SH * 3 // 2, SW), dtype=torch.uint8, device="cuda") for _ in range(NF)]
out = [torch.empty((DH, 5888), dtype=torch.uint8, device="cuda") for _ in range(NF)]
REP = 8
sp = (C.c_void_p * (4 * NF * REP))(); dp = (C.c_void_p * (4 * NF * REP))()
for i in range(NF * REP):
    f = i % NF
    sp[4 * i], sp[4 * i + 1], dp[4 * i] = nv12[f].data_ptr(), nv12[f].data_ptr() + SW * SH, out[f].data_ptr()
c = lib.gmat_sws_getContext(SW, SH, PIX_FMT["nv12"], DW, DH, PIX_FMT["rgb24"], SWS["bicubic"], None)
lib.gmat_sws_setFused(c, MODE)
for S in (1, 2, 3, 4, 6, 8):
    st = (C.c_void_p * S)()
    for s in range(S):
        h = C.c_void_p(); lib.gmat_stream_create(C.byref(h)); st[s] = h
    def go():
        r = lib.gmat_sws_scale_batch(c, NF * REP, C.cast(sp, C.POINTER(C.c_void_p)), ints([SW, SW]),
                                     C.cast(dp, C.POINTER(C.c_void_p)), ints([5888]), C.cast(st, C.POINTER(C.c_void_p)), S, 0)
        assert r == NF * REP, r
    go(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); go(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = min(best, dt / (NF * REP) * 1e6)
    print(f"mode {MODE} streams={S}: {best:7.2f} us/frame  {SW*SH/best/1e3:7.1f} Gpix/s  {18662400/best/1e3:7.1f} GB/s  kernel {lib.gmat_sws_lastKernel(c).decode()}")
