#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of the single-context YUV scaler (tuning aid)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, numpy as np, gmat_amd
from gmat_amd.lib import PIX_FMT, SWS, planes, ints
lib = gmat_amd.load()
SW, SH, DW, DH = 3840, 2160, 1920, 1080
src = torch.randint(0, 256, (SH * 3 // 2, SW), dtype=torch.uint8, device="cuda")
dst = torch.empty((DH, 5888), dtype=torch.uint8, device="cuda")
prof = torch.zeros((4096, 8), dtype=torch.int64, device="cuda")
c = lib.gmat_sws_getContext(SW, SH, PIX_FMT["nv12"], DW, DH, PIX_FMT["rgb24"], SWS["bicubic"], None)
lib.gmat_sws_setProfileBuffer(c, prof.data_ptr())
for _ in range(3):
    lib.gmat_sws_scale(c, planes([src.data_ptr(), src.data_ptr() + SW * SH]), ints([SW, SW]), 0, SH, planes([dst.data_ptr()]), ints([5888]))
torch.cuda.synchronize()
p = prof.cpu().numpy()
p = p[p[:, 5] != 0]
d = np.diff(p[:, :6], axis=1)
names = ["phase1 load", "barrier1", "phase2 hfilt", "barrier2", "phase3 vfilt+store"]
print("blocks", len(p), "kernel span (ticks):", p[:, 5].max() - p[:, 0].min())
for i, n in enumerate(names):
    print(f"{n:20s} mean {d[:, i].mean():9.0f}  p50 {np.median(d[:, i]):9.0f}  p95 {np.percentile(d[:, i], 95):9.0f}")
print(f"{'block total':20s} mean {(p[:,5]-p[:,0]).mean():9.0f}")
