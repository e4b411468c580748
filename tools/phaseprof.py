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
pall = prof.cpu().numpy()
# per-XCD (block b runs on XCD b % 8; shader clocks are only comparable within one XCD)
for x in range(8):
    px = pall[x::8]; px = px[px[:, 5] != 0]
    span = px[:, 5].max() - px[:, 0].min()
    busy = (px[:, 5] - px[:, 0]).sum()
    print(f"XCD {x}: blocks {len(px)}  span {span} ticks  sum(block time)/span = {busy/span:.1f} blocks in flight = {busy/span/32:.2f} per CU")
p = pall[pall[:, 5] != 0]
d = np.diff(p[:, :6], axis=1)
names = ["phase1 load", "barrier1", "phase2 hfilt", "barrier2", "phase3 vfilt+store"]
print("blocks", len(p), "kernel span (ticks):", p[:, 5].max() - p[:, 0].min())
for i, n in enumerate(names):
    print(f"{n:20s} mean {d[:, i].mean():9.0f}  p50 {np.median(d[:, i]):9.0f}  p95 {np.percentile(d[:, i], 95):9.0f}")
rt = (pall[:, 7] - pall[:, 6])[pall[:, 5] != 0]
print("block time by 100MHz realtime clock: mean %.0f ticks = %.2f us; shader ticks per us = %.0f" % (rt.mean(), rt.mean() / 100.0, (p[:,5]-p[:,0]).mean() / (rt.mean() / 100.0)))
ok = pall[pall[:, 5] != 0]
print("frame span by realtime clock: %.2f us" % ((ok[:, 7].max() - ok[:, 6].min()) / 100.0))
starts = np.sort(ok[:, 6] - ok[:, 6].min()); ends = np.sort(ok[:, 7] - ok[:, 6].min())
for t in range(0, int(ends.max()) + 1, 100):
    print("  t=%4.1fus  started %4d  finished %4d  in flight %4d" % (t / 100.0, (starts <= t).sum(), (ends <= t).sum(), (starts <= t).sum() - (ends <= t).sum()))
print(f"{'block total':20s} mean {(p[:,5]-p[:,0]).mean():9.0f}")
