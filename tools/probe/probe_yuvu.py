"""parity probe of the quad-lane walker (k_scale_yuvu.hip) over a few up-scale geometries; PROBE_HIP=1: on the GPU"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import harness
from harness import SWS
from gmat_amd.lib import load
os.environ.setdefault("GMAT_QUAD_WALKER", "2")
orc = harness.load_oracle(ROOT + "/oracle/liborc.so")
HIP = os.environ.get("PROBE_HIP") == "1"
lib = load() if HIP else load(ROOT + "/tests/hipemu/build/libgmat_hip_emu.so")
dev = harness.Dev(lib, "hip" if HIP else "emu")
geoms = [(160, 90, 240, 136), (128, 72, 320, 180), (128, 72, 256, 144), (200, 120, 300, 180), (320, 180, 480, 270), (64, 36, 200, 100), (96, 54, 400, 300), (256, 64, 260, 66), (384,216,640,360)]
if len(sys.argv) > 1:
    geoms = [tuple(int(v) for v in sys.argv[1].split(","))]
fails = 0
for geom in geoms:
    sw, sh, dw, dh = geom
    for sf in ("nv12", "yuv420p"):
        for df in ("rgb24", "bgra", sf):
            for flags in ("bicubic", "bilinear", "lanczos"):
                src = harness.synth_planes(orc, sf, sw, sh, seed=5)
                want = orc.sws(src, sw, sh, sf, dw, dh, df, SWS[flags])
                d = dev.upload_planes(src, 256, 0)
                got, pads, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, SWS[flags], dst_align=64, dst_extra=0)
                for p in d: p.free()
                bad = sum(int((g != w).sum()) for g, w in zip(got, want))
                padbad = any((pd != 0xCD).any() for pd in pads)
                if bad or padbad: fails += 1
                print(geom, sf, df, flags, kernel, "BAD %d" % bad if bad else "ok", "PAD!" if padbad else "")
print("fails", fails)
