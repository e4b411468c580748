import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, harness, gmat_amd
lib = gmat_amd.load(); orc = harness.load_oracle('oracle/liborc.so'); dev = harness.Dev(lib, 'hip')
for (w, h) in [(260, 34), (256, 64), (640, 360), (640, 64), (256, 360), (1920, 1080)]:
    src = harness.synth_planes(orc, 'nv12', w, h, seed=7)
    want = orc.yuv2rgb(src, w, h, 'nv12', 'rgb24')
    for sync in (0, 1):
        d = dev.upload_planes(src, 256)
        if sync: lib.gmat_device_sync()
        # verify upload
        back = [p.download() for p in d]
        up_ok = all((b == s).all() for b, s in zip(back, src))
        got, pads, k = dev.sws(d, w, h, 'nv12', w, h, 'rgb24', dst_align=256)
        bad = np.argwhere(got[0] != want)
        print(w, h, 'sync', sync, 'upload_ok', up_ok, 'bad', len(bad), bad[:3].tolist(), got[0][0, :6], want[0, :6])
