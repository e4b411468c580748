"""why does bench.py see 8.4 us/frame where kbench_ops sees 6.7?  (tuning aid)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import gmat_amd
import bench
lib = gmat_amd.load()
stream = C.c_void_p(); lib.gmat_stream_create(C.byref(stream))

def t(r, reps=64):
    tm = C.c_void_p(); lib.gmat_timer_create(C.byref(tm))
    for _ in range(4): r.step(3)
    lib.gmat_stream_sync(stream)
    best = 1e9
    for _ in range(3):
        lib.gmat_timer_begin(tm, stream)
        for _ in range(reps): r.step(3)
        lib.gmat_timer_end(tm, stream)
        ms = C.c_float(); lib.gmat_timer_elapsed_ms(tm, C.byref(ms))
        best = min(best, ms.value / reps / r.frames * 1e3)
    return best

for reps in (64, 30, 8):
    r = bench.Runner(lib, torch, stream, 32, 2, False, seed=1, branches=1)
    print("Runner 32 frames, 1 stream, reps", reps, "%.2f us/frame" % t(r, reps), flush=True)
    r.close()
if "--dist" in sys.argv:
    from gmat_amd import dist as gdist
    gdist.init("nccl")
    r = bench.Runner(lib, torch, stream, 32, 2, False, seed=1, branches=1)
    print("after dist.init: %.2f us/frame" % t(r), flush=True)
    r.close()
# bench's own timed()
r = bench.Runner(lib, torch, stream, 32, 2, False, seed=1, branches=1)
wall, ms = bench.timed(lib, torch, None, r, stream, 30, 5, 1)
print("bench.timed 30 steps: %.2f us/frame (device), wall %.2f" % (ms / 30 / 32 * 1e3, wall / 30 / 32 * 1e6))
wall, ms = bench.timed(lib, torch, None, r, stream, 200, 5, 1)
print("bench.timed 200 steps: %.2f us/frame (device), wall %.2f" % (ms / 200 / 32 * 1e3, wall / 200 / 32 * 1e6))
r.close()
