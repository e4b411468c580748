#!/usr/bin/env python3
"""bench.py — headline benchmark of the pixel-transform hot path on MI355X.

Workload (BASELINE.json configs[2], the config `metric` is quoted on): synthetic 3840x2160 NV12
frames, device resident -> RGB24 -> 1920x1080 bicubic RGB24, through the C ABI (gmat_sws_scale_batch).
A "step" is PASSES (default 12) passes over FRAMES (default 256) distinct frame pairs — 4.8 GB, larger than the 256 MiB
Infinity Cache, so HBM is measured, not cache; 12 x 256 frames = 10.4 ms a step, so that the driver's 20 steps are a timed region
of > 200 ms (round 3's was 17 ms: one pass a step).

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torchrun environment re-launches itself as N ranks (one per GPU) under
torch.distributed.run on 127.0.0.1; under torchrun it uses RANK / LOCAL_RANK / WORLD_SIZE as given.  Every
rank converts its own independent streams on its own GPU (hipSetDevice(LOCAL_RANK); the reference selects the
device per stream the same way, libavutil/hwcontext_cuda.c:395-434); RCCL carries only the start / stop barrier
and the MAX of the wall time.  `--dry` runs the same code on the CPU-emulated build of the library with gloo and
a tiny geometry (plumbing check in a GPU-less container; its numbers mean nothing and say so).

Output (rank 0).  Secondary measurements are printed on EARLIER lines, one `BENCH_DETAIL {json}` line per block
(chained, other_configs, c_harness, host_pipeline, per_rank, cpu_configs0, ...) and collected in bench_detail.json;
the LAST stdout line is ONE compact JSON object (< 2 KB): metric / value / unit / n_gpus / steps / warmup / ms_per_step
/ dtype / data / config{workload, ...} plus
  roofline      : the dominant kernel; achieved = algorithmic bytes per launch / average launch duration from
                  HIP events on the launch stream over the timed steps; traffic = HBM bytes per launch from a live
                  rocprofv3 PMC pass of this run (FETCH_SIZE x2 + WRITE_SIZE, separate passes), else from the
                  committed pass of this round (`traffic_source` says which).
  cpu_baseline  : libswscale arithmetic on the host cores, rank 0, bounded sample: stock libswscale.so when the
                  box has one ("reference"), else the C oracle ("port").
`value` = source gigapixels per second over all ranks (weak scaling).  A step = PASSES x (FRAMES / 32) launch sets of 32
frames each over a working set of distinct frame pairs (default 256 pairs = 4.8 GB: HBM, not Infinity Cache), every launch
set spread over TWO streams (config.workload says so; `roofline` is the same kernel on ONE stream).
`--same-device` (rehearsal of configs[4] on a 1-GPU box): every rank uses device 0 and the control plane is gloo (RCCL refuses
two ranks on one device); the launcher, per-rank NUMA binding, N working sets and the compact last line run as on 8 GPUs.
"""
import argparse
import ctypes as C
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PRE_WARM_MS = 40            # untimed clock-ramp load in front of the W warm-up steps
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r03_traffic.json")     # this round's committed PMC pass (fallback)
LAUNCH_FRAMES = int(os.environ.get("GMAT_BENCH_LAUNCH_FRAMES", "32"))          # frames one launch set carries (kYuv2xMaxFrames; the variable: experiments only)
DETAIL = {}                 # secondary blocks: printed on earlier lines + bench_detail.json, never in the last line


def detail(key, block):
    """a secondary block: its own earlier stdout line, and bench_detail.json at the end"""
    DETAIL[key] = block
    print("BENCH_DETAIL " + json.dumps({key: block}), flush=True)


class Geo:
    """the workload's geometry and its algorithmic byte counts"""

    def __init__(self, sw, sh, dw, dh):
        self.sw, self.sh, self.dw, self.dh = sw, sh, dw, dh
        self.px = sw * sh
        self.nv12 = self.px * 3 // 2
        self.rgb_src = self.px * 3
        self.rgb_dst = dw * dh * 3
        self.alg_fused = self.nv12 + self.rgb_dst          # 18,662,400 B at 4K -> 1080p
        self.alg_convert = self.nv12 + self.rgb_src        # 37,324,800 B
        self.alg_scale = self.rgb_src + self.rgb_dst       # 31,104,000 B


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=256,
                    help="distinct frame pairs of the working set (a multiple of 32; 256 pairs = 4.8 GB)")
    ap.add_argument("--passes", type=int, default=12,
                    help="passes over the working set that make ONE step (12 x 256 frames = 10.4 ms: 20 steps > 200 ms)")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal: every rank on device 0, control plane on gloo (N ranks on a 1-GPU box)")
    ap.add_argument("--branches", type=int, default=2,
                    help="concurrent HIP streams the independent frames of a launch set are spread over (headline `value`)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-detail", "--no-chained", dest="no_detail", action="store_true",
                    help="skip the secondary GPU measurements (chained forms, other configs, C harness)")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the pinned-host upload/compute/download leg")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the live rocprofv3 PMC traffic pass")
    ap.add_argument("--cpu-frames", type=int, default=128)
    ap.add_argument("--dry", action="store_true",
                    help="CPU plumbing check: emulated library, gloo, tiny frames; numbers are not measurements")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch_as_ranks(n):
    """`python bench.py --gpus N` from a plain shell: become N ranks under torch.distributed.run"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


class DevMem:
    """device frames through the C ABI (gmat_malloc); torch is not in the data path"""

    def __init__(self, lib):
        self.lib, self.blocks = lib, []

    def alloc(self, nbytes, fill=None):
        p = C.c_void_p()
        if self.lib.gmat_malloc(C.byref(p), nbytes) != 0 or not p.value:
            raise RuntimeError(f"gmat_malloc({nbytes}) failed")
        self.blocks.append(p.value)
        if fill is not None:
            if self.lib.gmat_memcpy_h2d(p.value, fill.ctypes.data, nbytes) != 0:
                raise RuntimeError("gmat_memcpy_h2d failed")
        return p.value

    def free(self):
        for b in self.blocks:
            self.lib.gmat_free(b)
        self.blocks = []


def random_bytes(n, seed):
    import numpy as np
    return np.random.default_rng(seed).integers(0, 256, n, dtype=np.uint8)


class Runner:
    """One implementation of the workload: a context and its working set of `frames` distinct frame pairs.
    step() = one pass over the working set: frames / 32 launch sets, each ONE C call (gmat_sws_scale_batch) that hands every
    stream its share of the set's 32 frames as one launch (grid.y = frame)."""

    def __init__(self, lib, geo, stream, frames, fused, seed, branches=1, share=None, passes=1):
        from gmat_amd.lib import PIX_FMT, SWS, ints
        self.lib, self.stream, self.frames, self.geo, self.passes = lib, stream, frames, geo, max(1, passes)
        self.ctx = lib.gmat_sws_getContext(geo.sw, geo.sh, PIX_FMT["nv12"], geo.dw, geo.dh, PIX_FMT["rgb24"],
                                           SWS["bicubic"] | SWS["hwaccel"], None)
        if not self.ctx:
            raise RuntimeError("gmat_sws_getContext failed")
        if lib.gmat_sws_setFused(self.ctx, int(fused)) != 0:
            raise RuntimeError('gmat_sws_setFused failed')
        lib.gmat_sws_setStream(self.ctx, stream)
        src_ls = (geo.sw + 255) // 256 * 256                 # AVHWFramesContext row alignment
        dst_ls = (geo.dw * 3 + 255) // 256 * 256
        self.owner = share is None
        if share is None:
            self.mem = DevMem(lib)
            nsrc = src_ls * (geo.sh * 3 // 2)
            content = [random_bytes(nsrc, seed * 1000 + i) for i in range(min(frames, 8))]     # content does not affect timing
            self.src = [self.mem.alloc(nsrc, content[i % len(content)]) for i in range(frames)]
            self.dst = [self.mem.alloc(dst_ls * geo.dh) for _ in range(frames)]
        else:                                                # another context over the same frames (the serial roofline run)
            self.mem, self.src, self.dst = None, share.src[:frames], share.dst[:frames]
        self.src_ls, self.dst_ls = src_ls, dst_ls
        n = frames
        self.sp = (C.c_void_p * (4 * n))()
        self.dp = (C.c_void_p * (4 * n))()
        for i in range(n):
            self.sp[4 * i], self.sp[4 * i + 1] = self.src[i], self.src[i] + src_ls * geo.sh      # UV directly after Y
            self.dp[4 * i] = self.dst[i]
        self.ss, self.ds = ints([src_ls, src_ls]), ints([dst_ls])
        self.branches = branches
        self.streams = (C.c_void_p * max(1, branches))()
        self.streams[0] = stream
        for b in range(1, branches):
            h = C.c_void_p()
            lib.gmat_stream_create(C.byref(h))
            self.streams[b] = h
        self.per_launch = min(LAUNCH_FRAMES, n)
        self.sets = [(f0, min(self.per_launch, n - f0)) for f0 in range(0, n, self.per_launch)]
        vp = C.POINTER(C.c_void_p)
        step_sz = C.sizeof(C.c_void_p) * 4
        self.set_args = [(cnt, C.cast(C.byref(self.sp, f0 * step_sz), vp), C.cast(C.byref(self.dp, f0 * step_sz), vp))
                         for f0, cnt in self.sets]
        self.pstreams = C.cast(self.streams, vp)

    def step(self, flags=0):
        """flags: GMAT_BATCH_FORK (1) on the step's first launch set, GMAT_BATCH_JOIN (2) on its last one"""
        lib, last = self.lib, len(self.set_args) - 1
        for p in range(self.passes):
            for k, (cnt, sp, dp) in enumerate(self.set_args):
                fl = (flags & 1 if k == 0 and p == 0 else 0) | (flags & 2 if k == last and p == self.passes - 1 else 0)
                r = lib.gmat_sws_scale_batch(self.ctx, cnt, sp, self.ss, dp, self.ds, self.pstreams, self.branches, fl)
                if r != cnt:
                    raise RuntimeError(f"gmat_sws_scale_batch failed: {r}")

    def launches_per_step(self):
        return len(self.set_args) * self.passes

    def frames_per_step(self):
        return self.frames * self.passes

    def per_launch_ms(self, n):
        """n launches on streams[0], each between its OWN pair of HIP events (launches still back to back: an event is a
        packet between two dispatches): the distribution behind roofline.frac — frac_p50 / frac_min / frac_max"""
        lib = self.lib
        timers = []
        for _ in range(n):
            t = C.c_void_p()
            lib.gmat_timer_create(C.byref(t))
            timers.append(t)
        one = (C.c_void_p * 1)(self.stream)
        ps = C.cast(one, C.POINTER(C.c_void_p))
        k = 0
        for t in timers:
            cnt, sp, dp = self.set_args[k % len(self.set_args)]
            k += 1
            lib.gmat_timer_begin(t, self.stream)
            if lib.gmat_sws_scale_batch(self.ctx, cnt, sp, self.ss, dp, self.ds, ps, 1, 0) != cnt:
                raise RuntimeError("gmat_sws_scale_batch failed")
            lib.gmat_timer_end(t, self.stream)
        lib.gmat_stream_sync(self.stream)
        out = []
        for t in timers:
            ms = C.c_float()
            lib.gmat_timer_elapsed_ms(t, C.byref(ms))
            lib.gmat_timer_destroy(t)
            out.append(float(ms.value))
        return out

    def kernel(self):
        return self.lib.gmat_sws_lastKernel(self.ctx).decode()

    def frames_per_launch(self):
        return max(1, int(self.lib.gmat_sws_lastLaunchFrames(self.ctx)))

    def close(self):
        self.lib.gmat_device_sync()
        for b in range(1, self.branches):
            self.lib.gmat_stream_destroy(self.streams[b])
        self.lib.gmat_sws_freeContext(self.ctx)
        if self.owner:
            self.mem.free()


class Env:
    """what differs between the GPU run and the dry (CPU-emulated) run"""

    def __init__(self, dry, device):
        self.dry = dry
        if dry:
            self.sync = None
        else:
            import torch
            torch.cuda.set_device(device)
            self.sync = torch.cuda.synchronize

    def synchronize(self, lib):
        lib.gmat_device_sync()
        if self.sync:
            self.sync()                # the contract's torch.cuda.synchronize(); the work itself is on gmat's streams


wall_local = [0.0]          # this rank's own wall time of the last timed() (its return value is the MAX over ranks)
ctl_device = ["cpu"]        # where the control plane's tensors live: "cuda" under RCCL, "cpu" under gloo


def timed(lib, env, dist, runner, stream, steps, warmup, world, pre_warm_ms=PRE_WARM_MS):
    """W warm-up steps, then exactly K steps between barrier+synchronize pairs; returns
    (wall seconds MAX over ranks, device milliseconds from HIP events on the launch stream)."""
    timer = C.c_void_p()
    lib.gmat_timer_create(C.byref(timer))
    gc.collect()
    gc.disable()                        # no collector pauses inside the timed region
    # clock ramp: after the idle set-up phase the GPU needs several milliseconds of sustained load to reach its
    # working clocks.  Untimed, and in addition to the W warm-up steps; reported as config.pre_warm_ms.
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < pre_warm_ms * 1e-3:
        runner.step()
        lib.gmat_stream_sync(stream)
    for _ in range(warmup):
        runner.step()
    dist.barrier(world) if dist else None
    env.synchronize(lib)
    t0 = time.perf_counter()
    lib.gmat_timer_begin(timer, stream)
    for i in range(steps):
        # FORK (1): the side streams start after the begin event; JOIN (2): the end event covers them
        runner.step((1 if i == 0 else 0) | (2 if i == steps - 1 else 0))
    lib.gmat_timer_end(timer, stream)
    lib.gmat_stream_sync(stream)
    env.synchronize(lib)
    dist.barrier(world) if dist else None
    wall = time.perf_counter() - t0
    wall_local[0] = wall
    gc.enable()
    ms = C.c_float()
    lib.gmat_timer_elapsed_ms(timer, C.byref(ms))
    lib.gmat_timer_destroy(timer)
    if dist and world > 1:
        wall = dist.max_over_ranks(wall, world, device=ctl_device[0])
    return wall, float(ms.value)


def time_single_kernel(lib, env, runner_fn, stream, reps):
    """average duration (ms) of one repeated launch function, HIP events on the launch stream"""
    timer = C.c_void_p()
    lib.gmat_timer_create(C.byref(timer))
    env.synchronize(lib)
    gc.collect()
    gc.disable()
    for _ in range(8):
        runner_fn()
    lib.gmat_stream_sync(stream)
    lib.gmat_timer_begin(timer, stream)
    for _ in range(reps):
        runner_fn()
    lib.gmat_timer_end(timer, stream)
    gc.enable()
    ms = C.c_float()
    lib.gmat_timer_elapsed_ms(timer, C.byref(ms))
    lib.gmat_timer_destroy(timer)
    return float(ms.value) / reps


def committed_traffic(kernel, frames_per_launch):
    """HBM bytes per launch from this round's committed PMC collection (profiles/r03_traffic.json, tools/gpu_pass.sh pmc),
    scaled to the frames a launch of this run carries; None when the kernel has no entry."""
    try:
        d = json.load(open(TRAFFIC_FILE))
        for k, v in d["kernels"].items():
            if k.startswith(kernel):
                return int(v["traffic_bytes_per_frame"] * frames_per_launch)
    except Exception:
        pass
    return None


def live_traffic(kernel, frames_per_launch, budget_s=150):
    """HBM bytes per launch of `kernel` measured in THIS run: rocprofv3 --kernel-trace --pmc over tools/bin/x2bench's headline case
    (the same context, 32 frames per launch), FETCH_SIZE and WRITE_SIZE in separate passes as MI355X_MICROARCH.md prescribes;
    FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 bytes).  None when rocprofv3 or the
    harness is missing or a pass fails: the caller falls back to the committed pass and says so."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = os.path.join(ROOT, "tools", "bin", "x2bench")
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not prof or not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="gmat_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", X2BENCH_VERIFY="0", X2BENCH_JSON="1")
    vals = {}
    t0 = time.perf_counter()
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            left = budget_s - (time.perf_counter() - t0)
            if left < 20:
                return None
            out = os.path.join(tmp, ctr)
            r = subprocess.run([prof, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "p", "--",
                                exe, str(LAUNCH_FRAMES), "6", "nv12 4K->1080p rgb24 bicubic"],
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=left)
            if r.returncode != 0:
                return None
            got = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kernel in row.get("Kernel_Name", "") and row.get("Counter_Name") == ctr:
                        got.append(float(row["Counter_Value"]))
            if not got:
                return None
            vals[ctr] = sum(got) / len(got)                      # KiB per launch of LAUNCH_FRAMES frames
        per_frame = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024 / LAUNCH_FRAMES
        return {"bytes_per_launch": int(per_frame * frames_per_launch), "read_bytes_per_frame": int(2 * vals["FETCH_SIZE"] * 1024 / LAUNCH_FRAMES),
                "written_bytes_per_frame": int(vals["WRITE_SIZE"] * 1024 / LAUNCH_FRAMES), "seconds": round(time.perf_counter() - t0, 1)}
    except Exception:                                            # noqa: BLE001 - a counter pass must not take the headline down
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


local_device = [0]


def other_configs(lib, env, stream, geo, frames=16):
    """BASELINE configs[1] and configs[3], plus the 4:2:0 -> 4:2:0 down-scale of a transcode: one frame per call
    (what sws_scale / filter_frame do) and 32 frames per launch.  Secondary numbers; never the headline `value`."""
    from gmat_amd.lib import PIX_FMT, planes, ints
    res = {}
    mem = DevMem(lib)

    def frame_set(n, nbytes, seed):
        return [mem.alloc(nbytes, random_bytes(nbytes, seed + i) if i < 4 else None) for i in range(n)]

    def sws_case(name, sf, sw, sh, df, dw, dh, alg_bytes):
        src = frame_set(frames, sw * sh * 3 // 2, 11)
        dbytes = dw * dh * 3 if df == "rgb24" else dw * dh * 3 // 2
        dst = frame_set(frames, dbytes, 0)
        c = lib.gmat_sws_getContext(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], 4, None)
        lib.gmat_sws_setStream(c, stream)
        st = {"i": 0}

        def run():
            i = st["i"] = (st["i"] + 1) % frames
            s0, d0 = src[i], dst[i]
            dp = [d0] if df == "rgb24" else [d0, d0 + dw * dh]
            dl = [dw * 3] if df == "rgb24" else [dw, dw]
            lib.gmat_sws_scale(c, planes([s0, s0 + sw * sh]), ints([sw, sw]), 0, sh, planes(dp), ints(dl))
        ms = time_single_kernel(lib, env, run, stream, 4 * frames)
        res[name] = {"kernel": lib.gmat_sws_lastKernel(c).decode(), "avg_launch_us": round(ms * 1e3, 2),
                     "Gpix/s": round(sw * sh / (ms * 1e-3) / 1e9, 1), "algorithmic_bytes": alg_bytes,
                     "achieved_GBps": round(alg_bytes / (ms * 1e-3) / 1e9, 1),
                     "frac": round(alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        lib.gmat_sws_freeContext(c)

    def batch_case(name, sf, sw, sh, df, dw, dh, alg_bytes, n=32):
        src = frame_set(n, sw * sh * 3 // 2, 21)
        dst = frame_set(n, dw * dh * 3 if df == "rgb24" else dw * dh * 3 // 2, 0)
        c = lib.gmat_sws_getContext(sw, sh, PIX_FMT[sf], dw, dh, PIX_FMT[df], 4, None)
        sp = (C.c_void_p * (4 * n))(); dp = (C.c_void_p * (4 * n))()
        for i in range(n):
            sp[4 * i], sp[4 * i + 1], dp[4 * i] = src[i], src[i] + sw * sh, dst[i]
            if df != "rgb24":
                dp[4 * i + 1] = dst[i] + dw * dh
        streams = (C.c_void_p * 1)(stream)
        dstr = ints([dw * 3]) if df == "rgb24" else ints([dw, dw])

        def run():
            r = lib.gmat_sws_scale_batch(c, n, C.cast(sp, C.POINTER(C.c_void_p)), ints([sw, sw]), C.cast(dp, C.POINTER(C.c_void_p)),
                                         dstr, C.cast(streams, C.POINTER(C.c_void_p)), 1, 0)
            assert r == n
        ms = time_single_kernel(lib, env, run, stream, 64)
        fpl = int(lib.gmat_sws_lastLaunchFrames(c))
        res[name] = {"kernel": lib.gmat_sws_lastKernel(c).decode(), "frames_per_launch": fpl, "avg_launch_us": round(ms * 1e3, 2),
                     "us_per_frame": round(ms * 1e3 / n, 3), "Gpix/s": round(n * sw * sh / (ms * 1e-3) / 1e9, 1),
                     "algorithmic_bytes_per_launch": alg_bytes * n,
                     "achieved_GBps": round(alg_bytes * n / (ms * 1e-3) / 1e9, 1),
                     "frac": round(alg_bytes * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        lib.gmat_sws_freeContext(c)

    def queued_case(name, filt, opts, sf, sw, sh, alg_bytes, batch=32, nframes=256):
        """the AVFilter-shaped layer's queued form: one send_frame / receive_frame per frame, one launch per `batch` frames"""
        from gmat_amd.lib import GmatFrame
        fc = lib.gmat_hwframe_ctx_create(0 if env.dry else local_device[0], PIX_FMT[sf], sw, sh, batch + 2)
        f = lib.gmat_filter_alloc(filt.encode())
        for k, v in dict(opts, batch=batch).items():
            assert lib.gmat_filter_set_option(f, k.encode(), str(v).encode()) == 0
        assert lib.gmat_filter_init(f) == 0 and lib.gmat_filter_config_props(f, fc, stream) == 0
        out = C.POINTER(GmatFrame)()

        def one():
            fr = lib.gmat_frame_alloc()
            assert lib.gmat_hwframe_get_buffer(fc, fr) == 0         # content: whatever the pool block holds (timing only)
            assert lib.gmat_filter_send_frame(f, fr) == 0
            while lib.gmat_filter_receive_frame(f, C.byref(out)) == 0:
                lib.gmat_frame_free(C.byref(out))
        for _ in range(2 * batch):
            one()
        lib.gmat_stream_sync(stream)
        gc.collect(); gc.disable()
        t0 = time.perf_counter()
        for _ in range(nframes):
            one()
        lib.gmat_filter_flush(f)
        while lib.gmat_filter_receive_frame(f, C.byref(out)) == 0:
            lib.gmat_frame_free(C.byref(out))
        lib.gmat_stream_sync(stream)
        dt = time.perf_counter() - t0
        gc.enable()
        us = dt / nframes * 1e6
        res[name] = {"api": f"gmat_filter_send_frame / receive_frame, {filt} batch={batch}", "us_per_frame": round(us, 3),
                     "Gpix/s": round(sw * sh / us / 1e3, 1), "algorithmic_bytes": alg_bytes,
                     "achieved_GBps": round(alg_bytes / us / 1e3, 1), "frac": round(alg_bytes / us / 1e3 / HBM_PEAK_GBS, 4),
                     "timing": "host wall clock over the frames incl. pool get/put, stream synchronised at the end"}
        lib.gmat_filter_free(f)
        lib.gmat_hwframe_ctx_free(fc)

    hw, hh = geo.sw // 2, geo.sh // 2                 # 1920 x 1080 in the real run
    sws_case("configs[1]: 1080p nv12 -> rgb24", "nv12", hw, hh, "rgb24", hw, hh, hw * hh * 9 // 2)
    batch_case("configs[1], 32 frames per launch", "nv12", hw, hh, "rgb24", hw, hh, hw * hh * 9 // 2)
    sws_case("configs[2], one frame per call (sws_scale)", "nv12", geo.sw, geo.sh, "rgb24", geo.dw, geo.dh, geo.alg_fused)
    queued_case("configs[1], one frame per call, queued filter (batch 32)", "format_hip", {"pix_fmt": "rgb24"}, "nv12", hw, hh, hw * hh * 9 // 2)
    queued_case("configs[2], one frame per call, queued filter (batch 32)", "scale_hip", {"w": "iw/2", "h": "ih/2", "format": "rgb24"},
                "nv12", geo.sw, geo.sh, geo.alg_fused)
    tb = geo.nv12 + geo.dw * geo.dh * 3 // 2
    sws_case("transcode: 4K nv12 -> 1080p nv12 bicubic", "nv12", geo.sw, geo.sh, "nv12", geo.dw, geo.dh, tb)
    batch_case("transcode, 32 frames per launch", "nv12", geo.sw, geo.sh, "nv12", geo.dw, geo.dh, tb)
    # the other exact ratios between a decoder's frame and a network's input / a ladder rung (round 2's ratio walkers); batched only
    if geo.sw % 24 == 0 and geo.sh % 24 == 0 and geo.sw // 4 >= 64:
        for name, num, den in (("3:1", 1, 3), ("3:2", 2, 3), ("4:1", 1, 4)):
            dw, dh = geo.sw * num // den, geo.sh * num // den
            if dw % 8 or dh % 4:
                continue
            batch_case("nv12 %dx%d -> rgb24 %dx%d (%s), 32 frames per launch" % (geo.sw, geo.sh, dw, dh, name), "nv12", geo.sw, geo.sh, "rgb24", dw, dh,
                       geo.nv12 + dw * dh * 3)
        batch_case("nv12 %dx%d -> nv12 %dx%d (4:1), 32 frames per launch" % (geo.sw, geo.sh, geo.sw // 4, geo.sh // 4), "nv12", geo.sw, geo.sh, "nv12",
                   geo.sw // 4, geo.sh // 4, geo.nv12 + (geo.sw // 4) * (geo.sh // 4) * 3 // 2)
    # up-scales (round 4: the quad-lane walker, k_scale_yuvu.hip — any factor; the tiled kernel's beyond 1 : 2 before it)
    if geo.sw == 3840 and geo.sh == 2160:
        for name, sw, sh, dw, dh, df in (("720p -> 1080p nv12 (1:1.5)", 1280, 720, 1920, 1080, "nv12"), ("720p -> 1080p rgb24 (1:1.5)", 1280, 720, 1920, 1080, "rgb24"),
                                         ("720p -> 4K nv12 (1:3)", 1280, 720, 3840, 2160, "nv12"), ("1080p -> 4K rgb24 (1:2)", 1920, 1080, 3840, 2160, "rgb24")):
            batch_case("up-scale nv12 %s, 32 frames per launch" % name, "nv12", sw, sh, df, dw, dh, sw * sh * 3 // 2 + (dw * dh * 3 if df == "rgb24" else dw * dh * 3 // 2))
    # configs[3]: rotate(90) + hflip + 3x3 smooth as ONE kernel on 4K rgb24
    w, h = geo.sw, geo.sh
    src = frame_set(frames, w * h * 3, 31)
    dst = frame_set(frames, w * h * 3, 0)
    st = {"i": 0}

    def run4():
        i = st["i"] = (st["i"] + 1) % frames
        lib.gmat_rotate_flip_smooth(src[i], w * 3, dst[i], h * 3, w, h, 3, stream)
    ms = time_single_kernel(lib, env, run4, stream, 4 * frames)
    alg = 2 * w * h * 3
    res["configs[3]: 4K rgb24 rotate(90)+flip+3x3 smooth, fused"] = {
        "kernel": "smooth121_kernel<3,transposed,60,8>", "avg_launch_us": round(ms * 1e3, 2),
        "Gpix/s": round(w * h / (ms * 1e-3) / 1e9, 1), "algorithmic_bytes": alg,
        "achieved_GBps": round(alg / (ms * 1e-3) / 1e9, 1), "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    # the same into a destination with a hardware pool's pitch (rows aligned to 256 bytes, gframes.cpp; the dense frame's 6480 is not a multiple of
    # a 128-byte line): 128-row tiles whose pieces are whole lines, streaming stores
    pitch = (h * 3 + 255) // 256 * 256
    dstp = frame_set(frames, pitch * w, 0)

    def run4p():
        i = st["i"] = (st["i"] + 1) % frames
        lib.gmat_rotate_flip_smooth(src[i], w * 3, dstp[i], pitch, w, h, 3, stream)
    ms = time_single_kernel(lib, env, run4p, stream, 4 * frames)
    res["configs[3] into a pool frame's pitch (%d bytes a row)" % pitch] = {
        "kernel": "smooth121_kernel<3,transposed,60,16,128 rows>", "avg_launch_us": round(ms * 1e3, 2),
        "Gpix/s": round(w * h / (ms * 1e-3) / 1e9, 1), "algorithmic_bytes": alg,
        "achieved_GBps": round(alg / (ms * 1e-3) / 1e9, 1), "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    lib.gmat_device_sync()
    mem.free()
    return res


def chained_forms(lib, env, stream, geo, k3, pre_warm, branches, frames=LAUNCH_FRAMES):
    """convert-then-scale semantics (the reference GPU back-end's order of operations, the metric's literal arrows), two forms"""
    from gmat_amd.lib import PIX_FMT, planes, ints
    fz = Runner(lib, geo, stream, frames, 1, seed=2000, branches=branches)
    fwall, fms = timed(lib, env, None, fz, stream, k3, 2, 1, pre_warm)
    fn = k3 * frames
    fused_kernel = fz.kernel()
    fz.close()
    ch = Runner(lib, geo, stream, frames, 0, seed=2000, branches=1)
    cwall, cms = timed(lib, env, None, ch, stream, k3, 2, 1, pre_warm)
    n = k3 * frames
    ach_c = (geo.alg_convert + geo.alg_scale) * n / (cms * 1e-3) / 1e9
    # per-kernel timing: each kernel alone over the rotating frame set
    mem = DevMem(lib)
    rgb = [mem.alloc(geo.sh * ch.src_ls * 3) for _ in range(frames)]
    cc = lib.gmat_sws_getContext(geo.sw, geo.sh, PIX_FMT["nv12"], geo.sw, geo.sh, PIX_FMT["rgb24"], 0, None)
    sc = lib.gmat_sws_getContext(geo.sw, geo.sh, PIX_FMT["rgb24"], geo.dw, geo.dh, PIX_FMT["rgb24"], 4, None)
    lib.gmat_sws_setStream(cc, stream); lib.gmat_sws_setStream(sc, stream)
    state = {"i": 0}

    def k_conv():
        i = state["i"] = (state["i"] + 1) % frames
        b = ch.src[i]
        lib.gmat_sws_scale(cc, planes([b, b + ch.src_ls * geo.sh]), ints([ch.src_ls, ch.src_ls]), 0, geo.sh,
                           planes([rgb[i]]), ints([ch.src_ls * 3]))

    def k_scale():
        i = state["i"] = (state["i"] + 1) % frames
        lib.gmat_sws_scale(sc, planes([rgb[i]]), ints([ch.src_ls * 3]), 0, geo.sh,
                           planes([ch.dst[i]]), ints([ch.dst_ls]))

    t_conv = time_single_kernel(lib, env, k_conv, stream, 4 * frames)
    t_scale = time_single_kernel(lib, env, k_scale, stream, 4 * frames)
    res = {
        "semantics": "sws(NV12->RGB24, POINT) then sws(RGB24->RGB24, BICUBIC), bit-exact",
        "fused_kernel": {"kernel": fused_kernel, "value": round(fn * geo.px / fwall / 1e9, 3), "unit": "Gpix/s",
                         "achieved_GBps": round(geo.alg_fused * fn / (fms * 1e-3) / 1e9, 1),
                         "frac": round(geo.alg_fused * fn / (fms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "algorithmic_bytes_per_frame": geo.alg_fused},
        "two_kernels": {
            "value": round(n * geo.px / cwall / 1e9, 3), "unit": "Gpix/s",
            "achieved_GBps": round(ach_c, 1), "frac": round(ach_c / HBM_PEAK_GBS, 4),
            "algorithmic_bytes_per_frame": geo.alg_convert + geo.alg_scale,
            "kernels": {
                "yuv2rgb_kernel": {"avg_launch_us": round(t_conv * 1e3, 3),
                                   "achieved_GBps": round(geo.alg_convert / (t_conv * 1e-3) / 1e9, 1),
                                   "frac": round(geo.alg_convert / (t_conv * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                lib.gmat_sws_lastKernel(sc).decode(): {
                    "avg_launch_us": round(t_scale * 1e3, 3),
                    "achieved_GBps": round(geo.alg_scale / (t_scale * 1e-3) / 1e9, 1),
                    "frac": round(geo.alg_scale / (t_scale * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}}}
    lib.gmat_sws_freeContext(cc); lib.gmat_sws_freeContext(sc)
    ch.close()
    mem.free()
    return res


def c_harness(dry):
    """Per-call numbers from tools/bin/x2bench (C++ over the C ABI, built by __graft_entry__.build()): one frame per
    gmat_sws_scale / filter call, timed with HIP events on the launch stream.  The Python loops above cannot issue a call
    every few microseconds (the ctypes call costs about as much as the kernel), so their one-frame-per-call figures
    are upper bounds of the call rate of THIS harness, not of the library; these are the library's."""
    exe = os.path.join(ROOT, "tools", "bin", "x2bench")
    if dry or not os.path.exists(exe):
        return {"skipped": "dry run" if dry else "tools/bin/x2bench not built (run __graft_entry__.build())"}
    env = dict(os.environ, X2BENCH_JSON="1", X2BENCH_VERIFY="0")
    res = {"harness": "tools/x2bench.cpp, HIP events on the launch stream, best of 3 x 50 launches, 2 frame sets rotated"}
    for nf, key in ((1, "one frame per call"), (32, "32 frames per launch")):
        rows = []
        try:
            r = subprocess.run([exe, str(nf), "50", ""], env=env, capture_output=True, text=True, timeout=300)
            for line in r.stdout.splitlines():
                if line.startswith("{"):
                    rows.append(json.loads(line))
        except Exception as e:                               # noqa: BLE001 - a bench leg must not take the headline down
            rows.append({"error": repr(e)})
        res[key] = rows
    # a per-frame caller that alternates n streams of its own (one library call per frame, nothing added): the launch boundary of one
    # frame hides behind the kernel of the other (profiles/r03zz_per_call_caller_streams.txt)
    for n in (2, 3):
        rows = []
        try:
            r = subprocess.run([exe, "1", "100", ""], env=dict(env, X2BENCH_CALL_STREAMS=str(n)), capture_output=True, text=True, timeout=300)
            rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        except Exception as e:                               # noqa: BLE001
            rows = [{"error": repr(e)}]
        res["one frame per call, %d caller streams" % n] = rows
    # round 5: 16-bit samples and packed-RGB sources away from 2 : 1 (the band walker of k_scale_yuvg16.hip; the lines form / tiled kernels before)
    for flt, key in (("deep:", "10- / 16-bit sources and 10-bit destinations, 32 frames per launch"), ("rgbsrc: rgb24", "packed-RGB sources away from 2:1, 32 frames per launch"),
                     ("rgbsrc: bgra", "RGBA sources (read as they are; alpha scaled as a fourth line), 32 frames per call")):
        try:
            r = subprocess.run([exe, "32", "10", flt], env=dict(env, X2BENCH_SETS="4"), capture_output=True, text=True, timeout=300)
            res[key] = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        except Exception as e:                               # noqa: BLE001
            res[key] = [{"error": repr(e)}]
    try:
        r = subprocess.run([exe, "1", "50", "op: "], env=env, capture_output=True, text=True, timeout=300)
        res["filters, one 4K frame per launch"] = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and "op: " in l]
    except Exception as e:                                   # noqa: BLE001
        res["filters, one 4K frame per launch"] = [{"error": repr(e)}]
    # the same calls with two / three frames in flight (launches round-robin over that many streams, as `value` has its launch
    # sets): a launch boundary costs 1.6 us + the ramp of a 50 MB kernel, which a second stream hides (profiles/r03r_*)
    # the queued filter form (option batch / gmat_op_batch): N frames through ONE launch, grid dimension = frame
    for nf in (4, 16):
        key = "filters, %d 4K frames per launch" % nf
        try:
            r = subprocess.run([exe, str(nf), "30", "op: "], env=env, capture_output=True, text=True, timeout=300)
            res[key] = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and "op: " in l]
        except Exception as e:                               # noqa: BLE001
            res[key] = [{"error": repr(e)}]
    for n in (2, 3):
        key = "filters, one 4K frame per launch, %d frames in flight" % n
        try:
            r = subprocess.run([exe, "1", "50", "op: "], env=dict(env, X2BENCH_OP_STREAMS=str(n)), capture_output=True, text=True, timeout=300)
            res[key] = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and "op: " in l]
        except Exception as e:                               # noqa: BLE001
            res[key] = [{"error": repr(e)}]
    return res


def host_pipeline(lib, geo, dist, world, device, dry, nframes=96, depth=4):
    """PCIe-inclusive rate: pinned host NV12 in -> HBM -> scale -> HBM -> pinned host RGB24 out, copies on their own
    streams overlapping the kernels (gmat_pipeline_*, one pipeline per rank, all ranks at once).  Never `value`."""
    import numpy as np
    from gmat_amd.pipeline import FramePipeline
    p = FramePipeline(lib, geo.sw, geo.sh, "nv12", geo.dw, geo.dh, "rgb24", depth=depth, device=device)
    for k in range(depth):                                   # synthetic content in every ring slot
        f = p.host_input(k)
        for pl, rows in ((0, geo.sh), (1, geo.sh // 2)):
            v = np.ctypeslib.as_array(C.cast(f.data[pl], C.POINTER(C.c_uint8)), (rows, f.linesize[pl]))
            v[...] = np.random.default_rng(k * 2 + pl).integers(0, 256, v.shape, dtype=np.uint8)
    for _ in range(2 * depth):
        p.submit()
    p.drain()
    dist.barrier(world)
    t0 = time.perf_counter()
    for _ in range(nframes):
        p.submit()
    p.drain()
    dt = time.perf_counter() - t0
    dt = dist.max_over_ranks(dt, world, device=ctl_device[0])
    # what a caller with PAGEABLE frames adds (hwupload's av_image_copy into the pinned ring, integration/vf_hwupload_hip.c:126):
    # one 4K NV12 frame copied into a ring slot by this rank's (NUMA-bound) thread — a per-thread ceiling beside the PCIe one
    fill = None
    try:
        f = p.host_input(0)
        n0, n1 = f.linesize[0] * geo.sh, f.linesize[1] * (geo.sh // 2)
        srcbuf = np.random.default_rng(99).integers(0, 256, n0 + n1, dtype=np.uint8)
        reps = 2 if dry else 12
        C.memmove(f.data[0], srcbuf.ctypes.data, n0); C.memmove(f.data[1], srcbuf.ctypes.data + n0, n1)      # touch
        tf = time.perf_counter()
        for _ in range(reps):
            C.memmove(f.data[0], srcbuf.ctypes.data, n0); C.memmove(f.data[1], srcbuf.ctypes.data + n0, n1)
        tf = (time.perf_counter() - tf) / reps
        fill = {"ms_per_4k_nv12_frame": round(tf * 1e3, 3), "GBps": round((n0 + n1) / tf / 1e9, 2), "frames_per_s_per_thread": round(1.0 / tf, 1),
                "Gpix_per_s_per_thread": round(geo.px / tf / 1e9, 3)}
    except Exception as e:                                   # noqa: BLE001 - a bench leg must not take the headline down
        fill = {"error": repr(e)}
    p.close()
    return {"value": round(world * nframes * geo.px / dt / 1e9, 3), "unit": "Gpix/s", "frames_per_rank": nframes, "ranks": world,
            "pageable_fill_one_thread": fill,
            "ring_depth": depth, "pcie_GBps_in_per_gpu": round(nframes * geo.nv12 / dt / 1e9, 2),
            "pcie_GBps_out_per_gpu": round(nframes * geo.rgb_dst / dt / 1e9, 2),
            "note": "pinned host frames, upload / compute / download on three streams chained by events (C ABI "
                    "gmat_pipeline_*), one pipeline per GPU; bounded by PCIe Gen5 x16 (63 GB/s spec), not by the kernels"}


def cpu_port(geo, nframes):
    """The C oracle (a port of libswscale's arithmetic) on the host cores: ONE nv12 -> rgb24 bicubic context (the
    headline's semantics), output rows sliced over the cores."""
    import numpy as np
    import harness
    from concurrent.futures import ThreadPoolExecutor
    from gmat_amd.lib import PIX_FMT, SWS, planes, ints
    path = os.path.join(ROOT, "oracle", "liborc.so")
    if not os.path.exists(path):
        return None, None
    orc = harness.load_oracle(path)
    L = orc.L
    # N = ALL host cores (round 3 stopped at 64, and the rank is bound to its GPU's NUMA node for the GPU
    # legs: the binding is lifted for this one)
    try:
        os.sched_setaffinity(0, range(os.cpu_count() or 1))
    except OSError:
        pass
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    y = orc.lcg((geo.sh, geo.sw), 7)
    uv = orc.lcg((geo.sh // 2, geo.sw), 8)
    out = np.empty((geo.dh, geo.dw * 3), np.uint8)
    ctx = L.orc_sws_create(geo.sw, geo.sh, PIX_FMT["nv12"], geo.dw, geo.dh, PIX_FMT["rgb24"], SWS["bicubic"], None)
    band = (geo.dh + cores - 1) // cores

    def scale(i):
        y0, y1 = i * band, min(geo.dh, (i + 1) * band)
        if y0 < y1:
            L.orc_sws_scale_rows(ctx, planes([y.ctypes.data, uv.ctypes.data]), ints([geo.sw, geo.sw]),
                                 planes([out.ctypes.data]), ints([geo.dw * 3]), y0, y1)

    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(scale, range(cores)))                                            # warm
        t0 = time.perf_counter()
        for _ in range(nframes):
            list(ex.map(scale, range(cores)))
        dt = time.perf_counter() - t0
    L.orc_sws_free(ctx)
    head = {"value": round(nframes * geo.px / dt / 1e9, 4), "unit": "Gpix/s", "cores": cores, "kind": "port",
            "note": "scalar C port, NOT stock libswscale (no SIMD): 2-3x below a production host",
            "sample": f"{nframes} frames {geo.sw}x{geo.sh} nv12 -> {geo.dw}x{geo.dh} rgb24 bicubic, one context, the C oracle "
                      f"row-sliced over all {cores} host cores, {dt:.2f} s"}
    # BASELINE configs[0]: 1080p yuv420p -> rgb24 on ONE host thread, libswscale's unscaled fast path (yuv2rgb.c:346-374)
    w, h = geo.sw // 2, geo.sh // 2
    src = [orc.lcg((h, w), 1), orc.lcg((h // 2, w // 2), 2), orc.lcg((h // 2, w // 2), 3)]
    orc.yuv2rgb(src, w, h, "yuv420p", "rgb24")
    n0 = max(4, nframes // 8)
    t0 = time.perf_counter()
    for _ in range(n0):
        orc.yuv2rgb(src, w, h, "yuv420p", "rgb24")
    d0 = time.perf_counter() - t0
    cfg0 = {"value": round(n0 * w * h / d0 / 1e9, 4), "unit": "Gpix/s", "cores": 1, "kind": "port",
            "sample": f"BASELINE configs[0]: {n0} frames {w}x{h} yuv420p -> rgb24, one stream, one thread, the oracle's "
                      f"restatement of yuv2rgb_c_24_rgb (libswscale/yuv2rgb.c:346-374), {d0:.2f} s"}
    return head, cfg0


def cpu_reference(geo, nframes):
    """Stock libswscale, when the box has one (dlopen("libswscale.so.*")): one context per thread on
    independent frames.  None when no library loads — this image ships none, so the port above is the baseline."""
    import ctypes.util
    import glob
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    cands = [ctypes.util.find_library("swscale")] + sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libswscale.so.*")) + \
        sorted(glob.glob("/usr/local/lib/libswscale.so.*")) + sorted(glob.glob("/usr/lib64/libswscale.so.*"))
    lib = None
    for c in cands:
        if not c:
            continue
        try:
            lib = C.CDLL(c)
            break
        except OSError:
            continue
    if lib is None:
        return None
    lib.sws_getContext.restype = C.c_void_p
    lib.sws_getContext.argtypes = [C.c_int] * 7 + [C.c_void_p] * 3
    lib.sws_scale.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int,
                              C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    lib.sws_freeContext.argtypes = [C.c_void_p]
    from gmat_amd.lib import PIX_FMT, planes, ints
    cores = min(os.cpu_count() or 1, 64)
    per = max(1, nframes // cores)

    def work(i):
        rng = np.random.default_rng(i)
        y = rng.integers(0, 256, (geo.sh, geo.sw), dtype=np.uint8)
        uv = rng.integers(0, 256, (geo.sh // 2, geo.sw), dtype=np.uint8)
        out = np.empty((geo.dh, geo.dw * 3), np.uint8)
        ctx = lib.sws_getContext(geo.sw, geo.sh, PIX_FMT["nv12"], geo.dw, geo.dh, PIX_FMT["rgb24"], 4, None, None, None)
        if not ctx:
            return 0
        for _ in range(per):
            lib.sws_scale(ctx, planes([y.ctypes.data, uv.ctypes.data]), ints([geo.sw, geo.sw]), 0, geo.sh,
                          planes([out.ctypes.data]), ints([geo.dw * 3]))
        lib.sws_freeContext(ctx)
        return per

    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        done = sum(ex.map(work, range(cores)))
        dt = time.perf_counter() - t0
    if not done:
        return None
    return {"value": round(done * geo.px / dt / 1e9, 4), "unit": "Gpix/s", "cores": cores, "kind": "reference",
            "sample": f"{done} frames {geo.sw}x{geo.sh} nv12 -> {geo.dw}x{geo.dh} rgb24 SWS_BICUBIC through the box's stock "
                      f"libswscale (sws_scale), one context per thread on {cores} threads, {dt:.2f} s (context set-up included)"}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(relaunch_as_ranks(a.gpus))
    if not a.dry:
        import torch      # noqa: F401  BEFORE the library: both link libamdhip64, and torch must bring up the one it was built with
    import gmat_amd
    from gmat_amd import dist as gdist
    rank, local, world = gdist.env_rank()
    if a.dry:
        from gmat_amd.lib import load
        emu = os.path.join(ROOT, "tests", "hipemu", "build", "libgmat_hip_emu.so")
        if not os.path.exists(emu):
            subprocess.run(["make", "-C", os.path.dirname(os.path.dirname(emu))], check=True, capture_output=True)
        lib = load(emu)                                         # test infrastructure: kernel sources on CPU fibers
        geo = Geo(128, 32, 64, 16)
        a.frames, a.steps, a.warmup, a.passes = min(a.frames, 4), min(a.steps, 2), min(a.warmup, 1), min(a.passes, 2)
        a.no_detail = a.no_pmc = True
    else:
        lib = gmat_amd.load()                                   # raises if the HIP library is missing
        if lib.gmat_device_count() < 1:
            raise SystemExit("bench.py: no HIP device visible (use --dry for the CPU plumbing check)")
        geo = Geo(3840, 2160, 1920, 1080)
        a.frames = max(LAUNCH_FRAMES, a.frames // LAUNCH_FRAMES * LAUNCH_FRAMES)
    # --same-device: the 8-GPU launch rehearsed on one GPU — every rank drives device 0
    dev = local % max(1, lib.gmat_device_count()) if a.dry else 0 if a.same_device else local      # (the emulated build has two devices)
    env = Env(a.dry, dev)
    local_device[0] = dev
    if lib.gmat_set_device(dev) != 0:
        raise SystemExit("bench.py: gmat_set_device(%d) failed" % dev)
    # each GPU gets its own host thread and pinned staging ring — bind this rank to the host cores of its
    # GPU's NUMA node BEFORE any pinned allocation (first touch decides where the ring lives)
    numa = {"node": int(lib.gmat_device_numa_node(dev)), "cpus_bound": int(lib.gmat_bind_thread_to_device(dev))}
    # control plane only (a barrier and one MAX): RCCL, and gloo when RCCL cannot be brought up — it must never be what takes
    # configs[4] down (RCCL refuses two ranks on one device: --same-device asks for gloo outright)
    gdist.init("gloo" if (a.dry or a.same_device) else "nccl", fallback="gloo")
    backend = gdist.backend()
    ctl_device[0] = "cuda" if backend == "nccl" else "cpu"
    dist = gdist
    stream = C.c_void_p()
    lib.gmat_stream_create(C.byref(stream))
    pre_warm = 0 if a.dry else PRE_WARM_MS
    branches = a.branches

    # ---- headline: one libswscale-semantics context (mode 2); every launch set spread over `branches` streams
    head = Runner(lib, geo, stream, a.frames, 2, seed=1000 + rank, branches=branches, passes=a.passes)
    wall, dev_ms = timed(lib, env, dist, head, stream, a.steps, a.warmup, world, pre_warm)
    nframes = a.steps * head.frames_per_step()
    gpix = world * nframes * geo.px / wall / 1e9
    kname = head.kernel()
    gdev = ctl_device[0]
    per_rank_wall = gdist.gather_floats(wall_local[0], world, device=gdev)
    per_rank_dev = [int(v) for v in gdist.gather_floats(dev, world, device=gdev)]
    per_rank_numa = [int(v) for v in gdist.gather_floats(numa["node"], world, device=gdev)]
    per_rank_cpus = [int(v) for v in gdist.gather_floats(numa["cpus_bound"], world, device=gdev)]
    # ---- roofline of the dominant kernel: the same frames, launches strictly back to back on ONE stream, HIP events
    ser = Runner(lib, geo, stream, a.frames, 2, seed=0, branches=1, share=head, passes=a.passes)
    _, ser_ms = timed(lib, env, None, ser, stream, a.steps, a.warmup, 1, pre_warm)
    fpl = ser.frames_per_launch()         # frames one launch carries (grid.y): <= 32
    nlaunch = a.steps * ser.launches_per_step()
    # the distribution behind the average: every launch between its own pair of HIP events, right after the timed run (warm clocks)
    pl_ms = sorted(ser.per_launch_ms(8 if a.dry else 512))
    ser.close()
    ser_ms = max(ser_ms, 1e-6); dev_ms = max(dev_ms, 1e-6)
    ach = geo.alg_fused * nframes / (ser_ms * 1e-3) / 1e9
    ach_ovl = geo.alg_fused * nframes / (dev_ms * 1e-3) / 1e9
    traffic, traffic_src = None, None
    if rank == 0 and world == 1 and not a.no_pmc:
        lt = live_traffic(kname, fpl)
        if lt:
            traffic, traffic_src = lt["bytes_per_launch"], "live rocprofv3 --pmc: FETCH_SIZE x2 | WRITE_SIZE, separate passes"
            detail("traffic_live", lt)
    if traffic is None:
        traffic = committed_traffic(kname, fpl)
        traffic_src = "committed pass profiles/r03_traffic.json" if traffic else None
    out = {
        "metric": "Gpix/s (and % HBM roofline) for 4K nv12->rgb24->1080p bicubic at 1/2/4/8 GPUs",
        "value": round(gpix, 3), "unit": "Gpix/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(wall / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"{geo.sw}x{geo.sh} nv12 -> {geo.dw}x{geo.dh} rgb24 bicubic (BASELINE configs[2]), device-resident, "
                               f"= ONE libswscale context bit for bit; {head.per_launch}-frame launch sets over {branches} streams",
                   "frames_per_step": head.frames_per_step(), "passes_per_step": a.passes, "distinct_frame_pairs": a.frames,
                   "launch_sets_per_step": head.launches_per_step(), "frames_per_launch_set": head.per_launch,
                   "streams": branches, "working_set_MB": round(a.frames * (geo.nv12 + geo.rgb_dst) / 1e6),
                   "pre_warm_ms": pre_warm, "kernel": kname, "timed_region_s": round(wall, 4),
                   "parallelism": f"{world} process(es), one per GPU, independent streams, no collective; control plane {backend or 'none'}"
                                  + (", ALL RANKS ON DEVICE 0 (rehearsal)" if a.same_device else "")},
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "frames_per_launch": fpl, "algorithmic_bytes_per_launch": geo.alg_fused * fpl,
                     "avg_launch_us": round(ser_ms * 1e3 / nlaunch, 3), "launches_timed": nlaunch,
                     "timing": "HIP events, ONE stream, launches back to back",
                     "frac_p50": round(geo.alg_fused * fpl / (pl_ms[len(pl_ms) // 2] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "frac_min": round(geo.alg_fused * fpl / (pl_ms[-1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "frac_max": round(geo.alg_fused * fpl / (pl_ms[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "per_launch_events": len(pl_ms),
                     "frac_overlapped": round(ach_ovl / HBM_PEAK_GBS, 4)},
    }
    head.close()
    if a.dry:
        out["dry_run"] = True
        out["data"] = "synthetic (DRY RUN on the CPU-emulated library: plumbing only, not a measurement)"
    if rank == 0:
        detail("per_rank", {"wall_s": [round(w, 6) for w in per_rank_wall], "min": round(min(per_rank_wall), 6),
                            "max": round(max(per_rank_wall), 6), "device": per_rank_dev, "numa_node": per_rank_numa,
                            "cpus_bound": per_rank_cpus, "numa_rank0": numa})

    if rank == 0 and not a.no_detail:
        detail("chained", chained_forms(lib, env, stream, geo, max(3, a.steps // 3), pre_warm, branches))
        detail("other_configs", other_configs(lib, env, stream, geo))
        detail("c_harness", c_harness(a.dry))

    if not a.no_pipeline:
        hp = host_pipeline(lib, geo, dist, world, dev, a.dry, nframes=8 if a.dry else 96)
        if rank == 0:
            detail("host_pipeline", hp)
            out["host_pipeline_Gpix_s"] = hp["value"]
    gdist.finalize(world)                                   # the other ranks are done; rank 0 goes on to the CPU legs
    if rank == 0:
        out["cpu_baseline"] = None
        if not a.no_cpu:
            port, cfg0 = cpu_port(geo, 4 if a.dry else a.cpu_frames)
            ref = cpu_reference(geo, 4 if a.dry else a.cpu_frames)
            out["cpu_baseline"] = ref or port
            if ref:
                detail("cpu_baseline_port", port)
            detail("cpu_configs0", cfg0)
        out["detail"] = "BENCH_DETAIL lines above + bench_detail.json"
        try:
            ddir = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else ROOT
            with open(os.path.join(ddir, "bench_detail.json"), "w") as f:
                json.dump(dict(DETAIL, headline=out), f, indent=1)
        except OSError:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
