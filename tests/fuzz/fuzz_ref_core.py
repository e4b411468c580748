#!/usr/bin/env python3
"""Randomised differential test against the REFERENCE ITSELF (build container only: needs /root/reference).

The other fuzzers compare the library with oracle/ (the restatement, pinned by the reference's FATE checksums).  This one compares it with the
reference's own compiled code in one process, through the reference's own cores:
  part A  tools/build_ref_swscale.sh -> libswscale_core_caller: sws_getContext(... | SWS_HWACCEL_CUDA) + sws_scale over the adapter and the
          CPU-emulated library, against the SAME libswscale's CPU context — random geometries (exact 2:1 / 3:1 / 3:2 / 4:1 / 1:2 ratios, which no
          reference vector holds, a third of the time), every SWS algorithm, every format pair of swscale_cuda.c:34-44 the library serves;
  part B  tools/build_ref_avfilter.sh -> avfilter_graph_caller: hwupload_hip -> a GPU filter -> hwdownload against the reference's CPU filter
          (rotate at random angles / interpolations, crop, transpose, flips, median windows, the 3 x 3 smooth, scale_hip's four algorithms).
A context the library declines (sws_getContext fails) is counted, not failed.  usage: tests/fuzz/fuzz_ref_core.py [ncases] [seed] [--keep DIR]"""
import os, sys, random, subprocess, tempfile, collections

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1
keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
if not os.path.exists("/root/reference/ffmpeg-gpu/configure"):
    print("the reference tree is not present"); sys.exit(0)
base = keep or tempfile.mkdtemp(prefix="fuzzref")
dirs = {}
for name, script in (("sws", "build_ref_swscale.sh"), ("avf", "build_ref_avfilter.sh")):
    d = os.path.join(base, name)
    os.makedirs(d, exist_ok=True)
    r = subprocess.run([os.path.join(ROOT, "tools", script), d], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    dirs[name] = d
rng = random.Random(seed)
hist, fails = collections.Counter(), 0

ALGOS = {"fast_bilinear": 1, "bilinear": 2, "bicubic": 4, "x": 8, "point": 0x10, "area": 0x20, "bicublin": 0x40, "gauss": 0x80, "sinc": 0x100,
         "lanczos": 0x200, "spline": 0x400}
YUV8 = ["nv12", "yuv420p"]
RGB = ["rgb24", "bgr24", "rgba", "bgra"]
HI = ["p010le", "yuv420p10le", "p016le", "yuv420p16le", "yuv444p16le", "rgba64le", "bgra64le", "yuv444p"]     # (yuv444p rides here: 8 bits, no chroma subsampling)


def geometry(maxw=420, maxh=240):
    dw, dh = 2 * rng.randint(8, 100), 2 * rng.randint(4, 56)
    q = rng.random()
    if q < 0.35:
        num, den = rng.choice([(2, 1), (3, 1), (3, 2), (4, 1), (1, 2), (2, 1), (2, 1)])
        if den == 2 and (num * dw) % 4:
            dw += 2
        if den == 2 and (num * dh) % 4:
            dh += 2
        sw, sh = num * dw // den, num * dh // den
    else:
        rx = rng.choice([rng.uniform(0.4, 1.0), rng.uniform(1.0, 2.5), rng.uniform(2.5, 6.0)])
        ry = rx * rng.uniform(0.8, 1.25) if rng.random() < 0.7 else rng.uniform(0.4, 5.0)
        sw, sh = 2 * int(dw * rx / 2), 2 * int(dh * ry / 2)
    sw, sh = max(16, min(maxw * 2, sw)), max(8, min(maxh * 2, sh))
    if rng.random() < 0.15:
        sw, dw = sw | 1, dw | 1                                   # odd sizes (planar chroma rounds up)
    if rng.random() < 0.15:
        sh, dh = sh | 1, dh | 1
    return sw, sh, dw, dh


for case in range(n):
    if case % 2 == 0:
        # ---- part A: the libswscale core ----
        sw, sh, dw, dh = geometry()
        sf = rng.choice(YUV8 * 3 + RGB + HI)
        if sf in YUV8:
            df = rng.choice(RGB * 2 + YUV8 + ["p010le"])
        elif sf in RGB:
            df = rng.choice(RGB + YUV8 * 2)
        else:
            df = rng.choice([sf, sf, "nv12", "rgb24", "yuv420p", "bgra", rng.choice(HI)])
        if rng.random() < 0.15:
            df = rng.choice(HI)                                    # ... and the deep / 4:4:4 formats as destinations of everything
        if (sf in ("nv12", "p010le", "p016le") or df in ("nv12", "p010le", "p016le")) and ((sw | dw) & 1):
            sw, dw = sw & ~1, dw & ~1                              # semi-planar rows are pairs
        if (sf in RGB or sf in ("rgba64le", "bgra64le")) and (sw & 1):
            sw += 1                                               # rgb24ToUV_half_c reads pixel 2i + 1 of the last pair of an odd row: whatever follows it in memory
                                                                  # (input.c:849-866; oracle/orc_sws.c rgb8_chr states the clamp the library uses instead)
        algo = rng.choice(list(ALGOS) + ["bicubic", "bicubic", "bilinear", "lanczos"])
        if (sw, sh) == (dw, dh):
            continue
        fl = ALGOS[algo]
        for bit in (0x2000, 0x4000, 0x40000, 0x80000):             # SWS_FULL_CHR_H_INT / _INP, SWS_ACCURATE_RND, SWS_BITEXACT (swscale.h:84-90)
            if rng.random() < 0.2:
                fl |= bit
        cmd = [os.path.join(dirs["sws"], "libswscale_core_caller"), str(sw), str(sh), sf, str(dw), str(dh), df, str(fl), str(fl),
               str(rng.randint(1, 1 << 20))]
        what = "core %s %dx%d -> %s %dx%d %s flags 0x%x" % (sf, sw, sh, df, dw, dh, algo, fl)
        key = "core %s -> %s" % ("yuv" if sf in YUV8 else "rgb" if sf in RGB else "hi", "yuv" if df in YUV8 else "rgb" if df in RGB else "hi")
    else:
        # ---- part B: the filter graph ----
        w, h = rng.randint(24, 400), rng.randint(16, 240)
        fmt = rng.choice(RGB)
        q = rng.random()
        nfr = rng.choice([1, 2, 3])
        batch = rng.choice(["", "", ":batch=2"]) if nfr > 1 else ""
        if q < 0.3:
            ang = round(rng.uniform(-180, 180), rng.choice([0, 1, 3]))
            if ang % 90 == 0:
                ang += 0.5                                          # quarter turns are transposes here (output w x h swapped), not vf_rotate's same-size walk
            near = rng.random() < 0.3
            gpu, cpu = "rotate_hip=angle=%s%s%s" % (ang, ":interp=nearest" if near else "", batch), "rotate=%s*PI/180%s" % (ang, ":bilinear=0" if near else "")
            key = "graph rotate " + ("nearest" if near else "bilinear")
        elif q < 0.4:
            cw, ch = rng.randint(8, w), rng.randint(8, h)
            x, y = rng.randint(0, w - cw), rng.randint(0, h - ch)
            gpu, cpu = "crop_hip=w=%d:h=%d:x=%d:y=%d" % (cw, ch, x, y), "crop=%d:%d:%d:%d" % (cw, ch, x, y)
            key = "graph crop"
        elif q < 0.5:
            d = rng.randint(0, 3)
            gpu, cpu = "transpose_hip=dir=%d%s" % (d, batch), "transpose=dir=%d" % d
            key = "graph transpose"
        elif q < 0.58:
            code = rng.choice([0, 1, -1])
            gpu, cpu = "flip_hip=code=%d%s" % (code, batch), {0: "vflip", 1: "hflip", -1: "hflip,vflip"}[code]
            key = "graph flip"
        elif q < 0.72:
            r, rv = rng.randint(1, 4), rng.randint(1, 4)
            if rng.random() < 0.5:
                rv = r
            gpu = "smooth_hip=type=median:kw=%d:kh=%d" % (2 * r + 1, 2 * rv + 1)
            cpu = "format=gbrp%s,median=radius=%d:radiusV=%d,format=%s" % ("" if fmt in ("rgb24", "bgr24") else "", r, rv, fmt)
            if fmt in ("rgba", "bgra"):
                fmt = rng.choice(["rgb24", "bgr24"])                # gbrp drops alpha; the CPU median of gbrap would filter it too
                cpu = "format=gbrp,median=radius=%d:radiusV=%d,format=%s" % (r, rv, fmt)
            key = "graph median"
        elif q < 0.8:
            fmt = rng.choice(["rgb24", "bgr24"])
            g3 = "1 2 1 2 4 2 1 2 1"
            gpu, cpu = "smooth_hip" + batch.replace(":", "=", 1), "format=gbrp,convolution=%s:%s:%s:%s:0.0625:0.0625:0.0625:0.0625,format=%s" % (g3, g3, g3, g3, fmt)
            key = "graph smooth"
        else:
            sw, sh, dw, dh = geometry()
            w, h = sw, sh
            fmt = rng.choice(YUV8 * 2 + RGB)
            out = rng.choice(RGB + YUV8) if fmt in YUV8 else rng.choice([fmt, "nv12", "yuv420p"])
            if (fmt == "nv12" or out == "nv12") and ((w | dw) & 1):
                w, dw = w & ~1, dw & ~1
            if fmt in RGB and (w & 1):
                w += 1
            if (w, h) == (dw, dh):
                continue
            a = rng.choice(["bilinear", "bicubic", "lanczos", "nearest"])
            gpu = "scale_hip=w=%d:h=%d:interp_algo=%s:format=%s%s" % (dw, dh, a, out, batch)
            cpu = "scale=%d:%d:flags=%s" % (dw, dh, {"nearest": "neighbor"}.get(a, a))
            key = "graph scale"
            cmd = [os.path.join(dirs["avf"], "avfilter_graph_caller"), str(w), str(h), fmt, str(nfr), "hwupload_hip,%s,hwdownload,format=%s" % (gpu, out),
                   "%s,format=%s" % (cpu, out), str(rng.randint(1, 1 << 20))]
            gpu = None
        if gpu is not None:
            cmd = [os.path.join(dirs["avf"], "avfilter_graph_caller"), str(w), str(h), fmt, str(nfr), "hwupload_hip,%s,hwdownload,format=%s" % (gpu, fmt),
                   "%s,format=%s" % (cpu, fmt), str(rng.randint(1, 1 << 20))]
        what = " ".join(cmd[1:])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    if r.returncode == 0:
        hist[key] += 1
    elif "sws_getContext(SWS_HWACCEL_CUDA) failed" in r.stderr:
        hist["(declined) " + key] += 1
        if "--verbose" in sys.argv:
            print("declined:", what)
    else:
        fails += 1
        print("MISMATCH case", case, what, "->", (r.stdout + r.stderr).strip()[-400:].replace("\n", " | "))
for k, v in sorted(hist.items()):
    print("%6d  %s" % (v, k))
print("cases", n, "failures", fails)
sys.exit(1 if fails else 0)
