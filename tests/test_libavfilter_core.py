"""The reference's REAL libavfilter + libavutil drive the filters and the frame pools (build container only).

Round 4's tests/c/filter_caller.c is a MINIATURE libavfilter around integration/vf_gmat_hip.c, and integration/vf_hwupload_hip.c called
av_hwdevice_ctx_create(AV_HWDEVICE_TYPE_CUDA) on a device type nothing in this repository implemented.  tools/build_ref_avfilter.sh builds
the reference's libavfilter.a + libavutil.a + libswscale.a out of tree (a temporary directory, portable C, the CPU filters the parity
target names) and links tests/c/avfilter_graph_caller.c with
    integration/hwcontext_hip.c     as libavutil's ONE open symbol, ff_hwcontext_type_cuda (hwcontext.c:36-38): device, pools, transfers
    integration/vf_gmat_hip.c       crop / flip / rotate / transpose / smooth / scale / format _hip
    integration/vf_hwupload_hip.c   upload through the pinned ring
    integration/swscale_hip_adapter.c + the CPU-emulated library   (libswscale's nine open symbols)
The caller builds two graphs with libavfilter's public API — buffer -> hwupload_hip -> GPU filters -> hwdownload -> buffersink, and
buffer -> the reference's own CPU filters -> buffersink — feeds both the same frames and compares every byte: the parity target itself
(scale / transpose / hflip / vflip / crop / rotate / convolution / median as the reference compiles them) is the oracle here, in one process,
through the reference's format negotiation, config_props order, frame pools, activate() scheduling and EOF handling.
Nothing of this travels to the GPU box (no reference tree there)."""
import os
import sys
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/ffmpeg-gpu"

OPEN = {"ff_hwcontext_type_cuda", "ff_sws_init_swscale_cuda", "ff_sws_free_swscale_cuda", "ff_swscale_cuda", "ff_yuv2rgb_init_tables_cuda",
        "yuv2rgb_cuda", "rgb2yuv_cuda", "yuv2yuv_cuda", "rgb24tobgr24_cuda", "rgb2rgb_init_cuda"}
G3 = "1 2 1 2 4 2 1 2 1"
CONV = "format=gbrp,convolution=%s:%s:%s:%s:0.0625:0.0625:0.0625:0.0625,format=rgb24" % (G3, G3, G3, G3)      # vf_convolution.c is planar-only; gbrp <-> rgb24 is lossless


@pytest.fixture(scope="session")
def graph_caller(tmp_path_factory):
    if not os.path.exists(os.path.join(REF, "configure")):
        pytest.skip("the reference tree is not present (GPU box)")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "hipemu")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = tmp_path_factory.mktemp("refavf")
    r = subprocess.run([os.path.join(ROOT, "tools", "build_ref_avfilter.sh"), str(out)], capture_output=True, text=True, timeout=1800)
    if r.returncode == 77:
        pytest.skip("reference tree not present")
    log = open(out / "make.log").read()[-2000:] if (out / "make.log").exists() else ""
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:] + log
    return str(out)


def test_the_reference_libraries_leave_open_exactly_what_this_repository_supplies(graph_caller):
    """nm -u over libavutil.a + libswscale.a + libavfilter.a: libswscale's nine and libavutil's hardware context type"""
    syms = set(open(os.path.join(graph_caller, "open_symbols.txt")).read().split())
    assert syms - {"ff_get_unscaled_swscale_cuda"} == OPEN


# (w, h, source format, frames, GPU chain (hwupload_hip in front, hwdownload + format behind are added), the reference's CPU chain)
CASES = [
    # libgpuscale behind scale_hip / format_hip
    (320, 180, "nv12", 3, "scale_hip=w=160:h=90:interp_algo=bicubic:format=rgb24", "scale=160:90:flags=bicubic", "rgb24"),           # the headline's shape
    (320, 180, "nv12", 3, "scale_hip=w=iw/2:h=ih/2:format=bgra", "scale=160:90:flags=bicubic", "bgra"),                               # default algorithm, expressions
    (320, 180, "nv12", 3, "scale_hip=w=212:h=120", "scale=212:120:flags=bicubic", "nv12"),                                            # a ladder step, any ratio
    (322, 182, "yuv420p", 3, "scale_hip=w=160:h=90:interp_algo=lanczos", "scale=160:90:flags=lanczos", "yuv420p"),                    # V-before-U pool frames
    (160, 90, "nv12", 2, "scale_hip=w=240:h=136:interp_algo=bilinear", "scale=240:136:flags=bilinear", "nv12"),                       # up
    (320, 180, "nv12", 2, "scale_hip=w=214:h=120:format=yuv420p", "scale=214:120:flags=bicubic", "yuv420p"),                          # between the chroma layouts
    (321, 181, "yuv420p", 2, "scale_hip=w=161:h=91:format=rgb24", "scale=161:91:flags=bicubic", "rgb24"),                             # odd sizes: ceil(h / 2) chroma rows in the pool
    (320, 180, "yuv420p", 2, "format_hip=pix_fmt=rgb24", "scale=flags=bicubic", "rgb24"),                                             # same size: yuv2rgb_c_24_rgb
    (320, 180, "yuv420p", 2, "scale_hip=format=bgra", "scale=flags=bicubic", "bgra"),
    (320, 180, "rgb24", 2, "scale_hip=w=160:h=90", "scale=160:90:flags=bicubic", "rgb24"),                                            # packed RGB at both ends
    (320, 180, "rgb24", 2, "scale_hip=format=nv12", "scale=flags=bicubic", "nv12"),                                                   # rgb -> yuv
    (320, 180, "rgb24", 2, "scale_hip=w=160:h=90:format=nv12", "scale=160:90:flags=bicubic", "nv12"),
    (320, 180, "nv12", 2, "format_hip=pix_fmt=yuv420p", "scale=flags=bicubic", "yuv420p"),                                            # re-layout
    (320, 180, "p010le", 2, "scale_hip=w=160:h=90", "scale=160:90:flags=bicubic", "p010le"),
    # sources deeper than 8 bits into 8-bit planar frames: ff_dither_8x8_128 (round 4: found by this test's fuzzer, tests/test_parity_dither.py)
    (640, 360, "p010le", 2, "scale_hip=w=320:h=180:format=nv12", "scale=320:180:flags=bicubic", "nv12"),                             # the 2:1 walker <10to8>
    (640, 360, "yuv420p10le", 2, "scale_hip=w=320:h=180:format=yuv420p:interp_algo=lanczos", "scale=320:180:flags=lanczos", "yuv420p"),
    (640, 360, "yuv420p10le", 2, "scale_hip=w=320:h=180:format=nv12", "scale=320:180:flags=bicubic", "nv12"),                        # ... and the other chroma layout
    (320, 180, "p010le", 2, "scale_hip=w=212:h=120:format=yuv420p", "scale=212:120:flags=bicubic", "yuv420p"),
    (320, 180, "p010le", 2, "scale_hip=format=nv12", "scale=flags=bicubic", "nv12"),                                                 # equal size, generic lines
    (320, 180, "rgba64le", 2, "scale_hip=w=160:h=90:format=nv12", "scale=160:90:flags=bicubic", "nv12"),
    (320, 180, "yuv444p16le", 2, "scale_hip=w=160:h=90:format=yuv420p", "scale=160:90:flags=bicubic", "yuv420p"),
    (320, 180, "yuv420p10le", 2, "scale_hip=format=yuv420p", "scale=flags=bicubic", "yuv420p"),                                      # equal size AND layout: planarCopyWrapper's own tables
    (321, 181, "yuv420p16le", 2, "scale_hip=format=yuv420p", "scale=flags=bicubic", "yuv420p"),
    (321, 181, "yuv444p16le", 2, "scale_hip=format=yuv444p", "scale=flags=bicubic", "yuv444p"),
    (320, 180, "nv12", 6, "scale_hip=w=160:h=90:format=rgb24:batch=4", "scale=160:90:flags=bicubic", "rgb24"),                        # activate(): a full queue, then EOF flushes two
    (320, 180, "nv12", 3, "scale_hip=w=160:h=90,scale_hip=w=80:h=44:format=rgb24", "scale=160:90:flags=bicubic,scale=80:44:flags=bicubic", "rgb24"),   # frames contexts from filter to filter
    # the pixel filters (cfg4's operations), against the CPU filters SURVEY.md 8a row 14 names
    (320, 180, "rgb24", 2, "rotate_hip=angle=90", "transpose=dir=clock", "rgb24"),
    (320, 180, "rgb24", 2, "rotate_hip=angle=-90", "transpose=dir=cclock", "rgb24"),
    (320, 180, "rgb24", 2, "rotate_hip=angle=180", "hflip,vflip", "rgb24"),
    (322, 178, "bgra", 2, "transpose_hip=dir=0", "transpose=dir=cclock_flip", "bgra"),
    (322, 178, "rgb24", 2, "transpose_hip=dir=3", "transpose=dir=clock_flip", "rgb24"),
    (320, 180, "rgb24", 2, "flip_hip=code=1", "hflip", "rgb24"),
    (320, 180, "rgb24", 2, "flip_hip=code=0", "vflip", "rgb24"),
    (320, 180, "rgba", 2, "flip_hip=code=-1", "hflip,vflip", "rgba"),
    (320, 180, "rgb24", 2, "crop_hip=w=200:h=100:x=31:y=17", "crop=200:100:31:17", "rgb24"),
    (320, 180, "rgb24", 2, "crop_hip=w=200:h=100", "crop=200:100", "rgb24"),                                                          # both centre by default
    (320, 180, "rgb24", 2, "smooth_hip", CONV, "rgb24"),
    (320, 180, "rgb24", 2, "smooth_hip=type=median", "format=gbrp,median=radius=1,format=rgb24", "rgb24"),
    (320, 180, "rgb24", 2, "smooth_hip=type=median:kw=5:kh=5", "format=gbrp,median=radius=2,format=rgb24", "rgb24"),
    (320, 180, "rgb24", 2, "rotate_hip=angle=17", "rotate=17*PI/180", "rgb24"),                                                       # vf_rotate.c's 16.16 bilinear walk
    (320, 180, "bgra", 2, "rotate_hip=angle=-33.5", "rotate=-33.5*PI/180", "bgra"),
    (320, 180, "rgb24", 2, "rotate_hip=angle=45:interp=nearest", "rotate=45*PI/180:bilinear=0", "rgb24"),
    # BASELINE configs[3] as a filter graph, one launch per frame and queued
    (320, 180, "rgb24", 3, "rotate_hip=angle=90,flip_hip=code=1,smooth_hip", "transpose=dir=clock,hflip," + CONV, "rgb24"),
    (320, 180, "rgb24", 5, "rotate_hip=angle=90:batch=2,flip_hip=code=1:batch=2,smooth_hip=batch=2", "transpose=dir=clock,hflip," + CONV, "rgb24"),
    # decode-side shape: scale then the RGB filters, in one graph
    (320, 180, "nv12", 3, "scale_hip=w=160:h=90:format=rgb24,rotate_hip=angle=90,smooth_hip", "scale=160:90:flags=bicubic,format=rgb24,transpose=dir=clock," + CONV, "rgb24"),
]


# frames that carry their own colour description (AVFrame.colorspace / color_range): format_cuda converts with in->colorspace (vf_format_cuda.c:184-217),
# `scale` with in_color_matrix=auto takes matrix and source range from the frame and writes limited range (vf_scale.c:793-824); (AVCOL_SPC, AVCOL_RANGE)
TAGGED = [
    (320, 180, "nv12", 2, "scale_hip=w=160:h=90:format=rgb24", "scale=160:90:flags=bicubic", "rgb24", 1, 1),              # BT.709, limited
    (320, 180, "nv12", 2, "scale_hip=w=160:h=90:format=rgb24", "scale=160:90:flags=bicubic", "rgb24", 1, 2),              # BT.709, full-range source
    (320, 180, "yuv420p", 2, "format_hip=pix_fmt=bgra", "scale=flags=bicubic", "bgra", 9, 1),                             # BT.2020
    (320, 180, "yuv420p", 2, "format_hip=pix_fmt=rgb24", "scale=flags=bicubic", "rgb24", 7, 2),                           # SMPTE 240M, full range
    (320, 180, "nv12", 2, "scale_hip=w=160:h=90", "scale=160:90:flags=bicubic", "nv12", 1, 2),                            # YUV -> YUV: full -> limited range conversion
    (320, 180, "rgb24", 2, "scale_hip=w=160:h=90:format=nv12", "scale=160:90:flags=bicubic", "nv12", 1, 1),               # RGB -> YUV: the tag's matrix on the way out
    (320, 180, "nv12", 5, "scale_hip=w=160:h=90:format=rgb24:batch=2", "scale=160:90:flags=bicubic", "rgb24", 1, 2),      # ... through the queue
    # the output frame says what its samples are — a full-range source leaves the first scale_hip as LIMITED range tagged so, and a second
    # scale_hip (or a converter) must not compress it again; the caller also compares every output frame's color_range / colorspace with `scale`'s
    (320, 180, "nv12", 2, "scale_hip=w=240:h=136,scale_hip=w=160:h=90", "scale=240:136:flags=bicubic,scale=160:90:flags=bicubic", "nv12", 1, 2),
    (320, 180, "nv12", 5, "scale_hip=w=240:h=136:batch=2,scale_hip=w=160:h=90:format=rgb24:batch=3", "scale=240:136:flags=bicubic,scale=160:90:flags=bicubic", "rgb24", 1, 2),
    (320, 180, "yuv420p", 2, "scale_hip=w=160:h=90,format_hip=pix_fmt=rgb24", "scale=160:90:flags=bicubic,scale=flags=bicubic", "rgb24", 5, 2),
]


@pytest.mark.parametrize("case", CASES + TAGGED, ids=lambda c: ("%s_%dx%d_%s%s" % (c[2], c[0], c[1], c[4], "_cs%d_r%d" % c[7:9] if len(c) > 7 else "")).replace(" ", "_")[:90])
def test_real_libavfilter_drives_the_gpu_filters(graph_caller, case):
    w, h, fmt, n, gpu, cpu, out = case[:7]
    gpu_chain = "hwupload_hip,%s,hwdownload,format=%s" % (gpu, out)
    cpu_chain = "%s,format=%s" % (cpu, out)
    tags = ["2026", str(case[7]), str(case[8])] if len(case) > 7 else []
    r = subprocess.run([os.path.join(graph_caller, "avfilter_graph_caller"), str(w), str(h), fmt, str(n), gpu_chain, cpu_chain] + tags,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2500:]
    assert "0 mismatching bytes" in r.stdout, r.stdout


@pytest.mark.parametrize("fmt", ["nv12", "yuv420p", "rgb24", "bgra", "p010le", "yuv444p", "yuv420p10le", "rgba64le"])
def test_libavfilters_own_hwupload_and_hwdownload_round_trip(graph_caller, fmt):
    """libavfilter's generic hwupload (vf_hwupload.c: the device from the application, av_hwframe_ctx_alloc / _init, av_hwframe_get_buffer,
    av_hwframe_transfer_data) and hwdownload over integration/hwcontext_hip.c alone: pool layout (YUV420P: V before U, chroma pitch = luma
    pitch / 2), both transfer directions, odd sizes"""
    for w, h in ((320, 180), (322, 182), (161, 91)):
        if fmt in ("nv12", "p010le") and (w & 1):
            continue
        r = subprocess.run([os.path.join(graph_caller, "avfilter_graph_caller"), str(w), str(h), fmt, "3",
                            "hwupload,hwdownload,format=%s" % fmt, "null"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "0 mismatching bytes" in r.stdout, (w, h, (r.stdout + r.stderr)[-2000:])


@pytest.mark.parametrize("fmt", ["nv12", "yuv420p", "rgb24", "p010le"])
def test_transfer_between_two_hardware_frames(graph_caller, fmt):
    """av_hwframe_transfer_data with a hardware frame on BOTH sides (hwcontext.c:448-467) — cuda_transfer_data copies device to device
    (hwcontext_cuda.c:239-252); hip_transfer_data answered ENOSYS.  Upload, pool-to-pool copy, download: the bytes that went in."""
    for w, h in ((320, 180), (162, 90)):
        r = subprocess.run([os.path.join(graph_caller, "avfilter_graph_caller"), str(w), str(h), fmt, "1", "d2d", "null"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and " 0 mismatching bytes" in r.stdout, (w, h, (r.stdout + r.stderr)[-2000:])


def test_error_paths_of_the_real_graph(graph_caller):
    """what libavfilter does with a refusal: an option out of range fails avfilter_init_str; a software frame on a GPU filter's input fails
    format negotiation (the filters list AV_PIX_FMT_CUDA only, as vf_crop_nvcv.c:90-98 does)"""
    exe = os.path.join(graph_caller, "avfilter_graph_caller")
    r = subprocess.run([exe, "320", "180", "rgb24", "1", "hwupload_hip,flip_hip=code=7,hwdownload,format=rgb24", "null"], capture_output=True, text=True)
    assert r.returncode != 0 and "avfilter_init_str" in r.stderr
    r = subprocess.run([exe, "320", "180", "rgb24", "1", "flip_hip=code=1,format=rgb24", "null"], capture_output=True, text=True)
    assert r.returncode != 0 and "avfilter_graph_config" in r.stderr
    r = subprocess.run([exe, "320", "180", "rgb24", "1", "hwupload_hip,smooth_hip=kw=4,hwdownload,format=rgb24", "null"], capture_output=True, text=True)
    assert r.returncode != 0


def test_reference_fuzz_smoke(tmp_path):
    """tests/fuzz/fuzz_ref_core.py over both harnesses (the real libswscale core + the real libavfilter): 60 random cases against the reference's own
    compiled CPU code — geometries incl. the exact ratios no reference vector holds, all eleven SWS algorithms and four flag bits, 8 / 10 / 16-bit and
    64-bit RGB formats, rotate / crop / median / smooth / transpose graphs.  Larger runs are recorded in profiles/r04_fuzz_ref_core.txt."""
    if not os.path.exists(os.path.join(REF, "configure")):
        pytest.skip("the reference tree is not present (GPU box)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_ref_core.py"), "60", "20261001", "--keep", str(tmp_path)],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "failures 0" in r.stdout, (r.stdout + r.stderr)[-3000:]
