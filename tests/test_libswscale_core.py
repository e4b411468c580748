"""The reference's REAL libswscale core drives the back-end (build container only).

tools/build_ref_swscale.sh builds the reference's libswscale.a + libavutil.a out of tree (a temporary directory; SURVEY.md §8c
recipe, portable C) and links tests/c/libswscale_core_caller.c with integration/swscale_hip_adapter.c + the CPU-emulated build of
the library IN PLACE OF the nine symbols the reference's libswscale/cuda objects define — no stub of any of them.  The caller uses
what an application uses: sws_getContext(... | SWS_HWACCEL_CUDA) (utils.c:2087-2123 -> sws_init_context_cuda :2026-2060),
sws_setCudaStream (swscale.c:1249), sws_scale (swscale.c:1204 -> scale_internal :1017 convert_unscaled / :1042-1044 ff_swscale_cuda),
sws_freeContext_cuda (utils.c:2507-2510) — and compares every byte with the SAME library's CPU path in the same process.

What it found (round 4): SwsContext.cspace, which round 3's adapter handed to gmat_sws_setColorspace as it was, is set by NOTHING in
the core (0 after sws_alloc_context) — the hand-written SwsContext of tests/c/adapter_caller.c had 5 in it.  With the real core every
SCALED yuv -> rgb context came out in the wrong matrix (4971 of 6144 bytes); the reference's get_constants maps unlisted values to
BT.601 and the adapter now does too.  Nothing of this travels to the GPU box (no reference tree there)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/ffmpeg-gpu"
BICUBIC, BILINEAR, POINT, LANCZOS = 4, 2, 0x10, 0x200

NINE = {"ff_sws_init_swscale_cuda", "ff_sws_free_swscale_cuda", "ff_swscale_cuda", "ff_yuv2rgb_init_tables_cuda",
        "yuv2rgb_cuda", "rgb2yuv_cuda", "yuv2yuv_cuda", "rgb24tobgr24_cuda", "rgb2rgb_init_cuda"}


@pytest.fixture(scope="session")
def core_caller(tmp_path_factory):
    if not os.path.exists(os.path.join(REF, "configure")):
        pytest.skip("the reference tree is not present (GPU box)")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "hipemu")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = tmp_path_factory.mktemp("refsws")
    r = subprocess.run([os.path.join(ROOT, "tools", "build_ref_swscale.sh"), str(out)], capture_output=True, text=True, timeout=1200)
    if r.returncode == 77:
        pytest.skip("reference tree not present")
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:] + open(out / "make.log").read()[-2000:] if (out / "make.log").exists() else r.stderr
    return str(out)


def test_the_core_leaves_exactly_the_nine_back_end_symbols_open(core_caller):
    """nm -u of the reference's libswscale.a: what the adapter and the library have to provide — and nothing else
    with a _cuda suffix (ff_get_unscaled_swscale_cuda is the core's own, swscale_unscaled.c:2014)"""
    syms = set(open(os.path.join(core_caller, "open_symbols.txt")).read().split())
    assert syms - {"ff_get_unscaled_swscale_cuda"} == NINE


# (srcW, srcH, srcFmt, dstW, dstH, dstFmt, flags of the GPU context, flags of the CPU context)
CASES = [
    (64, 32, "nv12", 64, 32, "rgb24", BICUBIC, POINT),         # same size: yuvToRgbWrapperCuda -> yuv2rgb_cuda; the CPU's nearest-chroma arithmetic
    (66, 34, "yuv420p", 66, 34, "bgra", BICUBIC, BICUBIC),     # planar source: the CPU takes yuv2rgb_c_32 itself (swscale_unscaled.c:2094-2100)
    (256, 144, "nv12", 128, 72, "rgb24", BICUBIC, BICUBIC),    # 4K-shaped -> half size: ff_sws_init_swscale_cuda / ff_swscale_cuda (the headline)
    (128, 64, "nv12", 64, 32, "rgb24", LANCZOS, LANCZOS),
    (100, 60, "nv12", 150, 90, "rgb24", BILINEAR, BILINEAR),   # up-scale, the tiled kernel
    (96, 48, "yuv420p", 40, 20, "bgra", BICUBIC, BICUBIC),
    (80, 40, "rgb24", 80, 40, "nv12", BICUBIC, BICUBIC),       # RgbToYuvWrapperCuda -> rgb2yuv_cuda
    (80, 40, "bgra", 80, 40, "yuv420p", BICUBIC, BICUBIC),
    (64, 32, "nv12", 64, 32, "yuv420p", BICUBIC, BICUBIC),     # YuvToYuvWrapperCuda -> yuv2yuv_cuda
    (64, 32, "yuv420p", 64, 32, "nv12", BICUBIC, BICUBIC),
    (64, 32, "nv12", 64, 32, "p010le", BICUBIC, BICUBIC),
    (64, 32, "rgb24", 64, 32, "bgr24", BICUBIC, BICUBIC),      # rgbToRgbWrapperCuda -> rgb24tobgr24_cuda
    (120, 48, "rgb24", 60, 24, "rgb24", BICUBIC, BICUBIC),     # scaled, packed RGB at both ends
    (128, 64, "nv12", 64, 32, "nv12", BICUBIC, BICUBIC),       # scale_cuda's job through libswscale
    (120, 48, "rgb24", 60, 24, "nv12", BICUBIC, BICUBIC),
    # round 4, found by tests/fuzz/fuzz_ref_core.py over this caller: sources deeper than 8 bits dither their 8-bit planar output (swscale.c:263-264)
    (256, 144, "p010le", 128, 72, "nv12", BICUBIC, BICUBIC),
    (234, 122, "yuv420p10le", 124, 70, "nv12", 1, 1),
    (216, 40, "p016le", 100, 22, "yuv420p", BICUBIC, BICUBIC),
    (216, 40, "rgba64le", 100, 22, "yuv420p", BICUBIC, BICUBIC),
    (216, 40, "yuv444p16le", 100, 22, "nv12", LANCZOS, LANCZOS),
    # ... and the three algorithms of scale_algorithms[] rounds 1-3 refused: SWS_X, SWS_BICUBLIN, SWS_SPLINE
    (256, 144, "nv12", 128, 72, "rgb24", 8, 8),
    (200, 120, "yuv420p", 68, 42, "bgra", 0x40, 0x40),
    (116, 32, "rgb24", 142, 44, "rgb24", 0x40, 0x40),            # RGB -> RGB: the chroma banks differ from the luma banks (the plane scaler, not scale_rgb_kernel)
    (160, 90, "nv12", 240, 136, "nv12", 0x400, 0x400),
    (164, 68, "rgba", 92, 50, "nv12", 0x400 | 0x40000, 0x400 | 0x40000),
    # round 6: SAME size with a deep source — the core's convert_unscaled calls yuv2yuv_cuda, whose reference switch has no arm for P010 / P016 sources and
    # writes nothing (yuv2yuv_cuda.cu:324-366); this library's symbol serves every pair of the list with what the CPU context computes (scale19_unit_kernel)
    (64, 18, "p010le", 64, 18, "nv12", BICUBIC, BICUBIC),
    (136, 22, "p016le", 136, 22, "p010le", BICUBIC, BICUBIC),
    (66, 10, "p010le", 66, 10, "yuv420p", BICUBIC, BICUBIC),
    (64, 18, "p016le", 64, 18, "nv12", BILINEAR, BILINEAR),
    (64, 18, "yuv420p", 64, 18, "p016le", BICUBIC, BICUBIC),
    (64, 18, "nv12", 64, 18, "rgba64le", BICUBIC, BICUBIC),      # ... and yuv2rgb_cuda's 64-bit output at equal size (scale19_unit64_kernel)
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s_%dx%d_to_%s_%dx%d_%x" % (c[2], c[0], c[1], c[5], c[3], c[4], c[6]))
def test_real_libswscale_core_drives_the_back_end(core_caller, case):
    sw, sh, sf, dw, dh, df, gflags, cflags = case
    r = subprocess.run([os.path.join(core_caller, "libswscale_core_caller"), str(sw), str(sh), sf, str(dw), str(dh), df, str(gflags), str(cflags)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-1500:]
    assert ": 0 of " in r.stdout
