"""The reference's own `ffmpeg` PROGRAM runs this repository's filters by name (build container only).

tools/build_ref_ffmpeg.sh configures the reference out of tree (fftools, libavfilter, libavdevice's lavfi, rawvideo, framecrc / framemd5), puts
integration/hwcontext_hip.c into its libavutil.a, integration/vf_gmat_hip.c + vf_hwupload_hip.c into its libavfilter.a (the eight filters appended to the
GENERATED filter_list.c), the adapter + the CPU-emulated library where libswscale's nine symbols are open, and links `ffmpeg`.  The command lines below are
what a GMAT user types (doc/FFmpeg_GPU_Filter_Implementation.md, the nvcv filter examples) with _hip for _nvcv / _cuda; each is run beside its CPU
counterpart and the framecrc / framemd5 streams — checksums of every output frame, made by the reference's own muxers — must be identical.
Nothing of this travels to the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/ffmpeg-gpu"
G3 = "1 2 1 2 4 2 1 2 1"
CONV = "format=gbrp,convolution=%s:%s:%s:%s:0.0625:0.0625:0.0625:0.0625,format=rgb24" % (G3, G3, G3, G3)


@pytest.fixture(scope="session")
def ffmpeg(tmp_path_factory):
    if not os.path.exists(os.path.join(REF, "configure")):
        pytest.skip("the reference tree is not present (GPU box)")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "hipemu")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = tmp_path_factory.mktemp("refffmpeg")
    r = subprocess.run([os.path.join(ROOT, "tools", "build_ref_ffmpeg.sh"), str(out)], capture_output=True, text=True, timeout=2400)
    if r.returncode == 77:
        pytest.skip("reference tree not present")
    logs = "".join(open(out / f).read()[-1500:] for f in ("make_libs.log", "make_ffmpeg.log") if (out / f).exists())
    assert r.returncode == 0 and (out / "ffmpeg").exists(), (r.stdout + r.stderr)[-2000:] + logs
    return str(out / "ffmpeg")


def _run(exe, pre, src, vf, muxer="framecrc", frames=4):
    cmd = [exe, "-hide_banner", "-loglevel", "error", "-filter_threads", "1"] + pre + ["-f", "lavfi", "-i", src, "-frames:v", str(frames), "-vf", vf, "-f", muxer, "-"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, " ".join(cmd) + "\n" + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l and not l.startswith("#")]
    assert len(lines) == frames, r.stdout
    return lines, [l for l in r.stdout.splitlines() if l.startswith("#dimensions") or l.startswith("#codec_id")]


def test_the_program_lists_the_filters(ffmpeg):
    out = subprocess.run([ffmpeg, "-hide_banner", "-filters"], capture_output=True, text=True).stdout
    for name in ("crop_hip", "flip_hip", "rotate_hip", "transpose_hip", "smooth_hip", "scale_hip", "format_hip", "hwupload_hip"):
        assert (" %s " % name) in out, name
    out = subprocess.run([ffmpeg, "-hide_banner", "-h", "filter=scale_hip"], capture_output=True, text=True).stdout
    for opt in ("interp_algo", "passthrough", "force_original_aspect_ratio", "force_divisible_by", "batch"):      # vf_scale_cuda.c:586-603 + the queue
        assert opt in out, opt


SRC = "testsrc2=size=640x360:rate=5"
# (lavfi source, GPU -vf, CPU -vf)
CASES = [
    (SRC, "format=nv12,hwupload_hip,scale_hip=w=320:h=180:format=rgb24,hwdownload,format=rgb24", "format=nv12,scale=320:180:flags=bicubic,format=rgb24"),     # configs[2]'s shape
    (SRC, "format=nv12,hwupload_hip,scale_hip=w=iw/2:h=ih/2:format=rgb24:batch=3,hwdownload,format=rgb24", "format=nv12,scale=320:180:flags=bicubic,format=rgb24"),
    (SRC, "format=nv12,hwupload_hip,scale_hip=w=426:h=240:interp_algo=lanczos,hwdownload,format=nv12", "format=nv12,scale=426:240:flags=lanczos,format=nv12"),  # a ladder step
    (SRC, "format=yuv420p,hwupload_hip,format_hip=pix_fmt=bgra,hwdownload,format=bgra", "format=yuv420p,scale=flags=bicubic,format=bgra"),
    (SRC, "format=rgb24,hwupload_hip,rotate_hip=angle=90,flip_hip=code=1,smooth_hip,hwdownload,format=rgb24", "format=rgb24,transpose=dir=clock,hflip," + CONV),   # configs[3]
    (SRC, "format=rgb24,hwupload_hip,rotate_hip=angle=90:batch=2,flip_hip=code=1:batch=2,smooth_hip=batch=2,hwdownload,format=rgb24", "format=rgb24,transpose=dir=clock,hflip," + CONV),
    (SRC, "format=rgb24,hwupload_hip,crop_hip=w=300:h=200:x=17:y=31,rotate_hip=angle=17,hwdownload,format=rgb24", "format=rgb24,crop=300:200:17:31,rotate=17*PI/180"),
    (SRC, "format=rgb24,hwupload_hip,smooth_hip=type=median:kw=5:kh=5,hwdownload,format=rgb24", "format=rgb24,format=gbrp,median=radius=2,format=rgb24"),
    ("yuvtestsrc=size=320x240:rate=5", "format=p010le,hwupload_hip,scale_hip=w=160:h=120:format=nv12,hwdownload,format=nv12", "format=p010le,scale=160:120:flags=bicubic,format=nv12"),
    ("rgbtestsrc=size=320x240:rate=5", "format=rgb24,hwupload_hip,scale_hip=w=160:h=120:format=nv12,hwdownload,format=nv12", "format=rgb24,scale=160:120:flags=bicubic,format=nv12"),
    # decode-side graph: scale on the GPU, then the RGB filters, one download
    (SRC, "format=nv12,hwupload_hip,scale_hip=w=320:h=180:format=rgb24,transpose_hip=dir=1,smooth_hip,hwdownload,format=rgb24",
          "format=nv12,scale=320:180:flags=bicubic,format=rgb24,transpose=dir=clock," + CONV),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c[1].split("hwupload_hip,")[1].split(",hwdownload")[0].replace(" ", "_")[:70])
@pytest.mark.parametrize("muxer", ["framecrc", "framemd5"])
def test_gpu_command_line_equals_cpu_command_line(ffmpeg, case, muxer):
    src, gpu, cpu = case
    g, gh = _run(ffmpeg, [], src, gpu, muxer)
    c, ch = _run(ffmpeg, [], src, cpu, muxer)
    assert gh == ch                                   # the stream's dimensions and codec as the muxer prints them
    assert g == c, "\n".join(g) + "\n--\n" + "\n".join(c)


def test_init_hw_device_and_libavfilters_own_hwupload(ffmpeg):
    """`-init_hw_device cuda=gpu:0 -filter_hw_device gpu` (fftools/ffmpeg_hw.c -> av_hwdevice_ctx_create -> integration/hwcontext_hip.c) with
    libavfilter's generic hwupload in front of scale_hip: the device of the command line, not one the filter made"""
    pre = ["-init_hw_device", "cuda=gpu:0", "-filter_hw_device", "gpu"]
    g, _ = _run(ffmpeg, pre, SRC, "format=nv12,hwupload,scale_hip=w=320:h=180,hwdownload,format=nv12")
    c, _ = _run(ffmpeg, [], SRC, "format=nv12,scale=320:180:flags=bicubic,format=nv12")
    assert g == c
    r = subprocess.run([ffmpeg, "-hide_banner", "-loglevel", "error", "-init_hw_device", "cuda=gpu:7", "-f", "lavfi", "-i", SRC, "-frames:v", "1", "-f", "null", "-"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "HIP device 7 requested" in r.stderr              # one (emulated) device: the program reports the refusal
