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
    for name in SRC:
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
