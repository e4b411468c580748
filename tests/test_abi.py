"""The C-ABI library loads and exports every symbol include/gmat_hip.h declares (no GPU needed)."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gmat_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = text.replace("#define GMAT_API", "")
    return sorted(set(re.findall(r"GMAT_API[^;(]*?\b(\w+)\s*\(", text)))


def test_header_and_binding_agree():
    from gmat_amd.lib import ABI_SYMBOLS
    assert sorted(ABI_SYMBOLS) == _declared_symbols()


def test_product_library_exports_every_declared_symbol():
    import ctypes
    from gmat_amd.lib import lib_path, load
    if not os.path.exists(lib_path()):
        import __graft_entry__ as g
        g.build()
    lib = load()                       # attaches all prototypes -> raises if a symbol is missing
    for name in _declared_symbols():
        assert getattr(lib, name) is not None
    assert b"gfx950" in lib.gmat_version()


def test_missing_library_fails_loudly(tmp_path):
    from gmat_amd.lib import load, GmatError
    with pytest.raises(GmatError, match="no CPU fallback"):
        load(str(tmp_path / "libgmat_hip.so"))


def test_product_never_references_the_oracle():
    """No source under gmat_amd/ may include, link or dlopen anything from oracle/."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gmat_amd")):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".so", ".o", ".pyc")):
                continue
            text = open(os.path.join(dirpath, f), errors="ignore").read()
            if re.search(r"liborc|orc_\w+\(|oracle/|#include\s+\"orc", text):
                bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_contexts_and_the_stateless_cache_are_per_device():
    """a context's tables live on the device current at creation (the reference makes the stream's device current around
    every call, hwcontext_cuda.c:395-434): using it under another device is refused, and the stateless entry points
    (rgb2yuv_cuda ...) keep one cached context per (geometry, device).  Runs on the emulator's two devices."""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import harness
    from gmat_amd.lib import load, PIX_FMT, planes, ints
    emu = os.path.join(ROOT, "tests", "hipemu", "build", "libgmat_hip_emu.so")
    if not os.path.exists(emu):
        pytest.skip("emulated build missing")
    lib = load(emu)
    assert lib.gmat_device_count() == 2
    dev = harness.Dev(lib, "emu")
    w, h = 64, 16
    assert lib.gmat_set_device(0) == 0
    c = lib.gmat_sws_getContext(w, h, PIX_FMT["nv12"], w // 2, h // 2, PIX_FMT["rgb24"], 4, None)
    src = dev.planes_like("nv12", w, h, 16)
    dst = dev.planes_like("rgb24", w // 2, h // 2, 16)
    args = (planes([p.ptr for p in src]), ints([p.stride for p in src]), 0, h, planes([p.ptr for p in dst]), ints([p.stride for p in dst]))
    assert lib.gmat_sws_scale(c, *args) == h // 2
    assert lib.gmat_set_device(1) == 0
    assert lib.gmat_sws_scale(c, *args) < 0                      # wrong device: refused, nothing launched
    # the stateless entry point builds a second context for device 1 instead of reusing device 0's
    rgb = dev.planes_like("rgb24", w, h, 16)
    yuv = dev.planes_like("nv12", w, h, 16)
    call = lambda: lib.rgb2yuv_cuda(planes([rgb[0].ptr]), ints([rgb[0].stride]), planes([p.ptr for p in yuv]),
                                    ints([p.stride for p in yuv]), w, h, PIX_FMT["rgb24"], PIX_FMT["nv12"], None)
    assert call() == 0
    assert lib.gmat_set_device(0) == 0
    assert call() == 0
    assert lib.gmat_sws_scale(c, *args) == h // 2
    lib.gmat_sws_freeContext(c)
    for p in src + dst + rgb + yuv:
        p.free()
