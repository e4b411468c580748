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


def test_filter_follows_its_links_device_not_the_current_one():
    """a scale_hip configured for frames of device 1 while device 0 is current must create its scaler (tables,
    output pool) on device 1 and make that device current around every frame — the reference pushes the frames' CUDA context at
    the top of config_props and filter_frame (vf_scale_cuda.c:292-294, :553).  Both the per-frame and the queued batch=K entry
    points; the batched launch refuses a foreign device instead of launching with another GPU's tables."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gmat_amd.lib import load, PIX_FMT, GmatFrame, planes, ints
    emu = os.path.join(ROOT, "tests", "hipemu", "build", "libgmat_hip_emu.so")
    if not os.path.exists(emu):
        pytest.skip("emulated build missing")
    lib = load(emu)
    w, h = 64, 16
    assert lib.gmat_set_device(1) == 0
    fc1 = lib.gmat_hwframe_ctx_create(1, PIX_FMT["nv12"], w, h, 4)
    assert lib.gmat_set_device(0) == 0
    fc0 = lib.gmat_hwframe_ctx_create(0, PIX_FMT["nv12"], w, h, 4)
    assert fc0 and fc1

    def make(batch):
        f = lib.gmat_filter_alloc(b"scale_hip")
        for k, v in (("w", "iw/2"), ("h", "ih/2"), ("format", "rgb24"), ("batch", str(batch))):
            assert lib.gmat_filter_set_option(f, k.encode(), v.encode()) == 0
        assert lib.gmat_filter_init(f) == 0
        return f

    # device 0 is current; the link is on device 1
    f = make(1)
    assert lib.gmat_filter_config_props(f, fc1, None) == 0
    assert lib.gmat_set_device(0) == 0
    for _ in range(2):
        fr = lib.gmat_frame_alloc()
        assert lib.gmat_hwframe_get_buffer(fc1, fr) == 0
        out = C.POINTER(GmatFrame)()
        assert lib.gmat_set_device(0) == 0                      # some other filter of the graph made device 0 current meanwhile
        assert lib.gmat_filter_frame(f, fr, C.byref(out)) == 0
        lib.gmat_frame_free(C.byref(out))
    lib.gmat_filter_free(f)
    # the queued form: batch = 2 launches ONE kernel for two frames
    f = make(2)
    assert lib.gmat_set_device(0) == 0
    assert lib.gmat_filter_config_props(f, fc1, None) == 0
    for i in range(4):
        fr = lib.gmat_frame_alloc()
        assert lib.gmat_hwframe_get_buffer(fc1, fr) == 0
        assert lib.gmat_set_device(0) == 0
        assert lib.gmat_filter_send_frame(f, fr) == 0
        out = C.POINTER(GmatFrame)()
        while lib.gmat_filter_receive_frame(f, C.byref(out)) == 0:
            lib.gmat_frame_free(C.byref(out))
    lib.gmat_filter_free(f)
    # the batch entry point of a bare context checks the device like gmat_sws_scale does
    assert lib.gmat_set_device(0) == 0
    c = lib.gmat_sws_getContext(w, h, PIX_FMT["nv12"], w // 2, h // 2, PIX_FMT["rgb24"], 4, None)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import harness
    dev = harness.Dev(lib, "emu")
    srcs = [dev.planes_like("nv12", w, h, 16) for _ in range(2)]
    dsts = [dev.planes_like("rgb24", w // 2, h // 2, 16) for _ in range(2)]
    sp = (C.c_void_p * 8)(); dp = (C.c_void_p * 8)()
    for i in range(2):
        sp[4 * i], sp[4 * i + 1], dp[4 * i] = srcs[i][0].ptr, srcs[i][1].ptr, dsts[i][0].ptr
    streams = (C.c_void_p * 1)(None)
    vp = C.POINTER(C.c_void_p)
    call = lambda: lib.gmat_sws_scale_batch(c, 2, C.cast(sp, vp), ints([p.stride for p in srcs[0]]), C.cast(dp, vp),
                                            ints([dsts[0][0].stride]), C.cast(streams, vp), 1, 0)
    assert call() == 2
    assert lib.gmat_set_device(1) == 0
    assert call() < 0
    assert lib.gmat_set_device(0) == 0
    lib.gmat_sws_freeContext(c)
    for ps in srcs + dsts:
        for p in ps:
            p.free()
    lib.gmat_hwframe_ctx_free(fc0); lib.gmat_hwframe_ctx_free(fc1)


def test_filter_and_pool_calls_restore_the_callers_device():
    """the reference pushes AND pops the frames' context around every call (vf_scale_cuda.c:292-294,:553-571,
    hwcontext_cuda.c:231-276).  A thread on device 0 that runs a filter (and takes pool frames) of device 1 is still on device 0
    afterwards: its own bare context of device 0 keeps working with no gmat_set_device in between."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import harness
    from gmat_amd.lib import load, PIX_FMT, GmatFrame, planes, ints
    emu = os.path.join(ROOT, "tests", "hipemu", "build", "libgmat_hip_emu.so")
    if not os.path.exists(emu):
        pytest.skip("emulated build missing")
    lib = load(emu)
    dev = harness.Dev(lib, "emu")
    w, h = 64, 16
    assert lib.gmat_set_device(0) == 0
    c = lib.gmat_sws_getContext(w, h, PIX_FMT["nv12"], w // 2, h // 2, PIX_FMT["rgb24"], 4, None)
    src = dev.planes_like("nv12", w, h, 16)
    dst = dev.planes_like("rgb24", w // 2, h // 2, 16)
    args = (planes([p.ptr for p in src]), ints([p.stride for p in src]), 0, h, planes([p.ptr for p in dst]), ints([p.stride for p in dst]))
    fc1 = lib.gmat_hwframe_ctx_create(1, PIX_FMT["nv12"], w, h, 0)          # empty pool: get_buffer allocates on device 1
    assert fc1 and lib.gmat_sws_scale(c, *args) == h // 2                    # still on device 0 after the pool was created
    for batch in (1, 2):
        f = lib.gmat_filter_alloc(b"scale_hip")
        for k, v in (("w", "iw/2"), ("h", "ih/2"), ("format", "rgb24"), ("batch", str(batch))):
            assert lib.gmat_filter_set_option(f, k.encode(), v.encode()) == 0
        assert lib.gmat_filter_init(f) == 0 and lib.gmat_filter_config_props(f, fc1, None) == 0
        assert lib.gmat_sws_scale(c, *args) == h // 2
        for _ in range(2):
            fr = lib.gmat_frame_alloc()
            assert lib.gmat_hwframe_get_buffer(fc1, fr) == 0
            assert lib.gmat_sws_scale(c, *args) == h // 2
            assert lib.gmat_filter_send_frame(f, fr) == 0
            assert lib.gmat_sws_scale(c, *args) == h // 2
            out = C.POINTER(GmatFrame)()
            while lib.gmat_filter_receive_frame(f, C.byref(out)) == 0:
                lib.gmat_frame_free(C.byref(out))
        assert lib.gmat_filter_flush(f) >= 0
        lib.gmat_filter_free(f)
        assert lib.gmat_sws_scale(c, *args) == h // 2
    lib.gmat_sws_freeContext(c)
    lib.gmat_hwframe_ctx_free(fc1)
    for p in src + dst:
        p.free()
