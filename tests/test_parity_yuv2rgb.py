"""HIP colour-conversion kernels vs the oracle, through the C ABI (bit-exact).

Runs twice: `emu` (same kernel sources on the CPU emulation, in the not-gpu suite) and
`hip` (the product library on a real MI355X, -m gpu).
"""
import ctypes as C

import numpy as np
import pytest

from harness import PIX_FMT, SWS, planes, ints, synth_planes

SIZES = [(64, 16), (260, 34), (130, 7), (3, 3), (1, 1), (517, 9)]


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("src_fmt", ["nv12", "yuv420p"])
@pytest.mark.parametrize("dst_fmt", ["rgb24", "bgr24", "rgba", "bgra"])
def test_yuv2rgb_bit_exact(dev, orc, w, h, src_fmt, dst_fmt):
    src = synth_planes(orc, src_fmt, w, h, seed=7)
    want = orc.yuv2rgb(src, w, h, src_fmt, dst_fmt)
    for align, extra in [(256, 0), (1, 0), (1, 5)]:          # aligned, tight, deliberately misaligned
        d_src = dev.upload_planes(src, align, extra)
        got, pads, kernel = dev.sws(d_src, w, h, src_fmt, w, h, dst_fmt, dst_align=align if align > 1 else 1,
                                    dst_extra=extra)
        assert kernel == "yuv2rgb_kernel"
        assert (got[0] == want).all(), f"mismatch at align={align} extra={extra}"
        assert (pads[0] == 0xCD).all(), "kernel wrote into the row padding"
        for p in d_src:
            p.free()


def test_yuv2rgb_colorspace_and_range(dev, orc):
    w, h = 96, 20
    src = synth_planes(orc, "nv12", w, h, seed=11)
    d_src = dev.upload_planes(src, 64)
    for cs, full in [(1, 0), (9, 0), (5, 1)]:
        want = orc.yuv2rgb(src, w, h, "nv12", "rgb24", cs, full)
        got, _, _ = dev.sws(d_src, w, h, "nv12", w, h, "rgb24", colorspace=(cs, full))
        assert (got[0] == want).all()


def test_reference_entry_point_yuv2rgb_cuda(dev, orc):
    # the plain-pointer symbol libswscale core binds (swscale_unscaled.c:1980)
    w, h = 128, 24
    src = synth_planes(orc, "nv12", w, h, seed=5)
    d_src = dev.upload_planes(src, 256)
    dst = dev.planes_like("rgb24", w, h, 256)
    r = dev.lib.yuv2rgb_cuda(planes([p.ptr for p in d_src]), ints([p.stride for p in d_src]),
                             planes([p.ptr for p in dst]), ints([p.stride for p in dst]), w, h,
                             PIX_FMT["nv12"], PIX_FMT["rgb24"], None)
    assert r == 0
    assert (dst[0].download() == orc.yuv2rgb(src, w, h, "nv12", "rgb24")).all()


@pytest.mark.parametrize("w,h", [(64, 8), (130, 6), (5, 3)])
def test_rgb24_bgr24_swap(dev, orc, w, h):
    src = synth_planes(orc, "rgb24", w, h, seed=3)
    want = np.zeros_like(src[0])
    orc.L.orc_rgb24_swap_rb(src[0].ctypes.data, src[0].strides[0], want.ctypes.data, want.strides[0], w, h)
    for align, extra in [(256, 0), (1, 1)]:
        d_src = dev.upload_planes(src, align, extra)
        got, pads, kernel = dev.sws(d_src, w, h, "rgb24", w, h, "bgr24", dst_align=align, dst_extra=extra)
        assert kernel == "swap_rb24_kernel"
        assert (got[0] == want).all() and (pads[0] == 0xCD).all()


_ORDER = {"rgb24": "rgb", "bgr24": "bgr", "rgba": "rgba", "bgra": "bgra"}


@pytest.mark.parametrize("w,h", [(64, 8), (130, 6), (5, 3), (1, 1), (257, 2)])
@pytest.mark.parametrize("pair", [("rgb24", "rgba"), ("rgb24", "bgra"), ("bgr24", "rgba"), ("bgr24", "bgra"),
                                  ("rgba", "rgb24"), ("rgba", "bgr24"), ("bgra", "rgb24"), ("bgra", "bgr24"),
                                  ("rgba", "bgra"), ("bgra", "rgba")])
def test_packed_rgb_repack(dev, orc, w, h, pair):
    """24 <-> 32 and 32 <-> 32 bit packed RGB at equal size: rgbToRgbWrapper's byte moves (alpha 255 when created,
    dropped when removed, kept 32 -> 32).  Checked against the oracle's restatement and against the channel names."""
    sf, df = pair
    src = synth_planes(orc, sf, w, h, seed=5)
    want = np.zeros((h, w * len(_ORDER[df])), np.uint8)
    assert orc.L.orc_rgb_repack(src[0].ctypes.data, src[0].strides[0], PIX_FMT[sf], want.ctypes.data, want.strides[0],
                                PIX_FMT[df], w, h) == 0
    # the same thing said with channel names
    spx = src[0].reshape(h, w, len(_ORDER[sf]))
    chan = {c: spx[:, :, i] for i, c in enumerate(_ORDER[sf])}
    chan.setdefault("a", np.full((h, w), 255, np.uint8))
    assert (want.reshape(h, w, -1) == np.stack([chan[c] for c in _ORDER[df]], axis=2)).all()
    for align, extra in [(256, 0), (1, 1), (4, 0)]:
        d_src = dev.upload_planes(src, align, extra)
        got, pads, kernel = dev.sws(d_src, w, h, sf, w, h, df, dst_align=align, dst_extra=extra)
        assert kernel == "repack_rgb_kernel"
        assert (got[0] == want).all() and (pads[0] == 0xCD).all()


@pytest.mark.parametrize("x0,twin", [("rgb0", "rgba"), ("bgr0", "bgra")])
def test_0alpha_formats_are_their_alpha_twins(dev, orc, x0, twin):
    """RGB0 / BGR0 (scale_cuda's 0bgr32 / 0rgb32 on little endian): libswscale runs them as RGBA / BGRA (handle_0alpha,
    utils.c:1121-1144) — the 4th byte is not read, a created one is 255, and padding that becomes a real alpha is set
    to 255 (swscale.c:959-978)"""
    w, h = 70, 22
    nv = synth_planes(orc, "nv12", w, h, seed=91)
    d = dev.upload_planes(nv, 64)
    a, _, _ = dev.sws(d, w, h, "nv12", w, h, x0)
    b, _, _ = dev.sws(d, w, h, "nv12", w, h, twin)
    assert (a[0] == b[0]).all() and (a[0].reshape(h, w, 4)[:, :, 3] == 255).all()
    a, _, _ = dev.sws(d, w, h, "nv12", 48, 20, x0)
    b, _, _ = dev.sws(d, w, h, "nv12", 48, 20, twin)
    assert (a[0] == b[0]).all()
    src = synth_planes(orc, twin, w, h, seed=92)                       # random 4th byte
    ds = dev.upload_planes(src, 64)
    for df in ("nv12", "rgb24", "yuv444p"):
        a, _, _ = dev.sws(ds, w, h, x0, 40, 18, df) if df == "rgb24" else dev.sws(ds, w, h, x0, w, h, df)
        b, _, _ = dev.sws(ds, w, h, twin, 40, 18, df) if df == "rgb24" else dev.sws(ds, w, h, twin, w, h, df)
        assert all((p == q).all() for p, q in zip(a, b)), df
    # equal layout, padding -> alpha: the byte becomes 255; alpha -> padding and padding -> padding: plain copies
    a, _, k = dev.sws(ds, w, h, x0, w, h, twin)
    assert (a[0].reshape(h, w, 4)[:, :, :3] == src[0].reshape(h, w, 4)[:, :, :3]).all() and (a[0].reshape(h, w, 4)[:, :, 3] == 255).all()
    a, _, _ = dev.sws(ds, w, h, twin, w, h, x0)
    assert (a[0] == src[0]).all()
    other = "bgra" if twin == "rgba" else "rgba"
    a, _, _ = dev.sws(ds, w, h, x0, w, h, other)
    px = src[0].reshape(h, w, 4)
    assert (a[0].reshape(h, w, 4)[:, :, :3] == px[:, :, [2, 1, 0]]).all() and (a[0].reshape(h, w, 4)[:, :, 3] == 255).all()


@pytest.mark.parametrize("pair", [("rgb24", "bgra"), ("bgr24", "rgba")])
def test_24_to_32_with_bitexact_runs_the_generic_scaler(dev, orc, pair):
    """findRgbConvFn returns no converter for 24 -> 32 bit under SWS_BITEXACT (swscale_unscaled.c:1571-1574): the
    context is a generic (lossy, through YUV) one even at equal size"""
    sf, df = pair
    w, h = 70, 22
    flags = SWS["bicubic"] | SWS["bitexact"]
    src = synth_planes(orc, sf, w, h, seed=6)
    want = orc.sws(src, w, h, sf, w, h, df, flags)
    d_src = dev.upload_planes(src, 64)
    got, pads, kernel = dev.sws(d_src, w, h, sf, w, h, df, flags, dst_align=64)
    assert (got[0] == want[0]).all(), kernel
    # (round 5: a width that is not a multiple of four no longer keeps an RGB -> RGB context off the block-cooperative form)
    assert kernel.startswith("scale_rgb_kernel") or kernel == "scale_yuvg_rgbsrc_blk_kernel", kernel


@pytest.mark.parametrize("w,h", [(64, 8), (36, 6)])
def test_nv12_to_rgbpf32_cswscale_shape(dev, orc, w, h):
    # metrans/app/CSwscale.c: tightly packed nv12 in, three stacked float planes out, value = u8/255
    src = synth_planes(orc, "nv12", w, h, seed=9)
    want = orc.nv12_to_rgbpf32(src, w, h)
    packed = np.concatenate([src[0].reshape(-1), src[1].reshape(-1)])
    from harness import DevBuf, ptr
    din = DevBuf(dev, packed.size)
    dev.lib.gmat_memcpy_h2d(din.ptr, ptr(packed), packed.size)
    dout = DevBuf(dev, 12 * w * h)
    ctx = dev.lib.SwscaleCuda_Nv12ToRgbpf32_Init(w, h)
    assert ctx
    r = dev.lib.SwscaleCuda_Nv12ToRgbpf32_Convert(ctx, din.ptr, w, dout.ptr, 4 * w, w, h, None)
    assert r == h
    host = np.empty((3, h, w), np.float32)
    dev.lib.gmat_device_sync()
    dev.lib.gmat_memcpy_d2h(ptr(host), dout.ptr, host.nbytes)
    dev.lib.SwscaleCuda_Nv12ToRgbpf32_Delete(ctx)
    assert (host.view(np.uint32) == want.view(np.uint32)).all()      # bit-identical floats
    din.free(); dout.free()


def test_error_behaviour(dev):
    lib = dev.lib
    assert not lib.gmat_sws_getContext(0, 10, 23, 10, 10, 2, 0, None)             # invalid dimension
    assert not lib.gmat_sws_getContext(16, 16, 999, 16, 16, 2, 0, None)           # unsupported format
    c = lib.gmat_sws_getContext(16, 16, PIX_FMT["nv12"], 16, 16, PIX_FMT["rgb24"], 0, None)
    assert c
    assert lib.gmat_sws_scale(c, None, None, 0, 16, None, None) < 0                # NULL parameters
    src = dev.planes_like("nv12", 16, 16)
    dst = dev.planes_like("rgb24", 16, 16)
    r = lib.gmat_sws_scale(c, planes([p.ptr for p in src]), ints([p.stride for p in src]), 4, 16,
                           planes([p.ptr for p in dst]), ints([p.stride for p in dst]))
    assert r < 0                                                                    # a slice beyond the frame (swscale.c:902-907);
    # one inside it converts the whole frame like ff_swscale_cuda: tests/test_reference_entry_points.py
    lib.gmat_sws_freeContext(c)


def test_error_behaviour_newer_entry_points(dev):
    lib = dev.lib
    assert lib.gmat_rotate(None, 0, None, 0, 8, 8, 8, 8, 3, 0.5, 1, None, None) < 0          # NULL frames
    src = dev.planes_like("rgb24", 8, 8)[0]
    dst = dev.planes_like("rgb24", 8, 8)[0]
    assert lib.gmat_rotate(src.ptr, src.stride, dst.ptr, dst.stride, 8, 8, 8, 8, 5, 0.5, 1, None, None) < 0    # bpp
    assert lib.gmat_transpose(src.ptr, src.stride, dst.ptr, dst.stride, 8, 8, 3, 7, None) < 0                  # dir
    # 4:2:0 -> 4:2:0 at another size needs every destination plane
    c = lib.gmat_sws_getContext(32, 16, PIX_FMT["nv12"], 16, 8, PIX_FMT["yuv420p"], 4, None)
    assert c
    s = dev.planes_like("nv12", 32, 16)
    d = dev.planes_like("yuv420p", 16, 8)
    r = lib.gmat_sws_scale(c, planes([p.ptr for p in s]), ints([p.stride for p in s]), 0, 16,
                           planes([d[0].ptr, d[1].ptr]), ints([d[0].stride, d[1].stride]))
    assert r < 0
    lib.gmat_sws_freeContext(c)
    # (round 3: the 19-bit path has its RGB readers — tests/test_parity_rgb64_src.py; a format outside the table is still refused,
    # and the library-internal plane formats are not accepted from a caller)
    for c in (lib.gmat_sws_getContext(32, 16, PIX_FMT["rgb24"], 16, 8, PIX_FMT["p016le"], 0, None),
              lib.gmat_sws_getContext(32, 16, PIX_FMT["rgb24"], 16, 8, PIX_FMT["rgba64le"], 0, None)):
        assert c
        lib.gmat_sws_freeContext(c)
    assert not lib.gmat_sws_getContext(32, 16, 0x47520064, 16, 8, PIX_FMT["rgb24"], 0, None)
    assert not lib.gmat_sws_getContext(32, 16, 0x47520008, 16, 8, PIX_FMT["p016le"], 0, None)
    assert not lib.gmat_sws_getContext(32, 16, 1, 16, 8, PIX_FMT["rgb24"], 0, None)          # AV_PIX_FMT_YUYV422


@pytest.mark.parametrize("w,h", [(64, 16), (130, 34), (33, 9), (256, 32), (260, 18), (516, 40), (68, 16)])
def test_rgbpf32_back_to_8bit(dev, orc, w, h):
    """format_hip's other direction: planar float RGB -> rgb24 / bgr24 / nv12.  A float frame made by the nv12 ->
    rgbpf32le converter returns to exactly the 8-bit RGB the integer converter gives, and 4:2:0 outputs equal the
    RGB24 -> YUV path applied to that frame."""
    src = synth_planes(orc, "nv12", w, h, seed=71)
    rgb = orc.yuv2rgb(src, w, h, "nv12", "rgb24")
    # the float frame as the C ABI lays it out: three stacked planes in one buffer (CSwscale.c:25-28)
    stacked = orc.nv12_to_rgbpf32(src, w, h).reshape(3 * h, w).view(np.uint8).reshape(3 * h, 4 * w)
    d_pf = dev.upload_planes([np.ascontiguousarray(stacked)], 16)[0]
    assert d_pf.stride == 4 * w or d_pf.stride % 16 == 0
    # upload_planes pads rows: the plane stride is stride * h, so re-pack with that stride in mind
    if d_pf.stride != 4 * w:
        d_pf.free()
        host = np.zeros((3 * h, (4 * w + 15) // 16 * 16), np.uint8)
        host[:, :4 * w] = stacked
        d_pf = dev.upload_planes([host], 1)[0]
    lib = dev.lib
    for dst_fmt in ("rgb24", "bgr24", "nv12", "yuv420p"):
        c = lib.gmat_sws_getContext(w, h, PIX_FMT["rgbpf32le"], w, h, PIX_FMT[dst_fmt], 0, None)
        assert c, dst_fmt
        dst = dev.planes_like(dst_fmt, w, h, 64)
        r = lib.gmat_sws_scale(c, planes([d_pf.ptr]), ints([d_pf.stride]), 0, h, planes([p.ptr for p in dst]),
                               ints([p.stride for p in dst]))
        assert r == h
        got = [p.download() for p in dst]
        kernel = lib.gmat_sws_lastKernel(c).decode()
        lib.gmat_sws_freeContext(c)
        if dst_fmt in ("nv12", "yuv420p"):      # round 4: ONE kernel from the floats to the planes where the strip converter's rule holds
            assert (kernel == "pf32_to_yuv420s_kernel") == (w % 4 == 0 and w >= 64 and h % 2 == 0 and h >= 16), (kernel, w, h)
        if dst_fmt == "rgb24":
            assert (got[0] == rgb).all()
        elif dst_fmt == "bgr24":
            assert (got[0].reshape(h, w, 3) == rgb.reshape(h, w, 3)[:, :, ::-1]).all()
        else:
            want = orc.sws([rgb], w, h, "rgb24", w, h, dst_fmt)
            for g, wv in zip(got, want):
                assert (g == wv).all(), dst_fmt
        for p in dst:
            p.free()


@pytest.mark.parametrize("w,h", [(64, 8), (36, 6), (38, 7), (37, 9), (130, 10), (260, 12), (16, 2), (514, 4)])
@pytest.mark.parametrize("pitch", ["tight", "lines", "odd"])
def test_nv12_to_rgbpf32_every_shape_single_and_batched(dev, orc, w, h, pitch):
    """nv12 -> planar float RGB (format_cuda's / CSwscale's tensor; k_yuv2rgb.hip nv12_to_rgbpf32_kernel, round 4: dword loads, the chroma terms
    once per sample pair, u8 / 255 from an LDS table, streaming 16-byte stores, grid.z = frame): widths on and off the four-pixel groups, odd
    widths and heights (the last chroma pair and row shared), float rows on 16-byte lines and off them (the per-float path), one call and a
    batch of three frames through gmat_sws_scale_batch — bit-identical floats, nothing written past a row"""
    from harness import DevPlane, ptr
    lib = dev.lib
    fstride = {"tight": 4 * w, "lines": (4 * w + 63) // 64 * 64, "odd": 4 * w + 4}[pitch]
    nf = 3
    srcs, wants, ins, outs = [], [], [], []
    for f in range(nf):
        src = synth_planes(orc, "nv12", w, h, seed=40 + f)
        srcs.append(src)
        wants.append(orc.nv12_to_rgbpf32(src, w, h))
        ins.append(dev.upload_planes(src, 4 if pitch != "odd" else 1, 0 if pitch != "odd" else 1))
        outs.append(DevPlane(dev, 3 * h, 4 * w, fstride))
    c = lib.gmat_sws_getContext(w, h, PIX_FMT["nv12"], w, h, PIX_FMT["rgbpf32le"], SWS["hwaccel"], None)
    assert c

    def check(f):
        got = outs[f].download().view(np.float32).reshape(3, h, w)
        assert (got.view(np.uint32) == wants[f].view(np.uint32)).all(), (w, h, pitch, f)
        assert (outs[f].download(with_padding=True)[:, 4 * w:] == 0xCD).all()
    r = lib.gmat_sws_scale(c, planes([p.ptr for p in ins[0]]), ints([p.stride for p in ins[0]]), 0, h, planes([outs[0].ptr]), ints([fstride]))
    assert r == h
    lib.gmat_device_sync()
    assert lib.gmat_sws_lastKernel(c).decode() == "nv12_to_rgbpf32_kernel"
    check(0)
    lib.gmat_memset(outs[0].ptr, 0xCD, 3 * h * fstride)
    sp = (C.c_void_p * (4 * nf))(); dp = (C.c_void_p * (4 * nf))()
    for f in range(nf):
        sp[4 * f], sp[4 * f + 1], dp[4 * f] = ins[f][0].ptr, ins[f][1].ptr, outs[f].ptr
    r = lib.gmat_sws_scale_batch(c, nf, C.cast(sp, C.POINTER(C.c_void_p)), ints([p.stride for p in ins[0]]), C.cast(dp, C.POINTER(C.c_void_p)),
                                 ints([fstride]), C.cast((C.c_void_p * 1)(None), C.POINTER(C.c_void_p)), 1, 0)
    assert r == nf, r
    lib.gmat_device_sync()
    assert int(lib.gmat_sws_lastLaunchFrames(c)) == nf
    for f in range(nf):
        check(f)
    lib.gmat_sws_freeContext(c)
    for p in outs:
        p.buf.free()
    for pl in ins:
        for p in pl:
            p.free()


@pytest.mark.parametrize("w,h", [(64, 16), (256, 32), (260, 18), (772, 22)])
@pytest.mark.parametrize("dst_fmt", ["nv12", "yuv420p"])
def test_rgbpf32_any_floats_to_420(dev, orc, w, h, dst_fmt):
    """the fused float -> 4:2:0 kernel (k_rgb2yuv.hip pf32_to_yuv420s_kernel) on floats a network may hand back: below 0, above 1, NaN,
    values between the 8-bit steps.  Expected: u8 = (int)(clamp(f, 0, 1) * 255 + 0.5) in single precision (NaN -> 0: fmaxf), then libswscale's
    RGB24 -> 4:2:0 lines (the oracle's) — and the same bytes as the two-kernel path it replaces (GMAT_SCALE_NO_STRIP=1)."""
    import os
    rng = np.random.default_rng(5 + w)
    f = rng.uniform(-0.3, 1.3, (3, h, w)).astype(np.float32)
    f[:, ::7, ::5] = np.float32(np.nan)
    f[:, 1::9, 2::11] = (rng.integers(0, 256, f[:, 1::9, 2::11].shape) / np.float32(255.0)).astype(np.float32)
    with np.errstate(invalid="ignore"):
        cl = np.minimum(np.maximum(np.where(np.isnan(f), np.float32(0), f), np.float32(0)), np.float32(1))
        q = (cl * np.float32(255.0) + np.float32(0.5)).astype(np.int32).astype(np.uint8)
    rgb = np.ascontiguousarray(q.transpose(1, 2, 0).reshape(h, 3 * w))
    want = orc.sws([rgb], w, h, "rgb24", w, h, dst_fmt)
    host = np.ascontiguousarray(f.reshape(3 * h, w)).view(np.uint8).reshape(3 * h, 4 * w)
    d_pf = dev.upload_planes([host], 1)[0]
    lib = dev.lib
    names = []
    for no_strip in ("0", "1"):
        os.environ["GMAT_SCALE_NO_STRIP"] = no_strip
        try:
            c = lib.gmat_sws_getContext(w, h, PIX_FMT["rgbpf32le"], w, h, PIX_FMT[dst_fmt], 0, None)
            assert c
            dst = dev.planes_like(dst_fmt, w, h, 64)
            r = lib.gmat_sws_scale(c, planes([d_pf.ptr]), ints([d_pf.stride]), 0, h, planes([p.ptr for p in dst]), ints([p.stride for p in dst]))
            assert r == h
            got = [p.download() for p in dst]
            pads = [p.download(with_padding=True)[:, p.row_bytes:] for p in dst]
            names.append(lib.gmat_sws_lastKernel(c).decode())
            lib.gmat_sws_freeContext(c)
            for g, wv, pd in zip(got, want, pads):
                assert (g == wv).all(), (dst_fmt, names[-1])
                assert (pd == 0xCD).all()
            for p in dst:
                p.free()
        finally:
            os.environ.pop("GMAT_SCALE_NO_STRIP", None)
    assert names[0] == "pf32_to_yuv420s_kernel" and names[1] != names[0], names
    d_pf.free()
