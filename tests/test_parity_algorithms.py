"""SWS_X ("experimental"), SWS_BICUBLIN and SWS_SPLINE (round 4): the three entries of scale_algorithms[] (utils.c:353-365) rounds 1-3 refused.
Each is held to the oracle here and — in the build container — to the reference's own libswscale through its real core (tests/fuzz/fuzz_ref_core.py
draws all eleven algorithms; tests/test_libswscale_core.py has fixed cases).  Known answers: the raised-cosine lobe of SWS_X at integer distances, the
spline kernel's interpolation property, BICUBLIN's luma banks = SWS_BICUBIC's and chroma banks = SWS_BILINEAR's."""
import ctypes as C

import numpy as np
import pytest

from harness import SWS, PIX_FMT, synth_planes

ALGOS = ["x", "bicublin", "spline"]
PAIRS = [("nv12", "rgb24"), ("yuv420p", "bgra"), ("nv12", "nv12"), ("yuv420p", "nv12"), ("rgb24", "rgb24"), ("bgra", "rgba"), ("rgb24", "nv12"),
         ("p010le", "nv12"), ("nv12", "p010le"), ("rgba64le", "rgb24"), ("yuv444p", "rgb24")]
GEOMS = [(256, 144, 128, 72), (200, 120, 68, 42), (160, 90, 240, 136), (192, 108, 128, 72), (640, 64, 212, 24)]


def _filters(lib, c):
    out = []
    for which in range(4):
        cnt = C.c_int()
        size = lib.gmat_sws_getFilter(c, which, None, None, 0, C.byref(cnt))
        assert size > 0
        coef = (C.c_int16 * (size * cnt.value))(); pos = (C.c_int32 * cnt.value)()
        assert lib.gmat_sws_getFilter(c, which, coef, pos, size * cnt.value, C.byref(cnt)) == size
        out.append((np.array(coef).reshape(cnt.value, size), np.array(pos)))
    return out


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("pair", PAIRS)
def test_against_the_oracle(dev, orc, algo, pair):
    sf, df = pair
    for geom in GEOMS:
        sw, sh, dw, dh = geom
        src = synth_planes(orc, sf, sw, sh, seed=121)
        if sf == "p010le":
            for p in src:
                p.view("<u2")[...] &= 0xFFC0
        want = orc.sws(src, sw, sh, sf, dw, dh, df, SWS[algo])
        d = dev.upload_planes(src, 64)
        got, pads, kernel = dev.sws(d, sw, sh, sf, dw, dh, df, SWS[algo], dst_align=64)
        for p in d:
            p.free()
        for i, (g, w) in enumerate(zip(got, want)):
            assert (g == w).all(), f"{kernel} {sf} -> {df} {geom} {algo} plane {i}: {int((g != w).sum())} bytes"
            assert (pads[i] == 0xCD).all()


@pytest.mark.parametrize("geom", [(256, 144, 128, 72), (200, 120, 68, 42), (160, 90, 240, 136)])
def test_filter_banks(dev, orc, geom):
    """the banks the kernels read are the oracle's (initFilter restated twice, independently), and BICUBLIN's are the two algorithms' it names"""
    lib = dev.lib
    sw, sh, dw, dh = geom

    def banks(algo):
        c = lib.gmat_sws_getContext(sw, sh, PIX_FMT["nv12"], dw, dh, PIX_FMT["nv12"], SWS[algo] | SWS["hwaccel"], None)
        assert c
        f = _filters(lib, c)
        lib.gmat_sws_freeContext(c)
        return f
    for algo in ALGOS:
        mine, theirs = banks(algo), orc.sws_filters(sw, sh, "nv12", dw, dh, "nv12", SWS[algo])
        for (mc, mp), (tc, tp) in zip(mine, theirs):
            assert mc.shape == tc.shape and (mc == tc).all() and (mp == tp).all(), algo
    bl, bc, bi = banks("bicublin"), banks("bicubic"), banks("bilinear")
    for which, ref in ((0, bc), (2, bc), (1, bi), (3, bi)):          # hLum, vLum: bicubic; hChr, vChr: bilinear (utils.c:1830, 1841, 1860, 1869)
        assert bl[which][0].shape == ref[which][0].shape and (bl[which][0] == ref[which][0]).all() and (bl[which][1] == ref[which][1]).all()


def test_known_answers_of_the_kernels(dev):
    """1 : 2 up-scale: output 2 k sits a quarter sample left of source k, so the taps are the kernel at distances .25, .75, 1.25 ...; the kernels
    themselves (utils.c:497-508, 322-334 + 535-537) evaluated here in floating point, normalised like initFilter (sum 16384), agree with every
    interior row of the bank to the unit the error diffusion of the normalisation moves (+-1)"""
    lib = dev.lib
    sw, dw = 64, 128
    def kx(d):
        c = np.cos(d * np.pi) if d < 1.0 else -1.0
        return c * 0.5 + 0.5
    def ksp(d):
        a, b, c, e = 1.0, 0.0, -2.196152422706632, 2.196152422706632 - 1.0
        while d > 1.0:
            a, b, c, e = 0.0, b + 2 * c + 3 * e, c + 3 * e, -b - 3 * c - 6 * e
            d -= 1.0
        return ((e * d + c) * d + b) * d + a
    for algo, k in (("x", kx), ("spline", ksp)):
        c = lib.gmat_sws_getContext(sw, 32, PIX_FMT["nv12"], dw, 64, PIX_FMT["nv12"], SWS[algo] | SWS["hwaccel"], None)
        assert c
        (coef, pos) = _filters(lib, c)[0]
        lib.gmat_sws_freeContext(c)
        for i in (40, 41, 62, 63):                                   # interior rows of both phases
            centre = (i + 0.5) * sw / dw - 0.5
            w = np.array([k(abs(pos[i] + j - centre)) for j in range(coef.shape[1])])
            w = w / w.sum() * 16384
            assert np.abs(coef[i] - w).max() <= 1.5, (algo, i, coef[i].tolist(), np.round(w, 1).tolist())
