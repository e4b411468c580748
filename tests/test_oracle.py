"""Oracle pinning (CPU only): the restatement against what the reference is known to produce.

Pins available without a reference build:
  * known-answer filter rows the survey recorded from the reference itself
  * literal constants in the reference sources
  * LUT path == closed form over all 2^24 YUV triples (two independent restatements)
"""
import numpy as np
import pytest

from harness import SWS


def test_known_answer_bicubic_3840_to_1920(orc):
    # hLumFilterSize=8, row 100 pos=197, row 0 folded at pos 0; vertical rows /4
    (hl, hlp), (hc, hcp), (vl, vlp), (vc, vcp) = orc.sws_filters(3840, 2160, "rgb24", 1920, 1080, "rgb24")
    assert hl.shape == (1920, 8) and vl.shape == (1080, 8)
    assert hlp[100] == 197
    assert hl[100].tolist() == [-230, -692, 1972, 7142, 7142, 1972, -692, -230]
    assert vl[100].tolist() == [-58, -172, 492, 1786, 1786, 492, -172, -58]
    assert hlp[0] == 0 and hl[0].tolist() == [8192, 7142, 1972, -692, -230, 0, 0, 0]
    assert (hl.sum(axis=1) == 16384).all() and (vl.sum(axis=1) == 4096).all()
    # rgb24 -> rgb24 at 2:1: chroma is pair-averaged on input, 1 tap horizontally, same vertical filter
    assert hc.shape == (1920, 1) and (hc == 16384).all() and (hcp == np.arange(1920)).all()
    assert (vc == vl).all() and (vcp == vlp).all()


def test_known_answer_lanczos_taps(orc):
    (hl, _), _, (vl, _), _ = orc.sws_filters(3840, 2160, "rgb24", 1920, 1080, "rgb24", SWS["lanczos"])
    assert hl.shape[1] == 12 and vl.shape[1] == 12          # "Lanczos: 12 taps"


def test_known_answer_nv12_unscaled_vchr(orc):
    # nv12 -> rgb24 with the default bicubic flags has vChrFilterSize=4,
    # e.g. {-115, 985, 3572, -346}
    _, _, (vl, _), (vc, _) = orc.sws_filters(1920, 1080, "nv12", 1920, 1080, "rgb24")
    assert vl.shape[1] == 1 and vc.shape[1] == 4
    rows = {tuple(r) for r in vc.tolist()}
    assert (-115, 985, 3572, -346) in rows


def test_known_answer_fused_filter_sizes(orc):
    # one nv12 2160p -> rgb24 1080p context has hLum=hChr=vLum=8, vChr=1
    (hl, _), (hc, _), (vl, _), (vc, _) = orc.sws_filters(3840, 2160, "nv12", 1920, 1080, "rgb24")
    assert (hl.shape[1], hc.shape[1], vl.shape[1], vc.shape[1]) == (8, 8, 8, 1)


@pytest.mark.parametrize("cs,full", [(5, 0), (1, 0), (9, 0), (5, 1)])
def test_lut_equals_closed_form_exhaustive(orc, cs, full):
    assert orc.L.orc_yuv2rgb_selfcheck(orc.y2r(cs, full)) == 0


def test_bt601_table_offsets_regression(orc):
    # yuv2rgb.c:806 uses yoffs = 326 (not 384 - 16*255/219), so the table path maps video white
    # (235,128,128) to 253 and mid grey 126 to 126: value = ((326 + Y) * cy - (400 << 16) + 0x8000) >> 16
    # with cy = 65536*255/219.  Regression values derived from that expression, not from a reference run.
    cy = (65536 * 255) // 219
    y = np.array([[16, 235, 126, 200]], np.uint8)
    uv = np.array([[128, 128, 128, 128]], np.uint8)
    out = orc.yuv2rgb([y, uv], 4, 1, "nv12", "rgb24").reshape(4, 3)
    for i, Y in enumerate([16, 235, 126, 200]):
        want = min(max(((326 + Y) * cy - (400 << 16) + 0x8000) >> 16, 0), 255)
        assert out[i].tolist() == [want] * 3
    assert out[0].tolist() == [0, 0, 0] and out[1].tolist() == [253, 253, 253]


def test_point_generic_equals_fast_path(orc):
    # nv12 -> rgb24 with SWS_POINT is byte-identical to the yuv2rgb.c fast path
    w, h = 64, 48
    src = [orc.lcg((h, w), 3), orc.lcg((h // 2, w), 4)]
    fast = orc.yuv2rgb(src, w, h, "nv12", "rgb24")
    gen = orc.sws(src, w, h, "nv12", w, h, "rgb24", SWS["point"])[0]
    assert (fast == gen).all()
    # and the default (bicubic) flags differ, as the survey observed
    bic = orc.sws(src, w, h, "nv12", w, h, "rgb24", SWS["bicubic"])[0]
    assert (fast != bic).any()


def test_yuv420p_equals_nv12(orc):
    w, h = 32, 16
    y, u, v = orc.lcg((h, w), 1), orc.lcg((h // 2, w // 2), 2), orc.lcg((h // 2, w // 2), 3)
    uv = np.empty((h // 2, w), np.uint8)
    uv[:, 0::2], uv[:, 1::2] = u, v
    a = orc.yuv2rgb([y, u, v], w, h, "yuv420p", "bgr24")
    b = orc.yuv2rgb([y, uv], w, h, "nv12", "bgr24")
    assert (a == b).all()


def test_scaler_flat_field_is_preserved(orc):
    # a constant RGB frame must come out constant (filters sum to one; exercises every stage)
    src = np.tile(np.array([200, 100, 50], np.uint8), (40, 64))
    out = orc.sws([src], 64, 40, "rgb24", 32, 20, "rgb24")[0].reshape(20, 32, 3)
    assert (out == out[0, 0]).all()
    assert np.abs(out[0, 0].astype(int) - [200, 100, 50]).max() <= 2    # rgb->yuv->rgb round trip
