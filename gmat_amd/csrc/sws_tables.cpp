// sws_tables.cpp — see sws_tables.h.  Host-only; no device code here.
#include "sws_tables.h"
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace gmat {

namespace {

// {crv, cbu, cgu, cgv} per colour-space index (ITU-R matrices scaled by 65536*255/224),
// values as tabulated in libswscale/yuv2rgb.c:48-60.
const int32_t kInvCoeffs[11][4] = {
    {117489, 138438, 13975, 34925}, {117489, 138438, 13975, 34925}, {104597, 132201, 25675, 53279},
    {104597, 132201, 25675, 53279}, {104448, 132798, 24759, 53109}, {104597, 132201, 25675, 53279},
    {104597, 132201, 25675, 53279}, {117579, 136230, 16907, 35559}, {0, 0, 0, 0},
    {110013, 140363, 12277, 42626}, {110013, 140363, 12277, 42626},
};

const int32_t *inv_coeffs(int cs)
{
    if (cs < 0 || cs > 10 || cs == 8) cs = GMAT_SWS_CS_DEFAULT;
    return kInvCoeffs[cs];
}

int32_t q16_to_i16(int64_t v)   // round a 16.16 value to int16 with saturation (yuv2rgb.c:762-772)
{
    int r = (int)((v + (1 << 15)) >> 16);
    return r < -0x7FFF ? -0x8000 : r > 0x7FFF ? 0x7FFF : r;
}

int64_t div_round(int64_t a, int64_t b) { return a >= 0 ? (a + (b >> 1)) / b : (a - (b >> 1)) / b; }

} // namespace

Yuv2RgbConsts make_yuv2rgb_consts(int colorspace, bool full_range, int brightness, int contrast, int saturation)
{
    const int32_t *t = inv_coeffs(colorspace);
    int64_t crv = t[0], cbu = t[1], cgu = -(int64_t)t[2], cgv = -(int64_t)t[3];
    int64_t cy = 1 << 16, oy = 0;
    const int luma_headroom = 512;
    const int yoffs = (full_range ? 384 : 326) + luma_headroom;

    if (full_range) {
        for (int64_t *p : {&crv, &cbu, &cgu, &cgv}) *p = (*p * 224) / 255;
    } else {
        cy = (cy * 255) / 219;
        oy = 16 << 16;
    }
    cy = (cy * contrast) >> 16;
    for (int64_t *p : {&crv, &cbu, &cgu, &cgv}) *p = (*p * contrast * saturation) >> 32;
    oy -= 256 * (int64_t)brightness;

    Yuv2RgbConsts k{};
    k.y_coeff  = q16_to_i16(cy * (1 << 13));
    k.y_offset = q16_to_i16(oy * (1 << 9));
    k.v2r = q16_to_i16(crv * (1 << 13));
    k.v2g = q16_to_i16(cgv * (1 << 13));
    k.u2g = q16_to_i16(cgu * (1 << 13));
    k.u2b = q16_to_i16(cbu * (1 << 13));

    const int64_t d = std::max<int64_t>(cy, 1);
    for (int64_t *p : {&crv, &cbu, &cgu, &cgv}) *p = ((*p * (1 << 16)) + 0x8000) / d;

    const int64_t yb0 = -(384LL << 16) - luma_headroom * cy - oy;
    k.base = (int32_t)(yb0 + 0x8000);
    k.cy   = (int32_t)cy;
    k.crv = (int32_t)crv; k.cbu = (int32_t)cbu; k.cgu = (int32_t)cgu; k.cgv = (int32_t)cgv;
    k.offR = (int32_t)(yoffs - (crv >> 9));
    k.offG = (int32_t)(yoffs - (cgu >> 9) - (cgv >> 9));
    k.offB = (int32_t)(yoffs - (cbu >> 9));
    return k;
}

Rgb2YuvConsts make_rgb2yuv_consts(int colorspace)
{
    const int32_t *t = inv_coeffs(colorspace);
    const int shift = 15;                       // RGB2YUV_SHIFT
    Rgb2YuvConsts k{};
    if (!std::memcmp(t, inv_coeffs(GMAT_SWS_CS_DEFAULT), 4 * sizeof(int32_t))) {
        // BT.601 literals, utils.c:845-856
        const double s = 1 << shift;
        k.by =  (int)(0.114 * 219 / 255 * s + 0.5);
        k.bv = -(int)(0.081 * 224 / 255 * s + 0.5);
        k.bu =  (int)(0.500 * 224 / 255 * s + 0.5);
        k.gy =  (int)(0.587 * 219 / 255 * s + 0.5);
        k.gv = -(int)(0.419 * 224 / 255 * s + 0.5);
        k.gu = -(int)(0.331 * 224 / 255 * s + 0.5);
        k.ry =  (int)(0.299 * 219 / 255 * s + 0.5);
        k.rv =  (int)(0.500 * 224 / 255 * s + 0.5);
        k.ru = -(int)(0.169 * 224 / 255 * s + 0.5);
        return k;
    }
    const int64_t ONE = 65536;
    const int64_t vr = t[0], ub = t[1], ug = -(int64_t)t[2], vg = -(int64_t)t[3];
    const int64_t cy = ONE * 255 / 219;
    const int64_t W = div_round(ONE * ONE * ug, ub), V = div_round(ONE * ONE * vg, vr), Z = ONE * ONE - W - V;
    const int64_t Cy = div_round(cy * Z, ONE), Cu = div_round(ub * Z, ONE), Cv = div_round(vr * Z, ONE);
    const int64_t S = 1 << shift;
    k.ry = (int32_t)-div_round(S * V, Cy);
    k.gy = (int32_t) div_round(S * ONE * ONE, Cy);
    k.by = (int32_t)-div_round(S * W, Cy);
    k.ru = (int32_t) div_round(S * V, Cu);
    k.gu = (int32_t)-div_round(S * ONE * ONE, Cu);
    k.bu = (int32_t) div_round(S * (Z + W), Cu);
    k.rv = (int32_t) div_round(S * (V + Z), Cv);
    k.gv = (int32_t)-div_round(S * ONE * ONE, Cv);
    k.bv = (int32_t) div_round(S * W, Cv);
    return k;
}

// ------------------------------------------------------------------------------------------
// Filter construction.  Works on a vector of per-output windows of 64-bit taps.
// ------------------------------------------------------------------------------------------
namespace {

struct Window { int32_t pos; std::vector<int64_t> tap; };

int floor_log2(unsigned v) { int n = 0; while (v >>= 1) n++; return n; }

// kernel weight for distance d (30-bit fixed point, already divided by the scale when minifying)
int64_t kernel_weight(int flags, const double param[2], int64_t d, int inc, int64_t fone)
{
    const double fd = d * (1.0 / (1 << 30));
    const bool p0 = param[0] != GMAT_SWS_PARAM_DEFAULT, p1 = param[1] != GMAT_SWS_PARAM_DEFAULT;
    if (flags & GMAT_SWS_BICUBIC) {
        const int64_t B = (int64_t)((p0 ? param[0] : 0.0) * (1 << 24));
        const int64_t C = (int64_t)((p1 ? param[1] : 0.6) * (1 << 24));
        int64_t w = 0;
        if (d < (1LL << 31)) {
            const int64_t dd = (d * d) >> 30, ddd = (dd * d) >> 30;
            if (d < (1LL << 30))
                w = (12 * (1 << 24) - 9 * B - 6 * C) * ddd + (-18 * (1 << 24) + 12 * B + 6 * C) * dd +
                    (6 * (1 << 24) - 2 * B) * (1LL << 30);
            else
                w = (-B - 6 * C) * ddd + (6 * B + 30 * C) * dd + (-12 * B - 48 * C) * d +
                    (8 * B + 24 * C) * (1LL << 30);
        }
        return w / ((1LL << 54) / fone);
    }
    if (flags & GMAT_SWS_X) {                                     // "experimental", utils.c:497-508: a raised-cosine lobe, sharpened by param[0]
        const double A = p0 ? param[0] : 1.0;
        double c = fd < 1.0 ? std::cos(fd * M_PI) : -1.0;
        c = c < 0.0 ? -std::pow(-c, A) : std::pow(c, A);
        return (int64_t)((c * 0.5 + 0.5) * fone);
    }
    if (flags & GMAT_SWS_AREA) {
        const int64_t d2 = d - (1 << 29);
        int64_t w;
        if (d2 * inc < -(1LL << (29 + 16)))      w = 1LL << (30 + 16);
        else if (d2 * inc < (1LL << (29 + 16)))  w = -d2 * inc + (1LL << (29 + 16));
        else                                     w = 0;
        return w * (fone >> (30 + 16));
    }
    if (flags & GMAT_SWS_GAUSS) {
        const double p = p0 ? param[0] : 3.0;
        return (int64_t)(std::exp2(-p * fd * fd) * fone);
    }
    if (flags & GMAT_SWS_SINC)
        return (int64_t)((d ? std::sin(fd * M_PI) / (fd * M_PI) : 1.0) * fone);
    if (flags & GMAT_SWS_LANCZOS) {
        const double p = p0 ? param[0] : 3.0;
        if (fd > p) return 0;
        return (int64_t)((d ? std::sin(fd * M_PI) * std::sin(fd * M_PI / p) / (fd * fd * M_PI * M_PI / p) : 1.0) * fone);
    }
    if (flags & GMAT_SWS_BILINEAR) {
        int64_t w = (1 << 30) - d;
        if (w < 0) w = 0;
        return w * (fone >> 30);
    }
    // SWS_SPLINE, utils.c:322-334, 535-537: the natural cubic spline's kernel, one polynomial piece per unit of distance, each
    // piece's coefficients derived from the previous one's (getSplineCoeff's recursion, written as the loop it is)
    double a = 1.0, b = 0.0, c = -2.196152422706632, e = -c - 1.0, dist = fd;
    while (dist > 1.0) {
        const double nb = b + 2.0 * c + 3.0 * e, nc = c + 3.0 * e, ne = -b - 3.0 * c - 6.0 * e;
        a = 0.0; b = nb; c = nc; e = ne;
        dist -= 1.0;
    }
    return (int64_t)((((e * dist + c) * dist + b) * dist + a) * fone);
}

int support_factor(int flags, const double param[2])
{
    int f = -1;
    // scale_algorithms[] in its own order (utils.c:353-365), the first entry with a factor: BICUBLIN, FAST_BILINEAR, POINT, LANCZOS carry none
    if      (flags & GMAT_SWS_AREA)     f = 1;
    else if (flags & GMAT_SWS_BICUBIC)  f = 4;
    else if (flags & GMAT_SWS_BILINEAR) f = 2;
    else if (flags & GMAT_SWS_GAUSS)    f = 8;
    else if (flags & GMAT_SWS_SINC)     f = 20;
    else if (flags & GMAT_SWS_SPLINE)   f = 20;
    else if (flags & GMAT_SWS_X)        f = 8;
    if (flags & GMAT_SWS_LANCZOS)
        f = param[0] != GMAT_SWS_PARAM_DEFAULT ? (int)std::ceil(2 * param[0]) : 6;
    return f;
}

} // namespace

void pack_filter_pairs(FilterBank &fb)
{
    bool any_odd = false;
    for (int32_t p : fb.pos) any_odd |= (p & 1) != 0;
    fb.pairs = (fb.taps + (any_odd ? 1 : 0) + 1) / 2;
    fb.packed.assign((size_t)fb.count * fb.pairs, 0);
    fb.pos_even.resize(fb.count);
    std::vector<int16_t> row((size_t)fb.pairs * 2);
    for (int i = 0; i < fb.count; i++) {
        const int lead = fb.pos[i] & 1;
        std::fill(row.begin(), row.end(), (int16_t)0);
        for (int j = 0; j < fb.taps; j++) row[lead + j] = fb.coef[(size_t)i * fb.taps + j];
        fb.pos_even[i] = fb.pos[i] - lead;
        for (int k = 0; k < fb.pairs; k++)
            fb.packed[(size_t)i * fb.pairs + k] =
                (int32_t)((uint32_t)(uint16_t)row[2 * k] | ((uint32_t)(uint16_t)row[2 * k + 1] << 16));
    }
}

int build_filter(FilterBank &out, int inc, int src_len, int dst_len, int one, int flags,
                 const double param_in[2], int src_pos, int dst_pos)
{
    double param[2] = {GMAT_SWS_PARAM_DEFAULT, GMAT_SWS_PARAM_DEFAULT};
    if (param_in) { param[0] = param_in[0]; param[1] = param_in[1]; }
    if (src_len < 1 || dst_len < 1) return GMAT_ERR(EINVAL);

    const int ratio_log = std::min(floor_log2((unsigned)std::max(src_len / dst_len, 1)), 8);
    const int64_t fone = 1LL << (54 - ratio_log);
    std::vector<Window> win(dst_len);
    int taps;

    // ---- raw windows -------------------------------------------------------------------
    if (std::abs(inc - 0x10000) < 10 && src_pos == dst_pos) {
        taps = 1;
        for (int i = 0; i < dst_len; i++) win[i] = {i, {fone}};
    } else if (flags & GMAT_SWS_POINT) {
        taps = 1;
        int64_t x = ((dst_pos * (int64_t)inc) >> 8) - ((src_pos * 0x8000LL) >> 7);
        for (int i = 0; i < dst_len; i++, x += inc)
            win[i] = {(int32_t)((x + (1 << 15)) >> 16), {fone}};
    } else if ((inc <= (1 << 16) && (flags & GMAT_SWS_AREA)) || (flags & GMAT_SWS_FAST_BILINEAR)) {
        taps = 2;
        int64_t x = ((dst_pos * (int64_t)inc) >> 8) - ((src_pos * 0x8000LL) >> 7);
        for (int i = 0; i < dst_len; i++, x += inc) {
            int xx = (int)((x - (1LL << 15) + (1 << 15)) >> 16);
            win[i].pos = xx;
            win[i].tap.resize(2);
            for (int j = 0; j < 2; j++, xx++) {
                int64_t w = fone - std::llabs((int64_t)xx * (1 << 16) - x) * (fone >> 16);
                win[i].tap[j] = std::max<int64_t>(w, 0);
            }
        }
    } else {
        const int sf = support_factor(flags, param);
        if (sf <= 0) return GMAT_ERR(EINVAL);
        taps = inc <= (1 << 16) ? 1 + sf : 1 + (int)(((int64_t)sf * src_len + dst_len - 1) / dst_len);
        taps = std::max(std::min(taps, src_len - 2), 1);
        int64_t x = ((dst_pos * (int64_t)inc) >> 7) - ((src_pos * 0x10000LL) >> 7);
        for (int i = 0; i < dst_len; i++, x += 2 * (int64_t)inc) {
            int xx = (int)((x - (taps - 2) * (1LL << 16)) / (1 << 17));
            win[i].pos = xx;
            win[i].tap.resize(taps);
            for (int j = 0; j < taps; j++, xx++) {
                int64_t d = std::llabs(((int64_t)xx * (1 << 17)) - x) << 13;
                if (inc > (1 << 16)) d = d * dst_len / src_len;
                win[i].tap[j] = kernel_weight(flags, param, d, inc, fone);
            }
        }
    }

    // ---- trim negligible taps; keep positions monotone ------------------------------------
    const double cutoff = 0.002 * (double)fone;       // SWS_MAX_REDUCE_CUTOFF
    int keep = 0;
    for (int i = dst_len - 1; i >= 0; i--) {
        std::vector<int64_t> &t = win[i].tap;
        int64_t acc = 0;
        for (int j = 0; j < taps; j++) {
            acc += std::llabs(t[0]);
            if ((double)acc > cutoff) break;
            if (i < dst_len - 1 && win[i].pos >= win[i + 1].pos) break;
            t.erase(t.begin());
            t.push_back(0);
            win[i].pos++;
        }
        int live = taps;
        acc = 0;
        for (int j = taps - 1; j > 0; j--) {
            acc += std::llabs(t[j]);
            if ((double)acc > cutoff) break;
            live--;
        }
        keep = std::max(keep, live);
    }
    if (keep <= 0) return GMAT_ERR(EINVAL);
    if (keep >= 256) return GMAT_ERR(ENOSYS);         // the reference would cascade (utils.c:648-652)
    for (Window &w : win) w.tap.resize(keep);         // alignment 1: no padding taps survive
    taps = keep;

    // ---- fold taps that hang over either border back inside --------------------------------
    for (Window &w : win) {
        std::vector<int64_t> &t = w.tap;
        if (w.pos < 0) {
            for (int j = 1; j < taps; j++) {
                const int left = std::max(j + w.pos, 0);
                t[left] += t[j];
                t[j] = 0;
            }
            w.pos = 0;
        }
        if (w.pos + taps > src_len) {
            const int shift = w.pos + std::min(taps - src_len, 0);
            int64_t acc = 0;
            for (int j = taps - 1; j >= 0; j--)
                if (w.pos + j >= src_len) { acc += t[j]; t[j] = 0; }
            for (int j = taps - 1; j >= 0; j--) t[j] = j < shift ? 0 : t[j - shift];
            w.pos -= shift;
            t[src_len - 1 - w.pos] += acc;
        }
    }

    // ---- normalise each row to `one` with error feedback -----------------------------------
    out.taps = taps;
    out.count = dst_len;
    out.coef.assign((size_t)dst_len * taps, 0);
    out.pos.resize(dst_len);
    for (int i = 0; i < dst_len; i++) {
        const std::vector<int64_t> &t = win[i].tap;
        int64_t sum = 0;
        for (int64_t v : t) sum += v;
        sum = (sum + one / 2) / one;
        if (!sum) sum = 1;
        int64_t err = 0;
        for (int j = 0; j < taps; j++) {
            const int64_t v = t[j] + err;
            const int q = (int)div_round(v, sum);
            out.coef[(size_t)i * taps + j] = (int16_t)q;
            err = v - q * sum;
        }
        out.pos[i] = win[i].pos;
    }
    pack_filter_pairs(out);
    return 0;
}

static int local_chroma_pos(int sub, int pos)      // utils.c:338-345 with the default (unset) position
{
    if (pos == -1 || pos <= -513) pos = (128 << sub) - 128;
    return (pos + 128) >> sub;
}

int build_scale_plan(ScalePlan &p, int srcW, int srcH, int srcFormat, int dstW, int dstH, int dstFormat,
                     int flags, const double param[2], const int *chrPos)
{
    // the planes of a 64-bit packed RGB source keep an RGB source's chroma decisions (utils.c:1427-1557 look at the format the caller gave)
    const bool src_rgb = is_packed_rgb(srcFormat) || is_priv_planes(srcFormat), dst_rgb = is_packed_rgb(dstFormat) || is_rgb64(dstFormat);
    const bool src444 = srcFormat == GMAT_PIX_FMT_YUV444P || srcFormat == GMAT_PIX_FMT_YUV444P16LE;
    const bool dst444 = dstFormat == GMAT_PIX_FMT_YUV444P || dstFormat == GMAT_PIX_FMT_YUV444P16LE;
    if (!(src_rgb || is_yuv420(srcFormat) || src444 || is_p01x(srcFormat) || pl16_depth(srcFormat)) ||
        !(dst_rgb || is_yuv420(dstFormat) || dst444 || is_p01x(dstFormat) || pl16_depth(dstFormat))) return GMAT_ERR(ENOSYS);
    static const int unset[4] = {-513, -513, -513, -513};
    if (!chrPos) chrPos = unset;
    if (srcW < 1 || srcH < 1 || dstW < 1 || dstH < 1) return GMAT_ERR(EINVAL);
    const int algo_mask = 0x7FF;
    if (!(flags & algo_mask)) flags |= GMAT_SWS_BICUBIC;

    p.srcW = srcW; p.srcH = srcH; p.dstW = dstW; p.dstH = dstH;
    p.srcFormat = srcFormat; p.dstFormat = dstFormat;
    p.lumXInc = (int)((((int64_t)srcW << 16) + (dstW >> 1)) / dstW);
    p.lumYInc = (int)((((int64_t)srcH << 16) + (dstH >> 1)) / dstH);
    p.chrSrcHSub = p.chrSrcVSub = (src_rgb || src444) ? 0 : 1;
    p.chrDstHSub = p.chrDstVSub = (dst_rgb || dst444) ? 0 : 1;

    if (dst_rgb) {
        if (!(flags & GMAT_SWS_FULL_CHR_H_INT)) {
            if (dstW & 1) flags |= GMAT_SWS_FULL_CHR_H_INT;
            // non-subsampled sources force full chroma interpolation (utils.c:1439-1447)
            if ((src_rgb || src444) && !(flags & GMAT_SWS_FAST_BILINEAR)) flags |= GMAT_SWS_FULL_CHR_H_INT;
        }
        if (!(flags & GMAT_SWS_FULL_CHR_H_INT)) p.chrDstHSub = 1;
    } else {
        flags &= ~GMAT_SWS_FULL_CHR_H_INT;
    }
    if (src_rgb && !(flags & GMAT_SWS_FULL_CHR_H_INP) &&
        ((dstW >> p.chrDstHSub) <= (srcW >> 1) || (flags & GMAT_SWS_FAST_BILINEAR)))
        p.chrSrcHSub = 1;
    p.flags = flags;

    p.chrSrcW = ceil_rshift(srcW, p.chrSrcHSub);
    p.chrSrcH = ceil_rshift(srcH, p.chrSrcVSub);
    p.chrDstW = ceil_rshift(dstW, p.chrDstHSub);
    p.chrDstH = ceil_rshift(dstH, p.chrDstVSub);
    p.chrXInc = (int)((((int64_t)p.chrSrcW << 16) + (p.chrDstW >> 1)) / p.chrDstW);
    p.chrYInc = (int)((((int64_t)p.chrSrcH << 16) + (p.chrDstH >> 1)) / p.chrDstH);

    int r;
    // SWS_BICUBLIN: bicubic luma banks, bilinear chroma banks (utils.c:1830, 1841, 1860, 1869)
    const int lflags = (flags & GMAT_SWS_BICUBLIN) ? (flags | GMAT_SWS_BICUBIC) : flags;
    const int cflags = (flags & GMAT_SWS_BICUBLIN) ? (flags | GMAT_SWS_BILINEAR) : flags;
    if ((r = build_filter(p.hLum, p.lumXInc, srcW, dstW, 1 << 14, lflags, param,
                          local_chroma_pos(0, 0), local_chroma_pos(0, 0))) < 0) return r;
    if ((r = build_filter(p.hChr, p.chrXInc, p.chrSrcW, p.chrDstW, 1 << 14, cflags, param,
                          local_chroma_pos(p.chrSrcHSub, chrPos[0]), local_chroma_pos(p.chrDstHSub, chrPos[2]))) < 0) return r;
    if ((flags & GMAT_SWS_FAST_BILINEAR) && is_yuv8_src(srcFormat) && !is_dst16(dstFormat)) {
        // 8-bit samples into 15-bit lines with SWS_FAST_BILINEAR: libswscale leaves hScale8To15_c for ff_hyscale_fast_c /
        // ff_hcscale_fast_c (swscale.c:566-574, hscale_fast_bilinear.c:23-55) — xpos += xInc, the two samples at xpos >> 16 blended
        // with the 7-bit xalpha = (xpos & 0xFFFF) >> 9 (luma weights 128 - xalpha and xalpha, chroma (xalpha ^ 127) and xalpha), and
        // src[srcW - 1] * 128 wherever (i * xInc) >> 16 reaches the last sample.  That is hScale8To15_c over this two-tap bank.
        auto fast_bank = [](FilterBank &fb, int inc, int srcLen, int dstLen, int wsum) {
            fb.taps = 2; fb.count = dstLen;
            fb.coef.assign((size_t)dstLen * 2, 0); fb.pos.assign(dstLen, 0);
            unsigned xpos = 0;
            for (int i = 0; i < dstLen; i++, xpos += (unsigned)inc) {
                const unsigned xx = xpos >> 16, xa = (xpos & 0xFFFF) >> 9;
                if ((int)(((unsigned)i * (unsigned)inc) >> 16) >= srcLen - 1) {
                    fb.pos[i] = srcLen >= 2 ? srcLen - 2 : 0;
                    fb.coef[2 * i] = srcLen >= 2 ? 0 : 16384; fb.coef[2 * i + 1] = srcLen >= 2 ? 16384 : 0;
                } else {
                    fb.pos[i] = (int)xx;
                    fb.coef[2 * i] = (int16_t)((wsum - (int)xa) << 7); fb.coef[2 * i + 1] = (int16_t)(xa << 7);
                }
            }
            pack_filter_pairs(fb);
        };
        fast_bank(p.hLum, p.lumXInc, srcW, dstW, 128);
        fast_bank(p.hChr, p.chrXInc, p.chrSrcW, p.chrDstW, 127);
    }
    if ((r = build_filter(p.vLum, p.lumYInc, srcH, dstH, 1 << 12, lflags, param,
                          local_chroma_pos(0, 0), local_chroma_pos(0, 0))) < 0) return r;
    if ((r = build_filter(p.vChr, p.chrYInc, p.chrSrcH, p.chrDstH, 1 << 12, cflags, param,
                          local_chroma_pos(p.chrSrcVSub, chrPos[1]), local_chroma_pos(p.chrDstVSub, chrPos[3]))) < 0) return r;
    return 0;
}

} // namespace gmat
