/*
 * oracle/orc_vf.c — TEST INFRASTRUCTURE ONLY (see orc.h).
 *
 * CPU counterparts of the reference's GPU filters.  The reference's
 * crop_nvcv / flip_nvcv / rotate_nvcv / smooth_nvcv delegate their arithmetic to CV-CUDA
 * 0.3.1_beta, which is not vendored (README.md:46-52, configure:6533) and has no test in
 * the tree: parity for those is unpinned, so the build defines the operations by the
 * in-tree CPU filters:
 *   transpose   libavfilter/vf_transpose.c:267-327 (dir bit0: read source bottom-up,
 *               bit1: write destination bottom-up; names :374-379
 *               0=cclock_flip 1=clock 2=cclock 3=clock_flip)
 *   hflip       libavfilter/vf_hflip.c:89-117
 *   vflip       libavfilter/vf_vflip.c:108-127
 *   crop        libavfilter/vf_crop.c (pointer offset)
 *   3x3 smooth  libavfilter/vf_convolution.c filter_3x3 :495-512, border setup_3x3 :555-569
 */
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include "orc.h"

void orc_transpose(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride,
                   int in_w, int in_h, int bpp, int dir)
{
    /* out is in_h wide and in_w tall: out(x, y) = in'(col = y, row = x) */
    const int out_w = in_h, out_h = in_w;
    int x, y;
    for (y = 0; y < out_h; y++) {
        int oy = (dir & 2) ? out_h - 1 - y : y;
        uint8_t *drow = dst + (long)oy * dst_stride;
        for (x = 0; x < out_w; x++) {
            int sy = (dir & 1) ? in_h - 1 - x : x;
            memcpy(drow + (long)x * bpp, src + (long)sy * src_stride + (long)y * bpp, bpp);
        }
    }
}

void orc_hflip(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int w, int h, int bpp)
{
    int x, y;
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++)
            memcpy(dst + (long)y * dst_stride + (long)x * bpp,
                   src + (long)y * src_stride + (long)(w - 1 - x) * bpp, bpp);
}

void orc_vflip(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int w, int h, int bpp)
{
    int y;
    for (y = 0; y < h; y++)
        memcpy(dst + (long)y * dst_stride, src + (long)(h - 1 - y) * src_stride, (size_t)w * bpp);
}

void orc_crop(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride,
              int x, int y, int w, int h, int bpp)
{
    int r;
    for (r = 0; r < h; r++)
        memcpy(dst + (long)r * dst_stride, src + (long)(y + r) * src_stride + (long)x * bpp, (size_t)w * bpp);
}

void orc_conv3x3(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride,
                 int w, int h, int bpp, const int matrix[9], float rdiv, float bias)
{
    int x, y, ch, i;
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++)
            for (ch = 0; ch < bpp; ch++) {
                int sum = 0;
                for (i = 0; i < 9; i++) {
                    int xoff = abs(x + ((i % 3) - 1));
                    int yoff = abs(y + (i / 3) - 1);
                    xoff = xoff >= w ? 2 * w - 1 - xoff : xoff;
                    yoff = yoff >= h ? 2 * h - 1 - yoff : yoff;
                    sum += src[(long)yoff * src_stride + (long)xoff * bpp + ch] * matrix[i];
                }
                sum = (int)(sum * rdiv + bias + 0.5f);
                dst[(long)y * dst_stride + (long)x * bpp + ch] =
                    (uint8_t)(sum < 0 ? 0 : sum > 255 ? 255 : sum);
            }
}

void orc_rgb24_swap_rb(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int w, int h)
{
    int x, y;
    for (y = 0; y < h; y++) {
        const uint8_t *s = src + (long)y * src_stride;
        uint8_t *d = dst + (long)y * dst_stride;
        for (x = 0; x < w; x++) {
            uint8_t r = s[3 * x], g = s[3 * x + 1], b = s[3 * x + 2];
            d[3 * x] = b; d[3 * x + 1] = g; d[3 * x + 2] = r;
        }
    }
}

/* ---- 3x3 median per channel -------------------------------------------------------------------------
 * vf_median.c + median_template.c at radius = radiusV = 1, percentile 0.5 (t = 4, vf_median.c:125): the value at
 * which the cumulative histogram of the window exceeds t, i.e. the 5th smallest of the nine; the window's rows are
 * max(0, y-1) .. min(h-1, y+1) (median_template.c:101-111) and its columns are clamped the same way (:116-118,
 * :133-147: column 0 is added radius extra times, column width-1 stands in past the right edge).  Packed pixels
 * are handled per channel (the CPU filter takes planar formats only; the nvcv filter takes packed RGB). */
void orc_median3x3(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int w, int h, int bpp)
{
    int x, y, ch, i, j;
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++)
            for (ch = 0; ch < bpp; ch++) {
                uint8_t v[9];
                int n = 0;
                for (j = -1; j <= 1; j++)
                    for (i = -1; i <= 1; i++) {
                        int yy = y + j < 0 ? 0 : y + j > h - 1 ? h - 1 : y + j;
                        int xx = x + i < 0 ? 0 : x + i > w - 1 ? w - 1 : x + i;
                        v[n++] = src[(long)yy * src_stride + (long)xx * bpp + ch];
                    }
                for (i = 1; i < 9; i++) {                      /* insertion sort */
                    uint8_t k = v[i];
                    for (j = i - 1; j >= 0 && v[j] > k; j--) v[j + 1] = v[j];
                    v[j + 1] = k;
                }
                dst[(long)y * dst_stride + (long)x * bpp + ch] = v[4];
            }
}

/* ---- packed 24/32-bit RGB re-packing at equal size -------------------------------------------------
 * rgbToRgbWrapper (swscale_unscaled.c:1579-1640) with the converter findRgbConvFn picks (:1458-1577) on a
 * little-endian host.  With RGB24 in the "RGB in int" class, BGR24 in the "BGR in int" class, RGBA = BGR32 and
 * BGRA = RGB32 (pixfmt.h), the table reduces to:
 *   same class,  24 -> 32   rgb24to32        rgb2rgb.c:170-188           d = { s[2], s[1], s[0], 255 }
 *   other class, 24 -> 32   rgb24tobgr32_c   rgb2rgb_template.c:31-53    d = { s[0], s[1], s[2], 255 }
 *   same class,  32 -> 24   rgb32to24        rgb2rgb.c:152-168           d = { s[2], s[1], s[0] }
 *   other class, 32 -> 24   rgb32tobgr24_c   rgb2rgb_template.c:55-77    d = { s[0], s[1], s[2] }
 *   RGBA <-> BGRA           shuffle_bytes_2103_c  rgb2rgb_template.c:317-329   bytes 0 and 2 exchanged, alpha kept
 * Returns 0, or -1 for a pair that has no special converter here (equal formats are a copy, 24 <-> 24 is
 * orc_rgb24_swap_rb).  NOTE (swscale_unscaled.c:1571-1574): with SWS_BITEXACT the 24 -> 32 converters are not
 * used at all; such a context runs the generic scaler, i.e. orc_sws_create. */
static int orc_rgb_in_int(int fmt) { return fmt == ORC_PIX_RGB24 || fmt == ORC_PIX_BGRA; }   /* else: BGR in int */

int orc_rgb_repack(const uint8_t *src, int src_stride, int src_fmt, uint8_t *dst, int dst_stride, int dst_fmt,
                   int w, int h)
{
    const int s32 = src_fmt == ORC_PIX_RGBA || src_fmt == ORC_PIX_BGRA;
    const int d32 = dst_fmt == ORC_PIX_RGBA || dst_fmt == ORC_PIX_BGRA;
    const int s24 = src_fmt == ORC_PIX_RGB24 || src_fmt == ORC_PIX_BGR24;
    const int d24 = dst_fmt == ORC_PIX_RGB24 || dst_fmt == ORC_PIX_BGR24;
    const int same_class = orc_rgb_in_int(src_fmt) == orc_rgb_in_int(dst_fmt);
    int x, y;
    if (!(s32 || s24) || !(d32 || d24) || src_fmt == dst_fmt || (s24 && d24))
        return -1;
    for (y = 0; y < h; y++) {
        const uint8_t *s = src + (long)y * src_stride;
        uint8_t *d = dst + (long)y * dst_stride;
        for (x = 0; x < w; x++) {
            if (s32 && d32) {                         /* shuffle_bytes_2103 */
                d[4 * x] = s[4 * x + 2]; d[4 * x + 1] = s[4 * x + 1]; d[4 * x + 2] = s[4 * x]; d[4 * x + 3] = s[4 * x + 3];
            } else if (s24) {                         /* rgb24to32 / rgb24tobgr32 */
                d[4 * x]     = same_class ? s[3 * x + 2] : s[3 * x];
                d[4 * x + 1] = s[3 * x + 1];
                d[4 * x + 2] = same_class ? s[3 * x] : s[3 * x + 2];
                d[4 * x + 3] = 255;
            } else {                                  /* rgb32to24 / rgb32tobgr24 */
                d[3 * x]     = same_class ? s[4 * x + 2] : s[4 * x];
                d[3 * x + 1] = s[4 * x + 1];
                d[3 * x + 2] = same_class ? s[4 * x] : s[4 * x + 2];
            }
        }
    }
    return 0;
}

/* ---- vf_rotate.c: arbitrary-angle rotation in 16.16 fixed point ---------------------------------
 * int_sin                 vf_rotate.c:198-218   Taylor series on angles scaled by 2^20, result scaled by 2^16
 * interpolate_bilinear8   vf_rotate.c:224-249
 * filter_slice            vf_rotate.c:410-498   (general branch: the four exact quarter-turn branches are the
 *                                                transposes/flips restated above)
 * filter_frame            vf_rotate.c:500-548   angle_int = angle * FIXP * 16; per-plane xi / yi / xprime / yprime
 * One plane of `bpp`-byte pixels; the caller passes plane dimensions (chroma planes: ceil-shifted sizes).
 * fill == NULL: pixels whose source position is out of range are left untouched (fillcolor=none). */
#define ROT_FIXP  (1 << 16)
#define ROT_FIXP2 (1 << 20)
#define ROT_INT_PI 3294199

static int64_t rot_int_sin(int64_t a)
{
    int64_t a2, res = 0;
    int i;
    if (a < 0) a = ROT_INT_PI - a;
    a %= 2 * ROT_INT_PI;
    if (a >= ROT_INT_PI * 3 / 2) a -= 2 * ROT_INT_PI;
    if (a >= ROT_INT_PI / 2) a = ROT_INT_PI - a;
    a2 = (a * a) / ROT_FIXP2;
    for (i = 2; i < 11; i += 2) {
        res += a;
        a = -a * a2 / (ROT_FIXP2 * i * (i + 1));
    }
    return (res + 8) >> 4;
}

void orc_rotate_sincos(double angle_rad, int *s, int *c)
{
    int angle_int = (int)(angle_rad * ROT_FIXP * 16);
    *s = (int)rot_int_sin(angle_int);
    *c = (int)rot_int_sin(angle_int + ROT_INT_PI / 2);
}

static int rot_clip(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

void orc_rotate(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride,
                int inw, int inh, int outw, int outh, int bpp, double angle_rad, int bilinear,
                const uint8_t *fill)
{
    int s, c, i, j, k;
    int xi, yi, xprime, yprime;
    orc_rotate_sincos(angle_rad, &s, &c);
    xi = -(outw - 1) * c / 2; yi = (outw - 1) * s / 2;
    xprime = -(outh - 1) * s / 2;
    yprime = -(outh - 1) * c / 2;
    if (fill)                                    /* ff_fill_rectangle over the whole output, :527-529 */
        for (j = 0; j < outh; j++)
            for (i = 0; i < outw; i++)
                for (k = 0; k < bpp; k++) dst[(long)j * dst_stride + i * bpp + k] = fill[k];
    for (j = 0; j < outh; j++) {
        int x = xprime + xi + ROT_FIXP * (inw - 1) / 2;
        int y = yprime + yi + ROT_FIXP * (inh - 1) / 2;
        for (i = 0; i < outw; i++) {
            int x1 = x >> 16, y1 = y >> 16;
            if (x1 >= -1 && x1 <= inw && y1 >= -1 && y1 <= inh) {
                uint8_t *pout = dst + (long)j * dst_stride + i * bpp;
                if (bilinear) {
                    int int_x = rot_clip(x >> 16, 0, inw - 1), int_y = rot_clip(y >> 16, 0, inh - 1);
                    int frac_x = x & 0xFFFF, frac_y = y & 0xFFFF;
                    int int_x1 = int_x + 1 < inw - 1 ? int_x + 1 : inw - 1;
                    int int_y1 = int_y + 1 < inh - 1 ? int_y + 1 : inh - 1;
                    for (k = 0; k < bpp; k++) {
                        int s00 = src[bpp * int_x  + k + (long)src_stride * int_y];
                        int s01 = src[bpp * int_x1 + k + (long)src_stride * int_y];
                        int s10 = src[bpp * int_x  + k + (long)src_stride * int_y1];
                        int s11 = src[bpp * int_x1 + k + (long)src_stride * int_y1];
                        int s0 = ((1 << 16) - frac_x) * s00 + frac_x * s01;
                        int s1 = ((1 << 16) - frac_x) * s10 + frac_x * s11;
                        pout[k] = (uint8_t)(((int64_t)((1 << 16) - frac_y) * s0 + (int64_t)frac_y * s1) >> 32);
                    }
                } else {
                    int x2 = rot_clip(x1, 0, inw - 1), y2 = rot_clip(y1, 0, inh - 1);
                    for (k = 0; k < bpp; k++) pout[k] = src[(long)y2 * src_stride + x2 * bpp + k];
                }
            }
            x += c;
            y -= s;
        }
        xprime += s;
        yprime += c;
    }
}

/* ---- rotate_nvcv's remaining options: interp = cubic | area, shift_x / shift_y (vf_rotate_nvcv.c:79-88,:114-135) ------------------
 * The reference hands them to CV-CUDA's rotate operator, whose arithmetic no reference test pins and whose source is not in the
 * tree (PARITY UNPINNED).  The build defines them on top of vf_rotate.c's fixed-point walk above — this is
 * the statement of that rule, the product implements the same integers:
 *   shift   the rotated image is translated by (shift_x, shift_y) output pixels: out(i, j) = rot(i - shift_x, j - shift_y).  With
 *           S = llrint(shift * 65536) the walk's start moves by  x0 -= (Sx c + Sy s) >> 16,  y0 -= (Sy c - Sx s) >> 16  (64-bit
 *           products, arithmetic shifts); the rotation stays about the centre (the reference rotates about the corner and leaves the
 *           re-centring to the user's shift: SURVEY.md section 0, defect 11).
 *   cubic   Catmull-Rom (a = -1/2) on the 4 x 4 neighbourhood of (x >> 16, y >> 16) with clamped indices, the fractions cut to 8
 *           bits f = (x & 0xFFFF) >> 8.  Integer weights of 14 fractional bits:  n0 = -f^3 + 512 f^2 - 65536 f,
 *           n1 = 3 f^3 - 1280 f^2 + 2^25, n3 = f^3 - 256 f^2;  w_k = (n_k + 1024) >> 11 for k = 0, 1, 3, w_2 = 16384 - w_0 - w_1 - w_3.
 *           Rows first: h_r = sum_k w_x[k] p[r][k];  then  out = clip_u8((sum_r w_y[r] h_r + 2^27) >> 28)  in 64 bits.
 *   area    = linear, as cv::warpAffine does for INTER_AREA.
 * interp: 0 nearest, 1 linear, 2 cubic. */
static void rot_cubic_w(int f, int w[4])
{
    const int64_t f2 = (int64_t)f * f, f3 = f2 * f;
    const int64_t n0 = -f3 + 512 * f2 - 65536 * (int64_t)f, n1 = 3 * f3 - 1280 * f2 + ((int64_t)1 << 25), n3 = f3 - 256 * f2;
    w[0] = (int)((n0 + 1024) >> 11); w[1] = (int)((n1 + 1024) >> 11); w[3] = (int)((n3 + 1024) >> 11);
    w[2] = 16384 - w[0] - w[1] - w[3];
}

void orc_rotate2(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride,
                 int inw, int inh, int outw, int outh, int bpp, double angle_rad, int interp,
                 double shift_x, double shift_y, const uint8_t *fill)
{
    int s, c, i, j, k;
    int xi, yi, xprime, yprime;
    /* (a translation of more than a million pixels is the same all-background frame as one of a million) */
    const double lim = 1.0e6;
    const int64_t Sx = llrint(fmin(fmax(shift_x, -lim), lim) * 65536.0), Sy = llrint(fmin(fmax(shift_y, -lim), lim) * 65536.0);
    int64_t tx, ty, reach;
    orc_rotate_sincos(angle_rad, &s, &c);
    xi = -(outw - 1) * c / 2; yi = (outw - 1) * s / 2;
    xprime = -(outh - 1) * s / 2;
    yprime = -(outh - 1) * c / 2;
    if (fill)
        for (j = 0; j < outh; j++)
            for (i = 0; i < outw; i++)
                for (k = 0; k < bpp; k++) dst[(long)j * dst_stride + i * bpp + k] = fill[k];
    /* a start beyond the reach of every output pixel: nothing maps into the source (and the 32-bit walk below would wrap) */
    tx = (Sx * c + Sy * s) >> 16; ty = (Sy * c - Sx * s) >> 16;
    reach = ((int64_t)inw + inh + outw + outh + 4) * ROT_FIXP;
    {
        const int64_t X0 = (int64_t)xprime + xi + ROT_FIXP * (inw - 1) / 2 - tx, Y0 = (int64_t)yprime + yi + ROT_FIXP * (inh - 1) / 2 - ty;
        if (X0 > reach || X0 < -reach || Y0 > reach || Y0 < -reach) return;
    }
    for (j = 0; j < outh; j++) {
        int x = xprime + xi + ROT_FIXP * (inw - 1) / 2 - (int)tx;
        int y = yprime + yi + ROT_FIXP * (inh - 1) / 2 - (int)ty;
        for (i = 0; i < outw; i++) {
            int x1 = x >> 16, y1 = y >> 16;
            if (x1 >= -1 && x1 <= inw && y1 >= -1 && y1 <= inh) {
                uint8_t *pout = dst + (long)j * dst_stride + i * bpp;
                if (interp == 2) {
                    int wx[4], wy[4], r, t;
                    rot_cubic_w((x & 0xFFFF) >> 8, wx); rot_cubic_w((y & 0xFFFF) >> 8, wy);
                    for (k = 0; k < bpp; k++) {
                        int64_t v = 0;
                        for (r = 0; r < 4; r++) {
                            const int yy = rot_clip(y1 - 1 + r, 0, inh - 1);
                            int h = 0;
                            for (t = 0; t < 4; t++) h += wx[t] * src[(long)yy * src_stride + rot_clip(x1 - 1 + t, 0, inw - 1) * bpp + k];
                            v += (int64_t)wy[r] * h;
                        }
                        v = (v + ((int64_t)1 << 27)) >> 28;
                        pout[k] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
                    }
                } else if (interp == 1) {
                    int int_x = rot_clip(x >> 16, 0, inw - 1), int_y = rot_clip(y >> 16, 0, inh - 1);
                    int frac_x = x & 0xFFFF, frac_y = y & 0xFFFF;
                    int int_x1 = int_x + 1 < inw - 1 ? int_x + 1 : inw - 1;
                    int int_y1 = int_y + 1 < inh - 1 ? int_y + 1 : inh - 1;
                    for (k = 0; k < bpp; k++) {
                        int s00 = src[bpp * int_x  + k + (long)src_stride * int_y];
                        int s01 = src[bpp * int_x1 + k + (long)src_stride * int_y];
                        int s10 = src[bpp * int_x  + k + (long)src_stride * int_y1];
                        int s11 = src[bpp * int_x1 + k + (long)src_stride * int_y1];
                        int s0 = ((1 << 16) - frac_x) * s00 + frac_x * s01;
                        int s1 = ((1 << 16) - frac_x) * s10 + frac_x * s11;
                        pout[k] = (uint8_t)(((int64_t)((1 << 16) - frac_y) * s0 + (int64_t)frac_y * s1) >> 32);
                    }
                } else {
                    int x2 = rot_clip(x1, 0, inw - 1), y2 = rot_clip(y1, 0, inh - 1);
                    for (k = 0; k < bpp; k++) pout[k] = src[(long)y2 * src_stride + x2 * bpp + k];
                }
            }
            x += c;
            y -= s;
        }
        xprime += s;
        yprime += c;
    }
}

/* ---- median of a kw x kh window per channel: vf_median.c + median_template.c at radius = (kw - 1) / 2, radiusV = (kh - 1) / 2,
 * percentile 0.5: t = 2 r rV + r + rV (vf_median.c:125), the output is the value at which the cumulative histogram of the window
 * exceeds t, i.e. its (t + 1)-th smallest of (2 r + 1)(2 rV + 1) samples; window rows max(0, y - rV) .. min(h - 1, y + rV)
 * (median_template.c:97-106) and columns are clamped the same way (:110,:122,:133-147); a radius larger than the plane allows is
 * clipped to (size - 1) / 2 first (check_params, vf_median.c:111-123). */
void orc_median(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int w, int h, int bpp, int kw, int kh)
{
    int r = (kw - 1) / 2, rv = (kh - 1) / 2, x, y, ch, i, j;
    if (w < 2 * r + 1) r = (w - 1) / 2;
    if (h < 2 * rv + 1) rv = (h - 1) / 2;
    {
        const int t = 2 * r * rv + r + rv;
        for (y = 0; y < h; y++)
            for (x = 0; x < w; x++)
                for (ch = 0; ch < bpp; ch++) {
                    int hist[256] = {0}, sum = 0, v;
                    for (j = -rv; j <= rv; j++)
                        for (i = -r; i <= r; i++) {
                            int yy = y + j < 0 ? 0 : y + j > h - 1 ? h - 1 : y + j;
                            int xx = x + i < 0 ? 0 : x + i > w - 1 ? w - 1 : x + i;
                            hist[src[(long)yy * src_stride + (long)xx * bpp + ch]]++;
                        }
                    for (v = 0; v < 256; v++) { sum += hist[v]; if (sum > t) break; }
                    dst[(long)y * dst_stride + (long)x * bpp + ch] = (uint8_t)v;
                }
    }
}

/* planarCopyWrapper, 8-bit samples into a deeper planar format of the same layout (swscale_unscaled.c:1844-1862, COPY816):
 * shiftonly = chroma planes, and luma of a limited-range source: v << (depth - 8); luma of a full-range source:
 * v << (depth - 8) | v >> (16 - depth).  One plane per call. */
void orc_plane_copy_up(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int w, int h, int depth, int shiftonly)
{
    int x, y;
    for (y = 0; y < h; y++) {
        const uint8_t *s = src + (long)y * src_stride;
        uint16_t *d = (uint16_t *)(dst + (long)y * dst_stride);
        for (x = 0; x < w; x++)
            d[x] = (uint16_t)(shiftonly ? s[x] << (depth - 8) : (s[x] << (depth - 8)) | (s[x] >> (2 * 8 - depth)));
    }
}

/* planarCopyWrapper, a deeper planar format into the 8-bit one of the same layout (swscale_unscaled.c:1743-1800 DITHER_COPY, dithers :40-113):
 * shift = depth - 8, d = dithers[shift - 1][row & 7][x & 7];  shiftonly (chroma planes, luma of a limited-range source):
 * t = (v + d) >> shift, out = t - (t >> 8);  luma of a full-range source: out = (v - (v >> 8) + d) >> shift.  Only the two tables the
 * served formats reach are restated: shift 2 (10 bits) and shift 8 (16 bits: the same values as ff_dither_8x8_128).  One plane per call;
 * taken at equal size when the two ranges agree (utils.c:1996-2000, swscale_unscaled.c:2293-2309). */
static const uint8_t orc_dithers_2[2][2] = { { 1, 2 }, { 3, 0 } };                         /* dithers[1]: rows / columns alternate */
static const uint8_t orc_dithers_8[8][8] = {
    {  36, 68,  60, 92,  34, 66,  58, 90 }, { 100,  4, 124, 28,  98,  2, 122, 26 }, {  52, 84,  44, 76,  50, 82,  42, 74 },
    { 116, 20, 108, 12, 114, 18, 106, 10 }, {  32, 64,  56, 88,  38, 70,  62, 94 }, {  96,  0, 120, 24, 102,  6, 126, 30 },
    {  48, 80,  40, 72,  54, 86,  46, 78 }, { 112, 16, 104,  8, 118, 22, 110, 14 },
};
void orc_plane_copy_down(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int w, int h, int depth, int shiftonly)
{
    const int shift = depth - 8;
    int x, y;
    for (y = 0; y < h; y++) {
        const uint16_t *s = (const uint16_t *)(src + (long)y * src_stride);
        uint8_t *d = dst + (long)y * dst_stride;
        for (x = 0; x < w; x++) {
            const unsigned dith = shift == 2 ? orc_dithers_2[y & 1][x & 1] : orc_dithers_8[y & 7][x & 7];
            unsigned t;
            if (shiftonly) { t = (s[x] + dith) >> shift; d[x] = (uint8_t)(t - (t >> 8)); }
            else           { t = s[x]; d[x] = (uint8_t)((t - (t >> 8) + dith) >> shift); }
        }
    }
}

/* planar8ToP01xleWrapper, swscale_unscaled.c:286-324 */
void orc_yuv420_to_p01x(const uint8_t *const src[4], const int src_stride[4], uint8_t *const dst[4],
                        const int dst_stride[4], int w, int h, int src_nv12)
{
    int x, y;
    for (y = 0; y < h; y++) {
        const uint8_t *s = src[0] + (long)y * src_stride[0];
        uint16_t *d = (uint16_t *)(dst[0] + (long)y * dst_stride[0]);
        for (x = 0; x < w; x++) d[x] = (uint16_t)(s[x] | (s[x] << 8));
        if (!(y & 1)) {
            uint16_t *duv = (uint16_t *)(dst[1] + (long)(y / 2) * dst_stride[1]);
            for (x = 0; x < w / 2; x++) {
                int u, v;
                if (src_nv12) {
                    u = src[1][(long)(y / 2) * src_stride[1] + 2 * x];
                    v = src[1][(long)(y / 2) * src_stride[1] + 2 * x + 1];
                } else {
                    u = src[1][(long)(y / 2) * src_stride[1] + x];
                    v = src[2][(long)(y / 2) * src_stride[2] + x];
                }
                duv[2 * x]     = (uint16_t)(u | (u << 8));
                duv[2 * x + 1] = (uint16_t)(v | (v << 8));
            }
        }
    }
}


/* ---- Gaussian blur, kw x kh with sigmaX / sigmaY and OpenCV border rules -----------------------------------------
 * What smooth_nvcv type=gaussian asks of CV-CUDA (vf_smooth_nvcv.c:88-105 options, :290-294 the operator call:
 * kernel size, sigma, border mode).  CV-CUDA is not in the reference tree and no reference test pins its output:
 * PARITY UNPINNED.  This restates the published OpenCV rule CV-CUDA documents itself as following:
 *   kernel  cv::getGaussianKernel: sigma <= 0 -> 0.3*((k-1)*0.5 - 1) + 0.8, fixed tables for k = 1,3,5,7, else
 *           exp(-x*x / (2 sigma^2)) / sum; sigmaY <= 0 -> sigmaX
 *   border  cv::borderInterpolate: 0 constant(0) 1 replicate 2 reflect 3 wrap 4 reflect101
 *   sum     float32, raster order over the window, term = (ky[j]*kx[i]) * pixel; out = clip((int)(sum + 0.5f)) */
static int orc_border(int p, int len, int border)
{
    if (p >= 0 && p < len) return p;
    if (border == 1) return p < 0 ? 0 : len - 1;
    if (border == 2 || border == 4) {
        int delta = border == 4;
        if (len == 1) return 0;
        while (p < 0 || p >= len) {
            if (p < 0) p = -p - 1 + delta;
            else p = len - 1 - (p - len) - delta;
        }
        return p;
    }
    if (border == 3) {
        while (p < 0) p += len;
        return p % len;
    }
    return -1;
}

static void orc_gauss_1d(int k, double sigma, float *out)
{
    static const double tab[4][7] = {{1.0}, {0.25, 0.5, 0.25}, {0.0625, 0.25, 0.375, 0.25, 0.0625},
                                     {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125}};
    double cf[256], total = 0, sg, s2;
    int i;
    if (sigma <= 0 && k <= 7) {
        for (i = 0; i < k; i++) out[i] = (float)tab[k >> 1][i];
        return;
    }
    sg = sigma > 0 ? sigma : 0.3 * ((k - 1) * 0.5 - 1) + 0.8;
    s2 = -0.5 / (sg * sg);
    for (i = 0; i < k; i++) {
        double x = i - (k - 1) * 0.5;
        cf[i] = exp(s2 * x * x);
        total += cf[i];
    }
    for (i = 0; i < k; i++) out[i] = (float)(cf[i] / total);
}

int orc_gauss_blur(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int w, int h, int bpp,
                   int kw, int kh, double sigma_x, double sigma_y, int border)
{
    float kx[256], ky[256];
    int x, y, ch, i, j;
    if (kw < 1 || kh < 1 || kw > 255 || kh > 255 || !(kw & 1) || !(kh & 1) || border < 0 || border > 4) return -1;
    orc_gauss_1d(kw, sigma_x, kx);
    orc_gauss_1d(kh, sigma_y > 0 ? sigma_y : sigma_x, ky);
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++)
            for (ch = 0; ch < bpp; ch++) {
                volatile float sum = 0.0f;
                int v;
                for (j = 0; j < kh; j++) {
                    int yy = orc_border(y + j - kh / 2, h, border);
                    for (i = 0; i < kw; i++) {
                        int xx = orc_border(x + i - kw / 2, w, border);
                        volatile float wgt = ky[j] * kx[i];
                        volatile float px = (yy < 0 || xx < 0) ? 0.0f : (float)src[(long)yy * src_stride + (long)xx * bpp + ch];
                        volatile float term = wgt * px;
                        sum = sum + term;
                    }
                }
                sum = sum + 0.5f;
                v = (int)sum;
                dst[(long)y * dst_stride + (long)x * bpp + ch] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            }
    return 0;
}
