/*
 * oracle/orc_sws.c — TEST INFRASTRUCTURE ONLY (see orc.h).
 *
 * CPU restatement of libswscale's *generic* scaler for the formats on the hot path:
 *   src: RGB24, BGR24, RGBA, BGRA, RGBA64LE, BGRA64LE, NV12, YUV420P, ...   dst: RGB24, BGR24, RGBA, BGRA, NV12, YUV420P, ...
 *
 * Follows (reference tree, ffmpeg-gpu/libswscale):
 *   sws_init_single_context   utils.c:1293-2020  chroma sub-sampling decisions :1427-1557,
 *                                                increments :1412-1413,:1588-1589,
 *                                                filter creation :1820-1875
 *   get_local_pos             utils.c:338-345
 *   fill_rgb2yuv_table        utils.c:765-858
 *   rgb24ToY_c/ToUV_c/ToUV_half_c, bgr24 twins, nvXXtoUV_c   input.c:795-866, :675-690
 *   rgb32 / bgr32 readers (rgb16_32To{Y,UV,UV_half}_c_template, S = RGB2YUV_SHIFT + 8)   input.c:246-390: the 24-bit
 *                                                            formulas on the same three channels (every term << 8)
 *   rgb64To{Y,UV,UV_half}_c_template (RGBA64LE / BGRA64LE)   input.c:36-121
 *   rgbaToA_c, rgba64leToA_c                                 input.c:413-449;  needAlpha  utils.c:1902
 *   alpha in the packed writers                              output.c:1025-1470 (64-bit), :1685-1828, :2037-2200 (8-bit)
 *   hScale8To15_c / hScale16To15_c                         swscale.c:93-136
 *   ff_hyscale_fast_c / ff_hcscale_fast_c (SWS_FAST_BILINEAR on 8-bit sources, selected swscale.c:566-574)   hscale_fast_bilinear.c:23-55
 *   swscale() row schedule                                 swscale.c:372-389
 *   packed_vscale selection                                vscale.c:108-170
 *   yuv2rgb_{X,2,1}_c_template + yuv2rgb_write             output.c:1554-1828
 *   yuv2rgb_full_{X,2,1}_c_template + yuv2rgb_write_full   output.c:1886-1935,:2037-2200
 *   yuv2planeX_8_c / yuv2plane1_8_c / yuv2nv12cX_c         output.c:400-450
 *   lum/chr planar vscale                                  vscale.c:30-105
 *
 * It models the portable C build (cpu_flags == 0, filterAlign == 1).  It always takes the
 * generic path: the special unscaled converters (ff_get_unscaled_swscale) are restated
 * separately (orc_yuv2rgb_frame, orc_rgb24_swap_rb).
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

struct OrcSws {
    int src_w, src_h, dst_w, dst_h, src_fmt, dst_fmt, flags;
    int src_full_range;                       /* planarCopyWrapper's luma rule (orc_sws_scale) */
    int chr_src_hsub, chr_src_vsub, chr_dst_hsub, chr_dst_vsub;
    int chr_src_w, chr_src_h, chr_dst_w, chr_dst_h;
    int lum_x_inc, lum_y_inc, chr_x_inc, chr_y_inc;
    int16_t *h_lum, *h_chr, *v_lum, *v_chr;
    int32_t *h_lum_pos, *h_chr_pos, *v_lum_pos, *v_chr_pos;
    int h_lum_size, h_chr_size, v_lum_size, v_chr_size;
    int src_is_rgb, dst_is_rgb;
    int src_px;                /* packed RGB source: bytes per pixel (3, 4, 8) */
    int need_alpha;            /* both ends carry alpha: the alpha plane is scaled with the luma filters (utils.c:1902) */
    int range_conv;            /* 0 none, 1 limited->full (ToJpeg), 2 full->limited (FromJpeg) */
    int32_t ry, gy, by, ru, gu, bu, rv, gv, bv;
    OrcYuv2Rgb y2r;
};

#define RGB2YUV_SHIFT 15    /* swscale_internal.h:452 */

static int is_rgb(int f)  { return f == ORC_PIX_RGB24 || f == ORC_PIX_BGR24 || f == ORC_PIX_RGBA || f == ORC_PIX_BGRA; }
static int is_p01x(int f) { return f == ORC_PIX_P010LE || f == ORC_PIX_P016LE; }
void orc_plane_copy_down(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int w, int h, int depth, int shiftonly);   /* orc_vf.c */
static int is_rgb64(int f) { return f == ORC_PIX_RGBA64LE || f == ORC_PIX_BGRA64LE; }
static int has_alpha(int f) { return f == ORC_PIX_RGBA || f == ORC_PIX_BGRA || is_rgb64(f); }
static int is_dst16(int f) { return f == ORC_PIX_P016LE || f == ORC_PIX_YUV444P16LE || f == ORC_PIX_YUV420P16LE || is_rgb64(f); }   /* 19-bit lines */
static int is_pl16_dst(int f) { return f == ORC_PIX_YUV444P16LE || f == ORC_PIX_YUV420P16LE; }          /* ... with planar chroma */
static int is_yuv(int f)  { return f == ORC_PIX_NV12 || f == ORC_PIX_YUV420P || f == ORC_PIX_YUV444P; }
/* planar YUV in 16-bit containers, native endian: no input converter (input.c:1523-1528 is HAVE_BIGENDIAN only), the
 * samples go to hScale16To15_c / hScale16To19_c as they are with sh derived from the depth (swscale.c:93-119,:63-91) */
static int pl16_depth(int f) { return f == ORC_PIX_YUV444P16LE || f == ORC_PIX_YUV420P16LE ? 16 : f == ORC_PIX_YUV420P10LE ? 10 : 0; }
static int fmt_sub(int f) { return (f == ORC_PIX_NV12 || f == ORC_PIX_YUV420P || is_p01x(f) || f == ORC_PIX_YUV420P16LE || f == ORC_PIX_YUV420P10LE) ? 1 : 0; }   /* log2_chroma_w == log2_chroma_h here */
static unsigned rl16(const uint8_t *p) { return (unsigned)p[0] | ((unsigned)p[1] << 8); }
static int ceil_rshift(int a, int b) { return -((-a) >> b); }
static int clip_u8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
static int clip_uintp2_30(int a)
{
    /* av_clip_uintp2_c(a, 30), libavutil/common.h */
    if (a & ~((1 << 30) - 1)) return (~a) >> 31 & ((1 << 30) - 1);
    return a;
}

static int get_local_pos(int chr_subsample, int pos)
{
    if (pos == -1 || pos <= -513)
        pos = (128 << chr_subsample) - 128;
    pos += 128;
    return pos >> chr_subsample;
}

static int64_t rounded_div(int64_t a, int64_t b)
{
    return a >= 0 ? (a + (b >> 1)) / b : (a - (b >> 1)) / b;
}

extern const int32_t *orc_get_coefficients(int colorspace);

static void fill_rgb2yuv(OrcSws *c, int colorspace)
{
    const int32_t *table = orc_get_coefficients(colorspace);
    const int32_t *def   = orc_get_coefficients(5);
    int64_t W, V, Z, Cy, Cu, Cv;
    int64_t vr = table[0], ub = table[1], ug = -table[2], vg = -table[3];
    const int64_t ONE = 65536;
    int64_t cy = ONE;

    cy = cy * 255 / 219;                         /* dstRange forced to 0, utils.c:811 */
    W = rounded_div(ONE * ONE * ug, ub);
    V = rounded_div(ONE * ONE * vg, vr);
    Z = ONE * ONE - W - V;
    Cy = rounded_div(cy * Z, ONE);
    Cu = rounded_div(ub * Z, ONE);
    Cv = rounded_div(vr * Z, ONE);

    c->ry = (int32_t)-rounded_div((1 << RGB2YUV_SHIFT) * V, Cy);
    c->gy = (int32_t) rounded_div((1 << RGB2YUV_SHIFT) * ONE * ONE, Cy);
    c->by = (int32_t)-rounded_div((1 << RGB2YUV_SHIFT) * W, Cy);
    c->ru = (int32_t) rounded_div((1 << RGB2YUV_SHIFT) * V, Cu);
    c->gu = (int32_t)-rounded_div((1 << RGB2YUV_SHIFT) * ONE * ONE, Cu);
    c->bu = (int32_t) rounded_div((1 << RGB2YUV_SHIFT) * (Z + W), Cu);
    c->rv = (int32_t) rounded_div((1 << RGB2YUV_SHIFT) * (V + Z), Cv);
    c->gv = (int32_t)-rounded_div((1 << RGB2YUV_SHIFT) * ONE * ONE, Cv);
    c->bv = (int32_t) rounded_div((1 << RGB2YUV_SHIFT) * W, Cv);

    if (!memcmp(table, def, 4 * sizeof(int32_t))) {               /* utils.c:845-856 */
        c->by =  ((int)(0.114 * 219 / 255 * (1 << RGB2YUV_SHIFT) + 0.5));
        c->bv = (-(int)(0.081 * 224 / 255 * (1 << RGB2YUV_SHIFT) + 0.5));
        c->bu =  ((int)(0.500 * 224 / 255 * (1 << RGB2YUV_SHIFT) + 0.5));
        c->gy =  ((int)(0.587 * 219 / 255 * (1 << RGB2YUV_SHIFT) + 0.5));
        c->gv = (-(int)(0.419 * 224 / 255 * (1 << RGB2YUV_SHIFT) + 0.5));
        c->gu = (-(int)(0.331 * 224 / 255 * (1 << RGB2YUV_SHIFT) + 0.5));
        c->ry =  ((int)(0.299 * 219 / 255 * (1 << RGB2YUV_SHIFT) + 0.5));
        c->rv =  ((int)(0.500 * 224 / 255 * (1 << RGB2YUV_SHIFT) + 0.5));
        c->ru = (-(int)(0.169 * 224 / 255 * (1 << RGB2YUV_SHIFT) + 0.5));
    }
}

void orc_sws_free(OrcSws *c)
{
    if (!c) return;
    free(c->h_lum); free(c->h_chr); free(c->v_lum); free(c->v_chr);
    free(c->h_lum_pos); free(c->h_chr_pos); free(c->v_lum_pos); free(c->v_chr_pos);
    free(c);
}

OrcSws *orc_sws_create(int src_w, int src_h, int src_fmt, int dst_w, int dst_h, int dst_fmt,
                       int flags, const double param[2])
{
    static const int def_pos[4] = { -513, -513, -513, -513 };
    return orc_sws_create_ex(src_w, src_h, src_fmt, dst_w, dst_h, dst_fmt, flags, param, def_pos, 0, 0);
}

/* chr_pos = { src_h_chr_pos, src_v_chr_pos, dst_h_chr_pos, dst_v_chr_pos } (AVOptions of the same
 * names, options.c:67-70; -513 = unset).  src_range / dst_range: 1 = full ("jpeg") range; only the
 * yuv->yuv lum/chrConvertRange hooks are restated (swscale.c:157-187, selection :530-556), the
 * RGB ends keep the limited-range BT.601 tables. */
OrcSws *orc_sws_create_ex(int src_w, int src_h, int src_fmt, int dst_w, int dst_h, int dst_fmt,
                          int flags, const double param[2], const int chr_pos[4], int src_range, int dst_range)
{
    OrcSws *c;
    int scaler_mask = ORC_SWS_FAST_BILINEAR | ORC_SWS_BILINEAR | ORC_SWS_BICUBIC | 8 | ORC_SWS_POINT |
                      ORC_SWS_AREA | 0x40 | 0x80 | 0x100 | ORC_SWS_LANCZOS | 0x400;

    /* P010LE / P016LE as sources and as destinations: P010LE (dstBpc = 10 <= 14) keeps the 15-bit intermediates,
     * P016LE switches to the 19-bit ones (scale_to_p016 below) */
    if (!(is_rgb(src_fmt) || is_rgb64(src_fmt) || is_yuv(src_fmt) || is_p01x(src_fmt) || pl16_depth(src_fmt)) ||
        !(is_rgb(dst_fmt) || is_yuv(dst_fmt) || is_p01x(dst_fmt) || is_dst16(dst_fmt) || dst_fmt == ORC_PIX_YUV420P10LE))
        return NULL;

    if (src_w < 1 || src_h < 1 || dst_w < 1 || dst_h < 1)
        return NULL;
    c = (OrcSws *)calloc(1, sizeof(*c));
    if (!c) return NULL;

    if (!(flags & scaler_mask))
        flags |= ORC_SWS_BICUBIC;                          /* utils.c:1370-1381 */
    c->src_w = src_w; c->src_h = src_h; c->dst_w = dst_w; c->dst_h = dst_h;
    c->src_fmt = src_fmt; c->dst_fmt = dst_fmt;
    c->src_is_rgb = is_rgb(src_fmt) || is_rgb64(src_fmt);
    c->src_px = is_rgb64(src_fmt) ? 8 : (src_fmt == ORC_PIX_RGBA || src_fmt == ORC_PIX_BGRA) ? 4 : 3;
    c->need_alpha = has_alpha(src_fmt) && has_alpha(dst_fmt);
    c->dst_is_rgb = is_rgb(dst_fmt) || is_rgb64(dst_fmt);

    c->lum_x_inc = (int)((((int64_t)src_w << 16) + (dst_w >> 1)) / dst_w);
    c->lum_y_inc = (int)((((int64_t)src_h << 16) + (dst_h >> 1)) / dst_h);

    c->chr_src_hsub = c->chr_src_vsub = fmt_sub(src_fmt);
    c->chr_dst_hsub = c->chr_dst_vsub = fmt_sub(dst_fmt);
    c->src_full_range = src_range;
    if (src_range != dst_range) {
        /* RGB ends have their range forced to 0 (utils.c:902-1030): an RGB source with a full-range YUV destination
         * is the limited -> full conversion of the 15-bit lines */
        if (c->dst_is_rgb || (c->src_is_rgb && src_range)) { free(c); return NULL; }
        c->range_conv = dst_range ? 1 : 2;
    }

    if (c->dst_is_rgb && !(flags & ORC_SWS_FULL_CHR_H_INT)) {      /* utils.c:1431-1448 */
        if (dst_w & 1)
            flags |= ORC_SWS_FULL_CHR_H_INT;
        if (c->chr_src_hsub == 0 && c->chr_src_vsub == 0 && !(flags & ORC_SWS_FAST_BILINEAR))
            flags |= ORC_SWS_FULL_CHR_H_INT;
    }
    if (!c->dst_is_rgb)
        flags &= ~ORC_SWS_FULL_CHR_H_INT;                          /* utils.c:1483-1517 */
    if (c->dst_is_rgb && !(flags & ORC_SWS_FULL_CHR_H_INT))
        c->chr_dst_hsub = 1;                                       /* :1519-1520 */
    if (c->src_is_rgb && !(flags & ORC_SWS_FULL_CHR_H_INP) &&
        ((dst_w >> c->chr_dst_hsub) <= (src_w >> 1) || (flags & ORC_SWS_FAST_BILINEAR)))
        c->chr_src_hsub = 1;                                       /* :1529-1545 */
    c->flags = flags;

    c->chr_src_w = ceil_rshift(src_w, c->chr_src_hsub);
    c->chr_src_h = ceil_rshift(src_h, c->chr_src_vsub);
    c->chr_dst_w = ceil_rshift(dst_w, c->chr_dst_hsub);
    c->chr_dst_h = ceil_rshift(dst_h, c->chr_dst_vsub);
    c->chr_x_inc = (int)((((int64_t)c->chr_src_w << 16) + (c->chr_dst_w >> 1)) / c->chr_dst_w);
    c->chr_y_inc = (int)((((int64_t)c->chr_src_h << 16) + (c->chr_dst_h >> 1)) / c->chr_dst_h);

    /* filters, utils.c:1820-1875; SWS_BICUBLIN (0x40): bicubic luma banks, bilinear chroma banks (:1830, 1841, 1860, 1869) */
    const int lflags = (flags & 0x40) ? (flags | ORC_SWS_BICUBIC) : flags, cflags = (flags & 0x40) ? (flags | ORC_SWS_BILINEAR) : flags;
    if (orc_init_filter(&c->h_lum, &c->h_lum_pos, &c->h_lum_size, c->lum_x_inc, src_w, dst_w, 1, 1 << 14,
                        lflags, param, get_local_pos(0, 0), get_local_pos(0, 0)) < 0) goto fail;
    if (orc_init_filter(&c->h_chr, &c->h_chr_pos, &c->h_chr_size, c->chr_x_inc, c->chr_src_w, c->chr_dst_w,
                        1, 1 << 14, cflags, param,
                        get_local_pos(c->chr_src_hsub, chr_pos[0]), get_local_pos(c->chr_dst_hsub, chr_pos[2])) < 0) goto fail;
    if (orc_init_filter(&c->v_lum, &c->v_lum_pos, &c->v_lum_size, c->lum_y_inc, src_h, dst_h, 1, 1 << 12,
                        lflags, param, get_local_pos(0, 0), get_local_pos(0, 0)) < 0) goto fail;
    if (orc_init_filter(&c->v_chr, &c->v_chr_pos, &c->v_chr_size, c->chr_y_inc, c->chr_src_h, c->chr_dst_h,
                        1, 1 << 12, cflags, param,
                        get_local_pos(c->chr_src_vsub, chr_pos[1]), get_local_pos(c->chr_dst_vsub, chr_pos[3])) < 0) goto fail;

    /* SWS_FAST_BILINEAR with 8-bit samples and 15-bit lines (srcBpc == 8 && dstBpc <= 14, swscale.c:566-574; an RGB source has
     * srcBpc 16): the horizontal scalers are ff_hyscale_fast_c / ff_hcscale_fast_c (hscale_fast_bilinear.c:23-55), which walk
     * xpos += xInc and blend src[xx], src[xx + 1] with the 7-bit xalpha = (xpos & 0xFFFF) >> 9:
     *     luma    (src[xx] << 7) + (src[xx + 1] - src[xx]) * xalpha   =  (src[xx] * (128 - xalpha) + src[xx + 1] * xalpha)
     *     chroma  src[xx] * (xalpha ^ 127) + src[xx + 1] * xalpha                (weights that sum to 127, not 128)
     *     every output with (i * xInc) >> 16 >= srcW - 1 is src[srcW - 1] * 128
     * i.e. hScale8To15_c over a two-tap bank {w0 << 7, w1 << 7}: the banks initFilter made are replaced by that one */
    if ((flags & ORC_SWS_FAST_BILINEAR) && !c->src_is_rgb && !is_p01x(src_fmt) && !pl16_depth(src_fmt) && !is_dst16(dst_fmt)) {
        int pass;
        for (pass = 0; pass < 2; pass++) {
            const int dw = pass ? c->chr_dst_w : dst_w, sw = pass ? c->chr_src_w : src_w, inc = pass ? c->chr_x_inc : c->lum_x_inc;
            int16_t *f = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)dw);
            int32_t *pos = (int32_t *)malloc(sizeof(int32_t) * (size_t)dw);
            unsigned xpos = 0;
            int i;
            if (!f || !pos) { free(f); free(pos); goto fail; }
            for (i = 0; i < dw; i++, xpos += (unsigned)inc) {
                const unsigned xx = xpos >> 16, xa = (xpos & 0xFFFF) >> 9;
                if ((int)(((unsigned)i * (unsigned)inc) >> 16) >= sw - 1) {          /* the tail rule */
                    pos[i] = sw >= 2 ? sw - 2 : 0;
                    f[2 * i] = sw >= 2 ? 0 : 16384; f[2 * i + 1] = sw >= 2 ? 16384 : 0;
                } else {
                    pos[i] = (int32_t)xx;
                    f[2 * i] = (int16_t)(((pass ? 127 : 128) - xa) << 7); f[2 * i + 1] = (int16_t)(xa << 7);
                }
            }
            if (pass) { free(c->h_chr); free(c->h_chr_pos); c->h_chr = f; c->h_chr_pos = pos; c->h_chr_size = 2; }
            else      { free(c->h_lum); free(c->h_lum_pos); c->h_lum = f; c->h_lum_pos = pos; c->h_lum_size = 2; }
        }
    }

    /* colour tables: BT.601, limited range on both sides (sws_setColorspaceDetails defaults,
     * utils.c:902-1030: RGB ends have their range forced to 0) */
    orc_yuv2rgb_init(&c->y2r, 5, 0, 0, 1 << 16, 1 << 16);
    fill_rgb2yuv(c, 5);
    return c;
fail:
    orc_sws_free(c);
    return NULL;
}

/* sws_setColorspaceDetails (utils.c:902-1030) for the two directions that have a matrix: a YUV source uses its
 * table for the YUV -> RGB stage (ff_yuv2rgb_c_init_tables), a YUV destination its table for the RGB -> YUV stage
 * (fill_rgb2yuv_table, utils.c:765-858).  Limited range on the YUV side; RGB -> RGB contexts keep the defaults. */
int orc_sws_set_colorspace(OrcSws *c, int colorspace)
{
    if (!c || colorspace < 0 || colorspace > 10) return -1;
    if (c->src_is_rgb && !c->dst_is_rgb) fill_rgb2yuv(c, colorspace);
    else if (!c->src_is_rgb && c->dst_is_rgb) orc_yuv2rgb_init(&c->y2r, colorspace, 0, 0, 1 << 16, 1 << 16);
    else return -1;
    return 0;
}

int orc_sws_filter(const OrcSws *c, int which, const int16_t **coef, const int32_t **pos, int *size, int *count)
{
    switch (which) {
    case 0: *coef = c->h_lum; *pos = c->h_lum_pos; *size = c->h_lum_size; *count = c->dst_w;     return 0;
    case 1: *coef = c->h_chr; *pos = c->h_chr_pos; *size = c->h_chr_size; *count = c->chr_dst_w; return 0;
    case 2: *coef = c->v_lum; *pos = c->v_lum_pos; *size = c->v_lum_size; *count = c->dst_h;     return 0;
    case 3: *coef = c->v_chr; *pos = c->v_chr_pos; *size = c->v_chr_size; *count = c->chr_dst_h; return 0;
    }
    return -1;
}

int orc_sws_info(const OrcSws *c, int *chr_src_w, int *chr_src_h, int *chr_dst_w, int *chr_dst_h, int *flags)
{
    *chr_src_w = c->chr_src_w; *chr_src_h = c->chr_src_h;
    *chr_dst_w = c->chr_dst_w; *chr_dst_h = c->chr_dst_h;
    *flags = c->flags;
    return 0;
}

/* ---- input stage: one source row -> 15-bit horizontally scaled lines -------------------- */

static void hscale8(int16_t *dst, int dst_w, const uint8_t *src, const int16_t *filter,
                    const int32_t *pos, int fs)
{
    int i, j;
    for (i = 0; i < dst_w; i++) {
        int val = 0;
        for (j = 0; j < fs; j++)
            val += ((int)src[pos[i] + j]) * filter[fs * i + j];
        val >>= 7;
        dst[i] = (int16_t)(val < (1 << 15) - 1 ? val : (1 << 15) - 1);
    }
}

static void hscale16(int16_t *dst, int dst_w, const uint16_t *src, const int16_t *filter,
                     const int32_t *pos, int fs, int sh)
{
    int i, j;
    for (i = 0; i < dst_w; i++) {
        int val = 0;
        for (j = 0; j < fs; j++)
            val += src[pos[i] + j] * filter[fs * i + j];
        val >>= sh;
        dst[i] = (int16_t)(val < (1 << 15) - 1 ? val : (1 << 15) - 1);
    }
}

static int imin(int a, int b) { return a < b ? a : b; }

static void range_lum(const OrcSws *c, int16_t *d, int w)
{
    int i;
    if (c->range_conv == 1)       /* lumRangeToJpeg_c, swscale.c:176-181 */
        for (i = 0; i < w; i++) d[i] = (int16_t)((imin(d[i], 30189) * 19077 - 39057361) >> 14);
    else if (c->range_conv == 2)  /* lumRangeFromJpeg_c, :183-188 */
        for (i = 0; i < w; i++) d[i] = (int16_t)((d[i] * 14071 + 33561947) >> 14);
}

static void range_chr(const OrcSws *c, int16_t *u, int16_t *v, int w)
{
    int i;
    if (c->range_conv == 1)       /* chrRangeToJpeg_c, :157-164 */
        for (i = 0; i < w; i++) {
            u[i] = (int16_t)((imin(u[i], 30775) * 4663 - 9289992) >> 12);
            v[i] = (int16_t)((imin(v[i], 30775) * 4663 - 9289992) >> 12);
        }
    else if (c->range_conv == 2)  /* chrRangeFromJpeg_c, :166-173 */
        for (i = 0; i < w; i++) {
            u[i] = (int16_t)((u[i] * 1799 + 4081085) >> 11);
            v[i] = (int16_t)((v[i] * 1799 + 4081085) >> 11);
        }
}

/* the 8-bit packed RGB readers: rgb24ToY_c / ToUV_c / ToUV_half_c and their bgr / 32-bit twins (input.c:795-866, :246-390) */
static void rgb8_lum(const OrcSws *c, const uint8_t *row, uint16_t *tmp)
{
    const int px = c->src_px;
    int ro = (c->src_fmt == ORC_PIX_RGB24 || c->src_fmt == ORC_PIX_RGBA) ? 0 : 2, bo = 2 - ro, i;
    for (i = 0; i < c->src_w; i++) {
        int r = row[px * i + ro], g = row[px * i + 1], b = row[px * i + bo];
        tmp[i] = (uint16_t)((c->ry * r + c->gy * g + c->by * b + (32 << (RGB2YUV_SHIFT - 1)) +
                             (1 << (RGB2YUV_SHIFT - 7))) >> (RGB2YUV_SHIFT - 6));
    }
}

static void rgb8_chr(const OrcSws *c, const uint8_t *row, uint16_t *tmp_u, uint16_t *tmp_v)
{
    const int px = c->src_px;
    int ro = (c->src_fmt == ORC_PIX_RGB24 || c->src_fmt == ORC_PIX_RGBA) ? 0 : 2, bo = 2 - ro, i;
    if (c->chr_src_hsub) {
        for (i = 0; i < c->chr_src_w; i++) {
            /* the reference reads pixel 2i+1 unconditionally; for odd widths that is the
             * padding of formatConvBuffer-less packed input, i.e. the next bytes of the row.
             * Odd source widths with half-chroma input are therefore not bit-defined and
             * the oracle clamps to the last pixel. */
            int i1 = 2 * i + 1 < c->src_w ? 2 * i + 1 : c->src_w - 1;
            int r = row[2 * px * i + ro] + row[px * i1 + ro];
            int g = row[2 * px * i + 1]  + row[px * i1 + 1];
            int b = row[2 * px * i + bo] + row[px * i1 + bo];
            tmp_u[i] = (uint16_t)((c->ru * r + c->gu * g + c->bu * b + (256 << RGB2YUV_SHIFT) +
                                   (1 << (RGB2YUV_SHIFT - 6))) >> (RGB2YUV_SHIFT - 5));
            tmp_v[i] = (uint16_t)((c->rv * r + c->gv * g + c->bv * b + (256 << RGB2YUV_SHIFT) +
                                   (1 << (RGB2YUV_SHIFT - 6))) >> (RGB2YUV_SHIFT - 5));
        }
    } else {
        for (i = 0; i < c->chr_src_w; i++) {
            int r = row[px * i + ro], g = row[px * i + 1], b = row[px * i + bo];
            tmp_u[i] = (uint16_t)((c->ru * r + c->gu * g + c->bu * b + (256 << (RGB2YUV_SHIFT - 1)) +
                                   (1 << (RGB2YUV_SHIFT - 7))) >> (RGB2YUV_SHIFT - 6));
            tmp_v[i] = (uint16_t)((c->rv * r + c->gv * g + c->bv * b + (256 << (RGB2YUV_SHIFT - 1)) +
                                   (1 << (RGB2YUV_SHIFT - 7))) >> (RGB2YUV_SHIFT - 6));
        }
    }
}

/* RGBA64LE / BGRA64LE readers (input.c:36-121): 16-bit channels, 16-bit results.
 *   Y  = (ry*r + gy*g + by*b + (0x2001 << 14)) >> 15
 *   UV = (ru*r + gu*g + bu*b + (0x10001 << 14)) >> 15, the _half form on (p0 + p1 + 1) >> 1 per channel
 * The sums are formed in the reference's own types: unsigned for Y (its channels are unsigned int), int for U / V. */
static void rgb64_lum(const OrcSws *c, const uint8_t *row, uint16_t *tmp)
{
    const int ro = c->src_fmt == ORC_PIX_RGBA64LE ? 0 : 4, bo = 4 - ro;
    int i;
    for (i = 0; i < c->src_w; i++) {
        unsigned r = rl16(row + 8 * i + ro), g = rl16(row + 8 * i + 2), b = rl16(row + 8 * i + bo);
        tmp[i] = (uint16_t)(((unsigned)c->ry * r + (unsigned)c->gy * g + (unsigned)c->by * b + (0x2001u << (RGB2YUV_SHIFT - 1))) >> RGB2YUV_SHIFT);
    }
}

static void rgb64_chr(const OrcSws *c, const uint8_t *row, uint16_t *tmp_u, uint16_t *tmp_v)
{
    const int ro = c->src_fmt == ORC_PIX_RGBA64LE ? 0 : 4, bo = 4 - ro;
    int i;
    for (i = 0; i < c->chr_src_w; i++) {
        int r, g, b;
        if (c->chr_src_hsub) {
            /* odd widths: pixel 2i+1 of the last pair is past the row in the reference; clamped here as for the 8-bit readers */
            const int i1 = 2 * i + 1 < c->src_w ? 2 * i + 1 : c->src_w - 1;
            r = (int)(rl16(row + 16 * i + ro) + rl16(row + 8 * i1 + ro) + 1) >> 1;
            g = (int)(rl16(row + 16 * i + 2)  + rl16(row + 8 * i1 + 2)  + 1) >> 1;
            b = (int)(rl16(row + 16 * i + bo) + rl16(row + 8 * i1 + bo) + 1) >> 1;
        } else {
            r = (int)rl16(row + 8 * i + ro); g = (int)rl16(row + 8 * i + 2); b = (int)rl16(row + 8 * i + bo);
        }
        tmp_u[i] = (uint16_t)((int)((unsigned)(c->ru * r) + (unsigned)(c->gu * g) + (unsigned)(c->bu * b) + (0x10001u << (RGB2YUV_SHIFT - 1))) >> RGB2YUV_SHIFT);
        tmp_v[i] = (uint16_t)((int)((unsigned)(c->rv * r) + (unsigned)(c->gv * g) + (unsigned)(c->bv * b) + (0x10001u << (RGB2YUV_SHIFT - 1))) >> RGB2YUV_SHIFT);
    }
}

/* the alpha samples as the horizontal scaler sees them: rgbaToA_c (a << 6 | a >> 2, input.c:442-449) for the 8-bit formats,
 * rgba64leToA_c (the sample, :413-421) for the 64-bit ones */
static void alpha_samples(const OrcSws *c, const uint8_t *row, uint16_t *tmp)
{
    int i;
    if (is_rgb64(c->src_fmt))
        for (i = 0; i < c->src_w; i++) tmp[i] = (uint16_t)rl16(row + 8 * i + 6);
    else
        for (i = 0; i < c->src_w; i++) tmp[i] = (uint16_t)(row[4 * i + 3] << 6 | row[4 * i + 3] >> 2);
}

/* lum_h_scale's alpha leg (hscale.c:66-76): the luma filter, hScale16To15_c with the source's sh; no range conversion */
static void alpha_line(const OrcSws *c, const uint8_t *const src[4], const int stride[4], int y, int16_t *out, uint16_t *tmp)
{
    alpha_samples(c, src[0] + (long)y * stride[0], tmp);
    hscale16(out, c->dst_w, tmp, c->h_lum, c->h_lum_pos, c->h_lum_size, is_rgb64(c->src_fmt) ? 15 : 13);
}

static void lum_line(const OrcSws *c, const uint8_t *const src[4], const int stride[4], int y,
                     int16_t *out, uint16_t *tmp)
{
    const uint8_t *row = src[0] + (long)y * stride[0];
    if (is_rgb64(c->src_fmt)) {
        rgb64_lum(c, row, tmp);
        hscale16(out, c->dst_w, tmp, c->h_lum, c->h_lum_pos, c->h_lum_size, 15);      /* sh = depth - 1 (swscale.c:93-119) */
    } else if (c->src_is_rgb) {
        rgb8_lum(c, row, tmp);
        hscale16(out, c->dst_w, tmp, c->h_lum, c->h_lum_pos, c->h_lum_size, 13);
    } else if (pl16_depth(c->src_fmt)) {
        /* native-endian 16-bit planar samples need no input converter; hScale16To15_c with sh = depth - 1 */
        int i;
        for (i = 0; i < c->src_w; i++) tmp[i] = (uint16_t)rl16(row + 2 * i);
        hscale16(out, c->dst_w, tmp, c->h_lum, c->h_lum_pos, c->h_lum_size, pl16_depth(c->src_fmt) - 1);
    } else if (is_p01x(c->src_fmt)) {
        /* p010LEToY_c (input.c:698-705): the 10 significant bits are the high ones, >> 6; P016LE has no converter
         * on a little-endian host (input.c:1523-1528 sits under HAVE_BIGENDIAN): the 16-bit samples as they are.
         * Then hScale16To15_c with sh = depth - 1 (swscale.c:93-119). */
        const int p010 = c->src_fmt == ORC_PIX_P010LE;
        int i;
        for (i = 0; i < c->src_w; i++)
            tmp[i] = (uint16_t)(p010 ? rl16(row + 2 * i) >> 6 : rl16(row + 2 * i));
        hscale16(out, c->dst_w, tmp, c->h_lum, c->h_lum_pos, c->h_lum_size, p010 ? 9 : 15);
    } else {
        hscale8(out, c->dst_w, row, c->h_lum, c->h_lum_pos, c->h_lum_size);
    }
    range_lum(c, out, c->dst_w);               /* hscale.c:60-61 */
}

static void chr_line(const OrcSws *c, const uint8_t *const src[4], const int stride[4], int y,
                     int16_t *out_u, int16_t *out_v, uint16_t *tmp_u, uint16_t *tmp_v)
{
    int i;
    if (is_rgb64(c->src_fmt)) {
        rgb64_chr(c, src[0] + (long)y * stride[0], tmp_u, tmp_v);
        hscale16(out_u, c->chr_dst_w, tmp_u, c->h_chr, c->h_chr_pos, c->h_chr_size, 15);
        hscale16(out_v, c->chr_dst_w, tmp_v, c->h_chr, c->h_chr_pos, c->h_chr_size, 15);
    } else if (c->src_is_rgb) {
        rgb8_chr(c, src[0] + (long)y * stride[0], tmp_u, tmp_v);
        hscale16(out_u, c->chr_dst_w, tmp_u, c->h_chr, c->h_chr_pos, c->h_chr_size, 13);
        hscale16(out_v, c->chr_dst_w, tmp_v, c->h_chr, c->h_chr_pos, c->h_chr_size, 13);
    } else if (pl16_depth(c->src_fmt)) {
        for (i = 0; i < c->chr_src_w; i++) {
            tmp_u[i] = (uint16_t)rl16(src[1] + (long)y * stride[1] + 2 * i);
            tmp_v[i] = (uint16_t)rl16(src[2] + (long)y * stride[2] + 2 * i);
        }
        hscale16(out_u, c->chr_dst_w, tmp_u, c->h_chr, c->h_chr_pos, c->h_chr_size, pl16_depth(c->src_fmt) - 1);
        hscale16(out_v, c->chr_dst_w, tmp_v, c->h_chr, c->h_chr_pos, c->h_chr_size, pl16_depth(c->src_fmt) - 1);
    } else if (is_p01x(c->src_fmt)) {
        /* p010LEToUV_c / p016LEToUV_c (input.c:716-747): interleaved 16-bit U, V */
        const uint8_t *row = src[1] + (long)y * stride[1];
        const int p010 = c->src_fmt == ORC_PIX_P010LE;
        for (i = 0; i < c->chr_src_w; i++) {
            tmp_u[i] = (uint16_t)(p010 ? rl16(row + 4 * i) >> 6 : rl16(row + 4 * i));
            tmp_v[i] = (uint16_t)(p010 ? rl16(row + 4 * i + 2) >> 6 : rl16(row + 4 * i + 2));
        }
        hscale16(out_u, c->chr_dst_w, tmp_u, c->h_chr, c->h_chr_pos, c->h_chr_size, p010 ? 9 : 15);
        hscale16(out_v, c->chr_dst_w, tmp_v, c->h_chr, c->h_chr_pos, c->h_chr_size, p010 ? 9 : 15);
    } else if (c->src_fmt == ORC_PIX_NV12) {
        const uint8_t *row = src[1] + (long)y * stride[1];
        uint8_t *u8 = (uint8_t *)tmp_u, *v8 = (uint8_t *)tmp_v;
        for (i = 0; i < c->chr_src_w; i++) {
            u8[i] = row[2 * i];
            v8[i] = row[2 * i + 1];
        }
        hscale8(out_u, c->chr_dst_w, u8, c->h_chr, c->h_chr_pos, c->h_chr_size);
        hscale8(out_v, c->chr_dst_w, v8, c->h_chr, c->h_chr_pos, c->h_chr_size);
    } else {
        hscale8(out_u, c->chr_dst_w, src[1] + (long)y * stride[1], c->h_chr, c->h_chr_pos, c->h_chr_size);
        hscale8(out_v, c->chr_dst_w, src[2] + (long)y * stride[2], c->h_chr, c->h_chr_pos, c->h_chr_size);
    }
    range_chr(c, out_u, out_v, c->chr_dst_w);  /* hscale.c:193-194 */
}

/* ---- output stage ------------------------------------------------------------------------ */

/* A: the alpha byte (255 without an alpha plane: the tables' own alpha / yuv2rgb_write_full's `hasAlpha ? A : 255`) */
static void put_rgb(uint8_t *d, int fmt, int R, int G, int B, int A)
{
    switch (fmt) {
    case ORC_PIX_RGB24: d[0] = (uint8_t)R; d[1] = (uint8_t)G; d[2] = (uint8_t)B; break;
    case ORC_PIX_BGR24: d[0] = (uint8_t)B; d[1] = (uint8_t)G; d[2] = (uint8_t)R; break;
    case ORC_PIX_RGBA:  d[0] = (uint8_t)R; d[1] = (uint8_t)G; d[2] = (uint8_t)B; d[3] = (uint8_t)A; break;
    case ORC_PIX_BGRA:  d[0] = (uint8_t)B; d[1] = (uint8_t)G; d[2] = (uint8_t)R; d[3] = (uint8_t)A; break;
    }
}

static void write_full(const OrcSws *c, uint8_t *d, int Y, int U, int V, int A)
{
    /* yuv2rgb_write_full, output.c:1886-1935 */
    int R, G, B;
    Y -= c->y2r.y_offset;
    Y *= c->y2r.y_coeff;
    Y += 1 << 21;
    R = (int)((unsigned)Y + (unsigned)(V * c->y2r.v2r));
    G = (int)((unsigned)Y + (unsigned)(V * c->y2r.v2g) + (unsigned)(U * c->y2r.u2g));
    B = (int)((unsigned)Y + (unsigned)(U * c->y2r.u2b));
    if ((R | G | B) & 0xC0000000) {
        R = clip_uintp2_30(R);
        G = clip_uintp2_30(G);
        B = clip_uintp2_30(B);
    }
    put_rgb(d, c->dst_fmt, R >> 22, G >> 22, B >> 22, A);
}

static void write_lut(const OrcSws *c, uint8_t *d, int Y, int U, int V, int A)
{
    /* yuv2rgb_write through table_rV/gU/gV/bU, output.c:1554-1600 */
    const OrcYuv2Rgb *t = &c->y2r;
    const uint8_t *r = t->y_table + t->off_rV[V + ORC_TABLE_HEADROOM];
    const uint8_t *g = t->y_table + t->off_gU[U + ORC_TABLE_HEADROOM] + t->off_gV[V + ORC_TABLE_HEADROOM];
    const uint8_t *b = t->y_table + t->off_bU[U + ORC_TABLE_HEADROOM];
    put_rgb(d, c->dst_fmt, r[Y], g[Y], b[Y], A);
}

/* lum[j], chr_u[j], chr_v[j] are the vertical taps' lines (already offset to the first tap); alp[j] the alpha lines on
 * the luma taps, NULL without an alpha plane.  Alpha per form (output.c):
 *   yuv2rgb_X_c :1709-1720      (2^18 + sum) >> 19, both of a pair clipped when either has bit 8 set
 *   yuv2rgb_2_c :1761-1766      (a0 * yalpha1 + a1 * yalpha) >> 19, clipped
 *   yuv2rgb_1_c :1794-1799      uvalpha < 2048: (a0 * 255 + 16384) >> 15;  :1816-1821 else: (a0 + 64) >> 7; clipped
 *   yuv2rgb_full_X_c :2069-2077 (2^18 + sum) >> 19;  _full_2_c :2117-2121  (a0 * yalpha1 + a1 * yalpha + 2^18) >> 19;
 *   yuv2rgb_full_1_c :2154-2158,:2171-2175  (a0 + 64) >> 7;  each clipped when bit 8 is set */
static void out_packed_row(const OrcSws *c, uint8_t *dest, int dst_y,
                           const int16_t *const *lum, const int16_t *const *chr_u,
                           const int16_t *const *chr_v, const int16_t *const *alp)
{
    const int step = (c->dst_fmt == ORC_PIX_RGB24 || c->dst_fmt == ORC_PIX_BGR24) ? 3 : 4;
    const int chr_y = dst_y >> c->chr_dst_vsub;
    const int lfs = c->v_lum_size, cfs = c->v_chr_size;
    const int16_t *lf = c->v_lum + dst_y * lfs;
    const int16_t *cf = c->v_chr + chr_y * cfs;
    const int full = !!(c->flags & ORC_SWS_FULL_CHR_H_INT);
    int i, j, mode, yalpha = 0, uvalpha = 0;

    /* packed_vscale(), vscale.c:135-167 */
    if (lfs == 1 && cfs == 1) {
        mode = 1; uvalpha = 0;
    } else if (lfs == 1 && cfs == 2 && cf[1] + cf[0] == 4096 && (unsigned)cf[1] <= 4096U) {
        mode = 1; uvalpha = cf[1];
    } else if (lfs == 2 && cfs == 2 && lf[1] + lf[0] == 4096 && (unsigned)lf[1] <= 4096U &&
               cf[1] + cf[0] == 4096 && (unsigned)cf[1] <= 4096U) {
        mode = 2; yalpha = lf[1]; uvalpha = cf[1];
    } else {
        mode = 3;
    }

    if (full) {
        for (i = 0; i < c->dst_w; i++) {
            int Y, U, V, A = 255;
            if (alp) {
                if (mode == 1) A = (alp[0][i] + 64) >> 7;
                else if (mode == 2) A = (alp[0][i] * (4096 - yalpha) + alp[1][i] * yalpha + (1 << 18)) >> 19;
                else { A = 1 << 18; for (j = 0; j < lfs; j++) A += alp[j][i] * lf[j]; A >>= 19; }
                if (A & 0x100) A = clip_u8(A);
            }
            if (mode == 1) {
                Y = lum[0][i] * 4;
                if (uvalpha < 2048) {
                    U = (chr_u[0][i] - (128 << 7)) * 4;
                    V = (chr_v[0][i] - (128 << 7)) * 4;
                } else {
                    U = (chr_u[0][i] + chr_u[1][i] - (128 << 8)) * 2;
                    V = (chr_v[0][i] + chr_v[1][i] - (128 << 8)) * 2;
                }
            } else if (mode == 2) {
                int yalpha1 = 4096 - yalpha, uvalpha1 = 4096 - uvalpha;
                Y = (lum[0][i] * yalpha1 + lum[1][i] * yalpha) >> 10;
                U = (chr_u[0][i] * uvalpha1 + chr_u[1][i] * uvalpha - (128 << 19)) >> 10;
                V = (chr_v[0][i] * uvalpha1 + chr_v[1][i] * uvalpha - (128 << 19)) >> 10;
            } else {
                Y = 1 << 9;
                U = (1 << 9) - (128 << 19);
                V = (1 << 9) - (128 << 19);
                for (j = 0; j < lfs; j++) Y += lum[j][i] * lf[j];
                for (j = 0; j < cfs; j++) {
                    U += chr_u[j][i] * cf[j];
                    V += chr_v[j][i] * cf[j];
                }
                Y >>= 10; U >>= 10; V >>= 10;
            }
            write_full(c, dest + i * step, Y, U, V, A);
        }
    } else {
        for (i = 0; i < ((c->dst_w + 1) >> 1); i++) {
            int Y1, Y2, U, V, A1 = 255, A2 = 255;
            if (alp) {
                const int16_t *a0 = alp[0];
                if (mode == 1 && uvalpha < 2048) {
                    A1 = clip_u8((a0[i * 2] * 255 + 16384) >> 15); A2 = clip_u8((a0[i * 2 + 1] * 255 + 16384) >> 15);
                } else if (mode == 1) {
                    A1 = clip_u8((a0[i * 2] + 64) >> 7); A2 = clip_u8((a0[i * 2 + 1] + 64) >> 7);
                } else if (mode == 2) {
                    A1 = clip_u8((a0[i * 2] * (4096 - yalpha) + alp[1][i * 2] * yalpha) >> 19);
                    A2 = clip_u8((a0[i * 2 + 1] * (4096 - yalpha) + alp[1][i * 2 + 1] * yalpha) >> 19);
                } else {
                    A1 = A2 = 1 << 18;
                    for (j = 0; j < lfs; j++) { A1 += alp[j][i * 2] * lf[j]; A2 += alp[j][i * 2 + 1] * lf[j]; }
                    A1 >>= 19; A2 >>= 19;
                    if ((A1 | A2) & 0x100) { A1 = clip_u8(A1); A2 = clip_u8(A2); }
                }
            }
            if (mode == 1) {
                Y1 = (lum[0][i * 2] + 64) >> 7;
                Y2 = (lum[0][i * 2 + 1] + 64) >> 7;
                if (uvalpha < 2048) {
                    U = (chr_u[0][i] + 64) >> 7;
                    V = (chr_v[0][i] + 64) >> 7;
                } else {
                    U = (chr_u[0][i] + chr_u[1][i] + 128) >> 8;
                    V = (chr_v[0][i] + chr_v[1][i] + 128) >> 8;
                }
            } else if (mode == 2) {
                int yalpha1 = 4096 - yalpha, uvalpha1 = 4096 - uvalpha;
                Y1 = (lum[0][i * 2]     * yalpha1 + lum[1][i * 2]     * yalpha) >> 19;
                Y2 = (lum[0][i * 2 + 1] * yalpha1 + lum[1][i * 2 + 1] * yalpha) >> 19;
                U  = (chr_u[0][i] * uvalpha1 + chr_u[1][i] * uvalpha) >> 19;
                V  = (chr_v[0][i] * uvalpha1 + chr_v[1][i] * uvalpha) >> 19;
            } else {
                Y1 = Y2 = U = V = 1 << 18;
                for (j = 0; j < lfs; j++) {
                    Y1 += lum[j][i * 2]     * lf[j];
                    Y2 += lum[j][i * 2 + 1] * lf[j];
                }
                for (j = 0; j < cfs; j++) {
                    U += chr_u[j][i] * cf[j];
                    V += chr_v[j][i] * cf[j];
                }
                Y1 >>= 19; Y2 >>= 19; U >>= 19; V >>= 19;
            }
            /* non-full packed output requires an even dst_w (odd forces full chroma) */
            write_lut(c, dest + (2 * i) * step, Y1, U, V, A1);
            write_lut(c, dest + (2 * i + 1) * step, Y2, U, V, A2);
        }
    }
}

/* swscale.c:36-46: the ordered dither of 8-bit planar output for sources deeper than 8 bits (row 8 repeats row 0 and is never indexed here) */
static const uint8_t orc_dither_8x8_128[8][8] = {
    {  36, 68,  60, 92,  34, 66,  58, 90 }, { 100,  4, 124, 28,  98,  2, 122, 26 },
    {  52, 84,  44, 76,  50, 82,  42, 74 }, { 116, 20, 108, 12, 114, 18, 106, 10 },
    {  32, 64,  56, 88,  38, 70,  62, 94 }, {  96,  0, 120, 24, 102,  6, 126, 30 },
    {  48, 80,  40, 72,  54, 86,  46, 78 }, { 112, 16, 104,  8, 118, 22, 110, 14 },
};
static const uint8_t orc_pb_64[8] = { 64, 64, 64, 64, 64, 64, 64, 64 };     /* sws_pb_64, swscale.c:48-50 */

/* swscale.c:263-264, 349-351, 482-485: should_dither = isNBPS(src) || is16BPS(src) — a source of 9 .. 16 bits (P010 / P016, planar 10 / 16 bit,
 * RGBA64 / BGRA64) dithers its 8-bit planar output with row dstY & 7 (luma) / chrDstY & 7 (chroma) of the table; an 8-bit source uses the constant 64.
 * Round 4: rounds 1-3 used 64 for every source — found by running the reference's real libswscale core (tests/fuzz/fuzz_ref_core.py). */
static int should_dither(const OrcSws *c) { return is_p01x(c->src_fmt) || pl16_depth(c->src_fmt) != 0 || is_rgb64(c->src_fmt); }
static const uint8_t *dither_row(const OrcSws *c, int y) { return should_dither(c) ? orc_dither_8x8_128[y & 7] : orc_pb_64; }

static void out_plane_row(uint8_t *dest, int w, const int16_t *filter, int fs,
                          const int16_t *const *src, const uint8_t *dither, int offset)
{
    /* lum_planar_vscale / chr_planar_vscale, vscale.c:30-105: 1 tap -> yuv2plane1_8_c (output.c:415-423), else yuv2planeX_8_c (:400-413);
     * the V plane's dither is offset by 3 (vscale.c:98,101) */
    int i, j;
    if (fs == 1) {
        for (i = 0; i < w; i++)
            dest[i] = (uint8_t)clip_u8((src[0][i] + dither[(i + offset) & 7]) >> 7);
    } else {
        for (i = 0; i < w; i++) {
            int val = dither[(i + offset) & 7] << 12;
            for (j = 0; j < fs; j++)
                val += src[j][i] * filter[j];
            dest[i] = (uint8_t)clip_u8(val >> 19);
        }
    }
}

static void out_nv12_chroma_row(uint8_t *dest, int w, const int16_t *filter, int fs,
                                const int16_t *const *su, const int16_t *const *sv, const uint8_t *dither)
{
    /* yuv2nv12cX_c, output.c:425-458 — always the X form, also for one tap (vscale.c:83-85) */
    int i, j;
    for (i = 0; i < w; i++) {
        int u = dither[i & 7] << 12, v = dither[(i + 3) & 7] << 12;
        for (j = 0; j < fs; j++) {
            u += su[j][i] * filter[j];
            v += sv[j][i] * filter[j];
        }
        dest[2 * i]     = (uint8_t)clip_u8(u >> 19);
        dest[2 * i + 1] = (uint8_t)clip_u8(v >> 19);
    }
}

/* P010 output, output.c:459-519 (output_pixel: clip to 10 bits, << 6, little endian):
 *   yuv2p010l1_c   val = src + (1 << 4);            >> 5
 *   yuv2p010lX_c   val = (1 << 16) + sum src*filter; >> 17
 *   yuv2p010cX_c   the same per chroma plane, U and V interleaved — always the X form (vscale.c:83-85) */
static void put_p010(uint8_t *d, int val, int shift)
{
    int v = val >> shift;
    v = v < 0 ? 0 : v > 1023 ? 1023 : v;
    v <<= 6;
    d[0] = (uint8_t)(v & 0xFF); d[1] = (uint8_t)(v >> 8);
}

/* planar 10-bit output (YUV420P10LE): yuv2plane1_10_c / yuv2planeX_10_c (output.c:330-384) — P010's arithmetic with the
 * sample left in the low bits; every plane, chroma included, goes through this (no dither at this depth) */
static void put_pl10(uint8_t *d, int val, int shift)
{
    int v = val >> shift;
    v = v < 0 ? 0 : v > 1023 ? 1023 : v;
    d[0] = (uint8_t)(v & 0xFF); d[1] = (uint8_t)(v >> 8);
}
static void out_pl10_row(uint8_t *dest, int w, const int16_t *filter, int fs, const int16_t *const *src)
{
    int i, j;
    if (fs == 1) {
        for (i = 0; i < w; i++) put_pl10(dest + 2 * i, src[0][i] + (1 << 4), 5);
    } else {
        for (i = 0; i < w; i++) {
            int val = 1 << 16;
            for (j = 0; j < fs; j++) val += src[j][i] * filter[j];
            put_pl10(dest + 2 * i, val, 17);
        }
    }
}

static void out_p010_luma_row(uint8_t *dest, int w, const int16_t *filter, int fs, const int16_t *const *src)
{
    int i, j;
    if (fs == 1) {
        for (i = 0; i < w; i++) put_p010(dest + 2 * i, src[0][i] + (1 << 4), 5);
    } else {
        for (i = 0; i < w; i++) {
            int val = 1 << 16;
            for (j = 0; j < fs; j++) val += src[j][i] * filter[j];
            put_p010(dest + 2 * i, val, 17);
        }
    }
}

static void out_p010_chroma_row(uint8_t *dest, int w, const int16_t *filter, int fs,
                                const int16_t *const *su, const int16_t *const *sv)
{
    int i, j;
    for (i = 0; i < w; i++) {
        int u = 1 << 16, v = 1 << 16;
        for (j = 0; j < fs; j++) { u += su[j][i] * filter[j]; v += sv[j][i] * filter[j]; }
        put_p010(dest + 4 * i, u, 17);
        put_p010(dest + 4 * i + 2, v, 17);
    }
}

/* ---- 16-bit destination (P016LE): 19-bit lines held in int32 (dstBpc = 16, utils.c:1561-1570) -----------------
 *   hScale8To19_c    swscale.c:138-153   min(sum >> 3, 2^19 - 1)
 *   hScale16To19_c   swscale.c:63-91     min(sum >> (depth - 5), 2^19 - 1)
 *   yuv2plane1_16_c  output.c:143-155    clip_uint16((src + 4) >> 3)
 *   yuv2planeX_16_c  output.c:157-181    0x8000 + clip_int16(((1 << 14) - 0x40000000 + sum src * (unsigned)filter) >> 15)
 *   yuv2nv12cX_16_c  output.c:183-211    the X form per chroma plane, interleaved, also for one tap
 * Whole frames only. */
static void hscale19(int32_t *dst, int dst_w, const uint16_t *src, const int16_t *filter, const int32_t *pos, int fs, int sh)
{
    int i, j;
    for (i = 0; i < dst_w; i++) {
        int val = 0;
        for (j = 0; j < fs; j++) val += src[pos[i] + j] * filter[fs * i + j];
        val >>= sh;
        dst[i] = val < (1 << 19) - 1 ? val : (1 << 19) - 1;
    }
}

/* range conversion of the 19-bit lines (swscale.c:189-226, selected :545-552), in the reference's own 32-bit arithmetic — the chroma
 * ToJpeg product passes 2^31 on its way and only the difference fits */
static void range19(const OrcSws *c, int32_t *d, int w, int chroma)
{
    int i;
    if (c->range_conv == 1 && !chroma)
        for (i = 0; i < w; i++) d[i] = (int)((unsigned)imin(d[i], 30189 << 4) * 4769U - (unsigned)(39057361 << 2)) >> 12;
    else if (c->range_conv == 2 && !chroma)
        for (i = 0; i < w; i++) d[i] = (int)((unsigned)d[i] * (unsigned)(14071 / 4) + (unsigned)((33561947 << 4) / 4)) >> 12;
    else if (c->range_conv == 1)
        for (i = 0; i < w; i++) d[i] = (int)((unsigned)imin(d[i], 30775 << 4) * 4663U - (unsigned)(9289992 << 4)) >> 12;
    else if (c->range_conv == 2)
        for (i = 0; i < w; i++) d[i] = (int)((unsigned)d[i] * 1799U + (unsigned)(4081085 << 4)) >> 11;
}

static void put16(uint8_t *d, int v) { d[0] = (uint8_t)(v & 0xFF); d[1] = (uint8_t)(v >> 8); }

static int planeX16(const int32_t *const *src, const int16_t *filter, int fs, int i)
{
    unsigned val = (1u << 14) - 0x40000000u;
    int j, v;
    for (j = 0; j < fs; j++) val += (unsigned)src[j][i] * (unsigned)(int)filter[j];
    v = (int)val >> 15;
    v = v < -32768 ? -32768 : v > 32767 ? 32767 : v;
    return 0x8000 + v;
}

/* the 19-bit lines of a whole frame: luma src_h x dst_w, chroma chr_src_h x chr_dst_w per plane */
static int make_lines19(OrcSws *c, const uint8_t *const src[4], const int src_stride[4], int32_t **pl, int32_t **pu, int32_t **pv, int32_t **pa)
{
    const int dw = c->dst_w, cdw = c->chr_dst_w, sh8 = 3;
    const int pl16 = pl16_depth(c->src_fmt) != 0;            /* planar 16-bit containers, read as they are (native endian) */
    const int src16 = is_p01x(c->src_fmt), p010 = c->src_fmt == ORC_PIX_P010LE;
    const int r64 = is_rgb64(c->src_fmt);                    /* rgb64To*_c give 16-bit lines: hScale16To19_c, sh = 16 - 5 */
    const int r8 = c->src_is_rgb && !r64;                    /* an 8-bit RGB source's 16-bit lines: sh = 9 (swscale.c:74-76) */
    const int sh = r8 ? 9 : r64 ? 11 : pl16 ? pl16_depth(c->src_fmt) - 5 : src16 ? (p010 ? 10 : 16) - 5 : sh8;
    int32_t *la = (pa && c->need_alpha) ? (int32_t *)malloc(sizeof(int32_t) * (size_t)dw * c->src_h) : NULL;
    int32_t *ly = (int32_t *)malloc(sizeof(int32_t) * (size_t)dw * c->src_h);
    int32_t *lu = (int32_t *)malloc(sizeof(int32_t) * (size_t)cdw * c->chr_src_h);
    int32_t *lv = (int32_t *)malloc(sizeof(int32_t) * (size_t)cdw * c->chr_src_h);
    uint16_t *t0 = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(c->src_w + 16));
    uint16_t *t1 = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(c->src_w + 16));
    int y, i;
    if (!ly || !lu || !lv || !t0 || !t1) { free(ly); free(lu); free(lv); free(t0); free(t1); return -1; }
    if (pa) *pa = la;
    for (y = 0; y < c->src_h; y++) {
        const uint8_t *row = src[0] + (long)y * src_stride[0];
        if (r64) {
            rgb64_lum(c, row, t0);
        } else if (r8) {
            rgb8_lum(c, row, t0);
        } else {
            for (i = 0; i < c->src_w; i++)
                t0[i] = (uint16_t)((src16 || pl16) ? (p010 ? rl16(row + 2 * i) >> 6 : rl16(row + 2 * i)) : row[i]);
        }
        hscale19(ly + (size_t)y * dw, dw, t0, c->h_lum, c->h_lum_pos, c->h_lum_size, sh);
        range19(c, ly + (size_t)y * dw, dw, 0);
        if (la) {
            alpha_samples(c, row, t0);
            hscale19(la + (size_t)y * dw, dw, t0, c->h_lum, c->h_lum_pos, c->h_lum_size, sh);
        }
    }
    for (y = 0; y < c->chr_src_h; y++) {
        if (r64) rgb64_chr(c, src[0] + (long)y * src_stride[0], t0, t1);
        if (r8) rgb8_chr(c, src[0] + (long)y * src_stride[0], t0, t1);
        for (i = 0; i < c->chr_src_w && !r64 && !r8; i++) {
            if (pl16) {
                t0[i] = (uint16_t)rl16(src[1] + (long)y * src_stride[1] + 2 * i);
                t1[i] = (uint16_t)rl16(src[2] + (long)y * src_stride[2] + 2 * i);
            } else if (src16) {
                const uint8_t *row = src[1] + (long)y * src_stride[1];
                t0[i] = (uint16_t)(p010 ? rl16(row + 4 * i) >> 6 : rl16(row + 4 * i));
                t1[i] = (uint16_t)(p010 ? rl16(row + 4 * i + 2) >> 6 : rl16(row + 4 * i + 2));
            } else if (c->src_fmt == ORC_PIX_NV12) {
                const uint8_t *row = src[1] + (long)y * src_stride[1];
                t0[i] = row[2 * i]; t1[i] = row[2 * i + 1];
            } else {
                t0[i] = src[1][(long)y * src_stride[1] + i]; t1[i] = src[2][(long)y * src_stride[2] + i];
            }
        }
        hscale19(lu + (size_t)y * cdw, cdw, t0, c->h_chr, c->h_chr_pos, c->h_chr_size, sh);
        hscale19(lv + (size_t)y * cdw, cdw, t1, c->h_chr, c->h_chr_pos, c->h_chr_size, sh);
        range19(c, lu + (size_t)y * cdw, cdw, 1);
        range19(c, lv + (size_t)y * cdw, cdw, 1);
    }
    free(t0); free(t1);
    *pl = ly; *pu = lu; *pv = lv;
    return 0;
}

static int scale_to_p016(OrcSws *c, const uint8_t *const src[4], const int src_stride[4], uint8_t *const dst[4],
                         const int dst_stride[4])
{
    const int dw = c->dst_w, cdw = c->chr_dst_w;
    int32_t *ly = NULL, *lu = NULL, *lv = NULL;
    const int32_t **lp = (const int32_t **)malloc(sizeof(*lp) * (size_t)(c->v_lum_size + c->v_chr_size) * 2);
    int y, i, j, ret = -1;
    if (!lp || make_lines19(c, src, src_stride, &ly, &lu, &lv, NULL) < 0) goto done;
    for (y = 0; y < c->dst_h; y++) {
        uint8_t *d = dst[0] + (long)y * dst_stride[0];
        for (j = 0; j < c->v_lum_size; j++) {
            int r = c->v_lum_pos[y] + j;
            lp[j] = ly + (size_t)(r < c->src_h ? r : c->src_h - 1) * dw;
        }
        for (i = 0; i < dw; i++) {
            if (c->v_lum_size == 1) {
                int v = (lp[0][i] + 4) >> 3;
                put16(d + 2 * i, v < 0 ? 0 : v > 65535 ? 65535 : v);
            } else {
                put16(d + 2 * i, planeX16(lp, c->v_lum + y * c->v_lum_size, c->v_lum_size, i));
            }
        }
    }
    for (y = 0; y < c->chr_dst_h; y++) {
        uint8_t *d = dst[1] + (long)y * dst_stride[1];
        const int32_t **up = lp, **vp = lp + c->v_chr_size;
        if (is_pl16_dst(c->dst_fmt)) {                      /* planar chroma: yuv2plane1_16_c / yuv2planeX_16_c per plane */
            uint8_t *dv = dst[2] + (long)y * dst_stride[2];
            for (j = 0; j < c->v_chr_size; j++) {
                int r = c->v_chr_pos[y] + j;
                if (r >= c->chr_src_h) r = c->chr_src_h - 1;
                up[j] = lu + (size_t)r * cdw; vp[j] = lv + (size_t)r * cdw;
            }
            for (i = 0; i < cdw; i++) {
                if (c->v_chr_size == 1) {
                    int a = (up[0][i] + 4) >> 3, b = (vp[0][i] + 4) >> 3;
                    put16(d + 2 * i, a < 0 ? 0 : a > 65535 ? 65535 : a);
                    put16(dv + 2 * i, b < 0 ? 0 : b > 65535 ? 65535 : b);
                } else {
                    put16(d + 2 * i, planeX16(up, c->v_chr + y * c->v_chr_size, c->v_chr_size, i));
                    put16(dv + 2 * i, planeX16(vp, c->v_chr + y * c->v_chr_size, c->v_chr_size, i));
                }
            }
            continue;
        }
        for (j = 0; j < c->v_chr_size; j++) {
            int r = c->v_chr_pos[y] + j;
            if (r >= c->chr_src_h) r = c->chr_src_h - 1;
            up[j] = lu + (size_t)r * cdw; vp[j] = lv + (size_t)r * cdw;
        }
        for (i = 0; i < cdw; i++) {
            put16(d + 4 * i, planeX16(up, c->v_chr + y * c->v_chr_size, c->v_chr_size, i));
            put16(d + 4 * i + 2, planeX16(vp, c->v_chr + y * c->v_chr_size, c->v_chr_size, i));
        }
    }
    ret = c->dst_h;
done:
    free(ly); free(lu); free(lv); free((void *)lp);
    return ret;
}

/* ---- RGBA64LE / BGRA64LE destinations on the 19-bit lines -------------------------------------------------------
 * packed_vscale's choice of form (vscale.c:135-167) and the three forms of each chroma mode, output.c:
 *   yuv2rgba64_X_c :1025-1105   _2_c :1107-1170   _1_c :1172-1272   (one chroma sample per pixel PAIR)
 *   yuv2rgba64_full_X_c :1275-1337, _full_2_c, _full_1_c             (one chroma sample per pixel)
 * and their common colour stage: Y = (Y - y_offset) * y_coeff + (1 << 13); R = V * v2r; G = V * v2g + U * u2g;
 * B = U * u2b; channel = clip_uintp2(X + Y, 30) >> 14.  Alpha: 0xFFFF << 14 without an alpha plane, else per form (the same
 * rule in the half- and full-chroma twins): X :1052-1064 ((-2^30 + sum) >> 1) + 0x20002000; _2 :1144-1150
 * ((a0 * yalpha1 + a1 * yalpha) >> 1) + 2^13; _1 :1196-1202,:1242-1248 (a0 << 11) + 2^13; then clip_uintp2(A, 30) >> 14. */
static int clip_uintp2_30b(int a) { return (a & ~0x3FFFFFFF) ? (~a >> 31) & 0x3FFFFFFF : a; }

static void put_rgba64(const OrcSws *c, uint8_t *d, int Y, int U, int V, int A)
{
    int R, G, B, o[3], k;
    Y -= c->y2r.y_offset;
    Y *= c->y2r.y_coeff;
    Y += 1 << 13;
    R = V * c->y2r.v2r;
    G = V * c->y2r.v2g + U * c->y2r.u2g;
    B = U * c->y2r.u2b;
    o[0] = clip_uintp2_30b((c->dst_fmt == ORC_PIX_RGBA64LE ? R : B) + Y) >> 14;
    o[1] = clip_uintp2_30b(G + Y) >> 14;
    o[2] = clip_uintp2_30b((c->dst_fmt == ORC_PIX_RGBA64LE ? B : R) + Y) >> 14;
    for (k = 0; k < 3; k++) put16(d + 2 * k, o[k]);
    put16(d + 6, clip_uintp2_30b(A) >> 14);
}

static int scale_to_rgba64(OrcSws *c, const uint8_t *const src[4], const int src_stride[4], uint8_t *const dst[4],
                           const int dst_stride[4])
{
    const int dw = c->dst_w, cdw = c->chr_dst_w, full = (c->flags & ORC_SWS_FULL_CHR_H_INT) != 0;
    const int lfs = c->v_lum_size, cfs = c->v_chr_size;
    int32_t *ly = NULL, *lu = NULL, *lv = NULL, *la = NULL;
    const int32_t **lp = (const int32_t **)malloc(sizeof(*lp) * (size_t)(2 * lfs + 2 * cfs));
    int y, i, j, ret = -1;
    if (!lp || make_lines19(c, src, src_stride, &ly, &lu, &lv, &la) < 0) goto done;
    for (y = 0; y < c->dst_h; y++) {
        uint8_t *d = dst[0] + (long)y * dst_stride[0];
        const int16_t *lf = c->v_lum + y * lfs, *cf = c->v_chr + y * cfs;
        const int32_t **up = lp + lfs, **vp = up + cfs, **ap = vp + cfs;
        const int chr2 = cfs == 2 && cf[0] + cf[1] == 4096 && (unsigned)cf[1] <= 4096u;
        const int lum2 = lfs == 2 && lf[0] + lf[1] == 4096 && (unsigned)lf[1] <= 4096u;
        for (j = 0; j < lfs; j++) {
            int r = c->v_lum_pos[y] + j;
            lp[j] = ly + (size_t)(r < c->src_h ? r : c->src_h - 1) * dw;
            if (la) ap[j] = la + (size_t)(r < c->src_h ? r : c->src_h - 1) * dw;
        }
        for (j = 0; j < cfs; j++) {
            int r = c->v_chr_pos[y] + j;
            if (r >= c->chr_src_h) r = c->chr_src_h - 1;
            up[j] = lu + (size_t)r * cdw; vp[j] = lv + (size_t)r * cdw;
        }
        for (i = 0; i < dw; i++) {
            const int ci = full ? i : i >> 1;
            int Y, U, V, A = 0xffff << 14;
            if (lfs == 1 && (cfs == 1 || chr2)) {                         /* yuv2packed1, uvalpha = 0 or cf[1] */
                const int uvalpha = cfs == 1 ? 0 : cf[1];
                Y = lp[0][i] >> 2;
                if (la) A = (ap[0][i] << 11) + (1 << 13);
                if (uvalpha < 2048) {
                    U = (up[0][ci] - (128 << 11)) >> 2;
                    V = (vp[0][ci] - (128 << 11)) >> 2;
                } else {
                    U = (up[0][ci] + up[1][ci] - (128 << 12)) >> 3;
                    V = (vp[0][ci] + vp[1][ci] - (128 << 12)) >> 3;
                }
            } else if (lum2 && chr2) {                                     /* yuv2packed2 */
                Y = (lp[0][i] * (4096 - lf[1]) + lp[1][i] * lf[1]) >> 14;
                if (la) A = ((ap[0][i] * (4096 - lf[1]) + ap[1][i] * lf[1]) >> 1) + (1 << 13);
                U = (up[0][ci] * (4096 - cf[1]) + up[1][ci] * cf[1] - (128 << 23)) >> 14;
                V = (vp[0][ci] * (4096 - cf[1]) + vp[1][ci] * cf[1] - (128 << 23)) >> 14;
            } else {                                                       /* yuv2packedX, 32-bit wrap-around sums */
                unsigned ay = (unsigned)-0x40000000, au = (unsigned)-(128 << 23), av = au;
                for (j = 0; j < lfs; j++) ay += (unsigned)lp[j][i] * (unsigned)(int)lf[j];
                for (j = 0; j < cfs; j++) {
                    au += (unsigned)up[j][ci] * (unsigned)(int)cf[j];
                    av += (unsigned)vp[j][ci] * (unsigned)(int)cf[j];
                }
                Y = ((int)ay >> 14) + 0x10000;
                if (la) {
                    unsigned aa = (unsigned)-0x40000000;
                    for (j = 0; j < lfs; j++) aa += (unsigned)ap[j][i] * (unsigned)(int)lf[j];
                    A = ((int)aa >> 1) + 0x20002000;
                }
                U = (int)au >> 14;
                V = (int)av >> 14;
            }
            put_rgba64(c, d + 8 * i, Y, U, V, A);
        }
    }
    ret = c->dst_h;
done:
    free(ly); free(lu); free(lv); free(la); free((void *)lp);
    return ret;
}

int orc_sws_scale_rows(OrcSws *c, const uint8_t *const src[4], const int src_stride[4],
                       uint8_t *const dst[4], const int dst_stride[4], int y0, int y1)
{
    int lum_first, lum_last, chr_first, chr_last, y, j, ret = -1;
    int cy0, cy1;
    int16_t *lum_buf = NULL, *u_buf = NULL, *v_buf = NULL, *a_buf = NULL;
    uint16_t *tmp = NULL, *tmp_u = NULL, *tmp_v = NULL;
    const int16_t **lp = NULL, **up = NULL, **vp = NULL, **ap = NULL;
    const int dst_w = c->dst_w, cdw = c->chr_dst_w;

    if (c->dst_fmt == ORC_PIX_P016LE || is_pl16_dst(c->dst_fmt))
        return (y0 <= 0 && y1 >= c->dst_h) ? scale_to_p016(c, src, src_stride, dst, dst_stride) : -1;
    if (is_rgb64(c->dst_fmt))
        return (y0 <= 0 && y1 >= c->dst_h) ? scale_to_rgba64(c, src, src_stride, dst, dst_stride) : -1;
    if (y0 < 0) y0 = 0;
    if (y1 > c->dst_h) y1 = c->dst_h;
    if (y0 >= y1) return 0;
    /* planar 4:2:0 output rows come in pairs sharing one chroma row */
    if (!c->dst_is_rgb && c->chr_dst_vsub && (y0 & 1)) return -1;

    cy0 = y0 >> c->chr_dst_vsub;
    cy1 = ((y1 - 1) >> c->chr_dst_vsub) + 1;
    lum_first = c->v_lum_pos[y0];
    lum_last  = c->v_lum_pos[y1 - 1] + c->v_lum_size - 1;
    chr_first = c->v_chr_pos[cy0];
    chr_last  = c->v_chr_pos[cy1 - 1] + c->v_chr_size - 1;
    if (lum_last >= c->src_h)     lum_last = c->src_h - 1;
    if (chr_last >= c->chr_src_h) chr_last = c->chr_src_h - 1;

    lum_buf = (int16_t *)malloc(sizeof(int16_t) * (size_t)dst_w * (lum_last - lum_first + 1));
    a_buf   = (int16_t *)malloc(sizeof(int16_t) * (size_t)dst_w * (lum_last - lum_first + 1));
    ap = (const int16_t **)malloc(sizeof(*ap) * c->v_lum_size);
    u_buf   = (int16_t *)malloc(sizeof(int16_t) * (size_t)cdw * (chr_last - chr_first + 1));
    v_buf   = (int16_t *)malloc(sizeof(int16_t) * (size_t)cdw * (chr_last - chr_first + 1));
    tmp     = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(c->src_w + 16));
    tmp_u   = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(c->src_w + 16));
    tmp_v   = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(c->src_w + 16));
    lp = (const int16_t **)malloc(sizeof(*lp) * c->v_lum_size);
    up = (const int16_t **)malloc(sizeof(*up) * c->v_chr_size);
    vp = (const int16_t **)malloc(sizeof(*vp) * c->v_chr_size);
    if (!lum_buf || !a_buf || !ap || !u_buf || !v_buf || !tmp || !tmp_u || !tmp_v || !lp || !up || !vp)
        goto done;

    for (y = lum_first; y <= lum_last; y++)
        lum_line(c, src, src_stride, y, lum_buf + (size_t)(y - lum_first) * dst_w, tmp);
    for (y = lum_first; y <= lum_last && c->need_alpha; y++)
        alpha_line(c, src, src_stride, y, a_buf + (size_t)(y - lum_first) * dst_w, tmp);
    for (y = chr_first; y <= chr_last; y++)
        chr_line(c, src, src_stride, y, u_buf + (size_t)(y - chr_first) * cdw,
                 v_buf + (size_t)(y - chr_first) * cdw, tmp_u, tmp_v);

    for (y = y0; y < y1; y++) {
        const int chr_y = y >> c->chr_dst_vsub;
        for (j = 0; j < c->v_lum_size; j++) {
            int r = c->v_lum_pos[y] + j;
            if (r >= c->src_h) r = c->src_h - 1;
            lp[j] = lum_buf + (size_t)(r - lum_first) * dst_w;
            ap[j] = a_buf + (size_t)(r - lum_first) * dst_w;
        }
        for (j = 0; j < c->v_chr_size; j++) {
            int r = c->v_chr_pos[chr_y] + j;
            if (r >= c->chr_src_h) r = c->chr_src_h - 1;
            up[j] = u_buf + (size_t)(r - chr_first) * cdw;
            vp[j] = v_buf + (size_t)(r - chr_first) * cdw;
        }
        if (c->dst_is_rgb) {
            out_packed_row(c, dst[0] + (long)y * dst_stride[0], y, lp, up, vp, c->need_alpha ? ap : NULL);
        } else if (c->dst_fmt == ORC_PIX_YUV420P10LE) {
            out_pl10_row(dst[0] + (long)y * dst_stride[0], dst_w, c->v_lum + y * c->v_lum_size, c->v_lum_size, lp);
            if (!(y & 1)) {
                out_pl10_row(dst[1] + (long)chr_y * dst_stride[1], cdw, c->v_chr + chr_y * c->v_chr_size, c->v_chr_size, up);
                out_pl10_row(dst[2] + (long)chr_y * dst_stride[2], cdw, c->v_chr + chr_y * c->v_chr_size, c->v_chr_size, vp);
            }
        } else if (c->dst_fmt == ORC_PIX_P010LE) {
            out_p010_luma_row(dst[0] + (long)y * dst_stride[0], dst_w, c->v_lum + y * c->v_lum_size, c->v_lum_size, lp);
            if (!(y & 1))
                out_p010_chroma_row(dst[1] + (long)chr_y * dst_stride[1], cdw, c->v_chr + chr_y * c->v_chr_size,
                                    c->v_chr_size, up, vp);
        } else {
            out_plane_row(dst[0] + (long)y * dst_stride[0], dst_w, c->v_lum + y * c->v_lum_size,
                          c->v_lum_size, lp, dither_row(c, y), 0);
            if (!c->chr_dst_vsub || !(y & 1)) {
                const int16_t *cf = c->v_chr + chr_y * c->v_chr_size;
                if (c->dst_fmt == ORC_PIX_NV12) {
                    out_nv12_chroma_row(dst[1] + (long)chr_y * dst_stride[1], cdw, cf, c->v_chr_size, up, vp, dither_row(c, chr_y));
                } else {
                    out_plane_row(dst[1] + (long)chr_y * dst_stride[1], cdw, cf, c->v_chr_size, up, dither_row(c, chr_y), 0);
                    out_plane_row(dst[2] + (long)chr_y * dst_stride[2], cdw, cf, c->v_chr_size, vp, dither_row(c, chr_y), 3);
                }
            }
        }
    }
    ret = y1 - y0;
done:
    free(lum_buf); free(a_buf); free(u_buf); free(v_buf); free(tmp); free(tmp_u); free(tmp_v);
    free((void *)lp); free((void *)up); free((void *)vp); free((void *)ap);
    return ret;
}

int orc_sws_scale(OrcSws *c, const uint8_t *const src[4], const int src_stride[4],
                  uint8_t *const dst[4], const int dst_stride[4])
{
    /* bounded working set: process in bands of 64 output rows */
    int y, band = 64;
    /* the one unscaled special converter a context of THIS entry point takes by itself (the others are the callers' choice: orc_plane_copy_up,
     * orc_yuv420_to_p01x, orc_yuv2rgb): equal size, a deeper planar format into the 8-bit one of the same layout, ranges equal (utils.c:1996-2000,
     * swscale_unscaled.c:2293-2309) -> planarCopyWrapper's dithered copy, luma rule by the source's range */
    if (c->src_w == c->dst_w && c->src_h == c->dst_h && !c->range_conv &&
        (((c->src_fmt == ORC_PIX_YUV420P10LE || c->src_fmt == ORC_PIX_YUV420P16LE) && c->dst_fmt == ORC_PIX_YUV420P) ||
         (c->src_fmt == ORC_PIX_YUV444P16LE && c->dst_fmt == ORC_PIX_YUV444P))) {
        const int depth = pl16_depth(c->src_fmt), sub = c->src_fmt == ORC_PIX_YUV444P16LE ? 0 : 1;
        const int cw = (c->src_w + sub) >> sub, ch = (c->src_h + sub) >> sub;
        orc_plane_copy_down(src[0], src_stride[0], dst[0], dst_stride[0], c->src_w, c->src_h, depth, !c->src_full_range);
        orc_plane_copy_down(src[1], src_stride[1], dst[1], dst_stride[1], cw, ch, depth, 1);
        orc_plane_copy_down(src[2], src_stride[2], dst[2], dst_stride[2], cw, ch, depth, 1);
        return c->dst_h;
    }
    if (c->dst_fmt == ORC_PIX_P016LE || is_pl16_dst(c->dst_fmt))
        return scale_to_p016(c, src, src_stride, dst, dst_stride);
    if (is_rgb64(c->dst_fmt))
        return scale_to_rgba64(c, src, src_stride, dst, dst_stride);
    for (y = 0; y < c->dst_h; y += band) {
        int r = orc_sws_scale_rows(c, src, src_stride, dst, dst_stride, y,
                                   y + band < c->dst_h ? y + band : c->dst_h);
        if (r < 0) return r;
    }
    return c->dst_h;
}
