/*
 * oracle/orc_yuv2rgb.c — TEST INFRASTRUCTURE ONLY (see orc.h).
 *
 * Restates libswscale/yuv2rgb.c of the reference tree:
 *   ff_yuv2rgb_coeffs            yuv2rgb.c:48-60
 *   fill_table / fill_gv_table   yuv2rgb.c:737-760
 *   roundToInt16                 yuv2rgb.c:762-772
 *   ff_yuv2rgb_c_init_tables     yuv2rgb.c:774-1030 (24/32 bpp branches)
 *   LOADCHROMA / PUTRGB24        yuv2rgb.c:69-91
 *   yuv2rgb_c_24_rgb/_bgr/_32    yuv2rgb.c:240-405
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

/* yuv2rgb.c:48-60 — {crv, cbu, cgu, cgv} per SWS_CS_* index */
static const int32_t orc_coeffs[11][4] = {
    { 117489, 138438, 13975, 34925 },
    { 117489, 138438, 13975, 34925 },
    { 104597, 132201, 25675, 53279 },
    { 104597, 132201, 25675, 53279 },
    { 104448, 132798, 24759, 53109 },
    { 104597, 132201, 25675, 53279 },
    { 104597, 132201, 25675, 53279 },
    { 117579, 136230, 16907, 35559 },
    {      0,      0,     0,     0 },
    { 110013, 140363, 12277, 42626 },
    { 110013, 140363, 12277, 42626 },
};

const int32_t *orc_get_coefficients(int colorspace)
{
    /* sws_getCoefficients, yuv2rgb.c:62-67 */
    if (colorspace > 10 || colorspace < 0 || colorspace == 8)
        colorspace = 5;
    return orc_coeffs[colorspace];
}

static int clip_u8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

static int16_t round_to_int16(int64_t f)
{
    int r = (int)((f + (1 << 15)) >> 16);
    if (r < -0x7FFF) return (int16_t)0x8000;
    if (r >  0x7FFF) return 0x7FFF;
    return (int16_t)r;
}

int orc_yuv2rgb_init(OrcYuv2Rgb *t, int colorspace, int full_range,
                     int brightness, int contrast, int saturation)
{
    const int32_t *inv = orc_get_coefficients(colorspace);
    const int table_plane_size = 1024 + 2 * ORC_TABLE_LUMA_HEADROOM;
    int64_t crv =  inv[0], cbu =  inv[1], cgu = -inv[2], cgv = -inv[3];
    int64_t cy = 1 << 16, oy = 0, yb;
    int i;

    memset(t, 0, sizeof(*t));
    t->full_range = full_range;
    t->yoffs = (full_range ? 384 : 326) + ORC_TABLE_LUMA_HEADROOM;

    if (!full_range) {
        cy = (cy * 255) / 219;
        oy = 16 << 16;
    } else {
        crv = (crv * 224) / 255;
        cbu = (cbu * 224) / 255;
        cgu = (cgu * 224) / 255;
        cgv = (cgv * 224) / 255;
    }
    cy  = (cy  * contrast)              >> 16;
    crv = (crv * contrast * saturation) >> 32;
    cbu = (cbu * contrast * saturation) >> 32;
    cgu = (cgu * contrast * saturation) >> 32;
    cgv = (cgv * contrast * saturation) >> 32;
    oy -= 256 * brightness;

    t->y_coeff  = round_to_int16(cy  * (1 << 13));
    t->y_offset = round_to_int16(oy  * (1 <<  9));
    t->v2r      = round_to_int16(crv * (1 << 13));
    t->v2g      = round_to_int16(cgv * (1 << 13));
    t->u2g      = round_to_int16(cgu * (1 << 13));
    t->u2b      = round_to_int16(cbu * (1 << 13));

    {
        int64_t d = cy > 1 ? cy : 1;
        crv = ((crv * (1 << 16)) + 0x8000) / d;
        cbu = ((cbu * (1 << 16)) + 0x8000) / d;
        cgu = ((cgu * (1 << 16)) + 0x8000) / d;
        cgv = ((cgv * (1 << 16)) + 0x8000) / d;
    }
    t->cy = cy; t->oy = oy;
    t->crv = crv; t->cbu = cbu; t->cgu = cgu; t->cgv = cgv;

    yb = -(384 << 16) - ORC_TABLE_LUMA_HEADROOM * cy - oy;
    t->yb0 = yb;
    for (i = 0; i < table_plane_size; i++) {
        t->y_table[i] = (uint8_t)clip_u8((int)((yb + 0x8000) >> 16));
        yb += cy;
    }
    /* fill_table(table, 1, inc, y_table + yoffs): pointer = base - (inc>>9) + (clip(i-HR)*inc >> 16) */
    for (i = 0; i < 256 + 2 * ORC_TABLE_HEADROOM; i++) {
        int64_t k = clip_u8(i - ORC_TABLE_HEADROOM);
        t->off_rV[i] = (int32_t)(t->yoffs - (crv >> 9) + ((k * crv) >> 16));
        t->off_gU[i] = (int32_t)(t->yoffs - (cgu >> 9) + ((k * cgu) >> 16));
        t->off_bU[i] = (int32_t)(t->yoffs - (cbu >> 9) + ((k * cbu) >> 16));
        t->off_gV[i] = (int32_t)(-(cgv >> 9) + ((k * cgv) >> 16));       /* fill_gv_table */
    }
    return 0;
}

void orc_yuv2rgb_lut_px(const OrcYuv2Rgb *t, int Y, int U, int V, uint8_t rgb[3])
{
    const uint8_t *r = t->y_table + t->off_rV[V + ORC_TABLE_HEADROOM];
    const uint8_t *g = t->y_table + t->off_gU[U + ORC_TABLE_HEADROOM] + t->off_gV[V + ORC_TABLE_HEADROOM];
    const uint8_t *b = t->y_table + t->off_bU[U + ORC_TABLE_HEADROOM];
    rgb[0] = r[Y]; rgb[1] = g[Y]; rgb[2] = b[Y];
}

static int closed_T(const OrcYuv2Rgb *t, int64_t idx)
{
    return clip_u8((int)((t->yb0 + idx * t->cy + 0x8000) >> 16));
}

void orc_yuv2rgb_closed_px(const OrcYuv2Rgb *t, int Y, int U, int V, uint8_t rgb[3])
{
    int64_t ir = t->yoffs - (t->crv >> 9) + (((int64_t)V * t->crv) >> 16) + Y;
    int64_t ig = t->yoffs - (t->cgu >> 9) + (((int64_t)U * t->cgu) >> 16)
                          - (t->cgv >> 9) + (((int64_t)V * t->cgv) >> 16) + Y;
    int64_t ib = t->yoffs - (t->cbu >> 9) + (((int64_t)U * t->cbu) >> 16) + Y;
    rgb[0] = (uint8_t)closed_T(t, ir);
    rgb[1] = (uint8_t)closed_T(t, ig);
    rgb[2] = (uint8_t)closed_T(t, ib);
}

long orc_yuv2rgb_selfcheck(const OrcYuv2Rgb *t)
{
    long bad = 0;
    int y, u, v;
    const int n = 1024 + 2 * ORC_TABLE_LUMA_HEADROOM;
    for (v = 0; v < 256; v++)
        for (u = 0; u < 256; u++) {
            /* the LUT indices must stay inside the table for every luma value */
            int lo = t->off_rV[v + 512], hi = lo + 255;
            int g0 = t->off_gU[u + 512] + t->off_gV[v + 512];
            int b0 = t->off_bU[u + 512];
            if (lo < 0 || hi >= n || g0 < 0 || g0 + 255 >= n || b0 < 0 || b0 + 255 >= n) {
                bad += 256;
                continue;
            }
            for (y = 0; y < 256; y++) {
                uint8_t a[3], b[3];
                orc_yuv2rgb_lut_px(t, y, u, v, a);
                orc_yuv2rgb_closed_px(t, y, u, v, b);
                bad += (a[0] != b[0]) | (a[1] != b[1]) | (a[2] != b[2]);
            }
        }
    return bad;
}

static void put_px(uint8_t *d, const uint8_t rgb[3], int dst_fmt)
{
    switch (dst_fmt) {
    case ORC_PIX_RGB24: d[0] = rgb[0]; d[1] = rgb[1]; d[2] = rgb[2]; break;
    case ORC_PIX_BGR24: d[0] = rgb[2]; d[1] = rgb[1]; d[2] = rgb[0]; break;
    case ORC_PIX_RGBA:  d[0] = rgb[0]; d[1] = rgb[1]; d[2] = rgb[2]; d[3] = 255; break;
    case ORC_PIX_BGRA:  d[0] = rgb[2]; d[1] = rgb[1]; d[2] = rgb[0]; d[3] = 255; break;
    }
}

int orc_yuv2rgb_frame(const OrcYuv2Rgb *t, const uint8_t *const src[4], const int src_stride[4],
                      uint8_t *dst, int dst_stride, int w, int h, int src_fmt, int dst_fmt)
{
    int step, x, y;
    switch (dst_fmt) {
    case ORC_PIX_RGB24: case ORC_PIX_BGR24: step = 3; break;
    case ORC_PIX_RGBA:  case ORC_PIX_BGRA:  step = 4; break;
    default: return -1;
    }
    if (src_fmt != ORC_PIX_NV12 && src_fmt != ORC_PIX_YUV420P)
        return -1;
    for (y = 0; y < h; y++) {
        const uint8_t *py = src[0] + (long)y * src_stride[0];
        uint8_t *d = dst + (long)y * dst_stride;
        for (x = 0; x < w; x++) {
            int U, V;
            uint8_t rgb[3];
            if (src_fmt == ORC_PIX_NV12) {
                const uint8_t *puv = src[1] + (long)(y >> 1) * src_stride[1] + 2 * (x >> 1);
                U = puv[0]; V = puv[1];
            } else {
                U = src[1][(long)(y >> 1) * src_stride[1] + (x >> 1)];
                V = src[2][(long)(y >> 1) * src_stride[2] + (x >> 1)];
            }
            orc_yuv2rgb_lut_px(t, py[x], U, V, rgb);
            put_px(d + x * step, rgb, dst_fmt);
        }
    }
    return 0;
}

int orc_nv12_to_rgbpf32(const OrcYuv2Rgb *t, const uint8_t *const src[4], const int src_stride[4],
                        float *dst, int dst_stride_bytes, int w, int h)
{
    int x, y, k;
    for (y = 0; y < h; y++) {
        const uint8_t *py = src[0] + (long)y * src_stride[0];
        for (x = 0; x < w; x++) {
            const uint8_t *puv = src[1] + (long)(y >> 1) * src_stride[1] + 2 * (x >> 1);
            uint8_t rgb[3];
            orc_yuv2rgb_lut_px(t, py[x], puv[0], puv[1], rgb);
            for (k = 0; k < 3; k++) {
                float *plane = (float *)((uint8_t *)dst + (long)k * dst_stride_bytes * h);
                float *row = (float *)((uint8_t *)plane + (long)y * dst_stride_bytes);
                row[x] = (float)rgb[k] / 255.0f;
            }
        }
    }
    return 0;
}

void orc_fill_lcg(uint8_t *p, long n, uint32_t seed)
{
    uint32_t s = seed;
    long i;
    for (i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        p[i] = (uint8_t)(s >> 24);
    }
}

void orc_free(void *p) { free(p); }
