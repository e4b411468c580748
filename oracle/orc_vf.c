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
