/*
 * oracle/orc_vsynth.c — TEST INFRASTRUCTURE ONLY (see orc.h).
 *
 * Restatement of the reference's synthetic test-video generator, so that the reference's own
 * FATE golden values (under tests/ref/fate and tests/ref/pixfmt) that are computed over "vsynth1"
 * can be reproduced here and used to pin the oracle:
 *   tests/videogen.c:29-42    myrnd      — seed = seed*314159 + 1; n==256 ? seed>>24 : seed%n
 *   tests/videogen.c:52-67    int_cos    — 1 - x^2 cosine approximation, 8 fractional bits
 *   tests/videogen.c:80-146   gen_image  — gradient background, 26x26 noise block, 10 moving
 *                                          noisy rectangles whose walk continues frame to frame
 *   tests/utils.c:36-103      rgb24_to_yuv420p — JPEG-matrix 8-bit fixed point, 2x2 chroma mean
 *   tests/utils.c:160-173     put_pixel  — clipped to the picture, channels stored mod 256
 * Output layout = tests/data/vsynth1.yuv (utils.c:139-148): per frame Y, then all U rows, then
 * all V rows, tightly packed.  tests/vsynth1/NN.pgm holds the same planes (U|V rows side by side).
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

static unsigned vs_rnd(unsigned *seed_ptr, int n)
{
    unsigned seed = *seed_ptr * 314159u + 1u, val;
    val = n == 256 ? seed >> 24 : seed % (unsigned)n;
    *seed_ptr = seed;
    return val;
}

static int vs_cos(int a)
{
    int v, neg = 0;
    a &= 255;
    if (a >= 128) a = 256 - a;
    if (a > 64) { neg = -1; a = 128 - a; }
    v = 256 - ((a * a) >> 4);
    return (v ^ neg) - neg;
}

typedef struct { int x, y, w, h, r, g, b; } VsObj;

static void vs_put(uint8_t *rgb, int w, int h, int x, int y, int r, int g, int b)
{
    uint8_t *p;
    if (x < 0 || x >= w || y < 0 || y >= h) return;
    p = rgb + ((long)y * w + x) * 3;
    p[0] = (uint8_t)r; p[1] = (uint8_t)g; p[2] = (uint8_t)b;
}

static void vs_to_yuv420p(uint8_t *lum, uint8_t *cb, uint8_t *cr, const uint8_t *rgb, int w, int h)
{
    /* FIX(x) = (int)(x * 256 + 0.5): 0.299->77 0.587->150 0.114->29 0.16874->43 0.33126->85 0.5->128
     * 0.41869->107 0.08131->21 */
    int x, y, k;
    for (y = 0; y < h; y += 2) {
        for (x = 0; x < w; x += 2) {
            int r1 = 0, g1 = 0, b1 = 0;
            for (k = 0; k < 4; k++) {
                const uint8_t *p = rgb + ((long)(y + (k >> 1)) * w + x + (k & 1)) * 3;
                int r = p[0], g = p[1], b = p[2];
                r1 += r; g1 += g; b1 += b;
                lum[(long)(y + (k >> 1)) * w + x + (k & 1)] = (uint8_t)((77 * r + 150 * g + 29 * b + 128) >> 8);
            }
            cb[(long)(y >> 1) * (w >> 1) + (x >> 1)] = (uint8_t)(((-43 * r1 - 85 * g1 + 128 * b1 + 4 * 128 - 1) >> 10) + 128);
            cr[(long)(y >> 1) * (w >> 1) + (x >> 1)] = (uint8_t)(((128 * r1 - 107 * g1 - 21 * b1 + 4 * 128 - 1) >> 10) + 128);
        }
    }
}

/* Writes frames [0, nframes) of vsynth1 as consecutive yuv420p frames (w*h*3/2 bytes each). */
int orc_vsynth1(uint8_t *out, int w, int h, int nframes)
{
    VsObj objs[10];
    unsigned seed = 1, seed1;
    uint8_t *rgb;
    int num, i, x, y;
    if (w < 2 || h < 2 || (w & 1) || (h & 1)) return -1;
    rgb = (uint8_t *)malloc((size_t)w * h * 3);
    if (!rgb) return -1;
    for (num = 0; num < nframes; num++) {
        int dx, dy;
        if (num == 0) {
            for (i = 0; i < 10; i++) {
                objs[i].x = (int)vs_rnd(&seed, w);
                objs[i].y = (int)vs_rnd(&seed, h);
                objs[i].w = (int)vs_rnd(&seed, w / 4) + 10;
                objs[i].h = (int)vs_rnd(&seed, h / 4) + 10;
                objs[i].r = (int)vs_rnd(&seed, 256);
                objs[i].g = (int)vs_rnd(&seed, 256);
                objs[i].b = (int)vs_rnd(&seed, 256);
            }
        }
        dx = vs_cos(num * 256 / 50) * 35;
        dy = vs_cos(num * 256 / 50 + 256 / 10) * 30;
        for (y = 0; y < h; y++)
            for (x = 0; x < w; x++) {
                int x1 = (x << 8) + dx, y1 = (y << 8) + dy;
                vs_put(rgb, w, h, x, y, ((y1 * 7) >> 8) & 0xff, (((x1 + y1) * 9) >> 8) & 0xff, ((x1 * 5) >> 8) & 0xff);
            }
        seed1 = (unsigned)num;
        for (y = 0; y < 26; y++)
            for (x = 0; x < 26; x++) {
                int r = (int)vs_rnd(&seed1, 256), g = (int)vs_rnd(&seed1, 256), b = (int)vs_rnd(&seed1, 256);
                vs_put(rgb, w, h, x + 10, y + 30, r, g, b);
            }
        for (i = 0; i < 10; i++) {
            VsObj *p = &objs[i];
            seed1 = (unsigned)i;
            for (y = 0; y < p->h; y++)
                for (x = 0; x < p->w; x++) {
                    int r = p->r + (int)vs_rnd(&seed1, 50);
                    int g = p->g + (int)vs_rnd(&seed1, 50);
                    int b = p->b + (int)vs_rnd(&seed1, 50);
                    vs_put(rgb, w, h, x + p->x, y + p->y, r, g, b);
                }
            p->x += (int)vs_rnd(&seed, 21) - 10;
            p->y += (int)vs_rnd(&seed, 21) - 10;
        }
        {
            uint8_t *f = out + (size_t)num * ((size_t)w * h * 3 / 2);
            vs_to_yuv420p(f, f + (size_t)w * h, f + (size_t)w * h + (size_t)(w / 2) * (h / 2), rgb, w, h);
        }
    }
    free(rgb);
    return nframes;
}
