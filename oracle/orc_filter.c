/*
 * oracle/orc_filter.c — TEST INFRASTRUCTURE ONLY (see orc.h).
 *
 * Restates initFilter(), libswscale/utils.c:367-763 of the reference tree, for the case
 * srcFilter == dstFilter == NULL (the only one libgpuscale and vf_scale reach) and
 * cpu_flags == 0 (the portable C build: no MMX/AltiVec special cases, utils.c:622-642).
 *
 * Stages, in the reference's order:
 *   1. raw 64-bit coefficients per output sample            utils.c:392-552
 *   2. (src/dst filter convolution: identity here)          utils.c:554-581
 *   3. trim near-zero taps, find the minimal common size    utils.c:583-620
 *   4. align, zero padding under SWS_BITEXACT               utils.c:643-670
 *   5. fold taps that fall outside [0,srcW) ("fix borders") utils.c:673-714
 *   6. error-diffused normalisation to `one`                utils.c:721-741
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "orc.h"

#define ORC_MAX_REDUCE_CUTOFF 0.002   /* swscale.h:97 */

static int64_t i64abs(int64_t v) { return v < 0 ? -v : v; }

static int ilog2(unsigned v)
{
    int n = 0;
    while (v >>= 1) n++;
    return n;
}

/* ROUNDED_DIV from libavutil/common.h:60 */
static int64_t rounded_div(int64_t a, int64_t b)
{
    return a >= 0 ? (a + (b >> 1)) / b : (a - (b >> 1)) / b;
}

/* getSplineCoeff, utils.c:322-334 */
static double spline_coeff(double a, double b, double c, double d, double dist)
{
    if (dist <= 1.0)
        return ((d * dist + c) * dist + b) * dist + a;
    return spline_coeff(0.0, b + 2.0 * c + 3.0 * d, c + 3.0 * d, -b - 3.0 * c - 6.0 * d, dist - 1.0);
}

int orc_init_filter(int16_t **out_filter, int32_t **filter_pos, int *out_filter_size,
                    int x_inc, int src_w, int dst_w, int filter_align, int one,
                    int flags, const double param_in[2], int src_pos, int dst_pos)
{
    int i, j;
    int filter_size, filter2_size, min_filter_size;
    int64_t *filter = NULL, *filter2 = NULL;
    int32_t *pos = NULL;
    int16_t *outf = NULL;
    double param[2] = { ORC_SWS_PARAM_DEFAULT, ORC_SWS_PARAM_DEFAULT };
    int lg = ilog2((unsigned)(src_w / dst_w > 0 ? src_w / dst_w : 1));
    const int64_t fone = 1LL << (54 - (lg < 8 ? lg : 8));
    int ret = -1;

    /* av_log2(0) == 0 in libavutil (intmath.h: v|1) */
    if (param_in) { param[0] = param_in[0]; param[1] = param_in[1]; }

    pos = (int32_t *)malloc(sizeof(int32_t) * (dst_w + 3));
    if (!pos) goto fail;

    if (abs(x_inc - 0x10000) < 10 && src_pos == dst_pos) {          /* unscaled, :394-404 */
        filter_size = 1;
        filter = (int64_t *)calloc((size_t)dst_w * filter_size, sizeof(int64_t));
        if (!filter) goto fail;
        for (i = 0; i < dst_w; i++) {
            filter[i] = fone;
            pos[i]    = i;
        }
    } else if (flags & ORC_SWS_POINT) {                              /* :405-420 */
        int64_t x_dst_in_src;
        filter_size = 1;
        filter = (int64_t *)malloc(sizeof(int64_t) * dst_w);
        if (!filter) goto fail;
        x_dst_in_src = ((dst_pos * (int64_t)x_inc) >> 8) - ((src_pos * 0x8000LL) >> 7);
        for (i = 0; i < dst_w; i++) {
            int xx = (int)((x_dst_in_src - ((int64_t)(filter_size - 1) << 15) + (1 << 15)) >> 16);
            pos[i]    = xx;
            filter[i] = fone;
            x_dst_in_src += x_inc;
        }
    } else if ((x_inc <= (1 << 16) && (flags & ORC_SWS_AREA)) ||
               (flags & ORC_SWS_FAST_BILINEAR)) {                    /* :421-446 */
        int64_t x_dst_in_src;
        filter_size = 2;
        filter = (int64_t *)malloc(sizeof(int64_t) * dst_w * filter_size);
        if (!filter) goto fail;
        x_dst_in_src = ((dst_pos * (int64_t)x_inc) >> 8) - ((src_pos * 0x8000LL) >> 7);
        for (i = 0; i < dst_w; i++) {
            int xx = (int)((x_dst_in_src - ((int64_t)(filter_size - 1) << 15) + (1 << 15)) >> 16);
            pos[i] = xx;
            for (j = 0; j < filter_size; j++) {
                int64_t coeff = fone - i64abs((int64_t)xx * (1 << 16) - x_dst_in_src) * (fone >> 16);
                if (coeff < 0) coeff = 0;
                filter[i * filter_size + j] = coeff;
                xx++;
            }
            x_dst_in_src += x_inc;
        }
    } else {                                                         /* :447-552 */
        int64_t x_dst_in_src;
        int size_factor = -1;
        /* scale_algorithms[], utils.c:353-365, first match with a positive factor */
        if      (flags & ORC_SWS_AREA)     size_factor = 1;
        else if (flags & ORC_SWS_BICUBIC)  size_factor = 4;
        else if (flags & ORC_SWS_BILINEAR) size_factor = 2;
        else if (flags & 0x80)             size_factor = 8;    /* GAUSS  */
        else if (flags & 0x100)            size_factor = 20;   /* SINC   */
        else if (flags & 0x400)            size_factor = 20;   /* SPLINE */
        else if (flags & 8)                size_factor = 8;    /* X      */
        if (flags & ORC_SWS_LANCZOS)
            size_factor = param[0] != ORC_SWS_PARAM_DEFAULT ? (int)ceil(2 * param[0]) : 6;
        if (size_factor <= 0) goto fail;

        if (x_inc <= 1 << 16)
            filter_size = 1 + size_factor;
        else
            filter_size = 1 + (int)(((int64_t)size_factor * src_w + dst_w - 1) / dst_w);
        if (filter_size > src_w - 2) filter_size = src_w - 2;
        if (filter_size < 1)         filter_size = 1;

        filter = (int64_t *)malloc(sizeof(int64_t) * (size_t)dst_w * filter_size);
        if (!filter) goto fail;
        x_dst_in_src = ((dst_pos * (int64_t)x_inc) >> 7) - ((src_pos * 0x10000LL) >> 7);
        for (i = 0; i < dst_w; i++) {
            int xx = (int)((x_dst_in_src - (filter_size - 2) * (1LL << 16)) / (1 << 17));
            pos[i] = xx;
            for (j = 0; j < filter_size; j++) {
                int64_t d = i64abs(((int64_t)xx * (1 << 17)) - x_dst_in_src) << 13;
                double floatd;
                int64_t coeff;

                if (x_inc > 1 << 16)
                    d = d * dst_w / src_w;
                floatd = d * (1.0 / (1 << 30));

                if (flags & ORC_SWS_BICUBIC) {
                    int64_t B = (int64_t)((param[0] != ORC_SWS_PARAM_DEFAULT ? param[0] :   0) * (1 << 24));
                    int64_t C = (int64_t)((param[1] != ORC_SWS_PARAM_DEFAULT ? param[1] : 0.6) * (1 << 24));
                    if (d >= 1LL << 31) {
                        coeff = 0;
                    } else {
                        int64_t dd  = (d  * d) >> 30;
                        int64_t ddd = (dd * d) >> 30;
                        if (d < 1LL << 30)
                            coeff =  (12 * (1 << 24) -  9 * B - 6 * C) * ddd +
                                    (-18 * (1 << 24) + 12 * B + 6 * C) *  dd +
                                      (6 * (1 << 24) -  2 * B)         * (1 << 30);
                        else
                            coeff =      (-B -  6 * C) * ddd +
                                      (6 * B + 30 * C) * dd  +
                                    (-12 * B - 48 * C) * d   +
                                      (8 * B + 24 * C) * (1 << 30);
                    }
                    coeff /= (1LL << 54) / fone;
                } else if (flags & 8) {                          /* SWS_X, :497-508 */
                    double A = param[0] != ORC_SWS_PARAM_DEFAULT ? param[0] : 1.0;
                    double cc;
                    if (floatd < 1.0) cc = cos(floatd * M_PI);
                    else              cc = -1.0;
                    if (cc < 0.0) cc = -pow(-cc, A);
                    else          cc = pow(cc, A);
                    coeff = (int64_t)((cc * 0.5 + 0.5) * fone);
                } else if (flags & ORC_SWS_AREA) {
                    int64_t d2 = d - (1 << 29);
                    if (d2 * x_inc < -(1LL << (29 + 16)))
                        coeff = 1LL << (30 + 16);
                    else if (d2 * x_inc < (1LL << (29 + 16)))
                        coeff = -d2 * x_inc + (1LL << (29 + 16));
                    else
                        coeff = 0;
                    coeff *= fone >> (30 + 16);
                } else if (flags & 0x80) {                       /* GAUSS */
                    double p = param[0] != ORC_SWS_PARAM_DEFAULT ? param[0] : 3.0;
                    coeff = (int64_t)(exp2(-p * floatd * floatd) * fone);
                } else if (flags & 0x100) {                      /* SINC */
                    coeff = (int64_t)((d ? sin(floatd * M_PI) / (floatd * M_PI) : 1.0) * fone);
                } else if (flags & ORC_SWS_LANCZOS) {
                    double p = param[0] != ORC_SWS_PARAM_DEFAULT ? param[0] : 3.0;
                    coeff = (int64_t)((d ? sin(floatd * M_PI) * sin(floatd * M_PI / p) /
                                       (floatd * floatd * M_PI * M_PI / p) : 1.0) * fone);
                    if (floatd > p)
                        coeff = 0;
                } else if (flags & ORC_SWS_BILINEAR) {
                    coeff = (1 << 30) - d;
                    if (coeff < 0) coeff = 0;
                    coeff *= fone >> 30;
                } else if (flags & 0x400) {                      /* SWS_SPLINE, :535-537 */
                    double p = -2.196152422706632;
                    coeff = (int64_t)(spline_coeff(1.0, 0.0, p, -p - 1.0, floatd) * fone);
                } else {
                    goto fail;
                }
                filter[i * filter_size + j] = coeff;
                xx++;
            }
            x_dst_in_src += 2 * x_inc;
        }
    }

    /* stage 2: no src/dst vectors -> filter2 is a copy (:554-581) */
    filter2_size = filter_size;
    filter2 = (int64_t *)calloc((size_t)dst_w * filter2_size, sizeof(int64_t));
    if (!filter2) goto fail;
    memcpy(filter2, filter, sizeof(int64_t) * (size_t)dst_w * filter_size);
    free(filter); filter = NULL;

    /* stage 3 (:583-620) */
    min_filter_size = 0;
    for (i = dst_w - 1; i >= 0; i--) {
        int min = filter2_size;
        int64_t cut_off = 0;

        for (j = 0; j < filter2_size; j++) {
            int k;
            cut_off += i64abs(filter2[i * filter2_size]);
            if (cut_off > ORC_MAX_REDUCE_CUTOFF * fone)
                break;
            if (i < dst_w - 1 && pos[i] >= pos[i + 1])
                break;
            for (k = 1; k < filter2_size; k++)
                filter2[i * filter2_size + k - 1] = filter2[i * filter2_size + k];
            filter2[i * filter2_size + k - 1] = 0;
            pos[i]++;
        }

        cut_off = 0;
        for (j = filter2_size - 1; j > 0; j--) {
            cut_off += i64abs(filter2[i * filter2_size + j]);
            if (cut_off > ORC_MAX_REDUCE_CUTOFF * fone)
                break;
            min--;
        }
        if (min > min_filter_size)
            min_filter_size = min;
    }
    if (min_filter_size <= 0) goto fail;

    /* stage 4 (:643-670) */
    filter_size = (min_filter_size + (filter_align - 1)) & (~(filter_align - 1));
    filter = (int64_t *)malloc(sizeof(int64_t) * (size_t)dst_w * filter_size);
    if (!filter) goto fail;
    /* RETCODE_USE_CASCADE (:648-652): MAX_FILTER_SIZE*16/16 = 256 taps */
    if (filter_size >= 256) goto fail;
    for (i = 0; i < dst_w; i++)
        for (j = 0; j < filter_size; j++) {
            if (j >= filter2_size)
                filter[i * filter_size + j] = 0;
            else
                filter[i * filter_size + j] = filter2[i * filter2_size + j];
            if ((flags & ORC_SWS_BITEXACT) && j >= min_filter_size)
                filter[i * filter_size + j] = 0;
        }

    /* stage 5 (:673-714) */
    for (i = 0; i < dst_w; i++) {
        if (pos[i] < 0) {
            for (j = 1; j < filter_size; j++) {
                int left = j + pos[i] > 0 ? j + pos[i] : 0;
                filter[i * filter_size + left] += filter[i * filter_size + j];
                filter[i * filter_size + j]     = 0;
            }
            pos[i] = 0;
        }
        if (pos[i] + filter_size > src_w) {
            int shift = pos[i] + (filter_size - src_w < 0 ? filter_size - src_w : 0);
            int64_t acc = 0;
            for (j = filter_size - 1; j >= 0; j--) {
                if (pos[i] + j >= src_w) {
                    acc += filter[i * filter_size + j];
                    filter[i * filter_size + j] = 0;
                }
            }
            for (j = filter_size - 1; j >= 0; j--) {
                if (j < shift)
                    filter[i * filter_size + j] = 0;
                else
                    filter[i * filter_size + j] = filter[i * filter_size + j - shift];
            }
            pos[i] -= shift;
            filter[i * filter_size + src_w - 1 - pos[i]] += acc;
        }
    }

    /* stage 6 (:716-741) */
    outf = (int16_t *)calloc((size_t)filter_size * (dst_w + 3), sizeof(int16_t));
    if (!outf) goto fail;
    for (i = 0; i < dst_w; i++) {
        int64_t error = 0, sum = 0;
        for (j = 0; j < filter_size; j++)
            sum += filter[i * filter_size + j];
        sum = (sum + one / 2) / one;
        if (!sum) sum = 1;
        for (j = 0; j < filter_size; j++) {
            int64_t v = filter[i * filter_size + j] + error;
            int int_v = (int)rounded_div(v, sum);
            outf[i * filter_size + j] = (int16_t)int_v;
            error = v - int_v * sum;
        }
    }
    pos[dst_w + 0] = pos[dst_w + 1] = pos[dst_w + 2] = pos[dst_w - 1];
    for (i = 0; i < filter_size; i++) {
        int k = (dst_w - 1) * filter_size + i;
        outf[k + 1 * filter_size] = outf[k + 2 * filter_size] = outf[k + 3 * filter_size] = outf[k];
    }

    *out_filter = outf; outf = NULL;
    *filter_pos = pos;  pos = NULL;
    *out_filter_size = filter_size;
    ret = 0;
fail:
    free(filter);
    free(filter2);
    free(outf);
    free(pos);
    return ret;
}
