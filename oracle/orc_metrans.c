/*
 * oracle/orc_metrans.c — CPU restatement of the MeTrans kernels whose arithmetic IS in the reference tree
 * (paths relative to /root/reference/metrans/include/NvCodec).  TEST INFRASTRUCTURE ONLY (see orc.h).
 *
 *   orc_mt_scale_nv12_bicubic   ScaleNv12_Bicubic_Kernel + BicubicLuma / BicubicChroma / BicubicCoefficient,
 *                               Resize_bicubic.cu:83-159
 *   orc_mt_u8_to_u16 / orc_mt_u16_to_u8   ConvertUInt8ToUInt16Kernel / ConvertUInt16ToUInt8Kernel, BitDepth.cu:15-29
 *
 * PARITY PIN STATUS: **parity unpinned**.  The reference holds no test, golden frame or checksum for these kernels and CUDA
 * cannot run here, so nothing checks this restatement against an output of the reference.  It follows the source operation
 * by operation in IEEE float32 with every product and sum rounded separately (-ffp-contract=off / ISO C: no fused multiply-add).
 * nvcc contracts a * b + c into fma by default, so the reference's own result can differ from this one where a sum lands within
 * an ulp of an integer: the test tolerance is +-1 LSB (BASELINE.json's bound for the float bicubic path); the HIP kernel is built
 * without contraction too and in practice agrees bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include "orc.h"

/* Resize_bicubic.cu:83-87 */
static float bicubic_coefficient(float d)
{
    d = fabsf(d);
    const float a = -0.5f;
    return d > 2.0f ? 0 : (d > 1.0f ? a * d * d * d - 5.0f * a * d * d + 8.0f * a * d - 4.0f * a
                                    : (a + 2.0f) * d * d * d - (a + 3.0f) * d * d + 1.0f);
}

static float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

/* one channel of BicubicLuma (:89-108) / BicubicChroma (:110-133): `step` bytes between samples, `n_cols` x `n_rows` the plane
 * the caller clamped the coordinates to.  A tap whose weight is exactly 0 is not read (the reference reads column / row n there,
 * one past the plane, and multiplies it by 0). */
static uint8_t bicubic_sample(const uint8_t *src, int pitch, int step, float fx, float fy)
{
    const int sx0 = (int)fx - 1, sy0 = (int)fy - 1;
    float cx[4], cy[4];
    for (int i = 0; i < 4; i++) {
        cx[i] = bicubic_coefficient((float)(sx0 + i) - fx);
        cy[i] = bicubic_coefficient((float)(sy0 + i) - fy);
    }
    float r = 0;
    for (int y = 0; y < 4; y++) {
        float rx = 0;
        for (int x = 0; x < 4; x++) {
            const float s = (cx[x] == 0.0f || cy[y] == 0.0f) ? 0.0f : (float)src[(long)pitch * (sy0 + y) + (long)(sx0 + x) * step];
            rx += s * cx[x];
        }
        r += rx * cy[y];
    }
    return (uint8_t)fmaxf(fminf(r, 255.0f), 0.0f);
}

/* ScaleNv12_Bicubic_Kernel (:135-159): one thread per 2 x 2 luma block and its chroma pair; chroma plane at base + pitch * height */
int orc_mt_scale_nv12_bicubic(const uint8_t *src, int src_pitch, int src_w, int src_h, uint8_t *dst, int dst_pitch, int dst_w, int dst_h)
{
    if (src_w < 8 || src_h < 8) return -1;
    const float fx_scale = (float)src_w / dst_w, fy_scale = (float)src_h / dst_h;
    const uint8_t *csrc = src + (long)src_h * src_pitch;
    for (int iy = 0; iy < dst_h / 2; iy++)
        for (int ix = 0; ix < dst_w / 2; ix++) {
            const int x = ix * 2;
            for (int dy = 0; dy < 2; dy++) {
                const int y = iy * 2 + dy;
                for (int dx = 0; dx < 2; dx++)
                    dst[(long)y * dst_pitch + x + dx] =
                        bicubic_sample(src, src_pitch, 1, clampf((x + dx) * fx_scale, 2.0f, (float)(src_w - 2)),
                                       clampf(y * fy_scale, 2.0f, (float)(src_h - 2)));
            }
            const float cfx = clampf(ix * fx_scale, 2.0f, (float)(src_w / 2 - 2)), cfy = clampf(iy * fy_scale, 2.0f, (float)(src_h / 2 - 2));
            uint8_t *d = dst + (long)(dst_h + iy) * dst_pitch + ix * 2;
            /* uchar2 plane of pitch nSrcPitch / 2 pairs: the same bytes, two bytes a sample */
            d[0] = bicubic_sample(csrc, src_pitch / 2 * 2, 2, cfx, cfy);
            d[1] = bicubic_sample(csrc + 1, src_pitch / 2 * 2, 2, cfx, cfy);
        }
    return 0;
}

/* BitDepth.cu:15-21: *(uchar2 *)&dpUInt16[x] = uchar2{0, dpUInt8[x]} */
void orc_mt_u8_to_u16(const uint8_t *src, uint16_t *dst, long n)
{
    for (long i = 0; i < n; i++) { uint8_t *b = (uint8_t *)&dst[i]; b[0] = 0; b[1] = src[i]; }
}

/* BitDepth.cu:23-29: dpUInt8[x] = ((uchar2 *)&dpUInt16[x])->y */
void orc_mt_u16_to_u8(const uint16_t *src, uint8_t *dst, long n)
{
    for (long i = 0; i < n; i++) dst[i] = ((const uint8_t *)&src[i])[1];
}
