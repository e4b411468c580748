// k_rgb64.hip — RGBA64LE / BGRA64LE as SOURCES, and the alpha plane of contexts whose both ends carry alpha (gfx950).
//
// libswscale reads a 64-bit packed RGB source through rgb64ToY_c / rgb64ToUV_c / rgb64ToUV_half_c (input.c:36-121) into 16-bit
// lines, then runs the generic scaler on them exactly as on a planar 16-bit source (hScale16To15_c / hScale16To19_c with
// sh from the 16-bit depth, swscale.c:63-119).  So the source side is ONE conversion kernel into Y / U / V planes of 16-bit
// samples in HBM (U, V at the chroma width the context decided: halved when chrSrcHSubSample is set, utils.c:1529-1545; never
// halved vertically), and the context behind it is the planar-16 one that exists already.
//
// Alpha (needAlpha = both ends have an alpha channel, utils.c:1902): the source's alpha samples (rgbaToA_c: a << 6 | a >> 2 for
// the 8-bit formats, rgba64leToA_c: the sample for the 64-bit ones, input.c:413-449) go through the LUMA filters — horizontally
// in lum_h_scale's alpha leg (hscale.c:66-80), vertically inside the packed writer, each form of which has its own rounding
// (output.c, quoted at alpha8_out_kernel).  Here: the existing one-sample-per-thread horizontal pass of k_scale16.hip into
// int32 lines, then alpha8_out_kernel / the alpha operand of vrgba64_kernel.  Completeness paths, not fast ones.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

namespace gmat {

// One thread per chroma sample: its one (full chroma) or two (half) pixels' luma, and U / V.
//   Y  = (ry*r + gy*g + by*b + (0x2001 << 14)) >> 15                       rgb64ToY_c_template        input.c:36-50
//   UV = (ru*r + gu*g + bu*b + (0x10001 << 14)) >> 15                      rgb64ToUV_c_template       :52-69
//        the same on (p0 + p1 + 1) >> 1 per channel                        rgb64ToUV_half_c_template  :71-87
// An odd width with halved chroma reads pixel 2i+1 of the last pair past the row in the reference (not bit-defined); the last
// pixel is used twice here, as the 8-bit readers of this library do.
__global__ __launch_bounds__(256) void rgb64_planes_kernel(const uint8_t *src, int ss, int w, int h, int chrW, int half, int bgr,
                                                           Rgb2YuvConsts k, uint8_t *py, int ys, uint8_t *pu, int us, uint8_t *pv, int vs)
{
    const int cx = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (cx >= chrW || y >= h) return;
    const unsigned short *row = reinterpret_cast<const unsigned short *>(src + (size_t)y * ss);
    unsigned short *oy = reinterpret_cast<unsigned short *>(py + (size_t)y * ys);
    const int ro = bgr ? 2 : 0, bo = 2 - ro;
    auto luma = [&](int x) {
        const unsigned r = row[4 * x + ro], g = row[4 * x + 1], b = row[4 * x + bo];
        oy[x] = (unsigned short)(((unsigned)k.ry * r + (unsigned)k.gy * g + (unsigned)k.by * b + (0x2001u << 14)) >> 15);
    };
    int r, g, b;
    if (half) {
        const int x0 = 2 * cx, x1 = min(2 * cx + 1, w - 1);
        luma(x0);
        if (x1 != x0) luma(x1);
        r = (int)(row[4 * x0 + ro] + row[4 * x1 + ro] + 1) >> 1;
        g = (int)(row[4 * x0 + 1] + row[4 * x1 + 1] + 1) >> 1;
        b = (int)(row[4 * x0 + bo] + row[4 * x1 + bo] + 1) >> 1;
    } else {
        luma(cx);
        r = row[4 * cx + ro]; g = row[4 * cx + 1]; b = row[4 * cx + bo];
    }
    reinterpret_cast<unsigned short *>(pu + (size_t)y * us)[cx] = (unsigned short)((k.ru * r + k.gu * g + k.bu * b + (0x10001 << 14)) >> 15);
    reinterpret_cast<unsigned short *>(pv + (size_t)y * vs)[cx] = (unsigned short)((k.rv * r + k.gv * g + k.bv * b + (0x10001 << 14)) >> 15);
}

// The 8-bit packed RGB twin, for 16-bit destinations: rgb24ToY_c / ToUV_c / ToUV_half_c (input.c:795-866; the 32-bit readers are the same
// formulas on the same three channels, :246-390) into planes of 16-bit samples — the lines hScale16To19_c then shifts by 9.
//   Y  = (ry*r + gy*g + by*b + (32 << 14) + (1 << 8)) >> 9
//   UV = (ru*r + gu*g + bu*b + (256 << 14) + (1 << 8)) >> 9;   half: on the SUMS of a pixel pair, (.. + (256 << 15) + (1 << 9)) >> 10
__global__ __launch_bounds__(256) void rgb8_planes_kernel(const uint8_t *src, int ss, int w, int h, int chrW, int half, int bgr, int px,
                                                          Rgb2YuvConsts k, uint8_t *py, int ys, uint8_t *pu, int us, uint8_t *pv, int vs)
{
    const int cx = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (cx >= chrW || y >= h) return;
    const uint8_t *row = src + (size_t)y * ss;
    unsigned short *oy = reinterpret_cast<unsigned short *>(py + (size_t)y * ys);
    const int ro = bgr ? 2 : 0, bo = 2 - ro;
    auto luma = [&](int x) {
        const int r = row[px * x + ro], g = row[px * x + 1], b = row[px * x + bo];
        oy[x] = (unsigned short)((k.ry * r + k.gy * g + k.by * b + (32 << 14) + (1 << 8)) >> 9);
    };
    int u, v;
    if (half) {
        const int x0 = 2 * cx, x1 = min(2 * cx + 1, w - 1);
        luma(x0);
        if (x1 != x0) luma(x1);
        const int r = row[px * x0 + ro] + row[px * x1 + ro], g = row[px * x0 + 1] + row[px * x1 + 1], b = row[px * x0 + bo] + row[px * x1 + bo];
        u = (k.ru * r + k.gu * g + k.bu * b + (256 << 15) + (1 << 9)) >> 10;
        v = (k.rv * r + k.gv * g + k.bv * b + (256 << 15) + (1 << 9)) >> 10;
    } else {
        luma(cx);
        const int r = row[px * cx + ro], g = row[px * cx + 1], b = row[px * cx + bo];
        u = (k.ru * r + k.gu * g + k.bu * b + (256 << 14) + (1 << 8)) >> 9;
        v = (k.rv * r + k.gv * g + k.bv * b + (256 << 14) + (1 << 8)) >> 9;
    }
    reinterpret_cast<unsigned short *>(pu + (size_t)y * us)[cx] = (unsigned short)u;
    reinterpret_cast<unsigned short *>(pv + (size_t)y * vs)[cx] = (unsigned short)v;
}

// (round 6) both kernels above four pixels a thread: the format sweep (profiles/r06_sweep_before.txt) found every context behind them — packed RGB into
// 16-bit YUV / RGBA64, RGBA64 sources — at 25-38 us a 1080p frame with the scaler behind them at 6-10: a thread a chroma sample, byte and 16-bit loads, 16-bit stores.
// Here: 12 / 16 / 32 source bytes a thread in dwords where the row allows (al4: rows on 4-byte addresses; else bytes), eight luma bytes and four or eight chroma
// bytes stored at once (the planes are the context's own: rows on 256-byte addresses).  The ragged last group of a row takes the formulas pixel by pixel.
// BPS 1: rgb8_planes_kernel's arithmetic (PX 3 | 4 bytes a pixel); BPS 2: rgb64_planes_kernel's (8 bytes a pixel)
template <int BPS, int PX>
__global__ __launch_bounds__(256) void rgb_planes4_kernel(const uint8_t *src, int ss, int w, int h, int chrW, int half, int bgr, int al4,
                                                          Rgb2YuvConsts k, uint8_t *py, int ys, uint8_t *pu, int us, uint8_t *pv, int vs)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const uint8_t *row = src + (size_t)y * ss;
    unsigned short *oy = reinterpret_cast<unsigned short *>(py + (size_t)y * ys);
    unsigned short *ou = reinterpret_cast<unsigned short *>(pu + (size_t)y * us), *ov = reinterpret_cast<unsigned short *>(pv + (size_t)y * vs);
    const int ro = bgr ? 2 : 0, bo = 2 - ro;
    auto sample = [&](int xx, int ch) -> int {
        if (BPS == 2) return reinterpret_cast<const unsigned short *>(row)[4 * xx + ch];
        return row[PX * xx + ch];
    };
    // (v_mul_i32_i24: samples below 2^16 x coefficients below 2^15 — the low 32 bits of the product, full rate; v_mul_lo_u32 is a quarter of it)
    auto lum = [&](int r, int g, int b) -> unsigned {
        const unsigned acc = (unsigned)__mul24(k.ry, r) + (unsigned)__mul24(k.gy, g) + (unsigned)__mul24(k.by, b);
        if (BPS == 2) return (acc + (0x2001u << 14)) >> 15;
        return (unsigned)((int)(acc + (unsigned)((32 << 14) + (1 << 8))) >> 9);
    };
    // chroma of one pixel (full) or of a pixel pair (half): the 8-bit readers work on the pair's SUMS, the 16-bit ones on its rounded means
    auto chr = [&](int r0, int g0, int b0, int r1, int g1, int b1, bool pair, unsigned &u, unsigned &v) {
        int r, g, b, add, sh;
        if (BPS == 2)  { r = pair ? (r0 + r1 + 1) >> 1 : r0; g = pair ? (g0 + g1 + 1) >> 1 : g0; b = pair ? (b0 + b1 + 1) >> 1 : b0; add = 0x10001 << 14; sh = 15; }
        else if (pair) { r = r0 + r1; g = g0 + g1; b = b0 + b1; add = (256 << 15) + (1 << 9); sh = 10; }
        else           { r = r0; g = g0; b = b0; add = (256 << 14) + (1 << 8); sh = 9; }
        u = (unsigned)((int)((unsigned)__mul24(k.ru, r) + (unsigned)__mul24(k.gu, g) + (unsigned)__mul24(k.bu, b) + (unsigned)add) >> sh) & 0xFFFFu;
        v = (unsigned)((int)((unsigned)__mul24(k.rv, r) + (unsigned)__mul24(k.gv, g) + (unsigned)__mul24(k.bv, b) + (unsigned)add) >> sh) & 0xFFFFu;
    };
    if (x + 4 <= w) {
        int c[4][3];                                                            // [pixel][r, g, b]
        if (al4) {
            constexpr int NDW = BPS == 2 ? 8 : PX;                             // dwords of four pixels: 8 (64-bit), 3 (24-bit), 4 (32-bit)
            unsigned d[NDW];
            const unsigned *p = reinterpret_cast<const unsigned *>(row + (size_t)x * (BPS == 2 ? 8 : PX));
#pragma unroll
            for (int i = 0; i < NDW; i++) d[i] = p[i];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    if (BPS == 2) { const int hw = 4 * i + ch; c[i][ch] = (int)(d[hw >> 1] >> (16 * (hw & 1)) & 0xFFFFu); }
                    else          { const int bt = PX * i + ch; c[i][ch] = (int)(d[bt >> 2] >> (8 * (bt & 3)) & 0xFFu); }
                }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) c[i][ch] = sample(x + i, ch);
        }
        // (red and blue by a select, not by an index: `c[i][ro]` with a run-time ro put the array into scratch — 24.6 -> 43.4 us a frame on the first try, r06o)
        int R[4], G[4], B[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { R[i] = bgr ? c[i][2] : c[i][0]; G[i] = c[i][1]; B[i] = bgr ? c[i][0] : c[i][2]; }
        unsigned yy[4];
#pragma unroll
        for (int i = 0; i < 4; i++) yy[i] = lum(R[i], G[i], B[i]) & 0xFFFFu;
        *reinterpret_cast<uint2 *>(oy + x) = make_uint2(yy[0] | yy[1] << 16, yy[2] | yy[3] << 16);
        if (half) {
            unsigned u0, v0, u1, v1;
            chr(R[0], G[0], B[0], R[1], G[1], B[1], true, u0, v0);
            chr(R[2], G[2], B[2], R[3], G[3], B[3], true, u1, v1);
            *reinterpret_cast<unsigned *>(ou + (x >> 1)) = u0 | u1 << 16;
            *reinterpret_cast<unsigned *>(ov + (x >> 1)) = v0 | v1 << 16;
        } else {
            unsigned u[4], v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) chr(R[i], G[i], B[i], 0, 0, 0, false, u[i], v[i]);
            *reinterpret_cast<uint2 *>(ou + x) = make_uint2(u[0] | u[1] << 16, u[2] | u[3] << 16);
            *reinterpret_cast<uint2 *>(ov + x) = make_uint2(v[0] | v[1] << 16, v[2] | v[3] << 16);
        }
    } else {
        // the row's ragged end: one to three pixels, as the kernels above (an odd width with halved chroma uses the last pixel twice)
        for (int xx = x; xx < w; xx++) oy[xx] = (unsigned short)lum(sample(xx, ro), sample(xx, 1), sample(xx, bo));
        if (half) {
            for (int cx = x >> 1; cx < chrW; cx++) {
                const int x0 = 2 * cx, x1 = min(2 * cx + 1, w - 1);
                unsigned u, v;
                chr(sample(x0, ro), sample(x0, 1), sample(x0, bo), sample(x1, ro), sample(x1, 1), sample(x1, bo), true, u, v);
                ou[cx] = (unsigned short)u; ov[cx] = (unsigned short)v;
            }
        } else {
            for (int xx = x; xx < w; xx++) {
                unsigned u, v;
                chr(sample(xx, ro), sample(xx, 1), sample(xx, bo), 0, 0, 0, false, u, v);
                ou[xx] = (unsigned short)u; ov[xx] = (unsigned short)v;
            }
        }
    }
}

int launch_rgb8_planes(const uint8_t *src, int ss, int w, int h, int chrW, int half, int bgr, int px, const Rgb2YuvConsts &k,
                       uint8_t *py, int ys, uint8_t *pu, int us, uint8_t *pv, int vs, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return 0;
    const bool planesAl = ((((uintptr_t)py | (uintptr_t)ys | (uintptr_t)pu | (uintptr_t)us | (uintptr_t)pv | (uintptr_t)vs) & 7) == 0);
    const char *kq = GMAT_KNOB("GMAT_RGB_PLANES4");
    if (planesAl && (px == 3 || px == 4) && !(kq && atoi(kq) == 0)) {
        const dim3 grid((w + 1023) / 1024, h), block(256);
        const int al4 = ((((uintptr_t)src | (uintptr_t)ss) & 3) == 0) ? 1 : 0;
        if (px == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(rgb_planes4_kernel<1, 3>), grid, block, 0, stream, src, ss, w, h, chrW, half, bgr, al4, k, py, ys, pu, us, pv, vs);
        else         hipLaunchKernelGGL(HIP_KERNEL_NAME(rgb_planes4_kernel<1, 4>), grid, block, 0, stream, src, ss, w, h, chrW, half, bgr, al4, k, py, ys, pu, us, pv, vs);
        GMAT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const dim3 grid((chrW + 255) / 256, h), block(256);
    hipLaunchKernelGGL(rgb8_planes_kernel, grid, block, 0, stream, src, ss, w, h, chrW, half, bgr, px, k, py, ys, pu, us, pv, vs);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_rgb64_planes(const uint8_t *src, int ss, int w, int h, int chrW, int half, int bgr, const Rgb2YuvConsts &k,
                        uint8_t *py, int ys, uint8_t *pu, int us, uint8_t *pv, int vs, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return 0;
    const bool planesAl = ((((uintptr_t)py | (uintptr_t)ys | (uintptr_t)pu | (uintptr_t)us | (uintptr_t)pv | (uintptr_t)vs) & 7) == 0);
    const char *kq = GMAT_KNOB("GMAT_RGB_PLANES4");
    if (planesAl && !(kq && atoi(kq) == 0)) {
        const dim3 grid((w + 1023) / 1024, h), block(256);
        const int al4 = ((((uintptr_t)src | (uintptr_t)ss) & 3) == 0) ? 1 : 0;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(rgb_planes4_kernel<2, 4>), grid, block, 0, stream, src, ss, w, h, chrW, half, bgr, al4, k, py, ys, pu, us, pv, vs);
        GMAT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const dim3 grid((chrW + 255) / 256, h), block(256);
    hipLaunchKernelGGL(rgb64_planes_kernel, grid, block, 0, stream, src, ss, w, h, chrW, half, bgr, k, py, ys, pu, us, pv, vs);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// The alpha byte of an RGBA / BGRA destination from the 15-bit alpha lines (srcH x dstW int32), one pixel per thread.  The form
// per output row is packed_vscale's (vscale.c:135-167), chosen on the host from the luma AND chroma filters as for the colour
// channels: form[y] = m | yalpha << 8 (yalpha = the luma filter's second coefficient), first[y] = the luma filter's position.
//   m  writer                          output.c      alpha                                           clipped to 8 bits
//   0  yuv2rgb_X_c                     :1709-1720    (2^18 + sum a * f) >> 19                        when bit 8 is set (*)
//   4  yuv2rgb_full_X_c                :2069-2077    the same                                        when bit 8 is set
//   1  yuv2rgb_1_c, uvalpha < 2048     :1794-1799    (a * 255 + 16384) >> 15                         always
//   2  yuv2rgb_1_c, uvalpha >= 2048    :1816-1821    (a + 64) >> 7                                   always
//   3  yuv2rgb_2_c                     :1761-1766    (a0 * (4096 - yalpha) + a1 * yalpha) >> 19      always
//   5  yuv2rgb_full_2_c                :2117-2121    (a0 * (4096 - yalpha) + a1 * yalpha + 2^18) >> 19   when bit 8 is set
//   6  yuv2rgb_full_1_c                :2154-2158,:2171-2175   (a + 64) >> 7                         when bit 8 is set
// (*) the half-chroma X writer tests the OR of a pixel pair; the partner's bit can only matter for a value >= 512 without bit 8,
// i.e. a filter row whose coefficients sum to more than 2^28 / 32767 = 2 * 4096 in absolute value, which initFilter does not make.
__global__ __launch_bounds__(256) void alpha8_out_kernel(const int32_t *la, int lineW, int lineH, DevFilter f, const int32_t *form,
                                                         const int32_t *first, uint8_t *dst, int ds, int dstW, int dstH)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dstW || y >= dstH) return;
    const int m = form[y] & 0xFF, ya = form[y] >> 8;
    int A;
    if (m == 0 || m == 4) {
        const int p0 = f.pos_even[y];
        A = 1 << 18;
        for (int k = 0; k < f.pairs; k++) {
            const int cf = f.packed[(size_t)y * f.pairs + k];
            const int r0 = min(p0 + 2 * k, lineH - 1), r1 = min(p0 + 2 * k + 1, lineH - 1);
            A += la[(size_t)r0 * lineW + x] * (int)(short)(cf & 0xFFFF) + la[(size_t)r1 * lineW + x] * (cf >> 16);
        }
        A >>= 19;
    } else {
        const int r0 = min(first[y], lineH - 1), r1 = min(first[y] + 1, lineH - 1);
        const int a0 = la[(size_t)r0 * lineW + x];
        if (m == 1)                A = (a0 * 255 + 16384) >> 15;
        else if (m == 2 || m == 6) A = (a0 + 64) >> 7;
        else                       A = (a0 * (4096 - ya) + la[(size_t)r1 * lineW + x] * ya + (m == 5 ? (1 << 18) : 0)) >> 19;
    }
    const bool always = m >= 1 && m <= 3;
    if (always || (A & 0x100)) A = clip_u8(A);
    dst[(size_t)y * ds + 4 * x + 3] = (uint8_t)A;
}

int launch_alpha8_out(const int32_t *la, int lineW, int lineH, const DevFilter &f, const int32_t *form, const int32_t *first,
                      uint8_t *dst, int ds, int dstW, int dstH, hipStream_t stream)
{
    if (dstW <= 0 || dstH <= 0) return 0;
    const dim3 grid((dstW + 255) / 256, dstH), block(256);
    hipLaunchKernelGGL(alpha8_out_kernel, grid, block, 0, stream, la, lineW, lineH, f, form, first, dst, ds, dstW, dstH);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
