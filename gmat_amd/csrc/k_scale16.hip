// k_scale16.hip — libswscale's generic scaler for 16-bit destinations (P016LE, RGBA64LE / BGRA64LE), gfx950.
//
// A 16-bit destination switches the CPU path to 19-bit intermediate lines held in int32 (dstBpc = 16, utils.c:1561-1570):
//   horizontal   hScale8To19_c    min(sum >> 3, 2^19 - 1)                          swscale.c:138-153
//                hScale16To19_c   min(sum >> (depth - 5), 2^19 - 1)                swscale.c:63-91
//   vertical     yuv2planeX_16_c  0x8000 + clip_int16(((1 << 14) - 0x40000000 + sum src * (unsigned)filter) >> 15)
//                                 in 32-bit wrap-around arithmetic                 output.c:157-181
//                yuv2plane1_16_c  clip_uint16((src + 4) >> 3) — the same value as the X form with coefficient 4096
//                yuv2nv12cX_16_c  the X form per chroma plane, interleaved, also for one tap   output.c:183-211
// The 15-bit kernels of k_scale_yuv.hip keep their lines as int16 pairs for v_dot2; 19-bit lines do not fit, so this
// path is a plain two-pass one: pass 1 filters every source row horizontally into an int32 plane in HBM (srcH x dstW),
// pass 2 filters that plane vertically and stores 16-bit samples.  One output sample per thread, taps read through
// the caches.  A completeness path (scale_cuda lists P016 as an output, vf_scale_cuda.c:45-54), not a fast one.
#include <hip/hip_runtime.h>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

namespace gmat {

// kind: 0 = 8-bit samples, sample stride `step` bytes; 10 / 16 = 16-bit samples (P010: >> 6), sample stride `step` bytes;
//       110 = 16-bit containers holding 10 bits in the low end (YUV420P10LE): as they are, sh of a 10-bit source;
//       208 = 8-bit alpha samples as rgbaToA_c hands them to the scaler (a << 6 | a >> 2, input.c:442-449);
//       14 = the 16-bit lines of an 8-bit packed RGB source (rgb24ToY_c ...): raw 16-bit samples, hScale16To19_c's sh = 9
// maxv: 2^19 - 1 (hScale*To19_c) or 2^15 - 1 (hScale16To15_c, swscale.c:93-119 — the alpha lines of an 8-bit destination)
__global__ __launch_bounds__(256) void hscale19_kernel(const uint8_t *src, int ss, int kind, int step, int srcW, int srcH,
                                                       DevFilter f, int32_t *dst, int dstW, int sh, int maxv, int rc)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dstW || y >= srcH) return;
    const uint8_t *row = src + (size_t)y * ss;
    const int p0 = f.pos_even[x];
    auto sample = [&](int i) -> int {                                 // a tap past the plane has coefficient 0
        i = min(i, srcW - 1);
        if (kind == 0) return row[(size_t)i * step];
        if (kind == 208) { const int a = row[(size_t)i * step]; return a << 6 | a >> 2; }
        const unsigned v = *reinterpret_cast<const unsigned short *>(row + (size_t)i * step);
        return kind == 10 ? (int)(v >> 6) : (int)v;
    };
    // (round 5, last hour: measured for the first time — P016 1080p -> 720p 41 us a frame, 70 % of it here — and latency-bound: a loop over f.pairs waits for every
    // pair's loads before it asks for the next.  Up to eight pairs: all of a thread's loads are issued together — NP >= f.pairs of them, the pairs past the
    // filter's own with coefficient 0 on a clamped, valid sample — then summed in the same order)
    int val = 0;
    auto taps = [&](auto np_c) {
        constexpr int NP = decltype(np_c)::value;
        int cf[NP], s0[NP], s1[NP];
#pragma unroll
        for (int k = 0; k < NP; k++) cf[k] = k < f.pairs ? f.packed[(size_t)x * f.pairs + k] : 0;
#pragma unroll
        for (int k = 0; k < NP; k++) { s0[k] = sample(p0 + 2 * k); s1[k] = sample(p0 + 2 * k + 1); }
#pragma unroll
        for (int k = 0; k < NP; k++) val += s0[k] * (int)(short)(cf[k] & 0xFFFF) + s1[k] * (cf[k] >> 16);
    };
    if (f.pairs <= 2)      taps(std::integral_constant<int, 2>());
    else if (f.pairs <= 4) taps(std::integral_constant<int, 4>());
    else if (f.pairs <= 6) taps(std::integral_constant<int, 6>());
    else if (f.pairs <= 8) taps(std::integral_constant<int, 8>());
    else
        for (int k = 0; k < f.pairs; k++) {
            const int cf = f.packed[(size_t)x * f.pairs + k];
            val += sample(p0 + 2 * k) * (int)(short)(cf & 0xFFFF) + sample(p0 + 2 * k + 1) * (cf >> 16);
        }
    int v = min(val >> sh, maxv);
    // range conversion of a 19-bit line (lum / chrRange{To,From}Jpeg16_c, swscale.c:189-226) in the reference's 32-bit arithmetic: the
    // chroma ToJpeg product passes 2^31 on its way, only the difference fits
    if (rc == 1)      v = (int)((unsigned)min(v, 30189 << 4) * 4769u - (unsigned)(39057361 << 2)) >> 12;
    else if (rc == 2) v = (int)((unsigned)v * (unsigned)(14071 / 4) + (unsigned)((33561947 << 4) / 4)) >> 12;
    else if (rc == 3) v = (int)((unsigned)min(v, 30775 << 4) * 4663u - (unsigned)(9289992 << 4)) >> 12;
    else if (rc == 4) v = (int)((unsigned)v * 1799u + (unsigned)(4081085 << 4)) >> 11;
    dst[(size_t)y * dstW + x] = v;
}

// planes == 1: one plane -> 16-bit samples at dst + 2x.  planes == 2: U and V lines -> interleaved 16-bit pairs at dst + 4x.
__global__ __launch_bounds__(256) void vscale16_kernel(const int32_t *lineA, const int32_t *lineB, int lineW, int lineH, DevFilter f,
                                                       uint8_t *dst, int ds, int dstW, int dstH, int planes)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dstW || y >= dstH) return;
    const int p0 = f.pos_even[y];
    unsigned a = (1u << 14) - 0x40000000u, b = a;
    // (as hscale19_kernel: up to eight pairs a thread asks for all of its lines' samples at once — the loop waited for each pair)
    auto taps = [&](auto np_c) {
        constexpr int NP = decltype(np_c)::value;
        int cf[NP];
        unsigned a0[NP], a1[NP], b0[NP], b1[NP];
#pragma unroll
        for (int k = 0; k < NP; k++) cf[k] = k < f.pairs ? f.packed[(size_t)y * f.pairs + k] : 0;     // (the coefficient of a row: wave-uniform)
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const int r0 = min(p0 + 2 * k, lineH - 1), r1 = min(p0 + 2 * k + 1, lineH - 1);
            a0[k] = (unsigned)lineA[(size_t)r0 * lineW + x]; a1[k] = (unsigned)lineA[(size_t)r1 * lineW + x];
            if (planes == 2) { b0[k] = (unsigned)lineB[(size_t)r0 * lineW + x]; b1[k] = (unsigned)lineB[(size_t)r1 * lineW + x]; } else b0[k] = b1[k] = 0;
        }
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const unsigned c0 = (unsigned)(int)(short)(cf[k] & 0xFFFF), c1 = (unsigned)(cf[k] >> 16);
            a += a0[k] * c0 + a1[k] * c1;
            b += b0[k] * c0 + b1[k] * c1;
        }
    };
    if (f.pairs <= 2)      taps(std::integral_constant<int, 2>());
    else if (f.pairs <= 4) taps(std::integral_constant<int, 4>());
    else if (f.pairs <= 6) taps(std::integral_constant<int, 6>());
    else if (f.pairs <= 8) taps(std::integral_constant<int, 8>());
    else
        for (int k = 0; k < f.pairs; k++) {
            const int cf = f.packed[(size_t)y * f.pairs + k];
            const unsigned c0 = (unsigned)(int)(short)(cf & 0xFFFF), c1 = (unsigned)(cf >> 16);
            const int r0 = min(p0 + 2 * k, lineH - 1), r1 = min(p0 + 2 * k + 1, lineH - 1);
            a += (unsigned)lineA[(size_t)r0 * lineW + x] * c0 + (unsigned)lineA[(size_t)r1 * lineW + x] * c1;
            if (planes == 2) b += (unsigned)lineB[(size_t)r0 * lineW + x] * c0 + (unsigned)lineB[(size_t)r1 * lineW + x] * c1;
        }
    const int va = min(max((int)a >> 15, -32768), 32767) + 0x8000;
    if (planes == 1) {
        reinterpret_cast<unsigned short *>(dst + (size_t)y * ds)[x] = (unsigned short)va;
    } else {
        const int vb = min(max((int)b >> 15, -32768), 32767) + 0x8000;
        unsigned short *d = reinterpret_cast<unsigned short *>(dst + (size_t)y * ds) + 2 * x;
        d[0] = (unsigned short)va; d[1] = (unsigned short)vb;
    }
}

// RGBA64LE / BGRA64LE from the 19-bit lines: yuv2rgba64_X_c / _full_X_c (output.c:1025-1105, :1275-1337); the 1- and
// 2-tap forms (_1_c, _2_c) are the same values with the effective coefficients the host prepares (gsws.cpp):
//   Y = ((-2^30 + sum lum * f) >> 14) + 2^16;  U, V = (-(128 << 23) + sum chr * f) >> 14      (32-bit wrap-around sums)
//   Y = (Y - y_offset) * y_coeff + (1 << 13);  R = V * v2r;  G = V * v2g + U * u2g;  B = U * u2b
//   channel = clip_uintp2(X + Y, 30) >> 14
//   alpha: 0xFFFF without an alpha plane (la == nullptr); else the X form's ((-2^30 + sum a * f) >> 1) + 0x20002000 -> clip_uintp2(., 30)
//   >> 14 (:1052-1064) on the same effective luma coefficients — the _2 form ((a0 * yalpha1 + a1 * yalpha) >> 1) + 2^13 (:1144-1150)
//   and the _1 form (a << 11) + 2^13 (:1196-1202) are that value exactly (2^30 is even; one tap of 4096 is the shift by 11 and 1)
// One pixel per thread; chrShift = 1: one chroma sample per pixel pair.
__global__ __launch_bounds__(256) void vrgba64_kernel(const int32_t *ly, const int32_t *lu, const int32_t *lv, int lumW, int lumH,
                                                      int chrW, int chrH, DevFilter fl, DevFilter fc, int chrShift, uint8_t *dst,
                                                      int ds, int dstW, int dstH, int bgr, Yuv2RgbConsts k, const int32_t *la)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dstW || y >= dstH) return;
    const int cx = x >> chrShift;
    unsigned ay = (unsigned)-0x40000000, au = (unsigned)-(128 << 23), av = au, aa = ay;
    const int pl = fl.pos_even[y], pc = fc.pos_even[y];
    for (int t = 0; t < fl.pairs; t++) {
        const int cf = fl.packed[(size_t)y * fl.pairs + t];
        const int r0 = min(pl + 2 * t, lumH - 1), r1 = min(pl + 2 * t + 1, lumH - 1);
        ay += (unsigned)ly[(size_t)r0 * lumW + x] * (unsigned)(int)(short)(cf & 0xFFFF) + (unsigned)ly[(size_t)r1 * lumW + x] * (unsigned)(cf >> 16);
        if (la) aa += (unsigned)la[(size_t)r0 * lumW + x] * (unsigned)(int)(short)(cf & 0xFFFF) + (unsigned)la[(size_t)r1 * lumW + x] * (unsigned)(cf >> 16);
    }
    for (int t = 0; t < fc.pairs; t++) {
        const int cf = fc.packed[(size_t)y * fc.pairs + t];
        const unsigned c0 = (unsigned)(int)(short)(cf & 0xFFFF), c1 = (unsigned)(cf >> 16);
        const int r0 = min(pc + 2 * t, chrH - 1), r1 = min(pc + 2 * t + 1, chrH - 1);
        au += (unsigned)lu[(size_t)r0 * chrW + cx] * c0 + (unsigned)lu[(size_t)r1 * chrW + cx] * c1;
        av += (unsigned)lv[(size_t)r0 * chrW + cx] * c0 + (unsigned)lv[(size_t)r1 * chrW + cx] * c1;
    }
    int Y = ((int)ay >> 14) + 0x10000;
    const int U = (int)au >> 14, V = (int)av >> 14;
    Y = (Y - k.y_offset) * k.y_coeff + (1 << 13);
    const int R = V * k.v2r, G = V * k.v2g + U * k.u2g, B = U * k.u2b;
    auto ch = [&](int v) -> unsigned { return (unsigned)min(max(v + Y, 0), 0x3FFFFFFF) >> 14; };
    const unsigned c0 = ch(bgr ? B : R), c1 = ch(G), c2 = ch(bgr ? R : B);
    unsigned short *d = reinterpret_cast<unsigned short *>(dst + (size_t)y * ds) + 4 * x;
    d[0] = (unsigned short)c0; d[1] = (unsigned short)c1; d[2] = (unsigned short)c2;
    d[3] = la ? (unsigned short)((unsigned)min(max(((int)aa >> 1) + 0x20002000, 0), 0x3FFFFFFF) >> 14) : (unsigned short)0xFFFF;
}

int launch_vrgba64(const int32_t *ly, const int32_t *lu, const int32_t *lv, int lumW, int lumH, int chrW, int chrH, const DevFilter &fl,
                   const DevFilter &fc, int chrShift, uint8_t *dst, int ds, int dstW, int dstH, int bgr, const Yuv2RgbConsts &k,
                   hipStream_t stream, const int32_t *la)
{
    if (dstW <= 0 || dstH <= 0) return 0;
    const dim3 grid((dstW + 255) / 256, dstH), block(256);
    hipLaunchKernelGGL(vrgba64_kernel, grid, block, 0, stream, ly, lu, lv, lumW, lumH, chrW, chrH, fl, fc, chrShift, dst, ds, dstW, dstH, bgr, k, la);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_hscale19(const uint8_t *src, int ss, int kind, int step, int srcW, int srcH, const DevFilter &f, int32_t *dst, int dstW,
                    hipStream_t stream, int to15, int rangeConv)
{
    if (dstW <= 0 || srcH <= 0) return 0;
    // hScale8To19_c: 3; hScale16To19_c: depth - 1 - 4; hScale16To15_c (to15): depth - 1, 13 for the 14-bit alpha of an 8-bit format
    // (an 8-bit RGB source's lines — kind 14, and its alpha, kind 208 — go to 19 bits by 9: swscale.c:74-76)
    const int sh = to15 ? (kind == 208 ? 13 : kind % 100 - 1) : kind == 0 ? 3 : kind == 208 ? 9 : kind % 100 - 5;
    const dim3 grid((dstW + 255) / 256, srcH), block(256);
    hipLaunchKernelGGL(hscale19_kernel, grid, block, 0, stream, src, ss, kind, step, srcW, srcH, f, dst, dstW, sh, to15 ? (1 << 15) - 1 : (1 << 19) - 1, rangeConv);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_vscale16(const int32_t *lineA, const int32_t *lineB, int lineW, int lineH, const DevFilter &f, uint8_t *dst, int ds,
                    int dstW, int dstH, hipStream_t stream)
{
    if (dstW <= 0 || dstH <= 0) return 0;
    const dim3 grid((dstW + 255) / 256, dstH), block(256);
    hipLaunchKernelGGL(vscale16_kernel, grid, block, 0, stream, lineA, lineB, lineW, lineH, f, dst, ds, dstW, dstH, lineB ? 2 : 1);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
