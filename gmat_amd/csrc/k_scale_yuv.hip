// k_scale_yuv.hip — libswscale's generic scaler for 8-bit YUV 4:2:0 sources (NV12, YUV420P) with
// packed-RGB output, as one fused LDS-tiled kernel for gfx950.  Integer arithmetic, bit-exact with
// what ONE libswscale context (sws_getContext(nv12 -> rgb24, different size)) computes on the CPU:
//   horizontal   hScale8To15_c: min(sum(src*f) >> 7, 32767), luma and both chroma planes   swscale.c:122-136
//   vertical+out half chroma : yuv2rgb_X_c / _2_c / _1_c + the yuv2rgb.c tables (closed form)  output.c:1680-1828
//                full chroma : yuv2rgb_full_X_c / _2_c / _1_c + yuv2rgb_write_full             output.c:1886-2200
//   form choice  packed_vscale                                                                  vscale.c:135-167
// The 1-tap and 2-tap special forms are folded into per-row accumulator start values and an
// "effective" vertical chroma filter prepared on the host (yuvscale_prepare), so the kernel runs one
// generic loop.
//
// HBM traffic per 4K->1080p frame: 12.4 MB of NV12 in, 6.2 MB of RGB24 out (2.25 B per source pixel).
// Block = 256 threads = one TW x TH output tile; phases: (1) window load, u8 -> int16 in LDS;
// (2) horizontal v_dot2c_i32_i16 over dword pairs, two source rows per item, row-pair-interleaved
// int16 result; (3) vertical dot2 from ds_read_b128 vectors + colour stage + 12/16-byte stores.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

namespace gmat {

constexpr int kYMaxPairs = 8;
constexpr int kLumaBatch = 6, kChromaBatch = 3;

__device__ __forceinline__ unsigned pk16(int lo, int hi) { return ((unsigned)lo & 0xFFFF) | ((unsigned)hi << 16); }

template <bool FAST>
__device__ __forceinline__ unsigned ld_y4(const YuvScaleArgs &a, int srow, int col)
{
    const uint8_t *row = a.y + (size_t)srow * a.ys;
    if (FAST) return *reinterpret_cast<const unsigned *>(row + col);
    unsigned v = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) v |= (unsigned)row[min(col + i, a.srcW - 1)] << (8 * i);
    return v;
}

// two chroma samples starting at even chroma column cc: returns U0 | V0<<8 | U1<<16 | V1<<24
template <bool FAST>
__device__ __forceinline__ unsigned ld_uv2(const YuvScaleArgs &a, int crow, int cc)
{
    if (FAST) {
        if (a.nv12) return *reinterpret_cast<const unsigned *>(a.u + (size_t)crow * a.us + 2 * cc);
        const unsigned uu = *reinterpret_cast<const unsigned short *>(a.u + (size_t)crow * a.us + cc);
        const unsigned vv = *reinterpret_cast<const unsigned short *>(a.v + (size_t)crow * a.vs + cc);
        return (uu & 0xFF) | ((vv & 0xFF) << 8) | ((uu >> 8) << 16) | ((vv >> 8) << 24);
    }
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int c = min(cc + i, a.chrSrcW - 1);
        unsigned U, V;
        if (a.nv12) { const uint8_t *q = a.u + (size_t)crow * a.us + 2 * c; U = q[0]; V = q[1]; }
        else        { U = a.u[(size_t)crow * a.us + c]; V = a.v[(size_t)crow * a.vs + c]; }
        r |= (U | (V << 8)) << (16 * i);
    }
    return r;
}

template <bool FASTL, bool FASTC>
__device__ __forceinline__ void yuv_phase1(const YuvScaleArgs &a, int tid, int c0L, int ncL, int r0L, int nrL,
                                           int c0C, int ncC, int r0C, int nrC, unsigned short *ly,
                                           unsigned short *lu, unsigned short *lv)
{
    // luma: 4 pixels per dword
    {
        const int ng = ncL >> 2, total = nrL * ng;
        for (int base = tid; base < total; base += 256 * kLumaBatch) {
            unsigned raw[kLumaBatch];
            int off[kLumaBatch];
#pragma unroll
            for (int j = 0; j < kLumaBatch; j++) {
                const int g = base + j * 256, gg = min(g, total - 1);
                const int r = gg / ng, cg = gg - r * ng;
                off[j] = g < total ? r * a.colsL + 4 * cg : -1;
                raw[j] = ld_y4<FASTL>(a, min(r0L + r, a.srcH - 1), c0L + 4 * cg);
            }
#pragma unroll
            for (int j = 0; j < kLumaBatch; j++) {
                if (off[j] < 0) continue;
                const unsigned v = raw[j];
                *reinterpret_cast<uint2 *>(ly + off[j]) =
                    make_uint2((v & 0xFF) | ((v & 0xFF00) << 8), ((v >> 16) & 0xFF) | ((v >> 24) << 16));
            }
        }
    }
    // chroma: 2 samples of each plane per dword (NV12) / pair of ushorts (planar)
    {
        const int ng = ncC >> 1, total = nrC * ng;
        for (int base = tid; base < total; base += 256 * kChromaBatch) {
            unsigned raw[kChromaBatch];
            int off[kChromaBatch];
#pragma unroll
            for (int j = 0; j < kChromaBatch; j++) {
                const int g = base + j * 256, gg = min(g, total - 1);
                const int r = gg / ng, cg = gg - r * ng;
                off[j] = g < total ? r * a.colsC + 2 * cg : -1;
                raw[j] = ld_uv2<FASTC>(a, min(r0C + r, a.chrSrcH - 1), c0C + 2 * cg);
            }
#pragma unroll
            for (int j = 0; j < kChromaBatch; j++) {
                if (off[j] < 0) continue;
                const unsigned v = raw[j];
                *reinterpret_cast<unsigned *>(lu + off[j]) = (v & 0xFF) | (v & 0xFF0000);
                *reinterpret_cast<unsigned *>(lv + off[j]) = ((v >> 8) & 0xFF) | ((v >> 8) & 0xFF0000);
            }
        }
    }
}

// 16-byte form of phase 1 (rows and window starts 16-byte aligned, window inside the frame): a thread owns up to
// K16 chunks of 16 source bytes, ALL of its loads are issued before the first LDS write.  The dword-per-lane form
// above needs several dependent HBM round trips per tile (6 / 3 groups in flight), which dominated the generic
// scalers the same way it dominated the 3x3 smooth (k_transform.hip).
constexpr int K16L = 5, K16C = 3;

__device__ __forceinline__ bool yuv_phase1_16(const YuvScaleArgs &a, int tid, int c0L, int ncL, int r0L, int nrL,
                                              int c0C, int ncC, int r0C, int nrC, unsigned short *ly,
                                              unsigned short *lu, unsigned short *lv)
{
    const int nL = ncL >> 4, totL = nrL * nL;           // luma chunks: 16 samples
    const int nC = ncC >> 3, totC = nrC * nC;           // chroma chunks: 8 samples of each plane
    if (totL > 256 * K16L || totC > 256 * K16C) return false;      // block-uniform
    uint4 vl[K16L], vc[K16C];
    int ol[K16L], oc[K16C];
#pragma unroll
    for (int k = 0; k < K16L; k++) {
        const int id = tid + 256 * k;
        ol[k] = -1; vl[k] = make_uint4(0u, 0u, 0u, 0u);
        if (id < totL) {
            const int r = id / nL, c = id - r * nL;
            ol[k] = r * a.colsL + 16 * c;
            vl[k] = *reinterpret_cast<const uint4 *>(a.y + (size_t)min(r0L + r, a.srcH - 1) * a.ys + c0L + 16 * c);
        }
    }
#pragma unroll
    for (int k = 0; k < K16C; k++) {
        const int id = tid + 256 * k;
        oc[k] = -1; vc[k] = make_uint4(0u, 0u, 0u, 0u);
        if (id < totC) {
            const int r = id / nC, c = id - r * nC;
            const size_t crow = (size_t)min(r0C + r, a.chrSrcH - 1);
            oc[k] = r * a.colsC + 8 * c;
            if (a.nv12) {
                vc[k] = *reinterpret_cast<const uint4 *>(a.u + crow * a.us + 2 * (c0C + 8 * c));    // U0 V0 U1 V1 ...
            } else {
                const uint2 tu = *reinterpret_cast<const uint2 *>(a.u + crow * a.us + c0C + 8 * c);
                const uint2 tv = *reinterpret_cast<const uint2 *>(a.v + crow * a.vs + c0C + 8 * c);
                // interleave to the NV12 byte order so the unpacking below is common
                vc[k].x = __builtin_amdgcn_perm(tv.x, tu.x, 0x05010400u); vc[k].y = __builtin_amdgcn_perm(tv.x, tu.x, 0x07030602u);
                vc[k].z = __builtin_amdgcn_perm(tv.y, tu.y, 0x05010400u); vc[k].w = __builtin_amdgcn_perm(tv.y, tu.y, 0x07030602u);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < K16L; k++) {
        if (ol[k] < 0) continue;
        uint4 *d = reinterpret_cast<uint4 *>(ly + ol[k]);
        const uint4 v = vl[k];
        d[0] = make_uint4(__builtin_amdgcn_perm(0u, v.x, 0x0C010C00u), __builtin_amdgcn_perm(0u, v.x, 0x0C030C02u),
                          __builtin_amdgcn_perm(0u, v.y, 0x0C010C00u), __builtin_amdgcn_perm(0u, v.y, 0x0C030C02u));
        d[1] = make_uint4(__builtin_amdgcn_perm(0u, v.z, 0x0C010C00u), __builtin_amdgcn_perm(0u, v.z, 0x0C030C02u),
                          __builtin_amdgcn_perm(0u, v.w, 0x0C010C00u), __builtin_amdgcn_perm(0u, v.w, 0x0C030C02u));
    }
#pragma unroll
    for (int k = 0; k < K16C; k++) {
        if (oc[k] < 0) continue;
        const uint4 v = vc[k];
        // U samples are bytes 0 and 2 of each dword, V samples bytes 1 and 3
        *reinterpret_cast<uint4 *>(lu + oc[k]) =
            make_uint4(__builtin_amdgcn_perm(0u, v.x, 0x0C020C00u), __builtin_amdgcn_perm(0u, v.y, 0x0C020C00u),
                       __builtin_amdgcn_perm(0u, v.z, 0x0C020C00u), __builtin_amdgcn_perm(0u, v.w, 0x0C020C00u));
        *reinterpret_cast<uint4 *>(lv + oc[k]) =
            make_uint4(__builtin_amdgcn_perm(0u, v.x, 0x0C030C01u), __builtin_amdgcn_perm(0u, v.y, 0x0C030C01u),
                       __builtin_amdgcn_perm(0u, v.z, 0x0C030C01u), __builtin_amdgcn_perm(0u, v.w, 0x0C030C01u));
    }
    return true;
}

// phase 1 for packed RGB24 / BGR24 sources with a YUV destination (src16 == 3): the LDS image holds what the CPU's
// input stage hands to hScale16To15_c (sh = 13 for RGB sources, swscale.c:93-119):
//   luma    rgb24ToY_c                     input.c:815-828   (ry*r + gy*g + by*b + (32 << 14) + (1 << 8)) >> 9
//   chroma  rgb24ToUV_c / rgb24ToUV_half_c input.c:830-866   per pixel, or on the sum of a horizontal pixel pair (>> 10)
// (the pair's second pixel is clamped to the last one for odd widths — the reference reads one pixel past the row there,
// which is not defined).  Both chroma planes have the
// source's height.  Two samples per item.
__device__ __forceinline__ void yuv_phase1_rgb(const YuvScaleArgs &a, int tid, int c0L, int ncL, int r0L, int nrL,
                                               int c0C, int ncC, int r0C, int nrC, unsigned short *ly,
                                               unsigned short *lu, unsigned short *lv)
{
    const int ro = a.rgbBgr ? 2 : 0, bo = 2 - ro;
    // 4 pixels of one row: three dwords when the rows are 4-byte aligned and the group lies inside the row (window
    // starts are multiples of 16 luma / 8 chroma samples, so groups start on 12-byte boundaries), else byte loads
    // with the column clamped to the last pixel
    auto load4 = [&](const uint8_t *row, int col, int (&r)[4], int (&g)[4], int (&b)[4]) {
        if (a.srcAligned && col >= 0 && col + 4 <= a.srcW) {
            const uint3 v = *reinterpret_cast<const uint3 *>(row + (size_t)col * 3);
            const unsigned c0[4] = {v.x & 0xFF, v.x >> 24, (v.y >> 16) & 0xFF, (v.z >> 8) & 0xFF};
            const unsigned c1[4] = {(v.x >> 8) & 0xFF, v.y & 0xFF, v.y >> 24, (v.z >> 16) & 0xFF};
            const unsigned c2[4] = {(v.x >> 16) & 0xFF, (v.y >> 8) & 0xFF, v.z & 0xFF, v.z >> 24};
#pragma unroll
            for (int i = 0; i < 4; i++) { r[i] = (int)(a.rgbBgr ? c2[i] : c0[i]); g[i] = (int)c1[i]; b[i] = (int)(a.rgbBgr ? c0[i] : c2[i]); }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint8_t *px = row + 3 * min(max(col + i, 0), a.srcW - 1);
                r[i] = px[ro]; g[i] = px[1]; b[i] = px[bo];
            }
        }
    };
    {
        const int ng = ncL >> 2, total = nrL * ng;           // luma: 4 pixels per item
        for (int it = tid; it < total; it += 256) {
            const int rr = it / ng, cg = it - rr * ng;
            int r[4], g[4], b[4];
            load4(a.y + (size_t)min(r0L + rr, a.srcH - 1) * a.ys, c0L + 4 * cg, r, g, b);
            const unsigned y0 = (unsigned)rgb_to_y14(a.r2y, r[0], g[0], b[0]), y1 = (unsigned)rgb_to_y14(a.r2y, r[1], g[1], b[1]);
            const unsigned y2 = (unsigned)rgb_to_y14(a.r2y, r[2], g[2], b[2]), y3 = (unsigned)rgb_to_y14(a.r2y, r[3], g[3], b[3]);
            *reinterpret_cast<uint2 *>(ly + rr * a.colsL + 4 * cg) = make_uint2(y0 | (y1 << 16), y2 | (y3 << 16));
        }
    }
    if (a.chrHalf) {
        const int ng = ncC >> 1, total = nrC * ng;           // chroma from pixel pairs: 2 samples = 4 pixels per item
        for (int it = tid; it < total; it += 256) {
            const int rr = it / ng, cg = it - rr * ng, cc = c0C + 2 * cg;
            int r[4], g[4], b[4];
            // chroma sample ci takes pixels 2ci and min(2ci + 1, srcW - 1); samples past the plane repeat the last one
            const int ci0 = min(cc, a.chrSrcW - 1), ci1 = min(cc + 1, a.chrSrcW - 1);
            if (ci1 == cc + 1 && 2 * cc + 4 <= a.srcW) {
                load4(a.y + (size_t)min(r0C + rr, a.srcH - 1) * a.ys, 2 * cc, r, g, b);
            } else {
                const uint8_t *row = a.y + (size_t)min(r0C + rr, a.srcH - 1) * a.ys;
                const int px[4] = {2 * ci0, min(2 * ci0 + 1, a.srcW - 1), 2 * ci1, min(2 * ci1 + 1, a.srcW - 1)};
#pragma unroll
                for (int i = 0; i < 4; i++) { const uint8_t *p = row + 3 * px[i]; r[i] = p[ro]; g[i] = p[1]; b[i] = p[bo]; }
            }
            const unsigned u0 = (unsigned)rgbsum_to_u14(a.r2y, r[0] + r[1], g[0] + g[1], b[0] + b[1]);
            const unsigned u1 = (unsigned)rgbsum_to_u14(a.r2y, r[2] + r[3], g[2] + g[3], b[2] + b[3]);
            const unsigned v0 = (unsigned)rgbsum_to_v14(a.r2y, r[0] + r[1], g[0] + g[1], b[0] + b[1]);
            const unsigned v1 = (unsigned)rgbsum_to_v14(a.r2y, r[2] + r[3], g[2] + g[3], b[2] + b[3]);
            *reinterpret_cast<unsigned *>(lu + rr * a.colsC + 2 * cg) = u0 | (u1 << 16);
            *reinterpret_cast<unsigned *>(lv + rr * a.colsC + 2 * cg) = v0 | (v1 << 16);
        }
    } else {
        const int ng = ncC >> 2, total = nrC * ng;           // one chroma sample per pixel: 4 per item
        for (int it = tid; it < total; it += 256) {
            const int rr = it / ng, cg = it - rr * ng;
            int r[4], g[4], b[4];
            load4(a.y + (size_t)min(r0C + rr, a.srcH - 1) * a.ys, c0C + 4 * cg, r, g, b);
            unsigned u[4], v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { u[i] = (unsigned)rgb_to_u14(a.r2y, r[i], g[i], b[i]); v[i] = (unsigned)rgb_to_v14(a.r2y, r[i], g[i], b[i]); }
            *reinterpret_cast<uint2 *>(lu + rr * a.colsC + 4 * cg) = make_uint2(u[0] | (u[1] << 16), u[2] | (u[3] << 16));
            *reinterpret_cast<uint2 *>(lv + rr * a.colsC + 4 * cg) = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
        }
    }
}

// phase 1 for 16-bit semi-planar sources (P010LE / P016LE): luma plane of 16-bit samples, chroma plane of interleaved
// 16-bit (U, V) pairs.  The LDS image holds what hScale16To15_c multiplies (swscale.c:93-119):
//   P010: sample >> 6 (p010LEToY_c / p010LEToUV_c, input.c:698-725) — 10 bits, non-negative as int16
//   P016: the sample itself; 16 unsigned bits do not fit the signed v_dot2 operand, so the image holds sample - 32768
//         (sample ^ 0x8000) and the horizontal accumulators start at 32768 * sum(coefficients) = 2^29 (YuvScaleArgs::hBias)
// Two samples per dword when the rows are 4-byte aligned, else 16-bit loads; columns past the plane repeat the last one.
__device__ __forceinline__ void yuv_phase1_src16(const YuvScaleArgs &a, int tid, int c0L, int ncL, int r0L, int nrL,
                                                 int c0C, int ncC, int r0C, int nrC, unsigned short *ly,
                                                 unsigned short *lu, unsigned short *lv)
{
    const bool p010 = a.src16 == 10;                         // 16 (P016LE) and 17 (planar 16-bit): the samples as they are
    const bool pl10 = a.src16 == 18;                         // planar 10-bit (YUV420P10LE): as they are, and they fit int16
    auto conv = [&](unsigned v) -> unsigned { return p010 ? (v >> 6) & 0x03FF03FFu : pl10 ? v : v ^ 0x80008000u; };   // both halves at once
    {
        const int ng = ncL >> 1, total = nrL * ng;           // items of 2 luma samples
        for (int it = tid; it < total; it += 256) {
            const int r = it / ng, cg = it - r * ng, col = c0L + 2 * cg;
            const uint8_t *row = a.y + (size_t)min(r0L + r, a.srcH - 1) * a.ys;
            unsigned v;
            if (a.srcAligned && col + 2 <= a.srcW) {
                v = *reinterpret_cast<const unsigned *>(row + 2 * col);
            } else {
                const unsigned s0 = *reinterpret_cast<const unsigned short *>(row + 2 * min(col, a.srcW - 1));
                const unsigned s1 = *reinterpret_cast<const unsigned short *>(row + 2 * min(col + 1, a.srcW - 1));
                v = s0 | (s1 << 16);
            }
            *reinterpret_cast<unsigned *>(ly + r * a.colsL + 2 * cg) = conv(v);
        }
    }
    {
        const int ng = ncC >> 1, total = nrC * ng;           // items of 2 chroma samples of each plane
        for (int it = tid; it < total; it += 256) {
            const int r = it / ng, cg = it - r * ng, cc = c0C + 2 * cg;
            const uint8_t *row = a.u + (size_t)min(r0C + r, a.chrSrcH - 1) * a.us;
            unsigned p0, p1;                                    // U | V << 16 of chroma samples cc, cc + 1
            if (a.src16 >= 17) {                                // planar: U and V planes of 16-bit samples
                const uint8_t *rowv = a.v + (size_t)min(r0C + r, a.chrSrcH - 1) * a.vs;
                const int k0 = min(cc, a.chrSrcW - 1), k1 = min(cc + 1, a.chrSrcW - 1);
                p0 = *reinterpret_cast<const unsigned short *>(row + 2 * k0) | ((unsigned)*reinterpret_cast<const unsigned short *>(rowv + 2 * k0) << 16);
                p1 = *reinterpret_cast<const unsigned short *>(row + 2 * k1) | ((unsigned)*reinterpret_cast<const unsigned short *>(rowv + 2 * k1) << 16);
            } else if (a.srcAligned && cc + 2 <= a.chrSrcW) {
                const uint2 t = *reinterpret_cast<const uint2 *>(row + 4 * cc);     // 4-byte aligned: two dword loads
                p0 = t.x; p1 = t.y;
            } else {
                const int k0 = min(cc, a.chrSrcW - 1), k1 = min(cc + 1, a.chrSrcW - 1);
                const unsigned short *q0 = reinterpret_cast<const unsigned short *>(row + 4 * k0);
                const unsigned short *q1 = reinterpret_cast<const unsigned short *>(row + 4 * k1);
                p0 = q0[0] | ((unsigned)q0[1] << 16); p1 = q1[0] | ((unsigned)q1[1] << 16);
            }
            p0 = conv(p0); p1 = conv(p1);
            *reinterpret_cast<unsigned *>(lu + r * a.colsC + 2 * cg) = (p0 & 0xFFFFu) | (p1 << 16);
            *reinterpret_cast<unsigned *>(lv + r * a.colsC + 2 * cg) = (p0 >> 16) | (p1 & 0xFFFF0000u);
        }
    }
}

// MODE 0: packed RGB out, half chroma (LUT form)   1: packed RGB out, full chroma
//      2: YUV 4:2:0 out (NV12 or YUV420P): the tile is TW x TH luma outputs plus the TW/2 x TH/2 chroma
//         outputs under them; vChr is indexed by CHROMA row; yuv2planeX_8_c / yuv2nv12cX_c (output.c:400-450)
//      3: YUV 4:4:4 planar out: chroma tile = luma tile
// LONG: horizontal filters longer than 2*kYMaxPairs taps (down-scale ratios beyond ~3.7:1).  A separate
// instantiation: with the tail loops compiled into the common variant every geometry paid for them (the 2x
// up-scale went from 38.7 to 47.3 us).
template <int TW, int MODE, bool LONG, bool SRC16>
__global__ __launch_bounds__(256) void scale_yuv_kernel(YuvScaleArgs a, Yuv2xFrames fr)
{
    {   // grid.y = frame of the batch (1 for a single frame): plane pointers from the kernel-argument segment
        const int f = blockIdx.y;
        a.y = fr.y[f]; a.u = fr.u[f]; a.v = fr.v[f];
        a.dst = fr.dst[f]; a.dstU = fr.dstU[f]; a.dstV = fr.dstV[f];
    }
    HIP_DYNAMIC_SHARED(uint4, lds_base)
    constexpr bool FULL = MODE == 1 || MODE == 3;       // chroma at full output width
    constexpr bool YUVOUT = MODE >= 2;
    constexpr int CVS = MODE == 2 ? 1 : 0;              // vertical chroma subsampling of the destination
    constexpr int CWD = FULL ? TW : TW / 2;            // chroma samples per output tile row

    int tcol, trow;
    {
        const int ntiles = a.ntx * a.nty;
        int lin = blockIdx.x;
        if (a.xcdRemap) {
            const int chunk = (ntiles + 7) >> 3;
            lin = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
        }
        if (lin >= ntiles) return;
        tcol = __builtin_amdgcn_readfirstlane(lin / a.nty);     // wave-uniform: table look-ups below become scalar loads
        trow = lin - tcol * a.nty;
    }
    const int tid = threadIdx.x;
    // optional phase timestamps (tuning aid): 6 x u64 per block, written by thread 0
    unsigned long long *prof = a.prof ? a.prof + (size_t)blockIdx.x * 8 : nullptr;
#define GMAT_STAMP(i) do { if (prof && tid == 0) prof[i] = __builtin_readcyclecounter(); } while (0)
    GMAT_STAMP(0);
    const int tx0 = tcol * TW, ty0 = trow * a.TH;
    const int tcx0 = FULL ? tx0 : tx0 >> 1;
    const int c0L = uniform_load(a.colStartL, tcol), ncL = uniform_load(a.colCountL, tcol);
    const int r0L = uniform_load(a.rowStartL, trow), nrL = uniform_load(a.rowCountL, trow);
    const int c0C = uniform_load(a.colStartC, tcol), ncC = uniform_load(a.colCountC, tcol);
    const int r0C = uniform_load(a.rowStartC, trow), nrC = uniform_load(a.rowCountC, trow);

    unsigned short *ly = reinterpret_cast<unsigned short *>(lds_base);
    unsigned short *lu = ly + a.rowsL * a.colsL;
    unsigned short *lv = lu + a.rowsC * a.colsC;
    int *hy = reinterpret_cast<int *>(lv + a.rowsC * a.colsC);
    int *hu = hy + (a.rowsL >> 1) * TW;
    int *hv = hu + (a.rowsC >> 1) * CWD;

    // ---- prologue: issue the loads of every filter coefficient / position this thread will use in
    // phases 2 and 3 NOW, so their latency overlaps the phase-1 pixel loads instead of serialising
    // behind the barriers (measured: phases 2/3 were dominated by these small dependent loads) ----
    const int xo2 = tid % TW, gx2 = min(tx0 + xo2, a.dstW - 1);
    const int xc2 = tid % CWD, gc2 = min(tcx0 + xc2, a.chrDstW - 1);
    int lc[kYMaxPairs], cc[kYMaxPairs];
#pragma unroll
    for (int k = 0; k < kYMaxPairs; k++) {
        lc[k] = k < a.hLum.pairs ? a.hLum.packed[(size_t)gx2 * a.hLum.pairs + k] : 0;
        cc[k] = k < a.hChr.pairs ? a.hChr.packed[(size_t)gc2 * a.hChr.pairs + k] : 0;
    }
    const int lpos = a.hLum.pos_even[gx2] - c0L;
    const int cpos = a.hChr.pos_even[gc2] - c0C;
    constexpr int QW = TW / 4;
    const int q = tid % QW;
    int yl = tid / QW;
    int vl[kYMaxPairs], vc[kYMaxPairs], vpL = 0, vpC = 0, lr = 0, cr = 0;
    auto load_row = [&](int yo) {
#pragma unroll
        for (int k = 0; k < kYMaxPairs; k++) {
            vl[k] = k < a.vLum.pairs ? a.vLum.packed[(size_t)yo * a.vLum.pairs + k] : 0;
            vc[k] = (!YUVOUT && k < a.vChr.pairs) ? a.vChr.packed[(size_t)yo * a.vChr.pairs + k] : 0;
        }
        vpL = (a.vLum.pos_even[yo] - r0L) >> 1;
        lr = a.vLum.round[yo];
        if (!YUVOUT) {                                   // MODE 2 indexes vChr by chroma row (phase 3)
            vpC = (a.vChr.pos_even[yo] - r0C) >> 1;
            cr = a.vChr.round[yo];
        }
    };
    load_row(min(ty0 + yl, a.dstH - 1));

    // ================= phase 1 ================================================================
    {
        const bool fl = a.srcAligned && c0L + ncL <= a.srcW, fc = a.srcAligned && c0C + ncC <= a.chrSrcW;
        if constexpr (SRC16) {
            if (a.src16 == 3) yuv_phase1_rgb(a, tid, c0L, ncL, r0L, nrL, c0C, ncC, r0C, nrC, ly, lu, lv);
            else              yuv_phase1_src16(a, tid, c0L, ncL, r0L, nrL, c0C, ncC, r0C, nrC, ly, lu, lv);
        }
        else if (fl && fc && a.srcAligned16 && yuv_phase1_16(a, tid, c0L, ncL, r0L, nrL, c0C, ncC, r0C, nrC, ly, lu, lv)) {}
        else if (fl && fc) yuv_phase1<true, true>(a, tid, c0L, ncL, r0L, nrL, c0C, ncC, r0C, nrC, ly, lu, lv);
        else          yuv_phase1<false, false>(a, tid, c0L, ncL, r0L, nrL, c0C, ncC, r0C, nrC, ly, lu, lv);
    }
    GMAT_STAMP(1);
    __syncthreads();
    GMAT_STAMP(2);

    // ================= phase 2: horizontal filters ==============================================
    // 8-bit sources: hScale8To15_c, sum >> 7.  16-bit sources: hScale16To15_c, (start + sum) >> (depth - 1)
    const int HS = SRC16 ? a.hShift : 7, HB = SRC16 ? a.hBias : 0;
    {   // luma: item = (row pair, output column)
        const int xo = xo2;
        for (int rp = tid / TW; rp < (nrL >> 1); rp += 256 / TW) {
            const int *p0 = reinterpret_cast<const int *>(ly + (2 * rp) * a.colsL + lpos);
            const int *p1 = reinterpret_cast<const int *>(ly + (2 * rp + 1) * a.colsL + lpos);
            int s0 = HB, s1 = HB;
#pragma unroll
            for (int k = 0; k < kYMaxPairs; k++)
                if (k < a.hLum.pairs) { s0 = dot2(p0[k], lc[k], s0); s1 = dot2(p1[k], lc[k], s1); }
            if (LONG) for (int k = kYMaxPairs; k < a.hLum.pairs; k++) {         // filters longer than 16 taps (ratios beyond ~3.7:1)
                const int cf = a.hLum.packed[(size_t)gx2 * a.hLum.pairs + k];
                s0 = dot2(p0[k], cf, s0); s1 = dot2(p1[k], cf, s1);
            }
            int l0 = min(s0 >> HS, 32767), l1 = min(s1 >> HS, 32767);
            if (a.rangeConv == 1) {          // lumRangeToJpeg_c, swscale.c:176-181 (applied to the h-scaled line, hscale.c:60)
                l0 = (m24(min(l0, 30189), 19077) - 39057361) >> 14; l1 = (m24(min(l1, 30189), 19077) - 39057361) >> 14;
            } else if (a.rangeConv == 2) {   // lumRangeFromJpeg_c, :183-188
                l0 = (m24(l0, 14071) + 33561947) >> 14; l1 = (m24(l1, 14071) + 33561947) >> 14;
            }
            hy[rp * TW + xo] = (int)pk16(l0, l1);
        }
    }
    {   // chroma: item = (row pair, chroma output column), both planes
        const int xc = xc2;
        for (int rp = tid / CWD; rp < (nrC >> 1); rp += 256 / CWD) {
            const int *u0 = reinterpret_cast<const int *>(lu + (2 * rp) * a.colsC + cpos);
            const int *u1 = reinterpret_cast<const int *>(lu + (2 * rp + 1) * a.colsC + cpos);
            const int *v0 = reinterpret_cast<const int *>(lv + (2 * rp) * a.colsC + cpos);
            const int *v1 = reinterpret_cast<const int *>(lv + (2 * rp + 1) * a.colsC + cpos);
            int su0 = HB, su1 = HB, sv0 = HB, sv1 = HB;
#pragma unroll
            for (int k = 0; k < kYMaxPairs; k++)
                if (k < a.hChr.pairs) {
                    su0 = dot2(u0[k], cc[k], su0); su1 = dot2(u1[k], cc[k], su1);
                    sv0 = dot2(v0[k], cc[k], sv0); sv1 = dot2(v1[k], cc[k], sv1);
                }
            if (LONG) for (int k = kYMaxPairs; k < a.hChr.pairs; k++) {
                const int cf = a.hChr.packed[(size_t)gc2 * a.hChr.pairs + k];
                su0 = dot2(u0[k], cf, su0); su1 = dot2(u1[k], cf, su1);
                sv0 = dot2(v0[k], cf, sv0); sv1 = dot2(v1[k], cf, sv1);
            }
            int cu0 = min(su0 >> HS, 32767), cu1 = min(su1 >> HS, 32767), cv0 = min(sv0 >> HS, 32767), cv1 = min(sv1 >> HS, 32767);
            if (a.rangeConv == 1) {          // chrRangeToJpeg_c, swscale.c:157-164 (hscale.c:193)
                cu0 = (m24(min(cu0, 30775), 4663) - 9289992) >> 12; cu1 = (m24(min(cu1, 30775), 4663) - 9289992) >> 12;
                cv0 = (m24(min(cv0, 30775), 4663) - 9289992) >> 12; cv1 = (m24(min(cv1, 30775), 4663) - 9289992) >> 12;
            } else if (a.rangeConv == 2) {   // chrRangeFromJpeg_c, :166-173
                cu0 = (m24(cu0, 1799) + 4081085) >> 11; cu1 = (m24(cu1, 1799) + 4081085) >> 11;
                cv0 = (m24(cv0, 1799) + 4081085) >> 11; cv1 = (m24(cv1, 1799) + 4081085) >> 11;
            }
            hu[rp * CWD + xc] = (int)pk16(cu0, cu1);
            hv[rp * CWD + xc] = (int)pk16(cv0, cv1);
        }
    }
    GMAT_STAMP(3);
    __syncthreads();
    GMAT_STAMP(4);

    // ================= phase 3: vertical filters + colour stage + store =========================
    {
        const int bpp = (a.dstFormat == GMAT_PIX_FMT_RGBA || a.dstFormat == GMAT_PIX_FMT_BGRA) ? 4 : 3;
        const bool swap_rb = a.dstFormat == GMAT_PIX_FMT_BGR24 || a.dstFormat == GMAT_PIX_FMT_BGRA;
        for (; yl < a.TH; yl += 256 / QW) {
            const int yo = ty0 + yl, xo = tx0 + 4 * q;
            if (yo >= a.dstH || xo >= a.dstW) continue;
            if (yl >= 256 / QW) load_row(yo);            // later passes of tall tiles (TH > 256/QW)
            int Y[4] = {lr, lr, lr, lr};
#pragma unroll
            for (int k = 0; k < kYMaxPairs; k++) {
                if (k < a.vLum.pairs) {
                    const int4 v = *reinterpret_cast<const int4 *>(hy + (vpL + k) * TW + 4 * q);
                    Y[0] = dot2(v.x, vl[k], Y[0]); Y[1] = dot2(v.y, vl[k], Y[1]);
                    Y[2] = dot2(v.z, vl[k], Y[2]); Y[3] = dot2(v.w, vl[k], Y[3]);
                }
            }
            for (int k = kYMaxPairs; k < a.vLum.pairs; k++) {       // filters longer than 15 taps
                const int cf = a.vLum.packed[(size_t)yo * a.vLum.pairs + k];
                const int4 v = *reinterpret_cast<const int4 *>(hy + (vpL + k) * TW + 4 * q);
                Y[0] = dot2(v.x, cf, Y[0]); Y[1] = dot2(v.y, cf, Y[1]);
                Y[2] = dot2(v.z, cf, Y[2]); Y[3] = dot2(v.w, cf, Y[3]);
            }
            if (YUVOUT) {
                if (a.dst16) {
                    // yuv2p010lX_c / l1_c: clip_uintp2((1 << 16 + sum) >> 17, 10) << 6; lr holds the 1 << 16
                    unsigned short *d16 = reinterpret_cast<unsigned short *>(a.dst + (size_t)yo * a.ds) + xo;
                    unsigned w[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) w[i] = (unsigned)min(max(Y[i] >> 17, 0), 1023) << a.dstShift;   // P010: << 6, planar: as it is
                    const int nx = min(4, a.dstW - xo);
                    if (a.dstAligned && nx == 4) *reinterpret_cast<uint2 *>(d16) = make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16));
                    else for (int i = 0; i < nx; i++) d16[i] = (unsigned short)w[i];
                    continue;
                }
                // yuv2planeX_8_c: clip_u8((dither << 12 + sum) >> 19); lr holds the 64 << 12 of an 8-bit source
                if (a.dither8) {
#pragma unroll
                    for (int i = 0; i < 4; i++) Y[i] += dither_delta(xo + i, yo);
                }
                uint8_t *d = a.dst + (size_t)yo * a.ds + xo;
                const unsigned o = (unsigned)clip_u8_shr(Y[0], 19) | ((unsigned)clip_u8_shr(Y[1], 19) << 8) |
                                   ((unsigned)clip_u8_shr(Y[2], 19) << 16) | ((unsigned)clip_u8_shr(Y[3], 19) << 24);
                const int nx = min(4, a.dstW - xo);
                if (a.dstAligned && nx == 4) *reinterpret_cast<unsigned *>(d) = o;
                else for (int i = 0; i < nx; i++) d[i] = (uint8_t)(o >> (8 * i));
                continue;
            }
            unsigned px[4];
            if (FULL) {
                int U[4] = {cr, cr, cr, cr}, V[4] = {cr, cr, cr, cr};
                auto acc4 = [&](int k, int cf) {
                    const int4 u = *reinterpret_cast<const int4 *>(hu + (vpC + k) * CWD + 4 * q);
                    const int4 v = *reinterpret_cast<const int4 *>(hv + (vpC + k) * CWD + 4 * q);
                    U[0] = dot2(u.x, cf, U[0]); U[1] = dot2(u.y, cf, U[1]); U[2] = dot2(u.z, cf, U[2]); U[3] = dot2(u.w, cf, U[3]);
                    V[0] = dot2(v.x, cf, V[0]); V[1] = dot2(v.y, cf, V[1]); V[2] = dot2(v.z, cf, V[2]); V[3] = dot2(v.w, cf, V[3]);
                };
#pragma unroll
                for (int k = 0; k < kYMaxPairs; k++) if (k < a.vChr.pairs) acc4(k, vc[k]);
                for (int k = kYMaxPairs; k < a.vChr.pairs; k++) acc4(k, a.vChr.packed[(size_t)yo * a.vChr.pairs + k]);
#pragma unroll
                for (int i = 0; i < 4; i++) px[i] = yuv_to_rgb_full(a.y2r, Y[i] >> 10, U[i] >> 10, V[i] >> 10);
            } else {
                int U[2] = {cr, cr}, V[2] = {cr, cr};
                auto acc2 = [&](int k, int cf) {
                    const uint2 u = *reinterpret_cast<const uint2 *>(hu + (vpC + k) * CWD + 2 * q);
                    const uint2 v = *reinterpret_cast<const uint2 *>(hv + (vpC + k) * CWD + 2 * q);
                    U[0] = dot2((int)u.x, cf, U[0]); U[1] = dot2((int)u.y, cf, U[1]);
                    V[0] = dot2((int)v.x, cf, V[0]); V[1] = dot2((int)v.y, cf, V[1]);
                };
#pragma unroll
                for (int k = 0; k < kYMaxPairs; k++) if (k < a.vChr.pairs) acc2(k, vc[k]);
                for (int k = kYMaxPairs; k < a.vChr.pairs; k++) acc2(k, a.vChr.packed[(size_t)yo * a.vChr.pairs + k]);
                // table_rV/gU/gV/bU are indexed with av_clip_uint8 (yuv2rgb.c:737-760)
                const ChromaTerms t0 = chroma_terms(a.y2r, clip_u8_shr(U[0], 19), clip_u8_shr(V[0], 19));
                const ChromaTerms t1 = chroma_terms(a.y2r, clip_u8_shr(U[1], 19), clip_u8_shr(V[1], 19));
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int ya = m24(Y[i] >> 19, a.y2r.cy), yb = m24(Y[i + 2] >> 19, a.y2r.cy);
                    px[i]     = (unsigned)luma_chan(t0.r, ya) | ((unsigned)luma_chan(t0.g, ya) << 8) | ((unsigned)luma_chan(t0.b, ya) << 16);
                    px[i + 2] = (unsigned)luma_chan(t1.r, yb) | ((unsigned)luma_chan(t1.g, yb) << 8) | ((unsigned)luma_chan(t1.b, yb) << 16);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                unsigned c = px[i];
                if (swap_rb) c = ((c & 0xFF) << 16) | (c & 0xFF00) | ((c >> 16) & 0xFF);
                px[i] = c | 0xFF000000u;
            }
            uint8_t *d = a.dst + (size_t)yo * a.ds + (size_t)xo * bpp;
            const int nx = min(4, a.dstW - xo);
            if (a.dstAligned && nx == 4) {
                if (bpp == 4) {
                    *reinterpret_cast<uint4 *>(d) = make_uint4(px[0], px[1], px[2], px[3]);
                } else {
                    uint3 o3;
                    o3.x = (px[0] & 0xFFFFFF) | (px[1] << 24);
                    o3.y = ((px[1] >> 8) & 0xFFFF) | (px[2] << 16);
                    o3.z = ((px[2] >> 16) & 0xFF) | (px[3] << 8);
                    *reinterpret_cast<uint3 *>(d) = o3;
                }
            } else {
                for (int i = 0; i < nx; i++) {
                    d[i * bpp + 0] = (uint8_t)px[i];
                    d[i * bpp + 1] = (uint8_t)(px[i] >> 8);
                    d[i * bpp + 2] = (uint8_t)(px[i] >> 16);
                    if (bpp == 4) d[i * bpp + 3] = 255;
                }
            }
        }
    }
    if (YUVOUT) {
        // chroma rows of the tile: item = (chroma row, group of 4 chroma columns), both planes
        constexpr int QC = CWD / 4;
        const int tcy0 = ty0 >> CVS;
        for (int it = tid; it < (a.TH >> CVS) * QC; it += 256) {
            const int cyl = it / QC, qc = it - cyl * QC;
            const int cy = tcy0 + cyl, cx = tcx0 + 4 * qc;
            if (cy >= a.chrDstH || cx >= a.chrDstW) continue;
            const int vp = (a.vChr.pos_even[cy] - r0C) >> 1;
            const int rnd = a.vChr.round[cy];
            int U[4] = {rnd, rnd, rnd, rnd}, V[4] = {rnd, rnd, rnd, rnd};
            for (int k = 0; k < a.vChr.pairs; k++) {
                const int cf = a.vChr.packed[(size_t)cy * a.vChr.pairs + k];
                const int4 u = *reinterpret_cast<const int4 *>(hu + (vp + k) * CWD + 4 * qc);
                const int4 v = *reinterpret_cast<const int4 *>(hv + (vp + k) * CWD + 4 * qc);
                U[0] = dot2(u.x, cf, U[0]); U[1] = dot2(u.y, cf, U[1]); U[2] = dot2(u.z, cf, U[2]); U[3] = dot2(u.w, cf, U[3]);
                V[0] = dot2(v.x, cf, V[0]); V[1] = dot2(v.y, cf, V[1]); V[2] = dot2(v.z, cf, V[2]); V[3] = dot2(v.w, cf, V[3]);
            }
            const int nx = min(4, a.chrDstW - cx);
            if (a.dst16 == 2) {                                     // YUV420P10LE: yuv2planeX_10_c on each chroma plane
                unsigned short *du = reinterpret_cast<unsigned short *>(a.dstU + (size_t)cy * a.dsU) + cx;
                unsigned short *dv = reinterpret_cast<unsigned short *>(a.dstV + (size_t)cy * a.dsV) + cx;
                unsigned u[4], v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) { u[i] = (unsigned)min(max(U[i] >> 17, 0), 1023); v[i] = (unsigned)min(max(V[i] >> 17, 0), 1023); }
                if (a.dstAligned && nx == 4) {
                    *reinterpret_cast<uint2 *>(du) = make_uint2(u[0] | (u[1] << 16), u[2] | (u[3] << 16));
                    *reinterpret_cast<uint2 *>(dv) = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
                } else for (int i = 0; i < nx; i++) { du[i] = (unsigned short)u[i]; dv[i] = (unsigned short)v[i]; }
                continue;
            }
            if (a.dst16) {                                          // yuv2p010cX_c: 16-bit U, V interleaved
                unsigned w[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    w[i] = ((unsigned)min(max(U[i] >> 17, 0), 1023) << 6) | ((unsigned)min(max(V[i] >> 17, 0), 1023) << 22);
                unsigned *d32 = reinterpret_cast<unsigned *>(a.dstU + (size_t)cy * a.dsU) + cx;
                if (a.dstAligned && nx == 4) *reinterpret_cast<uint4 *>(d32) = make_uint4(w[0], w[1], w[2], w[3]);
                else for (int i = 0; i < nx; i++) {
                    unsigned short *d16 = reinterpret_cast<unsigned short *>(a.dstU + (size_t)cy * a.dsU) + 2 * (cx + i);
                    d16[0] = (unsigned short)w[i]; d16[1] = (unsigned short)(w[i] >> 16);
                }
                continue;
            }
            unsigned ub[4], vb[4];
            if (a.dither8) {                                        // chrDither8 = row chrDstY & 7; V three columns on (vscale.c:98,101, output.c:433-434)
#pragma unroll
                for (int i = 0; i < 4; i++) { U[i] += dither_delta(cx + i, cy); V[i] += dither_delta(cx + i + 3, cy); }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) { ub[i] = (unsigned)clip_u8_shr(U[i], 19); vb[i] = (unsigned)clip_u8_shr(V[i], 19); }
            if (a.dstNv12) {
                uint8_t *d = a.dstU + (size_t)cy * a.dsU + 2 * cx;
                if (a.dstAligned && nx == 4) {
                    *reinterpret_cast<uint2 *>(d) = make_uint2(ub[0] | (vb[0] << 8) | (ub[1] << 16) | (vb[1] << 24),
                                                               ub[2] | (vb[2] << 8) | (ub[3] << 16) | (vb[3] << 24));
                } else {
                    for (int i = 0; i < nx; i++) { d[2 * i] = (uint8_t)ub[i]; d[2 * i + 1] = (uint8_t)vb[i]; }
                }
            } else {
                uint8_t *du = a.dstU + (size_t)cy * a.dsU + cx, *dv = a.dstV + (size_t)cy * a.dsV + cx;
                if (a.dstAligned && nx == 4) {
                    *reinterpret_cast<unsigned *>(du) = ub[0] | (ub[1] << 8) | (ub[2] << 16) | (ub[3] << 24);
                    *reinterpret_cast<unsigned *>(dv) = vb[0] | (vb[1] << 8) | (vb[2] << 16) | (vb[3] << 24);
                } else {
                    for (int i = 0; i < nx; i++) { du[i] = (uint8_t)ub[i]; dv[i] = (uint8_t)vb[i]; }
                }
            }
        }
    }
    GMAT_STAMP(5);
#undef GMAT_STAMP
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int yenv(const char *name, int dflt)
{
    const char *v = ::gmat::knob(name);
    return v && *v ? atoi(v) : dflt;
}

static void windows(const FilterBank &fb, int tile, int ntiles, int count, int align,
                    std::vector<int32_t> &start, std::vector<int32_t> &cnt, int &mx)
{
    start.resize(ntiles); cnt.resize(ntiles);
    for (int t = 0; t < ntiles; t++) {
        int lo = INT32_MAX, hi = 0;
        for (int i = t * tile; i < std::min((t + 1) * tile, count); i++) {
            lo = std::min(lo, fb.pos_even[i]);
            hi = std::max(hi, fb.pos_even[i] + 2 * fb.pairs);
        }
        if (lo == INT32_MAX) { lo = 0; hi = align; }       // tile beyond this plane's extent
        lo &= ~(align - 1);
        start[t] = lo;
        cnt[t] = align_up(hi - lo, align);
        mx = std::max(mx, cnt[t]);
    }
}

int yuvscale_prepare(const ScalePlan &p, YuvScaleTiling &t)
{
    const bool out444 = p.dstFormat == GMAT_PIX_FMT_YUV444P;
    const bool out10 = is_dst10(p.dstFormat);                                // 4:2:0 with 16-bit stores (P010LE, YUV420P10LE)
    const int yuvOut = (is_yuv420(p.dstFormat) || out10) ? 1 : out444 ? 2 : 0;   // 1: 4:2:0   2: planar 4:4:4
    const bool rgbSrc = p.srcFormat == GMAT_PIX_FMT_RGB24 || p.srcFormat == GMAT_PIX_FMT_BGR24;      // YUV destinations, and the RGB ones the RGB scaler has no writer for
    const bool pl16 = pl16_depth(p.srcFormat) != 0;
    if (!(is_yuv8_src(p.srcFormat) || is_p01x(p.srcFormat) || pl16 || rgbSrc) || !(is_packed_rgb(p.dstFormat) || yuvOut)) return GMAT_ERR(ENOSYS);
    if (is_p01x(p.srcFormat) || pl16) {
        // the P016 image is biased by -32768, undone by a start value that assumes every horizontal row sums to
        // 16384 (initFilter normalises exactly, utils.c:721-741)
        for (const FilterBank *fb : {&p.hLum, &p.hChr})
            for (int x = 0; x < fb->count; x++) {
                int sum = 0;
                for (int j = 0; j < fb->taps; j++) sum += fb->coef[(size_t)x * fb->taps + j];
                if (sum != 16384) return GMAT_ERR(ENOSYS);
            }
    }
    if (p.hLum.pairs > 64 || p.hChr.pairs > 64) return GMAT_ERR(ENOSYS);      // 128 taps: ratios up to ~30:1 (bicubic)
    const int full = ((p.flags & GMAT_SWS_FULL_CHR_H_INT) || out444) ? 1 : 0;  // chroma tile as wide as the luma tile
    if (full ? p.chrDstW != p.dstW : p.chrDstW != (p.dstW + 1) / 2) return GMAT_ERR(ENOSYS);
    if (p.chrDstH != (yuvOut == 1 ? (p.dstH + 1) / 2 : p.dstH)) return GMAT_ERR(ENOSYS);
    t.fullChroma = full;
    t.yuvOut = yuvOut;

    // ---- vertical special forms (vscale.c:135-167) -> per-row start values + effective chroma taps
    const int sh_one = full ? (1 << 9) : (1 << 18);
    const int chr_bias = full ? -(128 << 19) : 0;
    // planar 8-bit output: dither 64 (swscale.c:349-351), >> 19; P010: 1 << 16, >> 17 (output.c:481-519)
    const int planar_one = out10 ? (1 << 16) : (64 << 12);
    t.lumRound.assign(p.dstH, yuvOut ? planar_one : sh_one);
    t.chrRound.assign(p.chrDstH, yuvOut ? planar_one : sh_one + chr_bias);
    t.vChrEff = p.vChr;
    t.vLumEff = p.vLum;
    const int lfs = p.vLum.taps, cfs = p.vChr.taps;
    for (int y = 0; y < p.dstH && !yuvOut; y++) {
        int16_t *lf = &t.vLumEff.coef[(size_t)y * lfs];
        int16_t *cf = &t.vChrEff.coef[(size_t)y * cfs];
        const bool chr2 = cfs == 2 && cf[0] + cf[1] == 4096 && (unsigned)cf[1] <= 4096u;
        const bool lum2 = lfs == 2 && lf[0] + lf[1] == 4096 && (unsigned)lf[1] <= 4096u;
        if (lfs == 1 && cfs == 1) {                  // yuv2packed1, uvalpha 0: the samples themselves — the
            lf[0] = 4096; cf[0] = 4096;              // coefficients are not read (they can be 0 for degenerate rows)
        } else if (lfs == 1 && chr2) {               // yuv2packed1 with uvalpha = cf[1]
            lf[0] = 4096;
            if (cf[1] < 2048) { cf[0] = 4096; cf[1] = 0; }
            else              { cf[0] = 2048; cf[1] = 2048; }
        } else if (lum2 && chr2) {                   // yuv2packed2: no rounding constant
            t.lumRound[y] = 0;
            t.chrRound[y] = chr_bias;
        }
    }
    if (yuvOut) {
        // planar output (vscale.c:30-105): a 1-tap filter goes through yuv2plane1_8_c, which does not read the
        // coefficient; NV12 chroma always takes yuv2nv12cX_c, which does
        if (lfs == 1) std::fill(t.vLumEff.coef.begin(), t.vLumEff.coef.end(), (int16_t)4096);
        if (cfs == 1 && p.dstFormat != GMAT_PIX_FMT_NV12 && p.dstFormat != GMAT_PIX_FMT_P010LE) std::fill(t.vChrEff.coef.begin(), t.vChrEff.coef.end(), (int16_t)4096);
    }
    pack_filter_pairs(t.vChrEff);
    pack_filter_pairs(t.vLumEff);

    const int forceTW = yenv("GMAT_SCALE_TW", 0), forceTH = yenv("GMAT_SCALE_TH", 0);
    const int tws[] = {64, 32};
    // first the tilings that leave room for several blocks per CU; large down-scale ratios (wide windows) may use
    // the whole 64 KB a workgroup can address
    const int caps[] = {yenv("GMAT_SCALE_LDS_CAP", 40 * 1024), 64 * 1024};
    for (int ldsCap : caps)
    for (int TW : tws) {
        if (forceTW && TW != forceTW) continue;
        const int cwd = full ? TW : TW / 2;
        const int ntx = (p.dstW + TW - 1) / TW;
        int colsL = 0, colsC = 0;
        // window starts and lengths in whole 16-byte chunks (16 luma samples / 8 samples of each chroma plane)
        windows(p.hLum, TW, ntx, p.dstW, 16, t.colStartL, t.colCountL, colsL);
        windows(p.hChr, cwd, ntx, p.chrDstW, 8, t.colStartC, t.colCountC, colsC);
        colsL = align_up(colsL, 16); colsC = align_up(colsC, 8);
        const int ths[] = {32, 16, 8, 4, 2, 1};
        for (int TH : ths) {
            if (forceTH && TH != forceTH) continue;
            if (!forceTH && TH > 16) continue;
            if (yuvOut == 1 && (TH & 1)) continue;         // a tile holds TH/2 chroma rows
            const int nty = (p.dstH + TH - 1) / TH;
            int rowsL = 0, rowsC = 0;
            windows(t.vLumEff, TH, nty, p.dstH, 2, t.rowStartL, t.rowCountL, rowsL);
            if (yuvOut == 1) windows(t.vChrEff, TH / 2, nty, p.chrDstH, 2, t.rowStartC, t.rowCountC, rowsC);
            else        windows(t.vChrEff, TH, nty, p.dstH, 2, t.rowStartC, t.rowCountC, rowsC);
            const int bytes = rowsL * colsL * 2 + 2 * rowsC * colsC * 2 + (rowsL / 2) * TW * 4 + 2 * (rowsC / 2) * cwd * 4;
            if (bytes > ldsCap && !(forceTH && bytes <= 64 * 1024)) continue;
            t.TW = TW; t.TH = TH; t.ntx = ntx; t.nty = nty;
            t.rowsL = rowsL; t.colsL = colsL; t.rowsC = rowsC; t.colsC = colsC; t.ldsBytes = bytes;
            t.xcdRemap = yenv("GMAT_SCALE_XCD", 1);
            return 0;
        }
    }
    // no tile's window fits a workgroup's LDS (a one-row tile beyond ~ 20 : 1): TW = 0 — the context lives if the lines form (k_scale_yuvl.hip)
    // serves it (init_yuv_scaler), and the tiled kernel's place in the table answers ENOSYS
    t.TW = 0; t.TH = 0; t.ntx = 0; t.nty = 0; t.ldsBytes = 0;
    return 0;
}

const char *yuvscale_kernel_name(const YuvScaleTiling &t)
{
    if (t.yuvOut == 2) return t.TW == 64 ? "scale_yuv_kernel<64,yuv444>" : "scale_yuv_kernel<32,yuv444>";
    if (t.yuvOut) return t.TW == 64 ? "scale_yuv_kernel<64,yuv>" : "scale_yuv_kernel<32,yuv>";
    if (t.TW == 64) return t.fullChroma ? "scale_yuv_kernel<64,full>" : "scale_yuv_kernel<64,half>";
    return t.fullChroma ? "scale_yuv_kernel<32,full>" : "scale_yuv_kernel<32,half>";
}

int launch_scale_yuv(const YuvScaleArgs &a, const YuvScaleTiling &t, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    const int ntiles = t.ntx * t.nty;
    if (ntiles <= 0) return 0;
    Yuv2xFrames one;
    if (!frames) {
        one.y[0] = a.y; one.u[0] = a.u; one.v[0] = a.v; one.dst[0] = a.dst; one.dstU[0] = a.dstU; one.dstV[0] = a.dstV;
        frames = &one; nframes = 1;
    }
    if (nframes < 1 || nframes > kYuv2xMaxFrames) return GMAT_ERR(EINVAL);
    const Yuv2xFrames &fr = *frames;
    const dim3 grid(t.xcdRemap ? 8 * ((ntiles + 7) / 8) : ntiles, nframes), block(256);
    const size_t lds = (size_t)t.ldsBytes;
    const bool longH = a.hLum.pairs > kYMaxPairs || a.hChr.pairs > kYMaxPairs;
#define GMAT_LAUNCH_YUV(TW_, MODE_) \
    do { if (a.src16 && longH) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv_kernel<TW_, MODE_, true, true>), grid, block, lds, stream, a, fr); \
         else if (a.src16)  hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv_kernel<TW_, MODE_, false, true>), grid, block, lds, stream, a, fr); \
         else if (longH) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv_kernel<TW_, MODE_, true, false>), grid, block, lds, stream, a, fr); \
         else       hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv_kernel<TW_, MODE_, false, false>), grid, block, lds, stream, a, fr); } while (0)
    const int mode = t.yuvOut == 2 ? 3 : t.yuvOut ? 2 : t.fullChroma ? 1 : 0;
    if (t.TW == 64) { if (mode == 3) GMAT_LAUNCH_YUV(64, 3); else if (mode == 2) GMAT_LAUNCH_YUV(64, 2); else if (mode == 1) GMAT_LAUNCH_YUV(64, 1); else GMAT_LAUNCH_YUV(64, 0); }
    else if (t.TW == 32) { if (mode == 3) GMAT_LAUNCH_YUV(32, 3); else if (mode == 2) GMAT_LAUNCH_YUV(32, 2); else if (mode == 1) GMAT_LAUNCH_YUV(32, 1); else GMAT_LAUNCH_YUV(32, 0); }
    else return GMAT_ERR(EINVAL);
#undef GMAT_LAUNCH_YUV
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
