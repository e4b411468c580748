// k_scale_rgb2s.hip — strip-walking form of the exact 2:1 packed-RGB -> packed-RGB scaler for gfx950: rgb24 4K -> 1080p
// bicubic, the second half of the metric's literal "nv12 -> rgb24 -> 1080p" chain (the reference's structure,
// libswscale/cuda/swscale_cuda.c:352-371).  One libswscale context's arithmetic, bit-exact (same as k_scale.hip):
//   input     rgb24ToY_c, rgb24ToUV_half_c on pixel pairs                      input.c:815-866
//   luma      hScale16To15_c: min(sum >> 13, 32767), 8 taps at 2:1             swscale.c:93-119
//   chroma    one tap of 16384 (half-width chroma plane -> full-width output): min(2u, 32767)
//   vertical  yuv2rgb_full_X_c + yuv2rgb_write_full on luma AND chroma (8 taps) output.c:2037-2082,1886-1935
// Same design as k_scale_yuv2s.hip: a wave owns 256 output columns and walks down its strip with the horizontally
// filtered luma rows and the chroma rows it still needs in registers (48 VGPRs of window), pixels come straight from
// global memory (three 4-byte aligned dwordx4 per row and lane: 16 pixels, of which 14 are the lane's window), borders
// are the interior filter on an edge-replicated frame (checked on the host), every coefficient is a kernel argument.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

namespace gmat {

constexpr int R2_STRIP = 256;

#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned r2_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ uint4 r2_ld16(const uint8_t *p) { const r2_u32x4 v = *reinterpret_cast<const r2_u32x4 *>(p); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ unsigned r2_ld4(const uint8_t *p) { return *reinterpret_cast<const unsigned *>(p); }
typedef unsigned r2_u32x2 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ uint2 r2_ld8(const uint8_t *p) { const r2_u32x2 v = *reinterpret_cast<const r2_u32x2 *>(p); return make_uint2(v.x, v.y); }
#else
static inline uint4 r2_ld16(const uint8_t *p) { uint4 v; std::memcpy(&v, p, 16); return v; }
static inline uint2 r2_ld8(const uint8_t *p) { uint2 v; std::memcpy(&v, p, 8); return v; }
static inline unsigned r2_ld4(const uint8_t *p) { unsigned v; std::memcpy(&v, p, 4); return v; }
#endif

__device__ __forceinline__ int r2_dot2(int ab, int cd, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, ab), __builtin_bit_cast(short2v, cd), acc, true);   // VOP3P form, see k_scale_yuv2s.hip
}

struct R2Row { unsigned d[12]; };              // 16 pixels = 48 bytes of one source row, from pixel 2 * xo - 4

// DST: 0 rgb24, 1 bgr24, 2 rgba, 3 bgra
template <int DST>
__global__ __launch_bounds__(256) void scale_rgb2s_kernel(Rgb2sArgs a, Yuv2xFrames fr)
{
    constexpr bool BGR = (DST & 1) != 0;
    constexpr int BPP = DST >= 2 ? 4 : 3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = a.nseg * a.nsg;
    int lin = blockIdx.x;
    if (a.xcdRemap) {
        const int chunk = (nblk + 7) >> 3;
        lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    }
    if (lin >= nblk) return;
    const int seg = __builtin_amdgcn_readfirstlane(lin / a.nsg);
    const int X0 = ((lin - seg * a.nsg) * 4 + wave) * R2_STRIP;
    if (X0 >= a.dstW) return;
    const int y0 = seg * a.segRows;
    const int nOut = min(a.segRows, a.dstH - y0);
    const int nIter = nOut + 3;
    const uint8_t *ps = fr.y[blockIdx.y];
    uint8_t *pd = fr.dst[blockIdx.y];

    const int xo = X0 + 4 * lane;
    const bool active = xo < a.dstW;
    const int xc = active ? xo : a.dstW - 4;
    const bool edgeWave = X0 == 0 || X0 + R2_STRIP >= a.dstW;
    const int want = 6 * xc - 12;                                // byte offset of pixel 2 * xc - 4
    const int off = min(max(want, 0), 3 * a.srcW - 48);
    const int sh = want - off;                                   // -12 left frame edge, +12 right frame edge (3 dwords)
    const unsigned uoff = (unsigned)off;
    const unsigned dstOff = (unsigned)xo * BPP;

    auto load_row = [&](int r, R2Row &R) {
        const unsigned ro = (unsigned)min(max(r, 0), a.srcH - 1) * (unsigned)a.ss + uoff;
        const uint4 v0 = r2_ld16(ps + ro), v1 = r2_ld16(ps + (unsigned)(ro + 16)), v2 = r2_ld16(ps + (unsigned)(ro + 32));
        R.d[0] = v0.x; R.d[1] = v0.y; R.d[2] = v0.z; R.d[3] = v0.w; R.d[4] = v1.x; R.d[5] = v1.y; R.d[6] = v1.z; R.d[7] = v1.w;
        R.d[8] = v2.x; R.d[9] = v2.y; R.d[10] = v2.z; R.d[11] = v2.w;
    };
    // frame-edge lanes: move the dwords into window position (4 pixels = 3 dwords) and replicate the edge pixel
    auto fix_row = [&](R2Row &R) {
        if (sh < 0) {
            const unsigned p = R.d[0];
#pragma unroll
            for (int k = 11; k >= 3; k--) R.d[k] = R.d[k - 3];
            R.d[0] = __builtin_amdgcn_perm(p, p, 0x00020100u); R.d[1] = __builtin_amdgcn_perm(p, p, 0x01000201u); R.d[2] = __builtin_amdgcn_perm(p, p, 0x02010002u);
        } else if (sh > 0) {
            const unsigned p = R.d[11];
#pragma unroll
            for (int k = 0; k < 9; k++) R.d[k] = R.d[k + 3];
            R.d[9] = __builtin_amdgcn_perm(p, p, 0x01030201u); R.d[10] = __builtin_amdgcn_perm(p, p, 0x02010302u); R.d[11] = __builtin_amdgcn_perm(p, p, 0x03020103u);
        }
    };

    // One source row: 14-bit Y of the 14 window pixels as 7 odd-aligned pairs -> 4 horizontal sums; 14-bit U / V of the
    // lane's 4 pixel pairs.  (first, second) channel pair and third channel as the bytes come: the coefficients arrive
    // in that order, so BGR24 sources need no code.
    auto convert_row = [&](const R2Row &R, int (&hs)[4], int (&u14)[4], int (&v14)[4]) {
        int y[16], fs[16], th[16];
#pragma unroll
        for (int i = 1; i <= 14; i++) {
            const int o = 3 * i, d = o >> 2, b = o & 3;
            const unsigned lo = R.d[d], hi = R.d[d + 1 < 12 ? d + 1 : d];
            fs[i] = (int)__builtin_amdgcn_perm(hi, lo, 0x0C000C00u | (unsigned)b | ((unsigned)(b + 1) << 16));
            th[i] = (int)__builtin_amdgcn_perm(hi, lo, 0x0C0C0C00u | (unsigned)(b + 2));
            // rgb24ToY_c: (ry*r + gy*g + by*b + (32 << 14) + (1 << 8)) >> 9
            y[i] = r2_dot2(fs[i], a.cY01, m24(th[i], a.cY2) + ((32 << 14) + (1 << 8))) >> 9;
        }
        int p[7];
#pragma unroll
        for (int k = 0; k < 7; k++) p[k] = (int)((unsigned)y[2 * k + 1] | ((unsigned)y[2 * k + 2] << 16));      // 0 <= y < 2^15
#pragma unroll
        for (int j = 0; j < 4; j++)
            hs[j] = r2_dot2(p[j + 3], a.hL[3], r2_dot2(p[j + 2], a.hL[2], r2_dot2(p[j + 1], a.hL[1], r2_dot2(p[j], a.hL[0], 0))));
#pragma unroll
        for (int c = 0; c < 4; c++) {
            // rgb24ToUV_half_c on the sum of pixels 2c, 2c + 1 of the lane: (ru*r + gu*g + bu*b + (256 << 15) + (1 << 9)) >> 10
            const int i0 = 4 + 2 * c;
            const int fsum = fs[i0] + fs[i0 + 1];                   // two 9-bit sums in the halves: no carry across
            const int tsum = th[i0] + th[i0 + 1];
            u14[c] = r2_dot2(fsum, a.cU01, m24(tsum, a.cU2) + ((256 << 15) + (1 << 9))) >> 10;
            v14[c] = r2_dot2(fsum, a.cV01, m24(tsum, a.cV2) + ((256 << 15) + (1 << 9))) >> 10;
        }
    };

    int hwY[4][4], hwU[4][4], hwV[4][4];                          // [slot][output]: (row 2m-1 | row 2m << 16), 15-bit lines
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) hwY[s][j] = hwU[s][j] = hwV[s][j] = 0;

    R2Row bufA[2], bufB[2];                                       // ping-pong: rows 2m-1 and 2m of the current / next pair
    load_row(2 * (y0 - 1) - 1, bufA[0]);
    load_row(2 * (y0 - 1), bufB[0]);

    auto body = [&](const int j, auto slot_c, auto edge_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr bool EDGE = decltype(edge_c)::value;
        R2Row ra = bufA[SLOT & 1], rb = bufB[SLOT & 1];
        if (j + 1 < nIter) {
            const int m = y0 + j;                                   // pair y0 - 1 + (j + 1)
            load_row(2 * m - 1, bufA[(SLOT + 1) & 1]);
            load_row(2 * m, bufB[(SLOT + 1) & 1]);
        }
        if (EDGE) { fix_row(ra); fix_row(rb); }
        {
            int sa[4], sb[4], ua[4], va[4], ub[4], vb[4];
            convert_row(ra, sa, ua, va);
            convert_row(rb, sb, ub, vb);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                // hScale16To15_c: min(val >> 13, 32767)
                hwY[SLOT][q] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(sa[q] >> 13, sb[q] >> 13));
                // the one-tap chroma filter: min((u * 16384) >> 13, 32767) = min(2u, 32767), u < 2^15
                hwU[SLOT][q] = (int)((unsigned)min(2 * ua[q], 32767) | ((unsigned)min(2 * ub[q], 32767) << 16));
                hwV[SLOT][q] = (int)((unsigned)min(2 * va[q], 32767) | ((unsigned)min(2 * vb[q], 32767) << 16));
            }
        }
        if (j >= 3) {
            const int yo = y0 + j - 3;
            unsigned c0[4], c1[4], c2[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int Y = a.rnd, U = a.rnd - (128 << 19), V = a.rnd - (128 << 19);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    Y = r2_dot2(hwY[(SLOT + 1 + k) & 3][q], a.vL[k], Y);
                    U = r2_dot2(hwU[(SLOT + 1 + k) & 3][q], a.vL[k], U);
                    V = r2_dot2(hwV[(SLOT + 1 + k) & 3][q], a.vL[k], V);
                }
                Y >>= 10; U >>= 10; V >>= 10;
                // yuv2rgb_write_full (output.c:1886-1935); every factor is below 2^23, so the 24-bit multiplier gives the
                // same low 32 bits as the C expression, and av_clip_uintp2(x, 30) is a median with the range ends
                const int yy = m24(Y - a.y2r.y_offset, a.y2r.y_coeff) + (1 << 21);
                const int R = yy + m24(V, a.y2r.v2r);
                const int G = yy + m24(V, a.y2r.v2g) + m24(U, a.y2r.u2g);
                const int B = yy + m24(U, a.y2r.u2b);
                const unsigned r8 = (unsigned)min(max(R, 0), 0x3FFFFFFF) >> 6, g8 = (unsigned)min(max(G, 0), 0x3FFFFFFF) >> 6,
                               b8 = (unsigned)min(max(B, 0), 0x3FFFFFFF) >> 6;      // the byte sits in bits 16..23
                c0[q] = BGR ? b8 : r8; c1[q] = g8; c2[q] = BGR ? r8 : b8;
            }
            if (active) {
                uint8_t *d = pd + (unsigned)((unsigned)yo * (unsigned)a.ds + dstOff);
    #define R2_B2PAIR(lo, hi) __builtin_amdgcn_perm((hi), (lo), 0x0C0C0602u)
                if (BPP == 4) {
                    uint4 o4;
                    o4.x = R2_B2PAIR(c0[0], c1[0]) | (R2_B2PAIR(c2[0], 0u) << 16) | 0xFF000000u;
                    o4.y = R2_B2PAIR(c0[1], c1[1]) | (R2_B2PAIR(c2[1], 0u) << 16) | 0xFF000000u;
                    o4.z = R2_B2PAIR(c0[2], c1[2]) | (R2_B2PAIR(c2[2], 0u) << 16) | 0xFF000000u;
                    o4.w = R2_B2PAIR(c0[3], c1[3]) | (R2_B2PAIR(c2[3], 0u) << 16) | 0xFF000000u;
                    *reinterpret_cast<uint4 *>(d) = o4;
                } else {
                    uint3 o3;
                    o3.x = R2_B2PAIR(c0[0], c1[0]) | (R2_B2PAIR(c2[0], c0[1]) << 16);
                    o3.y = R2_B2PAIR(c1[1], c2[1]) | (R2_B2PAIR(c0[2], c1[2]) << 16);
                    o3.z = R2_B2PAIR(c2[2], c0[3]) | (R2_B2PAIR(c1[3], c2[3]) << 16);
                    *reinterpret_cast<uint3 *>(d) = o3;
                }
    #undef R2_B2PAIR
            }
        }
    };
    auto run = [&](auto edge_c) {
        for (int j0 = 0; j0 < nIter; j0 += 4) {
            body(j0, std::integral_constant<int, 0>(), edge_c);
            if (j0 + 1 < nIter) body(j0 + 1, std::integral_constant<int, 1>(), edge_c);
            if (j0 + 2 < nIter) body(j0 + 2, std::integral_constant<int, 2>(), edge_c);
            if (j0 + 3 < nIter) body(j0 + 3, std::integral_constant<int, 3>(), edge_c);
        }
    };
    if (edgeWave) run(std::true_type()); else run(std::false_type());
}

// ---------------------------------------------------------------------------------------------
// scale_rgb2h_kernel: the same arithmetic with the converted samples SHARED between neighbouring lanes.
// In the kernel above a lane loads and converts its whole 14-pixel window (48 bytes, rgb24ToY_c on 14 pixels) although only 8
// pixels are its own: its neighbours convert the other 6 again.  Here a lane loads its own 8 pixels (24 bytes: 2 loads per row
// instead of 3, half the row buffers), converts them once, packs the 14-bit Y samples as the odd-aligned pairs (1,2) (3,4) (5,6)
// plus (0 | 7), and takes the three samples either side of its window from the lanes beside it with four DPP wave shifts.
// Lanes 0 and 63 only provide: a wave makes 62 x 4 = 248 output columns.  A lane's 8 pixels are entirely inside or entirely
// outside the frame (widths are multiples of 8), so edge replication is "a lane outside presents the edge pixel's sample".
// The chroma plane (pixel pairs, one tap) needs no neighbours.
constexpr int H2_OUT = 248;

struct H2Row { unsigned d[6]; };               // the lane's own 8 pixels of one source row: 24 bytes of packed RGB, or 8 Y + 8 chroma bytes

__device__ __forceinline__ int h2_from_left(int v)  { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true); }   // lane - 1 (wave_shr:1)
__device__ __forceinline__ int h2_from_right(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, true); }   // lane + 1 (wave_shl:1)

//
// SRC: 0 packed rgb24 / bgr24; 1 NV12, 2 YUV420P — the FUSED convert-then-scale form: sws(YUV -> RGB24 at the source size, the
// unscaled converter of yuv2rgb.c with its nearest chroma) followed by sws(RGB24 -> RGB at half the size), the reference GPU
// back-end's order of operations (swscale_cuda.c:352-371), without the 25 MB RGB24 intermediate of a 4K frame: a lane converts its
// own 8 pixels of a row with the first context's table arithmetic (px_math.h chroma_terms / luma_chan, the chroma terms from LDS
// tables as in k_scale_yuv2s.hip) and hands (r | g << 16), b to the second context's input stage.
template <int DST, int SRC>
__global__ __launch_bounds__(256) void scale_rgb2h_kernel(Rgb2sArgs a, Yuv2xFrames fr)
{
    constexpr bool BGR = (DST & 1) != 0;
    constexpr int BPP = DST >= 2 ? 4 : 3;
    __shared__ int2 lutV[SRC ? 256 : 1], lutU[SRC ? 256 : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (SRC != 0) {
        // term_R = lutV[V].x, term_G = lutV[V].y + lutU[U].x, term_B = lutU[U].y; channel = byte 2 of clamp(term + Y * cy, 0, 0xFFFFFF)
        const Yuv2RgbConsts &k = a.y2r;
        lutV[tid] = make_int2(k.base + m24(k.offR + (m24(tid, k.crv) >> 16), k.cy), m24(m24(tid, k.cgv) >> 16, k.cy));
        lutU[tid] = make_int2(k.base + m24(k.offG + (m24(tid, k.cgu) >> 16), k.cy), k.base + m24(k.offB + (m24(tid, k.cbu) >> 16), k.cy));
        __syncthreads();
    }
    const int nblk = a.nseg * a.nsg;
    int lin = blockIdx.x;
    if (a.xcdRemap) {
        const int chunk = (nblk + 7) >> 3;
        lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    }
    if (lin >= nblk) return;
    const int seg = __builtin_amdgcn_readfirstlane(lin / a.nsg);
    const int X0 = ((lin - seg * a.nsg) * 4 + wave) * H2_OUT;
    if (X0 >= a.dstW) return;
    const int y0 = seg * a.segRows;
    const int nOut = min(a.segRows, a.dstH - y0);
    const int nIter = nOut + 3;
    const uint8_t *ps = fr.y[blockIdx.y];
    uint8_t *pd = fr.dst[blockIdx.y];

    const int xo = X0 + 4 * (lane - 1);                          // the lane's 4 output columns; lane 0 / 63: its neighbour's halo
    const bool stores = lane >= 1 && lane <= 62 && xo < a.dstW;
    const bool outL = xo < 0, outR = xo >= a.dstW;               // own pixels 2 xo .. 2 xo + 7 lie outside the frame
    const bool edgeWave = X0 == 0 || X0 + H2_OUT + 4 > a.dstW;   // wave-uniform: the wave holds an outside lane
    const unsigned xl = (unsigned)min(max(xo, 0), a.dstW - 4);   // outside lanes load the frame's first / last 8 pixels
    const unsigned uoff = 6u * xl;
    const unsigned dstOff = (unsigned)xo * BPP;
    const uint8_t *pu = fr.u[blockIdx.y], *pv = fr.v[blockIdx.y];

    auto load_row = [&](int r, H2Row &R) {
        const int rc = min(max(r, 0), a.srcH - 1);
        if constexpr (SRC == 0) {
            const unsigned ro = (unsigned)rc * (unsigned)a.ss + uoff;
            const uint4 v0 = r2_ld16(ps + ro);
            const uint2 v1 = r2_ld8(ps + (unsigned)(ro + 16));
            R.d[0] = v0.x; R.d[1] = v0.y; R.d[2] = v0.z; R.d[3] = v0.w; R.d[4] = v1.x; R.d[5] = v1.y;
        } else {
            // 8 luma bytes of the row, the 4 chroma samples of row rc / 2 under them (the unscaled converter's nearest chroma)
            const uint2 yv = r2_ld8(ps + (unsigned)((unsigned)rc * (unsigned)a.ss + 2u * xl));
            R.d[0] = yv.x; R.d[1] = yv.y; R.d[4] = R.d[5] = 0u;
            if constexpr (SRC == 1) {
                const uint2 c = r2_ld8(pu + (unsigned)((unsigned)(rc >> 1) * (unsigned)a.us + 2u * xl));
                R.d[2] = c.x; R.d[3] = c.y;
            } else {
                R.d[2] = r2_ld4(pu + (unsigned)((unsigned)(rc >> 1) * (unsigned)a.us + xl));
                R.d[3] = r2_ld4(pv + (unsigned)((unsigned)(rc >> 1) * (unsigned)a.vs + xl));
            }
        }
    };

    // One source row: 14-bit Y of the lane's 8 pixels -> with the neighbours' samples 7 odd-aligned pairs -> 4 horizontal sums;
    // 14-bit U / V of the lane's 4 pixel pairs.
    auto convert_row = [&](const H2Row &R, auto edge_c, int (&hs)[4], int (&u14)[4], int (&v14)[4]) {
        int y[8], fs[8], th[8];
        if constexpr (SRC == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int o = 3 * i, d = o >> 2, b = o & 3;
                const unsigned lo = R.d[d], hi = R.d[d + 1 < 6 ? d + 1 : d];
                fs[i] = (int)__builtin_amdgcn_perm(hi, lo, 0x0C000C00u | (unsigned)b | ((unsigned)(b + 1) << 16));
                th[i] = (int)__builtin_amdgcn_perm(hi, lo, 0x0C0C0C00u | (unsigned)(b + 2));
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) {                           // chroma sample c covers the lane's pixels 2c, 2c + 1
                unsigned U, V;
                if constexpr (SRC == 1) { const unsigned w = R.d[2 + (c >> 1)]; U = (w >> (16 * (c & 1))) & 0xFFu; V = (w >> (16 * (c & 1) + 8)) & 0xFFu; }
                else                    { U = (R.d[2] >> (8 * c)) & 0xFFu; V = (R.d[3] >> (8 * c)) & 0xFFu; }
                const int2 tv = lutV[V], tu = lutU[U];
                const int tr = tv.x, tg = tv.y + tu.x, tb = tu.y;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int i = 2 * c + h;
                    const int ycy = m24((int)((R.d[i >> 2] >> (8 * (i & 3))) & 0xFFu), a.y2r.cy);
                    const int r8 = luma_chan(tr, ycy), g8 = luma_chan(tg, ycy), b8 = luma_chan(tb, ycy);
                    fs[i] = r8 | (g8 << 16);                        // the intermediate is RGB24: (r, g) and b, as the bytes would come
                    th[i] = b8;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++)   // rgb24ToY_c: (ry*r + gy*g + by*b + (32 << 14) + (1 << 8)) >> 9
            y[i] = r2_dot2(fs[i], a.cY01, m24(th[i], a.cY2) + ((32 << 14) + (1 << 8))) >> 9;
        if constexpr (decltype(edge_c)::value) {                  // a lane outside the frame: every sample is the edge pixel's
            const int ye = outL ? y[0] : y[7];
#pragma unroll
            for (int i = 0; i < 8; i++) y[i] = (outL || outR) ? ye : y[i];
        }
        const int A0 = (int)((unsigned)y[1] | ((unsigned)y[2] << 16)), A1 = (int)((unsigned)y[3] | ((unsigned)y[4] << 16)),
                  A2 = (int)((unsigned)y[5] | ((unsigned)y[6] << 16)), B = (int)((unsigned)y[0] | ((unsigned)y[7] << 16));     // 0 <= y < 2^15
        const int lA2 = h2_from_left(A2), lB = h2_from_left(B), rA0 = h2_from_right(A0), rB = h2_from_right(B);
        int p[7];
        p[0] = lA2;                                                                      // the left lane's samples 5, 6
        p[1] = (int)__builtin_amdgcn_perm((unsigned)B, (unsigned)lB, 0x05040302u);        // its sample 7 | own sample 0
        p[2] = A0; p[3] = A1; p[4] = A2;
        p[5] = (int)__builtin_amdgcn_perm((unsigned)rB, (unsigned)B, 0x05040302u);        // own sample 7 | the right lane's sample 0
        p[6] = rA0;                                                                      // the right lane's samples 1, 2
#pragma unroll
        for (int j = 0; j < 4; j++)
            hs[j] = r2_dot2(p[j + 3], a.hL[3], r2_dot2(p[j + 2], a.hL[2], r2_dot2(p[j + 1], a.hL[1], r2_dot2(p[j], a.hL[0], 0))));
#pragma unroll
        for (int c = 0; c < 4; c++) {
            // rgb24ToUV_half_c on the sum of pixels 2c, 2c + 1 of the lane: (ru*r + gu*g + bu*b + (256 << 15) + (1 << 9)) >> 10
            const int fsum = fs[2 * c] + fs[2 * c + 1];             // two 9-bit sums in the halves: no carry across
            const int tsum = th[2 * c] + th[2 * c + 1];
            u14[c] = r2_dot2(fsum, a.cU01, m24(tsum, a.cU2) + ((256 << 15) + (1 << 9))) >> 10;
            v14[c] = r2_dot2(fsum, a.cV01, m24(tsum, a.cV2) + ((256 << 15) + (1 << 9))) >> 10;
        }
    };

    int hwY[4][4], hwU[4][4], hwV[4][4];                          // [slot][output]: (row 2m-1 | row 2m << 16), 15-bit lines
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) hwY[s][j] = hwU[s][j] = hwV[s][j] = 0;

    H2Row bufA[2], bufB[2];                                       // ping-pong: rows 2m-1 and 2m of the current / next pair
    load_row(2 * (y0 - 1) - 1, bufA[0]);
    load_row(2 * (y0 - 1), bufB[0]);

    auto body = [&](const int j, auto slot_c, auto edge_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        const H2Row ra = bufA[SLOT & 1], rb = bufB[SLOT & 1];
        if (j + 1 < nIter) {
            const int m = y0 + j;                                   // pair y0 - 1 + (j + 1)
            load_row(2 * m - 1, bufA[(SLOT + 1) & 1]);
            load_row(2 * m, bufB[(SLOT + 1) & 1]);
        }
        {
            int sa[4], sb[4], ua[4], va[4], ub[4], vb[4];
            convert_row(ra, edge_c, sa, ua, va);
            convert_row(rb, edge_c, sb, ub, vb);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                // hScale16To15_c: min(val >> 13, 32767)
                hwY[SLOT][q] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(sa[q] >> 13, sb[q] >> 13));
                // the one-tap chroma filter: min((u * 16384) >> 13, 32767) = min(2u, 32767), u < 2^15
                hwU[SLOT][q] = (int)((unsigned)min(2 * ua[q], 32767) | ((unsigned)min(2 * ub[q], 32767) << 16));
                hwV[SLOT][q] = (int)((unsigned)min(2 * va[q], 32767) | ((unsigned)min(2 * vb[q], 32767) << 16));
            }
        }
        if (j >= 3) {
            const int yo = y0 + j - 3;
            unsigned c0[4], c1[4], c2[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int Y = a.rnd, U = a.rnd - (128 << 19), V = a.rnd - (128 << 19);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    Y = r2_dot2(hwY[(SLOT + 1 + k) & 3][q], a.vL[k], Y);
                    U = r2_dot2(hwU[(SLOT + 1 + k) & 3][q], a.vL[k], U);
                    V = r2_dot2(hwV[(SLOT + 1 + k) & 3][q], a.vL[k], V);
                }
                Y >>= 10; U >>= 10; V >>= 10;
                // yuv2rgb_write_full (output.c:1886-1935), as in the kernel above
                const int yy = m24(Y - a.y2r.y_offset, a.y2r.y_coeff) + (1 << 21);
                const int R = yy + m24(V, a.y2r.v2r);
                const int G = yy + m24(V, a.y2r.v2g) + m24(U, a.y2r.u2g);
                const int Bc = yy + m24(U, a.y2r.u2b);
                const unsigned r8 = (unsigned)min(max(R, 0), 0x3FFFFFFF) >> 6, g8 = (unsigned)min(max(G, 0), 0x3FFFFFFF) >> 6,
                               b8 = (unsigned)min(max(Bc, 0), 0x3FFFFFFF) >> 6;     // the byte sits in bits 16..23
                c0[q] = BGR ? b8 : r8; c1[q] = g8; c2[q] = BGR ? r8 : b8;
            }
            if (stores) {
                uint8_t *d = pd + (unsigned)((unsigned)yo * (unsigned)a.ds + dstOff);
    #define R2_B2PAIR(lo, hi) __builtin_amdgcn_perm((hi), (lo), 0x0C0C0602u)
                if (BPP == 4) {
                    uint4 o4;
                    o4.x = R2_B2PAIR(c0[0], c1[0]) | (R2_B2PAIR(c2[0], 0u) << 16) | 0xFF000000u;
                    o4.y = R2_B2PAIR(c0[1], c1[1]) | (R2_B2PAIR(c2[1], 0u) << 16) | 0xFF000000u;
                    o4.z = R2_B2PAIR(c0[2], c1[2]) | (R2_B2PAIR(c2[2], 0u) << 16) | 0xFF000000u;
                    o4.w = R2_B2PAIR(c0[3], c1[3]) | (R2_B2PAIR(c2[3], 0u) << 16) | 0xFF000000u;
                    *reinterpret_cast<uint4 *>(d) = o4;
                } else {
                    uint3 o3;
                    o3.x = R2_B2PAIR(c0[0], c1[0]) | (R2_B2PAIR(c2[0], c0[1]) << 16);
                    o3.y = R2_B2PAIR(c1[1], c2[1]) | (R2_B2PAIR(c0[2], c1[2]) << 16);
                    o3.z = R2_B2PAIR(c2[2], c0[3]) | (R2_B2PAIR(c1[3], c2[3]) << 16);
                    *reinterpret_cast<uint3 *>(d) = o3;
                }
    #undef R2_B2PAIR
            }
        }
    };
    auto run = [&](auto edge_c) {
        for (int j0 = 0; j0 < nIter; j0 += 4) {
            body(j0, std::integral_constant<int, 0>(), edge_c);
            if (j0 + 1 < nIter) body(j0 + 1, std::integral_constant<int, 1>(), edge_c);
            if (j0 + 2 < nIter) body(j0 + 2, std::integral_constant<int, 2>(), edge_c);
            if (j0 + 3 < nIter) body(j0 + 3, std::integral_constant<int, 3>(), edge_c);
        }
    };
    if (edgeWave) run(std::true_type()); else run(std::false_type());
}

// ---------------------------------------------------------------------------------------------
// scale_rgb2y_kernel: packed RGB24 / BGR24 -> NV12 / YUV420P at exactly half the size, ONE libswscale context (rgb24ToY_c and
// rgb24ToUV_half_c, hScale16To15_c with sh = 13 on both planes, yuv2planeX_8_c / yuv2nv12cX_c).  The front end is scale_rgb2h_kernel's:
// a lane converts its own 8 pixels of a row, lanes 0 and 63 only provide halo samples, DPP shifts hand the neighbours' samples over.
// Chroma here is decimated on both axes: the pixel-pair samples (4 per lane) go through the 8-tap 2:1 filter to 2 outputs per lane,
// and the 4:1 vertical filter has 16 taps on rows [4c - 6, 4c + 9].  Rows arrive as pairs (2m - 1, 2m) — the luma window's
// alignment — so a chroma row collects 9 row pairs m = 2c - 3 .. 2c + 5 (the first and the last half used) in running sums: five
// chroma rows are open at any time, each pair feeds 5 / 4 of them with one v_dot2 per sample, and no 16-row window is kept.
// ---------------------------------------------------------------------------------------------
template <bool NV>
__global__ __launch_bounds__(256) void scale_rgb2y_kernel(Rgb2yArgs a, Yuv2xFrames fr)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lin = blockIdx.x;
    if (a.xcdRemap) {
        const int chunk = (a.nblk + 7) >> 3;
        lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    }
    if (lin >= a.nblk) return;
    const int unit = lin * 4 + wave;                             // (segment, strip) units packed densely: the waves share nothing
    if (unit >= a.nseg * a.nstrips) return;
    const int seg = __builtin_amdgcn_readfirstlane(unit / a.nstrips);
    const int X0 = (unit - seg * a.nstrips) * H2_OUT;
    const int c0 = seg * a.segRowsC, chrH = a.dstH >> 1;
    const int nC = min(a.segRowsC, chrH - c0);
    const int nIter = 2 * nC + 7;                                // row pairs m0 .. m0 + 2 nC + 6
    const int m0 = 2 * c0 - 3;
    const uint8_t *ps = fr.y[blockIdx.y];
    uint8_t *py = fr.dst[blockIdx.y], *pu = fr.dstU[blockIdx.y], *pv = fr.dstV[blockIdx.y];

    const int xo = X0 + 4 * (lane - 1);                          // the lane's 4 luma / 2 chroma output columns; lane 0 / 63: halo only
    const bool stores = lane >= 1 && lane <= 62 && xo < a.dstW;
    const bool outL = xo < 0, outR = xo >= a.dstW;               // own pixels 2 xo .. 2 xo + 7 lie outside the frame
    const bool edgeWave = X0 == 0 || X0 + H2_OUT + 4 > a.dstW;   // wave-uniform: the wave holds an outside lane
    const unsigned xl = (unsigned)min(max(xo, 0), a.dstW - 4);   // outside lanes load the frame's first / last 8 pixels
    const unsigned uoff = 6u * xl;

    auto load_row = [&](int r, H2Row &R) {
        const int rc = min(max(r, 0), a.srcH - 1);
        const unsigned ro = (unsigned)rc * (unsigned)a.ss + uoff;
        const uint4 v0 = r2_ld16(ps + ro);
        const uint2 v1 = r2_ld8(ps + (unsigned)(ro + 16));
        R.d[0] = v0.x; R.d[1] = v0.y; R.d[2] = v0.z; R.d[3] = v0.w; R.d[4] = v1.x; R.d[5] = v1.y;
    };

    // One source row: 14-bit Y of the lane's 8 pixels and 14-bit U / V of its 4 pixel pairs -> with the neighbours' samples the
    // odd-aligned pairs of the 8-tap windows -> 4 luma and 2 + 2 chroma horizontal sums
    auto convert_row = [&](const H2Row &R, auto edge_c, int (&hs)[4], int (&hu)[2], int (&hv)[2]) {
        int y[8], u[4], v[4], fs[8], th[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int o = 3 * i, d = o >> 2, b = o & 3;
            const unsigned lo = R.d[d], hi = R.d[d + 1 < 6 ? d + 1 : d];
            fs[i] = (int)__builtin_amdgcn_perm(hi, lo, 0x0C000C00u | (unsigned)b | ((unsigned)(b + 1) << 16));
            th[i] = (int)__builtin_amdgcn_perm(hi, lo, 0x0C0C0C00u | (unsigned)(b + 2));
        }
#pragma unroll
        for (int i = 0; i < 8; i++)   // rgb24ToY_c: (ry*r + gy*g + by*b + (32 << 14) + (1 << 8)) >> 9
            y[i] = r2_dot2(fs[i], a.cY01, m24(th[i], a.cY2) + ((32 << 14) + (1 << 8))) >> 9;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            // rgb24ToUV_half_c on the sum of pixels 2c, 2c + 1 of the lane: (ru*r + gu*g + bu*b + (256 << 15) + (1 << 9)) >> 10
            const int fsum = fs[2 * c] + fs[2 * c + 1];             // two 9-bit sums in the halves: no carry across
            const int tsum = th[2 * c] + th[2 * c + 1];
            u[c] = r2_dot2(fsum, a.cU01, m24(tsum, a.cU2) + ((256 << 15) + (1 << 9))) >> 10;
            v[c] = r2_dot2(fsum, a.cV01, m24(tsum, a.cV2) + ((256 << 15) + (1 << 9))) >> 10;
        }
        if constexpr (decltype(edge_c)::value) {                  // a lane outside the frame: every sample is the edge sample
            const int ye = outL ? y[0] : y[7], ue = outL ? u[0] : u[3], ve = outL ? v[0] : v[3];
#pragma unroll
            for (int i = 0; i < 8; i++) y[i] = (outL || outR) ? ye : y[i];
#pragma unroll
            for (int i = 0; i < 4; i++) { u[i] = (outL || outR) ? ue : u[i]; v[i] = (outL || outR) ? ve : v[i]; }
        }
        {
            const int A0 = (int)((unsigned)y[1] | ((unsigned)y[2] << 16)), A1 = (int)((unsigned)y[3] | ((unsigned)y[4] << 16)),
                      A2 = (int)((unsigned)y[5] | ((unsigned)y[6] << 16)), B = (int)((unsigned)y[0] | ((unsigned)y[7] << 16));     // 0 <= y < 2^15
            const int lA2 = h2_from_left(A2), lB = h2_from_left(B), rA0 = h2_from_right(A0), rB = h2_from_right(B);
            int p[7];
            p[0] = lA2;                                                                      // the left lane's samples 5, 6
            p[1] = (int)__builtin_amdgcn_perm((unsigned)B, (unsigned)lB, 0x05040302u);        // its sample 7 | own sample 0
            p[2] = A0; p[3] = A1; p[4] = A2;
            p[5] = (int)__builtin_amdgcn_perm((unsigned)rB, (unsigned)B, 0x05040302u);        // own sample 7 | the right lane's sample 0
            p[6] = rA0;                                                                      // the right lane's samples 1, 2
#pragma unroll
            for (int j = 0; j < 4; j++)
                hs[j] = r2_dot2(p[j + 3], a.hL[3], r2_dot2(p[j + 2], a.hL[2], r2_dot2(p[j + 1], a.hL[1], r2_dot2(p[j], a.hL[0], 0))));
        }
        auto chroma_h = [&](const int (&s)[4], int (&h)[2]) {
            // chroma output c of the lane: pair samples 2c - 3 .. 2c + 4
            const int A = (int)((unsigned)s[1] | ((unsigned)s[2] << 16)), B = (int)((unsigned)s[0] | ((unsigned)s[3] << 16));
            const int lA = h2_from_left(A), lB = h2_from_left(B), rA = h2_from_right(A), rB = h2_from_right(B);
            int p[5];
            p[0] = lA;                                                                       // the left lane's samples 1, 2
            p[1] = (int)__builtin_amdgcn_perm((unsigned)B, (unsigned)lB, 0x05040302u);        // its sample 3 | own sample 0
            p[2] = A;
            p[3] = (int)__builtin_amdgcn_perm((unsigned)rB, (unsigned)B, 0x05040302u);        // own sample 3 | the right lane's sample 0
            p[4] = rA;                                                                       // the right lane's samples 1, 2
#pragma unroll
            for (int j = 0; j < 2; j++)
                h[j] = r2_dot2(p[j + 3], a.hC[3], r2_dot2(p[j + 2], a.hC[2], r2_dot2(p[j + 1], a.hC[1], r2_dot2(p[j], a.hC[0], 0))));
        };
        chroma_h(u, hu);
        chroma_h(v, hv);
    };

    int hwY[4][4];                                                // [slot][output]: (row 2m-1 | row 2m << 16), 15-bit lines
    int acc[5][4];                                                // open chroma rows x (U0, U1, V0, V1)
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) hwY[s][j] = 0;
#pragma unroll
    for (int s = 0; s < 5; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[s][j] = 0;

    H2Row bufA[2], bufB[2];                                       // ping-pong: rows 2m-1 and 2m of the current / next pair
    load_row(2 * m0 - 1, bufA[0]);
    load_row(2 * m0, bufB[0]);

    auto body = [&](const int j, auto slot_c, auto edge_c) {
        constexpr int SLOT = decltype(slot_c)::value;             // j & 3; m = m0 + j is odd when j is even
        const H2Row ra = bufA[SLOT & 1], rb = bufB[SLOT & 1];
        const int m = m0 + j;
        if (j + 1 < nIter) {
            load_row(2 * m + 1, bufA[(SLOT + 1) & 1]);
            load_row(2 * m + 2, bufB[(SLOT + 1) & 1]);
        }
        int pc[4];                                                // this pair's chroma lines (U0, U1, V0, V1)
        {
            int sa[4], sb[4], ua[2], va[2], ub[2], vb[2];
            convert_row(ra, edge_c, sa, ua, va);
            convert_row(rb, edge_c, sb, ub, vb);
            // hScale16To15_c: min(val >> 13, 32767)
#pragma unroll
            for (int q = 0; q < 4; q++) hwY[SLOT][q] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(sa[q] >> 13, sb[q] >> 13));
#pragma unroll
            for (int q = 0; q < 2; q++) {
                pc[q] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(ua[q] >> 13, ub[q] >> 13));
                pc[2 + q] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(va[q] >> 13, vb[q] >> 13));
            }
        }
        if (j >= 5 && j < 5 + 2 * nC) {                           // luma row m - 2: pairs m - 3 .. m sit in slots SLOT + 1 .. SLOT + 4 (mod 4)
            const int yo = m - 2;
            unsigned yb[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int Y = a.rnd;
#pragma unroll
                for (int k = 0; k < 4; k++) Y = r2_dot2(hwY[(SLOT + 1 + k) & 3][q], a.vL[k], Y);
                yb[q] = (unsigned)clip_u8_shr(Y, 19);
            }
            if (stores) *reinterpret_cast<unsigned *>(py + ((unsigned)yo * (unsigned)a.ys + (unsigned)xo)) = yb[0] | (yb[1] << 8) | (yb[2] << 16) | (yb[3] << 24);
        }
        if constexpr ((SLOT & 1) == 0) {
            // m = 2t + 1: the pair opens chroma row t + 2 (taps -, 0), feeds t + 1 (3, 4), t (7, 8), t - 1 (11, 12) and closes t - 2 (15, -)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                acc[4][q] = r2_dot2(pc[q], a.vE[0], a.rnd);
                acc[3][q] = r2_dot2(pc[q], a.vE[2], acc[3][q]);
                acc[2][q] = r2_dot2(pc[q], a.vE[4], acc[2][q]);
                acc[1][q] = r2_dot2(pc[q], a.vE[6], acc[1][q]);
                acc[0][q] = r2_dot2(pc[q], a.vE[8], acc[0][q]);
            }
            if (j >= 8) {                                         // chroma row (m - 5) / 2 = c0 + (j - 8) / 2
                const int cy = c0 + ((j - 8) >> 1);
                const unsigned u0 = (unsigned)clip_u8_shr(acc[0][0], 19), u1 = (unsigned)clip_u8_shr(acc[0][1], 19),
                               v0 = (unsigned)clip_u8_shr(acc[0][2], 19), v1 = (unsigned)clip_u8_shr(acc[0][3], 19);
                if (stores) {
                    if (NV) {
                        *reinterpret_cast<unsigned *>(pu + ((unsigned)cy * (unsigned)a.us + (unsigned)xo)) = u0 | (v0 << 8) | (u1 << 16) | (v1 << 24);
                    } else {
                        *reinterpret_cast<unsigned short *>(pu + ((unsigned)cy * (unsigned)a.us + (unsigned)(xo >> 1))) = (unsigned short)(u0 | (u1 << 8));
                        *reinterpret_cast<unsigned short *>(pv + ((unsigned)cy * (unsigned)a.vs + (unsigned)(xo >> 1))) = (unsigned short)(v0 | (v1 << 8));
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int q = 0; q < 4; q++) acc[s][q] = acc[s + 1][q];
        } else {
            // m = 2t: feeds chroma rows t + 1 (taps 1, 2), t (5, 6), t - 1 (9, 10), t - 2 (13, 14)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                acc[3][q] = r2_dot2(pc[q], a.vE[1], acc[3][q]);
                acc[2][q] = r2_dot2(pc[q], a.vE[3], acc[2][q]);
                acc[1][q] = r2_dot2(pc[q], a.vE[5], acc[1][q]);
                acc[0][q] = r2_dot2(pc[q], a.vE[7], acc[0][q]);
            }
        }
    };
    auto run = [&](auto edge_c) {
        for (int j0 = 0; j0 < nIter; j0 += 4) {
            body(j0, std::integral_constant<int, 0>(), edge_c);
            if (j0 + 1 < nIter) body(j0 + 1, std::integral_constant<int, 1>(), edge_c);
            if (j0 + 2 < nIter) body(j0 + 2, std::integral_constant<int, 2>(), edge_c);
            if (j0 + 3 < nIter) body(j0 + 3, std::integral_constant<int, 3>(), edge_c);
        }
    };
    if (edgeWave) run(std::true_type()); else run(std::false_type());
}

// which of the two kernels a launch uses (GMAT_RGB2_SHARED=0: the one that converts the whole window per lane)
static bool rgb2_shared() { const char *e = GMAT_KNOB("GMAT_RGB2_SHARED"); return !(e && !atoi(e)); }
const char *rgb2s_kernel_name() { return rgb2_shared() ? "scale_rgb2h_kernel" : "scale_rgb2s_kernel"; }
bool rgb2h_takes_yuv() { return rgb2_shared(); }

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int rgb2s_prepare(const ScalePlan &p, Rgb2sTables &t)
{
    t = Rgb2sTables();
    const char *off = GMAT_KNOB("GMAT_SCALE_NO_STRIP");
    if (off && atoi(off)) return 0;
    if (!(p.srcFormat == GMAT_PIX_FMT_RGB24 || p.srcFormat == GMAT_PIX_FMT_BGR24)) return 0;
    if (!(p.dstFormat == GMAT_PIX_FMT_RGB24 || p.dstFormat == GMAT_PIX_FMT_BGR24 || p.dstFormat == GMAT_PIX_FMT_RGBA ||
          p.dstFormat == GMAT_PIX_FMT_BGRA)) return 0;
    if (p.srcW != 2 * p.dstW || p.srcH != 2 * p.dstH || p.srcW % 8 || p.srcW < 32 || p.dstH < 8) return 0;
    // half-width chroma plane (pixel pairs) at full height, full-size chroma at the output
    if (!p.chrSrcHSub || p.chrSrcW * 2 != p.srcW || p.chrSrcH != p.srcH || p.chrDstW != p.dstW || p.chrDstH != p.dstH) return 0;
    // identity chroma filter: one tap of 16384 on sample x
    if (p.hChr.taps != 1) return 0;
    for (int x = 0; x < p.hChr.count; x++)
        if (p.hChr.pos[x] != x || p.hChr.coef[x] != 16384) return 0;
    if (!filter_is_edge_replication(p.hLum, p.srcW, t.hL)) return 0;
    if (!filter_is_edge_replication(p.vLum, p.srcH, t.vL)) return 0;
    // the vertical chroma filter of a non-subsampled source is the luma one (utils.c:1838-1873)
    if (p.vChr.taps != p.vLum.taps || p.vChr.pos != p.vLum.pos || p.vChr.coef != p.vLum.coef) return 0;
    t.ok = 1;
    return 0;
}

int launch_scale_rgb2s(const Rgb2sArgs &a0, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (!frames || nframes < 1 || nframes > kYuv2xMaxFrames) return GMAT_ERR(EINVAL);
    Rgb2sArgs a = a0;
    const char *segStr = GMAT_KNOB("GMAT_STRIP_ROWS");
    const int segEnv = segStr ? atoi(segStr) : 0;
    const bool shared = rgb2_shared();
    const int nstrips = shared ? (a.dstW + H2_OUT - 1) / H2_OUT : (a.dstW + R2_STRIP - 1) / R2_STRIP;
    a.nsg = (nstrips + 3) / 4;
    int seg = segEnv > 0 ? segEnv : 0;
    if (!seg) {
        // as launch_scale_yuv2s; 3 waves per SIMD are resident (162 VGPRs), two rounds of them balance best (measured:
        // 16..48 rows at 32 frames 9.05 us per frame, 64 rows 9.8)
        const long rows = (long)a.dstH * nstrips * nframes;
        seg = (int)std::min(48L, std::max(3L, (rows + 6143) / 6144));
        // short launches want longer segments than that, of 1 (mod 4) rows (rows + 3 iterations through a loop unrolled by 4):
        // scale_rgb2h_kernel at 1 / 4 / 8 frames per launch 15.4 -> 14.1, 10.3 -> 9.3, 9.5 -> 9.2 us per frame with 9 / 17 / 17 rows
        // (profiles/r02u_yuv2p_rows_mod4.txt, second table)
        if (shared) seg = std::max(seg, (int)std::min(17L, std::max(9L, (rows + 959) / 960)));
    }
    a.segRows = seg;
    a.nseg = (a.dstH + seg - 1) / seg;
    const int nblk = a.nseg * a.nsg;
    const dim3 grid(a.xcdRemap ? 8 * ((nblk + 7) / 8) : nblk, nframes), block(256);
    const Yuv2xFrames &fr = *frames;
    if (a.srcKind != 0 && !shared) return GMAT_ERR(EINVAL);      // the YUV front end exists in the shared-conversion kernel only
    if (shared) {
#define GMAT_H2(D, S) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_rgb2h_kernel<D, S>), grid, block, 0, stream, a, fr)
#define GMAT_H2_SRC(D) do { if (a.srcKind == 0) GMAT_H2(D, 0); else if (a.srcKind == 1) GMAT_H2(D, 1); else GMAT_H2(D, 2); } while (0)
        switch (a.dstFormat) {
        case GMAT_PIX_FMT_RGB24: GMAT_H2_SRC(0); break;
        case GMAT_PIX_FMT_BGR24: GMAT_H2_SRC(1); break;
        case GMAT_PIX_FMT_RGBA:  GMAT_H2_SRC(2); break;
        case GMAT_PIX_FMT_BGRA:  GMAT_H2_SRC(3); break;
        default: return GMAT_ERR(EINVAL);
        }
#undef GMAT_H2_SRC
#undef GMAT_H2
        GMAT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    switch (a.dstFormat) {
    case GMAT_PIX_FMT_RGB24: hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_rgb2s_kernel<0>), grid, block, 0, stream, a, fr); break;
    case GMAT_PIX_FMT_BGR24: hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_rgb2s_kernel<1>), grid, block, 0, stream, a, fr); break;
    case GMAT_PIX_FMT_RGBA:  hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_rgb2s_kernel<2>), grid, block, 0, stream, a, fr); break;
    case GMAT_PIX_FMT_BGRA:  hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_rgb2s_kernel<3>), grid, block, 0, stream, a, fr); break;
    default: return GMAT_ERR(EINVAL);
    }
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// scale_rgb2y_kernel takes a context when all four filters are "the middle row on an edge-replicated line" with the window
// alignments the kernel assumes, and the geometry is whole lanes
int rgb2y_prepare(const ScalePlan &p, Rgb2yTables &t)
{
    t = Rgb2yTables();
    const char *off = GMAT_KNOB("GMAT_SCALE_NO_STRIP");
    if (off && atoi(off)) return 0;
    if (!(p.srcFormat == GMAT_PIX_FMT_RGB24 || p.srcFormat == GMAT_PIX_FMT_BGR24)) return 0;
    if (!(p.dstFormat == GMAT_PIX_FMT_NV12 || p.dstFormat == GMAT_PIX_FMT_YUV420P)) return 0;
    if (p.srcW != 2 * p.dstW || p.srcH != 2 * p.dstH || p.dstW % 4 || p.dstW < 64 || (p.dstH & 1) || p.dstH < 16) return 0;
    // chroma from pixel pairs at full height, decimated to half the destination on both axes
    if (!p.chrSrcHSub || p.chrSrcW * 2 != p.srcW || p.chrSrcH != p.srcH || p.chrDstW * 2 != p.dstW || p.chrDstH * 2 != p.dstH) return 0;
    if (!filter_is_edge_replication(p.hLum, p.srcW, t.hL)) return 0;
    if (!filter_is_edge_replication(p.hChr, p.chrSrcW, t.hC)) return 0;
    if (!filter_is_edge_replication(p.vLum, p.srcH, t.vL)) return 0;
    int32_t vc[8];
    if (!filter_is_edge_replication_ratio(p.vChr, p.chrSrcH, 4, 6, 8, vc)) return 0;
    // taps k = 0 .. 15 on rows 4c - 6 + k; row pairs (2m - 1, 2m): pair i of 9 holds taps (2i - 1, 2i)
    int tap[18] = {0};
    for (int k = 0; k < 8; k++) { tap[1 + 2 * k] = (int16_t)(vc[k] & 0xFFFF); tap[2 + 2 * k] = (int16_t)((uint32_t)vc[k] >> 16); }
    for (int i = 0; i < 9; i++) t.vE[i] = (int32_t)((uint32_t)(uint16_t)tap[2 * i] | ((uint32_t)(uint16_t)tap[2 * i + 1] << 16));
    t.ok = 1;
    return 0;
}

int launch_scale_rgb2y(const Rgb2yArgs &a0, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (!frames || nframes < 1 || nframes > kYuv2xMaxFrames) return GMAT_ERR(EINVAL);
    Rgb2yArgs a = a0;
    a.nstrips = (a.dstW + H2_OUT - 1) / H2_OUT;
    const char *segStr = GMAT_KNOB("GMAT_STRIP_ROWS");              // tuning / test override: chroma rows per segment
    int seg = segStr ? atoi(segStr) : 0;
    if (seg <= 0) {
        // a segment re-converts 14 source rows of warm-up (7 row pairs) on top of its 4 rows per chroma row; about 3 waves per SIMD.
        // Measured on 4K -> 1080p (profiles/r02x_rgb2y.txt): 32 frames per launch 45 rows 7.6 us per frame (34: 7.8, 54: 8.4, 8: 8.7);
        // 8 frames 12 rows 9.7 (8: 11.1, 27: 13.3); 1 frame 5 rows 19.5 (8: 24.7, 2: 25.7)
        const long rows = (long)(a.dstH >> 1) * a.nstrips * nframes;
        seg = (int)std::min(64L, std::max(5L, (rows + 3071) / 3072));
    }
    a.segRowsC = seg;
    a.nseg = ((a.dstH >> 1) + seg - 1) / seg;
    a.nblk = (a.nseg * a.nstrips + 3) / 4;
    a.xcdRemap = 1;
    const dim3 grid(8 * ((a.nblk + 7) / 8), nframes), block(256);
    if (a.nv12) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_rgb2y_kernel<true>), grid, block, 0, stream, a, *frames);
    else        hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_rgb2y_kernel<false>), grid, block, 0, stream, a, *frames);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
