// k_scale_yuvl.hip — the LINES form of libswscale's generic scaler for 8-bit YUV sources (NV12, YUV420P, YUV444P): two launches through a
// frame of horizontally filtered 15-bit lines.  Round 4.  Integer arithmetic, bit-exact with ONE libswscale context:
//   pass H   hScale8To15_c: min(sum(src * f) >> 7, 32767) of EVERY source row of the three planes          swscale.c:122-136
//            (+ lum / chrRangeTo / FromJpeg_c on the lines, as hscale.c:60,193 applies them)                 swscale.c:157-188
//   pass V   the vertical filters + the output stage of k_scale_yuv.hip's phase 3, the same expressions:
//            yuv2rgb_X_c / _2_c / _1_c + the yuv2rgb.c tables' closed form, yuv2rgb_full_X_c + yuv2rgb_write_full,
//            yuv2planeX_8_c / yuv2nv12cX_c                                                                  output.c:400-450,1680-2200
// What it is for: the down-scales no walker takes — beyond 6.1 : 1 (a 4K frame to a thumbnail: 4 r taps an axis at r : 1), filters the band
// walker's tables do not hold, range conversion — which the LDS-tiled kernel serves at 0.03 - 0.1 of the roofline because a tile's window
// grows with the ratio on BOTH axes (4K -> 640 x 360 rgb24: 63 us a frame), and which it REFUSED once a one-row tile's window passed 64 KB
// (beyond ~ 20 : 1).  A large down-scale is a reduction: the work is the horizontal pass (4 multiply-adds per SOURCE byte whatever the
// ratio) and its output is small (srcH x dstW samples), so the lines go through HBM / the L2 instead of LDS:
//   pass H   a lane owns ONE output column for a run of source row pairs: its coefficient pairs stay in registers (re-based on the host to
//            the 4-byte aligned window start, so the bytes -> int16 pairs step is two v_perm_b32 a dword and nothing else), the window is
//            read straight from the row (adjacent lanes' windows overlap 4 x: L1 hits), two rows at a time, one packed (row 2p, row 2p + 1)
//            dword out;
//   pass V   a lane owns four adjacent outputs of one row; the vertical coefficients are wave-uniform scalar loads, the lines 16-byte loads.
// HBM per 4K -> 480 x 270 rgb24 frame: 12.4 MB in, 0.4 MB out, 3.1 MB of lines written once and read ~ 4 x from the L2.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

namespace gmat {

__device__ __forceinline__ unsigned l_pk16(int lo, int hi) { return ((unsigned)lo & 0xFFFF) | ((unsigned)hi << 16); }

// bytes 0,1 / 2,3 of a dword as an int16 pair; the U / V samples of two interleaved chroma pairs
__device__ __forceinline__ int pair_lo(unsigned d) { return (int)__builtin_amdgcn_perm(0u, d, 0x0c010c00u); }
__device__ __forceinline__ int pair_hi(unsigned d) { return (int)__builtin_amdgcn_perm(0u, d, 0x0c030c02u); }
__device__ __forceinline__ int pair_u(unsigned d)  { return (int)__builtin_amdgcn_perm(0u, d, 0x0c020c00u); }
__device__ __forceinline__ int pair_v(unsigned d)  { return (int)__builtin_amdgcn_perm(0u, d, 0x0c030c01u); }

__device__ __forceinline__ int l_lum_range(int v, int rc)
{
    if (rc == 1) return (m24(min(v, 30189), 19077) - 39057361) >> 14;      // lumRangeToJpeg_c, swscale.c:176-181
    if (rc == 2) return (m24(v, 14071) + 33561947) >> 14;                  // lumRangeFromJpeg_c, :183-188
    return v;
}
__device__ __forceinline__ int l_chr_range(int v, int rc)
{
    if (rc == 1) return (m24(min(v, 30775), 4663) - 9289992) >> 12;        // chrRangeToJpeg_c, swscale.c:157-164
    if (rc == 2) return (m24(v, 1799) + 4081085) >> 11;                    // chrRangeFromJpeg_c, :166-173
    return v;
}

// ---- pass H ----------------------------------------------------------------------------------------------------------------------------
// A plane as a raw buffer resource: lane offset in a VGPR, row offset in the instruction's scalar offset, and a dword past the plane's last
// one reads as 0 (a window re-based to a 4-byte aligned start may overhang the row's last whole dword by its padding taps, whose coefficients
// are zero; the last lanes of a segment's 16-byte pieces overhang by more).
struct LPlane {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ LPlane(const uint8_t *p, unsigned bytes) : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p), 0, bytes, 0x00020000)) {}
    __device__ __forceinline__ uint4 ld16(unsigned off, unsigned row) const { const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, off, row, 0); return make_uint4(v.x, v.y, v.z, v.w); }
#else
    // hipcc's host pass (never executed) and the CPU emulation of the test suite: the descriptor's range check restated
    const uint8_t *p; unsigned n;
    __host__ __device__ LPlane(const uint8_t *q, unsigned bytes) : p(q), n(bytes) {}
    __host__ __device__ unsigned dw(size_t o) const { unsigned v = 0; if (o + 4 <= n) std::memcpy(&v, p + o, 4); return v; }
    __host__ __device__ uint4 ld16(unsigned off, unsigned row) const { const size_t o = (size_t)row + off; return make_uint4(dw(o), dw(o + 4), dw(o + 8), dw(o + 12)); }
#endif
};

// One wave = one ITEM: 64 output columns of a plane x `rp` source row pairs.  Items of a frame, in order: luma [0, nItemL), then the chroma
// planes — interleaved chroma: [nItemL, nItemL + nItemC) makes U and V lines together; planar: U items, then V items.
// Per row pair: the wave reads the bytes its 64 windows span ONCE, 16 bytes a lane (kLineNld pieces of 1 KB at most; the first version read
// every window dword by dword straight from the row: lanes r bytes apart, a quarter of the L1's rate — 12.4 us a 4K frame), through a
// wave-private LDS image (no barrier: one wave, and the LDS runs a wave's instructions in order), the next pair's bytes in flight during this
// pair's arithmetic.  P = coefficient pairs a lane holds (the context's longest re-based filter, rounded up to an instance).
constexpr int kLineNld = 5;

// N dwords of a window image, R dwords a read (the image address is a multiple of 4 R bytes: the host re-based the window so)
template <int N, int R>
__device__ __forceinline__ void read_window(const unsigned *w, unsigned (&d)[N])
{
    static_assert(N % R == 0, "window dwords");
    if constexpr (R == 4) {
#pragma unroll
        for (int j = 0; j < N / 4; j++) { const uint4 v = reinterpret_cast<const uint4 *>(w)[j]; d[4 * j] = v.x; d[4 * j + 1] = v.y; d[4 * j + 2] = v.z; d[4 * j + 3] = v.w; }
    } else if constexpr (R == 2) {
#pragma unroll
        for (int j = 0; j < N / 2; j++) { const uint2 v = reinterpret_cast<const uint2 *>(w)[j]; d[2 * j] = v.x; d[2 * j + 1] = v.y; }
    } else {
#pragma unroll
        for (int j = 0; j < N; j++) d[j] = w[j];
    }
}

// RW: dwords a lane reads from the image at once in a byte plane's items (1 | 2 | 4: the windows start on 4 RW-byte boundaries; the interleaved
// chroma items, whose lanes are twice as far apart, read 2 | 4 | 4) — a lane stride of r bytes read dword by dword is a 2-way bank conflict from
// r = 8 on (measured: the LDS busy 62 % of the kernel, half of it conflicts); NLD: 1 KB pieces a row at most.
template <int P, int RW, int NLD>
__global__ __launch_bounds__(256) void scale_yuvl_h_kernel(YuvLArgs a, Yuv2xFrames fr)
{
    constexpr int RB = (P / 2) % RW == 0 ? RW : 1;                // byte planes: P / 2 dwords a row
    constexpr int RU = RW == 1 ? 2 : (P % 4 == 0 ? 4 : 2);       // interleaved chroma: P dwords a row
    HIP_DYNAMIC_SHARED(uint4, lds_base)
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave));
    if (item >= a.nItem) return;
    int32_t *inter = a.inter + (size_t)f * a.frameInts;
    const bool lumaJob = item < a.nItemL;
    const int  cjob = lumaJob ? 0 : item - a.nItemL;
    const bool vPlane = !lumaJob && !a.nv12 && cjob >= a.nItemC;              // planar chroma: the V plane's items
    const int  it = lumaJob ? item : vPlane ? cjob - a.nItemC : cjob;
    const int  ncol = lumaJob ? a.nColL : a.nColC;
    const int  chunk = it / ncol, cg = it - chunk * ncol;
    const int  W = lumaJob ? a.dstW : a.chrDstW, H = lumaJob ? a.srcH : a.chrSrcH;
    const int  pitch = lumaJob ? a.pitchL : a.pitchC, pairRows = lumaJob ? a.pairRowsL : a.pairRowsC;
    const int  gx = min(cg * 64 + lane, W - 1);
    const int32_t *tab = lumaJob ? a.hL : a.hC;
    const unsigned off = (unsigned)(lumaJob ? a.offL : a.offC)[gx];
    int cf[P];
#pragma unroll
    for (int k = 0; k < P; k++) cf[k] = tab[(size_t)k * pitch + cg * 64 + lane];        // (pair k of every column: coalesced)
    const int p0 = chunk * a.rp, p1 = min(p0 + a.rp, pairRows);
    const int rc = a.rangeConv;
    const bool bytePlane = lumaJob || !a.nv12;
    const bool d4 = bytePlane && (lumaJob ? a.dot4L : a.dot4C) != 0;           // the table is in the signed-byte form (yuvl_prepare)
    const unsigned bias = d4 ? 0x80808080u : 0u;                               // ... and the row image holds sample - 128
    const int k4 = d4 ? tab[(size_t)P * pitch + cg * 64 + lane] : 0;           // 128 sum(c)
    const uint8_t *plane = lumaJob ? fr.y[f] : vPlane ? fr.v[f] : fr.u[f];
    const unsigned stride = (unsigned)(lumaJob ? a.ys : vPlane ? a.vs : a.us);
    const unsigned rowBytes = (unsigned)(lumaJob ? a.srcW : a.nv12 ? 2 * a.chrSrcW : a.chrSrcW);
    const LPlane pl(plane, (unsigned)(H - 1) * stride + ((rowBytes + 3u) & ~3u));
    // the segment of a row the wave's windows span (they start in column order): [seg0, segEnd), seg0 on a 16-byte boundary of the row
    const unsigned seg0 = (unsigned)__builtin_amdgcn_readlane((int)off, 0) & ~15u;
    const unsigned segEnd = (unsigned)__builtin_amdgcn_readlane((int)off, 63) + (bytePlane ? 2u * P : 4u * P);
    const int nld = __builtin_amdgcn_readfirstlane((int)((segEnd - seg0 + 1023u) >> 10));          // <= a.nld <= NLD (the host's maximum)
    const unsigned rowDw = (unsigned)a.nld * 256u;
    unsigned *img = reinterpret_cast<unsigned *>(lds_base) + (unsigned)wave * 2u * rowDw;
    const unsigned lo = seg0 + 16u * lane;
    // the window's place in the image, opaque per role: seen as ONE address in both roles' branches the compiler hoists the reads they share
    // above the branch dword by dword, and the wide reads are gone
    auto window = [&]() {
        unsigned wo = (off - seg0) >> 2;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(wo));
#endif
        return (const unsigned *)(img + wo);
    };

    uint4 raw[2][NLD];
    auto request = [&](int p) {
        const unsigned r0 = (unsigned)(2 * p) * stride, r1 = (unsigned)min(2 * p + 1, H - 1) * stride;
#pragma unroll
        for (int i = 0; i < NLD; i++)
            if (i < nld && lo + 1024u * i < segEnd) { raw[0][i] = pl.ld16(lo + 1024u * i, r0); raw[1][i] = pl.ld16(lo + 1024u * i, r1); }
    };
    if (p0 < p1) request(p0);
    for (int p = p0; p < p1; p++) {
        __builtin_amdgcn_wave_barrier();         // (emulation: the lanes of a wave are fibers; on the GPU the LDS runs a wave's instructions in order)
#pragma unroll
        for (int i = 0; i < NLD; i++)
            if (i < nld && lo + 1024u * i < segEnd) {
                *reinterpret_cast<uint4 *>(img + 256 * i + 4 * lane) = make_uint4(raw[0][i].x ^ bias, raw[0][i].y ^ bias, raw[0][i].z ^ bias, raw[0][i].w ^ bias);
                *reinterpret_cast<uint4 *>(img + rowDw + 256 * i + 4 * lane) = make_uint4(raw[1][i].x ^ bias, raw[1][i].y ^ bias, raw[1][i].z ^ bias, raw[1][i].w ^ bias);
            }
        __builtin_amdgcn_wave_barrier();
        if (p + 1 < p1) request(p + 1);
        if (bytePlane) {
            const unsigned *win = window();
            unsigned d0[P / 2], d1[P / 2];
            read_window<P / 2, RB>(win, d0);
            read_window<P / 2, RB>(win + rowDw, d1);
            int s0 = 0, s1 = 0;
            if (d4) {
                int h0 = 0, h1 = 0;
                s0 = s1 = k4;
#pragma unroll
                for (int j = 0; j < P / 2; j++) {
                    h0 = dot4s((int)d0[j], cf[2 * j], h0); s0 = dot4s((int)d0[j], cf[2 * j + 1], s0);
                    h1 = dot4s((int)d1[j], cf[2 * j], h1); s1 = dot4s((int)d1[j], cf[2 * j + 1], s1);
                }
                s0 += h0 * 256; s1 += h1 * 256;
            } else {
#pragma unroll
            for (int j = 0; j < P / 2; j++) {
                s0 = dot2(pair_lo(d0[j]), cf[2 * j], s0); s0 = dot2(pair_hi(d0[j]), cf[2 * j + 1], s0);
                s1 = dot2(pair_lo(d1[j]), cf[2 * j], s1); s1 = dot2(pair_hi(d1[j]), cf[2 * j + 1], s1);
            }
            }
            int l0 = min(s0 >> 7, 32767), l1 = min(s1 >> 7, 32767);
            if (rc) {
                if (lumaJob) { l0 = l_lum_range(l0, rc); l1 = l_lum_range(l1, rc); }
                else         { l0 = l_chr_range(l0, rc); l1 = l_chr_range(l1, rc); }
            }
            inter[(lumaJob ? 0 : vPlane ? a.baseV : a.baseU) + (size_t)p * pitch + cg * 64 + lane] = (int)l_pk16(l0, l1);
        } else {
            const unsigned *win = window();
            int uv[2][2];
#pragma unroll
            for (int r = 0; r < 2; r++) {                        // a row at a time: P dwords of window
                unsigned d[P];
                read_window<P, RU>(win + r * rowDw, d);
                int u = 0, v = 0;
#pragma unroll
                for (int j = 0; j < P; j++) { u = dot2(pair_u(d[j]), cf[j], u); v = dot2(pair_v(d[j]), cf[j], v); }
                u = min(u >> 7, 32767); v = min(v >> 7, 32767);
                if (rc) { u = l_chr_range(u, rc); v = l_chr_range(v, rc); }
                uv[r][0] = u; uv[r][1] = v;
            }
            inter[a.baseU + (size_t)p * pitch + cg * 64 + lane] = (int)l_pk16(uv[0][0], uv[1][0]);
            inter[a.baseV + (size_t)p * pitch + cg * 64 + lane] = (int)l_pk16(uv[0][1], uv[1][1]);
        }
    }
}

// ---- pass H for 16-bit samples (P010LE / P016LE: a luma plane and a plane of interleaved (U, V) pairs; planar 10 / 16 bit) ------------------
// hScale16To15_c (swscale.c:93-119): min((sum(src * f)) >> sh, 32767), sh = depth - 1.  What the taps multiply is k_scale_yuv.hip's 16-bit image
// (yuv_phase1_src16): P010 sample >> 6; planar 10 bit as it is; 16 bits as sample - 32768 (the signed v_dot2 operand) with the accumulator
// started at 32768 * 16384 (hBias; the host checks that every filter row sums to 16384).  A dword of a plane is one coefficient pair; of the
// interleaved plane two dwords make a U pair and a V pair.  Same structure as the 8-bit kernel, a ROW at a time (a lane stride of 2 r or 4 r
// bytes: up to kLineNld16 pieces a row), the window in groups of eight dwords.  Not tuned: it is what serves these sources beyond the tiled
// kernel's reach (refused from ~ 10 : 1 before).
constexpr int kLineNld16 = 9;

template <int P>
__global__ __launch_bounds__(256) void scale_yuvl_h16_kernel(YuvLArgs a, Yuv2xFrames fr)
{
    HIP_DYNAMIC_SHARED(uint4, lds_base)
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave));
    if (item >= a.nItem) return;
    int32_t *inter = a.inter + (size_t)f * a.frameInts;
    const bool semi = a.src16 < 17;                                            // P010 / P016: interleaved chroma
    const bool lumaJob = item < a.nItemL;
    const int  cjob = lumaJob ? 0 : item - a.nItemL;
    const bool vPlane = !lumaJob && !semi && cjob >= a.nItemC;
    const int  it = lumaJob ? item : vPlane ? cjob - a.nItemC : cjob;
    const int  ncol = lumaJob ? a.nColL : a.nColC;
    const int  chunk = it / ncol, cg = it - chunk * ncol;
    const int  W = lumaJob ? a.dstW : a.chrDstW, H = lumaJob ? a.srcH : a.chrSrcH;
    const int  pitch = lumaJob ? a.pitchL : a.pitchC, pairRows = lumaJob ? a.pairRowsL : a.pairRowsC;
    const int  gx = min(cg * 64 + lane, W - 1);
    const int32_t *tab = lumaJob ? a.hL : a.hC;
    const unsigned off = (unsigned)(lumaJob ? a.offL : a.offC)[gx];
    int cf[P];
#pragma unroll
    for (int k = 0; k < P; k++) cf[k] = tab[(size_t)k * pitch + cg * 64 + lane];
    const int p0 = chunk * a.rp, p1 = min(p0 + a.rp, pairRows);
    const int rc = a.rangeConv;
    const bool uvJob = !lumaJob && semi;
    const uint8_t *plane = lumaJob ? fr.y[f] : vPlane ? fr.v[f] : fr.u[f];
    const unsigned stride = (unsigned)(lumaJob ? a.ys : vPlane ? a.vs : a.us);
    const unsigned rowBytes = 2u * (unsigned)(lumaJob ? a.srcW : semi ? 2 * a.chrSrcW : a.chrSrcW);
    const LPlane pl(plane, (unsigned)(H - 1) * stride + ((rowBytes + 3u) & ~3u));
    const unsigned seg0 = (unsigned)__builtin_amdgcn_readlane((int)off, 0) & ~15u;
    const unsigned segEnd = (unsigned)__builtin_amdgcn_readlane((int)off, 63) + (uvJob ? 8u * P : 4u * P);
    const int nld = __builtin_amdgcn_readfirstlane((int)((segEnd - seg0 + 1023u) >> 10));
    unsigned *img = reinterpret_cast<unsigned *>(lds_base) + (unsigned)wave * (unsigned)a.nld * 256u;
    const unsigned *win = img + ((off - seg0) >> 2);                           // (a multiple of 8 bytes: the host re-based the windows so)
    const unsigned lo = seg0 + 16u * lane;
    const int kind = a.src16, sh = a.hShift, bias = a.hBias;
    auto conv = [&](unsigned v) -> int { return (int)(kind == 10 ? (v >> 6) & 0x03FF03FFu : kind == 18 ? v : v ^ 0x80008000u); };

    // the next ROW's bytes in flight during this row's arithmetic (as the 8-bit kernel's next pair)
    uint4 raw[kLineNld16];
    auto request = [&](int q) {
        const unsigned rowOff = (unsigned)min(q, H - 1) * stride;
#pragma unroll
        for (int i = 0; i < kLineNld16; i++)
            if (i < nld && lo + 1024u * i < segEnd) raw[i] = pl.ld16(lo + 1024u * i, rowOff);
    };
    if (p0 < p1) request(2 * p0);
    for (int p = p0; p < p1; p++) {
        int out[2][2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < kLineNld16; i++)
                if (i < nld && lo + 1024u * i < segEnd) *reinterpret_cast<uint4 *>(img + 256 * i + 4 * lane) = raw[i];
            __builtin_amdgcn_wave_barrier();
            if (2 * p + r + 1 < 2 * p1) request(2 * p + r + 1);
            int s0 = bias, s1 = bias;
            if (!uvJob) {
#pragma unroll
                for (int j0 = 0; j0 < P; j0 += 8) {
                    unsigned d[8];
#pragma unroll
                    for (int j = 0; j < 8; j += 2) if (j0 + j < P) { const uint2 t = *reinterpret_cast<const uint2 *>(win + j0 + j); d[j] = t.x; d[j + 1] = t.y; }
#pragma unroll
                    for (int j = 0; j < 8; j++) if (j0 + j < P) s0 = dot2(conv(d[j]), cf[j0 + j], s0);
                }
            } else {
#pragma unroll
                for (int j0 = 0; j0 < P; j0 += 4) {
                    unsigned d[8];
#pragma unroll
                    for (int j = 0; j < 4; j++) if (j0 + j < P) { const uint2 t = *reinterpret_cast<const uint2 *>(win + 2 * (j0 + j)); d[2 * j] = t.x; d[2 * j + 1] = t.y; }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (j0 + j < P) {
                            s0 = dot2(conv(__builtin_amdgcn_perm(d[2 * j + 1], d[2 * j], 0x05040100u)), cf[j0 + j], s0);       // U of two samples
                            s1 = dot2(conv(__builtin_amdgcn_perm(d[2 * j + 1], d[2 * j], 0x07060302u)), cf[j0 + j], s1);       // V
                        }
                }
            }
            int v0 = min(s0 >> sh, 32767), v1 = min(s1 >> sh, 32767);
            if (rc) { v0 = lumaJob ? l_lum_range(v0, rc) : l_chr_range(v0, rc); v1 = l_chr_range(v1, rc); }
            out[r][0] = v0; out[r][1] = v1;
        }
        const size_t o = (size_t)p * pitch + cg * 64 + lane;
        if (!uvJob) inter[(lumaJob ? 0 : vPlane ? a.baseV : a.baseU) + o] = (int)l_pk16(out[0][0], out[1][0]);
        else { inter[a.baseU + o] = (int)l_pk16(out[0][0], out[1][0]); inter[a.baseV + o] = (int)l_pk16(out[0][1], out[1][1]); }
    }
}

// ---- pass V ----------------------------------------------------------------------------------------------------------------------------
// MODE 0: packed RGB, half chroma (LUT form)   1: packed RGB, full chroma   2: YUV 4:2:0 (NV12 / YUV420P)   3: planar YUV 4:4:4
// A block = 4 output rows x 256 columns (a wave a row, a lane four adjacent outputs); MODE 2: its 2 chroma rows x 128 columns after them,
// a thread a column.
template <int MODE>
__global__ __launch_bounds__(256) void scale_yuvl_v_kernel(YuvLArgs a, Yuv2xFrames fr)
{
    constexpr bool FULL = MODE == 1 || MODE == 3;
    constexpr bool YUVOUT = MODE >= 2;
    const int f = blockIdx.y;
    const int32_t *inter = a.inter + (size_t)f * a.frameInts;
    const int32_t *hy = inter, *hu = inter + a.baseU, *hv = inter + a.baseV;
    uint8_t *dst = fr.dst[f], *dstU = fr.dstU[f], *dstV = fr.dstV[f];
    const int brow = blockIdx.x / a.nColV, bcol = blockIdx.x - brow * a.nColV;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int yo = __builtin_amdgcn_readfirstlane(brow * 4 + wave);
    const int xo = bcol * 256 + 4 * lane;
    const DevFilter &VL = a.vLum, &VC = a.vChr;

    if (yo < a.dstH && xo < a.dstW) {
        const int vpL = uniform_load(VL.pos_even, yo) >> 1, lr = uniform_load(VL.round, yo);
        int Y[4] = {lr, lr, lr, lr};
        const int32_t *ly = hy + (size_t)vpL * a.pitchL + xo;
#pragma unroll 8
        for (int k = 0; k < VL.pairs; k++) {
            const int cf = uniform_load(VL.packed, yo * VL.pairs + k);
            const int4 v = *reinterpret_cast<const int4 *>(ly + (size_t)k * a.pitchL);
            Y[0] = dot2(v.x, cf, Y[0]); Y[1] = dot2(v.y, cf, Y[1]); Y[2] = dot2(v.z, cf, Y[2]); Y[3] = dot2(v.w, cf, Y[3]);
        }
        const int nx = min(4, a.dstW - xo);
        if (YUVOUT && a.dst16) {
            // yuv2p010lX_c / yuv2planeX_10_c: clip_uintp2((1 << 16 + sum) >> 17, 10) (P010: << 6); lr holds the 1 << 16
            unsigned short *d16 = reinterpret_cast<unsigned short *>(dst + (size_t)yo * a.ds) + xo;
            unsigned w[4];
#pragma unroll
            for (int i = 0; i < 4; i++) w[i] = (unsigned)min(max(Y[i] >> 17, 0), 1023) << a.dstShift;
            if (a.dstAligned && nx == 4) *reinterpret_cast<uint2 *>(d16) = make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16));
            else for (int i = 0; i < nx; i++) d16[i] = (unsigned short)w[i];
        } else if (YUVOUT) {
            // yuv2planeX_8_c: clip_u8((dither << 12 + sum) >> 19); lr holds the 64 << 12 of an 8-bit source, a deeper one's ordered dither on top
            if (a.dither8) {
#pragma unroll
                for (int i = 0; i < 4; i++) Y[i] += dither_delta(xo + i, yo);
            }
            uint8_t *d = dst + (size_t)yo * a.ds + xo;
            const unsigned o = (unsigned)clip_u8_shr(Y[0], 19) | ((unsigned)clip_u8_shr(Y[1], 19) << 8) |
                               ((unsigned)clip_u8_shr(Y[2], 19) << 16) | ((unsigned)clip_u8_shr(Y[3], 19) << 24);
            if (a.dstAligned && nx == 4) *reinterpret_cast<unsigned *>(d) = o;
            else for (int i = 0; i < nx; i++) d[i] = (uint8_t)(o >> (8 * i));
        } else {
            const int vpC = uniform_load(VC.pos_even, yo) >> 1, cr = uniform_load(VC.round, yo);
            unsigned px[4];
            if (FULL) {
                int U[4] = {cr, cr, cr, cr}, V[4] = {cr, cr, cr, cr};
                const int32_t *lu = hu + (size_t)vpC * a.pitchC + xo, *lv = hv + (size_t)vpC * a.pitchC + xo;
#pragma unroll 8
            for (int k = 0; k < VC.pairs; k++) {
                    const int cf = uniform_load(VC.packed, yo * VC.pairs + k);
                    const int4 u = *reinterpret_cast<const int4 *>(lu + (size_t)k * a.pitchC);
                    const int4 v = *reinterpret_cast<const int4 *>(lv + (size_t)k * a.pitchC);
                    U[0] = dot2(u.x, cf, U[0]); U[1] = dot2(u.y, cf, U[1]); U[2] = dot2(u.z, cf, U[2]); U[3] = dot2(u.w, cf, U[3]);
                    V[0] = dot2(v.x, cf, V[0]); V[1] = dot2(v.y, cf, V[1]); V[2] = dot2(v.z, cf, V[2]); V[3] = dot2(v.w, cf, V[3]);
                }
#pragma unroll
                for (int i = 0; i < 4; i++) px[i] = yuv_to_rgb_full(a.y2r, Y[i] >> 10, U[i] >> 10, V[i] >> 10);
            } else {
                int U[2] = {cr, cr}, V[2] = {cr, cr};
                const int32_t *lu = hu + (size_t)vpC * a.pitchC + (xo >> 1), *lv = hv + (size_t)vpC * a.pitchC + (xo >> 1);
#pragma unroll 8
            for (int k = 0; k < VC.pairs; k++) {
                    const int cf = uniform_load(VC.packed, yo * VC.pairs + k);
                    const uint2 u = *reinterpret_cast<const uint2 *>(lu + (size_t)k * a.pitchC);
                    const uint2 v = *reinterpret_cast<const uint2 *>(lv + (size_t)k * a.pitchC);
                    U[0] = dot2((int)u.x, cf, U[0]); U[1] = dot2((int)u.y, cf, U[1]);
                    V[0] = dot2((int)v.x, cf, V[0]); V[1] = dot2((int)v.y, cf, V[1]);
                }
                // table_rV / gU / gV / bU are indexed with av_clip_uint8 (yuv2rgb.c:737-760)
                const ChromaTerms t0 = chroma_terms(a.y2r, clip_u8_shr(U[0], 19), clip_u8_shr(V[0], 19));
                const ChromaTerms t1 = chroma_terms(a.y2r, clip_u8_shr(U[1], 19), clip_u8_shr(V[1], 19));
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int ya = m24(Y[i] >> 19, a.y2r.cy), yb = m24(Y[i + 2] >> 19, a.y2r.cy);
                    px[i]     = (unsigned)luma_chan(t0.r, ya) | ((unsigned)luma_chan(t0.g, ya) << 8) | ((unsigned)luma_chan(t0.b, ya) << 16);
                    px[i + 2] = (unsigned)luma_chan(t1.r, yb) | ((unsigned)luma_chan(t1.g, yb) << 8) | ((unsigned)luma_chan(t1.b, yb) << 16);
                }
            }
            const int bpp = (a.dstFormat == GMAT_PIX_FMT_RGBA || a.dstFormat == GMAT_PIX_FMT_BGRA) ? 4 : 3;
            const bool swap_rb = a.dstFormat == GMAT_PIX_FMT_BGR24 || a.dstFormat == GMAT_PIX_FMT_BGRA;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                unsigned c = px[i];
                if (swap_rb) c = ((c & 0xFF) << 16) | (c & 0xFF00) | ((c >> 16) & 0xFF);
                px[i] = c | 0xFF000000u;
            }
            uint8_t *d = dst + (size_t)yo * a.ds + (size_t)xo * bpp;
            if (a.dstAligned && nx == 4) {
                if (bpp == 4) *reinterpret_cast<uint4 *>(d) = make_uint4(px[0], px[1], px[2], px[3]);
                else {
                    uint3 o3;
                    o3.x = (px[0] & 0xFFFFFF) | (px[1] << 24);
                    o3.y = ((px[1] >> 8) & 0xFFFF) | (px[2] << 16);
                    o3.z = ((px[2] >> 16) & 0xFF) | (px[3] << 8);
                    *reinterpret_cast<uint3 *>(d) = o3;
                }
            } else {
                for (int i = 0; i < nx; i++) {
                    d[i * bpp + 0] = (uint8_t)px[i]; d[i * bpp + 1] = (uint8_t)(px[i] >> 8); d[i * bpp + 2] = (uint8_t)(px[i] >> 16);
                    if (bpp == 4) d[i * bpp + 3] = 255;
                }
            }
        }
    }
    if (YUVOUT) {
        // chroma: MODE 2: rows brow * 2 + (wave >> 1), 128 columns a row, a thread a column; MODE 3: the luma rows' own geometry, a thread 4 columns
        constexpr int CVS = MODE == 2 ? 1 : 0;
        const int cy = __builtin_amdgcn_readfirstlane(CVS ? brow * 2 + (wave >> 1) : brow * 4 + wave);
        if (cy >= a.chrDstH) return;
        const int vp = uniform_load(VC.pos_even, cy) >> 1, rnd = uniform_load(VC.round, cy);
        if (CVS) {
            const int cx = bcol * 128 + (threadIdx.x & 127);
            if (cx >= a.chrDstW) return;
            int U = rnd, V = rnd;
            const int32_t *lu = hu + (size_t)vp * a.pitchC + cx, *lv = hv + (size_t)vp * a.pitchC + cx;
#pragma unroll 8
            for (int k = 0; k < VC.pairs; k++) {
                const int cf = uniform_load(VC.packed, cy * VC.pairs + k);
                U = dot2(lu[(size_t)k * a.pitchC], cf, U);
                V = dot2(lv[(size_t)k * a.pitchC], cf, V);
            }
            if (a.dst16) {                                          // yuv2p010cX_c (interleaved, << 6) / yuv2planeX_10_c per plane
                const unsigned u10 = (unsigned)min(max(U >> 17, 0), 1023), v10 = (unsigned)min(max(V >> 17, 0), 1023);
                if (a.dst16 == 1) reinterpret_cast<unsigned *>(dstU + (size_t)cy * a.dsU)[cx] = (u10 << 6) | (v10 << 22);
                else {
                    reinterpret_cast<unsigned short *>(dstU + (size_t)cy * a.dsU)[cx] = (unsigned short)u10;
                    reinterpret_cast<unsigned short *>(dstV + (size_t)cy * a.dsV)[cx] = (unsigned short)v10;
                }
                return;
            }
            if (a.dither8) { U += dither_delta(cx, cy); V += dither_delta(cx + 3, cy); }       // chrDither8: V three columns on (vscale.c:98,101, output.c:433-434)
            const unsigned ub = (unsigned)clip_u8_shr(U, 19), vb = (unsigned)clip_u8_shr(V, 19);
            if (a.dstNv12) {
                uint8_t *d = dstU + (size_t)cy * a.dsU + 2 * cx;
                if (a.dstAligned) *reinterpret_cast<unsigned short *>(d) = (unsigned short)(ub | (vb << 8));
                else { d[0] = (uint8_t)ub; d[1] = (uint8_t)vb; }
            } else {
                dstU[(size_t)cy * a.dsU + cx] = (uint8_t)ub;
                dstV[(size_t)cy * a.dsV + cx] = (uint8_t)vb;
            }
        } else {
            if (xo >= a.chrDstW) return;
            int U[4] = {rnd, rnd, rnd, rnd}, V[4] = {rnd, rnd, rnd, rnd};
            const int32_t *lu = hu + (size_t)vp * a.pitchC + xo, *lv = hv + (size_t)vp * a.pitchC + xo;
#pragma unroll 8
            for (int k = 0; k < VC.pairs; k++) {
                const int cf = uniform_load(VC.packed, cy * VC.pairs + k);
                const int4 u = *reinterpret_cast<const int4 *>(lu + (size_t)k * a.pitchC);
                const int4 v = *reinterpret_cast<const int4 *>(lv + (size_t)k * a.pitchC);
                U[0] = dot2(u.x, cf, U[0]); U[1] = dot2(u.y, cf, U[1]); U[2] = dot2(u.z, cf, U[2]); U[3] = dot2(u.w, cf, U[3]);
                V[0] = dot2(v.x, cf, V[0]); V[1] = dot2(v.y, cf, V[1]); V[2] = dot2(v.z, cf, V[2]); V[3] = dot2(v.w, cf, V[3]);
            }
            if (a.dither8) {
#pragma unroll
                for (int i = 0; i < 4; i++) { U[i] += dither_delta(xo + i, cy); V[i] += dither_delta(xo + i + 3, cy); }
            }
            const int nx = min(4, a.chrDstW - xo);
            uint8_t *du = dstU + (size_t)cy * a.dsU + xo, *dv = dstV + (size_t)cy * a.dsV + xo;
            const unsigned ou = (unsigned)clip_u8_shr(U[0], 19) | ((unsigned)clip_u8_shr(U[1], 19) << 8) | ((unsigned)clip_u8_shr(U[2], 19) << 16) | ((unsigned)clip_u8_shr(U[3], 19) << 24);
            const unsigned ov = (unsigned)clip_u8_shr(V[0], 19) | ((unsigned)clip_u8_shr(V[1], 19) << 8) | ((unsigned)clip_u8_shr(V[2], 19) << 16) | ((unsigned)clip_u8_shr(V[3], 19) << 24);
            if (a.dstAligned && nx == 4) { *reinterpret_cast<unsigned *>(du) = ou; *reinterpret_cast<unsigned *>(dv) = ov; }
            else for (int i = 0; i < nx; i++) { du[i] = (uint8_t)(ou >> (8 * i)); dv[i] = (uint8_t)(ov >> (8 * i)); }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static const int kLineInstances[] = {8, 12, 16, 20, 24, 32, 40, 48, 56, 72};

// coefficient pairs of a filter re-based to a window that starts on a multiple of `align` SAMPLES (up to align - 1 leading zero taps);
// tab[k][column]: pair k of every column side by side (pitch = the lines frame's); off: the window's byte offset in the row
static void rebase(const FilterBank &fb, int P, int pitch, int align, int bytesPerSample, std::vector<int32_t> &tab, std::vector<int32_t> &off)
{
    tab.assign((size_t)P * pitch, 0);
    off.resize(fb.count);
    for (int i = 0; i < fb.count; i++) {
        const int lead = fb.pos[i] & (align - 1);
        off[i] = (fb.pos[i] - lead) * bytesPerSample;
        int16_t w[2 * 72 + 2] = {0};
        for (int t = 0; t < fb.taps; t++) w[lead + t] = fb.coef[(size_t)i * fb.taps + t];
        for (int k = 0; k < P; k++) tab[(size_t)k * pitch + i] = (int32_t)(((uint32_t)(uint16_t)w[2 * k]) | ((uint32_t)(uint16_t)w[2 * k + 1] << 16));
    }
}

int yuvl_prepare(const ScalePlan &p, const YuvScaleTiling &g, YuvLTables &t)
{
    t = YuvLTables();
    const bool semi16 = is_p01x(p.srcFormat);                                   // 16-bit samples, interleaved chroma
    const bool pl16 = pl16_depth(p.srcFormat) != 0 && p.srcFormat != GMAT_PIX_FMT_PRIV_RGB8_PLANES;      // ... planar (incl. the planes of an RGBA64 frame)
    if (!(is_yuv8_src(p.srcFormat) || semi16 || pl16)) return 0;
    if (!(is_packed_rgb(p.dstFormat) || is_yuv420(p.dstFormat) || p.dstFormat == GMAT_PIX_FMT_YUV444P || is_dst10(p.dstFormat))) return 0;
    if (p.hLum.taps > 128 || p.hChr.taps > 128) return 0;
    if (semi16 || pl16) {
        // windows on 8-byte boundaries: 4 samples of a plane, 2 (U, V) pairs of the interleaved plane; a row at a time, up to kLineNld16 pieces
        const int need = std::max((p.hLum.taps + 3 + 1) / 2, (p.hChr.taps + (semi16 ? 1 : 3) + 1) / 2);
        int P = 0;
        for (int v : kLineInstances) if (v >= need) { P = v; break; }
        if (!P) return 0;
        t.P = P; t.RW = 2; t.wide = 1;
        t.pitchL = align_up(p.dstW, 64); t.pitchC = align_up(p.chrDstW, 64);
        rebase(p.hLum, P, t.pitchL, 4, 2, t.hL, t.offL);
        if (semi16) rebase(p.hChr, P, t.pitchC, 2, 4, t.hC, t.offC);
        else        rebase(p.hChr, P, t.pitchC, 4, 2, t.hC, t.offC);
        for (size_t i = 1; i < t.offL.size(); i++) if (t.offL[i] < t.offL[i - 1]) return 0;
        for (size_t i = 1; i < t.offC.size(); i++) if (t.offC[i] < t.offC[i - 1]) return 0;
        auto pieces16 = [&](const std::vector<int32_t> &off, int winBytes) {
            int n = 1;
            for (size_t c0 = 0; c0 < off.size(); c0 += 64) {
                const size_t c1 = std::min(off.size(), c0 + 64) - 1;
                n = std::max(n, (int)(((unsigned)off[c1] + winBytes - ((unsigned)off[c0] & ~15u) + 1023u) >> 10));
            }
            return n;
        };
        t.nld = std::max(pieces16(t.offL, 4 * P), pieces16(t.offC, semi16 ? 8 * P : 4 * P));
        if (t.nld > kLineNld16) return 0;
    } else {
    const bool nv12 = p.srcFormat == GMAT_PIX_FMT_NV12;
    // dwords a lane reads from a row's image at once: as many as its neighbour's window starts further on (a byte plane's lanes are srcW / dstW
    // bytes apart); the windows start on that boundary: samples 4 RW of a byte plane, 2 RU of the interleaved chroma plane (two bytes a sample)
    const int RW = p.srcW >= 16 * p.dstW ? 4 : p.srcW >= 8 * p.dstW ? 2 : 1, RU = RW == 1 ? 2 : 4;
    const int alignB = 4 * RW, alignU = 2 * RU;
    int need = (p.hLum.taps + alignB - 1 + 1) / 2;
    need = std::max(need, (p.hChr.taps + (nv12 ? alignU : alignB) - 1 + 1) / 2);
    int P = 0;
    for (int v : kLineInstances) if (v >= need && (v / 2) % RW == 0) { P = v; break; }
    if (!P) return 0;
    t.P = P; t.RW = RW;
    t.pitchL = align_up(p.dstW, 64); t.pitchC = align_up(p.chrDstW, 64);
    rebase(p.hLum, P, t.pitchL, alignB, 1, t.hL, t.offL);
    if (nv12) rebase(p.hChr, P, t.pitchC, alignU, 2, t.hC, t.offC);
    else      rebase(p.hChr, P, t.pitchC, alignB, 1, t.hC, t.offC);
    // windows start in column order (the kernel's edge test reads the wave's last lane)
    for (size_t i = 1; i < t.offL.size(); i++) if (t.offL[i] < t.offL[i - 1]) return 0;
    for (size_t i = 1; i < t.offC.size(); i++) if (t.offC[i] < t.offC[i - 1]) return 0;
    // 1 KB pieces of a row a wave's 64 windows span, at most (LDS: two rows of them a wave)
    auto pieces = [&](const std::vector<int32_t> &off, int winBytes) {
        int n = 1;
        for (size_t c0 = 0; c0 < off.size(); c0 += 64) {
            const size_t c1 = std::min(off.size(), c0 + 64) - 1;
            n = std::max(n, (int)(((unsigned)off[c1] + winBytes - ((unsigned)off[c0] & ~15u) + 1023u) >> 10));
        }
        return n;
    };
    t.nld = std::max(pieces(t.offL, 2 * P), pieces(t.offC, nv12 ? 4 * P : 2 * P));
    if (t.nld > (RW == 1 ? 2 : RW == 2 ? 3 : kLineNld)) return 0;     // (the instances: RW 1 | 2 | 4 with 2 | 3 | 5 pieces)
    // the signed-byte form of a byte plane's table: c = 256 ch + cl with both halves in [-128, 127], the samples biased by -128 where the wave stores its
    // row image — sum(s c) = 256 sum(s' ch) + sum(s' cl) + 128 sum(c): two v_dot4c_i32_i8 a dword instead of two v_perm_b32 + two v_dot2.  Slot 2 j of a
    // column: the ch bytes of taps 4 j .. 4 j + 3, slot 2 j + 1 their cl bytes; row P: 128 sum(c).  Bit-exact (201 GPU cases); a third fewer VALU
    // instructions in the luma items and the SAME time at 8 : 1 and 12 : 1 (pass H is not bound by its arithmetic: FINDINGS R4-lines) — it pays for the longest
    // filters only (32 frames a launch, 4K -> 160 x 90: 5.24 -> 4.98 us a frame; -> 480 x 270 4.77 -> 4.85): from 40 pairs on.  GMAT_LINES_DOT4=0 | 1: never | always
    const char *d4 = GMAT_KNOB("GMAT_LINES_DOT4");
    auto to_dot4 = [&](std::vector<int32_t> &tab, int pitch, int count) {
        std::vector<int32_t> out((size_t)(P + 1) * pitch, 0);
        for (int x = 0; x < count; x++) {
            long sum = 0;
            for (int j = 0; j < P / 2; j++) {
                const uint32_t a = (uint32_t)tab[(size_t)(2 * j) * pitch + x], b = (uint32_t)tab[(size_t)(2 * j + 1) * pitch + x];
                const int c[4] = {(int16_t)(a & 0xFFFF), (int16_t)(a >> 16), (int16_t)(b & 0xFFFF), (int16_t)(b >> 16)};
                uint32_t hi = 0, lo = 0;
                for (int i = 0; i < 4; i++) {
                    const int ch = (c[i] + 128) >> 8, cl = c[i] - 256 * ch;
                    if (ch < -128 || ch > 127) return false;
                    hi |= (uint32_t)(uint8_t)ch << (8 * i); lo |= (uint32_t)(uint8_t)cl << (8 * i);
                    sum += c[i];
                }
                out[(size_t)(2 * j) * pitch + x] = (int32_t)hi; out[(size_t)(2 * j + 1) * pitch + x] = (int32_t)lo;
            }
            if (128 * sum > 0x3FFFFFFF || 128 * sum < -0x3FFFFFFF) return false;
            out[(size_t)P * pitch + x] = (int32_t)(128 * sum);
        }
        tab.swap(out);
        return true;
    };
    if (d4 ? atoi(d4) != 0 : P >= 40) {
        t.dot4L = to_dot4(t.hL, t.pitchL, p.dstW) ? 1 : 0;
        if (!nv12) t.dot4C = to_dot4(t.hC, t.pitchC, p.chrDstW) ? 1 : 0;
    }
    }
    t.yuvOut = g.yuvOut; t.fullChroma = g.fullChroma;
    t.pairRowsL = (p.srcH + 1) / 2; t.pairRowsC = (p.chrSrcH + 1) / 2;
    // rows the frame holds: a vertical window's padded pairs may end one row pair past the plane (zero coefficients: the rows are never written)
    int rowsL = t.pairRowsL, rowsC = t.pairRowsC;
    for (int i = 0; i < g.vLumEff.count; i++) rowsL = std::max(rowsL, g.vLumEff.pos_even[i] / 2 + g.vLumEff.pairs);
    for (int i = 0; i < g.vChrEff.count; i++) rowsC = std::max(rowsC, g.vChrEff.pos_even[i] / 2 + g.vChrEff.pairs);
    t.baseU = (size_t)rowsL * t.pitchL;
    t.baseV = t.baseU + (size_t)rowsC * t.pitchC;
    t.frameInts = t.baseV + (size_t)rowsC * t.pitchC;
    t.ok = 1;
    return 0;
}

int launch_scale_yuvl(const YuvLArgs &a0, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (!frames || nframes < 1 || nframes > kYuv2xMaxFrames || !a0.inter || a0.nld < 1 || a0.nld > (a0.src16 ? kLineNld16 : a0.RW == 1 ? 2 : a0.RW == 2 ? 3 : kLineNld)) return GMAT_ERR(EINVAL);
    YuvLArgs a = a0;
    const Yuv2xFrames &fr = *frames;
    a.nColL = a.pitchL / 64; a.nColC = a.pitchC / 64;
    // row pairs a wave: long runs amortise the lane's coefficient loads, short ones fill the chip
    const char *rpStr = GMAT_KNOB("GMAT_LINES_RP");
    int rp = rpStr ? std::max(1, atoi(rpStr)) : 8;
    // (measured, profiles/r04_lines.txt: 32 frames a launch 8 / 16 / 32 pairs 5.0 / 5.2 / 5.7 us a 4K frame; one frame a launch 1 / 2 / 4 / 8 pairs 21.0 / 21.5 / 21.7 / 25.1)
    // (round 4, after the per-call event was gone: one 4K frame -> 480 x 270 at 1 / 2 / 3 / 4 / 8 pairs 14.1 / 13.6 / 13.5 / 14.2 / 17.1, -> 160 x 90 16.9 / 17.1 / 16.8 / 18.4 / 23.5)
    if (!rpStr) while (rp > 1 && (long)(a.nColL + a.nColC) * ((a.pairRowsL + rp - 1) / rp) * nframes < 4096) rp >>= 1;
    a.rp = rp;
    a.nItemL = a.nColL * ((a.pairRowsL + rp - 1) / rp);
    a.nItemC = a.nColC * ((a.pairRowsC + rp - 1) / rp);
    a.nItem = a.nItemL + ((a.src16 ? a.src16 < 17 : a.nv12) ? 1 : 2) * a.nItemC;
    if (a.src16) {
        const dim3 grid((a.nItem + 3) / 4, nframes), block(256);
        const size_t lds = (size_t)4 * a.nld * 1024;
        switch (a.P) {
#define GMAT_LH16(P_) case P_: hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvl_h16_kernel<P_>), grid, block, lds, stream, a, fr); break
        GMAT_LH16(8); GMAT_LH16(12); GMAT_LH16(16); GMAT_LH16(20); GMAT_LH16(24); GMAT_LH16(32); GMAT_LH16(40); GMAT_LH16(48); GMAT_LH16(56); GMAT_LH16(72);
#undef GMAT_LH16
        default: return GMAT_ERR(EINVAL);
        }
        GMAT_HIP_CHECK(hipGetLastError());
    } else {
        const dim3 grid((a.nItem + 3) / 4, nframes), block(256);
        const size_t lds = (size_t)4 * 2 * a.nld * 1024;
#define GMAT_LH(P_) case P_: if (a.RW == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvl_h_kernel<P_, 1, 2>), grid, block, lds, stream, a, fr); \
                     else if (a.RW == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvl_h_kernel<P_, 2, 3>), grid, block, lds, stream, a, fr); \
                     else hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvl_h_kernel<P_, 4, kLineNld>), grid, block, lds, stream, a, fr); break
        switch (a.P) {
        GMAT_LH(8); GMAT_LH(12); GMAT_LH(16); GMAT_LH(20); GMAT_LH(24); GMAT_LH(32); GMAT_LH(40); GMAT_LH(48); GMAT_LH(56); GMAT_LH(72);
        default: return GMAT_ERR(EINVAL);
        }
#undef GMAT_LH
        GMAT_HIP_CHECK(hipGetLastError());
    }
    {
        a.nColV = (a.dstW + 255) / 256;
        const dim3 grid(a.nColV * ((a.dstH + 3) / 4), nframes), block(256);
        const int mode = a.yuvOut == 2 ? 3 : a.yuvOut ? 2 : a.fullChroma ? 1 : 0;
        switch (mode) {
        case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvl_v_kernel<0>), grid, block, 0, stream, a, fr); break;
        case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvl_v_kernel<1>), grid, block, 0, stream, a, fr); break;
        case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvl_v_kernel<2>), grid, block, 0, stream, a, fr); break;
        default: hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvl_v_kernel<3>), grid, block, 0, stream, a, fr); break;
        }
        GMAT_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

} // namespace gmat
