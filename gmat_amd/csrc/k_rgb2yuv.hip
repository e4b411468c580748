// k_rgb2yuv.hip — packed RGB -> YUV 4:2:0 (same size) and NV12 <-> YUV420P re-layouts for gfx950.
//
// Replaces rgb2yuv_cuda -> color2nv12 / color2yuv420 (libswscale/cuda/yuv2rgb_cuda.cu:909-947,671-739) and
// yuv2yuv_cuda (libswscale/cuda/yuv2yuv_cuda.cu:320-371; always returns -1 in the reference) with libswscale's
// CPU arithmetic for an RGB24 -> NV12/YUV420P context of equal size (generic path):
//   luma    rgb24ToY_c                      input.c:815-828   Y14 = (ry*r+gy*g+by*b+(32<<14)+(1<<8))>>9
//           hScale16To15_c with one tap     swscale.c:93-119  min(2*Y14, 32767)
//           yuv2plane1_8_c                  output.c:400-408  clip_u8((v + 64) >> 7)
//   chroma  rgb24ToUV_half_c                input.c:849-866   pair-summed pixels, >>10
//           vertical filter (8 taps for the default bicubic 2:1) + yuv2planeX_8_c / yuv2nv12cX_c
//                                           output.c:385-430  clip_u8(((64<<12) + sum) >> 19)
// One block = 128 x 32 luma pixels: phase 1 reads the 38-row source window as 12-byte groups, writes Y
// for the 32 central rows straight to HBM and the half-resolution 15-bit chroma of all rows to LDS;
// phase 2 runs the vertical chroma taps (v_dot2c_i32_i16 over row pairs) and stores U/V.
// Traffic: 3 B/px in, 1.5 B/px out.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

namespace gmat {

constexpr int R2Y_TW = 128, R2Y_TH = 32, R2Y_CW = R2Y_TW / 2, R2Y_CH = R2Y_TH / 2;

struct Rgb2YuvArgs {
    const uint8_t *src; int ss, bgr, srcAligned;
    int toJpeg;                          // lum/chrRangeToJpeg_c on the 15-bit values before the output stage
    uint8_t *y, *u, *v; int ys, us, vs, nv12;
    int w, h, cw, ch;
    DevFilter vChr;                      // chroma vertical filter over SOURCE rows, count = ch
    const int32_t *rowStart, *rowCount;  // per tile row: source row window of the chroma taps (even start)
    int maxRows;
    Rgb2YuvConsts k;
};

__global__ __launch_bounds__(256) void rgb2yuv420_kernel(Rgb2YuvArgs a)
{
    HIP_DYNAMIC_SHARED(uint4, lds_base)
    int *hu = reinterpret_cast<int *>(lds_base);            // [maxRows/2][64] row-pair interleaved int16
    int *hv = hu + (a.maxRows >> 1) * R2Y_CW;
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * R2Y_TW, y0 = blockIdx.y * R2Y_TH;
    const int r0 = a.rowStart[blockIdx.y], nr = a.rowCount[blockIdx.y];   // chroma window rows (even r0)
    // the luma rows [y0, y0+TH) are always inside [r0, r0+nr) for a 2:1 vertical chroma filter; when they
    // are not (other flags) the missing rows are handled by the loop bounds below
    const int lo = min(r0, y0), hiRow = max(r0 + nr, min(y0 + R2Y_TH, a.h));
    const int rows = hiRow - lo;
    short *hu16 = reinterpret_cast<short *>(hu), *hv16 = reinterpret_cast<short *>(hv);

    // ---- phase 1: units of 4 pixels x 1 row; 32 groups per row ----
    const int total = rows * 32;
    for (int g = tid; g < total; g += 256) {
        const int rr = g >> 5, cg = g & 31;
        const int srow = lo + rr;
        const int sr = min(max(srow, 0), a.h - 1);
        const int col = x0 + 4 * cg;
        if (col >= a.w) continue;
        const uint8_t *row = a.src + (size_t)sr * a.ss;
        int r[4], gg[4], b[4];
        if (a.srcAligned && col + 4 <= a.w) {
            const uint3 v = *reinterpret_cast<const uint3 *>(row + (size_t)col * 3);
            r[0] = v.x & 0xFF;         gg[0] = (v.x >> 8) & 0xFF;  b[0] = (v.x >> 16) & 0xFF;
            r[1] = v.x >> 24;          gg[1] = v.y & 0xFF;         b[1] = (v.y >> 8) & 0xFF;
            r[2] = (v.y >> 16) & 0xFF; gg[2] = v.y >> 24;          b[2] = v.z & 0xFF;
            r[3] = (v.z >> 8) & 0xFF;  gg[3] = (v.z >> 16) & 0xFF; b[3] = v.z >> 24;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int c = min(col + i, a.w - 1);
                r[i] = row[3 * c]; gg[i] = row[3 * c + 1]; b[i] = row[3 * c + 2];
            }
        }
        if (a.bgr) {
#pragma unroll
            for (int i = 0; i < 4; i++) { const int t = r[i]; r[i] = b[i]; b[i] = t; }
        }
        // luma of the tile's own rows
        if (srow >= y0 && srow < y0 + R2Y_TH && srow < a.h && col < a.w) {
            unsigned yb = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int y14 = rgb_to_y14(a.k, r[i], gg[i], b[i]);
                int l = min(2 * y14, 32767);
                if (a.toJpeg) l = (m24(min(l, 30189), 19077) - 39057361) >> 14;          // lumRangeToJpeg_c, swscale.c:176-181
                yb |= (unsigned)clip_u8_shr(l + 64, 7) << (8 * i);
            }
            uint8_t *d = a.y + (size_t)srow * a.ys + col;
            if (col + 4 <= a.w && ((((uintptr_t)a.y | (uintptr_t)a.ys) & 3) == 0)) *reinterpret_cast<unsigned *>(d) = yb;
            else for (int i = 0; i < min(4, a.w - col); i++) d[i] = (uint8_t)(yb >> (8 * i));
        }
        // chroma of the window rows
        const int wr = srow - r0;
        if (wr >= 0 && wr < nr) {
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int rs = r[2 * i] + r[2 * i + 1], gs = gg[2 * i] + gg[2 * i + 1], bs = b[2 * i] + b[2 * i + 1];
                const int o = (((wr >> 1) * R2Y_CW + 2 * cg + i) << 1) + (wr & 1);
                int cu = min(2 * rgbsum_to_u14(a.k, rs, gs, bs), 32767), cv = min(2 * rgbsum_to_v14(a.k, rs, gs, bs), 32767);
                if (a.toJpeg) {                                                              // chrRangeToJpeg_c, swscale.c:157-164
                    cu = (m24(min(cu, 30775), 4663) - 9289992) >> 12; cv = (m24(min(cv, 30775), 4663) - 9289992) >> 12;
                }
                hu16[o] = (short)cu;
                hv16[o] = (short)cv;
            }
        }
    }
    __syncthreads();

    // ---- phase 2: vertical chroma taps, 4 chroma samples per thread ----
    {
        const int q = tid & 15, yl = tid >> 4;                 // 16 groups x 16 chroma rows
        const int cy = (y0 >> 1) + yl, cx = (x0 >> 1) + 4 * q;
        if (cy < a.ch && cx < a.cw) {
            const int vp = (a.vChr.pos_even[cy] - r0) >> 1;
            int U[4], V[4];
#pragma unroll
            for (int i = 0; i < 4; i++) U[i] = V[i] = a.vChr.round[cy];
            for (int k = 0; k < a.vChr.pairs; k++) {
                const int cf = a.vChr.packed[(size_t)cy * a.vChr.pairs + k];
                const int4 u = *reinterpret_cast<const int4 *>(hu + (vp + k) * R2Y_CW + 4 * q);
                const int4 v = *reinterpret_cast<const int4 *>(hv + (vp + k) * R2Y_CW + 4 * q);
                U[0] = dot2(u.x, cf, U[0]); U[1] = dot2(u.y, cf, U[1]); U[2] = dot2(u.z, cf, U[2]); U[3] = dot2(u.w, cf, U[3]);
                V[0] = dot2(v.x, cf, V[0]); V[1] = dot2(v.y, cf, V[1]); V[2] = dot2(v.z, cf, V[2]); V[3] = dot2(v.w, cf, V[3]);
            }
            // one tap of 4096: ((64<<12) + 4096*s) >> 19 == (s + 64) >> 7, i.e. yuv2plane1_8_c falls out of the X form
            const int sh = 19;
            const int n = min(4, a.cw - cx);
            unsigned ub[4], vb[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { ub[i] = (unsigned)clip_u8_shr(U[i], sh); vb[i] = (unsigned)clip_u8_shr(V[i], sh); }
            if (a.nv12) {
                uint8_t *d = a.u + (size_t)cy * a.us + 2 * cx;
                if (n == 4 && ((((uintptr_t)a.u | (uintptr_t)a.us) & 7) == 0)) {
                    *reinterpret_cast<uint2 *>(d) = make_uint2(ub[0] | (vb[0] << 8) | (ub[1] << 16) | (vb[1] << 24),
                                                               ub[2] | (vb[2] << 8) | (ub[3] << 16) | (vb[3] << 24));
                } else {
                    for (int i = 0; i < n; i++) { d[2 * i] = (uint8_t)ub[i]; d[2 * i + 1] = (uint8_t)vb[i]; }
                }
            } else {
                uint8_t *du = a.u + (size_t)cy * a.us + cx, *dv = a.v + (size_t)cy * a.vs + cx;
                if (n == 4 && ((((uintptr_t)a.u | (uintptr_t)a.us | (uintptr_t)a.v | (uintptr_t)a.vs) & 3) == 0)) {
                    *reinterpret_cast<unsigned *>(du) = ub[0] | (ub[1] << 8) | (ub[2] << 16) | (ub[3] << 24);
                    *reinterpret_cast<unsigned *>(dv) = vb[0] | (vb[1] << 8) | (vb[2] << 16) | (vb[3] << 24);
                } else {
                    for (int i = 0; i < n; i++) { du[i] = (uint8_t)ub[i]; dv[i] = (uint8_t)vb[i]; }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// rgb2yuv420s_kernel: the strip-walking form of the same conversion (round 2).  The tiled kernel above re-converts the 6 halo rows
// of every 32-row tile and passes the chroma through LDS: 14.7 us per 4K frame (0.32 of the HBM roofline).  Here a wave owns a strip
// of 512 pixel columns and walks down the source rows in pairs; a lane loads its own 8 pixels of each row (24 bytes, two loads),
// writes their luma at once (8 bytes) and keeps the pair-summed 15-bit chroma of the last four row pairs (2m-1 | 2m) in registers for
// the 2:1 vertical filter (8 taps as 4 int16 pairs, kernel arguments; borders = the interior filter on an edge-replicated plane,
// checked on the host with filter_is_edge_replication).  No horizontal filter exists on this path (identity: min(2 v, 32767)), so
// nothing crosses lanes.  A segment of `segRows` chroma rows re-creates its vertical window with 3 warm-up row pairs whose luma
// belongs to the neighbour and is neither computed nor written.
constexpr int Y2S_STRIP = 512;

#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned y2s_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned y2s_u32x2 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ uint4 y2s_ld16(const uint8_t *p) { const y2s_u32x4 v = *reinterpret_cast<const y2s_u32x4 *>(p); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint2 y2s_ld8(const uint8_t *p) { const y2s_u32x2 v = *reinterpret_cast<const y2s_u32x2 *>(p); return make_uint2(v.x, v.y); }
__device__ __forceinline__ void y2s_st8(uint8_t *p, unsigned lo, unsigned hi) { st_stream(p, make_uint2(lo, hi)); }
#else
static inline uint4 y2s_ld16(const uint8_t *p) { uint4 v; std::memcpy(&v, p, 16); return v; }
static inline uint2 y2s_ld8(const uint8_t *p) { uint2 v; std::memcpy(&v, p, 8); return v; }
static inline void y2s_st8(uint8_t *p, unsigned lo, unsigned hi) { std::memcpy(p, &lo, 4); std::memcpy(p + 4, &hi, 4); }
#endif
__device__ __forceinline__ int y2s_dot2(int ab, int cd, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, ab), __builtin_bit_cast(short2v, cd), acc, true);   // VOP3P form, see k_scale_yuv2s.hip
}

struct Rgb2YuvStripArgs {                        // plane pointers: Yuv2xFrames (y[] = the packed source, dst / dstU / dstV = the planes), blockIdx.y = frame
    int ss, toJpeg;
    int ys, us, vs;
    int w, h;
    int32_t cY01, cY2, cU01, cU2, cV01, cV2;     // coefficients in the byte order of the pixels: (first, second) as an int16 pair, third alone
    int32_t vC[4];                               // the vertical chroma filter on [2y - 3, 2y + 4] as 4 int16 pairs
    int rnd;
    int segRows, nseg, nstrips, nblk, xcdRemap;
};
struct Y2sRow { unsigned d[8]; };                 // eight pixels: 24 bytes (six dwords) or — PX = 4, RGBA / BGRA read as they are (round 6) — 32

// JPEG: full-range destination (lum / chrRangeToJpeg_c) — a template parameter: as a run-time flag the compiler made it a scalar branch
// per PIXEL (26 branches per iteration; the first build ran at 16 us per 4K frame, slower than the tiled kernel)
// NOSAT: the host has verified from the coefficients that no 8-bit input can reach the saturations of the limited-range path — the 14-bit
// luma stays below 16352 and the 14-bit chroma below 16384 (true for every matrix of fill_rgb2yuv_table: the luma row sums to
// 219/255 * 2^15) — so min(2 v, 32767) and the 8-bit clip of the one-tap luma output are dead and ((2 y + 64) >> 7) = (y + 32) >> 6.
// PX: bytes a source pixel (3; 4: RGBA / BGRA sources, rgb32ToY / ToUV read the same three channels — round 6: such a context ran a 32 -> 24-bit pass in
// front of this kernel, two launches a frame and no batches: 13.1 us a 1080p frame where RGB24 takes 2.2, profiles/r06_sweep_before.txt)
template <bool NV, bool JPEG, bool NOSAT, int PX>
__global__ __launch_bounds__(256) void rgb2yuv420s_kernel(Rgb2YuvStripArgs a, Yuv2xFrames fr)
{
    const uint8_t *psrc = fr.y[blockIdx.y];
    uint8_t *py = fr.dst[blockIdx.y], *pu = fr.dstU[blockIdx.y], *pv = fr.dstV[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    int lin = blockIdx.x;
    if (a.xcdRemap) {
        const int chunk = (a.nblk + 7) >> 3;
        lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    }
    if (lin >= a.nblk) return;
    const int unit = lin * 4 + wave;                            // (segment, strip) units packed densely: the waves share nothing
    if (unit >= a.nseg * a.nstrips) return;
    const int seg = __builtin_amdgcn_readfirstlane(unit / a.nstrips);
    const int X0 = (unit - seg * a.nstrips) * Y2S_STRIP;
    const int c0 = seg * a.segRows, ch = a.h >> 1;
    const int nOut = min(a.segRows, ch - c0);
    const int nIter = nOut + 3;                                 // 3 warm-up row pairs fill the vertical window

    const int xo = X0 + 8 * lane;
    const bool active = xo < a.w;
    const unsigned xc = (unsigned)(active ? xo : a.w - 8);      // idle lanes shadow the last group (loads stay inside the rows)

    auto load_row = [&](int r, Y2sRow &R) {
        const uint8_t *p = psrc + ((unsigned)min(max(r, 0), a.h - 1) * (unsigned)a.ss + (unsigned)PX * xc);
        const uint4 v0 = y2s_ld16(p);
        R.d[0] = v0.x; R.d[1] = v0.y; R.d[2] = v0.z; R.d[3] = v0.w;
        if (PX == 4) { const uint4 v1 = y2s_ld16(p + 16); R.d[4] = v1.x; R.d[5] = v1.y; R.d[6] = v1.z; R.d[7] = v1.w; }
        else         { const uint2 v1 = y2s_ld8(p + 16); R.d[4] = v1.x; R.d[5] = v1.y; }
    };
    // one source row: luma of the 8 pixels written if the row belongs to this segment; 15-bit U / V of the 4 pixel pairs returned
    auto convert_row = [&](const Y2sRow &R, int row, bool luma, int (&cu)[4], int (&cv)[4]) {
        int fs[8], th[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int o = PX * i, d = o >> 2, b = o & 3;
            const unsigned lo = R.d[d], hi = R.d[PX == 4 ? d : d + 1 < 6 ? d + 1 : d];
            fs[i] = (int)__builtin_amdgcn_perm(hi, lo, 0x0C000C00u | (unsigned)b | ((unsigned)(b + 1) << 16));
            th[i] = (int)__builtin_amdgcn_perm(hi, lo, 0x0C0C0C00u | (unsigned)(b + 2));
        }
        if (luma) {                                             // wave-uniform
            unsigned yb[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                // rgb24ToY_c, hScale16To15_c with one tap, (lumRangeToJpeg_c), yuv2plane1_8_c
                const int y14 = y2s_dot2(fs[i], a.cY01, m24(th[i], a.cY2) + ((32 << 14) + (1 << 8))) >> 9;
                if constexpr (NOSAT && !JPEG) {
                    yb[i] = (unsigned)(y14 + 32) >> 6;
                } else {
                    int l = min(2 * y14, 32767);
                    if constexpr (JPEG) l = (m24(min(l, 30189), 19077) - 39057361) >> 14;
                    yb[i] = (unsigned)clip_u8_shr(l + 64, 7);
                }
            }
            if (active) y2s_st8(py + ((unsigned)row * (unsigned)a.ys + (unsigned)xo), yb[0] | (yb[1] << 8) | (yb[2] << 16) | (yb[3] << 24),
                                yb[4] | (yb[5] << 8) | (yb[6] << 16) | (yb[7] << 24));
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            // rgb24ToUV_half_c on the sum of the pair's pixels, hScale16To15_c with one tap, (chrRangeToJpeg_c)
            const int fsum = fs[2 * c] + fs[2 * c + 1];             // two 9-bit sums in the halves: no carry across
            const int tsum = th[2 * c] + th[2 * c + 1];
            int u = 2 * (y2s_dot2(fsum, a.cU01, m24(tsum, a.cU2) + ((256 << 15) + (1 << 9))) >> 10);
            int v = 2 * (y2s_dot2(fsum, a.cV01, m24(tsum, a.cV2) + ((256 << 15) + (1 << 9))) >> 10);
            if constexpr (!NOSAT) { u = min(u, 32767); v = min(v, 32767); }
            if constexpr (JPEG) { u = (m24(min(u, 30775), 4663) - 9289992) >> 12; v = (m24(min(v, 30775), 4663) - 9289992) >> 12; }
            cu[c] = u; cv[c] = v;
        }
    };

    int hwU[4][4], hwV[4][4];                                   // [slot][pixel pair]: (row 2m-1 | row 2m << 16)
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int c = 0; c < 4; c++) hwU[s][c] = hwV[s][c] = 0;
    Y2sRow bufA[2], bufB[2];                                    // ping-pong: rows 2m-1 and 2m of the current / next pair
#pragma unroll
    for (int i = 0; i < 8; i++) bufA[1].d[i] = bufB[1].d[i] = 0u;
    load_row(2 * (c0 - 1) - 1, bufA[0]);
    load_row(2 * (c0 - 1), bufB[0]);

    auto body = [&](const int j, auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value;           // j & 3, static after unrolling
        const Y2sRow ra = bufA[SLOT & 1], rb = bufB[SLOT & 1];
        const int m = c0 - 1 + j;                               // this iteration's pair: rows 2m - 1, 2m
        if (j + 1 < nIter) {
            load_row(2 * m + 1, bufA[(SLOT + 1) & 1]);
            load_row(2 * m + 2, bufB[(SLOT + 1) & 1]);
        }
        {
            // row 2m - 1 belongs to chroma row m - 1, row 2m to chroma row m: luma only for the rows of this segment
            int ua[4], va[4], ub[4], vb[4];
            convert_row(ra, 2 * m - 1, m - 1 >= c0 && m - 1 < c0 + nOut, ua, va);
            convert_row(rb, 2 * m, m >= c0 && m < c0 + nOut, ub, vb);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                hwU[SLOT][c] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(ua[c], ub[c]));
                hwV[SLOT][c] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(va[c], vb[c]));
            }
        }
        if (j >= 3) {
            const int cy = c0 + j - 3;                          // pairs cy-1 .. cy+2 sit in slots SLOT+1 .. SLOT+4 (mod 4)
            unsigned ub8[4], vb8[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                int U = a.rnd, V = a.rnd;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    U = y2s_dot2(hwU[(SLOT + 1 + k) & 3][c], a.vC[k], U);
                    V = y2s_dot2(hwV[(SLOT + 1 + k) & 3][c], a.vC[k], V);
                }
                ub8[c] = (unsigned)clip_u8_shr(U, 19); vb8[c] = (unsigned)clip_u8_shr(V, 19);
            }
            if (active) {
                if (NV) {
                    y2s_st8(pu + ((unsigned)cy * (unsigned)a.us + (unsigned)xo), ub8[0] | (vb8[0] << 8) | (ub8[1] << 16) | (vb8[1] << 24),
                            ub8[2] | (vb8[2] << 8) | (ub8[3] << 16) | (vb8[3] << 24));
                } else {
                    st_stream(pu + ((unsigned)cy * (unsigned)a.us + (unsigned)(xo >> 1)), (unsigned)(ub8[0] | (ub8[1] << 8) | (ub8[2] << 16) | (ub8[3] << 24)));
                    st_stream(pv + ((unsigned)cy * (unsigned)a.vs + (unsigned)(xo >> 1)), (unsigned)(vb8[0] | (vb8[1] << 8) | (vb8[2] << 16) | (vb8[3] << 24)));
                }
            }
        }
    };
    for (int j0 = 0; j0 < nIter; j0 += 4) {
        body(j0, std::integral_constant<int, 0>());
        if (j0 + 1 < nIter) body(j0 + 1, std::integral_constant<int, 1>());
        if (j0 + 2 < nIter) body(j0 + 2, std::integral_constant<int, 2>());
        if (j0 + 3 < nIter) body(j0 + 3, std::integral_constant<int, 3>());
    }
}

// the strip form takes a frame when the chroma taps are the replicated 8-tap window, the geometry is whole lanes and rows pair up,
// and every plane can be moved in dwords
bool rgb2yuv420_strip_takes(const Rgb2YuvLaunch &L)
{
    const char *off = GMAT_KNOB("GMAT_SCALE_NO_STRIP");
    if (off && atoi(off)) return false;
    if (!L.stripOk || L.w % 8 || L.w < 64 || (L.h & 1) || L.h < 16) return false;
    uintptr_t all = (uintptr_t)L.src | (uintptr_t)L.ss | (uintptr_t)L.y | (uintptr_t)L.ys | (uintptr_t)L.u | (uintptr_t)L.us;
    if (!L.nv12) all |= (uintptr_t)L.v | (uintptr_t)L.vs;
    return (all & 3) == 0;
}

// frames == nullptr: the one frame described by L; else nframes frames of L's geometry and strides (all of them pass rgb2yuv420_strip_takes)
int launch_rgb2yuv420s(const Rgb2YuvLaunch &L, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (nframes < 1 || nframes > kYuv2xMaxFrames) return GMAT_ERR(EINVAL);
    Yuv2xFrames one;
    if (!frames) {
        std::memset(&one, 0, sizeof(one));
        one.y[0] = L.src; one.dst[0] = L.y; one.dstU[0] = L.u; one.dstV[0] = L.v;
        frames = &one; nframes = 1;
    }
    Rgb2YuvStripArgs a;
    std::memset(&a, 0, sizeof(a));
    a.ss = L.ss; a.toJpeg = L.toJpeg;
    a.ys = L.ys; a.us = L.us; a.vs = L.vs; a.w = L.w; a.h = L.h;
    auto pk = [](int lo, int hi) { return (int32_t)((uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16)); };
    const Rgb2YuvConsts &q = L.k;
    if (L.bgr) { a.cY01 = pk(q.by, q.gy); a.cY2 = q.ry; a.cU01 = pk(q.bu, q.gu); a.cU2 = q.ru; a.cV01 = pk(q.bv, q.gv); a.cV2 = q.rv; }
    else       { a.cY01 = pk(q.ry, q.gy); a.cY2 = q.by; a.cU01 = pk(q.ru, q.gu); a.cU2 = q.bu; a.cV01 = pk(q.rv, q.gv); a.cV2 = q.bv; }
    for (int k = 0; k < 4; k++) a.vC[k] = L.vC[k];
    a.rnd = 64 << 12;                                           // yuv2planeX_8_c / yuv2nv12cX_c dither
    a.nstrips = (L.w + Y2S_STRIP - 1) / Y2S_STRIP;
    const char *segStr = GMAT_KNOB("GMAT_STRIP_ROWS");              // tuning / test override (chroma rows per segment), read per launch
    int seg = segStr ? atoi(segStr) : 0;
    if (seg <= 0) {
        // measured on one 4K frame per launch (profiles/r02o_rgb2yuv_strip.txt): 5 chroma rows 11.9 us, 3: 12.2, 4: 13.3, 6: 13.0, 8: 14.8
        // — an odd count makes the iteration count (rows + 3) a multiple of the unrolled loop's period more often; 1080p: 3 rows
        const long rows = (long)(L.h >> 1) * a.nstrips * nframes;   // wave-rows (chroma)
        // (round 4, 32 frames a launch, 4K: 6 ... 12 rows 219-222 us a launch, 14: 226, 16: 229, 24: 237, 31 — the old cap — 232: profiles/r04_rows_all.txt)
        seg = (int)std::min(9L, std::max(3L, (rows + 1727) / 1728)) | 1;      // (1080p: 9 rows 60.5 us a 32-frame launch, 11: 64, 31: 67)
    }
    a.segRows = seg;
    a.nseg = ((L.h >> 1) + seg - 1) / seg;
    a.nblk = (a.nseg * a.nstrips + 3) / 4;
    a.xcdRemap = 1;
    const dim3 grid(8 * ((a.nblk + 7) / 8), nframes), block(256);
    // can an 8-bit pixel reach a saturation?  luma: 0 <= y14 <= 16351 keeps 2 y14 + 64 below 2^15; chroma (pixel PAIRS): 0 <= u14 <= 16383
    auto lo_hi = [](int c0, int c1, int c2, long scale, long add, int sh, long &lo, long &hi) {
        lo = (scale * (std::min(c0, 0) + std::min(c1, 0) + std::min(c2, 0)) + add) >> sh;
        hi = (scale * (std::max(c0, 0) + std::max(c1, 0) + std::max(c2, 0)) + add) >> sh;
    };
    long ylo, yhi, ulo, uhi, vlo, vhi;
    lo_hi(q.ry, q.gy, q.by, 255, (32 << 14) + (1 << 8), 9, ylo, yhi);
    lo_hi(q.ru, q.gu, q.bu, 510, (256L << 15) + (1 << 9), 10, ulo, uhi);
    lo_hi(q.rv, q.gv, q.bv, 510, (256L << 15) + (1 << 9), 10, vlo, vhi);
    const char *ns = GMAT_KNOB("GMAT_R2Y_NOSAT");                   // test knob: 0 = the variant that keeps every saturation
    const bool nosat = !(ns && !atoi(ns)) && ylo >= 0 && yhi <= 16351 && ulo >= 0 && uhi <= 16383 && vlo >= 0 && vhi <= 16383;
#define GMAT_Y2S(NV_, J_) do { if (L.px == 4) { if (nosat) hipLaunchKernelGGL(HIP_KERNEL_NAME(rgb2yuv420s_kernel<NV_, J_, true, 4>), grid, block, 0, stream, a, *frames); \
                                                 else       hipLaunchKernelGGL(HIP_KERNEL_NAME(rgb2yuv420s_kernel<NV_, J_, false, 4>), grid, block, 0, stream, a, *frames); } \
                               else if (nosat) hipLaunchKernelGGL(HIP_KERNEL_NAME(rgb2yuv420s_kernel<NV_, J_, true, 3>), grid, block, 0, stream, a, *frames); \
                               else            hipLaunchKernelGGL(HIP_KERNEL_NAME(rgb2yuv420s_kernel<NV_, J_, false, 3>), grid, block, 0, stream, a, *frames); } while (0)
    if (L.toJpeg) { if (L.nv12) GMAT_Y2S(true, true); else GMAT_Y2S(false, true); }
    else          { if (L.nv12) GMAT_Y2S(true, false); else GMAT_Y2S(false, false); }
#undef GMAT_Y2S
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// planar float RGB (rgbpf32le: three stacked planes, value = u8 / 255) -> 8-bit 4:2:0 in ONE kernel (round 4).  format_cuda's way back from a
// network's output tensor (vf_format_cuda.c:184-217: ... tensorrt, format_cuda=nv12, encode).  Rounds 1-3 quantised the floats into the
// context's RGB24 intermediate (rgbpf32_to_rgb24_kernel) and ran the RGB -> 4:2:0 converter behind it: 162 MB of traffic for a 4K frame whose
// compulsory bytes are 112 MB, two launches, 25.8 us for a 1080p frame.  Here rgb2yuv420s_kernel's walk with the quantisation in its load
// stage: a wave owns a strip of 256 pixel columns, a lane 4 pixels of a row — one 16-byte load per plane, u8 = (int)(clamp(f, 0, 1) * 255 + 0.5)
// exactly as rgbpf32_to_rgb24_kernel computes it, then rgb24ToY_c / rgb24ToUV_half_c and the 8-tap vertical chroma filter operation by operation.
constexpr int F2S_STRIP = 256;
struct F2sRow { float4 p[3]; };
__device__ __forceinline__ int f2s_quant(float f) { return (int)(__fadd_rn(__fmul_rn(fminf(fmaxf(f, 0.0f), 1.0f), 255.0f), 0.5f)); }

template <bool NV, bool JPEG, bool NOSAT>
__global__ __launch_bounds__(256) void pf32_to_yuv420s_kernel(Rgb2YuvStripArgs a, Yuv2xFrames fr)
{
    const uint8_t *psrc = fr.y[blockIdx.y];
    uint8_t *py = fr.dst[blockIdx.y], *pu = fr.dstU[blockIdx.y], *pv = fr.dstV[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    int lin = blockIdx.x;
    if (a.xcdRemap) {
        const int chunk = (a.nblk + 7) >> 3;
        lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    }
    if (lin >= a.nblk) return;
    const int unit = lin * 4 + wave;
    if (unit >= a.nseg * a.nstrips) return;
    const int seg = __builtin_amdgcn_readfirstlane(unit / a.nstrips);
    const int X0 = (unit - seg * a.nstrips) * F2S_STRIP;
    const int c0 = seg * a.segRows, ch = a.h >> 1;
    const int nOut = min(a.segRows, ch - c0);
    const int nIter = nOut + 3;                                 // 3 warm-up row pairs fill the vertical window
    const int xo = X0 + 4 * lane;
    const bool active = xo < a.w;
    const unsigned xc = (unsigned)(active ? xo : a.w - 4);      // idle lanes shadow the last group
    const size_t plane = (size_t)a.ss * a.h;

    auto load_row = [&](int r, F2sRow &R) {
        const uint8_t *p = psrc + ((size_t)min(max(r, 0), a.h - 1) * (unsigned)a.ss + 4u * xc);
#pragma unroll
        for (int k = 0; k < 3; k++) R.p[k] = *reinterpret_cast<const float4 *>(p + k * plane);
    };
    // one source row: luma of the 4 pixels written if the row belongs to this segment; 15-bit U / V of the 2 pixel pairs returned
    auto convert_row = [&](const F2sRow &R, int row, bool luma, int (&cu)[2], int (&cv)[2]) {
        const float f0[4] = {R.p[0].x, R.p[0].y, R.p[0].z, R.p[0].w}, f1[4] = {R.p[1].x, R.p[1].y, R.p[1].z, R.p[1].w}, f2[4] = {R.p[2].x, R.p[2].y, R.p[2].z, R.p[2].w};
        int fs[4], th[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { fs[i] = f2s_quant(f0[i]) | (f2s_quant(f1[i]) << 16); th[i] = f2s_quant(f2[i]); }
        if (luma) {                                             // wave-uniform
            unsigned yb[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                // rgb24ToY_c, hScale16To15_c with one tap, (lumRangeToJpeg_c), yuv2plane1_8_c
                const int y14 = y2s_dot2(fs[i], a.cY01, m24(th[i], a.cY2) + ((32 << 14) + (1 << 8))) >> 9;
                if constexpr (NOSAT && !JPEG) {
                    yb[i] = (unsigned)(y14 + 32) >> 6;
                } else {
                    int l = min(2 * y14, 32767);
                    if constexpr (JPEG) l = (m24(min(l, 30189), 19077) - 39057361) >> 14;
                    yb[i] = (unsigned)clip_u8_shr(l + 64, 7);
                }
            }
            if (active) st_stream(py + ((unsigned)row * (unsigned)a.ys + (unsigned)xo), (unsigned)(yb[0] | (yb[1] << 8) | (yb[2] << 16) | (yb[3] << 24)));
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            // rgb24ToUV_half_c on the sum of the pair's pixels, hScale16To15_c with one tap, (chrRangeToJpeg_c)
            const int fsum = fs[2 * c] + fs[2 * c + 1];             // two 9-bit sums in the halves: no carry across
            const int tsum = th[2 * c] + th[2 * c + 1];
            int u = 2 * (y2s_dot2(fsum, a.cU01, m24(tsum, a.cU2) + ((256 << 15) + (1 << 9))) >> 10);
            int v = 2 * (y2s_dot2(fsum, a.cV01, m24(tsum, a.cV2) + ((256 << 15) + (1 << 9))) >> 10);
            if constexpr (!NOSAT) { u = min(u, 32767); v = min(v, 32767); }
            if constexpr (JPEG) { u = (m24(min(u, 30775), 4663) - 9289992) >> 12; v = (m24(min(v, 30775), 4663) - 9289992) >> 12; }
            cu[c] = u; cv[c] = v;
        }
    };

    int hwU[4][2], hwV[4][2];                                   // [slot][pixel pair]: (row 2m-1 | row 2m << 16)
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int c = 0; c < 2; c++) hwU[s][c] = hwV[s][c] = 0;
    F2sRow bufA[2], bufB[2];                                    // ping-pong: rows 2m-1 and 2m of the current / next pair
#pragma unroll
    for (int k = 0; k < 3; k++) bufA[1].p[k] = bufB[1].p[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    load_row(2 * (c0 - 1) - 1, bufA[0]);
    load_row(2 * (c0 - 1), bufB[0]);

    auto body = [&](const int j, auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value;           // j & 3, static after unrolling
        const F2sRow ra = bufA[SLOT & 1], rb = bufB[SLOT & 1];
        const int m = c0 - 1 + j;                               // this iteration's pair: rows 2m - 1, 2m
        if (j + 1 < nIter) {
            load_row(2 * m + 1, bufA[(SLOT + 1) & 1]);
            load_row(2 * m + 2, bufB[(SLOT + 1) & 1]);
        }
        {
            int ua[2], va[2], ub[2], vb[2];
            convert_row(ra, 2 * m - 1, m - 1 >= c0 && m - 1 < c0 + nOut, ua, va);
            convert_row(rb, 2 * m, m >= c0 && m < c0 + nOut, ub, vb);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                hwU[SLOT][c] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(ua[c], ub[c]));
                hwV[SLOT][c] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(va[c], vb[c]));
            }
        }
        if (j >= 3) {
            const int cy = c0 + j - 3;                          // pairs cy-1 .. cy+2 sit in slots SLOT+1 .. SLOT+4 (mod 4)
            unsigned ub8[2], vb8[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                int U = a.rnd, V = a.rnd;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    U = y2s_dot2(hwU[(SLOT + 1 + k) & 3][c], a.vC[k], U);
                    V = y2s_dot2(hwV[(SLOT + 1 + k) & 3][c], a.vC[k], V);
                }
                ub8[c] = (unsigned)clip_u8_shr(U, 19); vb8[c] = (unsigned)clip_u8_shr(V, 19);
            }
            if (active) {
                if (NV) {
                    st_stream(pu + ((unsigned)cy * (unsigned)a.us + (unsigned)xo), (unsigned)(ub8[0] | (vb8[0] << 8) | (ub8[1] << 16) | (vb8[1] << 24)));
                } else {
                    *reinterpret_cast<uint16_t *>(pu + ((unsigned)cy * (unsigned)a.us + (unsigned)(xo >> 1))) = (uint16_t)(ub8[0] | (ub8[1] << 8));
                    *reinterpret_cast<uint16_t *>(pv + ((unsigned)cy * (unsigned)a.vs + (unsigned)(xo >> 1))) = (uint16_t)(vb8[0] | (vb8[1] << 8));
                }
            }
        }
    };
    for (int j0 = 0; j0 < nIter; j0 += 4) {
        body(j0, std::integral_constant<int, 0>());
        if (j0 + 1 < nIter) body(j0 + 1, std::integral_constant<int, 1>());
        if (j0 + 2 < nIter) body(j0 + 2, std::integral_constant<int, 2>());
        if (j0 + 3 < nIter) body(j0 + 3, std::integral_constant<int, 3>());
    }
}

// the fused form takes a frame when the strip converter would (replicated 8-tap chroma window, rows pair up) and the float rows can be read 16
// bytes at a time; L.src = the first float plane, L.ss = a float row's pitch in bytes (the planes are L.ss * L.h apart)
bool pf32_to_yuv420_strip_takes(const Rgb2YuvLaunch &L)
{
    const char *off = GMAT_KNOB("GMAT_SCALE_NO_STRIP");
    if (off && atoi(off)) return false;
    if (!L.stripOk || L.w % 4 || L.w < 64 || (L.h & 1) || L.h < 16) return false;
    if ((((uintptr_t)L.src | (uintptr_t)L.ss) & 15) != 0) return false;
    uintptr_t all = (uintptr_t)L.y | (uintptr_t)L.ys | (uintptr_t)L.u | (uintptr_t)L.us;
    if (!L.nv12) return (all & 3) == 0 && ((((uintptr_t)L.u | (uintptr_t)L.us | (uintptr_t)L.v | (uintptr_t)L.vs) & 1) == 0);
    return (all & 3) == 0;
}

int launch_pf32_to_yuv420s(const Rgb2YuvLaunch &L, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (nframes < 1 || nframes > kYuv2xMaxFrames) return GMAT_ERR(EINVAL);
    Yuv2xFrames one;
    if (!frames) {
        std::memset(&one, 0, sizeof(one));
        one.y[0] = L.src; one.dst[0] = L.y; one.dstU[0] = L.u; one.dstV[0] = L.v;
        frames = &one; nframes = 1;
    }
    Rgb2YuvStripArgs a;
    std::memset(&a, 0, sizeof(a));
    a.ss = L.ss; a.toJpeg = L.toJpeg;
    a.ys = L.ys; a.us = L.us; a.vs = L.vs; a.w = L.w; a.h = L.h;
    auto pk = [](int lo, int hi) { return (int32_t)((uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16)); };
    const Rgb2YuvConsts &q = L.k;
    a.cY01 = pk(q.ry, q.gy); a.cY2 = q.by; a.cU01 = pk(q.ru, q.gu); a.cU2 = q.bu; a.cV01 = pk(q.rv, q.gv); a.cV2 = q.bv;      // planes in R, G, B order
    for (int k = 0; k < 4; k++) a.vC[k] = L.vC[k];
    a.rnd = 64 << 12;                                           // yuv2planeX_8_c / yuv2nv12cX_c dither
    a.nstrips = (L.w + F2S_STRIP - 1) / F2S_STRIP;
    const char *segStr = GMAT_KNOB("GMAT_STRIP_ROWS");
    int seg = segStr ? atoi(segStr) : 0;
    if (seg <= 0) {
        // measured, 8 frames a launch (profiles/r04_rgbpf32.txt): 3 chroma rows 29.4 us a 4K frame, 5: 24.7, 7: 23.5, 11: 24.6, 15: 24.8, 23: 24.2,
        // 31: 25.1 — short segments sweep the three float planes in raster order; 7 rows at most
        const long rows = (long)(L.h >> 1) * a.nstrips * nframes;   // wave-rows (chroma)
        seg = (int)std::min(7L, std::max(3L, (rows + 3455) / 3456)) | 1;
    }
    a.segRows = seg;
    a.nseg = ((L.h >> 1) + seg - 1) / seg;
    a.nblk = (a.nseg * a.nstrips + 3) / 4;
    a.xcdRemap = 1;
    const dim3 grid(8 * ((a.nblk + 7) / 8), nframes), block(256);
    auto lo_hi = [](int c0, int c1, int c2, long scale, long add, int sh, long &lo, long &hi) {
        lo = (scale * (std::min(c0, 0) + std::min(c1, 0) + std::min(c2, 0)) + add) >> sh;
        hi = (scale * (std::max(c0, 0) + std::max(c1, 0) + std::max(c2, 0)) + add) >> sh;
    };
    long ylo, yhi, ulo, uhi, vlo, vhi;
    lo_hi(q.ry, q.gy, q.by, 255, (32 << 14) + (1 << 8), 9, ylo, yhi);
    lo_hi(q.ru, q.gu, q.bu, 510, (256L << 15) + (1 << 9), 10, ulo, uhi);
    lo_hi(q.rv, q.gv, q.bv, 510, (256L << 15) + (1 << 9), 10, vlo, vhi);
    const char *ns = GMAT_KNOB("GMAT_R2Y_NOSAT");
    const bool nosat = !(ns && !atoi(ns)) && ylo >= 0 && yhi <= 16351 && ulo >= 0 && uhi <= 16383 && vlo >= 0 && vhi <= 16383;
#define GMAT_F2S(NV_, J_) do { if (nosat) hipLaunchKernelGGL(HIP_KERNEL_NAME(pf32_to_yuv420s_kernel<NV_, J_, true>), grid, block, 0, stream, a, *frames); \
                               else       hipLaunchKernelGGL(HIP_KERNEL_NAME(pf32_to_yuv420s_kernel<NV_, J_, false>), grid, block, 0, stream, a, *frames); } while (0)
    if (L.toJpeg) { if (L.nv12) GMAT_F2S(true, true); else GMAT_F2S(false, true); }
    else          { if (L.nv12) GMAT_F2S(true, false); else GMAT_F2S(false, false); }
#undef GMAT_F2S
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// NV12 <-> YUV420P chroma re-layout (nv12ToPlanarWrapper / planarToNv12Wrapper, swscale_unscaled.c).
// A thread moves 4 chroma samples of each plane: 8 interleaved bytes <-> 4 + 4 planar bytes (v_perm_b32).
__global__ __launch_bounds__(256) void uv_deinterleave_kernel(const uint8_t *uv, int uvs, uint8_t *u, int us, uint8_t *v, int vs,
                                                              int cw, int ch, int aligned)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x >= cw || y >= ch) return;
    const uint8_t *s = uv + (size_t)y * uvs + 2 * x;
    uint8_t *du = u + (size_t)y * us + x, *dv = v + (size_t)y * vs + x;
    if (aligned && x + 4 <= cw) {
        const uint2 t = *reinterpret_cast<const uint2 *>(s);            // U0 V0 U1 V1 | U2 V2 U3 V3
        *reinterpret_cast<unsigned *>(du) = __builtin_amdgcn_perm(t.y, t.x, 0x06040200u);
        *reinterpret_cast<unsigned *>(dv) = __builtin_amdgcn_perm(t.y, t.x, 0x07050301u);
    } else {
        for (int i = 0; i < min(4, cw - x); i++) { du[i] = s[2 * i]; dv[i] = s[2 * i + 1]; }
    }
}

__global__ __launch_bounds__(256) void uv_interleave_kernel(const uint8_t *u, int us, const uint8_t *v, int vs, uint8_t *uv, int uvs,
                                                            int cw, int ch, int aligned)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x >= cw || y >= ch) return;
    const uint8_t *su = u + (size_t)y * us + x, *sv = v + (size_t)y * vs + x;
    uint8_t *d = uv + (size_t)y * uvs + 2 * x;
    if (aligned && x + 4 <= cw) {
        const unsigned a = *reinterpret_cast<const unsigned *>(su), b = *reinterpret_cast<const unsigned *>(sv);
        *reinterpret_cast<uint2 *>(d) = make_uint2(__builtin_amdgcn_perm(b, a, 0x05010400u), __builtin_amdgcn_perm(b, a, 0x07030602u));
    } else {
        for (int i = 0; i < min(4, cw - x); i++) { d[2 * i] = su[i]; d[2 * i + 1] = sv[i]; }
    }
}

// 8 -> 16 bit expansion of planar8ToP01xleWrapper (swscale_unscaled.c:286-324): every sample t becomes the
// little-endian 16-bit value t | t << 8 (P010LE and P016LE alike; no 10-bit mask on the CPU).  One plane
// (b == nullptr: n bytes per row -> n words) or two planes interleaved (U, V -> U16 V16 pairs).
__global__ __launch_bounds__(256) void widen8to16_kernel(const uint8_t *a, int sa, const uint8_t *b, int sb, uint8_t *d, int ds,
                                                         int n, int h, int aligned)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x >= n || y >= h) return;
    const uint8_t *pa = a + (size_t)y * sa + x;
    if (!b) {
        uint8_t *o = d + (size_t)y * ds + 2 * x;
        if (aligned && x + 4 <= n) {
            const unsigned t = *reinterpret_cast<const unsigned *>(pa);
            *reinterpret_cast<uint2 *>(o) = make_uint2(__builtin_amdgcn_perm(0u, t, 0x01010000u), __builtin_amdgcn_perm(0u, t, 0x03030202u));
        } else {
            for (int i = 0; i < min(4, n - x); i++) { o[2 * i] = pa[i]; o[2 * i + 1] = pa[i]; }
        }
    } else {
        const uint8_t *pb = b + (size_t)y * sb + x;
        uint8_t *o = d + (size_t)y * ds + 4 * x;
        if (aligned && x + 4 <= n) {
            const unsigned u = *reinterpret_cast<const unsigned *>(pa), v = *reinterpret_cast<const unsigned *>(pb);
            *reinterpret_cast<uint4 *>(o) = make_uint4(__builtin_amdgcn_perm(v, u, 0x04040000u), __builtin_amdgcn_perm(v, u, 0x05050101u),
                                                       __builtin_amdgcn_perm(v, u, 0x06060202u), __builtin_amdgcn_perm(v, u, 0x07070303u));
        } else {
            for (int i = 0; i < min(4, n - x); i++) {
                o[4 * i] = o[4 * i + 1] = pa[i];
                o[4 * i + 2] = o[4 * i + 3] = pb[i];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
int rgb2yuv_prepare(const ScalePlan &p, Rgb2YuvPlan &t)
{
    // same-size RGB -> 4:2:0 only: identity horizontal filters, one-tap vertical luma
    if (p.srcW != p.dstW || p.srcH != p.dstH) return GMAT_ERR(ENOSYS);
    if (p.hLum.taps != 1 || p.hChr.taps != 1 || p.vLum.taps != 1 || !p.chrSrcHSub) return GMAT_ERR(ENOSYS);
    for (int i = 0; i < p.hChr.count; i++) if (p.hChr.pos[i] != i) return GMAT_ERR(ENOSYS);
    t.nty = (p.dstH + R2Y_TH - 1) / R2Y_TH;
    t.ntx = (p.dstW + R2Y_TW - 1) / R2Y_TW;
    t.rowStart.resize(t.nty); t.rowCount.resize(t.nty);
    t.maxRows = 0;
    for (int ty = 0; ty < t.nty; ty++) {
        int lo = INT32_MAX, hi = 0;
        for (int cy = ty * R2Y_CH; cy < std::min((ty + 1) * R2Y_CH, p.chrDstH); cy++) {
            lo = std::min(lo, p.vChr.pos_even[cy]);
            hi = std::max(hi, p.vChr.pos_even[cy] + 2 * p.vChr.pairs);
        }
        if (lo == INT32_MAX) { lo = 0; hi = 2; }
        t.rowStart[ty] = lo;
        t.rowCount[ty] = align_up(hi - lo, 2);
        t.maxRows = std::max(t.maxRows, t.rowCount[ty]);
    }
    if (t.maxRows * R2Y_CW * 4 > 60 * 1024) return GMAT_ERR(ENOSYS);
    // accumulator start: yuv2planeX_8_c / yuv2nv12cX_c dither 64<<12; the one-tap planar form adds 64 before >>7
    t.round.assign(p.chrDstH, 64 << 12);
    // the strip form: the vertical chroma filter as the replicated window [2y - 3, 2y + 4]
    t.stripOk = (p.srcH % 2 == 0 && p.chrDstH * 2 == p.srcH && filter_is_edge_replication(p.vChr, p.srcH, t.vC)) ? 1 : 0;
    return 0;
}

int launch_rgb2yuv420(const Rgb2YuvLaunch &L, hipStream_t stream)
{
    if (rgb2yuv420_strip_takes(L)) return launch_rgb2yuv420s(L, stream, nullptr, 1);
    Rgb2YuvArgs a;
    a.src = L.src; a.ss = L.ss; a.bgr = L.bgr; a.toJpeg = L.toJpeg;
    a.srcAligned = ((((uintptr_t)L.src | (uintptr_t)L.ss) & 3) == 0);
    a.y = L.y; a.u = L.u; a.v = L.v; a.ys = L.ys; a.us = L.us; a.vs = L.vs; a.nv12 = L.nv12;
    a.w = L.w; a.h = L.h; a.cw = (L.w + 1) / 2; a.ch = (L.h + 1) / 2;
    a.vChr = L.vChr; a.rowStart = L.rowStart; a.rowCount = L.rowCount; a.maxRows = L.maxRows; a.k = L.k;
    const dim3 grid((L.w + R2Y_TW - 1) / R2Y_TW, (L.h + R2Y_TH - 1) / R2Y_TH), block(256);
    const size_t lds = (size_t)L.maxRows * R2Y_CW * 4;      // two planes of (maxRows/2) x 64 dwords
    hipLaunchKernelGGL(rgb2yuv420_kernel, grid, block, lds, stream, a);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- packed RGB -> planar YUV 4:4:4 at equal size ------------------------------------------------------------
// The CPU generic path of an RGB24 -> YUV444P context: no chroma subsampling at either end, so every filter has
// one tap and each output sample depends on one pixel only:
//   rgb24ToY_c / rgb24ToUV_c (input.c:815-847) -> hScale16To15_c with one tap: min(2 * v14, 32767)
//   -> yuv2plane1_8_c (output.c:400-408): clip_u8((v + 64) >> 7)
// 4 pixels per thread (12 bytes in, one dword per plane out).
__global__ __launch_bounds__(256) void rgb2yuv444_kernel(const uint8_t *src, int ss, int bgr, uint8_t *y, int ys, uint8_t *u, int us,
                                                         uint8_t *v, int vs, int w, int h, int aligned, Rgb2YuvConsts k)
{
    const int col = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (col >= w || row >= h) return;
    const uint8_t *rp = src + (size_t)row * ss;
    int r[4], g[4], b[4];
    const bool full = col + 4 <= w;
    if (aligned && full) {
        const uint3 q = *reinterpret_cast<const uint3 *>(rp + (size_t)col * 3);
        r[0] = q.x & 0xFF;         g[0] = (q.x >> 8) & 0xFF;  b[0] = (q.x >> 16) & 0xFF;
        r[1] = q.x >> 24;          g[1] = q.y & 0xFF;         b[1] = (q.y >> 8) & 0xFF;
        r[2] = (q.y >> 16) & 0xFF; g[2] = q.y >> 24;          b[2] = q.z & 0xFF;
        r[3] = (q.z >> 8) & 0xFF;  g[3] = (q.z >> 16) & 0xFF; b[3] = q.z >> 24;
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = min(col + i, w - 1);
            r[i] = rp[3 * c]; g[i] = rp[3 * c + 1]; b[i] = rp[3 * c + 2];
        }
    }
    if (bgr) {
#pragma unroll
        for (int i = 0; i < 4; i++) { const int t = r[i]; r[i] = b[i]; b[i] = t; }
    }
    unsigned yb = 0, ub = 0, vb = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        yb |= (unsigned)clip_u8_shr(min(2 * rgb_to_y14(k, r[i], g[i], b[i]), 32767) + 64, 7) << (8 * i);
        ub |= (unsigned)clip_u8_shr(min(2 * rgb_to_u14(k, r[i], g[i], b[i]), 32767) + 64, 7) << (8 * i);
        vb |= (unsigned)clip_u8_shr(min(2 * rgb_to_v14(k, r[i], g[i], b[i]), 32767) + 64, 7) << (8 * i);
    }
    uint8_t *dy = y + (size_t)row * ys + col, *du = u + (size_t)row * us + col, *dv = v + (size_t)row * vs + col;
    if (aligned && full) {
        *reinterpret_cast<unsigned *>(dy) = yb;
        *reinterpret_cast<unsigned *>(du) = ub;
        *reinterpret_cast<unsigned *>(dv) = vb;
    } else {
        for (int i = 0; i < min(4, w - col); i++) {
            dy[i] = (uint8_t)(yb >> (8 * i)); du[i] = (uint8_t)(ub >> (8 * i)); dv[i] = (uint8_t)(vb >> (8 * i));
        }
    }
}

int launch_rgb2yuv444(const uint8_t *src, int ss, int bgr, uint8_t *y, int ys, uint8_t *u, int us, uint8_t *v, int vs, int w, int h,
                      const Rgb2YuvConsts &k, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return 0;
    const int aligned = ((((uintptr_t)src | (uintptr_t)ss | (uintptr_t)y | (uintptr_t)ys | (uintptr_t)u | (uintptr_t)us |
                           (uintptr_t)v | (uintptr_t)vs) & 3) == 0);
    const dim3 grid((w + 255) / 256, (h + 3) / 4), block(256);
    hipLaunchKernelGGL(rgb2yuv444_kernel, grid, block, 0, stream, src, ss, bgr, y, ys, u, us, v, vs, w, h, aligned, k);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// NV12 <-> YUV420P in ONE launch (round 4): rows [0, h) copy the luma plane, rows [h, h + ch) (de)interleave the chroma — 16 bytes a lane each way,
// every byte read once and written once (streaming loads and stores), grid.z = frame.  Rounds 1-3: a 2-D copy and a chroma kernel, 7.2 / 6.3 us
// for a 4K frame; this is what nvdec's NV12 meets in front of every planar consumer (scale_cuda=format=yuv420p).
template <bool TO_PLANAR, bool FRAMES>
__global__ __launch_bounds__(256) void yuv420_relayout_kernel(const uint8_t *y, int ys, const uint8_t *a0, int s0, const uint8_t *a1, int s1,
                                                              uint8_t *dy, int dys, uint8_t *d0, int ds0, uint8_t *d1, int ds1, int w, int h, int cw, int ch,
                                                              Yuv2xFrames fr)
{
    if (FRAMES) {
        const int f = blockIdx.z;
        y = fr.y[f]; a0 = fr.u[f]; a1 = fr.v[f]; dy = fr.dst[f]; d0 = fr.dstU[f]; d1 = fr.dstV[f];
    }
    const int x = (blockIdx.x * 256 + threadIdx.x) * 16, r = blockIdx.y;
    if (r < h) {
        if (x >= w) return;
        const uint8_t *s = y + (size_t)r * ys + x;
        uint8_t *d = dy + (size_t)r * dys + x;
        if (x + 16 <= w) st_stream(d, ld_stream(s, uint4()));
        else for (int i = 0; i < w - x; i++) d[i] = s[i];
        return;
    }
    const int cr = r - h;
    if (cr >= ch) return;
    if (TO_PLANAR) {                                    // 16 interleaved bytes -> 8 + 8 planar bytes; x counts interleaved bytes
        if (x >= 2 * cw) return;
        const uint8_t *s = a0 + (size_t)cr * s0 + x;
        uint8_t *du = d0 + (size_t)cr * ds0 + (x >> 1), *dv = d1 + (size_t)cr * ds1 + (x >> 1);
        if (x + 16 <= 2 * cw) {
            const uint4 t = ld_stream(s, uint4());          // U0 V0 U1 V1 | U2 V2 U3 V3 | ...
            st_stream(du, make_uint2(__builtin_amdgcn_perm(t.y, t.x, 0x06040200u), __builtin_amdgcn_perm(t.w, t.z, 0x06040200u)));
            st_stream(dv, make_uint2(__builtin_amdgcn_perm(t.y, t.x, 0x07050301u), __builtin_amdgcn_perm(t.w, t.z, 0x07050301u)));
        } else for (int i = 0; i < cw - (x >> 1); i++) { du[i] = s[2 * i]; dv[i] = s[2 * i + 1]; }
    } else {                                            // 8 + 8 planar bytes -> 16 interleaved bytes
        if (x >= 2 * cw) return;
        const uint8_t *su = a0 + (size_t)cr * s0 + (x >> 1), *sv = a1 + (size_t)cr * s1 + (x >> 1);
        uint8_t *d = d0 + (size_t)cr * ds0 + x;
        if (x + 16 <= 2 * cw) {
            const uint2 a = ld_stream(su, uint2()), b = ld_stream(sv, uint2());
            st_stream(d, make_uint4(__builtin_amdgcn_perm(b.x, a.x, 0x05010400u), __builtin_amdgcn_perm(b.x, a.x, 0x07030602u),
                                    __builtin_amdgcn_perm(b.y, a.y, 0x05010400u), __builtin_amdgcn_perm(b.y, a.y, 0x07030602u)));
        } else for (int i = 0; i < cw - (x >> 1); i++) { d[2 * i] = su[i]; d[2 * i + 1] = sv[i]; }
    }
}

// every plane pointer and pitch of every frame can be moved 16 (interleaved side, luma) / 8 (planar chroma) bytes at a time
bool yuv420_relayout_takes(int toPlanar, const uint8_t *y, int ys, const uint8_t *a0, int s0, const uint8_t *a1, int s1,
                           const uint8_t *dy, int dys, const uint8_t *d0, int ds0, const uint8_t *d1, int ds1)
{
    uintptr_t wide = (uintptr_t)y | (uintptr_t)ys | (uintptr_t)dy | (uintptr_t)dys, narrow;
    if (toPlanar) { wide |= (uintptr_t)a0 | (uintptr_t)s0; narrow = (uintptr_t)d0 | (uintptr_t)ds0 | (uintptr_t)d1 | (uintptr_t)ds1; }
    else          { wide |= (uintptr_t)d0 | (uintptr_t)ds0; narrow = (uintptr_t)a0 | (uintptr_t)s0 | (uintptr_t)a1 | (uintptr_t)s1; }
    return (wide & 15) == 0 && (narrow & 7) == 0;
}

// one frame (frames == nullptr) or nframes frames of this geometry and these pitches (y / u / v = the source planes, dst / dstU / dstV the destination's)
int launch_yuv420_relayout(int toPlanar, const uint8_t *y, int ys, const uint8_t *a0, int s0, const uint8_t *a1, int s1,
                           uint8_t *dy, int dys, uint8_t *d0, int ds0, uint8_t *d1, int ds1, int w, int h, hipStream_t stream,
                           const Yuv2xFrames *frames, int nframes)
{
    if (w <= 0 || h <= 0) return 0;
    if (frames && (nframes < 1 || nframes > kYuv2xMaxFrames)) return GMAT_ERR(EINVAL);
    const int cw = (w + 1) / 2, ch = (h + 1) / 2;
    const int rowBytes = std::max(w, 2 * cw);
    const dim3 grid((rowBytes + 4095) / 4096, h + ch, frames ? nframes : 1), block(256);
    Yuv2xFrames none; none.y[0] = nullptr;
#define GMAT_RL(TP_) do { if (frames) hipLaunchKernelGGL(HIP_KERNEL_NAME(yuv420_relayout_kernel<TP_, true>), grid, block, 0, stream, y, ys, a0, s0, a1, s1, dy, dys, d0, ds0, d1, ds1, w, h, cw, ch, *frames); \
                            else        hipLaunchKernelGGL(HIP_KERNEL_NAME(yuv420_relayout_kernel<TP_, false>), grid, block, 0, stream, y, ys, a0, s0, a1, s1, dy, dys, d0, ds0, d1, ds1, w, h, cw, ch, none); } while (0)
    if (toPlanar) GMAT_RL(true); else GMAT_RL(false);
#undef GMAT_RL
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_uv_relayout(int toPlanar, const uint8_t *a0, int s0, const uint8_t *a1, int s1, uint8_t *d0, int ds0, uint8_t *d1,
                       int ds1, int cw, int ch, hipStream_t stream)
{
    if (cw <= 0 || ch <= 0) return 0;
    const dim3 grid((cw + 1023) / 1024, ch), block(256);
    if (toPlanar) {
        const int al = ((((uintptr_t)a0 | (uintptr_t)s0) & 7) == 0) && ((((uintptr_t)d0 | (uintptr_t)ds0 | (uintptr_t)d1 | (uintptr_t)ds1) & 3) == 0);
        hipLaunchKernelGGL(uv_deinterleave_kernel, grid, block, 0, stream, a0, s0, d0, ds0, d1, ds1, cw, ch, al);
    } else {
        const int al = ((((uintptr_t)d0 | (uintptr_t)ds0) & 7) == 0) && ((((uintptr_t)a0 | (uintptr_t)s0 | (uintptr_t)a1 | (uintptr_t)s1) & 3) == 0);
        hipLaunchKernelGGL(uv_interleave_kernel, grid, block, 0, stream, a0, s0, a1, s1, d0, ds0, cw, ch, al);
    }
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// planarCopyWrapper's 8-bit -> deeper planar copy (swscale_unscaled.c:1844-1862, COPY816), one plane: v << (depth - 8), with the
// top bits replicated into the new low bits (| v >> (16 - depth)) for the luma of a full-range source.  Four samples per thread.
__global__ __launch_bounds__(256) void plane_copy_up_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int depth, int replicate)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const uint8_t *s = src + (size_t)y * ss + x;
    unsigned short *d = reinterpret_cast<unsigned short *>(dst + (size_t)y * ds) + x;
    for (int i = 0; i < min(4, w - x); i++) {
        const unsigned v = s[i];
        d[i] = (unsigned short)((v << (depth - 8)) | (replicate ? v >> (16 - depth) : 0u));
    }
}

// planarCopyWrapper's deeper -> 8-bit planar copy (swscale_unscaled.c:1743-1800, DITHER_COPY), one plane: shift = depth - 8 and
// d = dithers[shift - 1][y & 7][x & 7] (:40-113) — shift 2: 1 2 / 3 0 alternating, shift 8: ff_dither_8x8_128's values (dither_8x8_128).
// shiftonly (chroma, luma of a limited-range source): t = (v + d) >> shift, t - (t >> 8); else (v - (v >> 8) + d) >> shift.  Four samples a thread.
__global__ __launch_bounds__(256) void plane_copy_down_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int depth, int shiftonly)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const unsigned short *s = reinterpret_cast<const unsigned short *>(src + (size_t)y * ss) + x;
    uint8_t *d = dst + (size_t)y * ds + x;
    const int shift = depth - 8;
    for (int i = 0; i < min(4, w - x); i++) {
        const unsigned v = s[i];
        const unsigned dith = shift == 2 ? (unsigned)(((y & 1) ? 3 : 1) ^ (((x + i) & 1) ? 3 : 0)) : (unsigned)dither_8x8_128(x + i, y);
        unsigned t;
        if (shiftonly) { t = (v + dith) >> shift; t -= t >> 8; }
        else           { t = (v - (v >> 8) + dith) >> shift; }
        d[i] = (uint8_t)t;
    }
}

int launch_plane_copy_down(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int depth, int shiftonly, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return 0;
    const dim3 grid((w + 1023) / 1024, h), block(256);
    hipLaunchKernelGGL(plane_copy_down_kernel, grid, block, 0, stream, src, ss, dst, ds, w, h, depth, shiftonly);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// (round 6) the three planes of a frame in ONE launch (three launches of four samples a thread before: yuv420p10le -> yuv420p at 1080p 9.7 us a frame, 0.12 of the
// roofline, most of it the gaps between six-microsecond-class launches — profiles/r06_sweep_before.txt): rows [0, h) of the luma plane, then the two chroma
// planes' (cw x ch each); a lane: eight samples — sixteen bytes in, eight out — where the rows allow (al: source rows on 16-byte, destination rows on 8-byte addresses)
struct PlaneDown3 { const uint8_t *src[3]; uint8_t *dst[3]; int ss[3], ds[3]; };
__global__ __launch_bounds__(256) void plane_copy_down3_kernel(PlaneDown3 P, int w, int h, int cw, int ch, int depth, int shiftonlyY, int al)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 8;
    int r = blockIdx.y;
    const int pl = r < h ? 0 : r < h + ch ? 1 : 2;
    if (pl == 1) r -= h; else if (pl == 2) r -= h + ch;
    const int pw = pl ? cw : w;
    if (x >= pw) return;
    const uint8_t *srow = (pl == 0 ? P.src[0] : pl == 1 ? P.src[1] : P.src[2]) + (size_t)r * (pl == 0 ? P.ss[0] : pl == 1 ? P.ss[1] : P.ss[2]);
    uint8_t *drow = (pl == 0 ? P.dst[0] : pl == 1 ? P.dst[1] : P.dst[2]) + (size_t)r * (pl == 0 ? P.ds[0] : pl == 1 ? P.ds[1] : P.ds[2]);
    const int shift = depth - 8, shiftonly = pl ? 1 : shiftonlyY;
    auto one = [&](unsigned v, int xx) -> unsigned {
        const unsigned dith = shift == 2 ? (unsigned)(((r & 1) ? 3 : 1) ^ ((xx & 1) ? 3 : 0)) : (unsigned)dither_8x8_128(xx, r);
        unsigned t;
        if (shiftonly) { t = (v + dith) >> shift; t -= t >> 8; }
        else           { t = (v - (v >> 8) + dith) >> shift; }
        return t & 0xFFu;
    };
    if (al && x + 8 <= pw) {
        const uint4 v = *reinterpret_cast<const uint4 *>(srow + 2 * (size_t)x);
        const unsigned q[4] = {v.x, v.y, v.z, v.w};
        unsigned o[2] = {0, 0};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            o[i >> 1] |= one(q[i] & 0xFFFFu, x + 2 * i) << (16 * (i & 1));
            o[i >> 1] |= one(q[i] >> 16, x + 2 * i + 1) << (16 * (i & 1) + 8);
        }
        *reinterpret_cast<uint2 *>(drow + x) = make_uint2(o[0], o[1]);
    } else {
        const unsigned short *s16 = reinterpret_cast<const unsigned short *>(srow);
        for (int i = 0; i < min(8, pw - x); i++) drow[x + i] = (uint8_t)one(s16[x + i], x + i);
    }
}

int launch_planes_copy_down(const uint8_t *const src[3], const int ss[3], uint8_t *const dst[3], const int ds[3], int w, int h, int cw, int ch, int depth,
                            int shiftonlyY, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return 0;
    PlaneDown3 P;
    uintptr_t sa = 0, da = 0;
    for (int i = 0; i < 3; i++) { P.src[i] = src[i]; P.dst[i] = dst[i]; P.ss[i] = ss[i]; P.ds[i] = ds[i]; sa |= (uintptr_t)src[i] | (uintptr_t)ss[i]; da |= (uintptr_t)dst[i] | (uintptr_t)ds[i]; }
    const dim3 grid((w + 2047) / 2048, h + 2 * ch), block(256);
    hipLaunchKernelGGL(plane_copy_down3_kernel, grid, block, 0, stream, P, w, h, cw, ch, depth, shiftonlyY, ((sa & 15) == 0 && (da & 7) == 0) ? 1 : 0);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_plane_copy_up(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int depth, int replicate, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return 0;
    const dim3 grid((w + 1023) / 1024, h), block(256);
    hipLaunchKernelGGL(plane_copy_up_kernel, grid, block, 0, stream, src, ss, dst, ds, w, h, depth, replicate);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// NV12 -> P010LE / P016LE at equal size (round 4).  libswscale has no special converter for a semi-planar 8-bit source (planar8ToP01xleWrapper takes
// PLANAR ones only, swscale_unscaled.c:2108-2112): the generic lines run with one-tap filters — hScale8To15_c t << 7 then yuv2p010l1_c
// ((t << 7) + 16 >> 5) << 6, or hScale8To19_c t << 11 then yuv2p016 ((t << 11) + 4 >> 3) — every sample of BOTH planes becomes t << 8, and the
// chroma keeps its interleaving.  Rounds 1-3 ran the tiled plane scaler for it (35 us a 4K frame, 0.13 of the roofline); this is the copy it is:
// a lane turns 8 bytes into 8 words, rows [0, h) from the luma plane and [h, h + ch) from the chroma plane (both w bytes wide).
__global__ __launch_bounds__(256) void nv12_shift8_kernel(const uint8_t *y, int ys, const uint8_t *uv, int uvs, uint8_t *dy, int dys, uint8_t *duv, int duvs,
                                                          int w, int h, int ch, int aligned)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 8, r = blockIdx.y;
    if (x >= w || r >= h + ch) return;
    const uint8_t *s = r < h ? y + (size_t)r * ys + x : uv + (size_t)(r - h) * uvs + x;
    uint8_t *d = r < h ? dy + (size_t)r * dys + 2 * x : duv + (size_t)(r - h) * duvs + 2 * x;
    if (aligned && x + 8 <= w) {
        const uint2 t = ld_stream(s, uint2());                          // (each byte is read once and each word written once: streaming both ways)
        st_stream(d, make_uint4(__builtin_amdgcn_perm(0u, t.x, 0x010C000Cu), __builtin_amdgcn_perm(0u, t.x, 0x030C020Cu),
                                __builtin_amdgcn_perm(0u, t.y, 0x010C000Cu), __builtin_amdgcn_perm(0u, t.y, 0x030C020Cu)));
    } else {
        for (int i = 0; i < min(8, w - x); i++) { d[2 * i] = 0; d[2 * i + 1] = s[i]; }
    }
}

int launch_nv12_shift8(const uint8_t *y, int ys, const uint8_t *uv, int uvs, uint8_t *dy, int dys, uint8_t *duv, int duvs, int w, int h, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return 0;
    const int ch = (h + 1) / 2, wb = 2 * ((w + 1) / 2);                // the chroma row's bytes (an odd width: one more than the luma row's)
    const int al = ((((uintptr_t)y | (uintptr_t)ys | (uintptr_t)uv | (uintptr_t)uvs) & 7) == 0) &&
                   ((((uintptr_t)dy | (uintptr_t)dys | (uintptr_t)duv | (uintptr_t)duvs) & 15) == 0) && w == wb;
    if (w != wb) {                                                      // odd width: the two planes differ in width — two launches of the one-plane form
        const dim3 g1((w + 2047) / 2048, h), g2((wb + 2047) / 2048, ch), block(256);
        hipLaunchKernelGGL(nv12_shift8_kernel, g1, block, 0, stream, y, ys, uv, uvs, dy, dys, duv, duvs, w, h, 0, 0);
        hipLaunchKernelGGL(nv12_shift8_kernel, g2, block, 0, stream, uv, uvs, uv, uvs, duv, duvs, duv, duvs, wb, ch, 0, 0);
    } else {
        const dim3 grid((w + 2047) / 2048, h + ch), block(256);
        hipLaunchKernelGGL(nv12_shift8_kernel, grid, block, 0, stream, y, ys, uv, uvs, dy, dys, duv, duvs, w, h, ch, al);
    }
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_widen8to16(const uint8_t *a, int sa, const uint8_t *b, int sb, uint8_t *d, int ds, int n, int h, hipStream_t stream)
{
    if (n <= 0 || h <= 0) return 0;
    const dim3 grid((n + 1023) / 1024, h), block(256);
    const int outAlign = b ? 15 : 7;
    const int al = ((((uintptr_t)d | (uintptr_t)ds) & outAlign) == 0) && ((((uintptr_t)a | (uintptr_t)sa) & 3) == 0) &&
                   (!b || (((uintptr_t)b | (uintptr_t)sb) & 3) == 0);
    hipLaunchKernelGGL(widen8to16_kernel, grid, block, 0, stream, a, sa, b, sb, d, ds, n, h, al);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
