// k_scale.hip — libswscale's generic scaler with packed-RGB output as ONE fused, LDS-tiled kernel
// for gfx950 (MI355X).  Integer arithmetic, bit-exact with the portable C build of libswscale.
//
// What it replaces: ff_swscale_cuda's "convert, then CV-CUDA resize" (libswscale/cuda/swscale_cuda.c:
// 273-479; always bilinear, third-party arithmetic) by the arithmetic of the CPU scaler:
//   input  stage  rgb24ToY_c / rgb24ToUV_c / rgb24ToUV_half_c          input.c:815-866
//   horizontal    hScale16To15_c: min(sum(src*f) >> 13, 32767)         swscale.c:93-119
//   vertical+out  yuv2rgb_full_X_c + yuv2rgb_write_full                output.c:2037-2082,1886-1935
//                 (1-tap and 2-tap special forms folded in through the `round` table)
// srcKind 1 additionally runs the nearest-chroma yuv2rgb stage (yuv2rgb.c) in front, in registers,
// so NV12 -> RGB -> scaled RGB never materialises the full-size RGB frame in HBM.
//
// One 256-thread block produces a TW x TH output tile in three phases separated by barriers:
//   1. LOAD    : the source window [rowStart,+rows) x [colStart,+cols) is read with 12-byte
//                (global_load_dwordx3, 4 pixels) coalesced loads, converted to 14-bit Y/U/V and
//                written to LDS as int16 (chroma optionally pair-averaged).
//   2. H-FILTER: thread (xo, r) runs the horizontal taps with v_dot2c_i32_i16 on dword pairs read
//                from LDS; its coefficient pairs stay in registers for all rows.  Results go to LDS
//                with two source rows interleaved per dword so that phase 3 can use dot2 too.
//   3. V-FILTER: thread (4 output pixels, 1 output row) reads ds_read_b128 row-pair vectors,
//                accumulates the vertical taps, runs the colour stage and writes 12 (rgb24) or
//                16 (rgba) contiguous bytes -> 192/256-byte runs per 16 lanes.
// Odd filter positions are handled on the host by prepending a zero tap (FilterBank::packed), so the
// kernel never re-aligns data.  Taps beyond a row's real window multiply finite in-range samples by
// zero coefficients.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

namespace gmat {

constexpr int kMaxPairs = 8;      // <= 15 taps after the parity shift (lanczos-3 at 2:1 needs 7)
constexpr int kUnitsInFlight = 3; // phase-1 work units whose global loads are issued before any is consumed

// A phase-1 work unit is 4 source pixels x 2 source rows.  RGB source: two 12-byte loads.  YUV 4:2:0
// source: two luma dwords + ONE chroma dword (both rows share it: the row window starts on an even row).
struct RawUnit { uint3 a, b; };          // RGB: rows A,B.  YUV: a.x = Y(row A), a.y = Y(row B), a.z = U0 V0 U1 V1

__device__ __forceinline__ unsigned pack16(int lo, int hi) { return ((unsigned)lo & 0xFFFF) | ((unsigned)hi << 16); }

template <bool FAST>
__device__ __forceinline__ uint3 load_rgb12(const ScaleArgs &a, int srow, int col)
{
    const uint8_t *row = a.src0 + (size_t)srow * a.ss0;
    if (FAST) return *reinterpret_cast<const uint3 *>(row + (size_t)col * 3);
    unsigned b[12];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = min(col + i, a.srcW - 1);
        b[3 * i] = row[3 * c]; b[3 * i + 1] = row[3 * c + 1]; b[3 * i + 2] = row[3 * c + 2];
    }
    uint3 v;
    v.x = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
    v.y = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
    v.z = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
    return v;
}

template <bool FAST>
__device__ __forceinline__ unsigned load_y4(const ScaleArgs &a, int srow, int col)
{
    const uint8_t *yrow = a.src0 + (size_t)srow * a.ss0;
    if (FAST) return *reinterpret_cast<const unsigned *>(yrow + col);
    unsigned v = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) v |= (unsigned)yrow[min(col + i, a.srcW - 1)] << (8 * i);
    return v;
}

template <bool FAST>
__device__ __forceinline__ unsigned load_uv4(const ScaleArgs &a, int srow, int col)
{
    const size_t crow = (size_t)(srow >> 1);
    if (FAST) {
        if (a.srcNv12) return *reinterpret_cast<const unsigned *>(a.src1 + crow * a.ss1 + col);
        const unsigned uu = *reinterpret_cast<const unsigned short *>(a.src1 + crow * a.ss1 + (col >> 1));
        const unsigned vv = *reinterpret_cast<const unsigned short *>(a.src2 + crow * a.ss2 + (col >> 1));
        return (uu & 0xFF) | ((vv & 0xFF) << 8) | ((uu >> 8) << 16) | ((vv >> 8) << 24);
    }
    unsigned v = 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int cc = min(col + 2 * i, a.srcW - 1) >> 1;
        unsigned U, V;
        if (a.srcNv12) { const uint8_t *q = a.src1 + crow * a.ss1 + 2 * cc; U = q[0]; V = q[1]; }
        else           { U = a.src1[crow * a.ss1 + cc]; V = a.src2[crow * a.ss2 + cc]; }
        v |= (U | (V << 8)) << (16 * i);
    }
    return v;
}

struct Px4 { int r[4], g[4], b[4]; };

__device__ __forceinline__ void unpack_rgb12(const uint3 v, bool bgr, Px4 &p)
{
    p.r[0] = v.x & 0xFF;         p.g[0] = (v.x >> 8) & 0xFF;  p.b[0] = (v.x >> 16) & 0xFF;
    p.r[1] = v.x >> 24;          p.g[1] = v.y & 0xFF;         p.b[1] = (v.y >> 8) & 0xFF;
    p.r[2] = (v.y >> 16) & 0xFF; p.g[2] = v.y >> 24;          p.b[2] = v.z & 0xFF;
    p.r[3] = (v.z >> 8) & 0xFF;  p.g[3] = (v.z >> 16) & 0xFF; p.b[3] = v.z >> 24;
    if (bgr) {
#pragma unroll
        for (int i = 0; i < 4; i++) { const int t = p.r[i]; p.r[i] = p.b[i]; p.b[i] = t; }
    }
}

__device__ __forceinline__ void yuv_row_to_rgb(const Yuv2RgbConsts &k, unsigned y4, const ChromaTerms &c0,
                                               const ChromaTerms &c1, Px4 &p)
{
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int ya = m24((int)((y4 >> (8 * i)) & 0xFF), k.cy), yb = m24((int)((y4 >> (8 * i + 16)) & 0xFF), k.cy);
        p.r[i] = luma_chan(c0.r, ya); p.g[i] = luma_chan(c0.g, ya); p.b[i] = luma_chan(c0.b, ya);
        p.r[i + 2] = luma_chan(c1.r, yb); p.g[i + 2] = luma_chan(c1.g, yb); p.b[i + 2] = luma_chan(c1.b, yb);
    }
}

// One converted row of a unit: 4 luma samples + chroma (2 pair-averaged or 4 full) in 14-bit form.
struct Row14 { int y[4], u[4], v[4]; };

__device__ __forceinline__ void rgb_row_to_14(const ScaleArgs &a, const Px4 &p, Row14 &o)
{
#pragma unroll
    for (int i = 0; i < 4; i++) o.y[i] = rgb_to_y14(a.r2y, p.r[i], p.g[i], p.b[i]);
    if (a.chrHalf) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int rs = p.r[2 * i] + p.r[2 * i + 1], gs = p.g[2 * i] + p.g[2 * i + 1], bs = p.b[2 * i] + p.b[2 * i + 1];
            o.u[i] = rgbsum_to_u14(a.r2y, rs, gs, bs);
            o.v[i] = rgbsum_to_v14(a.r2y, rs, gs, bs);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            o.u[i] = rgb_to_u14(a.r2y, p.r[i], p.g[i], p.b[i]);
            o.v[i] = rgb_to_v14(a.r2y, p.r[i], p.g[i], p.b[i]);
        }
    }
}

// phase 1 of scale_rgb_kernel: load the tile's source window, convert, store 14-bit planes to LDS.
// FAST = every 4-pixel group of the window lies inside the frame and rows are dword aligned (block-
// uniform), so all loads are unconditional vector loads and kUnitsInFlight units are in flight at once.
template <int TW, int SRCKIND, bool FAST>
__device__ __forceinline__ void scale_phase1(const ScaleArgs &a, int tid, int tx0, int c0, int nc, int r0, int nr,
                                             unsigned ngMagic, int strideCols, int CW, unsigned short *ly,
                                             unsigned short *lu, unsigned short *lv, int *hu, int *hv)
{
    const int ng = nc >> 2;
        const int total = (nr >> 1) * ng;                // units: row pair x 4-pixel group
        for (int base = tid; base < total; base += 256 * kUnitsInFlight) {
            RawUnit raw[kUnitsInFlight];
            int urp[kUnitsInFlight], ucg[kUnitsInFlight];
#pragma unroll
            for (int j = 0; j < kUnitsInFlight; j++) {
                const int g = base + j * 256;
                const int gg = min(g, total - 1);        // tail lanes reload the last unit (never stored)
                // gg / ng by multiplication; the 32-bit magic of ng == 1 (2^32 + 1) does not exist
                const int rp = ng == 1 ? gg : (int)__umulhi((unsigned)gg, ngMagic);
                const int cg = gg - rp * ng;
                urp[j] = g < total ? rp : -1;
                ucg[j] = cg;
                const int srowA = min(r0 + 2 * rp, a.srcH - 1), srowB = min(r0 + 2 * rp + 1, a.srcH - 1);
                const int col = c0 + 4 * cg;
                if (SRCKIND == 0) {
                    raw[j].a = load_rgb12<FAST>(a, srowA, col);
                    raw[j].b = load_rgb12<FAST>(a, srowB, col);
                } else {
                    raw[j].a.x = load_y4<FAST>(a, srowA, col);
                    raw[j].a.y = load_y4<FAST>(a, srowB, col);
                    raw[j].a.z = load_uv4<FAST>(a, srowA, col);
                }
            }
#pragma unroll
            for (int j = 0; j < kUnitsInFlight; j++) {
                if (urp[j] < 0) continue;
                const int rp = urp[j], cg = ucg[j];
                Px4 pa, pb;
                if (SRCKIND == 0) {
                    unpack_rgb12(raw[j].a, a.srcBgr != 0, pa);
                    unpack_rgb12(raw[j].b, a.srcBgr != 0, pb);
                } else {
                    const unsigned uv = raw[j].a.z;
                    const ChromaTerms t0 = chroma_terms(a.y2r, uv & 0xFF, (uv >> 8) & 0xFF);
                    const ChromaTerms t1 = chroma_terms(a.y2r, (uv >> 16) & 0xFF, uv >> 24);
                    yuv_row_to_rgb(a.y2r, raw[j].a.x, t0, t1, pa);
                    yuv_row_to_rgb(a.y2r, raw[j].a.y, t0, t1, pb);
                }
                Row14 ra, rb;
                rgb_row_to_14(a, pa, ra);
                rgb_row_to_14(a, pb, rb);
                unsigned short *ya = ly + (2 * rp) * strideCols + 4 * cg;
                *reinterpret_cast<uint2 *>(ya) = make_uint2(pack16(ra.y[0], ra.y[1]), pack16(ra.y[2], ra.y[3]));
                *reinterpret_cast<uint2 *>(ya + strideCols) = make_uint2(pack16(rb.y[0], rb.y[1]), pack16(rb.y[2], rb.y[3]));
                if (a.chromaDirect) {
                    // hScale16To15 with the single tap 16384: min((x * 16384) >> 13, 32767) = min(2x, 32767);
                    // written straight into the row-pair-interleaved h-filtered planes
                    const int n = a.chrHalf ? 2 : 4;
                    const int cc0 = (a.chrHalf ? ((c0 + 4 * cg) >> 1) : (c0 + 4 * cg)) - tx0;
                    int *pu = hu + rp * TW + cc0, *pv = hv + rp * TW + cc0;
                    if (cc0 >= 0 && cc0 < TW)         { pu[0] = (int)pack16(min(2 * ra.u[0], 32767), min(2 * rb.u[0], 32767)); pv[0] = (int)pack16(min(2 * ra.v[0], 32767), min(2 * rb.v[0], 32767)); }
                    if (cc0 + 1 >= 0 && cc0 + 1 < TW) { pu[1] = (int)pack16(min(2 * ra.u[1], 32767), min(2 * rb.u[1], 32767)); pv[1] = (int)pack16(min(2 * ra.v[1], 32767), min(2 * rb.v[1], 32767)); }
                    if (n == 4) {
                        if (cc0 + 2 >= 0 && cc0 + 2 < TW) { pu[2] = (int)pack16(min(2 * ra.u[2], 32767), min(2 * rb.u[2], 32767)); pv[2] = (int)pack16(min(2 * ra.v[2], 32767), min(2 * rb.v[2], 32767)); }
                        if (cc0 + 3 >= 0 && cc0 + 3 < TW) { pu[3] = (int)pack16(min(2 * ra.u[3], 32767), min(2 * rb.u[3], 32767)); pv[3] = (int)pack16(min(2 * ra.v[3], 32767), min(2 * rb.v[3], 32767)); }
                    }
                } else if (a.chrHalf) {
                    unsigned short *ua = lu + (2 * rp) * CW + 2 * cg, *va = lv + (2 * rp) * CW + 2 * cg;
                    *reinterpret_cast<unsigned *>(ua) = pack16(ra.u[0], ra.u[1]);
                    *reinterpret_cast<unsigned *>(va) = pack16(ra.v[0], ra.v[1]);
                    *reinterpret_cast<unsigned *>(ua + CW) = pack16(rb.u[0], rb.u[1]);
                    *reinterpret_cast<unsigned *>(va + CW) = pack16(rb.v[0], rb.v[1]);
                } else {
                    unsigned short *ua = lu + (2 * rp) * CW + 4 * cg, *va = lv + (2 * rp) * CW + 4 * cg;
                    *reinterpret_cast<uint2 *>(ua) = make_uint2(pack16(ra.u[0], ra.u[1]), pack16(ra.u[2], ra.u[3]));
                    *reinterpret_cast<uint2 *>(va) = make_uint2(pack16(ra.v[0], ra.v[1]), pack16(ra.v[2], ra.v[3]));
                    *reinterpret_cast<uint2 *>(ua + CW) = make_uint2(pack16(rb.u[0], rb.u[1]), pack16(rb.u[2], rb.u[3]));
                    *reinterpret_cast<uint2 *>(va + CW) = make_uint2(pack16(rb.v[0], rb.v[1]), pack16(rb.v[2], rb.v[3]));
                }
            }
        }
}

// LONG: horizontal filters longer than 2*kMaxPairs taps, compiled only into their own instantiation
template <int TW, int SRCKIND, bool LONG>
__global__ __launch_bounds__(256) void scale_rgb_kernel(ScaleArgs a, int strideCols, int maxRows)
{
    HIP_DYNAMIC_SHARED(uint4, lds_base)
    char *lds = reinterpret_cast<char *>(lds_base);

    // ---- tile selection; optional XCD-aware order: each XCD walks a contiguous column-major run
    // of tiles so vertically adjacent tiles (which share filter-support rows) meet in one L2 ----
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
    const int tx0 = tcol * TW, ty0 = trow * a.TH;
    const int c0 = uniform_load(a.colStart, tcol), nc = uniform_load(a.colCount, tcol);
    const int r0 = uniform_load(a.rowStart, trow), nr = uniform_load(a.rowCount, trow);
    const unsigned ngMagic = (unsigned)uniform_load(a.colMagic, tcol);

    // LDS carve-up.  With chromaDirect (1-tap identity chroma filter, e.g. every 2:1 RGB down-scale) the
    // chroma planes skip the staging arrays and phase 2 entirely.
    const int CW = a.chromaDirect ? 0 : (a.chrHalf ? (strideCols >> 1) : strideCols);
    unsigned short *ly = reinterpret_cast<unsigned short *>(lds);
    unsigned short *lu = ly + maxRows * strideCols;
    unsigned short *lv = lu + maxRows * CW;
    int *hy = reinterpret_cast<int *>(lv + maxRows * CW);
    int *hu = hy + (maxRows >> 1) * TW;
    int *hv = hu + (maxRows >> 1) * TW;

    // ================= phase 1: load + input conversion ========================================
    if (a.srcAligned && c0 + nc <= a.srcW)
        scale_phase1<TW, SRCKIND, true>(a, tid, tx0, c0, nc, r0, nr, ngMagic, strideCols, CW, ly, lu, lv, hu, hv);
    else
        scale_phase1<TW, SRCKIND, false>(a, tid, tx0, c0, nc, r0, nr, ngMagic, strideCols, CW, ly, lu, lv, hu, hv);
    __syncthreads();

    // ================= phase 2: horizontal filter (two source rows per item) ====================
    {
        const int xo = tid % TW;
        const int gx = min(tx0 + xo, a.dstW - 1);          // out-of-frame columns recompute the last one
        int lc[kMaxPairs], cc[kMaxPairs];
#pragma unroll
        for (int k = 0; k < kMaxPairs; k++) {
            lc[k] = k < a.hLum.pairs ? a.hLum.packed[(size_t)gx * a.hLum.pairs + k] : 0;
            cc[k] = (!a.chromaDirect && k < a.hChr.pairs) ? a.hChr.packed[(size_t)gx * a.hChr.pairs + k] : 0;
        }
        const int lpos = a.hLum.pos_even[gx] - c0;
        const int cpos = a.hChr.pos_even[gx] - (a.chrHalf ? (c0 >> 1) : c0);
        for (int rp = tid / TW; rp < (nr >> 1); rp += 256 / TW) {
            const int *py0 = reinterpret_cast<const int *>(ly + (2 * rp) * strideCols + lpos);
            const int *py1 = reinterpret_cast<const int *>(ly + (2 * rp + 1) * strideCols + lpos);
            int s0 = 0, s1 = 0;
#pragma unroll
            for (int k = 0; k < kMaxPairs; k++) {
                if (k < a.hLum.pairs) { s0 = dot2(py0[k], lc[k], s0); s1 = dot2(py1[k], lc[k], s1); }
            }
            if (LONG) for (int k = kMaxPairs; k < a.hLum.pairs; k++) {          // filters longer than 16 taps (large ratios)
                const int cf = a.hLum.packed[(size_t)gx * a.hLum.pairs + k];
                s0 = dot2(py0[k], cf, s0); s1 = dot2(py1[k], cf, s1);
            }
            hy[rp * TW + xo] = (int)pack16(min(s0 >> 13, 32767), min(s1 >> 13, 32767));
            if (!a.chromaDirect) {
                const int *pu0 = reinterpret_cast<const int *>(lu + (2 * rp) * CW + cpos);
                const int *pu1 = reinterpret_cast<const int *>(lu + (2 * rp + 1) * CW + cpos);
                const int *pv0 = reinterpret_cast<const int *>(lv + (2 * rp) * CW + cpos);
                const int *pv1 = reinterpret_cast<const int *>(lv + (2 * rp + 1) * CW + cpos);
                int u0 = 0, u1 = 0, v0 = 0, v1 = 0;
#pragma unroll
                for (int k = 0; k < kMaxPairs; k++) {
                    if (k < a.hChr.pairs) {
                        u0 = dot2(pu0[k], cc[k], u0); u1 = dot2(pu1[k], cc[k], u1);
                        v0 = dot2(pv0[k], cc[k], v0); v1 = dot2(pv1[k], cc[k], v1);
                    }
                }
                if (LONG) for (int k = kMaxPairs; k < a.hChr.pairs; k++) {
                    const int cf = a.hChr.packed[(size_t)gx * a.hChr.pairs + k];
                    u0 = dot2(pu0[k], cf, u0); u1 = dot2(pu1[k], cf, u1);
                    v0 = dot2(pv0[k], cf, v0); v1 = dot2(pv1[k], cf, v1);
                }
                hu[rp * TW + xo] = (int)pack16(min(u0 >> 13, 32767), min(u1 >> 13, 32767));
                hv[rp * TW + xo] = (int)pack16(min(v0 >> 13, 32767), min(v1 >> 13, 32767));
            }
        }
    }
    __syncthreads();

    // ================= phase 3: vertical filter + colour stage + store ==========================
    {
        constexpr int QW = TW / 4;                       // 4-pixel groups per tile row
        const int q = tid % QW;
        const int bpp = (a.dstFormat == GMAT_PIX_FMT_RGBA || a.dstFormat == GMAT_PIX_FMT_BGRA) ? 4 : 3;
        const bool swap_rb = a.dstFormat == GMAT_PIX_FMT_BGR24 || a.dstFormat == GMAT_PIX_FMT_BGRA;
        for (int yl = tid / QW; yl < a.TH; yl += 256 / QW) {
            const int yo = ty0 + yl;
            const int xo = tx0 + 4 * q;
            if (yo >= a.dstH || xo >= a.dstW) continue;
            const int vp = (a.vLum.pos_even[yo] - r0) >> 1;
            const int rnd = a.vLum.round[yo];
            int Y[4], U[4], V[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { Y[i] = rnd; U[i] = rnd - (128 << 19); V[i] = U[i]; }
            // vertical taps in chunks of kMaxPairs pairs (one chunk for every filter up to 15 taps)
            for (int k0 = 0; k0 < a.vLum.pairs; k0 += kMaxPairs) {
                int cf[kMaxPairs];
#pragma unroll
                for (int k = 0; k < kMaxPairs; k++)
                    cf[k] = k0 + k < a.vLum.pairs ? a.vLum.packed[(size_t)yo * a.vLum.pairs + k0 + k] : 0;
#pragma unroll
                for (int k = 0; k < kMaxPairs; k++) {
                    if (k0 + k < a.vLum.pairs) {
                        const int o = (vp + k0 + k) * TW + 4 * q;
                        const int4 vy = *reinterpret_cast<const int4 *>(hy + o);
                        const int4 vu = *reinterpret_cast<const int4 *>(hu + o);
                        const int4 vv = *reinterpret_cast<const int4 *>(hv + o);
                        Y[0] = dot2(vy.x, cf[k], Y[0]); Y[1] = dot2(vy.y, cf[k], Y[1]); Y[2] = dot2(vy.z, cf[k], Y[2]); Y[3] = dot2(vy.w, cf[k], Y[3]);
                        U[0] = dot2(vu.x, cf[k], U[0]); U[1] = dot2(vu.y, cf[k], U[1]); U[2] = dot2(vu.z, cf[k], U[2]); U[3] = dot2(vu.w, cf[k], U[3]);
                        V[0] = dot2(vv.x, cf[k], V[0]); V[1] = dot2(vv.y, cf[k], V[1]); V[2] = dot2(vv.z, cf[k], V[2]); V[3] = dot2(vv.w, cf[k], V[3]);
                    }
                }
            }
            unsigned px[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                unsigned c = yuv_to_rgb_full(a.y2r, Y[i] >> 10, U[i] >> 10, V[i] >> 10);
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
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int lds_bytes_for(int TW, int rows, int cols, int chrHalf, int chromaDirect)
{
    const int cw = chromaDirect ? 0 : (chrHalf ? cols / 2 : cols);
    return rows * cols * 2 + 2 * rows * cw * 2 + 3 * (rows / 2) * TW * 4;
}

static int env_int(const char *name, int dflt)
{
    const char *v = ::gmat::knob(name);
    return v && *v ? atoi(v) : dflt;
}

int scale_pick_tiling(const ScalePlan &p, ScaleTiling &t)
{
    if (p.hLum.pairs > 64 || p.hChr.pairs > 64) return GMAT_ERR(ENOSYS);       // 128 taps
    if (p.chrDstW != p.dstW || p.chrDstH != p.dstH) return GMAT_ERR(ENOSYS);   // full-chroma output only
    // the kernel walks ONE vertical bank for the three lines of a pixel: the chroma bank must be the luma bank (it is for every algorithm
    // but SWS_BICUBLIN, whose chroma banks are bilinear: the plane scaler with its RGB loader takes those contexts)
    if (p.vChr.taps != p.vLum.taps || p.vChr.count != p.vLum.count || p.vChr.pos != p.vLum.pos || p.vChr.coef != p.vLum.coef) return GMAT_ERR(ENOSYS);
    const int half = p.chrSrcHSub;
    // identity chroma filter (one tap of 16384 at pos[i] == i): chroma needs no horizontal pass
    int direct = p.hChr.taps == 1 && p.hChr.pairs == 1;
    for (int i = 0; direct && i < p.hChr.count; i++)
        direct = p.hChr.pos[i] == i && p.hChr.coef[i] == 16384;
    if (env_int("GMAT_SCALE_NO_DIRECT", 0)) direct = 0;
    const int forceTW = env_int("GMAT_SCALE_TW", 0), forceTH = env_int("GMAT_SCALE_TH", 0);
    // first the tilings that leave room for several blocks per CU, then (large ratios) anything up to 64 KB
    const int caps[] = {env_int("GMAT_SCALE_LDS_CAP", 40 * 1024), 64 * 1024};
    const int tws[] = {64, 32};
    for (int ldsCap : caps)
    for (int TW : tws) {
        if (forceTW && TW != forceTW) continue;
        // column windows
        const int ntx = (p.dstW + TW - 1) / TW;
        std::vector<int32_t> cs(ntx), cn(ntx);
        int maxCols = 0;
        for (int tc = 0; tc < ntx; tc++) {
            int lo = INT32_MAX, hi = 0;
            for (int x = tc * TW; x < std::min((tc + 1) * TW, p.dstW); x++) {
                lo = std::min(lo, p.hLum.pos_even[x]);
                hi = std::max(hi, p.hLum.pos_even[x] + 2 * p.hLum.pairs);
                lo = std::min(lo, p.hChr.pos_even[x] << half);
                hi = std::max(hi, (p.hChr.pos_even[x] + 2 * p.hChr.pairs) << half);
            }
            lo &= ~3;
            cs[tc] = lo;
            cn[tc] = align_up(hi - lo, 4);
            maxCols = std::max(maxCols, cn[tc]);
        }
        maxCols = align_up(maxCols, 8);
        const int ths[] = {32, 16, 8, 4, 2, 1};
        for (int TH : ths) {
            if (forceTH && TH != forceTH) continue;
            if (!forceTH && TH > 16) continue;           // 32 only on request (tuning)
            const int nty = (p.dstH + TH - 1) / TH;
            std::vector<int32_t> rs(nty), rn(nty);
            int maxRows = 0;
            for (int tr = 0; tr < nty; tr++) {
                int lo = INT32_MAX, hi = 0;
                for (int y = tr * TH; y < std::min((tr + 1) * TH, p.dstH); y++) {
                    lo = std::min(lo, p.vLum.pos_even[y]);
                    hi = std::max(hi, p.vLum.pos_even[y] + 2 * p.vLum.pairs);
                }
                rs[tr] = lo;                              // even by construction
                rn[tr] = align_up(hi - lo, 2);
                maxRows = std::max(maxRows, rn[tr]);
            }
            const int bytes = lds_bytes_for(TW, maxRows, maxCols, half, direct);
            if (bytes > ldsCap && !(forceTH && bytes <= 64 * 1024)) continue;
            t.TW = TW; t.TH = TH; t.ntx = ntx; t.nty = nty;
            t.maxRows = maxRows; t.maxCols = maxCols; t.ldsBytes = bytes;
            t.xcdRemap = env_int("GMAT_SCALE_XCD", 1);
            t.colStart = cs; t.colCount = cn; t.rowStart = rs; t.rowCount = rn;
            t.chromaDirect = direct;
            t.colMagic.resize(ntx);
            for (int i = 0; i < ntx; i++) t.colMagic[i] = (int32_t)(uint32_t)((1ull << 32) / (uint64_t)(cn[i] / 4) + 1);
            return 0;
        }
    }
    return GMAT_ERR(ENOSYS);
}

const char *scale_kernel_name(const ScaleArgs &a, const ScaleTiling &t)
{
    if (t.TW == 64) return a.srcKind ? "scale_rgb_kernel<64,yuv>" : "scale_rgb_kernel<64,rgb>";
    return a.srcKind ? "scale_rgb_kernel<32,yuv>" : "scale_rgb_kernel<32,rgb>";
}

int launch_scale_rgb(const ScaleArgs &a, const ScaleTiling &t, hipStream_t stream)
{
    const int ntiles = t.ntx * t.nty;
    if (ntiles <= 0) return 0;
    const int nblocks = t.xcdRemap ? 8 * ((ntiles + 7) / 8) : ntiles;
    const dim3 grid(nblocks), block(256);
    const size_t lds = (size_t)t.ldsBytes;
    const bool longH = a.hLum.pairs > kMaxPairs || a.hChr.pairs > kMaxPairs;
#define GMAT_LAUNCH_SCALE(TW_, KIND_)                                                                              \
    do { if (longH) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_rgb_kernel<TW_, KIND_, true>), grid, block, lds, stream, a, \
                                       t.maxCols, t.maxRows);                                                          \
         else       hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_rgb_kernel<TW_, KIND_, false>), grid, block, lds, stream, a, \
                                       t.maxCols, t.maxRows); } while (0)
    if (t.TW == 64) { if (a.srcKind) GMAT_LAUNCH_SCALE(64, 1); else GMAT_LAUNCH_SCALE(64, 0); }
    else if (t.TW == 32) { if (a.srcKind) GMAT_LAUNCH_SCALE(32, 1); else GMAT_LAUNCH_SCALE(32, 0); }
    else return GMAT_ERR(EINVAL);
#undef GMAT_LAUNCH_SCALE
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
