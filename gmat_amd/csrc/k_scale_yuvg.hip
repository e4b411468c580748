// k_scale_yuvg.hip — ONE polyphase band walker for 8-bit YUV 4:2:0 sources at ANY scaling ratio (gfx950).
//
// libswscale's single-context semantics (swscale.c:234-520): every plane scaled separately — hScale8To15_c (swscale.c:122-136),
// the vertical filters of vscale.c / output.c, the LUT colour stage of yuv2rgb.c — bit-exact, for whatever tables initFilter
// (utils.c:367-763) produced: any ratio, any SWS algorithm, borders folded the way the table says.  Round 2 served the exact
// ratios 2:1, 3:1, 3:2, 4:1 and 1:2 with one hand-specialised walker each and everything else (4K -> 900p, 1080p -> 432p ...)
// with the tiled kernel of round 1 at 0.10-0.14 of the HBM roofline.  This kernel needs no structure in
// the ratio:
//   * a LANE owns one output COLUMN of a plane for a band of output rows.  Its horizontal taps are therefore loop-invariant: the
//     coefficient pairs sit in registers, the source window starts at the lane's own 4-byte aligned address pos[x] & ~3, and the
//     byte pairs come out of the loaded dwords with v_perm_b32 through two per-lane selector registers (the window's byte phase
//     pos[x] & 3 is the only thing that differs between lanes; the dword indices are static).  No LDS, no positions in the loop.
//   * the vertical filter runs as K running sums per lane: source rows are consumed in pairs (cvt_pk = hScale8To15_c's
//     saturation), a pair feeds each of the K open output rows with one v_dot2 whose coefficient pair is WAVE-UNIFORM (one scalar
//     load per open row and pair from a table the host lays out per source row pair); when the first open row has seen its last
//     source row it leaves through the output stage and the sums shift down by one.  Multi-row closes (up-scaling axes, e.g. the
//     chroma of an RGB destination at ratios below 2:1) fall out of the loop structure: the next row is simply complete already.
//   * bands are short (raster-like order, see k_scale_yuv2s.hip's launcher) and odd bands walk upward through mirrored tables.
// Packed RGB destinations: even lanes filter the U sample of their column pair, odd lanes the V sample (the chroma plane of an
// RGB destination has half the output width), one lane exchange per output row.  4:2:0 destinations: plane jobs of the same
// walker, luma and chroma workgroups in one launch.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

namespace gmat {

// ---- a plane as a raw buffer resource: lane offset in a loop-invariant VGPR, row offset in the instruction's scalar offset,
//      reads past the plane's last byte return 0 (the windows are whole dwords and may overhang the last row by up to 7 bytes)
struct GPlane {
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ GPlane(const uint8_t *p, unsigned bytes) : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p), 0, bytes, 0x00020000)) {}
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    __device__ __forceinline__ void ld4(unsigned lane, unsigned row, unsigned *w) const { const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, lane, row, 0); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
    __device__ __forceinline__ void ld2(unsigned lane, unsigned row, unsigned *w) const { const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, lane, row, 0); w[0] = v.x; w[1] = v.y; }
    __device__ __forceinline__ void ld1(unsigned lane, unsigned row, unsigned *w) const { w[0] = __builtin_amdgcn_raw_buffer_load_b32(r, lane, row, 0); }
    __device__ __forceinline__ void st1(unsigned d, unsigned lane, unsigned row) const { __builtin_amdgcn_raw_buffer_store_b32(d, r, lane, row, 0); }
#else
    // hipcc's host pass (never executed) and the CPU emulation of the test suite
    uint8_t *p; unsigned n;
    __host__ __device__ GPlane(const uint8_t *q, unsigned bytes) : p(const_cast<uint8_t *>(q)), n(bytes) {}
    __host__ __device__ unsigned dw(size_t o) const { unsigned v = 0; if (o + 4 <= n) std::memcpy(&v, p + o, 4); return v; }
    __host__ __device__ void ld4(unsigned lane, unsigned row, unsigned *w) const { for (int i = 0; i < 4; i++) w[i] = dw((size_t)row + lane + 4 * i); }
    __host__ __device__ void ld2(unsigned lane, unsigned row, unsigned *w) const { for (int i = 0; i < 2; i++) w[i] = dw((size_t)row + lane + 4 * i); }
    __host__ __device__ void ld1(unsigned lane, unsigned row, unsigned *w) const { w[0] = dw((size_t)row + lane); }
    __host__ __device__ void st1(unsigned d, unsigned lane, unsigned row) const { if ((size_t)row + lane + 4 <= n) std::memcpy(p + (size_t)row + lane, &d, 4); }
#endif
    template <int NW> __device__ __forceinline__ void ld(unsigned lane, unsigned row, unsigned (&w)[NW]) const
    {
        static_assert(NW >= 2 && NW <= 12, "window dwords");
        int i = 0;
#pragma unroll
        for (; i + 4 <= NW; i += 4) ld4(lane + 4u * i, row, w + i);
        if (NW - i >= 2) { ld2(lane + 4u * i, row, w + i); i += 2; }
        if (NW - i == 1) ld1(lane + 4u * i, row, w + i);
    }
};

__device__ __forceinline__ int g_dot2(int ab, int cd, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, ab), __builtin_bit_cast(short2v, cd), acc, true);
}
__device__ __forceinline__ unsigned g_sat_pk_u8_i16(unsigned v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(v));
    return r;
#else
    const int lo = (int16_t)(v & 0xFFFFu), hi = (int16_t)(v >> 16);
    return (unsigned)std::min(std::max(lo, 0), 255) | ((unsigned)std::min(std::max(hi, 0), 255) << 8);
#endif
}

// window dwords of a lane: component stride 1 (a luma or planar chroma row): byte pairs (o + 2t, o + 2t + 1), t < P, o = pos & 3;
// component stride 2 (one component of NV12's UV row): (o + 4t, o + 4t + 2), o = (2 pos + comp) & 3
template <int P, bool S2> struct GWin { static constexpr int NW = S2 ? P + 1 : ((P - 1) >> 1) + 2; };

// one horizontally filtered sample of this lane's column: hScale8To15_c's sum (before its >> 7)
template <int P, bool S2>
__device__ __forceinline__ int g_hsum(const unsigned (&w)[GWin<P, S2>::NW], const int (&cf)[P], unsigned selE, unsigned selO)
{
    int s = 0;
#pragma unroll
    for (int t = 0; t < P; t++) {
        const int pr = S2 ? (int)__builtin_amdgcn_perm(w[t + 1], w[t], selE)
                          : (int)__builtin_amdgcn_perm(w[(t >> 1) + 1], w[t >> 1], (t & 1) ? selO : selE);
        s = g_dot2(pr, cf[t], s);
    }
    return s;
}

// One axis-pair of a lane: horizontal window + coefficients, K vertical running sums, the vertical program of its plane class.
template <int P, int K, bool S2>
struct GWalk {
    static constexpr int NW = GWin<P, S2>::NW;
    int cf[P];
    unsigned selE, selO, voff;
    int acc[K];
    unsigned cur[2][NW];                 // the row pair about to be consumed (requested one step ahead)
    int m;                               // next row pair (walking coordinates)
    // plane + program (wave-uniform)
    const int32_t *vcoef, *vlast, *vround;
    int srcRows, stride, rows, up;

    __device__ __forceinline__ void setup(const int32_t *hTab, const int32_t *posTab, int col, int comp)
    {
        const int pos = uniform_or_lane(posTab, col);
        const int b0 = S2 ? 2 * pos + comp : pos;
        const unsigned o = (unsigned)b0 & 3u;
        voff = (unsigned)b0 & ~3u;
        selE = S2 ? (0x0C000C00u | o | ((o + 2) << 16)) : (0x0C000C00u | o | ((o + 1) << 16));
        selO = 0x0C000C00u | (o + 2) | ((o + 3) << 16);
#pragma unroll
        for (int t = 0; t < P; t++) cf[t] = hTab[(size_t)col * P + t];
    }
    static __device__ __forceinline__ int uniform_or_lane(const int32_t *t, int i) { return t[i]; }

    __device__ __forceinline__ unsigned row_off(int rw) const      // walking-coordinate source row -> byte offset of the actual row
    {
        const int r = min(rw, srcRows - 1);
        return (unsigned)(up ? srcRows - 1 - r : r) * (unsigned)stride;
    }
    __device__ __forceinline__ void request(const GPlane &pl, int mm)
    {
        pl.template ld<NW>(voff, row_off(2 * mm), cur[0]);
        pl.template ld<NW>(voff, row_off(2 * mm + 1), cur[1]);
    }
    __device__ __forceinline__ void init_acc(int y)
    {
#pragma unroll
        for (int i = 0; i < K; i++) acc[i] = y + i < rows ? uniform_load(vround, y + i) : 0;
    }
    // consume the requested pair m, request pair m + 1 (req: which plane this lane reads — a lane-parity choice for planar chroma
    // under an RGB destination, so that only the LOADS diverge, never the arithmetic)
    template <class Req> __device__ __forceinline__ void step(Req &&req)
    {
        unsigned a[NW], b[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) { a[i] = cur[0][i]; b[i] = cur[1][i]; }
        const int32_t *vc = vcoef + (size_t)m * K;
        m++;
        req(m);
        const int h0 = g_hsum<P, S2>(a, cf, selE, selO) >> 7, h1 = g_hsum<P, S2>(b, cf, selE, selO) >> 7;
        const int hp = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(h0, h1));      // min(., 32767) of hScale8To15_c
#pragma unroll
        for (int i = 0; i < K; i++) acc[i] = g_dot2(hp, uniform_load(vc, i), acc[i]);
    }
    __device__ __forceinline__ void shift(int y)                  // row y has left: slot i now stands for row y + 1 + i
    {
#pragma unroll
        for (int i = 0; i + 1 < K; i++) acc[i] = acc[i + 1];
        acc[K - 1] = y + K < rows ? uniform_load(vround, y + K) : 0;
    }
};

// ---- packed RGB destinations --------------------------------------------------------------------------------------------------
// block = 4 waves = 4 adjacent strips of 64 output columns of one band; grid.y = frame
template <int P, int K, bool NV12>
__global__ __launch_bounds__(256) void scale_yuvg_rgb_kernel(YuvGArgs a, Yuv2xFrames fr)
{
    __shared__ int2 lutV[256], lutU[256];
    __shared__ unsigned stage[4][66];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const Yuv2RgbConsts &k = a.y2r;
        const int i = tid;
        lutV[i] = make_int2(k.base + m24(k.offR + (m24(i, k.crv) >> 16), k.cy), m24(m24(i, k.cgv) >> 16, k.cy));
        lutU[i] = make_int2(k.base + m24(k.offG + (m24(i, k.cgu) >> 16), k.cy), k.base + m24(k.offB + (m24(i, k.cbu) >> 16), k.cy));
    }
    __syncthreads();
    int lin = blockIdx.x;
    if (a.xcdRemap) { const int chunk = (a.nblk + 7) >> 3; lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3); }
    if (lin >= a.nblk) return;
    const int band = __builtin_amdgcn_readfirstlane(lin / a.nsg);
    const int X0 = ((lin - band * a.nsg) * 4 + wave) * 64;
    if (X0 >= a.dstW) return;
    const int up = a.updown & band & 1;
    const int y0 = band * a.bandRows, y1 = min(y0 + a.bandRows, a.dstH);
    const int ya = up ? a.dstH - y1 : y0, yb = up ? a.dstH - y0 : y1;            // walking coordinates
    const int f = blockIdx.y;
    // exact valid bytes of each plane (row bytes are multiples of 4 by the host rule): a window dword past them reads as 0
    const unsigned crb = (unsigned)(NV12 ? 2 * a.chrSrcW : a.chrSrcW);
    const GPlane bY(fr.y[f], (unsigned)a.ys * (unsigned)(a.srcH - 1) + (unsigned)a.srcW);
    const GPlane bU(fr.u[f], (unsigned)a.us * (unsigned)(a.chrSrcH - 1) + crb), bV(NV12 ? fr.u[f] : fr.v[f], (unsigned)a.us * (unsigned)(a.chrSrcH - 1) + crb);
    const int bpp = (a.dstFormat == GMAT_PIX_FMT_RGBA || a.dstFormat == GMAT_PIX_FMT_BGRA) ? 4 : 3;
    const bool bgr = a.dstFormat == GMAT_PIX_FMT_BGR24 || a.dstFormat == GMAT_PIX_FMT_BGRA;
    const GPlane bD(fr.dst[f], (unsigned)a.ds * (unsigned)(a.dstH - 1) + (unsigned)(a.dstW * bpp));

    const int x = X0 + lane, xc = min(x, a.dstW - 1), par = lane & 1;
    GWalk<P, K, false> L;
    GWalk<P, K, NV12> C;
    L.setup(a.hL, a.posL, xc, 0);
    C.setup(a.hC, a.posC, min(xc >> 1, a.chrDstW - 1), par);
    L.vcoef = a.vcoefL[up]; L.vlast = a.vlastL[up]; L.vround = a.vroundL[up]; L.srcRows = a.srcH; L.stride = a.ys; L.rows = a.dstH; L.up = up;
    C.vcoef = a.vcoefC[up]; C.vlast = a.vlastC[up]; C.vround = a.vroundC[up]; C.srcRows = a.chrSrcH; C.stride = a.us; C.rows = a.dstH; C.up = up;   // planar: us == vs (host rule)
    // the first pairs this band needs, and the first row whose sums must be tracked from their start
    L.m = uniform_load(a.vfirstL[up], ya); C.m = uniform_load(a.vfirstC[up], ya);
    int y = min(uniform_load(a.vyLoL[up], L.m), uniform_load(a.vyLoC[up], C.m));
    L.init_acc(y); C.init_acc(y);
    auto reqL = [&](int mm) { L.request(bY, mm); };
    auto reqC = [&](int mm) { if (NV12 || par == 0) C.request(bU, mm); else C.request(bV, mm); };
    reqL(L.m);
    reqC(C.m);
    const unsigned dsel = (unsigned)(lane % 3 == 0 ? 0x04020100u : lane % 3 == 1 ? 0x05040201u : 0x06050402u);
    const int p0 = min((4 * lane) / 3, 62);                      // lanes >= 48 store nothing
    for (; y < yb; y++) {
        const int lastL = uniform_load(L.vlast, y), lastC = uniform_load(C.vlast, y);
        while (L.m <= lastL) L.step(reqL);
        while (C.m <= lastC) C.step(reqC);
        if (y >= ya) {
            const int Y = L.acc[0] >> 19;
            const int mine = clip_u8_shr(C.acc[0], 19), other = __shfl_xor(mine, 1);
            const int U = par ? other : mine, V = par ? mine : other;
            const int2 tv = lutV[V], tu = lutU[U];
            const int ycy = m24(Y, a.y2r.cy);
            const unsigned cr = (unsigned)((bgr ? tu.y : tv.x) + ycy), cg = (unsigned)(tv.y + tu.x + ycy), cb = (unsigned)((bgr ? tv.x : tu.y) + ycy);
            const unsigned rg = g_sat_pk_u8_i16(__builtin_amdgcn_perm(cg, cr, 0x07060302u));
            const unsigned ba = g_sat_pk_u8_i16(__builtin_amdgcn_perm(0x00FF0000u, cb, 0x07060302u));
            const unsigned px = __builtin_amdgcn_perm(ba, rg, 0x05040100u);
            const unsigned drow = (unsigned)(up ? a.dstH - 1 - y : y) * (unsigned)a.ds;
            if (bpp == 4) {
                if (x < a.dstW) bD.st1(px, 4u * (unsigned)x, drow);
            } else {
                // 64 pixels x 3 bytes = 48 dwords: through the wave's own LDS row (no barrier: one wave, in-order LDS)
                stage[wave][lane] = px;
                __builtin_amdgcn_wave_barrier();
                const unsigned d0 = stage[wave][p0], d1 = stage[wave][p0 + 1];
                __builtin_amdgcn_wave_barrier();
                const unsigned o = __builtin_amdgcn_perm(d1, d0, dsel);
                const int nb = 3 * min(64, a.dstW - X0);                 // bytes of this strip's row
                if (4 * lane + 4 <= nb) bD.st1(o, 3u * (unsigned)X0 + 4u * (unsigned)lane, drow);
                else if (4 * lane < nb) {                                 // a width that is not a multiple of 4: the last bytes one by one
                    uint8_t *d = fr.dst[f] + (size_t)drow + 3u * (unsigned)X0 + 4u * (unsigned)lane;
                    for (int i = 0; i < nb - 4 * lane; i++) d[i] = (uint8_t)(o >> (8 * i));
                }
            }
        }
        L.shift(y); C.shift(y);
    }
}

// ---- 4:2:0 destinations: plane jobs ----------------------------------------------------------------------------------------------
// job 0: the luma plane (lane = column).  job 1: chroma — NV12 -> NV12: lane = (column, component) of the interleaved plane;
// planar -> planar: two jobs (U, V), lane = column.  Blocks [0, nblkL) are luma, the rest chroma.
template <int P, int K, bool NV12>
__global__ __launch_bounds__(256) void scale_yuvg_planes_kernel(YuvGArgs a, Yuv2xFrames fr)
{
    __shared__ unsigned stage[4][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lin = blockIdx.x;
    if (a.xcdRemap) { const int chunk = (a.nblk + 7) >> 3; lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3); }
    if (lin >= a.nblk) return;
    const int f = blockIdx.y;
    // which job
    int job = 0, rel = lin;
    if (lin >= a.nblkL) { rel = lin - a.nblkL; job = 1; if (!NV12 && rel >= a.nblkC) { rel -= a.nblkC; job = 2; } }
    const int nsg = job ? a.nsgC : a.nsg;
    const int band = __builtin_amdgcn_readfirstlane(rel / nsg);
    const int B0 = ((rel - band * nsg) * 4 + wave) * 64;             // first BYTE column of this wave in the destination plane row
    const int rowBytes = job == 0 ? a.dstW : NV12 ? 2 * a.chrDstW : a.chrDstW;
    if (B0 >= rowBytes) return;
    const int rows = job ? a.chrDstH : a.dstH, srcRows = job ? a.chrSrcH : a.srcH;
    const int up = a.updown & band & 1;
    const int bandRows = job ? a.bandRowsC : a.bandRows;
    const int y0 = band * bandRows, y1 = min(y0 + bandRows, rows);
    const int ya = up ? rows - y1 : y0, yb = up ? rows - y0 : y1;
    const uint8_t *sp = job == 0 ? fr.y[f] : job == 1 ? fr.u[f] : fr.v[f];
    uint8_t *dp = job == 0 ? fr.dst[f] : job == 1 ? fr.dstU[f] : fr.dstV[f];
    const int ss = job == 0 ? a.ys : job == 1 ? a.us : a.vs, dstride = job == 0 ? a.ds : job == 1 ? a.dsU : a.dsV;
    const int srcRowBytes = job == 0 ? a.srcW : NV12 ? 2 * a.chrSrcW : a.chrSrcW;
    const GPlane bS(sp, (unsigned)ss * (unsigned)(srcRows - 1) + (unsigned)srcRowBytes), bD(dp, (unsigned)dstride * (unsigned)(rows - 1) + (unsigned)rowBytes);
    const int bcol = min(B0 + lane, rowBytes - 1);
    auto run = [&](auto s2_c) {
        constexpr bool S2 = decltype(s2_c)::value;
        GWalk<P, K, S2> W;
        W.setup(job ? a.hC : a.hL, job ? a.posC : a.posL, S2 ? bcol >> 1 : bcol, S2 ? bcol & 1 : 0);
        W.vcoef = job ? a.vcoefC[up] : a.vcoefL[up]; W.vlast = job ? a.vlastC[up] : a.vlastL[up]; W.vround = job ? a.vroundC[up] : a.vroundL[up];
        W.srcRows = srcRows; W.stride = ss; W.rows = rows; W.up = up;
        W.m = uniform_load(job ? a.vfirstC[up] : a.vfirstL[up], ya);
        int y = uniform_load(job ? a.vyLoC[up] : a.vyLoL[up], W.m);
        W.init_acc(y);
        auto req = [&](int mm) { W.request(bS, mm); };
        req(W.m);
        for (; y < yb; y++) {
            const int last = uniform_load(W.vlast, y);
            while (W.m <= last) W.step(req);
            if (y >= ya) {
                // yuv2planeX_8_c / yuv2nv12cX_c: clip_u8((dither << 12 + sum) >> 19), the dither in the row's start value
                const unsigned v = (unsigned)clip_u8_shr(W.acc[0], 19);
                const unsigned drow = (unsigned)(up ? rows - 1 - y : y) * (unsigned)dstride;
                uint8_t *sb = reinterpret_cast<uint8_t *>(stage[wave]);
                sb[lane] = (uint8_t)v;
                __builtin_amdgcn_wave_barrier();
                const unsigned o = stage[wave][lane & 15];
                __builtin_amdgcn_wave_barrier();
                const int nb = min(64, rowBytes - B0);
                if (lane < 16) {
                    if (4 * lane + 4 <= nb) bD.st1(o, (unsigned)B0 + 4u * (unsigned)lane, drow);
                    else if (4 * lane < nb) {
                        uint8_t *d = dp + (size_t)drow + (unsigned)B0 + 4u * (unsigned)lane;
                        for (int i = 0; i < nb - 4 * lane; i++) d[i] = (uint8_t)(o >> (8 * i));
                    }
                }
            }
            W.shift(y);
        }
    };
    if (NV12 && job == 1) run(std::true_type()); else run(std::false_type());
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// The vertical program of one plane class in one walking direction (up: everything mirrored, so the kernel always counts upward):
//   first[y]    row pair in which output row y's window starts
//   last[y]     row pair holding its last source row
//   yLo[m]      first output row not complete before pair m (the row slot 0 of the running sums stands for while pair m is consumed)
//   coef[m][i]  int16 pair (tap on source row 2m, tap on 2m + 1) of output row yLo[m] + i, zero outside its window
static int build_vprog(const FilterBank &fb, const std::vector<int32_t> &round, int srcRows, bool up, YuvGVProg &v)
{
    const int rows = fb.count, taps = fb.taps;
    std::vector<int> pos(rows);
    std::vector<int16_t> cf((size_t)rows * taps);
    for (int y = 0; y < rows; y++) {
        const int ys = up ? rows - 1 - y : y;
        // effective window without leading / trailing zero taps would shorten the sums' lifetime; keep the table's own window
        pos[y] = up ? srcRows - (fb.pos[ys] + taps) : fb.pos[ys];
        for (int t = 0; t < taps; t++) cf[(size_t)y * taps + t] = fb.coef[(size_t)ys * taps + (up ? taps - 1 - t : t)];
        if (pos[y] < 0 || pos[y] + taps > srcRows + 1) return 0;
    }
    for (int y = 1; y < rows; y++) if (pos[y] < pos[y - 1]) return 0;            // the walk needs monotone windows
    const int M = (srcRows + 1) / 2 + 1;
    v.first.assign(rows, 0); v.last.assign(rows, 0); v.round.assign(rows, 0);
    for (int y = 0; y < rows; y++) {
        v.first[y] = pos[y] >> 1;
        v.last[y] = (pos[y] + taps - 1) >> 1;
        v.round[y] = round[up ? rows - 1 - y : y];
    }
    v.yLo.assign(M + 1, rows);
    {
        int y = 0;
        for (int m = 0; m <= M; m++) { while (y < rows && v.last[y] < m) y++; v.yLo[m] = y; }
    }
    int K = 1;
    for (int m = 0; m < M; m++) {
        int hi = v.yLo[m] - 1;
        for (int y = v.yLo[m]; y < rows && v.first[y] <= m; y++) hi = y;
        K = std::max(K, hi - v.yLo[m] + 1);
    }
    v.K = K; v.M = M; v.rows = rows;
    v.pos = pos; v.cf = cf; v.taps = taps;
    return 1;
}
static void fill_vcoef(YuvGVProg &v, int K)
{
    v.coef.assign((size_t)(v.M + 1) * K, 0);
    for (int m = 0; m < v.M; m++)
        for (int i = 0; i < K; i++) {
            const int y = v.yLo[m] + i;
            if (y >= v.rows) break;
            int lo = 0, hi = 0;
            const int t0 = 2 * m - v.pos[y], t1 = t0 + 1;
            if (t0 >= 0 && t0 < v.taps) lo = v.cf[(size_t)y * v.taps + t0];
            if (t1 >= 0 && t1 < v.taps) hi = v.cf[(size_t)y * v.taps + t1];
            v.coef[(size_t)m * K + i] = (int32_t)((uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16));
        }
}

static const int kGP[] = {4, 6, 8, 10}, kGK[] = {4, 6, 7, 9};

int yuvg_prepare(const ScalePlan &p, const YuvScaleTiling &g, YuvGTables &t)
{
    t = YuvGTables();
    const char *off = getenv("GMAT_SCALE_NO_GENERIC_WALKER");
    if (off && atoi(off)) return 0;
    const bool rgbOut = p.dstFormat == GMAT_PIX_FMT_RGB24 || p.dstFormat == GMAT_PIX_FMT_BGR24 || p.dstFormat == GMAT_PIX_FMT_RGBA || p.dstFormat == GMAT_PIX_FMT_BGRA;
    const bool yuvOut = p.dstFormat == GMAT_PIX_FMT_NV12 || p.dstFormat == GMAT_PIX_FMT_YUV420P;
    if (!(p.srcFormat == GMAT_PIX_FMT_NV12 || p.srcFormat == GMAT_PIX_FMT_YUV420P) || !(rgbOut || yuvOut)) return 0;
    if (rgbOut && (g.fullChroma || g.yuvOut)) return 0;
    if (yuvOut && g.yuvOut != 1) return 0;
    if (yuvOut && ((p.srcFormat == GMAT_PIX_FMT_NV12) != (p.dstFormat == GMAT_PIX_FMT_NV12))) return 0;   // same chroma layout on both sides
    if (p.dstW < 16 || p.dstH < 8 || p.srcW < 16 || p.srcH < 8) return 0;
    // whole dwords inside every source row (the windows are dword loads checked against the plane's exact size)
    if (p.srcW % 4 || (p.srcFormat == GMAT_PIX_FMT_NV12 ? (2 * p.chrSrcW) % 4 : p.chrSrcW % 4)) return 0;
    // RGB: one chroma sample per pixel pair and per output row (the LUT form); 4:2:0: the chroma planes of the destination
    if (rgbOut && (p.chrDstW != (p.dstW + 1) / 2 || p.chrDstH != p.dstH)) return 0;
    // horizontal: coefficient pairs on the table's own windows (a window may start anywhere; the last pair of an odd tap count is padded)
    auto hpack = [&](const FilterBank &fb, int srcLen, std::vector<int32_t> &out, int P) {
        out.assign((size_t)fb.count * P, 0);
        for (int x = 0; x < fb.count; x++) {
            if (fb.pos[x] < 0 || fb.pos[x] + fb.taps > srcLen) return false;
            for (int k = 0; k < P; k++) {
                const int t0 = 2 * k, t1 = t0 + 1;
                const int lo = t0 < fb.taps ? fb.coef[(size_t)x * fb.taps + t0] : 0, hi = t1 < fb.taps ? fb.coef[(size_t)x * fb.taps + t1] : 0;
                out[(size_t)x * P + k] = (int32_t)((uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16));
            }
        }
        return true;
    };
    const int needP = (std::max(p.hLum.taps, p.hChr.taps) + 1) / 2;
    int P = 0;
    for (int c : kGP) if (c >= needP) { P = c; break; }
    if (!P) return 0;
    if (!hpack(p.hLum, p.srcW, t.hL, P) || !hpack(p.hChr, p.chrSrcW, t.hC, P)) return 0;
    t.posL = p.hLum.pos; t.posC = p.hChr.pos;
    for (int up = 0; up < 2; up++) {
        if (!build_vprog(g.vLumEff, g.lumRound, p.srcH, up != 0, t.vL[up])) return 0;
        if (!build_vprog(g.vChrEff, g.chrRound, p.chrSrcH, up != 0, t.vC[up])) return 0;
    }
    const int needK = std::max(std::max(t.vL[0].K, t.vL[1].K), std::max(t.vC[0].K, t.vC[1].K));
    int K = 0;
    for (int c : kGK) if (c >= needK) { K = c; break; }
    if (!K) { if (getenv("GMAT_DEBUG_WALKER")) logf(LOG_ERROR, "yuvg: %dx%d -> %dx%d declined: K needed %d", p.srcW, p.srcH, p.dstW, p.dstH, needK); return 0; }
    for (int up = 0; up < 2; up++) { fill_vcoef(t.vL[up], K); fill_vcoef(t.vC[up], K); }
    t.P = P; t.K = K; t.yuvOut = yuvOut;
    if (getenv("GMAT_DEBUG_WALKER")) logf(LOG_ERROR, "yuvg: %dx%d -> %dx%d taps h %d/%d v %d/%d -> P %d, K needed %d (L %d/%d C %d/%d) -> %d", p.srcW, p.srcH, p.dstW, p.dstH,
                                          p.hLum.taps, p.hChr.taps, g.vLumEff.taps, g.vChrEff.taps, P, needK, t.vL[0].K, t.vL[1].K, t.vC[0].K, t.vC[1].K, K);
    t.ok = 1;
    return 0;
}

int launch_scale_yuvg(const YuvGArgs &a0, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (!frames || nframes < 1 || nframes > kYuv2xMaxFrames) return GMAT_ERR(EINVAL);
    YuvGArgs a = a0;
    const char *rowsStr = getenv("GMAT_STRIP_ROWS");          // tuning / test override, read per launch
    const int rowsEnv = rowsStr ? atoi(rowsStr) : 0;
    const char *ud = getenv("GMAT_STRIP_UPDOWN");
    a.updown = !(ud && !atoi(ud));
    const int nstrips = (a.dstW + 63) / 64;
    a.nsg = (nstrips + 3) / 4;
    // band height: short bands in raster order (see k_scale_yuv2s.hip's launcher); a lone small frame wants enough waves to fill the chip
    const long wr = (long)a.dstH * nstrips * nframes;
    int rows = rowsEnv > 0 ? rowsEnv : (int)std::min(16L, std::max(4L, (wr + 6143) / 6144));
    a.bandRows = rows;
    a.nbands = (a.dstH + rows - 1) / rows;
    a.nblkL = a.nbands * a.nsg;
    a.nblk = a.nblkL;
    if (a.yuvOut) {
        const int cbytes = a.nv12 ? 2 * a.chrDstW : a.chrDstW;
        a.nsgC = ((cbytes + 63) / 64 + 3) / 4;
        a.bandRowsC = std::max(2, rows / 2);
        a.nbandsC = (a.chrDstH + a.bandRowsC - 1) / a.bandRowsC;
        a.nblkC = a.nbandsC * a.nsgC;
        a.nblk = a.nblkL + (a.nv12 ? 1 : 2) * a.nblkC;
    }
    const dim3 grid(a.xcdRemap ? 8 * ((a.nblk + 7) / 8) : a.nblk, nframes), block(256);
    const Yuv2xFrames &fr = *frames;
#define GMAT_G_K(P_, K_) do { \
        if (a.yuvOut) { if (a.nv12) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_planes_kernel<P_, K_, true>), grid, block, 0, stream, a, fr); \
                        else        hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_planes_kernel<P_, K_, false>), grid, block, 0, stream, a, fr); } \
        else          { if (a.nv12) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_rgb_kernel<P_, K_, true>), grid, block, 0, stream, a, fr); \
                        else        hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_rgb_kernel<P_, K_, false>), grid, block, 0, stream, a, fr); } } while (0)
#define GMAT_G_P(P_) do { switch (a.K) { case 4: GMAT_G_K(P_, 4); break; case 6: GMAT_G_K(P_, 6); break; case 7: GMAT_G_K(P_, 7); break; default: GMAT_G_K(P_, 9); } } while (0)
    switch (a.P) { case 4: GMAT_G_P(4); break; case 6: GMAT_G_P(6); break; case 8: GMAT_G_P(8); break; default: GMAT_G_P(10); }
#undef GMAT_G_P
#undef GMAT_G_K
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
