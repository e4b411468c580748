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

struct Px4 { int r[4], g[4], b[4]; };

// ---- phase 1 helpers -------------------------------------------------------------------------
__device__ __forceinline__ void load_rgb4(const ScaleArgs &a, int srow, int col, Px4 &p)
{
    const uint8_t *row = a.src0 + (size_t)srow * a.ss0;
    if (a.srcAligned && col + 4 <= a.srcW) {
        const uint3 v = *reinterpret_cast<const uint3 *>(row + (size_t)col * 3);
        p.r[0] = v.x & 0xFF;         p.g[0] = (v.x >> 8) & 0xFF;  p.b[0] = (v.x >> 16) & 0xFF;
        p.r[1] = v.x >> 24;          p.g[1] = v.y & 0xFF;         p.b[1] = (v.y >> 8) & 0xFF;
        p.r[2] = (v.y >> 16) & 0xFF; p.g[2] = v.y >> 24;          p.b[2] = v.z & 0xFF;
        p.r[3] = (v.z >> 8) & 0xFF;  p.g[3] = (v.z >> 16) & 0xFF; p.b[3] = v.z >> 24;
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = min(col + i, a.srcW - 1);
            p.r[i] = row[3 * c]; p.g[i] = row[3 * c + 1]; p.b[i] = row[3 * c + 2];
        }
    }
    if (a.srcBgr) {
#pragma unroll
        for (int i = 0; i < 4; i++) { const int t = p.r[i]; p.r[i] = p.b[i]; p.b[i] = t; }
    }
}

__device__ __forceinline__ void load_yuv4(const ScaleArgs &a, int srow, int col, Px4 &p)
{
    const uint8_t *yrow = a.src0 + (size_t)srow * a.ss0;
    const size_t crow = (size_t)(srow >> 1);
    int Y[4], U[2], V[2];
    if (a.srcAligned && col + 4 <= a.srcW) {
        const unsigned y4 = *reinterpret_cast<const unsigned *>(yrow + col);
        Y[0] = y4 & 0xFF; Y[1] = (y4 >> 8) & 0xFF; Y[2] = (y4 >> 16) & 0xFF; Y[3] = y4 >> 24;
        if (a.srcNv12) {
            const unsigned uv = *reinterpret_cast<const unsigned *>(a.src1 + crow * a.ss1 + col);
            U[0] = uv & 0xFF; V[0] = (uv >> 8) & 0xFF; U[1] = (uv >> 16) & 0xFF; V[1] = uv >> 24;
        } else {
            const unsigned short uu = *reinterpret_cast<const unsigned short *>(a.src1 + crow * a.ss1 + (col >> 1));
            const unsigned short vv = *reinterpret_cast<const unsigned short *>(a.src2 + crow * a.ss2 + (col >> 1));
            U[0] = uu & 0xFF; U[1] = uu >> 8; V[0] = vv & 0xFF; V[1] = vv >> 8;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) Y[i] = yrow[min(col + i, a.srcW - 1)];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int cc = min(col + 2 * i, a.srcW - 1) >> 1;
            if (a.srcNv12) {
                const uint8_t *q = a.src1 + crow * a.ss1 + 2 * cc;
                U[i] = q[0]; V[i] = q[1];
            } else {
                U[i] = a.src1[crow * a.ss1 + cc];
                V[i] = a.src2[crow * a.ss2 + cc];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const ChromaTerms t = chroma_terms(a.y2r, U[i], V[i]);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int ycy = Y[2 * i + j] * a.y2r.cy;
            p.r[2 * i + j] = luma_chan(t.r, ycy);
            p.g[2 * i + j] = luma_chan(t.g, ycy);
            p.b[2 * i + j] = luma_chan(t.b, ycy);
        }
    }
}

__device__ __forceinline__ unsigned pack16(int lo, int hi) { return ((unsigned)lo & 0xFFFF) | ((unsigned)hi << 16); }

template <int TW, int SRCKIND>
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
            if ((int)(blockIdx.x >> 3) >= chunk) return;
        }
        if (lin >= ntiles) return;
        tcol = lin / a.nty;
        trow = lin - tcol * a.nty;
    }
    const int tid = threadIdx.x;
    const int tx0 = tcol * TW, ty0 = trow * a.TH;
    const int c0 = a.colStart[tcol], nc = a.colCount[tcol];
    const int r0 = a.rowStart[trow], nr = a.rowCount[trow];

    const int CW = a.chrHalf ? (strideCols >> 1) : strideCols;      // chroma row stride (samples)
    unsigned short *ly = reinterpret_cast<unsigned short *>(lds);
    unsigned short *lu = ly + maxRows * strideCols;
    unsigned short *lv = lu + maxRows * CW;
    int *hy = reinterpret_cast<int *>(lv + maxRows * CW);
    int *hu = hy + (maxRows >> 1) * TW;
    int *hv = hu + (maxRows >> 1) * TW;

    // ================= phase 1: load + input conversion ========================================
    {
        const int ng = nc >> 2;
        const int total = nr * ng;
        for (int g = tid; g < total; g += 256) {
            const int r = g / ng, cg = g - r * ng;
            const int srow = min(r0 + r, a.srcH - 1);
            const int col = c0 + 4 * cg;
            Px4 p;
            if (SRCKIND == 0) load_rgb4(a, srow, col, p);
            else              load_yuv4(a, srow, col, p);
            int y[4];
#pragma unroll
            for (int i = 0; i < 4; i++) y[i] = rgb_to_y14(a.r2y, p.r[i], p.g[i], p.b[i]);
            *reinterpret_cast<uint2 *>(ly + r * strideCols + 4 * cg) =
                make_uint2(pack16(y[0], y[1]), pack16(y[2], y[3]));
            if (a.chrHalf) {
                int u[2], v[2];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int rs = p.r[2 * i] + p.r[2 * i + 1], gs = p.g[2 * i] + p.g[2 * i + 1],
                              bs = p.b[2 * i] + p.b[2 * i + 1];
                    u[i] = rgbsum_to_u14(a.r2y, rs, gs, bs);
                    v[i] = rgbsum_to_v14(a.r2y, rs, gs, bs);
                }
                *reinterpret_cast<unsigned *>(lu + r * CW + 2 * cg) = pack16(u[0], u[1]);
                *reinterpret_cast<unsigned *>(lv + r * CW + 2 * cg) = pack16(v[0], v[1]);
            } else {
                int u[4], v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    u[i] = rgb_to_u14(a.r2y, p.r[i], p.g[i], p.b[i]);
                    v[i] = rgb_to_v14(a.r2y, p.r[i], p.g[i], p.b[i]);
                }
                *reinterpret_cast<uint2 *>(lu + r * CW + 4 * cg) = make_uint2(pack16(u[0], u[1]), pack16(u[2], u[3]));
                *reinterpret_cast<uint2 *>(lv + r * CW + 4 * cg) = make_uint2(pack16(v[0], v[1]), pack16(v[2], v[3]));
            }
        }
    }
    __syncthreads();

    // ================= phase 2: horizontal filter ==============================================
    {
        const int xo = tid % TW;
        const int gx = min(tx0 + xo, a.dstW - 1);          // out-of-frame columns recompute the last one
        int lc[kMaxPairs], cc[kMaxPairs];
#pragma unroll
        for (int k = 0; k < kMaxPairs; k++) {
            lc[k] = k < a.hLum.pairs ? a.hLum.packed[(size_t)gx * a.hLum.pairs + k] : 0;
            cc[k] = k < a.hChr.pairs ? a.hChr.packed[(size_t)gx * a.hChr.pairs + k] : 0;
        }
        const int lpos = a.hLum.pos_even[gx] - c0;
        const int cpos = a.hChr.pos_even[gx] - (a.chrHalf ? (c0 >> 1) : c0);
        short *hy16 = reinterpret_cast<short *>(hy);
        short *hu16 = reinterpret_cast<short *>(hu);
        short *hv16 = reinterpret_cast<short *>(hv);
        for (int r = tid / TW; r < nr; r += 256 / TW) {
            const int *py = reinterpret_cast<const int *>(ly + r * strideCols + lpos);
            const int *pu = reinterpret_cast<const int *>(lu + r * CW + cpos);
            const int *pv = reinterpret_cast<const int *>(lv + r * CW + cpos);
            int sy = 0, su = 0, sv = 0;
#pragma unroll
            for (int k = 0; k < kMaxPairs; k++) {
                if (k < a.hLum.pairs) sy = dot2(py[k], lc[k], sy);
                if (k < a.hChr.pairs) {
                    su = dot2(pu[k], cc[k], su);
                    sv = dot2(pv[k], cc[k], sv);
                }
            }
            const int o = (((r >> 1) * TW + xo) << 1) + (r & 1);
            hy16[o] = (short)min(sy >> 13, 32767);
            hu16[o] = (short)min(su >> 13, 32767);
            hv16[o] = (short)min(sv >> 13, 32767);
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
            for (int k = 0; k < a.vLum.pairs; k++) {
                const int cf = a.vLum.packed[(size_t)yo * a.vLum.pairs + k];
                const int o = (vp + k) * TW + 4 * q;
                const int4 vy = *reinterpret_cast<const int4 *>(hy + o);
                const int4 vu = *reinterpret_cast<const int4 *>(hu + o);
                const int4 vv = *reinterpret_cast<const int4 *>(hv + o);
                Y[0] = dot2(vy.x, cf, Y[0]); Y[1] = dot2(vy.y, cf, Y[1]); Y[2] = dot2(vy.z, cf, Y[2]); Y[3] = dot2(vy.w, cf, Y[3]);
                U[0] = dot2(vu.x, cf, U[0]); U[1] = dot2(vu.y, cf, U[1]); U[2] = dot2(vu.z, cf, U[2]); U[3] = dot2(vu.w, cf, U[3]);
                V[0] = dot2(vv.x, cf, V[0]); V[1] = dot2(vv.y, cf, V[1]); V[2] = dot2(vv.z, cf, V[2]); V[3] = dot2(vv.w, cf, V[3]);
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
static int lds_bytes_for(int TW, int rows, int cols, int chrHalf)
{
    const int cw = chrHalf ? cols / 2 : cols;
    return rows * cols * 2 + 2 * rows * cw * 2 + 3 * (rows / 2) * TW * 4;
}

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

int scale_pick_tiling(const ScalePlan &p, ScaleTiling &t)
{
    if (p.hLum.pairs > kMaxPairs || p.hChr.pairs > kMaxPairs) return GMAT_ERR(ENOSYS);
    if (p.chrDstW != p.dstW || p.chrDstH != p.dstH) return GMAT_ERR(ENOSYS);   // full-chroma output only
    const int half = p.chrSrcHSub;
    const int forceTW = env_int("GMAT_SCALE_TW", 0), forceTH = env_int("GMAT_SCALE_TH", 0);
    const int ldsCap = env_int("GMAT_SCALE_LDS_CAP", 40 * 1024);
    const int tws[] = {64, 32};
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
            const int bytes = lds_bytes_for(TW, maxRows, maxCols, half);
            if (bytes > ldsCap && !(forceTH && bytes <= 64 * 1024)) continue;
            t.TW = TW; t.TH = TH; t.ntx = ntx; t.nty = nty;
            t.maxRows = maxRows; t.maxCols = maxCols; t.ldsBytes = bytes;
            t.xcdRemap = env_int("GMAT_SCALE_XCD", 1);
            t.colStart = cs; t.colCount = cn; t.rowStart = rs; t.rowCount = rn;
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
#define GMAT_LAUNCH_SCALE(TW_, KIND_)                                                              \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_rgb_kernel<TW_, KIND_>), grid, block, lds, stream, a, \
                       t.maxCols, t.maxRows)
    if (t.TW == 64) { if (a.srcKind) GMAT_LAUNCH_SCALE(64, 1); else GMAT_LAUNCH_SCALE(64, 0); }
    else if (t.TW == 32) { if (a.srcKind) GMAT_LAUNCH_SCALE(32, 1); else GMAT_LAUNCH_SCALE(32, 0); }
    else return GMAT_ERR(EINVAL);
#undef GMAT_LAUNCH_SCALE
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
