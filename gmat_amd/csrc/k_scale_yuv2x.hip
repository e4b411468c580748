// k_scale_yuv2x.hip — 2:1 horizontal specialisation of the single-context YUV scaler (k_scale_yuv.hip)
// for gfx950.  Same arithmetic, same results (bit-exact with one libswscale context); what changes is
// the data movement, which the regular 2:1 geometry makes possible:
//   * the host re-expresses every horizontal filter row on the regular window [2x + w0, 2x + w0 + 10)
//     (zero taps trimmed, border rows keep their folded coefficients — yuv2x_prepare), so no per-output
//     position is needed and 4 adjacent outputs share one 8-dword LDS window (4 x ds_read_b64 per row);
//   * pixels enter as 16-byte loads (16 luma samples / 8 UV pairs per lane), whole rows per wave, row
//     arithmetic on the scalar unit;
//   * the tile's coefficient rows are staged in LDS once per block (5 x ds_read_b128 per item).
// Tile 64 x 16 outputs, 256 threads, ~27 KB LDS -> 5-6 blocks per CU.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

namespace gmat {

constexpr int X2_TW = 64, X2_TH = 16, X2_P = 5;
constexpr int X2_COLSL = 160, X2_COLSC = 80;           // LDS row lengths (int16 samples)

__device__ __forceinline__ unsigned x2pk(int lo, int hi) { return ((unsigned)lo & 0xFFFF) | ((unsigned)hi << 16); }

// expands 4 bytes to two dwords of int16 pairs
__device__ __forceinline__ uint2 x2_widen(unsigned v)
{
    return make_uint2((v & 0xFF) | ((v & 0xFF00) << 8), ((v >> 16) & 0xFF) | ((v >> 24) << 16));
}

// 4 adjacent outputs x 2 rows from one regular window: w0/w1 hold 8 dwords of row 0 / row 1,
// c[j*5 + k] the k-th coefficient pair of output j.  Returns 4 dwords (row0 | row1 << 16).
__device__ __forceinline__ uint4 x2_hfilter4(const int (&w0)[8], const int (&w1)[8], const int (&c)[20])
{
    unsigned o[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int s0 = 0, s1 = 0;
#pragma unroll
        for (int k = 0; k < X2_P; k++) {
            s0 = dot2(w0[j + k], c[j * X2_P + k], s0);
            s1 = dot2(w1[j + k], c[j * X2_P + k], s1);
        }
        o[j] = x2pk(min(s0 >> 7, 32767), min(s1 >> 7, 32767));
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

__global__ __launch_bounds__(256) void scale_yuv2x_kernel(Yuv2xArgs a, int rowsL, int rowsC)
{
    HIP_DYNAMIC_SHARED(uint4, lds_base)
    int tcol, trow;
    {
        const int ntiles = a.ntx * a.nty;
        int lin = blockIdx.x;
        if (a.xcdRemap) {
            const int chunk = (ntiles + 7) >> 3;
            lin = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
        }
        if (lin >= ntiles) return;
        tcol = lin / a.nty;
        trow = lin - tcol * a.nty;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long *prof = a.prof ? a.prof + (size_t)blockIdx.x * 8 : nullptr;
#define X2_STAMP(i) do { if (prof && tid == 0) prof[i] = __builtin_readcyclecounter(); } while (0)
    X2_STAMP(0);
    if (prof && tid == 0) prof[6] = __builtin_amdgcn_s_memrealtime();     // 100 MHz reference clock
    const int tx0 = tcol * X2_TW, ty0 = trow * X2_TH, tcx0 = tx0 >> 1;
    const int r0L = a.rowStartL[trow], nrL = a.rowCountL[trow];
    const int r0C = a.rowStartC[trow], nrC = a.rowCountC[trow];
    const int wl = 2 * tx0 + a.w0L, wc = 2 * tcx0 + a.w0C;     // first window column (luma / chroma samples)
    const int c0L = wl & ~15, c0C = wc & ~7;                     // 16-byte aligned window starts
    const int eL = (wl - c0L) >> 1, eC = (wc - c0C) >> 1;       // window offset inside an LDS row, in dwords (even)

    unsigned short *ly = reinterpret_cast<unsigned short *>(lds_base);
    unsigned short *lu = ly + rowsL * X2_COLSL;
    unsigned short *lv = lu + rowsC * X2_COLSC;
    int *hy = reinterpret_cast<int *>(lv + rowsC * X2_COLSC);
    int *hu = hy + (rowsL >> 1) * X2_TW;
    int *hv = hu + (rowsC >> 1) * (X2_TW / 2);
    int *cL = hv + (rowsC >> 1) * (X2_TW / 2);                   // [64][5]
    int *cC = cL + X2_TW * X2_P;                                 // [32][5]

    // ---- prologue: vertical coefficients of this thread's output row (phase 3), issued first ----
    constexpr int QW = X2_TW / 4;
    const int q = tid % QW, yl = tid / QW;                        // 16 x 16 threads: one item each
    const int yo = ty0 + yl, yoc = min(yo, a.dstH - 1);
    int vl[X2_P], vc0, vc1;
#pragma unroll
    for (int k = 0; k < X2_P; k++) vl[k] = k < a.vLum.pairs ? a.vLum.packed[(size_t)yoc * a.vLum.pairs + k] : 0;
    vc0 = a.vChr.packed[(size_t)yoc * a.vChr.pairs];
    vc1 = a.vChr.pairs > 1 ? a.vChr.packed[(size_t)yoc * a.vChr.pairs + 1] : 0;
    const int vpL = (a.vLum.pos_even[yoc] - r0L) >> 1, vpC = (a.vChr.pos_even[yoc] - r0C) >> 1;
    const int lr = a.vLum.round[yoc], cr = a.vChr.round[yoc];

    // ================= phase 1: 16-byte loads, whole rows per wave ===============================
    {
        // tile coefficient rows -> LDS (80 + 40 lanes x 16 B)
        if (tid < 80)       reinterpret_cast<uint4 *>(cL)[tid] = reinterpret_cast<const uint4 *>(a.hLreg + (size_t)tx0 * X2_P)[tid];
        else if (tid < 120) reinterpret_cast<uint4 *>(cC)[tid - 80] = reinterpret_cast<const uint4 *>(a.hCreg + (size_t)tcx0 * X2_P)[tid - 80];

        const int rs = lane / 10, g = lane - rs * 10;            // 6 rows x 10 groups per wave (lanes 60..63 idle)
        const bool act = lane < 60;
        // luma: up to two passes in flight
        {
            const int col = min(max(c0L + 16 * g, 0), a.srcW - 16);
            const uint8_t *base = a.y + col;
            for (int rb = 0; rb < nrL; rb += 48) {
                const int ra = rb + wave * 6 + rs, rb2 = ra + 24;
                uint4 va = make_uint4(0, 0, 0, 0), vb = va;
                if (act && ra < nrL)  va = *reinterpret_cast<const uint4 *>(base + (size_t)min(max(r0L + ra, 0), a.srcH - 1) * a.ys);
                if (act && rb2 < nrL) vb = *reinterpret_cast<const uint4 *>(base + (size_t)min(max(r0L + rb2, 0), a.srcH - 1) * a.ys);
                if (act && ra < nrL) {
                    uint4 *d = reinterpret_cast<uint4 *>(ly + ra * X2_COLSL + 16 * g);
                    const uint2 p0 = x2_widen(va.x), p1 = x2_widen(va.y), p2 = x2_widen(va.z), p3 = x2_widen(va.w);
                    d[0] = make_uint4(p0.x, p0.y, p1.x, p1.y);
                    d[1] = make_uint4(p2.x, p2.y, p3.x, p3.y);
                }
                if (act && rb2 < nrL) {
                    uint4 *d = reinterpret_cast<uint4 *>(ly + rb2 * X2_COLSL + 16 * g);
                    const uint2 p0 = x2_widen(vb.x), p1 = x2_widen(vb.y), p2 = x2_widen(vb.z), p3 = x2_widen(vb.w);
                    d[0] = make_uint4(p0.x, p0.y, p1.x, p1.y);
                    d[1] = make_uint4(p2.x, p2.y, p3.x, p3.y);
                }
            }
        }
        // chroma: 8 samples of each plane per lane
        {
            const int cc = min(max(c0C + 8 * g, 0), a.chrSrcW - 8);
            for (int rb = 0; rb < nrC; rb += 24) {
                const int r = rb + wave * 6 + rs;
                if (act && r < nrC) {
                    const size_t crow = (size_t)min(max(r0C + r, 0), a.chrSrcH - 1);
                    unsigned u01, u23, u45, u67, v01, v23, v45, v67;
                    if (a.nv12) {
                        const uint4 t = *reinterpret_cast<const uint4 *>(a.u + crow * a.us + 2 * cc);   // U0 V0 U1 V1 ...
                        u01 = (t.x & 0xFF) | (t.x & 0xFF0000);  v01 = ((t.x >> 8) & 0xFF) | ((t.x >> 8) & 0xFF0000);
                        u23 = (t.y & 0xFF) | (t.y & 0xFF0000);  v23 = ((t.y >> 8) & 0xFF) | ((t.y >> 8) & 0xFF0000);
                        u45 = (t.z & 0xFF) | (t.z & 0xFF0000);  v45 = ((t.z >> 8) & 0xFF) | ((t.z >> 8) & 0xFF0000);
                        u67 = (t.w & 0xFF) | (t.w & 0xFF0000);  v67 = ((t.w >> 8) & 0xFF) | ((t.w >> 8) & 0xFF0000);
                    } else {
                        const uint2 tu = *reinterpret_cast<const uint2 *>(a.u + crow * a.us + cc);
                        const uint2 tv = *reinterpret_cast<const uint2 *>(a.v + crow * a.vs + cc);
                        const uint2 a0 = x2_widen(tu.x), a1 = x2_widen(tu.y), b0 = x2_widen(tv.x), b1 = x2_widen(tv.y);
                        u01 = a0.x; u23 = a0.y; u45 = a1.x; u67 = a1.y;
                        v01 = b0.x; v23 = b0.y; v45 = b1.x; v67 = b1.y;
                    }
                    *reinterpret_cast<uint4 *>(lu + r * X2_COLSC + 8 * g) = make_uint4(u01, u23, u45, u67);
                    *reinterpret_cast<uint4 *>(lv + r * X2_COLSC + 8 * g) = make_uint4(v01, v23, v45, v67);
                }
            }
        }
    }
    X2_STAMP(1);
    __syncthreads();
    X2_STAMP(2);

    // ================= phase 2: horizontal filters, 4 outputs x 2 rows per item ===================
    {
        const int nL = (nrL >> 1) * 16, nC = (nrC >> 1) * 8;      // luma items, chroma items per plane
        const int total = nL + 2 * nC;
        for (int it = tid; it < total; it += 256) {
            const unsigned short *srcp;
            const int *cf;
            int *dstp;
            int colsS, e, g, rp;
            if (it < nL) {
                rp = it >> 4; g = it & 15;
                srcp = ly; colsS = X2_COLSL; e = eL; cf = cL; dstp = hy + rp * X2_TW + 4 * g;
            } else {
                const int j = it - nL, pl = j >= nC, jj = pl ? j - nC : j;
                rp = jj >> 3; g = jj & 7;
                srcp = pl ? lv : lu; colsS = X2_COLSC; e = eC; cf = cC; dstp = (pl ? hv : hu) + rp * (X2_TW / 2) + 4 * g;
            }
            const int *r0p = reinterpret_cast<const int *>(srcp + (2 * rp) * colsS) + 4 * g + e;
            const int *r1p = reinterpret_cast<const int *>(srcp + (2 * rp + 1) * colsS) + 4 * g + e;
            int w0[8], w1[8], c[20];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint2 t0 = reinterpret_cast<const uint2 *>(r0p)[i], t1 = reinterpret_cast<const uint2 *>(r1p)[i];
                w0[2 * i] = (int)t0.x; w0[2 * i + 1] = (int)t0.y; w1[2 * i] = (int)t1.x; w1[2 * i + 1] = (int)t1.y;
            }
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const int4 t = reinterpret_cast<const int4 *>(cf + 4 * g * X2_P)[i];
                c[4 * i] = t.x; c[4 * i + 1] = t.y; c[4 * i + 2] = t.z; c[4 * i + 3] = t.w;
            }
            *reinterpret_cast<uint4 *>(dstp) = x2_hfilter4(w0, w1, c);
        }
    }
    X2_STAMP(3);
    __syncthreads();
    X2_STAMP(4);

    // ================= phase 3: vertical filters + colour stage + store ==========================
    {
        const int xo = tx0 + 4 * q;
        if (yo < a.dstH && xo < a.dstW) {
            int Y[4] = {lr, lr, lr, lr}, U[2] = {cr, cr}, V[2] = {cr, cr};
#pragma unroll
            for (int k = 0; k < X2_P; k++) {
                if (k < a.vLum.pairs) {
                    const int4 v = *reinterpret_cast<const int4 *>(hy + (vpL + k) * X2_TW + 4 * q);
                    Y[0] = dot2(v.x, vl[k], Y[0]); Y[1] = dot2(v.y, vl[k], Y[1]);
                    Y[2] = dot2(v.z, vl[k], Y[2]); Y[3] = dot2(v.w, vl[k], Y[3]);
                }
            }
            {
                const uint2 u = *reinterpret_cast<const uint2 *>(hu + vpC * (X2_TW / 2) + 2 * q);
                const uint2 v = *reinterpret_cast<const uint2 *>(hv + vpC * (X2_TW / 2) + 2 * q);
                U[0] = dot2((int)u.x, vc0, U[0]); U[1] = dot2((int)u.y, vc0, U[1]);
                V[0] = dot2((int)v.x, vc0, V[0]); V[1] = dot2((int)v.y, vc0, V[1]);
            }
            if (a.vChr.pairs > 1) {
                const uint2 u = *reinterpret_cast<const uint2 *>(hu + (vpC + 1) * (X2_TW / 2) + 2 * q);
                const uint2 v = *reinterpret_cast<const uint2 *>(hv + (vpC + 1) * (X2_TW / 2) + 2 * q);
                U[0] = dot2((int)u.x, vc1, U[0]); U[1] = dot2((int)u.y, vc1, U[1]);
                V[0] = dot2((int)v.x, vc1, V[0]); V[1] = dot2((int)v.y, vc1, V[1]);
            }
            const ChromaTerms t0 = chroma_terms(a.y2r, clip_u8(U[0] >> 19), clip_u8(V[0] >> 19));
            const ChromaTerms t1 = chroma_terms(a.y2r, clip_u8(U[1] >> 19), clip_u8(V[1] >> 19));
            unsigned px[4];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int ya = m24(Y[i] >> 19, a.y2r.cy), yb = m24(Y[i + 2] >> 19, a.y2r.cy);
                px[i]     = (unsigned)luma_chan(t0.r, ya) | ((unsigned)luma_chan(t0.g, ya) << 8) | ((unsigned)luma_chan(t0.b, ya) << 16);
                px[i + 2] = (unsigned)luma_chan(t1.r, yb) | ((unsigned)luma_chan(t1.g, yb) << 8) | ((unsigned)luma_chan(t1.b, yb) << 16);
            }
            const int bpp = (a.dstFormat == GMAT_PIX_FMT_RGBA || a.dstFormat == GMAT_PIX_FMT_BGRA) ? 4 : 3;
            if (a.dstFormat == GMAT_PIX_FMT_BGR24 || a.dstFormat == GMAT_PIX_FMT_BGRA) {
#pragma unroll
                for (int i = 0; i < 4; i++) px[i] = ((px[i] & 0xFF) << 16) | (px[i] & 0xFF00) | ((px[i] >> 16) & 0xFF);
            }
            uint8_t *d = a.dst + (size_t)yo * a.ds + (size_t)xo * bpp;
            const int nx = min(4, a.dstW - xo);
            if (a.dstAligned && nx == 4) {
                if (bpp == 4) {
                    *reinterpret_cast<uint4 *>(d) = make_uint4(px[0] | 0xFF000000u, px[1] | 0xFF000000u, px[2] | 0xFF000000u, px[3] | 0xFF000000u);
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
    X2_STAMP(5);
    if (prof && tid == 0) prof[7] = __builtin_amdgcn_s_memrealtime();
#undef X2_STAMP
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// Re-express a filter bank on the regular window [2x + w0, 2x + w0 + 2*X2_P).  Returns false when some
// row's non-zero taps do not fit.  `padded` = number of rows to emit (rows past count repeat the last).
static bool regularise(const FilterBank &fb, int padded, int &w0, std::vector<int32_t> &out)
{
    int lo = INT32_MAX;
    std::vector<int> first(fb.count), last(fb.count);
    for (int x = 0; x < fb.count; x++) {
        int f = -1, l = -1;
        for (int j = 0; j < fb.taps; j++)
            if (fb.coef[(size_t)x * fb.taps + j]) { if (f < 0) f = j; l = j; }
        if (f < 0) { f = l = 0; }
        first[x] = fb.pos[x] + f; last[x] = fb.pos[x] + l;
        lo = std::min(lo, first[x] - 2 * x);
    }
    w0 = lo & ~3;                                    // floor to a multiple of 4 (also for negatives)
    for (int x = 0; x < fb.count; x++)
        if (last[x] - 2 * x - w0 >= 2 * X2_P) return false;
    out.assign((size_t)padded * X2_P, 0);
    for (int xx = 0; xx < padded; xx++) {
        const int x = std::min(xx, fb.count - 1);
        int16_t win[2 * X2_P] = {0};
        for (int j = 0; j < fb.taps; j++) {
            const int16_t cv = fb.coef[(size_t)x * fb.taps + j];
            if (!cv) continue;
            win[fb.pos[x] + j - 2 * x - w0] = cv;
        }
        for (int k = 0; k < X2_P; k++)
            out[(size_t)xx * X2_P + k] = (int32_t)((uint32_t)(uint16_t)win[2 * k] | ((uint32_t)(uint16_t)win[2 * k + 1] << 16));
    }
    return true;
}

int yuv2x_prepare(const ScalePlan &p, const YuvScaleTiling &g, Yuv2xTables &t)
{
    t.ok = 0;
    const char *off = getenv("GMAT_SCALE_NO_2X");
    if (off && atoi(off)) return 0;
    if (g.fullChroma || g.TW != X2_TW || g.TH != X2_TH) return 0;
    if (p.srcW % 16 || p.chrSrcW % 8 || p.srcW < 16) return 0;
    if (p.vLum.pairs > X2_P || g.vChrEff.pairs > 2) return 0;
    t.ntx = g.ntx; t.nty = g.nty;
    if (!regularise(p.hLum, t.ntx * X2_TW, t.w0L, t.hLreg)) return 0;
    if (!regularise(p.hChr, t.ntx * (X2_TW / 2), t.w0C, t.hCreg)) return 0;
    // every tile's regular window must fit the fixed LDS row lengths
    for (int tc = 0; tc < t.ntx; tc++) {
        const int wl = 2 * tc * X2_TW + t.w0L, wc = 2 * tc * (X2_TW / 2) + t.w0C;
        if ((wl - (wl & ~15)) + 2 * (X2_TW - 1) + 2 * X2_P > X2_COLSL) return 0;
        if ((wc - (wc & ~7)) + 2 * (X2_TW / 2 - 1) + 2 * X2_P > X2_COLSC) return 0;
    }
    const int bytes = g.rowsL * X2_COLSL * 2 + 2 * g.rowsC * X2_COLSC * 2 + (g.rowsL / 2) * X2_TW * 4 +
                      2 * (g.rowsC / 2) * (X2_TW / 2) * 4 + (X2_TW + X2_TW / 2) * X2_P * 4;
    if (bytes > 64 * 1024) return 0;
    t.ok = bytes;
    return 0;
}

int launch_scale_yuv2x(const Yuv2xArgs &a, int rowsL, int rowsC, int ldsBytes, hipStream_t stream)
{
    const int ntiles = a.ntx * a.nty;
    if (ntiles <= 0) return 0;
    const dim3 grid(a.xcdRemap ? 8 * ((ntiles + 7) / 8) : ntiles), block(256);
    hipLaunchKernelGGL(scale_yuv2x_kernel, grid, block, (size_t)ldsBytes, stream, a, rowsL, rowsC);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
