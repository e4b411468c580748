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
#include <cstring>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

// The LDS image allows at most 6 blocks per CU = 6 waves per SIMD: tell the compiler, so it budgets registers and
// schedules for that occupancy instead of the maximum (measured: 6.50 -> 6.41 us per frame, 6.58 -> 6.40 for YUV output)
#if !defined(X2_WAVES_ATTR) && defined(__HIP__)
#define X2_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(6, 6)))
#elif !defined(X2_WAVES_ATTR)
#define X2_WAVES_ATTR
#endif

namespace gmat {

constexpr int X2_TW = 64, X2_TH = 16;
constexpr int X2_VRC = 12;                              // ints per chroma-row vertical record (YUV output)
constexpr int X2_COLSC = 80;                            // chroma LDS row length (int16 samples)
// Per-variant sizes.  The luma LDS row holds the widest regular window, 15 + 2*63 + 2P samples: 151 for P = 5, so 152
// (the last 16-sample chunk of a row is stored as its first half only), 160 for P = 8.  The per-output-row record is
//   P = 5: [0..4] luma pairs  [5..6] chroma pairs  [8] luma window row  [9] chroma window row  [10],[11] start values
//   P = 8: [0..7] luma pairs  [8..11] chroma pairs [12] ...             [13] ...               [14],[15]
// Both exist to keep the P = 5 block at 27 136 B of LDS: hipOccupancyMaxActiveBlocksPerMultiprocessor steps from 6 to
// 5 blocks per CU between 27 264 and 27 520 B on gfx950.  (Measured effect on the batched headline launch: none —
// SQ_WAVE_CYCLES shows the same 4.3 resident blocks per CU on average either way)
constexpr int x2_colsl(int P) { return P == 5 ? 152 : 160; }
constexpr int x2_vr(int P) { return P == 5 ? 12 : 16; }
constexpr int x2_vr_chroma(int P) { return P == 5 ? 5 : 8; }
constexpr int x2_vr_misc(int P) { return P == 5 ? 8 : 12; }

__device__ __forceinline__ unsigned x2pk(int lo, int hi) { return ((unsigned)lo & 0xFFFF) | ((unsigned)hi << 16); }

// expands 4 bytes to two dwords of int16 pairs: two v_perm_b32 (selector 0x0C = constant zero byte)
__device__ __forceinline__ uint2 x2_widen(unsigned v)
{
    return make_uint2(__builtin_amdgcn_perm(0u, v, 0x0C010C00u), __builtin_amdgcn_perm(0u, v, 0x0C030C02u));
}

typedef short x2_short2 __attribute__((ext_vector_type(2)));

// 4 adjacent outputs x 2 rows from one regular window: w0/w1 hold 8 dwords of row 0 / row 1,
// c[j*5 + k] the k-th coefficient pair of output j.  Returns 4 dwords (row0 | row1 << 16).
template <int P>
__device__ __forceinline__ uint4 x2_hfilter4(const int (&w0)[(4 + P) & ~1], const int (&w1)[(4 + P) & ~1], const int (&c)[4 * P])
{
    unsigned o[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int s0 = 0, s1 = 0;
#pragma unroll
        for (int k = 0; k < P; k++) {
            s0 = dot2(w0[j + k], c[j * P + k], s0);
            s1 = dot2(w1[j + k], c[j * P + k], s1);
        }
        // hScale8To15_c stores min(val >> 7, 32767) into an int16.  v_cvt_pk_i16_i32 saturates both ways and
        // packs in one instruction; the lower bound cannot trigger (255 * sum of negative taps >> 7 > -32768
        // for every kernel initFilter can emit with one == 16384).
        o[j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_i16(s0 >> 7, s1 >> 7));
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// YUVOUT = false: packed RGB out (colour stage fused).  true: NV12 / YUV420P out — the tile is 64 x 16 luma
// outputs plus the 32 x 8 chroma outputs under them, vertical chroma filter indexed by chroma row
// (yuv2planeX_8_c / yuv2nv12cX_c, output.c:400-450).
// P = coefficient pairs per output on the regular window of 2*P samples: 5 covers bicubic / bilinear (8 taps + the
// parity slot), 8 covers Lanczos-3 (12 taps; the window origin is a multiple of 4 samples, which costs up to 3).
// UNI: the block's tile(s) lie in the interior of an exact 2:1 geometry (Yuv2xUniform): no coefficient tables, no
// per-row records — the coefficients are kernel arguments (SGPR operands of the dot products), the window rows
// closed forms.  The general path stages both in LDS.
template <bool YUVOUT, int P, int TILES, bool UNI>
__device__ __forceinline__ void x2_tiles(const Yuv2xArgs &a, int rowsL, int rowsC, int tcol, int trow0, uint4 *lds_base)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long *prof = a.prof ? a.prof + (size_t)blockIdx.x * 8 : nullptr;
#define X2_STAMP(i) do { if (prof && tid == 0 && tt == 0) prof[i] = __builtin_readcyclecounter(); } while (0)
    const int tx0 = tcol * X2_TW, tcx0 = tx0 >> 1;
    const int wl = 2 * tx0 + a.w0L, wc = 2 * tcx0 + a.w0C;     // first window column (luma / chroma samples)
    const int c0L = wl & ~15, c0C = wc & ~7;                     // 16-byte aligned window starts
    const int eL = (wl - c0L) >> 1, eC = (wc - c0C) >> 1;       // window offset inside an LDS row, in dwords (even)

    unsigned short *ly = reinterpret_cast<unsigned short *>(lds_base);
    constexpr int X2_COLSL = x2_colsl(P), X2_VR = x2_vr(P), VR_M = x2_vr_misc(P);
    unsigned short *lu = ly + rowsL * X2_COLSL;
    unsigned short *lv = lu + rowsC * X2_COLSC;
    int *hy = reinterpret_cast<int *>(lv + rowsC * X2_COLSC);
    int *hu = hy + (rowsL >> 1) * X2_TW;
    int *hv = hu + (rowsC >> 1) * (X2_TW / 2);
    int *cL = hv + (rowsC >> 1) * (X2_TW / 2);                   // [64][P]
    int *cC = cL + X2_TW * P;                                    // [32][P]
    int *vr = cC + (X2_TW / 2) * P;                              // [16][X2_VR] vertical records of the tile's rows
    int *vrc = vr + X2_TH * X2_VR;                               // YUVOUT: [8][X2_VRC] records of the tile's chroma rows

    constexpr int QW = X2_TW / 4;
    const int q = tid % QW, yl = tid / QW;                        // 16 x 16 threads: one phase-3 item each

    // ---- phase 1, split in two: issue the loads of a tile into registers / commit them to LDS ----------------
    constexpr int NL = UNI ? 0 : X2_TW * P / 4, NC = UNI ? 0 : (X2_TW / 2) * P / 4, NV = UNI ? 0 : X2_TH * X2_VR / 4,
                  NVC = (YUVOUT && !UNI) ? (X2_TH / 2) * X2_VRC / 4 : 0;
    constexpr int NT = NL + NC + NV + NVC;                       // 16-byte table chunks: one per thread, a second for
    static_assert(NT <= 512, "two table chunks per thread");      // the first NT - 256 threads of the widest variant
    const int rs = (lane * 205) >> 11, g = lane - rs * 10;       // lane / 10 for lane < 64: 6 rows x 10 groups per wave
    const bool act = lane < 60;
    const int rowA = wave * 6 + rs, rowB = rowA + 24;            // nrL <= 48, nrC <= 24 (yuv2x_prepare)
    const unsigned colL = (unsigned)min(max(c0L + 16 * g, 0), a.srcW - 16);
    const unsigned colC = (unsigned)min(max(c0C + 8 * g, 0), a.chrSrcW - 8);
    // the Lanczos 4:2:0-output variant filters 8 chroma rows from 26-28 source rows: a second chroma slot (rows 24..47)
    constexpr bool C2 = YUVOUT && P == 8;
    struct TileRegs { uint4 va, vb, tc, tc2, tv0, tv1; int r0L, nrL, r0C, nrC; };
    // table chunk i of tile row `trow`: the coefficient rows (first tile of the block only) and the vertical records
    auto tab_src = [&](int i, int trow) -> const uint4 * {
        const int ty0 = trow * X2_TH;
        if (i < NL)           return reinterpret_cast<const uint4 *>(a.hLreg + (size_t)tx0 * P) + i;
        if (i < NL + NC)      return reinterpret_cast<const uint4 *>(a.hCreg + (size_t)tcx0 * P) + (i - NL);
        if (i < NL + NC + NV) return reinterpret_cast<const uint4 *>(a.vrec + (size_t)ty0 * X2_VR) + (i - NL - NC);
        return reinterpret_cast<const uint4 *>(a.vrecC + (size_t)(ty0 >> 1) * X2_VRC) + (i - NL - NC - NV);
    };
    auto tab_dst = [&](int i) -> uint4 * {
        if (i < NL)           return reinterpret_cast<uint4 *>(cL) + i;
        if (i < NL + NC)      return reinterpret_cast<uint4 *>(cC) + (i - NL);
        if (i < NL + NC + NV) return reinterpret_cast<uint4 *>(vr) + (i - NL - NC);
        return reinterpret_cast<uint4 *>(vrc) + (i - NL - NC - NV);
    };
    auto issue = [&](int trow, bool first) -> TileRegs {
        TileRegs R;
        if constexpr (UNI) {           // interior tile rows: the windows step by a constant (checked on the host)
            R.r0L = a.uni.r0L + trow * a.uni.dL; R.nrL = a.uni.nrL;
            R.r0C = a.uni.r0C + trow * a.uni.dC; R.nrC = a.uni.nrC;
        } else {
            R.r0L = uniform_load(a.rowStartL, trow); R.nrL = uniform_load(a.rowCountL, trow);
            R.r0C = uniform_load(a.rowStartC, trow); R.nrC = uniform_load(a.rowCountC, trow);
        }
        // the table loads are ISSUED first and stored to LDS only after the pixel loads have been issued too, so a
        // block pays one memory round trip, not two (a load -> store loop in this place cost 7 %)
        R.tv0 = R.tv1 = make_uint4(0u, 0u, 0u, 0u);
        if (tid < NT && (first || tid >= NL + NC)) R.tv0 = *tab_src(tid, trow);
        if (NT > 256 && tid + 256 < NT) R.tv1 = *tab_src(tid + 256, trow);
        // all three pixel loads are issued before anything is consumed; rows are clamped for the load and only the
        // store is predicated
        const unsigned offA = (unsigned)min(max(R.r0L + min(rowA, R.nrL - 1), 0), a.srcH - 1) * (unsigned)a.ys + colL;
        const unsigned offB = (unsigned)min(max(R.r0L + min(rowB, R.nrL - 1), 0), a.srcH - 1) * (unsigned)a.ys + colL;
        const unsigned crow = (unsigned)min(max(R.r0C + min(rowA, R.nrC - 1), 0), a.chrSrcH - 1);
        R.va = *reinterpret_cast<const uint4 *>(a.y + offA);
        R.vb = *reinterpret_cast<const uint4 *>(a.y + offB);
        R.tc2 = make_uint4(0u, 0u, 0u, 0u);
        if (C2) {
            const unsigned crow2 = (unsigned)min(max(R.r0C + min(rowB, R.nrC - 1), 0), a.chrSrcH - 1);
            if (a.nv12) {
                R.tc2 = *reinterpret_cast<const uint4 *>(a.u + crow2 * (unsigned)a.us + 2 * colC);
            } else {
                const uint2 tu = *reinterpret_cast<const uint2 *>(a.u + crow2 * (unsigned)a.us + colC);
                const uint2 tv = *reinterpret_cast<const uint2 *>(a.v + crow2 * (unsigned)a.vs + colC);
                R.tc2.x = __builtin_amdgcn_perm(tv.x, tu.x, 0x05010400u); R.tc2.y = __builtin_amdgcn_perm(tv.x, tu.x, 0x07030602u);
                R.tc2.z = __builtin_amdgcn_perm(tv.y, tu.y, 0x05010400u); R.tc2.w = __builtin_amdgcn_perm(tv.y, tu.y, 0x07030602u);
            }
        }
        if (a.nv12) {
            R.tc = *reinterpret_cast<const uint4 *>(a.u + crow * (unsigned)a.us + 2 * colC);       // U0 V0 U1 V1 ...
        } else {
            const uint2 tu = *reinterpret_cast<const uint2 *>(a.u + crow * (unsigned)a.us + colC);
            const uint2 tv = *reinterpret_cast<const uint2 *>(a.v + crow * (unsigned)a.vs + colC);
            // interleave to the NV12 byte order so the rest is common
            R.tc.x = __builtin_amdgcn_perm(tv.x, tu.x, 0x05010400u); R.tc.y = __builtin_amdgcn_perm(tv.x, tu.x, 0x07030602u);
            R.tc.z = __builtin_amdgcn_perm(tv.y, tu.y, 0x05010400u); R.tc.w = __builtin_amdgcn_perm(tv.y, tu.y, 0x07030602u);
        }
        return R;
    };
    auto commit = [&](bool first, const TileRegs &R) {
        if (tid < NT && (first || tid >= NL + NC)) *tab_dst(tid) = R.tv0;
        if (NT > 256 && tid + 256 < NT) *tab_dst(tid + 256) = R.tv1;
        if (act && rowA < R.nrL) {
            uint4 *d = reinterpret_cast<uint4 *>(ly + rowA * X2_COLSL + 16 * g);
            const uint2 p0 = x2_widen(R.va.x), p1 = x2_widen(R.va.y), p2 = x2_widen(R.va.z), p3 = x2_widen(R.va.w);
            d[0] = make_uint4(p0.x, p0.y, p1.x, p1.y);
            if (X2_COLSL == 160 || g < 9) d[1] = make_uint4(p2.x, p2.y, p3.x, p3.y);
        }
        if (act && rowB < R.nrL) {
            uint4 *d = reinterpret_cast<uint4 *>(ly + rowB * X2_COLSL + 16 * g);
            const uint2 p0 = x2_widen(R.vb.x), p1 = x2_widen(R.vb.y), p2 = x2_widen(R.vb.z), p3 = x2_widen(R.vb.w);
            d[0] = make_uint4(p0.x, p0.y, p1.x, p1.y);
            if (X2_COLSL == 160 || g < 9) d[1] = make_uint4(p2.x, p2.y, p3.x, p3.y);
        }
        if (C2 && act && rowB < R.nrC) {
            *reinterpret_cast<uint4 *>(lu + rowB * X2_COLSC + 8 * g) =
                make_uint4(__builtin_amdgcn_perm(0u, R.tc2.x, 0x0C020C00u), __builtin_amdgcn_perm(0u, R.tc2.y, 0x0C020C00u),
                           __builtin_amdgcn_perm(0u, R.tc2.z, 0x0C020C00u), __builtin_amdgcn_perm(0u, R.tc2.w, 0x0C020C00u));
            *reinterpret_cast<uint4 *>(lv + rowB * X2_COLSC + 8 * g) =
                make_uint4(__builtin_amdgcn_perm(0u, R.tc2.x, 0x0C030C01u), __builtin_amdgcn_perm(0u, R.tc2.y, 0x0C030C01u),
                           __builtin_amdgcn_perm(0u, R.tc2.z, 0x0C030C01u), __builtin_amdgcn_perm(0u, R.tc2.w, 0x0C030C01u));
        }
        if (act && rowA < R.nrC) {
            // U samples are bytes 0 and 2 of each dword, V samples bytes 1 and 3
            *reinterpret_cast<uint4 *>(lu + rowA * X2_COLSC + 8 * g) =
                make_uint4(__builtin_amdgcn_perm(0u, R.tc.x, 0x0C020C00u), __builtin_amdgcn_perm(0u, R.tc.y, 0x0C020C00u),
                           __builtin_amdgcn_perm(0u, R.tc.z, 0x0C020C00u), __builtin_amdgcn_perm(0u, R.tc.w, 0x0C020C00u));
            *reinterpret_cast<uint4 *>(lv + rowA * X2_COLSC + 8 * g) =
                make_uint4(__builtin_amdgcn_perm(0u, R.tc.x, 0x0C030C01u), __builtin_amdgcn_perm(0u, R.tc.y, 0x0C030C01u),
                           __builtin_amdgcn_perm(0u, R.tc.z, 0x0C030C01u), __builtin_amdgcn_perm(0u, R.tc.w, 0x0C030C01u));
        }
    };

    TileRegs cur, nxt = TileRegs();
    {
        const int tt = 0;
        X2_STAMP(0);
        if (prof && tid == 0) prof[6] = __builtin_amdgcn_s_memrealtime();     // 100 MHz reference clock
    }
    cur = issue(trow0, true);
#pragma unroll
    for (int tt = 0; tt < TILES; tt++) {
        const int trow = trow0 + tt;
        if (trow >= a.nty) break;                                  // block-uniform
        const int ty0 = trow * X2_TH, yo = ty0 + yl;
        const int r0L = cur.r0L, nrL = cur.nrL, r0C = cur.r0C, nrC = cur.nrC;
        commit(tt == 0, cur);
        X2_STAMP(1);
        __syncthreads();
        X2_STAMP(2);
        const bool more = tt + 1 < TILES && trow + 1 < a.nty;
        if (more) nxt = issue(trow + 1, false);                    // in flight during phases 2 and 3

        // ================= phase 2: horizontal filters, 4 outputs x 2 rows per item ===================
        {
            // The luma item range is padded to whole waves so that every wave iteration is entirely luma or
            // entirely chroma: the choice is a scalar branch and each side has compile-time row lengths
            // (the per-lane select cost ~25 VALU instructions per item before).
            const int nL = (nrL >> 1) * 16, nC = (nrC >> 1) * 8;      // luma items, chroma items per plane
            const int nLw = (nL + 63) & ~63;
            const int total = nLw + 2 * nC;
            auto item = [&](const unsigned short *srcp, int colsS, int e, const int *cf, const int *cfu, int *dstp, int rp, int g) {
                const uint2 *r0p = reinterpret_cast<const uint2 *>(srcp + (2 * rp) * colsS);
                const uint2 *r1p = reinterpret_cast<const uint2 *>(srcp + (2 * rp + 1) * colsS);
                const int pair0 = 2 * g + (e >> 1);                       // 8-byte pair index of the window start
                // 4 adjacent outputs share a window of 6 + 2P samples = 3 + P dwords, read as 8-byte pairs
                constexpr int NWD = (4 + P) & ~1;
                int w0[NWD], w1[NWD], c[4 * P];
    #pragma unroll
                for (int i = 0; i < NWD / 2; i++) {
                    const uint2 t0 = r0p[pair0 + i], t1 = r1p[pair0 + i];
                    w0[2 * i] = (int)t0.x; w0[2 * i + 1] = (int)t0.y; w1[2 * i] = (int)t1.x; w1[2 * i + 1] = (int)t1.y;
                }
    #pragma unroll
                for (int i = 0; i < P; i++) {
                    if constexpr (UNI) {
                        c[i] = c[P + i] = c[2 * P + i] = c[3 * P + i] = cfu[i];      // the same row for the 4 outputs (SGPRs)
                    } else {
                        const int4 t = reinterpret_cast<const int4 *>(cf + 4 * g * P)[i];
                        c[4 * i] = t.x; c[4 * i + 1] = t.y; c[4 * i + 2] = t.z; c[4 * i + 3] = t.w;
                    }
                }
                *reinterpret_cast<uint4 *>(dstp) = x2_hfilter4<P>(w0, w1, c);
            };
            for (int it = tid; it < total; it += 256) {
                const int wbase = __builtin_amdgcn_readfirstlane(it);     // tid of the wave's first lane is a multiple of 64
                if (wbase < nLw) {
                    if (it < nL) {
                        const int rp = it >> 4, g = it & 15;
                        item(ly, X2_COLSL, eL, cL, a.uni.hL, hy + rp * X2_TW + 4 * g, rp, g);
                    }
                } else {
                    const int j = it - nLw, pl = j >= nC, jj = pl ? j - nC : j;
                    const int rp = jj >> 3, g = jj & 7;
                    item(lu + pl * (rowsC * X2_COLSC), X2_COLSC, eC, cC, a.uni.hC, hu + pl * ((rowsC >> 1) * (X2_TW / 2)) + rp * (X2_TW / 2) + 4 * g, rp, g);
                }
            }
        }
        X2_STAMP(3);
        __syncthreads();
        X2_STAMP(4);

        // ================= phase 3: vertical filters + colour stage + store ==========================
        if constexpr (YUVOUT) {
            const int xo = tx0 + 4 * q;
            {   // luma: 4 outputs of row yo
                int vl8[8], vpL, lr;
                if constexpr (UNI) {
    #pragma unroll
                    for (int k = 0; k < 8; k++) vl8[k] = a.uni.vL[k];
                    vpL = (2 * yo + a.uni.aL - r0L) >> 1; lr = a.uni.lr;
                } else {
                    const int4 ra = reinterpret_cast<const int4 *>(vr + yl * X2_VR)[0], rb = reinterpret_cast<const int4 *>(vr + yl * X2_VR)[1],
                               rd = reinterpret_cast<const int4 *>(vr + yl * X2_VR)[VR_M / 4];
                    vl8[0] = ra.x; vl8[1] = ra.y; vl8[2] = ra.z; vl8[3] = ra.w; vl8[4] = rb.x; vl8[5] = rb.y; vl8[6] = rb.z; vl8[7] = rb.w;
                    vpL = (rd.x - r0L) >> 1; lr = rd.z;
                }
                int Y[4] = {lr, lr, lr, lr};
    #pragma unroll
                for (int k = 0; k < P; k++) {
                    if (k < a.vLpairs) {
                        const int4 v = *reinterpret_cast<const int4 *>(hy + (vpL + k) * X2_TW + 4 * q);
                        Y[0] = dot2(v.x, vl8[k], Y[0]); Y[1] = dot2(v.y, vl8[k], Y[1]);
                        Y[2] = dot2(v.z, vl8[k], Y[2]); Y[3] = dot2(v.w, vl8[k], Y[3]);
                    }
                }
                if (yo < a.dstH && xo < a.dstW) {
                    // clip_u8(v >> 19) = byte 2 of clamp(v >> 3, 0, 0xFFFFFF)
                    const unsigned y0 = (unsigned)min(max(Y[0] >> 3, 0), 0xFFFFFF), y1 = (unsigned)min(max(Y[1] >> 3, 0), 0xFFFFFF);
                    const unsigned y2 = (unsigned)min(max(Y[2] >> 3, 0), 0xFFFFFF), y3 = (unsigned)min(max(Y[3] >> 3, 0), 0xFFFFFF);
                    const unsigned o = __builtin_amdgcn_perm(y1, y0, 0x0C0C0602u) | (__builtin_amdgcn_perm(y3, y2, 0x0C0C0602u) << 16);
                    uint8_t *d = a.dst + (size_t)yo * a.ds + xo;
                    const int nx = min(4, a.dstW - xo);
                    if (a.dstAligned && nx == 4) *reinterpret_cast<unsigned *>(d) = o;
                    else for (int i = 0; i < nx; i++) d[i] = (uint8_t)(o >> (8 * i));
                }
            }
            {   // chroma: thread (q, yl) computes plane (yl & 1) of chroma row yl >> 1, columns 2q and 2q + 1;
                // the U and V halves of an NV12 dword meet through one cross-lane exchange (lanes l and l ^ 16)
                const int pl = yl & 1, cyl = yl >> 1;
                const int cy = (ty0 >> 1) + cyl, cx = tcx0 + 2 * q;
                int vcp[8], vp, crnd;
                if constexpr (UNI) {
    #pragma unroll
                    for (int k = 0; k < 8; k++) vcp[k] = a.uni.vCy[k];
                    vp = (2 * cy + a.uni.aCy - r0C) >> 1; crnd = a.uni.cr;
                } else {
                    const int4 ca = reinterpret_cast<const int4 *>(vrc + cyl * X2_VRC)[0], cb = reinterpret_cast<const int4 *>(vrc + cyl * X2_VRC)[1],
                               cc3 = reinterpret_cast<const int4 *>(vrc + cyl * X2_VRC)[2];
                    vcp[0] = ca.x; vcp[1] = ca.y; vcp[2] = ca.z; vcp[3] = ca.w; vcp[4] = cb.x; vcp[5] = cb.y; vcp[6] = cb.z; vcp[7] = cb.w;
                    vp = (cc3.x - r0C) >> 1; crnd = cc3.y;
                }
                const int *hp = pl ? hv : hu;
                int C0 = crnd, C1 = crnd;
    #pragma unroll
                for (int k = 0; k < P; k++) {
                    if (k < a.vCpairs) {
                        const uint2 t = *reinterpret_cast<const uint2 *>(hp + (vp + k) * (X2_TW / 2) + 2 * q);
                        C0 = dot2((int)t.x, vcp[k], C0); C1 = dot2((int)t.y, vcp[k], C1);
                    }
                }
                const unsigned c0 = (unsigned)min(max(C0 >> 3, 0), 0xFFFFFF), c1 = (unsigned)min(max(C1 >> 3, 0), 0xFFFFFF);
                const unsigned mine = __builtin_amdgcn_perm(c1, c0, 0x0C0C0602u);        // sample 0 | sample 1 << 8
                const bool inside = cy < a.chrDstH && cx < a.chrDstW;
                const int nx = min(2, a.chrDstW - cx);
                if (a.dstNv12) {
                    const unsigned other = (unsigned)__shfl_xor((int)mine, 16);           // V pair for the U lanes
                    if (inside && pl == 0) {
                        // U0 V0 U1 V1
                        const unsigned o = __builtin_amdgcn_perm(other, mine, 0x05010400u);
                        uint8_t *d = a.dstU + (size_t)cy * a.dsU + 2 * cx;
                        if (a.dstAligned && nx == 2) *reinterpret_cast<unsigned *>(d) = o;
                        else for (int i = 0; i < 2 * nx; i++) d[i] = (uint8_t)(o >> (8 * i));
                    }
                } else if (inside) {
                    uint8_t *d = (pl ? a.dstV + (size_t)cy * a.dsV : a.dstU + (size_t)cy * a.dsU) + cx;
                    if (a.dstAligned && nx == 2) *reinterpret_cast<unsigned short *>(d) = (unsigned short)mine;
                    else for (int i = 0; i < nx; i++) d[i] = (uint8_t)(mine >> (8 * i));
                }
            }
        } else {
            const int xo = tx0 + 4 * q;
            if (yo < a.dstH && xo < a.dstW) {
                // this row's record: 5 luma pairs, 2 chroma pairs, window positions, accumulator start values
                int vl[8], vcc[4], vpL, vpC, lr, cr;
                if constexpr (UNI) {
                    // interior rows: one vertical luma row, two chroma rows (by row parity), window rows in closed form
    #pragma unroll
                    for (int k = 0; k < 8; k++) vl[k] = a.uni.vL[k];
                    const bool odd = (yo & 1) != 0;
    #pragma unroll
                    for (int k = 0; k < 4; k++) vcc[k] = odd ? a.uni.vC[1][k] : a.uni.vC[0][k];
                    vpL = (2 * yo + a.uni.aL - r0L) >> 1; vpC = (((yo + a.uni.aC) & ~1) - r0C) >> 1;
                    lr = a.uni.lr; cr = a.uni.cr;
                } else {
                    const int4 ra = reinterpret_cast<const int4 *>(vr + yl * X2_VR)[0], rb = reinterpret_cast<const int4 *>(vr + yl * X2_VR)[1],
                               rd = reinterpret_cast<const int4 *>(vr + yl * X2_VR)[VR_M / 4];
                    const int4 rc = P == 5 ? make_int4(rb.y, rb.z, 0, 0) : reinterpret_cast<const int4 *>(vr + yl * X2_VR)[2];
                    vl[0] = ra.x; vl[1] = ra.y; vl[2] = ra.z; vl[3] = ra.w; vl[4] = rb.x; vl[5] = rb.y; vl[6] = rb.z; vl[7] = rb.w;
                    vcc[0] = rc.x; vcc[1] = rc.y; vcc[2] = rc.z; vcc[3] = rc.w;
                    vpL = (rd.x - r0L) >> 1; vpC = (rd.y - r0C) >> 1; lr = rd.z; cr = rd.w;
                }
                int Y[4] = {lr, lr, lr, lr}, U[2] = {cr, cr}, V[2] = {cr, cr};
    #pragma unroll
                for (int k = 0; k < P; k++) {
                    if (k < a.vLpairs) {
                        const int4 v = *reinterpret_cast<const int4 *>(hy + (vpL + k) * X2_TW + 4 * q);
                        Y[0] = dot2(v.x, vl[k], Y[0]); Y[1] = dot2(v.y, vl[k], Y[1]);
                        Y[2] = dot2(v.z, vl[k], Y[2]); Y[3] = dot2(v.w, vl[k], Y[3]);
                    }
                }
                constexpr int VCP = P == 5 ? 2 : 4;           // vertical chroma pairs the variant provides for
    #pragma unroll
                for (int k = 0; k < VCP; k++) {
                    if (k == 0 || k < a.vCpairs) {
                        const uint2 u = *reinterpret_cast<const uint2 *>(hu + (vpC + k) * (X2_TW / 2) + 2 * q);
                        const uint2 v = *reinterpret_cast<const uint2 *>(hv + (vpC + k) * (X2_TW / 2) + 2 * q);
                        U[0] = dot2((int)u.x, vcc[k], U[0]); U[1] = dot2((int)u.y, vcc[k], U[1]);
                        V[0] = dot2((int)v.x, vcc[k], V[0]); V[1] = dot2((int)v.y, vcc[k], V[1]);
                    }
                }
                ChromaTerms t0 = chroma_terms(a.y2r, clip_u8(U[0] >> 19), clip_u8(V[0] >> 19));
                ChromaTerms t1 = chroma_terms(a.y2r, clip_u8(U[1] >> 19), clip_u8(V[1] >> 19));
                const int bpp = (a.dstFormat == GMAT_PIX_FMT_RGBA || a.dstFormat == GMAT_PIX_FMT_BGRA) ? 4 : 3;
                if (a.dstFormat == GMAT_PIX_FMT_BGR24 || a.dstFormat == GMAT_PIX_FMT_BGRA) {
                    int t = t0.r; t0.r = t0.b; t0.b = t;
                    t = t1.r; t1.r = t1.b; t1.b = t;
                }
                // channel value = byte 2 of clamp(term + Y*cy, 0, 0xFFFFFF); bytes are gathered with v_perm_b32
                unsigned c0[4], c1[4], c2[4];                 // first / second / third channel of the 4 pixels
    #pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int ya = m24(Y[i] >> 19, a.y2r.cy), yb = m24(Y[i + 2] >> 19, a.y2r.cy);
                    c0[i] = (unsigned)min(max(t0.r + ya, 0), 0xFFFFFF); c1[i] = (unsigned)min(max(t0.g + ya, 0), 0xFFFFFF);
                    c2[i] = (unsigned)min(max(t0.b + ya, 0), 0xFFFFFF);
                    c0[i + 2] = (unsigned)min(max(t1.r + yb, 0), 0xFFFFFF); c1[i + 2] = (unsigned)min(max(t1.g + yb, 0), 0xFFFFFF);
                    c2[i + 2] = (unsigned)min(max(t1.b + yb, 0), 0xFFFFFF);
                }
                // perm(hi, lo, sel): byte2(lo) -> byte 0, byte2(hi) -> byte 1, upper half zero
    #define X2_B2PAIR(lo, hi) __builtin_amdgcn_perm((hi), (lo), 0x0C0C0602u)
                uint8_t *d = a.dst + (size_t)yo * a.ds + (size_t)xo * bpp;
                const int nx = min(4, a.dstW - xo);
                if (a.dstAligned && nx == 4) {
                    if (bpp == 4) {
                        uint4 o4;
                        o4.x = X2_B2PAIR(c0[0], c1[0]) | (X2_B2PAIR(c2[0], 0u) << 16) | 0xFF000000u;
                        o4.y = X2_B2PAIR(c0[1], c1[1]) | (X2_B2PAIR(c2[1], 0u) << 16) | 0xFF000000u;
                        o4.z = X2_B2PAIR(c0[2], c1[2]) | (X2_B2PAIR(c2[2], 0u) << 16) | 0xFF000000u;
                        o4.w = X2_B2PAIR(c0[3], c1[3]) | (X2_B2PAIR(c2[3], 0u) << 16) | 0xFF000000u;
                        *reinterpret_cast<uint4 *>(d) = o4;
                    } else {
                        uint3 o3;           // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
                        o3.x = X2_B2PAIR(c0[0], c1[0]) | (X2_B2PAIR(c2[0], c0[1]) << 16);
                        o3.y = X2_B2PAIR(c1[1], c2[1]) | (X2_B2PAIR(c0[2], c1[2]) << 16);
                        o3.z = X2_B2PAIR(c2[2], c0[3]) | (X2_B2PAIR(c1[3], c2[3]) << 16);
#if defined(__HIP_DEVICE_COMPILE__)
                        {   // one 12-byte store: left to the compiler, the store is tail-merged with the RGBA path's and
                            // split into 8 + 4 bytes (two store instructions per wave; SQ_INSTS_VMEM_WR doubled)
                            typedef unsigned x2_u32x3 __attribute__((ext_vector_type(3)));
                            const x2_u32x3 pk = {o3.x, o3.y, o3.z};
                            asm volatile("global_store_dwordx3 %0, %1, off" :: "v"(d), "v"(pk) : "memory");
                        }
#else
                        *reinterpret_cast<uint3 *>(d) = o3;
#endif
                    }
                } else {
                    for (int i = 0; i < nx; i++) {
                        d[i * bpp + 0] = (uint8_t)(c0[i] >> 16);
                        d[i * bpp + 1] = (uint8_t)(c1[i] >> 16);
                        d[i * bpp + 2] = (uint8_t)(c2[i] >> 16);
                        if (bpp == 4) d[i * bpp + 3] = 255;
                    }
                }
    #undef X2_B2PAIR
            }
        }
        X2_STAMP(5);
        if (prof && tid == 0 && tt == 0) prof[7] = __builtin_amdgcn_s_memrealtime();
        if (more) {
            __syncthreads();                                       // phase 3 has read vr before the next commit overwrites it
            cur = nxt;
        }
    }
#undef X2_STAMP
}

template <bool YUVOUT, int P, int TILES>
__global__ __launch_bounds__(256) X2_WAVES_ATTR void scale_yuv2x_kernel(Yuv2xArgs a, Yuv2xFrames fr, int rowsL, int rowsC)
{
    // grid.y = frame of the batch: the plane pointers come from the kernel-argument segment (scalar loads)
    {
        const int f = blockIdx.y;
        a.y = fr.y[f]; a.u = fr.u[f]; a.v = fr.v[f];
        a.dst = fr.dst[f]; a.dstU = fr.dstU[f]; a.dstV = fr.dstV[f];
    }
    // TILES vertically adjacent tiles per block, software-pipelined: the pixel (and record) loads of tile t+1 are
    // issued right after tile t's rows have been committed to LDS and stay in flight during its phases 2 and 3,
    // so only the first tile of a block waits for HBM.
    HIP_DYNAMIC_SHARED(uint4, lds_base)
    int tcol, trow0;
    {
        const int ntyB = (a.nty + TILES - 1) / TILES, ntiles = a.ntx * ntyB;
        int lin = blockIdx.x;
        if (a.xcdRemap) {
            const int chunk = (ntiles + 7) >> 3;
            lin = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
        }
        if (lin >= ntiles) return;
        // the division runs on the VALU; readfirstlane tells the compiler the result is wave-uniform, so the
        // per-tile table look-ups below become scalar loads and the tile address arithmetic scalar code
        tcol = __builtin_amdgcn_readfirstlane(lin / ntyB);
        trow0 = (lin - tcol * ntyB) * TILES;
    }
#ifndef X2U_FORCE_GENERAL
    if constexpr (TILES == 1) {
        if (tcol >= a.uni.tcLo && tcol <= a.uni.tcHi && trow0 >= a.uni.trLo && trow0 <= a.uni.trHi && !a.prof) {   // block-uniform
            x2_tiles<YUVOUT, P, TILES, true>(a, rowsL, rowsC, tcol, trow0, lds_base);
            return;
        }
    }
#endif
    x2_tiles<YUVOUT, P, TILES, false>(a, rowsL, rowsC, tcol, trow0, lds_base);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// Re-express a filter bank on the regular window [2x + w0, 2x + w0 + 2*P).  Returns false when some
// row's non-zero taps do not fit.  `padded` = number of rows to emit (rows past count repeat the last).
static bool regularise(const FilterBank &fb, int padded, int P, int &w0, std::vector<int32_t> &out)
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
        if (last[x] - 2 * x - w0 >= 2 * P) return false;
    out.assign((size_t)padded * P, 0);
    std::vector<int16_t> win(2 * P);
    for (int xx = 0; xx < padded; xx++) {
        const int x = std::min(xx, fb.count - 1);
        std::fill(win.begin(), win.end(), (int16_t)0);
        for (int j = 0; j < fb.taps; j++) {
            const int16_t cv = fb.coef[(size_t)x * fb.taps + j];
            if (!cv) continue;
            win[fb.pos[x] + j - 2 * x - w0] = cv;
        }
        for (int k = 0; k < P; k++)
            out[(size_t)xx * P + k] = (int32_t)((uint32_t)(uint16_t)win[2 * k] | ((uint32_t)(uint16_t)win[2 * k + 1] << 16));
    }
    return true;
}

int yuv2x_prepare(const ScalePlan &p, const YuvScaleTiling &g, Yuv2xTables &t)
{
    t.ok = 0;
    const char *off = GMAT_KNOB("GMAT_SCALE_NO_2X");
    if (off && atoi(off)) return 0;
    if (g.fullChroma || g.yuvOut == 2 || g.TW != X2_TW || g.TH != X2_TH) return 0;
    if (!is_yuv420(p.srcFormat) || is_dst10(p.dstFormat)) return 0;                    // the tile geometry assumes half-size chroma planes
    if (p.srcW % 16 || p.chrSrcW % 8 || p.srcW < 16) return 0;
    if (g.rowsL > 48 || g.rowsC > 48) return 0;               // phase 1 covers 48 luma rows and 24 chroma rows per tile (48 in
                                                              // the Lanczos 4:2:0-output variant, checked below)
    t.ntx = g.ntx; t.nty = g.nty;
    // smallest window that holds every row: 10 samples (bicubic, bilinear ...) or 16 (Lanczos-3)
    int P = 0;
    for (int cand : {5, 8}) {
        if (g.vLumEff.pairs > cand || g.vChrEff.pairs > (g.yuvOut ? cand : (cand == 5 ? 2 : 4))) continue;
        if (!regularise(p.hLum, t.ntx * X2_TW, cand, t.w0L, t.hLreg)) continue;
        if (!regularise(p.hChr, t.ntx * (X2_TW / 2), cand, t.w0C, t.hCreg)) continue;
        // every tile's regular window must fit the fixed LDS row lengths
        bool fits = true;
        for (int tc = 0; tc < t.ntx && fits; tc++) {
            const int wl = 2 * tc * X2_TW + t.w0L, wc = 2 * tc * (X2_TW / 2) + t.w0C;
            fits = (wl - (wl & ~15)) + 2 * (X2_TW - 1) + 2 * cand <= x2_colsl(cand) &&
                   (wc - (wc & ~7)) + 2 * (X2_TW / 2 - 1) + 2 * cand <= X2_COLSC;
        }
        if (fits) { P = cand; break; }
    }
    if (!P) return 0;
    if (g.rowsC > 24 && !(g.yuvOut && P == 8)) return 0;
    const int X2_VR = x2_vr(P), VR_C = x2_vr_chroma(P), VR_M = x2_vr_misc(P);
    const int bytes = g.rowsL * x2_colsl(P) * 2 + 2 * g.rowsC * X2_COLSC * 2 + (g.rowsL / 2) * X2_TW * 4 +
                      2 * (g.rowsC / 2) * (X2_TW / 2) * 4 + (X2_TW + X2_TW / 2) * P * 4 + X2_TH * X2_VR * 4 +
                      (g.yuvOut ? (X2_TH / 2) * X2_VRC * 4 : 0);
    if (bytes > 64 * 1024) return 0;
    // per-output-row records for phase 3 (rows past dstH repeat the last one; never stored):
    // luma pairs, chroma pairs (RGB output), luma / chroma window row, luma / chroma accumulator start — at the
    // positions x2_vr_chroma(P) / x2_vr_misc(P) give (layout at the top of this file)
    t.vrec.assign((size_t)t.nty * X2_TH * X2_VR, 0);
    for (int yy = 0; yy < t.nty * X2_TH; yy++) {
        const int y = std::min(yy, p.dstH - 1);
        int32_t *r = &t.vrec[(size_t)yy * X2_VR];
        for (int k = 0; k < g.vLumEff.pairs; k++) r[k] = g.vLumEff.packed[(size_t)y * g.vLumEff.pairs + k];
        r[VR_M] = g.vLumEff.pos_even[y];
        r[VR_M + 2] = g.lumRound[y];
        if (g.yuvOut) continue;                              // chroma rows have their own records (vrecC)
        for (int k = 0; k < g.vChrEff.pairs; k++) r[VR_C + k] = g.vChrEff.packed[(size_t)y * g.vChrEff.pairs + k];
        r[VR_M + 1] = g.vChrEff.pos_even[y];
        r[VR_M + 3] = g.chrRound[y];
    }
    t.vrecC.clear();
    if (g.yuvOut) {
        // [0..6] chroma pairs   [8] window row   [9] accumulator start
        t.vrecC.assign((size_t)t.nty * (X2_TH / 2) * X2_VRC, 0);
        for (int cyy = 0; cyy < t.nty * (X2_TH / 2); cyy++) {
            const int cy = std::min(cyy, p.chrDstH - 1);
            int32_t *r = &t.vrecC[(size_t)cyy * X2_VRC];
            for (int k = 0; k < g.vChrEff.pairs; k++) r[k] = g.vChrEff.packed[(size_t)cy * g.vChrEff.pairs + k];
            r[8] = g.vChrEff.pos_even[cy];
            r[9] = g.chrRound[cy];
        }
    }
    // ---- interior tiles with uniform coefficients (RGB output) --------------------------------------------------
    t.uni = Yuv2xUniform();
    if (t.ntx >= 3 && t.nty >= 3) {
        Yuv2xUniform u;
        const int mc = t.ntx / 2, mr = t.nty / 2;                       // the middle tile provides the candidate rows
        const int32_t *HL = &t.hLreg[(size_t)mc * X2_TW * P], *HC = &t.hCreg[(size_t)mc * (X2_TW / 2) * P];
        auto col_uniform = [&](int tc) {
            for (int x = 0; x < X2_TW; x++)
                if (std::memcmp(&t.hLreg[((size_t)tc * X2_TW + x) * P], HL, P * 4)) return false;
            for (int x = 0; x < X2_TW / 2; x++)
                if (std::memcmp(&t.hCreg[((size_t)tc * (X2_TW / 2) + x) * P], HC, P * 4)) return false;
            return true;
        };
        const int ym = mr * X2_TH;
        const int lp = g.vLumEff.pairs, cp = g.vChrEff.pairs;
        const int aL = g.vLumEff.pos_even[ym] - 2 * ym;
        // vertical chroma: indexed by OUTPUT row for RGB output (window row (y + aC) & ~1, the pair row alternates with
        // the parity of y), by CHROMA row for 4:2:0 output (window row 2 * cy + aCy, one pair row)
        const int cym = ym / 2;
        const int aC = g.yuvOut ? 0 : g.vChrEff.pos_even[ym] - ym;      // ym and pos_even are even
        const int aCy = g.yuvOut ? g.vChrEff.pos_even[cym] - 2 * cym : 0;
        auto row_uniform = [&](int tr) {
            for (int y = tr * X2_TH; y < (tr + 1) * X2_TH; y++) {
                if (y >= p.dstH) return false;
                if (g.vLumEff.pos_even[y] != 2 * y + aL) return false;
                if (std::memcmp(&g.vLumEff.packed[(size_t)y * lp], &g.vLumEff.packed[(size_t)ym * lp], lp * 4)) return false;
                if (g.lumRound[y] != g.lumRound[ym]) return false;
                if (!g.yuvOut) {
                    const int yr = ym + (y & 1);                                    // reference row of the same parity
                    if (g.vChrEff.pos_even[y] != ((y + aC) & ~1) || g.chrRound[y] != g.chrRound[ym]) return false;
                    if (std::memcmp(&g.vChrEff.packed[(size_t)y * cp], &g.vChrEff.packed[(size_t)yr * cp], cp * 4)) return false;
                }
            }
            for (int cy = tr * (X2_TH / 2); g.yuvOut && cy < (tr + 1) * (X2_TH / 2); cy++) {
                if (cy >= p.chrDstH) return false;
                if (g.vChrEff.pos_even[cy] != 2 * cy + aCy || g.chrRound[cy] != g.chrRound[cym]) return false;
                if (std::memcmp(&g.vChrEff.packed[(size_t)cy * cp], &g.vChrEff.packed[(size_t)cym * cp], cp * 4)) return false;
            }
            return true;
        };
        bool ok = lp <= 8 && cp <= (g.yuvOut ? 8 : 4) && col_uniform(mc) && row_uniform(mr) && (aL & 1) == 0 && (aCy & 1) == 0;
        // all four coefficient rows of a quad identical is implied by col_uniform (every column equals HL)
        if (ok) {
            u.tcLo = u.tcHi = mc; u.trLo = u.trHi = mr;
            while (u.tcLo > 0 && col_uniform(u.tcLo - 1)) u.tcLo--;
            while (u.tcHi + 1 < t.ntx && col_uniform(u.tcHi + 1)) u.tcHi++;
            while (u.trLo > 0 && row_uniform(u.trLo - 1)) u.trLo--;
            while (u.trHi + 1 < t.nty && row_uniform(u.trHi + 1)) u.trHi++;
            for (int k = 0; k < P; k++) { u.hL[k] = HL[k]; u.hC[k] = HC[k]; }
            for (int k = 0; k < lp; k++) u.vL[k] = g.vLumEff.packed[(size_t)ym * lp + k];
            for (int k = 0; k < cp; k++) {
                if (g.yuvOut) {
                    u.vCy[k] = g.vChrEff.packed[(size_t)cym * cp + k];
                } else {
                    u.vC[0][k] = g.vChrEff.packed[(size_t)ym * cp + k];
                    u.vC[1][k] = g.vChrEff.packed[(size_t)(ym + 1) * cp + k];
                }
            }
            u.aL = aL; u.aC = aC; u.aCy = aCy; u.lr = g.lumRound[ym]; u.cr = g.chrRound[g.yuvOut ? cym : ym];
            // row windows of the interior tile rows as r0 + trow * d (same count): shrink the row range to where that holds
            u.dL = mr + 1 < t.nty ? g.rowStartL[mr + 1] - g.rowStartL[mr] : 0;
            u.dC = mr + 1 < t.nty ? g.rowStartC[mr + 1] - g.rowStartC[mr] : 0;
            u.r0L = g.rowStartL[mr] - mr * u.dL; u.r0C = g.rowStartC[mr] - mr * u.dC;
            u.nrL = g.rowCountL[mr]; u.nrC = g.rowCountC[mr];
            auto win_ok = [&](int tr) {
                return g.rowStartL[tr] == u.r0L + tr * u.dL && g.rowCountL[tr] == u.nrL &&
                       g.rowStartC[tr] == u.r0C + tr * u.dC && g.rowCountC[tr] == u.nrC;
            };
            int lo = mr, hi = mr;
            while (lo > u.trLo && win_ok(lo - 1)) lo--;
            while (hi < u.trHi && win_ok(hi + 1)) hi++;
            u.trLo = lo; u.trHi = hi;
            t.uni = u;
        }
        if (GMAT_KNOB("GMAT_DEBUG_UNI"))
            logf(LOG_ERROR, "yuv2x uniform tiles: cols [%d, %d] of %d, rows [%d, %d] of %d (P = %d)", t.uni.tcLo, t.uni.tcHi, t.ntx, t.uni.trLo,
                 t.uni.trHi, t.nty, P);
    }
    t.yuvOut = g.yuvOut;
    t.P = P;
    t.vLpairs = g.vLumEff.pairs; t.vCpairs = g.vChrEff.pairs;
    t.ok = bytes;
    return 0;
}

int launch_scale_yuv2x(const Yuv2xArgs &a, int rowsL, int rowsC, int ldsBytes, hipStream_t stream, const Yuv2xFrames *frames,
                       int nframes)
{
    Yuv2xFrames one;
    if (!frames) {
        std::memset(&one, 0, sizeof(one));
        one.y[0] = a.y; one.u[0] = a.u; one.v[0] = a.v; one.dst[0] = a.dst; one.dstU[0] = a.dstU; one.dstV[0] = a.dstV;
        frames = &one; nframes = 1;
    }
    if (nframes < 1 || nframes > kYuv2xMaxFrames) return GMAT_ERR(EINVAL);
    const Yuv2xFrames &fr = *frames;
    // GMAT_SCALE_TILES=2 runs two vertically adjacent tiles per block with the second tile's loads in flight during
    // the first tile's phases 2 and 3.  Measured on MI355X: 12.3-12.4 us against 12.1 us for one tile per block
    // (82 VGPRs instead of 48, and five co-resident blocks per CU already overlap each other's load phases), so
    // the default stays 1.
    static const int tilesEnv = GMAT_KNOB("GMAT_SCALE_TILES") ? atoi(GMAT_KNOB("GMAT_SCALE_TILES")) : 1;
    const int tilesPerBlock = tilesEnv == 2 ? 2 : 1;
    const int ntiles = a.ntx * ((a.nty + tilesPerBlock - 1) / tilesPerBlock);
    if (ntiles <= 0) return 0;
    const dim3 grid(a.xcdRemap ? 8 * ((ntiles + 7) / 8) : ntiles, nframes), block(256);
#define GMAT_X2(Y_, P_) do { if (tilesPerBlock == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv2x_kernel<Y_, P_, 2>), grid, block, (size_t)ldsBytes, stream, a, fr, rowsL, rowsC); \
                              else hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv2x_kernel<Y_, P_, 1>), grid, block, (size_t)ldsBytes, stream, a, fr, rowsL, rowsC); } while (0)
    if (a.P == 5)      { if (a.yuvOut) GMAT_X2(true, 5); else GMAT_X2(false, 5); }
    else if (a.P == 8) { if (a.yuvOut) GMAT_X2(true, 8); else GMAT_X2(false, 8); }
    else return GMAT_ERR(EINVAL);
#undef GMAT_X2
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
