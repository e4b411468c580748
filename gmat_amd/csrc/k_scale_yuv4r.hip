// k_scale_yuv4r.hip — NV12 / YUV420P at exactly a quarter of the size into packed RGB (4K -> 960x540, 1080p -> 480x270: a decoder's frame into a
// low-resolution analysis pass), with the arithmetic of ONE libswscale context (hScale8To15_c on both planes, yuv2rgb_X_c's vertical
// sums and table stage), bit-exact.  The generic plane scaler serves it at ~0.1 of the HBM roofline.
//
// At 4:1 the bicubic filter has ONE phase of 16 taps on [4x - 6, 4x + 9], on both axes of the luma and — an RGB destination keeps
// half-width chroma at the output's full height — on the horizontal axis of the chroma; the chroma's vertical axis is a 2:1 scale, 8
// taps on [2y - 3, 2y + 4].  libswscale folds taps outside the plane onto the edge sample; the host checks that every table row IS the
// interior row on an edge-replicated line (filter_is_edge_replication_ratio / filter_is_edge_replication) and passes the taps as int16
// pairs in the kernel arguments.
//   * a wave owns a strip of 256 output columns; a lane makes 4 pixels of an output row.  Its luma window is the 32 bytes at 4x - 8 (16
//     byte pairs, every pair one half of a dword: output j takes the pairs 2j + 1 .. 2j + 8), its UV window the 48 bytes at position
//     4c - 8 (a dword = the (U, V) of a position pair: output i takes the dwords 2i + 1 .. 2i + 8);
//   * both vertical filters are RUNNING SUMS (see scale_yuv3r_kernel, k_scale_yuv3x1.hip): step y takes the luma rows 4y + 6 .. 4y + 9
//     and the chroma rows 2y + 3, 2y + 4; rows are packed in pairs by v_cvt_pk_i16_i32 (hScale8To15_c's saturation) and a pair feeds
//     each of the four open output rows with one v_dot2; output row y closes at the end of its step.  Slots are static after unrolling
//     four steps; the rows just consumed are re-requested one step ahead;
//   * edge lanes load from a base shifted into the row and repair their registers with selects.
// Parity: held to the oracle (tests/test_parity_down4rgb.py, together with the generic kernel on the same matrix); no vector the
// reference holds is a 4:1 scale.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

namespace gmat {

constexpr int D4_STRIP = 256;                  // output columns per wave: 64 lanes x 4

#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned d4_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ uint4 d4_ld16(const uint8_t *p) { const d4_u32x4 v = *reinterpret_cast<const d4_u32x4 *>(p); return make_uint4(v.x, v.y, v.z, v.w); }
#else
static inline uint4 d4_ld16(const uint8_t *p) { uint4 v; std::memcpy(&v, p, 16); return v; }
#endif
__device__ __forceinline__ int d4_dot2(int packed_ab, int packed_cd, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, packed_ab), __builtin_bit_cast(short2v, packed_cd), acc, true);
}
__device__ __forceinline__ unsigned d4_rep(unsigned v, unsigned sel) { return __builtin_amdgcn_perm(v, v, sel); }

// NV: NV12 source (interleaved UV); else YUV420P (a lane's windows on the U and V planes are the 24 bytes at position 4c - 8 of each)
template <int DST, bool NV>
__global__ __launch_bounds__(256) void scale_yuv4r_kernel(Yuv4rArgs a, Yuv2xFrames fr)
{
    constexpr bool BGR = (DST & 1) != 0;
    constexpr int BPP = DST >= 2 ? 4 : 3;
    __shared__ int2 lutV[256], lutU[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        // term_R = lutV[V].x, term_G = lutV[V].y + lutU[U].x, term_B = lutU[U].y; channel = byte 2 of clamp(term + Y * cy, 0, 0xFFFFFF)
        const Yuv2RgbConsts &k = a.y2r;
        lutV[tid] = make_int2(k.base + m24(k.offR + (m24(tid, k.crv) >> 16), k.cy), m24(m24(tid, k.cgv) >> 16, k.cy));
        lutU[tid] = make_int2(k.base + m24(k.offG + (m24(tid, k.cgu) >> 16), k.cy), k.base + m24(k.offB + (m24(tid, k.cbu) >> 16), k.cy));
        __syncthreads();
    }
    int lin = blockIdx.x;
    if (a.xcdRemap) {
        const int chunk = (a.nblk + 7) >> 3;
        lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    }
    if (lin >= a.nblk) return;
    const int unit = lin * 4 + wave;                             // (segment, strip) units packed densely: the waves share nothing
    if (unit >= a.nseg * a.nstrips) return;
    const int seg = __builtin_amdgcn_readfirstlane(unit / a.nstrips);
    const int X0 = (unit - seg * a.nstrips) * D4_STRIP;
    const int y0 = seg * a.segRows, nOut = min(a.segRows, a.dstH - y0);
    const int nSteps = nOut + 3;                                 // steps y0 - 3 .. y0 + nOut - 1: three open the sums of row y0
    // Odd segments walk UPWARD (k_scale_yuv3x1.hip): the same steps over the rows in descending order — step y then takes the FIRST rows of
    // row y's windows, 4y - 3 .. 4y - 6 and 2y - 2, 2y - 3 — with the tap pairs taken in reverse and their halves swapped.  A segment and its
    // neighbour then reach their common boundary at the same time and the warm-up rows are read from HBM once (L2 hits for the other).
    const int up = a.updown & seg & 1;
    auto sw16 = [](int32_t v) { return (int32_t)(((uint32_t)v >> 16) | ((uint32_t)v << 16)); };
    int32_t vvL[8], vvC[4];                                      // wave-uniform: scalar registers
#pragma unroll
    for (int k = 0; k < 8; k++) { const int32_t r = sw16(a.vL[7 - k]); vvL[k] = a.vL[k] ^ ((a.vL[k] ^ r) & -up); }
#pragma unroll
    for (int k = 0; k < 4; k++) { const int32_t r = sw16(a.vC[3 - k]); vvC[k] = a.vC[k] ^ ((a.vC[k] ^ r) & -up); }
    const int yFirst = up ? y0 + nOut + 2 : y0 - 3, yDir = up ? -1 : 1;
    // rows of step y: luma lumaBase(y) + yDir * r, chroma chrBase(y) + yDir * r; never outside the rows the segment needs
    const int lLo = 4 * y0 - 6, lHi = 4 * (y0 + nOut - 1) + 9, cLo = 2 * y0 - 3, cHi = 2 * (y0 + nOut - 1) + 4;
    auto lumaRow = [&](int y, int r) { return min(max(up ? 4 * y - 3 - r : 4 * y + 6 + r, lLo), lHi); };
    auto chrRow = [&](int y, int r) { return min(max(up ? 2 * y - 2 - r : 2 * y + 3 + r, cLo), cHi); };
    const int srcW = 4 * a.dstW, srcH = 4 * a.dstH, chrH = srcH >> 1;
    const uint8_t *py = fr.y[blockIdx.y], *puv = fr.u[blockIdx.y], *pv = fr.v[blockIdx.y];
    uint8_t *pd = fr.dst[blockIdx.y];

    const int xo = X0 + 4 * lane;
    const bool active = xo < a.dstW;
    const int xc = active ? xo : a.dstW - 4;                     // idle lanes shadow the last group
    const bool edgeWave = X0 == 0 || 4 * (X0 + D4_STRIP) + 8 > srcW;
    const bool isLeft = xc == 0, isRight = xc == a.dstW - 4;
    // luma window: bytes 4 xc - 8 .. + 31; at the frame's edges it reaches 8 bytes outside: those lanes load 8 bytes further in
    const unsigned boL = (unsigned)(4 * xc - 8), lboL = boL + (isLeft ? 8u : 0u) - (isRight ? 8u : 0u);
    // UV window: positions 2 xc - 8 .. + 23 (48 bytes); 8 positions outside at the edges: 16 bytes further in
    const unsigned boC = 2u * (unsigned)(2 * xc - 8), lboC = boC + (isLeft ? 16u : 0u) - (isRight ? 16u : 0u);
    // planar chroma: 24 bytes of each plane at position 2 xc - 8; 8 bytes further in at the edges
    const unsigned boP = (unsigned)(2 * xc - 8), lboP = boP + (isLeft ? 8u : 0u) - (isRight ? 8u : 0u);
    (void)srcW;

    auto loadL = [&](int row, unsigned (&d)[8], auto edge_c) {
        const uint8_t *p = py + ((unsigned)min(max(row, 0), srcH - 1) * (unsigned)a.ys + (decltype(edge_c)::value ? lboL : boL));
        const uint4 t = d4_ld16(p), u = d4_ld16(p + 16);
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; d[4] = u.x; d[5] = u.y; d[6] = u.z; d[7] = u.w;
    };
    auto loadC = [&](int row, unsigned (&d)[12], auto edge_c) {
        const unsigned rc = (unsigned)min(max(row, 0), chrH - 1);
        if constexpr (NV) {
            const uint8_t *p = puv + (rc * (unsigned)a.us + (decltype(edge_c)::value ? lboC : boC));
            const uint4 t = d4_ld16(p), u = d4_ld16(p + 16), v = d4_ld16(p + 32);
            d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; d[4] = u.x; d[5] = u.y; d[6] = u.z; d[7] = u.w; d[8] = v.x; d[9] = v.y; d[10] = v.z; d[11] = v.w;
        } else {
            // d[0 .. 5]: the U window, d[6 .. 11]: the V window
            const unsigned off = decltype(edge_c)::value ? lboP : boP;
            const uint8_t *pu_ = puv + (rc * (unsigned)a.us + off), *pv_ = pv + (rc * (unsigned)a.vs + off);
            const uint4 t = d4_ld16(pu_), v = d4_ld16(pv_);
            const unsigned t4 = *reinterpret_cast<const unsigned *>(pu_ + 16), t5 = *reinterpret_cast<const unsigned *>(pu_ + 20);
            const unsigned v4 = *reinterpret_cast<const unsigned *>(pv_ + 16), v5 = *reinterpret_cast<const unsigned *>(pv_ + 20);
            d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; d[4] = t4; d[5] = t5; d[6] = v.x; d[7] = v.y; d[8] = v.z; d[9] = v.w; d[10] = v4; d[11] = v5;
        }
    };
    // hScale8To15_c of a luma row: the lane's 4 sums >> 7 (the saturation is the pack's)
    auto hrowL = [&](const unsigned (&src)[8], auto edge_c, int (&s)[4]) {
        unsigned d[8];
#pragma unroll
        for (int i = 0; i < 8; i++) d[i] = src[i];
        if constexpr (decltype(edge_c)::value) {
            const unsigned first = d4_rep(src[0], 0x00000000u), last = d4_rep(src[7], 0x03030303u);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const unsigned fromLeft = i < 2 ? first : src[i - 2], fromRight = i > 5 ? last : src[i + 2];
                d[i] = isLeft ? fromLeft : isRight ? fromRight : src[i];
            }
        }
        int H[16];                                               // byte pairs (2h, 2h + 1) of the window as int16 pairs; H[0] and H[15] carry no tap
#pragma unroll
        for (int h = 1; h < 15; h++)
            H[h] = (int)__builtin_amdgcn_perm(0u, d[h >> 1], (h & 1) ? 0x0C030C02u : 0x0C010C00u);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int acc = 0;
#pragma unroll
            for (int m = 0; m < 8; m++) acc = d4_dot2(H[2 * j + 1 + m], a.hL[m], acc);
            s[j] = acc >> 7;
        }
    };
    // ... of a chroma row: U0 V0 U1 V1 (>> 7)
    auto hrowC = [&](const unsigned (&src)[12], auto edge_c, int (&s)[4]) {
        unsigned d[12];
#pragma unroll
        for (int i = 0; i < 12; i++) d[i] = src[i];
        int pU[12], pV[12];                                      // position pairs (2i, 2i + 1) per channel; pairs 0 and 11 carry no tap
        if constexpr (NV) {
            if constexpr (decltype(edge_c)::value) {
                const unsigned first = d4_rep(src[0], 0x01000100u), last = d4_rep(src[11], 0x03020302u);
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    const unsigned fromLeft = i < 4 ? first : src[i - 4], fromRight = i > 7 ? last : src[i + 4];
                    d[i] = isLeft ? fromLeft : isRight ? fromRight : src[i];
                }
            }
#pragma unroll
            for (int i = 1; i < 11; i++) {
                pU[i] = (int)__builtin_amdgcn_perm(0u, d[i], 0x0C020C00u);
                pV[i] = (int)__builtin_amdgcn_perm(0u, d[i], 0x0C030C01u);
            }
        } else {
            if constexpr (decltype(edge_c)::value) {
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                    const unsigned first = d4_rep(src[6 * pl], 0x00000000u), last = d4_rep(src[6 * pl + 5], 0x03030303u);
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        const unsigned fromLeft = i < 2 ? first : src[6 * pl + i - 2], fromRight = i > 3 ? last : src[6 * pl + i + 2];
                        d[6 * pl + i] = isLeft ? fromLeft : isRight ? fromRight : src[6 * pl + i];
                    }
                }
            }
#pragma unroll
            for (int i = 1; i < 11; i++) {                          // pair i = the half i & 1 of the plane's dword i >> 1
                pU[i] = (int)__builtin_amdgcn_perm(0u, d[i >> 1], (i & 1) ? 0x0C030C02u : 0x0C010C00u);
                pV[i] = (int)__builtin_amdgcn_perm(0u, d[6 + (i >> 1)], (i & 1) ? 0x0C030C02u : 0x0C010C00u);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            int u = 0, v = 0;
#pragma unroll
            for (int m = 0; m < 8; m++) { u = d4_dot2(pU[2 * i + 1 + m], a.hC[m], u); v = d4_dot2(pV[2 * i + 1 + m], a.hC[m], v); }
            s[2 * i] = u >> 7; s[2 * i + 1] = v >> 7;
        }
    };

    int accL[4][4], accC[4][4];                                  // [slot][sample]: open output rows; luma 4 columns, chroma U0 V0 U1 V1
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int q = 0; q < 4; q++) accL[s][q] = accC[s][q] = 0;
    unsigned bufL[4][8], bufC[2][12];
    const unsigned dstOff = (unsigned)xo * BPP;

    auto emit = [&](int yo, const int (&YS)[4], const int (&CS)[4]) {
        if (yo < y0 || yo >= y0 + nOut) return;                  // wave-uniform: the warm-up steps
        const int iU[2] = {clip_u8_shr(CS[0], 19), clip_u8_shr(CS[2], 19)}, iV[2] = {clip_u8_shr(CS[1], 19), clip_u8_shr(CS[3], 19)};
        unsigned c0[4], c1[4], c2[4];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int2 tv = lutV[iV[c]], tu = lutU[iU[c]];
            const int tr = BGR ? tu.y : tv.x, tg = tv.y + tu.x, tb = BGR ? tv.x : tu.y;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int q = 2 * c + h;
                const int yc = m24(YS[q] >> 19, a.y2r.cy);
                c0[q] = (unsigned)min(max(tr + yc, 0), 0xFFFFFF);
                c1[q] = (unsigned)min(max(tg + yc, 0), 0xFFFFFF);
                c2[q] = (unsigned)min(max(tb + yc, 0), 0xFFFFFF);
            }
        }
        if (active) {
            uint8_t *d = pd + (unsigned)((unsigned)yo * (unsigned)a.ds + dstOff);
#define D4_B2PAIR(lo, hi) __builtin_amdgcn_perm((hi), (lo), 0x0C0C0602u)
            if (BPP == 4) {
                uint4 o4;
                o4.x = D4_B2PAIR(c0[0], c1[0]) | (D4_B2PAIR(c2[0], 0u) << 16) | 0xFF000000u;
                o4.y = D4_B2PAIR(c0[1], c1[1]) | (D4_B2PAIR(c2[1], 0u) << 16) | 0xFF000000u;
                o4.z = D4_B2PAIR(c0[2], c1[2]) | (D4_B2PAIR(c2[2], 0u) << 16) | 0xFF000000u;
                o4.w = D4_B2PAIR(c0[3], c1[3]) | (D4_B2PAIR(c2[3], 0u) << 16) | 0xFF000000u;
                st_stream(d, o4);
            } else {
                uint3 o3;           // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
                o3.x = D4_B2PAIR(c0[0], c1[0]) | (D4_B2PAIR(c2[0], c0[1]) << 16);
                o3.y = D4_B2PAIR(c1[1], c2[1]) | (D4_B2PAIR(c0[2], c1[2]) << 16);
                o3.z = D4_B2PAIR(c2[2], c0[3]) | (D4_B2PAIR(c1[3], c2[3]) << 16);
                st_stream(d, o3);
            }
#undef D4_B2PAIR
        }
    };

    auto body = [&](const int i, auto ph_c, auto edge_c) {
        constexpr int PH = decltype(ph_c)::value;                // i & 3: names the slots
        const int y = yFirst + yDir * i;                         // the output row this step closes
        // slots: row y in PH, the next three rows of the walk in PH + 1, PH + 2 and (opening here) PH + 3 (mod 4)
        constexpr int S0 = PH & 3, S1 = (PH + 1) & 3, S2 = (PH + 2) & 3, S3 = (PH + 3) & 3;
        int YS[4], CS[4];
        // luma quad u = y + 1: rows 4u + 2 .. 4u + 5, taps (12 + r) / (8 + r) / (4 + r) / r for the rows y, y + 1, y + 2, y + 3
        {
            int hA[4], hB[4], hC[4], hD[4];
            hrowL(bufL[0], edge_c, hA); loadL(lumaRow(y + yDir, 0), bufL[0], edge_c);
            hrowL(bufL[1], edge_c, hB); loadL(lumaRow(y + yDir, 1), bufL[1], edge_c);
            hrowL(bufL[2], edge_c, hC); loadL(lumaRow(y + yDir, 2), bufL[2], edge_c);
            hrowL(bufL[3], edge_c, hD); loadL(lumaRow(y + yDir, 3), bufL[3], edge_c);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int ab = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(hA[q], hB[q]));
                const int cd = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(hC[q], hD[q]));
                YS[q] = d4_dot2(cd, vvL[7], d4_dot2(ab, vvL[6], accL[S0][q]));
                accL[S1][q] = d4_dot2(cd, vvL[5], d4_dot2(ab, vvL[4], accL[S1][q]));
                accL[S2][q] = d4_dot2(cd, vvL[3], d4_dot2(ab, vvL[2], accL[S2][q]));
                accL[S3][q] = d4_dot2(cd, vvL[1], d4_dot2(ab, vvL[0], a.lr));
            }
        }
        // chroma pair m = y + 2: rows 2m - 1, 2m, taps (6, 7) / (4, 5) / (2, 3) / (0, 1) for the rows y, y + 1, y + 2, y + 3
        {
            int hA[4], hB[4];
            hrowC(bufC[0], edge_c, hA); loadC(chrRow(y + yDir, 0), bufC[0], edge_c);
            hrowC(bufC[1], edge_c, hB); loadC(chrRow(y + yDir, 1), bufC[1], edge_c);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int ab = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(hA[q], hB[q]));
                CS[q] = d4_dot2(ab, vvC[3], accC[S0][q]);
                accC[S1][q] = d4_dot2(ab, vvC[2], accC[S1][q]);
                accC[S2][q] = d4_dot2(ab, vvC[1], accC[S2][q]);
                accC[S3][q] = d4_dot2(ab, vvC[0], a.cr);
            }
        }
        emit(y, YS, CS);
    };
    auto run = [&](auto edge_c) {
#pragma unroll
        for (int r = 0; r < 4; r++) loadL(lumaRow(yFirst, r), bufL[r], edge_c);
        loadC(chrRow(yFirst, 0), bufC[0], edge_c); loadC(chrRow(yFirst, 1), bufC[1], edge_c);
        for (int i0 = 0; i0 < nSteps; i0 += 4) {
            body(i0, std::integral_constant<int, 0>(), edge_c);
            if (i0 + 1 < nSteps) body(i0 + 1, std::integral_constant<int, 1>(), edge_c);
            if (i0 + 2 < nSteps) body(i0 + 2, std::integral_constant<int, 2>(), edge_c);
            if (i0 + 3 < nSteps) body(i0 + 3, std::integral_constant<int, 3>(), edge_c);
        }
    };
    if (edgeWave) run(std::true_type()); else run(std::false_type());
}

// ---------------------------------------------------------------------------------------------
// scale_yuv4x1_kernel: the same ratio from 4:2:0 into 4:2:0 (NV12 -> NV12, YUV420P -> YUV420P: the 540p / 270p rungs of a transcoding
// ladder), every plane walked on its own with the windows, pair packing and running sums of the kernel above; the chroma plane is
// scaled 4:1 on both axes here, and yuv2planeX_8_c / yuv2nv12cX_c clip the sums to 8 bits.
// ---------------------------------------------------------------------------------------------
struct D4Plane {
    const uint8_t *src; uint8_t *dst;
    int ss, ds, dstW, dstH;                    // widths in samples (UV plane: in UV positions)
    int32_t h[8], v[8];
    int rnd;
};

// NW = dwords of a lane's window per row; LOAD(row, d, edge_c) / HROW(d, edge_c, s[4]) / STORE(y, w[4]) are the plane kind's
template <int NW, typename Load, typename HRow, typename Store>
__device__ __forceinline__ void d4_walk(const D4Plane &P, int y0, int nOut, bool edgeWave, int up, Load &&load, HRow &&hrow, Store &&store)
{
    auto sw16 = [](int32_t v) { return (int32_t)(((uint32_t)v >> 16) | ((uint32_t)v << 16)); };
    int32_t vv[8];                                               // upward segments: the tap pairs reversed, halves swapped (see the kernel above)
#pragma unroll
    for (int k = 0; k < 8; k++) { const int32_t r = sw16(P.v[7 - k]); vv[k] = P.v[k] ^ ((P.v[k] ^ r) & -up); }
    const int yFirst = up ? y0 + nOut + 2 : y0 - 3, yDir = up ? -1 : 1;
    const int rLo = 4 * y0 - 6, rHi = 4 * (y0 + nOut - 1) + 9;
    auto rowOf = [&](int y, int r) { return min(max(up ? 4 * y - 3 - r : 4 * y + 6 + r, rLo), rHi); };
    int acc[4][4];
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[s][q] = 0;
    unsigned buf[4][NW];
    const int nSteps = nOut + 3;
    auto body = [&](const int i, auto ph_c, auto edge_c) {
        constexpr int PH = decltype(ph_c)::value;
        constexpr int S0 = PH & 3, S1 = (PH + 1) & 3, S2 = (PH + 2) & 3, S3 = (PH + 3) & 3;
        const int y = yFirst + yDir * i;
        int hA[4], hB[4], hC[4], hD[4];
        hrow(buf[0], edge_c, hA); load(rowOf(y + yDir, 0), buf[0], edge_c);
        hrow(buf[1], edge_c, hB); load(rowOf(y + yDir, 1), buf[1], edge_c);
        hrow(buf[2], edge_c, hC); load(rowOf(y + yDir, 2), buf[2], edge_c);
        hrow(buf[3], edge_c, hD); load(rowOf(y + yDir, 3), buf[3], edge_c);
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int ab = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(hA[q], hB[q]));
            const int cd = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(hC[q], hD[q]));
            w[q] = (unsigned)clip_u8_shr(d4_dot2(cd, vv[7], d4_dot2(ab, vv[6], acc[S0][q])), 19);
            acc[S1][q] = d4_dot2(cd, vv[5], d4_dot2(ab, vv[4], acc[S1][q]));
            acc[S2][q] = d4_dot2(cd, vv[3], d4_dot2(ab, vv[2], acc[S2][q]));
            acc[S3][q] = d4_dot2(cd, vv[1], d4_dot2(ab, vv[0], P.rnd));
        }
        if (y >= y0 && y < y0 + nOut) store(y, w);               // wave-uniform
    };
    auto run = [&](auto edge_c) {
#pragma unroll
        for (int r = 0; r < 4; r++) load(rowOf(yFirst, r), buf[r], edge_c);
        for (int i0 = 0; i0 < nSteps; i0 += 4) {
            body(i0, std::integral_constant<int, 0>(), edge_c);
            if (i0 + 1 < nSteps) body(i0 + 1, std::integral_constant<int, 1>(), edge_c);
            if (i0 + 2 < nSteps) body(i0 + 2, std::integral_constant<int, 2>(), edge_c);
            if (i0 + 3 < nSteps) body(i0 + 3, std::integral_constant<int, 3>(), edge_c);
        }
    };
    if (edgeWave) run(std::true_type()); else run(std::false_type());
}

// one single-channel plane: the output rows [y0, y0 + nOut) of the strip at X0
__device__ __forceinline__ void d4_walk_plane(const D4Plane &P, int X0, int y0, int nOut, int up, int lane)
{
    const int srcW = 4 * P.dstW, srcH = 4 * P.dstH;
    const int xo = X0 + 4 * lane;
    const bool active = xo < P.dstW;
    const int xc = active ? xo : P.dstW - 4;
    const bool edgeWave = X0 == 0 || 4 * (X0 + D4_STRIP) + 8 > srcW;
    const bool isLeft = xc == 0, isRight = xc == P.dstW - 4;
    const unsigned bo = (unsigned)(4 * xc - 8), lbo = bo + (isLeft ? 8u : 0u) - (isRight ? 8u : 0u);
    auto load = [&](int row, unsigned (&d)[8], auto edge_c) {
        const uint8_t *p = P.src + ((unsigned)min(max(row, 0), srcH - 1) * (unsigned)P.ss + (decltype(edge_c)::value ? lbo : bo));
        const uint4 t = d4_ld16(p), u = d4_ld16(p + 16);
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; d[4] = u.x; d[5] = u.y; d[6] = u.z; d[7] = u.w;
    };
    auto hrow = [&](const unsigned (&src)[8], auto edge_c, int (&s)[4]) {
        unsigned d[8];
#pragma unroll
        for (int i = 0; i < 8; i++) d[i] = src[i];
        if constexpr (decltype(edge_c)::value) {
            const unsigned first = d4_rep(src[0], 0x00000000u), last = d4_rep(src[7], 0x03030303u);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const unsigned fromLeft = i < 2 ? first : src[i - 2], fromRight = i > 5 ? last : src[i + 2];
                d[i] = isLeft ? fromLeft : isRight ? fromRight : src[i];
            }
        }
        int H[16];
#pragma unroll
        for (int h = 1; h < 15; h++)
            H[h] = (int)__builtin_amdgcn_perm(0u, d[h >> 1], (h & 1) ? 0x0C030C02u : 0x0C010C00u);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int acc = 0;
#pragma unroll
            for (int m = 0; m < 8; m++) acc = d4_dot2(H[2 * j + 1 + m], P.h[m], acc);
            s[j] = acc >> 7;
        }
    };
    auto store = [&](int y, const unsigned (&w)[4]) {
        if (active) *reinterpret_cast<unsigned *>(P.dst + ((unsigned)y * (unsigned)P.ds + (unsigned)xo)) = w[0] | (w[1] << 8) | (w[2] << 16) | (w[3] << 24);
    };
    d4_walk<8>(P, y0, nOut, edgeWave, up, load, hrow, store);
}

// NV12's interleaved UV plane: a lane makes 2 UV output positions (4 bytes)
__device__ __forceinline__ void d4_walk_uv(const D4Plane &P, int X0, int y0, int nOut, int up, int lane)
{
    const int srcW = 4 * P.dstW, srcH = 4 * P.dstH;              // in UV positions
    const int co = X0 + 2 * lane;
    const bool active = co < P.dstW;
    const int cc = active ? co : P.dstW - 2;
    const bool edgeWave = X0 == 0 || 4 * (X0 + D4_STRIP / 2) + 8 > srcW;
    const bool isLeft = cc == 0, isRight = cc == P.dstW - 2;
    const unsigned bo = 2u * (unsigned)(4 * cc - 8), lbo = bo + (isLeft ? 16u : 0u) - (isRight ? 16u : 0u);
    auto load = [&](int row, unsigned (&d)[12], auto edge_c) {
        const uint8_t *p = P.src + ((unsigned)min(max(row, 0), srcH - 1) * (unsigned)P.ss + (decltype(edge_c)::value ? lbo : bo));
        const uint4 t = d4_ld16(p), u = d4_ld16(p + 16), v = d4_ld16(p + 32);
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; d[4] = u.x; d[5] = u.y; d[6] = u.z; d[7] = u.w; d[8] = v.x; d[9] = v.y; d[10] = v.z; d[11] = v.w;
    };
    auto hrow = [&](const unsigned (&src)[12], auto edge_c, int (&s)[4]) {
        unsigned d[12];
#pragma unroll
        for (int i = 0; i < 12; i++) d[i] = src[i];
        if constexpr (decltype(edge_c)::value) {
            const unsigned first = d4_rep(src[0], 0x01000100u), last = d4_rep(src[11], 0x03020302u);
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const unsigned fromLeft = i < 4 ? first : src[i - 4], fromRight = i > 7 ? last : src[i + 4];
                d[i] = isLeft ? fromLeft : isRight ? fromRight : src[i];
            }
        }
        int pU[12], pV[12];
#pragma unroll
        for (int i = 1; i < 11; i++) {
            pU[i] = (int)__builtin_amdgcn_perm(0u, d[i], 0x0C020C00u);
            pV[i] = (int)__builtin_amdgcn_perm(0u, d[i], 0x0C030C01u);
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            int u = 0, v = 0;
#pragma unroll
            for (int m = 0; m < 8; m++) { u = d4_dot2(pU[2 * i + 1 + m], P.h[m], u); v = d4_dot2(pV[2 * i + 1 + m], P.h[m], v); }
            s[2 * i] = u >> 7; s[2 * i + 1] = v >> 7;
        }
    };
    auto store = [&](int y, const unsigned (&w)[4]) {
        if (active) *reinterpret_cast<unsigned *>(P.dst + ((unsigned)y * (unsigned)P.ds + 2u * (unsigned)co)) = w[0] | (w[1] << 8) | (w[2] << 16) | (w[3] << 24);
    };
    d4_walk<12>(P, y0, nOut, edgeWave, up, load, hrow, store);
}

__device__ __forceinline__ D4Plane d4_plane(const uint8_t *src, uint8_t *dst, int ss, int ds, int dstW, int dstH,
                                            const int32_t (&h)[8], const int32_t (&v)[8], int rnd)
{
    D4Plane P;
    P.src = src; P.dst = dst; P.ss = ss; P.ds = ds; P.dstW = dstW; P.dstH = dstH; P.rnd = rnd;
#pragma unroll
    for (int k = 0; k < 8; k++) { P.h[k] = h[k]; P.v[k] = v[k]; }
    return P;
}

// blockIdx.x: [0, nblkL) luma workgroups, then the chroma workgroups; a wave's unit of work is one (segment, strip) pair, packed
// densely (unit = 4 * workgroup + wave, segment-major).  blockIdx.y = frame.
template <bool NV>
__global__ __launch_bounds__(256) void scale_yuv4x1_kernel(Yuv4x1Args a, Yuv2xFrames fr)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    int lin = blockIdx.x;
    if (a.xcdRemap) {
        const int chunk = (a.nblk + 7) >> 3;
        lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    }
    if (lin >= a.nblk) return;
    const int f = blockIdx.y;
    if (lin < a.nblkL) {
        const int unit = lin * 4 + wave;
        if (unit >= a.nsegL * a.nsgL) return;
        const int seg = __builtin_amdgcn_readfirstlane(unit / a.nsgL);
        const int X0 = (unit - seg * a.nsgL) * D4_STRIP;
        const int y0 = seg * a.segRows;
        const D4Plane P = d4_plane(fr.y[f], fr.dst[f], a.ys, a.ds, a.dstW, a.dstH, a.hL, a.vL, a.lr);
        d4_walk_plane(P, X0, y0, min(a.segRows, a.dstH - y0), a.updown & seg & 1, lane);
        return;
    }
    int unit = (lin - a.nblkL) * 4 + wave;
    const int per = a.nsegC * a.nsgC;                            // units of one chroma plane
    if (NV) {
        if (unit >= per) return;
        const int seg = __builtin_amdgcn_readfirstlane(unit / a.nsgC);
        const int X0 = (unit - seg * a.nsgC) * (D4_STRIP / 2);
        const int y0 = seg * a.segRows;
        const D4Plane P = d4_plane(fr.u[f], fr.dstU[f], a.us, a.dsU, a.chrDstW, a.chrDstH, a.hC, a.vC, a.cr);
        d4_walk_uv(P, X0, y0, min(a.segRows, a.chrDstH - y0), a.updown & seg & 1, lane);
    } else {
        if (unit >= 2 * per) return;
        const int pl = __builtin_amdgcn_readfirstlane(unit >= per ? 1 : 0);
        unit -= pl * per;
        const int seg = __builtin_amdgcn_readfirstlane(unit / a.nsgC);
        const int X0 = (unit - seg * a.nsgC) * D4_STRIP;
        const int y0 = seg * a.segRows;
        const D4Plane P = d4_plane(pl ? fr.v[f] : fr.u[f], pl ? fr.dstV[f] : fr.dstU[f], pl ? a.vs : a.us, pl ? a.dsV : a.dsU,
                                   a.chrDstW, a.chrDstH, a.hC, a.vC, a.cr);
        d4_walk_plane(P, X0, y0, min(a.segRows, a.chrDstH - y0), a.updown & seg & 1, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int yuv4r_prepare(const ScalePlan &p, const YuvScaleTiling &g, Yuv4rTables &t)
{
    t = Yuv4rTables();
    const char *off = GMAT_KNOB("GMAT_SCALE_NO_STRIP");
    if (off && atoi(off)) return 0;
    if (g.fullChroma || g.yuvOut) return 0;
    if (p.srcFormat != GMAT_PIX_FMT_NV12 && p.srcFormat != GMAT_PIX_FMT_YUV420P) return 0;
    if (!(p.dstFormat == GMAT_PIX_FMT_RGB24 || p.dstFormat == GMAT_PIX_FMT_BGR24 || p.dstFormat == GMAT_PIX_FMT_RGBA ||
          p.dstFormat == GMAT_PIX_FMT_BGRA)) return 0;
    if (p.srcW != 4 * p.dstW || p.srcH != 4 * p.dstH || p.dstW % 4 || p.dstW < 32 || p.dstH < 8) return 0;
    if (p.chrSrcW * 2 != p.srcW || p.chrSrcH * 2 != p.srcH || p.chrDstW * 2 != p.dstW || p.chrDstH != p.dstH) return 0;
    if (!filter_is_edge_replication_ratio(p.hLum, p.srcW, 4, 6, 8, t.hL)) return 0;
    if (!filter_is_edge_replication_ratio(p.hChr, p.chrSrcW, 4, 6, 8, t.hC)) return 0;
    if (!filter_is_edge_replication_ratio(g.vLumEff, p.srcH, 4, 6, 8, t.vL)) return 0;
    if (!filter_is_edge_replication(g.vChrEff, p.chrSrcH, t.vC)) return 0;
    for (int y = 0; y < p.dstH; y++) if (g.lumRound[y] != g.lumRound[0]) return 0;
    for (int y = 0; y < p.chrDstH; y++) if (g.chrRound[y] != g.chrRound[0]) return 0;
    t.lr = g.lumRound[0]; t.cr = g.chrRound[0];
    t.ok = 1;
    return 0;
}

int launch_scale_yuv4r(const Yuv4rArgs &a0, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (!frames || nframes < 1 || nframes > kYuv2xMaxFrames) return GMAT_ERR(EINVAL);
    Yuv4rArgs a = a0;
    a.nstrips = (a.dstW + D4_STRIP - 1) / D4_STRIP;
    const char *segStr = GMAT_KNOB("GMAT_STRIP_ROWS");              // tuning / test override: output rows per segment
    int seg = segStr ? atoi(segStr) : 0;
    if (seg <= 0) {
        // a segment of n output rows walks n + 3 steps of 4 luma + 2 chroma rows.  Measured on 4K -> 960x540 (profiles/r02ze_down4rgb.txt):
        // 32 frames per launch 17 rows 3.6 - 3.8 us per frame (9: 4.05, 29: 4.2); one frame 3 rows 10.0 us (5: 12.4, 2: 13.1)
        const long rows = (long)a.dstH * a.nstrips * nframes;
        seg = (int)std::min(45L, std::max(3L, (rows + 4095) / 4096));
    }
    a.segRows = seg;
    a.nseg = (a.dstH + seg - 1) / seg;
    a.nblk = (a.nseg * a.nstrips + 3) / 4;
    a.xcdRemap = 1;
    { const char *ud = GMAT_KNOB("GMAT_STRIP_UPDOWN"); a.updown = !(ud && !atoi(ud)); }      // 0: every segment walks downward (test / measurement)
    const dim3 grid(8 * ((a.nblk + 7) / 8), nframes), block(256);
#define GMAT_D4R(D) do { if (a.nv12) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv4r_kernel<D, true>), grid, block, 0, stream, a, *frames); \
                        else        hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv4r_kernel<D, false>), grid, block, 0, stream, a, *frames); } while (0)
    switch (a.dstFormat) {
    case GMAT_PIX_FMT_RGB24: GMAT_D4R(0); break;
    case GMAT_PIX_FMT_BGR24: GMAT_D4R(1); break;
    case GMAT_PIX_FMT_RGBA:  GMAT_D4R(2); break;
    case GMAT_PIX_FMT_BGRA:  GMAT_D4R(3); break;
    default: return GMAT_ERR(EINVAL);
    }
#undef GMAT_D4R
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

int yuv4x1_prepare(const ScalePlan &p, const YuvScaleTiling &g, Yuv4x1Tables &t)
{
    t = Yuv4x1Tables();
    const char *off = GMAT_KNOB("GMAT_SCALE_NO_STRIP");
    if (off && atoi(off)) return 0;
    if (g.yuvOut != 1) return 0;
    const bool nv = p.srcFormat == GMAT_PIX_FMT_NV12 && p.dstFormat == GMAT_PIX_FMT_NV12;
    const bool pl = p.srcFormat == GMAT_PIX_FMT_YUV420P && p.dstFormat == GMAT_PIX_FMT_YUV420P;
    if (!nv && !pl) return 0;
    // a lane makes 4 samples of a plane (2 positions of the UV plane)
    if (p.srcW != 4 * p.dstW || p.srcH != 4 * p.dstH || p.dstW % (nv ? 4 : 8) || p.dstW < 64 || p.dstH < 16 || (p.dstH & 1)) return 0;
    if (p.chrDstW * 2 != p.dstW || p.chrDstH * 2 != p.dstH || p.chrSrcW != 4 * p.chrDstW || p.chrSrcH != 4 * p.chrDstH) return 0;
    if (!filter_is_edge_replication_ratio(p.hLum, p.srcW, 4, 6, 8, t.hL)) return 0;
    if (!filter_is_edge_replication_ratio(p.hChr, p.chrSrcW, 4, 6, 8, t.hC)) return 0;
    if (!filter_is_edge_replication_ratio(g.vLumEff, p.srcH, 4, 6, 8, t.vL)) return 0;
    if (!filter_is_edge_replication_ratio(g.vChrEff, p.chrSrcH, 4, 6, 8, t.vC)) return 0;
    for (int y = 0; y < p.dstH; y++) if (g.lumRound[y] != g.lumRound[0]) return 0;
    for (int y = 0; y < p.chrDstH; y++) if (g.chrRound[y] != g.chrRound[0]) return 0;
    t.lr = g.lumRound[0]; t.cr = g.chrRound[0];
    t.ok = 1;
    return 0;
}

int launch_scale_yuv4x1(const Yuv4x1Args &a0, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (!frames || nframes < 1 || nframes > kYuv2xMaxFrames) return GMAT_ERR(EINVAL);
    Yuv4x1Args a = a0;
    const int nstripsL = (a.dstW + D4_STRIP - 1) / D4_STRIP;
    const int nstripsC = a.nv12 ? (a.chrDstW + D4_STRIP / 2 - 1) / (D4_STRIP / 2) : (a.chrDstW + D4_STRIP - 1) / D4_STRIP;
    const int nplC = a.nv12 ? 1 : 2;
    a.nsgL = nstripsL; a.nsgC = nstripsC;
    const char *segStr = GMAT_KNOB("GMAT_STRIP_ROWS");              // tuning / test override: output rows per segment (every plane)
    int seg = segStr ? atoi(segStr) : 0;
    if (seg <= 0) {
        const long rows = ((long)a.dstH * nstripsL + (long)a.chrDstH * nstripsC * nplC) * nframes;      // wave-rows (output)
        // re-swept at the end of round 3 (32 frames a launch, rows 4 / 6 / 8 / 12 / 16 / 25 / 32 / 45: 4.20 / 3.86 / 3.83 / 4.07 / 3.88 / 4.11 / 4.51 / 5.28 us
        // per 4K -> 540p nv12 frame): short segments, as everywhere since the bands walk up and down
        seg = (int)std::min(8L, std::max(3L, (rows + 4095) / 4096));
    }
    a.segRows = seg;
    a.nsegL = (a.dstH + seg - 1) / seg;
    a.nsegC = (a.chrDstH + seg - 1) / seg;
    a.nblkL = (a.nsegL * a.nsgL + 3) / 4;
    a.nblk = a.nblkL + (a.nsegC * a.nsgC * nplC + 3) / 4;
    a.xcdRemap = 1;
    { const char *ud = GMAT_KNOB("GMAT_STRIP_UPDOWN"); a.updown = !(ud && !atoi(ud)); }
    const dim3 grid(8 * ((a.nblk + 7) / 8), nframes), block(256);
    if (a.nv12) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv4x1_kernel<true>), grid, block, 0, stream, a, *frames);
    else        hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv4x1_kernel<false>), grid, block, 0, stream, a, *frames);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
