// k_scale_yuv2p.hip — strip-walking form of the exact 2:1 YUV 4:2:0 -> YUV 4:2:0 scaler for gfx950: scale_cuda's main job
// (vf_scale_cuda.c:428-501: 4:2:0 in, 4:2:0 out at another size) at the transcode ratio 4K -> 1080p, with the arithmetic of
// ONE libswscale context (hScale8To15_c per plane, yuv2planeX_8_c / yuv2nv12cX_c vertically, swscale.c:234-520, output.c:
// 400-450), bit-exact.
//
// The design of k_scale_yuv2s.hip without a colour stage: every plane is scaled on its own, so a wave owns a strip of ONE
// plane — 256 output columns of a single-channel plane (Y, or U / V of YUV420P), or 128 output positions of NV12's
// interleaved UV plane — and walks down it with the horizontally filtered row pairs it still needs in a 4-deep register
// window.  Source bytes come straight from global memory (16-byte loads at 4-byte alignment), are widened onto the
// odd-aligned pair grid with v_perm_b32, and every coefficient is a kernel argument (the borders are the interior filter on
// an edge-replicated plane, checked on the host coefficient by coefficient: filter_is_edge_replication).  A workgroup is
// four strips of one plane; luma and chroma workgroups of all frames share one launch (blockIdx.x picks the plane).
// Parity: held to the oracle (tests/test_parity_planes2p.py, together with the tiled kernel on the same matrix).  None of
// the reference's own vectors is a 2:1 4:2:0 -> 4:2:0 scale (FATE filter-scale200 / -scale500 / -crop_scale and
// filter-pixfmts-scale are other ratios): they pin the oracle's code for this path, not this ratio.
//   bytes per output pixel: 4 luma + 2 chroma source bytes read, 1.5 written: 7.5 B per output pixel, 15.6 MB per 4K frame
//   (the tiled kernel it supersedes for this case, scale_yuv2x_kernel<yuv>, re-read a 7-row halo per 16-row tile).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

namespace gmat {

constexpr int P2_STRIP = 256;                  // output columns per wave of a single-channel plane: 64 lanes x 4
constexpr int P2_STRIP_UV = 128;               // output UV positions per wave of the interleaved plane: 64 lanes x 2

#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned p2_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned p2_u32x2 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ uint4 p2_ld16(const uint8_t *p) { const p2_u32x4 v = *reinterpret_cast<const p2_u32x4 *>(p); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint2 p2_ld8(const uint8_t *p) { const p2_u32x2 v = *reinterpret_cast<const p2_u32x2 *>(p); return make_uint2(v.x, v.y); }
#else
static inline uint4 p2_ld16(const uint8_t *p) { uint4 v; std::memcpy(&v, p, 16); return v; }
static inline uint2 p2_ld8(const uint8_t *p) { uint2 v; std::memcpy(&v, p, 8); return v; }
#endif

// three-operand v_dot2_i32_i16 (clamp bit set: no tied accumulator, see k_scale_yuv2s.hip)
__device__ __forceinline__ int p2_dot2(int packed_ab, int packed_cd, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, packed_ab), __builtin_bit_cast(short2v, packed_cd), acc, true);
}
__device__ __forceinline__ int p2_pair12(unsigned lo) { return (int)__builtin_amdgcn_perm(0u, lo, 0x0C020C01u); }
__device__ __forceinline__ int p2_pair30(unsigned hi, unsigned lo) { return (int)__builtin_amdgcn_perm(hi, lo, 0x0C040C03u); }
__device__ __forceinline__ unsigned p2_rep(unsigned v, unsigned sel) { return __builtin_amdgcn_perm(v, v, sel); }

// ---- the samples --------------------------------------------------------------------------------------------------
// 8-bit: hScale8To15_c (>> 7).  10 bits in 16-bit containers: p010LEToY_c / p010LEToUV_c (P010: sample >> 6, input.c:698-725;
// planar YUV420P10LE: as they are), hScale16To15_c with sh = 9 (swscale.c:93-119).  Destinations: yuv2planeX_8_c /
// yuv2nv12cX_c: clip_u8((64 << 12 + sum) >> 19); yuv2planeX_10_c / yuv2p010lX_c / cX_c: clip10((1 << 16 + sum) >> 17), P010
// << 6 (output.c:330-519).  The horizontal stage depends on the source depth only, the vertical one on the destination's:
// the walkers are templates <S16, D16> and serve 8 -> 8, 10 -> 10, 8 -> 10 and 10 -> 8.
__device__ __forceinline__ unsigned p2_shr6(unsigned v) { return (v >> 6) & 0x03FF03FFu; }        // both halves: v_pk_lshrrev_b16
__device__ __forceinline__ int p2_odd(unsigned hi, unsigned lo) { return (int)((lo >> 16) | (hi << 16)); }   // v_alignbit_b32

#ifndef P2_UVD_STREAM
#define P2_UVD_STREAM false      // its 248-byte strips are not whole lines: plain stores 0.8 % ahead (3.208 -> 3.183 us per nv12 frame, three alternating rounds)
#endif
struct P2Plane {
    const uint8_t *src; uint8_t *dst;
    int ss, ds, srcW, srcH, dstW;              // strides in bytes, widths in samples (UV plane: in UV positions)
    const int32_t *h, *v;                      // NP int16 pairs each on the odd-aligned window [2x - (NP - 1), 2x + NP]
    int rnd;                                   // vertical accumulator start (8-bit: the dither term << 12; 10-bit: 1 << 16)
    int srcHi6, dstHi6;                        // P010: the 10 significant bits are the high ones (>> 6 in, << 6 out)
    // up: the segment is walked from its LAST output row to its first (odd segments: the halo rows two neighbouring segments share
    // are then requested at the same time, one HBM read and one L2 hit — the headline's finding).  The walker
    // itself never knows: it walks the vertically mirrored plane downward — row r stands for row srcH - 1 - r (dstH - 1 - r), the
    // row pair (2m - 1, 2m) is the mirrored pair H/2 - m with its halves swapped, so `v` holds the pairs reversed and swapped.
    int dstH, up;
    int dither;                                // 8-bit out of a deeper source (S16 && !D16): 0 the constant in rnd, 1 this plane's own columns (luma, U),
                                               // 2 three columns on (V), 3 interleaved U0 V0 U1 V1 (U: its column, V: three on) — dither_8x8_128
    __device__ __forceinline__ int srow(int r) const { return up ? srcH - 1 - r : r; }
    __device__ __forceinline__ int drow(int r) const { return up ? dstH - 1 - r : r; }
};
// value q of a lane's four (single plane: samples x + q; interleaved: U(x), V(x), U(x + 1), V(x + 1)) on destination row y
__device__ __forceinline__ int p2_dither(int mode, int x, int y, int q)
{
    const int col = mode == 3 ? x + (q >> 1) + 3 * (q & 1) : x + q + (mode == 2 ? 3 : 0);
    return dither_delta(col, y);
}

__device__ __forceinline__ unsigned p2_ld4(const uint8_t *p) { return *reinterpret_cast<const unsigned *>(p); }

// vertical stage of 4 values + store: D16 ? 8 bytes (4 x 16 bit) : 4 bytes.  Slot SLOT holds the newest row pair, the
// oldest is SLOT + 1 (mod NP).
template <bool D16, int NP, int SLOT, bool STREAM = true, bool DITH = false>
__device__ __forceinline__ void p2_vstore(const P2Plane &P, const int (&hw)[NP][4], bool active, unsigned byteOff, int x = 0, int y = 0)
{
    unsigned w[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int acc = P.rnd;
        if constexpr (DITH) { if (P.dither) acc += p2_dither(P.dither, x, y, q); }
#pragma unroll
        for (int k = 0; k < NP; k++) acc = p2_dot2(hw[(SLOT + 1 + k) % NP][q], P.v[k], acc);
        if (D16) {
            w[q] = (unsigned)min(max(acc, 0), (1024 << 17) - 1) >> 17;               // clamp, then shift (see clip_u8_shr)
            if (P.dstHi6) w[q] <<= 6;
        } else {
            w[q] = (unsigned)clip_u8_shr(acc, 19);
        }
    }
    if (!active) return;
    if (!STREAM) {                                              // a walker whose strips are not whole lines (the shared-load UV walker: 248 bytes)
        if (D16) *reinterpret_cast<uint2 *>(P.dst + byteOff) = make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16));
        else     *reinterpret_cast<unsigned *>(P.dst + byteOff) = w[0] | (w[1] << 8) | (w[2] << 16) | (w[3] << 24);
        return;
    }
    if (D16) st_stream(P.dst + byteOff, make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16)));
    else     st_stream(P.dst + byteOff, (unsigned)(w[0] | (w[1] << 8) | (w[2] << 16) | (w[3] << 24)));
}

// the row loop, unrolled by NP so that the slot of an iteration is a compile-time constant
template <int NP, typename Body, typename Edge>
__device__ __forceinline__ void p2_rows(int nIter, Body &&body, Edge edge_c)
{
    for (int j0 = 0; j0 < nIter; j0 += NP) {
        body(j0, std::integral_constant<int, 0>(), edge_c);
        if (j0 + 1 < nIter) body(j0 + 1, std::integral_constant<int, 1>(), edge_c);
        if (j0 + 2 < nIter) body(j0 + 2, std::integral_constant<int, 2>(), edge_c);
        if (j0 + 3 < nIter) body(j0 + 3, std::integral_constant<int, 3>(), edge_c);
        if constexpr (NP > 4) {
            if (j0 + 4 < nIter) body(j0 + 4, std::integral_constant<int, 4>(), edge_c);
            if (j0 + 5 < nIter) body(j0 + 5, std::integral_constant<int, 5>(), edge_c);
        }
    }
}

// ---- one single-channel plane: srcW x srcH -> dstW x dstH (exactly half), rows [y0, y0 + nOut) of the strip at X0 ----
// NP = 4: 16 samples from sample 2xc - 4 (8-bit: 4 dwords, 16-bit: 8).  NP = 6 (Lanczos-3, 12 taps): 8-bit 24 samples from
// 2xc - 8 (6 dwords), 16-bit 20 samples from 2xc - 6 (10 dwords).
struct P2Row { unsigned d[10]; };

template <bool S16, bool D16, int NP>
__device__ __forceinline__ void p2_walk_plane(const P2Plane &P, int X0, int y0, int nOut, int lane)
{
    constexpr int BL = S16 ? (NP == 4 ? 4 : 6) : (NP == 4 ? 4 : 8);      // samples between the window base and 2xc
    constexpr int ND = S16 ? (NP == 4 ? 8 : 10) : (NP == 4 ? 4 : 6);     // dwords of the window
    constexpr int SPD = S16 ? 2 : 4;                                     // samples per dword
    constexpr int SHR = S16 ? 9 : 7;
    const int xo = X0 + 4 * lane;
    const bool active = xo < P.dstW;
    const int xc = active ? xo : P.dstW - 4;                    // idle lanes shadow the last group (loads stay inside the rows)
    const int want = 2 * xc - BL;                               // first sample of the window
    const int off = min(max(want, 0), P.srcW - ND * SPD);       // NP = 4: the whole window clamped, then shifted back in registers
    const int sh = want - off;                                  //         -4 at the left plane edge, +4 at the right one
    const unsigned uoff = (S16 ? 2u : 1u) * (unsigned)off;
    const int wd0 = want / SPD - (want < 0 ? 1 : 0) * ((-want % SPD) != 0);   // dword index of the window base (want is a multiple of SPD)
    const int lastDw = P.srcW / SPD - 1;
    const int nIter = nOut + NP - 1;                            // NP - 1 warm-up row pairs fill the vertical window
    const int m0 = y0 - (NP / 2 - 1);                           // row pair of iteration 0 (pair m = rows 2m - 1, 2m)
    const bool hi6 = P.srcHi6 != 0;

    auto load1 = [&](int row, P2Row &r, auto edge_c) {
        const unsigned o = (unsigned)P.srow(min(max(row, 0), P.srcH - 1)) * (unsigned)P.ss;
        if (NP == 4 || !decltype(edge_c)::value) {
            const unsigned b = o + (NP == 4 ? uoff : (S16 ? 2u : 1u) * (unsigned)want);
            const uint4 t0 = p2_ld16(P.src + b);
            r.d[0] = t0.x; r.d[1] = t0.y; r.d[2] = t0.z; r.d[3] = t0.w;
            if (ND > 4) {
                if (ND >= 8) { const uint4 t1 = p2_ld16(P.src + (unsigned)(b + 16u)); r.d[4] = t1.x; r.d[5] = t1.y; r.d[6] = t1.z; r.d[7] = t1.w; }
                if (ND == 6) { const uint2 t1 = p2_ld8(P.src + (unsigned)(b + 16u)); r.d[4] = t1.x; r.d[5] = t1.y; }
                if (ND == 10) { const uint2 t2 = p2_ld8(P.src + (unsigned)(b + 32u)); r.d[8] = t2.x; r.d[9] = t2.y; }
            }
        } else {
            // NP = 6 in a wave that touches a plane edge: two lanes a side overlap the edge by different amounts, so every
            // dword comes from its own clamped address (the edge sample is replicated in hrow)
#pragma unroll
            for (int i = 0; i < ND; i++) r.d[i] = p2_ld4(P.src + (unsigned)(o + 4u * (unsigned)min(max(wd0 + i, 0), lastDw)));
        }
    };
    auto load = [&](int m, P2Row &ra, P2Row &rb, auto edge_c) { load1(2 * m - 1, ra, edge_c); load1(2 * m, rb, edge_c); };
    // horizontal filter of one row: 4 outputs from NP + 3 odd-aligned pairs
    // EK: 0 interior wave, 1 the plane's left edge, 2 its right edge, 3 both.  The fix-ups of the edge lanes are SELECTS on lane masks,
    // one per dword when the wave touches one edge only: as `if (sh < 0) ...` on per-lane values they were divergent branches, and the
    // edge strips' waves ran 22-26 % longer per row than the others (profiles/r02q_headline_wave_durations.txt)
    const bool edgeL = sh < 0, edgeR = sh > 0;
    auto hrow = [&](const P2Row &R, auto edge_c, int (&s)[4]) {
        constexpr int EK = decltype(edge_c)::value;
        constexpr bool EDGE = EK != 0;
        unsigned d[ND];
#pragma unroll
        for (int k = 0; k < ND; k++) d[k] = R.d[k];
        if (S16 && hi6) {
#pragma unroll
            for (int k = 0; k < ND; k++) d[k] = p2_shr6(d[k]);
        }
        if constexpr (EDGE && NP == 4) {
            // the window shifted by SD dwords (4 samples) towards the plane, the edge sample replicated into the dwords that left it
            constexpr int SD = S16 ? 2 : 1;
            unsigned o[ND];
#pragma unroll
            for (int k = 0; k < ND; k++) o[k] = d[k];
            const unsigned first = p2_rep(o[0], S16 ? 0x01000100u : 0x00000000u), last = p2_rep(o[ND - 1], S16 ? 0x03020302u : 0x03030303u);
#pragma unroll
            for (int k = 0; k < ND; k++) {
                const unsigned fromLeft = k < SD ? first : o[k - SD < 0 ? 0 : k - SD], fromRight = k >= ND - SD ? last : o[k + SD >= ND ? ND - 1 : k + SD];
                if constexpr (EK == 1)      d[k] = edgeL ? fromLeft : o[k];
                else if constexpr (EK == 2) d[k] = edgeR ? fromRight : o[k];
                else                        d[k] = edgeL ? fromLeft : edgeR ? fromRight : o[k];
            }
        }
        if constexpr (EDGE && NP != 4) {
#pragma unroll
            for (int i = 0; i < ND; i++) {
                const int idx = wd0 + i;                        // the dword this one stands for; outside the row: the edge sample
                const unsigned lo = S16 ? p2_rep(d[i], 0x01000100u) : p2_rep(d[i], 0x00000000u);
                const unsigned hi = S16 ? p2_rep(d[i], 0x03020302u) : p2_rep(d[i], 0x03030303u);
                d[i] = idx < 0 ? lo : idx > lastDw ? hi : d[i];
            }
        }
        int p[NP + 3];
        if (S16) {
#pragma unroll
            for (int k = 0; k < NP + 3; k++) p[k] = p2_odd(d[k + 1], d[k]);      // samples (base + 1 + 2k, + 2 + 2k)
        } else {
            constexpr int O0 = BL - (NP - 1);                   // byte of the first pair: 1 (NP = 4) or 3 (NP = 6)
#pragma unroll
            for (int k = 0; k < NP + 3; k++) {
                const int o = O0 + 2 * k;
                p[k] = (o & 3) == 1 ? p2_pair12(d[o >> 2]) : p2_pair30(d[(o >> 2) + 1], d[o >> 2]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int acc = 0;
#pragma unroll
            for (int k = 0; k < NP; k++) acc = p2_dot2(p[j + k], P.h[k], acc);
            s[j] = acc;
        }
    };

    int hw[NP][4];                                              // [slot][output]: (row 2m-1 | row 2m << 16), 15-bit lines
#pragma unroll
    for (int s = 0; s < NP; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) hw[s][j] = 0;
    P2Row bufA[2], bufB[2];                                     // ping-pong: iteration j consumes [j & 1], prefetches the other
#pragma unroll
    for (int i = 0; i < 10; i++) bufA[0].d[i] = bufB[0].d[i] = bufA[1].d[i] = bufB[1].d[i] = 0u;

    auto body = [&](const int j, auto slot_c, auto edge_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        // with NP even the parity of j equals the parity of SLOT
        if (j + 1 < nIter) load(m0 + j + 1, bufA[(SLOT + 1) & 1], bufB[(SLOT + 1) & 1], edge_c);
        {
            int sa[4], sb[4];
            hrow(bufA[SLOT & 1], edge_c, sa);
            hrow(bufB[SLOT & 1], edge_c, sb);
#pragma unroll
            for (int q = 0; q < 4; q++)                        // hScale8To15_c / hScale16To15_c: min(val >> 7 | 9, 32767)
                hw[SLOT][q] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(sa[q] >> SHR, sb[q] >> SHR));
        }
        if (j >= NP - 1) {
            const int yo = y0 + j - (NP - 1);
            p2_vstore<D16, NP, SLOT, true, S16 && !D16>(P, hw, active, (unsigned)((unsigned)P.drow(yo) * (unsigned)P.ds + (D16 ? 2u : 1u) * (unsigned)xo), xo, P.drow(yo));
        }
    };
    {
        const int ek = (X0 == 0 ? 1 : 0) | (X0 + P2_STRIP + (NP == 4 ? 0 : 8) >= P.dstW ? 2 : 0);       // wave-uniform
        auto go = [&](auto ec) { load(m0, bufA[0], bufB[0], ec); p2_rows<NP>(nIter, body, ec); };
        if (ek == 0) go(std::integral_constant<int, 0>());
        else if (ek == 1) go(std::integral_constant<int, 1>());
        else if (ek == 2) go(std::integral_constant<int, 2>());
        else go(std::integral_constant<int, 3>());
    }
}

// ---- the interleaved UV plane (NV12: 2 bytes a position; P010: one dword): a lane makes 2 UV outputs a row ------------
// window: positions [2cc - NP, 2cc + NP + 4): 8-bit NP + 2 dwords (6 / 8), 16-bit 2 NP + 4 dwords (12 / 16)
struct P2RowUV { unsigned d[16]; };

template <bool S16, bool D16, int NP>
__device__ __forceinline__ void p2_walk_uv(const P2Plane &P, int X0, int y0, int nOut, int lane)
{
    constexpr int ND = S16 ? 2 * NP + 4 : NP + 2;
    constexpr int SHR = S16 ? 9 : 7;
    const int co = X0 + 2 * lane;
    const bool active = co < P.dstW;
    const int cc = active ? co : P.dstW - 2;
    const int want = 2 * cc - NP;                               // first position of the window (even)
    // NP = 4, 8-bit: two loads with their own clamps (bytes); NP = 4, 16-bit: one 48-byte window clamped as a whole
    const int offA = max(4 * cc - 8, 0), shA = 4 * cc - 8 - offA;               // -8 bytes: left edge
    const int offB = min(4 * cc + 8, 2 * P.srcW - 8), shB = 4 * cc + 8 - offB;  // +8 bytes: right edge
    const int off16 = min(max(want, 0), P.srcW - 12);
    const int sh16 = want - off16;                              // -4 / +4 positions
    const int wd0 = S16 ? want : want / 2;                      // dword index of the window base (want even: exact, also when negative)
    const int lastDw = (S16 ? P.srcW : P.srcW / 2) - 1;
    const int nIter = nOut + NP - 1;
    const int m0 = y0 - (NP / 2 - 1);

    auto load1 = [&](int row, P2RowUV &r, auto edge_c) {
        const unsigned o = (unsigned)P.srow(min(max(row, 0), P.srcH - 1)) * (unsigned)P.ss;
        if (NP == 4) {
            if (S16) {
                const unsigned b = o + 4u * (unsigned)off16;
                const uint4 t0 = p2_ld16(P.src + b), t1 = p2_ld16(P.src + (unsigned)(b + 16u)), t2 = p2_ld16(P.src + (unsigned)(b + 32u));
                r.d[0] = t0.x; r.d[1] = t0.y; r.d[2] = t0.z; r.d[3] = t0.w; r.d[4] = t1.x; r.d[5] = t1.y; r.d[6] = t1.z; r.d[7] = t1.w;
                r.d[8] = t2.x; r.d[9] = t2.y; r.d[10] = t2.z; r.d[11] = t2.w;
            } else {
                const uint4 t0 = p2_ld16(P.src + (unsigned)(o + (unsigned)offA));
                const uint2 t1 = p2_ld8(P.src + (unsigned)(o + (unsigned)offB));
                r.d[0] = t0.x; r.d[1] = t0.y; r.d[2] = t0.z; r.d[3] = t0.w; r.d[4] = t1.x; r.d[5] = t1.y;
            }
        } else if (!decltype(edge_c)::value) {
            const unsigned b = o + 4u * (unsigned)wd0;
#pragma unroll
            for (int i = 0; i < ND / 4; i++) {
                const uint4 t = p2_ld16(P.src + (unsigned)(b + 16u * i));
                r.d[4 * i] = t.x; r.d[4 * i + 1] = t.y; r.d[4 * i + 2] = t.z; r.d[4 * i + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < ND; i++) r.d[i] = p2_ld4(P.src + (unsigned)(o + 4u * (unsigned)min(max(wd0 + i, 0), lastDw)));
        }
    };
    auto load = [&](int m, P2RowUV &ra, P2RowUV &rb, auto edge_c) { load1(2 * m - 1, ra, edge_c); load1(2 * m, rb, edge_c); };
    // horizontal filter of one row: 2 U and 2 V outputs from NP + 1 odd-aligned pairs per channel
    const bool edge16L = sh16 < 0, edge16R = sh16 > 0, edgeAL = shA < 0, edgeBR = shB > 0;      // lane masks, see p2_walk_plane
    auto hrow = [&](const P2RowUV &R, auto edge_c, int (&su)[2], int (&sv)[2]) {
        constexpr int EK = decltype(edge_c)::value;
        constexpr bool EDGE = EK != 0;
        unsigned e[ND];
#pragma unroll
        for (int k = 0; k < ND; k++) e[k] = R.d[k];
        if (S16) {
#pragma unroll
            for (int k = 0; k < ND; k++) e[k] = p2_shr6(e[k]);  // p010LEToUV_c: both samples of the position >> 6
        }
        if constexpr (EDGE && NP == 4) {
            if constexpr (S16) {
                unsigned o[12];                                     // 12 positions (one dword each) shifted by 4, the edge position replicated
#pragma unroll
                for (int k = 0; k < 12; k++) o[k] = e[k];
#pragma unroll
                for (int k = 0; k < 12; k++) {
                    const unsigned fromLeft = k < 4 ? o[0] : o[k - 4 < 0 ? 0 : k - 4], fromRight = k >= 8 ? o[11] : o[k + 4 > 11 ? 11 : k + 4];
                    if constexpr (EK == 1)      e[k] = edge16L ? fromLeft : o[k];
                    else if constexpr (EK == 2) e[k] = edge16R ? fromRight : o[k];
                    else                        e[k] = edge16L ? fromLeft : edge16R ? fromRight : o[k];
                }
            } else {
                if constexpr ((EK & 1) != 0) {
                    const unsigned r = p2_rep(e[0], 0x01000100u), e0 = e[0], e1 = e[1];
                    e[0] = edgeAL ? r : e0; e[1] = edgeAL ? r : e1; e[2] = edgeAL ? e0 : e[2]; e[3] = edgeAL ? e1 : e[3];
                }
                if constexpr ((EK & 2) != 0) {
                    const unsigned rr = p2_rep(e[5], 0x03020302u);
                    e[4] = edgeBR ? rr : e[4]; e[5] = edgeBR ? rr : e[5];
                }
            }
        }
        if constexpr (EDGE && NP != 4) {
#pragma unroll
            for (int i = 0; i < ND; i++) {
                const int idx = wd0 + i;
                const unsigned lo = S16 ? e[i] : p2_rep(e[i], 0x01000100u);      // a 16-bit position is the whole dword
                const unsigned hi = S16 ? e[i] : p2_rep(e[i], 0x03020302u);
                e[i] = idx < 0 ? lo : idx > lastDw ? hi : e[i];
            }
        }
        int pU[NP + 1], pV[NP + 1];
#pragma unroll
        for (int k = 0; k < NP + 1; k++) {
            if (S16) {                          // positions (base + 1 + 2k, + 2 + 2k) = e[2k+1], e[2k+2]
                pU[k] = (int)__builtin_amdgcn_perm(e[2 * k + 2], e[2 * k + 1], 0x05040100u);
                pV[k] = (int)__builtin_amdgcn_perm(e[2 * k + 2], e[2 * k + 1], 0x07060302u);
            } else {                            // bytes 2,3 of e[k] and 0,1 of e[k+1]
                pU[k] = (int)__builtin_amdgcn_perm(e[k + 1], e[k], 0x0C040C02u);
                pV[k] = (int)__builtin_amdgcn_perm(e[k + 1], e[k], 0x0C050C03u);
            }
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            int au = 0, av = 0;
#pragma unroll
            for (int k = 0; k < NP; k++) { au = p2_dot2(pU[c + k], P.h[k], au); av = p2_dot2(pV[c + k], P.h[k], av); }
            su[c] = au; sv[c] = av;
        }
    };

    int hw[NP][4];                                              // [slot][U0, V0, U1, V1]: (row 2m-1 | row 2m << 16)
#pragma unroll
    for (int s = 0; s < NP; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) hw[s][j] = 0;
    P2RowUV bufA[2], bufB[2];
#pragma unroll
    for (int i = 0; i < 16; i++) bufA[0].d[i] = bufB[0].d[i] = bufA[1].d[i] = bufB[1].d[i] = 0u;

    auto body = [&](const int j, auto slot_c, auto edge_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        if (j + 1 < nIter) load(m0 + j + 1, bufA[(SLOT + 1) & 1], bufB[(SLOT + 1) & 1], edge_c);
        {
            int ua[2], va[2], ub[2], vb[2];
            hrow(bufA[SLOT & 1], edge_c, ua, va);
            hrow(bufB[SLOT & 1], edge_c, ub, vb);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                hw[SLOT][2 * c + 0] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(ua[c] >> SHR, ub[c] >> SHR));
                hw[SLOT][2 * c + 1] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(va[c] >> SHR, vb[c] >> SHR));
            }
        }
        if (j >= NP - 1) {
            const int yo = y0 + j - (NP - 1);                   // yuv2nv12cX_c / yuv2p010cX_c: U0 V0 U1 V1
            p2_vstore<D16, NP, SLOT, true, S16 && !D16>(P, hw, active, (unsigned)((unsigned)P.drow(yo) * (unsigned)P.ds + (D16 ? 4u : 2u) * (unsigned)co), co, P.drow(yo));
        }
    };
    {
        const int ek = (X0 == 0 ? 1 : 0) | (X0 + P2_STRIP_UV + (NP == 4 ? 0 : 8) >= P.dstW ? 2 : 0);    // wave-uniform
        auto go = [&](auto ec) { load(m0, bufA[0], bufB[0], ec); p2_rows<NP>(nIter, body, ec); };
        if (ek == 0) go(std::integral_constant<int, 0>());
        else if (ek == 1) go(std::integral_constant<int, 1>());
        else if (ek == 2) go(std::integral_constant<int, 2>());
        else go(std::integral_constant<int, 3>());
    }
}

// ---- the interleaved 8-bit UV plane with SHARED loads (8 taps): a lane loads only its own 4 source positions (8 bytes, one load per row)
// and takes the two dwords either side of them from the lanes beside it by DPP wave shifts.  p2_walk_uv's windows overlap threefold
// between lanes and cost two loads per row; a wave-level load occupies the CU's address path ~17 cycles whatever its width
// (tools/ubench/load_rate.hip), which put that path at ~85 % for the chroma waves of an NV12 transcode.  Lanes 0 and 63 only provide:
// a wave makes 62 x 2 = 124 UV outputs.  A lane's 4 positions are entirely inside or entirely outside the plane (its width is even),
// so edge replication is "a lane outside presents the edge position" (as in scale_rgb2h_kernel).
constexpr int P2_STRIP_UVD = 124;

template <bool D16>                             // instantiated for 8-bit destinations only (see the kernel)
__device__ __forceinline__ void p2_walk_uvd(const P2Plane &P, int X0, int y0, int nOut, int lane)
{
    constexpr int NP = 4, SHR = 7;
    const int co = X0 + 2 * (lane - 1);                         // the lane's 2 UV outputs; lanes 0 / 63: their neighbours' halo
    const bool stores = lane >= 1 && lane <= 62 && co < P.dstW;
    const bool outL = co < 0, outR = co >= P.dstW;              // own positions 2 co .. 2 co + 3 lie outside the plane
    const bool edgeWave = X0 == 0 || X0 + P2_STRIP_UVD >= P.dstW;   // wave-uniform: the wave holds an outside lane
    const unsigned ob = 4u * (unsigned)min(max(co, 0), P.dstW - 2);     // outside lanes load the plane's first / last 4 positions
    const int nIter = nOut + NP - 1;
    const int m0 = y0 - (NP / 2 - 1);

    auto load = [&](int m, uint2 &ra, uint2 &rb) {
        ra = p2_ld8(P.src + ((unsigned)P.srow(min(max(2 * m - 1, 0), P.srcH - 1)) * (unsigned)P.ss + ob));
        rb = p2_ld8(P.src + ((unsigned)P.srow(min(max(2 * m, 0), P.srcH - 1)) * (unsigned)P.ss + ob));
    };
    auto hrow = [&](const uint2 &R, auto edge_c, int (&su)[2], int (&sv)[2]) {
        unsigned d0 = R.x, d1 = R.y;
        if constexpr (decltype(edge_c)::value) {
            const unsigned first = p2_rep(d0, 0x01000100u), last = p2_rep(d1, 0x03020302u);
            d0 = outL ? first : outR ? last : d0; d1 = outL ? first : outR ? last : d1;
        }
        unsigned e[6];                                          // positions 2 co - 4 .. 2 co + 7, two per dword
        e[0] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)d0, 0x138, 0xF, 0xF, true);      // lane - 1
        e[1] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)d1, 0x138, 0xF, 0xF, true);
        e[2] = d0; e[3] = d1;
        e[4] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)d0, 0x130, 0xF, 0xF, true);      // lane + 1
        e[5] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)d1, 0x130, 0xF, 0xF, true);
        int pU[NP + 1], pV[NP + 1];
#pragma unroll
        for (int k = 0; k < NP + 1; k++) {                      // bytes 2,3 of e[k] and 0,1 of e[k+1]
            pU[k] = (int)__builtin_amdgcn_perm(e[k + 1], e[k], 0x0C040C02u);
            pV[k] = (int)__builtin_amdgcn_perm(e[k + 1], e[k], 0x0C050C03u);
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            int au = 0, av = 0;
#pragma unroll
            for (int k = 0; k < NP; k++) { au = p2_dot2(pU[c + k], P.h[k], au); av = p2_dot2(pV[c + k], P.h[k], av); }
            su[c] = au; sv[c] = av;
        }
    };

    int hw[NP][4];                                              // [slot][U0, V0, U1, V1]: (row 2m-1 | row 2m << 16)
#pragma unroll
    for (int s = 0; s < NP; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) hw[s][j] = 0;
    uint2 bufA[2], bufB[2];
    bufA[1] = bufB[1] = make_uint2(0u, 0u);
    load(m0, bufA[0], bufB[0]);

    auto body = [&](const int j, auto slot_c, auto edge_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        if (j + 1 < nIter) load(m0 + j + 1, bufA[(SLOT + 1) & 1], bufB[(SLOT + 1) & 1]);
        {
            int ua[2], va[2], ub[2], vb[2];
            hrow(bufA[SLOT & 1], edge_c, ua, va);
            hrow(bufB[SLOT & 1], edge_c, ub, vb);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                hw[SLOT][2 * c + 0] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(ua[c] >> SHR, ub[c] >> SHR));
                hw[SLOT][2 * c + 1] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(va[c] >> SHR, vb[c] >> SHR));
            }
        }
        if (j >= NP - 1) {
            const int yo = y0 + j - (NP - 1);                   // yuv2nv12cX_c / yuv2p010cX_c: U0 V0 U1 V1
            p2_vstore<D16, NP, SLOT, P2_UVD_STREAM>(P, hw, stores, (unsigned)((unsigned)P.drow(yo) * (unsigned)P.ds + (D16 ? 4u : 2u) * (unsigned)co));
        }
    };
    if (edgeWave) p2_rows<NP>(nIter, body, std::true_type());
    else          p2_rows<NP>(nIter, body, std::false_type());
}

// ---- chroma across layouts: planar U, V planes <-> one interleaved UV plane (NV12 -> YUV420P: a hardware decoder's frames into a
// software encoder; YUV420P -> NV12: the reverse).  The UV walker with the other side's load or store stage: a lane still makes 2
// UV outputs a row and keeps [U0, V0, U1, V1] in its window slots.
//   SPL: the source is planar — per plane 2 NP + 2 samples from sample 2cc - 4 (NP = 4) / 2cc - 8 | 2cc - 6 (NP = 6; 8 | 16 bit)
//   !SPL: the source is interleaved, the destination planar: 2 bytes (16 bit: a dword) to each plane
struct P2Cross {
    const uint8_t *srcU, *srcV; uint8_t *dstU, *dstV;          // interleaved side: ...U is the UV plane, ...V unused
    int ssU, ssV, dsU, dsV, srcW, srcH, dstW;                   // widths in UV positions
    const int32_t *h, *v;
    int rnd, srcHi6, dstHi6;
    int dstH, up;                                               // see P2Plane
    int dither;                                                 // != 0: U0 V0 U1 V1 dithered (P2Plane::dither mode 3)
    __device__ __forceinline__ int srow(int r) const { return up ? srcH - 1 - r : r; }
    __device__ __forceinline__ int drow(int r) const { return up ? dstH - 1 - r : r; }
};
struct P2RowX { unsigned u[16], v[8]; };                        // interleaved source: u = the UV dwords

template <bool S16, bool D16, int NP, bool SPL>
__device__ __forceinline__ void p2_walk_uvx(const P2Cross &P, int X0, int y0, int nOut, int lane)
{
    constexpr int SHR = S16 ? 9 : 7;
    // planar source: dwords per plane and samples between the window base and 2cc
    constexpr int BLP = S16 ? (NP == 4 ? 4 : 6) : (NP == 4 ? 4 : 8);
    constexpr int NDP = S16 ? (NP == 4 ? 6 : 8) : (NP == 4 ? 3 : 5);
    constexpr int SPD = S16 ? 2 : 4;
    // interleaved source: NP + 2 dwords (16 bit: 2 NP + 4), window base position 2cc - NP
    constexpr int NDI = S16 ? 2 * NP + 4 : NP + 2;
    const int co = X0 + 2 * lane;
    const bool active = co < P.dstW;
    const int cc = active ? co : P.dstW - 2;
    const int want = SPL ? 2 * cc - BLP : 2 * cc - NP;          // first sample (planar) / position (interleaved) of the window
    const int wd0 = SPL ? (want >= 0 ? want / SPD : -((-want + SPD - 1) / SPD)) : (S16 ? want : (want >= 0 ? want / 2 : -((-want + 1) / 2)));
    const int lastDw = SPL ? P.srcW / SPD - 1 : (S16 ? P.srcW : P.srcW / 2) - 1;
    // a wave is an edge wave when any of its windows leaves the row: every dword then comes from its own clamped address
    const bool edgeWave = X0 == 0 || X0 + P2_STRIP_UV + 8 >= P.dstW;
    const int nIter = nOut + NP - 1;
    const int m0 = y0 - (NP / 2 - 1);

    auto load1 = [&](int row, P2RowX &r, auto edge_c) {
        const int rr = P.srow(min(max(row, 0), P.srcH - 1));
        const unsigned ou = (unsigned)rr * (unsigned)P.ssU, ov = (unsigned)rr * (unsigned)P.ssV;
        if constexpr (decltype(edge_c)::value) {
#pragma unroll
            for (int i = 0; i < (SPL ? NDP : NDI); i++) {
                const unsigned c = 4u * (unsigned)min(max(wd0 + i, 0), lastDw);
                r.u[i] = p2_ld4(P.srcU + (unsigned)(ou + c));
                if (SPL) r.v[i] = p2_ld4(P.srcV + (unsigned)(ov + c));
            }
        } else if constexpr (SPL) {
            const unsigned b = 4u * (unsigned)wd0;
            const uint4 tu = p2_ld16(P.srcU + (unsigned)(ou + b)), tv = p2_ld16(P.srcV + (unsigned)(ov + b));   // NDP = 3: the 4th dword is not used
            r.u[0] = tu.x; r.u[1] = tu.y; r.u[2] = tu.z; r.u[3] = tu.w; r.v[0] = tv.x; r.v[1] = tv.y; r.v[2] = tv.z; r.v[3] = tv.w;
            if (NDP > 4) {
                const uint4 tu2 = p2_ld16(P.srcU + (unsigned)(ou + b + 16u)), tv2 = p2_ld16(P.srcV + (unsigned)(ov + b + 16u));
                r.u[4] = tu2.x; r.u[5] = tu2.y; r.u[6] = tu2.z; r.u[7] = tu2.w; r.v[4] = tv2.x; r.v[5] = tv2.y; r.v[6] = tv2.z; r.v[7] = tv2.w;
            }
        } else {
            const unsigned b = ou + 4u * (unsigned)wd0;
#pragma unroll
            for (int i = 0; i < NDI / 4; i++) {
                const uint4 t = p2_ld16(P.srcU + (unsigned)(b + 16u * i));
                r.u[4 * i] = t.x; r.u[4 * i + 1] = t.y; r.u[4 * i + 2] = t.z; r.u[4 * i + 3] = t.w;
            }
            if (NDI % 4) { const uint2 t = p2_ld8(P.srcU + (unsigned)(b + 16u * (NDI / 4))); r.u[NDI - 2] = t.x; r.u[NDI - 1] = t.y; }
        }
    };
    auto load = [&](int m, P2RowX &ra, P2RowX &rb, auto edge_c) { load1(2 * m - 1, ra, edge_c); load1(2 * m, rb, edge_c); };
    auto hrow = [&](const P2RowX &R, auto edge_c, int (&su)[2], int (&sv)[2]) {
        constexpr bool EDGE = decltype(edge_c)::value;
        int pU[NP + 1], pV[NP + 1];
        if constexpr (SPL) {
            unsigned du[NDP + 1], dv[NDP + 1];
#pragma unroll
            for (int k = 0; k < NDP; k++) { du[k] = R.u[k]; dv[k] = R.v[k]; }
            du[NDP] = dv[NDP] = 0u;
            if constexpr (EDGE) {
#pragma unroll
                for (int i = 0; i < NDP; i++) {
                    const int idx = wd0 + i;
                    const unsigned selLo = S16 ? 0x01000100u : 0x00000000u, selHi = S16 ? 0x03020302u : 0x03030303u;
                    du[i] = idx < 0 ? p2_rep(du[i], selLo) : idx > lastDw ? p2_rep(du[i], selHi) : du[i];
                    dv[i] = idx < 0 ? p2_rep(dv[i], selLo) : idx > lastDw ? p2_rep(dv[i], selHi) : dv[i];
                }
            }
            if constexpr (S16) {
#pragma unroll
                for (int k = 0; k < NP + 1; k++) { pU[k] = p2_odd(du[k + 1], du[k]); pV[k] = p2_odd(dv[k + 1], dv[k]); }
            } else {
                constexpr int O0 = BLP - (NP - 1);              // byte of the first pair: 1 (NP = 4) or 3 (NP = 6)
#pragma unroll
                for (int k = 0; k < NP + 1; k++) {
                    const int o = O0 + 2 * k;
                    pU[k] = (o & 3) == 1 ? p2_pair12(du[o >> 2]) : p2_pair30(du[(o >> 2) + 1], du[o >> 2]);
                    pV[k] = (o & 3) == 1 ? p2_pair12(dv[o >> 2]) : p2_pair30(dv[(o >> 2) + 1], dv[o >> 2]);
                }
            }
        } else {
            unsigned e[NDI];
#pragma unroll
            for (int k = 0; k < NDI; k++) e[k] = R.u[k];
            if (S16) {
#pragma unroll
                for (int k = 0; k < NDI; k++) e[k] = p2_shr6(e[k]);
            }
            if constexpr (EDGE) {
#pragma unroll
                for (int i = 0; i < NDI; i++) {
                    const int idx = wd0 + i;
                    const unsigned lo = S16 ? e[i] : p2_rep(e[i], 0x01000100u), hi = S16 ? e[i] : p2_rep(e[i], 0x03020302u);
                    e[i] = idx < 0 ? lo : idx > lastDw ? hi : e[i];
                }
            }
#pragma unroll
            for (int k = 0; k < NP + 1; k++) {
                if (S16) {
                    pU[k] = (int)__builtin_amdgcn_perm(e[2 * k + 2], e[2 * k + 1], 0x05040100u);
                    pV[k] = (int)__builtin_amdgcn_perm(e[2 * k + 2], e[2 * k + 1], 0x07060302u);
                } else {
                    pU[k] = (int)__builtin_amdgcn_perm(e[k + 1], e[k], 0x0C040C02u);
                    pV[k] = (int)__builtin_amdgcn_perm(e[k + 1], e[k], 0x0C050C03u);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            int au = 0, av = 0;
#pragma unroll
            for (int k = 0; k < NP; k++) { au = p2_dot2(pU[c + k], P.h[k], au); av = p2_dot2(pV[c + k], P.h[k], av); }
            su[c] = au; sv[c] = av;
        }
    };

    int hw[NP][4];                                              // [slot][U0, V0, U1, V1]: (row 2m-1 | row 2m << 16)
#pragma unroll
    for (int s = 0; s < NP; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) hw[s][j] = 0;
    P2RowX bufA[2], bufB[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) bufA[i].u[k] = bufB[i].u[k] = 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) bufA[i].v[k] = bufB[i].v[k] = 0u;
    }

    auto body = [&](const int j, auto slot_c, auto edge_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        if (j + 1 < nIter) load(m0 + j + 1, bufA[(SLOT + 1) & 1], bufB[(SLOT + 1) & 1], edge_c);
        {
            int ua[2], va[2], ub[2], vb[2];
            hrow(bufA[SLOT & 1], edge_c, ua, va);
            hrow(bufB[SLOT & 1], edge_c, ub, vb);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                hw[SLOT][2 * c + 0] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(ua[c] >> SHR, ub[c] >> SHR));
                hw[SLOT][2 * c + 1] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(va[c] >> SHR, vb[c] >> SHR));
            }
        }
        if (j >= NP - 1) {
            const int yo = y0 + j - (NP - 1);
            unsigned w[4];                                      // U0 V0 U1 V1
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int acc = P.rnd;
                if constexpr (S16 && !D16) { if (P.dither) acc += p2_dither(3, co, P.drow(yo), q); }
#pragma unroll
                for (int k = 0; k < NP; k++) acc = p2_dot2(hw[(SLOT + 1 + k) % NP][q], P.v[k], acc);
                if (D16) { w[q] = (unsigned)min(max(acc, 0), (1024 << 17) - 1) >> 17; if (P.dstHi6) w[q] <<= 6; }
                else     w[q] = (unsigned)clip_u8_shr(acc, 19);
            }
            if (active) {
                if constexpr (SPL) {                            // planar in, interleaved out
                    uint8_t *d = P.dstU + (unsigned)((unsigned)P.drow(yo) * (unsigned)P.dsU + (D16 ? 4u : 2u) * (unsigned)co);
                    if (D16) st_stream(d, make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16)));
                    else     st_stream(d, (unsigned)(w[0] | (w[1] << 8) | (w[2] << 16) | (w[3] << 24)));
                } else {                                        // interleaved in, planar out: 2 samples to each plane
                    uint8_t *du = P.dstU + (unsigned)((unsigned)P.drow(yo) * (unsigned)P.dsU + (D16 ? 2u : 1u) * (unsigned)co);
                    uint8_t *dv = P.dstV + (unsigned)((unsigned)P.drow(yo) * (unsigned)P.dsV + (D16 ? 2u : 1u) * (unsigned)co);
                    if (D16) { *reinterpret_cast<unsigned *>(du) = w[0] | (w[2] << 16); *reinterpret_cast<unsigned *>(dv) = w[1] | (w[3] << 16); }
                    else     { *reinterpret_cast<unsigned short *>(du) = (unsigned short)(w[0] | (w[2] << 8));
                               *reinterpret_cast<unsigned short *>(dv) = (unsigned short)(w[1] | (w[3] << 8)); }
                }
            }
        }
    };
    if (edgeWave) { load(m0, bufA[0], bufB[0], std::true_type());  p2_rows<NP>(nIter, body, std::true_type()); }
    else          { load(m0, bufA[0], bufB[0], std::false_type()); p2_rows<NP>(nIter, body, std::false_type()); }
}

// the kernel of the mixed-layout pairs: SNV: the SOURCE is interleaved (NV12 | P010LE) and the destination planar; !SNV: the reverse
template <bool SNV, bool S16, bool D16, int NP>
__global__ __launch_bounds__(256) void scale_yuv2px_kernel(Yuv2pArgs a, Yuv2xFrames fr)
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
    const int sHi = (SNV && S16) ? 1 : 0, dHi = (!SNV && D16) ? 1 : 0;     // P010 is the interleaved 10-bit format
    if (lin < a.nblkL) {
        const int seg = __builtin_amdgcn_readfirstlane(lin / a.nsgL);
        const int X0 = ((lin - seg * a.nsgL) * 4 + wave) * P2_STRIP;
        if (X0 >= a.dstW) return;
        const int y0 = seg * a.segRowsL, n = min(a.segRowsL, a.dstH - y0), up = a.updown & seg & 1;
        const P2Plane P = {fr.y[f], fr.dst[f], a.ys, a.ds, a.srcW, a.srcH, a.dstW, a.hL, up ? a.vLup : a.vL, a.lr, sHi, dHi, a.dstH, up, a.dither8 ? 1 : 0};
        p2_walk_plane<S16, D16, NP>(P, X0, up ? a.dstH - (y0 + n) : y0, n, lane);
        return;
    }
    lin -= a.nblkL;
    const int seg = __builtin_amdgcn_readfirstlane(lin / a.nsgC);
    const int X0 = ((lin - seg * a.nsgC) * 4 + wave) * P2_STRIP_UV;
    if (X0 >= a.chrDstW) return;
    const int y0 = seg * a.segRowsC, n = min(a.segRowsC, a.chrDstH - y0), up = a.updown & seg & 1;
    const P2Cross P = {fr.u[f], fr.v[f], fr.dstU[f], fr.dstV[f], a.us, a.vs, a.dsU, a.dsV, a.chrSrcW, a.chrSrcH, a.chrDstW,
                       a.hC, up ? a.vCup : a.vC, a.cr, sHi, dHi, a.chrDstH, up, a.dither8};
    p2_walk_uvx<S16, D16, NP, !SNV>(P, X0, up ? a.chrDstH - (y0 + n) : y0, n, lane);
}

// blockIdx.x: [0, nblkL) luma workgroups (segment-major, 4 strips each), then the chroma workgroups — interleaved (NV): of the
// UV plane, planar: of U, then of V.  blockIdx.y = frame.  S16 / D16: 10 bits in 16-bit containers on that side (interleaved:
// P010LE, bits in the high end; planar: YUV420P10LE, low end).  NP: coefficient pairs per filter (4: 8 taps; 6: Lanczos-3).
template <bool NV, bool S16, bool D16, int NP>
__global__ __launch_bounds__(256) void scale_yuv2p_kernel(Yuv2pArgs a, Yuv2xFrames fr)
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
    const int sHi = (NV && S16) ? 1 : 0, dHi = (NV && D16) ? 1 : 0;
    if (lin < a.nblkL) {
        const int seg = __builtin_amdgcn_readfirstlane(lin / a.nsgL);
        const int X0 = ((lin - seg * a.nsgL) * 4 + wave) * P2_STRIP;
        if (X0 >= a.dstW) return;
        const int y0 = seg * a.segRowsL, n = min(a.segRowsL, a.dstH - y0), up = a.updown & seg & 1;
        const P2Plane P = {fr.y[f], fr.dst[f], a.ys, a.ds, a.srcW, a.srcH, a.dstW, a.hL, up ? a.vLup : a.vL, a.lr, sHi, dHi, a.dstH, up, a.dither8 ? 1 : 0};
        p2_walk_plane<S16, D16, NP>(P, X0, up ? a.dstH - (y0 + n) : y0, n, lane);
        return;
    }
    lin -= a.nblkL;
    if (NV) {
        constexpr bool UVD = !S16 && !D16 && NP == 4;           // 8-bit on both sides, 8 taps: the shared-load walker (124 outputs per wave); measured
                                                                // 3.99 -> 3.59 us per nv12 -> nv12 frame, but 4.19 -> 4.33 for nv12 -> p010: not there
        const int seg = __builtin_amdgcn_readfirstlane(lin / a.nsgC);
        const int X0 = ((lin - seg * a.nsgC) * 4 + wave) * (UVD ? P2_STRIP_UVD : P2_STRIP_UV);
        if (X0 >= a.chrDstW) return;
        const int y0 = seg * a.segRowsC, n = min(a.segRowsC, a.chrDstH - y0), up = a.updown & seg & 1;
        const P2Plane P = {fr.u[f], fr.dstU[f], a.us, a.dsU, a.chrSrcW, a.chrSrcH, a.chrDstW, a.hC, up ? a.vCup : a.vC, a.cr, sHi, dHi, a.chrDstH, up, a.dither8 ? 3 : 0};
        if constexpr (UVD) p2_walk_uvd<D16>(P, X0, up ? a.chrDstH - (y0 + n) : y0, n, lane);
        else               p2_walk_uv<S16, D16, NP>(P, X0, up ? a.chrDstH - (y0 + n) : y0, n, lane);
    } else {
        const int per = a.nsegC * a.nsgC;
        const int pl = __builtin_amdgcn_readfirstlane(lin >= per ? 1 : 0);
        lin -= pl * per;
        const int seg = __builtin_amdgcn_readfirstlane(lin / a.nsgC);
        const int X0 = ((lin - seg * a.nsgC) * 4 + wave) * P2_STRIP;
        if (X0 >= a.chrDstW) return;
        const int y0 = seg * a.segRowsC, n = min(a.segRowsC, a.chrDstH - y0), up = a.updown & seg & 1;
        const P2Plane P = {pl ? fr.v[f] : fr.u[f], pl ? fr.dstV[f] : fr.dstU[f], pl ? a.vs : a.us, pl ? a.dsV : a.dsU,
                           a.chrSrcW, a.chrSrcH, a.chrDstW, a.hC, up ? a.vCup : a.vC, a.cr, 0, 0, a.chrDstH, up, a.dither8 ? (pl ? 2 : 1) : 0};
        p2_walk_plane<S16, D16, NP>(P, X0, up ? a.chrDstH - (y0 + n) : y0, n, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int yuv2p_prepare(const ScalePlan &p, const YuvScaleTiling &g, Yuv2pTables &t)
{
    t = Yuv2pTables();
    const char *off = GMAT_KNOB("GMAT_SCALE_NO_STRIP");
    if (off && atoi(off)) return 0;
    if (g.yuvOut == 2) {
        // 8-bit 4:2:0 -> planar 4:4:4 at exactly 2:1: the chroma planes keep their size, and when their filters are the identity (one
        // tap of 16384 / 4096 on sample x: (u << 7 + 64) >> 7 = u) the chroma is a re-layout and only the luma plane is scaled
        if (!(p.srcFormat == GMAT_PIX_FMT_NV12 || p.srcFormat == GMAT_PIX_FMT_YUV420P) || p.dstFormat != GMAT_PIX_FMT_YUV444P) return 0;
        if (p.srcW != 2 * p.dstW || p.srcH != 2 * p.dstH || p.srcW % 16 || p.srcW < 64 || p.dstH < 16) return 0;
        if (p.chrSrcW != p.chrDstW || p.chrSrcH != p.chrDstH || p.chrDstW != p.dstW || p.chrDstH != p.dstH) return 0;
        if (p.hChr.taps != 1 || g.vChrEff.taps != 1) return 0;
        for (int x = 0; x < p.hChr.count; x++) if (p.hChr.pos[x] != x || p.hChr.coef[x] != 16384) return 0;
        for (int y = 0; y < g.vChrEff.count; y++) if (g.vChrEff.pos[y] != y || g.vChrEff.coef[y] != 4096) return 0;
        if (!filter_is_edge_replication_np(p.hLum, p.srcW, 4, t.hL) || !filter_is_edge_replication_np(g.vLumEff, p.srcH, 4, t.vL)) return 0;
        for (int y = 0; y < p.dstH; y++) if (g.lumRound[y] != g.lumRound[0]) return 0;
        for (int y = 0; y < p.chrDstH; y++) if (g.chrRound[y] != g.lumRound[0]) return 0;     // the same dither on every plane
        t.np = 4; t.lr = g.lumRound[0]; t.snv = p.srcFormat == GMAT_PIX_FMT_NV12;
        t.ok444 = 1;
        return 0;
    }
    if (g.yuvOut != 1) return 0;                                 // 4:2:0 destinations only (8-bit, or 10 bits on the 15-bit lines)
    // 8- or 10-bit 4:2:0 on both sides, interleaved (NV12 | P010LE) or planar (YUV420P | YUV420P10LE) chroma, any pairing
    const bool sNv = p.srcFormat == GMAT_PIX_FMT_NV12 || p.srcFormat == GMAT_PIX_FMT_P010LE;
    const bool dNv = p.dstFormat == GMAT_PIX_FMT_NV12 || p.dstFormat == GMAT_PIX_FMT_P010LE;
    const bool sPl = p.srcFormat == GMAT_PIX_FMT_YUV420P || p.srcFormat == GMAT_PIX_FMT_YUV420P10LE;
    const bool dPl = p.dstFormat == GMAT_PIX_FMT_YUV420P || p.dstFormat == GMAT_PIX_FMT_YUV420P10LE;
    if (!((sNv || sPl) && (dNv || dPl))) return 0;
    t.cross = (sNv != dNv) ? 1 : 0;                              // mixed chroma layouts: scale_yuv2px_kernel
    t.snv = sNv ? 1 : 0;
    t.srcDepth = (p.srcFormat == GMAT_PIX_FMT_P010LE || p.srcFormat == GMAT_PIX_FMT_YUV420P10LE) ? 10 : 8;
    t.dstDepth = (p.dstFormat == GMAT_PIX_FMT_P010LE || p.dstFormat == GMAT_PIX_FMT_YUV420P10LE) ? 10 : 8;
    if (p.srcW != 2 * p.dstW || p.srcH != 2 * p.dstH || p.srcW % 16 || p.srcW < 64 || p.dstH < 16) return 0;
    if (p.chrSrcW * 2 != p.srcW || p.chrSrcH * 2 != p.srcH || p.chrDstW * 2 != p.dstW || p.chrDstH * 2 != p.dstH) return 0;
    // 8-tap filters (bicubic, bilinear, ...) on 4 coefficient pairs, else Lanczos-3's 12 taps on 6; all four filters alike
    t.np = 0;
    for (int np : {4, 6}) {
        if (np == 6 && (p.srcW < 128 || p.dstH < 24)) break;   // the wider window wants planes at least 24 chroma samples across
        if (filter_is_edge_replication_np(p.hLum, p.srcW, np, t.hL) && filter_is_edge_replication_np(p.hChr, p.chrSrcW, np, t.hC) &&
            filter_is_edge_replication_np(g.vLumEff, p.srcH, np, t.vL) && filter_is_edge_replication_np(g.vChrEff, p.chrSrcH, np, t.vC)) {
            t.np = np;
            break;
        }
    }
    if (!t.np) return 0;
    for (int y = 0; y < p.dstH; y++) if (g.lumRound[y] != g.lumRound[0]) return 0;
    for (int y = 0; y < p.chrDstH; y++) if (g.chrRound[y] != g.chrRound[0]) return 0;
    t.lr = g.lumRound[0]; t.cr = g.chrRound[0];
    t.ok = 1;
    return 0;
}

int launch_scale_yuv2p(const Yuv2pArgs &a0, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (!frames || nframes < 1 || nframes > kYuv2xMaxFrames) return GMAT_ERR(EINVAL);
    Yuv2pArgs a = a0;
    const char *segStr = GMAT_KNOB("GMAT_STRIP_ROWS");              // tuning / test override, read per launch
    const int segEnv = segStr ? atoi(segStr) : 0;
    const int nstripsL = (a.dstW + P2_STRIP - 1) / P2_STRIP;
    const bool uvw = a.nv12 || a.cross;                          // the chroma runs on a UV walker (2 outputs per lane, one "plane" of workgroups)
    const bool uvd = a.nv12 && !a.cross && a.srcDepth == 8 && a.dstDepth == 8 && a.np == 4;   // scale_yuv2p_kernel's shared-load UV walker: 124 outputs per wave
    const int nstripsC = uvd ? (a.chrDstW + P2_STRIP_UVD - 1) / P2_STRIP_UVD
                       : uvw ? (a.chrDstW + P2_STRIP_UV - 1) / P2_STRIP_UV : (a.chrDstW + P2_STRIP - 1) / P2_STRIP;
    const int nplC = uvw ? 1 : 2;
    a.nsgL = (nstripsL + 3) / 4; a.nsgC = (nstripsC + 3) / 4;
    int seg = segEnv > 0 ? segEnv : 0;
    if (!seg) {
        // measured on 4K -> 1080p NV12 (profiles/r02c_yuv2p_rows_sweep.txt), best luma rows per segment by frames per
        // launch: 1 -> 3, 2 -> 3, 4 -> 8, 8 -> 16, 16 -> 16, 32 -> 16..24.  Short launches want short segments (a wave's run
        // time is its segment, and the 3 warm-up row pairs are paid from parallelism that would idle anyway); beyond 16 rows
        // the waves get long enough for the tail of the launch to show.  wave-rows / 8640 is within 3 % of the best everywhere.
        const long rows = ((long)a.dstH * nstripsL + (a.lumaOnly ? 0L : (long)a.chrDstH * nstripsC * nplC)) * nframes;
        seg = (int)std::min(16L, std::max(3L, (rows + 8639) / 8640));
        // Re-measured late in round 2 (profiles/r02u_yuv2p_rows_mod4.txt): segment lengths with rows + 3 a multiple of the row loop's
        // unroll factor 4 (5, 9, 13, 17) are the good ones — no partial pass through the unrolled body — and short launches want LONGER
        // segments than the rule above gave: best 5 / 9 / 13 / 17 / 17 / 17 rows at 1 / 2 / 4 / 8 / 16 / 32 frames per launch
        // (nv12 -> nv12: 8.17 -> 7.37, 6.93 -> 5.77, 4.97 -> 4.61, 4.12 -> 3.96 us per frame).  The 16-bit forms want them longer still:
        // p010 -> p010 best 9 / 17 / 17 rows at 1 / 4 / 32 frames (13.1 -> 11.0, 8.45 -> 7.20, 6.39 -> 6.26 us per frame).
        if (a.np == 4) {
            const long per = (a.srcDepth == 8 && a.dstDepth == 8) ? 2880 : 1440;
            seg = (int)std::min(17L, std::max(5L, (rows + per - 1) / per));
            seg = ((seg - 1 + 3) / 4) * 4 + 1;
        }
        // the 6-pair (Lanczos) form pays 5 warm-up row pairs per segment instead of 3: twice the rows (measured best 6 / 12-24 /
        // 32 at 1 / 4 / 32 frames per launch, profiles/r02f_yuv2p_lanczos_rows_sweep.txt)
        if (a.np == 6) seg = std::min(32, std::max(6, 2 * seg));
    }
    // Chroma segments: as many ROWS as luma's when both sides are 8-bit (half the chroma warm-up, equal wave lifetimes: nv12 4K ->
    // 1080p 4.12 -> 3.98 us, yuv420p 4.19 -> 3.89 us per frame), half as many for the 16-bit forms (their chroma waves are the
    // slowest of the launch and longer ones make its tail: p010 6.5 -> 7.5 us with equal rows).  GMAT_P2_CHROMA_SEG = 0 | 1 overrides.
    const char *cse = GMAT_KNOB("GMAT_P2_CHROMA_SEG");
    const bool equalC = cse ? atoi(cse) != 0 : (a.srcDepth == 8 && a.dstDepth == 8 && !a.cross);   // the cross-layout walker: 4.23 -> 4.99 us with equal rows
    // odd segments walk upward (GMAT_STRIP_UPDOWN=0: all downward): the mirrored plane's pairs are the plane's, reversed, halves swapped
    const char *ud = GMAT_KNOB("GMAT_STRIP_UPDOWN");
    a.updown = (ud ? atoi(ud) != 0 : true) && a.srcH == 2 * a.dstH && a.chrSrcH == 2 * a.chrDstH;
    for (int k = 0; k < a.np; k++) {
        const uint32_t l = (uint32_t)a.vL[a.np - 1 - k], c = (uint32_t)a.vC[a.np - 1 - k];
        a.vLup[k] = (int32_t)((l >> 16) | (l << 16)); a.vCup[k] = (int32_t)((c >> 16) | (c << 16));
    }
    a.segRowsL = seg; a.segRowsC = equalC ? seg : std::max(2, (seg + 1) / 2);
    a.nsegL = (a.dstH + a.segRowsL - 1) / a.segRowsL;
    a.nsegC = (a.chrDstH + a.segRowsC - 1) / a.segRowsC;
    a.nblkL = a.nsegL * a.nsgL;
    a.nblk = a.nblkL + (a.lumaOnly ? 0 : a.nsegC * a.nsgC * nplC);
    const dim3 grid(a.xcdRemap ? 8 * ((a.nblk + 7) / 8) : a.nblk, nframes), block(256);
#define GMAT_P2N(NV_, S_, D_) do { if (a.np == 6) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv2p_kernel<NV_, S_, D_, 6>), grid, block, 0, stream, a, *frames); \
                                   else           hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv2p_kernel<NV_, S_, D_, 4>), grid, block, 0, stream, a, *frames); } while (0)
#define GMAT_P2X(NV_, S_, D_) do { if (a.np == 6) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv2px_kernel<NV_, S_, D_, 6>), grid, block, 0, stream, a, *frames); \
                                   else           hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv2px_kernel<NV_, S_, D_, 4>), grid, block, 0, stream, a, *frames); } while (0)
    const int sel = (a.nv12 ? 4 : 0) | (a.srcDepth == 10 ? 2 : 0) | (a.dstDepth == 10 ? 1 : 0);
    if (a.cross) {
        switch (sel) {
        case 0: GMAT_P2X(false, false, false); break; case 1: GMAT_P2X(false, false, true); break;
        case 2: GMAT_P2X(false, true, false);  break; case 3: GMAT_P2X(false, true, true);  break;
        case 4: GMAT_P2X(true, false, false);  break; case 5: GMAT_P2X(true, false, true);  break;
        case 6: GMAT_P2X(true, true, false);   break; default: GMAT_P2X(true, true, true);  break;
        }
        GMAT_HIP_CHECK(hipGetLastError());
        return 0;
    }
#undef GMAT_P2X
    switch (sel) {
    case 0: GMAT_P2N(false, false, false); break; case 1: GMAT_P2N(false, false, true); break;
    case 2: GMAT_P2N(false, true, false);  break; case 3: GMAT_P2N(false, true, true);  break;
    case 4: GMAT_P2N(true, false, false);  break; case 5: GMAT_P2N(true, false, true);  break;
    case 6: GMAT_P2N(true, true, false);   break; default: GMAT_P2N(true, true, true);  break;
    }
#undef GMAT_P2N
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
