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

struct P2Plane {
    const uint8_t *src; uint8_t *dst;
    int ss, ds, srcW, srcH, dstW;              // strides in bytes, widths in samples (UV plane: in UV positions)
    const int32_t *h, *v;                      // 4 int16 pairs each on the odd-aligned window [2x - 3, 2x + 4]
    int rnd;                                   // vertical accumulator start (8-bit: the dither term << 12; 10-bit: 1 << 16)
    int srcHi6, dstHi6;                        // P010: the 10 significant bits are the high ones (>> 6 in, << 6 out)
};

// vertical stage of 4 values + store: D16 ? 8 bytes (4 x 16 bit) : 4 bytes
template <bool D16>
__device__ __forceinline__ void p2_vstore(const P2Plane &P, const int (&hw)[4][4], int s1, int s2, int s3, int s4,
                                          int32_t v0, int32_t v1, int32_t v2, int32_t v3, bool active, unsigned byteOff)
{
    unsigned w[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int acc = P.rnd;
        acc = p2_dot2(hw[s1][q], v0, acc); acc = p2_dot2(hw[s2][q], v1, acc);
        acc = p2_dot2(hw[s3][q], v2, acc); acc = p2_dot2(hw[s4][q], v3, acc);
        if (D16) {
            w[q] = (unsigned)min(max(acc, 0), (1024 << 17) - 1) >> 17;               // clamp, then shift (see clip_u8_shr)
            if (P.dstHi6) w[q] <<= 6;
        } else {
            w[q] = (unsigned)clip_u8_shr(acc, 19);
        }
    }
    if (!active) return;
    if (D16) *reinterpret_cast<uint2 *>(P.dst + byteOff) = make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16));
    else     *reinterpret_cast<unsigned *>(P.dst + byteOff) = w[0] | (w[1] << 8) | (w[2] << 16) | (w[3] << 24);
}

// ---- one single-channel plane: srcW x srcH -> dstW x dstH (exactly half), rows [y0, y0 + nOut) of the strip at X0 ----
struct P2Row { uint4 a, b; };                  // 16 samples from sample 2xc - 4 (8-bit: a only)

template <bool S16, bool D16>
__device__ __forceinline__ void p2_walk_plane(const P2Plane &P, int X0, int y0, int nOut, int lane)
{
    const int xo = X0 + 4 * lane;
    const bool active = xo < P.dstW;
    const int xc = active ? xo : P.dstW - 4;                    // idle lanes shadow the last group (loads stay inside the rows)
    const bool edgeWave = X0 == 0 || X0 + P2_STRIP >= P.dstW;
    const int want = 2 * xc - 4;                                // samples [2xc - 4, 2xc + 12) of the row
    const int off = min(max(want, 0), P.srcW - 16);
    const int sh = want - off;                                  // -4 at the left plane edge, +4 at the right one
    const unsigned uoff = (S16 ? 2u : 1u) * (unsigned)off;
    const int nIter = nOut + 3;                                 // 3 warm-up row pairs fill the vertical window
    const int32_t h0 = P.h[0], h1 = P.h[1], h2 = P.h[2], h3 = P.h[3];
    const int32_t v0 = P.v[0], v1 = P.v[1], v2 = P.v[2], v3 = P.v[3];
    const bool hi6 = P.srcHi6 != 0;

    auto load1 = [&](int row, P2Row &r) {
        const unsigned o = (unsigned)min(max(row, 0), P.srcH - 1) * (unsigned)P.ss + uoff;
        r.a = p2_ld16(P.src + o);
        if (S16) r.b = p2_ld16(P.src + (unsigned)(o + 16u));
    };
    auto load = [&](int m, P2Row &ra, P2Row &rb) { load1(2 * m - 1, ra); load1(2 * m, rb); };
    // horizontal filter of one row: 4 outputs from 7 odd-aligned pairs
    auto hrow = [&](const P2Row &R, auto edge_c, int (&s)[4]) {
        int p[7];
        if (S16) {
            unsigned d[8] = {R.a.x, R.a.y, R.a.z, R.a.w, R.b.x, R.b.y, R.b.z, R.b.w};
            if (hi6) {
#pragma unroll
                for (int k = 0; k < 8; k++) d[k] = p2_shr6(d[k]);
            }
            if (decltype(edge_c)::value) {
                if (sh < 0) {                                   // 4 samples = 2 dwords to the right, first sample replicated
                    const unsigned r = p2_rep(d[0], 0x01000100u);
#pragma unroll
                    for (int k = 7; k >= 2; k--) d[k] = d[k - 2];
                    d[0] = d[1] = r;
                } else if (sh > 0) {
                    const unsigned r = p2_rep(d[7], 0x03020302u);
#pragma unroll
                    for (int k = 0; k < 6; k++) d[k] = d[k + 2];
                    d[6] = d[7] = r;
                }
            }
#pragma unroll
            for (int k = 0; k < 7; k++) p[k] = p2_odd(d[k + 1], d[k]);      // samples (2k-3, 2k-2) rel. to 2xc
        } else {
            uint4 L = R.a;
            if (decltype(edge_c)::value) {
                if (sh < 0) L = make_uint4(p2_rep(L.x, 0x00000000u), L.x, L.y, L.z);
                else if (sh > 0) L = make_uint4(L.y, L.z, L.w, p2_rep(L.w, 0x03030303u));
            }
            p[0] = p2_pair12(L.x); p[1] = p2_pair30(L.y, L.x); p[2] = p2_pair12(L.y); p[3] = p2_pair30(L.z, L.y);
            p[4] = p2_pair12(L.z); p[5] = p2_pair30(L.w, L.z); p[6] = p2_pair12(L.w);
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
            s[j] = p2_dot2(p[j + 3], h3, p2_dot2(p[j + 2], h2, p2_dot2(p[j + 1], h1, p2_dot2(p[j], h0, 0))));
    };

    int hw[4][4];                                               // [slot][output]: (row 2m-1 | row 2m << 16), 15-bit lines
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) hw[s][j] = 0;
    P2Row bufA[2], bufB[2];                                     // ping-pong: iteration j consumes [j & 1], prefetches the other
    bufA[0].b = bufB[0].b = bufA[1].a = bufA[1].b = bufB[1].a = bufB[1].b = make_uint4(0u, 0u, 0u, 0u);
    load(y0 - 1, bufA[0], bufB[0]);

    auto body = [&](const int j, auto slot_c, auto edge_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        if (j + 1 < nIter) load(y0 + j, bufA[(SLOT + 1) & 1], bufB[(SLOT + 1) & 1]);
        {
            int sa[4], sb[4];
            hrow(bufA[SLOT & 1], edge_c, sa);
            hrow(bufB[SLOT & 1], edge_c, sb);
#pragma unroll
            for (int q = 0; q < 4; q++)                        // hScale8To15_c / hScale16To15_c: min(val >> 7 | 9, 32767)
                hw[SLOT][q] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(sa[q] >> (S16 ? 9 : 7), sb[q] >> (S16 ? 9 : 7)));
        }
        if (j >= 3) {
            const int yo = y0 + j - 3;
            p2_vstore<D16>(P, hw, (SLOT + 1) & 3, (SLOT + 2) & 3, (SLOT + 3) & 3, SLOT & 3, v0, v1, v2, v3, active,
                           (unsigned)((unsigned)yo * (unsigned)P.ds + (D16 ? 2u : 1u) * (unsigned)xo));
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

// ---- the interleaved UV plane (NV12: 2 bytes a position; P010: one dword): a lane makes 2 UV outputs a row ------------
struct P2RowUV { uint4 a, b, c; };             // 8-bit: a = 16 bytes from position 2c - 4, b.xy = 8 bytes from 2c + 4; 16-bit: 12 dwords

template <bool S16, bool D16>
__device__ __forceinline__ void p2_walk_uv(const P2Plane &P, int X0, int y0, int nOut, int lane)
{
    const int co = X0 + 2 * lane;
    const bool active = co < P.dstW;
    const int cc = active ? co : P.dstW - 2;
    const bool edgeWave = X0 == 0 || X0 + P2_STRIP_UV >= P.dstW;
    // 8-bit: two loads with their own clamps (bytes); 16-bit: one 48-byte window, positions [2cc - 4, 2cc + 8)
    const int offA = max(4 * cc - 8, 0), shA = 4 * cc - 8 - offA;               // -8 bytes: left edge
    const int offB = min(4 * cc + 8, 2 * P.srcW - 8), shB = 4 * cc + 8 - offB;  // +8 bytes: right edge
    const int want = 2 * cc - 4;
    const int off16 = min(max(want, 0), P.srcW - 12);
    const int sh16 = want - off16;                              // -4 / +4 positions
    const unsigned uoffA = S16 ? 4u * (unsigned)off16 : (unsigned)offA, uoffB = (unsigned)offB;
    const int nIter = nOut + 3;
    const int32_t h0 = P.h[0], h1 = P.h[1], h2 = P.h[2], h3 = P.h[3];
    const int32_t v0 = P.v[0], v1 = P.v[1], v2 = P.v[2], v3 = P.v[3];

    auto load1 = [&](int row, P2RowUV &r) {
        const unsigned o = (unsigned)min(max(row, 0), P.srcH - 1) * (unsigned)P.ss;
        r.a = p2_ld16(P.src + (unsigned)(o + uoffA));
        if (S16) { r.b = p2_ld16(P.src + (unsigned)(o + uoffA + 16u)); r.c = p2_ld16(P.src + (unsigned)(o + uoffA + 32u)); }
        else     { const uint2 t = p2_ld8(P.src + (unsigned)(o + uoffB)); r.b = make_uint4(t.x, t.y, 0u, 0u); }
    };
    auto load = [&](int m, P2RowUV &ra, P2RowUV &rb) { load1(2 * m - 1, ra); load1(2 * m, rb); };
    // horizontal filter of one row: 2 U and 2 V outputs from 5 odd-aligned pairs per channel
    auto hrow = [&](const P2RowUV &R, auto edge_c, int (&su)[2], int (&sv)[2]) {
        int pU[5], pV[5];
        if (S16) {
            unsigned e[12] = {R.a.x, R.a.y, R.a.z, R.a.w, R.b.x, R.b.y, R.b.z, R.b.w, R.c.x, R.c.y, R.c.z, R.c.w};
#pragma unroll
            for (int k = 0; k < 12; k++) e[k] = p2_shr6(e[k]);  // p010LEToUV_c: both samples of the position >> 6
            if (decltype(edge_c)::value) {
                if (sh16 < 0) {
#pragma unroll
                    for (int k = 11; k >= 4; k--) e[k] = e[k - 4];
                    e[1] = e[2] = e[3] = e[0];
                } else if (sh16 > 0) {
#pragma unroll
                    for (int k = 0; k < 8; k++) e[k] = e[k + 4];
                    e[8] = e[9] = e[10] = e[11];
                }
            }
#pragma unroll
            for (int k = 0; k < 5; k++) {       // positions (2k-3, 2k-2) rel. to 2cc = e[2k+1], e[2k+2]
                pU[k] = (int)__builtin_amdgcn_perm(e[2 * k + 2], e[2 * k + 1], 0x05040100u);
                pV[k] = (int)__builtin_amdgcn_perm(e[2 * k + 2], e[2 * k + 1], 0x07060302u);
            }
        } else {
            unsigned e[6] = {R.a.x, R.a.y, R.a.z, R.a.w, R.b.x, R.b.y};
            if (decltype(edge_c)::value) {
                if (shA < 0) { const unsigned r = p2_rep(e[0], 0x01000100u); e[3] = e[1]; e[2] = e[0]; e[0] = e[1] = r; }
                if (shB > 0) { e[4] = e[5] = p2_rep(e[5], 0x03020302u); }
            }
#pragma unroll
            for (int k = 0; k < 5; k++) {       // samples (2k-3, 2k-2) rel. to 2c: bytes 2,3 of e[k] and 0,1 of e[k+1]
                pU[k] = (int)__builtin_amdgcn_perm(e[k + 1], e[k], 0x0C040C02u);
                pV[k] = (int)__builtin_amdgcn_perm(e[k + 1], e[k], 0x0C050C03u);
            }
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            su[c] = p2_dot2(pU[c + 3], h3, p2_dot2(pU[c + 2], h2, p2_dot2(pU[c + 1], h1, p2_dot2(pU[c], h0, 0))));
            sv[c] = p2_dot2(pV[c + 3], h3, p2_dot2(pV[c + 2], h2, p2_dot2(pV[c + 1], h1, p2_dot2(pV[c], h0, 0))));
        }
    };

    int hw[4][4];                                               // [slot][U0, V0, U1, V1]: (row 2m-1 | row 2m << 16)
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) hw[s][j] = 0;
    P2RowUV bufA[2], bufB[2];
#pragma unroll
    for (int i = 0; i < 2; i++) bufA[i].a = bufA[i].b = bufA[i].c = bufB[i].a = bufB[i].b = bufB[i].c = make_uint4(0u, 0u, 0u, 0u);
    load(y0 - 1, bufA[0], bufB[0]);

    auto body = [&](const int j, auto slot_c, auto edge_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        if (j + 1 < nIter) load(y0 + j, bufA[(SLOT + 1) & 1], bufB[(SLOT + 1) & 1]);
        {
            int ua[2], va[2], ub[2], vb[2];
            hrow(bufA[SLOT & 1], edge_c, ua, va);
            hrow(bufB[SLOT & 1], edge_c, ub, vb);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                hw[SLOT][2 * c + 0] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(ua[c] >> (S16 ? 9 : 7), ub[c] >> (S16 ? 9 : 7)));
                hw[SLOT][2 * c + 1] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(va[c] >> (S16 ? 9 : 7), vb[c] >> (S16 ? 9 : 7)));
            }
        }
        if (j >= 3) {
            const int yo = y0 + j - 3;                          // yuv2nv12cX_c / yuv2p010cX_c: U0 V0 U1 V1
            p2_vstore<D16>(P, hw, (SLOT + 1) & 3, (SLOT + 2) & 3, (SLOT + 3) & 3, SLOT & 3, v0, v1, v2, v3, active,
                           (unsigned)((unsigned)yo * (unsigned)P.ds + (D16 ? 4u : 2u) * (unsigned)co));
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

// blockIdx.x: [0, nblkL) luma workgroups (segment-major, 4 strips each), then the chroma workgroups — interleaved (NV): of the
// UV plane, planar: of U, then of V.  blockIdx.y = frame.  S16 / D16: 10 bits in 16-bit containers on that side (interleaved:
// P010LE, bits in the high end; planar: YUV420P10LE, low end).
template <bool NV, bool S16, bool D16>
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
        const int y0 = seg * a.segRowsL;
        const P2Plane P = {fr.y[f], fr.dst[f], a.ys, a.ds, a.srcW, a.srcH, a.dstW, a.hL, a.vL, a.lr, sHi, dHi};
        p2_walk_plane<S16, D16>(P, X0, y0, min(a.segRowsL, a.dstH - y0), lane);
        return;
    }
    lin -= a.nblkL;
    if (NV) {
        const int seg = __builtin_amdgcn_readfirstlane(lin / a.nsgC);
        const int X0 = ((lin - seg * a.nsgC) * 4 + wave) * P2_STRIP_UV;
        if (X0 >= a.chrDstW) return;
        const int y0 = seg * a.segRowsC;
        const P2Plane P = {fr.u[f], fr.dstU[f], a.us, a.dsU, a.chrSrcW, a.chrSrcH, a.chrDstW, a.hC, a.vC, a.cr, sHi, dHi};
        p2_walk_uv<S16, D16>(P, X0, y0, min(a.segRowsC, a.chrDstH - y0), lane);
    } else {
        const int per = a.nsegC * a.nsgC;
        const int pl = __builtin_amdgcn_readfirstlane(lin >= per ? 1 : 0);
        lin -= pl * per;
        const int seg = __builtin_amdgcn_readfirstlane(lin / a.nsgC);
        const int X0 = ((lin - seg * a.nsgC) * 4 + wave) * P2_STRIP;
        if (X0 >= a.chrDstW) return;
        const int y0 = seg * a.segRowsC;
        const P2Plane P = {pl ? fr.v[f] : fr.u[f], pl ? fr.dstV[f] : fr.dstU[f], pl ? a.vs : a.us, pl ? a.dsV : a.dsU,
                           a.chrSrcW, a.chrSrcH, a.chrDstW, a.hC, a.vC, a.cr, 0, 0};
        p2_walk_plane<S16, D16>(P, X0, y0, min(a.segRowsC, a.chrDstH - y0), lane);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int yuv2p_prepare(const ScalePlan &p, const YuvScaleTiling &g, Yuv2pTables &t)
{
    t = Yuv2pTables();
    const char *off = getenv("GMAT_SCALE_NO_STRIP");
    if (off && atoi(off)) return 0;
    if (g.yuvOut != 1) return 0;                                 // 4:2:0 destinations only (8-bit, or 10 bits on the 15-bit lines)
    // same chroma layout on both sides: interleaved (NV12 | P010LE) -> (NV12 | P010LE), planar (YUV420P | YUV420P10LE) -> the same
    const bool sNv = p.srcFormat == GMAT_PIX_FMT_NV12 || p.srcFormat == GMAT_PIX_FMT_P010LE;
    const bool dNv = p.dstFormat == GMAT_PIX_FMT_NV12 || p.dstFormat == GMAT_PIX_FMT_P010LE;
    const bool sPl = p.srcFormat == GMAT_PIX_FMT_YUV420P || p.srcFormat == GMAT_PIX_FMT_YUV420P10LE;
    const bool dPl = p.dstFormat == GMAT_PIX_FMT_YUV420P || p.dstFormat == GMAT_PIX_FMT_YUV420P10LE;
    if (!((sNv && dNv) || (sPl && dPl))) return 0;
    t.srcDepth = (p.srcFormat == GMAT_PIX_FMT_P010LE || p.srcFormat == GMAT_PIX_FMT_YUV420P10LE) ? 10 : 8;
    t.dstDepth = (p.dstFormat == GMAT_PIX_FMT_P010LE || p.dstFormat == GMAT_PIX_FMT_YUV420P10LE) ? 10 : 8;
    if (p.srcW != 2 * p.dstW || p.srcH != 2 * p.dstH || p.srcW % 16 || p.srcW < 64 || p.dstH < 16) return 0;
    if (p.chrSrcW * 2 != p.srcW || p.chrSrcH * 2 != p.srcH || p.chrDstW * 2 != p.dstW || p.chrDstH * 2 != p.dstH) return 0;
    if (!filter_is_edge_replication(p.hLum, p.srcW, t.hL)) return 0;
    if (!filter_is_edge_replication(p.hChr, p.chrSrcW, t.hC)) return 0;
    if (!filter_is_edge_replication(g.vLumEff, p.srcH, t.vL)) return 0;
    if (!filter_is_edge_replication(g.vChrEff, p.chrSrcH, t.vC)) return 0;
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
    const char *segStr = getenv("GMAT_STRIP_ROWS");              // tuning / test override, read per launch
    const int segEnv = segStr ? atoi(segStr) : 0;
    const int nstripsL = (a.dstW + P2_STRIP - 1) / P2_STRIP;
    const int nstripsC = a.nv12 ? (a.chrDstW + P2_STRIP_UV - 1) / P2_STRIP_UV : (a.chrDstW + P2_STRIP - 1) / P2_STRIP;
    const int nplC = a.nv12 ? 1 : 2;
    a.nsgL = (nstripsL + 3) / 4; a.nsgC = (nstripsC + 3) / 4;
    int seg = segEnv > 0 ? segEnv : 0;
    if (!seg) {
        // measured on 4K -> 1080p NV12 (profiles/r02c_yuv2p_rows_sweep.txt), best luma rows per segment by frames per
        // launch: 1 -> 3, 2 -> 3, 4 -> 8, 8 -> 16, 16 -> 16, 32 -> 16..24.  Short launches want short segments (a wave's run
        // time is its segment, and the 3 warm-up row pairs are paid from parallelism that would idle anyway); beyond 16 rows
        // the waves get long enough for the tail of the launch to show.  wave-rows / 8640 is within 3 % of the best everywhere.
        const long rows = ((long)a.dstH * nstripsL + (long)a.chrDstH * nstripsC * nplC) * nframes;
        seg = (int)std::min(16L, std::max(3L, (rows + 8639) / 8640));
    }
    a.segRowsL = seg; a.segRowsC = std::max(2, (seg + 1) / 2);
    a.nsegL = (a.dstH + a.segRowsL - 1) / a.segRowsL;
    a.nsegC = (a.chrDstH + a.segRowsC - 1) / a.segRowsC;
    a.nblkL = a.nsegL * a.nsgL;
    a.nblk = a.nblkL + a.nsegC * a.nsgC * nplC;
    const dim3 grid(a.xcdRemap ? 8 * ((a.nblk + 7) / 8) : a.nblk, nframes), block(256);
#define GMAT_P2(NV_, S_, D_) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv2p_kernel<NV_, S_, D_>), grid, block, 0, stream, a, *frames)
    const int sel = (a.nv12 ? 4 : 0) | (a.srcDepth == 10 ? 2 : 0) | (a.dstDepth == 10 ? 1 : 0);
    switch (sel) {
    case 0: GMAT_P2(false, false, false); break; case 1: GMAT_P2(false, false, true); break;
    case 2: GMAT_P2(false, true, false);  break; case 3: GMAT_P2(false, true, true);  break;
    case 4: GMAT_P2(true, false, false);  break; case 5: GMAT_P2(true, false, true);  break;
    case 6: GMAT_P2(true, true, false);   break; default: GMAT_P2(true, true, true);  break;
    }
#undef GMAT_P2
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
