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

// ---- one single-channel plane: srcW x srcH -> dstW x dstH (exactly half), rows [y0, y0 + nOut) of the strip at X0 ----
struct P2Plane {
    const uint8_t *src; uint8_t *dst;
    int ss, ds, srcW, srcH, dstW;
    const int32_t *h, *v;                      // 4 int16 pairs each on the odd-aligned window [2x - 3, 2x + 4]
    int rnd;                                   // vertical accumulator start (the dither term << 12)
};

__device__ __forceinline__ void p2_walk_plane(const P2Plane &P, int X0, int y0, int nOut, int lane)
{
    const int xo = X0 + 4 * lane;
    const bool active = xo < P.dstW;
    const int xc = active ? xo : P.dstW - 4;                    // idle lanes shadow the last group (loads stay inside the rows)
    const bool edgeWave = X0 == 0 || X0 + P2_STRIP >= P.dstW;
    const int want = 2 * xc - 4;                                // bytes [2xc - 4, 2xc + 12) of the row
    const int off = min(max(want, 0), P.srcW - 16);
    const int sh = want - off;                                  // -4 at the left plane edge, +4 at the right one
    const unsigned uoff = (unsigned)off;
    const int nIter = nOut + 3;                                 // 3 warm-up row pairs fill the vertical window
    const int32_t h0 = P.h[0], h1 = P.h[1], h2 = P.h[2], h3 = P.h[3];
    const int32_t v0 = P.v[0], v1 = P.v[1], v2 = P.v[2], v3 = P.v[3];

    auto load = [&](int m, uint4 &la, uint4 &lb) {
        const int ra = min(max(2 * m - 1, 0), P.srcH - 1), rb = min(max(2 * m, 0), P.srcH - 1);
        la = p2_ld16(P.src + (unsigned)((unsigned)ra * (unsigned)P.ss + uoff));
        lb = p2_ld16(P.src + (unsigned)((unsigned)rb * (unsigned)P.ss + uoff));
    };
    auto fix = [&](uint4 L, auto edge_c) -> uint4 {
        if (decltype(edge_c)::value) {
            if (sh < 0) L = make_uint4(p2_rep(L.x, 0x00000000u), L.x, L.y, L.z);
            else if (sh > 0) L = make_uint4(L.y, L.z, L.w, p2_rep(L.w, 0x03030303u));
        }
        return L;
    };
    auto hrow = [&](const uint4 &L, int (&s)[4]) {
        int p[7];
        p[0] = p2_pair12(L.x); p[1] = p2_pair30(L.y, L.x); p[2] = p2_pair12(L.y); p[3] = p2_pair30(L.z, L.y);
        p[4] = p2_pair12(L.z); p[5] = p2_pair30(L.w, L.z); p[6] = p2_pair12(L.w);
#pragma unroll
        for (int j = 0; j < 4; j++)
            s[j] = p2_dot2(p[j + 3], h3, p2_dot2(p[j + 2], h2, p2_dot2(p[j + 1], h1, p2_dot2(p[j], h0, 0))));
    };

    int hw[4][4];                                               // [slot][output]: (row 2m-1 | row 2m << 16) after hScale8To15_c
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) hw[s][j] = 0;
    uint4 bufA[2], bufB[2];                                     // ping-pong: iteration j consumes [j & 1], prefetches the other
    bufA[1] = bufB[1] = make_uint4(0u, 0u, 0u, 0u);
    load(y0 - 1, bufA[0], bufB[0]);

    auto body = [&](const int j, auto slot_c, auto edge_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        if (j + 1 < nIter) load(y0 + j, bufA[(SLOT + 1) & 1], bufB[(SLOT + 1) & 1]);
        {
            int sa[4], sb[4];
            hrow(fix(bufA[SLOT & 1], edge_c), sa);
            hrow(fix(bufB[SLOT & 1], edge_c), sb);
#pragma unroll
            for (int q = 0; q < 4; q++)                        // hScale8To15_c: min(val >> 7, 32767)
                hw[SLOT][q] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(sa[q] >> 7, sb[q] >> 7));
        }
        if (j >= 3) {
            const int yo = y0 + j - 3;
            unsigned o = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {                      // yuv2planeX_8_c: clip_u8((dither << 12 + sum) >> 19)
                int acc = P.rnd;
                acc = p2_dot2(hw[(SLOT + 1) & 3][q], v0, acc); acc = p2_dot2(hw[(SLOT + 2) & 3][q], v1, acc);
                acc = p2_dot2(hw[(SLOT + 3) & 3][q], v2, acc); acc = p2_dot2(hw[(SLOT + 4) & 3][q], v3, acc);
                o |= (unsigned)clip_u8_shr(acc, 19) << (8 * q);
            }
            if (active) *reinterpret_cast<unsigned *>(P.dst + (unsigned)((unsigned)yo * (unsigned)P.ds + (unsigned)xo)) = o;
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

// ---- NV12's interleaved UV plane: chrSrcW x chrSrcH sample pairs -> half of each; a lane makes 2 UV outputs a row ------
struct P2PlaneUV {
    const uint8_t *src; uint8_t *dst;
    int ss, ds, srcW, srcH, dstW;              // widths in UV positions
    const int32_t *h, *v;
    int rnd;
};
struct P2RowUV { uint4 a; uint2 b; };          // 16 bytes from sample 2c - 4, 8 bytes from sample 2c + 4 (c = first output)

__device__ __forceinline__ void p2_walk_uv(const P2PlaneUV &P, int X0, int y0, int nOut, int lane)
{
    const int co = X0 + 2 * lane;
    const bool active = co < P.dstW;
    const int cc = active ? co : P.dstW - 2;
    const bool edgeWave = X0 == 0 || X0 + P2_STRIP_UV >= P.dstW;
    const int offA = max(4 * cc - 8, 0), shA = 4 * cc - 8 - offA;               // -8: left edge
    const int offB = min(4 * cc + 8, 2 * P.srcW - 8), shB = 4 * cc + 8 - offB;  // +8: right edge
    const unsigned uoffA = (unsigned)offA, uoffB = (unsigned)offB;
    const int nIter = nOut + 3;
    const int32_t h0 = P.h[0], h1 = P.h[1], h2 = P.h[2], h3 = P.h[3];
    const int32_t v0 = P.v[0], v1 = P.v[1], v2 = P.v[2], v3 = P.v[3];

    auto load = [&](int m, P2RowUV &ra, P2RowUV &rb) {
        const unsigned oa = (unsigned)min(max(2 * m - 1, 0), P.srcH - 1) * (unsigned)P.ss;
        const unsigned ob = (unsigned)min(max(2 * m, 0), P.srcH - 1) * (unsigned)P.ss;
        ra.a = p2_ld16(P.src + (unsigned)(oa + uoffA)); ra.b = p2_ld8(P.src + (unsigned)(oa + uoffB));
        rb.a = p2_ld16(P.src + (unsigned)(ob + uoffA)); rb.b = p2_ld8(P.src + (unsigned)(ob + uoffB));
    };
    // horizontal filter of one row: 2 U and 2 V outputs from 5 odd-aligned pairs per channel
    auto hrow = [&](const P2RowUV &R, auto edge_c, int (&su)[2], int (&sv)[2]) {
        unsigned e[6] = {R.a.x, R.a.y, R.a.z, R.a.w, R.b.x, R.b.y};
        if (decltype(edge_c)::value) {
            if (shA < 0) { const unsigned r = p2_rep(e[0], 0x01000100u); e[3] = e[1]; e[2] = e[0]; e[0] = e[1] = r; }
            if (shB > 0) { e[4] = e[5] = p2_rep(e[5], 0x03020302u); }
        }
        int pU[5], pV[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {           // samples (2k-3, 2k-2) rel. to 2c: bytes 2,3 of e[k] and 0,1 of e[k+1]
            pU[k] = (int)__builtin_amdgcn_perm(e[k + 1], e[k], 0x0C040C02u);
            pV[k] = (int)__builtin_amdgcn_perm(e[k + 1], e[k], 0x0C050C03u);
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
    bufA[1].a = bufB[1].a = make_uint4(0u, 0u, 0u, 0u); bufA[1].b = bufB[1].b = make_uint2(0u, 0u);
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
                hw[SLOT][2 * c + 0] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(ua[c] >> 7, ub[c] >> 7));
                hw[SLOT][2 * c + 1] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(va[c] >> 7, vb[c] >> 7));
            }
        }
        if (j >= 3) {
            const int yo = y0 + j - 3;
            unsigned o = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {                      // yuv2nv12cX_c: U0 V0 U1 V1, clip_u8((dither << 12 + sum) >> 19)
                int acc = P.rnd;
                acc = p2_dot2(hw[(SLOT + 1) & 3][q], v0, acc); acc = p2_dot2(hw[(SLOT + 2) & 3][q], v1, acc);
                acc = p2_dot2(hw[(SLOT + 3) & 3][q], v2, acc); acc = p2_dot2(hw[(SLOT + 4) & 3][q], v3, acc);
                o |= (unsigned)clip_u8_shr(acc, 19) << (8 * q);
            }
            if (active) *reinterpret_cast<unsigned *>(P.dst + (unsigned)((unsigned)yo * (unsigned)P.ds + 2u * (unsigned)co)) = o;
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

// ---------------------------------------------------------------------------------------------------------------------
// 10-bit samples in 16-bit containers: P010LE -> P010LE (HDR transcode) and YUV420P10LE -> YUV420P10LE.  The same walkers with
// 2 bytes per sample: p010LEToY_c / p010LEToUV_c (sample >> 6, input.c:698-725; planar samples as they are), hScale16To15_c
// with sh = 9 (swscale.c:93-119), yuv2p010l1_c / lX_c / cX_c or yuv2planeX_10_c vertically: clip10((1 << 16 + sum) >> 17), P010
// << 6 (output.c:330-384,459-519).  The samples are int16 pairs already; the odd-aligned pair grid is one v_alignbit_b32 per
// pair, P010's >> 6 one packed 16-bit shift per dword.
// ---------------------------------------------------------------------------------------------------------------------
struct P2Plane16 {
    const uint8_t *src; uint8_t *dst;
    int ss, ds, srcW, srcH, dstW;              // strides in bytes, widths in samples
    const int32_t *h, *v;
    int rnd, hi6;                              // hi6: P010 (significant bits are the high ones: >> 6 in, << 6 out)
};
struct P2Row16 { uint4 a, b; };                // 16 samples from sample 2xc - 4

__device__ __forceinline__ unsigned p2_shr6(unsigned v) { return (v >> 6) & 0x03FF03FFu; }        // both halves: v_pk_lshrrev_b16
__device__ __forceinline__ int p2_odd(unsigned hi, unsigned lo) { return (int)((lo >> 16) | (hi << 16)); }   // v_alignbit_b32

__device__ __forceinline__ void p2_walk_plane16(const P2Plane16 &P, int X0, int y0, int nOut, int lane)
{
    const int xo = X0 + 4 * lane;
    const bool active = xo < P.dstW;
    const int xc = active ? xo : P.dstW - 4;
    const bool edgeWave = X0 == 0 || X0 + P2_STRIP >= P.dstW;
    const int want = 2 * xc - 4;                                // samples [2xc - 4, 2xc + 12) of the row
    const int off = min(max(want, 0), P.srcW - 16);
    const int sh = want - off;                                  // -4 / +4 samples at the plane edges
    const unsigned uoff = 2u * (unsigned)off;
    const int nIter = nOut + 3;
    const int32_t h0 = P.h[0], h1 = P.h[1], h2 = P.h[2], h3 = P.h[3];
    const int32_t v0 = P.v[0], v1 = P.v[1], v2 = P.v[2], v3 = P.v[3];
    const bool hi6 = P.hi6 != 0;

    auto load = [&](int m, P2Row16 &ra, P2Row16 &rb) {
        const unsigned oa = (unsigned)min(max(2 * m - 1, 0), P.srcH - 1) * (unsigned)P.ss + uoff;
        const unsigned ob = (unsigned)min(max(2 * m, 0), P.srcH - 1) * (unsigned)P.ss + uoff;
        ra.a = p2_ld16(P.src + oa); ra.b = p2_ld16(P.src + (unsigned)(oa + 16u));
        rb.a = p2_ld16(P.src + ob); rb.b = p2_ld16(P.src + (unsigned)(ob + 16u));
    };
    auto hrow = [&](const P2Row16 &R, auto edge_c, int (&s)[4]) {
        unsigned d[8] = {R.a.x, R.a.y, R.a.z, R.a.w, R.b.x, R.b.y, R.b.z, R.b.w};
        if (hi6) {
#pragma unroll
            for (int k = 0; k < 8; k++) d[k] = p2_shr6(d[k]);
        }
        if (decltype(edge_c)::value) {
            if (sh < 0) {                                       // 4 samples = 2 dwords to the right, first sample replicated
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
        int p[7];                                               // pairs (2k-3, 2k-2) rel. to 2xc: hi half of d[k], lo half of d[k+1]
#pragma unroll
        for (int k = 0; k < 7; k++) p[k] = p2_odd(d[k + 1], d[k]);
#pragma unroll
        for (int j = 0; j < 4; j++)
            s[j] = p2_dot2(p[j + 3], h3, p2_dot2(p[j + 2], h2, p2_dot2(p[j + 1], h1, p2_dot2(p[j], h0, 0))));
    };

    int hw[4][4];
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) hw[s][j] = 0;
    P2Row16 bufA[2], bufB[2];
    bufA[1].a = bufA[1].b = bufB[1].a = bufB[1].b = make_uint4(0u, 0u, 0u, 0u);
    load(y0 - 1, bufA[0], bufB[0]);

    auto body = [&](const int j, auto slot_c, auto edge_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        if (j + 1 < nIter) load(y0 + j, bufA[(SLOT + 1) & 1], bufB[(SLOT + 1) & 1]);
        {
            int sa[4], sb[4];
            hrow(bufA[SLOT & 1], edge_c, sa);
            hrow(bufB[SLOT & 1], edge_c, sb);
#pragma unroll
            for (int q = 0; q < 4; q++)                        // hScale16To15_c: min(val >> 9, 32767)
                hw[SLOT][q] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(sa[q] >> 9, sb[q] >> 9));
        }
        if (j >= 3) {
            const int yo = y0 + j - 3;
            unsigned w[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int acc = P.rnd;
                acc = p2_dot2(hw[(SLOT + 1) & 3][q], v0, acc); acc = p2_dot2(hw[(SLOT + 2) & 3][q], v1, acc);
                acc = p2_dot2(hw[(SLOT + 3) & 3][q], v2, acc); acc = p2_dot2(hw[(SLOT + 4) & 3][q], v3, acc);
                w[q] = (unsigned)min(max(acc, 0), (1024 << 17) - 1) >> 17;          // clamp, then shift (see clip_u8_shr)
                if (hi6) w[q] <<= 6;
            }
            if (active)
                *reinterpret_cast<uint2 *>(P.dst + (unsigned)((unsigned)yo * (unsigned)P.ds + 2u * (unsigned)xo)) =
                    make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16));
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

// P010's interleaved UV plane: a position is one dword (U | V << 16); a lane makes 2 UV outputs from 12 positions a row
struct P2RowUV16 { uint4 a, b, c; };

__device__ __forceinline__ void p2_walk_uv16(const P2Plane16 &P, int X0, int y0, int nOut, int lane)
{
    const int co = X0 + 2 * lane;
    const bool active = co < P.dstW;
    const int cc = active ? co : P.dstW - 2;
    const bool edgeWave = X0 == 0 || X0 + P2_STRIP_UV >= P.dstW;
    const int want = 2 * cc - 4;                                // positions [2cc - 4, 2cc + 8)
    const int off = min(max(want, 0), P.srcW - 12);
    const int sh = want - off;                                  // -4 / +4 positions
    const unsigned uoff = 4u * (unsigned)off;
    const int nIter = nOut + 3;
    const int32_t h0 = P.h[0], h1 = P.h[1], h2 = P.h[2], h3 = P.h[3];
    const int32_t v0 = P.v[0], v1 = P.v[1], v2 = P.v[2], v3 = P.v[3];

    auto load = [&](int m, P2RowUV16 &ra, P2RowUV16 &rb) {
        const unsigned oa = (unsigned)min(max(2 * m - 1, 0), P.srcH - 1) * (unsigned)P.ss + uoff;
        const unsigned ob = (unsigned)min(max(2 * m, 0), P.srcH - 1) * (unsigned)P.ss + uoff;
        ra.a = p2_ld16(P.src + oa); ra.b = p2_ld16(P.src + (unsigned)(oa + 16u)); ra.c = p2_ld16(P.src + (unsigned)(oa + 32u));
        rb.a = p2_ld16(P.src + ob); rb.b = p2_ld16(P.src + (unsigned)(ob + 16u)); rb.c = p2_ld16(P.src + (unsigned)(ob + 32u));
    };
    auto hrow = [&](const P2RowUV16 &R, auto edge_c, int (&su)[2], int (&sv)[2]) {
        unsigned e[12] = {R.a.x, R.a.y, R.a.z, R.a.w, R.b.x, R.b.y, R.b.z, R.b.w, R.c.x, R.c.y, R.c.z, R.c.w};
#pragma unroll
        for (int k = 0; k < 12; k++) e[k] = p2_shr6(e[k]);      // p010LEToUV_c: both samples of the position >> 6
        if (decltype(edge_c)::value) {
            if (sh < 0) {
#pragma unroll
                for (int k = 11; k >= 4; k--) e[k] = e[k - 4];
                e[1] = e[2] = e[3] = e[0];
            } else if (sh > 0) {
#pragma unroll
                for (int k = 0; k < 8; k++) e[k] = e[k + 4];
                e[8] = e[9] = e[10] = e[11];
            }
        }
        int pU[5], pV[5];                                       // positions (2k-3, 2k-2) rel. to 2cc = e[2k+1], e[2k+2]
#pragma unroll
        for (int k = 0; k < 5; k++) {
            pU[k] = (int)__builtin_amdgcn_perm(e[2 * k + 2], e[2 * k + 1], 0x05040100u);
            pV[k] = (int)__builtin_amdgcn_perm(e[2 * k + 2], e[2 * k + 1], 0x07060302u);
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            su[c] = p2_dot2(pU[c + 3], h3, p2_dot2(pU[c + 2], h2, p2_dot2(pU[c + 1], h1, p2_dot2(pU[c], h0, 0))));
            sv[c] = p2_dot2(pV[c + 3], h3, p2_dot2(pV[c + 2], h2, p2_dot2(pV[c + 1], h1, p2_dot2(pV[c], h0, 0))));
        }
    };

    int hw[4][4];                                               // [slot][U0, V0, U1, V1]
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) hw[s][j] = 0;
    P2RowUV16 bufA[2], bufB[2];
    bufA[1].a = bufA[1].b = bufA[1].c = bufB[1].a = bufB[1].b = bufB[1].c = make_uint4(0u, 0u, 0u, 0u);
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
                hw[SLOT][2 * c + 0] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(ua[c] >> 9, ub[c] >> 9));
                hw[SLOT][2 * c + 1] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(va[c] >> 9, vb[c] >> 9));
            }
        }
        if (j >= 3) {
            const int yo = y0 + j - 3;
            unsigned w[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {                      // yuv2p010cX_c: U0 V0 U1 V1
                int acc = P.rnd;
                acc = p2_dot2(hw[(SLOT + 1) & 3][q], v0, acc); acc = p2_dot2(hw[(SLOT + 2) & 3][q], v1, acc);
                acc = p2_dot2(hw[(SLOT + 3) & 3][q], v2, acc); acc = p2_dot2(hw[(SLOT + 4) & 3][q], v3, acc);
                w[q] = ((unsigned)min(max(acc, 0), (1024 << 17) - 1) >> 17) << 6;
            }
            if (active)
                *reinterpret_cast<uint2 *>(P.dst + (unsigned)((unsigned)yo * (unsigned)P.ds + 4u * (unsigned)co)) =
                    make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16));
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

// the 16-bit-container twin of scale_yuv2p_kernel: P010 = true: P010LE -> P010LE, false: YUV420P10LE -> YUV420P10LE
template <bool P010>
__global__ __launch_bounds__(256) void scale_yuv2p16_kernel(Yuv2pArgs a, Yuv2xFrames fr)
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
        const int seg = __builtin_amdgcn_readfirstlane(lin / a.nsgL);
        const int X0 = ((lin - seg * a.nsgL) * 4 + wave) * P2_STRIP;
        if (X0 >= a.dstW) return;
        const int y0 = seg * a.segRowsL;
        const P2Plane16 P = {fr.y[f], fr.dst[f], a.ys, a.ds, a.srcW, a.srcH, a.dstW, a.hL, a.vL, a.lr, P010 ? 1 : 0};
        p2_walk_plane16(P, X0, y0, min(a.segRowsL, a.dstH - y0), lane);
        return;
    }
    lin -= a.nblkL;
    if (P010) {
        const int seg = __builtin_amdgcn_readfirstlane(lin / a.nsgC);
        const int X0 = ((lin - seg * a.nsgC) * 4 + wave) * P2_STRIP_UV;
        if (X0 >= a.chrDstW) return;
        const int y0 = seg * a.segRowsC;
        const P2Plane16 P = {fr.u[f], fr.dstU[f], a.us, a.dsU, a.chrSrcW, a.chrSrcH, a.chrDstW, a.hC, a.vC, a.cr, 1};
        p2_walk_uv16(P, X0, y0, min(a.segRowsC, a.chrDstH - y0), lane);
    } else {
        const int per = a.nsegC * a.nsgC;
        const int pl = __builtin_amdgcn_readfirstlane(lin >= per ? 1 : 0);
        lin -= pl * per;
        const int seg = __builtin_amdgcn_readfirstlane(lin / a.nsgC);
        const int X0 = ((lin - seg * a.nsgC) * 4 + wave) * P2_STRIP;
        if (X0 >= a.chrDstW) return;
        const int y0 = seg * a.segRowsC;
        const P2Plane16 P = {pl ? fr.v[f] : fr.u[f], pl ? fr.dstV[f] : fr.dstU[f], pl ? a.vs : a.us, pl ? a.dsV : a.dsU,
                             a.chrSrcW, a.chrSrcH, a.chrDstW, a.hC, a.vC, a.cr, 0};
        p2_walk_plane16(P, X0, y0, min(a.segRowsC, a.chrDstH - y0), lane);
    }
}

// blockIdx.x: [0, nblkL) luma workgroups (segment-major, 4 strips each), then the chroma workgroups — NV12: of the UV
// plane, planar: of U, then of V.  blockIdx.y = frame.
template <bool NV12>
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
    if (lin < a.nblkL) {
        const int seg = __builtin_amdgcn_readfirstlane(lin / a.nsgL);
        const int X0 = ((lin - seg * a.nsgL) * 4 + wave) * P2_STRIP;
        if (X0 >= a.dstW) return;
        const int y0 = seg * a.segRowsL;
        const P2Plane P = {fr.y[f], fr.dst[f], a.ys, a.ds, a.srcW, a.srcH, a.dstW, a.hL, a.vL, a.lr};
        p2_walk_plane(P, X0, y0, min(a.segRowsL, a.dstH - y0), lane);
        return;
    }
    lin -= a.nblkL;
    if (NV12) {
        const int seg = __builtin_amdgcn_readfirstlane(lin / a.nsgC);
        const int X0 = ((lin - seg * a.nsgC) * 4 + wave) * P2_STRIP_UV;
        if (X0 >= a.chrDstW) return;
        const int y0 = seg * a.segRowsC;
        const P2PlaneUV P = {fr.u[f], fr.dstU[f], a.us, a.dsU, a.chrSrcW, a.chrSrcH, a.chrDstW, a.hC, a.vC, a.cr};
        p2_walk_uv(P, X0, y0, min(a.segRowsC, a.chrDstH - y0), lane);
    } else {
        const int per = a.nsegC * a.nsgC;
        const int pl = __builtin_amdgcn_readfirstlane(lin >= per ? 1 : 0);
        lin -= pl * per;
        const int seg = __builtin_amdgcn_readfirstlane(lin / a.nsgC);
        const int X0 = ((lin - seg * a.nsgC) * 4 + wave) * P2_STRIP;
        if (X0 >= a.chrDstW) return;
        const int y0 = seg * a.segRowsC;
        const P2Plane P = {pl ? fr.v[f] : fr.u[f], pl ? fr.dstV[f] : fr.dstU[f], pl ? a.vs : a.us, pl ? a.dsV : a.dsU,
                           a.chrSrcW, a.chrSrcH, a.chrDstW, a.hC, a.vC, a.cr};
        p2_walk_plane(P, X0, y0, min(a.segRowsC, a.chrDstH - y0), lane);
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
    const bool nv = p.srcFormat == GMAT_PIX_FMT_NV12 && p.dstFormat == GMAT_PIX_FMT_NV12;
    const bool pl = p.srcFormat == GMAT_PIX_FMT_YUV420P && p.dstFormat == GMAT_PIX_FMT_YUV420P;
    const bool nv10 = p.srcFormat == GMAT_PIX_FMT_P010LE && p.dstFormat == GMAT_PIX_FMT_P010LE;
    const bool pl10 = p.srcFormat == GMAT_PIX_FMT_YUV420P10LE && p.dstFormat == GMAT_PIX_FMT_YUV420P10LE;
    if (!nv && !pl && !nv10 && !pl10) return 0;                  // same chroma layout and sample size on both sides
    t.depth = (nv10 || pl10) ? 10 : 8;
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
    if (a.depth == 10) {
        if (a.nv12) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv2p16_kernel<true>), grid, block, 0, stream, a, *frames);
        else        hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv2p16_kernel<false>), grid, block, 0, stream, a, *frames);
    } else if (a.nv12) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv2p_kernel<true>), grid, block, 0, stream, a, *frames);
    else               hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv2p_kernel<false>), grid, block, 0, stream, a, *frames);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
