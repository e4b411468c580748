// k_scale19.hip — libswscale's generic scaler for 16-bit YUV destinations (P016LE, YUV420P16LE, YUV444P16LE) in ONE launch, gfx950.
//
// The arithmetic is k_scale16.hip's (dstBpc = 16, utils.c:1561-1570): 19-bit lines held in int32,
//   horizontal   hScale8To19_c    min(sum >> 3, 2^19 - 1)                          swscale.c:138-153
//                hScale16To19_c   min(sum >> (depth - 5), 2^19 - 1)                swscale.c:63-91
//   vertical     yuv2planeX_16_c  0x8000 + clip_int16(((1 << 14) - 0x40000000 + sum src * (unsigned)filter) >> 15)
//                                 in 32-bit wrap-around arithmetic                 output.c:157-181
//                yuv2nv12cX_16_c  the X form per chroma plane, interleaved         output.c:183-211
// Round 5's last hour timed that two-pass path for the first time: P016LE 1080p -> 720p 35-41 us a frame, 0.03 of the roofline — five or
// six launches a frame, two-byte gathers through the vector cache, the int32 lines of a whole frame through HBM.  Here a block owns a tile
// of 64 output columns x TH output rows of a plane (or of both chroma planes):
//   stage   the source rows the tile's vertical taps span, the samples its horizontal windows span, global -> LDS; whatever the source is
//           (8- or 16-bit samples, interleaved or planar chroma, P010's shift) a staged row is a plane's row of 16-bit sample PAIRS, one
//           dword a pair, 16-bit samples biased by -2^15 so that v_dot2_i32_i16 takes them (the bias returns as 2^15 sum(c) a column);
//   pass H  a lane owns a column (its coefficient pairs stay in registers), a wave walks the staged rows four at a time: a pair of taps is
//           one ds_read_b32 and one v_dot2_i32_i16; the 19-bit line values go to LDS;
//   pass V  a thread owns four columns of an output row (two of each chroma plane where the destination interleaves them): ds_read_b128 of
//           the lines, v_mad_i32_i24 (a 19-bit line x a 13-bit coefficient: the low 32 bits of the product ARE the 32-bit wrap-around
//           product), the row's coefficients unpacked once a tile in LDS; four samples stored.
// The lines never leave the CU; a frame is one launch, a batch of frames one launch too (grid.y).  First form of this file (profiles/r06a,
// r06b): sample-by-sample LDS reads and v_mul_i32_i24 in pass H, column pairs in pass V — 45 + 35 VALU instructions an output, 9.0 us a
// 1080p -> 720p frame batched: bound by instruction issue (a wave64 VALU instruction is 4 cycles on a SIMD here).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

// measurement builds (tools/build_variant.sh s19pN k_scale19.hip -DS19_PROBE=N; wrong pixels): 1 no staging, 2 no pass H, 3 no pass V (stores stay), 4 nothing (the launch's floor)
#ifndef S19_PROBE
#define S19_PROBE 0
#endif

namespace gmat {

namespace {

// lum / chrRange{To,From}Jpeg16_c on a 19-bit line value (swscale.c:189-226), as hscale19_kernel states them
__device__ __forceinline__ int s19_range(int v, int rc)
{
    if (rc == 1)      v = (int)((unsigned)min(v, 30189 << 4) * 4769u - (unsigned)(39057361 << 2)) >> 12;
    else if (rc == 2) v = (int)((unsigned)v * (unsigned)(14071 / 4) + (unsigned)((33561947 << 4) / 4)) >> 12;
    else if (rc == 3) v = (int)((unsigned)min(v, 30775 << 4) * 4663u - (unsigned)(9289992 << 4)) >> 12;
    else if (rc == 4) v = (int)((unsigned)v * 1799u + (unsigned)(4081085 << 4)) >> 11;
    // the 15-bit lines' forms (lum / chrRange{To,From}Jpeg_c, swscale.c:157-188, as scale_yuv_kernel states them)
    else if (rc == 5) v = (__mul24(min(v, 30189), 19077) - 39057361) >> 14;
    else if (rc == 6) v = (__mul24(v, 14071) + 33561947) >> 14;
    else if (rc == 7) v = (__mul24(min(v, 30775), 4663) - 9289992) >> 12;
    else if (rc == 8) v = (__mul24(v, 1799) + 4081085) >> 11;
    return v;
}

__device__ __forceinline__ int s19_dot2(unsigned samples, int coefs, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, samples), __builtin_bit_cast(short2v, coefs), acc, false);
}

// four samples of every component a source row image carries (one staging unit) -> sample pairs, 16-bit samples as v_dot2_i32_i16 takes them.
// LAYOUT 0: 8-bit planar (4 bytes -> two pairs); 1: 8-bit interleaved (8 bytes -> two pairs each of A and B); 2: 16-bit planar (8 bytes);
// 3: 16-bit interleaved (16 bytes)
template <int LAYOUT, bool SHR6>
__device__ __forceinline__ void s19_pairs_t(const unsigned *v, unsigned xorv, unsigned &a0, unsigned &a1, unsigned &b0, unsigned &b1)
{
    b0 = b1 = 0;
    if (LAYOUT == 0) {
        a0 = (v[0] & 0xFFu) | (v[0] & 0xFF00u) << 8; a1 = (v[0] >> 16 & 0xFFu) | (v[0] >> 24) << 16;
    } else if (LAYOUT == 1) {                                                   // u0 v0 u1 v1 | u2 v2 u3 v3
        a0 = v[0] & 0x00FF00FFu; b0 = v[0] >> 8 & 0x00FF00FFu;
        a1 = v[1] & 0x00FF00FFu; b1 = v[1] >> 8 & 0x00FF00FFu;
    } else if (LAYOUT == 2) {
        a0 = v[0]; a1 = v[1];
    } else {                                                                    // (u0 v0) (u1 v1) (u2 v2) (u3 v3)
        a0 = (v[0] & 0xFFFFu) | v[1] << 16; b0 = v[0] >> 16 | (v[1] & 0xFFFF0000u);
        a1 = (v[2] & 0xFFFFu) | v[3] << 16; b1 = v[2] >> 16 | (v[3] & 0xFFFF0000u);
    }
    if (LAYOUT >= 2) {
        // (P010's ten bits, kind 10: xorv is 0 there — a template argument: as a run-time flag the shift cost two selects a dword in the commit, r06v)
        if (SHR6) { a0 = a0 >> 6 & 0x03FF03FFu; a1 = a1 >> 6 & 0x03FF03FFu; b0 = b0 >> 6 & 0x03FF03FFu; b1 = b1 >> 6 & 0x03FF03FFu; }
        else      { a0 ^= xorv; a1 ^= xorv; b0 ^= xorv; b1 ^= xorv; }
    }
}

template <int LAYOUT>
__device__ __forceinline__ void s19_pairs(const unsigned *v, unsigned xorv, bool shr6, unsigned &a0, unsigned &a1, unsigned &b0, unsigned &b1)
{
    if (LAYOUT >= 2 && shr6) s19_pairs_t<LAYOUT, true>(v, xorv, a0, a1, b0, b1);
    else                     s19_pairs_t<LAYOUT, false>(v, xorv, a0, a1, b0, b1);
}

// keeps an offset that advances by a constant a step as ONE v_add a step: left alone the compiler re-derives every step's offset as (row + step rows) x stride,
// a v_mad_u64_u32 each (r06v)
#if defined(__HIP_DEVICE_COMPILE__)
#define S19_OPAQUE(v) asm volatile("" : "+v"(v))
#else
#define S19_OPAQUE(v) ((void)0)
#endif

// acc + a b on 24-bit operands as ONE v_mad_i32_i24 (the low 32 bits of the product: the wrap-around 32-bit multiply-add for a 19-bit line and a 13-bit
// coefficient).  Written as `acc += __mul24(..) + __mul24(..)` the compiler makes two v_mul_i32_i24 and a v_add3_u32 of a tap pair — 12 instructions a pair and
// column quad where 8 do (profiles/r06_scale19_history.txt r06r)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ unsigned s19_mad(int a, int b, unsigned acc) { unsigned r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc)); return r; }
#else
__device__ __forceinline__ unsigned s19_mad(int a, int b, unsigned acc) { return acc + (unsigned)__mul24(a, b); }      // (the emulator, and hipcc's host pass)
#endif

// Staging: rows [0, gn) of one source row image -> staged rows.  sp: the image's row r0 + g0; B0: byte offset of the tile's first staged
// sample group; out0 / out1: the staged rows of the component(s) the image carries, PP dwords a row; nunit units a row.
// The fast form (every row address a multiple of 4, a row holds at least one unit) is split in two so that a group's loads are in flight while
// the block filters the group before it: s19_issue asks for EVERY unit of the group at once — a lane takes unit (lane mod LPR) of rows
// (lane / LPR) + 4 RPW u, u < U, CP column passes when a row has more than 64 units; SLOTS = CP U register slots of NDW dwords — with no branch
// between the loads (a row past the group's last, a unit past the row's last repeat the last one: the same dwords are stored twice; a unit
// past the row's last WHOLE one reads that one and stores zeros); s19_commit turns the registers into sample pairs in LDS.  The ONE unit that
// straddles a row's end is patched byte by byte afterwards, a thread a row.  32-bit offsets from sp: a plane spans less than 4 GB.
template <int LAYOUT, int SLOTS, int CP, int pfBase>
__device__ __forceinline__ void s19_issue(const uint8_t *sp, unsigned stride, int rowBytes, int B0, int gn, int nunit, int lshift,
                                          unsigned (&pf)[16], int lane, int wave)
{
    constexpr int NDW = LAYOUT == 0 ? 1 : LAYOUT == 3 ? 4 : 2, UB = 4 * NDW, U = SLOTS / CP;
    const int LPR = 1 << lshift, RPW = 64 >> lshift;
    const int ul = lane & (LPR - 1), row0 = wave * RPW + (lane >> lshift);
    const int offLast = (rowBytes / UB - 1) * UB;                               // the last whole unit of a row from its start
    const unsigned gstep = (unsigned)(4 * RPW) * stride;
#pragma unroll
    for (int cp = 0; cp < CP; cp++) {
        const int u0 = min(ul + cp * LPR, nunit - 1);
        const unsigned offc = (unsigned)min(B0 + u0 * UB, offLast);
        const unsigned gLast = (unsigned)(gn - 1) * stride + offc;
        unsigned g = (unsigned)row0 * stride + offc;
#pragma unroll
        for (int u = 0; u < U; u++) {
            S19_OPAQUE(g);
            const unsigned *gp = reinterpret_cast<const unsigned *>(sp + min(g, gLast));
#pragma unroll
            for (int i = 0; i < NDW; i++) pf[pfBase + (cp * U + u) * NDW + i] = gp[i];
            g += gstep;
        }
    }
}

// (a unit past the row's last whole one is stored as it was read — the last whole one's samples, where round 6's first form stored zeros at two selects a slot: every
// tap there has a zero coefficient, check_banks, and the straddling unit is rewritten by the patch pass)
template <int LAYOUT, int SLOTS, int CP, int pfBase, bool SHR6>
__device__ __forceinline__ void s19_commit(int gn, int nunit, int lshift, unsigned xorv,
                                           unsigned *out0, unsigned *out1, int PP, const unsigned (&pf)[16], int lane, int wave)
{
    constexpr int NDW = LAYOUT == 0 ? 1 : LAYOUT == 3 ? 4 : 2, U = SLOTS / CP;
    const int LPR = 1 << lshift, RPW = 64 >> lshift;
    const int ul = lane & (LPR - 1), row0 = wave * RPW + (lane >> lshift);
    const unsigned lstep = (unsigned)(4 * RPW * PP);
#pragma unroll
    for (int cp = 0; cp < CP; cp++) {
        const int u0 = min(ul + cp * LPR, nunit - 1);
        const unsigned lLast = (unsigned)((gn - 1) * PP + 2 * u0);
        unsigned l = (unsigned)(row0 * PP + 2 * u0);
#pragma unroll
        for (int u = 0; u < U; u++) {
            S19_OPAQUE(l);
            unsigned a0, a1, b0, b1, z[NDW];
#pragma unroll
            for (int i = 0; i < NDW; i++) z[i] = pf[pfBase + (cp * U + u) * NDW + i];
            s19_pairs_t<LAYOUT, SHR6>(z, xorv, a0, a1, b0, b1);
            const unsigned lo = min(l, lLast);
            *reinterpret_cast<uint2 *>(out0 + lo) = make_uint2(a0, a1);
            if (LAYOUT == 1 || LAYOUT == 3) *reinterpret_cast<uint2 *>(out1 + lo) = make_uint2(b0, b1);
            l += lstep;
        }
    }
}

// the byte-by-byte forms: one unit of every row (the fast form's straddling unit: unit >= 0), or every unit (any alignment: unit < 0)
template <int LAYOUT>
__device__ __forceinline__ void s19_stage_bytes(const uint8_t *sp, int stride, int rowBytes, int B0, int gn, int nunit, int unit, unsigned xorv, bool shr6,
                                                unsigned *out0, unsigned *out1, int PP, int tid)
{
    constexpr int NDW = LAYOUT == 0 ? 1 : LAYOUT == 3 ? 4 : 2, UB = 4 * NDW;
    const int per = unit >= 0 ? 1 : nunit;
    for (int it = tid; it < gn * per; it += 256) {
        const int rr = it / per, u0 = unit >= 0 ? unit : it - rr * per;
        unsigned z[NDW], a0, a1, b0, b1;
        const uint8_t *rowp = sp + (size_t)rr * stride;
        for (int i = 0; i < NDW; i++) {
            z[i] = 0;
            for (int q = 0; q < 4; q++)
                if (B0 + u0 * UB + 4 * i + q < rowBytes) z[i] |= (unsigned)rowp[B0 + u0 * UB + 4 * i + q] << (8 * q);   // (past the row: zero — those taps' coefficients are)
        }
        s19_pairs<LAYOUT>(z, xorv, shr6, a0, a1, b0, b1);
        *reinterpret_cast<uint2 *>(out0 + (size_t)rr * PP + 2 * u0) = make_uint2(a0, a1);
        if (LAYOUT == 1 || LAYOUT == 3) *reinterpret_cast<uint2 *>(out1 + (size_t)rr * PP + 2 * u0) = make_uint2(b0, b1);
    }
}

// the instance of a (layout, images) pair: 16 register dwords a lane in all
#define S19_FOR_LAYOUT(CALL)                                                      \
    do {                                                                          \
        if (J.layout == 0) { if (J.nraw == 1) { CALL(0, 16) } else { CALL(0, 8) } } \
        else if (J.layout == 1) { CALL(1, 8) }                                    \
        else if (J.layout == 2) { if (J.nraw == 1) { CALL(2, 8) } else { CALL(2, 4) } } \
        else { CALL(3, 4) }                                                       \
    } while (0)

// pass H of four staged rows (rows past the group's last repeat it) for the lane's column; rc0: the lane's window in the group's first row
template <int NP, bool RC>
__device__ __forceinline__ void s19_hrows(const unsigned *rc0, int PP, int rb, int gn, const int *cf, const int32_t *cfg, int hp, unsigned bias,
                                          int sh, int maxv, int rc, int32_t *lrow, int lane)
{
    int acc[4] = {0, 0, 0, 0};
    int ro[4];
#pragma unroll
    for (int u = 0; u < 4; u++) ro[u] = min(rb + 4 * u, gn - 1);
    if (NP > 0) {
        unsigned s[4][NP > 0 ? NP : 1];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int k = 0; k < (NP > 0 ? NP : 1); k++) s[u][k] = rc0[ro[u] * PP + k];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int k = 0; k < (NP > 0 ? NP : 1); k++) acc[u] = s19_dot2(s[u][k], cf[k], acc[u]);
    } else {
        for (int k = 0; k < hp; k++) {
            const int cc = cfg[k];
#pragma unroll
            for (int u = 0; u < 4; u++) acc[u] = s19_dot2(rc0[ro[u] * PP + k], cc, acc[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        int v = min((int)((unsigned)acc[u] + bias) >> sh, maxv);       // (the sums start at 0: a biased sum stays inside 2^31 while sum |c| < 2^16, plan_job)
        if (RC) v = s19_range(v, rc);
        lrow[ro[u] * kS19TW + lane] = v;
    }
}

// the plane pointers of the item's frame, read from the frame table in the kernel's own body (handing the table itself to the functions below
// by reference made the compiler copy its 1.5 KB into every lane's scratch: 98 KB written a wave, the kernel seven times slower — r06l): src0 / src1 =
// the job's source row images, dst0 / dst1 = its components' destinations, by value

// stage + pass H of ONE job's tile (tx, ty): the tile rows' vertical tables into `vtab`, the 19-bit lines of its components into `lines`
// (kS19TW ints a row, component c from row c * nrLines on); `raw`: the staged rows.  Ends behind a barrier.
template <int NP>
__device__ __forceinline__ void s19_lines(const S19Args &a, int jx, const uint8_t *src0, const uint8_t *src1, int tx, int ty,
                                          int32_t *vtab, unsigned *raw, int32_t *lines)
{
    const S19Job &J = a.job[jx];
    // (the wave's number as a scalar: as tid >> 6 it is a vector register to the compiler, and pass H's row indices, their v_mul_lo_u32 by the row pitch and the
    // loop's bounds were vector work — 35 of 51 VALU instructions a row quad, r06v)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = tx * J.TW, y0 = ty * J.TH;
    const int r0 = J.rowStart[ty], nr = J.rowCount[ty], c0 = J.colStart[tx];
    const int vp = J.v.pairs, PP = J.PP;

    const int nunit = PP >> 1;
    const int sgb = J.layout == 0 ? 1 : J.layout == 3 ? 4 : 2;                  // bytes a sample group (a sample of every component of the row image)
    const int B0 = c0 * sgb, UBrt = 4 * sgb;
    const bool fast = a.srcAl4 != 0 && J.rowBytes >= UBrt;
    const bool shr6 = J.kind == 10;
    // the unit that straddles the row's end (fast form only; block-uniform)
    const int upart = (J.rowBytes > B0 && (J.rowBytes - B0) % UBrt != 0 && (J.rowBytes - B0) / UBrt < nunit) ? (J.rowBytes - B0) / UBrt : -1;
    unsigned pf[16];
#pragma unroll
    for (int i = 0; i < 16; i++) pf[i] = 0;
    // (the two row images of planar chroma take register slots 0-7 and 8-15: the slot base is a template argument — a run-time base made
    // the compiler index the register file through M0)
    auto issue = [&](int g0) {
        const int gn = min(J.G, nr - g0);
        {
            const uint8_t *sp = src0 + (size_t)(r0 + g0) * J.rawStride[0];
#define S19_ISSUE(L_, S_) if (J.cp2) s19_issue<L_, S_, 2, 0>(sp, (unsigned)J.rawStride[0], J.rowBytes, B0, gn, nunit, J.lshift, pf, lane, wave); \
                          else       s19_issue<L_, S_, 1, 0>(sp, (unsigned)J.rawStride[0], J.rowBytes, B0, gn, nunit, J.lshift, pf, lane, wave);
            S19_FOR_LAYOUT(S19_ISSUE);
#undef S19_ISSUE
        }
        if (J.nraw == 2) {
            const uint8_t *sp = src1 + (size_t)(r0 + g0) * J.rawStride[1];
            if (J.layout == 0) { if (J.cp2) s19_issue<0, 8, 2, 8>(sp, (unsigned)J.rawStride[1], J.rowBytes, B0, gn, nunit, J.lshift, pf, lane, wave);
                                 else       s19_issue<0, 8, 1, 8>(sp, (unsigned)J.rawStride[1], J.rowBytes, B0, gn, nunit, J.lshift, pf, lane, wave); }
            else               { if (J.cp2) s19_issue<2, 4, 2, 8>(sp, (unsigned)J.rawStride[1], J.rowBytes, B0, gn, nunit, J.lshift, pf, lane, wave);
                                 else       s19_issue<2, 4, 1, 8>(sp, (unsigned)J.rawStride[1], J.rowBytes, B0, gn, nunit, J.lshift, pf, lane, wave); }
        }
    };
    if (fast && S19_PROBE != 1) issue(0);                                     // the first group's rows: asked for before anything else is

    // the lane's column: first pair of its window in a staged row, coefficient pairs, the bias the staged samples carry
    const int hp = J.h.pairs;
    const int x = x0 + lane;
    const bool xin = lane < J.TW && x < J.dstW;
    const int q0 = xin ? (J.h.pos_even[x] - c0) >> 1 : 0;
    const int32_t *cfg = J.h.packed + (size_t)(xin ? x : 0) * hp;
    // (ONE number of register pairs a launch — the larger job's: per-job instances inside one kernel were tried, r06s: 102 VGPRs against 74, every case 10-20 % slower)
    int cf[NP > 0 ? NP : 1];
#pragma unroll
    for (int k = 0; k < (NP > 0 ? NP : 1); k++) cf[k] = (NP > 0 && xin && k < hp) ? cfg[k] : 0;
    unsigned bias = 0;
    if (J.xorv) {
        int sum = 0;
        if (NP > 0) {
#pragma unroll
            for (int k = 0; k < (NP > 0 ? NP : 1); k++) sum += (int)(short)(cf[k] & 0xFFFF) + (cf[k] >> 16);
        } else if (xin)
            for (int k = 0; k < hp; k++) sum += (int)(short)(cfg[k] & 0xFFFF) + (cfg[k] >> 16);
        bias = (unsigned)sum << 15;
    }
    {   // the vertical tables of the tile's rows (read back in pass V, two barriers from here)
        const int nrow = min(J.TH, J.dstH - y0);
        for (int i = tid; i < nrow * vp; i += 256) {
            const int c = J.v.packed[(size_t)y0 * vp + i];
            vtab[2 * J.TH + 2 * i] = (int)(short)(c & 0xFFFF); vtab[2 * J.TH + 2 * i + 1] = c >> 16;
        }
        // a row's window start and its sums' start value: yuv2planeX_16_c's constant, or (15-bit lines) the bank's own — the dither of
        // yuv2planeX_8_c << 12, 2^16 for the 10-bit writers, the one- and two-tap forms' (YuvScaleTiling::lumRound / chrRound)
        if (tid < nrow) { vtab[tid] = J.v.pos_even[y0 + tid]; vtab[J.TH + tid] = J.outMode ? J.v.round[y0 + tid] : (int)((1u << 14) - 0x40000000u); }
    }

    for (int g0 = 0; g0 < nr; g0 += J.G) {
        const int gn = min(J.G, nr - g0);
        if (g0) __syncthreads();                                               // the previous group's windows have been read
        if (S19_PROBE != 1 && fast) {
            unsigned *o0 = raw, *o1 = raw + (size_t)J.G * PP;
#define S19_COMMIT_(L_, S_, C_, B_, O0_, O1_) do { if (L_ >= 2 && shr6) s19_commit<L_, S_, C_, B_, true>(gn, nunit, J.lshift, J.xorv, O0_, O1_, PP, pf, lane, wave); \
                                                 else                  s19_commit<L_, S_, C_, B_, false>(gn, nunit, J.lshift, J.xorv, O0_, O1_, PP, pf, lane, wave); } while (0)
#define S19_COMMIT(L_, S_) if (J.cp2) S19_COMMIT_(L_, S_, 2, 0, o0, o1); else S19_COMMIT_(L_, S_, 1, 0, o0, o1);
            S19_FOR_LAYOUT(S19_COMMIT);
#undef S19_COMMIT
            if (J.nraw == 2) {
                if (J.layout == 0) { if (J.cp2) S19_COMMIT_(0, 8, 2, 8, o1, o1); else S19_COMMIT_(0, 8, 1, 8, o1, o1); }
                else               { if (J.cp2) S19_COMMIT_(2, 4, 2, 8, o1, o1); else S19_COMMIT_(2, 4, 1, 8, o1, o1); }
            }
#undef S19_COMMIT_
        }
        if (S19_PROBE != 1 && (!fast || upart >= 0)) {
            if (fast) __syncthreads();                                          // after the zeros the fast form left in the straddling unit
            for (int i = 0; i < J.nraw; i++) {
                unsigned *o0 = raw + (size_t)i * J.G * PP, *o1 = raw + (size_t)J.G * PP;
                const uint8_t *sp = (i ? src1 : src0) + (size_t)(r0 + g0) * J.rawStride[i];
#define S19_BYTES(L_, S_) s19_stage_bytes<L_>(sp, J.rawStride[i], J.rowBytes, B0, gn, nunit, fast ? upart : -1, J.xorv, shr6, o0, o1, PP, tid);
                S19_FOR_LAYOUT(S19_BYTES);
#undef S19_BYTES
            }
        }
        __syncthreads();
        if (fast && S19_PROBE != 1 && g0 + J.G < nr) issue(g0 + J.G);          // in flight while this group is filtered
        // pass H: four staged rows a wave at a time (their window reads in flight together)
        for (int c = 0; c < (S19_PROBE == 2 ? 0 : J.ncomp); c++) {
            const unsigned *rc0 = raw + (size_t)c * J.G * PP + q0;
            int32_t *lrow = lines + (size_t)(c * J.nrLines + g0) * kS19TW;
            if (J.rc == 0) for (int rb = wave; rb < gn; rb += 16) s19_hrows<NP, false>(rc0, PP, rb, gn, cf, cfg, xin ? hp : 0, bias, J.sh, J.maxv, 0, lrow, lane);
            else           for (int rb = wave; rb < gn; rb += 16) s19_hrows<NP, true>(rc0, PP, rb, gn, cf, cfg, xin ? hp : 0, bias, J.sh, J.maxv, J.rc, lrow, lane);
        }
    }
    __syncthreads();
}

// one sample from its sum: s19_planes' writers (outMode 0: 16-bit, 1: 8-bit with the deep sources' dither, 2: 10-bit << outShift)
__device__ __forceinline__ unsigned s19_outv(int mode, int osh, int dith, unsigned acc, int xx, int yy, int vplane)
{
    if (mode == 0) return (unsigned)(min(max((int)acc >> 15, -32768), 32767) + 0x8000);
    if (mode == 2) return (unsigned)min(max((int)acc >> 17, 0), 1023) << osh;
    int v = (int)acc;
    if (dith) v += dither_delta(xx + 3 * vplane, yy);
    return (unsigned)clip_u8_shr(v, 19);
}

// pass V of a YUV destination's job: the tile's output rows from its lines.  outMode 0: 16-bit samples (yuv2planeX_16_c / yuv2nv12cX_16_c); the 15-bit
// lines' writers (round 6, second half: the pairs no walker serves — planar <-> semi-planar with a deep end, 4 : 4 : 4 <-> 4 : 2 : 0): 1: 8-bit samples,
// clip_u8((start + sum + dither) >> 19) — yuv2planeX_8_c / yuv2nv12cX_c, the dither of a deep source ff_dither_8x8_128 by (x, y), the V plane's three columns on
// (output.c:400-450, vscale.c:98-101) —, 2: 10-bit samples in 16-bit stores, clip_uintp2((start + sum) >> 17, 10) << outShift (yuv2planeX_10_c; P010: << 6,
// yuv2p010lX_c / cX_c, output.c:459-519)
__device__ __forceinline__ void s19_planes(const S19Args &a, int jx, uint8_t *dst0, uint8_t *dst1, int tx, int ty,
                                           const int32_t *vtab, const int32_t *lines)
{
    const S19Job &J = a.job[jx];
    const int tid = threadIdx.x;
    const int x0 = tx * J.TW, y0 = ty * J.TH, r0 = J.rowStart[ty], vp = J.v.pairs;
    // The rows of `lines` from nr on (a window's padded taps past the plane) hold whatever the LDS held: their coefficients are zero
    // and v_mad_i32_i24 of anything by zero adds zero.
    const int yEnd = min(y0 + J.TH, J.dstH);
    const int mode = J.outMode, osh = J.outShift, dith = J.dither8;
    // one sample from its sum; (xx, yy): its place in the plane (the dither's), vplane: the V plane's three columns
    auto outv = [&](unsigned acc, int xx, int yy, int vplane) -> unsigned { return s19_outv(mode, osh, dith, acc, xx, yy, vplane); };
    if (J.ileave) {
        // a thread: columns (2 cp, 2 cp + 1) of both chroma planes, 32 pairs a row, 8 rows a step; (U, V) pairs stored
        const int cp = tid & 31, xo = x0 + 2 * cp;
        if (xo < J.dstW)
            for (int y = y0 + (tid >> 5); y < yEnd; y += 8) {
                const int32_t *la = lines + (size_t)(vtab[y - y0] - r0) * kS19TW + 2 * cp, *lb = la + (size_t)J.nrLines * kS19TW;
                const int32_t *vc = vtab + 2 * J.TH + (y - y0) * 2 * vp;
                const unsigned k0 = (unsigned)vtab[J.TH + (y - y0)];
                unsigned u0 = k0, u1 = k0, v0 = k0, v1 = k0;
#pragma unroll 2
                for (int k = 0; k < (S19_PROBE == 3 ? 0 : vp); k++) {
                    const int2 cc = *reinterpret_cast<const int2 *>(vc + 2 * k);
                    const int2 a0 = *reinterpret_cast<const int2 *>(la + (2 * k) * kS19TW), a1 = *reinterpret_cast<const int2 *>(la + (2 * k + 1) * kS19TW);
                    const int2 b0 = *reinterpret_cast<const int2 *>(lb + (2 * k) * kS19TW), b1 = *reinterpret_cast<const int2 *>(lb + (2 * k + 1) * kS19TW);
                    u0 = s19_mad(a1.x, cc.y, s19_mad(a0.x, cc.x, u0)); u1 = s19_mad(a1.y, cc.y, s19_mad(a0.y, cc.x, u1));
                    v0 = s19_mad(b1.x, cc.y, s19_mad(b0.x, cc.x, v0)); v1 = s19_mad(b1.y, cc.y, s19_mad(b0.y, cc.x, v1));
                }
                const unsigned pu0 = outv(u0, xo, y, 0), pv0 = outv(v0, xo, y, 1), pu1 = outv(u1, xo + 1, y, 0), pv1 = outv(v1, xo + 1, y, 1);
                const bool two = xo + 1 < J.dstW;
                if (mode == 1) {
                    uint8_t *d = dst0 + J.dstOff[0] + (size_t)y * J.ds[0] + 2 * (size_t)xo;
                    if (a.dstAl4 && two) *reinterpret_cast<unsigned *>(d) = pu0 | pv0 << 8 | pu1 << 16 | pv1 << 24;
                    else { d[0] = (uint8_t)pu0; d[1] = (uint8_t)pv0; if (two) { d[2] = (uint8_t)pu1; d[3] = (uint8_t)pv1; } }
                } else {
                    const unsigned d0 = pu0 | pv0 << 16, d1 = pu1 | pv1 << 16;
                    uint8_t *d = dst0 + J.dstOff[0] + (size_t)y * J.ds[0] + 4 * (size_t)xo;
                    if (a.dstAl4) {
                        reinterpret_cast<unsigned *>(d)[0] = d0;
                        if (two) reinterpret_cast<unsigned *>(d)[1] = d1;
                    } else {
                        unsigned short *d16 = reinterpret_cast<unsigned short *>(d);
                        d16[0] = (unsigned short)d0; d16[1] = (unsigned short)(d0 >> 16);
                        if (two) { d16[2] = (unsigned short)d1; d16[3] = (unsigned short)(d1 >> 16); }
                    }
                }
            }
    } else {
        // a thread: columns 4 q .. 4 q + 3 of a plane, 16 quads a row, 16 rows a step
        const int q = tid & 15, xo = x0 + 4 * q;
        if (xo < J.dstW)
            for (int y = y0 + (tid >> 4); y < yEnd; y += 16) {
                const int32_t *vc = vtab + 2 * J.TH + (y - y0) * 2 * vp;
                const int p0 = vtab[y - y0] - r0;
                const unsigned k0 = (unsigned)vtab[J.TH + (y - y0)];
                for (int c = 0; c < J.ncomp; c++) {
                    const int32_t *la = lines + (size_t)(c * J.nrLines + p0) * kS19TW + 4 * q;
                    unsigned o0 = k0, o1 = k0, o2 = k0, o3 = k0;
#pragma unroll 2
                    for (int k = 0; k < (S19_PROBE == 3 ? 0 : vp); k++) {
                        const int2 cc = *reinterpret_cast<const int2 *>(vc + 2 * k);
                        const int4 a0 = *reinterpret_cast<const int4 *>(la + (2 * k) * kS19TW), a1 = *reinterpret_cast<const int4 *>(la + (2 * k + 1) * kS19TW);
                        o0 = s19_mad(a1.x, cc.y, s19_mad(a0.x, cc.x, o0)); o1 = s19_mad(a1.y, cc.y, s19_mad(a0.y, cc.x, o1));
                        o2 = s19_mad(a1.z, cc.y, s19_mad(a0.z, cc.x, o2)); o3 = s19_mad(a1.w, cc.y, s19_mad(a0.w, cc.x, o3));
                    }
                    const unsigned w0 = outv(o0, xo, y, c), w1 = outv(o1, xo + 1, y, c), w2 = outv(o2, xo + 2, y, c), w3 = outv(o3, xo + 3, y, c);
                    const int n = min(4, J.dstW - xo);
                    if (mode == 1) {
                        uint8_t *d = (c ? dst1 : dst0) + J.dstOff[c] + (size_t)y * J.ds[c] + (size_t)xo;
                        if (a.dstAl4 && n == 4) *reinterpret_cast<unsigned *>(d) = w0 | w1 << 8 | w2 << 16 | w3 << 24;
                        else { d[0] = (uint8_t)w0; if (n > 1) d[1] = (uint8_t)w1; if (n > 2) d[2] = (uint8_t)w2; if (n > 3) d[3] = (uint8_t)w3; }
                    } else {
                        const unsigned d0 = w0 | w1 << 16, d1 = w2 | w3 << 16;
                        uint8_t *d = (c ? dst1 : dst0) + J.dstOff[c] + (size_t)y * J.ds[c] + 2 * (size_t)xo;
                        if (a.dstAl4 && n == 4) { reinterpret_cast<unsigned *>(d)[0] = d0; reinterpret_cast<unsigned *>(d)[1] = d1; }
                        else {
                            unsigned short *d16 = reinterpret_cast<unsigned short *>(d);
                            d16[0] = (unsigned short)d0;
                            if (n > 1) d16[1] = (unsigned short)(d0 >> 16);
                            if (n > 2) d16[2] = (unsigned short)d1;
                            if (n > 3) d16[3] = (unsigned short)(d1 >> 16);
                        }
                    }
                }
            }
    }
}

// yuv2rgba64_X_c's colour stage from the three sums (output.c:1062-1100), one pixel: two dwords.  The five products are 24-bit ones (v_mul_i32_i24 / v_mad_i32_i24, full
// rate; v_mul_lo_u32 is a quarter of it and was 20 of a pixel's 50 issue slots, r06y5): Y, U, V are sums >> 14 — within 2^18 — and the coefficients 16-bit values, so the low 32
// bits of the 24-bit product ARE the 32-bit wrap-around product libswscale computes
__device__ __forceinline__ void s19_rgb64_pixel(const Yuv2RgbConsts &k, int bgr, unsigned ay, unsigned au, unsigned av, unsigned &lo, unsigned &hi)
{
    const int U = (int)au >> 14, V = (int)av >> 14;
    const unsigned Y = s19_mad(((int)ay >> 14) + 0x10000 - k.y_offset, k.y_coeff, 1u << 13);
    const unsigned R = s19_mad(V, k.v2r, Y), G = s19_mad(U, k.u2g, s19_mad(V, k.v2g, Y)), B = s19_mad(U, k.u2b, Y);
    auto ch = [&](unsigned v) -> unsigned { return (unsigned)min(max((int)v, 0), 0x3FFFFFFF) >> 14; };
    const unsigned c0 = ch(bgr ? B : R), c1 = ch(G), c2 = ch(bgr ? R : B);
    lo = c0 | c1 << 16; hi = c2 | 0xFFFF0000u;
}

// pass V + colour stage of an RGBA64LE / BGRA64LE destination: yuv2rgba64_X_c / _full_X_c (output.c:1025-1105, :1275-1337; the 1- and 2-tap forms are
// the same sums on the effective coefficients, as vrgba64_kernel states them):
//   Y = ((-2^30 + sum lum f) >> 14) + 2^16;  U, V = (-(128 << 23) + sum chr f) >> 14     (32-bit wrap-around sums)
//   Y = (Y - y_offset) y_coeff + 2^13;  R = V v2r;  G = V v2g + U u2g;  B = U u2b;  channel = clip_uintp2(X + Y, 30) >> 14;  alpha 0xFFFF
// A thread: pixels (2 cp, 2 cp + 1) of an output row, 32 pairs a row, 8 rows a step; sixteen bytes stored.  vtL / vtC, linesY / linesC: the luma
// job's and the chroma job's tables and lines (chroma columns: x >> chrShift from the chroma tile's first one)
__device__ __forceinline__ void s19_rgb64(const S19Args &a, uint8_t *dst0, int tx, int ty, const int32_t *vtL, const int32_t *vtC,
                                          const int32_t *linesY, const int32_t *linesC)
{
    const S19Job &L = a.job[0], &C = a.job[1];
    const int tid = threadIdx.x, cp = tid & 31;
    const int x0 = tx * L.TW, y0 = ty * L.TH, xo = x0 + 2 * cp;
    const int r0L = L.rowStart[ty], r0C = C.rowStart[ty], vpL = L.v.pairs, vpC = C.v.pairs;
    const int yEnd = min(y0 + L.TH, L.dstH);
    const int cs = a.chrShift, cx = cs ? cp : 2 * cp;                            // the pair's (first) chroma column in the chroma tile
    const Yuv2RgbConsts &k = a.y2r;
    if (xo >= L.dstW) return;
    for (int y = y0 + (tid >> 5); y < yEnd; y += 8) {
        const int32_t *ly = linesY + (vtL[y - y0] - r0L) * kS19TW + 2 * cp;
        const int32_t *lu = linesC + (vtC[y - y0] - r0C) * kS19TW + cx, *lv = lu + C.nrLines * kS19TW;
        const int32_t *cl = vtL + 2 * L.TH + (y - y0) * 2 * vpL, *cc = vtC + 2 * C.TH + (y - y0) * 2 * vpC;
        unsigned ay0 = (unsigned)-0x40000000, ay1 = ay0, au0 = (unsigned)-(128 << 23), au1 = au0, av0 = au0, av1 = au0;
#pragma unroll 2
        for (int t = 0; t < vpL; t++) {
            const int2 c2 = *reinterpret_cast<const int2 *>(cl + 2 * t);
            const int2 p0 = *reinterpret_cast<const int2 *>(ly + (2 * t) * kS19TW), p1 = *reinterpret_cast<const int2 *>(ly + (2 * t + 1) * kS19TW);
            ay0 = s19_mad(p1.x, c2.y, s19_mad(p0.x, c2.x, ay0)); ay1 = s19_mad(p1.y, c2.y, s19_mad(p0.y, c2.x, ay1));
        }
        if (cs) {
#pragma unroll 2
            for (int t = 0; t < vpC; t++) {
                const int2 c2 = *reinterpret_cast<const int2 *>(cc + 2 * t);
                au0 = s19_mad(lu[(2 * t + 1) * kS19TW], c2.y, s19_mad(lu[(2 * t) * kS19TW], c2.x, au0));
                av0 = s19_mad(lv[(2 * t + 1) * kS19TW], c2.y, s19_mad(lv[(2 * t) * kS19TW], c2.x, av0));
            }
            au1 = au0; av1 = av0;
        } else {
#pragma unroll 2
            for (int t = 0; t < vpC; t++) {
                const int2 c2 = *reinterpret_cast<const int2 *>(cc + 2 * t);
                const int2 u0 = *reinterpret_cast<const int2 *>(lu + (2 * t) * kS19TW), u1 = *reinterpret_cast<const int2 *>(lu + (2 * t + 1) * kS19TW);
                const int2 v0 = *reinterpret_cast<const int2 *>(lv + (2 * t) * kS19TW), v1 = *reinterpret_cast<const int2 *>(lv + (2 * t + 1) * kS19TW);
                au0 = s19_mad(u1.x, c2.y, s19_mad(u0.x, c2.x, au0)); au1 = s19_mad(u1.y, c2.y, s19_mad(u0.y, c2.x, au1));
                av0 = s19_mad(v1.x, c2.y, s19_mad(v0.x, c2.x, av0)); av1 = s19_mad(v1.y, c2.y, s19_mad(v0.y, c2.x, av1));
            }
        }
        auto pixel = [&](unsigned ay, unsigned au, unsigned av, unsigned &lo, unsigned &hi) { s19_rgb64_pixel(k, a.rgb64 == 2, ay, au, av, lo, hi); };
        unsigned d[4];
        pixel(ay0, au0, av0, d[0], d[1]);
        pixel(ay1, au1, av1, d[2], d[3]);
        uint8_t *dp = dst0 + (size_t)y * L.ds[0] + 8 * (size_t)xo;
        const bool two = xo + 1 < L.dstW;
        if (a.dstAl4) {
            reinterpret_cast<unsigned *>(dp)[0] = d[0]; reinterpret_cast<unsigned *>(dp)[1] = d[1];
            if (two) { reinterpret_cast<unsigned *>(dp)[2] = d[2]; reinterpret_cast<unsigned *>(dp)[3] = d[3]; }
        } else {
            unsigned short *d16 = reinterpret_cast<unsigned short *>(dp);
            for (int i = 0; i < (two ? 4 : 2); i++) { d16[2 * i] = (unsigned short)d[i]; d16[2 * i + 1] = (unsigned short)(d[i] >> 16); }
        }
    }
}

// the frame table of a ONE-frame launch of the unit forms: 48 bytes of kernel arguments instead of 1.5 KB (what sws_scale() / filter_frame() send: P010 -> NV12 1080p alone
// 5.4 -> 5.0 us, r06y8)
struct S19Frame1 {
    const uint8_t *y[1], *u[1], *v[1];
    uint8_t *dst[1], *dstU[1], *dstV[1];
};

template <int NP>
__global__ __launch_bounds__(256) void scale19_kernel(S19Args a, Yuv2xFrames fr)
{
    HIP_DYNAMIC_SHARED(uint4, lds_base)
    uint8_t *lds = reinterpret_cast<uint8_t *>(lds_base);
    if (S19_PROBE == 4) return;
    // workgroups go round the eight XCDs in launch order (observed, MI355X_MICROARCH.md): each XCD takes one contiguous range of (frame, tile)
    // items, so that the tiles that share source lines — neighbours across, the rows two tile rows overlap in — meet in ONE L2
    int item = blockIdx.x;
    if (a.xcdRemap) {
        const int per = (int)gridDim.x >> 3;
        if (item < per * 8) item = (item & 7) * per + (item >> 3);
    }
    // a YUV destination: the item is a tile of ONE job (the luma plane's, or the two chroma planes').  A packed 64-bit destination: the item is a tile
    // of pixels — the luma job's lines, then the chroma job's (their staged rows share LDS), then the colour stage.  ONE inlined copy of s19_lines
    // serves both (three copies: 171 VGPRs and 1.5 KB of scratch)
    const int nb = a.rgb64 ? a.job[0].nblk : a.job[0].nblk + a.job[1].nblk;
    const int f = item / nb;
    int b = item - f * nb;
    int ji = 0;
    if (!a.rgb64 && b >= a.job[0].nblk) { ji = 1; b -= a.job[0].nblk; }
    const int ntx = a.job[ji].ntx;
    const int tx = b % ntx, ty = b / ntx;
    // LDS: the tile rows' vertical tables (window start, then 2 vp coefficients a row) of the job(s), the staged rows (ncomp x G x PP dwords), the lines
    const int vt0 = a.job[ji].vtBytes;
    int32_t *vtab0 = reinterpret_cast<int32_t *>(lds), *vtab1 = reinterpret_cast<int32_t *>(lds + vt0);
    unsigned *raw = reinterpret_cast<unsigned *>(lds + vt0 + (a.rgb64 ? a.job[1].vtBytes : 0));
    int32_t *lines0 = a.rgb64 ? reinterpret_cast<int32_t *>(lds + a.linesOff) : reinterpret_cast<int32_t *>(raw + (size_t)a.job[ji].ncomp * a.job[ji].G * a.job[ji].PP);
    int32_t *lines1 = lines0 + a.job[0].nrLines * kS19TW;
    auto src_ptr = [&](int sel) -> const uint8_t * { return sel == 0 ? fr.y[f] : sel == 1 ? fr.u[f] : fr.v[f]; };
    auto dst_ptr = [&](int sel) -> uint8_t * { return sel == 0 ? fr.dst[f] : sel == 1 ? fr.dstU[f] : fr.dstV[f]; };
    const int nj = a.rgb64 ? 2 : 1;
    for (int j = 0; j < nj; j++)
        s19_lines<NP>(a, ji + j, src_ptr(a.job[ji + j].rawSel[0]), src_ptr(a.job[ji + j].rawSel[1]), tx, ty, j ? vtab1 : vtab0, raw, j ? lines1 : lines0);
    if (a.rgb64) s19_rgb64(a, fr.dst[f], tx, ty, vtab0, vtab1, lines0, lines1);
    else         s19_planes(a, ji, dst_ptr(a.job[ji].dstSel[0]), dst_ptr(a.job[ji].dstSel[1]), tx, ty, vtab0, lines0);
}

// ---- the unit form: equal size, every bank a one-tap identity --------------------------------------------------------------------------------
// yuv2yuv_cuda's space (libswscale/cuda/yuv2yuv_cuda.cu:324-366: same-size conversions between the 4:2:0 / 4:4:4 layouts and depths) wherever libswscale
// has no unscaled special converter and runs its generic scaler with one-tap filters: a sample's 15- / 19-bit line value is min(s 2^14 >> sh, maxv), the
// range conversion, then the writer's one product and shift — no neighbour, no LDS.  A thread: eight samples of a row of the luma plane, or of both chroma
// planes; 8 / 16 / 32-byte loads and stores where the planes sit on 16-byte addresses and pitches, sample by sample elsewhere and on a row's ragged end.
// S19Job is the tile form's (layouts, plane selectors, pitches, shift / clamp / range / writer); unitCoef / unitRound: the vertical bank's one coefficient and the
// sums' start value (the same for every row: s19_unit_plan checks).
template <int LAYOUT>
__device__ __forceinline__ void s19u_load(const uint8_t *p0, const uint8_t *p1, bool fast, int n, bool shr6, int (&sa)[8], int (&sb)[8])
{
#pragma unroll
    for (int i = 0; i < 8; i++) sa[i] = sb[i] = 0;
    if (fast) {
        if (LAYOUT == 0) {
            const uint2 v = *reinterpret_cast<const uint2 *>(p0);
            const unsigned w[2] = {v.x, v.y};
#pragma unroll
            for (int i = 0; i < 8; i++) sa[i] = (int)(w[i >> 2] >> (8 * (i & 3)) & 0xFFu);
            if (p1) {
                const uint2 q = *reinterpret_cast<const uint2 *>(p1);
                const unsigned z[2] = {q.x, q.y};
#pragma unroll
                for (int i = 0; i < 8; i++) sb[i] = (int)(z[i >> 2] >> (8 * (i & 3)) & 0xFFu);
            }
        } else if (LAYOUT == 1) {
            const uint4 v = *reinterpret_cast<const uint4 *>(p0);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 8; i++) { sa[i] = (int)(w[i >> 1] >> (16 * (i & 1)) & 0xFFu); sb[i] = (int)(w[i >> 1] >> (16 * (i & 1) + 8) & 0xFFu); }
        } else if (LAYOUT == 2) {
            const uint4 v = *reinterpret_cast<const uint4 *>(p0);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 8; i++) sa[i] = (int)(w[i >> 1] >> (16 * (i & 1)) & 0xFFFFu);
            if (p1) {
                const uint4 q = *reinterpret_cast<const uint4 *>(p1);
                const unsigned z[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int i = 0; i < 8; i++) sb[i] = (int)(z[i >> 1] >> (16 * (i & 1)) & 0xFFFFu);
            }
        } else {
            const uint4 v0 = reinterpret_cast<const uint4 *>(p0)[0], v1 = reinterpret_cast<const uint4 *>(p0)[1];
            const unsigned w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int i = 0; i < 8; i++) { sa[i] = (int)(w[i] & 0xFFFFu); sb[i] = (int)(w[i] >> 16); }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (i < n) {                                                        // (unrolled: a run-time index into the arrays would put them in scratch)
                if (LAYOUT == 0)      { sa[i] = p0[i]; if (p1) sb[i] = p1[i]; }
                else if (LAYOUT == 1) { sa[i] = p0[2 * i]; sb[i] = p0[2 * i + 1]; }
                else if (LAYOUT == 2) { sa[i] = reinterpret_cast<const unsigned short *>(p0)[i]; if (p1) sb[i] = reinterpret_cast<const unsigned short *>(p1)[i]; }
                else                  { sa[i] = reinterpret_cast<const unsigned short *>(p0)[2 * i]; sb[i] = reinterpret_cast<const unsigned short *>(p0)[2 * i + 1]; }
            }
    }
    if (LAYOUT >= 2 && shr6) {
#pragma unroll
        for (int i = 0; i < 8; i++) { sa[i] >>= 6; sb[i] >>= 6; }
    }
}

// MODE: the job's outMode, RC: a range conversion of the line values — template arguments: as run-time values every sample paid their branches (6.2 -> 3.3 us a 1080p
// frame with them in the loop, r06y; the instruction count, not the bytes, was the bound)
template <int MODE, bool RC>
__device__ __forceinline__ void s19u_samples(const S19Job &J, const int (&s)[8], int y, int vplane, unsigned (&o)[8])
{
    // min(s 2^14 >> sh, maxv): one shift either way
    const int shl = J.sh <= 14 ? 14 - J.sh : 0, shr = J.sh > 14 ? J.sh - 14 : 0;
    const int osh = J.outShift, cv = J.unitCoef, dmask = J.dither8 ? -1 : 0;
    const unsigned k0 = (unsigned)J.unitRound;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int l = min((s[i] << shl) >> shr, J.maxv);
        if (RC) l = s19_range(l, J.rc);
        const unsigned acc = s19_mad(l, cv, k0);
        // (a thread's first column is a multiple of 8: the dither's column is the sample's index — a constant the compiler folds; the deep sources' flag as a mask, not a branch)
        if (MODE == 1) o[i] = (unsigned)clip_u8_shr((int)acc + (dither_delta(i + 3 * vplane, y) & dmask), 19);
        else           o[i] = s19_outv(MODE, osh, 0, acc, 0, 0, 0);
    }
}

template <class FR>
__global__ __launch_bounds__(256) void scale19_unit_kernel(S19Args a, FR fr)
{
    const int nb = a.unitBlk[0] + a.unitBlk[1];
    const int f = blockIdx.x / nb;
    int b = blockIdx.x - f * nb, ji = 0;
    if (b >= a.unitBlk[0]) { ji = 1; b -= a.unitBlk[0]; }
    const S19Job &J = a.job[ji];
    const int upr = (J.dstW + 7) >> 3;
    const int idx = b * 256 + (int)threadIdx.x;
    // idx / upr by the host's multiplier (exact for idx < 2^31: launch_scale19): the compiler's division is thirty instructions of a thread that has a hundred
    const int y = upr == 1 ? idx : (int)(__umulhi((unsigned)idx, a.unitMul[ji]) >> a.unitShr[ji]), x0 = (idx - y * upr) * 8;
    if (y >= J.dstH) return;
    const int n = min(8, J.dstW - x0);
    auto src_ptr = [&](int sel) -> const uint8_t * { return sel == 0 ? fr.y[f] : sel == 1 ? fr.u[f] : fr.v[f]; };
    auto dst_ptr = [&](int sel) -> uint8_t * { return sel == 0 ? fr.dst[f] : sel == 1 ? fr.dstU[f] : fr.dstV[f]; };
    const int sgb = J.layout == 0 ? 1 : J.layout == 3 ? 4 : 2;                  // bytes a sample group of a source row image
    // (32-bit offsets from the plane pointers: a plane spans less than 4 GB)
    const uint8_t *p0 = src_ptr(J.rawSel[0]) + ((unsigned)y * (unsigned)J.rawStride[0] + (unsigned)(x0 * sgb));
    const uint8_t *p1 = (J.ncomp == 2 && J.nraw == 2) ? src_ptr(J.rawSel[1]) + ((unsigned)y * (unsigned)J.rawStride[1] + (unsigned)(x0 * sgb)) : nullptr;
    const bool fastS = a.srcAl16 && n == 8, fastD = a.dstAl16 && n == 8;
    int sa[8], sb[8];
    if (J.layout == 0)      s19u_load<0>(p0, p1, fastS, n, false, sa, sb);
    else if (J.layout == 1) s19u_load<1>(p0, p1, fastS, n, false, sa, sb);
    else if (J.layout == 2) s19u_load<2>(p0, p1, fastS, n, J.kind == 10, sa, sb);
    else                    s19u_load<3>(p0, p1, fastS, n, J.kind == 10, sa, sb);
    const int mode = J.outMode;
    unsigned oa[8], ob[8];
#pragma unroll
    for (int i = 0; i < 8; i++) ob[i] = 0;
#define S19U_SAMPLES(M_, R_) do { s19u_samples<M_, R_>(J, sa, y, 0, oa); if (J.ncomp == 2) s19u_samples<M_, R_>(J, sb, y, 1, ob); } while (0)
    if (J.rc) { if (mode == 0) S19U_SAMPLES(0, true); else if (mode == 1) S19U_SAMPLES(1, true); else S19U_SAMPLES(2, true); }
    else      { if (mode == 0) S19U_SAMPLES(0, false); else if (mode == 1) S19U_SAMPLES(1, false); else S19U_SAMPLES(2, false); }
#undef S19U_SAMPLES
    const int ob8 = mode == 1 ? 1 : 2;                                           // bytes an output sample
    if (J.ileave) {
        uint8_t *d = dst_ptr(J.dstSel[0]) + ((unsigned)J.dstOff[0] + (unsigned)y * (unsigned)J.ds[0] + (unsigned)(x0 * 2 * ob8));
        if (mode == 1) {
            if (fastD) {
                unsigned w[4];
#pragma unroll
                for (int k = 0; k < 4; k++) w[k] = oa[2 * k] | ob[2 * k] << 8 | oa[2 * k + 1] << 16 | ob[2 * k + 1] << 24;
                *reinterpret_cast<uint4 *>(d) = make_uint4(w[0], w[1], w[2], w[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) if (i < n) { d[2 * i] = (uint8_t)oa[i]; d[2 * i + 1] = (uint8_t)ob[i]; }
            }
        } else {
            if (fastD) {
                reinterpret_cast<uint4 *>(d)[0] = make_uint4(oa[0] | ob[0] << 16, oa[1] | ob[1] << 16, oa[2] | ob[2] << 16, oa[3] | ob[3] << 16);
                reinterpret_cast<uint4 *>(d)[1] = make_uint4(oa[4] | ob[4] << 16, oa[5] | ob[5] << 16, oa[6] | ob[6] << 16, oa[7] | ob[7] << 16);
            } else {
                unsigned short *d16 = reinterpret_cast<unsigned short *>(d);
#pragma unroll
                for (int i = 0; i < 8; i++) if (i < n) { d16[2 * i] = (unsigned short)oa[i]; d16[2 * i + 1] = (unsigned short)ob[i]; }
            }
        }
    } else {
        auto store = [&](int c, const unsigned (&o)[8]) {
            uint8_t *d = dst_ptr(J.dstSel[c]) + ((unsigned)J.dstOff[c] + (unsigned)y * (unsigned)J.ds[c] + (unsigned)(x0 * ob8));
            if (mode == 1) {
                if (fastD) *reinterpret_cast<uint2 *>(d) = make_uint2(o[0] | o[1] << 8 | o[2] << 16 | o[3] << 24, o[4] | o[5] << 8 | o[6] << 16 | o[7] << 24);
                else {
#pragma unroll
                    for (int i = 0; i < 8; i++) if (i < n) d[i] = (uint8_t)o[i];
                }
            } else {
                if (fastD) *reinterpret_cast<uint4 *>(d) = make_uint4(o[0] | o[1] << 16, o[2] | o[3] << 16, o[4] | o[5] << 16, o[6] | o[7] << 16);
                else {
                    unsigned short *d16 = reinterpret_cast<unsigned short *>(d);
#pragma unroll
                    for (int i = 0; i < 8; i++) if (i < n) d16[i] = (unsigned short)o[i];
                }
            }
        };
        store(0, oa);
        if (J.ncomp == 2) store(1, ob);
    }
}

// ---- the unit form of a packed 64-bit destination: yuv2rgb_cuda's RGBA64 / BGRA64 outputs at equal size (yuv2rgb_cuda.cu:862-907) ---------------------------
// libswscale has no unscaled converter into the 64-bit formats: the generic path with identity horizontal banks, an identity vertical luma bank and the chroma's
// vertical filter (4 : 2 : 0 rows interpolated: four taps under bicubic, one under point).  A thread: eight pixels of a row — their luma samples, per chroma tap
// the four (4 : 4 : 4: eight) chroma samples under them straight from the tap's row, yuv2rgba64_X_c's sums and colour stage as s19_rgb64 states them, 64 bytes
// stored.  Only for planes and pitches on 16-byte addresses and widths of whole units (launch_scale19: anything else is the tile form's).
// LC: the chroma job's layout (0 / 2: planar 8- / 16-bit, 1 / 3: interleaved); CS: chroma columns = pixel columns >> CS
template <int LC, int CS>
__device__ __forceinline__ void s19u64_thread(const S19Args &a, const uint8_t *sy, const uint8_t *su, const uint8_t *sv, int x0, int y, unsigned (&d)[16])
{
    const S19Job &L = a.job[0], &C = a.job[1];
    constexpr int NC = CS ? 4 : 8;                                              // chroma samples of a component under the thread's pixels
    const bool shr6 = L.kind == 10;
    const int shl = L.sh <= 14 ? 14 - L.sh : 0, shr = L.sh > 14 ? L.sh - 14 : 0, maxv = L.maxv;
    // luma: eight samples, one line value each
    unsigned ay[8];
    {
        int sa[8], sb[8];
        if (L.layout == 0) s19u_load<0>(sy + ((unsigned)y * (unsigned)L.rawStride[0] + (unsigned)x0), nullptr, true, 8, false, sa, sb);
        else               s19u_load<2>(sy + ((unsigned)y * (unsigned)L.rawStride[0] + (unsigned)(2 * x0)), nullptr, true, 8, shr6, sa, sb);
#pragma unroll
        for (int i = 0; i < 8; i++) ay[i] = s19_mad(min((sa[i] << shl) >> shr, maxv), L.unitCoef, (unsigned)-0x40000000);
    }
    // chroma: the row's taps
    unsigned au[NC], av[NC];
#pragma unroll
    for (int j = 0; j < NC; j++) au[j] = av[j] = (unsigned)-(128 << 23);
    const int vp = C.v.pairs, pos = C.v.pos_even[y];
    const int cx = CS ? x0 >> 1 : x0;
    constexpr int csb = LC == 0 ? 1 : LC == 3 ? 4 : 2;                          // bytes a chroma sample group
    for (int t = 0; t < 2 * vp; t++) {
        const int pk = C.v.packed[(size_t)y * vp + (t >> 1)];
        const int cf = (t & 1) ? pk >> 16 : (int)(short)(pk & 0xFFFF);
        const int rc = min(pos + t, C.srcH - 1);                                // (a padded tap past the plane: its coefficient is zero)
        int cu[8], cv[8];
        if (CS) {
            // four samples a component: half a unit — 4 / 8 / 8 / 16 bytes
            unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0;
            if (LC == 0)      { r0 = *reinterpret_cast<const unsigned *>(su + ((unsigned)rc * (unsigned)C.rawStride[0] + (unsigned)cx));
                                r1 = *reinterpret_cast<const unsigned *>(sv + ((unsigned)rc * (unsigned)C.rawStride[1] + (unsigned)cx)); }
            else if (LC == 1) { const uint2 q = *reinterpret_cast<const uint2 *>(su + ((unsigned)rc * (unsigned)C.rawStride[0] + (unsigned)(2 * cx))); r0 = q.x; r1 = q.y; }
            else if (LC == 2) { const uint2 q = *reinterpret_cast<const uint2 *>(su + ((unsigned)rc * (unsigned)C.rawStride[0] + (unsigned)(2 * cx)));
                                const uint2 w = *reinterpret_cast<const uint2 *>(sv + ((unsigned)rc * (unsigned)C.rawStride[1] + (unsigned)(2 * cx))); r0 = q.x; r1 = q.y; r2 = w.x; r3 = w.y; }
            else              { const uint4 q = *reinterpret_cast<const uint4 *>(su + ((unsigned)rc * (unsigned)C.rawStride[0] + (unsigned)(4 * cx))); r0 = q.x; r1 = q.y; r2 = q.z; r3 = q.w; }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (LC == 0)      { cu[j] = (int)(r0 >> (8 * j) & 0xFFu); cv[j] = (int)(r1 >> (8 * j) & 0xFFu); }
                else if (LC == 1) { const unsigned w = j < 2 ? r0 : r1; cu[j] = (int)(w >> (16 * (j & 1)) & 0xFFu); cv[j] = (int)(w >> (16 * (j & 1) + 8) & 0xFFu); }
                else if (LC == 2) { cu[j] = (int)((j < 2 ? r0 : r1) >> (16 * (j & 1)) & 0xFFFFu); cv[j] = (int)((j < 2 ? r2 : r3) >> (16 * (j & 1)) & 0xFFFFu); }
                else              { const unsigned w = j == 0 ? r0 : j == 1 ? r1 : j == 2 ? r2 : r3; cu[j] = (int)(w & 0xFFFFu); cv[j] = (int)(w >> 16); }
                if (LC >= 2 && shr6) { cu[j] >>= 6; cv[j] >>= 6; }
            }
        } else {
            s19u_load<LC>(su + ((unsigned)rc * (unsigned)C.rawStride[0] + (unsigned)(cx * csb)), sv + ((unsigned)rc * (unsigned)C.rawStride[1] + (unsigned)(cx * csb)), true, 8,
                          LC >= 2 && shr6, cu, cv);
        }
#pragma unroll
        for (int j = 0; j < NC; j++) {
            au[j] = s19_mad(min((cu[j] << shl) >> shr, maxv), cf, au[j]);
            av[j] = s19_mad(min((cv[j] << shl) >> shr, maxv), cf, av[j]);
        }
    }
    const Yuv2RgbConsts &k = a.y2r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int j = CS ? i >> 1 : i;
        s19_rgb64_pixel(k, a.rgb64 == 2, ay[i], au[j], av[j], d[2 * i], d[2 * i + 1]);
    }
}

template <class FR>
__global__ __launch_bounds__(256) void scale19_unit64_kernel(S19Args a, FR fr)
{
    // a thread's eight pixels are 64 bytes: stored as they are, a wave's store instruction would write 16 bytes of every 64 over 4 KB (7.1 us a 1080p frame, 0.35 of the
    // roofline, whatever the arithmetic cost: r06y5 / r06y6).  The pixels go through LDS instead, and store q of a wave writes dwords [256 q, 256 q + 256) of the wave's
    // 4 KB — whole lines; the units of a wave are consecutive ones of the frame, their rows found again per store
    __shared__ uint4 xch[256 * 4];
    const int nb = a.unitBlk[0];
    const int f = blockIdx.x / nb, b = blockIdx.x - f * nb;
    const S19Job &L = a.job[0], &C = a.job[1];
    const int upr = L.dstW >> 3;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = b * 256 + tid;
    auto row_of = [&](int i) -> int { return upr == 1 ? i : (int)(__umulhi((unsigned)i, a.unitMul[0]) >> a.unitShr[0]); };
    const int y = row_of(idx), x0 = (idx - y * upr) * 8;
    unsigned d[16];
#pragma unroll
    for (int i = 0; i < 16; i++) d[i] = 0;
    if (y < L.dstH) {
        const uint8_t *sy = fr.y[f], *su = fr.u[f], *sv = fr.v[f];
        if (a.chrShift) {
            if (C.layout == 0)      s19u64_thread<0, 1>(a, sy, su, sv, x0, y, d);
            else if (C.layout == 1) s19u64_thread<1, 1>(a, sy, su, sv, x0, y, d);
            else if (C.layout == 2) s19u64_thread<2, 1>(a, sy, su, sv, x0, y, d);
            else                    s19u64_thread<3, 1>(a, sy, su, sv, x0, y, d);
        } else {
            if (C.layout == 0)      s19u64_thread<0, 0>(a, sy, su, sv, x0, y, d);
            else if (C.layout == 2) s19u64_thread<2, 0>(a, sy, su, sv, x0, y, d);
        }
    }
    uint4 *mine = xch + wave * 256;
#pragma unroll
    for (int q = 0; q < 4; q++) mine[lane * 4 + q] = make_uint4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
    __syncthreads();
    uint8_t *dst = fr.dst[f];
    const int wbase = b * 256 + wave * 64;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int e = q * 64 + lane, ts = e >> 2, slot = e & 3;               // the wave's dword-quad e: unit ts of the wave, its pixels 2 slot, 2 slot + 1
        const int is = wbase + ts, ys = row_of(is), xs = (is - ys * upr) * 8;
        if (ys < L.dstH) *reinterpret_cast<uint4 *>(dst + ((unsigned)ys * (unsigned)L.ds[0] + (unsigned)(8 * xs + 16 * slot))) = mine[e];
    }
}

// ---- 16-bit 4:2:0 sources into packed 8-bit RGB at equal size (kernels.h UnitRgbArgs) -------------------------------------------------------------------------
// A thread: eight pixels of a row.  Luma: Y = (roundL + line coefL) >> 19; chroma: per tap of the row's vertical bank the four (U, V) samples under the pixels from
// the tap's row, U / V = clip_u8((round + sum) >> 19); the pixel from yuv2rgb.c's tables in closed form (chroma_terms / luma_chan).  PX bytes a pixel; the wave's
// pixels go through LDS and leave as 8-byte pieces in address order (whole lines a store, as scale19_unit64_kernel's)
// SRC: 0 planar chroma, 1 interleaved (P016LE), 2 interleaved with ten bits in the high end (P010LE) — template arguments: as run-time flags their branches sat around every
// tap and sample (P010 6.6 us a 1080p frame against the planar sources' 5.4, r06y10)
template <int PX, int SRC, class FR>
__global__ __launch_bounds__(256) void unit_rgb_kernel(UnitRgbArgs a, FR fr)
{
    __shared__ uint2 xch[256 * PX];
    const int f = blockIdx.x / a.blocks, b = blockIdx.x - f * a.blocks;
    const int upr = a.w >> 3;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = b * 256 + tid;
    auto row_of = [&](int i) -> int { return upr == 1 ? i : (int)(__umulhi((unsigned)i, a.rowMul) >> a.rowShr); };
    const int y = row_of(idx), x0 = (idx - y * upr) * 8;
    unsigned d[2 * PX];
#pragma unroll
    for (int i = 0; i < 2 * PX; i++) d[i] = 0;
    if (y < a.h) {
        const uint8_t *sy = fr.y[f], *su = fr.u[f], *sv = fr.v[f];
        int ly[8];
        {
            int sb[8];
            s19u_load<2>(sy + ((unsigned)y * (unsigned)a.ys + (unsigned)(2 * x0)), nullptr, true, 8, SRC == 2, ly, sb);
        }
        int U[4], V[4];
        const int vp = a.vChr.pairs, pos = a.vChr.pos_even[y], rnd = a.vChr.round[y];
#pragma unroll
        for (int j = 0; j < 4; j++) U[j] = V[j] = rnd;
        const int cx = x0 >> 1;
        for (int t = 0; t < 2 * vp; t++) {
            const int pk = a.vChr.packed[(size_t)y * vp + (t >> 1)];
            const int cf = (t & 1) ? pk >> 16 : (int)(short)(pk & 0xFFFF);
            if (cf == 0) continue;                                              // (the pairs' padding, a bank's zero taps: a third of a bicubic bank's slots at equal size)
            const int rc = min(pos + t, a.chrH - 1);                            // (a padded tap past the plane: its coefficient is zero)
            unsigned r0, r1, r2, r3;
            if (SRC) { const uint4 q = *reinterpret_cast<const uint4 *>(su + ((unsigned)rc * (unsigned)a.us + (unsigned)(4 * cx))); r0 = q.x; r1 = q.y; r2 = q.z; r3 = q.w; }
            else { const uint2 q = *reinterpret_cast<const uint2 *>(su + ((unsigned)rc * (unsigned)a.us + (unsigned)(2 * cx)));
                   const uint2 w = *reinterpret_cast<const uint2 *>(sv + ((unsigned)rc * (unsigned)a.vs + (unsigned)(2 * cx))); r0 = q.x; r1 = q.y; r2 = w.x; r3 = w.y; }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int cu, cv;
                if (SRC) { const unsigned w = j == 0 ? r0 : j == 1 ? r1 : j == 2 ? r2 : r3; cu = (int)(w & 0xFFFFu); cv = (int)(w >> 16); }
                else        { cu = (int)((j < 2 ? r0 : r1) >> (16 * (j & 1)) & 0xFFFFu); cv = (int)((j < 2 ? r2 : r3) >> (16 * (j & 1)) & 0xFFFFu); }
                if (SRC == 2) { cu >>= 6; cv >>= 6; }
                U[j] = (int)s19_mad(min((cu << a.shl) >> a.shr, a.maxv), cf, (unsigned)U[j]);
                V[j] = (int)s19_mad(min((cv << a.shl) >> a.shr, a.maxv), cf, (unsigned)V[j]);
            }
        }
        unsigned px[8];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // table_rV / gU / gV / bU are indexed with av_clip_uint8 (yuv2rgb.c:737-760)
            const ChromaTerms t = chroma_terms(a.y2r, clip_u8_shr(U[j], 19), clip_u8_shr(V[j], 19));
#pragma unroll
            for (int i = 2 * j; i < 2 * j + 2; i++) {
                const int Y = (int)s19_mad(min((ly[i] << a.shl) >> a.shr, a.maxv), a.coefL, (unsigned)a.roundL) >> 19;
                const int ya = m24(Y, a.y2r.cy);
                const unsigned r = (unsigned)luma_chan(t.r, ya), g = (unsigned)luma_chan(t.g, ya), bl = (unsigned)luma_chan(t.b, ya);
                px[i] = a.bgr ? (bl | g << 8 | r << 16) : (r | g << 8 | bl << 16);
            }
        }
        if (PX == 4) {
#pragma unroll
            for (int i = 0; i < 8; i++) d[i] = px[i] | 0xFF000000u;
        } else {
            // eight 3-byte pixels in six dwords
            d[0] = px[0] | px[1] << 24;          d[1] = px[1] >> 8 | px[2] << 16;     d[2] = px[2] >> 16 | px[3] << 8;
            d[3] = px[4] | px[5] << 24;          d[4] = px[5] >> 8 | px[6] << 16;     d[5] = px[6] >> 16 | px[7] << 8;
        }
    }
    uint2 *mine = xch + wave * 64 * PX;
#pragma unroll
    for (int q = 0; q < PX; q++) mine[lane * PX + q] = make_uint2(d[2 * q], d[2 * q + 1]);
    __syncthreads();
    uint8_t *dst = fr.dst[f];
    const int wbase = b * 256 + wave * 64;
#pragma unroll
    for (int q = 0; q < PX; q++) {
        const int e = q * 64 + lane, ts = e / PX, slot = e - ts * PX;          // the wave's 8-byte piece e: unit ts of the wave, its piece `slot`
        const int is = wbase + ts, ys = row_of(is), xs = (is - ys * upr) * 8;
        if (ys < a.h) *reinterpret_cast<uint2 *>(dst + ((unsigned)ys * (unsigned)a.ds + (unsigned)(PX * xs + 8 * slot))) = mine[e];
    }
}

// ---- the planner (host) ---------------------------------------------------------------------------------------------------------------------
// Refuses (ENOSYS: the two passes take the context) a bank whose padded taps past the plane are not all zero or whose coefficients could
// carry a biased sum past 2^31, and rows of more than 128 staging units.
int check_banks(const FilterBank &h, const FilterBank &v, int srcW, int srcH, int dstW, int dstH)
{
    if (h.count != dstW || v.count != dstH || h.pairs < 1 || v.pairs < 1) return GMAT_ERR(EINVAL);
    for (int x = 0; x < dstW; x++) {
        long sumAbs = 0;
        for (int k = 0; k < h.pairs; k++) {
            const int32_t c = h.packed[(size_t)x * h.pairs + k];
            const int c0 = (int16_t)(c & 0xFFFF), c1 = c >> 16;
            sumAbs += std::abs(c0) + std::abs(c1);
            if ((c0 && h.pos_even[x] + 2 * k >= srcW) || (c1 && h.pos_even[x] + 2 * k + 1 >= srcW)) return GMAT_ERR(ENOSYS);
        }
        if (sumAbs > 65535 || h.pos_even[x] < 0 || (h.pos_even[x] & 1)) return GMAT_ERR(ENOSYS);
    }
    for (int y = 0; y < dstH; y++) {
        if (v.pos_even[y] < 0 || v.pos_even[y] >= srcH) return GMAT_ERR(EINVAL);
        for (int k = 0; k < v.pairs; k++) {
            const int32_t c = v.packed[(size_t)y * v.pairs + k];
            if (((c & 0xFFFF) && v.pos_even[y] + 2 * k >= srcH) || ((c >> 16) && v.pos_even[y] + 2 * k + 1 >= srcH)) return GMAT_ERR(ENOSYS);
        }
    }
    return 0;
}

// the tile columns of a job (TW output columns a tile): the first staged sample of each, the sample pairs a staged row holds, how the stage's
// lanes share a row; returns the rows one s19_issue covers
int plan_cols(const FilterBank &h, int dstW, int TW, int nraw, int layout, int hpairs, S19Job &J, std::vector<int32_t> &colStart)
{
    J.TW = TW;
    J.ntx = (dstW + TW - 1) / TW;
    colStart.assign(J.ntx, 0);
    long PP = 4;
    for (int tx = 0; tx < J.ntx; tx++) {
        long lo = LONG_MAX, hi = 0;
        for (int x = tx * TW; x < std::min(dstW, (tx + 1) * TW); x++) {
            lo = std::min(lo, (long)h.pos_even[x]);
            hi = std::max(hi, (long)h.pos_even[x] + 2 * hpairs);
        }
        const long c0 = lo & ~3L;
        colStart[tx] = (int32_t)c0;
        PP = std::max(PP, ((hi - c0 + 7) >> 3) * 4);                           // whole units of four samples = two pairs; rows of whole 16 bytes
    }
    if (PP / 2 > 128) return GMAT_ERR(ENOSYS);                                 // (a lane stages at most two units of a row)
    J.PP = (int)PP;
    J.lshift = 0;
    while ((1 << J.lshift) < PP / 2 && J.lshift < 6) J.lshift++;              // lanes a staged row's units take (a power of two)
    J.cp2 = PP / 2 > 64;
    // rows one s19_issue covers: 16 register dwords a lane = slots of a unit, over the row images and the column passes
    const int ndw = layout == 0 ? 1 : layout == 3 ? 4 : 2;
    return 4 * (64 >> J.lshift) * std::max(1, 16 / (ndw * nraw) / (J.cp2 ? 2 : 1));
}

// the tile rows of a job at TH output rows a tile
void plan_rows(const FilterBank &v, int srcH, int dstH, int TH, std::vector<int32_t> &rs, std::vector<int32_t> &rc, int &nrMax, int &nrLines)
{
    const int nty = (dstH + TH - 1) / TH;
    rs.assign(nty, 0); rc.assign(nty, 0);
    nrMax = 1; nrLines = 1;
    for (int ty = 0; ty < nty; ty++) {
        int lo = INT_MAX, hi = 0;
        for (int y = ty * TH; y < std::min(dstH, (ty + 1) * TH); y++) {
            lo = std::min(lo, v.pos_even[y]);
            hi = std::max(hi, v.pos_even[y] + 2 * v.pairs);
        }
        nrLines = std::max(nrLines, hi - lo);
        hi = std::min(hi, srcH);
        rs[ty] = lo; rc[ty] = hi - lo;
        nrMax = std::max(nrMax, hi - lo);
    }
}

// rows staged at once: all of a tile's when they fit a third of the budget (at least 4 KB) and one s19_issue, else as many (a multiple of 4, the waves)
int plan_group(const S19Job &J, int ncomp, int nrMax, int gcap, int budget)
{
    const long rawBudget = std::max(4096L, (long)budget / 3), rowB = (long)ncomp * J.PP * 4;
    int G = (int)std::min<long>(std::min(nrMax, gcap), rawBudget / rowB);
    if (G < nrMax && G >= 4) G &= ~3;
    return G;
}

static const int kS19TH[] = {64, 48, 40, 32, 24, 16, 12, 8, 4, 2, 1};

// one job of a YUV destination on its own: the tallest tile the LDS budget holds; returns its LDS bytes
int plan_job(const FilterBank &h, const FilterBank &v, int srcW, int srcH, int dstW, int dstH, int ncomp, int nraw, int layout, int hpairs, int budget, int thCap,
             S19Job &J, std::vector<int32_t> &colStart, std::vector<int32_t> &rowStart, std::vector<int32_t> &rowCount)
{
    if (int r = check_banks(h, v, srcW, srcH, dstW, dstH); r < 0) return r;
    const int gcap = plan_cols(h, dstW, kS19TW, nraw, layout, hpairs, J, colStart);
    if (gcap < 0) return gcap;
    for (int TH : kS19TH) {
        if (TH > thCap) continue;
        std::vector<int32_t> rs, rc;
        int nrMax, nrLines;
        plan_rows(v, srcH, dstH, TH, rs, rc, nrMax, nrLines);
        const long linesBytes = (long)ncomp * nrLines * kS19TW * 4, vtBytes = ((long)TH * (2 * v.pairs + 2) * 4 + 15) & ~15L;
        const int G = plan_group(J, ncomp, nrMax, gcap, budget);
        if (G < 1) { if (TH == 1) return GMAT_ERR(ENOSYS); continue; }
        const long total = vtBytes + (long)G * ncomp * J.PP * 4 + linesBytes;
        if (total > budget && TH > 1) continue;
        if (total > 65536) return GMAT_ERR(ENOSYS);
        J.TH = TH; J.nty = (int)rs.size(); J.nblk = J.ntx * J.nty; J.nrMax = nrMax; J.nrLines = nrLines; J.G = G; J.vtBytes = (int)vtBytes;
        rowStart.swap(rs); rowCount.swap(rc);
        return (int)((total + 15) & ~15L);
    }
    return GMAT_ERR(ENOSYS);
}

} // namespace

// rgb64: 0 a YUV destination (two jobs of their own tiles); 1 RGBA64LE, 2 BGRA64LE (one grid of 64-column tiles: the luma job's lines, then the chroma job's —
// 64 >> chrShift columns — behind the same staged-row LDS, then the colour stage); chrShift: 1 = one chroma sample a pixel pair (ignored for YUV destinations)
static int s19_prepare_budget(const ScalePlan &p, const FilterBank &vl, const FilterBank &vc, int bps, int kind, int srcSemi, int dstSemi, int rgb64, int chrShift, S19Tables &t,
                              int outMode, int outShift, int hsh, int budget, int thCap)
{
    t.ok = 0;
    // hScale8To19_c: 3; hScale16To19_c: depth - 5 (kind 14: 9); the 15-bit lines (outMode != 0): hScale8To15_c's 7 / hScale16To15_c's depth - 1, the caller's
    const int sh = outMode ? hsh : kind == 0 ? 3 : kind % 100 - 5;
    const unsigned xorv = (bps == 2 && kind != 10) ? 0x80008000u : 0u;            // 16-bit samples as v_dot2_i32_i16 takes them (P010's ten bits fit as they are)
    S19Job &L = t.job[0], &C = t.job[1];
    std::memset(&L, 0, sizeof(L)); std::memset(&C, 0, sizeof(C));
    L.ncomp = 1; L.nraw = 1; L.ileave = 0; L.layout = bps == 2 ? 2 : 0;
    L.rawSel[0] = 0;
    L.kind = kind; L.xorv = xorv; L.rowBytes = p.srcW * bps;
    L.srcW = p.srcW; L.srcH = p.srcH; L.dstW = p.dstW; L.dstH = p.dstH;
    L.dstSel[0] = 0; L.dstOff[0] = 0;
    L.sh = sh; L.maxv = outMode ? 32767 : (1 << 19) - 1; L.outMode = outMode; L.outShift = outShift;
    C.ncomp = 2; C.nraw = srcSemi ? 1 : 2; C.ileave = dstSemi ? 1 : 0; C.layout = (bps == 2 ? 2 : 0) + (srcSemi ? 1 : 0);
    C.rawSel[0] = 1; C.rawSel[1] = 2;
    C.kind = kind; C.xorv = xorv; C.rowBytes = p.chrSrcW * bps * (srcSemi ? 2 : 1);
    C.srcW = p.chrSrcW; C.srcH = p.chrSrcH; C.dstW = p.chrDstW; C.dstH = p.chrDstH;
    C.dstSel[0] = 1; C.dstSel[1] = dstSemi ? 1 : 2;
    C.dstOff[0] = 0; C.dstOff[1] = dstSemi ? (outMode == 1 ? 1 : 2) : 0;
    C.sh = sh; C.maxv = outMode ? 32767 : (1 << 19) - 1; C.outMode = outMode; C.outShift = outShift;
    const int hp = std::max(p.hLum.pairs, p.hChr.pairs);
    t.np = hp <= 4 ? 4 : hp <= 8 ? 8 : 0;
    L.np = C.np = t.np;
    const int hpL = t.np ? t.np : p.hLum.pairs, hpC = t.np ? t.np : p.hChr.pairs;
    t.rgb64 = rgb64; t.chrShift = chrShift; t.linesOff = 0;
    if (!rgb64) {
        int r = plan_job(p.hLum, vl, p.srcW, p.srcH, p.dstW, p.dstH, 1, 1, L.layout, hpL, budget, thCap, L, t.colStart[0], t.rowStart[0], t.rowCount[0]);
        if (r < 0) return r;
        int lds = r;
        r = plan_job(p.hChr, vc, p.chrSrcW, p.chrSrcH, p.chrDstW, p.chrDstH, 2, C.nraw, C.layout, hpC, budget, thCap, C, t.colStart[1], t.rowStart[1], t.rowCount[1]);
        if (r < 0) return r;
        t.ldsBytes = std::max(lds, r);
    } else {
        // one grid of tiles: 64 pixels across, the chroma job's tile 64 >> chrShift of ITS columns; the vertical chroma bank has a row an output row
        if (vc.count != p.dstH || p.chrDstW != ((p.dstW + (1 << chrShift) - 1) >> chrShift)) return GMAT_ERR(ENOSYS);
        C.dstH = p.dstH;
        if (int r = check_banks(p.hLum, vl, p.srcW, p.srcH, p.dstW, p.dstH); r < 0) return r;
        if (int r = check_banks(p.hChr, vc, p.chrSrcW, p.chrSrcH, p.chrDstW, p.dstH); r < 0) return r;
        const int gcapL = plan_cols(p.hLum, p.dstW, kS19TW, 1, L.layout, hpL, L, t.colStart[0]);
        const int gcapC = plan_cols(p.hChr, p.chrDstW, kS19TW >> chrShift, C.nraw, C.layout, hpC, C, t.colStart[1]);
        if (gcapL < 0) return gcapL;
        if (gcapC < 0) return gcapC;
        if (L.ntx != C.ntx) return GMAT_ERR(ENOSYS);
        bool done = false;
        for (int TH : kS19TH) {
            if (TH > thCap) continue;
            std::vector<int32_t> rsL, rcL, rsC, rcC;
            int nrL, nlL, nrC, nlC;
            plan_rows(vl, p.srcH, p.dstH, TH, rsL, rcL, nrL, nlL);
            plan_rows(vc, p.chrSrcH, p.dstH, TH, rsC, rcC, nrC, nlC);
            const int GL = plan_group(L, 1, nrL, gcapL, budget), GC = plan_group(C, 2, nrC, gcapC, budget);
            if (GL < 1 || GC < 1) { if (TH == 1) return GMAT_ERR(ENOSYS); continue; }
            const long vtL = ((long)TH * (2 * vl.pairs + 2) * 4 + 15) & ~15L, vtC = ((long)TH * (2 * vc.pairs + 2) * 4 + 15) & ~15L;
            const long rawB = std::max((long)GL * L.PP * 4, (long)GC * 2 * C.PP * 4);
            const long total = vtL + vtC + rawB + ((long)nlL + 2L * nlC) * kS19TW * 4;
            if (total > budget && TH > 1) continue;
            if (total > 65536) return GMAT_ERR(ENOSYS);
            L.TH = C.TH = TH; L.nty = C.nty = (int)rsL.size(); L.nblk = C.nblk = L.ntx * L.nty;
            L.nrMax = nrL; L.nrLines = nlL; L.G = GL; L.vtBytes = (int)vtL;
            C.nrMax = nrC; C.nrLines = nlC; C.G = GC; C.vtBytes = (int)vtC;
            t.rowStart[0].swap(rsL); t.rowCount[0].swap(rcL); t.rowStart[1].swap(rsC); t.rowCount[1].swap(rcC);
            t.linesOff = (int)(vtL + vtC + rawB);
            t.ldsBytes = (int)((total + 15) & ~15L);
            done = true;
            break;
        }
        if (!done) return GMAT_ERR(ENOSYS);
    }
    if (GMAT_KNOB("GMAT_S19_DEBUG"))
        for (int j = 0; j < 2; j++)
            fprintf(stderr, "s19 job %d: %d x %d -> %d x %d layout %d np %d vp %d TW %d TH %d tiles %d x %d nrMax %d nrLines %d PP %d G %d lshift %d lds %d rgb64 %d\n", j, t.job[j].srcW, t.job[j].srcH,
                    t.job[j].dstW, t.job[j].dstH, t.job[j].layout, t.np, j ? vc.pairs : vl.pairs, t.job[j].TW, t.job[j].TH, t.job[j].ntx, t.job[j].nty, t.job[j].nrMax, t.job[j].nrLines, t.job[j].PP, t.job[j].G, t.job[j].lshift, t.ldsBytes, rgb64);
    t.ok = 1;
    return 0;
}

// The LDS a block may take (measured, profiles/r06_scale19_history.txt r06v: 16 ... 48 KB over the twelve dst16 cases): 32 KB — five blocks a CU — wherever a tile
// of that size spends at least 85 % of its staged rows on rows of its own (tile rows x the vertical ratio / the rows its windows span): every 1.5 : 1 and 2 : 1
// bicubic case is flat or best there.  A tile that does not — long vertical windows: 3 : 1 lanczos 23.9 us a frame at 32 KB, 20.0 at 40, 19.5 at 48; a 64-bit
// destination's three line sets, 4K -> 1080p 23.1 / 20.5 / 21.5 — takes 40, then 48 KB.  GMAT_S19_LDS pins the number.
int s19_prepare(const ScalePlan &p, const FilterBank &vl, const FilterBank &vc, int bps, int kind, int srcSemi, int dstSemi, int rgb64, int chrShift, S19Tables &t,
                int outMode, int outShift, int hsh)
{
    const char *ks = GMAT_KNOB("GMAT_S19_LDS"), *kr = GMAT_KNOB("GMAT_S19_ROWS");
    const int thCap = kr ? std::max(1, atoi(kr)) : 64;
    t.ok = 0;
    if (ks) return s19_prepare_budget(p, vl, vc, bps, kind, srcSemi, dstSemi, rgb64, chrShift, t, outMode, outShift, hsh, std::min(std::max(atoi(ks), 4096), 65536), thCap);
    int r = GMAT_ERR(ENOSYS);
    for (int budget : {32768, 40960, 49152}) {
        S19Tables cand;
        const int rc = s19_prepare_budget(p, vl, vc, bps, kind, srcSemi, dstSemi, rgb64, chrShift, cand, outMode, outShift, hsh, budget, thCap);
        if (rc < 0) { if (!t.ok) r = rc; continue; }
        const S19Job &L = cand.job[0];
        const bool full = (long)L.TH * L.srcH * 100 >= 85L * L.nrMax * L.dstH || L.nty == 1;
        t = std::move(cand); r = 0;
        if (full) break;
    }
    return r;
}

// the bank's one coefficient when output i is source sample i and nothing else, for every i, with the same coefficient; 0 otherwise
static int s19_identity_coef(const FilterBank &fb, int n)
{
    if (fb.count != n || fb.pairs < 1) return 0;
    int coef = 0;
    for (int i = 0; i < n; i++) {
        const int k = i - fb.pos_even[i];
        if (k < 0 || k >= 2 * fb.pairs) return 0;
        for (int t = 0; t < 2 * fb.pairs; t++) {
            const int32_t pk = fb.packed[(size_t)i * fb.pairs + (t >> 1)];
            const int c = (t & 1) ? pk >> 16 : (int)(int16_t)(pk & 0xFFFF);
            if ((c != 0) != (t == k)) return 0;
            if (t == k) { if (coef && c != coef) return 0; coef = c; }
        }
    }
    return coef;
}

void s19_unit_plan(const ScalePlan &p, const FilterBank &vl, const FilterBank &vc, const int32_t *lumRound, const int32_t *chrRound, S19Tables &t)
{
    t.unit = 0;
    const char *ku = GMAT_KNOB("GMAT_S19_UNIT");
    if (!t.ok || (ku && atoi(ku) == 0)) return;
    if (t.rgb64) {
        // a packed 64-bit destination: identity horizontal banks, an identity vertical luma bank; the chroma's vertical bank is whatever it is (scale19_unit64_kernel);
        // planar 4 : 4 : 4 or any 4 : 2 : 0 layout, rows of whole units
        S19Job &L = t.job[0], &C = t.job[1];
        if (L.srcW != L.dstW || L.srcH != L.dstH || C.srcW != C.dstW || (L.dstW & 7)) return;
        if (C.dstW != (t.chrShift ? (L.dstW + 1) >> 1 : L.dstW) || (!t.chrShift && (C.layout & 1))) return;
        if (s19_identity_coef(p.hLum, L.dstW) != 16384 || s19_identity_coef(p.hChr, C.dstW) != 16384) return;
        const int cv = s19_identity_coef(vl, L.dstH);
        if (cv <= 0 || cv >= (1 << 15) || vc.count != L.dstH || vc.pairs < 1) return;
        L.unitCoef = cv; L.unitRound = 0;
        t.unit = 2;
        return;
    }
    for (int j = 0; j < 2; j++) {
        S19Job &J = t.job[j];
        if (J.srcW != J.dstW || J.srcH != J.dstH) return;
        if (s19_identity_coef(j ? p.hChr : p.hLum, J.dstW) != 16384) return;
        const int cv = s19_identity_coef(j ? vc : vl, J.dstH);
        if (cv <= 0 || cv >= (1 << 15)) return;
        const int32_t *rnd = j ? chrRound : lumRound;
        int k0 = (int)((1u << 14) - 0x40000000u);                               // yuv2planeX_16_c's constant (s19_lines)
        if (J.outMode) {
            if (!rnd) return;
            k0 = rnd[0];
            for (int y = 1; y < J.dstH; y++) if (rnd[y] != k0) return;
        }
        J.unitCoef = cv; J.unitRound = k0;
    }
    t.unit = 1;
}

int launch_scale19(const S19Args &a0, int np, int ldsBytes, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (!frames || nframes < 1 || nframes > kYuv2xMaxFrames || ldsBytes < 1 || ldsBytes > 65536) return GMAT_ERR(EINVAL);
    S19Args a = a0;
    if (a.unit == 2 && !(a.srcAl16 && a.dstAl16)) a.unit = 0;                   // (the 64-bit unit form has no sample-by-sample path: the tile form takes odd planes)
    if (a.unit) {
        if (a.unit == 2) a.job[1].dstW = 0, a.job[1].dstH = 0;                   // (one grid of pixel units: no blocks of the chroma job's own)
        for (int j = 0; j < 2; j++) {
            const long upr = ((long)a.job[j].dstW + 7) >> 3;
            a.unitBlk[j] = (int)((upr * a.job[j].dstH + 255) / 256);
            if ((long)a.unitBlk[j] * 256 >= (1L << 31)) return GMAT_ERR(EINVAL);
            // n / upr = (n m) >> (31 + L), L = ceil(log2 upr), m = floor(2^(31 + L) / upr) + 1 < 2^32: exact for n < 2^31 (the excess n e / (upr 2^(31 + L)) < 2^-L <= 1 / upr)
            int L = 0;
            while ((1L << L) < upr) L++;
            a.unitMul[j] = upr > 1 ? (unsigned)(((1ULL << (31 + L)) / (unsigned long long)upr) + 1) : 0u;
            a.unitShr[j] = upr > 1 ? L - 1 : 0;
        }
        S19Frame1 f1;
        f1.y[0] = frames->y[0]; f1.u[0] = frames->u[0]; f1.v[0] = frames->v[0]; f1.dst[0] = frames->dst[0]; f1.dstU[0] = frames->dstU[0]; f1.dstV[0] = frames->dstV[0];
        if (a.unit == 2) {
            a.job[1].dstW = a0.job[1].dstW; a.job[1].dstH = a0.job[1].dstH;
            if (nframes == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale19_unit64_kernel<S19Frame1>), dim3(a.unitBlk[0]), dim3(256), 0, stream, a, f1);
            else              hipLaunchKernelGGL(HIP_KERNEL_NAME(scale19_unit64_kernel<Yuv2xFrames>), dim3(a.unitBlk[0] * nframes), dim3(256), 0, stream, a, *frames);
        } else {
            if (nframes == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale19_unit_kernel<S19Frame1>), dim3(a.unitBlk[0] + a.unitBlk[1]), dim3(256), 0, stream, a, f1);
            else              hipLaunchKernelGGL(HIP_KERNEL_NAME(scale19_unit_kernel<Yuv2xFrames>), dim3((a.unitBlk[0] + a.unitBlk[1]) * nframes), dim3(256), 0, stream, a, *frames);
        }
        GMAT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const dim3 grid((a.rgb64 ? a.job[0].nblk : a.job[0].nblk + a.job[1].nblk) * nframes), block(256);
    const Yuv2xFrames &fr = *frames;
    const char *kx = GMAT_KNOB("GMAT_SCALE_XCD");
    a.xcdRemap = kx ? atoi(kx) != 0 : 1;
    // (a 48-byte frame table for one-frame launches of THIS kernel was measured, r06y8: nothing — its one-frame time is the blocks' latency chains)
    switch (np) {
    case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(scale19_kernel<4>), grid, block, ldsBytes, stream, a, fr); break;
    case 8: hipLaunchKernelGGL(HIP_KERNEL_NAME(scale19_kernel<8>), grid, block, ldsBytes, stream, a, fr); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(scale19_kernel<0>), grid, block, ldsBytes, stream, a, fr); break;
    }
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

void unit_rgb_plan(const ScalePlan &p, const FilterBank &vl, const int32_t *lumRound, int fullChroma, UnitRgbPlan &u)
{
    u.ok = 0;
    const char *ku = GMAT_KNOB("GMAT_S19_UNIT");
    if ((ku && atoi(ku) == 0) || fullChroma || !lumRound) return;
    if (p.srcW != p.dstW || p.srcH != p.dstH || (p.dstW & 7) || p.chrDstW != (p.dstW + 1) / 2 || p.chrSrcW != p.chrDstW) return;
    if (s19_identity_coef(p.hLum, p.dstW) != 16384 || s19_identity_coef(p.hChr, p.chrDstW) != 16384) return;
    const int cv = s19_identity_coef(vl, p.dstH);
    if (cv <= 0 || cv >= (1 << 15)) return;
    for (int y = 1; y < p.dstH; y++) if (lumRound[y] != lumRound[0]) return;
    u.coefL = cv; u.roundL = lumRound[0];
    u.ok = 1;
}

int launch_unit_rgb(const UnitRgbArgs &a0, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (!frames || nframes < 1 || nframes > kYuv2xMaxFrames || (a0.w & 7) || a0.w < 8 || a0.h < 1 || (a0.px != 3 && a0.px != 4)) return GMAT_ERR(EINVAL);
    UnitRgbArgs a = a0;
    const long upr = a.w >> 3;
    a.blocks = (int)((upr * a.h + 255) / 256);
    if ((long)a.blocks * 256 >= (1L << 31)) return GMAT_ERR(EINVAL);
    int L = 0;
    while ((1L << L) < upr) L++;
    a.rowMul = upr > 1 ? (unsigned)(((1ULL << (31 + L)) / (unsigned long long)upr) + 1) : 0u;     // (launch_scale19's multiplier)
    a.rowShr = upr > 1 ? L - 1 : 0;
    const dim3 grid(a.blocks * nframes), block(256);
    const int srck = a.semi ? (a.shr6 ? 2 : 1) : 0;
#define URGB_LAUNCH(PX_, S_) hipLaunchKernelGGL(HIP_KERNEL_NAME(unit_rgb_kernel<PX_, S_, Yuv2xFrames>), grid, block, 0, stream, a, *frames)
    if (a.px == 3) { if (srck == 0) URGB_LAUNCH(3, 0); else if (srck == 1) URGB_LAUNCH(3, 1); else URGB_LAUNCH(3, 2); }
    else           { if (srck == 0) URGB_LAUNCH(4, 0); else if (srck == 1) URGB_LAUNCH(4, 1); else URGB_LAUNCH(4, 2); }
#undef URGB_LAUNCH
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
