// k_scale_yuv3x1.hip — strip-walking form of the exact 3:1 down-scale of 8-bit YUV 4:2:0 (4K -> 720p, 1080p -> 360p), NV12 -> NV12
// and YUV420P -> YUV420P, with the arithmetic of ONE libswscale context (hScale8To15_c per plane, yuv2planeX_8_c / yuv2nv12cX_c
// vertically, swscale.c:234-520, output.c:400-450), bit-exact.  The generic plane scaler spends 12 us on a 4K -> 720p frame (0.14
// of the HBM roofline): 11-tap rows through LDS tiles, coefficients from memory.
//
// At 3:1 the bicubic filter has ONE phase of 11 taps: output x reads source [3x - 4, 3x + 6], and so do the rows.  libswscale folds
// taps outside the plane onto the edge sample; the host checks that every table row IS the interior row on an edge-replicated line
// (filter_is_edge_replication_ratio, true for all four tables of a bicubic 3:1 context) and passes the 11 coefficients as 6 int16
// pairs in the kernel arguments.
//   * a wave owns a strip of 256 output columns (a lane: 4 adjacent outputs from 20 source bytes at the 4-aligned offset 3x - 4) and
//     walks down the SOURCE rows in steps of three: the next two steps' rows are in flight while this step's are filtered (18 byte
//     pairs by v_perm_b32 and 24 v_dot2 per row), each row packed with the row above by v_cvt_pk_i16_i32 (which is also
//     hScale8To15_c's saturation) into a ring of the last 12 row pairs (n-1 | n);
//   * after every step one output row leaves: 6 v_dot2 per sample over the pairs ending at rows 3y-3, 3y-1 .. 3y+7 (the last tap's
//     partner row 3y + 7 has coefficient 0);
//   * pixels never pass through LDS; ring slots, load buffers and byte selectors are static after unrolling four steps.
// Parity: held to the oracle (tests/test_parity_down3.py, together with the generic kernel on the same matrix); no vector the
// reference holds is a 3:1 scale.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

namespace gmat {

constexpr int D3_STRIP = 256;                  // output columns per wave of a single-channel plane: 64 lanes x 4
constexpr int D3_STRIP_UV = 128;               // output UV positions per wave of the interleaved plane: 64 lanes x 2

// 4 waves per SIMD = at most 128 VGPRs: without the bound the allocator takes 129 and a whole wave per SIMD is lost
#if defined(__HIP__)
#define D3_FOUR_WAVES __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define D3_FOUR_WAVES
#endif
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned d3_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ uint4 d3_ld16(const uint8_t *p) { const d3_u32x4 v = *reinterpret_cast<const d3_u32x4 *>(p); return make_uint4(v.x, v.y, v.z, v.w); }
#else
static inline uint4 d3_ld16(const uint8_t *p) { uint4 v; std::memcpy(&v, p, 16); return v; }
#endif
__device__ __forceinline__ unsigned d3_ld4(const uint8_t *p) { return *reinterpret_cast<const unsigned *>(p); }

// One dword to base + off.  On the device the store is issued from inline assembly, out of the compiler's sight: gfx9 counts loads
// and stores in the same vmcnt, and LLVM's wait insertion treats a counter with two kinds of events pending as out of order — with a
// store in flight every wait for a load becomes s_waitcnt vmcnt(0), which also waits for the rows just requested for two steps
// ahead (the trace showed exactly that at the top of every step).  Seen as loads only, the counter is in order and the waits are
// counted ones; the stores the compiler does not know about make its counts stricter than necessary, never laxer (the counter
// then includes them, loads still return in order among themselves).  Nothing in the kernel reads the destination.
__device__ __forceinline__ void d3_st4(uint8_t *base, unsigned off, unsigned v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2" : : "v"(off), "v"(v), "s"(base));      // s_nop 4: see g_st in k_scale_yuvg.hip (VALU-written scalar operand)
#else
    *reinterpret_cast<unsigned *>(base + off) = v;
#endif
}

__device__ __forceinline__ int d3_dot2(int packed_ab, int packed_cd, int acc)      // three-operand v_dot2_i32_i16 (see k_scale_yuv2s.hip)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, packed_ab), __builtin_bit_cast(short2v, packed_cd), acc, true);
}
__device__ __forceinline__ unsigned d3_rep(unsigned v, unsigned sel) { return __builtin_amdgcn_perm(v, v, sel); }
// cond ? a : b on VALUES (see u2_blend in k_scale_yuv1x2.hip: a ?: on members of an in-memory struct selects an ADDRESS into scratch memory)
__device__ __forceinline__ int32_t d3_blend(bool cond, int32_t a, int32_t b) { return b ^ ((a ^ b) & -(int32_t)cond); }

struct D3Plane {
    const uint8_t *src; uint8_t *dst;
    int ss, ds, dstW, srcW, srcH;              // widths in samples (UV plane: in UV positions)
    int32_t h[6], v[6];                        // 11 taps as int16 pairs, the 12th coefficient is 0
    int rnd;
};

// bytes O, O + 1 of a lane's 20-byte window d[0 .. 4], widened to an int16 pair (byte 20 does not exist: its coefficient is 0)
template <int O>
__device__ __forceinline__ int d3_pair(const unsigned (&d)[5])
{
    constexpr int dw = O >> 2, b = O & 3;
    if constexpr (b < 3) return (int)__builtin_amdgcn_perm(0u, d[dw], 0x0C000C00u | ((unsigned)(b + 1) << 16) | (unsigned)b);
    else if constexpr (dw < 4) return (int)__builtin_amdgcn_perm(d[dw + 1], d[dw], 0x0C040C03u);
    else return (int)__builtin_amdgcn_perm(0u, d[dw], 0x0C0C0C03u);
}

// output sample J of the lane: taps at bytes 3J .. 3J + 10
template <int J>
__device__ __forceinline__ int d3_hsum(const D3Plane &P, const unsigned (&d)[5])
{
    int s = d3_dot2(d3_pair<3 * J>(d), P.h[0], 0);
    s = d3_dot2(d3_pair<3 * J + 2>(d), P.h[1], s);
    s = d3_dot2(d3_pair<3 * J + 4>(d), P.h[2], s);
    s = d3_dot2(d3_pair<3 * J + 6>(d), P.h[3], s);
    s = d3_dot2(d3_pair<3 * J + 8>(d), P.h[4], s);
    return d3_dot2(d3_pair<3 * J + 10>(d), P.h[5], s);
}

// The walk shared by both plane kinds.  NW = dwords of a lane's window per row; LOAD(row, d, edge_c) / HROW(d, edge_c, s[4]) /
// STORE(y, w[4]) are the plane kind's.  Step s of a segment handles the source rows 3 y0 - 4 + 3 s + {0, 1, 2}; from step 3 on
// output row y0 + s - 3 leaves.  PH = s mod 4 names the ring slots (PH * 3 + r) the step writes, PAR = s & 1 its load buffers.
//
// Odd segments walk UPWARD (up != 0): the same steps over the rows in descending order, the coefficient pairs taken in reverse with
// their halves swapped (a pair is (row before | row now), and "before" is then the row below), output rows counted down.  Segment s
// (down) and s + 1 (up) then reach their common boundary at the same time — both at the end of their walk — and s + 1 and s + 2 both
// START at theirs: the 9 rows either side of a boundary that both segments need are read twice within a few microseconds on the
// same XCD (L2 hits) instead of once at the start of one wave and once at the end of another (two HBM reads).
template <int NW, typename Load, typename HRow, typename Store>
__device__ __forceinline__ void d3_walk(const D3Plane &P, int y0, int nOut, bool edgeWave, int up, Load &&load, HRow &&hrow, Store &&store)
{
    int ring[12][4], prev[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        prev[q] = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) ring[i][q] = 0;
    }
    unsigned buf[2][3][NW];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int i = 0; i < NW; i++) buf[a][r][i] = 0u;
    const int nStart = 3 * y0 - 4;
    const int nLast = 3 * (y0 + nOut - 1) + 7;                  // the last source row of the segment (before clamping into the plane)
    const int nSteps = nOut + 3;
    const int rowBase = up ? nLast : nStart, rowDir = up ? -1 : 1, yBase = up ? y0 + nOut - 1 : y0;
    auto rowOf = [&](int i) { return min(max(rowBase + rowDir * i, nStart), nLast); };     // i = 3 * step + r, never outside the segment's rows
    int32_t vv[6];                                              // wave-uniform: scalar registers
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const uint32_t rev = (uint32_t)P.v[5 - k];
        vv[k] = d3_blend(up != 0, (int32_t)((rev >> 16) | (rev << 16)), P.v[k]);
    }

    auto body = [&](const int s, auto ph_c, auto edge_c) {
        constexpr int PH = decltype(ph_c)::value, PAR = PH & 1;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            int hs[4];
            hrow(buf[PAR][r], edge_c, hs);
            // rolling prefetch: the registers just consumed receive the same row of step s + 2.  Unconditional (a branch here costs
            // the in-order vmcnt its slack: 7.2 instead of 4.8 us per frame on the UV plane), but never past the segment's last row:
            // the loads of the last two steps re-read that row (cache hits) instead of 6 rows of the neighbour's (HBM traffic)
            load(rowOf(3 * (s + 2) + r), buf[PAR][r], edge_c);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int cur = hs[q] >> 7;                     // hScale8To15_c: min(val >> 7, 32767) — the pack saturates
                ring[PH * 3 + r][q] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(prev[q], cur));
                prev[q] = cur;
            }
        }
        if (s >= 3) {
            // the pairs ending 10, 8, 6, 4, 2, 0 rows above the step's last row (slots of the steps s-3 .. s)
            constexpr int A = ((PH + 1) & 3) * 3 + 1, B = ((PH + 2) & 3) * 3, C = ((PH + 2) & 3) * 3 + 2, D = ((PH + 3) & 3) * 3 + 1,
                          E = PH * 3, F = PH * 3 + 2;
            unsigned w[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int acc = d3_dot2(ring[A][q], vv[0], P.rnd);
                acc = d3_dot2(ring[B][q], vv[1], acc);
                acc = d3_dot2(ring[C][q], vv[2], acc);
                acc = d3_dot2(ring[D][q], vv[3], acc);
                acc = d3_dot2(ring[E][q], vv[4], acc);
                acc = d3_dot2(ring[F][q], vv[5], acc);
                w[q] = (unsigned)clip_u8_shr(acc, 19);
            }
            store(yBase + rowDir * (s - 3), w);
        }
    };
    auto run = [&](auto edge_c) {
#pragma unroll
        for (int r = 0; r < 3; r++) load(rowOf(r), buf[0][r], edge_c);
#pragma unroll
        for (int r = 0; r < 3; r++) load(rowOf(3 + r), buf[1][r], edge_c);
        for (int s0 = 0; s0 < nSteps; s0 += 4) {
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
            body(s0, I0(), edge_c);
            if (s0 + 1 < nSteps) body(s0 + 1, I1(), edge_c);
            if (s0 + 2 < nSteps) body(s0 + 2, I2(), edge_c);
            if (s0 + 3 < nSteps) body(s0 + 3, I3(), edge_c);
        }
    };
    // prev must hold row 3 y0 - 5 before the first pair is formed, but that pair is never used: any value does
    if (edgeWave) run(std::true_type()); else run(std::false_type());
}

// ---- one single-channel plane: the output rows [y0, y0 + nOut) of the strip at X0 ------------------------------------------
__device__ __forceinline__ void d3_walk_plane(const D3Plane &P, int X0, int y0, int nOut, int up, int lane)
{
    const int xo = X0 + 4 * lane;
    const bool active = xo < P.dstW;
    const int xc = active ? xo : P.dstW - 4;                    // idle lanes shadow the last group
    const bool edgeWave = X0 == 0 || 3 * (X0 + D3_STRIP) + 8 > P.srcW;   // a window of this wave may leave the row
    const unsigned bo = (unsigned)(3 * xc - 4);                 // byte offset of the window base, a multiple of 4 (negative in lane 0 of the first strip)

    // Edge waves: only the lane at x = 0 (its window starts 4 bytes before the row) and the lanes at the last group (theirs ends 4 bytes
    // after it) reach outside; they load the same 20 bytes one dword further in / out and shift the registers back, the dword that
    // falls outside becomes the replicated edge sample.  (The first version loaded each dword of an edge wave from its own clamped
    // address: 5 load instructions per row instead of 2 — a wave-level load occupies the CU's address path for ~17 cycles whatever its
    // width (tools/ubench/load_rate.hip), and 2 of the 5 strips of a 1280-wide plane are edge waves.)
    const bool isLeft = xc == 0, isRight = xc == P.dstW - 4;
    const unsigned lbo = bo + (isLeft ? 4u : 0u) - (isRight ? 4u : 0u);
    auto load = [&](int row, unsigned (&d)[5], auto edge_c) {
        const unsigned o = (unsigned)min(max(row, 0), P.srcH - 1) * (unsigned)P.ss;
        const uint8_t *p = P.src + (o + (decltype(edge_c)::value ? lbo : bo));      // interior waves: bo >= 0
        const uint4 t = d3_ld16(p);
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
        d[4] = d3_ld4(p + 16);
    };
    auto hrow = [&](const unsigned (&src)[5], auto edge_c, int (&s)[4]) {
        unsigned d[5] = {src[0], src[1], src[2], src[3], src[4]};
        if constexpr (decltype(edge_c)::value) {
            const unsigned first = d3_rep(src[0], 0x00000000u), last = d3_rep(src[4], 0x03030303u);
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const unsigned fromLeft = i == 0 ? first : src[i - 1], fromRight = i == 4 ? last : src[i + 1];
                d[i] = isLeft ? fromLeft : isRight ? fromRight : src[i];
            }
        }
        s[0] = d3_hsum<0>(P, d); s[1] = d3_hsum<1>(P, d); s[2] = d3_hsum<2>(P, d); s[3] = d3_hsum<3>(P, d);
    };
    auto store = [&](int y, const unsigned (&w)[4]) {
        if (active) d3_st4(P.dst, (unsigned)y * (unsigned)P.ds + (unsigned)xo, w[0] | (w[1] << 8) | (w[2] << 16) | (w[3] << 24));
    };
    d3_walk<5>(P, y0, nOut, edgeWave, up, load, hrow, store);
}

// ---- NV12's interleaved UV plane: a lane makes 2 UV output positions (4 bytes) from 15 source positions (8 dwords) ------------
__device__ __forceinline__ void d3_walk_uv(const D3Plane &P, int X0, int y0, int nOut, int up, int lane)
{
    const int co = X0 + 2 * lane;
    const bool active = co < P.dstW;
    const int cc = active ? co : P.dstW - 2;
    const bool edgeWave = X0 == 0 || 3 * (X0 + D3_STRIP_UV) + 8 > P.srcW;
    const unsigned bo = 2u * (unsigned)(3 * cc - 4);            // byte offset of the window base: a multiple of 4 (3 cc - 4 is even)

    // edge waves as in the plane walker: the lane at position 0 starts 2 dwords before the row, the lanes at the last pair end 3 dwords
    // after it
    const bool isLeft = cc == 0, isRight = cc == P.dstW - 2;
    const unsigned lbo = bo + (isLeft ? 8u : 0u) - (isRight ? 12u : 0u);
    auto load = [&](int row, unsigned (&d)[8], auto edge_c) {
        const unsigned o = (unsigned)min(max(row, 0), P.srcH - 1) * (unsigned)P.ss;
        const uint8_t *p = P.src + (o + (decltype(edge_c)::value ? lbo : bo));
        const uint4 t = d3_ld16(p);
        const uint4 u = d3_ld16(p + 16);
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; d[4] = u.x; d[5] = u.y; d[6] = u.z; d[7] = u.w;
    };
    // s = U0 V0 U1 V1: output position J reads source positions 3J .. 3J + 10 of the window (position i = bytes 2i (U), 2i + 1 (V))
    auto hrow = [&](const unsigned (&src)[8], auto edge_c, int (&s)[4]) {
        unsigned d[8];
#pragma unroll
        for (int i = 0; i < 8; i++) d[i] = src[i];
        if constexpr (decltype(edge_c)::value) {
            const unsigned first = d3_rep(src[0], 0x01000100u), last = d3_rep(src[7], 0x03020302u);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const unsigned fromLeft = i < 2 ? first : src[i - 2], fromRight = i > 4 ? last : src[i + 3];
                d[i] = isLeft ? fromLeft : isRight ? fromRight : src[i];
            }
        }
        int u0 = 0, v0 = 0, u1 = 0, v1 = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            // output 0: positions (2k, 2k + 1) = dword k; output 1: positions (2k + 3, 2k + 4) = high half of dword k + 1, low half of k + 2
            u0 = d3_dot2((int)__builtin_amdgcn_perm(0u, d[k], 0x0C020C00u), P.h[k], u0);
            v0 = d3_dot2((int)__builtin_amdgcn_perm(0u, d[k], 0x0C030C01u), P.h[k], v0);
            u1 = d3_dot2((int)__builtin_amdgcn_perm(d[k + 2], d[k + 1], 0x0C040C02u), P.h[k], u1);
            v1 = d3_dot2((int)__builtin_amdgcn_perm(d[k + 2], d[k + 1], 0x0C050C03u), P.h[k], v1);
        }
        s[0] = u0; s[1] = v0; s[2] = u1; s[3] = v1;
    };
    auto store = [&](int y, const unsigned (&w)[4]) {
        if (active) d3_st4(P.dst, (unsigned)y * (unsigned)P.ds + 2u * (unsigned)co, w[0] | (w[1] << 8) | (w[2] << 16) | (w[3] << 24));
    };
    d3_walk<8>(P, y0, nOut, edgeWave, up, load, hrow, store);
}

__device__ __forceinline__ D3Plane d3_plane(const uint8_t *src, uint8_t *dst, int ss, int ds, int dstW, int srcW, int srcH,
                                            const int32_t (&h)[6], const int32_t (&v)[6], int rnd)
{
    D3Plane P;
    P.src = src; P.dst = dst; P.ss = ss; P.ds = ds; P.dstW = dstW; P.srcW = srcW; P.srcH = srcH; P.rnd = rnd;
#pragma unroll
    for (int k = 0; k < 6; k++) { P.h[k] = h[k]; P.v[k] = v[k]; }
    return P;
}

// blockIdx.x: [0, nblkL) luma workgroups, then the chroma workgroups.  A wave's unit of work is one (segment, strip) pair; units are
// packed densely into workgroups (unit = 4 * workgroup + wave, segment-major), whatever the number of strips per row: the waves of a
// workgroup share nothing (no LDS), and 1280 columns are FIVE strips — grouped by row that is one full workgroup and one with a
// single live wave, which always lands on the same SIMD.  blockIdx.y = frame.  A segment is segRows output rows.
template <bool NV>
__global__ __launch_bounds__(256) D3_FOUR_WAVES void scale_yuv3x1_kernel(Yuv3x1Args a, Yuv2xFrames fr)
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
        const int X0 = (unit - seg * a.nsgL) * D3_STRIP;
        const int y0 = seg * a.segRowsL;
        const D3Plane P = d3_plane(fr.y[f], fr.dst[f], a.ys, a.ds, a.dstW, 3 * a.dstW, 3 * a.dstH, a.hL, a.vL, a.lr);
        d3_walk_plane(P, X0, y0, min(a.segRowsL, a.dstH - y0), a.updown & seg & 1, lane);
        return;
    }
    int unit = (lin - a.nblkL) * 4 + wave;
    const int per = a.nsegC * a.nsgC;                            // units of one chroma plane
    if (NV) {
        if (unit >= per) return;
        const int seg = __builtin_amdgcn_readfirstlane(unit / a.nsgC);
        const int X0 = (unit - seg * a.nsgC) * D3_STRIP_UV;
        const int y0 = seg * a.segRowsC;
        const D3Plane P = d3_plane(fr.u[f], fr.dstU[f], a.us, a.dsU, a.chrDstW, 3 * a.chrDstW, 3 * a.chrDstH, a.hC, a.vC, a.cr);
        d3_walk_uv(P, X0, y0, min(a.segRowsC, a.chrDstH - y0), a.updown & seg & 1, lane);
    } else {
        if (unit >= 2 * per) return;
        const int pl = __builtin_amdgcn_readfirstlane(unit >= per ? 1 : 0);
        unit -= pl * per;
        const int seg = __builtin_amdgcn_readfirstlane(unit / a.nsgC);
        const int X0 = (unit - seg * a.nsgC) * D3_STRIP;
        const int y0 = seg * a.segRowsC;
        const D3Plane P = d3_plane(pl ? fr.v[f] : fr.u[f], pl ? fr.dstV[f] : fr.dstU[f], pl ? a.vs : a.us, pl ? a.dsV : a.dsU,
                                   a.chrDstW, 3 * a.chrDstW, 3 * a.chrDstH, a.hC, a.vC, a.cr);
        d3_walk_plane(P, X0, y0, min(a.segRowsC, a.chrDstH - y0), a.updown & seg & 1, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// scale_yuv3r_kernel: NV12 at a third of the size into packed RGB (4K -> 720p, 1080p -> 360p: a decoder's frame into a network's
// input), ONE libswscale context: hScale8To15_c on both planes, yuv2rgb_X_c's vertical sums (>> 19) and table stage (output.c:1680-1731).
// An RGB destination keeps its chroma at half the output width and full output height, so chroma is 3:1 horizontally — the lane's
// window and filter of the UV walker above — and 3:2 vertically (k_scale_yuv3x2.hip: 6 taps, even rows [3k - 2, 3k + 3] with A, odd
// rows [3k - 1, 3k + 4] with B, output row 1 with the table's own row S).  A lane makes 4 pixels of an output row: 4 luma sums from its
// 20-byte window, 2 chroma pairs from its 32-byte UV window.  Both vertical filters run as RUNNING SUMS instead of rings of rows:
// step T takes luma rows 6T - 5 .. 6T and chroma rows 3T - 2 .. 3T; a luma triple (3u - 2, 3u - 1, 3u) feeds the four open output
// rows u - 2 .. u + 1 (the first two rows as a pair through v_dot2, the third by v_mad_i32_i24) and closes row u - 2; a chroma row
// feeds four open rows; output rows 2T - 3 and 2T - 2 leave in the middle and at the end of the step.  Accumulator slots are static
// after unrolling two steps.
// ---------------------------------------------------------------------------------------------
// GMAT_Y3R_FOUR_WAVES (a build-time A/B): bound the kernel to 128 VGPRs; it takes 160 unbounded (3 waves per SIMD, no scratch)
#if defined(GMAT_Y3R_FOUR_WAVES)
#define D3R_BOUND D3_FOUR_WAVES
#else
#define D3R_BOUND
#endif
template <int DST>
__global__ __launch_bounds__(256) D3R_BOUND void scale_yuv3r_kernel(Yuv3rArgs a, Yuv2xFrames fr)
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
    const int unit = lin * 4 + wave;                             // (segment, strip) units packed densely
    if (unit >= a.nseg * a.nstrips) return;
    const int seg = __builtin_amdgcn_readfirstlane(unit / a.nstrips);
    const int X0 = (unit - seg * a.nstrips) * D3_STRIP;
    const int y0 = seg * a.segRows, nOut = min(a.segRows, a.dstH - y0);          // y0 is even
    const int T0 = y0 >> 1, nT = ((y0 + nOut + 2) >> 1) - T0 + 1;
    const int srcW = 3 * a.dstW, srcH = 3 * a.dstH, chrH = srcH >> 1;
    const int lLast = 6 * (T0 + nT - 1), cLast = 3 * (T0 + nT - 1);        // the last rows the segment uses: nothing beyond is requested ahead
    const uint8_t *py = fr.y[blockIdx.y], *puv = fr.u[blockIdx.y];
    uint8_t *pd = fr.dst[blockIdx.y];

    const int xo = X0 + 4 * lane;
    const bool active = xo < a.dstW;
    const int xc = active ? xo : a.dstW - 4;                     // idle lanes shadow the last group
    const bool edgeWave = X0 == 0 || 3 * (X0 + D3_STRIP) + 16 > srcW;
    const bool isLeft = xc == 0, isRight = xc == a.dstW - 4;
    const unsigned boL = (unsigned)(3 * xc - 4), lboL = boL + (isLeft ? 4u : 0u) - (isRight ? 4u : 0u);
    const int cc = xc >> 1;                                      // first of the lane's two chroma positions
    const unsigned boC = 2u * (unsigned)(3 * cc - 4), lboC = boC + (isLeft ? 8u : 0u) - (isRight ? 12u : 0u);
    D3Plane PL, PC;                                              // only the horizontal taps are read through these
#pragma unroll
    for (int k = 0; k < 6; k++) { PL.h[k] = a.hL[k]; PC.h[k] = a.hC[k]; }

    auto loadL = [&](int row, unsigned (&d)[5], auto edge_c) {
        const unsigned o = (unsigned)min(max(row, 0), srcH - 1) * (unsigned)a.ys;
        const uint8_t *p = py + (o + (decltype(edge_c)::value ? lboL : boL));
        const uint4 t = d3_ld16(p);
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
        d[4] = d3_ld4(p + 16);
    };
    auto loadC = [&](int row, unsigned (&d)[8], auto edge_c) {
        const unsigned o = (unsigned)min(max(row, 0), chrH - 1) * (unsigned)a.us;
        const uint8_t *p = puv + (o + (decltype(edge_c)::value ? lboC : boC));
        const uint4 t = d3_ld16(p);
        const uint4 u = d3_ld16(p + 16);
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; d[4] = u.x; d[5] = u.y; d[6] = u.z; d[7] = u.w;
    };
    // hScale8To15_c of a luma row: the lane's 4 sums (before >> 7)
    auto hrowL = [&](const unsigned (&src)[5], auto edge_c, int (&s)[4]) {
        unsigned d[5] = {src[0], src[1], src[2], src[3], src[4]};
        if constexpr (decltype(edge_c)::value) {
            const unsigned first = d3_rep(src[0], 0x00000000u), last = d3_rep(src[4], 0x03030303u);
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const unsigned fromLeft = i == 0 ? first : src[i - 1], fromRight = i == 4 ? last : src[i + 1];
                d[i] = isLeft ? fromLeft : isRight ? fromRight : src[i];
            }
        }
        s[0] = d3_hsum<0>(PL, d); s[1] = d3_hsum<1>(PL, d); s[2] = d3_hsum<2>(PL, d); s[3] = d3_hsum<3>(PL, d);
    };
    // ... of a chroma row: U0 V0 U1 V1 as 15-bit lines (>> 7, min 32767)
    auto hrowC = [&](const unsigned (&src)[8], auto edge_c, int (&s)[4]) {
        unsigned d[8];
#pragma unroll
        for (int i = 0; i < 8; i++) d[i] = src[i];
        if constexpr (decltype(edge_c)::value) {
            const unsigned first = d3_rep(src[0], 0x01000100u), last = d3_rep(src[7], 0x03020302u);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const unsigned fromLeft = i < 2 ? first : src[i - 2], fromRight = i > 4 ? last : src[i + 3];
                d[i] = isLeft ? fromLeft : isRight ? fromRight : src[i];
            }
        }
        int u0 = 0, v0 = 0, u1 = 0, v1 = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            u0 = d3_dot2((int)__builtin_amdgcn_perm(0u, d[k], 0x0C020C00u), PC.h[k], u0);
            v0 = d3_dot2((int)__builtin_amdgcn_perm(0u, d[k], 0x0C030C01u), PC.h[k], v0);
            u1 = d3_dot2((int)__builtin_amdgcn_perm(d[k + 2], d[k + 1], 0x0C040C02u), PC.h[k], u1);
            v1 = d3_dot2((int)__builtin_amdgcn_perm(d[k + 2], d[k + 1], 0x0C050C03u), PC.h[k], v1);
        }
        s[0] = min(u0 >> 7, 32767); s[1] = min(v0 >> 7, 32767); s[2] = min(u1 >> 7, 32767); s[3] = min(v1 >> 7, 32767);
    };

    int accL[4][4], accC[4][4];                                  // [slot][sample]: luma 4 columns; chroma U0 V0 U1 V1
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int q = 0; q < 4; q++) accL[s][q] = accC[s][q] = 0;
    unsigned bufL[2][3][5], bufC[3][8];
    const unsigned dstOff = (unsigned)xo * BPP;

    // one RGB row: yuv2rgb_X_c_template's sums (Y unclipped, U / V clamped by the tables' headroom = clip_u8) and table stage
    auto emit = [&](int yo, const int (&YS)[4], const int (&CS)[4]) {
        if (yo < y0 || yo >= y0 + nOut) return;                  // wave-uniform
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
#define D3_B2PAIR(lo, hi) __builtin_amdgcn_perm((hi), (lo), 0x0C0C0602u)
            if (BPP == 4) {
                uint4 o4;
                o4.x = D3_B2PAIR(c0[0], c1[0]) | (D3_B2PAIR(c2[0], 0u) << 16) | 0xFF000000u;
                o4.y = D3_B2PAIR(c0[1], c1[1]) | (D3_B2PAIR(c2[1], 0u) << 16) | 0xFF000000u;
                o4.z = D3_B2PAIR(c0[2], c1[2]) | (D3_B2PAIR(c2[2], 0u) << 16) | 0xFF000000u;
                o4.w = D3_B2PAIR(c0[3], c1[3]) | (D3_B2PAIR(c2[3], 0u) << 16) | 0xFF000000u;
                st_stream(d, o4);
            } else {
                uint3 o3;           // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
                o3.x = D3_B2PAIR(c0[0], c1[0]) | (D3_B2PAIR(c2[0], c0[1]) << 16);
                o3.y = D3_B2PAIR(c1[1], c2[1]) | (D3_B2PAIR(c0[2], c1[2]) << 16);
                o3.z = D3_B2PAIR(c2[2], c0[3]) | (D3_B2PAIR(c1[3], c2[3]) << 16);
                st_stream(d, o3);
            }
#undef D3_B2PAIR
        }
    };

    // the luma triple of sub-step SUB of step i (rows 3u - 2 .. 3u, u = 2T - 1 + SUB): afterwards row u - 2 is complete in slot (U4 + 0) & 3
    auto luma_triple = [&](int T, auto u4_c, auto sub_c, auto edge_c, int (&YS)[4]) {
        constexpr int U4 = decltype(u4_c)::value, SUB = decltype(sub_c)::value;
        int hA[4], hB[4], hC[4];
        hrowL(bufL[SUB][0], edge_c, hA);
        loadL(min(6 * (T + 1) - 5 + 3 * SUB + 0, lLast), bufL[SUB][0], edge_c);
        hrowL(bufL[SUB][1], edge_c, hB);
        loadL(min(6 * (T + 1) - 5 + 3 * SUB + 1, lLast), bufL[SUB][1], edge_c);
        hrowL(bufL[SUB][2], edge_c, hC);
        loadL(min(6 * (T + 1) - 5 + 3 * SUB + 2, lLast), bufL[SUB][2], edge_c);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            // hScale8To15_c: min(val >> 7, 32767) — the pack saturates
            const int ab = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(hA[q] >> 7, hB[q] >> 7));
            const int c = min(hC[q] >> 7, 32767);
            accL[(U4 + 0) & 3][q] = m24(c, a.vS[0]) + d3_dot2(ab, a.vP[0], accL[(U4 + 0) & 3][q]);
            accL[(U4 + 1) & 3][q] = m24(c, a.vS[1]) + d3_dot2(ab, a.vP[1], accL[(U4 + 1) & 3][q]);
            accL[(U4 + 2) & 3][q] = m24(c, a.vS[2]) + d3_dot2(ab, a.vP[2], accL[(U4 + 2) & 3][q]);
            YS[q] = accL[(U4 + 0) & 3][q];
            accL[(U4 + 3) & 3][q] = m24(c, a.vS[3]) + d3_dot2(ab, a.vP[3], a.lr);      // row u + 1 opens
        }
    };
    // one chroma row into the running sums: k0 .. k3 = the taps of the rows in slots S0 .. S3; FRESH: slot S0 opens with this row
    auto chroma_row = [&](const int (&v)[4], auto s0_c, auto s1_c, auto s2_c, auto s3_c, auto fresh_c, int k0, int k1, int k2, int k3) {
        constexpr int S0 = decltype(s0_c)::value, S1 = decltype(s1_c)::value, S2 = decltype(s2_c)::value, S3 = decltype(s3_c)::value;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            accC[S0][q] = m24(v[q], k0) + (decltype(fresh_c)::value ? a.cr : accC[S0][q]);
            accC[S1][q] = m24(v[q], k1) + accC[S1][q];
            accC[S2][q] = m24(v[q], k2) + accC[S2][q];
            accC[S3][q] = m24(v[q], k3) + accC[S3][q];
        }
    };

    auto body = [&](const int i, auto par_c, auto edge_c) {
        constexpr int PAR = decltype(par_c)::value;              // i & 1: names the slots
        const int T = T0 + i;
        // output row 1 has its own taps (odd rows: slot of 2T - 3, 2T - 1, 2T + 1)
        const bool s_m3 = T == 2, s_m1 = T == 1, s_p1 = T == 0;
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        // chroma slots: output row o lives in slot o & 3; with T = T0 + i the row parity pattern only depends on i & 1 (a renaming)
        constexpr int E0 = (2 * PAR) & 3, Em2 = (2 * PAR + 2) & 3, Om1 = (2 * PAR + 3) & 3, Om3 = (2 * PAR + 1) & 3, Op1 = Om3;
        using SE0 = std::integral_constant<int, E0>; using SEm2 = std::integral_constant<int, Em2>;
        using SOm1 = std::integral_constant<int, Om1>; using SOm3 = std::integral_constant<int, Om3>; using SOp1 = std::integral_constant<int, Op1>;
        int YS[4], CS[4], cv[4];
        // ---- first half: luma rows 6T - 5 .. 6T - 3, chroma row 3T - 2 -> output row 2T - 3
        luma_triple(T, std::integral_constant<int, (2 * PAR) & 3>(), I0(), edge_c, YS);
        hrowC(bufC[0], edge_c, cv);
        loadC(min(3 * (T + 1) - 2, cLast), bufC[0], edge_c);
        // row 3T - 2: even 2T opens (A0), even 2T - 2 (A3), odd 2T - 1 (B2), odd 2T - 3 closes (B5)
        chroma_row(cv, SE0(), SEm2(), SOm1(), SOm3(), std::true_type(), a.cA[0], a.cA[3], s_m1 ? a.cS[2] : a.cB[2], s_m3 ? a.cS[5] : a.cB[5]);
#pragma unroll
        for (int q = 0; q < 4; q++) CS[q] = accC[Om3][q];
        emit(2 * T - 3, YS, CS);
        // ---- second half: luma rows 6T - 2 .. 6T, chroma rows 3T - 1, 3T -> output row 2T - 2
        luma_triple(T, std::integral_constant<int, (2 * PAR + 1) & 3>(), I1(), edge_c, YS);
        hrowC(bufC[1], edge_c, cv);
        loadC(min(3 * (T + 1) - 1, cLast), bufC[1], edge_c);
        // row 3T - 1: odd 2T + 1 opens (B0), even 2T (A1), even 2T - 2 (A4), odd 2T - 1 (B3)
        chroma_row(cv, SOp1(), SE0(), SEm2(), SOm1(), std::true_type(), s_p1 ? a.cS[0] : a.cB[0], a.cA[1], a.cA[4], s_m1 ? a.cS[3] : a.cB[3]);
        hrowC(bufC[2], edge_c, cv);
        loadC(min(3 * (T + 1), cLast), bufC[2], edge_c);
        // row 3T: odd 2T + 1 (B1), even 2T (A2), even 2T - 2 closes (A5), odd 2T - 1 (B4)
        chroma_row(cv, SOp1(), SE0(), SEm2(), SOm1(), std::false_type(), s_p1 ? a.cS[1] : a.cB[1], a.cA[2], a.cA[5], s_m1 ? a.cS[4] : a.cB[4]);
#pragma unroll
        for (int q = 0; q < 4; q++) CS[q] = accC[Em2][q];
        emit(2 * T - 2, YS, CS);
    };
    auto run = [&](auto edge_c) {
#pragma unroll
        for (int r = 0; r < 3; r++) { loadL(6 * T0 - 5 + r, bufL[0][r], edge_c); loadL(6 * T0 - 2 + r, bufL[1][r], edge_c); loadC(3 * T0 - 2 + r, bufC[r], edge_c); }
        for (int i0 = 0; i0 < nT; i0 += 2) {
            body(i0, std::integral_constant<int, 0>(), edge_c);
            if (i0 + 1 < nT) body(i0 + 1, std::integral_constant<int, 1>(), edge_c);
        }
    };
    if (edgeWave) run(std::true_type()); else run(std::false_type());
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int yuv3x1_prepare(const ScalePlan &p, const YuvScaleTiling &g, Yuv3x1Tables &t)
{
    t = Yuv3x1Tables();
    const char *off = GMAT_KNOB("GMAT_SCALE_NO_STRIP");
    if (off && atoi(off)) return 0;
    if (g.yuvOut != 1) return 0;
    const bool nv = p.srcFormat == GMAT_PIX_FMT_NV12 && p.dstFormat == GMAT_PIX_FMT_NV12;
    const bool pl = p.srcFormat == GMAT_PIX_FMT_YUV420P && p.dstFormat == GMAT_PIX_FMT_YUV420P;
    if (!nv && !pl) return 0;
    if (p.srcW != 3 * p.dstW || p.srcH != 3 * p.dstH || p.dstW % 8 || p.dstW < 32 || p.dstH < 12 || (p.dstH & 1)) return 0;   // dstH >= 12: the middle chroma row's window is interior
    if (p.chrDstW * 2 != p.dstW || p.chrDstH * 2 != p.dstH || p.chrSrcW != 3 * p.chrDstW || p.chrSrcH != 3 * p.chrDstH) return 0;
    if (!filter_is_edge_replication_ratio(p.hLum, p.srcW, 3, 4, 6, t.hL)) return 0;
    if (!filter_is_edge_replication_ratio(p.hChr, p.chrSrcW, 3, 4, 6, t.hC)) return 0;
    if (!filter_is_edge_replication_ratio(g.vLumEff, p.srcH, 3, 4, 6, t.vL)) return 0;
    if (!filter_is_edge_replication_ratio(g.vChrEff, p.chrSrcH, 3, 4, 6, t.vC)) return 0;
    // the 12th slot of the window (sample 3x + 7, row 3y + 7) must carry no weight: byte 20 of a lane's window is never loaded
    if ((t.hL[5] >> 16) || (t.hC[5] >> 16) || (t.vL[5] >> 16) || (t.vC[5] >> 16)) return 0;
    for (int y = 0; y < p.dstH; y++) if (g.lumRound[y] != g.lumRound[0]) return 0;
    for (int y = 0; y < p.chrDstH; y++) if (g.chrRound[y] != g.chrRound[0]) return 0;
    t.lr = g.lumRound[0]; t.cr = g.chrRound[0];
    t.ok = 1;
    return 0;
}

int launch_scale_yuv3x1(const Yuv3x1Args &a0, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (!frames || nframes < 1 || nframes > kYuv2xMaxFrames) return GMAT_ERR(EINVAL);
    Yuv3x1Args a = a0;
    const char *segStr = GMAT_KNOB("GMAT_STRIP_ROWS");              // tuning / test override (output rows per luma segment), read per launch
    const int segEnv = segStr ? atoi(segStr) : 0;
    const int nstripsL = (a.dstW + D3_STRIP - 1) / D3_STRIP;
    const int nstripsC = a.nv12 ? (a.chrDstW + D3_STRIP_UV - 1) / D3_STRIP_UV : (a.chrDstW + D3_STRIP - 1) / D3_STRIP;
    const int nplC = a.nv12 ? 1 : 2;
    a.nsgL = nstripsL; a.nsgC = nstripsC;                        // strips per row of segments
    int seg = segEnv > 0 ? segEnv : 0;
    if (!seg) {
        // a wave walks 3 seg + 9 source rows: the 9 warm-up rows argue for long segments, filling the chip for short ones
        const long rows = ((long)a.dstH * nstripsL + (long)a.chrDstH * nstripsC * nplC) * nframes;      // wave-rows (output)
        seg = (int)std::min(25L, std::max(6L, (rows + 6143) / 6144));    // 25 rows (28 steps: a multiple of the 4 unrolled) at 32 frames: 3.53 us, 28 rows 3.73
    }
    const char *ud = GMAT_KNOB("GMAT_STRIP_UPDOWN");                 // test / measurement knob: 0 = every segment walks downward
    a.updown = !(ud && !atoi(ud));
    a.segRowsL = seg; a.segRowsC = seg;                          // the same walk length on every plane: equal wave lifetimes, half the chroma warm-up
    a.nsegL = (a.dstH + a.segRowsL - 1) / a.segRowsL;
    a.nsegC = (a.chrDstH + a.segRowsC - 1) / a.segRowsC;
    a.nblkL = (a.nsegL * a.nsgL + 3) / 4;                        // four (segment, strip) units per workgroup
    a.nblk = a.nblkL + (a.nsegC * a.nsgC * nplC + 3) / 4;
    const dim3 grid(a.xcdRemap ? 8 * ((a.nblk + 7) / 8) : a.nblk, nframes), block(256);
    if (a.nv12) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv3x1_kernel<true>), grid, block, 0, stream, a, *frames);
    else        hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv3x1_kernel<false>), grid, block, 0, stream, a, *frames);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// scale_yuv3r_kernel takes an NV12 -> packed RGB context at exactly 3:1 whose four filters have the shapes the kernel assumes
int yuv3r_prepare(const ScalePlan &p, const YuvScaleTiling &g, Yuv3rTables &t)
{
    t = Yuv3rTables();
    const char *off = GMAT_KNOB("GMAT_SCALE_NO_STRIP");
    if (off && atoi(off)) return 0;
    if (g.fullChroma || g.yuvOut) return 0;
    if (p.srcFormat != GMAT_PIX_FMT_NV12) return 0;
    if (!(p.dstFormat == GMAT_PIX_FMT_RGB24 || p.dstFormat == GMAT_PIX_FMT_BGR24 || p.dstFormat == GMAT_PIX_FMT_RGBA ||
          p.dstFormat == GMAT_PIX_FMT_BGRA)) return 0;
    if (p.srcW != 3 * p.dstW || p.srcH != 3 * p.dstH || p.dstW % 4 || p.dstW < 32 || p.dstH < 12 || (p.dstH & 1)) return 0;
    // half-width chroma at the output's full height; the source's chroma is half the source on both axes
    if (p.chrSrcW * 2 != p.srcW || p.chrSrcH * 2 != p.srcH || p.chrDstW * 2 != p.dstW || p.chrDstH != p.dstH) return 0;
    if (!filter_is_edge_replication_ratio(p.hLum, p.srcW, 3, 4, 6, t.hL)) return 0;
    if (!filter_is_edge_replication_ratio(p.hChr, p.chrSrcW, 3, 4, 6, t.hC)) return 0;
    if (!filter_is_edge_replication_ratio(g.vLumEff, p.srcH, 3, 4, 6, t.vL)) return 0;
    if (!down32_axis(g.vChrEff, p.chrSrcH, t.vCA, t.vCB, t.vCS)) return 0;
    if ((t.hL[5] >> 16) || (t.hC[5] >> 16) || (t.vL[5] >> 16)) return 0;        // the 12th slot carries no weight
    for (int y = 0; y < p.dstH; y++) if (g.lumRound[y] != g.lumRound[0]) return 0;
    for (int y = 0; y < p.chrDstH; y++) if (g.chrRound[y] != g.chrRound[0]) return 0;
    t.lr = g.lumRound[0]; t.cr = g.chrRound[0];
    t.ok = 1;
    return 0;
}

int launch_scale_yuv3r(const Yuv3rArgs &a0, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (!frames || nframes < 1 || nframes > kYuv2xMaxFrames) return GMAT_ERR(EINVAL);
    Yuv3rArgs a = a0;
    a.nstrips = (a.dstW + D3_STRIP - 1) / D3_STRIP;
    const char *segStr = GMAT_KNOB("GMAT_STRIP_ROWS");              // tuning / test override: output rows per segment
    int seg = segStr ? atoi(segStr) : 0;
    if (seg <= 0) {
        // a segment of n output rows walks n / 2 + 2 steps of 6 luma + 3 chroma rows: two warm-up steps.  Measured on 4K -> 720p
        // (profiles/r02za_down3rgb.txt): 32 frames per launch 24 - 30 rows 4.3 - 4.4 us per frame (12: 4.7, 48: 4.7); one frame 4 rows 10.7 us (8: 14.4)
        const long rows = (long)a.dstH * a.nstrips * nframes;
        seg = (int)std::min(48L, std::max(4L, (rows + 4095) / 4096));
    }
    seg = (seg + 1) & ~1;                                        // segments start on even output rows
    a.segRows = seg;
    a.nseg = (a.dstH + seg - 1) / seg;
    a.nblk = (a.nseg * a.nstrips + 3) / 4;
    a.xcdRemap = 1;
    const dim3 grid(8 * ((a.nblk + 7) / 8), nframes), block(256);
    switch (a.dstFormat) {
    case GMAT_PIX_FMT_RGB24: hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv3r_kernel<0>), grid, block, 0, stream, a, *frames); break;
    case GMAT_PIX_FMT_BGR24: hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv3r_kernel<1>), grid, block, 0, stream, a, *frames); break;
    case GMAT_PIX_FMT_RGBA:  hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv3r_kernel<2>), grid, block, 0, stream, a, *frames); break;
    case GMAT_PIX_FMT_BGRA:  hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv3r_kernel<3>), grid, block, 0, stream, a, *frames); break;
    default: return GMAT_ERR(EINVAL);
    }
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
