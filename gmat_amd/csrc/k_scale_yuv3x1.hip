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
    asm volatile("global_store_dword %0, %1, %2" : : "v"(off), "v"(v), "s"(base));
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
// host side
// ---------------------------------------------------------------------------------------------
int yuv3x1_prepare(const ScalePlan &p, const YuvScaleTiling &g, Yuv3x1Tables &t)
{
    t = Yuv3x1Tables();
    const char *off = getenv("GMAT_SCALE_NO_STRIP");
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
    const char *segStr = getenv("GMAT_STRIP_ROWS");              // tuning / test override (output rows per luma segment), read per launch
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
    const char *ud = getenv("GMAT_STRIP_UPDOWN");                 // test / measurement knob: 0 = every segment walks downward
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

} // namespace gmat
