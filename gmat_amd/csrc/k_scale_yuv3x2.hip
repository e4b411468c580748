// k_scale_yuv3x2.hip — strip-walking form of the exact 3:2 down-scale of 8-bit YUV 4:2:0 (1080p -> 720p, 4K -> 1440p), NV12 -> NV12
// and YUV420P -> YUV420P, with the arithmetic of ONE libswscale context (hScale8To15_c per plane, yuv2planeX_8_c / yuv2nv12cX_c
// vertically, swscale.c:234-520, output.c:400-450), bit-exact.  The generic plane scaler spends 5.5 us on a 1080p -> 720p frame (0.10
// of the HBM roofline).
//
// At 3:2 the bicubic filter has 6 taps and TWO phases: output 2k reads source [3k - 2, 3k + 3] with coefficients A, output 2k + 1 reads
// [3k - 1, 3k + 4] with B (A mirrored) — and so do the rows: three source rows make two output rows.  libswscale folds taps outside the
// plane onto the edge sample, which is the interior filter on an edge-replicated line for every output but ONE: output 1 (second
// column, second row) has its own row in the table (-752, 6223, 10171, 1554, -812 where folding B gives -910, 6280, 10266, 1567, -819).
// The host checks every table row against that rule and passes A, B and the table's own row of output 1, per axis and plane kind, as
// kernel arguments.
//   * a wave owns a strip of 512 output columns (a lane: 8 adjacent outputs from the 20 source bytes at the 4-aligned offset
//     3 x / 2 - 4: 15 byte pairs by v_perm_b32, 24 v_dot2 per row) and walks down the SOURCE rows in steps of three, the next two steps'
//     rows in flight; every row is packed with the row above by v_cvt_pk_i16_i32 (hScale8To15_c's saturation) into row pairs (n-1 | n);
//   * step T handles rows 3T - 2, 3T - 1, 3T: after the first of them output row 2T - 3 leaves (pairs ending at rows 3T - 6, 3T - 4,
//     3T - 2), after the last one output row 2T - 2 (pairs ending at 3T - 4, 3T - 2, 3T) — five pair sets are alive at any time, their
//     slots static after unrolling two steps;
//   * the design notes of k_scale_yuv3x1.hip apply: units of (segment, strip) packed densely into workgroups, edge lanes loading from a
//     base shifted by the out-of-row dword and repairing their registers, stores issued from inline assembly (counted waits for the
//     loads), prefetch rows clamped to the segment's own rows.
// Parity: held to the oracle (tests/test_parity_down32.py, together with the generic kernel on the same matrix); no vector the reference
// holds is a 3:2 scale.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

namespace gmat {

constexpr int E3_STRIP = 512;                  // output columns per wave of a single-channel plane: 64 lanes x 8
constexpr int E3_STRIP_UV = 256;               // output UV positions per wave of the interleaved plane: 64 lanes x 4

// 4 waves per SIMD = at most 128 VGPRs
#if defined(__HIP__)
#define E3_FOUR_WAVES __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define E3_FOUR_WAVES
#endif
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned e3_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ uint4 e3_ld16(const uint8_t *p) { const e3_u32x4 v = *reinterpret_cast<const e3_u32x4 *>(p); return make_uint4(v.x, v.y, v.z, v.w); }
#else
static inline uint4 e3_ld16(const uint8_t *p) { uint4 v; std::memcpy(&v, p, 16); return v; }
#endif
__device__ __forceinline__ unsigned e3_ld4(const uint8_t *p) { return *reinterpret_cast<const unsigned *>(p); }

// 8 bytes to base + off, issued out of the compiler's sight on the device (see d3_st4 in k_scale_yuv3x1.hip)
__device__ __forceinline__ void e3_st8(uint8_t *base, unsigned off, unsigned lo, unsigned hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned e3_u32x2 __attribute__((ext_vector_type(2)));
    const e3_u32x2 v = {lo, hi};
    asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2" : : "v"(off), "v"(v), "s"(base));    // s_nop 4: see g_st in k_scale_yuvg.hip (VALU-written scalar operand)
#else
    std::memcpy(base + off, &lo, 4); std::memcpy(base + off + 4, &hi, 4);
#endif
}

__device__ __forceinline__ int e3_dot2(int packed_ab, int packed_cd, int acc)      // three-operand v_dot2_i32_i16 (see k_scale_yuv2s.hip)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, packed_ab), __builtin_bit_cast(short2v, packed_cd), acc, true);
}
__device__ __forceinline__ unsigned e3_rep(unsigned v, unsigned sel) { return __builtin_amdgcn_perm(v, v, sel); }
// cond ? a : b on VALUES (a ?: on members of an in-memory struct selects an ADDRESS into scratch memory, see k_scale_yuv1x2.hip)
__device__ __forceinline__ int32_t e3_blend(bool cond, int32_t a, int32_t b) { return b ^ ((a ^ b) & -(int32_t)cond); }

struct E3Plane {
    const uint8_t *src; uint8_t *dst;
    int ss, ds, dstW, srcW, srcH, dstH;        // widths in samples (UV plane: in UV positions)
    int32_t hA[3], hB[3], hS[3];               // horizontal: even outputs, odd outputs, output 1 — 6 taps as int16 pairs
    int32_t vA[3], vB[3], vS[3];               // vertical likewise (output rows)
    int rnd;
};

// bytes O, O + 1 of a lane's 20-byte window d[0 .. 4], widened to an int16 pair
template <int O>
__device__ __forceinline__ int e3_pair(const unsigned (&d)[5])
{
    constexpr int dw = O >> 2, b = O & 3;
    static_assert(O >= 0 && O <= 18, "pair outside the window");
    if constexpr (b < 3) return (int)__builtin_amdgcn_perm(0u, d[dw], 0x0C000C00u | ((unsigned)(b + 1) << 16) | (unsigned)b);
    else return (int)__builtin_amdgcn_perm(d[dw + 1], d[dw], 0x0C040C03u);
}

// The walk shared by both plane kinds: LOAD(row, d, edge_c) / HROW(d, edge_c, s[8]) / STORE(y, w[8]) are the plane kind's.
// The segment makes the output rows [y0, y0 + nOut), y0 even.  Step T of the segment is step n0 + T of the plane (n0 = y0 / 2).
template <typename Load, typename HRow, typename Store>
__device__ __forceinline__ void e3_walk(const E3Plane &P, int y0, int nOut, bool edgeWave, Load &&load, HRow &&hrow, Store &&store)
{
    // pair sets: r0 = (3T-3 | 3T-2) of the current step; r1[T & 1] = (3T-2 | 3T-1); r2[T & 1] = (3T-1 | 3T)
    int r0[8], r1[2][8], r2[2][8], prev[8];
#pragma unroll
    for (int q = 0; q < 8; q++) prev[q] = r0[q] = r1[0][q] = r1[1][q] = r2[0][q] = r2[1][q] = 0;
    unsigned buf[2][3][5];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int i = 0; i < 5; i++) buf[a][r][i] = 0u;
    const int n0 = y0 >> 1;
    const int nSteps = (nOut >> 1) + 2;                          // the last step contributes its first row only (output row y0 + nOut - 1)
    const int nStart = 3 * n0 - 2;                              // first row of step 0
    const int nLast = 3 * (n0 + nSteps - 1) - 2;                // last row the segment needs: the first row of its last step
    const int yEnd = y0 + nOut;
    auto rowOf = [&](int i) { return min(nStart + i, nLast); };  // never past the segment's rows (plane borders are clamped in LOAD)

    auto emit = [&](int y, const int (&p0)[8], const int (&p1)[8], const int (&p2)[8], int32_t c0, int32_t c1, int32_t c2) {
        unsigned w[8];
#pragma unroll
        for (int q = 0; q < 8; q++)
            w[q] = (unsigned)clip_u8_shr(e3_dot2(p2[q], c2, e3_dot2(p1[q], c1, e3_dot2(p0[q], c0, P.rnd))), 19);
        store(y, w);
    };
    auto body = [&](const int s, auto par_c, auto edge_c) {
        constexpr int PAR = decltype(par_c)::value;                // s & 1: load buffers and the slots of r1 / r2 this step writes
#pragma unroll
        for (int r = 0; r < 3; r++) {
            int hs[8];
            hrow(buf[PAR][r], edge_c, hs);
            load(rowOf(3 * (s + 2) + r), buf[PAR][r], edge_c);      // rolling prefetch: the same row of step s + 2
            int (&dst)[8] = r == 0 ? r0 : r == 1 ? r1[PAR] : r2[PAR];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int cur = hs[q] >> 7;                     // hScale8To15_c: min(val >> 7, 32767) — the pack saturates
                dst[q] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(prev[q], cur));
                prev[q] = cur;
            }
            if (r == 0) {
                // odd output row 2T - 3 (T = n0 + s): pairs ending at rows 3T - 6 (r2 of step T - 2), 3T - 4 (r1 of T - 1), 3T - 2 (r0)
                const int y = y0 + 2 * s - 3;
                if (y >= y0 && y < yEnd) {
                    const bool sp = y == 1;                     // wave-uniform: the table's own row of output row 1
                    emit(y, r2[PAR], r1[PAR ^ 1], r0, e3_blend(sp, P.vS[0], P.vB[0]), e3_blend(sp, P.vS[1], P.vB[1]), e3_blend(sp, P.vS[2], P.vB[2]));
                }
            }
            if (r == 2) {
                // even output row 2T - 2: pairs ending at rows 3T - 4 (r1 of step T - 1), 3T - 2 (r0), 3T (r2 of this step)
                const int y = y0 + 2 * s - 2;
                if (y >= y0 && y < yEnd) emit(y, r1[PAR ^ 1], r0, r2[PAR], P.vA[0], P.vA[1], P.vA[2]);
            }
        }
    };
    auto run = [&](auto edge_c) {
#pragma unroll
        for (int r = 0; r < 3; r++) load(rowOf(r), buf[0][r], edge_c);
#pragma unroll
        for (int r = 0; r < 3; r++) load(rowOf(3 + r), buf[1][r], edge_c);
        for (int s0 = 0; s0 < nSteps; s0 += 2) {
            body(s0, std::integral_constant<int, 0>(), edge_c);
            if (s0 + 1 < nSteps) body(s0 + 1, std::integral_constant<int, 1>(), edge_c);
        }
    };
    // prev must hold row 3 n0 - 3 before the first pair is formed, but that pair is never used: any value does
    if (edgeWave) run(std::true_type()); else run(std::false_type());
}

// ---- one single-channel plane: the output rows [y0, y0 + nOut) of the strip at X0 ------------------------------------------
__device__ __forceinline__ void e3_walk_plane(const E3Plane &P, int X0, int y0, int nOut, int lane)
{
    const int xo = X0 + 8 * lane;
    const bool active = xo < P.dstW;
    const int xc = active ? xo : P.dstW - 8;                    // idle lanes shadow the last group
    const bool edgeWave = X0 == 0 || 3 * (X0 + E3_STRIP) / 2 + 8 > P.srcW;     // a window of this wave may leave the row
    const unsigned bo = (unsigned)(3 * (xc >> 1) - 4);          // byte offset of the window base, a multiple of 4 (negative in the lane at x = 0)
    // edge waves: the lane at x = 0 (its window starts one dword before the row) and the lanes of the last group (theirs ends one dword
    // after it) load the same 20 bytes one dword further in / out and shift the registers back, the dword outside = the edge sample
    const bool isLeft = xc == 0, isRight = xc == P.dstW - 8;
    const unsigned lbo = bo + (isLeft ? 4u : 0u) - (isRight ? 4u : 0u);
    // output 1 of the row (lane 0 of the first strip, its second output) takes the table's own coefficients
    const bool first = xc == 0;
    const int32_t b1a = e3_blend(first, P.hS[0], P.hB[0]), b1b = e3_blend(first, P.hS[1], P.hB[1]), b1c = e3_blend(first, P.hS[2], P.hB[2]);

    auto load = [&](int row, unsigned (&d)[5], auto edge_c) {
        const unsigned o = (unsigned)min(max(row, 0), P.srcH - 1) * (unsigned)P.ss;
        const uint8_t *p = P.src + (o + (decltype(edge_c)::value ? lbo : bo));      // interior waves: bo >= 0
        const uint4 t = e3_ld16(p);
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
        d[4] = e3_ld4(p + 16);
    };
    // window byte b = source sample 3 (xc / 2) - 4 + b; output 2i of the lane reads bytes 3i + 2 .. 3i + 7, output 2i + 1 bytes 3i + 3 .. 3i + 8
    auto hrow = [&](const unsigned (&src)[5], auto edge_c, int (&s)[8]) {
        unsigned d[5] = {src[0], src[1], src[2], src[3], src[4]};
        if constexpr (decltype(edge_c)::value) {
            const unsigned firstS = e3_rep(src[0], 0x00000000u), lastS = e3_rep(src[4], 0x03030303u);
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const unsigned fromLeft = i == 0 ? firstS : src[i - 1], fromRight = i == 4 ? lastS : src[i + 1];
                d[i] = isLeft ? fromLeft : isRight ? fromRight : src[i];
            }
        }
        s[0] = e3_dot2(e3_pair<6>(d), P.hA[2], e3_dot2(e3_pair<4>(d), P.hA[1], e3_dot2(e3_pair<2>(d), P.hA[0], 0)));
        s[1] = e3_dot2(e3_pair<7>(d), b1c, e3_dot2(e3_pair<5>(d), b1b, e3_dot2(e3_pair<3>(d), b1a, 0)));
        s[2] = e3_dot2(e3_pair<9>(d), P.hA[2], e3_dot2(e3_pair<7>(d), P.hA[1], e3_dot2(e3_pair<5>(d), P.hA[0], 0)));
        s[3] = e3_dot2(e3_pair<10>(d), P.hB[2], e3_dot2(e3_pair<8>(d), P.hB[1], e3_dot2(e3_pair<6>(d), P.hB[0], 0)));
        s[4] = e3_dot2(e3_pair<12>(d), P.hA[2], e3_dot2(e3_pair<10>(d), P.hA[1], e3_dot2(e3_pair<8>(d), P.hA[0], 0)));
        s[5] = e3_dot2(e3_pair<13>(d), P.hB[2], e3_dot2(e3_pair<11>(d), P.hB[1], e3_dot2(e3_pair<9>(d), P.hB[0], 0)));
        s[6] = e3_dot2(e3_pair<15>(d), P.hA[2], e3_dot2(e3_pair<13>(d), P.hA[1], e3_dot2(e3_pair<11>(d), P.hA[0], 0)));
        s[7] = e3_dot2(e3_pair<16>(d), P.hB[2], e3_dot2(e3_pair<14>(d), P.hB[1], e3_dot2(e3_pair<12>(d), P.hB[0], 0)));
    };
    auto store = [&](int y, const unsigned (&w)[8]) {
        if (active) e3_st8(P.dst, (unsigned)y * (unsigned)P.ds + (unsigned)xo, w[0] | (w[1] << 8) | (w[2] << 16) | (w[3] << 24),
                           w[4] | (w[5] << 8) | (w[6] << 16) | (w[7] << 24));
    };
    e3_walk(P, y0, nOut, edgeWave, load, hrow, store);
}

// ---- NV12's interleaved UV plane: a lane makes 4 UV output positions (8 bytes) from 10 source positions (5 dwords) -------------
__device__ __forceinline__ void e3_walk_uv(const E3Plane &P, int X0, int y0, int nOut, int lane)
{
    const int co = X0 + 4 * lane;
    const bool active = co < P.dstW;
    const int cc = active ? co : P.dstW - 4;
    const bool edgeWave = X0 == 0 || 3 * (X0 + E3_STRIP_UV) / 2 + 4 > P.srcW;
    const unsigned bo = 2u * (unsigned)(3 * (cc >> 1) - 2);     // byte offset of the window base (position 3 cc / 2 - 2): a multiple of 4
    const bool isLeft = cc == 0, isRight = cc == P.dstW - 4;
    const unsigned lbo = bo + (isLeft ? 4u : 0u) - (isRight ? 4u : 0u);
    const bool first = cc == 0;
    const int32_t b1a = e3_blend(first, P.hS[0], P.hB[0]), b1b = e3_blend(first, P.hS[1], P.hB[1]), b1c = e3_blend(first, P.hS[2], P.hB[2]);

    auto load = [&](int row, unsigned (&d)[5], auto edge_c) {
        const unsigned o = (unsigned)min(max(row, 0), P.srcH - 1) * (unsigned)P.ss;
        const uint8_t *p = P.src + (o + (decltype(edge_c)::value ? lbo : bo));
        const uint4 t = e3_ld16(p);
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
        d[4] = e3_ld4(p + 16);
    };
    // window position p = source position 3 (cc / 2) - 2 + p = dword p / 2, half p % 2 (U: byte 0, V: byte 1 of the half).
    // output position 2i of the lane reads positions 3i .. 3i + 5, output 2i + 1 positions 3i + 1 .. 3i + 6
    auto hrow = [&](const unsigned (&src)[5], auto edge_c, int (&s)[8]) {
        unsigned d[5] = {src[0], src[1], src[2], src[3], src[4]};
        if constexpr (decltype(edge_c)::value) {
            const unsigned firstS = e3_rep(src[0], 0x01000100u), lastS = e3_rep(src[4], 0x03020302u);
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const unsigned fromLeft = i == 0 ? firstS : src[i - 1], fromRight = i == 4 ? lastS : src[i + 1];
                d[i] = isLeft ? fromLeft : isRight ? fromRight : src[i];
            }
        }
        int pU[9], pV[9];                                       // the position pairs (p, p + 1), p = 0 .. 8, per channel
#pragma unroll
        for (int p = 0; p < 9; p++) {
            if (p & 1) {
                pU[p] = (int)__builtin_amdgcn_perm(d[(p + 1) >> 1], d[p >> 1], 0x0C040C02u);
                pV[p] = (int)__builtin_amdgcn_perm(d[(p + 1) >> 1], d[p >> 1], 0x0C050C03u);
            } else {
                pU[p] = (int)__builtin_amdgcn_perm(0u, d[p >> 1], 0x0C020C00u);
                pV[p] = (int)__builtin_amdgcn_perm(0u, d[p >> 1], 0x0C030C01u);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int e = 3 * i, o = 3 * i + 1;
            const int32_t oa = i == 0 ? b1a : P.hB[0], ob = i == 0 ? b1b : P.hB[1], oc = i == 0 ? b1c : P.hB[2];
            s[4 * i + 0] = e3_dot2(pU[e + 4], P.hA[2], e3_dot2(pU[e + 2], P.hA[1], e3_dot2(pU[e], P.hA[0], 0)));
            s[4 * i + 1] = e3_dot2(pV[e + 4], P.hA[2], e3_dot2(pV[e + 2], P.hA[1], e3_dot2(pV[e], P.hA[0], 0)));
            s[4 * i + 2] = e3_dot2(pU[o + 4], oc, e3_dot2(pU[o + 2], ob, e3_dot2(pU[o], oa, 0)));
            s[4 * i + 3] = e3_dot2(pV[o + 4], oc, e3_dot2(pV[o + 2], ob, e3_dot2(pV[o], oa, 0)));
        }
    };
    auto store = [&](int y, const unsigned (&w)[8]) {
        if (active) e3_st8(P.dst, (unsigned)y * (unsigned)P.ds + 2u * (unsigned)co, w[0] | (w[1] << 8) | (w[2] << 16) | (w[3] << 24),
                           w[4] | (w[5] << 8) | (w[6] << 16) | (w[7] << 24));
    };
    e3_walk(P, y0, nOut, edgeWave, load, hrow, store);
}

__device__ __forceinline__ E3Plane e3_plane(const uint8_t *src, uint8_t *dst, int ss, int ds, int dstW, int dstH,
                                            const int32_t (&hA)[3], const int32_t (&hB)[3], const int32_t (&hS)[3],
                                            const int32_t (&vA)[3], const int32_t (&vB)[3], const int32_t (&vS)[3], int rnd)
{
    E3Plane P;
    P.src = src; P.dst = dst; P.ss = ss; P.ds = ds; P.dstW = dstW; P.dstH = dstH; P.srcW = 3 * (dstW >> 1); P.srcH = 3 * (dstH >> 1); P.rnd = rnd;
#pragma unroll
    for (int k = 0; k < 3; k++) { P.hA[k] = hA[k]; P.hB[k] = hB[k]; P.hS[k] = hS[k]; P.vA[k] = vA[k]; P.vB[k] = vB[k]; P.vS[k] = vS[k]; }
    return P;
}

// blockIdx.x: [0, nblkL) luma workgroups, then the chroma workgroups; a wave's unit of work is one (segment, strip) pair, packed
// densely (unit = 4 * workgroup + wave, segment-major).  blockIdx.y = frame.  A segment is segRows output rows (even).
template <bool NV>
__global__ __launch_bounds__(256) E3_FOUR_WAVES void scale_yuv3x2_kernel(Yuv3x2Args a, Yuv2xFrames fr)
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
        const int X0 = (unit - seg * a.nsgL) * E3_STRIP;
        const int y0 = seg * a.segRowsL;
        const E3Plane P = e3_plane(fr.y[f], fr.dst[f], a.ys, a.ds, a.dstW, a.dstH, a.hLA, a.hLB, a.hLS, a.vLA, a.vLB, a.vLS, a.lr);
        e3_walk_plane(P, X0, y0, min(a.segRowsL, a.dstH - y0), lane);
        return;
    }
    int unit = (lin - a.nblkL) * 4 + wave;
    const int per = a.nsegC * a.nsgC;                            // units of one chroma plane
    if (NV) {
        if (unit >= per) return;
        const int seg = __builtin_amdgcn_readfirstlane(unit / a.nsgC);
        const int X0 = (unit - seg * a.nsgC) * E3_STRIP_UV;
        const int y0 = seg * a.segRowsC;
        const E3Plane P = e3_plane(fr.u[f], fr.dstU[f], a.us, a.dsU, a.chrDstW, a.chrDstH, a.hCA, a.hCB, a.hCS, a.vCA, a.vCB, a.vCS, a.cr);
        e3_walk_uv(P, X0, y0, min(a.segRowsC, a.chrDstH - y0), lane);
    } else {
        if (unit >= 2 * per) return;
        const int pl = __builtin_amdgcn_readfirstlane(unit >= per ? 1 : 0);
        unit -= pl * per;
        const int seg = __builtin_amdgcn_readfirstlane(unit / a.nsgC);
        const int X0 = (unit - seg * a.nsgC) * E3_STRIP;
        const int y0 = seg * a.segRowsC;
        const E3Plane P = e3_plane(pl ? fr.v[f] : fr.u[f], pl ? fr.dstV[f] : fr.dstU[f], pl ? a.vs : a.us, pl ? a.dsV : a.dsU,
                                   a.chrDstW, a.chrDstH, a.hCA, a.hCB, a.hCS, a.vCA, a.vCB, a.vCS, a.cr);
        e3_walk_plane(P, X0, y0, min(a.segRowsC, a.chrDstH - y0), lane);
    }
}

// ---------------------------------------------------------------------------------------------
// scale_yuv32r_kernel: NV12 at two thirds of the size into packed RGB (1080p -> 720p, 4K -> 1440p), ONE libswscale context.  The lane
// mapping and both horizontal filters are the plane walkers' above: 8 luma outputs from a 20-byte window, 4 chroma positions (an RGB
// destination keeps half-width chroma) from a 20-byte UV window.  Vertically the luma is the 3:2 filter as RUNNING SUMS (as the
// chroma of scale_yuv3r_kernel, k_scale_yuv3x1.hip: a row feeds the four open output rows, A / B / S taps by the row's place in step T,
// output rows 2T - 3 and 2T - 2 close after the first and the last row of the step); the chroma — half the source's rows into all of
// the destination's — is a 3:4 UP-scale: 4 taps, four phases, output row 4k + 2 + i on the chroma rows 3k + (0, 1, 1, 2)[i] .. + 3, rows
// 0 and 1 with the table's own rows.  The last four chroma lines sit in a sliding window, U and V of a position packed in one register
// (v_dot2 with the tap in one half picks the channel); two steps push three chroma rows.  Segments start on multiples of 4 output rows.
// ---------------------------------------------------------------------------------------------
// 171 VGPRs = two waves per SIMD.  Bounded to 168 (three waves) the allocator spills 8 - 12 bytes per lane and the kernel is no faster
// (2.23 against 2.16 us per 1080p frame, one frame per launch 13.8 against 12.3: profiles/r02zc_down32rgb.txt)
template <int DST>
__global__ __launch_bounds__(256) void scale_yuv32r_kernel(Yuv32rArgs a, Yuv2xFrames fr)
{
    constexpr bool BGR = (DST & 1) != 0;
    constexpr int BPP = DST >= 2 ? 4 : 3;
    __shared__ int2 lutV[256], lutU[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
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
    const int unit = lin * 4 + wave;
    if (unit >= a.nseg * a.nstrips) return;
    const int seg = __builtin_amdgcn_readfirstlane(unit / a.nstrips);
    const int X0 = (unit - seg * a.nstrips) * E3_STRIP;
    const int y0 = seg * a.segRows, nOut = min(a.segRows, a.dstH - y0);          // y0 is a multiple of 4
    const int T0 = y0 >> 1, nT = ((y0 + nOut + 2) >> 1) - T0 + 1;                // T0 is even
    const int lLast = 3 * (T0 + nT - 1), cLast = 3 * ((T0 + nT - 1) >> 1) + 1;  // the last luma / chroma rows the segment uses: nothing beyond is requested ahead
    const int srcW = 3 * (a.dstW >> 1), srcH = 3 * (a.dstH >> 1), chrH = srcH >> 1;
    const uint8_t *py = fr.y[blockIdx.y], *puv = fr.u[blockIdx.y];
    uint8_t *pd = fr.dst[blockIdx.y];

    const int xo = X0 + 8 * lane;
    const bool active = xo < a.dstW;
    const int xc = active ? xo : a.dstW - 8;
    const bool edgeWave = X0 == 0 || 3 * (X0 + E3_STRIP) / 2 + 8 > srcW;
    const bool isLeft = xc == 0, isRight = xc == a.dstW - 8;
    const unsigned boL = (unsigned)(3 * (xc >> 1) - 4), lboL = boL + (isLeft ? 4u : 0u) - (isRight ? 4u : 0u);
    const int cc = xc >> 1;                                      // first of the lane's four chroma positions
    const unsigned boC = 2u * (unsigned)(3 * (cc >> 1) - 2), lboC = boC + (isLeft ? 4u : 0u) - (isRight ? 4u : 0u);
    // output column 1 (of luma and of chroma) takes the table's own row
    const int32_t l1a = e3_blend(isLeft, a.hLS[0], a.hLB[0]), l1b = e3_blend(isLeft, a.hLS[1], a.hLB[1]), l1c = e3_blend(isLeft, a.hLS[2], a.hLB[2]);
    const int32_t c1a = e3_blend(isLeft, a.hCS[0], a.hCB[0]), c1b = e3_blend(isLeft, a.hCS[1], a.hCB[1]), c1c = e3_blend(isLeft, a.hCS[2], a.hCB[2]);

    auto load5 = [&](const uint8_t *base, int stride, int row, int rows, unsigned off, unsigned (&d)[5]) {
        const uint8_t *p = base + ((unsigned)min(max(row, 0), rows - 1) * (unsigned)stride + off);
        const uint4 t = e3_ld16(p);
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
        d[4] = e3_ld4(p + 16);
    };
    auto loadL = [&](int row, unsigned (&d)[5], auto edge_c) { load5(py, a.ys, row, srcH, decltype(edge_c)::value ? lboL : boL, d); };
    auto loadC = [&](int row, unsigned (&d)[5], auto edge_c) { load5(puv, a.us, row, chrH, decltype(edge_c)::value ? lboC : boC, d); };
    // hScale8To15_c of a luma row: 8 lines (>> 7, min 32767)
    auto hrowL = [&](const unsigned (&src)[5], auto edge_c, int (&s)[8]) {
        unsigned d[5] = {src[0], src[1], src[2], src[3], src[4]};
        if constexpr (decltype(edge_c)::value) {
            const unsigned firstS = e3_rep(src[0], 0x00000000u), lastS = e3_rep(src[4], 0x03030303u);
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const unsigned fromLeft = i == 0 ? firstS : src[i - 1], fromRight = i == 4 ? lastS : src[i + 1];
                d[i] = isLeft ? fromLeft : isRight ? fromRight : src[i];
            }
        }
        s[0] = e3_dot2(e3_pair<6>(d), a.hLA[2], e3_dot2(e3_pair<4>(d), a.hLA[1], e3_dot2(e3_pair<2>(d), a.hLA[0], 0)));
        s[1] = e3_dot2(e3_pair<7>(d), l1c, e3_dot2(e3_pair<5>(d), l1b, e3_dot2(e3_pair<3>(d), l1a, 0)));
        s[2] = e3_dot2(e3_pair<9>(d), a.hLA[2], e3_dot2(e3_pair<7>(d), a.hLA[1], e3_dot2(e3_pair<5>(d), a.hLA[0], 0)));
        s[3] = e3_dot2(e3_pair<10>(d), a.hLB[2], e3_dot2(e3_pair<8>(d), a.hLB[1], e3_dot2(e3_pair<6>(d), a.hLB[0], 0)));
        s[4] = e3_dot2(e3_pair<12>(d), a.hLA[2], e3_dot2(e3_pair<10>(d), a.hLA[1], e3_dot2(e3_pair<8>(d), a.hLA[0], 0)));
        s[5] = e3_dot2(e3_pair<13>(d), a.hLB[2], e3_dot2(e3_pair<11>(d), a.hLB[1], e3_dot2(e3_pair<9>(d), a.hLB[0], 0)));
        s[6] = e3_dot2(e3_pair<15>(d), a.hLA[2], e3_dot2(e3_pair<13>(d), a.hLA[1], e3_dot2(e3_pair<11>(d), a.hLA[0], 0)));
        s[7] = e3_dot2(e3_pair<16>(d), a.hLB[2], e3_dot2(e3_pair<14>(d), a.hLB[1], e3_dot2(e3_pair<12>(d), a.hLB[0], 0)));
#pragma unroll
        for (int q = 0; q < 8; q++) s[q] = min(s[q] >> 7, 32767);
    };
    // ... of a chroma row: 4 positions, (U | V << 16) each
    auto hrowC = [&](const unsigned (&src)[5], auto edge_c, int (&w)[4]) {
        unsigned d[5] = {src[0], src[1], src[2], src[3], src[4]};
        if constexpr (decltype(edge_c)::value) {
            const unsigned firstS = e3_rep(src[0], 0x01000100u), lastS = e3_rep(src[4], 0x03020302u);
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const unsigned fromLeft = i == 0 ? firstS : src[i - 1], fromRight = i == 4 ? lastS : src[i + 1];
                d[i] = isLeft ? fromLeft : isRight ? fromRight : src[i];
            }
        }
        int pU[9], pV[9];
#pragma unroll
        for (int p = 0; p < 9; p++) {
            if (p & 1) {
                pU[p] = (int)__builtin_amdgcn_perm(d[(p + 1) >> 1], d[p >> 1], 0x0C040C02u);
                pV[p] = (int)__builtin_amdgcn_perm(d[(p + 1) >> 1], d[p >> 1], 0x0C050C03u);
            } else {
                pU[p] = (int)__builtin_amdgcn_perm(0u, d[p >> 1], 0x0C020C00u);
                pV[p] = (int)__builtin_amdgcn_perm(0u, d[p >> 1], 0x0C030C01u);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int e = 3 * i, o = 3 * i + 1;
            const int32_t oa = i == 0 ? c1a : a.hCB[0], ob = i == 0 ? c1b : a.hCB[1], oc = i == 0 ? c1c : a.hCB[2];
            const int ue = e3_dot2(pU[e + 4], a.hCA[2], e3_dot2(pU[e + 2], a.hCA[1], e3_dot2(pU[e], a.hCA[0], 0)));
            const int ve = e3_dot2(pV[e + 4], a.hCA[2], e3_dot2(pV[e + 2], a.hCA[1], e3_dot2(pV[e], a.hCA[0], 0)));
            const int uo = e3_dot2(pU[o + 4], oc, e3_dot2(pU[o + 2], ob, e3_dot2(pU[o], oa, 0)));
            const int vo = e3_dot2(pV[o + 4], oc, e3_dot2(pV[o + 2], ob, e3_dot2(pV[o], oa, 0)));
            // hScale8To15_c's min(val >> 7, 32767) is the pack's saturation
            w[2 * i + 0] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(ue >> 7, ve >> 7));
            w[2 * i + 1] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(uo >> 7, vo >> 7));
        }
    };

    int accL[4][8];                                              // open luma rows x the lane's 8 columns
    int win[4][4];                                               // the last four chroma lines x 4 positions, (U | V << 16)
#pragma unroll
    for (int s = 0; s < 4; s++) {
#pragma unroll
        for (int q = 0; q < 8; q++) accL[s][q] = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) win[s][q] = 0;
    }
    unsigned bufL[2][3][5], bufC[3][5];
    const unsigned dstOff = (unsigned)xo * BPP;

    auto push = [&](const unsigned (&raw)[5], auto edge_c) {
        int w[4];
        hrowC(raw, edge_c, w);
#pragma unroll
        for (int q = 0; q < 4; q++) { win[0][q] = win[1][q]; win[1][q] = win[2][q]; win[2][q] = win[3][q]; win[3][q] = w[q]; }
    };
    // one luma row into the running sums (see chroma_row of scale_yuv3r_kernel)
    auto luma_row = [&](const int (&v)[8], auto s0_c, auto s1_c, auto s2_c, auto s3_c, auto fresh_c, int k0, int k1, int k2, int k3) {
        constexpr int S0 = decltype(s0_c)::value, S1 = decltype(s1_c)::value, S2 = decltype(s2_c)::value, S3 = decltype(s3_c)::value;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            accL[S0][q] = m24(v[q], k0) + (decltype(fresh_c)::value ? a.lr : accL[S0][q]);
            accL[S1][q] = m24(v[q], k1) + accL[S1][q];
            accL[S2][q] = m24(v[q], k2) + accL[S2][q];
            accL[S3][q] = m24(v[q], k3) + accL[S3][q];
        }
    };
    // one RGB row from the closed luma sums in slot SL and the chroma window under the taps k0 .. k3
    auto emit = [&](int yo, auto sl_c, int k0, int k1, int k2, int k3) {
        constexpr int SL = decltype(sl_c)::value;
        if (yo < y0 || yo >= y0 + nOut) return;                  // wave-uniform
        int iU[4], iV[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int U = e3_dot2(win[3][c], k3 & 0xFFFF, e3_dot2(win[2][c], k2 & 0xFFFF, e3_dot2(win[1][c], k1 & 0xFFFF, e3_dot2(win[0][c], k0 & 0xFFFF, a.cr))));
            const int V = e3_dot2(win[3][c], k3 << 16, e3_dot2(win[2][c], k2 << 16, e3_dot2(win[1][c], k1 << 16, e3_dot2(win[0][c], k0 << 16, a.cr))));
            iU[c] = clip_u8_shr(U, 19); iV[c] = clip_u8_shr(V, 19);
        }
        unsigned c0[8], c1[8], c2[8];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int2 tv = lutV[iV[c]], tu = lutU[iU[c]];
            const int tr = BGR ? tu.y : tv.x, tg = tv.y + tu.x, tb = BGR ? tv.x : tu.y;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int q = 2 * c + h;
                const int yc = m24(accL[SL][q] >> 19, a.y2r.cy);
                c0[q] = (unsigned)min(max(tr + yc, 0), 0xFFFFFF);
                c1[q] = (unsigned)min(max(tg + yc, 0), 0xFFFFFF);
                c2[q] = (unsigned)min(max(tb + yc, 0), 0xFFFFFF);
            }
        }
        if (active) {
            uint8_t *d = pd + (unsigned)((unsigned)yo * (unsigned)a.ds + dstOff);
#define E3_B2PAIR(lo, hi) __builtin_amdgcn_perm((hi), (lo), 0x0C0C0602u)
#pragma unroll
            for (int g = 0; g < 2; g++) {
                const int b = 4 * g;
                if (BPP == 4) {
                    uint4 o4;
                    o4.x = E3_B2PAIR(c0[b + 0], c1[b + 0]) | (E3_B2PAIR(c2[b + 0], 0u) << 16) | 0xFF000000u;
                    o4.y = E3_B2PAIR(c0[b + 1], c1[b + 1]) | (E3_B2PAIR(c2[b + 1], 0u) << 16) | 0xFF000000u;
                    o4.z = E3_B2PAIR(c0[b + 2], c1[b + 2]) | (E3_B2PAIR(c2[b + 2], 0u) << 16) | 0xFF000000u;
                    o4.w = E3_B2PAIR(c0[b + 3], c1[b + 3]) | (E3_B2PAIR(c2[b + 3], 0u) << 16) | 0xFF000000u;
                    st_stream(d + 16 * g, o4);
                } else {
                    uint3 o3;
                    o3.x = E3_B2PAIR(c0[b + 0], c1[b + 0]) | (E3_B2PAIR(c2[b + 0], c0[b + 1]) << 16);
                    o3.y = E3_B2PAIR(c1[b + 1], c2[b + 1]) | (E3_B2PAIR(c0[b + 2], c1[b + 2]) << 16);
                    o3.z = E3_B2PAIR(c2[b + 2], c0[b + 3]) | (E3_B2PAIR(c1[b + 3], c2[b + 3]) << 16);
                    st_stream(d + 12 * g, o3);
                }
            }
#undef E3_B2PAIR
        }
    };

    auto body = [&](const int i, auto par_c, auto edge_c) {
        constexpr int PAR = decltype(par_c)::value;              // i & 1 = T & 1 (T0 is even)
        const int T = T0 + i;
        const bool s_m3 = T == 2, s_m1 = T == 1, s_p1 = T == 0;  // the luma taps of output row 1 (rows 2T - 3, 2T - 1, 2T + 1)
        constexpr int E0 = (2 * PAR) & 3, Em2 = (2 * PAR + 2) & 3, Om1 = (2 * PAR + 3) & 3, Om3 = (2 * PAR + 1) & 3, Op1 = Om3;
        using SE0 = std::integral_constant<int, E0>; using SEm2 = std::integral_constant<int, Em2>;
        using SOm1 = std::integral_constant<int, Om1>; using SOm3 = std::integral_constant<int, Om3>; using SOp1 = std::integral_constant<int, Op1>;
        int v[8];
        // luma row 3T - 2: even 2T opens (A0), even 2T - 2 (A3), odd 2T - 1 (B2), odd 2T - 3 closes (B5)
        hrowL(bufL[PAR][0], edge_c, v);
        loadL(min(3 * (T + 2) - 2, lLast), bufL[PAR][0], edge_c);
        luma_row(v, SE0(), SEm2(), SOm1(), SOm3(), std::true_type(), a.lA[0], a.lA[3], s_m1 ? a.lS[2] : a.lB[2], s_m3 ? a.lS[5] : a.lB[5]);
        if (PAR == 0) {
            // T = 2k: chroma row 3k - 1 -> output row 4k - 3 (phase 3; row 1 at T = 2 has its own taps)
            const int kap = T >> 1;
            push(bufC[0], edge_c);
            loadC(min(3 * (kap + 1) - 1, cLast), bufC[0], edge_c);
            emit(2 * T - 3, SOm3(), s_m3 ? a.cS1[0] : a.cP[3][0], s_m3 ? a.cS1[1] : a.cP[3][1], s_m3 ? a.cS1[2] : a.cP[3][2], s_m3 ? a.cS1[3] : a.cP[3][3]);
        } else {
            // T = 2k + 1: chroma row 3k + 1 -> output rows 4k - 1 (phase 1) and 4k (phase 2; row 0 at T = 1 has its own taps)
            const int kap = T >> 1;
            push(bufC[2], edge_c);
            loadC(min(3 * (kap + 1) + 1, cLast), bufC[2], edge_c);
            emit(2 * T - 3, SOm3(), a.cP[1][0], a.cP[1][1], a.cP[1][2], a.cP[1][3]);
        }
        // luma row 3T - 1: odd 2T + 1 opens (B0), even 2T (A1), even 2T - 2 (A4), odd 2T - 1 (B3)
        hrowL(bufL[PAR][1], edge_c, v);
        loadL(min(3 * (T + 2) - 1, lLast), bufL[PAR][1], edge_c);
        luma_row(v, SOp1(), SE0(), SEm2(), SOm1(), std::true_type(), s_p1 ? a.lS[0] : a.lB[0], a.lA[1], a.lA[4], s_m1 ? a.lS[3] : a.lB[3]);
        // luma row 3T: odd 2T + 1 (B1), even 2T (A2), even 2T - 2 closes (A5), odd 2T - 1 (B4)
        hrowL(bufL[PAR][2], edge_c, v);
        loadL(min(3 * (T + 2), lLast), bufL[PAR][2], edge_c);
        luma_row(v, SOp1(), SE0(), SEm2(), SOm1(), std::false_type(), s_p1 ? a.lS[1] : a.lB[1], a.lA[2], a.lA[5], s_m1 ? a.lS[4] : a.lB[4]);
        if (PAR == 0) {
            // chroma row 3k -> output row 4k - 2 (phase 0)
            const int kap = T >> 1;
            push(bufC[1], edge_c);
            loadC(min(3 * (kap + 1), cLast), bufC[1], edge_c);
            emit(2 * T - 2, SEm2(), a.cP[0][0], a.cP[0][1], a.cP[0][2], a.cP[0][3]);
        } else {
            emit(2 * T - 2, SEm2(), s_m1 ? a.cS0[0] : a.cP[2][0], s_m1 ? a.cS0[1] : a.cP[2][1], s_m1 ? a.cS0[2] : a.cP[2][2], s_m1 ? a.cS0[3] : a.cP[2][3]);
        }
    };
    auto run = [&](auto edge_c) {
        const int k0 = T0 >> 1;
        // the chroma window before step T0: rows 3 k0 - 4 .. 3 k0 - 2
        {
            unsigned pre[5];
#pragma unroll
            for (int r = 0; r < 3; r++) { loadC(3 * k0 - 4 + r, pre, edge_c); push(pre, edge_c); }
        }
        loadC(3 * k0 - 1, bufC[0], edge_c); loadC(3 * k0, bufC[1], edge_c); loadC(3 * k0 + 1, bufC[2], edge_c);
#pragma unroll
        for (int r = 0; r < 3; r++) { loadL(3 * T0 - 2 + r, bufL[0][r], edge_c); loadL(3 * (T0 + 1) - 2 + r, bufL[1][r], edge_c); }
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
// One axis of an exact 3:2 down-scale: the table row of output x on its nominal window (x = 2k: [3k - 2, 3k + 3], x = 2k + 1:
// [3k - 1, 3k + 4]).  Every output must equal "the middle row of its parity on an edge-replicated line", except output 1, whose table
// row is taken as it is.  A / B / S as 3 int16 pairs each.
bool down32_axis(const FilterBank &fb, int srcLen, int32_t (&A)[3], int32_t (&B)[3], int32_t (&S)[3])
{
    if (fb.count < 8 || (fb.count & 1) || 2 * srcLen != 3 * fb.count) return false;
    auto window = [&](int x, int (&w)[6]) -> bool {              // the table row of x on its window; false: a tap falls outside it
        const int ws = 3 * (x >> 1) - ((x & 1) ? 1 : 2);
        for (int k = 0; k < 6; k++) w[k] = 0;
        for (int j = 0; j < fb.taps; j++) {
            const int16_t c = fb.coef[(size_t)x * fb.taps + j];
            if (!c) continue;
            const int s = fb.pos[x] + j;
            if (s < 0 || s >= srcLen || s - ws < 0 || s - ws > 5) return false;
            w[s - ws] += c;
        }
        return true;
    };
    int nom[2][6];
    const int xm = (fb.count / 2) & ~1;
    if (!window(xm, nom[0]) || !window(xm + 1, nom[1])) return false;
    for (int x = 0; x < fb.count; x++) {
        int w[6], e[6] = {0, 0, 0, 0, 0, 0};
        if (!window(x, w)) return false;
        const int ws = 3 * (x >> 1) - ((x & 1) ? 1 : 2);
        for (int k = 0; k < 6; k++) {                            // the nominal row folded onto the clamped samples, on the window again
            const int s = std::min(std::max(ws + k, 0), srcLen - 1);
            e[s - ws] += nom[x & 1][k];
        }
        if (x == 1) {
            // in the kernel the out-of-range slot holds the replicated edge sample: its coefficient is 0 in the table row
            for (int k = 0; k < 3; k++) S[k] = (int32_t)((uint32_t)(uint16_t)w[2 * k] | ((uint32_t)(uint16_t)w[2 * k + 1] << 16));
        } else if (std::memcmp(w, e, sizeof(w)) != 0) {
            return false;
        }
    }
    for (int par = 0; par < 2; par++) {
        int32_t (&N)[3] = par ? B : A;
        for (int k = 0; k < 3; k++) N[k] = (int32_t)((uint32_t)(uint16_t)nom[par][2 * k] | ((uint32_t)(uint16_t)nom[par][2 * k + 1] << 16));
    }
    return true;
}

int yuv3x2_prepare(const ScalePlan &p, const YuvScaleTiling &g, Yuv3x2Tables &t)
{
    t = Yuv3x2Tables();
    const char *off = GMAT_KNOB("GMAT_SCALE_NO_STRIP");
    if (off && atoi(off)) return 0;
    if (g.yuvOut != 1) return 0;
    const bool nv = p.srcFormat == GMAT_PIX_FMT_NV12 && p.dstFormat == GMAT_PIX_FMT_NV12;
    const bool pl = p.srcFormat == GMAT_PIX_FMT_YUV420P && p.dstFormat == GMAT_PIX_FMT_YUV420P;
    if (!nv && !pl) return 0;
    // a lane makes 8 samples of a plane (4 positions of the UV plane): luma widths in multiples of 16 for planar chroma, of 8 for NV12;
    // rows come in pairs on both planes: heights in multiples of 4
    if (2 * p.srcW != 3 * p.dstW || 2 * p.srcH != 3 * p.dstH || p.dstW % (nv ? 8 : 16) || p.dstW < 64 || p.dstH % 4 || p.dstH < 16) return 0;
    if (p.chrDstW * 2 != p.dstW || p.chrDstH * 2 != p.dstH || 2 * p.chrSrcW != 3 * p.chrDstW || 2 * p.chrSrcH != 3 * p.chrDstH) return 0;
    if (!down32_axis(p.hLum, p.srcW, t.hLA, t.hLB, t.hLS)) return 0;
    if (!down32_axis(p.hChr, p.chrSrcW, t.hCA, t.hCB, t.hCS)) return 0;
    if (!down32_axis(g.vLumEff, p.srcH, t.vLA, t.vLB, t.vLS)) return 0;
    if (!down32_axis(g.vChrEff, p.chrSrcH, t.vCA, t.vCB, t.vCS)) return 0;
    for (int y = 0; y < p.dstH; y++) if (g.lumRound[y] != g.lumRound[0]) return 0;
    for (int y = 0; y < p.chrDstH; y++) if (g.chrRound[y] != g.chrRound[0]) return 0;
    t.lr = g.lumRound[0]; t.cr = g.chrRound[0];
    t.ok = 1;
    return 0;
}

int launch_scale_yuv3x2(const Yuv3x2Args &a0, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (!frames || nframes < 1 || nframes > kYuv2xMaxFrames) return GMAT_ERR(EINVAL);
    Yuv3x2Args a = a0;
    const char *segStr = GMAT_KNOB("GMAT_STRIP_ROWS");              // tuning / test override (output rows per segment), read per launch
    const int segEnv = segStr ? atoi(segStr) : 0;
    const int nstripsL = (a.dstW + E3_STRIP - 1) / E3_STRIP;
    const int nstripsC = a.nv12 ? (a.chrDstW + E3_STRIP_UV - 1) / E3_STRIP_UV : (a.chrDstW + E3_STRIP - 1) / E3_STRIP;
    const int nplC = a.nv12 ? 1 : 2;
    a.nsgL = nstripsL; a.nsgC = nstripsC;                        // strips per row of segments
    int seg = segEnv > 0 ? segEnv : 0;
    if (!seg) {
        // a wave walks 3 (seg / 2 + 2) source rows for seg output rows: two warm-up steps per segment.  Measured on 1080p -> 720p
        // (profiles/r02n_down32_rows_sweep.txt): one frame per launch 4 rows (5.2 us; 8 rows 6.6), 4 frames 8 rows, 32 frames 12-18 rows
        const long rows = ((long)a.dstH * nstripsL + (long)a.chrDstH * nstripsC * nplC) * nframes;      // wave-rows (output)
        seg = (int)std::min(24L, std::max(rows < 8192 ? 4L : 8L, (rows + 6143) / 6144));
    }
    seg = (seg + 1) & ~1;                                        // segments start on even output rows
    a.segRowsL = seg; a.segRowsC = seg;
    a.nsegL = (a.dstH + a.segRowsL - 1) / a.segRowsL;
    a.nsegC = (a.chrDstH + a.segRowsC - 1) / a.segRowsC;
    a.nblkL = (a.nsegL * a.nsgL + 3) / 4;                        // four (segment, strip) units per workgroup
    a.nblk = a.nblkL + (a.nsegC * a.nsgC * nplC + 3) / 4;
    const dim3 grid(a.xcdRemap ? 8 * ((a.nblk + 7) / 8) : a.nblk, nframes), block(256);
    if (a.nv12) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv3x2_kernel<true>), grid, block, 0, stream, a, *frames);
    else        hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv3x2_kernel<false>), grid, block, 0, stream, a, *frames);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// Is this axis the 3:4 UP-scale the kernel assumes: output y = 4k + 2 + i on the window [ws, ws + 3], ws = 3k + (0, 1, 1, 2)[i], every
// table row from row 2 on equal to its phase's middle row folded onto the clamped samples?  Rows 0 and 1 keep the table's own taps
// (on their windows [-2, 1] and [-1, 2], whose out-of-range slots the kernel fills with row 0: a tap there must be 0).
static bool up43_axis(const FilterBank &fb, int srcLen, int32_t (&P)[4][4], int32_t (&S0)[4], int32_t (&S1)[4])
{
    static const int off[4] = {0, 1, 1, 2};
    if (fb.count < 16 || (fb.count & 3) || 3 * fb.count != 4 * srcLen) return false;
    auto ws_of = [&](int y) { const int k = (y - 2) >> 2, i = (y - 2) & 3; return 3 * k + off[i]; };     // arithmetic shift: k = -1 for y = 0, 1
    auto window = [&](int y, int (&w)[4]) -> bool {
        const int ws = ws_of(y);
        for (int k = 0; k < 4; k++) w[k] = 0;
        for (int j = 0; j < fb.taps; j++) {
            const int16_t c = fb.coef[(size_t)y * fb.taps + j];
            if (!c) continue;
            const int s = fb.pos[y] + j;
            if (s < 0 || s >= srcLen || s - ws < 0 || s - ws > 3) return false;
            w[s - ws] += c;
        }
        return true;
    };
    int nom[4][4];
    const int ym = ((fb.count / 2) & ~3) + 2;
    for (int i = 0; i < 4; i++) if (!window(ym + i, nom[i])) return false;
    for (int y = 0; y < fb.count; y++) {
        int w[4];
        if (!window(y, w)) return false;
        if (y < 2) {
            int32_t (&S)[4] = y ? S1 : S0;
            for (int k = 0; k < 4; k++) S[k] = w[k];
            continue;
        }
        const int ws = ws_of(y), i = (y - 2) & 3;
        int e[4] = {0, 0, 0, 0};
        for (int k = 0; k < 4; k++) {
            const int s = std::min(std::max(ws + k, 0), srcLen - 1);
            if (s - ws < 0 || s - ws > 3) return false;
            e[s - ws] += nom[i][k];
        }
        if (std::memcmp(w, e, sizeof(w)) != 0) return false;
    }
    for (int i = 0; i < 4; i++) for (int k = 0; k < 4; k++) P[i][k] = nom[i][k];
    return true;
}

int yuv32r_prepare(const ScalePlan &p, const YuvScaleTiling &g, Yuv32rTables &t)
{
    t = Yuv32rTables();
    const char *off = GMAT_KNOB("GMAT_SCALE_NO_STRIP");
    if (off && atoi(off)) return 0;
    if (g.fullChroma || g.yuvOut) return 0;
    if (p.srcFormat != GMAT_PIX_FMT_NV12) return 0;
    if (!(p.dstFormat == GMAT_PIX_FMT_RGB24 || p.dstFormat == GMAT_PIX_FMT_BGR24 || p.dstFormat == GMAT_PIX_FMT_RGBA ||
          p.dstFormat == GMAT_PIX_FMT_BGRA)) return 0;
    if (2 * p.srcW != 3 * p.dstW || 2 * p.srcH != 3 * p.dstH || p.dstW % 8 || p.dstW < 64 || p.dstH % 4 || p.dstH < 16) return 0;
    if (p.chrSrcW * 2 != p.srcW || p.chrSrcH * 2 != p.srcH || p.chrDstW * 2 != p.dstW || p.chrDstH != p.dstH) return 0;
    if (!down32_axis(p.hLum, p.srcW, t.hLA, t.hLB, t.hLS)) return 0;
    if (!down32_axis(p.hChr, p.chrSrcW, t.hCA, t.hCB, t.hCS)) return 0;
    if (!down32_axis(g.vLumEff, p.srcH, t.vLA, t.vLB, t.vLS)) return 0;
    if (!up43_axis(g.vChrEff, p.chrSrcH, t.cP, t.cS0, t.cS1)) return 0;
    for (int y = 0; y < p.dstH; y++) if (g.lumRound[y] != g.lumRound[0]) return 0;
    for (int y = 0; y < p.chrDstH; y++) if (g.chrRound[y] != g.chrRound[0]) return 0;
    t.lr = g.lumRound[0]; t.cr = g.chrRound[0];
    t.ok = 1;
    return 0;
}

int launch_scale_yuv32r(const Yuv32rArgs &a0, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (!frames || nframes < 1 || nframes > kYuv2xMaxFrames) return GMAT_ERR(EINVAL);
    Yuv32rArgs a = a0;
    a.nstrips = (a.dstW + E3_STRIP - 1) / E3_STRIP;
    const char *segStr = GMAT_KNOB("GMAT_STRIP_ROWS");              // tuning / test override: output rows per segment
    int seg = segStr ? atoi(segStr) : 0;
    if (seg <= 0) {
        // measured on 1080p -> 720p (profiles/r02zc_down32rgb.txt): 32 frames per launch 24 rows 2.14 us per frame (12: 2.32, 48: 2.74);
        // one frame 4 rows 9.9 us (8: 13.8) — 1280 columns are three strips, a single frame is a few hundred waves
        const long rows = (long)a.dstH * a.nstrips * nframes;
        seg = (int)std::min(48L, std::max(4L, (rows + 3071) / 3072));
    }
    seg = (seg + 3) & ~3;                                        // segments start on multiples of 4 output rows (the chroma phases' period)
    a.segRows = seg;
    a.nseg = (a.dstH + seg - 1) / seg;
    a.nblk = (a.nseg * a.nstrips + 3) / 4;
    a.xcdRemap = 1;
    const dim3 grid(8 * ((a.nblk + 7) / 8), nframes), block(256);
    switch (a.dstFormat) {
    case GMAT_PIX_FMT_RGB24: hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv32r_kernel<0>), grid, block, 0, stream, a, *frames); break;
    case GMAT_PIX_FMT_BGR24: hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv32r_kernel<1>), grid, block, 0, stream, a, *frames); break;
    case GMAT_PIX_FMT_RGBA:  hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv32r_kernel<2>), grid, block, 0, stream, a, *frames); break;
    case GMAT_PIX_FMT_BGRA:  hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv32r_kernel<3>), grid, block, 0, stream, a, *frames); break;
    default: return GMAT_ERR(EINVAL);
    }
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
