// k_scale_yuv2s.hip — strip-walking form of the exact 2:1 YUV 4:2:0 -> packed RGB scaler for gfx950 (the headline:
// 4K nv12 -> 1080p rgb24 bicubic, one libswscale context: swscale.c:234-520 semantics, bit-exact).
//
// What the tiled 2:1 kernel (k_scale_yuv2x.hip) spends its time on is not arithmetic but movement: every 64 x 16
// tile re-loads and re-filters a 7-row vertical halo (19 % of its horizontal work, 1.32x the compulsory HBM traffic),
// widens every sample into LDS, reads it back, writes the horizontal result to LDS and reads it a third time.  Here
//   * a wave owns a strip of 256 output columns and WALKS DOWN it: the horizontally filtered rows it still needs
//     live in a 4-deep register window (16 VGPRs), so a source row is loaded and filtered once per strip segment;
//   * pixels never pass through LDS: each lane loads the 16 source bytes its 4 outputs need straight from global
//     memory (4-byte aligned dwordx4; neighbouring lanes overlap by 8 bytes, which the vector L1 absorbs) and
//     widens them with v_perm_b32 onto the odd-aligned pair grid (2x-3, 2x-2) ... (2x+3, 2x+4): 4 coefficient pairs
//     per output instead of the 5 an even-aligned window needs;
//   * no tables: libswscale folds the taps that fall outside the frame onto the edge sample (initFilter,
//     utils.c:601-640), which — when the host has checked it coefficient by coefficient, yuv2s_prepare — is the
//     interior filter applied to an edge-replicated frame.  Rows are replicated by clamping the row address, columns
//     by a byte permute in the two edge lanes, and every coefficient is a kernel argument (an SGPR operand of
//     v_dot2c_i32_i16);
//   * the per-chroma-sample colour terms (yuv2rgb.c's table_rV / gU / gV / bU in closed form) come from two 2 KB
//     LDS look-up tables built once per workgroup, not from 14 VALU instructions per sample.
// Waves of a workgroup share nothing but those tables: no barrier inside the row loop, occupancy set by VGPRs only.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

namespace gmat {

constexpr int S2_STRIP = 256;                  // output columns per wave: 64 lanes x 4

// Round 3: three cuts in the headline kernel's instruction stream, each behind a build-time switch so that the A/B of
// profiles/r03c_* can be repeated (tools/build_variant.sh): 155 -> ~140 VALU instructions per row and lane.
//   S2_SBASE   the row base of every load / store is a scalar pointer (SALU: s_mul, s_add_u32, s_addc_u32), the lane offset a
//              loop-invariant VGPR: no v_add_u32 per load (5 per iteration)
//   S2_LUT512  the colour tables are indexed by (sum + 128 * 16384) >> 14 on 512 entries whose first 128 / last 128 repeat
//              entries 0 / 255: the index clamp of yuv2rgb.c's tables (av_clip_uint8 in the chroma path) is in the table, not
//              a v_med3 per chroma sample (4 per iteration).  The host proves the index range from the coefficients.
//   S2_SATPK   channel = sat_u8(high half of (term + Y * cy)) packed two at a time by v_sat_pk_u8_i16 instead of a v_med3 per
//              channel and a byte permute (21 -> 15 per rgb24 row, 28 -> 20 per rgba row)
#ifndef S2_SBASE
#define S2_SBASE 1
#endif
#ifndef S2_LUT512
#define S2_LUT512 1
#endif
#ifndef S2_SATPK
#define S2_SATPK 1
#endif
constexpr int S2_LUT_N = S2_LUT512 ? 512 : 256, S2_LUT_BIAS = S2_LUT512 ? 128 : 0;
// S2_PROBE (measurement builds only, tools/build_variant.sh — the results are WRONG pixels): what the walker's time is made of.  Bit 0: no horizontal
// chroma filter (-26 VALU a row), bit 1: no vertical luma filter (-16), bit 2: the pair's second luma row is not filtered (-23), bit 3: the second luma row
// is not LOADED either (a third of the read bytes gone, with bit 2).  profiles/r05b_headline_probes.txt
#ifndef S2_PROBE
#define S2_PROBE 0
#endif

// ---- unaligned vector loads (4-byte aligned addresses) -----------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned s2_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned s2_u32x3 __attribute__((ext_vector_type(3), aligned(4)));
typedef unsigned s2_u32x2 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ uint4 s2_ld16(const uint8_t *p) { const s2_u32x4 v = *reinterpret_cast<const s2_u32x4 *>(p); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint3 s2_ld12(const uint8_t *p) { const s2_u32x3 v = *reinterpret_cast<const s2_u32x3 *>(p); return make_uint3(v.x, v.y, v.z); }
__device__ __forceinline__ uint2 s2_ld8(const uint8_t *p) { const s2_u32x2 v = *reinterpret_cast<const s2_u32x2 *>(p); return make_uint2(v.x, v.y); }
#else
static inline uint4 s2_ld16(const uint8_t *p) { uint4 v; std::memcpy(&v, p, 16); return v; }
static inline uint3 s2_ld12(const uint8_t *p) { uint3 v; std::memcpy(&v, p, 12); return v; }
static inline uint2 s2_ld8(const uint8_t *p) { uint2 v; std::memcpy(&v, p, 8); return v; }
#endif

// v_dot2_i32_i16 in its three-operand (VOP3P) form: with the clamp bit set the compiler cannot use the two-operand
// v_dot2c, whose tied accumulator costs a v_mov per chain start (12 per row here).  The clamp saturates the int32
// accumulate, which none of these sums (< 2^26) can reach, so the result is the same.
#ifndef S2_DOT2_CLAMP
#define S2_DOT2_CLAMP 1
#endif
__device__ __forceinline__ int s2_dot2(int packed_ab, int packed_cd, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, packed_ab), __builtin_bit_cast(short2v, packed_cd), acc, S2_DOT2_CLAMP != 0);
}

// two signed 16-bit halves -> two unsigned bytes with saturation, in the low half of the result (the upper half is never used:
// every caller selects bytes 0 and 1 with a v_perm_b32)
__device__ __forceinline__ unsigned s2_sat_pk_u8_i16(unsigned v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(v));
    return r;
#else
    const int lo = (int16_t)(v & 0xFFFFu), hi = (int16_t)(v >> 16);
    return (unsigned)std::min(std::max(lo, 0), 255) | ((unsigned)std::min(std::max(hi, 0), 255) << 8);
#endif
}

// S2_SBASE: a plane as a raw buffer resource (four SGPRs built once per wave from the wave-uniform plane pointer): the row offset
// travels in the instruction's scalar offset, the lane offset is a loop-invariant VGPR — no vector ALU per load or store.
// num_records = 2^32 - 1: the kernel clamps every row and column itself, the bounds check is not relied on.
// S2_LD_AUX / S2_ST_AUX: the cache-policy bits of the loads / stores (1 = sc0, 2 = nt, 16 = sc1).  The stores are non-temporal: the
// destination is written once, `nt` keeps it from displacing the source rows neighbouring bands share in the L2 — alternating runs on
// one box 116.8 -> 113.1 us per 32-frame launch, 0.640 -> 0.660 (sc0 / sc1 on the stores +-0; on the LOADS nt -17 %, sc1 -3 %, sc0 +-0:
// profiles/r03zs_cache_policy_ab.txt)
#ifndef S2_LD_AUX
#define S2_LD_AUX 0
#endif
#ifndef S2_ST_AUX
#define S2_ST_AUX (GMAT_NT_STORES ? 2 : 0)
#endif
struct S2Plane {
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ explicit S2Plane(const uint8_t *p) : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p), 0, 0xFFFFFFFFu, 0x00020000)) {}
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    typedef unsigned v3u __attribute__((ext_vector_type(3)));
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    __device__ __forceinline__ uint4 ld16(unsigned lane, unsigned row) const { const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, lane, row, S2_LD_AUX); return make_uint4(v.x, v.y, v.z, v.w); }
    __device__ __forceinline__ uint3 ld12(unsigned lane, unsigned row) const { const v3u v = __builtin_amdgcn_raw_buffer_load_b96(r, lane, row, S2_LD_AUX); return make_uint3(v.x, v.y, v.z); }
    __device__ __forceinline__ uint2 ld8(unsigned lane, unsigned row) const { const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, lane, row, S2_LD_AUX); return make_uint2(v.x, v.y); }
    __device__ __forceinline__ void st16(uint4 d, unsigned lane, unsigned row) const { v4u v = {d.x, d.y, d.z, d.w}; __builtin_amdgcn_raw_buffer_store_b128(v, r, lane, row, S2_ST_AUX); }
    __device__ __forceinline__ void st12(uint3 d, unsigned lane, unsigned row) const { v3u v = {d.x, d.y, d.z}; __builtin_amdgcn_raw_buffer_store_b96(v, r, lane, row, S2_ST_AUX); }
#else
    // hipcc's host pass (never executed) and the CPU emulation of the test suite
    uint8_t *p;
    __host__ __device__ explicit S2Plane(const uint8_t *q) : p(const_cast<uint8_t *>(q)) {}
    __host__ __device__ uint4 ld16(unsigned lane, unsigned row) const { uint4 v; std::memcpy(&v, p + (size_t)row + lane, 16); return v; }
    __host__ __device__ uint3 ld12(unsigned lane, unsigned row) const { uint3 v; std::memcpy(&v, p + (size_t)row + lane, 12); return v; }
    __host__ __device__ uint2 ld8(unsigned lane, unsigned row) const { uint2 v; std::memcpy(&v, p + (size_t)row + lane, 8); return v; }
    __host__ __device__ void st16(uint4 d, unsigned lane, unsigned row) const { std::memcpy(p + (size_t)row + lane, &d, 16); }
    __host__ __device__ void st12(uint3 d, unsigned lane, unsigned row) const { std::memcpy(p + (size_t)row + lane, &d, 12); }
#endif
};

// bytes (1,2) of lo -> one int16 pair; byte 3 of lo and byte 0 of hi -> the next one (selector 0x0C = constant zero)
__device__ __forceinline__ int s2_pair12(unsigned lo) { return (int)__builtin_amdgcn_perm(0u, lo, 0x0C020C01u); }
__device__ __forceinline__ int s2_pair30(unsigned hi, unsigned lo) { return (int)__builtin_amdgcn_perm(hi, lo, 0x0C040C03u); }
__device__ __forceinline__ unsigned s2_rep(unsigned v, unsigned sel) { return __builtin_amdgcn_perm(v, v, sel); }

// The pixels one row-loop iteration consumes: a luma row pair (2m-1, 2m) and one chroma row
struct S2Pix {
    uint4 la, lb;                  // 16 luma bytes of each row, from column 8t - 4 of the strip
    uint4 ca;                      // NV12: 8 UV pairs from chroma column 4t - 4;  planar: U bytes (x,y,z) from 4t - 4
    uint4 cb;                      // NV12: (x,y) the next 4 UV pairs;             planar: V bytes (x,y,z)
};

// DST: 0 rgb24, 1 bgr24, 2 rgba, 3 bgra
// S2_WAVES = n > 0: compile for n waves a SIMD (A/B).  74 VGPRs as allocated = 6 waves; forcing 7 (72 VGPRs, 6 dwords spilled) measured
// 3.87 against 3.61-3.77 us per frame, 8 (64 VGPRs) 5.3 us: occupancy is not what this kernel lacks.
#ifndef S2_WAVES
#define S2_WAVES 0
#endif
#if S2_WAVES > 0 && defined(__HIP__)
#define S2_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(S2_WAVES)))
#else
#define S2_WAVES_ATTR
#endif
template <bool NV12, int DST>
__global__ __launch_bounds__(256) S2_WAVES_ATTR void scale_yuv2s_kernel(Yuv2sArgs a, Yuv2xFrames fr)
{
    constexpr bool BGR = (DST & 1) != 0;
    constexpr int BPP = DST >= 2 ? 4 : 3;
    __shared__ int2 lutV[S2_LUT_N], lutU[S2_LUT_N];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- colour look-up tables: chan = byte 2 of clamp(term + Y * cy, 0, 0xFFFFFF) with
    //      term_R = lutV[V].x, term_G = lutV[V].y + lutU[U].x, term_B = lutU[U].y  (px_math.h chroma_terms, split by sample)
    //      S2_LUT512: entry e holds sample clip_u8(e - 128)
    {
        const Yuv2RgbConsts &k = a.y2r;
#pragma unroll
        for (int e = tid; e < S2_LUT_N; e += 256) {
            const int i = min(max(e - S2_LUT_BIAS, 0), 255);
            lutV[e] = make_int2(k.base + m24(k.offR + (m24(i, k.crv) >> 16), k.cy), m24(m24(i, k.cgv) >> 16, k.cy));
            lutU[e] = make_int2(k.base + m24(k.offG + (m24(i, k.cgu) >> 16), k.cy), k.base + m24(k.offB + (m24(i, k.cbu) >> 16), k.cy));
        }
    }
    __syncthreads();

    // ---- which strip segment ------------------------------------------------------------------------------------
    const int nblk = a.nseg * a.nsg;
    int lin = blockIdx.x;
    if (a.xcdRemap) {
        const int chunk = (nblk + 7) >> 3;
        lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    }
    if (lin >= nblk) return;
    const int seg = __builtin_amdgcn_readfirstlane(lin / a.nsg);
    const int X0 = ((lin - seg * a.nsg) * 4 + wave) * S2_STRIP;
    if (X0 >= a.dstW) return;                                  // wave-uniform; no barrier below
    const int y0 = seg * a.segRows;
    const int nOut = min(a.segRows, a.dstH - y0);
    const int nIter = nOut + 3;                                // 3 warm-up row pairs fill the vertical window
    // Odd segments walk UPWARD (a.updown): the rows two neighbouring segments both read — the 3 row pairs either side of their
    // joint — are then read at the same time (both at the start of their walks or both at the end) instead of a wave's lifetime
    // apart, so the second reader finds them in L2.  Iteration j handles row pair mStart + dir * j and, from j = 3 on, output row
    // yoStart + dir * (j - 3); the window's slots then hold the pairs of an output row youngest-first instead of oldest-first,
    // i.e. the vertical coefficient pairs are applied in reverse order (the rows inside a pair keep their order).  All scalar.
    const bool up = a.updown && (seg & 1);
    const int dir = up ? -1 : 1;
    const int mStart = up ? y0 + nOut + 1 : y0 - 1;
    const int yoStart = up ? y0 + nOut - 1 : y0;
    const int vl0 = up ? a.vL[3] : a.vL[0], vl1 = up ? a.vL[2] : a.vL[1], vl2 = up ? a.vL[1] : a.vL[2], vl3 = up ? a.vL[0] : a.vL[3];

    const uint8_t *py, *pu, *pv;
    uint8_t *pd;
    {
        const int f = blockIdx.y;
        py = fr.y[f]; pu = fr.u[f]; pv = fr.v[f]; pd = fr.dst[f];
    }

    // ---- per-lane constants -------------------------------------------------------------------------------------
    const int xo = X0 + 4 * lane;
    const bool active = xo < a.dstW;
    const int xc = active ? xo : a.dstW - 4;                    // idle lanes shadow the last group (loads stay inside the rows)
    // luma: bytes [2xc - 4, 2xc + 12) of the row; sources < 0 and >= srcW are the replicated edge samples
    const int wantL = 2 * xc - 4;
    const int offL = min(max(wantL, 0), a.srcW - 16);
    const int shL = wantL - offL;                               // -4 at the left frame edge, +4 at the right one
    // chroma samples [xc - 4, xc + 8) of the row
    int offA, offB, shA, shB;
    if (NV12) {
        offA = max(2 * xc - 8, 0);               shA = 2 * xc - 8 - offA;        // 16 bytes: samples xc-4 .. xc+3   (-8: left edge)
        offB = min(2 * xc + 8, 2 * a.chrSrcW - 8); shB = 2 * xc + 8 - offB;      //  8 bytes: samples xc+4 .. xc+7   (+8: right edge)
    } else {
        offA = min(max(xc - 4, 0), a.chrSrcW - 12); shA = xc - 4 - offA;         // 12 bytes of each plane (-4 / +4)
        offB = 0; shB = 0;
    }
    const unsigned dstOff = (unsigned)xo * BPP;
    // which lanes sit on a frame edge, as lane masks: the edge variant of the row loop repairs their registers with SELECTS on these
    // masks — written as `if (shL < 0) ...` the fix-ups became divergent branches (s_and_saveexec + s_cbranch around each), and the
    // waves of the two edge strips ran 22-26 % longer per row than the others although they execute only 11 % more VALU instructions
    // (per-wave time stamps, profiles/r02q_headline_wave_durations.txt); with one wave per slot the launch lasts as long as its slowest wave
    const bool edgeLL = shL < 0, edgeLR = shL > 0, edgeAL = shA < 0, edgeAR = shA > 0, edgeBR = shB > 0;

    // row pointers are wave-uniform (scalar unit), the lane offsets unsigned 32-bit: global_load with an SGPR base
    const unsigned uoffL = (unsigned)offL, uoffA = (unsigned)offA, uoffB = (unsigned)offB;
#if S2_SBASE
    const S2Plane bY(py), bU(pu), bV(NV12 ? pu : pv), bD(pd);
#endif
    auto load_luma = [&](int m, S2Pix &P) {
        const int ra = min(max(2 * m - 1, 0), a.srcH - 1), rb = min(max(2 * m, 0), a.srcH - 1);
        // one 32-bit offset per load = scalar row offset + lane offset (a plane is far below 4 GB): the frame pointer
        // stays the SGPR base of the global_load
#if S2_SBASE
        P.la = bY.ld16(uoffL, (unsigned)ra * (unsigned)a.ys);       // lane offset (loop-invariant VGPR) + scalar row offset
        if ((S2_PROBE & 8) == 0) P.lb = bY.ld16(uoffL, (unsigned)rb * (unsigned)a.ys); else P.lb = P.la;
#else
        P.la = s2_ld16(py + (unsigned)((unsigned)ra * (unsigned)a.ys + uoffL));
        P.lb = s2_ld16(py + (unsigned)((unsigned)rb * (unsigned)a.ys + uoffL));
#endif
    };
    auto load_chroma = [&](int cy, S2Pix &P) {
        const int r = min(max(cy, 0), a.chrSrcH - 1);
        if (NV12) {
            const unsigned ro = (unsigned)r * (unsigned)a.us;
#if S2_SBASE
            P.ca = bU.ld16(uoffA, ro);
            const uint2 t = bU.ld8(uoffB, ro);
#else
            P.ca = s2_ld16(pu + (unsigned)(ro + uoffA));
            const uint2 t = s2_ld8(pu + (unsigned)(ro + uoffB));
#endif
            P.cb = make_uint4(t.x, t.y, 0u, 0u);
        } else {
#if S2_SBASE
            const uint3 tu = bU.ld12(uoffA, (unsigned)r * (unsigned)a.us);
            const uint3 tv = bV.ld12(uoffA, (unsigned)r * (unsigned)a.vs);
#else
            const uint3 tu = s2_ld12(pu + (unsigned)((unsigned)r * (unsigned)a.us + uoffA));
            const uint3 tv = s2_ld12(pv + (unsigned)((unsigned)r * (unsigned)a.vs + uoffA));
#endif
            P.ca = make_uint4(tu.x, tu.y, tu.z, 0u);
            P.cb = make_uint4(tv.x, tv.y, tv.z, 0u);
        }
    };
    // frame-edge lanes: shift the dwords into window position and replicate the edge sample
    auto fix_luma = [&](uint4 L, auto edge_c) -> uint4 {
        constexpr int K = decltype(edge_c)::value;              // 0 interior wave, 1 the frame's left edge, 2 its right edge, 3 both (one strip)
        if constexpr (K == 1) {
            const unsigned first = s2_rep(L.x, 0x00000000u);
            L = make_uint4(edgeLL ? first : L.x, edgeLL ? L.x : L.y, edgeLL ? L.y : L.z, edgeLL ? L.z : L.w);
        } else if constexpr (K == 2) {
            const unsigned last = s2_rep(L.w, 0x03030303u);
            L = make_uint4(edgeLR ? L.y : L.x, edgeLR ? L.z : L.y, edgeLR ? L.w : L.z, edgeLR ? last : L.w);
        } else if constexpr (K == 3) {
            const unsigned first = s2_rep(L.x, 0x00000000u), last = s2_rep(L.w, 0x03030303u);
            L = make_uint4(edgeLL ? first : edgeLR ? L.y : L.x, edgeLL ? L.x : edgeLR ? L.z : L.y,
                           edgeLL ? L.y : edgeLR ? L.w : L.z, edgeLL ? L.z : edgeLR ? last : L.w);
        }
        return L;
    };

    // horizontal luma filter of one row: 4 outputs from 7 odd-aligned pairs
    auto hrow = [&](const uint4 &L, int (&s)[4]) {
        int p[7];
        p[0] = s2_pair12(L.x); p[1] = s2_pair30(L.y, L.x); p[2] = s2_pair12(L.y); p[3] = s2_pair30(L.z, L.y);
        p[4] = s2_pair12(L.z); p[5] = s2_pair30(L.w, L.z); p[6] = s2_pair12(L.w);
#pragma unroll
        for (int j = 0; j < 4; j++)
            s[j] = s2_dot2(p[j + 3], a.hL[3], s2_dot2(p[j + 2], a.hL[2], s2_dot2(p[j + 1], a.hL[1], s2_dot2(p[j], a.hL[0], 0))));
    };

    int hw[4][4];                                               // [slot][output]: (row 2m-1 | row 2m << 16) after hScale8To15_c
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) hw[s][j] = 0;

    S2Pix buf[2];                                               // ping-pong: iteration j consumes buf[j & 1], prefetches into the other
    buf[0].ca = buf[0].cb = buf[1].ca = buf[1].cb = make_uint4(0u, 0u, 0u, 0u);
    buf[1].la = buf[1].lb = make_uint4(0u, 0u, 0u, 0u);
    load_luma(mStart, buf[0]);

    // EDGE: the wave holds a frame-edge lane (first / last strip).  The whole row loop exists twice so that interior
    // waves carry none of the fix-up moves.
    auto body = [&](const int j, auto slot_c, auto edge_c) {
        constexpr int SLOT = decltype(slot_c)::value;           // j & 3, static after unrolling
        constexpr int EDGE = decltype(edge_c)::value;           // 0 interior, 1 left, 2 right, 3 both: see fix_luma
        const S2Pix &cur = buf[SLOT & 1];
        S2Pix &nxt = buf[(SLOT + 1) & 1];
        // ---- prefetch the next iteration's rows ------------------------------------------------------------
        if (j + 1 < nIter) {
            load_luma(mStart + dir * (j + 1), nxt);                 // the row pair of iteration j + 1
            if (j + 1 >= 3) load_chroma(yoStart + dir * (j - 2), nxt);   // chroma row of ITS output row
        }
        // ---- horizontal luma of pair m = y0 - 1 + j -> slot ------------------------------------------------
        {
            int sa[4], sb[4];
            hrow(fix_luma(cur.la, edge_c), sa);
            if ((S2_PROBE & 4) == 0) hrow(fix_luma(cur.lb, edge_c), sb);
            else { for (int q = 0; q < 4; q++) sb[q] = sa[q] ^ (int)cur.lb.x; }
#pragma unroll
            for (int q = 0; q < 4; q++)       // hScale8To15_c: min(val >> 7, 32767); the lower bound cannot trigger
                hw[SLOT][q] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(sa[q] >> 7, sb[q] >> 7));
        }
        if (j >= 3) {
            const int yo = yoStart + dir * (j - 3);
            // ---- vertical luma: pairs yo-1 .. yo+2 (walking up: yo+2 .. yo-1) sit in slots SLOT+1 .. SLOT+4 (mod 4) ------
            int Y[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int acc = a.lr;
                if ((S2_PROBE & 2) == 0) {
                acc = s2_dot2(hw[(SLOT + 1) & 3][q], vl0, acc);
                acc = s2_dot2(hw[(SLOT + 2) & 3][q], vl1, acc);
                acc = s2_dot2(hw[(SLOT + 3) & 3][q], vl2, acc);
                }
                acc = s2_dot2(hw[(SLOT + 4) & 3][q], vl3, acc);
                Y[q] = acc >> 19;
            }
            // ---- chroma of row yo: 2 outputs per plane from 5 odd-aligned pairs ---------------------------------
            int pU[5], pV[5];
            if (NV12) {
                unsigned e[6] = {cur.ca.x, cur.ca.y, cur.ca.z, cur.ca.w, cur.cb.x, cur.cb.y};
                if constexpr ((EDGE & 1) != 0) {
                    const unsigned r = s2_rep(e[0], 0x01000100u), e0 = e[0], e1 = e[1];
                    e[0] = edgeAL ? r : e0; e[1] = edgeAL ? r : e1; e[2] = edgeAL ? e0 : e[2]; e[3] = edgeAL ? e1 : e[3];
                }
                if constexpr ((EDGE & 2) != 0) {
                    const unsigned rr = s2_rep(e[5], 0x03020302u);
                    e[4] = edgeBR ? rr : e[4]; e[5] = edgeBR ? rr : e[5];
                }
#pragma unroll
                for (int k = 0; k < 5; k++) {       // samples (2k-3, 2k-2) rel. to xc: bytes 2,3 of e[k] and 0,1 of e[k+1]
                    pU[k] = (int)__builtin_amdgcn_perm(e[k + 1], e[k], 0x0C040C02u);
                    pV[k] = (int)__builtin_amdgcn_perm(e[k + 1], e[k], 0x0C050C03u);
                }
            } else {
                unsigned fu[3] = {cur.ca.x, cur.ca.y, cur.ca.z}, fv[3] = {cur.cb.x, cur.cb.y, cur.cb.z};
                if constexpr (EDGE != 0) {
                    const unsigned u0 = fu[0], u1 = fu[1], u2 = fu[2], v0 = fv[0], v1 = fv[1], v2 = fv[2];
                    if constexpr (EDGE == 1) {
                        fu[0] = edgeAL ? s2_rep(u0, 0x00000000u) : u0; fu[1] = edgeAL ? u0 : u1; fu[2] = edgeAL ? u1 : u2;
                        fv[0] = edgeAL ? s2_rep(v0, 0x00000000u) : v0; fv[1] = edgeAL ? v0 : v1; fv[2] = edgeAL ? v1 : v2;
                    } else if constexpr (EDGE == 2) {
                        fu[0] = edgeAR ? u1 : u0; fu[1] = edgeAR ? u2 : u1; fu[2] = edgeAR ? s2_rep(u2, 0x03030303u) : u2;
                        fv[0] = edgeAR ? v1 : v0; fv[1] = edgeAR ? v2 : v1; fv[2] = edgeAR ? s2_rep(v2, 0x03030303u) : v2;
                    } else {
                        const unsigned uf = s2_rep(u0, 0x00000000u), ul = s2_rep(u2, 0x03030303u), vf = s2_rep(v0, 0x00000000u), vl = s2_rep(v2, 0x03030303u);
                        fu[0] = edgeAL ? uf : edgeAR ? u1 : u0; fu[1] = edgeAL ? u0 : edgeAR ? u2 : u1; fu[2] = edgeAL ? u1 : edgeAR ? ul : u2;
                        fv[0] = edgeAL ? vf : edgeAR ? v1 : v0; fv[1] = edgeAL ? v0 : edgeAR ? v2 : v1; fv[2] = edgeAL ? v1 : edgeAR ? vl : v2;
                    }
                }
                pU[0] = s2_pair12(fu[0]); pU[1] = s2_pair30(fu[1], fu[0]); pU[2] = s2_pair12(fu[1]); pU[3] = s2_pair30(fu[2], fu[1]); pU[4] = s2_pair12(fu[2]);
                pV[0] = s2_pair12(fv[0]); pV[1] = s2_pair30(fv[1], fv[0]); pV[2] = s2_pair12(fv[1]); pV[3] = s2_pair30(fv[2], fv[1]); pV[4] = s2_pair12(fv[2]);
            }
            int iU[2], iV[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                // hScale8To15_c (>> 7, min 32767), the one-tap vertical filter (1 << 18) + h * 4096, >> 19 and the table
                // index clamp collapse into clip_u8((sum + 8192) >> 14): floor(floor(x / 128 + 64) / 128) = floor((x + 8192) / 16384)
                constexpr int R0 = 8192 + (S2_LUT_BIAS << 14);
#if (S2_PROBE & 1)
                const int su = s2_dot2(pU[c + 1], a.hC[1], R0), sv = s2_dot2(pV[c + 1], a.hC[1], R0);
#else
                const int su = s2_dot2(pU[c + 3], a.hC[3], s2_dot2(pU[c + 2], a.hC[2], s2_dot2(pU[c + 1], a.hC[1], s2_dot2(pU[c], a.hC[0], R0))));
                const int sv = s2_dot2(pV[c + 3], a.hC[3], s2_dot2(pV[c + 2], a.hC[2], s2_dot2(pV[c + 1], a.hC[1], s2_dot2(pV[c], a.hC[0], R0))));
#endif
#if S2_LUT512
                iU[c] = (su >> 14) & (S2_LUT_N - 1); iV[c] = (sv >> 14) & (S2_LUT_N - 1);     // in [0, 511] by the host's bound; the mask is free
#else
                iU[c] = clip_u8_shr(su, 14); iV[c] = clip_u8_shr(sv, 14);
#endif
            }
            // ---- colour stage + store ---------------------------------------------------------------------------
            unsigned c0[4], c1[4], c2[4];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int2 tv = lutV[iV[c]], tu = lutU[iU[c]];
                const int tr = BGR ? tu.y : tv.x, tg = tv.y + tu.x, tb = BGR ? tv.x : tu.y;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int q = 2 * c + h;
#if S2_SATPK
                    c0[q] = (unsigned)(tr + m24(Y[q], a.y2r.cy));      // |term + Y cy| < 2^27: the channel is sat_u8 of the high half
                    c1[q] = (unsigned)(tg + m24(Y[q], a.y2r.cy));
                    c2[q] = (unsigned)(tb + m24(Y[q], a.y2r.cy));
#else
                    c0[q] = (unsigned)min(max(tr + m24(Y[q], a.y2r.cy), 0), 0xFFFFFF);
                    c1[q] = (unsigned)min(max(tg + m24(Y[q], a.y2r.cy), 0), 0xFFFFFF);
                    c2[q] = (unsigned)min(max(tb + m24(Y[q], a.y2r.cy), 0), 0xFFFFFF);
#endif
                }
            }
            if (active) {
#if S2_SBASE
                const unsigned drow = (unsigned)yo * (unsigned)a.ds;
    #define S2_ST16(v) bD.st16((v), dstOff, drow)
    #define S2_ST12(v) bD.st12((v), dstOff, drow)
#else
                uint8_t *d = pd + (unsigned)((unsigned)yo * (unsigned)a.ds + dstOff);
    #define S2_ST16(v) (*reinterpret_cast<uint4 *>(d) = (v))
    #define S2_ST12(v) (*reinterpret_cast<uint3 *>(d) = (v))
#endif
#if S2_SATPK
    // two channels -> bytes (0, 1) of a register: the high halves side by side, then the saturating pack
    #define S2_SAT2(x, y) s2_sat_pk_u8_i16(__builtin_amdgcn_perm((y), (x), 0x07060302u))
    #define S2_JOIN(lo2, hi2) __builtin_amdgcn_perm((hi2), (lo2), 0x05040100u)
                if (BPP == 4) {
                    uint4 o4;
                    o4.x = S2_JOIN(S2_SAT2(c0[0], c1[0]), S2_SAT2(c2[0], 0x00FF0000u));
                    o4.y = S2_JOIN(S2_SAT2(c0[1], c1[1]), S2_SAT2(c2[1], 0x00FF0000u));
                    o4.z = S2_JOIN(S2_SAT2(c0[2], c1[2]), S2_SAT2(c2[2], 0x00FF0000u));
                    o4.w = S2_JOIN(S2_SAT2(c0[3], c1[3]), S2_SAT2(c2[3], 0x00FF0000u));
                    S2_ST16(o4);
                } else {
                    uint3 o3;           // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
                    o3.x = S2_JOIN(S2_SAT2(c0[0], c1[0]), S2_SAT2(c2[0], c0[1]));
                    o3.y = S2_JOIN(S2_SAT2(c1[1], c2[1]), S2_SAT2(c0[2], c1[2]));
                    o3.z = S2_JOIN(S2_SAT2(c2[2], c0[3]), S2_SAT2(c1[3], c2[3]));
                    S2_ST12(o3);
                }
    #undef S2_SAT2
    #undef S2_JOIN
            }
        }
    };
#else
    #define S2_B2PAIR(lo, hi) __builtin_amdgcn_perm((hi), (lo), 0x0C0C0602u)
                if (BPP == 4) {
                    uint4 o4;
                    o4.x = S2_B2PAIR(c0[0], c1[0]) | (S2_B2PAIR(c2[0], 0u) << 16) | 0xFF000000u;
                    o4.y = S2_B2PAIR(c0[1], c1[1]) | (S2_B2PAIR(c2[1], 0u) << 16) | 0xFF000000u;
                    o4.z = S2_B2PAIR(c0[2], c1[2]) | (S2_B2PAIR(c2[2], 0u) << 16) | 0xFF000000u;
                    o4.w = S2_B2PAIR(c0[3], c1[3]) | (S2_B2PAIR(c2[3], 0u) << 16) | 0xFF000000u;
                    S2_ST16(o4);
                } else {
                    uint3 o3;           // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
                    o3.x = S2_B2PAIR(c0[0], c1[0]) | (S2_B2PAIR(c2[0], c0[1]) << 16);
                    o3.y = S2_B2PAIR(c1[1], c2[1]) | (S2_B2PAIR(c0[2], c1[2]) << 16);
                    o3.z = S2_B2PAIR(c2[2], c0[3]) | (S2_B2PAIR(c1[3], c2[3]) << 16);
                    S2_ST12(o3);
                }
    #undef S2_B2PAIR
            }
        }
    };
#endif

#undef S2_ST16
#undef S2_ST12

    auto run = [&](auto edge_c) {
        for (int j0 = 0; j0 < nIter; j0 += 4) {
            body(j0, std::integral_constant<int, 0>(), edge_c);
            if (j0 + 1 < nIter) body(j0 + 1, std::integral_constant<int, 1>(), edge_c);
            if (j0 + 2 < nIter) body(j0 + 2, std::integral_constant<int, 2>(), edge_c);
            if (j0 + 3 < nIter) body(j0 + 3, std::integral_constant<int, 3>(), edge_c);
        }
    };
    // wave-uniform: which frame edges this wave's strip touches
    const int edgeKind = (X0 == 0 ? 1 : 0) | (X0 + S2_STRIP >= a.dstW ? 2 : 0);
    if (edgeKind == 0) run(std::integral_constant<int, 0>());
    else if (edgeKind == 1) run(std::integral_constant<int, 1>());
    else if (edgeKind == 2) run(std::integral_constant<int, 2>());
    else run(std::integral_constant<int, 3>());
}

// ---------------------------------------------------------------------------------------------------------------------
// ONE FRAME PER LAUNCH (round 4): the block-cooperative form.  What sws_scale() / filter_frame() issue is one frame, and a launch of
// one 4K -> 1080p frame is 2880 of the walker's waves for 6144 wave slots, each a CHAIN of 7 dependent memory round trips (one
// prefetched row pair per iteration) with 3 warm-up row pairs for its 3 output rows — the launch lasts as long as that chain, 5.2 us
// of a 6.8 us call (profiles/r04a_x2bench_1frame_baseline.txt), while the same frame inside a 32-frame launch costs 3.6 us.  Here a
// block of four waves owns ONE strip of 256 output columns and 4 * RW output rows:
//   * every wave issues ALL its loads at once — the RW + 1 row pairs it filters horizontally (pairs w, w + 4, ...: the 4 RW + 3 pairs of the
//     band dealt round) and the RW chroma rows of its own output rows — so a block pays ONE memory round trip, not 4 RW + 3;
//   * the horizontally filtered pairs (int16 pairs, 16 bytes a lane) go through LDS (1 KB a pair), one barrier, and every wave
//     makes its RW output rows from the RW + 3 pairs they touch: the 3 warm-up pairs are filtered once per 4 RW rows, not once per 3;
//   * the colour tables are built while the loads are in flight and share the barrier.
// Same arithmetic, operation by operation, as scale_yuv2s_kernel (the walker keeps every launch of more than a few frames: its
// register window moves nothing through LDS).
template <bool NV12, int DST, int RW>
__global__ __launch_bounds__(256) void scale_yuv2s_blk_kernel(Yuv2sArgs a, Yuv2xFrames fr)
{
    constexpr bool BGR = (DST & 1) != 0;
    constexpr int BPP = DST >= 2 ? 4 : 3;
    constexpr int R = 4 * RW, NPAIR = R + 3, PW = RW + 1;       // rows a block, row pairs it filters, pairs a wave
    __shared__ int2 lutV[S2_LUT_N], lutU[S2_LUT_N];
    __shared__ uint4 hwS[NPAIR][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nstrips = a.nsg;                                  // (this launcher: strips a row, not groups of four)
    const int nblk = a.nseg * nstrips;
    int lin = blockIdx.x;
    if (a.xcdRemap) {
        const int chunk = (nblk + 7) >> 3;
        lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    }
    if (lin >= nblk) return;                                    // block-uniform
    const int seg = lin / nstrips;
    const int X0 = (lin - seg * nstrips) * S2_STRIP;
    const int y0 = seg * R;

    const int f = blockIdx.y;
    const S2Plane bY(fr.y[f]), bU(fr.u[f]), bV(NV12 ? fr.u[f] : fr.v[f]), bD(fr.dst[f]);

    // ---- per-lane constants: as in scale_yuv2s_kernel -------------------------------------------------------------------
    const int xo = X0 + 4 * lane;
    const bool active = xo < a.dstW;
    const int xc = active ? xo : a.dstW - 4;
    const int wantL = 2 * xc - 4;
    const int offL = min(max(wantL, 0), a.srcW - 16);
    const int shL = wantL - offL;
    int offA, offB, shA, shB;
    if (NV12) {
        offA = max(2 * xc - 8, 0);               shA = 2 * xc - 8 - offA;
        offB = min(2 * xc + 8, 2 * a.chrSrcW - 8); shB = 2 * xc + 8 - offB;
    } else {
        offA = min(max(xc - 4, 0), a.chrSrcW - 12); shA = xc - 4 - offA;
        offB = 0; shB = 0;
    }
    const unsigned dstOff = (unsigned)xo * BPP;
    const bool edgeLL = shL < 0, edgeLR = shL > 0, edgeAL = shA < 0, edgeAR = shA > 0, edgeBR = shB > 0;
    const unsigned uoffL = (unsigned)offL, uoffA = (unsigned)offA, uoffB = (unsigned)offB;

    // ---- every load of this wave, back to back ------------------------------------------------------------------------------
    uint4 la[PW], lb[PW], ca[RW], cb[RW];
#pragma unroll
    for (int i = 0; i < PW; i++) {
        const int m = y0 - 1 + wave + 4 * i;                    // pair m = rows 2m - 1, 2m (wave 3's last one lies past the band: unused)
        const int ra = min(max(2 * m - 1, 0), a.srcH - 1), rb = min(max(2 * m, 0), a.srcH - 1);
        la[i] = bY.ld16(uoffL, (unsigned)ra * (unsigned)a.ys);
        lb[i] = bY.ld16(uoffL, (unsigned)rb * (unsigned)a.ys);
    }
#pragma unroll
    for (int i = 0; i < RW; i++) {
        const int r = min(max(y0 + RW * wave + i, 0), a.chrSrcH - 1);
        if (NV12) {
            const unsigned ro = (unsigned)r * (unsigned)a.us;
            ca[i] = bU.ld16(uoffA, ro);
            const uint2 t = bU.ld8(uoffB, ro);
            cb[i] = make_uint4(t.x, t.y, 0u, 0u);
        } else {
            const uint3 tu = bU.ld12(uoffA, (unsigned)r * (unsigned)a.us);
            const uint3 tv = bV.ld12(uoffA, (unsigned)r * (unsigned)a.vs);
            ca[i] = make_uint4(tu.x, tu.y, tu.z, 0u);
            cb[i] = make_uint4(tv.x, tv.y, tv.z, 0u);
        }
    }
    // ---- the colour tables, while the loads are in flight (scale_yuv2s_kernel's, entry by entry) -----------------------------
    {
        const Yuv2RgbConsts &k = a.y2r;
#pragma unroll
        for (int e = tid; e < S2_LUT_N; e += 256) {
            const int i = min(max(e - S2_LUT_BIAS, 0), 255);
            lutV[e] = make_int2(k.base + m24(k.offR + (m24(i, k.crv) >> 16), k.cy), m24(m24(i, k.cgv) >> 16, k.cy));
            lutU[e] = make_int2(k.base + m24(k.offG + (m24(i, k.cgu) >> 16), k.cy), k.base + m24(k.offB + (m24(i, k.cbu) >> 16), k.cy));
        }
    }

    auto fix_luma = [&](uint4 L, auto edge_c) -> uint4 {
        constexpr int K = decltype(edge_c)::value;
        if constexpr (K == 1) {
            const unsigned first = s2_rep(L.x, 0x00000000u);
            L = make_uint4(edgeLL ? first : L.x, edgeLL ? L.x : L.y, edgeLL ? L.y : L.z, edgeLL ? L.z : L.w);
        } else if constexpr (K == 2) {
            const unsigned last = s2_rep(L.w, 0x03030303u);
            L = make_uint4(edgeLR ? L.y : L.x, edgeLR ? L.z : L.y, edgeLR ? L.w : L.z, edgeLR ? last : L.w);
        } else if constexpr (K == 3) {
            const unsigned first = s2_rep(L.x, 0x00000000u), last = s2_rep(L.w, 0x03030303u);
            L = make_uint4(edgeLL ? first : edgeLR ? L.y : L.x, edgeLL ? L.x : edgeLR ? L.z : L.y,
                           edgeLL ? L.y : edgeLR ? L.w : L.z, edgeLL ? L.z : edgeLR ? last : L.w);
        }
        return L;
    };
    auto hrow = [&](const uint4 &L, int (&s)[4]) {
        int p[7];
        p[0] = s2_pair12(L.x); p[1] = s2_pair30(L.y, L.x); p[2] = s2_pair12(L.y); p[3] = s2_pair30(L.z, L.y);
        p[4] = s2_pair12(L.z); p[5] = s2_pair30(L.w, L.z); p[6] = s2_pair12(L.w);
#pragma unroll
        for (int j = 0; j < 4; j++)
            s[j] = s2_dot2(p[j + 3], a.hL[3], s2_dot2(p[j + 2], a.hL[2], s2_dot2(p[j + 1], a.hL[1], s2_dot2(p[j], a.hL[0], 0))));
    };

    auto go = [&](auto edge_c) {
        constexpr int EDGE = decltype(edge_c)::value;
        // ---- horizontal luma of this wave's pairs -> LDS -------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < PW; i++) {
            const int k = wave + 4 * i;
            if (k < NPAIR) {
                int sa[4], sb[4];
                hrow(fix_luma(la[i], edge_c), sa);
                hrow(fix_luma(lb[i], edge_c), sb);
                uint4 h;
                h.x = (unsigned)__builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(sa[0] >> 7, sb[0] >> 7));
                h.y = (unsigned)__builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(sa[1] >> 7, sb[1] >> 7));
                h.z = (unsigned)__builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(sa[2] >> 7, sb[2] >> 7));
                h.w = (unsigned)__builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(sa[3] >> 7, sb[3] >> 7));
                hwS[k][lane] = h;
            }
        }
        __syncthreads();                                        // the band's pairs and the colour tables
        // ---- this wave's output rows -----------------------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < RW; i++) {
            const int r = RW * wave + i, yo = y0 + r;
            if (yo >= a.dstH) break;                            // wave-uniform (the frame's last band)
            const uint4 h0 = hwS[r][lane], h1 = hwS[r + 1][lane], h2 = hwS[r + 2][lane], h3 = hwS[r + 3][lane];
            const unsigned q0[4] = {h0.x, h0.y, h0.z, h0.w}, q1[4] = {h1.x, h1.y, h1.z, h1.w}, q2[4] = {h2.x, h2.y, h2.z, h2.w}, q3[4] = {h3.x, h3.y, h3.z, h3.w};
            int Y[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int acc = a.lr;
                acc = s2_dot2((int)q0[q], a.vL[0], acc);
                acc = s2_dot2((int)q1[q], a.vL[1], acc);
                acc = s2_dot2((int)q2[q], a.vL[2], acc);
                acc = s2_dot2((int)q3[q], a.vL[3], acc);
                Y[q] = acc >> 19;
            }
            int pU[5], pV[5];
            if (NV12) {
                unsigned e[6] = {ca[i].x, ca[i].y, ca[i].z, ca[i].w, cb[i].x, cb[i].y};
                if constexpr ((EDGE & 1) != 0) {
                    const unsigned rr = s2_rep(e[0], 0x01000100u), e0 = e[0], e1 = e[1];
                    e[0] = edgeAL ? rr : e0; e[1] = edgeAL ? rr : e1; e[2] = edgeAL ? e0 : e[2]; e[3] = edgeAL ? e1 : e[3];
                }
                if constexpr ((EDGE & 2) != 0) {
                    const unsigned rr = s2_rep(e[5], 0x03020302u);
                    e[4] = edgeBR ? rr : e[4]; e[5] = edgeBR ? rr : e[5];
                }
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    pU[k] = (int)__builtin_amdgcn_perm(e[k + 1], e[k], 0x0C040C02u);
                    pV[k] = (int)__builtin_amdgcn_perm(e[k + 1], e[k], 0x0C050C03u);
                }
            } else {
                unsigned fu[3] = {ca[i].x, ca[i].y, ca[i].z}, fv[3] = {cb[i].x, cb[i].y, cb[i].z};
                if constexpr (EDGE != 0) {
                    const unsigned u0 = fu[0], u1 = fu[1], u2 = fu[2], v0 = fv[0], v1 = fv[1], v2 = fv[2];
                    if constexpr (EDGE == 1) {
                        fu[0] = edgeAL ? s2_rep(u0, 0x00000000u) : u0; fu[1] = edgeAL ? u0 : u1; fu[2] = edgeAL ? u1 : u2;
                        fv[0] = edgeAL ? s2_rep(v0, 0x00000000u) : v0; fv[1] = edgeAL ? v0 : v1; fv[2] = edgeAL ? v1 : v2;
                    } else if constexpr (EDGE == 2) {
                        fu[0] = edgeAR ? u1 : u0; fu[1] = edgeAR ? u2 : u1; fu[2] = edgeAR ? s2_rep(u2, 0x03030303u) : u2;
                        fv[0] = edgeAR ? v1 : v0; fv[1] = edgeAR ? v2 : v1; fv[2] = edgeAR ? s2_rep(v2, 0x03030303u) : v2;
                    } else {
                        const unsigned uf = s2_rep(u0, 0x00000000u), ul = s2_rep(u2, 0x03030303u), vf = s2_rep(v0, 0x00000000u), vl = s2_rep(v2, 0x03030303u);
                        fu[0] = edgeAL ? uf : edgeAR ? u1 : u0; fu[1] = edgeAL ? u0 : edgeAR ? u2 : u1; fu[2] = edgeAL ? u1 : edgeAR ? ul : u2;
                        fv[0] = edgeAL ? vf : edgeAR ? v1 : v0; fv[1] = edgeAL ? v0 : edgeAR ? v2 : v1; fv[2] = edgeAL ? v1 : edgeAR ? vl : v2;
                    }
                }
                pU[0] = s2_pair12(fu[0]); pU[1] = s2_pair30(fu[1], fu[0]); pU[2] = s2_pair12(fu[1]); pU[3] = s2_pair30(fu[2], fu[1]); pU[4] = s2_pair12(fu[2]);
                pV[0] = s2_pair12(fv[0]); pV[1] = s2_pair30(fv[1], fv[0]); pV[2] = s2_pair12(fv[1]); pV[3] = s2_pair30(fv[2], fv[1]); pV[4] = s2_pair12(fv[2]);
            }
            int iU[2], iV[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                constexpr int R0 = 8192 + (S2_LUT_BIAS << 14);
                const int su = s2_dot2(pU[c + 3], a.hC[3], s2_dot2(pU[c + 2], a.hC[2], s2_dot2(pU[c + 1], a.hC[1], s2_dot2(pU[c], a.hC[0], R0))));
                const int sv = s2_dot2(pV[c + 3], a.hC[3], s2_dot2(pV[c + 2], a.hC[2], s2_dot2(pV[c + 1], a.hC[1], s2_dot2(pV[c], a.hC[0], R0))));
#if S2_LUT512
                iU[c] = (su >> 14) & (S2_LUT_N - 1); iV[c] = (sv >> 14) & (S2_LUT_N - 1);
#else
                iU[c] = clip_u8_shr(su, 14); iV[c] = clip_u8_shr(sv, 14);
#endif
            }
            unsigned c0[4], c1[4], c2[4];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int2 tv = lutV[iV[c]], tu = lutU[iU[c]];
                const int tr = BGR ? tu.y : tv.x, tg = tv.y + tu.x, tb = BGR ? tv.x : tu.y;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int q = 2 * c + h;
                    c0[q] = (unsigned)(tr + m24(Y[q], a.y2r.cy));      // |term + Y cy| < 2^27: the channel is sat_u8 of the high half
                    c1[q] = (unsigned)(tg + m24(Y[q], a.y2r.cy));
                    c2[q] = (unsigned)(tb + m24(Y[q], a.y2r.cy));
                }
            }
            if (active) {
                const unsigned drow = (unsigned)yo * (unsigned)a.ds;
    #define S2_SAT2(x, y) s2_sat_pk_u8_i16(__builtin_amdgcn_perm((y), (x), 0x07060302u))
    #define S2_JOIN(lo2, hi2) __builtin_amdgcn_perm((hi2), (lo2), 0x05040100u)
                if (BPP == 4) {
                    uint4 o4;
                    o4.x = S2_JOIN(S2_SAT2(c0[0], c1[0]), S2_SAT2(c2[0], 0x00FF0000u));
                    o4.y = S2_JOIN(S2_SAT2(c0[1], c1[1]), S2_SAT2(c2[1], 0x00FF0000u));
                    o4.z = S2_JOIN(S2_SAT2(c0[2], c1[2]), S2_SAT2(c2[2], 0x00FF0000u));
                    o4.w = S2_JOIN(S2_SAT2(c0[3], c1[3]), S2_SAT2(c2[3], 0x00FF0000u));
                    bD.st16(o4, dstOff, drow);
                } else {
                    uint3 o3;           // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
                    o3.x = S2_JOIN(S2_SAT2(c0[0], c1[0]), S2_SAT2(c2[0], c0[1]));
                    o3.y = S2_JOIN(S2_SAT2(c1[1], c2[1]), S2_SAT2(c0[2], c1[2]));
                    o3.z = S2_JOIN(S2_SAT2(c2[2], c0[3]), S2_SAT2(c1[3], c2[3]));
                    bD.st12(o3, dstOff, drow);
                }
    #undef S2_SAT2
    #undef S2_JOIN
            }
        }
    };
    const int edgeKind = (X0 == 0 ? 1 : 0) | (X0 + S2_STRIP >= a.dstW ? 2 : 0);      // block-uniform: every wave meets the barrier inside go()
    if (edgeKind == 0) go(std::integral_constant<int, 0>());
    else if (edgeKind == 1) go(std::integral_constant<int, 1>());
    else if (edgeKind == 2) go(std::integral_constant<int, 2>());
    else go(std::integral_constant<int, 3>());
}

// ---------------------------------------------------------------------------------------------------------------------
// The same kernel written over the number of coefficient pairs NP, shipped for NP = 6 only (Lanczos-3: 12 taps on
// [2x - 5, 2x + 6], a 6-slot vertical window, 24 luma bytes per row and lane).  The 4-pair kernel above is NOT an instantiation
// of this template on purpose: written this way the 4-pair form compiles to a different schedule (70 instead of 78 VGPRs) that
// measured 10 % slower on the headline (4.58 against 4.10 us per 4K frame, same box as an untouched kernel that did not move).
// The pixels one row-loop iteration consumes: a luma row pair (2m-1, 2m) and one chroma row.
// NP = 4 (8 taps): 16 luma bytes of each row from column 2xc - 4; NV12: 24 chroma bytes (8 + 4 UV pairs) from sample xc - 4,
//                  planar: 12 bytes of each chroma plane from sample xc - 4.
// NP = 6 (Lanczos-3, 12 taps): 24 luma bytes from column 2xc - 8; NV12: 32 chroma bytes from sample xc - 6; planar: 24 bytes of
//                  each plane from sample xc - 8.
struct S2PixN {
    unsigned la[6], lb[6];         // luma rows 2m-1 and 2m
    unsigned ca[8];                // NV12: UV pairs;  planar: U bytes (first 3 / 6 dwords)
    unsigned cb[6];                // NV12 (NP = 4 only): ca continues in cb[0..1];  planar: V bytes
};

__device__ __forceinline__ unsigned s2_ld4(const uint8_t *p) { return *reinterpret_cast<const unsigned *>(p); }

// DST: 0 rgb24, 1 bgr24, 2 rgba, 3 bgra.  NP: coefficient pairs per filter — 4 for the 8-tap filters of an exact 2:1 scale
// (the headline), 6 for Lanczos-3.
template <bool NV12, int DST, int NP>
__global__ __launch_bounds__(256) void scale_yuv2s_np_kernel(Yuv2sArgs a, Yuv2xFrames fr)
{
    constexpr bool BGR = (DST & 1) != 0;
    constexpr int BPP = DST >= 2 ? 4 : 3;
    constexpr int NDL = NP == 4 ? 4 : 6;                        // luma dwords per row
    constexpr int BLL = NP == 4 ? 4 : 8;                        // luma bytes between the window base and 2xc
    constexpr int NDC = NV12 ? NP + 2 : (NP == 4 ? 3 : 6);      // chroma dwords per row (planar: per plane)
    __shared__ int2 lutV[256], lutU[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- colour look-up tables: chan = byte 2 of clamp(term + Y * cy, 0, 0xFFFFFF) with
    //      term_R = lutV[V].x, term_G = lutV[V].y + lutU[U].x, term_B = lutU[U].y  (px_math.h chroma_terms, split by sample)
    {
        const Yuv2RgbConsts &k = a.y2r;
        const int i = tid;
        lutV[i] = make_int2(k.base + m24(k.offR + (m24(i, k.crv) >> 16), k.cy), m24(m24(i, k.cgv) >> 16, k.cy));
        lutU[i] = make_int2(k.base + m24(k.offG + (m24(i, k.cgu) >> 16), k.cy), k.base + m24(k.offB + (m24(i, k.cbu) >> 16), k.cy));
    }
    __syncthreads();

    // ---- which strip segment ------------------------------------------------------------------------------------
    const int nblk = a.nseg * a.nsg;
    int lin = blockIdx.x;
    if (a.xcdRemap) {
        const int chunk = (nblk + 7) >> 3;
        lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    }
    if (lin >= nblk) return;
    const int seg = __builtin_amdgcn_readfirstlane(lin / a.nsg);
    const int X0 = ((lin - seg * a.nsg) * 4 + wave) * S2_STRIP;
    if (X0 >= a.dstW) return;                                  // wave-uniform; no barrier below
    const int y0 = seg * a.segRows;
    const int nOut = min(a.segRows, a.dstH - y0);
    const int nIter = nOut + NP - 1;                           // NP - 1 warm-up row pairs fill the vertical window
    const int m0 = y0 - (NP / 2 - 1);                          // row pair of iteration 0 (pair m = rows 2m - 1, 2m)

    const uint8_t *py, *pu, *pv;
    uint8_t *pd;
    {
        const int f = blockIdx.y;
        py = fr.y[f]; pu = fr.u[f]; pv = fr.v[f]; pd = fr.dst[f];
    }

    // ---- per-lane constants -------------------------------------------------------------------------------------
    const int xo = X0 + 4 * lane;
    const bool active = xo < a.dstW;
    const int xc = active ? xo : a.dstW - 4;                    // idle lanes shadow the last group (loads stay inside the rows)
    // wave-uniform: only these waves hold a lane whose window reaches past a frame edge.  The 12-tap window (NP = 6) overhangs
    // further: the chroma loads of a lane reach sample xc + 9, so the wave BEFORE the last one is an edge wave too when fewer
    // than 8 columns follow it
    const bool edgeWave = X0 == 0 || X0 + S2_STRIP + (NP == 4 ? 0 : 8) >= a.dstW;
    // luma: bytes [2xc - BLL, ...) of the row; sources < 0 and >= srcW are the replicated edge samples
    const int wantL = 2 * xc - BLL;
    const int offL = min(max(wantL, 0), a.srcW - 16);           // NP = 4: the window clamped as a whole, shifted back in registers
    const int shL = wantL - offL;                               //         -4 at the left frame edge, +4 at the right one
    const int wdL = wantL >> 2;                                 // NP = 6: dword index of the window base (wantL is a multiple of 8)
    const int lastL = (a.srcW >> 2) - 1;
    // chroma samples [xc - 4, xc + 8) (NP = 4) / [xc - 6 | xc - 8, ...) (NP = 6) of the row
    int offA, offB, shA, shB;
    if (NV12) {
        offA = max(2 * xc - 8, 0);               shA = 2 * xc - 8 - offA;        // 16 bytes: samples xc-4 .. xc+3   (-8: left edge)
        offB = min(2 * xc + 8, 2 * a.chrSrcW - 8); shB = 2 * xc + 8 - offB;      //  8 bytes: samples xc+4 .. xc+7   (+8: right edge)
    } else {
        offA = min(max(xc - 4, 0), a.chrSrcW - 12); shA = xc - 4 - offA;         // 12 bytes of each plane (-4 / +4)
        offB = 0; shB = 0;
    }
    const int wdC = NV12 ? (2 * xc - 12) >> 2 : (xc - 8) >> 2;  // NP = 6: dword index of the chroma window base (arithmetic shift: floor)
    const int lastC = ((NV12 ? 2 * a.chrSrcW : a.chrSrcW) >> 2) - 1;
    const unsigned dstOff = (unsigned)xo * BPP;

    // row pointers are wave-uniform (scalar unit), the lane offsets unsigned 32-bit: global_load with an SGPR base
    const unsigned uoffL = (unsigned)offL, uoffA = (unsigned)offA, uoffB = (unsigned)offB;
    auto load_luma = [&](int m, S2PixN &P, auto edge_c) {
        const int ra = min(max(2 * m - 1, 0), a.srcH - 1), rb = min(max(2 * m, 0), a.srcH - 1);
        // one 32-bit offset per load = scalar row offset + lane offset (a plane is far below 4 GB): the frame pointer
        // stays the SGPR base of the global_load
        if constexpr (NP == 4) {
            const uint4 ta = s2_ld16(py + (unsigned)((unsigned)ra * (unsigned)a.ys + uoffL));
            const uint4 tb = s2_ld16(py + (unsigned)((unsigned)rb * (unsigned)a.ys + uoffL));
            P.la[0] = ta.x; P.la[1] = ta.y; P.la[2] = ta.z; P.la[3] = ta.w;
            P.lb[0] = tb.x; P.lb[1] = tb.y; P.lb[2] = tb.z; P.lb[3] = tb.w;
        } else if constexpr (!decltype(edge_c)::value) {
            const unsigned oa = (unsigned)ra * (unsigned)a.ys + (unsigned)wantL, ob = (unsigned)rb * (unsigned)a.ys + (unsigned)wantL;
            const uint4 ta = s2_ld16(py + oa); const uint2 ta2 = s2_ld8(py + (unsigned)(oa + 16u));
            const uint4 tb = s2_ld16(py + ob); const uint2 tb2 = s2_ld8(py + (unsigned)(ob + 16u));
            P.la[0] = ta.x; P.la[1] = ta.y; P.la[2] = ta.z; P.la[3] = ta.w; P.la[4] = ta2.x; P.la[5] = ta2.y;
            P.lb[0] = tb.x; P.lb[1] = tb.y; P.lb[2] = tb.z; P.lb[3] = tb.w; P.lb[4] = tb2.x; P.lb[5] = tb2.y;
        } else {
            // a wave on a frame edge: two lanes a side overlap the edge by different amounts — every dword from its own
            // clamped address, the edge sample replicated where the index was clamped (fix_luma)
#pragma unroll
            for (int i = 0; i < NDL; i++) {
                const unsigned c = 4u * (unsigned)min(max(wdL + i, 0), lastL);
                P.la[i] = s2_ld4(py + (unsigned)((unsigned)ra * (unsigned)a.ys + c));
                P.lb[i] = s2_ld4(py + (unsigned)((unsigned)rb * (unsigned)a.ys + c));
            }
        }
    };
    auto load_chroma = [&](int cy, S2PixN &P, auto edge_c) {
        const int r = min(max(cy, 0), a.chrSrcH - 1);
        if constexpr (NP == 4) {
            if (NV12) {
                const unsigned ro = (unsigned)r * (unsigned)a.us;
                const uint4 t0 = s2_ld16(pu + (unsigned)(ro + uoffA));
                const uint2 t = s2_ld8(pu + (unsigned)(ro + uoffB));
                P.ca[0] = t0.x; P.ca[1] = t0.y; P.ca[2] = t0.z; P.ca[3] = t0.w; P.ca[4] = t.x; P.ca[5] = t.y;
            } else {
                const uint3 tu = s2_ld12(pu + (unsigned)((unsigned)r * (unsigned)a.us + uoffA));
                const uint3 tv = s2_ld12(pv + (unsigned)((unsigned)r * (unsigned)a.vs + uoffA));
                P.ca[0] = tu.x; P.ca[1] = tu.y; P.ca[2] = tu.z;
                P.cb[0] = tv.x; P.cb[1] = tv.y; P.cb[2] = tv.z;
            }
        } else if constexpr (!decltype(edge_c)::value) {
            if (NV12) {
                const unsigned o = (unsigned)r * (unsigned)a.us + 4u * (unsigned)wdC;
                const uint4 t0 = s2_ld16(pu + o), t1 = s2_ld16(pu + (unsigned)(o + 16u));
                P.ca[0] = t0.x; P.ca[1] = t0.y; P.ca[2] = t0.z; P.ca[3] = t0.w; P.ca[4] = t1.x; P.ca[5] = t1.y; P.ca[6] = t1.z; P.ca[7] = t1.w;
            } else {
                const unsigned ou = (unsigned)r * (unsigned)a.us + 4u * (unsigned)wdC, ov = (unsigned)r * (unsigned)a.vs + 4u * (unsigned)wdC;
                const uint4 u0 = s2_ld16(pu + ou); const uint2 u1 = s2_ld8(pu + (unsigned)(ou + 16u));
                const uint4 v0 = s2_ld16(pv + ov); const uint2 v1 = s2_ld8(pv + (unsigned)(ov + 16u));
                P.ca[0] = u0.x; P.ca[1] = u0.y; P.ca[2] = u0.z; P.ca[3] = u0.w; P.ca[4] = u1.x; P.ca[5] = u1.y;
                P.cb[0] = v0.x; P.cb[1] = v0.y; P.cb[2] = v0.z; P.cb[3] = v0.w; P.cb[4] = v1.x; P.cb[5] = v1.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NDC; i++) {
                const unsigned c = 4u * (unsigned)min(max(wdC + i, 0), lastC);
                P.ca[i] = s2_ld4(pu + (unsigned)((unsigned)r * (unsigned)a.us + c));
                if (!NV12) P.cb[i] = s2_ld4(pv + (unsigned)((unsigned)r * (unsigned)a.vs + c));
            }
        }
    };
    // frame-edge lanes: shift the dwords into window position and replicate the edge sample (NP = 4); replicate where the
    // address was clamped (NP = 6)
    auto fix_luma = [&](const unsigned (&src)[6], unsigned (&L)[NDL], auto edge_c) {
#pragma unroll
        for (int i = 0; i < NDL; i++) L[i] = src[i];
        if constexpr (decltype(edge_c)::value && NP == 4) {
            if (shL < 0) { const unsigned r = s2_rep(L[0], 0x00000000u); L[3] = L[2]; L[2] = L[1]; L[1] = L[0]; L[0] = r; }
            else if (shL > 0) { const unsigned r = s2_rep(L[3], 0x03030303u); L[0] = L[1]; L[1] = L[2]; L[2] = L[3]; L[3] = r; }
        }
        if constexpr (decltype(edge_c)::value && NP != 4) {
#pragma unroll
            for (int i = 0; i < NDL; i++) {
                const int idx = wdL + i;
                const unsigned lo = s2_rep(L[i], 0x00000000u), hi = s2_rep(L[i], 0x03030303u);
                L[i] = idx < 0 ? lo : idx > lastL ? hi : L[i];
            }
        }
    };

    // horizontal luma filter of one row: 4 outputs from NP + 3 odd-aligned pairs
    auto hrow = [&](const unsigned (&L)[NDL], int (&s)[4]) {
        int p[NP + 3];
        constexpr int O0 = BLL - (NP - 1);                      // byte of the first pair: 1 (NP = 4) or 3 (NP = 6)
#pragma unroll
        for (int k = 0; k < NP + 3; k++) {
            const int o = O0 + 2 * k;
            p[k] = (o & 3) == 1 ? s2_pair12(L[o >> 2]) : s2_pair30(L[(o >> 2) + 1], L[o >> 2]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if constexpr (NP == 4) {
                s[j] = s2_dot2(p[j + 3], a.hL[3], s2_dot2(p[j + 2], a.hL[2], s2_dot2(p[j + 1], a.hL[1], s2_dot2(p[j], a.hL[0], 0))));
            } else {
                int acc = 0;
#pragma unroll
                for (int k = 0; k < NP; k++) acc = s2_dot2(p[j + k], a.hL[k], acc);
                s[j] = acc;
            }
        }
    };

    int hw[NP][4];                                              // [slot][output]: (row 2m-1 | row 2m << 16) after hScale8To15_c
#pragma unroll
    for (int s = 0; s < NP; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) hw[s][j] = 0;

    S2PixN buf[2];                                               // ping-pong: iteration j consumes buf[j & 1], prefetches into the other
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int k = 0; k < 6; k++) buf[i].la[k] = buf[i].lb[k] = buf[i].cb[k] = 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) buf[i].ca[k] = 0u;
    }

    // EDGE: the wave holds a frame-edge lane (first / last strip).  The whole row loop exists twice so that interior
    // waves carry none of the fix-up moves.
    auto body = [&](const int j, auto slot_c, auto edge_c) {
        constexpr int SLOT = decltype(slot_c)::value;           // j mod NP, static after unrolling (NP even: j & 1 == SLOT & 1)
        constexpr bool EDGE = decltype(edge_c)::value;
        const S2PixN &cur = buf[SLOT & 1];
        S2PixN &nxt = buf[(SLOT + 1) & 1];
        // ---- prefetch the next iteration's rows ------------------------------------------------------------
        if (j + 1 < nIter) {
            load_luma(m0 + j + 1, nxt, edge_c);                                      // row pair of iteration j + 1
            if (j + 1 >= NP - 1) load_chroma(y0 + (j + 1) - (NP - 1), nxt, edge_c);  // chroma row of its output row
        }
        // ---- horizontal luma of this iteration's pair -> slot ----------------------------------------------
        {
            int sa[4], sb[4];
            unsigned La[NDL], Lb[NDL];
            fix_luma(cur.la, La, edge_c); fix_luma(cur.lb, Lb, edge_c);
            hrow(La, sa);
            hrow(Lb, sb);
#pragma unroll
            for (int q = 0; q < 4; q++)       // hScale8To15_c: min(val >> 7, 32767); the lower bound cannot trigger
                hw[SLOT][q] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(sa[q] >> 7, sb[q] >> 7));
        }
        if (j >= NP - 1) {
            const int yo = y0 + j - (NP - 1);
            // ---- vertical luma: the NP pairs of this output row sit in slots SLOT+1 .. SLOT+NP (mod NP), oldest first ------
            int Y[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int acc = a.lr;
#pragma unroll
                for (int k = 0; k < NP; k++) acc = s2_dot2(hw[(SLOT + 1 + k) % NP][q], a.vL[k], acc);
                Y[q] = acc >> 19;
            }
            // ---- chroma of row yo: 2 outputs per plane from NP + 1 odd-aligned pairs -----------------------------
            int pU[NP + 1], pV[NP + 1];
            if (NV12) {
                unsigned e[NP + 2];
#pragma unroll
                for (int k = 0; k < NP + 2; k++) e[k] = cur.ca[k];
                if constexpr (EDGE && NP == 4) {
                    if (shA < 0) { const unsigned r = s2_rep(e[0], 0x01000100u); e[3] = e[1]; e[2] = e[0]; e[0] = e[1] = r; }
                    if (shB > 0) { e[4] = e[5] = s2_rep(e[5], 0x03020302u); }
                }
                if constexpr (EDGE && NP != 4) {
#pragma unroll
                    for (int i = 0; i < NP + 2; i++) {
                        const int idx = wdC + i;
                        const unsigned lo = s2_rep(e[i], 0x01000100u), hi = s2_rep(e[i], 0x03020302u);
                        e[i] = idx < 0 ? lo : idx > lastC ? hi : e[i];
                    }
                }
#pragma unroll
                for (int k = 0; k < NP + 1; k++) {  // samples (2k-(NP-1), ...) rel. to xc: bytes 2,3 of e[k] and 0,1 of e[k+1]
                    pU[k] = (int)__builtin_amdgcn_perm(e[k + 1], e[k], 0x0C040C02u);
                    pV[k] = (int)__builtin_amdgcn_perm(e[k + 1], e[k], 0x0C050C03u);
                }
            } else {
                constexpr int NDP = NP == 4 ? 3 : 6;
                unsigned fu[NDP], fv[NDP];
#pragma unroll
                for (int k = 0; k < NDP; k++) { fu[k] = cur.ca[k]; fv[k] = cur.cb[k]; }
                if constexpr (EDGE && NP == 4) {
                    if (shA < 0) {
                        fu[2] = fu[1]; fu[1] = fu[0]; fu[0] = s2_rep(fu[0], 0x00000000u);
                        fv[2] = fv[1]; fv[1] = fv[0]; fv[0] = s2_rep(fv[0], 0x00000000u);
                    } else if (shA > 0) {
                        fu[0] = fu[1]; fu[1] = fu[2]; fu[2] = s2_rep(fu[2], 0x03030303u);
                        fv[0] = fv[1]; fv[1] = fv[2]; fv[2] = s2_rep(fv[2], 0x03030303u);
                    }
                }
                if constexpr (EDGE && NP != 4) {
#pragma unroll
                    for (int i = 0; i < NDP; i++) {
                        const int idx = wdC + i;
                        fu[i] = idx < 0 ? s2_rep(fu[i], 0x00000000u) : idx > lastC ? s2_rep(fu[i], 0x03030303u) : fu[i];
                        fv[i] = idx < 0 ? s2_rep(fv[i], 0x00000000u) : idx > lastC ? s2_rep(fv[i], 0x03030303u) : fv[i];
                    }
                }
                // NP = 4: window from sample xc - 4, first pair at byte 1;  NP = 6: from xc - 8, first pair (xc - 5, xc - 4) at byte 3
                constexpr int O0 = NP == 4 ? 1 : 3;
#pragma unroll
                for (int k = 0; k < NP + 1; k++) {
                    const int o = O0 + 2 * k;
                    pU[k] = (o & 3) == 1 ? s2_pair12(fu[o >> 2]) : s2_pair30(fu[(o >> 2) + 1], fu[o >> 2]);
                    pV[k] = (o & 3) == 1 ? s2_pair12(fv[o >> 2]) : s2_pair30(fv[(o >> 2) + 1], fv[o >> 2]);
                }
            }
            int iU[2], iV[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                // hScale8To15_c (>> 7, min 32767), the one-tap vertical filter (1 << 18) + h * 4096, >> 19 and the table
                // index clamp collapse into clip_u8((sum + 8192) >> 14): floor(floor(x / 128 + 64) / 128) = floor((x + 8192) / 16384)
                int su = 8192, sv = 8192;
                if constexpr (NP == 4) {
                    su = s2_dot2(pU[c + 3], a.hC[3], s2_dot2(pU[c + 2], a.hC[2], s2_dot2(pU[c + 1], a.hC[1], s2_dot2(pU[c], a.hC[0], 8192))));
                    sv = s2_dot2(pV[c + 3], a.hC[3], s2_dot2(pV[c + 2], a.hC[2], s2_dot2(pV[c + 1], a.hC[1], s2_dot2(pV[c], a.hC[0], 8192))));
                } else {
#pragma unroll
                    for (int k = 0; k < NP; k++) { su = s2_dot2(pU[c + k], a.hC[k], su); sv = s2_dot2(pV[c + k], a.hC[k], sv); }
                }
                iU[c] = clip_u8_shr(su, 14); iV[c] = clip_u8_shr(sv, 14);
            }
            // ---- colour stage + store ---------------------------------------------------------------------------
            unsigned c0[4], c1[4], c2[4];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int2 tv = lutV[iV[c]], tu = lutU[iU[c]];
                const int tr = BGR ? tu.y : tv.x, tg = tv.y + tu.x, tb = BGR ? tv.x : tu.y;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int q = 2 * c + h;
                    c0[q] = (unsigned)min(max(tr + m24(Y[q], a.y2r.cy), 0), 0xFFFFFF);
                    c1[q] = (unsigned)min(max(tg + m24(Y[q], a.y2r.cy), 0), 0xFFFFFF);
                    c2[q] = (unsigned)min(max(tb + m24(Y[q], a.y2r.cy), 0), 0xFFFFFF);
                }
            }
            if (active) {
                uint8_t *d = pd + (unsigned)((unsigned)yo * (unsigned)a.ds + dstOff);
    #define S2_B2PAIR(lo, hi) __builtin_amdgcn_perm((hi), (lo), 0x0C0C0602u)
                if (BPP == 4) {
                    uint4 o4;
                    o4.x = S2_B2PAIR(c0[0], c1[0]) | (S2_B2PAIR(c2[0], 0u) << 16) | 0xFF000000u;
                    o4.y = S2_B2PAIR(c0[1], c1[1]) | (S2_B2PAIR(c2[1], 0u) << 16) | 0xFF000000u;
                    o4.z = S2_B2PAIR(c0[2], c1[2]) | (S2_B2PAIR(c2[2], 0u) << 16) | 0xFF000000u;
                    o4.w = S2_B2PAIR(c0[3], c1[3]) | (S2_B2PAIR(c2[3], 0u) << 16) | 0xFF000000u;
                    *reinterpret_cast<uint4 *>(d) = o4;
                } else {
                    uint3 o3;           // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
                    o3.x = S2_B2PAIR(c0[0], c1[0]) | (S2_B2PAIR(c2[0], c0[1]) << 16);
                    o3.y = S2_B2PAIR(c1[1], c2[1]) | (S2_B2PAIR(c0[2], c1[2]) << 16);
                    o3.z = S2_B2PAIR(c2[2], c0[3]) | (S2_B2PAIR(c1[3], c2[3]) << 16);
                    *reinterpret_cast<uint3 *>(d) = o3;
                }
    #undef S2_B2PAIR
            }
        }
    };

    auto run = [&](auto edge_c) {
        load_luma(m0, buf[0], edge_c);
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
    };
    if (edgeWave) run(std::true_type()); else run(std::false_type());
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// Is this axis "the filter row `nominal` on the window [2x - (NP - 1), 2x + NP] of an edge-replicated line" for every output x?
// NP coefficient pairs: 4 for the 8-tap filters of an exact 2:1 scale (bicubic, bilinear, ...), 6 for Lanczos-3's 12 taps.
bool filter_is_edge_replication_np(const FilterBank &fb, int srcLen, int NP, int32_t *pairs)
{
    const int W = 2 * NP, L = NP - 1;                           // window size, samples to the left of 2x
    if (NP < 1 || NP > 8 || fb.count < W || srcLen != 2 * fb.count) return false;
    // the middle row provides the nominal coefficients
    const int xm = fb.count / 2;
    int nominal[16] = {0};
    for (int j = 0; j < fb.taps; j++) {
        const int16_t c = fb.coef[(size_t)xm * fb.taps + j];
        if (!c) continue;
        const int slot = fb.pos[xm] + j - (2 * xm - L);
        if (slot < 0 || slot >= W) return false;
        nominal[slot] = c;
    }
    std::vector<int> eff(2 * W), tab(2 * W);
    for (int x = 0; x < fb.count; x++) {
        // effective coefficient per source sample, window base 2x - L - NP (room for the taps the table may hold further out)
        const int base = 2 * x - L - NP;
        std::fill(eff.begin(), eff.end(), 0); std::fill(tab.begin(), tab.end(), 0);
        for (int k = 0; k < W; k++) {
            const int s = std::min(std::max(2 * x - L + k, 0), srcLen - 1);
            if (s - base < 0 || s - base >= 2 * W) return false;
            eff[s - base] += nominal[k];
        }
        for (int j = 0; j < fb.taps; j++) {
            const int16_t c = fb.coef[(size_t)x * fb.taps + j];
            if (!c) continue;
            const int s = fb.pos[x] + j;
            if (s < 0 || s >= srcLen || s - base < 0 || s - base >= 2 * W) return false;
            tab[s - base] += c;
        }
        if (eff != tab) return false;
    }
    for (int k = 0; k < NP; k++)
        pairs[k] = (int32_t)((uint32_t)(uint16_t)nominal[2 * k] | ((uint32_t)(uint16_t)nominal[2 * k + 1] << 16));
    return true;
}
bool filter_is_edge_replication(const FilterBank &fb, int srcLen, int32_t (&pairs)[4])
{
    return filter_is_edge_replication_np(fb, srcLen, 4, pairs);
}

// The same question for an exact R:1 down-scale: is every row "the middle row on the window [R x - L, R x - L + 2 NP) of an
// edge-replicated line"?  pairs: NP int16 pairs (a window slot without a tap: 0).
bool filter_is_edge_replication_ratio(const FilterBank &fb, int srcLen, int R, int L, int NP, int32_t *pairs)
{
    const int W = 2 * NP;
    if (R < 2 || NP < 1 || NP > 8 || fb.count < 4 || srcLen != R * fb.count) return false;
    const int xm = fb.count / 2;
    int nominal[16] = {0};
    for (int j = 0; j < fb.taps; j++) {
        const int16_t c = fb.coef[(size_t)xm * fb.taps + j];
        if (!c) continue;
        const int slot = fb.pos[xm] + j - (R * xm - L);
        if (slot < 0 || slot >= W) return false;
        nominal[slot] = c;
    }
    std::vector<int> eff(3 * W), tab(3 * W);
    for (int x = 0; x < fb.count; x++) {
        const int base = R * x - L - W;
        std::fill(eff.begin(), eff.end(), 0); std::fill(tab.begin(), tab.end(), 0);
        for (int k = 0; k < W; k++) {
            const int s = std::min(std::max(R * x - L + k, 0), srcLen - 1);
            if (s - base < 0 || s - base >= 3 * W) return false;
            eff[s - base] += nominal[k];
        }
        for (int j = 0; j < fb.taps; j++) {
            const int16_t c = fb.coef[(size_t)x * fb.taps + j];
            if (!c) continue;
            const int s = fb.pos[x] + j;
            if (s < 0 || s >= srcLen || s - base < 0 || s - base >= 3 * W) return false;
            tab[s - base] += c;
        }
        if (eff != tab) return false;
    }
    for (int k = 0; k < NP; k++)
        pairs[k] = (int32_t)((uint32_t)(uint16_t)nominal[2 * k] | ((uint32_t)(uint16_t)nominal[2 * k + 1] << 16));
    return true;
}

int yuv2s_prepare(const ScalePlan &p, const YuvScaleTiling &g, Yuv2sTables &t)
{
    t = Yuv2sTables();
    const char *off = GMAT_KNOB("GMAT_SCALE_NO_STRIP");
    if (off && atoi(off)) return 0;
    if (g.fullChroma || g.yuvOut) return 0;
    if (!is_yuv420(p.srcFormat)) return 0;
    if (!(p.dstFormat == GMAT_PIX_FMT_RGB24 || p.dstFormat == GMAT_PIX_FMT_BGR24 || p.dstFormat == GMAT_PIX_FMT_RGBA ||
          p.dstFormat == GMAT_PIX_FMT_BGRA)) return 0;
    if (p.srcW != 2 * p.dstW || p.srcH != 2 * p.dstH || p.srcW % 8 || p.srcW < 32 || p.dstH < 8) return 0;
    if (p.chrSrcW != p.dstW || p.chrSrcH != p.dstH || p.chrDstW * 2 != p.dstW || p.chrDstH != p.dstH) return 0;
    // 8-tap filters on 4 coefficient pairs (the headline), else Lanczos-3's 12 taps on 6
    t.np = 0;
    for (int np : {4, 6}) {
        if (np == 6 && (p.srcW < 128 || p.dstH < 12)) break;    // the wider window: chroma rows of at least 32 samples
        if (filter_is_edge_replication_np(p.hLum, p.srcW, np, t.hL) && filter_is_edge_replication_np(p.hChr, p.chrSrcW, np, t.hC) &&
            filter_is_edge_replication_np(g.vLumEff, p.srcH, np, t.vL)) { t.np = np; break; }
    }
    if (!t.np) return 0;
#if S2_LUT512
    // the 4-pair kernel indexes its 512-entry colour tables with (sum + 8192) >> 14 + 128 and no clamp: prove the range from the
    // coefficients (bicubic: [-29, 284]); a filter with more overshoot than half of full scale stays on the tiled kernel
    if (t.np == 4) {
        int pos = 0, neg = 0;
        for (int k = 0; k < 4; k++)
            for (int h = 0; h < 2; h++) {
                const int c = (int16_t)(((uint32_t)t.hC[k] >> (16 * h)) & 0xFFFF);
                (c > 0 ? pos : neg) += c;
            }
        if (((255 * neg + 8192) >> 14) < -S2_LUT_BIAS || ((255 * pos + 8192) >> 14) > S2_LUT_N - 1 - S2_LUT_BIAS) return 0;
    }
#endif
    // vertical chroma: one tap of 4096 on row y, rounding 1 << 18 (the kernel folds it into the horizontal accumulator)
    if (g.vChrEff.taps != 1) return 0;
    for (int y = 0; y < p.dstH; y++)
        if (g.vChrEff.pos[y] != y || g.vChrEff.coef[y] != 4096 || g.chrRound[y] != (1 << 18) || g.lumRound[y] != g.lumRound[0]) return 0;
    t.lr = g.lumRound[0];
    t.ok = 1;
    return 0;
}

// wave slots of the device: CUs x 4 SIMDs x the waves a SIMD holds of a kernel with this many VGPRs (512 a lane and SIMD, allocated
// in blocks of 8) — from hipDeviceProp, not a literal (MI355X: 256 x 4 x 6 = 6144 for the walker's 74 VGPRs)
static int wave_slots(int vgprs)
{
    static thread_local int cus[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int perSimd = std::min(8, 512 / ((vgprs + 7) / 8 * 8));
    return cus[dev] * 4 * perSimd;
}

// which form a launch of nframes frames takes: the block-cooperative kernel while the launch is short — up to 17 wave-rows (256 output columns of
// one row) per wave slot: twelve 4K -> 1080p frames, every launch of 1080p -> 540p frames — the walker beyond, and for the 6-pair filters always.
// GMAT_STRIP_BLOCK = n: up to n frames instead (0 = never).  Round 4 drew the line at 3 frames from x2bench's default rotation of 64 frame pairs
// (1.2 GB: partly served by the 256 MB Infinity Cache), where the two forms tie from 4 frames on; with every frame from HBM (X2BENCH_SETS = 8, bench.py's
// regime) the block form is ahead up to 12 frames — per frame, block form / walker, 4K -> 1080p: 3 frames 4.70 / 5.13 us, 4: 4.33 / 4.63, 6: 3.98 / 4.35,
// 8: 3.83 / 4.18, 12: 3.75 / 3.88, 16: 3.74 / 3.70, 32: 3.68 / 3.52; 1080p -> 540p: 8 frames 1.31 / 1.48, 32: 0.97 / 1.06 — and never behind in the cached
// regime (profiles/r05f_blk_frames.txt, r05g_blk_frames_small.txt).  Its blocks live for one memory round trip: short launches keep every CU
// fed where the walker's 13-iteration waves leave a tail.
bool yuv2s_block_form(const Yuv2sArgs &a, int nframes)
{
    if (a.np != 4) return false;
    if (const char *e = GMAT_KNOB("GMAT_STRIP_BLOCK")) return nframes <= atoi(e);
    const long waveRows = (long)a.dstH * ((a.dstW + S2_STRIP - 1) / S2_STRIP) * nframes;
    return waveRows <= 17L * wave_slots(74);
}

static int launch_scale_yuv2s_blk(Yuv2sArgs a, hipStream_t stream, const Yuv2xFrames &fr, int nframes)
{
    // rows a block: 4 waves x RW rows.  12 (RW = 3) filters 15 row pairs per 12 rows and gives a 4K -> 1080p frame 720 blocks = 2880
    // waves; a small frame takes shorter bands so that the launch still has a few waves for every SIMD (1080p -> 540p: RW = 1,
    // 2160 waves of one row each instead of 720 of three).  GMAT_STRIP_ROWS = 4 / 8 / 12 / 16 select RW = 1 .. 4 (A/B)
    const char *segStr = GMAT_KNOB("GMAT_STRIP_ROWS");
    const int segEnv = segStr ? atoi(segStr) : 0;
    const long waveRows = (long)a.dstH * ((a.dstW + S2_STRIP - 1) / S2_STRIP) * nframes, simds = wave_slots(74) / 6;
    const int rw = segEnv == 4 ? 1 : segEnv == 8 ? 2 : segEnv == 12 ? 3 : segEnv == 16 ? 4 : waveRows <= 3 * simds ? 1 : waveRows <= 6 * simds ? 2 : 3;
    a.segRows = 4 * rw;
    a.nseg = (a.dstH + a.segRows - 1) / a.segRows;
    a.nsg = (a.dstW + S2_STRIP - 1) / S2_STRIP;                 // strips a row
    a.updown = 0;
    const int nblk = a.nseg * a.nsg;
    const dim3 grid(a.xcdRemap ? 8 * ((nblk + 7) / 8) : nblk, nframes), block(256);
#define GMAT_S2B(N_, D_) do { if (rw == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv2s_blk_kernel<N_, D_, 1>), grid, block, 0, stream, a, fr); \
                              else if (rw == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv2s_blk_kernel<N_, D_, 2>), grid, block, 0, stream, a, fr); \
                              else if (rw == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv2s_blk_kernel<N_, D_, 4>), grid, block, 0, stream, a, fr); \
                              else hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv2s_blk_kernel<N_, D_, 3>), grid, block, 0, stream, a, fr); } while (0)
    const int d = a.dstFormat == GMAT_PIX_FMT_RGB24 ? 0 : a.dstFormat == GMAT_PIX_FMT_BGR24 ? 1 : a.dstFormat == GMAT_PIX_FMT_RGBA ? 2 : 3;
    if (a.nv12) { switch (d) { case 0: GMAT_S2B(true, 0); break; case 1: GMAT_S2B(true, 1); break; case 2: GMAT_S2B(true, 2); break; default: GMAT_S2B(true, 3); } }
    else        { switch (d) { case 0: GMAT_S2B(false, 0); break; case 1: GMAT_S2B(false, 1); break; case 2: GMAT_S2B(false, 2); break; default: GMAT_S2B(false, 3); } }
#undef GMAT_S2B
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_scale_yuv2s(const Yuv2sArgs &a0, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (!frames || nframes < 1 || nframes > kYuv2xMaxFrames) return GMAT_ERR(EINVAL);
    Yuv2sArgs a = a0;
    if (yuv2s_block_form(a, nframes)) return launch_scale_yuv2s_blk(a, stream, *frames, nframes);
    // Rows per strip segment.  A segment costs 3 warm-up row pairs (horizontal filter only) on top of its rows, so long segments
    // looked right — and round 2 ran 45-row segments, one wave per wave slot for a 32-frame launch.  Round 3 measured the access
    // pattern instead (tools/ubench/hbm_rw.hip, profiles/r03d_hbm_patterns.txt): thousands of waves each walking a long column
    // are thousands of concurrent row streams, and a pure data-movement kernel of that shape tops out at 4.8-4.9 TB/s on this
    // part whatever its prefetch depth or load width, while short bands swept in raster order reach 5.5 TB/s (the vertical halo
    // re-read comes from L2).  Short segments also end the one-round tail (a launch no longer lasts as long as its slowest wave).
    // Measured on the headline, same box: 45 rows 122.0 us, 24: 121.9, 16: 117.1, 12: 116.3, 10: 114.9, 8: 117.0, 6: 119.6 per
    // 32-frame launch (profiles/r03f_rows_updown.txt) — 12 rows at most; small launches keep the old rule (3 rows for one frame:
    // a launch cannot finish faster than one segment).
    const char *segStr = GMAT_KNOB("GMAT_STRIP_ROWS");          // tuning / test override, read per launch
    const int segEnv = segStr ? atoi(segStr) : 0;
    const int nstrips = (a.dstW + S2_STRIP - 1) / S2_STRIP;
    a.nsg = (nstrips + 3) / 4;
    int seg = segEnv > 0 ? segEnv : 0;
    if (!seg) {
        const long rows = (long)a.dstH * nstrips * nframes;      // wave-rows of the launch
        const long slots = wave_slots(74);                       // the walker's 74 VGPRs: 6 waves a SIMD
        // (round 4, re-swept with non-temporal stores, same box, three alternating repeats: 32 frames a launch 12 / 10 / 9 rows 0.639-0.644 / 0.658-0.660 /
        // 0.654-0.659 of the roofline, 16 frames 0.603-0.606 / 0.616-0.617 / 0.622-0.623: profiles/r04_headline_rows.txt)
        // — for launches of several rounds of such bands; a launch of ONE round of 12-row bands keeps them (1080p -> 540p, 32 frames: 10 / 12 / 14 rows 28.1 / 27.7 / 27.3 us)
        const long natural = std::max(3L, (rows + slots - 1) / slots);
        seg = (int)(natural > 12 ? (nframes >= 24 ? 10L : 9L) : natural);
        // the 6-pair kernel: 5 warm-up row pairs per segment instead of 3 want longer segments, its 109 VGPRs (4 waves per SIMD)
        // shorter ones; measured best 6 / 12 / 16 rows at 1 / 4 / 32 frames per launch (profiles/r02f_yuv2s_lanczos_rows_sweep.txt)
        if (a.np == 6) seg = std::min(16, std::max(6, 2 * seg));
    }
    a.segRows = seg;
    a.nseg = (a.dstH + seg - 1) / seg;
    const char *ud = GMAT_KNOB("GMAT_STRIP_UPDOWN");                 // test / measurement knob: 0 = every segment walks downward
    a.updown = a.np == 4 && !(ud && !atoi(ud));
    const int nblk = a.nseg * a.nsg;
    const dim3 grid(a.xcdRemap ? 8 * ((nblk + 7) / 8) : nblk, nframes), block(256);
    const Yuv2xFrames &fr = *frames;
#define GMAT_S2(N_, D_) do { if (a.np == 6) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv2s_np_kernel<N_, D_, 6>), grid, block, 0, stream, a, fr); \
                             else           hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuv2s_kernel<N_, D_>), grid, block, 0, stream, a, fr); } while (0)
    const int d = a.dstFormat == GMAT_PIX_FMT_RGB24 ? 0 : a.dstFormat == GMAT_PIX_FMT_BGR24 ? 1 : a.dstFormat == GMAT_PIX_FMT_RGBA ? 2 : 3;
    if (a.nv12) { switch (d) { case 0: GMAT_S2(true, 0); break; case 1: GMAT_S2(true, 1); break; case 2: GMAT_S2(true, 2); break; default: GMAT_S2(true, 3); } }
    else        { switch (d) { case 0: GMAT_S2(false, 0); break; case 1: GMAT_S2(false, 1); break; case 2: GMAT_S2(false, 2); break; default: GMAT_S2(false, 3); } }
#undef GMAT_S2
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
