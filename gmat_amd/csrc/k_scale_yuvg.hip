// k_scale_yuvg.hip — ONE polyphase band walker for 8-bit YUV 4:2:0 sources at ANY scaling ratio (gfx950).
//
// libswscale's single-context semantics (swscale.c:234-520): every plane scaled separately — hScale8To15_c (swscale.c:122-136),
// the vertical filters of vscale.c / output.c, the LUT colour stage of yuv2rgb.c — bit-exact, for whatever tables initFilter
// (utils.c:367-763) produced: any ratio, any SWS algorithm, borders folded the way the table says.  Round 2 served the exact
// ratios 2:1, 3:1, 3:2, 4:1 and 1:2 with one hand-specialised walker each and everything else (4K -> 900p, 1080p -> 432p ...)
// with the tiled kernel of round 1 at 0.10-0.14 of the HBM roofline.  This kernel needs no structure in
// the ratio:
//   * a LANE owns one output COLUMN of a plane for a band of output rows.  Its horizontal taps are therefore loop-invariant: the
//     coefficient pairs sit in registers, the source window starts at the lane's own 4-byte aligned address pos[x] & ~3, and the
//     byte pairs come out of the loaded dwords with v_perm_b32 through two per-lane selector registers (the window's byte phase
//     pos[x] & 3 is the only thing that differs between lanes; the dword indices are static).  No LDS, no positions in the loop.
//   * the vertical filter runs as K running sums per lane: source rows are consumed in pairs (cvt_pk = hScale8To15_c's
//     saturation), a pair feeds each of the K open output rows with one v_dot2 whose coefficient pair is WAVE-UNIFORM.  The host
//     lays the whole vertical program out per QUAD — four luma rows and the two chroma rows beside them: two luma pairs, one chroma
//     pair, their 3 K coefficient pairs, the first output row still open and the number of rows the quad completes — so the
//     kernel's loop is static (two quads unrolled: every register of the prefetch rings has a fixed name) and only the number of
//     rows leaving after a quad (0, 1, 2 ...) is decided at run time.  Multi-row closes (up-scaling axes, e.g. the chroma of an RGB
//     destination below 2:1) are just a larger count.
//   * bands are short (raster-like order, see k_scale_yuv2s.hip's launcher) and odd bands walk upward through mirrored tables.
// Packed RGB destinations: even lanes filter the U sample of their column pair, odd lanes the V sample (the chroma plane of an
// RGB destination has half the output width), one lane exchange per output row.  4:2:0 destinations: plane jobs of the same
// walker, luma and chroma workgroups in one launch.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

// G_BPS: bytes per SOURCE sample.  This file compiles twice: as it is (8-bit sources: NV12 / YUV420P), and from k_scale_yuvg16.hip with G_BPS = 2
// (round 5: P010LE / P016LE / YUV420P10LE / YUV420P16LE — hScale16To15_c, swscale.c:93-119, in front of the same vertical program and output stages;
// those sources sat on the lines form's two passes or the tiled kernel at 0.12-0.24 of the roofline).  What changes with the sample width is the
// horizontal stage alone — which bytes a coefficient pair meets (GWin, g_hsum, GStream::setup), how many dwords of a row a lane loads (SD), what the
// row image holds (g_conv) — and the output stage's 10-bit and dithered 8-bit forms; the 16-bit build lives in namespace gmat::g16.
#ifndef G_BPS
#define G_BPS 1
#endif
// G_PROBE (measurement builds only — WRONG pixels): bit 0: half the horizontal taps, bit 1: a pair's second row is not loaded (half the read bytes),
// bit 2: no vertical taps but a sum's first.  profiles/r05o_walker_probes.txt
#ifndef G_PROBE
#define G_PROBE 0
#endif
#if G_BPS == 2
#define G_NAME(n) n##16
#else
#define G_NAME(n) n
#endif
// G_PART (the 16-bit build is TWO translation units, so that its kernels compile side by side — k_scale_yuvg16.hip and k_scale_yuvg16b.hip): 0 = everything;
// 1 = everything but the launchers of the block-cooperative RGB-source kernels (whose instances are the launchers' alone); 2 = those launchers alone
#ifndef G_PART
#define G_PART 0
#endif

namespace gmat {
#if G_BPS == 2
namespace g16 {
#endif

// ---- a plane as a raw buffer resource: lane offset in a loop-invariant VGPR, row offset in the instruction's scalar offset,
//      reads past the plane's last byte return 0 (the windows are whole dwords and may overhang the last row by up to 7 bytes)
struct GPlane {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    __amdgpu_buffer_rsrc_t r;
    v4u words;                            // the same descriptor as four dwords, for the store issued from inline assembly
    __device__ __forceinline__ GPlane(const uint8_t *p, unsigned bytes) : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p), 0, bytes, 0x00020000))
    {
        // (readfirstlane: the descriptor is an "s" operand of the inline-assembly store below — a wave-uniform value the compiler chose to compute on the
        // vector ALU, e.g. a size that is a product of kernel arguments, reaches the assembler as v[n:n+3], "invalid operand for instruction")
        const unsigned long long a = (unsigned long long)p;
        words = (v4u){(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu)),
                      (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
    }
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    __device__ __forceinline__ void ld4(unsigned lane, unsigned row, unsigned *w) const { const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, lane, row, 0); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
    __device__ __forceinline__ void ld2(unsigned lane, unsigned row, unsigned *w) const { const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, lane, row, 0); w[0] = v.x; w[1] = v.y; }
    __device__ __forceinline__ void ld1(unsigned lane, unsigned row, unsigned *w) const { w[0] = __builtin_amdgcn_raw_buffer_load_b32(r, lane, row, 0); }
    __device__ __forceinline__ void st1(unsigned d, unsigned lane, unsigned row) const
    {
        // The store is issued from inline assembly so that the compiler does not see it: with a store pending beside loads LLVM treats
        // gfx9's shared vmcnt as out of order and waits for vmcnt(0) at every use of a loaded register — which also drains the rows
        // requested three steps ahead.  Seen as loads only, its waits are counted ones; an unseen store in the queue makes a counted
        // wait stricter than needed, never laxer (loads return in order; k_scale_yuv3x1.hip).
        // s_nop 4: a scalar operand written by a VALU instruction (v_readfirstlane_b32) may be read by a memory instruction five wait states
        // later at the earliest; the compiler keeps that distance for its own instructions and does not look into this string
        asm volatile("s_nop 4\n\tbuffer_store_dword %0, %1, %2, %3 offen" :: "v"(d), "v"(lane), "s"(words), "s"(row) : "memory");
    }
#else
    // hipcc's host pass (never executed) and the CPU emulation of the test suite
    uint8_t *p; unsigned n;
    __host__ __device__ GPlane(const uint8_t *q, unsigned bytes) : p(const_cast<uint8_t *>(q)), n(bytes) {}
    __host__ __device__ unsigned dw(size_t o) const { unsigned v = 0; if (o + 4 <= n) std::memcpy(&v, p + o, 4); return v; }
    __host__ __device__ void ld4(unsigned lane, unsigned row, unsigned *w) const { for (int i = 0; i < 4; i++) w[i] = dw((size_t)row + lane + 4 * i); }
    __host__ __device__ void ld2(unsigned lane, unsigned row, unsigned *w) const { for (int i = 0; i < 2; i++) w[i] = dw((size_t)row + lane + 4 * i); }
    __host__ __device__ void ld1(unsigned lane, unsigned row, unsigned *w) const { w[0] = dw((size_t)row + lane); }
    __host__ __device__ void st1(unsigned d, unsigned lane, unsigned row) const { if ((size_t)row + lane + 4 <= n) std::memcpy(p + (size_t)row + lane, &d, 4); }
#endif
    template <int NW> __device__ __forceinline__ void ld(unsigned lane, unsigned row, unsigned (&w)[NW]) const
    {
        static_assert(NW >= 2 && NW <= 16, "window dwords");
        int i = 0;
#pragma unroll
        for (; i + 4 <= NW; i += 4) ld4(lane + 4u * i, row, w + i);
        if (NW - i >= 2) { ld2(lane + 4u * i, row, w + i); i += 2; }
        if (NW - i == 1) ld1(lane + 4u * i, row, w + i);
    }
};

__device__ __forceinline__ int g_dot2(int ab, int cd, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, ab), __builtin_bit_cast(short2v, cd), acc, true);
}
__device__ __forceinline__ unsigned g_sat_pk_u8_i16(unsigned v)
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

// window dwords of a lane: component stride 1 (a luma or planar chroma row): byte pairs (o + 2t, o + 2t + 1), t < P, o = pos & 3;
// component stride 2 (one component of NV12's UV row): (o + 4t, o + 4t + 2), o = (2 pos + comp) & 3.
// 16-bit samples (G_BPS = 2): the host re-bases every window to an 8-BYTE boundary of the row (leading zero taps: up to three samples of a plane,
// one (U, V) position of P010's interleaved row) — a coefficient pair of a plane then meets one ALIGNED dword as it was loaded: no v_perm_b32, no
// selector, and the window is read two dwords at a time (ds_read_b64).  Dword by dword at the lanes' own offsets the same windows were a 2-way bank
// conflict (lanes 1.2 dwords apart at 2.4 : 1: SQ_LDS_BANK_CONFLICT 37.7 M cycles a 32-frame launch against 0 in the 8-bit build, SQ_WAIT_INST_LDS
// 63 M against 8 M: profiles/r05k_walker16_pmc.txt).  Stride 2: the component's half of dwords 2t and 2t + 1 (one v_perm_b32 a pair).
#if G_BPS == 2
template <int P, bool S2> struct GWin { static constexpr int NW = S2 ? 2 * P : (P + 1) & ~1; };
#else
template <int P, bool S2> struct GWin { static constexpr int NW = S2 ? P + 1 : ((P - 1) >> 1) + 2; };
#endif

// one horizontally filtered sample of this lane's column: hScale8To15_c's / hScale16To15_c's sum (before its shift), from `start`
template <int P, bool S2>
__device__ __forceinline__ int g_hsum(const unsigned (&w)[GWin<P, S2>::NW], const int (&cf)[P], unsigned selE, unsigned selO, int start)
{
    int s = start;
#pragma unroll
    for (int t = 0; t < ((G_PROBE & 1) ? (P + 1) / 2 : P); t++) {
#if G_BPS == 2
        const int pr = S2 ? (int)__builtin_amdgcn_perm(w[2 * t + 1], w[2 * t], selE) : (int)w[t];
#else
        const int pr = S2 ? (int)__builtin_amdgcn_perm(w[t + 1], w[t], selE)
                          : (int)__builtin_amdgcn_perm(w[(t >> 1) + 1], w[t >> 1], (t & 1) ? selO : selE);
#endif
        s = g_dot2(pr, cf[t], s);
    }
    return s;
}

// what a row image holds of the dwords as loaded (G_BPS = 2, YuvScaleArgs' kinds): P010 (10): sample >> 6; planar 10 bit (18): as it is; 16 bits (16
// semi-planar, 17 planar): sample - 32768 — the signed operand of v_dot2 — with the sums started at 32768 * 16384 (hBias; every filter row sums to 16384,
// checked on the host).  One or two VALU instructions per LOADED dword, not per window
struct GConv { int kind, sh, bias; };
// branch-free over the (wave-uniform) kind: both samples of the dword shifted right by 6 or 0 (one v_pk_lshrrev_b16), bit 15 flipped or not
typedef unsigned short g_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned g_conv(unsigned shr, unsigned flip, unsigned v)
{
    const g_us2 s = __builtin_bit_cast(g_us2, v) >> (g_us2){(unsigned short)shr, (unsigned short)shr};
    return __builtin_bit_cast(unsigned, s) ^ flip;
}
// first byte of the window of output column `pos` (component `comp` of an interleaved row) / of the aligned dwords the lane reads for it
template <bool S2> __device__ __host__ __forceinline__ int g_win_byte(int pos, int comp) { return G_BPS == 2 ? (S2 ? 4 * pos + 2 * comp : 2 * pos) : (S2 ? 2 * pos + comp : pos); }
template <bool S2> __device__ __host__ __forceinline__ int g_win_base(int pos, int comp) { return g_win_byte<S2>(pos, comp) & (G_BPS == 2 ? ~7 : ~3); }
// coefficient pairs of a stream in a kernel instantiated for P: 16-bit samples of an interleaved row lead with at most one position where a plane leads
// with three samples, so the same taps fit one pair fewer (and every pair there costs a v_perm_b32 besides its v_dot2)
template <int P, bool S2> struct GPairs { static constexpr int N = (G_BPS == 2 && S2) ? P - 1 : P; };

// One stream of a lane: its horizontal window (selectors, coefficient pairs), the wave's share of the row loads, a ring of R
// requested row pairs with STATIC slot names and the gathered windows of the pair consumed next.
//
// Where the source bytes come from.  The first build let every lane load its own window straight from global memory (NW dwords
// per row and lane, one row pair requested ahead) in a loop whose trip counts depended on the ratio: bit-exact, 9.8 us per 4K ->
// 900p frame (profiles/r03h_generic_walker_first.txt) — the compiler could count nothing: with dynamic loops around the loads and
// the output stores in the same queue every use of a loaded register waited for vmcnt(0) or (1), and the address arithmetic of
// each load was ~8 scalar instructions.  Now
//   * a wave loads each source row's UNIQUE bytes once — one dword per lane: 256 bytes cover 64 columns up to 3.7 : 1, SD = 2 dwords
//     per lane beyond that — R row pairs ahead into ring registers;
//   * a pair's bytes pass through a wave-private LDS row image on their way to the lanes' overlapping windows (ds_write_b32, then
//     NW dword reads per row at the lane's window offset).  No barrier: one wave, and the LDS executes its instructions in order;
//   * rows past the plane are simply requested: the buffer resource returns 0 for them, and their taps are 0 in every table.
// LK (16-bit build, round 5): what a lane loads.  0: SD dwords of the plane's row, 256 bytes apart.  1 / 2 / 3: a PACKED RGB24 / BGR24 source — the image
// holds what libswscale's input stage hands to hScale16To15_c (sh = 13): rgb24ToY_c's 14-bit luma (1: four pixels = three raw dwords -> two image
// dwords), rgb24ToUV_half_c's chroma of pixel PAIRS as one plane's line (2: eight pixels = six raw dwords -> two image dwords) or as (U, V) dwords like
// P010's interleaved row (3: four pixels -> two image dwords) — input.c:815-866; a lane then fills SD ADJACENT image dwords from 4 SDL contiguous bytes.
template <int P, bool S2, int SD, int R, int LK = 0>
struct GStream {
    static constexpr int NW = GWin<P, S2>::NW;
    static constexpr int IMG = 64 * SD;
    static constexpr int SDL = LK == 0 ? SD : LK == 2 ? 3 * SD : (3 * SD) / 2;      // raw dwords of a row a lane
    int cf[P];
    unsigned selE, selO;
    int winDw;                           // LDS dword index of this lane's window inside a row image
    int ldDw[SD];                        // LDS dword index this lane fills, per sub-load (LK != 0: ldDw[0] + s)
    unsigned ldOff[SDL];                 // byte offset in the source row this lane loads, per sub-load
    unsigned ring[R][2][SDL];
    int rc01, rc2, rk01, rk2;            // LK != 0: the converter's coefficients — bytes (0, 1) as an int16 pair and byte 2: luma or the plane's component (rc), V beside U (rk, LK = 3)
    unsigned win[2][NW];
    unsigned *img;                       // this wave's two row images of this stream: img[row * IMG + dword]
    unsigned reqOff, rowStep;            // (wave-uniform) byte offset of the next row to request, and of one row further along the walk
#if G_BPS == 2
    unsigned cvShr, cvFlip;              // 16-bit samples: what the image holds (see g_conv) ...
    int cvSh, cvBias;                    // ... hScale16To15_c's shift and the sums' start (8 bit: >> 7 from 0)
    __device__ __forceinline__ void set_conv(const GConv &c) { cvShr = c.kind == 10 ? 6u : 0u; cvFlip = (c.kind == 10 || c.kind == 18) ? 0u : 0x80008000u; cvSh = c.sh; cvBias = c.bias; }
#else
    __device__ __forceinline__ void set_conv(const GConv &) {}
#endif

    // col: this lane's output column of the plane; comp: component of an interleaved row (S2); returns the window's first dword (bytes)
    __device__ __forceinline__ int setup(const int32_t *hTab, const int32_t *posTab, int col, int comp)
    {
        const int pos = posTab[col];
        const int b0 = g_win_byte<S2>(pos, comp);
        const unsigned o = (unsigned)b0 & 3u;
#if G_BPS == 2
        selE = comp ? 0x07060302u : 0x05040100u;           // (stride 2 only)
        selO = 0u;
        (void)o;
#else
        selE = S2 ? (0x0C000C00u | o | ((o + 2) << 16)) : (0x0C000C00u | o | ((o + 1) << 16));
        selO = 0x0C000C00u | (o + 2) | ((o + 3) << 16);
#endif
#pragma unroll
        for (int t = 0; t < P; t++) cf[t] = hTab[(size_t)col * P + t];
        return g_win_base<S2>(pos, comp);
    }
    // the walk starts at row pair `pair` (walking coordinates) of a plane of `rows` rows
    __device__ __forceinline__ void start(int pair, int rows, int stride, int up)
    {
        rowStep = up ? 0u - (unsigned)stride : (unsigned)stride;
        reqOff = (unsigned)(up ? rows - 1 - 2 * pair : 2 * pair) * (unsigned)stride;
    }
    template <class Ld> __device__ __forceinline__ void request(Ld &&ld, unsigned (&dst)[2][SDL])
    {
#pragma unroll
        for (int s = 0; s < SDL; s++) { dst[0][s] = ld(ldOff[s], reqOff); dst[1][s] = (G_PROBE & 2) ? dst[0][s] : ld(ldOff[s], reqOff + rowStep); }
        reqOff += 2u * rowStep;
    }
    // four packed pixels (three dwords): bytes (0, 1) of each as an int16 pair, byte 2 on its own (k_scale_rgb2s.hip's unpacking)
    static __device__ __forceinline__ void rgb4(const unsigned *d, int (&fs)[4], int (&th)[4])
    {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int o = 3 * i, dw = o >> 2, b = o & 3;
            const unsigned lo = d[dw], hi = d[dw + 1 < 3 ? dw + 1 : dw];
            fs[i] = (int)__builtin_amdgcn_perm(hi, lo, 0x0C000C00u | (unsigned)b | ((unsigned)(b + 1) << 16));
            th[i] = (int)__builtin_amdgcn_perm(hi, lo, 0x0C0C0C00u | (unsigned)(b + 2));
        }
    }
    // a row pair's bytes: registers -> row images -> this lane's windows
    __device__ __forceinline__ void gather(const unsigned (&src)[2][SDL])
    {
        __builtin_amdgcn_wave_barrier();         // (emulation: the lanes of a wave are fibers; on the GPU the LDS runs a wave's instructions in order)
        if constexpr (LK == 0) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
#if G_BPS == 2
                for (int s = 0; s < SD; s++) img[r * IMG + ldDw[s]] = g_conv(cvShr, cvFlip, src[r][s]);
#else
                for (int s = 0; s < SD; s++) img[r * IMG + ldDw[s]] = src[r][s];
#endif
        } else {
            constexpr int KY = (32 << 14) + (1 << 8), KC = (256 << 15) + (1 << 9);      // rgb24ToY_c: >> 9; rgb24ToUV_half_c on a pair's sums: >> 10
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int g = 0; g < SD / 2; g++) {
                    unsigned o0, o1;
                    if constexpr (LK == 1) {
                        int fs[4], th[4], y[4];
                        rgb4(&src[r][3 * g], fs, th);
#pragma unroll
                        for (int i = 0; i < 4; i++) y[i] = g_dot2(fs[i], rc01, m24(th[i], rc2) + KY) >> 9;
                        o0 = (unsigned)y[0] | ((unsigned)y[1] << 16); o1 = (unsigned)y[2] | ((unsigned)y[3] << 16);
                    } else if constexpr (LK == 2) {
                        int c[4];
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            int fs[4], th[4];
                            rgb4(&src[r][6 * g + 3 * h], fs, th);
#pragma unroll
                            for (int q = 0; q < 2; q++)          // (two 9-bit sums in the halves: no carry across)
                                c[2 * h + q] = g_dot2(fs[2 * q] + fs[2 * q + 1], rc01, m24(th[2 * q] + th[2 * q + 1], rc2) + KC) >> 10;
                        }
                        o0 = (unsigned)c[0] | ((unsigned)c[1] << 16); o1 = (unsigned)c[2] | ((unsigned)c[3] << 16);
                    } else {
                        int fs[4], th[4];
                        rgb4(&src[r][3 * g], fs, th);
                        const int f0 = fs[0] + fs[1], t0 = th[0] + th[1], f1 = fs[2] + fs[3], t1 = th[2] + th[3];
                        o0 = (unsigned)(g_dot2(f0, rc01, m24(t0, rc2) + KC) >> 10) | ((unsigned)(g_dot2(f0, rk01, m24(t0, rk2) + KC) >> 10) << 16);
                        o1 = (unsigned)(g_dot2(f1, rc01, m24(t1, rc2) + KC) >> 10) | ((unsigned)(g_dot2(f1, rk01, m24(t1, rk2) + KC) >> 10) << 16);
                    }
                    *reinterpret_cast<uint2 *>(img + r * IMG + ldDw[0] + 2 * g) = make_uint2(o0, o1);
                }
        }
        __builtin_amdgcn_wave_barrier();
#if G_BPS == 2
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < NW; i += 2) {                // (winDw is even, the images 8-byte aligned: ds_read_b64)
                const uint2 t = *reinterpret_cast<const uint2 *>(img + r * IMG + winDw + i);
                win[r][i] = t.x; win[r][i + 1] = t.y;
            }
#else
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < NW; i++) win[r][i] = img[r * IMG + winDw + i];
#endif
    }
    // pair 0 gathered, pairs 1 .. R requested: pair k + 1 in ring slot (k + 1) % R
    template <class Ld> __device__ __forceinline__ void prime(Ld &&ld)
    {
        unsigned first[2][SDL];
        request(ld, first);
#pragma unroll
        for (int d = 1; d <= R; d++) request(ld, ring[d % R]);
        gather(first);
    }
    // the gathered pair's two horizontally filtered samples of this lane's column, packed: min(sum >> 7, 32767) each (hScale8To15_c)
    __device__ __forceinline__ int hpair() const
    {
#if G_BPS == 2
        const int h0 = g_hsum<P, S2>(win[0], cf, selE, selO, cvBias) >> cvSh, h1 = g_hsum<P, S2>(win[1], cf, selE, selO, cvBias) >> cvSh;
#else
        const int h0 = g_hsum<P, S2>(win[0], cf, selE, selO, 0) >> 7, h1 = g_hsum<P, S2>(win[1], cf, selE, selO, 0) >> 7;
#endif
        return __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(h0, h1));
    }
    // after a pair has been consumed: the next one (in ring slot S) becomes the gathered one, its slot is requested again R pairs on
    template <int S, class Ld> __device__ __forceinline__ void advance(Ld &&ld)
    {
        gather(ring[S]);
        request(ld, ring[S]);
    }
};

// occupancy hint.  G_WAVES = -1 (default): the instances with one dword of a row per lane (P <= 6) are compiled for 5 waves a SIMD —
// same-box A/B (profiles/r03u_waves5.txt): 4K -> 900p rgb24 0.372 -> 0.389, -> 768p 0.399 -> 0.408, 1080p -> 480p 0.295 -> 0.316, the
// others within 1 %; the 2-12 dwords this spills sit outside the row loop.  Forcing 5 waves on the two-dword instances (Lanczos)
// spills inside it (32 us per frame): they stay as the compiler allocates them.  G_WAVES = n > 0: n waves for every instance (A/B).
#ifndef G_WAVES
#define G_WAVES -1
#endif
#if !defined(__HIP__)
#define G_WAVES_ATTR                                          // (the CPU emulation of HIP compiles this file as plain C++)
#elif G_WAVES > 0
#define G_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(G_WAVES)))
#elif G_WAVES < 0
// one dword of a row per lane (P <= 6: ratios up to 3.7:1 with 4-tap algorithms): 94-100 VGPRs as compiled, 5 waves a SIMD fit in 96
// (16-bit samples: wider windows and rings — as the compiler allocates them: 98 VGPRs in the plane jobs at P = 6, 148 under an RGB destination; G_WAVES16 = n: A/B)
#ifndef G_WAVES16
#define G_WAVES16 1
#endif
#define G_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(G_BPS == 1 ? (P <= 6 && K <= 9 ? 5 : 1) : (P <= 6 && K <= 9 ? G_WAVES16 : 1))))
#else
#define G_WAVES_ATTR
#endif
// G_DEFER: the rows a quad completes leave during the NEXT quad, between a gather and its consumer (A/B switch)
#ifndef G_DEFER
#define G_DEFER 0
#endif
// G_RL: luma ring slots (row pairs requested ahead): 4 = two quads ahead, 2 = one quad ahead and 4 registers fewer
#ifndef G_RL
#define G_RL 4
#endif
// G_RP: ring slots of a plane job (row pairs requested ahead): 4 = two quads ahead, 2 = one quad ahead and 4 SD registers fewer (A/B)
// (16-bit samples: 2 — one quad ahead is as fast or 1-2 % faster, 8 % on an up-scale, and eight registers fewer: profiles/r05n_w16_ring.txt)
#ifndef G_RP
#define G_RP (G_BPS == 2 ? 2 : 4)
#endif
constexpr int kGHead = 2;               // dwords in front of a quad's coefficient pairs: first open output row, rows it completes

// ---- packed RGB destinations --------------------------------------------------------------------------------------------------
// block = 4 waves = 4 adjacent strips of 64 output columns of one band; grid.y = frame
template <int P, int K, bool NV12>
__global__ __launch_bounds__(256) G_WAVES_ATTR void scale_yuvg_rgb_kernel(YuvGArgs a, Yuv2xFrames fr)
{
    constexpr int SD = G_BPS == 2 ? (P >= 10 ? 4 : 2) : (P >= 8 ? 2 : 1), QS = kGHead + 3 * K;
    __shared__ int2 lutV[256], lutU[256];
    __shared__ __attribute__((aligned(16))) unsigned image[4][2][2 * 64 * SD];                 // [wave][luma | chroma][two row images]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const Yuv2RgbConsts &k = a.y2r;
        const int i = tid;
        lutV[i] = make_int2(k.base + m24(k.offR + (m24(i, k.crv) >> 16), k.cy), m24(m24(i, k.cgv) >> 16, k.cy));
        lutU[i] = make_int2(k.base + m24(k.offG + (m24(i, k.cgu) >> 16), k.cy), k.base + m24(k.offB + (m24(i, k.cbu) >> 16), k.cy));
    }
    __syncthreads();
    int lin = blockIdx.x;
    if (a.xcdRemap) { const int chunk = (a.nblk + 7) >> 3; lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3); }
    if (lin >= a.nblk) return;
    const int band = __builtin_amdgcn_readfirstlane(lin / a.nsg);
    const int X0 = ((lin - band * a.nsg) * 4 + wave) * 64;
    if (X0 >= a.dstW) return;
    const int up = a.updown & band & 1;
    // bandStep: 16.16 rows a band — bands whose heights differ by at most one row when the launcher balances them (no division here), else
    // bandRows << 16
    const int y0 = (int)(((unsigned)band * (unsigned)a.bandStep) >> 16), y1 = band + 1 >= a.nbands ? a.dstH : (int)(((unsigned)(band + 1) * (unsigned)a.bandStep) >> 16);
    const int ya = up ? a.dstH - y1 : y0, yb = up ? a.dstH - y0 : y1;            // walking coordinates
    const int f = blockIdx.y;
    // exact valid bytes of each plane (row bytes are multiples of 4 by the host rule): a dword past them reads as 0
    const unsigned crb = (unsigned)(NV12 ? 2 * a.chrSrcW : a.chrSrcW) * G_BPS;
    const GPlane bY(fr.y[f], (unsigned)a.ys * (unsigned)(a.srcH - 1) + (unsigned)a.srcW * G_BPS);
    const GPlane bU(fr.u[f], (unsigned)a.us * (unsigned)(a.chrSrcH - 1) + crb), bV(NV12 ? fr.u[f] : fr.v[f], (unsigned)a.us * (unsigned)(a.chrSrcH - 1) + crb);
    const int bpp = (a.dstFormat == GMAT_PIX_FMT_RGBA || a.dstFormat == GMAT_PIX_FMT_BGRA) ? 4 : 3;
    const bool bgr = a.dstFormat == GMAT_PIX_FMT_BGR24 || a.dstFormat == GMAT_PIX_FMT_BGRA;
    const GPlane bD(fr.dst[f], (unsigned)a.ds * (unsigned)(a.dstH - 1) + (unsigned)(a.dstW * bpp));

    const int x = X0 + lane, xc = min(x, a.dstW - 1), par = lane & 1;
    GStream<P, false, SD, G_RL> L;
    GStream<GPairs<P, NV12>::N, NV12, SD, 2> C;
    L.set_conv(GConv{a.src16, a.hShift, a.hBias}); C.set_conv(GConv{a.src16, a.hShift, a.hBias});
    {
        // luma: the wave's row segment starts at lane 0's window; every lane fills dword `lane` (+ 64) of the row image
        const int w0 = L.setup(a.hL, a.posL, xc, 0);
        const int seg = __builtin_amdgcn_readfirstlane(w0);
        L.winDw = (w0 - seg) >> 2;
#pragma unroll
        for (int s = 0; s < SD; s++) { L.ldDw[s] = lane + 64 * s; L.ldOff[s] = (unsigned)seg + 4u * (unsigned)(lane + 64 * s); }
        L.img = image[wave][0];
    }
    {
        // chroma column lane >> 1, component lane & 1.  NV12: one interleaved segment.  Planar: the U segment in the first half of
        // the row image (filled by lanes 0 .. 31), the V segment in the second (lanes 32 .. 63); a lane READS the half of its component
        const int w0 = C.setup(a.hC, a.posC, min(xc >> 1, a.chrDstW - 1), par);
        const int seg = __builtin_amdgcn_readfirstlane(w0);
        if (NV12) {
            C.winDw = (w0 - seg) >> 2;
#pragma unroll
            for (int s = 0; s < SD; s++) { C.ldDw[s] = lane + 64 * s; C.ldOff[s] = (unsigned)seg + 4u * (unsigned)(lane + 64 * s); }
        } else {
            C.winDw = par * 32 * SD + ((w0 - seg) >> 2);
#pragma unroll
            for (int s = 0; s < SD; s++) { const int j = (lane & 31) + 32 * s; C.ldDw[s] = (lane >> 5) * 32 * SD + j; C.ldOff[s] = (unsigned)seg + 4u * (unsigned)j; }
        }
        C.img = image[wave][1];
    }
    auto ldL = [&](unsigned off, unsigned row) { unsigned v; bY.ld1(off, row, &v); return v; };
    auto ldC = [&](unsigned off, unsigned row) {
        unsigned v;
        if (NV12 || lane < 32) bU.ld1(off, row, &v); else bV.ld1(off, row, &v);      // only the load diverges (planar: us == vs, host rule)
        return v;
    };
    // the vertical program: quads q0 .. q1 complete the band's rows; the sums start at the first row still open before q0
    const int32_t *prog = a.prog[up];
    const int q0 = uniform_load(a.qfirst[up], ya), q1 = uniform_load(a.qdone[up], yb - 1);
    int y = uniform_load(prog, q0 * QS);
    L.start(2 * q0, a.srcH, a.ys, up);
    C.start(q0, a.chrSrcH, a.us, up);
    L.prime(ldL);
    C.prime(ldC);
    int accL[K], accC[K];
#pragma unroll
    for (int i = 0; i < K; i++) { accL[i] = a.roundL; accC[i] = a.roundC; }
    // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3 from (own pixel = bytes 0..3, next lane's = bytes 4..7)
    const unsigned dsel = (unsigned)((lane & 3) == 0 ? 0x04020100u : (lane & 3) == 1 ? 0x05040201u : 0x06050402u);

    // lane i of every group of four fetches lane quad_perm[i] of it: one VALU instruction, no LDS round trip
#define GMAT_G_QUAD(v, ctrl) __builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xF, 0xF, true)
    auto emit = [&](int yw) {
        const int Y = accL[0] >> 19;
        const int mine = clip_u8_shr(accC[0], 19), other = GMAT_G_QUAD(mine, 0xB1);          // quad_perm:[1,0,3,2]: the pair's other component
        const int U = par ? other : mine, V = par ? mine : other;
        const int2 tv = lutV[V], tu = lutU[U];
        const int ycy = m24(Y, a.y2r.cy);
        const unsigned cr = (unsigned)((bgr ? tu.y : tv.x) + ycy), cg = (unsigned)(tv.y + tu.x + ycy), cb = (unsigned)((bgr ? tv.x : tu.y) + ycy);
        const unsigned rg = g_sat_pk_u8_i16(__builtin_amdgcn_perm(cg, cr, 0x07060302u));
        const unsigned ba = g_sat_pk_u8_i16(__builtin_amdgcn_perm(0x00FF0000u, cb, 0x07060302u));
        const unsigned px = __builtin_amdgcn_perm(ba, rg, 0x05040100u);
        const unsigned drow = (unsigned)(up ? a.dstH - 1 - yw : yw) * (unsigned)a.ds;
        if (bpp == 4) {
            if (x < a.dstW) bD.st1(px, 4u * (unsigned)x, drow);
        } else {
            // four pixels = three dwords: lanes 0, 1, 2 of each group of four store one, from their own pixel and the next lane's
            // (quad_perm:[1,2,3,3]) through a byte permute whose selector is the lane's place in the group
            const unsigned nxt = (unsigned)GMAT_G_QUAD(px, 0xF9);
            const unsigned o = __builtin_amdgcn_perm(nxt, px, dsel);
            const int nb = 3 * min(64, a.dstW - X0);                 // bytes of this strip's row
            const int ob = 12 * (lane >> 2) + 4 * (lane & 3);        // byte offset of this lane's dword in the strip's row
            if ((lane & 3) != 3) {
                if (ob + 4 <= nb) bD.st1(o, 3u * (unsigned)X0 + (unsigned)ob, drow);
                else if (ob < nb) {                                   // a width that is not a multiple of 4: the last bytes one by one
                    uint8_t *d = fr.dst[f] + (size_t)drow + 3u * (unsigned)X0 + (unsigned)ob;
                    for (int i = 0; i < nb - ob; i++) d[i] = (uint8_t)(o >> (8 * i));
                }
            }
        }
    };
#undef GMAT_G_QUAD
    // one quad: luma pairs 2q and 2q + 1, chroma pair q.  SA / SB / SC: the ring slots holding the pairs that follow (static: quads
    // alternate between two sets of slots).  Order matters — the wave barriers around a gather pin it in place: a gather's LDS round
    // trip is covered by work that does not need it (the other stream's horizontal filter, the output stage of the rows the PREVIOUS
    // quad completed: their sums are final, and they must leave before this quad's taps are added because the program's slots are
    // counted from the first row still open).
    int pend = 0;                               // rows completed by the previous quad, not yet out
    auto flush = [&]() {
        for (int k = 0; k < pend; k++, y++) {
            if (y >= ya && y < yb) emit(y);
#pragma unroll
            for (int i = 0; i + 1 < K; i++) { accL[i] = accL[i + 1]; accC[i] = accC[i + 1]; }
            accL[K - 1] = a.roundL; accC[K - 1] = a.roundC;
        }
    };
    auto quad = [&](int q, auto sa_c, auto sb_c, auto sc_c) {
        constexpr int SA = decltype(sa_c)::value, SB = decltype(sb_c)::value, SC = decltype(sc_c)::value;
        const int32_t *pq = prog + (size_t)q * QS;
        int cL0[K], cL1[K], cC[K];
#pragma unroll
        for (int i = 0; i < K; i++) { cL0[i] = uniform_load(pq, kGHead + i); cL1[i] = uniform_load(pq, kGHead + K + i); cC[i] = uniform_load(pq, kGHead + 2 * K + i); }
        const int ne = uniform_load(pq, 1);
        const int hp0 = L.hpair();
        L.template advance<SA>(ldL);
        const int hpc = C.hpair();
        C.template advance<SC>(ldC);
#if G_DEFER
        flush();
#endif
#pragma unroll
        for (int i = 0; i < K; i++) { accL[i] = g_dot2(hp0, cL0[i], accL[i]); accC[i] = g_dot2(hpc, cC[i], accC[i]); }
        const int hp1 = L.hpair();
        L.template advance<SB>(ldL);
#pragma unroll
        for (int i = 0; i < K; i++) accL[i] = g_dot2(hp1, cL1[i], accL[i]);
        pend = ne;
#if !G_DEFER
        flush();
#endif
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    for (int q = q0; q <= q1; q += 2) {
#if G_RL == 4
        quad(q, I1(), I2(), I1());
        if (q + 1 > q1) break;
        quad(q + 1, I3(), I0(), I0());
#else
        quad(q, I1(), I0(), I1());
        if (q + 1 > q1) break;
        quad(q + 1, I1(), I0(), I0());
#endif
    }
    flush();
}

// which bytes of its row a lane loads and which image dwords it fills: a plane's row (LK = 0: dword `lane` of the wave's segment and the ones 256
// bytes on) or packed RGB pixels (LK != 0: SD adjacent image dwords from 4 SDL contiguous bytes; seg = the segment's first image BYTE, a multiple of 8).
// job / semi pick the converter's coefficients: luma, one chroma plane's component, or U beside V
template <class Stream>
__device__ __forceinline__ void g_stream_loads(Stream &W, int seg, int lane, bool s2, const YuvGArgs &a, int job)
{
    constexpr int SD = Stream::IMG / 64, SDL = Stream::SDL;
    if constexpr (SDL == SD) {
#pragma unroll
        for (int s = 0; s < SD; s++) { W.ldDw[s] = lane + 64 * s; W.ldOff[s] = (unsigned)seg + 4u * (unsigned)(lane + 64 * s); }
    } else {
        const int e0 = s2 ? seg >> 2 : seg >> 1;                      // first sample (chroma position) of the segment
        const unsigned raw = (unsigned)((2 * SDL == 3 * SD && !s2) ? 3 * e0 : 6 * e0);      // luma: 3 bytes a sample; chroma of pixel pairs: 6
#pragma unroll
        for (int s = 0; s < SD; s++) W.ldDw[s] = lane * SD + s;
#pragma unroll
        for (int s = 0; s < SDL; s++) W.ldOff[s] = raw + 4u * (unsigned)(lane * SDL + s);
        const Rgb2YuvConsts &k = a.r2y;
        auto pk = [](int lo, int hi) { return (int)(((unsigned)lo & 0xFFFFu) | ((unsigned)hi << 16)); };
        const int c0 = job == 0 ? k.ry : (job == 2 ? k.rv : k.ru), c1 = job == 0 ? k.gy : (job == 2 ? k.gv : k.gu), c2 = job == 0 ? k.by : (job == 2 ? k.bv : k.bu);
        W.rc01 = a.rgbBgr ? pk(c2, c1) : pk(c0, c1); W.rc2 = a.rgbBgr ? c0 : c2;
        W.rk01 = a.rgbBgr ? pk(k.bv, k.gv) : pk(k.rv, k.gv); W.rk2 = a.rgbBgr ? k.rv : k.bv;
    }
}

// The output stage of a plane job: the lane's vertical sum -> its sample of the plane row -> the wave's dwords.  elem: the lane's element of the row
// (a luma / planar chroma sample, or a component of the interleaved row), e0 the wave's first, n the row's elements; dx: the lane's half of the
// ordered dither (8-bit output of a deeper source: dither_8x8_128 is affine over GF(2), px_math.h), y the REAL output row
struct GPlaneOut {
    int dst16, dstShift, dither8, xpart;
    __device__ __forceinline__ void set(const YuvGArgs &a, int col) { dst16 = a.dst16; dstShift = a.dstShift; dither8 = a.dither8; xpart = dither_8x8_128(col, 0); }
    __device__ __forceinline__ void store(const GPlane &bD, uint8_t *dp, int acc, int e0, int lane, int n, unsigned drow, int y) const
    {
        const int nb = min(64, n - e0);
        if (__builtin_amdgcn_readfirstlane(dst16)) {
            // yuv2p010l1_c / lX_c / cX_c, yuv2planeX_10_c (output.c:459-519): clip_uintp2((1 << 16 + sum) >> 17, 10), P010: << 6 — two lanes a dword
            const unsigned v = (unsigned)min(max(acc >> 17, 0), 1023) << dstShift;
            const unsigned o = v | ((unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xF5, 0xF, 0xF, true) << 16);                 // quad_perm:[1,1,3,3]
            if ((lane & 1) == 0) {
                if (lane + 2 <= nb) bD.st1(o, 2u * (unsigned)(e0 + lane), drow);
                else if (lane < nb) *reinterpret_cast<unsigned short *>(dp + (size_t)drow + 2u * (unsigned)(e0 + lane)) = (unsigned short)v;
            }
            return;
        }
        // yuv2planeX_8_c / yuv2nv12cX_c: clip_u8((dither << 12 + sum) >> 19); the sums started at 64 << 12, a deeper source's ordered dither on top
        if (G_BPS == 2 && __builtin_amdgcn_readfirstlane(dither8)) acc += ((xpart ^ dither_8x8_128(0, y) ^ 36) - 64) << 12;
        const unsigned v = (unsigned)clip_u8_shr(acc, 19);
        // four lanes' bytes -> one dword in lane 0 of each group of four: two quad permutes and two shift-ors, no LDS round trip
        const unsigned pr = v | ((unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xF5, 0xF, 0xF, true) << 8);          // quad_perm:[1,1,3,3]
        const unsigned o = pr | ((unsigned)__builtin_amdgcn_update_dpp(0, (int)pr, 0xAA, 0xF, 0xF, true) << 16);       // quad_perm:[2,2,2,2]
        if ((lane & 3) == 0) {
            if (lane + 4 <= nb) bD.st1(o, (unsigned)e0 + (unsigned)lane, drow);
            else if (lane < nb) {
                uint8_t *d = dp + (size_t)drow + (unsigned)e0 + (unsigned)lane;
                for (int i = 0; i < nb - lane; i++) d[i] = (uint8_t)(o >> (8 * i));
            }
        }
    }
};

// ---- 4:2:0 destinations: plane jobs ----------------------------------------------------------------------------------------------
// job 0: the luma plane (lane = column).  job 1: chroma — NV12 -> NV12: lane = (column, component) of the interleaved plane;
// planar -> planar: two jobs (U, V), lane = column.  Blocks [0, nblkL) are luma, the rest chroma.  A quad of a plane job is four of
// ITS rows (two row pairs; the third coefficient set of the program is unused).
// SRC = 1 (16-bit build): the source is ONE plane of packed RGB24 / BGR24 pixels (libswscale's generic path for an RGB source into a 4:2:0 frame,
// down-scaling: luma from every pixel, chroma from horizontal pixel PAIRS at full height — chrSrcHSubSample = 1, chrSrcVSubSample = 0, utils.c:1529-1545);
// every job reads that plane through its own converter (GStream's LK); NV12 then names the DESTINATION's chroma layout
template <int P, int K, bool NV12, int SRC = 0>
__global__ __launch_bounds__(256) G_WAVES_ATTR void scale_yuvg_planes_kernel(YuvGArgs a, Yuv2xFrames fr)
{
    constexpr int SD = G_BPS == 2 ? (P >= 10 ? 4 : 2) : (P >= 8 ? 2 : 1), QS = kGHead + 3 * K;
    __shared__ __attribute__((aligned(16))) unsigned image[4][2 * 64 * SD];                    // [wave][two row images]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lin = blockIdx.x;
    if (a.xcdRemap) { const int chunk = (a.nblk + 7) >> 3; lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3); }
    if (lin >= a.nblk) return;
    const int f = blockIdx.y;
    // which job
    int job = 0, rel = lin;
    if (lin >= a.nblkL) { rel = lin - a.nblkL; job = 1; if (!NV12 && rel >= a.nblkC) { rel -= a.nblkC; job = 2; } }
    const int nsg = job ? a.nsgC : a.nsg;
    const int band = __builtin_amdgcn_readfirstlane(rel / nsg);
    const int B0 = ((rel - band * nsg) * 4 + wave) * 64;             // first BYTE column of this wave in the destination plane row
    const int rowBytes = job == 0 ? a.dstW : NV12 ? 2 * a.chrDstW : a.chrDstW;
    if (B0 >= rowBytes) return;
    const int rows = job ? a.chrDstH : a.dstH, srcRows = job ? a.chrSrcH : a.srcH;
    const int up = a.updown & band & 1;
    const int bandRows = job ? a.bandRowsC : a.bandRows;
    const unsigned bstep = job ? (unsigned)a.bandStepC : (unsigned)a.bandStep;
    const int y0 = (int)(((unsigned)band * bstep) >> 16), y1 = band + 1 >= (job ? a.nbandsC : a.nbands) ? rows : (int)(((unsigned)(band + 1) * bstep) >> 16);
    (void)bandRows;
    const int ya = up ? rows - y1 : y0, yb = up ? rows - y0 : y1;
    const uint8_t *sp = (SRC || job == 0) ? fr.y[f] : job == 1 ? fr.u[f] : fr.v[f];
    uint8_t *dp = job == 0 ? fr.dst[f] : job == 1 ? fr.dstU[f] : fr.dstV[f];
    const int ss = (SRC || job == 0) ? a.ys : job == 1 ? a.us : a.vs, dstride = job == 0 ? a.ds : job == 1 ? a.dsU : a.dsV;
    const int srcRowBytes = SRC ? 3 * a.srcW : (job == 0 ? a.srcW : NV12 ? 2 * a.chrSrcW : a.chrSrcW) * G_BPS;
    const GPlane bS(sp, (unsigned)ss * (unsigned)(srcRows - 1) + (unsigned)srcRowBytes), bD(dp, (unsigned)dstride * (unsigned)(rows - 1) + (unsigned)rowBytes * (a.dst16 ? 2u : 1u));
    const int bcol = min(B0 + lane, rowBytes - 1);
    GPlaneOut out;                                                 // (the dither's column: a chroma sample's own, V three columns on — vscale.c:98,101, output.c:433-434)
    out.set(a, job == 0 ? bcol : NV12 ? (bcol >> 1) + 3 * (bcol & 1) : bcol + (job == 2 ? 3 : 0));
    const int32_t *prog = job ? a.progC[up] : a.prog[up];
    const int rnd = job ? a.roundC : a.roundL;
    auto run = [&](auto s2_c, auto lk_c) {
        constexpr bool S2 = decltype(s2_c)::value;
        constexpr int LK = decltype(lk_c)::value;
        GStream<GPairs<P, S2>::N, S2, SD, G_RP, LK> W;
        W.set_conv(GConv{a.src16, a.hShift, a.hBias});
        {
            const int w0 = W.setup(job ? a.hC : a.hL, job ? a.posC : a.posL, S2 ? bcol >> 1 : bcol, S2 ? bcol & 1 : 0);
            const int seg = __builtin_amdgcn_readfirstlane(w0);
            W.winDw = (w0 - seg) >> 2;
            g_stream_loads(W, seg, lane, S2, a, job);
            W.img = image[wave];
        }
        auto ld = [&](unsigned off, unsigned row) { unsigned v; bS.ld1(off, row, &v); return v; };
        const int q0 = uniform_load(job ? a.qfirstC[up] : a.qfirst[up], ya), q1 = uniform_load(job ? a.qdoneC[up] : a.qdone[up], yb - 1);
        int y = uniform_load(prog, q0 * QS);
        W.start(2 * q0, srcRows, ss, up);
        W.prime(ld);
        int acc[K];
#pragma unroll
        for (int i = 0; i < K; i++) acc[i] = rnd;
        auto emit = [&](int yw) {
            const int yr = up ? rows - 1 - yw : yw;
            out.store(bD, dp, acc[0], B0, lane, rowBytes, (unsigned)yr * (unsigned)dstride, yr);
        };
        int pend = 0;                           // rows completed by the previous quad, not yet out (see the RGB kernel)
        auto flush = [&]() {
            for (int k = 0; k < pend; k++, y++) {
                if (y >= ya && y < yb) emit(y);
#pragma unroll
                for (int i = 0; i + 1 < K; i++) acc[i] = acc[i + 1];
                acc[K - 1] = rnd;
            }
        };
        auto quad = [&](int q, auto sa_c, auto sb_c) {
            constexpr int SA = decltype(sa_c)::value, SB = decltype(sb_c)::value;
            const int32_t *pq = prog + (size_t)q * QS;
            int c0[K], c1[K];
#pragma unroll
            for (int i = 0; i < K; i++) { c0[i] = uniform_load(pq, kGHead + i); c1[i] = uniform_load(pq, kGHead + K + i); }
            const int ne = uniform_load(pq, 1);
            const int hp0 = W.hpair();
            W.template advance<SA>(ld);
#if G_DEFER
            flush();
#endif
#pragma unroll
            for (int i = 0; i < ((G_PROBE & 4) ? 1 : K); i++) acc[i] = g_dot2(hp0, c0[i], acc[i]);
            const int hp1 = W.hpair();
            W.template advance<SB>(ld);
#pragma unroll
            for (int i = 0; i < ((G_PROBE & 4) ? 1 : K); i++) acc[i] = g_dot2(hp1, c1[i], acc[i]);
            pend = ne;
#if !G_DEFER
            flush();
#endif
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        for (int q = q0; q <= q1; q += 2) {
#if G_RP == 4
            quad(q, I1(), I2());
            if (q + 1 > q1) break;
            quad(q + 1, I3(), I0());
#else
            quad(q, I1(), I0());
            if (q + 1 > q1) break;
            quad(q + 1, I1(), I0());
#endif
        }
        flush();
    };
    using LK0 = std::integral_constant<int, 0>;
#if G_BPS == 2
    if constexpr (SRC != 0) {
        if (job == 0) run(std::false_type(), std::integral_constant<int, 1>());
        else if (NV12) run(std::true_type(), std::integral_constant<int, 3>());
        else run(std::false_type(), std::integral_constant<int, 2>());
    } else
#endif
    if (NV12 && job == 1) run(std::true_type(), LK0()); else run(std::false_type(), LK0());
}

// ---- the block-cooperative form: a launch of ONE frame (or a few) --------------------------------------------------------------------
// What sws_scale() / filter_frame() issue is one frame, and a band walker is the wrong shape for it: too few waves to fill the chip
// unless the bands are short, and every short band pays its vertical windows' lead-in again, row after row in one wave's dependent
// chain (4K -> 720p rgb24 alone: 9.9 us a frame on the 3:1 walker against 3.5 batched).  Here a BLOCK owns 64 output columns of a
// band of rows and its four waves share the work twice:
//   1. horizontally — the band's source row pairs are dealt round the waves (pair i to wave i & 3); a wave requests ALL of its pairs
//      at once (J pairs of registers: no ring, nothing waits for anything but the data), filters each through its wave-private row
//      image exactly as the walker does (GStream::gather / hpair) and leaves the packed pair of 15-bit samples in LDS;
//   2. one barrier;
//   3. vertically — output row y0 + i belongs to wave i & 3: its taps are dot products over the LDS column of filtered pairs,
//      coefficient pairs by OUTPUT row from the host (wave-uniform: scalar loads, four pairs a group), then the walker's output stage.
// Every filtered row is computed once per band (the lead-in is still per band, but in parallel), no running sums, no program of quads.
// Same arithmetic as the walker operation by operation: hScale8To15_c's sums in tap order, the vertical sum from the same start value
// over the same taps (integer adds commute), the same colour stage.
constexpr int kGBlkJL = 8, kGBlkJC = 5;  // row pairs a wave requests at once: luma / the chroma of an RGB destination
constexpr int kGBlkPad = 16;             // LDS pair slots past the band's own: a row's last group of four may reach there (its taps are 0)
constexpr int kGBlkHead = 2;             // dwords in front of an output row's coefficient pairs: first row pair, last row pair

// all of a wave's row pairs requested, filtered and left in hs[pair slot][lane]; the band's pairs are [pa, pb], this wave's pa + wave + 4 j
template <int P, bool S2, int SD, int J, int LK, class Ld>
__device__ __forceinline__ void g_blk_hpass(GStream<P, S2, SD, J, LK> &W, Ld &&ld, int pa, int pb, int wave, int lane, unsigned stride, int (*hs)[64])
{
    // every slot is requested, the ones past the band's last pair as that pair again (a cache hit): with no load under a branch the
    // compiler counts its waits, and the first pair is filtered while the others are still on their way
#pragma unroll
    for (int j = 0; j < J; j++) {
        const unsigned off = (unsigned)(2 * min(pa + wave + 4 * j, pb)) * stride;
#pragma unroll
        for (int s = 0; s < GStream<P, S2, SD, J, LK>::SDL; s++) { W.ring[j][0][s] = ld(W.ldOff[s], off); W.ring[j][1][s] = ld(W.ldOff[s], off + stride); }
    }
#pragma unroll
    for (int j = 0; j < J; j++)
        if (pa + wave + 4 * j <= pb) {
            W.gather(W.ring[j]);
            hs[wave + 4 * j][lane] = W.hpair();
        }
}
// one output row's vertical sum of this lane's column: n4 groups of four coefficient pairs from row pair slot `base` on
__device__ __forceinline__ int g_blk_vsum(const int32_t *row, int n4, const int (*hs)[64], int base, int lane, int start)
{
    int acc = start;
    for (int g = 0; g < n4; g++) {
        const int c0 = uniform_load(row, kGBlkHead + 4 * g), c1 = uniform_load(row, kGBlkHead + 4 * g + 1);
        const int c2 = uniform_load(row, kGBlkHead + 4 * g + 2), c3 = uniform_load(row, kGBlkHead + 4 * g + 3);
        const int (*h)[64] = hs + base + 4 * g;
        acc = g_dot2(h[0][lane], c0, acc); acc = g_dot2(h[1][lane], c1, acc);
        acc = g_dot2(h[2][lane], c2, acc); acc = g_dot2(h[3][lane], c3, acc);
    }
    return acc;
}

// packed RGB destinations: block = 64 output columns x a.bandRows output rows; grid.y = frame
template <int P, bool NV12>
__global__ __launch_bounds__(256) void scale_yuvg_blk_rgb_kernel(YuvGArgs a, Yuv2xFrames fr)
{
    constexpr int SD = G_BPS == 2 ? (P >= 10 ? 4 : 2) : (P >= 8 ? 2 : 1), JL = kGBlkJL, JC = kGBlkJC;
    __shared__ int2 lutV[256], lutU[256];
    __shared__ __attribute__((aligned(16))) unsigned image[4][2][2 * 64 * SD];                 // [wave][luma | chroma][two row images]
    __shared__ int hLs[4 * JL + kGBlkPad][64], hCs[4 * JC + kGBlkPad][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const Yuv2RgbConsts &k = a.y2r;
        const int i = tid;
        lutV[i] = make_int2(k.base + m24(k.offR + (m24(i, k.crv) >> 16), k.cy), m24(m24(i, k.cgv) >> 16, k.cy));
        lutU[i] = make_int2(k.base + m24(k.offG + (m24(i, k.cgu) >> 16), k.cy), k.base + m24(k.offB + (m24(i, k.cbu) >> 16), k.cy));
    }
    int lin = blockIdx.x;
    if (a.xcdRemap) { const int chunk = (a.nblk + 7) >> 3; lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3); }
    if (lin >= a.nblk) return;
    const int band = lin / a.nsg;                                     // (a.nsg: 64-column strips per row of blocks in this form)
    const int X0 = (lin - band * a.nsg) * 64;
    const int y0 = band * a.bandRows, y1 = min(y0 + a.bandRows, a.dstH);
    const int f = blockIdx.y;
    const unsigned crb = (unsigned)(NV12 ? 2 * a.chrSrcW : a.chrSrcW) * G_BPS;
    const GPlane bY(fr.y[f], (unsigned)a.ys * (unsigned)(a.srcH - 1) + (unsigned)a.srcW * G_BPS);
    const GPlane bU(fr.u[f], (unsigned)a.us * (unsigned)(a.chrSrcH - 1) + crb), bV(NV12 ? fr.u[f] : fr.v[f], (unsigned)a.us * (unsigned)(a.chrSrcH - 1) + crb);
    const int bpp = (a.dstFormat == GMAT_PIX_FMT_RGBA || a.dstFormat == GMAT_PIX_FMT_BGRA) ? 4 : 3;
    const bool bgr = a.dstFormat == GMAT_PIX_FMT_BGR24 || a.dstFormat == GMAT_PIX_FMT_BGRA;
    const GPlane bD(fr.dst[f], (unsigned)a.ds * (unsigned)(a.dstH - 1) + (unsigned)(a.dstW * bpp));

    const int x = X0 + lane, xc = min(x, a.dstW - 1), par = lane & 1;
    GStream<P, false, SD, JL> L;
    GStream<GPairs<P, NV12>::N, NV12, SD, JC> C;
    L.set_conv(GConv{a.src16, a.hShift, a.hBias}); C.set_conv(GConv{a.src16, a.hShift, a.hBias});
    {
        // (the lane mapping of scale_yuvg_rgb_kernel; the row segment's start — lane 0's window — from a SCALAR load, so that the row
        // requests do not wait for the lanes' own table entries)
        const int w0 = L.setup(a.hL, a.posL, xc, 0);
        const int seg = g_win_base<false>(uniform_load(a.posL, X0), 0);
        L.winDw = (w0 - seg) >> 2;
#pragma unroll
        for (int s = 0; s < SD; s++) { L.ldDw[s] = lane + 64 * s; L.ldOff[s] = (unsigned)seg + 4u * (unsigned)(lane + 64 * s); }
        L.img = image[wave][0];
    }
    {
        const int w0 = C.setup(a.hC, a.posC, min(xc >> 1, a.chrDstW - 1), par);
        const int pos0 = uniform_load(a.posC, min(X0 >> 1, a.chrDstW - 1));
        const int seg = g_win_base<NV12>(pos0, 0);
        if (NV12) {
            C.winDw = (w0 - seg) >> 2;
#pragma unroll
            for (int s = 0; s < SD; s++) { C.ldDw[s] = lane + 64 * s; C.ldOff[s] = (unsigned)seg + 4u * (unsigned)(lane + 64 * s); }
        } else {
            C.winDw = par * 32 * SD + ((w0 - seg) >> 2);
#pragma unroll
            for (int s = 0; s < SD; s++) { const int j = (lane & 31) + 32 * s; C.ldDw[s] = (lane >> 5) * 32 * SD + j; C.ldOff[s] = (unsigned)seg + 4u * (unsigned)j; }
        }
        C.img = image[wave][1];
    }
    auto ldL = [&](unsigned off, unsigned row) { unsigned v; bY.ld1(off, row, &v); return v; };
    auto ldC = [&](unsigned off, unsigned row) {
        unsigned v;
        if (NV12 || lane < 32) bU.ld1(off, row, &v); else bV.ld1(off, row, &v);
        return v;
    };
    const int sL = kGBlkHead + 4 * a.n4L, sC = kGBlkHead + 4 * a.n4C;
    const int paL = uniform_load(a.vtL, y0 * sL), pbL = uniform_load(a.vtL, (y1 - 1) * sL + 1);
    const int paC = uniform_load(a.vtC, y0 * sC), pbC = uniform_load(a.vtC, (y1 - 1) * sC + 1);
    g_blk_hpass(L, ldL, paL, pbL, wave, lane, (unsigned)a.ys, hLs);
    g_blk_hpass(C, ldC, paC, pbC, wave, lane, (unsigned)a.us, hCs);
    __syncthreads();

    const unsigned dsel = (unsigned)((lane & 3) == 0 ? 0x04020100u : (lane & 3) == 1 ? 0x05040201u : 0x06050402u);
#define GMAT_G_QUAD(v, ctrl) __builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xF, 0xF, true)
    for (int y = y0 + wave; y < y1; y += 4) {
        const int32_t *rl = a.vtL + (size_t)y * sL, *rc = a.vtC + (size_t)y * sC;
        const int accL = g_blk_vsum(rl, a.n4L, hLs, uniform_load(rl, 0) - paL, lane, a.roundL);
        const int accC = g_blk_vsum(rc, a.n4C, hCs, uniform_load(rc, 0) - paC, lane, a.roundC);
        // the colour stage and the store of scale_yuvg_rgb_kernel's emit()
        const int Y = accL >> 19;
        const int mine = clip_u8_shr(accC, 19), other = GMAT_G_QUAD(mine, 0xB1);
        const int U = par ? other : mine, V = par ? mine : other;
        const int2 tv = lutV[V], tu = lutU[U];
        const int ycy = m24(Y, a.y2r.cy);
        const unsigned cr = (unsigned)((bgr ? tu.y : tv.x) + ycy), cg = (unsigned)(tv.y + tu.x + ycy), cb = (unsigned)((bgr ? tv.x : tu.y) + ycy);
        const unsigned rg = g_sat_pk_u8_i16(__builtin_amdgcn_perm(cg, cr, 0x07060302u));
        const unsigned ba = g_sat_pk_u8_i16(__builtin_amdgcn_perm(0x00FF0000u, cb, 0x07060302u));
        const unsigned px = __builtin_amdgcn_perm(ba, rg, 0x05040100u);
        const unsigned drow = (unsigned)y * (unsigned)a.ds;
        if (bpp == 4) {
            if (x < a.dstW) bD.st1(px, 4u * (unsigned)x, drow);
        } else {
            const unsigned nxt = (unsigned)GMAT_G_QUAD(px, 0xF9);
            const unsigned o = __builtin_amdgcn_perm(nxt, px, dsel);
            const int nb = 3 * min(64, a.dstW - X0);
            const int ob = 12 * (lane >> 2) + 4 * (lane & 3);
            if ((lane & 3) != 3) {
                if (ob + 4 <= nb) bD.st1(o, 3u * (unsigned)X0 + (unsigned)ob, drow);
                else if (ob < nb) {
                    uint8_t *d = fr.dst[f] + (size_t)drow + 3u * (unsigned)X0 + (unsigned)ob;
                    for (int i = 0; i < nb - ob; i++) d[i] = (uint8_t)(o >> (8 * i));
                }
            }
        }
    }
#undef GMAT_G_QUAD
}

// 4:2:0 destinations: block = 64 byte columns x a band of rows of ONE plane job (luma | interleaved chroma | U | V, as scale_yuvg_planes_kernel)
template <int P, bool NV12, int SRC = 0>
__global__ __launch_bounds__(256) void scale_yuvg_blk_planes_kernel(YuvGArgs a, Yuv2xFrames fr)
{
    constexpr int SD = G_BPS == 2 ? (P >= 10 ? 4 : 2) : (P >= 8 ? 2 : 1), J = kGBlkJL;
    __shared__ __attribute__((aligned(16))) unsigned image[4][2 * 64 * SD];
    __shared__ int hS[4 * J + kGBlkPad][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lin = blockIdx.x;
    if (a.xcdRemap) { const int chunk = (a.nblk + 7) >> 3; lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3); }
    if (lin >= a.nblk) return;
    const int f = blockIdx.y;
    int job = 0, rel = lin;
    if (lin >= a.nblkL) { rel = lin - a.nblkL; job = 1; if (!NV12 && rel >= a.nblkC) { rel -= a.nblkC; job = 2; } }
    const int nsg = job ? a.nsgC : a.nsg;
    const int band = rel / nsg;
    const int B0 = (rel - band * nsg) * 64;                         // first BYTE column of this block in the destination plane row
    const int rowBytes = job == 0 ? a.dstW : NV12 ? 2 * a.chrDstW : a.chrDstW;
    const int rows = job ? a.chrDstH : a.dstH, srcRows = job ? a.chrSrcH : a.srcH;
    const int bandRows = job ? a.bandRowsC : a.bandRows;
    const int y0 = band * bandRows, y1 = min(y0 + bandRows, rows);
    const uint8_t *sp = (SRC || job == 0) ? fr.y[f] : job == 1 ? fr.u[f] : fr.v[f];
    uint8_t *dp = job == 0 ? fr.dst[f] : job == 1 ? fr.dstU[f] : fr.dstV[f];
    const int ss = (SRC || job == 0) ? a.ys : job == 1 ? a.us : a.vs, dstride = job == 0 ? a.ds : job == 1 ? a.dsU : a.dsV;
    const int srcRowBytes = SRC ? 3 * a.srcW : (job == 0 ? a.srcW : NV12 ? 2 * a.chrSrcW : a.chrSrcW) * G_BPS;
    const GPlane bS(sp, (unsigned)ss * (unsigned)(srcRows - 1) + (unsigned)srcRowBytes), bD(dp, (unsigned)dstride * (unsigned)(rows - 1) + (unsigned)rowBytes * (a.dst16 ? 2u : 1u));
    const int bcol = min(B0 + lane, rowBytes - 1);
    GPlaneOut out;
    out.set(a, job == 0 ? bcol : NV12 ? (bcol >> 1) + 3 * (bcol & 1) : bcol + (job == 2 ? 3 : 0));
    const int32_t *vt = job ? a.vtC : a.vtL;
    const int n4 = job ? a.n4C : a.n4L, sV = kGBlkHead + 4 * n4;
    const int rnd = job ? a.roundC : a.roundL;
    auto run = [&](auto s2_c, auto lk_c) {
        constexpr bool S2 = decltype(s2_c)::value;
        constexpr int LK = decltype(lk_c)::value;
        GStream<GPairs<P, S2>::N, S2, SD, J, LK> W;
        W.set_conv(GConv{a.src16, a.hShift, a.hBias});
        {
            const int w0 = W.setup(job ? a.hC : a.hL, job ? a.posC : a.posL, S2 ? bcol >> 1 : bcol, S2 ? bcol & 1 : 0);
            const int pos0 = uniform_load(job ? a.posC : a.posL, S2 ? B0 >> 1 : B0);      // lane 0's window (B0 is even), by a scalar load
            const int seg = g_win_base<S2>(pos0, 0);
            W.winDw = (w0 - seg) >> 2;
            g_stream_loads(W, seg, lane, S2, a, job);
            W.img = image[wave];
        }
        auto ld = [&](unsigned off, unsigned row) { unsigned v; bS.ld1(off, row, &v); return v; };
        const int pa = uniform_load(vt, y0 * sV), pb = uniform_load(vt, (y1 - 1) * sV + 1);
        g_blk_hpass(W, ld, pa, pb, wave, lane, (unsigned)ss, hS);
        __syncthreads();
        for (int y = y0 + wave; y < y1; y += 4) {
            const int32_t *rv = vt + (size_t)y * sV;
            const int acc = g_blk_vsum(rv, n4, hS, uniform_load(rv, 0) - pa, lane, rnd);
            out.store(bD, dp, acc, B0, lane, rowBytes, (unsigned)y * (unsigned)dstride, y);       // (the output stage of scale_yuvg_planes_kernel's emit())
        }
    };
    using LK0 = std::integral_constant<int, 0>;
#if G_BPS == 2
    if constexpr (SRC != 0) {
        if (job == 0) run(std::false_type(), std::integral_constant<int, 1>());
        else if (NV12) run(std::true_type(), std::integral_constant<int, 3>());
        else run(std::false_type(), std::integral_constant<int, 2>());
    } else
#endif
    if (NV12 && job == 1) run(std::true_type(), LK0()); else run(std::false_type(), LK0());
}

#if G_BPS == 2
// ---- packed RGB -> packed RGB at any ratio the walker reaches (round 5) -----------------------------------------------------------------------------------
// libswscale's generic path for an RGB source into an RGB frame (the BASELINE's literal second stage, rgb24 -> rgb24, away from its exact 2 : 1):
// rgb24ToY_c per pixel, rgb24ToUV_c per pixel or rgb24ToUV_half_c per pixel pair (chrSrcHSubSample = 1 from 2 : 1 on), both at FULL height; hScale16To15_c
// (sh = 13) of the three lines; yuv2rgb_full_X_c over them — full-chroma output is forced for a non-subsampled source (utils.c:1439-1447) and the vertical
// chroma filter is the luma one (utils.c:1838-1873) — and yuv2rgb_write_full (output.c:1886-1935, 2037-2082).  A lane owns one output column of a band;
// its U and V lines come from ONE stream: the raw pixels are loaded once and the converter fills two row images.
// HALF: chroma from pixel pairs (eight pixels = six raw dwords -> two image dwords a plane) instead of from every pixel (four pixels = three raw dwords)
template <int P, int SD, int R, bool HALF>
struct GStreamUV {
    static constexpr int NW = (P + 1) & ~1, IMG = 64 * SD, SDL = HALF ? 3 * SD : (3 * SD) / 2;
    int cf[P];
    int winDw, ldDw0;
    unsigned ldOff[SDL];
    unsigned ring[R][2][SDL];
    unsigned win[2][2][NW];              // [U | V][row of the pair][window dword]
    unsigned *img;                       // [U | V][row][IMG]
    unsigned reqOff, rowStep;
    int u01, u2, v01, v2;
    __device__ __forceinline__ int setup(const int32_t *hTab, const int32_t *posTab, int col)
    {
#pragma unroll
        for (int t = 0; t < P; t++) cf[t] = hTab[(size_t)col * P + t];
        return g_win_base<false>(posTab[col], 0);
    }
    __device__ __forceinline__ void start(int pair, int stride) { rowStep = (unsigned)stride; reqOff = (unsigned)(2 * pair) * (unsigned)stride; }
    template <class Ld> __device__ __forceinline__ void request(Ld &&ld, unsigned (&dst)[2][SDL])
    {
#pragma unroll
        for (int s = 0; s < SDL; s++) { dst[0][s] = ld(ldOff[s], reqOff); dst[1][s] = ld(ldOff[s], reqOff + rowStep); }
        reqOff += 2u * rowStep;
    }
    __device__ __forceinline__ void gather(const unsigned (&src)[2][SDL])
    {
        typedef GStream<P, false, SD, R, 1> G1;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int g = 0; g < SD / 2; g++) {
                int cu[4], cv[4];
                if constexpr (HALF) {
                    constexpr int KC = (256 << 15) + (1 << 9);          // rgb24ToUV_half_c on a pair's sums: >> 10
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        int fs[4], th[4];
                        G1::rgb4(&src[r][6 * g + 3 * h], fs, th);
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            const int f = fs[2 * q] + fs[2 * q + 1], t = th[2 * q] + th[2 * q + 1];
                            cu[2 * h + q] = g_dot2(f, u01, m24(t, u2) + KC) >> 10;
                            cv[2 * h + q] = g_dot2(f, v01, m24(t, v2) + KC) >> 10;
                        }
                    }
                } else {
                    constexpr int KC = (256 << 14) + (1 << 8);          // rgb24ToUV_c: >> 9
                    int fs[4], th[4];
                    G1::rgb4(&src[r][3 * g], fs, th);
#pragma unroll
                    for (int i = 0; i < 4; i++) { cu[i] = g_dot2(fs[i], u01, m24(th[i], u2) + KC) >> 9; cv[i] = g_dot2(fs[i], v01, m24(th[i], v2) + KC) >> 9; }
                }
                *reinterpret_cast<uint2 *>(img + r * IMG + ldDw0 + 2 * g) = make_uint2((unsigned)cu[0] | ((unsigned)cu[1] << 16), (unsigned)cu[2] | ((unsigned)cu[3] << 16));
                *reinterpret_cast<uint2 *>(img + (2 + r) * IMG + ldDw0 + 2 * g) = make_uint2((unsigned)cv[0] | ((unsigned)cv[1] << 16), (unsigned)cv[2] | ((unsigned)cv[3] << 16));
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < NW; i += 2) {
                    const uint2 t = *reinterpret_cast<const uint2 *>(img + (2 * c + r) * IMG + winDw + i);
                    win[c][r][i] = t.x; win[c][r][i + 1] = t.y;
                }
    }
    template <class Ld> __device__ __forceinline__ void prime(Ld &&ld)
    {
        unsigned first[2][SDL];
        request(ld, first);
#pragma unroll
        for (int d = 1; d <= R; d++) request(ld, ring[d % R]);
        gather(first);
    }
    // the gathered pair's horizontally filtered U and V samples of this lane's column, packed by rows: min(sum >> 13, 32767) (hScale16To15_c)
    __device__ __forceinline__ void hpairs(int &hu, int &hv) const
    {
        int h[2][2];
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int r = 0; r < 2; r++) {
                int sum = 0;
#pragma unroll
                for (int t = 0; t < P; t++) sum = g_dot2((int)win[c][r][t], cf[t], sum);
                h[c][r] = sum >> 13;
            }
        hu = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(h[0][0], h[0][1]));
        hv = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(h[1][0], h[1][1]));
    }
    template <int S, class Ld> __device__ __forceinline__ void advance(Ld &&ld) { gather(ring[S]); request(ld, ring[S]); }
};

// block = 4 waves = 4 adjacent strips of 64 output columns of one band; grid.y = frame.  Bands walk downward.  a.prog[0] is the program of ONE plane (a quad =
// four source rows = two row pairs), shared by the three lines
template <int P, int K, bool HALF>
__global__ __launch_bounds__(256) void scale_yuvg_rgbsrc_kernel(YuvGArgs a, Yuv2xFrames fr)
{
    constexpr int PC = HALF ? P : P, SD = P >= 10 ? 4 : 2, QS = kGHead + 3 * K;
    __shared__ __attribute__((aligned(16))) unsigned imageY[4][2 * 64 * SD];
    __shared__ __attribute__((aligned(16))) unsigned imageC[4][4 * 64 * SD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lin = blockIdx.x;
    if (a.xcdRemap) { const int chunk = (a.nblk + 7) >> 3; lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3); }
    if (lin >= a.nblk) return;
    const int band = __builtin_amdgcn_readfirstlane(lin / a.nsg);
    const int X0 = ((lin - band * a.nsg) * 4 + wave) * 64;
    if (X0 >= a.dstW) return;
    const int ya = (int)(((unsigned)band * (unsigned)a.bandStep) >> 16), yb = band + 1 >= a.nbands ? a.dstH : (int)(((unsigned)(band + 1) * (unsigned)a.bandStep) >> 16);
    const int f = blockIdx.y;
    const GPlane bS(fr.y[f], (unsigned)a.ys * (unsigned)(a.srcH - 1) + 3u * (unsigned)a.srcW);
    const int bpp = (a.dstFormat == GMAT_PIX_FMT_RGBA || a.dstFormat == GMAT_PIX_FMT_BGRA) ? 4 : 3;
    const bool bgr = a.dstFormat == GMAT_PIX_FMT_BGR24 || a.dstFormat == GMAT_PIX_FMT_BGRA;
    const GPlane bD(fr.dst[f], (unsigned)a.ds * (unsigned)(a.dstH - 1) + (unsigned)(a.dstW * bpp));
    const int x = X0 + lane, xc = min(x, a.dstW - 1);

    GStream<P, false, SD, 2, 1> L;
    GStreamUV<PC, SD, 2, HALF> C;
    L.set_conv(GConv{18, 13, 0});
    {
        const int w0 = L.setup(a.hL, a.posL, xc, 0);
        const int seg = __builtin_amdgcn_readfirstlane(w0);
        L.winDw = (w0 - seg) >> 2;
        g_stream_loads(L, seg, lane, false, a, 0);
        L.img = imageY[wave];
    }
    {
        const int w0 = C.setup(a.hC, a.posC, HALF ? min(xc, a.dstW - 1) : xc);
        const int seg = __builtin_amdgcn_readfirstlane(w0);
        C.winDw = (w0 - seg) >> 2;
        const int e0 = seg >> 1;                                      // first chroma sample of the segment
        C.ldDw0 = lane * SD;
#pragma unroll
        for (int s2 = 0; s2 < GStreamUV<PC, SD, 2, HALF>::SDL; s2++)
            C.ldOff[s2] = (unsigned)((HALF ? 6 : 3) * e0) + 4u * (unsigned)(lane * GStreamUV<PC, SD, 2, HALF>::SDL + s2);
        const Rgb2YuvConsts &k = a.r2y;
        auto pk = [](int lo, int hi) { return (int)(((unsigned)lo & 0xFFFFu) | ((unsigned)hi << 16)); };
        C.u01 = a.rgbBgr ? pk(k.bu, k.gu) : pk(k.ru, k.gu); C.u2 = a.rgbBgr ? k.ru : k.bu;
        C.v01 = a.rgbBgr ? pk(k.bv, k.gv) : pk(k.rv, k.gv); C.v2 = a.rgbBgr ? k.rv : k.bv;
        C.img = imageC[wave];
    }
    auto ld = [&](unsigned off, unsigned row) { unsigned v; bS.ld1(off, row, &v); return v; };
    const int32_t *prog = a.prog[0];
    const int q0 = uniform_load(a.qfirst[0], ya), q1 = uniform_load(a.qdone[0], yb - 1);
    int y = uniform_load(prog, q0 * QS);
    L.start(2 * q0, a.srcH, a.ys, 0);
    C.start(2 * q0, a.ys);
    L.prime(ld);
    C.prime(ld);
    int accY[K], accU[K], accV[K];
    // yuv2rgb_full_X_c: Y from 1 << 9, U and V from (1 << 9) - (128 << 19); >> 10
#pragma unroll
    for (int i = 0; i < K; i++) { accY[i] = a.roundL; accU[i] = accV[i] = a.roundC; }
    const unsigned dsel = (unsigned)((lane & 3) == 0 ? 0x04020100u : (lane & 3) == 1 ? 0x05040201u : 0x06050402u);
#define GMAT_G_QUAD(v, ctrl) __builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xF, 0xF, true)
    auto emit = [&](int yw) {
        const int Y = accY[0] >> 10, U = accU[0] >> 10, V = accV[0] >> 10;
        // yuv2rgb_write_full (output.c:1886-1935): every factor below 2^23 (the 24-bit multiplier gives the C expression's low 32 bits), av_clip_uintp2(x, 30)
        const int yy = m24(Y - a.y2r.y_offset, a.y2r.y_coeff) + (1 << 21);
        const int R = yy + m24(V, a.y2r.v2r), G = yy + m24(V, a.y2r.v2g) + m24(U, a.y2r.u2g), B = yy + m24(U, a.y2r.u2b);
        const unsigned r8 = (unsigned)min(max(R, 0), 0x3FFFFFFF) >> 22, g8 = (unsigned)min(max(G, 0), 0x3FFFFFFF) >> 22, b8 = (unsigned)min(max(B, 0), 0x3FFFFFFF) >> 22;
        const unsigned px = (bgr ? b8 : r8) | (g8 << 8) | ((bgr ? r8 : b8) << 16) | 0xFF000000u;
        const unsigned drow = (unsigned)yw * (unsigned)a.ds;
        if (bpp == 4) {
            if (x < a.dstW) bD.st1(px, 4u * (unsigned)x, drow);
        } else {
            const unsigned nxt = (unsigned)GMAT_G_QUAD(px, 0xF9);
            const unsigned o = __builtin_amdgcn_perm(nxt, px, dsel);
            const int nb = 3 * min(64, a.dstW - X0);
            const int ob = 12 * (lane >> 2) + 4 * (lane & 3);
            if ((lane & 3) != 3) {
                if (ob + 4 <= nb) bD.st1(o, 3u * (unsigned)X0 + (unsigned)ob, drow);
                else if (ob < nb) {
                    uint8_t *d = fr.dst[f] + (size_t)drow + 3u * (unsigned)X0 + (unsigned)ob;
                    for (int i = 0; i < nb - ob; i++) d[i] = (uint8_t)(o >> (8 * i));
                }
            }
        }
    };
#undef GMAT_G_QUAD
    int pend = 0;
    auto flush = [&]() {
        for (int k = 0; k < pend; k++, y++) {
            if (y >= ya && y < yb) emit(y);
#pragma unroll
            for (int i = 0; i + 1 < K; i++) { accY[i] = accY[i + 1]; accU[i] = accU[i + 1]; accV[i] = accV[i + 1]; }
            accY[K - 1] = a.roundL; accU[K - 1] = accV[K - 1] = a.roundC;
        }
    };
    auto quad = [&](int q) {
        const int32_t *pq = prog + (size_t)q * QS;
        int c0[K], c1[K];
#pragma unroll
        for (int i = 0; i < K; i++) { c0[i] = uniform_load(pq, kGHead + i); c1[i] = uniform_load(pq, kGHead + K + i); }
        const int ne = uniform_load(pq, 1);
        int hu, hv;
        const int hy0 = L.hpair();
        L.template advance<1>(ld);
        C.hpairs(hu, hv);
        C.template advance<1>(ld);
#pragma unroll
        for (int i = 0; i < K; i++) { accY[i] = g_dot2(hy0, c0[i], accY[i]); accU[i] = g_dot2(hu, c0[i], accU[i]); accV[i] = g_dot2(hv, c0[i], accV[i]); }
        const int hy1 = L.hpair();
        L.template advance<0>(ld);
        C.hpairs(hu, hv);
        C.template advance<0>(ld);
#pragma unroll
        for (int i = 0; i < K; i++) { accY[i] = g_dot2(hy1, c1[i], accY[i]); accU[i] = g_dot2(hu, c1[i], accU[i]); accV[i] = g_dot2(hv, c1[i], accV[i]); }
        pend = ne;
        flush();
    };
    for (int q = q0; q <= q1; q++) quad(q);
    flush();
}

// ---- the converter of the block-cooperative RGB-source kernels: the two rows of a pair side by side in the wave's halves ------------------------------------
// rgb24ToY_c / rgb24ToUV_c / rgb24ToUV_half_c's coefficients for a lane's byte order: bytes (0, 1) of a pixel as an int16 pair, byte 2 on its own
struct GRgbConv {
    int y01, y2, u01, u2, v01, v2;
    __device__ __forceinline__ GRgbConv(const Rgb2YuvConsts &k, int bgr)
    {
        auto pk = [](int lo, int hi) { return (int)(((unsigned)lo & 0xFFFFu) | ((unsigned)hi << 16)); };
        y01 = bgr ? pk(k.by, k.gy) : pk(k.ry, k.gy); y2 = bgr ? k.ry : k.by;
        u01 = bgr ? pk(k.bu, k.gu) : pk(k.ru, k.gu); u2 = bgr ? k.ru : k.bu;
        v01 = bgr ? pk(k.bv, k.gv) : pk(k.rv, k.gv); v2 = bgr ? k.rv : k.bv;
    }
};
// PPL pixels of ONE row (raw: 3 PPL / 4 dwords) -> PPL luma samples and PPL (HALF: PPL / 2, of pixel pairs) samples of U and of V, into row `half` of the
// wave's images at the lane's group of samples.  wantY / wantC (wave-uniform): lines nobody reads are not made
// BPX = 4: RGBA / BGRA pixels (rgb32ToY / ToUV read the same three channels with the same coefficients and ignore the fourth byte, input.c rgb16_32
// templates); ia != nullptr: the alpha line too — rgbaToA_c's a << 6 | a >> 2 (input.c:442-449), filtered like the luma line
template <bool HALF, int PPL, int BPX = 3>
__device__ __forceinline__ void g_rgb_rows(const unsigned *raw, const GRgbConv &cv, unsigned *iy, unsigned *iu, unsigned *iv, int half, int grp, bool wantY, bool wantC,
                                           unsigned *ia = nullptr)
{
    typedef GStream<4, false, 2, 1, 1> G1;
    constexpr int IY = 16 * PPL, IC = HALF ? IY / 2 : IY, NC = HALF ? PPL / 2 : PPL;
    constexpr int KY = (32 << 14) + (1 << 8);                            // rgb24ToY_c: >> 9
    int ys[PPL], us[NC], vs[NC], as[PPL];
#pragma unroll
    for (int h = 0; h < PPL / 4; h++) {
        int fs[4], th[4];
        if constexpr (BPX == 3) G1::rgb4(raw + 3 * h, fs, th);
        else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const unsigned d = raw[4 * h + i];
                fs[i] = (int)__builtin_amdgcn_perm(d, d, 0x0C010C00u);
                th[i] = (int)__builtin_amdgcn_perm(d, d, 0x0C0C0C02u);
                as[4 * h + i] = (int)(((d >> 18) & 0x3FC0u) | (d >> 26));                       // a << 6 | a >> 2
            }
        }
        if (wantY) {
#pragma unroll
            for (int i = 0; i < 4; i++) ys[4 * h + i] = g_dot2(fs[i], cv.y01, m24(th[i], cv.y2) + KY) >> 9;
        }
        if (wantC) {
            if constexpr (HALF) {
                constexpr int KC = (256 << 15) + (1 << 9);               // rgb24ToUV_half_c on a pair's sums: >> 10
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int fq = fs[2 * q] + fs[2 * q + 1], tq = th[2 * q] + th[2 * q + 1];           // (two 9-bit sums in the halves: no carry across)
                    us[2 * h + q] = g_dot2(fq, cv.u01, m24(tq, cv.u2) + KC) >> 10;
                    vs[2 * h + q] = g_dot2(fq, cv.v01, m24(tq, cv.v2) + KC) >> 10;
                }
            } else {
                constexpr int KC = (256 << 14) + (1 << 8);               // rgb24ToUV_c: >> 9
#pragma unroll
                for (int i = 0; i < 4; i++) { us[4 * h + i] = g_dot2(fs[i], cv.u01, m24(th[i], cv.u2) + KC) >> 9; vs[4 * h + i] = g_dot2(fs[i], cv.v01, m24(th[i], cv.v2) + KC) >> 9; }
            }
        }
    }
    auto put = [&](unsigned *img, int pitch, const int *v, int n) {      // n samples of this lane's row into its image
        unsigned *d = img + half * pitch + (n / 2) * grp;
        if (n == 8) *reinterpret_cast<uint4 *>(d) = make_uint4((unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
                                                                (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16));
        else if (n == 4) *reinterpret_cast<uint2 *>(d) = make_uint2((unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16));
        else        *d = (unsigned)v[0] | ((unsigned)v[1] << 16);
    };
    if (wantY) put(iy, IY, ys, PPL);
    if (wantC) { put(iu, IC, us, NC); put(iv, IC, vs, NC); }
    if constexpr (BPX == 4) { if (ia) put(ia, IY, as, PPL); }
}
// a lane's column of the two rows of an image: hScale16To15_c (sh = 13) over 8-byte aligned windows, the two samples packed (the pack saturates at 32767)
template <int P>
__device__ __forceinline__ int g_rgb_hfilt(const unsigned *img, int pitch, int win, const int (&cf)[P])
{
    constexpr int NW = (P + 1) & ~1;
    int h[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        int sum = 0;
#pragma unroll
        for (int i = 0; i < NW; i += 2) {
            const uint2 t = *reinterpret_cast<const uint2 *>(img + r * pitch + win + i);
            sum = g_dot2((int)t.x, cf[i], sum);
            if (i + 1 < P) sum = g_dot2((int)t.y, cf[i + 1], sum);
        }
        h[r] = sum >> 13;
    }
    return __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(h[0], h[1]));
}

// ---- the same conversion, block-cooperative: a launch of few frames, up-scales, windows of any height (round 5) ---------------------------------------------
// block = 64 output columns x a.bandRows output rows; grid.y = frame.  scale_yuvg_blk_rgb_kernel's two phases with THREE lines behind one load of the pixels:
// (1) the band's source row pairs are dealt round the waves.  The two rows of a pair are converted SIDE BY SIDE — lanes 0-31 take row 2p, lanes 32-63 row
//     2p + 1, PPL pixels a lane (3 PPL / 4 raw dwords) — into the wave's row images of Y, U and V: half the loads and half the converter instructions of
//     the band walker's streams, which convert 256 samples of BOTH rows in every lane.  Then every lane filters its column of the two rows of the three lines
//     (8-byte aligned windows, ds_read_b64, one v_dot2 a coefficient pair) and leaves three packed pairs of 15-bit samples in LDS;
// (2) one barrier; output row y0 + i belongs to wave i & 3: ONE set of coefficient pairs (the chroma's vertical filter is the luma's) down three LDS columns,
//     yuv2rgb_full_X_c's shifts, yuv2rgb_write_full.
// HALF: chroma from pixel pairs (2 : 1 and beyond).  PPL = 4 where every block's segment fits 128 pixels, 8 otherwise (HALF: always 8).  J: row pairs a wave
// requests at once.  The filtered pairs sit in dynamic LDS: 3 x a.blkSlots x 64 dwords.
// BPX = 4: an RGBA / BGRA source read as it is (no 32 -> 24-bit pass in front); ALPHA: its fourth byte is a fourth line — with an alpha channel at both ends
// libswscale scales the alpha plane through the luma filters (needAlpha, utils.c:1902) and yuv2rgb_full_X_c writes (2^18 + sum) >> 19 (output.c:2069-2077)
template <int P, bool HALF, int PPL, int J, int BPX = 3, bool ALPHA = false>
__global__ __launch_bounds__(256) void scale_yuvg_rgbsrc_blk_kernel(YuvGArgs a, Yuv2xFrames fr)
{
    static_assert(PPL == 4 || PPL == 8, "pixels a lane");
    static_assert(!HALF || PPL == 8, "pixel pairs: eight pixels a lane");
    static_assert(!ALPHA || BPX == 4, "an alpha line: four bytes a pixel");
    constexpr int IY = 16 * PPL, IC = HALF ? IY / 2 : IY;    // dwords of a row image: 32 lanes x PPL samples (chroma of pixel pairs: half of them)
    constexpr int RD = (BPX * PPL) / 4;                      // raw dwords a lane
    __shared__ __attribute__((aligned(16))) unsigned imgY[4][2][IY], imgU[4][2][IC], imgV[4][2][IC], imgA[ALPHA ? 4 : 1][2][ALPHA ? IY : 2];
    HIP_DYNAMIC_SHARED(uint4, lds_base)
    int (*hY)[64] = reinterpret_cast<int (*)[64]>(lds_base);
    int (*hU)[64] = hY + a.blkSlots, (*hV)[64] = hU + a.blkSlots, (*hA)[64] = hV + a.blkSlots;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lin = blockIdx.x;
    if (a.xcdRemap) { const int chunk = (a.nblk + 7) >> 3; lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3); }
    if (lin >= a.nblk) return;
    const int band = lin / a.nsg;
    const int X0 = (lin - band * a.nsg) * 64;
    const int y0 = band * a.bandRows, y1 = min(y0 + a.bandRows, a.dstH);
    const int f = blockIdx.y;
    // (rounded up to a whole dword: a width that is not a multiple of four ends its last row inside one — the same page as the row's last byte)
    const GPlane bS(fr.y[f], (unsigned)a.ys * (unsigned)(a.srcH - 1) + (((unsigned)BPX * (unsigned)a.srcW + 3u) & ~3u));
    const int bpp = (a.dstFormat == GMAT_PIX_FMT_RGBA || a.dstFormat == GMAT_PIX_FMT_BGRA) ? 4 : 3;
    const bool bgr = a.dstFormat == GMAT_PIX_FMT_BGR24 || a.dstFormat == GMAT_PIX_FMT_BGRA;
    const GPlane bD(fr.dst[f], (unsigned)a.ds * (unsigned)(a.dstH - 1) + (unsigned)(a.dstW * bpp));
    const int x = X0 + lane, xc = min(x, a.dstW - 1);

    // this lane's column: coefficient pairs and windows (in samples, multiples of 4); the block's first pixel from SCALAR loads of lane 0's windows
    int cfY[P], cfC[P];
#pragma unroll
    for (int t = 0; t < P; t++) { cfY[t] = a.hL[(size_t)xc * P + t]; cfC[t] = a.hC[(size_t)xc * P + t]; }
    const int sy0 = uniform_load(a.posL, X0) & ~3, sc0 = uniform_load(a.posC, X0) & ~3;
    const int px0 = HALF ? min(sy0 & ~7, 2 * sc0) : min(sy0, sc0);
    const int winY = ((a.posL[xc] & ~3) - px0) >> 1, winC = ((a.posC[xc] & ~3) - (HALF ? px0 >> 1 : px0)) >> 1;
    const int half = lane >> 5, grp = lane & 31;
    const unsigned lbase = (unsigned)BPX * (unsigned)px0 + (unsigned)(BPX * PPL) * (unsigned)grp + (unsigned)half * (unsigned)a.ys;
    const GRgbConv cv(a.r2y, a.rgbBgr);
    unsigned *const iy = imgY[wave][0], *const iu = imgU[wave][0], *const iv = imgV[wave][0], *const ia = ALPHA ? imgA[ALPHA ? wave : 0][0] : nullptr;

    const int sV = kGBlkHead + 4 * a.n4L;
    const int pa = uniform_load(a.vtL, y0 * sV), pb = uniform_load(a.vtL, (y1 - 1) * sV + 1);
    // (1) every slot is requested (the ones past the band's last pair as that pair again: no load under a branch)
    unsigned ring[J][RD];
#pragma unroll
    for (int j = 0; j < J; j++) {
        const unsigned off = (unsigned)(2 * min(pa + wave + 4 * j, pb)) * (unsigned)a.ys;
#pragma unroll
        for (int s = 0; s < RD; s++) bS.ld1(lbase + 4u * (unsigned)s, off, &ring[j][s]);
    }
#pragma unroll
    for (int j = 0; j < J; j++)
        if (pa + wave + 4 * j <= pb) {
            __builtin_amdgcn_wave_barrier();
            g_rgb_rows<HALF, PPL, BPX>(ring[j], cv, iy, iu, iv, half, grp, true, true, ia);
            __builtin_amdgcn_wave_barrier();
            const int slot = wave + 4 * j;
            hY[slot][lane] = g_rgb_hfilt<P>(iy, IY, winY, cfY);
            hU[slot][lane] = g_rgb_hfilt<P>(iu, IC, winC, cfC);
            hV[slot][lane] = g_rgb_hfilt<P>(iv, IC, winC, cfC);
            if constexpr (ALPHA) hA[slot][lane] = g_rgb_hfilt<P>(ia, IY, winY, cfY);
        }
    __syncthreads();

    // (2)
    const unsigned dsel = (unsigned)((lane & 3) == 0 ? 0x04020100u : (lane & 3) == 1 ? 0x05040201u : 0x06050402u);
#define GMAT_G_QUAD(v, ctrl) __builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xF, 0xF, true)
    for (int y = y0 + wave; y < y1; y += 4) {
        const int32_t *rv = a.vtL + (size_t)y * sV;
        const int base = uniform_load(rv, 0) - pa;
        const int rnd = uniform_load(a.vtRnd, y);                          // (1 << 9, or 0 in a row of yuv2rgb_full_2_c)
        int accY = rnd, accU = rnd - (128 << 19), accV = accU, accA = 1 << 18;
        for (int g = 0; g < a.n4L; g++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int c = uniform_load(rv, kGBlkHead + 4 * g + i), s = base + 4 * g + i;
                accY = g_dot2(hY[s][lane], c, accY); accU = g_dot2(hU[s][lane], c, accU); accV = g_dot2(hV[s][lane], c, accV);
                if constexpr (ALPHA) accA = g_dot2(hA[s][lane], c, accA);
            }
        }
        unsigned a8 = 255u;
        if constexpr (ALPHA) { int A = accA >> 19; if (A & 0x100) A = min(max(A, 0), 255); a8 = (unsigned)A & 0xFFu; }       // (clipped when bit 8 is set, as the writer does)
        const int Y = accY >> 10, U = accU >> 10, V = accV >> 10;
        // yuv2rgb_write_full (output.c:1886-1935): scale_yuvg_rgbsrc_kernel's emit()
        const int yy = m24(Y - a.y2r.y_offset, a.y2r.y_coeff) + (1 << 21);
        const int R = yy + m24(V, a.y2r.v2r), G = yy + m24(V, a.y2r.v2g) + m24(U, a.y2r.u2g), B = yy + m24(U, a.y2r.u2b);
        const unsigned r8 = (unsigned)min(max(R, 0), 0x3FFFFFFF) >> 22, g8 = (unsigned)min(max(G, 0), 0x3FFFFFFF) >> 22, b8 = (unsigned)min(max(B, 0), 0x3FFFFFFF) >> 22;
        const unsigned px = (bgr ? b8 : r8) | (g8 << 8) | ((bgr ? r8 : b8) << 16) | (a8 << 24);
        const unsigned drow = (unsigned)y * (unsigned)a.ds;
        if (bpp == 4) {
            if (x < a.dstW) bD.st1(px, 4u * (unsigned)x, drow);
        } else {
            const unsigned nxt = (unsigned)GMAT_G_QUAD(px, 0xF9);
            const unsigned o = __builtin_amdgcn_perm(nxt, px, dsel);
            const int nb = 3 * min(64, a.dstW - X0);
            const int ob = 12 * (lane >> 2) + 4 * (lane & 3);
            if ((lane & 3) != 3) {
                if (ob + 4 <= nb) bD.st1(o, 3u * (unsigned)X0 + (unsigned)ob, drow);
                else if (ob < nb) {
                    uint8_t *d = fr.dst[f] + (size_t)drow + 3u * (unsigned)X0 + (unsigned)ob;
                    for (int i = 0; i < nb - ob; i++) d[i] = (uint8_t)(o >> (8 * i));
                }
            }
        }
    }
#undef GMAT_G_QUAD
}
// ---- packed RGB -> 4:2:0, block-cooperative and fused (round 5) -------------------------------------------------------------------------------------------------
// scale_yuvg_blk_planes_kernel<.., SRC = 1> gives an RGB source three kinds of blocks — luma, U, V (or the interleaved chroma) — and every one of them loads
// and unpacks the pixels for itself, both rows of a pair in every lane.  Here a block owns 64 luma columns x a.bandRows luma rows AND the chroma beside them
// (32 columns x half the rows of U and of V): one load of the band's pixels, the converter of scale_yuvg_rgbsrc_blk_kernel (a pair's rows side by side in the
// wave's halves), then lanes 0-63 filter the luma line's columns and — NV12: even / odd lanes, planar: the wave's halves — the U / V lines' columns; the
// filtered pairs of both sit in LDS, one barrier, and the waves deal out the band's luma rows, then its chroma rows (each its own vertical table and
// output stage: yuv2planeX_8_c / yuv2nv12cX_c).  The band's row pairs are the union of its luma and chroma windows; a pair outside one of them skips that line.
// BPX = 4: an RGBA / BGRA source read as it is (its fourth byte ignored)
template <int P, bool HALF, int PPL, int J, int BPX = 3>
__global__ __launch_bounds__(256) void scale_yuvg_rgb2p_blk_kernel(YuvGArgs a, Yuv2xFrames fr)
{
    static_assert(PPL == 4 || PPL == 8, "pixels a lane");
    // (pixel pairs at four pixels a lane: two chroma samples, one image dword — ratios below 1.75 : 1, whose 64 columns span fewer than 128 pixels)
    constexpr int IY = 16 * PPL, IC = HALF ? IY / 2 : IY, RD = (BPX * PPL) / 4;
    __shared__ __attribute__((aligned(16))) unsigned imgY[4][2][IY], imgU[4][2][IC], imgV[4][2][IC];
    HIP_DYNAMIC_SHARED(uint4, lds_base)
    int (*hY)[64] = reinterpret_cast<int (*)[64]>(lds_base);
    int (*hC)[64] = hY + a.blkSlots;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lin = blockIdx.x;
    if (a.xcdRemap) { const int chunk = (a.nblk + 7) >> 3; lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3); }
    if (lin >= a.nblk) return;
    const int band = lin / a.nsg;
    const int X0 = (lin - band * a.nsg) * 64, C0 = X0 >> 1;
    const int y0 = band * a.bandRows, y1 = min(y0 + a.bandRows, a.dstH);
    const int cy0 = y0 >> 1, cy1 = min((y1 + 1) >> 1, a.chrDstH);
    const int f = blockIdx.y;
    const bool nv12 = a.nv12 != 0;
    // (rounded up to a whole dword: a width that is not a multiple of four ends its last row inside one — the same page as the row's last byte)
    const GPlane bS(fr.y[f], (unsigned)a.ys * (unsigned)(a.srcH - 1) + (((unsigned)BPX * (unsigned)a.srcW + 3u) & ~3u));
    const GPlane bY(fr.dst[f], (unsigned)a.ds * (unsigned)(a.dstH - 1) + (unsigned)a.dstW);
    const int crow = nv12 ? 2 * a.chrDstW : a.chrDstW;
    const GPlane bU(fr.dstU[f], (unsigned)a.dsU * (unsigned)(a.chrDstH - 1) + (unsigned)crow);
    const GPlane bV(nv12 ? fr.dstU[f] : fr.dstV[f], (unsigned)(nv12 ? a.dsU : a.dsV) * (unsigned)(a.chrDstH - 1) + (unsigned)crow);

    // this lane's luma column and its chroma column (NV12: component = lane & 1 of column lane >> 1; planar: component = lane >> 5 of column lane & 31)
    const int xc = min(X0 + lane, a.dstW - 1);
    const int comp = nv12 ? lane & 1 : lane >> 5;
    const int cc = min(C0 + (nv12 ? lane >> 1 : lane & 31), a.chrDstW - 1);
    int cfY[P], cfC[P];
#pragma unroll
    for (int t = 0; t < P; t++) { cfY[t] = a.hL[(size_t)xc * P + t]; cfC[t] = a.hCp[(size_t)cc * P + t]; }
    const int sy0 = uniform_load(a.posL, X0) & ~3, sc0 = uniform_load(a.posC, min(C0, a.chrDstW - 1)) & ~3;
    const int px0 = HALF ? min(sy0 & ~7, 2 * sc0) : min(sy0, sc0);
    const int winY = ((a.posL[xc] & ~3) - px0) >> 1, winC = ((a.posC[cc] & ~3) - (HALF ? px0 >> 1 : px0)) >> 1;
    const int half = lane >> 5, grp = lane & 31;
    const unsigned lbase = (unsigned)BPX * (unsigned)px0 + (unsigned)(BPX * PPL) * (unsigned)grp + (unsigned)half * (unsigned)a.ys;
    const GRgbConv cv(a.r2y, a.rgbBgr);
    unsigned *const iy = imgY[wave][0], *const iu = imgU[wave][0], *const iv = imgV[wave][0];
    const unsigned *const ic = comp ? iv : iu;

    const int sL = kGBlkHead + 4 * a.n4L, sC = kGBlkHead + 4 * a.n4C;
    const int paL = uniform_load(a.vtL, y0 * sL), pbL = uniform_load(a.vtL, (y1 - 1) * sL + 1);
    const bool anyC = cy1 > cy0;
    const int paC = anyC ? uniform_load(a.vtC, cy0 * sC) : paL, pbC = anyC ? uniform_load(a.vtC, (cy1 - 1) * sC + 1) : paL - 1;
    const int pa = min(paL, paC), pb = max(pbL, pbC);
    unsigned ring[J][RD];
#pragma unroll
    for (int j = 0; j < J; j++) {
        const unsigned off = (unsigned)(2 * min(pa + wave + 4 * j, pb)) * (unsigned)a.ys;
#pragma unroll
        for (int s = 0; s < RD; s++) bS.ld1(lbase + 4u * (unsigned)s, off, &ring[j][s]);
    }
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int pair = pa + wave + 4 * j;
        if (pair <= pb) {
            const bool wantY = pair >= paL && pair <= pbL, wantC = pair >= paC && pair <= pbC;
            __builtin_amdgcn_wave_barrier();
            g_rgb_rows<HALF, PPL, BPX>(ring[j], cv, iy, iu, iv, half, grp, wantY, wantC);
            __builtin_amdgcn_wave_barrier();
            const int slot = wave + 4 * j;
            if (wantY) hY[slot][lane] = g_rgb_hfilt<P>(iy, IY, winY, cfY);
            if (wantC) hC[slot][lane] = g_rgb_hfilt<P>(ic, IC, winC, cfC);
        }
    }
    __syncthreads();

    // luma rows
    {
        GPlaneOut out;
        out.set(a, xc);
        for (int y = y0 + wave; y < y1; y += 4) {
            const int32_t *rv = a.vtL + (size_t)y * sL;
            const int acc = g_blk_vsum(rv, a.n4L, hY, uniform_load(rv, 0) - pa, lane, a.roundL);
            out.store(bY, fr.dst[f], acc, X0, lane, a.dstW, (unsigned)y * (unsigned)a.ds, y);
        }
    }
    // chroma rows
    {
        const int bcol = min(X0 + lane, crow - 1);
        GPlaneOut out;
        out.set(a, nv12 ? (bcol >> 1) + 3 * (bcol & 1) : cc + 3 * comp);
        for (int y = cy0 + wave; y < cy1; y += 4) {
            const int32_t *rv = a.vtC + (size_t)y * sC;
            int acc = g_blk_vsum(rv, a.n4C, hC, uniform_load(rv, 0) - pa, lane, a.roundC);
            if (nv12) { out.store(bU, fr.dstU[f], acc, X0, lane, crow, (unsigned)y * (unsigned)a.dsU, y); continue; }
            // planar: GPlaneOut::store's 8-bit stage, lanes 0-31 into U's row, lanes 32-63 into V's
            if (__builtin_amdgcn_readfirstlane(out.dither8)) acc += ((out.xpart ^ dither_8x8_128(0, y) ^ 36) - 64) << 12;
            const unsigned v = (unsigned)clip_u8_shr(acc, 19);
            const unsigned pr = v | ((unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xF5, 0xF, 0xF, true) << 8);
            const unsigned o = pr | ((unsigned)__builtin_amdgcn_update_dpp(0, (int)pr, 0xAA, 0xF, 0xF, true) << 16);
            const int l5 = lane & 31, nb = min(32, a.chrDstW - C0);
            if ((lane & 3) == 0 && l5 < nb) {
                const unsigned drowU = (unsigned)y * (unsigned)a.dsU, drowV = (unsigned)y * (unsigned)a.dsV;      // (wave-uniform: the store's scalar offset)
                if (l5 + 4 <= nb) { if (comp) bV.st1(o, (unsigned)(C0 + l5), drowV); else bU.st1(o, (unsigned)(C0 + l5), drowU); }
                else {
                    uint8_t *d = (comp ? fr.dstV[f] + (size_t)drowV : fr.dstU[f] + (size_t)drowU) + (unsigned)(C0 + l5);
                    for (int i = 0; i < nb - l5; i++) d[i] = (uint8_t)(o >> (8 * i));
                }
            }
        }
    }
}
#endif

#if G_BPS == 2
} // namespace g16
using namespace g16;
#endif

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// The vertical program of a plane class pair (luma + the chroma beside it; or one plane on its own, fc == nullptr: a "quad" is then
// four rows of that plane) in one walking direction (up: everything mirrored, so that the kernel always counts upward).  Per quad q:
//   [0]          the first output row not complete before q (the row slot 0 of the running sums stands for while q is consumed)
//   [1]          the number of output rows complete after q
//   [2 .. 2+K)   int16 pairs (tap on luma row 4q, tap on 4q + 1) of output rows [0] + i;  [2+K ..): rows 4q + 2, 4q + 3;
//   [2+2K ..)    (tap on chroma row 2q, tap on 2q + 1)
// qfirst[y] / qdone[y]: the quad in which output row y's windows start / after which it is complete.
static bool build_qprog(const FilterBank &fl, int srcRowsL, const FilterBank *fc, int srcRowsC, bool up, YuvGQProg &v)
{
    const int rows = fl.count;
    if (fc && fc->count != rows) return false;
    auto mirror = [&](const FilterBank &fb, int srcRows, std::vector<int> &pos, std::vector<int16_t> &cf) {
        pos.resize(rows); cf.resize((size_t)rows * fb.taps);
        for (int y = 0; y < rows; y++) {
            const int ys = up ? rows - 1 - y : y;
            pos[y] = up ? srcRows - (fb.pos[ys] + fb.taps) : fb.pos[ys];
            for (int t = 0; t < fb.taps; t++) cf[(size_t)y * fb.taps + t] = fb.coef[(size_t)ys * fb.taps + (up ? fb.taps - 1 - t : t)];
            if (pos[y] < 0 || pos[y] + fb.taps > srcRows) return false;
            if (y && pos[y] < pos[y - 1]) return false;                      // the walk needs monotone windows
        }
        return true;
    };
    std::vector<int> posL, posC;
    std::vector<int16_t> cfL, cfC;
    if (!mirror(fl, srcRowsL, posL, cfL)) return false;
    if (fc && !mirror(*fc, srcRowsC, posC, cfC)) return false;
    v.qfirst.assign(rows, 0); v.qdone.assign(rows, 0);
    int Q = 0;
    for (int y = 0; y < rows; y++) {
        int s = posL[y] >> 2, d = (posL[y] + fl.taps - 1) >> 2;
        if (fc) { s = std::min(s, posC[y] >> 1); d = std::max(d, (posC[y] + fc->taps - 1) >> 1); }
        v.qfirst[y] = s; v.qdone[y] = d;
        Q = std::max(Q, d + 1);
    }
    std::vector<int> yBase(Q + 1, rows);
    for (int q = 0, y = 0; q <= Q; q++) { while (y < rows && v.qdone[y] < q) y++; yBase[q] = y; }
    int K = 1;
    for (int q = 0; q < Q; q++) {
        int hi = yBase[q] - 1;
        for (int y = yBase[q]; y < rows && v.qfirst[y] <= q; y++) hi = y;
        K = std::max(K, hi - yBase[q] + 1);
    }
    v.K = K; v.Q = Q; v.rows = rows;
    v.yBase = yBase; v.posL = posL; v.posC = posC; v.cfL = cfL; v.cfC = cfC; v.tapsL = fl.taps; v.tapsC = fc ? fc->taps : 0;
    return true;
}
static void fill_qprog(YuvGQProg &v, int K)
{
    const int QS = kGHead + 3 * K;
    v.prog.assign((size_t)(v.Q + 1) * QS, 0);
    auto tap = [](const std::vector<int16_t> &cf, int taps, int y, int t) { return t >= 0 && t < taps ? (int)cf[(size_t)y * taps + t] : 0; };
    auto pk = [](int lo, int hi) { return (int32_t)((uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16)); };
    for (int q = 0; q <= v.Q; q++) {
        int32_t *r = &v.prog[(size_t)q * QS];
        r[0] = v.yBase[q];
        r[1] = q < v.Q ? v.yBase[q + 1] - v.yBase[q] : 0;
        if (q == v.Q) break;
        for (int i = 0; i < K; i++) {
            const int y = v.yBase[q] + i;
            if (y >= v.rows) break;
            r[kGHead + i] = pk(tap(v.cfL, v.tapsL, y, 4 * q - v.posL[y]), tap(v.cfL, v.tapsL, y, 4 * q + 1 - v.posL[y]));
            r[kGHead + K + i] = pk(tap(v.cfL, v.tapsL, y, 4 * q + 2 - v.posL[y]), tap(v.cfL, v.tapsL, y, 4 * q + 3 - v.posL[y]));
            if (v.tapsC) r[kGHead + 2 * K + i] = pk(tap(v.cfC, v.tapsC, y, 2 * q - v.posC[y]), tap(v.cfC, v.tapsC, y, 2 * q + 1 - v.posC[y]));
        }
    }
}

// K = output rows open at once.  The plane jobs have the registers for 12 and 15 as well (57-67 VGPRs at K = 9): 4:2:0 -> 4:2:0
// UP-scales up to 2:1 (720p -> 1080p needs 10, 1080p -> 1440p 10, 1:2 14-15); an RGB destination needs 14-22 there and stays tiled.
// P = 13 (round 4): 26 taps — bicubic down to 6.2 : 1 (4K -> 360p), bilinear / area twice as far
// the block-cooperative form's vertical table of a plane class (by output row, walking down): [first row pair, last row pair, 4 n4 coefficient pairs on
// the row pairs from the first one on]
static void g_blk_vtab(const FilterBank &fb, std::vector<int32_t> &out, int &n4)
{
    const int npv = fb.taps / 2 + 1;                          // row pairs a window can touch (it may start on an odd row)
    n4 = (npv + 3) / 4;
    const int stride = kGBlkHead + 4 * n4;
    out.assign((size_t)fb.count * stride, 0);
    for (int y = 0; y < fb.count; y++) {
        const int pos = fb.pos[y], p0 = pos >> 1;
        int32_t *r = &out[(size_t)y * stride];
        r[0] = p0; r[1] = (pos + fb.taps - 1) >> 1;
        for (int k = 0; k < 4 * n4; k++) {
            const int t0 = 2 * (p0 + k) - pos, t1 = t0 + 1;
            const int lo = t0 >= 0 && t0 < fb.taps ? fb.coef[(size_t)y * fb.taps + t0] : 0, hi = t1 >= 0 && t1 < fb.taps ? fb.coef[(size_t)y * fb.taps + t1] : 0;
            r[kGBlkHead + k] = (int32_t)((uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16));
        }
    }
}
// the tallest band (rows, <= cap) that fits J pairs a wave from ANY start row (slots: LDS pair slots behind the 4 J of the band's own)
static int g_blk_tallest(const std::vector<int32_t> &vt, int n4, int count, int J, int cap, int pad)
{
    const int stride = kGBlkHead + 4 * n4;
    int best = cap;
    for (int y0 = 0; y0 < count && best > 0; y0++) {
        const int pa = vt[(size_t)y0 * stride];
        int fit = 0;
        for (int y = y0; y < std::min(count, y0 + best); y++) {
            const int p0 = vt[(size_t)y * stride], pl = vt[(size_t)y * stride + 1];
            if (pl - pa + 1 > 4 * J || p0 - pa + 4 * n4 > 4 * J + pad) break;
            fit = y - y0 + 1;
        }
        if (y0 + fit < count || fit == best) best = std::min(best, fit);      // (a band cut short by the plane's end fits whatever its nominal height)
    }
    return best;
}

#if G_BPS == 2
static const int kGP[] = {4, 5, 6, 7, 8, 10, 13};           // (7: the 10 / 11 taps of 2.2 - 2.5 : 1 behind three leading zeros)
#else
static const int kGP[] = {4, 5, 6, 8, 10, 13};
#endif
static const int kGK[] = {4, 6, 7, 9}, kGKPlanes[] = {4, 6, 7, 9, 12, 15};

#if G_BPS == 2
// the fused block form of an RGB source into an 8-bit 4:2:0 frame (scale_yuvg_rgb2p_blk_kernel): the chroma's pairs on a PLANE's aligned windows, a
// block's luma and chroma windows inside 32 lanes x PPL pixels from its first pixel on, and per band height the row pairs its windows span together.
// t.vtL / t.vtC (g_blk_vtab) are there already
static void g_rgb2p_tables(const ScalePlan &p, int P, YuvGTables &t)
{
    if (GMAT_KNOB("GMAT_RGBSRC_NO_FUSED") && atoi(GMAT_KNOB("GMAT_RGBSRC_NO_FUSED"))) return;
    const int NW = (P + 1) & ~1;
    const bool half = p.chrSrcHSub != 0;
    bool ok = true;
    t.hCp.assign((size_t)p.hChr.count * P, 0);
    for (int x = 0; x < p.hChr.count && ok; x++) {
        const int lead = p.hChr.pos[x] & 3;
        if (p.hChr.pos[x] < 0 || p.hChr.pos[x] + p.hChr.taps > p.chrSrcW || lead + p.hChr.taps > 2 * P) ok = false;
        for (int k = 0; k < P && ok; k++) {
            const int t0 = 2 * k - lead, t1 = t0 + 1;
            const int lo = t0 >= 0 && t0 < p.hChr.taps ? p.hChr.coef[(size_t)x * p.hChr.taps + t0] : 0, hi = t1 >= 0 && t1 < p.hChr.taps ? p.hChr.coef[(size_t)x * p.hChr.taps + t1] : 0;
            t.hCp[(size_t)x * P + k] = (int32_t)((uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16));
        }
    }
    int ppl = 4;
    for (int c0 = 0; c0 < p.dstW && ok; c0 += 64) {
        const int c1 = std::min(c0 + 64, p.dstW) - 1, k0 = std::min(c0 >> 1, p.chrDstW - 1), k1 = std::min((c0 >> 1) + 31, p.chrDstW - 1);
        const int sy0 = p.hLum.pos[c0] & ~3, sc0 = p.hChr.pos[k0] & ~3;
        const int px0 = half ? std::min(sy0 & ~7, 2 * sc0) : std::min(sy0, sc0), pc0 = half ? px0 / 2 : px0;
        int endY = 0, endC = 0;
        for (int x = c0; x <= c1 && ok; x++) { const int b = p.hLum.pos[x] & ~3; ok = b >= px0; endY = std::max(endY, b + 2 * NW - px0); }
        for (int x = k0; x <= k1 && ok; x++) { const int b = p.hChr.pos[x] & ~3; ok = b >= pc0; endC = std::max(endC, b + 2 * NW - pc0); }
        while (ppl <= 8 && (endY > 32 * ppl || endC > (half ? 16 : 32) * ppl)) ppl *= 2;
        if (ppl > 8) ok = false;
    }
    if (!ok) { t.hCp.clear(); return; }
    const int sL = kGBlkHead + 4 * t.n4L, sC = kGBlkHead + 4 * t.n4C;
    for (int i = 1; i <= 16; i++) {
        const int B = 4 * i;
        int most = 0;
        for (int y0 = 0; y0 < p.dstH; y0 += B) {
            const int y1 = std::min(y0 + B, p.dstH), cy0 = y0 >> 1, cy1 = std::min((y1 + 1) >> 1, p.chrDstH);
            int pa = t.vtL[(size_t)y0 * sL], pb = t.vtL[(size_t)(y1 - 1) * sL + 1];
            if (cy1 > cy0) { pa = std::min(pa, t.vtC[(size_t)cy0 * sC]); pb = std::max(pb, t.vtC[(size_t)(cy1 - 1) * sC + 1]); }
            most = std::max(most, pb - pa + 1);
        }
        t.f2Pairs[i] = most;
    }
    t.f2PPL = ppl;
}
#endif

#if G_BPS == 2
// the plane jobs of a packed RGB source (scale_yuvg_planes_kernel / scale_yuvg_blk_planes_kernel with SRC = 1): launched for launch_scale_yuvg16 with its
// arguments and grid as they are; the instances live in the second translation unit
int launch_yuvg_planes_rgbsrc(const YuvGArgs &a, dim3 grid, hipStream_t stream, const Yuv2xFrames &fr, bool blk);
#if G_PART != 1
int launch_yuvg_planes_rgbsrc(const YuvGArgs &a, dim3 grid, hipStream_t stream, const Yuv2xFrames &fr, bool blk)
{
    const dim3 block(256);
#define GMAT_PB(P_) do { if (a.nv12) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_blk_planes_kernel<P_, true, 1>), grid, block, 0, stream, a, fr); \
                         else        hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_blk_planes_kernel<P_, false, 1>), grid, block, 0, stream, a, fr); } while (0)
#define GMAT_PL(P_, K_) do { if (a.nv12) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_planes_kernel<P_, K_, true, 1>), grid, block, 0, stream, a, fr); \
                             else        hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_planes_kernel<P_, K_, false, 1>), grid, block, 0, stream, a, fr); } while (0)
#define GMAT_PP(P_) do { if (blk) GMAT_PB(P_); else switch (a.K) { case 4: GMAT_PL(P_, 4); break; case 6: GMAT_PL(P_, 6); break; case 7: GMAT_PL(P_, 7); break; \
                                                                   case 9: GMAT_PL(P_, 9); break; case 12: GMAT_PL(P_, 12); break; default: GMAT_PL(P_, 15); } } while (0)
    switch (a.P) { case 4: GMAT_PP(4); break; case 5: GMAT_PP(5); break; case 6: GMAT_PP(6); break; case 7: GMAT_PP(7); break; case 8: GMAT_PP(8); break;
                   case 10: GMAT_PP(10); break; default: GMAT_PP(13); }
#undef GMAT_PP
#undef GMAT_PL
#undef GMAT_PB
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}
#endif
#endif

#if G_PART != 2
int G_NAME(yuvg_prepare)(const ScalePlan &p, const YuvScaleTiling &g, YuvGTables &t)
{
    t = YuvGTables();
    const char *off = GMAT_KNOB("GMAT_SCALE_NO_GENERIC_WALKER");
    if (off && atoi(off)) return 0;
    const bool rgbOut = p.dstFormat == GMAT_PIX_FMT_RGB24 || p.dstFormat == GMAT_PIX_FMT_BGR24 || p.dstFormat == GMAT_PIX_FMT_RGBA || p.dstFormat == GMAT_PIX_FMT_BGRA;
    // 4:2:0 destinations: 8 bits, or (round 5) 10 bits in 16-bit stores — P010LE / YUV420P10LE: the same 15-bit lines, yuv2p010lX_c / yuv2planeX_10_c's shift
    // (round 5, last: YUV444P -> YUV444P — a format of scale_cuda's list, vf_scale_cuda.c:45-54 — is three plane jobs like any planar frame: the chroma jobs' tables and
    // sizes simply are the full-size ones; it ran the lines form's two passes at 0.12 of the roofline)
    // (round 6: the other planar pairs with a 4:4:4 end — YUV444P -> YUV420P, YUV420P -> YUV444P — are three plane jobs too, the chroma jobs' source and destination
    // sizes whatever the plan says; the format sweep found them on the lines form's two launches, 0.11-0.13 of the roofline)
    const bool pl8s = p.srcFormat == GMAT_PIX_FMT_YUV420P || p.srcFormat == GMAT_PIX_FMT_YUV444P, pl8d = p.dstFormat == GMAT_PIX_FMT_YUV420P || p.dstFormat == GMAT_PIX_FMT_YUV444P;
    const bool p444 = G_BPS == 1 && pl8s && pl8d && (p.srcFormat == GMAT_PIX_FMT_YUV444P || p.dstFormat == GMAT_PIX_FMT_YUV444P);
    const bool yuvOut = p.dstFormat == GMAT_PIX_FMT_NV12 || p.dstFormat == GMAT_PIX_FMT_YUV420P || is_dst10(p.dstFormat) || p444;
    const bool semiDst = p.dstFormat == GMAT_PIX_FMT_NV12 || p.dstFormat == GMAT_PIX_FMT_P010LE;
#if G_BPS == 2
    // a packed RGB24 / BGR24 source into a 4:2:0 frame (the walker's own converter: GStream LK): the chroma of pixel PAIRS at full height — what libswscale
    // gives every such context that does not up-scale (utils.c:1529-1545); the jobs synthesise the DESTINATION's chroma layout, so "semi" follows it
    const bool rgbSrc = p.srcFormat == GMAT_PIX_FMT_RGB24 || p.srcFormat == GMAT_PIX_FMT_BGR24;
    if (rgbSrc && yuvOut && !is_dst10(p.dstFormat) && !p.chrSrcVSub && p.chrSrcH == p.srcH && g.yuvOut == 1 && p.dstW >= 16 && p.dstH >= 8 && p.srcW >= 16 && p.srcH >= 8 &&
        (p.chrSrcHSub ? (p.chrSrcW * 2 == p.srcW && (p.srcW & 3)) : p.chrSrcW == p.srcW)) {
        // an UP-scale (chroma from every pixel, utils.c:1529-1545), or a down-scale of a width that is not a multiple of four (the walker's streams load whole
        // groups of four pixels): the fused block form alone — no walker instance (t.K = 0), no plane jobs
        for (int v : g.lumRound) if (v != g.lumRound[0]) return 0;
        for (int v : g.chrRound) if (v != g.chrRound[0]) return 0;
        int need = 0;
        for (const FilterBank *fb : {&p.hLum, &p.hChr})
            for (int x = 0; x < fb->count; x++) {
                if (fb->pos[x] < 0 || fb->pos[x] + fb->taps > (fb == &p.hLum ? p.srcW : p.chrSrcW)) return 0;
                need = std::max(need, ((fb->pos[x] & 3) + fb->taps + 1) / 2);
            }
        if (need > 8) return 0;
        const int P = std::max(need, 4);
        t.hL.assign((size_t)p.hLum.count * P, 0);
        for (int x = 0; x < p.hLum.count; x++) {
            const int lead = p.hLum.pos[x] & 3;
            for (int k = 0; k < P; k++) {
                const int t0 = 2 * k - lead, t1 = t0 + 1;
                const int lo = t0 >= 0 && t0 < p.hLum.taps ? p.hLum.coef[(size_t)x * p.hLum.taps + t0] : 0, hi = t1 >= 0 && t1 < p.hLum.taps ? p.hLum.coef[(size_t)x * p.hLum.taps + t1] : 0;
                t.hL[(size_t)x * P + k] = (int32_t)((uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16));
            }
        }
        t.posL = p.hLum.pos; t.posC = p.hChr.pos;
        t.roundL = g.lumRound[0]; t.roundC = g.chrRound[0];
        g_blk_vtab(g.vLumEff, t.vtL, t.n4L);
        g_blk_vtab(g.vChrEff, t.vtC, t.n4C);
        g_rgb2p_tables(p, P, t);
        if (!t.f2PPL || t.f2Pairs[2] > 32) { t = YuvGTables(); return 0; }
        t.hC = t.hCp;                                                  // (the walker's table slot: never read without an instance of it)
        t.P = P; t.K = 0; t.yuvOut = 1;
        t.ok = 1;
        return 0;
    }
    if (rgbSrc && (!yuvOut || !p.chrSrcHSub || p.chrSrcVSub || p.chrSrcW * 2 != p.srcW || p.chrSrcH != p.srcH || (p.srcW & 3))) return 0;
    const bool semiSrc = rgbSrc ? semiDst : is_p01x(p.srcFormat);
    if (!(rgbSrc || semiSrc || p.srcFormat == GMAT_PIX_FMT_YUV420P10LE || p.srcFormat == GMAT_PIX_FMT_YUV420P16LE) || !(rgbOut || yuvOut)) return 0;
    if (const char *o16 = GMAT_KNOB("GMAT_SCALE_NO_WALKER16")) if (atoi(o16)) return 0;
    // the 16-bit image is biased by -32768 and the sums start at 32768 * 16384: every horizontal row sums to 16384 (initFilter normalises exactly)
    for (const FilterBank *fb : {&p.hLum, &p.hChr})
        for (int x = 0; x < fb->count; x++) {
            int sum = 0;
            for (int j = 0; j < fb->taps; j++) sum += fb->coef[(size_t)x * fb->taps + j];
            if (sum != 16384) return 0;
        }
#else
    const bool semiSrc = p.srcFormat == GMAT_PIX_FMT_NV12;
    if (!(p.srcFormat == GMAT_PIX_FMT_NV12 || p.srcFormat == GMAT_PIX_FMT_YUV420P || p444) || !(rgbOut || yuvOut)) return 0;
    if ((p.srcFormat == GMAT_PIX_FMT_YUV444P) != p444) return 0;
#endif
    if (rgbOut && (g.fullChroma || g.yuvOut)) return 0;
    if (yuvOut && g.yuvOut != (p.dstFormat == GMAT_PIX_FMT_YUV444P ? 2 : 1)) return 0;
    if (yuvOut && semiSrc != semiDst) return 0;                           // same chroma layout on both sides
    if (p.dstW < 16 || p.dstH < 8 || p.srcW < 16 || p.srcH < 8) return 0;
    // whole dwords inside every source row (the rows are dword loads checked against the plane's exact size)
    if ((p.srcW * G_BPS) % 4 || ((semiSrc ? 2 * p.chrSrcW : p.chrSrcW) * G_BPS) % 4) return 0;
    // RGB: one chroma sample per pixel pair and per output row (the LUT form); 4:2:0: the chroma planes of the destination
    if (rgbOut && (p.chrDstW != (p.dstW + 1) / 2 || p.chrDstH != p.dstH)) return 0;
    // the sums start at ONE value per plane class (true of every multi-tap vertical filter; the 1- and 2-tap special forms of
    // vscale.c:135-167 have per-row starts and stay on the tiled kernel)
    for (int v : g.lumRound) if (v != g.lumRound[0]) return 0;
    for (int v : g.chrRound) if (v != g.chrRound[0]) return 0;
    t.roundL = g.lumRound[0]; t.roundC = g.chrRound[0];
    // horizontal: coefficient pairs on the table's own windows (a window may start anywhere; the last pair of an odd tap count is padded)
    // (16-bit samples: the windows re-based to 8-byte boundaries — `lead` zero taps in front: pos & 3 samples of a plane, pos & 1 positions of an interleaved row)
    auto hpack = [&](const FilterBank &fb, int srcLen, std::vector<int32_t> &out, int P, bool s2) {
        out.assign((size_t)fb.count * P, 0);
        for (int x = 0; x < fb.count; x++) {
            if (fb.pos[x] < 0 || fb.pos[x] + fb.taps > srcLen) return false;
            const int lead = G_BPS == 2 ? (s2 ? fb.pos[x] & 1 : fb.pos[x] & 3) : 0;
            if (lead + fb.taps > 2 * P) return false;
            for (int k = 0; k < P; k++) {
                const int t0 = 2 * k - lead, t1 = t0 + 1;
                const int lo = t0 >= 0 && t0 < fb.taps ? fb.coef[(size_t)x * fb.taps + t0] : 0, hi = t1 >= 0 && t1 < fb.taps ? fb.coef[(size_t)x * fb.taps + t1] : 0;
                out[(size_t)x * P + k] = (int32_t)((uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16));
            }
        }
        return true;
    };
    auto pairs_needed = [&](const FilterBank &fb, bool s2) {
        int m = 0;
        for (int x = 0; x < fb.count; x++) m = std::max(m, (G_BPS == 2 ? (s2 ? fb.pos[x] & 1 : fb.pos[x] & 3) : 0) + fb.taps);
        return (m + 1) / 2;
    };
    // (the chroma stream of an interleaved 16-bit row runs on one pair fewer than the instance's P: GPairs)
    const int needP = std::max(pairs_needed(p.hLum, false), pairs_needed(p.hChr, semiSrc) + (G_BPS == 2 && semiSrc ? 1 : 0));
    int P = 0;
    for (int c : kGP) if (c >= needP) { P = c; break; }
    if (!P) return 0;
    if (!hpack(p.hLum, p.srcW, t.hL, P, false) || !hpack(p.hChr, p.chrSrcW, t.hC, G_BPS == 2 && semiSrc ? P - 1 : P, semiSrc)) return 0;
    t.posL = p.hLum.pos; t.posC = p.hChr.pos;
    int needK = 0;
    for (int up = 0; up < 2; up++) {
        if (rgbOut) {
            if (!build_qprog(g.vLumEff, p.srcH, &g.vChrEff, p.chrSrcH, up != 0, t.rgb[up])) return 0;
            needK = std::max(needK, t.rgb[up].K);
        } else {
            if (!build_qprog(g.vLumEff, p.srcH, nullptr, 0, up != 0, t.pl[up])) return 0;
            if (!build_qprog(g.vChrEff, p.chrSrcH, nullptr, 0, up != 0, t.pc[up])) return 0;
            needK = std::max(needK, std::max(t.pl[up].K, t.pc[up].K));
        }
    }
    int K = 0;
    if (yuvOut) { for (int c : kGKPlanes) if (c >= needK) { K = c; break; } }
    else        { for (int c : kGK) if (c >= needK) { K = c; break; } }
    // up-scales into RGB (four-tap filters: P = 4): the chroma of an RGB destination is up-scaled twice as far as its luma, 14-22 rows open
    if (!K && rgbOut && P == 4) for (int c : {15, 18, 22}) if (c >= needK) { K = c; break; }
    if (!K) { if (GMAT_KNOB("GMAT_DEBUG_WALKER")) logf(LOG_ERROR, "yuvg: %dx%d -> %dx%d declined: K needed %d", p.srcW, p.srcH, p.dstW, p.dstH, needK); return 0; }
    for (int up = 0; up < 2; up++) { if (rgbOut) fill_qprog(t.rgb[up], K); else { fill_qprog(t.pl[up], K); fill_qprog(t.pc[up], K); } }
    // the row segments: a wave's 64 (luma, planar chroma plane) or 32 (the chroma of an RGB destination, NV12's interleaved chroma)
    // consecutive columns must fit the row image of SD * 256 bytes (planar chroma halves under an RGB destination: SD * 128)
    {
        const int SD = G_BPS == 2 ? (P >= 10 ? 4 : 2) : (P >= 8 ? 2 : 1);     // (16-bit samples: eight pairs hold the 13 taps of 3 : 1 behind three leading zeros — 416 bytes a row)
        auto fits = [&](const FilterBank &fb, int cols, bool s2, int capBytes) {
            const int NW = G_BPS == 2 ? (s2 ? 2 * (P - 1) : (P + 1) & ~1) : (s2 ? P + 1 : ((P - 1) >> 1) + 2);
            for (int c0 = 0; c0 < fb.count; c0 += cols) {
                const int c1 = std::min(c0 + cols, fb.count) - 1;
                const int b0 = s2 ? g_win_base<true>(fb.pos[c0], 0) : g_win_base<false>(fb.pos[c0], 0);
                const int b1 = s2 ? g_win_base<true>(fb.pos[c1], 1) : g_win_base<false>(fb.pos[c1], 0);
                if (b1 + 4 * NW - b0 > capBytes) return false;
            }
            return true;
        };
        const bool nv12 = semiSrc;
        if (!fits(p.hLum, 64, false, 256 * SD)) return 0;
        if (rgbOut ? !fits(p.hChr, 32, nv12, nv12 ? 256 * SD : 128 * SD) : !fits(p.hChr, nv12 ? 32 : 64, nv12, 256 * SD)) return 0;
    }
    t.P = P; t.K = K; t.yuvOut = yuvOut;
    // the block-cooperative form's vertical tables (by output row, walking down) and the tallest band whose row pairs fit a block:
    // checked for EVERY start row, so that the launcher may cut bands of any height up to it anywhere
    {
        auto vtab = [&](const FilterBank &fb, std::vector<int32_t> &out, int &n4) { g_blk_vtab(fb, out, n4); };
        auto tallest = [&](const std::vector<int32_t> &vt, int n4, int count, int J, int cap) { return g_blk_tallest(vt, n4, count, J, cap, kGBlkPad); };
        vtab(g.vLumEff, t.vtL, t.n4L);
        vtab(g.vChrEff, t.vtC, t.n4C);
        if (rgbOut) {
            const int rows = std::min(tallest(t.vtL, t.n4L, g.vLumEff.count, kGBlkJL, 32), tallest(t.vtC, t.n4C, g.vChrEff.count, kGBlkJC, 32)) & ~3;
            t.blkRows = rows; t.blkRowsC = 0;
        } else {
            const int rl = tallest(t.vtL, t.n4L, g.vLumEff.count, kGBlkJL, 32) & ~3, rc = tallest(t.vtC, t.n4C, g.vChrEff.count, kGBlkJL, 16) & ~1;
            t.blkRows = std::min(rl, 2 * rc) & ~3; t.blkRowsC = t.blkRows / 2;
        }
        if (t.n4L > 4 || t.n4C > 4) t.blkRows = 0;                    // (windows of more than 16 row pairs: the walker alone)
        if (G_BPS == 2 && P >= 10) t.blkRows = 0;                     // (16-bit samples, four dwords of a row per lane: 64 registers of requested pairs)
    }
#if G_BPS == 2
    if (rgbSrc && yuvOut && !is_dst10(p.dstFormat) && P <= 8) g_rgb2p_tables(p, P, t);
#endif
    if (GMAT_KNOB("GMAT_DEBUG_WALKER")) logf(LOG_ERROR, "yuvg: %dx%d -> %dx%d taps h %d/%d v %d/%d -> P %d, K needed %d -> %d", p.srcW, p.srcH, p.dstW, p.dstH,
                                          p.hLum.taps, p.hChr.taps, g.vLumEff.taps, g.vChrEff.taps, P, needK, K);
    t.ok = 1;
    return 0;
}

// a short launch takes the block-cooperative form: up to 3 frames, or — small outputs — up to 2.8 M output pixels (round 5, every frame from HBM,
// walker / block form, us a launch: 4K -> 854 x 480 rgb24 4 frames 33.1 / 29.6, 8: 50.6 / 53.3; 1080p -> 768 x 432 4 frames 13.9 / 11.5, 8: 19.8 / 18.5,
// 16: 30.4 / 31.6; 4K -> 900p 4 frames 30.5 / 32.9: profiles/r05h_yuvg_blk_frames.txt).  GMAT_STRIP_BLOCK=n: launches of up to n frames; 0: never
bool G_NAME(yuvg_block_form)(const YuvGArgs &a, int nframes)
{
    if (a.blkRows < 4 || !a.vtL || !a.vtC) return false;
    if (const char *bs = GMAT_KNOB("GMAT_STRIP_BLOCK")) return nframes <= atoi(bs);
    return nframes <= 3 || (long)a.dstW * a.dstH * nframes <= 2800000L;
}

static int launch_scale_yuvg_blk(const YuvGArgs &a0, hipStream_t stream, const Yuv2xFrames &fr, int nframes)
{
    YuvGArgs a = a0;
    const char *rowsStr = GMAT_KNOB("GMAT_STRIP_ROWS");          // tuning / test override: output rows per band
    const int rowsEnv = rowsStr ? atoi(rowsStr) : 0;
    a.nsg = (a.dstW + 63) / 64;
    // band height: tall bands pay the vertical windows' lead-in less often, short ones fill the chip — a block per CU and SIMD pair at least
    int rows = a.blkRows;
    if (rowsEnv > 0) rows = std::min(a.blkRows, std::max(4, rowsEnv & ~3));
    else {
        // measured, one frame a launch (profiles/r04n_blk_rows_sweep.txt): the tallest band that still leaves four blocks a CU
        const int jobs = a.yuvOut ? 2 : 1;                           // (a 4:2:0 destination's chroma jobs: as many blocks again, half as tall)
        while (rows > 8 && (long)a.nsg * ((a.dstH + rows - 1) / rows) * nframes * jobs < 1024) rows -= 4;
    }
    a.bandRows = rows;
    a.nbands = (a.dstH + rows - 1) / rows;
    a.nblkL = a.nbands * a.nsg;
    a.nblk = a.nblkL;
    if (a.yuvOut) {
        const int cbytes = a.nv12 ? 2 * a.chrDstW : a.chrDstW;
        a.nsgC = (cbytes + 63) / 64;
        a.bandRowsC = rows / 2;
        a.nbandsC = (a.chrDstH + a.bandRowsC - 1) / a.bandRowsC;
        a.nblkC = a.nbandsC * a.nsgC;
        a.nblk = a.nblkL + (a.nv12 ? 1 : 2) * a.nblkC;
    }
    const dim3 grid(a.xcdRemap ? 8 * ((a.nblk + 7) / 8) : a.nblk, nframes), block(256);
#if G_BPS == 2
    if (a.yuvOut && a.src16 == 3) return launch_yuvg_planes_rgbsrc(a, grid, stream, fr, true);        // (their instances: k_scale_yuvg16b.hip)
#endif
#define GMAT_GB(P_) do { \
        if (a.yuvOut) { if (a.nv12) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_blk_planes_kernel<P_, true>), grid, block, 0, stream, a, fr); \
                        else        hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_blk_planes_kernel<P_, false>), grid, block, 0, stream, a, fr); } \
        else          { if (a.nv12) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_blk_rgb_kernel<P_, true>), grid, block, 0, stream, a, fr); \
                        else        hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_blk_rgb_kernel<P_, false>), grid, block, 0, stream, a, fr); } } while (0)
#if G_BPS == 2
    if (a.P == 7) { GMAT_GB(7); } else
#endif
    switch (a.P) { case 4: GMAT_GB(4); break; case 5: GMAT_GB(5); break; case 6: GMAT_GB(6); break; case 8: GMAT_GB(8); break; case 10: GMAT_GB(10); break; default: GMAT_GB(13); }
#undef GMAT_GB
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

#endif  // G_PART != 2
#if G_BPS == 2
// a packed RGB source into an 8-bit 4:2:0 frame on the fused block form: an up-scale always (no other form); a down-scale below 1.75 : 1 (four pixels a lane:
// a block's 64 columns span fewer than 128 pixels) always too — us a frame, plane jobs (walker or its block form) / fused, rgb24 1080p -> 720p nv12: one frame 10.2 /
// 8.7, two 7.7 / 7.1, four 8.6 / 5.7, 32: 5.84 / 3.46; beyond (eight pixels a lane) from three frames a launch on — 4K -> 720p: one 17.5 / 18.4, two 14.5 / 14.6,
// three 13.5 / 13.3, four 15.7 / 12.3, 32: 12.2 / 11.26 (profiles/r05v_rgb2p_fused.txt): alone, three kinds of smaller blocks fill the chip better than one whose
// chroma window is 40 % of its row pairs.  GMAT_RGBSRC_FUSED=n: from n frames a launch on (1: always, 0: never)
#if G_PART != 2
bool yuvg_rgb2p_fused(const YuvGArgs &a, int nframes)
{
    if (a.src16 != 3 || !a.yuvOut || !a.f2PPL || !a.hCp || !a.vtL || !a.vtC || a.f2Pairs[2] > 32) return false;
    if (!a.K || a.srcPx == 4) return true;                              // (an up-scale: no other form; four-byte pixels: no other form reads them)
    if (const char *fs = GMAT_KNOB("GMAT_RGBSRC_FUSED")) return atoi(fs) > 0 && nframes >= atoi(fs);
    return a.f2PPL == 4 || nframes >= 3;
}
#endif
int launch_scale_yuvg_rgb2p(const YuvGArgs &a0, hipStream_t stream, const Yuv2xFrames &fr, int nframes);
#if G_PART != 1
int launch_scale_yuvg_rgb2p(const YuvGArgs &a0, hipStream_t stream, const Yuv2xFrames &fr, int nframes)
{
    YuvGArgs a = a0;
    const char *rowsStr = GMAT_KNOB("GMAT_STRIP_ROWS");
    const int rowsEnv = rowsStr ? atoi(rowsStr) : 0;
    a.nsg = (a.dstW + 63) / 64;
    // the tallest band whose row pairs fit (32), then shorter ones until the launch has four blocks a CU (never below 8 rows)
    int rows = 8;
    for (int i = 16; i >= 2; i--) if (a.f2Pairs[i] <= 32) { rows = 4 * i; break; }
    if (rowsEnv > 0) rows = std::min(rows, std::max(8, rowsEnv & ~3));
    else while (rows > 8 && (long)a.nsg * ((a.dstH + rows - 1) / rows) * nframes < 1024) rows -= 4;
    a.bandRows = rows;
    a.nbands = (a.dstH + rows - 1) / rows;
    a.nblkL = a.nblk = a.nbands * a.nsg;
    const int J = a.f2Pairs[rows / 4] <= 16 ? 4 : 8;
    a.blkSlots = 4 * J + 4 * std::max(a.n4L, a.n4C);
    const size_t lds = (size_t)2 * a.blkSlots * 64 * 4;
    const dim3 grid(a.xcdRemap ? 8 * ((a.nblk + 7) / 8) : a.nblk, nframes), block(256);
    const bool half = a.chrSrcW != a.srcW;
    const bool px4 = a.srcPx == 4;
#define GMAT_R2J(P_, H_, L_, J_) do { if (px4) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_rgb2p_blk_kernel<P_, H_, L_, J_, 4>), grid, block, lds, stream, a, fr); \
                                      else     hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_rgb2p_blk_kernel<P_, H_, L_, J_>), grid, block, lds, stream, a, fr); } while (0)
#define GMAT_R2(P_, H_, L_) do { if (J == 4) GMAT_R2J(P_, H_, L_, 4); else GMAT_R2J(P_, H_, L_, 8); } while (0)
#define GMAT_R2P(P_) do { if (half) { if (a.f2PPL == 4) GMAT_R2(P_, true, 4); else GMAT_R2(P_, true, 8); } \
                          else      { if (a.f2PPL == 4) GMAT_R2(P_, false, 4); else GMAT_R2(P_, false, 8); } } while (0)
    switch (a.P) { case 4: GMAT_R2P(4); break; case 5: GMAT_R2P(5); break; case 6: GMAT_R2P(6); break; case 7: GMAT_R2P(7); break; default: GMAT_R2P(8); }
#undef GMAT_R2P
#undef GMAT_R2
#undef GMAT_R2J
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}
#endif  // G_PART != 1
#endif

#if G_PART != 2
int G_NAME(launch_scale_yuvg)(const YuvGArgs &a0, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (!frames || nframes < 1 || nframes > kYuv2xMaxFrames) return GMAT_ERR(EINVAL);
#if G_BPS == 2
    if (yuvg_rgb2p_fused(a0, nframes)) return launch_scale_yuvg_rgb2p(a0, stream, *frames, nframes);
#endif
    if (G_NAME(yuvg_block_form)(a0, nframes)) return launch_scale_yuvg_blk(a0, stream, *frames, nframes);
    YuvGArgs a = a0;
    const char *rowsStr = GMAT_KNOB("GMAT_STRIP_ROWS");          // tuning / test override, read per launch
    const int rowsEnv = rowsStr ? atoi(rowsStr) : 0;
    const char *ud = GMAT_KNOB("GMAT_STRIP_UPDOWN");
    a.updown = !(ud && !atoi(ud));
    const int nstrips = (a.dstW + 63) / 64;
    a.nsg = (nstrips + 3) / 4;
    // band height: a band pays its vertical windows' lead-in (taps / ratio output rows' worth of source rows) — far more than the
    // exact-ratio walkers' three row pairs — so its optimum sits higher: 4K -> 900p rgb24 at 8 / 16 / 24 / 32 rows 7.4 / 6.1 / 5.9 /
    // 5.7 us per frame (profiles/r03p_rows_sweep_all_strip_kernels.txt); a lone small frame wants enough waves to fill the chip
    const long wr = (long)a.dstH * nstrips * nframes;
    // (an up-scale's bands are cheap in source rows and dear in open sums: twice the height — 1080p -> 1440p alone 17.0 -> 14.3-15.9 us)
    const bool upV = a.dstH > a.srcH;
    // (round 4's last sweep, 32 frames a launch, 32 / 48 rows: an RGB destination 4K -> 900p 181 / 175 us a launch, lanczos 256 / 247.6, -> 640 x 360 142.3 / 127.5; a 4:2:0
    // destination -> 900p 173.5 / 178, -> 640 x 360 147.7 / 159: profiles/r04_rows_all.txt)
    const long cap = a.yuvOut ? 32 : 48;
    int rows = rowsEnv > 0 ? rowsEnv : upV ? (int)std::min(32L, std::max(8L, (wr + 3071) / 3072)) : (int)std::min(cap, std::max(4L, (wr + 6143) / 6144));
    // a full launch of down-scaling bands (the rule at its cap): BALANCED bands — round(rows / 32) of them, heights differing by at most one row (900
    // rows: 28 bands of 32 or 33 instead of 28 x 32 and a 4-row one that pays a whole lead-in); otherwise bands of exactly `rows` rows
    const bool balanced = rowsEnv <= 0 && !upV && rows == (int)cap && a.dstH >= 2 * cap && a.dstH < 32768;
    a.bandRows = rows;
    a.nbands = balanced ? (a.dstH + rows / 2) / rows : (a.dstH + rows - 1) / rows;
    a.bandStep = balanced ? (int)(((unsigned)a.dstH << 16) / (unsigned)a.nbands) : rows << 16;
    a.nblkL = a.nbands * a.nsg;
    a.nblk = a.nblkL;
    if (a.yuvOut) {
        const int cbytes = a.nv12 ? 2 * a.chrDstW : a.chrDstW;
        a.nsgC = ((cbytes + 63) / 64 + 3) / 4;
        a.bandRowsC = std::max(2, rows / 2);
        a.nbandsC = balanced ? std::max(1, (a.chrDstH + a.bandRowsC / 2) / a.bandRowsC) : (a.chrDstH + a.bandRowsC - 1) / a.bandRowsC;
        a.bandStepC = balanced ? (int)(((unsigned)a.chrDstH << 16) / (unsigned)a.nbandsC) : a.bandRowsC << 16;
        a.nblkC = a.nbandsC * a.nsgC;
        a.nblk = a.nblkL + (a.nv12 ? 1 : 2) * a.nblkC;
    }
    const dim3 grid(a.xcdRemap ? 8 * ((a.nblk + 7) / 8) : a.nblk, nframes), block(256);
    const Yuv2xFrames &fr = *frames;
#if G_BPS == 2
    if (a.yuvOut && a.src16 == 3) return launch_yuvg_planes_rgbsrc(a, grid, stream, fr, false);       // (their instances: k_scale_yuvg16b.hip)
#endif
#define GMAT_G_PL(P_, K_) do { if (a.nv12) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_planes_kernel<P_, K_, true>), grid, block, 0, stream, a, fr); \
                               else        hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_planes_kernel<P_, K_, false>), grid, block, 0, stream, a, fr); } while (0)
#define GMAT_G_RGB(P_, K_) do { if (a.nv12) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_rgb_kernel<P_, K_, true>), grid, block, 0, stream, a, fr); \
                                else        hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_rgb_kernel<P_, K_, false>), grid, block, 0, stream, a, fr); } while (0)
#define GMAT_G_P(P_) do { \
        if (a.yuvOut) switch (a.K) { case 4: GMAT_G_PL(P_, 4); break; case 6: GMAT_G_PL(P_, 6); break; case 7: GMAT_G_PL(P_, 7); break; case 9: GMAT_G_PL(P_, 9); break; \
                                     case 12: GMAT_G_PL(P_, 12); break; default: GMAT_G_PL(P_, 15); } \
        else          switch (a.K) { case 4: GMAT_G_RGB(P_, 4); break; case 6: GMAT_G_RGB(P_, 6); break; case 7: GMAT_G_RGB(P_, 7); break; default: GMAT_G_RGB(P_, 9); } } while (0)
#define GMAT_G_P4() do { \
        if (!a.yuvOut && a.K > 9) switch (a.K) { case 15: GMAT_G_RGB(4, 15); break; case 18: GMAT_G_RGB(4, 18); break; default: GMAT_G_RGB(4, 22); } \
        else GMAT_G_P(4); } while (0)
#if G_BPS == 2
    if (a.P == 7) { GMAT_G_P(7); } else
#endif
    switch (a.P) { case 4: GMAT_G_P4(); break; case 5: GMAT_G_P(5); break; case 6: GMAT_G_P(6); break; case 8: GMAT_G_P(8); break; case 10: GMAT_G_P(10); break; default: GMAT_G_P(13); }
#undef GMAT_G_P
#undef GMAT_G_P4
#undef GMAT_G_PL
#undef GMAT_G_RGB
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

#endif  // G_PART != 2

#if G_BPS == 2
// ---- packed RGB -> packed RGB (scale_yuvg_rgbsrc_kernel) ---------------------------------------------------------------------------------------------------
// p: the RGB scaler's plan (gsws.cpp init_scaler: full-chroma output, the chroma lines at full height and full or half width)
#if G_PART != 2
int yuvg_rgbsrc_prepare(const ScalePlan &p, YuvGTables &t)
{
    t = YuvGTables();
    const char *off = GMAT_KNOB("GMAT_SCALE_NO_GENERIC_WALKER");
    if (off && atoi(off)) return 0;
    if (const char *o16 = GMAT_KNOB("GMAT_SCALE_NO_WALKER16")) if (atoi(o16)) return 0;
    if (!(p.srcFormat == GMAT_PIX_FMT_RGB24 || p.srcFormat == GMAT_PIX_FMT_BGR24)) return 0;
    if (!(p.dstFormat == GMAT_PIX_FMT_RGB24 || p.dstFormat == GMAT_PIX_FMT_BGR24 || p.dstFormat == GMAT_PIX_FMT_RGBA || p.dstFormat == GMAT_PIX_FMT_BGRA)) return 0;
    if (p.dstW < 16 || p.dstH < 8 || p.srcW < 16 || p.srcH < 8) return 0;
    // (exactly 2 : 1 has its own strip walker, k_scale_rgb2s.hip, in front of these forms wherever it takes the frame: dword-aligned planes of three-byte pixels)
    // full-chroma output (forced for this source), chroma lines of the source's height and of its width or half of it
    if (p.chrDstW != p.dstW || p.chrDstH != p.dstH || p.chrSrcVSub || p.chrSrcH != p.srcH) return 0;
    // (any width, an even one where the chroma comes from pixel pairs: the block form; the walker's form wants whole groups of four / eight pixels a row)
    if (!(p.chrSrcHSub == 0 ? p.chrSrcW == p.srcW : p.chrSrcW * 2 == p.srcW)) return 0;
    const bool walkW = !(p.srcW & (p.chrSrcHSub ? 7 : 3));
    // yuv2rgb_full_X_c proper (the one- and two-tap special forms of vscale.c:135-167 stay on the tiled kernel); the chroma's vertical filter is the luma's
    // (round 5, later: the block form takes those too — packed_vscale's forms differ from yuv2rgb_full_X_c in the sums' start alone, which it reads per output row)
    if (p.vLum.taps < 1 || p.vChr.taps != p.vLum.taps || p.vChr.pos != p.vLum.pos || p.vChr.coef != p.vLum.coef) return 0;
    if (p.hLum.count != p.dstW || p.hChr.count != p.dstW) return 0;
    auto lead = [](const FilterBank &fb, int x) { return fb.pos[x] & 3; };
    auto pairs_needed = [&](const FilterBank &fb) { int m = 0; for (int x = 0; x < fb.count; x++) m = std::max(m, lead(fb, x) + fb.taps); return (m + 1) / 2; };
    const int needP = std::max(pairs_needed(p.hLum), pairs_needed(p.hChr));
    int P = 0;
    for (int c : kGP) if (c >= needP) { P = c; break; }
    if (!P) return 0;
    auto hpack = [&](const FilterBank &fb, int srcLen, std::vector<int32_t> &out) {
        out.assign((size_t)fb.count * P, 0);
        for (int x = 0; x < fb.count; x++) {
            if (fb.pos[x] < 0 || fb.pos[x] + fb.taps > srcLen) return false;
            const int ld = lead(fb, x);
            for (int k = 0; k < P; k++) {
                const int t0 = 2 * k - ld, t1 = t0 + 1;
                const int lo = t0 >= 0 && t0 < fb.taps ? fb.coef[(size_t)x * fb.taps + t0] : 0, hi = t1 >= 0 && t1 < fb.taps ? fb.coef[(size_t)x * fb.taps + t1] : 0;
                out[(size_t)x * P + k] = (int32_t)((uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16));
            }
        }
        return true;
    };
    if (!hpack(p.hLum, p.srcW, t.hL) || !hpack(p.hChr, p.chrSrcW, t.hC)) return 0;
    t.posL = p.hLum.pos; t.posC = p.hChr.pos;
    // the band walker's form (scale_yuvg_rgbsrc_kernel): the running sums of one plane's program, a wave's 64 columns inside the row image its lanes fill
    int K = 0;
    if (walkW && p.vLum.taps >= 3 && build_qprog(p.vLum, p.srcH, nullptr, 0, false, t.rgb[0])) {
        for (int c : kGK) if (c >= t.rgb[0].K) { K = c; break; }
        if (!K && t.rgb[0].K <= 12 && P <= 8) K = 12;          // (ratios near 1 and up-scales: ten or eleven rows open over a quad of four source rows)
        const int SD = P >= 10 ? 4 : 2, NW = (P + 1) & ~1;
        for (const FilterBank *fb : {&p.hLum, &p.hChr})
            for (int c0 = 0; c0 < fb->count; c0 += 64) {
                const int c1 = std::min(c0 + 64, fb->count) - 1;
                if (g_win_base<false>(fb->pos[c1], 0) + 4 * NW - g_win_base<false>(fb->pos[c0], 0) > 256 * SD) K = 0;
            }
        if (K) fill_qprog(t.rgb[0], K);
    }
    // the block-cooperative form (scale_yuvg_rgbsrc_blk_kernel): windows of up to 16 row pairs, any number of rows open at once (up-scales); a block's
    // 64 columns inside 32 lanes x PPL pixels of a row, from the block's first pixel (the kernel's px0) on
    if (P <= 8 && !(GMAT_KNOB("GMAT_RGBSRC_NO_BLOCK") && atoi(GMAT_KNOB("GMAT_RGBSRC_NO_BLOCK")))) {
        // packed_vscale's forms (vscale.c:135-160): one tap — the line as it is (yuv2rgb_full_1_c: the coefficient is never read: 4096 here, and the X form's
        // rounding constant drops out of the shift); two taps that are a proper blend — yuv2rgb_full_2_c, no rounding constant (output.c:2118-2120);
        // everything else yuv2rgb_full_X_c from 1 << 9.  The sums' start by output row: DevFilterStore::upload's rule for the tiled kernel (gsws.cpp)
        FilterBank vb = p.vLum;
        if (vb.taps == 1) std::fill(vb.coef.begin(), vb.coef.end(), (int16_t)4096);
        t.vtRnd.assign(vb.count, 1 << 9);
        if (vb.taps == 2)
            for (int y = 0; y < vb.count; y++) {
                const int f0 = vb.coef[(size_t)y * 2], f1 = vb.coef[(size_t)y * 2 + 1];
                if (f0 + f1 == 4096 && (unsigned)f1 <= 4096u) t.vtRnd[y] = 0;
            }
        g_blk_vtab(vb, t.vtL, t.n4L);
        const bool half = p.chrSrcHSub != 0;
        const int NW = (P + 1) & ~1;
        int ppl = half ? 8 : 4;
        for (int c0 = 0; c0 < p.dstW && ppl <= 8; c0 += 64) {
            const int c1 = std::min(c0 + 64, p.dstW) - 1;
            const int sy0 = p.hLum.pos[c0] & ~3, sc0 = p.hChr.pos[c0] & ~3;
            const int px0 = half ? std::min(sy0 & ~7, 2 * sc0) : std::min(sy0, sc0);
            int endY = 0, endC = 0;                                   // (window starts need not be monotonic at the borders: every column)
            bool before = false;
            for (int x = c0; x <= c1; x++) {
                endY = std::max(endY, (p.hLum.pos[x] & ~3) + 2 * NW - px0);
                endC = std::max(endC, (p.hChr.pos[x] & ~3) + 2 * NW - (half ? px0 / 2 : px0));
                before = before || (p.hLum.pos[x] & ~3) < px0 || (p.hChr.pos[x] & ~3) < (half ? px0 / 2 : px0);
            }
            if (before) { ppl = 16; break; }
            while (ppl <= 8 && (endY > 32 * ppl || endC > (half ? 16 : 32) * ppl)) ppl *= 2;
        }
        if (ppl <= 8 && t.n4L <= 4) {
            const int cap = p.dstH > p.srcH ? 64 : 32;
            t.blkRows = g_blk_tallest(t.vtL, t.n4L, p.vLum.count, 8, cap, 4 * t.n4L) & ~3;
            t.blkRows4 = g_blk_tallest(t.vtL, t.n4L, p.vLum.count, 4, cap, 4 * t.n4L) & ~3;
            t.blkPPL = ppl;
            if (t.blkRows < 4) t.blkRows = t.blkRows4 = 0;
        }
    }
    if (GMAT_KNOB("GMAT_DEBUG_WALKER")) logf(LOG_ERROR, "yuvg rgbsrc: %dx%d -> %dx%d taps h %d/%d v %d -> P %d, K needed %d -> %d; block form rows %d / %d, %d pixels a lane",
                                          p.srcW, p.srcH, p.dstW, p.dstH, p.hLum.taps, p.hChr.taps, p.vLum.taps, P, t.rgb[0].K, K, t.blkRows, t.blkRows4, t.blkPPL);
    if (!K && !t.blkRows) return 0;
    t.roundL = 1 << 9; t.roundC = (1 << 9) - (128 << 19);          // yuv2rgb_full_X_c (output.c:2037-2082)
    t.P = P; t.K = K; t.walkOk = K > 0; t.yuvOut = 0;
    t.ok = 1;
    return 0;
}

// a launch of nframes frames on the block-cooperative form: wherever it has an instance — it is in front of the tiled kernel AND of the walker's form at
// every launch size (us a frame, tiled or walker / this form: rgb24 1080p -> 720p alone 13.1 / 8.2, 32 frames a launch 6.55 / 4.20; 4K -> 900p 27.4 / 19.4,
// 22.0 / 12.6; 720p -> 1080p 13.7 / 10.4, 9.25 / 5.47: profiles/r05u_rgbrgb_block_form.txt).  GMAT_RGBSRC_BLOCK=n: launches of up to n frames where the
// walker has an instance too (0: never there; tests, A/B)
bool yuvg_rgbsrc_block_form(const YuvGArgs &a, int nframes)
{
    if (a.blkRows < 4 || !a.vtL) return false;
    if (!a.K) return true;
    if (const char *bs = GMAT_KNOB("GMAT_RGBSRC_BLOCK")) return nframes <= atoi(bs);
    return true;
}
#endif
int launch_scale_yuvg_rgbsrc_blk(const YuvGArgs &a0, hipStream_t stream, const Yuv2xFrames &fr, int nframes);
#if G_PART != 1
int launch_scale_yuvg_rgbsrc_blk(const YuvGArgs &a0, hipStream_t stream, const Yuv2xFrames &fr, int nframes)
{
    YuvGArgs a = a0;
    const char *rowsStr = GMAT_KNOB("GMAT_STRIP_ROWS");
    const int rowsEnv = rowsStr ? atoi(rowsStr) : 0;
    a.nsg = (a.dstW + 63) / 64;
    int rows = a.blkRows;
    if (rowsEnv > 0) rows = std::min(a.blkRows, std::max(4, rowsEnv & ~3));
    else while (rows > 8 && (long)a.nsg * ((a.dstH + rows - 1) / rows) * nframes < 1024) rows -= 4;      // (four blocks a CU, as launch_scale_yuvg_blk)
    a.bandRows = rows;
    a.nbands = (a.dstH + rows - 1) / rows;
    a.nblkL = a.nblk = a.nbands * a.nsg;
    const int J = rows <= a.blkRows4 ? 4 : 8;
    a.blkSlots = 4 * J + 4 * a.n4L;
    const bool px4 = a.srcPx == 4, alpha = px4 && a.srcAlpha;
    const size_t lds = (size_t)(alpha ? 4 : 3) * a.blkSlots * 64 * 4;
    const dim3 grid(a.xcdRemap ? 8 * ((a.nblk + 7) / 8) : a.nblk, nframes), block(256);
    const bool half = a.chrSrcW != a.srcW;
#define GMAT_RBJ(P_, H_, L_, J_) do { \
        if (alpha)    hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_rgbsrc_blk_kernel<P_, H_, L_, J_, 4, true>), grid, block, lds, stream, a, fr); \
        else if (px4) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_rgbsrc_blk_kernel<P_, H_, L_, J_, 4, false>), grid, block, lds, stream, a, fr); \
        else          hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_rgbsrc_blk_kernel<P_, H_, L_, J_>), grid, block, lds, stream, a, fr); } while (0)
#define GMAT_RB(P_, H_, L_) do { if (J == 4) GMAT_RBJ(P_, H_, L_, 4); else GMAT_RBJ(P_, H_, L_, 8); } while (0)
#define GMAT_RB_P(P_) do { if (half) GMAT_RB(P_, true, 8); else if (a.blkPPL == 4) GMAT_RB(P_, false, 4); else GMAT_RB(P_, false, 8); } while (0)
    switch (a.P) { case 4: GMAT_RB_P(4); break; case 5: GMAT_RB_P(5); break; case 6: GMAT_RB_P(6); break; case 7: GMAT_RB_P(7); break; default: GMAT_RB_P(8); }
#undef GMAT_RB_P
#undef GMAT_RB
#undef GMAT_RBJ
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}
#endif  // G_PART != 1

#if G_PART != 2
int launch_scale_yuvg_rgbsrc(const YuvGArgs &a0, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (!frames || nframes < 1 || nframes > kYuv2xMaxFrames) return GMAT_ERR(EINVAL);
    if (yuvg_rgbsrc_block_form(a0, nframes)) return launch_scale_yuvg_rgbsrc_blk(a0, stream, *frames, nframes);
    if (!a0.K) return GMAT_ERR(ENOSYS);
    YuvGArgs a = a0;
    const char *rowsStr = GMAT_KNOB("GMAT_STRIP_ROWS");
    const int rowsEnv = rowsStr ? atoi(rowsStr) : 0;
    const int nstrips = (a.dstW + 63) / 64;
    a.nsg = (nstrips + 3) / 4;
    const long wr = (long)a.dstH * nstrips * nframes;
    const int rows = rowsEnv > 0 ? rowsEnv : (int)std::min(48L, std::max(4L, (wr + 6143) / 6144));     // (the band walker's rule for an RGB destination)
    const bool balanced = rowsEnv <= 0 && rows == 48 && a.dstH >= 96 && a.dstH < 32768;
    a.bandRows = rows;
    a.nbands = balanced ? (a.dstH + rows / 2) / rows : (a.dstH + rows - 1) / rows;
    a.bandStep = balanced ? (int)(((unsigned)a.dstH << 16) / (unsigned)a.nbands) : rows << 16;
    a.nblkL = a.nbands * a.nsg;
    a.nblk = a.nblkL;
    const dim3 grid(a.xcdRemap ? 8 * ((a.nblk + 7) / 8) : a.nblk, nframes), block(256);
    const Yuv2xFrames &fr = *frames;
    const bool half = a.chrSrcW != a.srcW;
#define GMAT_RS(P_, K_) do { if (half) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_rgbsrc_kernel<P_, K_, true>), grid, block, 0, stream, a, fr); \
                             else      hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvg_rgbsrc_kernel<P_, K_, false>), grid, block, 0, stream, a, fr); } while (0)
#define GMAT_RS_P(P_) do { switch (a.K) { case 4: GMAT_RS(P_, 4); break; case 6: GMAT_RS(P_, 6); break; case 7: GMAT_RS(P_, 7); break; default: GMAT_RS(P_, 9); } } while (0)
#define GMAT_RS_P12(P_) do { if (a.K == 12) GMAT_RS(P_, 12); else GMAT_RS_P(P_); } while (0)
    switch (a.P) { case 4: GMAT_RS_P12(4); break; case 5: GMAT_RS_P12(5); break; case 6: GMAT_RS_P12(6); break; case 7: GMAT_RS_P12(7); break; case 8: GMAT_RS_P12(8); break;
                   case 10: GMAT_RS_P(10); break; default: GMAT_RS_P(13); }
#undef GMAT_RS_P12
#undef GMAT_RS_P
#undef GMAT_RS
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}
#endif  // G_PART != 2
#endif

} // namespace gmat
