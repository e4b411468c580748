// k_scale_yuvu.hip — the QUAD-LANE polyphase walker for 8-bit YUV 4:2:0 sources (gfx950): UP-scales of any factor, and whatever else has
// short filters (horizontal taps <= 8, a vertical window of at most five row pairs: down-scales up to 2 : 1).
//
// libswscale's single-context semantics (swscale.c:234-520): every plane scaled separately — hScale8To15_c (swscale.c:122-136), the
// vertical filters of vscale.c / output.c, the LUT colour stage of yuv2rgb.c — bit-exact, on the tables initFilter (utils.c:367-763) made.
//
// The band walker (k_scale_yuvg.hip) gives a lane ONE output column and keeps K running sums, one per output row open at once; every
// source row pair costs K v_dot2 and every row that leaves costs a shift of the K sums.  That is the right shape for down-scales (few
// rows open, many source rows) and the wrong one for up-scales: 720p -> 1080p has 12-15 rows open for filters of FOUR taps, pays 12-15
// dot products per pair where a row needs three, 11-14 register moves per row that leaves, and stops at 1 : 2 (22 rows open is what its
// instances carry).  Measured: 4.4 us per 1080p NV12 frame, 0.125 of the HBM roofline (profiles/r04j_*).  Here
//   * a lane owns FOUR adjacent outputs of a row (luma: 4 columns; NV12's interleaved chroma: 2 columns x U, V), so the output leaves as
//     whole dwords (4:2:0 destinations) or as 12 / 16 bytes of packed RGB with one colour-table lookup per chroma sample — no lane
//     exchange, no quarter-filled stores;
//   * the vertical filter is a GATHER: the horizontally filtered row pairs of the last R steps sit in a register ring (newest first),
//     and an output row is R v_dot2 per sample against coefficient pairs the host lays out BY OUTPUT ROW relative to the newest pair at
//     the step that completes the row.  How many rows are open at once no longer matters: any up-scale factor, no K;
//   * the horizontal stage is the band walker's: a wave loads each source row's unique bytes once (one or two dwords a lane, four row
//     pairs ahead), a pair's bytes pass through a wave-private LDS row image, every output reads its own window there (static dword
//     indices, the byte phase in a per-output selector for v_perm_b32).
// Bands walk downward only: an up-scale's halo rows are a few per cent of its (already small) source traffic.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

// U_BPS: bytes per SOURCE sample.  This file compiles twice (round 5, as k_scale_yuvg.hip does): as it is for NV12 / YUV420P, and from k_scale_yuvu16.hip with
// U_BPS = 2 for P010LE / P016LE / YUV420P10LE / YUV420P16LE (hScale16To15_c in front of the same rings and output stages; namespace gmat::u16, entry points *16).
// With 16-bit samples the host re-bases every window to an 8-byte boundary of the row: a plane's coefficient pair is an aligned dword as loaded (no
// v_perm_b32), windows are read with ds_read_b64, the interleaved chroma runs on one pair fewer (k_scale_yuvg.hip).
#ifndef U_BPS
#define U_BPS 1
#endif
#if U_BPS == 2
#define U_NAME(n) n##16
#else
#define U_NAME(n) n
#endif

namespace gmat {
#if U_BPS == 2
namespace u16 {
#endif

// ---- a plane as a raw buffer resource (k_scale_yuvg.hip's GPlane with 12- and 16-byte stores): reads past the plane's last byte return
//      0, the stores are issued from inline assembly so that the compiler's wait counts see loads only (FINDINGS.md R3-walker)
struct UPlane {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    typedef unsigned v3u __attribute__((ext_vector_type(3)));
    __amdgpu_buffer_rsrc_t r;
    v4u words;
    __device__ __forceinline__ UPlane(const uint8_t *p, unsigned bytes) : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p), 0, bytes, 0x00020000))
    {
        // (readfirstlane: an "s" operand of the inline-assembly stores must not be a value the compiler chose to compute on the vector ALU: k_scale_yuvg.hip GPlane)
        const unsigned long long a = (unsigned long long)p;
        words = (v4u){(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu)),
                      (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
    }
    __device__ __forceinline__ unsigned ld1(unsigned lane, unsigned row) const { return __builtin_amdgcn_raw_buffer_load_b32(r, lane, row, 0); }
    __device__ __forceinline__ void st2(uint2 d, unsigned lane, unsigned row) const
    {
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        const v2u v = {d.x, d.y};
        asm volatile("s_nop 4\n\tbuffer_store_dwordx2 %0, %1, %2, %3 offen" :: "v"(v), "v"(lane), "s"(words), "s"(row) : "memory");
    }
    // s_nop 4: a scalar operand written by a VALU instruction may be read by a memory instruction five wait states later at the earliest
    // (tests/test_isa_guard.py); s_nop 0 behind the wide stores: their data registers must not be overwritten by the next VALU instruction
    __device__ __forceinline__ void st1(unsigned d, unsigned lane, unsigned row) const
    {
        asm volatile("s_nop 4\n\tbuffer_store_dword %0, %1, %2, %3 offen" :: "v"(d), "v"(lane), "s"(words), "s"(row) : "memory");
    }
    __device__ __forceinline__ void st3(uint3 d, unsigned lane, unsigned row) const
    {
        const v3u v = {d.x, d.y, d.z};
        asm volatile("s_nop 4\n\tbuffer_store_dwordx3 %0, %1, %2, %3 offen\n\ts_nop 0" :: "v"(v), "v"(lane), "s"(words), "s"(row) : "memory");
    }
    __device__ __forceinline__ void st4(uint4 d, unsigned lane, unsigned row) const
    {
        const v4u v = {d.x, d.y, d.z, d.w};
        asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 0" :: "v"(v), "v"(lane), "s"(words), "s"(row) : "memory");
    }
#else
    // hipcc's host pass (never executed) and the CPU emulation of the test suite
    uint8_t *p; unsigned n;
    __host__ __device__ UPlane(const uint8_t *q, unsigned bytes) : p(const_cast<uint8_t *>(q)), n(bytes) {}
    __host__ __device__ unsigned ld1(unsigned lane, unsigned row) const { unsigned v = 0; if ((size_t)row + lane + 4 <= n) std::memcpy(&v, p + (size_t)row + lane, 4); return v; }
    __host__ __device__ void st1(unsigned d, unsigned lane, unsigned row) const { if ((size_t)row + lane + 4 <= n) std::memcpy(p + (size_t)row + lane, &d, 4); }
    __host__ __device__ void st2(uint2 d, unsigned lane, unsigned row) const { if ((size_t)row + lane + 8 <= n) std::memcpy(p + (size_t)row + lane, &d, 8); }
    __host__ __device__ void st3(uint3 d, unsigned lane, unsigned row) const { if ((size_t)row + lane + 12 <= n) std::memcpy(p + (size_t)row + lane, &d, 12); }
    __host__ __device__ void st4(uint4 d, unsigned lane, unsigned row) const { if ((size_t)row + lane + 16 <= n) std::memcpy(p + (size_t)row + lane, &d, 16); }
#endif
};

__device__ __forceinline__ int u_dot2(int ab, int cd, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, ab), __builtin_bit_cast(short2v, cd), acc, true);
}
// two signed 16-bit halves -> two unsigned bytes with saturation in the low half (the upper half is never used: every caller selects
// bytes 0 and 1 with a v_perm_b32)
__device__ __forceinline__ unsigned u_sat_pk_u8_i16(unsigned v)
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

constexpr int kUPad = 4;                 // dwords behind a row image: the last dwords of a window may lie there (their taps are 0)
constexpr int kUStrip = 256;             // output bytes (4:2:0 destination) / pixels (RGB) of a row a wave owns: 64 lanes x 4

// One stream of a lane: NC output columns with their horizontal windows (a stride-1 component: byte pairs (o + 2t, o + 2t + 1), t < P,
// o = pos & 3; S2 = both components of NV12's interleaved row: (o + 4t, o + 4t + 2) for U and one byte further for V, o = 2 pos & 3),
// the wave's share of the row loads, a ring of RR requested row pairs with static slot names, the gathered windows of the pair consumed
// next.  NOUT horizontally filtered samples leave per pair: NC (stride 1) or 2 NC (S2: U, V of column 0, U, V of column 1).
// first byte of the window of output column `pos`, and of the aligned dwords read for it (16-bit samples: 8-byte boundaries)
template <bool S2> __device__ __host__ __forceinline__ int u_win_base(int pos) { return U_BPS == 2 ? ((S2 ? 4 * pos : 2 * pos) & ~7) : ((S2 ? 2 * pos : pos) & ~3); }
typedef unsigned short u_us2 __attribute__((ext_vector_type(2)));
template <int P, bool S2, int NC, int SD, int RR>
struct UStream {
#if U_BPS == 2
    static constexpr int NW = S2 ? 2 * P : (P + 1) & ~1;
    unsigned cvShr, cvFlip; int cvSh, cvBias;     // what the row image holds (k_scale_yuvg.hip g_conv), hScale16To15_c's shift, the sums' start
    __device__ __forceinline__ void set_conv(int kind, int sh, int bias) { cvShr = kind == 10 ? 6u : 0u; cvFlip = (kind == 10 || kind == 18) ? 0u : 0x80008000u; cvSh = sh; cvBias = bias; }
#else
    static constexpr int NW = S2 ? P + 1 : ((P - 1) >> 1) + 2;
    __device__ __forceinline__ void set_conv(int, int, int) {}
#endif
    static constexpr int NOUT = S2 ? 2 * NC : NC;
    static constexpr int IMG = 64 * SD + kUPad;
    int cf[NC][P];
    unsigned sel[NC];                    // stride 1: the selector of even pairs (odd pairs: + 0x00020002);  S2: U's (V's: + 0x00010001)
    int winDw[NC];                       // LDS dword index of a column's window inside a row image
    int ldDw[SD];                        // LDS dword index this lane fills, per sub-load
    unsigned ldOff[SD];                  // byte offset in the source row this lane loads, per sub-load
    unsigned ring[RR][2][SD];
    unsigned win[NC][2][NW];
    unsigned *img;                       // this wave's two row images of this stream
    int reqOff, rowStep, lastOff;        // (wave-uniform) byte offset of the next row to request; one row; the plane's last row

    // column c of this lane is output column `col` of the plane; returns the first byte of its window (dword aligned)
    __device__ __forceinline__ int setup_col(int c, const int32_t *hTab, const int32_t *posTab, int col)
    {
        const int pos = posTab[col];
#if U_BPS == 2
        sel[c] = 0u;
#else
        const int b0 = S2 ? 2 * pos : pos;
        const unsigned o = (unsigned)b0 & 3u;
        sel[c] = S2 ? (0x0C000C00u | o | ((o + 2) << 16)) : (0x0C000C00u | o | ((o + 1) << 16));
#endif
#pragma unroll
        for (int t = 0; t < P; t++) cf[c][t] = hTab[(size_t)col * P + t];
        return u_win_base<S2>(pos);
    }
    // the walk starts at row pair `pair` (it may lie in front of the plane: an RGB destination's luma stream runs behind its chroma stream)
    __device__ __forceinline__ void start(int pair, int stride, int rows)
    {
        rowStep = stride;
        reqOff = 2 * pair * stride;
        lastOff = (rows - 1) * stride;
    }
    // rows in front of the plane and behind it are requested as its first / last row: no tap falls on them (scalar clamps: the resource's
    // own range check is not relied on)
    template <class Ld> __device__ __forceinline__ void request(Ld &&ld, unsigned (&dst)[2][SD])
    {
        const unsigned o0 = (unsigned)min(max(reqOff, 0), lastOff), o1 = (unsigned)min(max(reqOff + rowStep, 0), lastOff);
#pragma unroll
        for (int s = 0; s < SD; s++) { dst[0][s] = ld(ldOff[s], o0); dst[1][s] = ld(ldOff[s], o1); }
        reqOff += 2 * rowStep;
    }
    // a row pair's bytes: registers -> row images -> every column's windows
    __device__ __forceinline__ void gather(const unsigned (&src)[2][SD])
    {
        __builtin_amdgcn_wave_barrier();         // (emulation: the lanes of a wave are fibers; on the GPU the LDS runs a wave's instructions in order)
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
#if U_BPS == 2
            for (int s = 0; s < SD; s++) {
                const u_us2 h = __builtin_bit_cast(u_us2, src[r][s]) >> (u_us2){(unsigned short)cvShr, (unsigned short)cvShr};
                img[r * IMG + ldDw[s]] = __builtin_bit_cast(unsigned, h) ^ cvFlip;
            }
#else
            for (int s = 0; s < SD; s++) img[r * IMG + ldDw[s]] = src[r][s];
#endif
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int r = 0; r < 2; r++)
#if U_BPS == 2
#pragma unroll
                for (int i = 0; i < NW; i += 2) {            // (window bases are 8-byte aligned, IMG is even: ds_read_b64)
                    const uint2 t = *reinterpret_cast<const uint2 *>(img + r * IMG + winDw[c] + i);
                    win[c][r][i] = t.x; win[c][r][i + 1] = t.y;
                }
#else
#pragma unroll
                for (int i = 0; i < NW; i++) win[c][r][i] = img[r * IMG + winDw[c] + i];
#endif
    }
    // pair 0 gathered, pairs 1 .. RR requested: pair k + 1 in ring slot (k + 1) % RR
    template <class Ld> __device__ __forceinline__ void prime(Ld &&ld)
    {
        unsigned first[2][SD];
        request(ld, first);
#pragma unroll
        for (int d = 1; d <= RR; d++) request(ld, ring[d % RR]);
        gather(first);
    }
    // the gathered pair's horizontally filtered samples, packed (row 2k | row 2k + 1 << 16): min(sum >> 7, 32767) each (hScale8To15_c)
    __device__ __forceinline__ void hpairs(int (&out)[NOUT]) const
    {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if constexpr (S2) {
#pragma unroll
                for (int comp = 0; comp < 2; comp++) {
                    int h[2];
#if U_BPS == 2
                    const unsigned sc = comp ? 0x07060302u : 0x05040100u;        // the component's half of the (U, V) dwords 2t and 2t + 1
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        int s = cvBias;
#pragma unroll
                        for (int t = 0; t < P; t++) s = u_dot2((int)__builtin_amdgcn_perm(win[c][r][2 * t + 1], win[c][r][2 * t], sc), cf[c][t], s);
                        h[r] = s >> cvSh;
                    }
#else
                    const unsigned sc = sel[c] + (comp ? 0x00010001u : 0u);
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        int s = 0;
#pragma unroll
                        for (int t = 0; t < P; t++) s = u_dot2((int)__builtin_amdgcn_perm(win[c][r][t + 1], win[c][r][t], sc), cf[c][t], s);
                        h[r] = s >> 7;
                    }
#endif
                    out[2 * c + comp] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(h[0], h[1]));
                }
            } else {
                int h[2];
#if U_BPS == 2
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    int s = cvBias;
#pragma unroll
                    for (int t = 0; t < P; t++) s = u_dot2((int)win[c][r][t], cf[c][t], s);      // an aligned dword as loaded IS the pair
                    h[r] = s >> cvSh;
                }
#else
                const unsigned se = sel[c], so = sel[c] + 0x00020002u;
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    int s = 0;
#pragma unroll
                    for (int t = 0; t < P; t++)
                        s = u_dot2((int)__builtin_amdgcn_perm(win[c][r][(t >> 1) + 1], win[c][r][t >> 1], (t & 1) ? so : se), cf[c][t], s);
                    h[r] = s >> 7;
                }
#endif
                out[c] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(h[0], h[1]));
            }
        }
    }
    // after a pair has been consumed: the next one (in ring slot S) becomes the gathered one, its slot is requested again RR pairs on
    template <int S, class Ld> __device__ __forceinline__ void advance(Ld &&ld)
    {
        gather(ring[S]);
        request(ld, ring[S]);
    }
};

// coefficient pairs of a stream in a kernel instantiated for P: 16-bit samples of an interleaved row lead with at most one position where a plane leads with
// three samples (k_scale_yuvg.hip GPairs)
template <int P, bool S2> struct UPairs { static constexpr int N = (U_BPS == 2 && S2) ? P - 1 : P; };

// R coefficient pairs of output row y (wave-uniform: scalar loads)
template <int R> __device__ __forceinline__ void u_load_row(const int32_t *vt, int y, int (&c)[R])
{
#pragma unroll
    for (int j = 0; j < R; j++) c[j] = uniform_load(vt, y * R + j);
}

// ---- 4:2:0 destinations: plane jobs -----------------------------------------------------------------------------------------------
// job 0: the luma plane (a lane: 4 columns).  job 1: chroma — NV12 -> NV12: a lane = 2 columns x (U, V) of the interleaved plane;
// planar -> planar: two jobs (U, V), a lane = 4 columns.  Blocks [0, nblkL) are luma, the rest chroma.  A step is a row pair of the job's
// plane; R = ring depth (row pairs an output row's window can touch).
template <int P, int R, int SD, bool NV12>
__global__ __launch_bounds__(256) void scale_yuvu_planes_kernel(YuvUArgs a, Yuv2xFrames fr)
{
    constexpr int IMG = 64 * SD + kUPad;
    __shared__ unsigned image[4][2 * IMG];                        // [wave][two row images]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lin = blockIdx.x;
    if (a.xcdRemap) { const int chunk = (a.nblk + 7) >> 3; lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3); }
    if (lin >= a.nblk) return;
    const int f = blockIdx.y;
    int job = 0, rel = lin;
    if (lin >= a.nblkL) { rel = lin - a.nblkL; job = 1; if (!NV12 && rel >= a.nblkC) { rel -= a.nblkC; job = 2; } }
    const int nsg = job ? a.nsgC : a.nsg;
    const int band = __builtin_amdgcn_readfirstlane(rel / nsg);
    const int B0 = ((rel - band * nsg) * 4 + wave) * kUStrip;       // first BYTE column of this wave in the destination plane row
    const int rowBytes = job == 0 ? a.dstW : NV12 ? 2 * a.chrDstW : a.chrDstW;
    if (B0 >= rowBytes) return;
    const int rows = job ? a.chrDstH : a.dstH, srcRows = job ? a.chrSrcH : a.srcH;
    const int bandRows = job ? a.bandRowsC : a.bandRows;
    const int ya = band * bandRows, yb = min(ya + bandRows, rows);
    const uint8_t *sp = job == 0 ? fr.y[f] : job == 1 ? fr.u[f] : fr.v[f];
    uint8_t *dp = job == 0 ? fr.dst[f] : job == 1 ? fr.dstU[f] : fr.dstV[f];
    const int ss = job == 0 ? a.ys : job == 1 ? a.us : a.vs, dstride = job == 0 ? a.ds : job == 1 ? a.dsU : a.dsV;
    const int srcRowBytes = (job == 0 ? a.srcW : NV12 ? 2 * a.chrSrcW : a.chrSrcW) * U_BPS;
    const UPlane bS(sp, (unsigned)ss * (unsigned)(srcRows - 1) + (unsigned)srcRowBytes), bD(dp, (unsigned)dstride * (unsigned)(rows - 1) + (unsigned)rowBytes * (a.dst16 ? 2u : 1u));
    const int32_t *vt = job ? a.vtC : a.vtL, *endT = job ? a.endC : a.endL, *firstT = job ? a.firstC : a.firstL, *lastT = job ? a.lastC : a.lastL;
    const int rnd = job ? a.roundC : a.roundL;
    const int b0 = B0 + 4 * lane;                                   // this lane's first byte column

    auto run = [&](auto s2_c) {
        constexpr bool S2 = decltype(s2_c)::value;
        constexpr int NC = S2 ? 2 : 4;
        UStream<UPairs<P, S2>::N, S2, NC, SD, 4> W;
        W.set_conv(a.src16, a.hShift, a.hBias);
        {
            const int32_t *hTab = job ? a.hC : a.hL, *posTab = job ? a.posC : a.posL;
            const int ncols = S2 ? a.chrDstW : rowBytes;
            int w0[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) w0[c] = W.setup_col(c, hTab, posTab, min((S2 ? b0 >> 1 : b0) + c, ncols - 1));
            const int seg = __builtin_amdgcn_readfirstlane(w0[0]);     // the wave's row segment starts at lane 0's first window
#pragma unroll
            for (int c = 0; c < NC; c++) W.winDw[c] = (w0[c] - seg) >> 2;
#pragma unroll
            for (int s = 0; s < SD; s++) { W.ldDw[s] = lane + 64 * s; W.ldOff[s] = (unsigned)seg + 4u * (unsigned)(lane + 64 * s); }
            W.img = image[wave];
        }
        auto ld = [&](unsigned off, unsigned row) { return bS.ld1(off, row); };
        const int pa = uniform_load(firstT, ya), pb = uniform_load(lastT, yb - 1);
        W.start(pa, ss, srcRows);
        W.prime(ld);
        int hr[R][4];                                               // the ring: hr[j] = the pair j steps before the newest
#pragma unroll
        for (int j = 0; j < R; j++)
#pragma unroll
            for (int o = 0; o < 4; o++) hr[j][o] = 0;
        int y = ya;
        // coefficient pairs of the next row to leave: requested as soon as the previous row's sums are done, so that the scalar loads'
        // way is covered by that row's packing and store (one set of scalar registers, redefined in place: no copies)
        int cv[R];
        u_load_row<R>(vt, y, cv);

        auto step = [&](int p, auto slot_c) {
            constexpr int SLOT = decltype(slot_c)::value;
            const int ye = min(uniform_load(endT, p), yb);         // output rows complete after this pair
            int hp[4];
            W.hpairs(hp);
            W.template advance<SLOT>(ld);                          // (the gather's LDS round trip is covered by the rows leaving below)
#pragma unroll
            for (int j = R - 1; j > 0; j--)
#pragma unroll
                for (int o = 0; o < 4; o++) hr[j][o] = hr[j - 1][o];
#pragma unroll
            for (int o = 0; o < 4; o++) hr[0][o] = hp[o];
            while (y < ye) {
                // yuv2planeX_8_c / yuv2nv12cX_c: clip_u8((dither << 12 + sum) >> 19), the dither in the sums' start value
                int acc[4];
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    acc[o] = rnd;
#pragma unroll
                    for (int j = 0; j < R; j++) acc[o] = u_dot2(hr[j][o], cv[j], acc[o]);
                }
                u_load_row<R>(vt, min(y + 1, rows - 1), cv);
                const unsigned drow = (unsigned)y * (unsigned)dstride;
                if (__builtin_amdgcn_readfirstlane(a.dst16)) {
                    // yuv2p010l1_c / lX_c / cX_c, yuv2planeX_10_c (output.c:459-519): clip_uintp2((1 << 16 + sum) >> 17, 10), P010: << 6 — four samples = two dwords
                    unsigned w[4];
#pragma unroll
                    for (int o = 0; o < 4; o++) w[o] = (unsigned)min(max(acc[o] >> 17, 0), 1023) << a.dstShift;
                    if (b0 + 4 <= rowBytes) bD.st2(make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16)), 2u * (unsigned)b0, drow);
                    else if (b0 < rowBytes) {
                        unsigned short *q = reinterpret_cast<unsigned short *>(dp + (size_t)drow) + b0;
                        for (int i = 0; i < rowBytes - b0; i++) q[i] = (unsigned short)w[i];
                    }
                    y++;
                    continue;
                }
                if (U_BPS == 2 && __builtin_amdgcn_readfirstlane(a.dither8)) {
                    // 8-bit planar output of a deeper source: ff_dither_8x8_128 (swscale.c:263-264, 482-485) on top of the 64 << 12 the sums started at; the dither's
                    // column: a sample's own, V three columns on (vscale.c:98,101, output.c:433-434)
#pragma unroll
                    for (int o = 0; o < 4; o++) {
                        const int col = job == 0 ? b0 + o : NV12 ? ((b0 + o) >> 1) + 3 * ((b0 + o) & 1) : b0 + o + (job == 2 ? 3 : 0);
                        acc[o] += dither_delta(col, y);
                    }
                }
                const unsigned v0 = (unsigned)clip_u8_shr(acc[0], 19), v1 = (unsigned)clip_u8_shr(acc[1], 19);
                const unsigned v2 = (unsigned)clip_u8_shr(acc[2], 19), v3 = (unsigned)clip_u8_shr(acc[3], 19);
                const unsigned d = v0 | (v1 << 8) | (v2 << 16) | (v3 << 24);
                if (b0 + 4 <= rowBytes) bD.st1(d, (unsigned)b0, drow);
                else if (b0 < rowBytes) {                            // a row that is not whole dwords: its last bytes one by one
                    uint8_t *q = dp + (size_t)drow + (unsigned)b0;
                    for (int i = 0; i < rowBytes - b0; i++) q[i] = (uint8_t)(d >> (8 * i));
                }
                y++;
            }
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        for (int p = pa; p <= pb; p += 4) {
            step(p, I1());
            if (p + 1 > pb) break;
            step(p + 1, I2());
            if (p + 2 > pb) break;
            step(p + 2, I3());
            if (p + 3 > pb) break;
            step(p + 3, I0());
        }
    };
    if (NV12 && job == 1) run(std::true_type()); else run(std::false_type());
}

// ---- packed RGB destinations -----------------------------------------------------------------------------------------------------------
// block = 4 waves = 4 adjacent strips of 256 pixels of one band; grid.y = frame.  A lane: 4 pixels = 4 luma columns + the 2 chroma
// columns under them (RGB destinations keep the chroma plane at half the output width, one sample a pixel pair: yuv2rgb_X_c_template).
// A step: two luma row pairs and one chroma row pair (below); RL / RC = ring depths, counted from the step's newest pair of each stream.
template <int P, int RL, int RC, int SD, bool NV12>
__global__ __launch_bounds__(256) void scale_yuvu_rgb_kernel(YuvUArgs a, Yuv2xFrames fr)
{
    constexpr int IMG = 64 * SD + kUPad, RV = RL + RC;
    __shared__ int2 lutV[256], lutU[256];
    __shared__ unsigned image[4][2][2 * IMG];                     // [wave][luma | chroma][two row images]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        // chan = sat_u8(high half of (term + Y * cy)), term_R = lutV[V].x, term_G = lutV[V].y + lutU[U].x, term_B = lutU[U].y
        // (px_math.h chroma_terms split by sample: yuv2rgb.c's table_rV / gU / gV / bU in closed form)
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
    const int X0 = ((lin - band * a.nsg) * 4 + wave) * kUStrip;
    if (X0 >= a.dstW) return;
    const int ya = band * a.bandRows, yb = min(ya + a.bandRows, a.dstH);
    const int f = blockIdx.y;
    const unsigned crb = (unsigned)(NV12 ? 2 * a.chrSrcW : a.chrSrcW) * U_BPS;
    const UPlane bY(fr.y[f], (unsigned)a.ys * (unsigned)(a.srcH - 1) + (unsigned)a.srcW * U_BPS);
    const UPlane bU(fr.u[f], (unsigned)a.us * (unsigned)(a.chrSrcH - 1) + crb), bV(NV12 ? fr.u[f] : fr.v[f], (unsigned)a.us * (unsigned)(a.chrSrcH - 1) + crb);
    const int bpp = (a.dstFormat == GMAT_PIX_FMT_RGBA || a.dstFormat == GMAT_PIX_FMT_BGRA) ? 4 : 3;
    const bool bgr = a.dstFormat == GMAT_PIX_FMT_BGR24 || a.dstFormat == GMAT_PIX_FMT_BGRA;
    const UPlane bD(fr.dst[f], (unsigned)a.ds * (unsigned)(a.dstH - 1) + (unsigned)(a.dstW * bpp));

    const int x0 = X0 + 4 * lane;                                  // this lane's first pixel
    UStream<P, false, 4, SD, 4> L;
    // chroma: NV12: 2 columns x (U, V) of the interleaved row.  Planar: 4 "columns" = (U, column 0), (U, 1), (V, 0), (V, 1); the U segment
    // in the first half of the row image (filled by lanes 0 .. 31), the V segment in the second (lanes 32 .. 63)
    UStream<UPairs<P, NV12>::N, NV12, NV12 ? 2 : 4, SD, 2> C;
    L.set_conv(a.src16, a.hShift, a.hBias); C.set_conv(a.src16, a.hShift, a.hBias);
    {
        int w0[4];
#pragma unroll
        for (int c = 0; c < 4; c++) w0[c] = L.setup_col(c, a.hL, a.posL, min(x0 + c, a.dstW - 1));
        const int seg = __builtin_amdgcn_readfirstlane(w0[0]);
#pragma unroll
        for (int c = 0; c < 4; c++) L.winDw[c] = (w0[c] - seg) >> 2;
#pragma unroll
        for (int s = 0; s < SD; s++) { L.ldDw[s] = lane + 64 * s; L.ldOff[s] = (unsigned)seg + 4u * (unsigned)(lane + 64 * s); }
        L.img = image[wave][0];
    }
    {
        const int cc0 = x0 >> 1;
        if constexpr (NV12) {
            int w0[2];
#pragma unroll
            for (int c = 0; c < 2; c++) w0[c] = C.setup_col(c, a.hC, a.posC, min(cc0 + c, a.chrDstW - 1));
            const int seg = __builtin_amdgcn_readfirstlane(w0[0]);
#pragma unroll
            for (int c = 0; c < 2; c++) C.winDw[c] = (w0[c] - seg) >> 2;
#pragma unroll
            for (int s = 0; s < SD; s++) { C.ldDw[s] = lane + 64 * s; C.ldOff[s] = (unsigned)seg + 4u * (unsigned)(lane + 64 * s); }
        } else {
            int w0[4];
#pragma unroll
            for (int c = 0; c < 4; c++) w0[c] = C.setup_col(c, a.hC, a.posC, min(cc0 + (c & 1), a.chrDstW - 1));
            const int seg = __builtin_amdgcn_readfirstlane(w0[0]);
#pragma unroll
            for (int c = 0; c < 4; c++) C.winDw[c] = (c >> 1) * 32 * SD + ((w0[c] - seg) >> 2);
#pragma unroll
            for (int s = 0; s < SD; s++) { const int j = (lane & 31) + 32 * s; C.ldDw[s] = (lane >> 5) * 32 * SD + j; C.ldOff[s] = (unsigned)seg + 4u * (unsigned)j; }
        }
        C.img = image[wave][1];
    }
    auto ldL = [&](unsigned off, unsigned row) { return bY.ld1(off, row); };
    auto ldC = [&](unsigned off, unsigned row) {
        unsigned v;
        if (NV12 || lane < 32) v = bU.ld1(off, row); else v = bV.ld1(off, row);      // only the load diverges (planar: us == vs, host rule)
        return v;
    };
    // a step s: luma row pairs 2 (s - lead), 2 (s - lead) + 1 and chroma row pair s.  lead: the luma stream runs that many steps behind the
    // chroma stream (the host's choice, 0 .. 2): the chroma of an RGB destination is up-scaled twice as far as its luma, its windows end
    // later, and with the streams side by side every luma window would have to be kept a step longer (deeper rings, more dot products a
    // row).  Luma pairs in front of the plane (a band at the frame's top) read as 0 and carry no taps.
    const int q0 = uniform_load(a.firstL, ya), q1 = uniform_load(a.lastL, yb - 1);
    L.start(2 * (q0 - a.lead), a.ys, a.srcH);
    C.start(q0, a.us, a.chrSrcH);
    L.prime(ldL);
    C.prime(ldC);
    int hL[RL][4], hC[RC][4];                                      // the rings, newest pair first
#pragma unroll
    for (int j = 0; j < RL; j++)
#pragma unroll
        for (int o = 0; o < 4; o++) hL[j][o] = 0;
#pragma unroll
    for (int j = 0; j < RC; j++)
#pragma unroll
        for (int o = 0; o < 4; o++) hC[j][o] = 0;
    int y = ya;
    // RL luma + RC chroma coefficient pairs of the next row to leave: requested as soon as the previous row's sums are done, so that the
    // scalar loads' way is covered by that row's colour stage (one set of scalar registers, redefined in place: no copies)
    int cv[RV];
    u_load_row<RV>(a.vtL, y, cv);
    const int npx = a.dstW - x0;                                   // pixels this lane really has (>= 4: all four)

    // two channels -> bytes (0, 1) of a register: the high halves side by side, then the saturating pack
#define U_SAT2(x, y) u_sat_pk_u8_i16(__builtin_amdgcn_perm((y), (x), 0x07060302u))
#define U_JOIN(lo2, hi2) __builtin_amdgcn_perm((hi2), (lo2), 0x05040100u)
    // one step.  SA / SB / SC: the ring slots holding the pairs that follow (static: steps alternate between two sets of slots).  A
    // gather's LDS round trip is covered by work that does not need it: the other stream's horizontal filter, the rows leaving.
    auto quad = [&](int q, auto sa_c, auto sb_c, auto sc_c) {
        constexpr int SA = decltype(sa_c)::value, SB = decltype(sb_c)::value, SC = decltype(sc_c)::value;
        const int ye = min(uniform_load(a.endL, q), yb);
        int hp0[4], hpc[4], hp1[4];
        L.hpairs(hp0);
        L.template advance<SA>(ldL);
        C.hpairs(hpc);
        C.template advance<SC>(ldC);
        L.hpairs(hp1);
        L.template advance<SB>(ldL);
#pragma unroll
        for (int j = RL - 1; j > 1; j--)
#pragma unroll
            for (int o = 0; o < 4; o++) hL[j][o] = hL[j - 2][o];
#pragma unroll
        for (int o = 0; o < 4; o++) { hL[1][o] = hp0[o]; hL[0][o] = hp1[o]; }
#pragma unroll
        for (int j = RC - 1; j > 0; j--)
#pragma unroll
            for (int o = 0; o < 4; o++) hC[j][o] = hC[j - 1][o];
#pragma unroll
        for (int o = 0; o < 4; o++) hC[0][o] = hpc[o];
        while (y < ye) {
            int aL[4], aC[4];
#pragma unroll
            for (int o = 0; o < 4; o++) {
                aL[o] = a.roundL; aC[o] = a.roundC;
#pragma unroll
                for (int j = 0; j < RL; j++) aL[o] = u_dot2(hL[j][o], cv[j], aL[o]);
#pragma unroll
                for (int j = 0; j < RC; j++) aC[o] = u_dot2(hC[j][o], cv[RL + j], aC[o]);
            }
            u_load_row<RV>(a.vtL, min(y + 1, a.dstH - 1), cv);
            // chroma samples: NV12: (U, V) of column 0, then of column 1;  planar: U of columns 0, 1, then V
            int iU[2], iV[2];
            iU[0] = clip_u8_shr(aC[0], 19);
            iV[0] = clip_u8_shr(aC[NV12 ? 1 : 2], 19);
            iU[1] = clip_u8_shr(aC[NV12 ? 2 : 1], 19);
            iV[1] = clip_u8_shr(aC[3], 19);
            unsigned k0[4], k1[4], k2[4];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int2 tv = lutV[iV[h]], tu = lutU[iU[h]];
                const int tr = bgr ? tu.y : tv.x, tg = tv.y + tu.x, tb = bgr ? tv.x : tu.y;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int q4 = 2 * h + e;
                    const int ycy = m24(aL[q4] >> 19, a.y2r.cy);
                    k0[q4] = (unsigned)(tr + ycy); k1[q4] = (unsigned)(tg + ycy); k2[q4] = (unsigned)(tb + ycy);   // |term + Y cy| < 2^27: the channel is sat_u8 of the high half
                }
            }
            const unsigned drow = (unsigned)y * (unsigned)a.ds;
            if (bpp == 4) {
                uint4 o4;
                o4.x = U_JOIN(U_SAT2(k0[0], k1[0]), U_SAT2(k2[0], 0x00FF0000u));
                o4.y = U_JOIN(U_SAT2(k0[1], k1[1]), U_SAT2(k2[1], 0x00FF0000u));
                o4.z = U_JOIN(U_SAT2(k0[2], k1[2]), U_SAT2(k2[2], 0x00FF0000u));
                o4.w = U_JOIN(U_SAT2(k0[3], k1[3]), U_SAT2(k2[3], 0x00FF0000u));
                if (npx >= 4) bD.st4(o4, 4u * (unsigned)x0, drow);
                else if (npx > 0) {
                    const unsigned w[4] = {o4.x, o4.y, o4.z, o4.w};
                    uint8_t *d = fr.dst[f] + (size_t)drow + 4u * (unsigned)x0;
                    for (int i = 0; i < 4 * npx; i++) d[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
                }
            } else {
                uint3 o3;           // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
                o3.x = U_JOIN(U_SAT2(k0[0], k1[0]), U_SAT2(k2[0], k0[1]));
                o3.y = U_JOIN(U_SAT2(k1[1], k2[1]), U_SAT2(k0[2], k1[2]));
                o3.z = U_JOIN(U_SAT2(k2[2], k0[3]), U_SAT2(k1[3], k2[3]));
                if (npx >= 4) bD.st3(o3, 3u * (unsigned)x0, drow);
                else if (npx > 0) {
                    const unsigned w[3] = {o3.x, o3.y, o3.z};
                    uint8_t *d = fr.dst[f] + (size_t)drow + 3u * (unsigned)x0;
                    for (int i = 0; i < 3 * npx; i++) d[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
                }
            }
            y++;
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    for (int q = q0; q <= q1; q += 2) {
        quad(q, I1(), I2(), I1());
        if (q + 1 > q1) break;
        quad(q + 1, I3(), I0(), I0());
    }
#undef U_SAT2
#undef U_JOIN
}

#if U_BPS == 2
} // namespace u16
using namespace u16;
#endif

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// ring depths the kernels are instantiated for: 4:2:0 destinations R (4-tap filters need 3, 8-tap ones 5), RGB (RL, RC)
#if U_BPS == 2
static const int kUP[] = {4, 6};              // (16-bit samples: up to three leading zero taps in front of the 2 / 4 / 6 / 8 taps)
#else
static const int kUP[] = {2, 3, 4};
#endif
static const int kUR[] = {3, 5};
static const int kURgb[][2] = {{4, 3}, {4, 4}, {6, 4}, {8, 5}};      // (bilinear: 4 / 3; bicubic: 4 / 4 with the luma stream a step behind; Lanczos-3: 6 / 4)

int U_NAME(yuvu_prepare)(const ScalePlan &p, const YuvScaleTiling &g, YuvUTables &t)
{
    t = YuvUTables();
    const char *off = GMAT_KNOB("GMAT_SCALE_NO_QUAD_WALKER");
    if (off && atoi(off)) return 0;
    const bool rgbOut = p.dstFormat == GMAT_PIX_FMT_RGB24 || p.dstFormat == GMAT_PIX_FMT_BGR24 || p.dstFormat == GMAT_PIX_FMT_RGBA || p.dstFormat == GMAT_PIX_FMT_BGRA;
    // 4:2:0 destinations: 8 bits, or (round 5) 10 bits in 16-bit stores (P010LE / YUV420P10LE: the same 15-bit lines, another shift)
    const bool yuvOut = p.dstFormat == GMAT_PIX_FMT_NV12 || p.dstFormat == GMAT_PIX_FMT_YUV420P || is_dst10(p.dstFormat);
    const bool semiDst = p.dstFormat == GMAT_PIX_FMT_NV12 || p.dstFormat == GMAT_PIX_FMT_P010LE;
#if U_BPS == 2
    const bool nv12 = is_p01x(p.srcFormat);                                                 // (interleaved chroma)
    if (!(nv12 || p.srcFormat == GMAT_PIX_FMT_YUV420P10LE || p.srcFormat == GMAT_PIX_FMT_YUV420P16LE) || !(rgbOut || yuvOut)) return 0;
    if (const char *o16 = GMAT_KNOB("GMAT_SCALE_NO_WALKER16")) if (atoi(o16)) return 0;
    for (const FilterBank *fb : {&p.hLum, &p.hChr})                                         // (the 16-bit image's bias: every row sums to 16384)
        for (int x = 0; x < fb->count; x++) {
            int sum = 0;
            for (int j = 0; j < fb->taps; j++) sum += fb->coef[(size_t)x * fb->taps + j];
            if (sum != 16384) return 0;
        }
#else
    const bool nv12 = p.srcFormat == GMAT_PIX_FMT_NV12;
    if (!(nv12 || p.srcFormat == GMAT_PIX_FMT_YUV420P) || !(rgbOut || yuvOut)) return 0;
#endif
    if (rgbOut && (g.fullChroma || g.yuvOut)) return 0;
    if (yuvOut && g.yuvOut != 1) return 0;
    if (yuvOut && nv12 != semiDst) return 0;                                                // same chroma layout on both sides
    if (p.dstW < 16 || p.dstH < 8 || p.srcW < 16 || p.srcH < 8) return 0;
    // whole dwords inside every source row (the rows are dword loads checked against the plane's exact size)
    if ((p.srcW * U_BPS) % 4 || ((nv12 ? 2 * p.chrSrcW : p.chrSrcW) * U_BPS) % 4) return 0;
    if (rgbOut && (p.chrDstW != (p.dstW + 1) / 2 || p.chrDstH != p.dstH || p.dstW % 2)) return 0;
    if (yuvOut && (p.chrDstW != (p.dstW + 1) / 2 || p.chrDstH != (p.dstH + 1) / 2)) return 0;
    // the sums start at ONE value per plane class (the 1- and 2-tap special forms of vscale.c:135-167 have per-row starts)
    for (int v : g.lumRound) if (v != g.lumRound[0]) return 0;
    for (int v : g.chrRound) if (v != g.chrRound[0]) return 0;
    t.roundL = g.lumRound[0]; t.roundC = g.chrRound[0];
    if (g.vLumEff.count != p.dstH || g.vChrEff.count != (rgbOut ? p.dstH : p.chrDstH)) return 0;

    // ---- horizontal: coefficient pairs on the table's own windows -------------------------------------------------------------------
    // (16-bit samples: windows re-based to 8-byte boundaries — `lead` zero taps in front: pos & 3 samples of a plane, pos & 1 positions of an interleaved row —
    // and the interleaved chroma stream on one pair fewer than the instance's P: UPairs)
    auto lead = [&](const FilterBank &fb, int x, bool s2) { return U_BPS == 2 ? (s2 ? fb.pos[x] & 1 : fb.pos[x] & 3) : 0; };
    auto pairs_needed = [&](const FilterBank &fb, bool s2) { int m = 0; for (int x = 0; x < fb.count; x++) m = std::max(m, lead(fb, x, s2) + fb.taps); return (m + 1) / 2; };
    const int needP = std::max(pairs_needed(p.hLum, false), pairs_needed(p.hChr, nv12) + (U_BPS == 2 && nv12 ? 1 : 0));
    int P = 0;
    for (int c : kUP) if (c >= needP) { P = c; break; }
    if (!P) return 0;
    auto hpack = [&](const FilterBank &fb, int srcLen, std::vector<int32_t> &out, int PP, bool s2) {
        out.assign((size_t)fb.count * PP, 0);
        for (int x = 0; x < fb.count; x++) {
            if (fb.pos[x] < 0 || fb.pos[x] + fb.taps > srcLen) return false;
            if (x && fb.pos[x] < fb.pos[x - 1]) return false;                               // lane 0 holds a wave's first window
            const int ld = lead(fb, x, s2);
            for (int k = 0; k < PP; k++) {
                const int t0 = 2 * k - ld, t1 = t0 + 1;
                const int lo = t0 >= 0 && t0 < fb.taps ? fb.coef[(size_t)x * fb.taps + t0] : 0, hi = t1 >= 0 && t1 < fb.taps ? fb.coef[(size_t)x * fb.taps + t1] : 0;
                out[(size_t)x * PP + k] = (int32_t)((uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16));
            }
        }
        return true;
    };
    if (p.hLum.count != p.dstW || p.hChr.count != p.chrDstW) return 0;
    if (!hpack(p.hLum, p.srcW, t.hL, P, false) || !hpack(p.hChr, p.chrSrcW, t.hC, U_BPS == 2 && nv12 ? P - 1 : P, nv12)) return 0;
    t.posL = p.hLum.pos; t.posC = p.hChr.pos;
    // the row segments: the windows of a wave's columns lie inside the capDw dwords its lanes load (+ the pad for dwords whose taps are 0)
    auto fits = [&](const FilterBank &fb, int cols, bool s2, int capDw) {
        const int NW = U_BPS == 2 ? (s2 ? 2 * (P - 1) : (P + 1) & ~1) : (s2 ? P + 1 : ((P - 1) >> 1) + 2);
        for (int c0 = 0; c0 < fb.count; c0 += cols) {
            const int c1 = std::min(c0 + cols, fb.count) - 1;
            const int b0 = s2 ? u_win_base<true>(fb.pos[c0]) : u_win_base<false>(fb.pos[c0]), b1 = s2 ? u_win_base<true>(fb.pos[c1]) : u_win_base<false>(fb.pos[c1]);
            if (((b1 - b0) >> 2) + NW > capDw + kUPad) return false;
            const int lastSample = fb.pos[c1] + fb.taps - 1;                                                  // the last byte with a real tap
            const int lastTap = U_BPS == 2 ? (s2 ? 4 * lastSample + 3 : 2 * lastSample + 1) : (s2 ? 2 * lastSample + 1 : lastSample);
            if (lastTap >= b0 + 4 * capDw) return false;
        }
        return true;
    };
    int SD = 0;
    for (int sd : {1 * U_BPS, 2 * U_BPS}) {
        bool ok = fits(p.hLum, kUStrip, false, 64 * sd);
        if (rgbOut) ok = ok && fits(p.hChr, kUStrip / 2, nv12, nv12 ? 64 * sd : 32 * sd);
        else        ok = ok && fits(p.hChr, nv12 ? kUStrip / 2 : kUStrip, nv12, 64 * sd);
        if (ok) { SD = sd; break; }
    }
    if (!SD) return 0;

    // ---- vertical: rings and coefficient pairs by output row ---------------------------------------------------------------------------
    auto pk = [](int lo, int hi) { return (int32_t)((uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16)); };
    auto tap = [](const FilterBank &fb, int y, int srcRow) {
        const int k = srcRow - fb.pos[y];
        return k >= 0 && k < fb.taps ? (int)fb.coef[(size_t)y * fb.taps + k] : 0;
    };
    auto monotone = [](const FilterBank &fb, int srcRows) {
        for (int y = 0; y < fb.count; y++) {
            if (fb.pos[y] < 0 || fb.pos[y] + fb.taps > srcRows) return false;
            if (y && fb.pos[y] < fb.pos[y - 1]) return false;
        }
        return true;
    };
    if (!monotone(g.vLumEff, p.srcH) || !monotone(g.vChrEff, p.chrSrcH)) return 0;
    if (yuvOut) {
        // a step = a row pair of the plane
        auto plane = [&](const FilterBank &fb, std::vector<int32_t> &first, std::vector<int32_t> &last, int &need) {
            first.resize(fb.count); last.resize(fb.count); need = 1;
            for (int y = 0; y < fb.count; y++) {
                first[y] = fb.pos[y] >> 1; last[y] = (fb.pos[y] + fb.taps - 1) >> 1;
                need = std::max(need, last[y] - first[y] + 1);
            }
        };
        int needL = 0, needC = 0;
        plane(g.vLumEff, t.firstL, t.lastL, needL);
        plane(g.vChrEff, t.firstC, t.lastC, needC);
        int R = 0;
        for (int c : kUR) if (c >= std::max(needL, needC)) { R = c; break; }
        if (!R) return 0;
        auto fill = [&](const FilterBank &fb, const std::vector<int32_t> &last, std::vector<int32_t> &vt, std::vector<int32_t> &end) {
            vt.assign((size_t)fb.count * R, 0);
            const int steps = last.back() + 1;
            end.assign(steps + 1, 0);
            for (int y = 0; y < fb.count; y++) {
                for (int j = 0; j < R; j++) { const int pp = last[y] - j; vt[(size_t)y * R + j] = pk(tap(fb, y, 2 * pp), tap(fb, y, 2 * pp + 1)); }
                end[last[y]] = y + 1;
            }
            for (int s = 1; s <= steps; s++) end[s] = std::max(end[s], end[s - 1]);
        };
        fill(g.vLumEff, t.lastL, t.vtL, t.endL);
        fill(g.vChrEff, t.lastC, t.vtC, t.endC);
        t.RL = t.RC = R;
    } else {
        // a step s = luma row pairs 2 (s - lead), 2 (s - lead) + 1 and chroma row pair s.  lead = 0 .. 2, whichever needs the shallowest rings
        const FilterBank &fl = g.vLumEff, &fc = g.vChrEff;
        const int rows = p.dstH;
        int RL = 0, RC = 0, lead = 0;
        std::vector<int32_t> first(rows), last(rows);
        for (int ld = 0; ld <= 2; ld++) {
            int needL = 1, needC = 1;
            bool ok = true;
            for (int y = 0; y < rows && ok; y++) {
                first[y] = std::min((fl.pos[y] >> 2) + ld, fc.pos[y] >> 1);
                last[y] = std::max(((fl.pos[y] + fl.taps - 1) >> 2) + ld, (fc.pos[y] + fc.taps - 1) >> 1);
                if (y && (last[y] < last[y - 1] || first[y] < first[y - 1])) ok = false;
                needL = std::max(needL, 2 * (last[y] - ld) + 1 - (fl.pos[y] >> 1) + 1);
                needC = std::max(needC, last[y] - (fc.pos[y] >> 1) + 1);
            }
            if (GMAT_KNOB("GMAT_DEBUG_WALKER")) logf(LOG_ERROR, "yuvu: lead %d: rings needed %d / %d%s", ld, needL, needC, ok ? "" : " (not monotone)");
            if (!ok) continue;
            for (const auto &rc : kURgb)
                if (rc[0] >= needL && rc[1] >= needC) {
                    if (!RL || rc[0] + rc[1] < RL + RC) { RL = rc[0]; RC = rc[1]; lead = ld; t.firstL = first; t.lastL = last; }
                    break;
                }
        }
        if (!RL) { if (GMAT_KNOB("GMAT_DEBUG_WALKER")) logf(LOG_ERROR, "yuvu: %dx%d -> %dx%d declined: rings", p.srcW, p.srcH, p.dstW, p.dstH); return 0; }
        const int RV = RL + RC;
        t.vtL.assign((size_t)rows * RV, 0);
        const int steps = t.lastL.back() + 1;
        t.endL.assign(steps + 1, 0);
        for (int y = 0; y < rows; y++) {
            const int q = t.lastL[y];
            for (int j = 0; j < RL; j++) { const int pp = 2 * (q - lead) + 1 - j; t.vtL[(size_t)y * RV + j] = pk(tap(fl, y, 2 * pp), tap(fl, y, 2 * pp + 1)); }
            for (int j = 0; j < RC; j++) { const int pp = q - j; t.vtL[(size_t)y * RV + RL + j] = pk(tap(fc, y, 2 * pp), tap(fc, y, 2 * pp + 1)); }
            t.endL[q] = y + 1;
        }
        for (int s = 1; s <= steps; s++) t.endL[s] = std::max(t.endL[s], t.endL[s - 1]);
        t.lead = lead;
        t.RL = RL; t.RC = RC;
        // (the 4:2:0 tables are not used: one-element placeholders keep the uploads uniform)
        t.vtC.assign(1, 0); t.endC.assign(1, 0); t.firstC.assign(1, 0); t.lastC.assign(1, 0);
    }
    t.P = P; t.SD = SD; t.yuvOut = yuvOut;
    if (GMAT_KNOB("GMAT_DEBUG_WALKER")) logf(LOG_ERROR, "yuvu: %dx%d -> %dx%d taps h %d/%d v %d/%d -> P %d, SD %d, rings %d / %d, lead %d", p.srcW, p.srcH, p.dstW, p.dstH,
                                          p.hLum.taps, p.hChr.taps, g.vLumEff.taps, g.vChrEff.taps, P, SD, t.RL, t.RC, t.lead);
    t.ok = 1;
    return 0;
}

int U_NAME(launch_scale_yuvu)(const YuvUArgs &a0, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (!frames || nframes < 1 || nframes > kYuv2xMaxFrames) return GMAT_ERR(EINVAL);
    YuvUArgs a = a0;
    const char *rowsStr = GMAT_KNOB("GMAT_STRIP_ROWS");          // tuning / test override, read per launch
    const int rowsEnv = rowsStr ? atoi(rowsStr) : 0;
    const int nstrips = (a.dstW + kUStrip - 1) / kUStrip;
    a.nsg = (nstrips + 3) / 4;
    // band height: a band pays its ring's lead-in (R - 1 row pairs filtered for rows above it); a lone small frame wants enough waves
    const long wr = (long)a.dstH * nstrips * nframes * (a.yuvOut ? 2 : 1);
    // (round 4's last sweep, 32 frames a launch, 32 / 48 / 64 rows: 720p -> 1080p rgb24 81.3 / 78.4 / 84.2 us a launch, lanczos 100 / 94.8 / 102.5, 720p -> 4K rgb24 280 / 266.6 /
    // 264.4, nv12 177.6 / 167.8 / 167.4, 1080p -> 4K rgb24 292.8 / 281.5 / 274.6, 1080p -> 1440p nv12 105.6 / 102.6 / 107; 4:2:0 destinations of up to 1080 rows lose 1 % at 48:
    // profiles/r04_rows_all.txt)
    const long cap = (!a.yuvOut || a.dstH > 1080) ? 48 : 32;
    int rows = rowsEnv > 0 ? rowsEnv : (int)std::min(cap, std::max(4L, (wr + 4095) / 4096));
    if (a.yuvOut) rows = std::max(2, rows & ~1);
    a.bandRows = rows;
    a.nbands = (a.dstH + rows - 1) / rows;
    a.nblkL = a.nbands * a.nsg;
    a.nblk = a.nblkL;
    if (a.yuvOut) {
        const int cbytes = a.nv12 ? 2 * a.chrDstW : a.chrDstW;
        a.nsgC = ((cbytes + kUStrip - 1) / kUStrip + 3) / 4;
        a.bandRowsC = std::max(1, rows / 2);
        a.nbandsC = (a.chrDstH + a.bandRowsC - 1) / a.bandRowsC;
        a.nblkC = a.nbandsC * a.nsgC;
        a.nblk = a.nblkL + (a.nv12 ? 1 : 2) * a.nblkC;
    }
    const dim3 grid(a.xcdRemap ? 8 * ((a.nblk + 7) / 8) : a.nblk, nframes), block(256);
    const Yuv2xFrames &fr = *frames;
#define GMAT_U_PL(P_, R_, SD_) do { if (a.nv12) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvu_planes_kernel<P_, R_, SD_, true>), grid, block, 0, stream, a, fr); \
                                    else        hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvu_planes_kernel<P_, R_, SD_, false>), grid, block, 0, stream, a, fr); } while (0)
#define GMAT_U_RGB(P_, RL_, RC_, SD_) do { if (a.nv12) hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvu_rgb_kernel<P_, RL_, RC_, SD_, true>), grid, block, 0, stream, a, fr); \
                                           else        hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_yuvu_rgb_kernel<P_, RL_, RC_, SD_, false>), grid, block, 0, stream, a, fr); } while (0)
#define GMAT_U_SD(P_, SD_) do { \
        if (a.yuvOut) { if (a.RL <= 3) GMAT_U_PL(P_, 3, SD_); else GMAT_U_PL(P_, 5, SD_); } \
        else          { if (a.RL <= 4 && a.RC <= 3) GMAT_U_RGB(P_, 4, 3, SD_); else if (a.RL <= 4) GMAT_U_RGB(P_, 4, 4, SD_); else if (a.RL <= 6) GMAT_U_RGB(P_, 6, 4, SD_); else GMAT_U_RGB(P_, 8, 5, SD_); } } while (0)
#define GMAT_U_P(P_) do { if (a.SD == 1 * U_BPS) GMAT_U_SD(P_, 1 * U_BPS); else GMAT_U_SD(P_, 2 * U_BPS); } while (0)
#if U_BPS == 2
    if (a.P == 4) GMAT_U_P(4); else GMAT_U_P(6);
#else
    switch (a.P) { case 2: GMAT_U_P(2); break; case 3: GMAT_U_P(3); break; default: GMAT_U_P(4); }
#endif
#undef GMAT_U_P
#undef GMAT_U_SD
#undef GMAT_U_RGB
#undef GMAT_U_PL
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
