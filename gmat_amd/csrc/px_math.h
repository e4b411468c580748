// px_math.h — per-pixel fixed-point arithmetic shared by the kernels (device code, gfx950).
//
// Integer formulas only; every function states the libswscale expression it evaluates
// (paths relative to /root/reference/ffmpeg-gpu/libswscale).
#pragma once
#include <hip/hip_runtime.h>
#include "sws_tables.h"

namespace gmat {

typedef short short2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int clip_u8(int v) { return min(max(v, 0), 255); }   // v_med3_i32

// clip_u8(v >> s), written clamp-then-shift — ALWAYS use this for a clamped shift, never clip_u8(v >> s).
// hipcc (ROCm 7.2) turns the shift-then-clamp form of two neighbouring values into v_ashr_pk_u8_i32 and then ORs further
// bytes onto the result as if its upper 16 bits were zero; on gfx950 the instruction leaves the destination's upper half AS IT
// WAS.  Whether the output is right then depends on what the register allocator put there: rgb2yuv444_kernel was right by
// accident (destination = a source holding a 15-bit value), scale_yuv2p_kernel wrote garbage into bytes 2 and 3 of every
// output dword on hardware while the CPU emulation of the same source (plain C++ semantics) was bit-exact.
// tests/test_isa_guard.py disassembles the shipped library and fails on any v_ashr_pk_*.
__device__ __forceinline__ int clip_u8_shr(int v, int s) { return min(max(v, 0), (256 << s) - 1) >> s; }
// ff_dither_8x8_128[y & 7][x & 7] (swscale.c:36-46), the ordered dither libswscale gives 8-bit planar output when the SOURCE has more than
// 8 bits (should_dither, swscale.c:263-264, 482-485; 8-bit sources: the constant 64).  The table is affine over GF(2): 36 with x0 flipping
// 0x60, x1 0x18, x2 0x06, y0 0x40, y1 0x10, y2 0x04 — no memory (tests/test_parity_dither.py holds it against the table itself).
__host__ __device__ __forceinline__ int dither_8x8_128(int x, int y)
{
    return 36 ^ ((x & 1) * 0x60) ^ ((x & 2) * 0xC) ^ ((x & 4) + ((x & 4) >> 1)) ^ ((y & 1) * 0x40) ^ ((y & 2) * 8) ^ (y & 4);
}
// what the dither adds to a vertical accumulator that started at 64 << 12 (yuv2planeX_8_c / yuv2nv12cX_c: dither << 12)
__host__ __device__ __forceinline__ int dither_delta(int x, int y) { return (dither_8x8_128(x, y) - 64) << 12; }


// Streaming stores: a destination frame is written once and not read again by the launch; `nt` keeps its lines from displacing the
// source rows that neighbouring bands still share in the L2.  It pays where a wave writes WHOLE lines and costs where tiles of
// different blocks complete each other's lines in the L2 — measured per kernel, alternating builds on one box
// (profiles/r03zs_nt_stores_ab.txt): headline 116.8 -> 113.1 us per 32-frame launch (0.640 -> 0.660; sc0 / sc1 on the stores +-0, `nt` on the
// LOADS -17 %), 2:1 planes -1.5 ... -3 %, 3:1 / 3:2 RGB -1.9 / -4.4 %, the 1080p converter -8.4 %, hflip -8.5 %, 2-byte transpose -19 %
// (-10 % batched), median -7 ... -10 % (+4 % batched); NOT used where it lost: `smooth121_kernel` in all its forms (+20 ... +30 % batched:
// 60-column tiles), the 1-byte transpose (+15 % batched), the 3:1 / 3:2 plane walkers (+1.3 %), or did nothing (4:1, the band walker, the
// RGB-source scalers, rotate).  GMAT_NT_STORES=0 (a build flag) is the A/B.
#ifndef GMAT_NT_STORES
#define GMAT_NT_STORES 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && GMAT_NT_STORES
typedef unsigned nt_v2u __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned nt_v3u __attribute__((ext_vector_type(3), aligned(4)));
typedef unsigned nt_v4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void st_stream(void *p, unsigned v) { __builtin_nontemporal_store(v, reinterpret_cast<unsigned *>(p)); }
__device__ __forceinline__ void st_stream(void *p, uint2 v) { nt_v2u t = {v.x, v.y}; __builtin_nontemporal_store(t, reinterpret_cast<nt_v2u *>(p)); }
__device__ __forceinline__ void st_stream(void *p, uint3 v) { nt_v3u t = {v.x, v.y, v.z}; __builtin_nontemporal_store(t, reinterpret_cast<nt_v3u *>(p)); }
__device__ __forceinline__ void st_stream(void *p, uint4 v) { nt_v4u t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<nt_v4u *>(p)); }
// ld_stream: the same bit on a LOAD — only for bytes no other thread of the launch reads: a copy with both goes 0.62 -> 0.635 (0.78 ->
// 0.81 at 8 frames a launch), hflip 0.615 -> 0.625 (0.76 -> 0.815 at 16); the yuv -> rgb converter, one dword a lane, LOSES 3.5 % with it,
// and the headline, whose neighbouring bands share halo rows, 17 % (profiles/r03zs_nt_stores_ab.txt, last table)
__device__ __forceinline__ unsigned ld_stream(const void *p, unsigned) { return __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(p)); }
__device__ __forceinline__ uint2 ld_stream(const void *p, uint2) { const nt_v2u t = __builtin_nontemporal_load(reinterpret_cast<const nt_v2u *>(p)); return make_uint2(t.x, t.y); }
__device__ __forceinline__ uint3 ld_stream(const void *p, uint3) { const nt_v3u t = __builtin_nontemporal_load(reinterpret_cast<const nt_v3u *>(p)); return make_uint3(t.x, t.y, t.z); }
__device__ __forceinline__ uint4 ld_stream(const void *p, uint4) { const nt_v4u t = __builtin_nontemporal_load(reinterpret_cast<const nt_v4u *>(p)); return make_uint4(t.x, t.y, t.z, t.w); }
#else
__host__ __device__ __forceinline__ unsigned ld_stream(const void *p, unsigned) { return *reinterpret_cast<const unsigned *>(p); }
__host__ __device__ __forceinline__ uint2 ld_stream(const void *p, uint2) { return *reinterpret_cast<const uint2 *>(p); }
__host__ __device__ __forceinline__ uint3 ld_stream(const void *p, uint3) { return *reinterpret_cast<const uint3 *>(p); }
__host__ __device__ __forceinline__ uint4 ld_stream(const void *p, uint4) { return *reinterpret_cast<const uint4 *>(p); }
__host__ __device__ __forceinline__ void st_stream(void *p, unsigned v) { *reinterpret_cast<unsigned *>(p) = v; }
__host__ __device__ __forceinline__ void st_stream(void *p, uint2 v) { *reinterpret_cast<uint2 *>(p) = v; }
__host__ __device__ __forceinline__ void st_stream(void *p, uint3 v) { *reinterpret_cast<uint3 *>(p) = v; }
__host__ __device__ __forceinline__ void st_stream(void *p, uint4 v) { *reinterpret_cast<uint4 *>(p) = v; }
#endif

// 24-bit multiply (v_mul_i32_i24 / v_mad_i32_i24 run at full rate; v_mul_lo_u32 does not).  Every use
// below has |operands| < 2^23: pixel values <= 510, table constants <= 2^18.
__device__ __forceinline__ int m24(int a, int b) { return __mul24(a, b); }

// Load of a per-tile table entry whose index is wave-uniform.  The tables are written once at context creation
// and never while a kernel runs, so they may be read through the constant address space: the compiler then emits
// s_load_dword (scalar cache, result in an SGPR) instead of a vector load + v_readfirstlane, which shortens the
// dependent-load chain at the start of every workgroup.
__device__ __forceinline__ int uniform_load(const int32_t *p, int i)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(4))) const int32_t *const_ptr;
    return ((const_ptr)(unsigned long long)p)[i];
#else
    return p[i];
#endif
}

// 2-way int16 dot product with int32 accumulate: v_dot2c_i32_i16 (exact integer arithmetic).
__device__ __forceinline__ int dot2(int packed_ab, int packed_cd, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, packed_ab),
                                  __builtin_bit_cast(short2v, packed_cd), acc, false);
}

// 4-way signed-byte dot product with int32 accumulate: v_dot4c_i32_i8 (exact integer arithmetic).
__device__ __forceinline__ int dot4s(int bytes_a, int bytes_b, int acc)
{
    return __builtin_amdgcn_sdot4(bytes_a, bytes_b, acc, false);
}

// ---- yuv -> rgb through the closed form of the yuv2rgb.c tables ----------------------------
// table_rV[V][Y] = y_table[offR + ((V*crv)>>16) + Y], y_table[i] = clip_u8((yb0 + i*cy + 0x8000)>>16)
// (yuv2rgb.c:737-760, :958-971), so each channel is clip_u8((term + Y*cy) >> 16) with a per-chroma term.
struct ChromaTerms { int r, g, b; };

__device__ __forceinline__ ChromaTerms chroma_terms(const Yuv2RgbConsts &k, int U, int V)
{
    const int kr = k.offR + (m24(V, k.crv) >> 16);
    const int kg = k.offG + (m24(U, k.cgu) >> 16) + (m24(V, k.cgv) >> 16);
    const int kb = k.offB + (m24(U, k.cbu) >> 16);
    ChromaTerms t;
    t.r = k.base + m24(kr, k.cy);
    t.g = k.base + m24(kg, k.cy);
    t.b = k.base + m24(kb, k.cy);
    return t;
}

// clip_u8(v >> 16) written as clamp-then-shift.  The shift-then-clamp form makes hipcc (ROCm 7.2) pair two
// channels into v_ashr_pk_u8_i32, whose upper destination half is NOT cleared on gfx950 while the compiler
// assumes it is: packed RGB came out as (R|B) in the B byte (seen on hardware, round 1).
__device__ __forceinline__ int luma_chan(int term, int ycy) { return min(max(term + ycy, 0), 0xFFFFFF) >> 16; }

// ---- rgb -> 14-bit Y / U / V as the generic scaler's input stage does (input.c:795-866) -----
__device__ __forceinline__ int rgb_to_y14(const Rgb2YuvConsts &c, int r, int g, int b)
{
    // rgb24ToY_c: (ry*r + gy*g + by*b + (32<<14) + (1<<8)) >> 9
    return (m24(c.ry, r) + m24(c.gy, g) + m24(c.by, b) + (32 << 14) + (1 << 8)) >> 9;
}
__device__ __forceinline__ int rgb_to_u14(const Rgb2YuvConsts &c, int r, int g, int b)
{
    // rgb24ToUV_c: (ru*r + gu*g + bu*b + (256<<14) + (1<<8)) >> 9
    return (m24(c.ru, r) + m24(c.gu, g) + m24(c.bu, b) + (256 << 14) + (1 << 8)) >> 9;
}
__device__ __forceinline__ int rgb_to_v14(const Rgb2YuvConsts &c, int r, int g, int b)
{
    return (m24(c.rv, r) + m24(c.gv, g) + m24(c.bv, b) + (256 << 14) + (1 << 8)) >> 9;
}
// rgb24ToUV_half_c on the SUM of two horizontally adjacent pixels:
//   (ru*r + gu*g + bu*b + (256<<15) + (1<<9)) >> 10
__device__ __forceinline__ int rgbsum_to_u14(const Rgb2YuvConsts &c, int r, int g, int b)
{
    return (m24(c.ru, r) + m24(c.gu, g) + m24(c.bu, b) + (256 << 15) + (1 << 9)) >> 10;
}
__device__ __forceinline__ int rgbsum_to_v14(const Rgb2YuvConsts &c, int r, int g, int b)
{
    return (m24(c.rv, r) + m24(c.gv, g) + m24(c.bv, b) + (256 << 15) + (1 << 9)) >> 10;
}

// ---- full-chroma output stage: yuv2rgb_write_full (output.c:1886-1935) ------------------------
// Y, U, V are the vertically filtered values after their >>10.  Returns R | G<<8 | B<<16.
__device__ __forceinline__ unsigned yuv_to_rgb_full(const Yuv2RgbConsts &k, int Y, int U, int V)
{
    unsigned y = (unsigned)((Y - k.y_offset) * k.y_coeff) + (1u << 21);
    int R = (int)(y + (unsigned)(V * k.v2r));
    int G = (int)(y + (unsigned)(V * k.v2g) + (unsigned)(U * k.u2g));
    int B = (int)(y + (unsigned)(U * k.u2b));
    if ((R | G | B) & 0xC0000000) {
        // av_clip_uintp2(x, 30)
        R = (R & ~0x3FFFFFFF) ? ((~R) >> 31) & 0x3FFFFFFF : R;
        G = (G & ~0x3FFFFFFF) ? ((~G) >> 31) & 0x3FFFFFFF : G;
        B = (B & ~0x3FFFFFFF) ? ((~B) >> 31) & 0x3FFFFFFF : B;
    }
    return (unsigned)(R >> 22) | ((unsigned)(G >> 22) << 8) | ((unsigned)(B >> 22) << 16);
}

} // namespace gmat
