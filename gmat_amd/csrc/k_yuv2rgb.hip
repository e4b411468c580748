// k_yuv2rgb.hip — YUV 4:2:0 -> packed RGB colour conversion for gfx950 (MI355X).
//
// Replaces yuv2rgb_cuda -> nv122color / yuv4202color -> yuv2rgb_odd_kernel
// (libswscale/cuda/yuv2rgb_cuda.cu:862-907,548-562,207-340) with libswscale's CPU arithmetic:
// nearest chroma + the fixed-point tables of yuv2rgb.c, evaluated in closed form (px_math.h).
//
// Memory plan (HBM-bound, 4.5 B/px for rgb24): one thread converts a 4x2 pixel block.
//   loads : 2 x dword of Y (one per row) + 1 dword of interleaved UV (or 2 x ushort for planar)
//           -> a wave reads 256 contiguous bytes per load instruction
//   stores: 2 x global_store_dwordx3 (12 B = 4 rgb24 pixels) -> a wave writes 768 contiguous bytes
//           per instruction; rgba uses one dwordx4 per row.
// No LDS: nothing is reused beyond the 2x2 chroma sharing, which lives in registers.
#include <hip/hip_runtime.h>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

namespace gmat {

enum { OUT_RGB24 = 0, OUT_BGR24 = 1, OUT_RGBA = 2, OUT_BGRA = 3 };

template <int OUT>
__device__ __forceinline__ void store_px(uint8_t *d, int r, int g, int b)
{
    if (OUT == OUT_RGB24)      { d[0] = (uint8_t)r; d[1] = (uint8_t)g; d[2] = (uint8_t)b; }
    else if (OUT == OUT_BGR24) { d[0] = (uint8_t)b; d[1] = (uint8_t)g; d[2] = (uint8_t)r; }
    else if (OUT == OUT_RGBA)  { d[0] = (uint8_t)r; d[1] = (uint8_t)g; d[2] = (uint8_t)b; d[3] = 255; }
    else                       { d[0] = (uint8_t)b; d[1] = (uint8_t)g; d[2] = (uint8_t)r; d[3] = 255; }
}

// converts 4 luma samples (packed in `y4`) sharing two chroma pairs into 4 packed pixels and stores them
template <int OUT>
__device__ __forceinline__ void convert_row4(uint8_t *drow, unsigned y4, const ChromaTerms &c0,
                                             const ChromaTerms &c1, int cy)
{
    unsigned px[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int ycy = m24((int)((y4 >> (8 * i)) & 0xFF), cy);
        const ChromaTerms &c = i < 2 ? c0 : c1;
        const unsigned r = (unsigned)luma_chan(c.r, ycy), g = (unsigned)luma_chan(c.g, ycy),
                       b = (unsigned)luma_chan(c.b, ycy);
        if (OUT == OUT_RGB24 || OUT == OUT_RGBA) px[i] = r | (g << 8) | (b << 16) | 0xFF000000u;
        else                                     px[i] = b | (g << 8) | (r << 16) | 0xFF000000u;
    }
    if (OUT == OUT_RGBA || OUT == OUT_BGRA) {
        st_stream(drow, make_uint4(px[0], px[1], px[2], px[3]));
    } else {
        uint3 o;
        o.x = (px[0] & 0xFFFFFF) | (px[1] << 24);
        o.y = ((px[1] >> 8) & 0xFFFF) | (px[2] << 16);
        o.z = ((px[2] >> 16) & 0xFF) | (px[3] << 8);
        st_stream(drow, o);
    }
}

// BATCH: grid.z = frame, the plane pointers of up to 32 frames of one geometry come from the kernel-argument
// segment (Yuv2xFrames); strides and the alignment class are shared.
template <int OUT, bool BATCH>
__global__ __launch_bounds__(256) void yuv2rgb_kernel(YuvSrc s, Yuv2xFrames fr, uint8_t *dst, int ds, int w, int h,
                                                      Yuv2RgbConsts k, int aligned)
{
    constexpr int BPP = (OUT == OUT_RGBA || OUT == OUT_BGRA) ? 4 : 3;
    if (BATCH) {
        const int f = blockIdx.z;
        s.y = fr.y[f]; s.u = fr.u[f]; s.v = fr.v[f]; dst = fr.dst[f];
    }
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = (blockIdx.y * 4 + threadIdx.y) * 2;
    if (x >= w || y >= h) return;

    const uint8_t *py = s.y + (size_t)y * s.ys + x;
    const size_t crow = (size_t)(y >> 1);
    uint8_t *d0 = dst + (size_t)y * ds + (size_t)x * BPP;
    const bool two_rows = y + 1 < h;

    if (aligned && x + 4 <= w) {
        const unsigned y0 = *reinterpret_cast<const unsigned *>(py);
        const unsigned y1 = two_rows ? *reinterpret_cast<const unsigned *>(py + s.ys) : 0u;
        int u0, v0, u1, v1;
        if (s.nv12) {
            const unsigned uv = *reinterpret_cast<const unsigned *>(s.u + crow * s.us + x);
            u0 = uv & 0xFF; v0 = (uv >> 8) & 0xFF; u1 = (uv >> 16) & 0xFF; v1 = uv >> 24;
        } else {
            const unsigned short uu = *reinterpret_cast<const unsigned short *>(s.u + crow * s.us + (x >> 1));
            const unsigned short vv = *reinterpret_cast<const unsigned short *>(s.v + crow * s.vs + (x >> 1));
            u0 = uu & 0xFF; u1 = uu >> 8; v0 = vv & 0xFF; v1 = vv >> 8;
        }
        const ChromaTerms c0 = chroma_terms(k, u0, v0), c1 = chroma_terms(k, u1, v1);
        convert_row4<OUT>(d0, y0, c0, c1, k.cy);
        if (two_rows) convert_row4<OUT>(d0 + ds, y1, c0, c1, k.cy);
        return;
    }
    // edge / unaligned path: byte accesses, chroma index clamped by construction (x>>1 < ceil(w/2))
    const int nx = min(4, w - x);
    for (int r = 0; r < (two_rows ? 2 : 1); r++) {
        for (int i = 0; i < nx; i++) {
            const int xx = x + i;
            int U, V;
            if (s.nv12) {
                const uint8_t *p = s.u + crow * s.us + 2 * (xx >> 1);
                U = p[0]; V = p[1];
            } else {
                U = s.u[crow * s.us + (xx >> 1)];
                V = s.v[crow * s.vs + (xx >> 1)];
            }
            const ChromaTerms c = chroma_terms(k, U, V);
            const int ycy = m24((int)py[(size_t)r * s.ys + i], k.cy);
            store_px<OUT>(d0 + (size_t)r * ds + (size_t)i * BPP, luma_chan(c.r, ycy), luma_chan(c.g, ycy),
                          luma_chan(c.b, ycy));
        }
    }
}

// nv12 -> planar float RGB (AV_PIX_FMT_RGBPF32LE as GMAT defines it: three stacked planes,
// plane k at dst + k*ds*h), value = u8 / 255.0f — the layout and normalisation of
// nv122color_planar<RGBF32> (yuv2rgb_cuda.cu:381-545,564-570) with the integer colour stage.
// This is the tensor a network reads (format_cuda -> tensorrt in the reference's graphs; CSwscale.c): 12 bytes WRITTEN per pixel against 1.5 read,
// so the kernel is a streaming writer.  Round 4: a lane makes 4 pixels of 2 rows from three dword loads; the chroma terms once per sample pair
// (not per pixel and row); u8 / 255.0f comes from a 256-entry table in LDS built with the same IEEE division (24 divisions a lane were 60 % of
// the kernel's instructions); the 16-byte stores are non-temporal (a frame of floats is written once and read by the next kernel at the
// earliest); FRAMES: grid.z = frame.  1080p 8.4 -> see profiles/r04_rgbpf32.txt
template <bool FRAMES>
__global__ __launch_bounds__(256) void nv12_to_rgbpf32_kernel(YuvSrc s, Yuv2xFrames fr, uint8_t *dst, int ds, int w, int h,
                                                              Yuv2RgbConsts k, int aligned)
{
    __shared__ float unit[256];
    {
        const int i = threadIdx.y * 64 + threadIdx.x;
        unit[i] = (float)i / 255.0f;
    }
    __syncthreads();
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = (blockIdx.y * 4 + threadIdx.y) * 2;
    if (x >= w || y >= h) return;
    const uint8_t *py = s.y, *pu = s.u;
    if (FRAMES) { const int f = blockIdx.z; py = fr.y[f]; pu = fr.u[f]; dst = fr.dst[f]; }
    const size_t plane = (size_t)ds * h;
    const int nx = min(4, w - x);
    // the two chroma pairs of these 4 columns: one dword (U0 V0 U1 V1) where the row allows it
    unsigned uv;
    {
        const uint8_t *p = pu + (size_t)(y >> 1) * s.us + x;      // x is even: column x / 2 of the interleaved row
        if (aligned && nx == 4) uv = *reinterpret_cast<const unsigned *>(p);
        else {
            const int x1 = min(x + 2, ((w + 1) & ~1) - 2);         // the last pair again when the row ends here
            uv = (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)pu[(size_t)(y >> 1) * s.us + x1] << 16) | ((unsigned)pu[(size_t)(y >> 1) * s.us + x1 + 1] << 24);
        }
    }
    const ChromaTerms c0 = chroma_terms(k, (int)(uv & 0xFF), (int)((uv >> 8) & 0xFF)), c1 = chroma_terms(k, (int)((uv >> 16) & 0xFF), (int)(uv >> 24));
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (y + r >= h) break;
        unsigned yy;
        const uint8_t *prow = py + (size_t)(y + r) * s.ys + x;
        if (aligned && nx == 4) yy = *reinterpret_cast<const unsigned *>(prow);
        else { yy = 0; for (int i = 0; i < nx; i++) yy |= (unsigned)prow[i] << (8 * i); }
        float o[3][4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const ChromaTerms &c = i < 2 ? c0 : c1;
            const int ycy = m24((int)((yy >> (8 * i)) & 0xFF), k.cy);
            o[0][i] = unit[luma_chan(c.r, ycy)];
            o[1][i] = unit[luma_chan(c.g, ycy)];
            o[2][i] = unit[luma_chan(c.b, ycy)];
        }
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            float *row = reinterpret_cast<float *>(dst + pl * plane + (size_t)(y + r) * ds) + x;
            if (aligned && nx == 4) {
                st_stream(row, make_uint4(__builtin_bit_cast(unsigned, o[pl][0]), __builtin_bit_cast(unsigned, o[pl][1]),
                                          __builtin_bit_cast(unsigned, o[pl][2]), __builtin_bit_cast(unsigned, o[pl][3])));
            } else {
                for (int i = 0; i < nx; i++) row[i] = o[pl][i];
            }
        }
    }
}

// rgb24 <-> bgr24 (rgb24tobgr24_cuda, rgb2rgb_cuda_kernel.cu:7-41): 4 pixels = 3 dwords per thread
__global__ __launch_bounds__(256) void swap_rb24_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds,
                                                        int w, int h, int aligned)
{
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const uint8_t *s = src + (size_t)y * ss + (size_t)x * 3;
    uint8_t *d = dst + (size_t)y * ds + (size_t)x * 3;
    if (aligned && x + 4 <= w) {
        const uint3 v = ld_stream(s, uint3());          // (round 4: every byte read once and written once — streaming both ways)
        uint3 o;
        // bytes: R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3  ->  B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
        // (written with shifts; the compiler folds the byte moves into v_perm_b32 / v_bfi_b32)
        const unsigned b0 = (v.x >> 16) & 0xFF, g0 = (v.x >> 8) & 0xFF, r0 = v.x & 0xFF, r1 = v.x >> 24;
        const unsigned g1 = v.y & 0xFF, b1 = (v.y >> 8) & 0xFF, r2 = (v.y >> 16) & 0xFF, g2 = v.y >> 24;
        const unsigned b2 = v.z & 0xFF, r3 = (v.z >> 8) & 0xFF, g3 = (v.z >> 16) & 0xFF, b3 = v.z >> 24;
        o.x = b0 | (g0 << 8) | (r0 << 16) | (b1 << 24);
        o.y = g1 | (r1 << 8) | (b2 << 16) | (g2 << 24);
        o.z = r2 | (b3 << 8) | (g3 << 16) | (r3 << 24);
        st_stream(d, o);
        return;
    }
    const int nx = min(4, w - x);
    for (int i = 0; i < nx; i++) {
        const uint8_t r = s[3 * i], g = s[3 * i + 1], b = s[3 * i + 2];
        d[3 * i] = b; d[3 * i + 1] = g; d[3 * i + 2] = r;
    }
}

// RGB24/BGR24 <-> RGBA/BGRA and RGBA <-> BGRA at equal size: the byte moves of rgbToRgbWrapper
// (swscale_unscaled.c:1579-1640; rgb24to32 / rgb24tobgr32 / rgb32to24 / rgb32tobgr24 / shuffle_bytes_2103).
// Alpha is 255 when created, dropped when removed and kept 32 -> 32.  4 pixels per thread: 12 or 16 bytes in,
// 12 or 16 bytes out.
template <int SB, int DB>
__global__ __launch_bounds__(256) void repack_rgb_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int swap,
                                                         int aligned)
{
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const uint8_t *s = src + (size_t)y * ss + (size_t)x * SB;
    uint8_t *d = dst + (size_t)y * ds + (size_t)x * DB;
    unsigned px[4];                                   // c0 | c1 << 8 | c2 << 16 | a << 24 per pixel
    const bool full = aligned && x + 4 <= w;
    const int nx = min(4, w - x);
    if (full) {
        if (SB == 4) {
            const uint4 v = ld_stream(s, uint4());
            px[0] = v.x; px[1] = v.y; px[2] = v.z; px[3] = v.w;
        } else {
            const uint3 v = ld_stream(s, uint3());
            px[0] = v.x | 0xFF000000u;
            px[1] = (v.x >> 24) | (v.y << 8) | 0xFF000000u;
            px[2] = (v.y >> 16) | (v.z << 16) | 0xFF000000u;
            px[3] = (v.z >> 8) | 0xFF000000u;
        }
    } else {
        for (int i = 0; i < 4; i++) {
            const int j = min(i, nx - 1);
            px[i] = s[SB * j] | (s[SB * j + 1] << 8) | (s[SB * j + 2] << 16) | (SB == 4 ? (unsigned)s[SB * j + 3] << 24 : 0xFF000000u);
        }
    }
    if (swap & 1) {
#pragma unroll
        for (int i = 0; i < 4; i++) px[i] = (px[i] & 0xFF00FF00u) | ((px[i] >> 16) & 0xFF) | ((px[i] & 0xFF) << 16);
    }
    if (swap & 2) {                                   // a padding byte becomes a real alpha: 255 (swscale.c:959-978)
#pragma unroll
        for (int i = 0; i < 4; i++) px[i] |= 0xFF000000u;
    }
    if (full) {
        if (DB == 4) {
            st_stream(d, make_uint4(px[0], px[1], px[2], px[3]));
        } else {
            uint3 o;
            o.x = (px[0] & 0xFFFFFFu) | (px[1] << 24);
            o.y = ((px[1] >> 8) & 0xFFFFu) | (px[2] << 16);
            o.z = ((px[2] >> 16) & 0xFFu) | (px[3] << 8);
            st_stream(d, o);
        }
    } else {
        for (int i = 0; i < nx; i++)
            for (int b = 0; b < DB; b++) d[DB * i + b] = (uint8_t)(px[i] >> (8 * b));
    }
}

static inline bool aligned4(const void *p, int stride) { return (((uintptr_t)p | (uintptr_t)stride) & 3) == 0; }

int launch_yuv2rgb(const YuvSrc &s, uint8_t *dst, int ds, int w, int h, int dstFormat,
                   const Yuv2RgbConsts &k, hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (w <= 0 || h <= 0) return 0;
    if (frames && (nframes < 1 || nframes > kYuv2xMaxFrames)) return GMAT_ERR(EINVAL);
    const int bpp = bytes_per_pixel(dstFormat);
    if (!bpp) return GMAT_ERR(ENOSYS);
    // vector path: 4-byte aligned luma/chroma rows and dword-aligned output groups
    // (4 px * 3 B = 12 B keeps dword alignment; rgba needs 16 B)
    int aligned = 1;
    for (int f = 0; f < (frames ? nframes : 1); f++) {
        const uint8_t *py = frames ? frames->y[f] : s.y, *pu = frames ? frames->u[f] : s.u, *pv = frames ? frames->v[f] : s.v;
        const uint8_t *pd = frames ? frames->dst[f] : dst;
        int al = aligned4(py, s.ys) && aligned4(pd, ds);
        if (s.nv12) al = al && aligned4(pu, s.us);
        else        al = al && ((((uintptr_t)pu | (uintptr_t)pv | (uintptr_t)s.us | (uintptr_t)s.vs) & 1) == 0);
        if (bpp == 4) al = al && ((((uintptr_t)pd | (uintptr_t)ds) & 15) == 0);
        aligned = aligned && al;                // one class for the launch: the byte path is always correct
    }
    const dim3 block(64, 4), grid((w + 255) / 256, (h + 7) / 8, frames ? nframes : 1);
#define GMAT_Y2R(OUT_) do { \
        if (frames) hipLaunchKernelGGL(HIP_KERNEL_NAME(yuv2rgb_kernel<OUT_, true>), grid, block, 0, stream, s, *frames, dst, ds, w, h, k, aligned); \
        else { Yuv2xFrames none; none.y[0] = nullptr; \
               hipLaunchKernelGGL(HIP_KERNEL_NAME(yuv2rgb_kernel<OUT_, false>), grid, block, 0, stream, s, none, dst, ds, w, h, k, aligned); } } while (0)
    switch (dstFormat) {
    case GMAT_PIX_FMT_RGB24: GMAT_Y2R(OUT_RGB24); break;
    case GMAT_PIX_FMT_BGR24: GMAT_Y2R(OUT_BGR24); break;
    case GMAT_PIX_FMT_RGBA:  GMAT_Y2R(OUT_RGBA); break;
    case GMAT_PIX_FMT_BGRA:  GMAT_Y2R(OUT_BGRA); break;
    default: return GMAT_ERR(ENOSYS);
    }
#undef GMAT_Y2R
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// planar float RGB (three stacked planes, plane stride = ss * h) -> packed rgb24 / bgr24: the inverse of the
// normalisation above, u8 = (int)(clamp(f, 0, 1) * 255 + 0.5).  A frame produced by nv12_to_rgbpf32_kernel comes
// back to exactly the same bytes (k / 255.0f * 255 + 0.5 truncates to k for every k in 0..255).  The reference has
// no integer definition of this direction (rgbpf32_to_nv12, format_cuda_kernel.cu:624-630, is a float matrix):
// 4:2:0 outputs run this kernel into the context's RGB24 intermediate and then the RGB -> YUV path.
__global__ __launch_bounds__(256) void rgbpf32_to_rgb24_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h,
                                                               int bgr, int aligned)
{
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const size_t plane = (size_t)ss * h;
    unsigned c[3][4];
    const int nx = min(4, w - x);
#pragma unroll
    for (int pl = 0; pl < 3; pl++) {
        const float *row = reinterpret_cast<const float *>(src + pl * plane + (size_t)y * ss) + x;
        float f[4] = {0.f, 0.f, 0.f, 0.f};
        if (aligned && nx == 4) { const float4 v = *reinterpret_cast<const float4 *>(row); f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
        else for (int i = 0; i < nx; i++) f[i] = row[i];
#pragma unroll
        for (int i = 0; i < 4; i++) c[pl][i] = (unsigned)(int)(__fadd_rn(__fmul_rn(fminf(fmaxf(f[i], 0.0f), 1.0f), 255.0f), 0.5f));
    }
    const int r = bgr ? 2 : 0, b = 2 - r;
    uint8_t *d = dst + (size_t)y * ds + (size_t)x * 3;
    if (nx == 4 && ((((uintptr_t)dst | (uintptr_t)ds) & 3) == 0)) {
        uint3 o;
        o.x = c[r][0] | (c[1][0] << 8) | (c[b][0] << 16) | (c[r][1] << 24);
        o.y = c[1][1] | (c[b][1] << 8) | (c[r][2] << 16) | (c[1][2] << 24);
        o.z = c[b][2] | (c[r][3] << 8) | (c[1][3] << 16) | (c[b][3] << 24);
        *reinterpret_cast<uint3 *>(d) = o;
    } else {
        for (int i = 0; i < nx; i++) { d[3 * i] = (uint8_t)c[r][i]; d[3 * i + 1] = (uint8_t)c[1][i]; d[3 * i + 2] = (uint8_t)c[b][i]; }
    }
}

int launch_rgbpf32_to_rgb24(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int bgr, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return 0;
    const int aligned = ((((uintptr_t)src | (uintptr_t)ss) & 15) == 0);
    const dim3 block(64, 4), grid((w + 255) / 256, (h + 3) / 4);
    hipLaunchKernelGGL(rgbpf32_to_rgb24_kernel, grid, block, 0, stream, src, ss, dst, ds, w, h, bgr, aligned);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_nv12_to_rgbpf32(const YuvSrc &s, uint8_t *dst, int ds, int w, int h, const Yuv2RgbConsts &k,
                           hipStream_t stream, const Yuv2xFrames *frames, int nframes)
{
    if (w <= 0 || h <= 0) return 0;
    if (!s.nv12) return GMAT_ERR(ENOSYS);
    if (frames && (nframes < 1 || nframes > kYuv2xMaxFrames)) return GMAT_ERR(EINVAL);
    // vector path: dword-aligned luma / chroma rows, 16-byte aligned float rows, for every frame of the launch (the byte path is always correct)
    int aligned = (w % 2) == 0;
    for (int f = 0; f < (frames ? nframes : 1); f++) {
        const uint8_t *py = frames ? frames->y[f] : s.y, *pu = frames ? frames->u[f] : s.u, *pd = frames ? frames->dst[f] : dst;
        aligned = aligned && aligned4(py, s.ys) && aligned4(pu, s.us) && ((((uintptr_t)pd | (uintptr_t)ds) & 15) == 0);
    }
    const dim3 block(64, 4), grid((w + 255) / 256, (h + 7) / 8, frames ? nframes : 1);
    if (frames) hipLaunchKernelGGL(HIP_KERNEL_NAME(nv12_to_rgbpf32_kernel<true>), grid, block, 0, stream, s, *frames, dst, ds, w, h, k, aligned);
    else { Yuv2xFrames none; none.y[0] = nullptr;
           hipLaunchKernelGGL(HIP_KERNEL_NAME(nv12_to_rgbpf32_kernel<false>), grid, block, 0, stream, s, none, dst, ds, w, h, k, aligned); }
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_repack_rgb(const uint8_t *src, int ss, int srcBpp, uint8_t *dst, int ds, int dstBpp, int w, int h, int swapRB,
                      hipStream_t stream)
{
    if (w <= 0 || h <= 0) return 0;
    // dword groups for 24-bit rows, 16-byte groups for 32-bit rows
    const int aligned = ((((uintptr_t)src | (uintptr_t)ss) & (srcBpp == 4 ? 15 : 3)) == 0) &&
                        ((((uintptr_t)dst | (uintptr_t)ds) & (dstBpp == 4 ? 15 : 3)) == 0);
    const dim3 block(64, 4), grid((w + 255) / 256, (h + 3) / 4);
    if (srcBpp == 3 && dstBpp == 4)      hipLaunchKernelGGL(HIP_KERNEL_NAME(repack_rgb_kernel<3, 4>), grid, block, 0, stream, src, ss, dst, ds, w, h, swapRB, aligned);
    else if (srcBpp == 4 && dstBpp == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(repack_rgb_kernel<4, 3>), grid, block, 0, stream, src, ss, dst, ds, w, h, swapRB, aligned);
    else if (srcBpp == 4 && dstBpp == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(repack_rgb_kernel<4, 4>), grid, block, 0, stream, src, ss, dst, ds, w, h, swapRB, aligned);
    else return GMAT_ERR(ENOSYS);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_swap_rb24(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return 0;
    const int aligned = aligned4(src, ss) && aligned4(dst, ds);
    const dim3 block(64, 4), grid((w + 255) / 256, (h + 3) / 4);
    hipLaunchKernelGGL(swap_rb24_kernel, grid, block, 0, stream, src, ss, dst, ds, w, h, aligned);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
