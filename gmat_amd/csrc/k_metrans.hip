// k_metrans.hip — kernels only the MeTrans front-ends (include/gmat_metrans.h) need, for gfx950 (MI355X).
//
//   scale_nv12_bicubic_kernel   the reference's OWN bicubic (metrans/include/NvCodec/Resize_bicubic.cu:83-159): float 4 x 4, a = -0.5,
//                               source coordinate x * scale (no half-pixel centre, no anti-alias widening) clamped to [2, n - 2],
//                               truncating cast — restated operation by operation (and so is the test suite's checker)
//   nv12_to_planar_kernel       NV12 -> three stacked planes of 8-bit or float samples, R,G,B or B,G,R
//                               (YuvToRgbPlanarKernel, ColorSpace.cu:165-195) with the integer colour stage of k_yuv2rgb.hip
//   split_packed32_kernel       BGRA / RGBA -> the same stacked planes (behind the P016 sources, whose colours a context makes)
//   widen_shift8_kernel / narrow_shift8_kernel   ConvertUInt8ToUInt16 / ConvertUInt16ToUInt8 (BitDepth.cu:15-36): v << 8, v >> 8
//
// Memory plan of the bicubic: the reference gathers 16 bytes per output sample from global memory.  Its double sum is separable as written —
//   r = sum_y ( sum_x src[sy][sx] * cx[x] ) * cy[y]   (inner sum first, both sums from 0 in tap order)
// and the inner sum depends on (output column, source row) only.  A block therefore owns 256 output samples of a band of output rows: phase 1
// filters every source row the band touches ONCE into an LDS tile of floats (thread = column: its four coefficients are loop-invariant
// registers, its byte loads walk down a column so that a wave reads a contiguous span of each row), phase 2 takes four rows of that tile per
// output (thread = 4 adjacent columns: one ds_read_b128 per tap row, one dword store).  Same values in the same order as the reference's
// per-sample loop, 4 / scale source rows per output row instead of 4 (a 2:1 down-scale filters each source row once instead of twice,
// an up-scale far fewer), and the bytes of a source row come in as contiguous wave-wide reads.
#include <hip/hip_runtime.h>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

namespace gmat {

// BicubicCoefficient (Resize_bicubic.cu:83-87), a = -0.5, products and sums in the order written there
__device__ __forceinline__ float mt_bicubic_w(float d)
{
    d = fabsf(d);
    const float a = -0.5f;
    return d > 2.0f ? 0.0f
         : (d > 1.0f ? a * d * d * d - 5.0f * a * d * d + 8.0f * a * d - 4.0f * a
                     : (a + 2.0f) * d * d * d - (a + 3.0f) * d * d + 1.0f);
}

constexpr int kMtRows = 40;                 // source rows of the LDS tile: 40 x 256 floats = 40 KB
constexpr int kMtCols = 256;

struct MtBicubicArgs {
    const uint8_t *src; int ss;             // the plane (luma: bytes; chroma: U,V byte pairs), pitch in bytes
    uint8_t *dst; int ds;
    int srcN, srcRows;                      // samples per row per channel / rows of the SOURCE plane (the clamps' n)
    int outN, outRows;                      // samples per row per channel / rows written
    float fxScale, fyScale;
    int rowsPerBlock;                       // output rows a block owns (a multiple of 4)
    int dstAligned;                         // dword stores allowed
};

// C = 1: the luma plane; C = 2: the interleaved chroma plane (a column = one byte of a U,V pair)
template <int C>
__global__ __launch_bounds__(256) void scale_nv12_bicubic_kernel(MtBicubicArgs a)
{
    __shared__ float tile[kMtRows][kMtCols];
    const int t = threadIdx.x;
    const int e = blockIdx.x * kMtCols + t;                    // this thread's phase-1 column (byte of the output row)
    const int outBytes = a.outN * C;
    const float xmax = (float)(a.srcN - 2), ymax = (float)(a.srcRows - 2);
    // phase-1 constants of the column: min(max(x * fxScale, 2), n - 2), taps (int)fx - 1 .. + 2 (Resize_bicubic.cu:89-97,:143-156)
    const int px = min(e, outBytes - 1) / C, ch = C == 2 ? (e & 1) : 0;
    const float fx = fminf(fmaxf((float)px * a.fxScale, 2.0f), xmax);
    const int sx0 = (int)fx - 1;
    float cx[4]; int off[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        cx[k] = mt_bicubic_w((float)(sx0 + k) - fx);
        off[k] = min(sx0 + k, a.srcN - 1) * C + ch;            // the tap at n (coordinate clamped to n - 2) has weight exactly 0: read n - 1
    }
    const int yEnd = min(a.outRows, (int)(blockIdx.y + 1) * a.rowsPerBlock);
    int y0 = blockIdx.y * a.rowsPerBlock;
    while (y0 < yEnd) {
        // the rows of this pass: as many (in fours) as keep the source span within the tile; one group of four always fits
        // when 3 * fyScale + 4 <= kMtRows, and a single row always does
        const float fyA = fminf(fmaxf((float)y0 * a.fyScale, 2.0f), ymax);
        const int rLo = (int)fyA - 1;
        int y1 = y0;
        while (y1 < yEnd) {
            const int yl = min(y1 + 3, yEnd - 1);
            const float fyB = fminf(fmaxf((float)yl * a.fyScale, 2.0f), ymax);
            if ((int)fyB + 2 - rLo >= kMtRows) break;
            y1 = yl + 1;
        }
        if (y1 == y0) y1 = y0 + 1;                             // scale beyond the tile: row by row (4 source rows each)
        const float fyL = fminf(fmaxf((float)(y1 - 1) * a.fyScale, 2.0f), ymax);
        const int nRows = (int)fyL + 2 - rLo + 1;
        __syncthreads();                                       // the previous pass's readers are done
        if (e < outBytes) {
            for (int r = 0; r < nRows; r++) {
                const uint8_t *row = a.src + (size_t)min(rLo + r, a.srcRows - 1) * a.ss;
                float rx = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; k++) rx += (float)row[off[k]] * cx[k];
                tile[r][t] = rx;
            }
        }
        __syncthreads();
        // phase 2: thread = 4 adjacent columns of one row; 4 rows of the pass per step
        const int g = t & 63, e4 = blockIdx.x * kMtCols + 4 * g;
        for (int y = y0 + (t >> 6); y < y1; y += 4) {
            if (e4 >= outBytes) break;
            const float fy = fminf(fmaxf((float)y * a.fyScale, 2.0f), ymax);
            const int sy0 = (int)fy - 1;
            float r4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float cy = mt_bicubic_w((float)(sy0 + k) - fy);
                const float4 h = *reinterpret_cast<const float4 *>(&tile[sy0 + k - rLo][4 * g]);
                r4[0] += h.x * cy; r4[1] += h.y * cy; r4[2] += h.z * cy; r4[3] += h.w * cy;
            }
            unsigned o[4];
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = (unsigned)(int)fmaxf(fminf(r4[i], 255.0f), 0.0f);      // (uint8_t)max(min(r, 255), 0)
            uint8_t *d = a.dst + (size_t)y * a.ds + e4;
            if (a.dstAligned && e4 + 4 <= outBytes) *reinterpret_cast<unsigned *>(d) = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
            else for (int i = 0; i < 4 && e4 + i < outBytes; i++) d[i] = (uint8_t)o[i];
        }
        y0 = y1;
    }
}

int launch_scale_nv12_bicubic_ref(const uint8_t *src, int ss, int srcW, int srcH, uint8_t *dst, int ds, int dstW, int dstH, hipStream_t stream)
{
    // the reference clamps coordinates to [2, n - 2] on the luma plane and on the half-size chroma plane; below 8 x 8 its own
    // indices leave the frame
    if (!src || !dst || srcW < 8 || srcH < 8 || dstW < 2 || dstH < 2) return GMAT_ERR(EINVAL);
    const float fxScale = (float)srcW / (float)dstW, fyScale = (float)srcH / (float)dstH;       // Resize_bicubic.cu:140
    const int al = ((((uintptr_t)dst | (uintptr_t)ds) & 3) == 0);
    auto rows_per_block = [](float s, int rows) {
        int r = (int)((kMtRows - 5) / s) & ~3;                     // rows whose source span fits the tile
        r = std::max(4, std::min(r, 32));
        return std::min(r, (rows + 3) & ~3);
    };
    MtBicubicArgs L;
    L.src = src; L.ss = ss; L.dst = dst; L.ds = ds; L.srcN = srcW; L.srcRows = srcH;
    L.outN = 2 * (dstW / 2); L.outRows = 2 * (dstH / 2);           // ix < dstW / 2, iy < dstH / 2: an odd last column / row is not written
    L.fxScale = fxScale; L.fyScale = fyScale; L.rowsPerBlock = rows_per_block(fyScale, L.outRows); L.dstAligned = al;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_nv12_bicubic_kernel<1>), dim3((L.outN + kMtCols - 1) / kMtCols, (L.outRows + L.rowsPerBlock - 1) / L.rowsPerBlock),
                       dim3(256), 0, stream, L);
    MtBicubicArgs Cc = L;
    Cc.src = src + (size_t)srcH * ss; Cc.ss = ss / 2 * 2;   // a uchar2 plane of pitch nSrcPitch / 2 (:151)
    Cc.dst = dst + (size_t)dstH * ds;      // chroma at base + pitch * height on both sides (:150-152)
    Cc.srcN = srcW / 2; Cc.srcRows = srcH / 2; Cc.outN = dstW / 2; Cc.outRows = dstH / 2;
    Cc.rowsPerBlock = rows_per_block(fyScale, Cc.outRows);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_nv12_bicubic_kernel<2>), dim3((2 * Cc.outN + kMtCols - 1) / kMtCols, (Cc.outRows + Cc.rowsPerBlock - 1) / Cc.rowsPerBlock),
                       dim3(256), 0, stream, Cc);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- NV12 -> three stacked planes (plane k at dst + k * ds * h), 8-bit or float = u8 / 255 ----------------------------------------
// 4 x 2 pixels per thread like yuv2rgb_kernel: two dwords of luma, one of chroma; per plane and row one dword (8-bit) or 16 bytes (float)
template <bool F32, bool BGR>
__global__ __launch_bounds__(256) void nv12_to_planar_kernel(YuvSrc s, uint8_t *dst, int ds, int w, int h, Yuv2RgbConsts k, int aligned, int srcAligned)
{
    // round 4 (as nv12_to_rgbpf32_kernel, k_yuv2rgb.hip): u8 / 255.0f from a 256-entry LDS table built with the same IEEE division (12 divisions a
    // row and lane were most of the float form's instructions), dword loads where the rows allow them, streaming stores
    __shared__ float unit[256];
    if (F32) {
        const int i = threadIdx.y * 64 + threadIdx.x;
        unit[i] = (float)i / 255.0f;
        __syncthreads();
    }
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = (blockIdx.y * 4 + threadIdx.y) * 2;
    if (x >= w || y >= h) return;
    const size_t plane = (size_t)ds * h;
    const size_t crow = (size_t)(y >> 1);
    const int nx = min(4, w - x);
    ChromaTerms c[2];
    if (srcAligned && nx == 4) {
        const unsigned uv = *reinterpret_cast<const unsigned *>(s.u + crow * s.us + x);      // U0 V0 U1 V1 (x is a multiple of 4: pair x / 2 at byte x)
        c[0] = chroma_terms(k, (int)(uv & 0xFF), (int)((uv >> 8) & 0xFF));
        c[1] = chroma_terms(k, (int)((uv >> 16) & 0xFF), (int)(uv >> 24));
    } else {
        for (int i = 0; i < 2; i++) {
            const int xx = min(x + 2 * i, w - 1);
            const uint8_t *p = s.u + crow * s.us + 2 * (xx >> 1);
            c[i] = chroma_terms(k, p[0], p[1]);
        }
    }
    for (int r = 0; r < 2 && y + r < h; r++) {
        unsigned yy = 0;
        const uint8_t *prow = s.y + (size_t)(y + r) * s.ys + x;
        if (srcAligned && nx == 4) yy = *reinterpret_cast<const unsigned *>(prow);
        else for (int i = 0; i < 4; i++) yy |= (unsigned)prow[min(i, nx - 1)] << (8 * i);
        int o[3][4];
        for (int i = 0; i < 4; i++) {
            const int ycy = m24((int)((yy >> (8 * i)) & 0xFF), k.cy);
            const ChromaTerms &q = c[i >> 1];
            o[BGR ? 2 : 0][i] = luma_chan(q.r, ycy);
            o[1][i] = luma_chan(q.g, ycy);
            o[BGR ? 0 : 2][i] = luma_chan(q.b, ycy);
        }
        for (int pl = 0; pl < 3; pl++) {
            uint8_t *row = dst + pl * plane + (size_t)(y + r) * ds;
            if (F32) {
                float *f = reinterpret_cast<float *>(row) + x;
                if (aligned && nx == 4) st_stream(f, make_uint4(__builtin_bit_cast(unsigned, unit[o[pl][0]]), __builtin_bit_cast(unsigned, unit[o[pl][1]]),
                                                                __builtin_bit_cast(unsigned, unit[o[pl][2]]), __builtin_bit_cast(unsigned, unit[o[pl][3]])));
                else for (int i = 0; i < nx; i++) f[i] = unit[o[pl][i]];
            } else {
                if (aligned && nx == 4) st_stream(row + x, (unsigned)((unsigned)o[pl][0] | ((unsigned)o[pl][1] << 8) | ((unsigned)o[pl][2] << 16) | ((unsigned)o[pl][3] << 24)));
                else for (int i = 0; i < nx; i++) row[x + i] = (uint8_t)o[pl][i];
            }
        }
    }
}

int launch_nv12_to_planar(const YuvSrc &s, uint8_t *dst, int ds, int w, int h, const Yuv2RgbConsts &k, int f32, int bgr, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return 0;
    if (!s.nv12 || !dst) return GMAT_ERR(EINVAL);
    const size_t plane = (size_t)ds * h;
    const int aligned = ((((uintptr_t)dst | (uintptr_t)ds | plane) & (f32 ? 15 : 3)) == 0);
    const int srcAligned = ((((uintptr_t)s.y | (uintptr_t)s.ys | (uintptr_t)s.u | (uintptr_t)s.us) & 3) == 0) && (w % 2) == 0;
    const dim3 block(64, 4), grid((w + 255) / 256, (h + 7) / 8);
    if (f32 && bgr)       hipLaunchKernelGGL(HIP_KERNEL_NAME(nv12_to_planar_kernel<true, true>), grid, block, 0, stream, s, dst, ds, w, h, k, aligned, srcAligned);
    else if (f32)         hipLaunchKernelGGL(HIP_KERNEL_NAME(nv12_to_planar_kernel<true, false>), grid, block, 0, stream, s, dst, ds, w, h, k, aligned, srcAligned);
    else if (bgr)         hipLaunchKernelGGL(HIP_KERNEL_NAME(nv12_to_planar_kernel<false, true>), grid, block, 0, stream, s, dst, ds, w, h, k, aligned, srcAligned);
    else                  hipLaunchKernelGGL(HIP_KERNEL_NAME(nv12_to_planar_kernel<false, false>), grid, block, 0, stream, s, dst, ds, w, h, k, aligned, srcAligned);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- packed 32-bit pixels -> three stacked planes of their first three bytes, in byte order ------------------------------------------
template <bool F32>
__global__ __launch_bounds__(256) void split_packed32_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int aligned)
{
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const size_t plane = (size_t)ds * h;
    const int nx = min(4, w - x);
    unsigned px[4] = {0, 0, 0, 0};
    const uint8_t *sp = src + (size_t)y * ss + (size_t)x * 4;
    if (aligned && nx == 4) { const uint4 v = *reinterpret_cast<const uint4 *>(sp); px[0] = v.x; px[1] = v.y; px[2] = v.z; px[3] = v.w; }
    else for (int i = 0; i < nx; i++) px[i] = (unsigned)sp[4 * i] | ((unsigned)sp[4 * i + 1] << 8) | ((unsigned)sp[4 * i + 2] << 16);
    for (int pl = 0; pl < 3; pl++) {
        uint8_t *row = dst + pl * plane + (size_t)y * ds;
        unsigned b[4];
        for (int i = 0; i < 4; i++) b[i] = (px[i] >> (8 * pl)) & 0xFF;
        if (F32) {
            float *f = reinterpret_cast<float *>(row) + x;
            if (aligned && nx == 4) *reinterpret_cast<float4 *>(f) = make_float4((float)b[0] / 255.0f, (float)b[1] / 255.0f, (float)b[2] / 255.0f, (float)b[3] / 255.0f);
            else for (int i = 0; i < nx; i++) f[i] = (float)b[i] / 255.0f;
        } else {
            if (aligned && nx == 4) *reinterpret_cast<unsigned *>(row + x) = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
            else for (int i = 0; i < nx; i++) row[x + i] = (uint8_t)b[i];
        }
    }
}

int launch_split_packed32(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int f32, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return 0;
    if (!src || !dst) return GMAT_ERR(EINVAL);
    const size_t plane = (size_t)ds * h;
    const int aligned = ((((uintptr_t)src | (uintptr_t)ss) & 15) == 0) && ((((uintptr_t)dst | (uintptr_t)ds | plane) & (f32 ? 15 : 3)) == 0);
    const dim3 block(64, 4), grid((w + 255) / 256, (h + 3) / 4);
    if (f32) hipLaunchKernelGGL(HIP_KERNEL_NAME(split_packed32_kernel<true>), grid, block, 0, stream, src, ss, dst, ds, w, h, aligned);
    else     hipLaunchKernelGGL(HIP_KERNEL_NAME(split_packed32_kernel<false>), grid, block, 0, stream, src, ss, dst, ds, w, h, aligned);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- BitDepth.cu:15-36: dpUInt16[x] = {0, dpUInt8[x]} (little endian: v << 8) and its inverse (the high byte) --------------------------
// 16 samples per thread when both ends are 16-byte aligned; a grid-stride loop like the reference's
__global__ __launch_bounds__(256) void widen_shift8_kernel(const uint8_t *src, uint16_t *dst, long n, int aligned)
{
    const long stride = (long)gridDim.x * 256;
    if (aligned) {
        const long n16 = n >> 4;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + 16 * i);
            const unsigned in[4] = {v.x, v.y, v.z, v.w};
            unsigned o[8];
            for (int k = 0; k < 4; k++) {
                o[2 * k]     = ((in[k] & 0xFFu) << 8) | ((in[k] & 0xFF00u) << 16);
                o[2 * k + 1] = ((in[k] >> 8) & 0xFF00u) | (in[k] & 0xFF000000u);
            }
            uint4 *d = reinterpret_cast<uint4 *>(dst + 16 * i);
            d[0] = make_uint4(o[0], o[1], o[2], o[3]);
            d[1] = make_uint4(o[4], o[5], o[6], o[7]);
        }
        for (long i = (n16 << 4) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = (uint16_t)(src[i] << 8);
        return;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = (uint16_t)(src[i] << 8);
}

__global__ __launch_bounds__(256) void narrow_shift8_kernel(const uint16_t *src, uint8_t *dst, long n, int aligned)
{
    const long stride = (long)gridDim.x * 256;
    if (aligned) {
        const long n16 = n >> 4;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
            const uint4 *s4 = reinterpret_cast<const uint4 *>(src + 16 * i);
            const uint4 a = s4[0], b = s4[1];
            const unsigned in[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            unsigned o[4];
            for (int k = 0; k < 4; k++) {
                const unsigned lo = in[2 * k], hi = in[2 * k + 1];
                o[k] = ((lo >> 8) & 0xFFu) | ((lo >> 16) & 0xFF00u) | ((hi << 8) & 0xFF0000u) | (hi & 0xFF000000u);
            }
            *reinterpret_cast<uint4 *>(dst + 16 * i) = make_uint4(o[0], o[1], o[2], o[3]);
        }
        for (long i = (n16 << 4) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = (uint8_t)(src[i] >> 8);
        return;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = (uint8_t)(src[i] >> 8);
}

int launch_widen_shift8(const uint8_t *src, uint16_t *dst, long n, hipStream_t stream)
{
    if (n <= 0) return 0;
    if (!src || !dst || ((uintptr_t)dst & 1)) return GMAT_ERR(EINVAL);
    const int aligned = ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0);
    const long groups = (n + 16 * 256 - 1) / (16 * 256);
    hipLaunchKernelGGL(widen_shift8_kernel, dim3((unsigned)std::max(1L, std::min(groups, 4096L))), dim3(256), 0, stream, src, dst, n, aligned);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_narrow_shift8(const uint16_t *src, uint8_t *dst, long n, hipStream_t stream)
{
    if (n <= 0) return 0;
    if (!src || !dst || ((uintptr_t)src & 1)) return GMAT_ERR(EINVAL);
    const int aligned = ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0);
    const long groups = (n + 16 * 256 - 1) / (16 * 256);
    hipLaunchKernelGGL(narrow_shift8_kernel, dim3((unsigned)std::max(1L, std::min(groups, 4096L))), dim3(256), 0, stream, src, dst, n, aligned);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
