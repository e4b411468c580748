// k_scale16.hip — libswscale's generic scaler for 16-bit destinations (P016LE), gfx950.
//
// A 16-bit destination switches the CPU path to 19-bit intermediate lines held in int32 (dstBpc = 16, utils.c:1561-1570):
//   horizontal   hScale8To19_c    min(sum >> 3, 2^19 - 1)                          swscale.c:138-153
//                hScale16To19_c   min(sum >> (depth - 5), 2^19 - 1)                swscale.c:63-91
//   vertical     yuv2planeX_16_c  0x8000 + clip_int16(((1 << 14) - 0x40000000 + sum src * (unsigned)filter) >> 15)
//                                 in 32-bit wrap-around arithmetic                 output.c:157-181
//                yuv2plane1_16_c  clip_uint16((src + 4) >> 3) — the same value as the X form with coefficient 4096
//                yuv2nv12cX_16_c  the X form per chroma plane, interleaved, also for one tap   output.c:183-211
// The 15-bit kernels of k_scale_yuv.hip keep their lines as int16 pairs for v_dot2; 19-bit lines do not fit, so this
// path is a plain two-pass one: pass 1 filters every source row horizontally into an int32 plane in HBM (srcH x dstW),
// pass 2 filters that plane vertically and stores 16-bit samples.  One output sample per thread, taps read through
// the caches.  A completeness path (scale_cuda lists P016 as an output, vf_scale_cuda.c:45-54), not a fast one.
#include <hip/hip_runtime.h>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

namespace gmat {

// kind: 0 = 8-bit samples, sample stride `step` bytes; 10 / 16 = 16-bit samples (P010: >> 6), sample stride `step` bytes
__global__ __launch_bounds__(256) void hscale19_kernel(const uint8_t *src, int ss, int kind, int step, int srcW, int srcH,
                                                       DevFilter f, int32_t *dst, int dstW, int sh)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dstW || y >= srcH) return;
    const uint8_t *row = src + (size_t)y * ss;
    const int p0 = f.pos_even[x];
    int val = 0;
    for (int k = 0; k < f.pairs; k++) {
        const int cf = f.packed[(size_t)x * f.pairs + k];
        int s[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int i = min(p0 + 2 * k + j, srcW - 1);              // a tap past the plane has coefficient 0
            if (kind == 0) s[j] = row[(size_t)i * step];
            else {
                const unsigned v = *reinterpret_cast<const unsigned short *>(row + (size_t)i * step);
                s[j] = kind == 10 ? (int)(v >> 6) : (int)v;
            }
        }
        val += s[0] * (int)(short)(cf & 0xFFFF) + s[1] * (cf >> 16);
    }
    dst[(size_t)y * dstW + x] = min(val >> sh, (1 << 19) - 1);
}

// planes == 1: one plane -> 16-bit samples at dst + 2x.  planes == 2: U and V lines -> interleaved 16-bit pairs at dst + 4x.
__global__ __launch_bounds__(256) void vscale16_kernel(const int32_t *lineA, const int32_t *lineB, int lineW, int lineH, DevFilter f,
                                                       uint8_t *dst, int ds, int dstW, int dstH, int planes)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dstW || y >= dstH) return;
    const int p0 = f.pos_even[y];
    unsigned a = (1u << 14) - 0x40000000u, b = a;
    for (int k = 0; k < f.pairs; k++) {
        const int cf = f.packed[(size_t)y * f.pairs + k];
        const unsigned c0 = (unsigned)(int)(short)(cf & 0xFFFF), c1 = (unsigned)(cf >> 16);
        const int r0 = min(p0 + 2 * k, lineH - 1), r1 = min(p0 + 2 * k + 1, lineH - 1);
        a += (unsigned)lineA[(size_t)r0 * lineW + x] * c0 + (unsigned)lineA[(size_t)r1 * lineW + x] * c1;
        if (planes == 2) b += (unsigned)lineB[(size_t)r0 * lineW + x] * c0 + (unsigned)lineB[(size_t)r1 * lineW + x] * c1;
    }
    const int va = min(max((int)a >> 15, -32768), 32767) + 0x8000;
    if (planes == 1) {
        reinterpret_cast<unsigned short *>(dst + (size_t)y * ds)[x] = (unsigned short)va;
    } else {
        const int vb = min(max((int)b >> 15, -32768), 32767) + 0x8000;
        unsigned short *d = reinterpret_cast<unsigned short *>(dst + (size_t)y * ds) + 2 * x;
        d[0] = (unsigned short)va; d[1] = (unsigned short)vb;
    }
}

int launch_hscale19(const uint8_t *src, int ss, int kind, int step, int srcW, int srcH, const DevFilter &f, int32_t *dst, int dstW,
                    hipStream_t stream)
{
    if (dstW <= 0 || srcH <= 0) return 0;
    const int sh = kind == 0 ? 3 : kind - 5;                  // hScale8To19_c: 3; hScale16To19_c: depth - 1 - 4
    const dim3 grid((dstW + 255) / 256, srcH), block(256);
    hipLaunchKernelGGL(hscale19_kernel, grid, block, 0, stream, src, ss, kind, step, srcW, srcH, f, dst, dstW, sh);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_vscale16(const int32_t *lineA, const int32_t *lineB, int lineW, int lineH, const DevFilter &f, uint8_t *dst, int ds,
                    int dstW, int dstH, hipStream_t stream)
{
    if (dstW <= 0 || dstH <= 0) return 0;
    const dim3 grid((dstW + 255) / 256, dstH), block(256);
    hipLaunchKernelGGL(vscale16_kernel, grid, block, 0, stream, lineA, lineB, lineW, lineH, f, dst, ds, dstW, dstH, lineB ? 2 : 1);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
