// k_transform.hip — crop / flip / rotate(transposes) / 3x3 smooth on packed pixels for gfx950.
//
// Stands in for the CV-CUDA operators behind vf_crop_nvcv.c:277, vf_flip_nvcv.c:251,
// vf_rotate_nvcv.c:275, vf_smooth_nvcv.c:290-294 (third-party arithmetic, unpinned), defined
// instead by the in-tree CPU filters: vf_transpose.c:267-327, vf_hflip.c:89-117, vf_vflip.c:108-127,
// vf_convolution.c:495-512,555-569.  All HBM-bound byte permutations (2 B moved per byte of frame):
// rows enter and leave a block as dword runs, the permutation happens in LDS.
#include <hip/hip_runtime.h>
#include "common.h"
#include "kernels.h"

namespace gmat {

// ---- dword-granular row segment copy global -> LDS, safe at row ends ---------------------------
// copies bytes [b0, b0+n) of a source row into lds[0..n), reading aligned dwords where the whole
// dword lies inside [0, rowBytes) and single bytes elsewhere (out-of-row bytes read as the
// nearest in-row byte; callers overwrite halo pixels afterwards).
__device__ __forceinline__ void row_to_lds(const uint8_t *row, int rowBytes, int b0, int n, uint8_t *l,
                                           int lane, int nlanes, bool aligned)
{
    const int a0 = b0 & ~3;                         // may be negative: arithmetic & keeps floor semantics
    const int a1 = (b0 + n + 3) & ~3;
    for (int a = a0 + 4 * lane; a < a1; a += 4 * nlanes) {
        if (aligned && a >= 0 && a + 4 <= rowBytes && a >= b0 && a + 4 <= b0 + n) {
            const unsigned v = *reinterpret_cast<const unsigned *>(row + a);
            uint8_t *d = l + (a - b0);
            d[0] = (uint8_t)v; d[1] = (uint8_t)(v >> 8); d[2] = (uint8_t)(v >> 16); d[3] = (uint8_t)(v >> 24);
        } else {
            for (int i = 0; i < 4; i++) {
                const int b = a + i;
                if (b >= b0 && b < b0 + n) l[b - b0] = row[min(max(b, 0), rowBytes - 1)];
            }
        }
    }
}

// writes n bytes from LDS to a destination row segment as dwords where possible
__device__ __forceinline__ void lds_to_row(uint8_t *row, int b0, int n, const uint8_t *l, int lane, int nlanes,
                                           bool aligned)
{
    const int a0 = b0 & ~3, a1 = (b0 + n + 3) & ~3;
    for (int a = a0 + 4 * lane; a < a1; a += 4 * nlanes) {
        if (aligned && a >= b0 && a + 4 <= b0 + n) {
            const uint8_t *s = l + (a - b0);
            *reinterpret_cast<unsigned *>(row + a) =
                (unsigned)s[0] | ((unsigned)s[1] << 8) | ((unsigned)s[2] << 16) | ((unsigned)s[3] << 24);
        } else {
            for (int i = 0; i < 4; i++) {
                const int b = a + i;
                if (b >= b0 && b < b0 + n) row[b] = l[b - b0];
            }
        }
    }
}

// ---- transpose (+ optional source/destination vertical reversal = the four transpose dirs) -----
template <int BPP>
__global__ __launch_bounds__(256) void transpose_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds,
                                                        int inW, int inH, int dir, int aligned)
{
    constexpr int T = 64;
    constexpr int PITCH = T * BPP + 4;                  // +4 B: odd dword pitch, conflict-light columns
    __shared__ uint8_t tile[T * PITCH];
    __shared__ uint8_t orow[4][T * BPP];                // per-wave output row staging
    // input tile: columns [ix0, ix0+T) x rows [iy0, iy0+T) of the (possibly bottom-up) source
    const int ix0 = blockIdx.x * T, iy0 = blockIdx.y * T;
    const int tw = min(T, inW - ix0), th = min(T, inH - iy0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < th; r += 4) {
        const int sy = (dir & 1) ? inH - 1 - (iy0 + r) : iy0 + r;
        row_to_lds(src + (size_t)sy * ss, inW * BPP, ix0 * BPP, tw * BPP, tile + r * PITCH, lane, 64, aligned);
    }
    __syncthreads();
    // output: out(x = iy0 + r, y = ix0 + c) = tile[r][c]; out is inH wide, inW tall
    const int outH = inW;
    for (int c = wave; c < tw; c += 4) {
        for (int r = lane; r < th; r += 64)
            for (int b = 0; b < BPP; b++) orow[wave][r * BPP + b] = tile[r * PITCH + c * BPP + b];
        __builtin_amdgcn_wave_barrier();
        const int oy = (dir & 2) ? outH - 1 - (ix0 + c) : ix0 + c;
        lds_to_row(dst + (size_t)oy * ds, iy0 * BPP, th * BPP, orow[wave], lane, 64, aligned);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- flips: out(x, y) = in(fh ? w-1-x : x, fv ? h-1-y : y) -------------------------------------
template <int BPP>
__global__ __launch_bounds__(256) void flip_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds,
                                                   int w, int h, int fh, int fv, int aligned)
{
    constexpr int T = 256;                                // pixels per block row segment
    __shared__ uint8_t seg[4][T * BPP];
    __shared__ uint8_t out[4][T * BPP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int y = blockIdx.y * 4 + wave;
    const int x0 = blockIdx.x * T;
    if (y >= h) return;
    const int tw = min(T, w - x0);
    const int sy = fv ? h - 1 - y : y;
    const int sx0 = fh ? w - x0 - tw : x0;                // source segment start
    row_to_lds(src + (size_t)sy * ss, w * BPP, sx0 * BPP, tw * BPP, seg[wave], lane, 64, aligned);
    __builtin_amdgcn_wave_barrier();
    for (int p = lane; p < tw; p += 64) {
        const int sp = fh ? tw - 1 - p : p;
        for (int b = 0; b < BPP; b++) out[wave][p * BPP + b] = seg[wave][sp * BPP + b];
    }
    __builtin_amdgcn_wave_barrier();
    lds_to_row(dst + (size_t)y * ds, x0 * BPP, tw * BPP, out[wave], lane, 64, aligned);
}

// ---- 3x3 convolution with vf_convolution's borders; optional transposed store ------------------
// sum = sum_i c[i]*m[i];  out = clip_u8((int)(sum * rdiv + bias + 0.5f))   (vf_convolution.c:495-512)
// border (setup_3x3, :555-569): index -1 -> 1 (reflect-101), index n -> n-1 (edge repeated).
struct ConvParams { int m[9]; float rdiv, bias; };

template <int BPP, int TW, int TH, bool TRANSPOSED>
__global__ __launch_bounds__(256) void conv3x3_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds,
                                                      int w, int h, ConvParams cp, int aligned)
{
    constexpr int SP = (TW + 2) * BPP + 2;                 // source tile pitch (bytes)
    constexpr int RP = TW * BPP + (TRANSPOSED ? 4 : 0);    // result tile pitch
    __shared__ uint8_t st[(TH + 2) * SP];
    __shared__ uint8_t rt[TH * RP];
    __shared__ uint8_t orow[4][(TRANSPOSED ? TH : 1) * BPP];
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int tw = min(TW, w - x0), th = min(TH, h - y0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    for (int r = wave; r < th + 2; r += 4) {
        int yy = y0 + r - 1;
        yy = yy < 0 ? -yy : yy;
        yy = yy >= h ? 2 * h - 1 - yy : yy;
        yy = min(max(yy, 0), h - 1);
        row_to_lds(src + (size_t)yy * ss, w * BPP, (x0 - 1) * BPP, (tw + 2) * BPP, st + r * SP, lane, 64, aligned);
    }
    __syncthreads();
    // horizontal halo fix-up for tiles touching the frame's left / right edge
    if (x0 == 0 || x0 + tw == w) {
        for (int r = threadIdx.x; r < th + 2; r += 256) {
            uint8_t *row = st + r * SP;
            if (x0 == 0) {
                const int sp = w > 1 ? 2 : 1;              // pixel +1 sits at tile index 2; w==1: 2w-1-1 = 0 -> index 1
                for (int b = 0; b < BPP; b++) row[b] = row[sp * BPP + b];
            }
            if (x0 + tw == w)
                for (int b = 0; b < BPP; b++) row[(tw + 1) * BPP + b] = row[tw * BPP + b];
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < th * tw * BPP; i += 256) {
        const int r = i / (tw * BPP), cb = i - r * (tw * BPP);
        const uint8_t *p = st + r * SP + cb;               // top-left tap of the 3x3 window (this channel)
        int sum = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) sum += (int)p[(k / 3) * SP + (k % 3) * BPP] * cp.m[k];
        // separate multiply and adds: the CPU reference does not contract them into an fma
        const float f = __fadd_rn(__fadd_rn(__fmul_rn((float)sum, cp.rdiv), cp.bias), 0.5f);
        const int v = (int)f;
        rt[r * RP + cb] = (uint8_t)min(max(v, 0), 255);
    }
    __syncthreads();
    if (!TRANSPOSED) {
        for (int r = wave; r < th; r += 4)
            lds_to_row(dst + (size_t)(y0 + r) * ds, x0 * BPP, tw * BPP, rt + r * RP, lane, 64, aligned);
    } else {
        // out(x = y0 + r, y = x0 + c) = rt[r][c]
        for (int c = wave; c < tw; c += 4) {
            for (int r = lane; r < th; r += 64)
                for (int b = 0; b < BPP; b++) orow[wave][r * BPP + b] = rt[r * RP + c * BPP + b];
            __builtin_amdgcn_wave_barrier();
            lds_to_row(dst + (size_t)(x0 + c) * ds, y0 * BPP, th * BPP, orow[wave], lane, 64, aligned);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

static inline int al4(const void *a, int sa, const void *b, int sb)
{
    return ((((uintptr_t)a | (uintptr_t)sa | (uintptr_t)b | (uintptr_t)sb) & 3) == 0);
}

int launch_transpose(const uint8_t *src, int ss, uint8_t *dst, int ds, int inW, int inH, int bpp, int dir,
                     hipStream_t stream)
{
    if (inW <= 0 || inH <= 0) return 0;
    if (dir < 0 || dir > 3) return GMAT_ERR(EINVAL);
    const dim3 grid((inW + 63) / 64, (inH + 63) / 64), block(256);
    const int aligned = al4(src, ss, dst, ds);
    if (bpp == 3)      hipLaunchKernelGGL(transpose_kernel<3>, grid, block, 0, stream, src, ss, dst, ds, inW, inH, dir, aligned);
    else if (bpp == 4) hipLaunchKernelGGL(transpose_kernel<4>, grid, block, 0, stream, src, ss, dst, ds, inW, inH, dir, aligned);
    else return GMAT_ERR(ENOSYS);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_flip(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int bpp, int fh, int fv,
                hipStream_t stream)
{
    if (w <= 0 || h <= 0) return 0;
    const dim3 grid((w + 255) / 256, (h + 3) / 4), block(256);
    const int aligned = al4(src, ss, dst, ds);
    if (bpp == 3)      hipLaunchKernelGGL(flip_kernel<3>, grid, block, 0, stream, src, ss, dst, ds, w, h, fh, fv, aligned);
    else if (bpp == 4) hipLaunchKernelGGL(flip_kernel<4>, grid, block, 0, stream, src, ss, dst, ds, w, h, fh, fv, aligned);
    else return GMAT_ERR(ENOSYS);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_copy2d(const uint8_t *src, int ss, uint8_t *dst, int ds, int rowBytes, int h, hipStream_t stream)
{
    if (rowBytes <= 0 || h <= 0) return 0;
    GMAT_HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)ds, src, (size_t)ss, (size_t)rowBytes, (size_t)h,
                                    hipMemcpyDeviceToDevice, stream));
    return 0;
}

int launch_conv3x3(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int bpp, const int m[9],
                   float rdiv, float bias, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return 0;
    ConvParams cp;
    for (int i = 0; i < 9; i++) cp.m[i] = m[i];
    cp.rdiv = rdiv; cp.bias = bias;
    const dim3 grid((w + 63) / 64, (h + 15) / 16), block(256);
    const int aligned = al4(src, ss, dst, ds);
    if (bpp == 3)      hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_kernel<3, 64, 16, false>), grid, block, 0, stream, src, ss, dst, ds, w, h, cp, aligned);
    else if (bpp == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_kernel<4, 64, 16, false>), grid, block, 0, stream, src, ss, dst, ds, w, h, cp, aligned);
    else return GMAT_ERR(ENOSYS);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// rotate(90, clockwise) then horizontal flip is the plain transpose out(x, y) = in(y, x); the 3x3
// kernel 1 2 1 / 2 4 2 / 1 2 1 is symmetric and vf_convolution's border rule is the same on both
// axes, so smoothing commutes with the transpose: smooth the source tile, store it transposed.
int launch_rotate_flip_smooth(const uint8_t *src, int ss, uint8_t *dst, int ds, int inW, int inH, int bpp,
                              hipStream_t stream)
{
    if (inW <= 0 || inH <= 0) return 0;
    ConvParams cp;
    const int m[9] = {1, 2, 1, 2, 4, 2, 1, 2, 1};
    for (int i = 0; i < 9; i++) cp.m[i] = m[i];
    cp.rdiv = 1.0f / 16.0f; cp.bias = 0.0f;
    const dim3 grid((inW + 63) / 64, (inH + 63) / 64), block(256);
    const int aligned = al4(src, ss, dst, ds);
    if (bpp == 3)      hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_kernel<3, 64, 64, true>), grid, block, 0, stream, src, ss, dst, ds, inW, inH, cp, aligned);
    else if (bpp == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_kernel<4, 64, 64, true>), grid, block, 0, stream, src, ss, dst, ds, inW, inH, cp, aligned);
    else return GMAT_ERR(ENOSYS);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
