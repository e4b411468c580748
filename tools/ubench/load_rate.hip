// load_rate.hip — what the vector-memory path charges for the load shapes of the strip kernels, measured on cache-resident data:
// every wave re-reads the same few KB (L1 / L2 hits), 8 loads in flight, 4 waves per SIMD on every CU.  Prints the time per wave-level
// load instruction per CU (the TA / L1 path is shared by a CU's four SIMDs).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/load_rate.hip -o tools/bin/load_rate && tools/bin/load_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef unsigned u32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3), aligned(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2), aligned(4)));

// W = dwords per lane per load, STRIDE = bytes between lanes, OFF = byte offset of lane 0
template <int W, int STRIDE, int OFF>
__global__ __launch_bounds__(256) void load_kernel(const uint8_t *src, unsigned *out, int iters, int rowBytes)
{
    const int lane = threadIdx.x & 63;
    const uint8_t *p = src + (size_t)(blockIdx.x & 15) * 65536 + (threadIdx.x >> 6) * 16384 + OFF + lane * STRIDE;
    unsigned acc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint8_t *q = p + ((i + r) & 7) * rowBytes;
            if constexpr (W == 4) { const u32x4 v = *reinterpret_cast<const u32x4 *>(q); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
            else if constexpr (W == 3) { const u32x3 v = *reinterpret_cast<const u32x3 *>(q); acc ^= v.x ^ v.y ^ v.z; }
            else if constexpr (W == 2) { const u32x2 v = *reinterpret_cast<const u32x2 *>(q); acc ^= v.x ^ v.y; }
            else acc ^= *reinterpret_cast<const unsigned *>(q);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int W, int STRIDE, int OFF>
static void run(const char *name, const uint8_t *src, unsigned *out)
{
    const int nblk = 256 * 4, iters = 2048;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((load_kernel<W, STRIDE, OFF>), dim3(nblk), dim3(256), 0, 0, src, out, 16, 1536);
    hipEventRecord(e0);
    hipLaunchKernelGGL((load_kernel<W, STRIDE, OFF>), dim3(nblk), dim3(256), 0, 0, src, out, iters, 1536);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double perCu = (double)iters * 8 * 16;                 // wave-level load instructions per CU: 4 blocks x 4 waves
    const double ns = ms * 1e6 / perCu;
    printf("%-44s %.3f ms  %.2f ns = %.1f cycles per wave-load per CU;  %.1f useful B/clk/CU\n", name, ms, ns, ns * 2.4,
           (double)(W * 4 < STRIDE ? W * 4 : STRIDE) * 64 / (ns * 2.4));
}

int main()
{
    uint8_t *src; unsigned *out;
    hipMalloc(&src, 16 * 65536 + 65536); hipMemset(src, 1, 16 * 65536 + 65536);
    hipMalloc(&out, 1024 * 256 * 4);
    run<4, 16, 0>("dwordx4  stride 16  16-aligned", src, out);
    run<4, 16, 4>("dwordx4  stride 16  off 4 (yuv2s luma)", src, out);
    run<4, 12, 0>("dwordx4  stride 12  (yuv3x1 plane, 1st)", src, out);
    run<1, 12, 16>("dword    stride 12  (yuv3x1 plane, 2nd)", src, out);
    run<3, 12, 0>("dwordx3  stride 12  contiguous", src, out);
    run<4, 12, 8>("dwordx4  stride 12  off 8", src, out);
    run<2, 12, 0>("dwordx2  stride 12", src, out);
    run<2, 8, 0>("dwordx2  stride 8   contiguous", src, out);
    run<2, 8, 4>("dwordx2  stride 8   off 4", src, out);
    run<1, 4, 0>("dword    stride 4   contiguous", src, out);
    run<3, 8, 0>("dwordx3  stride 8   (yuv2p plane)", src, out);
    run<4, 24, 0>("dwordx4  stride 24  (yuv3x1 UV: x4 + x4)", src, out);
    run<4, 8, 0>("dwordx4  stride 8", src, out);
    return 0;
}
