// wg_rate.hip — how fast the chip starts workgroups: an (almost) empty kernel of N workgroups, T threads each, L bytes of LDS each;
// microseconds per launch from HIP events over 200 back-to-back launches.  One 4K rgb24 frame in 32 x 32 pixel tiles is 8160 workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/wg_rate.hip -o tools/bin/wg_rate && tools/bin/wg_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int L>
__global__ void k(int *out, int work)
{
    __shared__ int lds[L / 4 > 0 ? L / 4 : 1];
    int v = threadIdx.x;
    if (L > 0) { lds[threadIdx.x] = v; __syncthreads(); v = lds[(threadIdx.x + 1) % blockDim.x]; }
    for (int i = 0; i < work; i++) v = v * 3 + 1;
    if (v == 0x12345678) out[0] = v;
}

template <int L>
static void run(int n, int t, int work)
{
    int *out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k<L>, dim3(n), dim3(t), 0, 0, out, work);
    hipEventRecord(e0);
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k<L>, dim3(n), dim3(t), 0, 0, out, work);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("workgroups %6d x %4d threads, LDS %5d B, work %4d: %7.2f us per launch = %6.2f ns per workgroup\n", n, t, L, work, ms * 5, ms * 5000 / n);
    hipFree(out);
}

int main()
{
    for (int n : {2040, 8160, 16320, 32640}) {
        run<0>(n, 256, 0); run<8192>(n, 256, 0); run<8192>(n, 128, 0); run<0>(n, 64, 0); run<8192>(n, 256, 200);
    }
    return 0;
}
