// rot_probe.hip — the hardware questions behind the fp32 form of the arbitrary-angle rotate (FINDINGS R4-rotate):
//   1. with MODE.fp_round = toward zero, is cvt(fma(fy', D', s0')) == floor of the exact 40-bit blend for every input?  (v_fma_f32,
//      v_pk_fma_f32; conversion by v_cvt_pk_u8_f32, by v_cvt_u32_f32, by a 2^23 bias)
//   2. what do v_cvt_f32_ubyteN, v_pk_fma_f32, v_pk_add_f32, v_cvt_pk_u8_f32, v_fma_f64, v_cvt_f64_i32, v_cvt_u32_f64, v_dot2_u32_u16 and
//      v_mad_i64_i32 cost per wave instruction?
//   3. ("lds" argument, run apart: may fault) does ds_read_b32 / ds_read_b64 take an address that is only 2-byte aligned?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/rot_probe.hip -o tools/bin/rot_probe && tools/bin/rot_probe [lds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));
typedef short short2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void round_mode(int m)          // MODE[1:0]: 0 nearest even, 1 +inf, 2 -inf, 3 toward zero
{
    if (m == 3) __builtin_amdgcn_s_setreg(1 | (0 << 6) | (1 << 11), 3);
    else __builtin_amdgcn_s_setreg(1 | (0 << 6) | (1 << 11), 0);
}

// ---- 1. semantics -----------------------------------------------------------------------------------------------------------
// one case = (s00, s01, s10, s11, fx, fy); exact = (((65536 - fy) * s0 + fy * s1) >> 32), s0 = (65536 - fx) * s00 + fx * s01
__global__ void sem_kernel(const unsigned *cases, int n, unsigned *bad, int mode)
{
    round_mode(mode);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned a = cases[2 * i], b = cases[2 * i + 1];
    const unsigned fx = b & 0xFFFF, fy = b >> 16;
    const long long s00 = a & 0xFF, s01 = (a >> 8) & 0xFF, s10 = (a >> 16) & 0xFF, s11 = a >> 24;
    const long long s0 = (65536 - (long long)fx) * s00 + fx * s01, s1 = (65536 - (long long)fx) * s10 + fx * s11;
    const unsigned want = (unsigned)(((65536 - (long long)fy) * s0 + fy * s1) >> 32);
    const float f00 = (float)(a & 0xFF), f01 = (float)((a >> 8) & 0xFF);
    const float f10 = (float)((a >> 16) & 0xFF), f11 = (float)(a >> 24);
    const float fxs = (float)fx * (1.0f / 65536.0f), fys = (float)fy * (1.0f / 65536.0f);
    float2v lo = {f00, f10}, hi = {f01, f11}, fx2 = {fxs, fxs};
    const float2v d = hi - lo;
    const float2v s = __builtin_elementwise_fma(fx2, d, lo);             // v_pk_fma_f32, exact
    const float D = s.y - s.x;
    const float v = __builtin_fmaf(fys, D, s.x);                         // one rounding, toward zero
    float2v fy2 = {fys, fys}, D2 = {D, D}, sx2 = {s.x, s.x};
    const float2v vp = __builtin_elementwise_fma(fy2, D2, sx2);          // the packed form of the same
    unsigned r_pk = 0;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %2" : "=v"(r_pk) : "v"(v), "v"(0u));
    const unsigned r_u32 = (unsigned)v;                                  // v_cvt_u32_f32: truncates
    const unsigned r_bias = __builtin_bit_cast(unsigned, v + 8388608.0f) & 0xFF;
    const unsigned r_pkf = (unsigned)vp.y;
    if ((r_pk & 0xFF) != want) atomicAdd(&bad[0], 1);
    if (r_u32 != want) atomicAdd(&bad[1], 1);
    if (r_bias != want) atomicAdd(&bad[2], 1);
    if (r_pkf != want) atomicAdd(&bad[3], 1);
    // the double form of the whole blend (v_fma_f64: exact), for the rate comparison's sake
    const double dv = __builtin_fma((double)fy * (1.0 / 65536.0), (double)D, (double)s.x);
    if ((unsigned)dv != want) atomicAdd(&bad[4], 1);
}

// ---- 2. rates ---------------------------------------------------------------------------------------------------------------
template <int OP>
__device__ __forceinline__ void step(float2v &c, float2v a, float2v b, double &dc, double da, int &ic, int ia)
{
    if constexpr (OP == 0) c = __builtin_elementwise_fma(a, b, c);                                  // v_pk_fma_f32
    else if constexpr (OP == 1) c = c + a;                                                           // v_pk_add_f32
    else if constexpr (OP == 2) c.x = __builtin_fmaf(a.x, b.x, c.x);                                 // v_fma_f32
    else if constexpr (OP == 3) c.x = (float)(((__builtin_bit_cast(unsigned, c.x) + (unsigned)ia) >> 8) & 0xFF);   // cvt + add
    else if constexpr (OP == 4) { unsigned r; asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(r) : "v"(a.x), "v"(ic)); ic = (int)r; }
    else if constexpr (OP == 5) dc = __builtin_fma(da, da, dc);                                      // v_fma_f64
    else if constexpr (OP == 6) dc = (double)(ic + (int)__builtin_bit_cast(long long, dc));          // v_cvt_f64_i32 + add
    else if constexpr (OP == 7) ic = (int)(unsigned)(dc + (double)ic);                               // cvt_f64_i32, add_f64, cvt_u32_f64
    else if constexpr (OP == 8) ic = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned short, ia),
                                                                   __builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned short, ic), (unsigned)ic, false);
    else if constexpr (OP == 9) { long long t = (long long)ic * ia + __builtin_bit_cast(long long, dc); dc = __builtin_bit_cast(double, t); }   // v_mad_i64_i32
    else if constexpr (OP == 10) ic = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, ia), __builtin_bit_cast(short2v, ic), ic, false);
    else if constexpr (OP == 11) c.x = (float)(ic & 0xFFFF) + c.x;                                   // cvt (sdwa?) + add
    else if constexpr (OP == 12) ic = (int)(unsigned)c.x + ic;                                       // v_cvt_u32_f32 + add
    else if constexpr (OP == 13) c.x = __builtin_floorf(c.x) + a.x;                                  // v_floor_f32 + add
}

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(int *out, int iters, float a0, int ia0)
{
    float2v c[8]; double dc[8]; int ic[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { c[k] = float2v{(float)threadIdx.x + k, 1.0f}; dc[k] = 1.0 + k + threadIdx.x; ic[k] = threadIdx.x + k; }
    const float2v a = {a0, a0 * 0.5f}, b = {0.999f, 0.998f};
    const double da = 0.99 + a0 * 1e-9;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int k = 0; k < 8; k++) step<OP>(c[k], a, b, dc[k], da, ic[k], ia0 + k);
        }
    }
    float s = 0; double ds = 0; int is = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { s += c[k].x + c[k].y; ds += dc[k]; is ^= ic[k]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)s + (int)ds + is;
}

template <int OP>
static void run(const char *name, int wavesPerSimd, int instrPerStep)
{
    const int nblk = 256 * wavesPerSimd, iters = 2048;
    int *out;
    hipMalloc(&out, (size_t)nblk * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(nblk), dim3(256), 0, 0, out, 64, 3.0f, 5);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(nblk), dim3(256), 0, 0, out, iters, 3.0f, 5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double nstep = (double)iters * 64;
    const double ns = ms * 1e6 / (nstep * wavesPerSimd);
    printf("%-44s %d waves/SIMD: %.3f ms, %.2f cycles a step at 2.4 GHz (%d instr a step: %.2f each)\n", name, wavesPerSimd, ms, ns * 2.4, instrPerStep,
           ns * 2.4 / instrPerStep);
    hipFree(out);
}

// ---- 3. LDS alignment -------------------------------------------------------------------------------------------------------
__global__ void lds_kernel(unsigned *out)
{
    __shared__ unsigned buf[256];
    buf[threadIdx.x] = 0x03020100u + 0x04040404u * threadIdx.x;           // byte i of the array holds i (mod 256)
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)buf + 4 * threadIdx.x;
    unsigned r2, r1, lo, hi;
    asm volatile("ds_read_b32 %0, %1 offset:2\n s_waitcnt lgkmcnt(0)" : "=v"(r2) : "v"(addr));
    asm volatile("ds_read_b32 %0, %1 offset:1\n s_waitcnt lgkmcnt(0)" : "=v"(r1) : "v"(addr));
    unsigned long long r64;
    asm volatile("ds_read_b64 %0, %1 offset:2\n s_waitcnt lgkmcnt(0)" : "=v"(r64) : "v"(addr));
    lo = (unsigned)r64; hi = (unsigned)(r64 >> 32);
    unsigned long long r64b;
    asm volatile("ds_read_b64 %0, %1 offset:4\n s_waitcnt lgkmcnt(0)" : "=v"(r64b) : "v"(addr));
    out[6 * threadIdx.x] = r2; out[6 * threadIdx.x + 1] = r1; out[6 * threadIdx.x + 2] = lo; out[6 * threadIdx.x + 3] = hi;
    out[6 * threadIdx.x + 4] = (unsigned)r64b; out[6 * threadIdx.x + 5] = (unsigned)(r64b >> 32);
}

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "lds")) {
        unsigned *o; hipMalloc(&o, 64 * 6 * 4);
        hipLaunchKernelGGL(lds_kernel, dim3(1), dim3(64), 0, 0, o);
        std::vector<unsigned> h(64 * 6);
        const hipError_t e = hipMemcpy(h.data(), o, h.size() * 4, hipMemcpyDeviceToHost);
        printf("lds probe: %s\n", hipGetErrorString(e));
        for (int t : {0, 1, 5}) printf("  lane %d: b32@+2 %08x  b32@+1 %08x  b64@+2 %08x %08x  b64@+4 %08x %08x\n", t, h[6 * t], h[6 * t + 1], h[6 * t + 2], h[6 * t + 3], h[6 * t + 4], h[6 * t + 5]);
        return 0;
    }
    {   // semantics: 2^24 random cases + the corners
        const int n = 1 << 24;
        std::vector<unsigned> c(2 * (size_t)n);
        uint64_t st = 0x9E3779B97F4A7C15ull;
        for (size_t i = 0; i < c.size(); i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; c[i] = (unsigned)(st >> 16); }
        const unsigned corner[] = {0u, 0xFFFFu, 0xFFFF0000u, 0xFFFFFFFFu, 1u, 0x10000u, 0x8000u, 0x80000000u, 0x7FFF7FFFu, 0x00010001u};
        int k = 0;
        for (unsigned px : {0u, 0xFFFFFFFFu, 0xFF00FF00u, 0x00FF00FFu, 0xFFFF0000u, 0x0000FFFFu, 0xFF0000FFu, 0x00FFFF00u, 0x01FE01FEu, 0x80FF7F00u})
            for (unsigned f : corner) { c[2 * k] = px; c[2 * k + 1] = f; k++; }
        unsigned *dc, *bad;
        hipMalloc(&dc, c.size() * 4); hipMalloc(&bad, 32);
        hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice);
        for (int mode : {3, 0}) {
            hipMemset(bad, 0, 32);
            hipLaunchKernelGGL(sem_kernel, dim3(n / 256), dim3(256), 0, 0, dc, n, bad, mode);
            unsigned h[8]; hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost);
            printf("round mode %d (%s): mismatches of %d cases: cvt_pk_u8 %u, cvt_u32 %u, 2^23 bias %u, pk_fma + cvt_u32 %u, f64 fma %u\n", mode,
                   mode == 3 ? "toward zero" : "nearest even", n, h[0], h[1], h[2], h[3], h[4]);
        }
    }
    for (int w : {4}) {
        run<0>("v_pk_fma_f32", w, 1);
        run<1>("v_pk_add_f32", w, 1);
        run<2>("v_fma_f32", w, 1);
        run<3>("v_cvt_f32_ubyte1 + v_add_u32", w, 2);
        run<4>("v_cvt_pk_u8_f32", w, 1);
        run<5>("v_fma_f64", w, 1);
        run<6>("v_cvt_f64_i32 + v_add_u32", w, 2);
        run<7>("v_cvt_f64_i32 + v_add_f64 + v_cvt_u32_f64", w, 3);
        run<8>("v_dot2_u32_u16", w, 1);
        run<9>("v_mad_i64_i32", w, 1);
        run<10>("v_dot2_i32_i16", w, 1);
        run<11>("cvt_f32 of a 16-bit field + v_add_f32", w, 2);
        run<12>("v_cvt_u32_f32 + v_add_u32", w, 2);
        run<13>("v_floor_f32 + v_add_f32", w, 2);
    }
    return 0;
}
