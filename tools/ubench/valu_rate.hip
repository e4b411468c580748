// valu_rate.hip — issue rate of the VALU instructions the strip kernels are made of, measured: 8 independent chains per lane,
// 4 waves per SIMD on every SIMD of the chip, cycles per instruction per wave from s_memtime around the loop of wave 0.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/bin/valu_rate && tools/bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef short short2v __attribute__((ext_vector_type(2)));

template <int OP>
__device__ __forceinline__ int op(int a, int b, int c)
{
    if constexpr (OP == 0) return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), c, true);
    else if constexpr (OP == 1) return (int)__builtin_amdgcn_perm((unsigned)a, (unsigned)c, 0x0C040C03u) + 0 * b;
    else if constexpr (OP == 2) return (c << 1) + a;                                // v_lshl_add_u32
    else if constexpr (OP == 3) return __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(c, a));
    else if constexpr (OP == 4) return min(max(c, a), b);                           // v_med3_i32
    else if constexpr (OP == 5) return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), c, false);
    else if constexpr (OP == 6) return c * a + b;                                   // v_mad / v_mul_lo
    else if constexpr (OP == 7) return (c >> 7) ^ a;
    else if constexpr (OP == 8) return __builtin_amdgcn_update_dpp(0, c, 0x138, 0xF, 0xF, true) + 0 * (a + b);   // v_mov_b32_dpp wave_shr:1
    else if constexpr (OP == 9) return __builtin_amdgcn_update_dpp(0, c, 0x130, 0xF, 0xF, true) + 0 * (a + b);   // wave_shl:1
    else if constexpr (OP == 10) return __builtin_amdgcn_update_dpp(0, c, 0x111, 0xF, 0xF, true) + 0 * (a + b);  // row_shr:1
    else if constexpr (OP == 11) return __builtin_amdgcn_update_dpp(0, c, 0x101, 0xF, 0xF, true) + 0 * (a + b);  // row_shl:1
    else if constexpr (OP == 12) return __builtin_amdgcn_update_dpp(0, c, 0x13C, 0xF, 0xF, true) + 0 * (a + b);  // wave_ror:1
    else if constexpr (OP == 13) return __builtin_amdgcn_ds_bpermute(a & 0xFC, c) + 0 * b;                       // ds_bpermute_b32
    else if constexpr (OP == 14) {
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(u16x2, c), __builtin_bit_cast(u16x2, a))) + 0 * b;   // v_pk_min_u16
    }
    else if constexpr (OP == 16) return (int)__umulhi((unsigned)c, (unsigned)a) + 0 * b;                         // v_mul_hi_u32
    else if constexpr (OP == 17) return (int)(((unsigned long long)(unsigned)c * (unsigned)a + ((unsigned long long)(unsigned)b << 32)) >> 32);   // v_mad_u64_u32, high half
    else if constexpr (OP == 18) return (int)((unsigned)c * (unsigned)a);                                        // v_mul_lo_u32
    else if constexpr (OP == 19) return __builtin_amdgcn_sdot4(a, b, c, false);                                  // v_dot4_i32_i8
    else if constexpr (OP == 20) return (int)__builtin_amdgcn_udot4((unsigned)a, (unsigned)b, (unsigned)c, false); // v_dot4_u32_u8
    else return (int)min(min((unsigned)c, (unsigned)a), (unsigned)b);                                           // v_min3_u32
}

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(int *out, long long *cyc, int iters, int a0, int b0)
{
    int c[8];
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = threadIdx.x + k;
    const int a = a0 + (int)threadIdx.x, b = b0;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int k = 0; k < 8; k++) c[k] = op<OP>(OP == 4 || OP == 2 || OP == 7 ? c[(k + 1) & 7] : a, b, c[k]);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    int s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s ^= c[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char *name, int wavesPerSimd)
{
    const int nblk = 256 * wavesPerSimd, iters = 4096;          // a 256-thread block = one wave per SIMD of a CU
    int *out; long long *cyc;
    hipMalloc(&out, (size_t)nblk * 256 * 4); hipMalloc(&cyc, (size_t)nblk * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(nblk), dim3(256), 0, 0, out, cyc, 64, 3, 5);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(nblk), dim3(256), 0, 0, out, cyc, iters, 3, 5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ninstr = (double)iters * 64;                    // per wave
    // all waves of a SIMD share it: per-SIMD instruction count = wavesPerSimd * ninstr
    const double ns_per_instr_simd = ms * 1e6 / (ninstr * wavesPerSimd);
    printf("%-28s %d waves/SIMD: %.3f ms, %.3f ns per wave-instruction per SIMD = %.2f cycles at 2.4 GHz\n", name, wavesPerSimd, ms, ns_per_instr_simd,
           ns_per_instr_simd * 2.4);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w : {1, 4}) {
        run<0>("v_dot2_i32_i16 clamp", w);
        run<5>("v_dot2_i32_i16", w);
        run<1>("v_perm_b32", w);
        run<2>("v_lshl_add_u32", w);
        run<3>("v_cvt_pk_i16_i32", w);
        run<4>("v_max_i32 + v_min_i32 (2 instr)", w);
        run<6>("v_mul_lo + add", w);
        run<7>("v_ashr + xor (2 instr)", w);
        run<8>("v_mov_b32_dpp wave_shr:1", w);
        run<9>("v_mov_b32_dpp wave_shl:1", w);
        run<10>("v_mov_b32_dpp row_shr:1", w);
        run<11>("v_mov_b32_dpp row_shl:1", w);
        run<12>("v_mov_b32_dpp wave_ror:1", w);
        run<13>("ds_bpermute_b32", w);
        run<14>("v_pk_min_u16", w);
        run<15>("v_min3_u32", w);
        run<16>("v_mul_hi_u32", w);
        run<19>("v_dot4_i32_i8", w);
        run<20>("v_dot4_u32_u8", w);
        run<17>("v_mad_u64_u32 (high half)", w);
        run<18>("v_mul_lo_u32", w);
    }
    return 0;
}
