// Tuning experiment (not product code): issue cost of the integer VALU instructions the scalers are made of, and the
// bandwidth of the strip kernel's load pattern (4-byte aligned dwordx4, neighbouring lanes overlapping by 8 bytes).
//   hipcc -O3 --offload-arch=gfx950 tools/exp/ubench.hip -o ubench && ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorName(e), __LINE__); exit(1);} } while (0)

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// each variant: 8 independent instructions, repeated 64 x ITER times per wave
template <int OP>
__global__ __launch_bounds__(256) void rate(unsigned long long *out, int iters, int seed)
{
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    int b = seed * 77 + 1, c = threadIdx.x ^ 0x55;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) asm volatile(REP64("v_dot2c_i32_i16 %0, %8, %9\n v_dot2c_i32_i16 %1, %8, %9\n v_dot2c_i32_i16 %2, %8, %9\n v_dot2c_i32_i16 %3, %8, %9\n v_dot2c_i32_i16 %4, %8, %9\n v_dot2c_i32_i16 %5, %8, %9\n v_dot2c_i32_i16 %6, %8, %9\n v_dot2c_i32_i16 %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c));
        if (OP == 1) asm volatile(REP64("v_dot2_i32_i16 %0, %9, %8, %0 clamp\n v_dot2_i32_i16 %1, %9, %8, %1 clamp\n v_dot2_i32_i16 %2, %9, %8, %2 clamp\n v_dot2_i32_i16 %3, %9, %8, %3 clamp\n v_dot2_i32_i16 %4, %9, %8, %4 clamp\n v_dot2_i32_i16 %5, %9, %8, %5 clamp\n v_dot2_i32_i16 %6, %9, %8, %6 clamp\n v_dot2_i32_i16 %7, %9, %8, %7 clamp\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c));
        if (OP == 2) asm volatile(REP64("v_perm_b32 %0, %0, %9, %8\n v_perm_b32 %1, %1, %9, %8\n v_perm_b32 %2, %2, %9, %8\n v_perm_b32 %3, %3, %9, %8\n v_perm_b32 %4, %4, %9, %8\n v_perm_b32 %5, %5, %9, %8\n v_perm_b32 %6, %6, %9, %8\n v_perm_b32 %7, %7, %9, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c));
        if (OP == 3) asm volatile(REP64("v_mad_i32_i24 %0, %0, %8, %9\n v_mad_i32_i24 %1, %1, %8, %9\n v_mad_i32_i24 %2, %2, %8, %9\n v_mad_i32_i24 %3, %3, %8, %9\n v_mad_i32_i24 %4, %4, %8, %9\n v_mad_i32_i24 %5, %5, %8, %9\n v_mad_i32_i24 %6, %6, %8, %9\n v_mad_i32_i24 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c));
        if (OP == 4) asm volatile(REP64("v_med3_i32 %0, %0, %8, %9\n v_med3_i32 %1, %1, %8, %9\n v_med3_i32 %2, %2, %8, %9\n v_med3_i32 %3, %3, %8, %9\n v_med3_i32 %4, %4, %8, %9\n v_med3_i32 %5, %5, %8, %9\n v_med3_i32 %6, %6, %8, %9\n v_med3_i32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c));
        if (OP == 5) asm volatile(REP64("v_ashrrev_i32 %0, 1, %0\n v_ashrrev_i32 %1, 1, %1\n v_ashrrev_i32 %2, 1, %2\n v_ashrrev_i32 %3, 1, %3\n v_ashrrev_i32 %4, 1, %4\n v_ashrrev_i32 %5, 1, %5\n v_ashrrev_i32 %6, 1, %6\n v_ashrrev_i32 %7, 1, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c));
        if (OP == 6) asm volatile(REP64("v_cvt_pk_i16_i32 %0, %0, %9\n v_cvt_pk_i16_i32 %1, %1, %9\n v_cvt_pk_i16_i32 %2, %2, %9\n v_cvt_pk_i16_i32 %3, %3, %9\n v_cvt_pk_i16_i32 %4, %4, %9\n v_cvt_pk_i16_i32 %5, %5, %9\n v_cvt_pk_i16_i32 %6, %6, %9\n v_cvt_pk_i16_i32 %7, %7, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c));
        if (OP == 7) asm volatile(REP64("v_add_u32 %0, %0, %9\n v_add_u32 %1, %1, %9\n v_add_u32 %2, %2, %9\n v_add_u32 %3, %3, %9\n v_add_u32 %4, %4, %9\n v_add_u32 %5, %5, %9\n v_add_u32 %6, %6, %9\n v_add_u32 %7, %7, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c));
        if (OP == 8) asm volatile(REP64("v_mov_b32 %0, %9\n v_mov_b32 %1, %9\n v_mov_b32 %2, %9\n v_mov_b32 %3, %9\n v_mov_b32 %4, %9\n v_mov_b32 %5, %9\n v_mov_b32 %6, %9\n v_mov_b32 %7, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c));
        if (OP == 9) asm volatile(REP64("v_pk_min_i16 %0, %0, %9\n v_pk_min_i16 %1, %1, %9\n v_pk_min_i16 %2, %2, %9\n v_pk_min_i16 %3, %3, %9\n v_pk_min_i16 %4, %4, %9\n v_pk_min_i16 %5, %5, %9\n v_pk_min_i16 %6, %6, %9\n v_pk_min_i16 %7, %7, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c));
        if (OP == 10) asm volatile(REP64("v_fma_f32 %0, %0, %9, %0\n v_fma_f32 %1, %1, %9, %1\n v_fma_f32 %2, %2, %9, %2\n v_fma_f32 %3, %3, %9, %3\n v_fma_f32 %4, %4, %9, %4\n v_fma_f32 %5, %5, %9, %5\n v_fma_f32 %6, %6, %9, %6\n v_fma_f32 %7, %7, %9, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c));
        if (OP == 11) asm volatile(REP64("v_dot4_i32_i8 %0, %9, %8, %0\n v_dot4_i32_i8 %1, %9, %8, %1\n v_dot4_i32_i8 %2, %9, %8, %2\n v_dot4_i32_i8 %3, %9, %8, %3\n v_dot4_i32_i8 %4, %9, %8, %4\n v_dot4_i32_i8 %5, %9, %8, %5\n v_dot4_i32_i8 %6, %9, %8, %6\n v_dot4_i32_i8 %7, %9, %8, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x7fffffff) out[0] = 0;
}

// MODE 0: 16 B per lane, aligned, no overlap (lane stride 16)        -> unique bytes = requested
// MODE 1: 16 B per lane at 8*lane - 4 (4-byte aligned, 8 B overlap)  -> unique = half
// MODE 2: 8 B per lane, aligned, no overlap
// MODE 3: MODE 1 + a 16 B and an 8 B load of a second plane at 8*lane - 8 / + 8 (the NV12 chroma pattern: 24 B per 8 unique)
template <int MODE>
__global__ __launch_bounds__(256) void ldpat(const uint8_t *src, const uint8_t *src2, unsigned *sink, int rowBytes, int rows, int rowsPerWave)
{
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int stripsPerRow = rowBytes / (MODE == 0 ? 1024 : 512);
    const int strip = wave % stripsPerRow, seg = wave / stripsPerRow;
    const int r0 = seg * rowsPerWave;
    if (r0 >= rows) return;
    unsigned acc = 0;
    typedef unsigned u4 __attribute__((ext_vector_type(4), aligned(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2), aligned(4)));
    const size_t col = MODE == 0 ? (size_t)strip * 1024 + 16 * lane : (size_t)strip * 512 + 8 * lane;
#pragma unroll 4
    for (int r = r0; r < r0 + rowsPerWave && r < rows; r++) {
        const uint8_t *p = src + (size_t)r * rowBytes + col;
        if (MODE == 0) { const u4 v = *(const u4 *)p; acc += v.x ^ v.y ^ v.z ^ v.w; }
        if (MODE == 1 || MODE == 3) { const u4 v = *(const u4 *)(p + (col ? -4 : 0)); acc += v.x ^ v.y ^ v.z ^ v.w; }
        if (MODE == 2) { const u2 v = *(const u2 *)p; acc += v.x ^ v.y; }
        if (MODE == 3 && !(r & 1)) {
            const uint8_t *q = src2 + (size_t)(r >> 1) * rowBytes + col;
            const u4 v = *(const u4 *)(q + (col ? -8 : 0)); const u2 w = *(const u2 *)(q + (col + 16 < (size_t)rowBytes ? 8 : 0));
            acc += v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y;
        }
    }
    if (acc == 0x12345678) sink[0] = acc;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("%s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    unsigned long long *out; CK(hipMalloc(&out, 1 << 20));
    const char *names[] = {"v_dot2c_i32_i16 (VOP2)", "v_dot2_i32_i16 clamp (VOP3P)", "v_perm_b32", "v_mad_i32_i24", "v_med3_i32", "v_ashrrev_i32", "v_cvt_pk_i16_i32",
                           "v_add_u32", "v_mov_b32", "v_pk_min_i16", "v_fma_f32", "v_dot4_i32_i8"};
    for (int wavesPerSimd : {1, 2, 4}) {
        const int blocks = prop.multiProcessorCount * wavesPerSimd, iters = 32;
        printf("-- %d wave(s) per SIMD: cycles per wave-instruction (per SIMD, throughput)\n", wavesPerSimd);
        for (int op = 0; op < 12; op++) {
            for (int rep = 0; rep < 2; rep++) {
                switch (op) {
#define R(N) case N: hipLaunchKernelGGL(rate<N>, dim3(blocks), dim3(256), 0, 0, out, iters, rep + 3); break;
                R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11)
                }
                CK(hipDeviceSynchronize());
            }
            std::vector<unsigned long long> h(blocks * 4);
            CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
            double s = 0; unsigned long long mn = ~0ull;
            for (auto v : h) { s += v; mn = v < mn ? v : mn; }
            const double n = 512.0 * iters;
            printf("   %-30s avg %.2f  min %.2f  (x%d waves: %.2f per instruction issued on the SIMD)\n", names[op], s / h.size() / n, mn / n, wavesPerSimd,
                   s / h.size() / n / wavesPerSimd);
        }
    }
    // ---- load patterns: a 3840-byte-row "plane" of 2160 x 64 rows (531 MB) -----------------------------------------
    {
        const int rowBytes = 4096, rows = 2048 * 64;
        uint8_t *src, *src2; unsigned *sink;
        CK(hipMalloc(&src, (size_t)rowBytes * rows + 64)); CK(hipMalloc(&src2, (size_t)rowBytes * rows / 2 + 64)); CK(hipMalloc(&sink, 64));
        CK(hipMemset(src, 1, (size_t)rowBytes * rows + 64)); CK(hipMemset(src2, 1, (size_t)rowBytes * rows / 2 + 64));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const char *mn[] = {"16 B aligned, no overlap", "16 B @4-aligned, 8 B overlap", "8 B aligned, no overlap", "luma pattern + NV12 chroma pattern"};
        for (int rowsPerWave : {32, 128})
        for (int mode = 0; mode < 4; mode++) {
            const int stripsPerRow = rowBytes / (mode == 0 ? 1024 : 512);
            const int waves = stripsPerRow * ((rows + rowsPerWave - 1) / rowsPerWave), blocks = (waves + 3) / 4;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0));
                switch (mode) {
                case 0: hipLaunchKernelGGL(ldpat<0>, dim3(blocks), dim3(256), 0, 0, src, src2, sink, rowBytes, rows, rowsPerWave); break;
                case 1: hipLaunchKernelGGL(ldpat<1>, dim3(blocks), dim3(256), 0, 0, src, src2, sink, rowBytes, rows, rowsPerWave); break;
                case 2: hipLaunchKernelGGL(ldpat<2>, dim3(blocks), dim3(256), 0, 0, src, src2, sink, rowBytes, rows, rowsPerWave); break;
                default: hipLaunchKernelGGL(ldpat<3>, dim3(blocks), dim3(256), 0, 0, src, src2, sink, rowBytes, rows, rowsPerWave); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            }
            const double bytes = (double)rowBytes * rows * (mode == 3 ? 1.5 : 1.0);
            printf("load %-36s rows/wave %3d: %7.3f ms  %7.1f GB/s unique\n", mn[mode], rowsPerWave, best, bytes / best / 1e6);
        }
    }
    return 0;
}
