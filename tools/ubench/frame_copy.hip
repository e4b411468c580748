// tools/ubench/frame_copy.hip — what ONE 4K rgb24 frame per launch (24.9 MB in, 24.9 MB out) can reach, by kernel shape.
// Not product code.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/frame_copy.hip -o tools/bin/frame_copy
// The transform filters run one frame per launch (filter_frame()); their ceiling is not the 6.2 TB/s of a big streaming copy but what
// a 50 MB launch gets between two kernel boundaries.  Shapes:
//   empty      the launch-to-launch gap alone
//   oneshot    thread = one 16-byte load + store                                           (flip_direct_kernel's shape); oneshot_nt: the store non-temporal
//   loopP      thread = P 16-byte accesses, grid-stride
//   rows4      wave = 16 rows x 240 B, a dword per lane per row: 16 loads in flight, then 16 stores   (smooth121_kernel's shape)
//   rows16     wave = 16 rows x 1 KB, 16 bytes per lane per row
//   pipe       rows4 with two tiles per wave, the second tile's loads issued before the first tile's stores
// each at 1, 2, 4, 8 frames per launch (grid.y).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int W = 3840, H = 2160, PITCH = W * 3;           // 11520 bytes a row
constexpr size_t FB = (size_t)PITCH * H;
struct Frames { const uint8_t *s[8]; uint8_t *d[8]; };

__global__ void k_empty() {}
__global__ __launch_bounds__(256) void k_oneshot(Frames f)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i * 16 < FB) reinterpret_cast<uint4 *>(f.d[blockIdx.y])[i] = reinterpret_cast<const uint4 *>(f.s[blockIdx.y])[i];
}
// the same with a non-temporal store (round 3's last finding: px_math.h st_stream)
__global__ __launch_bounds__(256) void k_oneshot_nt(Frames f)
{
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i * 16 < FB) __builtin_nontemporal_store(reinterpret_cast<const v4u *>(f.s[blockIdx.y])[i], reinterpret_cast<v4u *>(f.d[blockIdx.y]) + i);
}
// ... and a non-temporal load as well
__global__ __launch_bounds__(256) void k_oneshot_nt2(Frames f)
{
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i * 16 < FB) __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const v4u *>(f.s[blockIdx.y]) + i), reinterpret_cast<v4u *>(f.d[blockIdx.y]) + i);
}
template <int P>
__global__ __launch_bounds__(256) void k_loop(Frames f)
{
    const size_t n = FB / 16, stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint4 v[P];
#pragma unroll
    for (int k = 0; k < P; k++) if (i + k * stride < n) v[k] = reinterpret_cast<const uint4 *>(f.s[blockIdx.y])[i + k * stride];
#pragma unroll
    for (int k = 0; k < P; k++) if (i + k * stride < n) reinterpret_cast<uint4 *>(f.d[blockIdx.y])[i + k * stride] = v[k];
}
// tiles of 60 dwords x 64 rows, 4 waves of 16 rows; 48 tile columns x 34 tile rows = 1632 tiles (rows 2160 = 33.75 tile rows)
__global__ __launch_bounds__(256) void k_rows4(Frames f)
{
    const int t = blockIdx.x, tx = t % 48, ty = t / 48, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane >= 60) return;
    const int y0 = ty * 64 + wave * 16;
    const uint8_t *s = f.s[blockIdx.y] + (size_t)tx * 240 + lane * 4;
    uint8_t *d = f.d[blockIdx.y] + (size_t)tx * 240 + lane * 4;
    unsigned v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) if (y0 + r < H) v[r] = *reinterpret_cast<const unsigned *>(s + (size_t)(y0 + r) * PITCH);
#pragma unroll
    for (int r = 0; r < 16; r++) if (y0 + r < H) *reinterpret_cast<unsigned *>(d + (size_t)(y0 + r) * PITCH) = v[r];
}
// tiles of 1 KB x 64 rows (the last tile column is 256 B wide: 11520 = 11 x 1024 + 256)
__global__ __launch_bounds__(256) void k_rows16(Frames f)
{
    const int t = blockIdx.x, tx = t % 12, ty = t / 12, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (tx * 1024 + lane * 16 >= PITCH) return;
    const int y0 = ty * 64 + wave * 16;
    const uint8_t *s = f.s[blockIdx.y] + (size_t)tx * 1024 + lane * 16;
    uint8_t *d = f.d[blockIdx.y] + (size_t)tx * 1024 + lane * 16;
    uint4 v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) if (y0 + r < H) v[r] = *reinterpret_cast<const uint4 *>(s + (size_t)(y0 + r) * PITCH);
#pragma unroll
    for (int r = 0; r < 16; r++) if (y0 + r < H) *reinterpret_cast<uint4 *>(d + (size_t)(y0 + r) * PITCH) = v[r];
}
// rows4, two vertically adjacent tiles per block, software-pipelined
__global__ __launch_bounds__(256) void k_pipe(Frames f)
{
    const int t = blockIdx.x, tx = t % 48, ty = (t / 48) * 2, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane >= 60) return;
    const uint8_t *s = f.s[blockIdx.y] + (size_t)tx * 240 + lane * 4;
    uint8_t *d = f.d[blockIdx.y] + (size_t)tx * 240 + lane * 4;
    unsigned a[16], b[16];
    const int ya = ty * 64 + wave * 16, yb = ya + 64;
#pragma unroll
    for (int r = 0; r < 16; r++) if (ya + r < H) a[r] = *reinterpret_cast<const unsigned *>(s + (size_t)(ya + r) * PITCH);
#pragma unroll
    for (int r = 0; r < 16; r++) if (yb + r < H) b[r] = *reinterpret_cast<const unsigned *>(s + (size_t)(yb + r) * PITCH);
#pragma unroll
    for (int r = 0; r < 16; r++) if (ya + r < H) *reinterpret_cast<unsigned *>(d + (size_t)(ya + r) * PITCH) = a[r];
#pragma unroll
    for (int r = 0; r < 16; r++) if (yb + r < H) *reinterpret_cast<unsigned *>(d + (size_t)(yb + r) * PITCH) = b[r];
}

// general tile copy: a wave covers N rows x (64 * sizeof(V)) bytes; a block's 4 waves sit side by side (HORIZ) or on top of each other;
// every load of the wave is issued before its first store.  XCD: blocks are renumbered so that each XCD (blockIdx % 8) walks a
// contiguous eighth of the frame in raster order.
template <typename V, int N, bool HORIZ, bool XCD>
__global__ __launch_bounds__(256) void k_tile(Frames f, int nbx, int nby)
{
    constexpr int WB = 64 * sizeof(V);                      // bytes a wave covers in a row
    int t = blockIdx.x;
    if (XCD) { const int nt = nbx * nby, chunk = (nt + 7) / 8; t = (t & 7) * chunk + (t >> 3); if (t >= nt) return; }
    const int bx = t % nbx, by = t / nbx, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = (HORIZ ? (bx * 4 + wave) : bx) * WB + lane * (int)sizeof(V);
    const int y0 = (HORIZ ? by : by * 4 + wave) * N;
    if (x >= PITCH) return;
    const uint8_t *s = f.s[blockIdx.y] + x;
    uint8_t *d = f.d[blockIdx.y] + x;
    V v[N];
#pragma unroll
    for (int r = 0; r < N; r++) if (y0 + r < H) v[r] = *reinterpret_cast<const V *>(s + (size_t)(y0 + r) * PITCH);
#pragma unroll
    for (int r = 0; r < N; r++) if (y0 + r < H) *reinterpret_cast<V *>(d + (size_t)(y0 + r) * PITCH) = v[r];
}
// the same walked as a stream: row r + 2 is requested before row r is stored (register ring of 3)
template <typename V, int N, bool XCD>
__global__ __launch_bounds__(256) void k_walk(Frames f, int nbx, int nby)
{
    constexpr int WB = 64 * sizeof(V);
    int t = blockIdx.x;
    if (XCD) { const int nt = nbx * nby, chunk = (nt + 7) / 8; t = (t & 7) * chunk + (t >> 3); if (t >= nt) return; }
    const int bx = t % nbx, by = t / nbx, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = (bx * 4 + wave) * WB + lane * (int)sizeof(V), y0 = by * N;
    if (x >= PITCH) return;
    const uint8_t *s = f.s[blockIdx.y] + x + (size_t)y0 * PITCH;
    uint8_t *d = f.d[blockIdx.y] + x + (size_t)y0 * PITCH;
    const int n = min(N, H - y0);
    V a = *reinterpret_cast<const V *>(s), b = *reinterpret_cast<const V *>(s + (size_t)min(1, n - 1) * PITCH);
    for (int r = 0; r < n; r++) {
        const V c = *reinterpret_cast<const V *>(s + (size_t)min(r + 2, n - 1) * PITCH);
        *reinterpret_cast<V *>(d + (size_t)r * PITCH) = a;
        a = b; b = c;
    }
}

int main()
{
    const int NSET = 16;
    std::vector<uint8_t *> src(NSET), dst(NSET);
    for (int i = 0; i < NSET; i++) { CK(hipMalloc(&src[i], FB)); CK(hipMalloc(&dst[i], FB)); CK(hipMemset(src[i], i + 1, FB)); CK(hipMemset(dst[i], 0, FB)); }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto frames = [&](int base, int nf) { Frames f; for (int k = 0; k < 8; k++) { f.s[k] = src[(base + k % nf) % NSET]; f.d[k] = dst[(base + k % nf) % NSET]; } return f; };
    auto timeit = [&](const char *name, int nf, auto &&launch) {
        for (int i = 0; i < 40; i++) launch(i, nf);                 // warm-up: clocks up
        CK(hipStreamSynchronize(st));
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 120; i++) launch(i, nf);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double us = best * 1e3 / 120 / nf;
        printf("%-10s frames/launch %d  %7.2f us/frame  %6.0f GB/s  frac %.3f\n", name, nf, us, 2.0 * FB / us / 1e3, 2.0 * FB / us / 1e3 / 8000.0);
        fflush(stdout);
    };
    timeit("empty", 1, [&](int, int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); });
    for (int nf : {1, 2, 4, 8}) {
        timeit("oneshot", nf, [&](int i, int n) { hipLaunchKernelGGL(k_oneshot, dim3((unsigned)((FB / 16 + 255) / 256), n), dim3(256), 0, st, frames(i * n, n)); });
        timeit("oneshot_nt2", nf, [&](int i, int n) { hipLaunchKernelGGL(k_oneshot_nt2, dim3((unsigned)((FB / 16 + 255) / 256), n), dim3(256), 0, st, frames(i * n, n)); });
        timeit("oneshot_nt", nf, [&](int i, int n) { hipLaunchKernelGGL(k_oneshot_nt, dim3((unsigned)((FB / 16 + 255) / 256), n), dim3(256), 0, st, frames(i * n, n)); });
        timeit("loop2", nf, [&](int i, int n) { hipLaunchKernelGGL(k_loop<2>, dim3((unsigned)((FB / 16 / 2 + 255) / 256), n), dim3(256), 0, st, frames(i * n, n)); });
        timeit("loop4", nf, [&](int i, int n) { hipLaunchKernelGGL(k_loop<4>, dim3((unsigned)((FB / 16 / 4 + 255) / 256), n), dim3(256), 0, st, frames(i * n, n)); });
        timeit("loop8", nf, [&](int i, int n) { hipLaunchKernelGGL(k_loop<8>, dim3((unsigned)((FB / 16 / 8 + 255) / 256), n), dim3(256), 0, st, frames(i * n, n)); });
        timeit("rows4", nf, [&](int i, int n) { hipLaunchKernelGGL(k_rows4, dim3(48 * 34, n), dim3(256), 0, st, frames(i * n, n)); });
        timeit("rows16", nf, [&](int i, int n) { hipLaunchKernelGGL(k_rows16, dim3(12 * 34, n), dim3(256), 0, st, frames(i * n, n)); });
        timeit("pipe", nf, [&](int i, int n) { hipLaunchKernelGGL(k_pipe, dim3(48 * 17, n), dim3(256), 0, st, frames(i * n, n)); });
    }
    auto tile = [&](const char *name, auto kern, int wb, int n, bool horiz) {
        const int nbx = horiz ? (PITCH + 4 * wb - 1) / (4 * wb) : (PITCH + wb - 1) / wb, nby = horiz ? (H + n - 1) / n : (H + 4 * n - 1) / (4 * n);
        for (int nf : {1, 4})
            timeit(name, nf, [&](int i, int m) { hipLaunchKernelGGL(kern, dim3(8 * ((nbx * nby + 7) / 8), m), dim3(256), 0, st, frames(i * m, m), nbx, nby); });
    };
#define TILE(V, VN, N) \
    tile("t" #VN "x" #N "H", k_tile<V, N, true, false>, 64 * sizeof(V), N, true); \
    tile("t" #VN "x" #N "Hx", k_tile<V, N, true, true>, 64 * sizeof(V), N, true); \
    tile("t" #VN "x" #N "V", k_tile<V, N, false, false>, 64 * sizeof(V), N, false); \
    tile("t" #VN "x" #N "Vx", k_tile<V, N, false, true>, 64 * sizeof(V), N, false);
    TILE(uint4, 16, 1) TILE(uint4, 16, 2) TILE(uint4, 16, 4) TILE(uint4, 16, 8)
    TILE(uint2, 8, 1) TILE(uint2, 8, 4) TILE(uint2, 8, 8)
    TILE(unsigned, 4, 1) TILE(unsigned, 4, 4) TILE(unsigned, 4, 16)
#define WALK(V, VN, N) \
    tile("w" #VN "x" #N, k_walk<V, N, false>, 64 * sizeof(V), N, true); \
    tile("w" #VN "x" #N "x", k_walk<V, N, true>, 64 * sizeof(V), N, true);
    WALK(uint4, 16, 8) WALK(uint4, 16, 16) WALK(uint4, 16, 32) WALK(uint2, 8, 16) WALK(unsigned, 4, 16)
    return 0;
}
