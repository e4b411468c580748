// k_transform.hip — crop / flip / rotate(transposes) / 3x3 smooth on packed pixels for gfx950.
//
// Stands in for the CV-CUDA operators behind vf_crop_nvcv.c:277, vf_flip_nvcv.c:251,
// vf_rotate_nvcv.c:275, vf_smooth_nvcv.c:290-294 (third-party arithmetic, unpinned), defined
// instead by the in-tree CPU filters: vf_transpose.c:267-327, vf_hflip.c:89-117, vf_vflip.c:108-127,
// vf_convolution.c:495-512,555-569.  All HBM-bound byte permutations (2 B moved per byte of frame):
// rows enter and leave a block as dword runs, the permutation happens in LDS.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <type_traits>
#include "common.h"
#include "kernels.h"
#include "px_math.h"

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
            if ((reinterpret_cast<uintptr_t>(d) & 3) == 0) {
                *reinterpret_cast<unsigned *>(d) = v;              // one ds_write_b32
            } else {
                d[0] = (uint8_t)v; d[1] = (uint8_t)(v >> 8); d[2] = (uint8_t)(v >> 16); d[3] = (uint8_t)(v >> 24);
            }
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
                (reinterpret_cast<uintptr_t>(s) & 3) == 0
                    ? *reinterpret_cast<const unsigned *>(s)                                   // one ds_read_b32
                    : ((unsigned)s[0] | ((unsigned)s[1] << 8) | ((unsigned)s[2] << 16) | ((unsigned)s[3] << 24));
        } else {
            for (int i = 0; i < 4; i++) {
                const int b = a + i;
                if (b >= b0 && b < b0 + n) row[b] = l[b - b0];
            }
        }
    }
}

// 16-byte tile loader.  Chunk c of tile row r = source bytes [a16 + 16c, a16 + 16c + 16) of row rowOf(r), stored at
// lds + r*pitch + 16c (pitch a multiple of 4).  CL lanes span a row's chunks (n16 <= CL), 256/CL rows per pass, K
// passes; ALL of a thread's loads are issued before its first LDS write — with the earlier dword-per-lane loader a
// 66-row tile cost three dependent HBM round trips per wave and the load phase alone took 21 of the 3x3 smooth's
// 37 us.  Chunks that stick out of [0, rowBytes) (tiles on the left / right frame edge) or unaligned sources are
// assembled bytewise with the column clamped.
template <int CL, int K, int NTH = 256, typename RowOf>
__device__ __forceinline__ void tile_to_lds16(const uint8_t *src, int ss, int rowBytes, int a16, int n16, int nrows,
                                              uint8_t *lds, int pitch, int tid, bool fast, RowOf rowOf)
{
    constexpr int RPP = NTH / CL;                          // rows per pass (NTH threads a tile)
    const int c = tid % CL, rb = tid / CL;
    const int a = a16 + 16 * c;
    const bool inside = fast && a >= 0 && a + 16 <= rowBytes;
    uint4 v[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int r = rb + RPP * k;
        v[k] = make_uint4(0u, 0u, 0u, 0u);
        if (c < n16 && r < nrows) {
            const uint8_t *row = src + (size_t)rowOf(r) * ss;
            if (inside) {
                v[k] = *reinterpret_cast<const uint4 *>(row + a);
            } else {
                unsigned w[4] = {0u, 0u, 0u, 0u};
                for (int i = 0; i < 16; i++) w[i >> 2] |= (unsigned)row[min(max(a + i, 0), rowBytes - 1)] << (8 * (i & 3));
                v[k] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int r = rb + RPP * k;
        if (c < n16 && r < nrows) {
            unsigned *d = reinterpret_cast<unsigned *>(lds + r * pitch + 16 * c);
            d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
        }
    }
}

// 16-byte tile store, the mirror of tile_to_lds16: tile row r (n bytes at lds + r*pitch) goes to byte offset b0 of
// destination row rowOf(r).  b0 is a multiple of 16 for every caller; whole chunks of 16-byte-aligned destinations
// are written with one global_store_dwordx4, the rest bytewise.
template <int CL, int K, typename RowOf>
__device__ __forceinline__ void lds_to_tile16(uint8_t *dst, int ds, int b0, int n, int nrows, const uint8_t *lds, int pitch,
                                              int tid, bool fast, RowOf rowOf)
{
    constexpr int RPP = 256 / CL;
    const int c = tid % CL, rb = tid / CL;
    const int n16 = (n + 15) >> 4;
    if (c >= n16) return;
    const bool whole = fast && 16 * c + 16 <= n;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int r = rb + RPP * k;
        if (r >= nrows) break;
        const unsigned *s = reinterpret_cast<const unsigned *>(lds + r * pitch + 16 * c);
        uint8_t *d = dst + (size_t)rowOf(r) * ds + b0 + 16 * c;
        if (whole) {
            *reinterpret_cast<uint4 *>(d) = make_uint4(s[0], s[1], s[2], s[3]);
        } else {
            const uint8_t *sb = reinterpret_cast<const uint8_t *>(s);
            for (int i = 0; i < min(16, n - 16 * c); i++) d[i] = sb[i];
        }
    }
}

// ---- transpose (+ optional source/destination vertical reversal = the four transpose dirs) -----
// out(x = iy, y = ix) = in(ix, iy); dir & 1 reads the source bottom-up, dir & 2 writes the destination bottom-up
// (vf_transpose.c:267-327).  Full tiles of 16-byte-aligned frames take a path without byte-granular LDS traffic:
//   BPP 3 / 4 — pixels sit in LDS as DWORDS (3-byte pixels are widened on the way in): the column read of the
//               transpose is one conflict-light ds_read_b32 per pixel, the output rows leave as 12 / 4 byte stores;
//   BPP 1     — the tile is cut into 4x4 byte blocks, each transposed in registers with 8 v_perm_b32; rows are
//               rotated by (row >> 2) dwords in LDS so the four row reads of a block are conflict-free.
// Everything else (partial tiles, unaligned frames, 2-byte samples) goes through the byte-wise path below it.
// NT: threads per tile.  128 x 128 tiles of 1- and 2-byte samples run with 1024 (full 128-byte lines on both sides AND as many waves in
// flight as the 64 x 64 tiling has: with 256 threads a 4K plane's 510 tiles left the chip short of waves — 12.4 us against 7.8).
template <int BPP, int T, int NT = 256>
__global__ __launch_bounds__(NT) void transpose_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds,
                                                        int inW, int inH, int dir, int aligned, OpFrames fr)
{
    src = fr.src[blockIdx.y]; dst = fr.dst[blockIdx.y];    // grid.y = frame
    constexpr int PITCH = T * BPP + 4;                  // +4 B: odd dword pitch, conflict-light columns
    constexpr int GENERIC_BYTES = T * PITCH + (NT / 64) * T * BPP;
    constexpr int PXP = T + 1;                          // dword-pixel tile pitch (BPP 3 / 4)
    constexpr int FAST_BYTES = BPP >= 3 ? T * PXP * 4 : T * T * BPP;
    constexpr int LDS_BYTES = GENERIC_BYTES > FAST_BYTES ? GENERIC_BYTES : FAST_BYTES;
    __shared__ __attribute__((aligned(16))) uint8_t smem[LDS_BYTES];
    // Tile order: workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2.  An output row segment of
    // one tile is only T*BPP bytes, so the tiles that complete a destination cache line are the neighbours along the
    // SOURCE row direction; giving every XCD a contiguous run of tiles in that order lets the partial lines merge in
    // one L2 instead of being written back separately by two (measured: 4K rgb24 transpose 18.3 -> 16.1 us).
    int tbx, tby;
    {
        const int nbx = (inW + T - 1) / T, nby = (inH + T - 1) / T, ntiles = nbx * nby;
        const int chunk = (ntiles + 7) >> 3;
        const int t = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
        if (t >= ntiles) return;
        tbx = t / nby;
        tby = t - tbx * nby;
    }
    // input tile: columns [ix0, ix0+T) x rows [iy0, iy0+T) of the (possibly bottom-up) source
    const int ix0 = tbx * T, iy0 = tby * T;
    const int tw = min(T, inW - ix0), th = min(T, inH - iy0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    const int outH = inW;
    auto srow = [&](int r) { return (dir & 1) ? inH - 1 - (iy0 + r) : iy0 + r; };
    auto orowOf = [&](int c) { return (dir & 2) ? outH - 1 - (ix0 + c) : ix0 + c; };
    const bool fast = tw == T && th == T && aligned &&
                      ((((uintptr_t)src | (uintptr_t)ss) & 15) == 0);           // block-uniform

    if (BPP >= 3 && fast) {
        static_assert(BPP < 3 || T == 64, "one 16-pixel group per thread");
        unsigned *px = reinterpret_cast<unsigned *>(smem);                      // [T][PXP] one dword per pixel
        {
            const int r = tid >> 2, g = tid & 3;                                // 64 rows x 4 groups of 16 pixels
            const uint8_t *p = src + (size_t)srow(r) * ss + (size_t)(ix0 + 16 * g) * BPP;
            unsigned o[16];
            if (BPP == 4) {
                const uint4 a0 = reinterpret_cast<const uint4 *>(p)[0], a1 = reinterpret_cast<const uint4 *>(p)[1],
                            a2 = reinterpret_cast<const uint4 *>(p)[2], a3 = reinterpret_cast<const uint4 *>(p)[3];
                o[0] = a0.x; o[1] = a0.y; o[2] = a0.z; o[3] = a0.w; o[4] = a1.x; o[5] = a1.y; o[6] = a1.z; o[7] = a1.w;
                o[8] = a2.x; o[9] = a2.y; o[10] = a2.z; o[11] = a2.w; o[12] = a3.x; o[13] = a3.y; o[14] = a3.z; o[15] = a3.w;
            } else {
                const uint4 a0 = reinterpret_cast<const uint4 *>(p)[0], a1 = reinterpret_cast<const uint4 *>(p)[1],
                            a2 = reinterpret_cast<const uint4 *>(p)[2];
                const unsigned w[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
#pragma unroll
                for (int q = 0; q < 4; q++) {                                   // 3 dwords -> 4 pixels [b0 b1 b2 0]
                    o[4 * q + 0] = __builtin_amdgcn_perm(w[3 * q + 1], w[3 * q + 0], 0x0C020100u);
                    o[4 * q + 1] = __builtin_amdgcn_perm(w[3 * q + 1], w[3 * q + 0], 0x0C050403u);
                    o[4 * q + 2] = __builtin_amdgcn_perm(w[3 * q + 2], w[3 * q + 1], 0x0C040302u);
                    o[4 * q + 3] = __builtin_amdgcn_perm(0u, w[3 * q + 2], 0x0C030201u);
                }
            }
#pragma unroll
            for (int i = 0; i < 16; i++) px[r * PXP + 16 * g + i] = o[i];
        }
        __syncthreads();
        if (BPP == 4) {
            // wave: 16 output rows; lane = source row = output x
#pragma unroll 4
            for (int k = 0; k < T / 4; k++) {
                const int c = wave * (T / 4) + k;
                reinterpret_cast<unsigned *>(dst + (size_t)orowOf(c) * ds)[iy0 + lane] = px[lane * PXP + c];
            }
        } else {
            // 16 lanes x 4 consecutive source rows = one output row of 64 pixels; a wave covers 4 output rows
            const int cc = lane >> 4, rq = lane & 15;
#pragma unroll
            for (int k = 0; k < T / 16; k++) {
                const int c = 16 * k + 4 * wave + cc;
                const unsigned A = px[(4 * rq + 0) * PXP + c], B = px[(4 * rq + 1) * PXP + c],
                               Cc = px[(4 * rq + 2) * PXP + c], D = px[(4 * rq + 3) * PXP + c];
                uint3 o3;
                o3.x = __builtin_amdgcn_perm(B, A, 0x04020100u);               // A0 A1 A2 B0
                o3.y = __builtin_amdgcn_perm(Cc, B, 0x05040201u);              // B1 B2 C0 C1
                o3.z = __builtin_amdgcn_perm(D, Cc, 0x06050402u);              // C2 D0 D1 D2
                *reinterpret_cast<uint3 *>(dst + (size_t)orowOf(c) * ds + (size_t)(iy0 + 4 * rq) * 3) = o3;
            }
        }
        return;
    }
    if (BPP == 2 && fast) {
        // 2-byte samples (a UV plane, 16-bit gray): blocks of 2 x 2 samples.  One dword of a source row holds two samples;
        // the dwords of rows 2*rblk and 2*rblk + 1 at column q give the dwords of output rows 2*q and 2*q + 1 at byte 4*rblk,
        // so a wave's store is one contiguous run per output row.
        constexpr int DW = T / 2;                                               // dwords per tile row
        constexpr int CPR = DW / 4, RPP = NT / CPR, NPASS = T / RPP;            // 16-byte chunks per row, rows per pass, passes
        unsigned *t32 = reinterpret_cast<unsigned *>(smem);                     // [T rows][DW dwords], row R rotated by R >> 1
        {
            const int c = tid % CPR, rb = tid / CPR;
            uint4 v[NPASS];
#pragma unroll
            for (int k = 0; k < NPASS; k++)
                v[k] = *reinterpret_cast<const uint4 *>(src + (size_t)srow(rb + RPP * k) * ss + (size_t)ix0 * 2 + 16 * c);
#pragma unroll
            for (int k = 0; k < NPASS; k++) {
                const int R = rb + RPP * k, rot = R >> 1;
                unsigned *row = t32 + R * DW;
                row[(4 * c + 0 + rot) & (DW - 1)] = v[k].x; row[(4 * c + 1 + rot) & (DW - 1)] = v[k].y;
                row[(4 * c + 2 + rot) & (DW - 1)] = v[k].z; row[(4 * c + 3 + rot) & (DW - 1)] = v[k].w;
            }
        }
        __syncthreads();
        constexpr int NW = NT / 64;
        constexpr int NR = T / 2, QPW = 64 / NR, NIT = DW / (NW * QPW);          // row pairs; q's per wave and pass; passes
        const int rblk = lane & (NR - 1);
#pragma unroll 4
        for (int it = 0; it < NIT; it++) {
            const int q = it * NW * QPW + wave * QPW + lane / NR;
            const unsigned d0 = t32[(2 * rblk + 0) * DW + ((q + rblk) & (DW - 1))], d1 = t32[(2 * rblk + 1) * DW + ((q + rblk) & (DW - 1))];
            const size_t xb = (size_t)iy0 * 2 + 4 * rblk;
            st_stream(dst + (size_t)orowOf(2 * q + 0) * ds + xb, __builtin_amdgcn_perm(d1, d0, 0x05040100u));
            st_stream(dst + (size_t)orowOf(2 * q + 1) * ds + xb, __builtin_amdgcn_perm(d1, d0, 0x07060302u));
        }
        return;
    }
    if (BPP == 1 && fast) {
        static_assert(BPP != 1 || T == 128 || T == 64, "NB x NB blocks of 4 x 4 bytes, NB = 32 | 16");
        constexpr int NB = T / 4;                                               // blocks (= dwords) per tile row
        constexpr int CPR = T / 16, RPP = NT / CPR, NPASS = T / RPP;           // 16-byte chunks per row, rows per pass of the NT threads, passes
        unsigned *t32 = reinterpret_cast<unsigned *>(smem);                     // [T rows][NB dwords], row R rotated by R >> 2
        {
            const int c = tid % CPR, rb = tid / CPR;
            uint4 v[NPASS];
#pragma unroll
            for (int k = 0; k < NPASS; k++)
                v[k] = *reinterpret_cast<const uint4 *>(src + (size_t)srow(rb + RPP * k) * ss + ix0 + 16 * c);
#pragma unroll
            for (int k = 0; k < NPASS; k++) {
                const int R = rb + RPP * k, rot = R >> 2;
                unsigned *row = t32 + R * NB;
                row[(4 * c + 0 + rot) & (NB - 1)] = v[k].x; row[(4 * c + 1 + rot) & (NB - 1)] = v[k].y;
                row[(4 * c + 2 + rot) & (NB - 1)] = v[k].z; row[(4 * c + 3 + rot) & (NB - 1)] = v[k].w;
            }
        }
        __syncthreads();
        // block (rblk, q): source rows 4*rblk .. +3, dword column q  ->  output rows 4*q .. +3, bytes 4*rblk .. +3
        constexpr int NW = NT / 64;
        constexpr int QPW = 64 / NB, NIT = NB / (NW * QPW);                      // q's per wave and pass; passes
        const int rblk = lane & (NB - 1);
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int q = it * NW * QPW + wave * QPW + lane / NB;
            const unsigned d0 = t32[(4 * rblk + 0) * NB + ((q + rblk) & (NB - 1))], d1 = t32[(4 * rblk + 1) * NB + ((q + rblk) & (NB - 1))],
                           d2 = t32[(4 * rblk + 2) * NB + ((q + rblk) & (NB - 1))], d3 = t32[(4 * rblk + 3) * NB + ((q + rblk) & (NB - 1))];
            const unsigned lo01 = __builtin_amdgcn_perm(d1, d0, 0x05010400u), hi01 = __builtin_amdgcn_perm(d1, d0, 0x07030602u);
            const unsigned lo23 = __builtin_amdgcn_perm(d3, d2, 0x05010400u), hi23 = __builtin_amdgcn_perm(d3, d2, 0x07030602u);
            const unsigned o0 = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u), o1 = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
            const unsigned o2 = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u), o3 = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
            const size_t xb = (size_t)iy0 + 4 * rblk;
            *reinterpret_cast<unsigned *>(dst + (size_t)orowOf(4 * q + 0) * ds + xb) = o0;
            *reinterpret_cast<unsigned *>(dst + (size_t)orowOf(4 * q + 1) * ds + xb) = o1;
            *reinterpret_cast<unsigned *>(dst + (size_t)orowOf(4 * q + 2) * ds + xb) = o2;
            *reinterpret_cast<unsigned *>(dst + (size_t)orowOf(4 * q + 3) * ds + xb) = o3;
        }
        return;
    }

    // ---- byte-wise path --------------------------------------------------------------------------------
    uint8_t *tile = smem;
    uint8_t (*orow)[T * BPP] = reinterpret_cast<uint8_t (*)[T * BPP]>(smem + T * PITCH);   // per-wave output row staging
    {
        // ix0 * BPP is a multiple of 16 for every (BPP, T) instantiated, so tile byte 0 is chunk-aligned
        static_assert((T * BPP) % 16 == 0 && T * BPP <= 256, "tile rows are whole 16-byte chunks");
        constexpr int CL = T * BPP / 16 <= 8 ? 8 : 16;
        constexpr int K = (T + NT / CL - 1) / (NT / CL);
        const bool fast16 = ((((uintptr_t)src | (uintptr_t)ss) & 15) == 0);
        tile_to_lds16<CL, K, NT>(src, ss, inW * BPP, ix0 * BPP, (tw * BPP + 15) >> 4, th, tile, PITCH, tid, fast16, srow);
    }
    __syncthreads();
    // output: out(x = iy0 + r, y = ix0 + c) = tile[r][c]; out is inH wide, inW tall
    for (int c = wave; c < tw; c += NT / 64) {
        for (int r = lane; r < th; r += 64)
            for (int b = 0; b < BPP; b++) orow[wave][r * BPP + b] = tile[r * PITCH + c * BPP + b];
        __builtin_amdgcn_wave_barrier();
        lds_to_row(dst + (size_t)orowOf(c) * ds, iy0 * BPP, th * BPP, orow[wave], lane, 64, aligned);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- flips: out(x, y) = in(fh ? w-1-x : x, fv ? h-1-y : y) -------------------------------------
template <int BPP>
__global__ __launch_bounds__(256) void flip_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds,
                                                   int w, int h, int fh, int fv, int aligned, OpFrames fr)
{
    src = fr.src[blockIdx.z]; dst = fr.dst[blockIdx.z];    // grid.z = frame
    constexpr int T = 256;                                // pixels per block row segment
    __shared__ __attribute__((aligned(16))) uint8_t seg[4][T * BPP];
    __shared__ __attribute__((aligned(16))) uint8_t out[4][T * BPP];
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

// Direct (LDS-free) flips for the common aligned case (rows dword aligned, width a multiple of 4 pixels):
// a thread moves 4 pixels; the horizontal mirror of a 4-pixel group is another aligned 4-pixel group, whose
// pixel order is reversed in registers with v_perm_b32.  A wave reads and writes 768 (rgb24) / 1024 (rgba)
// contiguous bytes per instruction.
template <int BPP>
__global__ __launch_bounds__(256) void flip_direct_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds,
                                                          int w, int h, int fh, int fv, OpFrames fr)
{
    src = fr.src[blockIdx.z]; dst = fr.dst[blockIdx.z];    // grid.z = frame
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const int sy = fv ? h - 1 - y : y, sx = fh ? w - 4 - x : x;
    const uint8_t *s = src + (size_t)sy * ss + (size_t)sx * BPP;
    uint8_t *d = dst + (size_t)y * ds + (size_t)x * BPP;
    if (BPP == 4) {
        const uint4 v = ld_stream(s, uint4());
        st_stream(d, fh ? make_uint4(v.w, v.z, v.y, v.x) : v);
    } else {
        const uint3 v = ld_stream(s, uint3());
        uint3 o = v;
        if (fh) {
            // pixels a b c d = bytes a0 a1 a2 b0 | b1 b2 c0 c1 | c2 d0 d1 d2  ->  d c b a
            o.x = __builtin_amdgcn_perm(v.y, v.z, 0x06030201u);                       // d0 d1 d2 c0
            const unsigned lo = __builtin_amdgcn_perm(v.z, v.y, 0x0C0C0403u);         // c1 c2
            const unsigned hi = __builtin_amdgcn_perm(v.y, v.x, 0x0C0C0403u);         // b0 b1
            o.y = __builtin_amdgcn_perm(hi, lo, 0x05040100u);                         // c1 c2 b0 b1
            o.z = __builtin_amdgcn_perm(v.y, v.x, 0x02010005u);                       // b2 a0 a1 a2
        }
        st_stream(d, o);
    }
}

// ---- 3x3 convolution with vf_convolution's borders; optional transposed store ------------------
// sum = sum_i c[i]*m[i];  out = clip_u8((int)(sum * rdiv + bias + 0.5f))   (vf_convolution.c:495-512)
// border (setup_3x3, :555-569): index -1 -> 1 (reflect-101), index n -> n-1 (edge repeated).
struct ConvParams { int m[9]; float rdiv, bias; int shift; unsigned half; };   // shift >= 0: rdiv == 2^-shift, bias == 0

// gathers bytes (i0, i1, i2) of the 12-byte window {w0, w1, w2} into one dword [b(i0), b(i1), b(i2), 0]
template <int I0, int I1, int I2>
__device__ __forceinline__ unsigned gather3(unsigned w0, unsigned w1, unsigned w2)
{
    static_assert(I0 < I1 && I1 < I2 && I2 < 12, "window indices");
    if (I2 < 8) return __builtin_amdgcn_perm(w1, w0, 0x0C000000u | (I2 << 16) | (I1 << 8) | I0);
    if (I0 >= 4) return __builtin_amdgcn_perm(w2, w1, 0x0C000000u | ((I2 - 4) << 16) | ((I1 - 4) << 8) | (I0 - 4));
    // spans all three dwords: first (I0, I1) from {w0,w1}, then add I2 from w2
    const unsigned t = __builtin_amdgcn_perm(w1, w0, 0x0C0C0000u | (I1 << 8) | I0);
    return __builtin_amdgcn_perm(w2, t, 0x0C000100u | ((I2 - 8 + 4) << 16));
}

template <int BPP, int TW, int TH, bool TRANSPOSED>
__global__ __launch_bounds__(256) void conv3x3_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds,
                                                      int w, int h, ConvParams cp, int aligned, int fastCoef)
{
    // The source tile holds pixels [x0-1, x0+TW] of rows [y0-1, y0+TH].  x0 is a multiple of 64, so the first
    // source byte (x0-1)*BPP is (16-BPP) (mod 16) — 15, 14, 13, 12 for 1-, 2-, 3-, 4-byte pixels: the LDS row keeps
    // that offset (OFF16), so that 16-byte global chunks land on 16-byte LDS chunks (tile_to_lds16) and global
    // dwords on LDS dwords (SHIFT = OFF16 mod 4 is the byte position of the first tile byte inside its dword).
    constexpr int OFF16 = (16 - BPP) & 15;
    constexpr int SHIFT = OFF16 & 3;
    constexpr int N16 = (OFF16 + (TW + 2) * BPP + 15) / 16;            // 16-byte chunks per tile row
    constexpr int SP = N16 * 16 + 16;                                  // source tile pitch (compute reads 2 dwords past)
    constexpr int RP = ((TW * BPP + 3) / 4) * 4 + 4;                   // result tile pitch
    static_assert((RP / 4) % 2 == 1, "odd dword pitch keeps column accesses conflict-light");
    __shared__ __attribute__((aligned(16))) uint8_t lds16[(TH + 2) * SP];
    uint8_t *st_raw = lds16 + (OFF16 & ~3);                            // dword base of the tile window
    // result tile: rows of RP bytes, or — for the transposed store — one row of PT bytes per tile COLUMN
    constexpr int PT = ((TH * BPP + 3) / 4) * 4 + 4;
    static_assert((PT / 4) % 2 == 1, "odd dword pitch");
    // the plain (row-major) store goes straight from registers to global memory: no result tile, one barrier less,
    // and with 15 KB of LDS all eight tiles of a CU are resident at once
    constexpr int RT_BYTES = TRANSPOSED ? TW * PT : 16;
    static_assert((TW * BPP + 3) / 4 <= 64, "one lane per dword column of the tile");
    __shared__ __attribute__((aligned(16))) uint8_t rt[RT_BYTES];
    uint8_t *st = st_raw + SHIFT;                                      // = lds16 + OFF16: tile byte 0 of row 0
    // XCD-aware tile order (see transpose_kernel): each XCD takes a contiguous run of tiles along the direction in
    // which neighbouring tiles share destination cache lines — the output row for the plain store, the source
    // column for the transposed store
    int tbx, tby;
    {
        const int nbx = (w + TW - 1) / TW, nby = (h + TH - 1) / TH, ntiles = nbx * nby;
        const int chunk = (ntiles + 7) >> 3;
        const int t = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
        if (t >= ntiles) return;
        if (TRANSPOSED) { tbx = t / nby; tby = t - tbx * nby; }
        else            { tby = t / nbx; tbx = t - tby * nbx; }
    }
    const int x0 = tbx * TW, y0 = tby * TH;
    const int tw = min(TW, w - x0), th = min(TH, h - y0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        constexpr int CL = N16 <= 16 ? 16 : 32;
        constexpr int K = (TH + 2 + 256 / CL - 1) / (256 / CL);
        static_assert(N16 <= CL, "chunks per row");
        const bool fast16 = ((((uintptr_t)src | (uintptr_t)ss) & 15) == 0);
        const int b0 = (x0 - 1) * BPP;                                 // first tile byte; b0 - OFF16 is a multiple of 16
        tile_to_lds16<CL, K>(src, ss, w * BPP, b0 - OFF16, (OFF16 + (tw + 2) * BPP + 15) >> 4, th + 2, lds16, SP,
                             (int)threadIdx.x, fast16, [&](int r) {
                                 int yy = y0 + r - 1;
                                 yy = yy < 0 ? -yy : yy;
                                 yy = yy >= h ? 2 * h - 1 - yy : yy;
                                 return min(max(yy, 0), h - 1);
                             });
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
    const int rowDwords = (tw * BPP + 3) >> 2;
    // result byte (r, cb) -> LDS: row-major, or transposed at pixel granularity
    auto put_dword = [&](int r, int d, unsigned o) {
        if (!TRANSPOSED) {
            // x0 * BPP is a multiple of 4: dword d of the tile row is an aligned dword of the destination row
            uint8_t *p = dst + (size_t)(y0 + r) * ds + (size_t)x0 * BPP + 4 * d;
            if (aligned && 4 * d + 4 <= tw * BPP) *reinterpret_cast<unsigned *>(p) = o;
            else for (int j = 0; j < 4 && 4 * d + j < tw * BPP; j++) p[j] = (uint8_t)(o >> (8 * j));
        } else if (BPP == 4) {
            *reinterpret_cast<unsigned *>(rt + d * PT + r * 4) = o;               // dword d is pixel d
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int cb = 4 * d + j, px = cb / BPP, b = cb - px * BPP;
                if (cb < tw * BPP) rt[px * PT + r * BPP + b] = (uint8_t)(o >> (8 * j));
            }
        }
    };
    if (fastCoef) {
        // coefficients in [0,255].  Lane = one DWORD column of the tile (4 channel bytes), wave = TH/4 consecutive
        // rows.  The lane walks down its column: each source row is read once (3 dwords), its horizontal tap
        // triples are gathered once with v_perm_b32 (window bytes SHIFT+j, +BPP, +2*BPP) and multiplied with
        // v_dot4_u32_u8 against the three matrix rows, feeding the three output rows it contributes to.
        const unsigned m0 = (unsigned)cp.m[0] | ((unsigned)cp.m[1] << 8) | ((unsigned)cp.m[2] << 16);
        const unsigned m1 = (unsigned)cp.m[3] | ((unsigned)cp.m[4] << 8) | ((unsigned)cp.m[5] << 16);
        const unsigned m2 = (unsigned)cp.m[6] | ((unsigned)cp.m[7] << 8) | ((unsigned)cp.m[8] << 16);
        constexpr int RPW = (TH + 3) / 4;
        const int r0 = wave * RPW, nrows = min(RPW, th - r0);          // wave-uniform
        const int d = lane;
        // accumulators start at the rounding constant of the integer epilogue (0 for the float epilogue)
        const unsigned init = cp.shift >= 0 ? cp.half : 0u;
        auto emit = [&](int r, const unsigned (&sum)[4]) {
            unsigned o = 0;
            if (cp.shift >= 0) {
                // rdiv == 2^-shift and bias == 0: (int)(sum * rdiv + 0.5f) == (sum + 2^(shift-1)) >> shift exactly
                // (sum < 2^24 is exact in float, the product and the + 0.5f are exact too)
#pragma unroll
                for (int j = 0; j < 4; j++) o |= min(sum[j] >> cp.shift, 255u) << (8 * j);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    // separate multiply and adds: the CPU reference does not contract them into an fma
                    const float f = __fadd_rn(__fadd_rn(__fmul_rn((float)(int)sum[j], cp.rdiv), cp.bias), 0.5f);
                    o |= (unsigned)min(max((int)f, 0), 255) << (8 * j);
                }
            }
            put_dword(r, d, o);
        };
        if (d < rowDwords && nrows > 0) {
            unsigned A[4], B[4], Cc[4];
#pragma unroll
            for (int j = 0; j < 4; j++) A[j] = B[j] = Cc[j] = init;
            // step(s): source tile row r0 + s; X = output row s-2 (gets matrix row 2, completes), Y = row s-1
            // (matrix row 1), Z = row s (matrix row 0, starts)
#define GMAT_CONV_STEP(s_, X, Y, Z)                                                                              \
            if ((s_) < nrows + 2) {                                                                              \
                const unsigned *p = reinterpret_cast<const unsigned *>(st_raw + (r0 + (s_)) * SP) + d;           \
                const unsigned w0 = p[0], w1 = p[1], w2 = p[2];                                                  \
                unsigned g[4];                                                                                   \
                g[0] = gather3<SHIFT + 0, SHIFT + 0 + BPP, SHIFT + 0 + 2 * BPP>(w0, w1, w2);                     \
                g[1] = gather3<SHIFT + 1, SHIFT + 1 + BPP, SHIFT + 1 + 2 * BPP>(w0, w1, w2);                     \
                g[2] = gather3<SHIFT + 2, SHIFT + 2 + BPP, SHIFT + 2 + 2 * BPP>(w0, w1, w2);                     \
                g[3] = gather3<SHIFT + 3, SHIFT + 3 + BPP, SHIFT + 3 + 2 * BPP>(w0, w1, w2);                     \
                _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                  \
                    X[j] = __builtin_amdgcn_udot4(g[j], m2, X[j], false);                                        \
                    Y[j] = __builtin_amdgcn_udot4(g[j], m1, Y[j], false);                                        \
                    Z[j] = __builtin_amdgcn_udot4(g[j], m0, init, false);                                        \
                }                                                                                                \
                if ((s_) >= 2) emit(r0 + (s_) - 2, X);                                                           \
            }
            for (int s = 0; s < nrows + 2; s += 3) {
                GMAT_CONV_STEP(s, A, B, Cc)
                GMAT_CONV_STEP(s + 1, B, Cc, A)
                GMAT_CONV_STEP(s + 2, Cc, A, B)
            }
#undef GMAT_CONV_STEP
        }
    } else {
        for (int i = threadIdx.x; i < th * tw * BPP; i += 256) {
            const int r = i / (tw * BPP), cb = i - r * (tw * BPP);
            const uint8_t *p = st + r * SP + cb;           // top-left tap of the 3x3 window (this channel)
            int sum = 0;
#pragma unroll
            for (int k = 0; k < 9; k++) sum += (int)p[(k / 3) * SP + (k % 3) * BPP] * cp.m[k];
            const float f = __fadd_rn(__fadd_rn(__fmul_rn((float)sum, cp.rdiv), cp.bias), 0.5f);
            if (!TRANSPOSED) dst[(size_t)(y0 + r) * ds + (size_t)x0 * BPP + cb] = (uint8_t)min(max((int)f, 0), 255);
            else { const int px = cb / BPP; rt[px * PT + r * BPP + (cb - px * BPP)] = (uint8_t)min(max((int)f, 0), 255); }
        }
    }
    if (TRANSPOSED) {
        __syncthreads();
        // out(x = y0 + r, y = x0 + c): the result tile is stored one LDS row per tile column
        const bool fast16 = ((((uintptr_t)dst | (uintptr_t)ds) & 15) == 0);
        constexpr int CL = (TH * BPP + 15) / 16 <= 16 ? 16 : 32;
        constexpr int K = (TW + 256 / CL - 1) / (256 / CL);
        lds_to_tile16<CL, K>(dst, ds, y0 * BPP, th * BPP, tw, rt, PT, (int)threadIdx.x, fast16, [&](int c) { return x0 + c; });
    }
}

static inline int al4(const void *a, int sa, const void *b, int sb)
{
    return ((((uintptr_t)a | (uintptr_t)sa | (uintptr_t)b | (uintptr_t)sb) & 3) == 0);
}

// the frame table of a launch: the caller's, or the one frame given by pointer; `loop` = call `one` per frame (kernels without a table)
static inline OpFrames op_frames(const uint8_t *src, uint8_t *dst, const OpFrames *frames)
{
    OpFrames f;
    if (frames) return *frames;
    std::memset(&f, 0, sizeof(f));
    f.src[0] = src; f.dst[0] = dst;
    return f;
}
template <class F> static inline int op_loop(const OpFrames *frames, int nframes, F &&one)
{
    for (int i = 0; i < nframes; i++) {
        const int r = one(frames->src[i], frames->dst[i]);
        if (r < 0) return r;
    }
    return 0;
}

int launch_flip(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int bpp, int fh, int fv,
                hipStream_t stream, const OpFrames *frames, int nframes)
{
    if (w <= 0 || h <= 0) return 0;
    if (nframes < 1 || nframes > kOpMaxFrames) return GMAT_ERR(EINVAL);
    if (frames) { src = frames->src[0]; dst = frames->dst[0]; }
    const OpFrames fr = op_frames(src, dst, frames);
    unsigned all = (unsigned)ss | (unsigned)ds;
    for (int i = 0; i < nframes; i++) all |= (unsigned)(uintptr_t)fr.src[i] | (unsigned)(uintptr_t)fr.dst[i];
    const int aligned = (all & 3) == 0;
    if (bpp >= 3 && aligned && w % 4 == 0 && (bpp == 3 || (all & 15) == 0)) {
        const dim3 dgrid((w + 255) / 256, (h + 3) / 4, nframes), dblock(64, 4);
        if (bpp == 3)      hipLaunchKernelGGL(flip_direct_kernel<3>, dgrid, dblock, 0, stream, src, ss, dst, ds, w, h, fh, fv, fr);
        else if (bpp == 4) hipLaunchKernelGGL(flip_direct_kernel<4>, dgrid, dblock, 0, stream, src, ss, dst, ds, w, h, fh, fv, fr);
        else return GMAT_ERR(ENOSYS);
        GMAT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const dim3 grid((w + 255) / 256, (h + 3) / 4, nframes), block(256);
    if (bpp == 3)      hipLaunchKernelGGL(flip_kernel<3>, grid, block, 0, stream, src, ss, dst, ds, w, h, fh, fv, aligned, fr);
    else if (bpp == 4) hipLaunchKernelGGL(flip_kernel<4>, grid, block, 0, stream, src, ss, dst, ds, w, h, fh, fv, aligned, fr);
    else if (bpp == 1) hipLaunchKernelGGL(flip_kernel<1>, grid, block, 0, stream, src, ss, dst, ds, w, h, fh, fv, aligned, fr);
    else if (bpp == 2) hipLaunchKernelGGL(flip_kernel<2>, grid, block, 0, stream, src, ss, dst, ds, w, h, fh, fv, aligned, fr);
    else return GMAT_ERR(ENOSYS);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// a strided copy (crop_hip; the identity turns of rotate_hip): 16 bytes a lane from a source of ANY alignment — a crop's x offset is
// whatever it is, and gfx950's global loads need none — to a 16-byte aligned destination with streaming stores.  The runtime's 2-D copy
// it replaces: 11.8 us per 4K rgb24 frame (profiles/r03zt_crop.txt).
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned cp_v4u1 __attribute__((ext_vector_type(4), aligned(1)));
__device__ __forceinline__ uint4 cp_ld16(const uint8_t *p) { const cp_v4u1 v = *reinterpret_cast<const cp_v4u1 *>(p); return make_uint4(v.x, v.y, v.z, v.w); }
#else
__host__ __device__ inline uint4 cp_ld16(const uint8_t *p) { uint4 v; std::memcpy(&v, p, 16); return v; }
#endif
__global__ __launch_bounds__(256) void copy2d_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds, int rowBytes, int h)
{
    const int c = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (y >= h || 16 * c >= rowBytes) return;
    const uint8_t *s = src + (size_t)y * ss + 16 * (size_t)c;
    uint8_t *d = dst + (size_t)y * ds + 16 * (size_t)c;
    if (16 * c + 16 <= rowBytes) st_stream(d, cp_ld16(s));
    else for (int i = 0; i < rowBytes - 16 * c; i++) d[i] = s[i];
}

int launch_copy2d(const uint8_t *src, int ss, uint8_t *dst, int ds, int rowBytes, int h, hipStream_t stream)
{
    if (rowBytes <= 0 || h <= 0) return 0;
    if (((((uintptr_t)dst | (uintptr_t)ds) & 15) == 0) && rowBytes >= 64 && src != dst) {
        const dim3 grid(((rowBytes + 15) / 16 + 63) / 64, (h + 3) / 4), block(64, 4);
        hipLaunchKernelGGL(copy2d_kernel, grid, block, 0, stream, src, ss, dst, ds, rowBytes, h);
        GMAT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    GMAT_HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)ds, src, (size_t)ss, (size_t)rowBytes, (size_t)h,
                                    hipMemcpyDeviceToDevice, stream));
    return 0;
}

// ---- the 1 2 1 / 2 4 2 / 1 2 1 smooth without a source tile in LDS ----------------------------------------------------
// smooth_hip's default matrix (rdiv 1/16, bias 0) is separable: out = (v121(h121(x)) + 8) >> 4, the same integer the
// general kernel sums.  A lane owns one DWORD column of a 48-dword tile (lanes 1..48; lanes 0 and 49 carry the halo
// dwords), a wave 16 output rows: it loads its 18 source rows straight from global memory (all loads in flight at once,
// 200 contiguous bytes per instruction), splits every dword into its even and odd bytes as two 16-bit fields (v_perm_b32)
// and does the arithmetic on both fields of a register at once — the sums stay below 4096, ordinary 32-bit adds never
// carry across.  The vertical sum comes first (registers only); the horizontal neighbours of a byte sit BPP bytes away,
// i.e. in the neighbouring lanes' registers: wave-wide DPP shifts (v_mov_b32 wave_shr:1 / wave_shl:1) and a funnel shift
// bring them in.  vf_convolution's borders (index -1 -> 1, index n -> n-1, :555-569) are a clamp of the row index
// (scalar) and, in the tiles on the left / right frame edge, a halo dword assembled from lanes 1, 2 / the last lane.
// TRANSPOSED (rotate 90 + hflip, see launch_rotate_flip_smooth): results go byte-wise into a transposed LDS tile (odd
// dword pitch), one block barrier, then rows of TH * BPP bytes leave as dwordx4 stores.  LDS 12.5 KB, <= 64 VGPRs: the
// 2040 tiles of a 4K frame are all resident at once (8 blocks per CU).
// IDENT: no filter — the plain transpose of 3- / 4-byte pixels through this kernel's loads and transposed tile (dir: vf_transpose's
// four directions, bit 0 reads the source bottom-up, bit 1 writes the destination bottom-up): 13.6 us per 4K rgb24 frame against
// transpose_kernel<3, 64>'s 16.0.
template <int BPP, bool TRANSPOSED, int TD, int RPW, bool IDENT = false, int NWH = 0, int THP = 64>
__global__ __launch_bounds__(NWH ? 64 * NWH : 64 * THP / RPW) void smooth121_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int dst16, int dir, OpFrames fr)
{
    src = fr.src[blockIdx.y]; dst = fr.dst[blockIdx.y];    // grid.y = frame
    constexpr int TH = THP, NS = RPW + 2, NT = 64 * TH / RPW; // tile: TD dwords x TH rows; RPW rows per wave; source rows per wave; threads
    static_assert(TD + 2 <= 64 && (TD * 4) % BPP == 0, "tile width: whole pixels, two halo lanes");
    constexpr int TWP = TD * 4 / BPP;                       // tile width in pixels
    constexpr int PT = TH * BPP + 4;                        // transposed tile pitch: 49 / 65 dwords (odd)
    static_assert(!TRANSPOSED || ((PT / 4) & 1), "odd dword pitch");
    __shared__ __attribute__((aligned(16))) uint8_t rt[TRANSPOSED ? TWP * PT + 256 : 16];      // + the pad of the idle lanes
    const int rowDwords = (w * BPP) >> 2;
    int tbx, tby;
    if constexpr (NWH > 0) {
        // the block's waves side by side: NWH column tiles of RPW rows, so that the rows a block writes are NWH * TD dwords long
        // (8 x 240 B = 15 whole lines) and only the block's own waves share a line
        const int nbx = (rowDwords + TD - 1) / TD, nbxB = (nbx + NWH - 1) / NWH, nbyB = (h + RPW - 1) / RPW, ntiles = nbxB * nbyB;
        const int chunk = (ntiles + 7) >> 3;
        const int t = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
        if (t >= ntiles) return;
        tby = t / nbxB; tbx = (t - tby * nbxB) * NWH + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
        if (tbx >= nbx) return;
    } else
    {   // XCD-aware tile order, as conv3x3_kernel
        const int nbx = (rowDwords + TD - 1) / TD, nby = (h + TH - 1) / TH, ntiles = nbx * nby;
        const int chunk = (ntiles + 7) >> 3;
        const int t = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
        if (t >= ntiles) return;
        if (TRANSPOSED) { tbx = t / nby; tby = t - tbx * nby; }
        else            { tby = t / nbx; tbx = t - tby * nbx; }
    }
    const int d0 = tbx * TD, nd = min(TD, rowDwords - d0);  // first dword and dword count of the tile
    const int y0 = NWH ? tby * RPW : tby * TH, th = min(NWH ? RPW : TH, h - y0);
    const int lane = threadIdx.x & 63;
    const int wave = NWH ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int r0 = wave * RPW, nrows = min(RPW, th - r0);   // wave-uniform
    const int di = lane - 1;                                // this lane's dword of the tile; -1 and nd are the halos
    if (nrows > 0) {
        const unsigned colOff = 4u * (unsigned)min(max(d0 + di, 0), rowDwords - 1);
        unsigned wv[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            int yy = y0 + r0 + s - 1;
            yy = yy < 0 ? -yy : yy;
            yy = yy >= h ? 2 * h - 1 - yy : yy;
            yy = min(max(yy, 0), h - 1);
            if (IDENT && (dir & 1)) yy = h - 1 - yy;
            wv[s] = *reinterpret_cast<const unsigned *>(src + ((unsigned)(yy * ss) + colOff));
        }
        if (d0 == 0) {                                      // pixel -1 := pixel 1
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)wv[s], 1), b = (unsigned)__builtin_amdgcn_readlane((int)wv[s], 2);
                const unsigned halo = BPP == 1 ? a << 16 : BPP == 2 ? a : BPP == 3 ? (a >> 16) | (b << 16) : b;
                wv[s] = lane == 0 ? halo : wv[s];
            }
        }
        if (d0 + nd == rowDwords) {                         // pixel w := pixel w - 1
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const unsigned last = (unsigned)__builtin_amdgcn_readlane((int)wv[s], nd);
                wv[s] = lane == nd + 1 ? last >> (8 * (4 - BPP)) : wv[s];
            }
        }
        // per-lane LDS pointers of the 4 result bytes (transposed store): pixel row * PT + channel; lanes outside the tile
        // write into a pad behind it (one dword each: no bank conflicts, no exec masking around the stores)
        const bool mine = di >= 0 && di < nd;
        uint8_t *pj[4] = {rt, rt, rt, rt};
        if (TRANSPOSED) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int cb = 4 * max(di, 0) + j, px = cb / BPP;
                pj[j] = rt + (mine ? px * PT + (cb - px * BPP) + r0 * BPP : TWP * PT + 4 * lane);
            }
        }
        auto rows = [&](auto full) {
            constexpr bool FULL = decltype(full)::value;    // all 16 rows of the wave exist: no per-row test
            unsigned e0 = __builtin_amdgcn_perm(0u, wv[0], 0x0C020C00u), o0 = __builtin_amdgcn_perm(0u, wv[0], 0x0C030C01u);
            unsigned e1 = __builtin_amdgcn_perm(0u, wv[1], 0x0C020C00u), o1 = __builtin_amdgcn_perm(0u, wv[1], 0x0C030C01u);
#pragma unroll
            for (int r = 0; r < RPW; r++) {
                const unsigned e2 = __builtin_amdgcn_perm(0u, wv[r + 2], 0x0C020C00u), o2 = __builtin_amdgcn_perm(0u, wv[r + 2], 0x0C030C01u);
                if (FULL || r < nrows) {
                    const unsigned ve = e0 + 2 * e1 + e2, vo = o0 + 2 * o1 + o2;
                    // the neighbouring lanes' sums; lanes 0 / 63 have no source and read 0 (bound_ctrl), neither is `mine`
                    const unsigned vep = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ve, 0x138, 0xF, 0xF, true);   // lane - 1
                    const unsigned vop = (unsigned)__builtin_amdgcn_update_dpp(0, (int)vo, 0x138, 0xF, 0xF, true);
                    const unsigned ven = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ve, 0x130, 0xF, 0xF, true);   // lane + 1
                    const unsigned von = (unsigned)__builtin_amdgcn_update_dpp(0, (int)vo, 0x130, 0xF, 0xF, true);
                    unsigned le, lo, re, ro;                // the fields BPP bytes to the left / right of the even / odd bytes
                    if (BPP == 1)      { le = (vop >> 16) | (vo << 16); lo = ve;                       re = vo;                        ro = (ve >> 16) | (ven << 16); }
                    else if (BPP == 2) { le = (vep >> 16) | (ve << 16); lo = (vop >> 16) | (vo << 16); re = (ve >> 16) | (ven << 16);  ro = (vo >> 16) | (von << 16); }
                    else if (BPP == 3) { le = vop;                      lo = (vep >> 16) | (ve << 16); re = (vo >> 16) | (von << 16);  ro = ven; }
                    else               { le = vep;                      lo = vop;                      re = ven;                       ro = von; }
                    const unsigned he = IDENT ? e1 : (le + 2 * ve + re + 0x00080008u) >> 4, ho = IDENT ? o1 : (lo + 2 * vo + ro + 0x00080008u) >> 4;
                    if (TRANSPOSED) {
                        if (BPP == 4) {
                            *reinterpret_cast<unsigned *>(pj[0] + r * 4) = __builtin_amdgcn_perm(ho, he, 0x06020400u);
                        } else {
                            pj[0][r * BPP] = (uint8_t)he; pj[2][r * BPP] = (uint8_t)(he >> 16);
                            pj[1][r * BPP] = (uint8_t)ho; pj[3][r * BPP] = (uint8_t)(ho >> 16);
                        }
                    } else if (mine) {
                        // (plain stores: with `nt` the side-by-side form loses 0 - 10 %, the stacked one 20 - 30 %)
                        *reinterpret_cast<unsigned *>(dst + ((unsigned)((y0 + r0 + r) * ds) + 4u * (unsigned)(d0 + di))) = __builtin_amdgcn_perm(ho, he, 0x06020400u);
                    }
                }
                e0 = e1; o0 = o1; e1 = e2; o1 = o2;
            }
        };
        if (nrows == RPW) rows(std::true_type());
        else              rows(std::false_type());
    }
    if (!TRANSPOSED) return;
    __syncthreads();
    // rows of the transposed tile: pixel column px of the source tile -> destination row x0 + px, bytes [y0*BPP, +th*BPP)
    const int npx = nd * 4 / BPP, x0 = d0 * 4 / BPP, nbytes = th * BPP;
    auto orow = [&](int r) { return (IDENT && (dir & 2)) ? w - 1 - r : r; };       // the destination is w rows tall
    constexpr int CW = TH * BPP / 16 <= 16 ? 16 : TH * BPP / 16 <= 32 ? 32 : 64;   // lanes across a row of the transposed tile
    const int tx = threadIdx.x % CW, ty = threadIdx.x / CW;
    if ((dst16 & 1) && (nbytes & 15) == 0) {                  // dst16 bit 0: 16-byte aligned destination rows; bit 1: streaming stores
        const int cpr = nbytes >> 4;                        // 16-byte chunks per row (<= CW)
        for (int px = ty; px < npx; px += NT / CW) {
            if (tx < cpr) {
                const unsigned *l = reinterpret_cast<const unsigned *>(rt + px * PT + 16 * tx);
                uint8_t *q = dst + (size_t)orow(x0 + px) * ds + (size_t)y0 * BPP + 16 * tx;
                if (dst16 & 2) st_stream(q, make_uint4(l[0], l[1], l[2], l[3]));
                else *reinterpret_cast<uint4 *>(q) = make_uint4(l[0], l[1], l[2], l[3]);
            }
        }
    } else {
        for (int px = ty; px < npx; px += NT / CW)
            lds_to_row(dst + (size_t)orow(x0 + px) * ds, y0 * BPP, nbytes, rt + px * PT, tx, CW, (y0 * BPP & 3) == 0);
    }
}

// a destination whose rows start on 128-byte lines (GMAT_SMOOTH_LINE_DST=0: treat none as such, A/B)
static bool smooth121_line_dst(const uint8_t *dst, int ds)
{
    const char *e = GMAT_KNOB("GMAT_SMOOTH_LINE_DST");
    return !(e && !atoi(e)) && ((((uintptr_t)dst | (uintptr_t)ds) & 127) == 0);
}

// the separable kernel's conditions: rows of whole dwords, dword-aligned pointers and pitches, at least 4 pixels a row
static bool smooth121_ok(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int bpp)
{
    const bool off = GMAT_KNOB("GMAT_NO_SMOOTH121") != nullptr;                        // A/B switch for the benches and tests
    return !off && bpp >= 1 && bpp <= 4 && ((w * bpp) & 3) == 0 && w >= 4 && al4(src, ss, dst, ds) && (int64_t)ss * h < (1ll << 31) &&
           (int64_t)ds * std::max(w, h) < (1ll << 31);
}

int launch_transpose(const uint8_t *src, int ss, uint8_t *dst, int ds, int inW, int inH, int bpp, int dir,
                     hipStream_t stream, const OpFrames *frames, int nframes)
{
    if (inW <= 0 || inH <= 0) return 0;
    if (nframes < 1 || nframes > kOpMaxFrames) return GMAT_ERR(EINVAL);
    if (frames) {                                            // alignment tests below see the batch's least aligned frame
        src = frames->src[0]; dst = frames->dst[0];
        for (int i = 1; i < nframes; i++) {
            src = (const uint8_t *)((uintptr_t)src | ((uintptr_t)frames->src[i] & 15)); dst = (uint8_t *)((uintptr_t)dst | ((uintptr_t)frames->dst[i] & 127));
        }
    }
    const OpFrames fr = op_frames(src, dst, frames);
    if (dir < 0 || dir > 3) return GMAT_ERR(EINVAL);
    // 1- and 2-byte samples (the planes of planar / semi-planar YUV) use 128 x 128 tiles so that a tile row is >= 128 B of contiguous
    // HBM traffic on BOTH sides, with 1024 threads a tile so that the chip holds as many waves as with 64 x 64 tiles (256 threads on
    // 128 x 128: 12.4 us per 4K luma plane; 64 x 64: 7.8 us; 128 x 128 with 1024 threads: 6.5 us).  GMAT_TRANSPOSE_TILE overrides (A/B).
    const char *te = GMAT_KNOB("GMAT_TRANSPOSE_TILE");             // 64 | 128 (256 threads) | 1128 (128 x 128, 1024 threads): A/B
    const int tv = te ? atoi(te) : 0;
    const int tiles128 = ((inW + 127) / 128) * ((inH + 127) / 128);
    const int T = bpp <= 2 ? ((tv ? tv == 64 : tiles128 < 2048) ? 64 : 128) : 64;
    const bool big = bpp <= 2 && (tv == 1128 || tv == 0);   // default (profiles/r03x_transpose_tiles.txt: 4K gray 7.8 -> 6.5 us, 2-byte 11.2 -> 10.0)
    const int TT = big ? 128 : T;
    const int ntiles = ((inW + TT - 1) / TT) * ((inH + TT - 1) / TT);
    const dim3 grid(8 * ((ntiles + 7) / 8), nframes), block(big ? 1024 : 256);
    const int aligned = al4(src, ss, dst, ds);
    if ((bpp == 3 || bpp == 4) && tv == 0 && smooth121_ok(src, ss, dst, ds, inW, inH, bpp)) {
        // packed RGB: the 3 x 3 smooth's transposed form with its filter left out (dword-per-lane row loads, byte-transposed LDS tile,
        // 16-byte row stores)
        const int td = bpp == 3 ? 60 : 62;
        const int nt = ((inW * bpp / 4 + td - 1) / td) * ((inH + 63) / 64);
        const dim3 g2(8 * ((nt + 7) / 8), nframes);
        const int dst16 = ((((uintptr_t)dst | (uintptr_t)ds) & 15) == 0);
        if (smooth121_line_dst(dst, ds)) {                   // whole-line pieces, streaming stores: see launch_rotate_flip_smooth
            if (bpp == 3) {
                const int ntt = ((inW * bpp / 4 + td - 1) / td) * ((inH + 127) / 128);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(smooth121_kernel<3, true, 60, 16, true, 0, 128>), dim3(8 * ((ntt + 7) / 8), nframes), dim3(512), 0, stream,
                                   src, ss, dst, ds, inW, inH, dst16 | 2, dir, fr);
            } else {
                hipLaunchKernelGGL(HIP_KERNEL_NAME(smooth121_kernel<4, true, 62, 16, true>), g2, dim3(256), 0, stream, src, ss, dst, ds, inW, inH, dst16 | 2, dir, fr);
            }
        }
        else if (bpp == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(smooth121_kernel<3, true, 60, 8, true>), g2, dim3(512), 0, stream, src, ss, dst, ds, inW, inH, dst16, dir, fr);
        else               hipLaunchKernelGGL(HIP_KERNEL_NAME(smooth121_kernel<4, true, 62, 16, true>), g2, dim3(256), 0, stream, src, ss, dst, ds, inW, inH, dst16, dir, fr);
        GMAT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (bpp == 3)      hipLaunchKernelGGL(HIP_KERNEL_NAME(transpose_kernel<3, 64>), grid, block, 0, stream, src, ss, dst, ds, inW, inH, dir, aligned, fr);
    else if (bpp == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(transpose_kernel<4, 64>), grid, block, 0, stream, src, ss, dst, ds, inW, inH, dir, aligned, fr);
    else if (bpp == 1 && big) hipLaunchKernelGGL(HIP_KERNEL_NAME(transpose_kernel<1, 128, 1024>), grid, block, 0, stream, src, ss, dst, ds, inW, inH, dir, aligned, fr);
    else if (bpp == 2 && big) hipLaunchKernelGGL(HIP_KERNEL_NAME(transpose_kernel<2, 128, 1024>), grid, block, 0, stream, src, ss, dst, ds, inW, inH, dir, aligned, fr);
    else if (bpp == 1 && T == 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(transpose_kernel<1, 128>), grid, block, 0, stream, src, ss, dst, ds, inW, inH, dir, aligned, fr);
    else if (bpp == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(transpose_kernel<1, 64>), grid, block, 0, stream, src, ss, dst, ds, inW, inH, dir, aligned, fr);
    else if (bpp == 2 && T == 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(transpose_kernel<2, 128>), grid, block, 0, stream, src, ss, dst, ds, inW, inH, dir, aligned, fr);
    else if (bpp == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(transpose_kernel<2, 64>), grid, block, 0, stream, src, ss, dst, ds, inW, inH, dir, aligned, fr);
    else return GMAT_ERR(ENOSYS);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_conv3x3(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int bpp, const int m[9],
                   float rdiv, float bias, hipStream_t stream, const OpFrames *frames, int nframes)
{
    if (w <= 0 || h <= 0) return 0;
    if (nframes < 1 || nframes > kOpMaxFrames) return GMAT_ERR(EINVAL);
    if (frames) {                                            // alignment tests below see the batch's least aligned frame
        src = frames->src[0]; dst = frames->dst[0];
        for (int i = 1; i < nframes; i++) {
            src = (const uint8_t *)((uintptr_t)src | ((uintptr_t)frames->src[i] & 15)); dst = (uint8_t *)((uintptr_t)dst | ((uintptr_t)frames->dst[i] & 127));
        }
    }
    const OpFrames fr = op_frames(src, dst, frames);
    ConvParams cp;
    int fast = 1;
    for (int i = 0; i < 9; i++) { cp.m[i] = m[i]; fast &= m[i] >= 0 && m[i] <= 255; }
    cp.rdiv = rdiv; cp.bias = bias;
    cp.shift = -1; cp.half = 0;
    for (int k = 0; k <= 16 && fast; k++)
        if (bias == 0.0f && rdiv == 1.0f / (float)(1 << k)) { cp.shift = k; cp.half = k ? 1u << (k - 1) : 0u; }
    static const int m121[9] = {1, 2, 1, 2, 4, 2, 1, 2, 1};
    if (cp.shift == 4 && std::memcmp(m, m121, sizeof(m121)) == 0 && smooth121_ok(src, ss, dst, ds, w, h, bpp)) {
        const int td = bpp == 3 ? 60 : 62;                   // tile width in dwords: whole pixels (60 dwords = 80 rgb24 pixels)
        // four waves of a block side by side (4 x 240 B a row, 16 rows): only a block's own waves share a 128-byte line.  With the waves
        // stacked (a 240 B x 64 row tile a block) every line was completed by ANOTHER block, on another XCD as often as not, and each L2
        // wrote its part: 16 frames a launch 10.3 -> 9.0 us rgb24, 3.0 -> 2.45 us gray (profiles/r03zu_smooth_layout.txt).
        // GMAT_SMOOTH_STACKED=1: the stacked form (A/B)
        const char *est = GMAT_KNOB("GMAT_SMOOTH_STACKED");
        if (!(est && atoi(est))) {
            constexpr int NWH = 4;
            const int nbx = (w * bpp / 4 + td - 1) / td;
            const dim3 bh(64 * NWH);
            // 1- and 2-byte planes: 8 rows a wave — a 4K luma plane is 2 160 waves at 16 rows, a quarter of what the chip holds (one frame
            // per launch 6.15 -> 5.73 us, 16 frames per launch unchanged)
            const int rpw = bpp <= 2 ? 8 : 16;
            const int ntb2 = ((nbx + NWH - 1) / NWH) * ((h + rpw - 1) / rpw);
            const dim3 g2(8 * ((ntb2 + 7) / 8), nframes);
            switch (bpp) {
            case 1:  hipLaunchKernelGGL(HIP_KERNEL_NAME(smooth121_kernel<1, false, 62, 8, false, NWH>), g2, bh, 0, stream, src, ss, dst, ds, w, h, 0, 0, fr); break;
            case 2:  hipLaunchKernelGGL(HIP_KERNEL_NAME(smooth121_kernel<2, false, 62, 8, false, NWH>), g2, bh, 0, stream, src, ss, dst, ds, w, h, 0, 0, fr); break;
            case 3:  hipLaunchKernelGGL(HIP_KERNEL_NAME(smooth121_kernel<3, false, 60, 16, false, NWH>), g2, bh, 0, stream, src, ss, dst, ds, w, h, 0, 0, fr); break;
            default: hipLaunchKernelGGL(HIP_KERNEL_NAME(smooth121_kernel<4, false, 62, 16, false, NWH>), g2, bh, 0, stream, src, ss, dst, ds, w, h, 0, 0, fr); break;
            }
            GMAT_HIP_CHECK(hipGetLastError());
            return 0;
        }
        const int nt = ((w * bpp / 4 + td - 1) / td) * ((h + 63) / 64);
        const dim3 g(8 * ((nt + 7) / 8), nframes), b(256);
        switch (bpp) {
        case 1:  hipLaunchKernelGGL(HIP_KERNEL_NAME(smooth121_kernel<1, false, 62, 16>), g, b, 0, stream, src, ss, dst, ds, w, h, 0, 0, fr); break;
        case 2:  hipLaunchKernelGGL(HIP_KERNEL_NAME(smooth121_kernel<2, false, 62, 16>), g, b, 0, stream, src, ss, dst, ds, w, h, 0, 0, fr); break;
        case 3:  hipLaunchKernelGGL(HIP_KERNEL_NAME(smooth121_kernel<3, false, 60, 16>), g, b, 0, stream, src, ss, dst, ds, w, h, 0, 0, fr); break;
        default: hipLaunchKernelGGL(HIP_KERNEL_NAME(smooth121_kernel<4, false, 62, 16>), g, b, 0, stream, src, ss, dst, ds, w, h, 0, 0, fr); break;
        }
        GMAT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (frames && nframes > 1)                              // the general matrix has no frame table: one launch per frame
        return op_loop(frames, nframes, [&](const uint8_t *s1, uint8_t *d1) { return launch_conv3x3(s1, ss, d1, ds, w, h, bpp, m, rdiv, bias, stream, nullptr, 1); });
    if (frames) { src = frames->src[0]; dst = frames->dst[0]; }
    const int TW = bpp <= 2 ? 128 : 64;
    const int ntiles = ((w + TW - 1) / TW) * ((h + 63) / 64);
    const dim3 grid(8 * ((ntiles + 7) / 8)), block(256);
    const int aligned = al4(src, ss, dst, ds);
    if (bpp == 3)      hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_kernel<3, 64, 64, false>), grid, block, 0, stream, src, ss, dst, ds, w, h, cp, aligned, fast);
    else if (bpp == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_kernel<4, 64, 64, false>), grid, block, 0, stream, src, ss, dst, ds, w, h, cp, aligned, fast);
    else if (bpp == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_kernel<1, 128, 64, false>), grid, block, 0, stream, src, ss, dst, ds, w, h, cp, aligned, fast);
    else if (bpp == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_kernel<2, 128, 64, false>), grid, block, 0, stream, src, ss, dst, ds, w, h, cp, aligned, fast);
    else return GMAT_ERR(ENOSYS);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- 3x3 median (smooth_nvcv type=median, vf_smooth_nvcv.c:82-105; semantics of the CPU median filter,
// vf_median.c + median_template.c at radius 1, percentile 0.5: per channel the 5th smallest of the 3x3 window, rows and
// columns clamped at the frame edges — the histogram walk there adds row max(0, y-1) / min(h-1, y+1) and column
// max(0, x-1) / min(w-1, x+1)).  A selection, no arithmetic.  Per output byte: sort the three columns with
// v_min3 / v_med3 / v_max3, then med3(max3(mins), med3(meds), min3(maxs)).  4 consecutive bytes of a row per thread;
// taps are unaligned dword loads at -bpp / 0 / +bpp bytes, byte-wise with clamped columns in the first and last group.
template <int BPP>
__global__ __launch_bounds__(256) void median3x3_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int dstAligned)
{
    // 4 consecutive bytes of TWO vertically adjacent output rows per thread: the rows y-1 .. y+2 are loaded once, and
    // the two windows share the sorted pair (row y, row y+1) of every column
    const int rb = w * BPP;
    const int i0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 2;
    if (i0 >= rb || y >= h) return;
    const uint8_t *rows[4] = {src + (size_t)max(y - 1, 0) * ss, src + (size_t)y * ss, src + (size_t)min(y + 1, h - 1) * ss,
                              src + (size_t)min(y + 2, h - 1) * ss};
    unsigned t[4][3];                                     // [row][column offset -1, 0, +1]: 4 bytes each
    const bool interior = i0 >= BPP && i0 + 4 + BPP <= rb;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (interior) {
            __builtin_memcpy(&t[r][0], rows[r] + i0 - BPP, 4);
            __builtin_memcpy(&t[r][1], rows[r] + i0, 4);
            __builtin_memcpy(&t[r][2], rows[r] + i0 + BPP, 4);
        } else {
            t[r][0] = t[r][1] = t[r][2] = 0;
            for (int b = 0; b < 4; b++) {
                const int i = min(i0 + b, rb - 1), px = i / BPP, ch = i - px * BPP;
                t[r][0] |= (unsigned)rows[r][max(px - 1, 0) * BPP + ch] << (8 * b);
                t[r][1] |= (unsigned)rows[r][i] << (8 * b);
                t[r][2] |= (unsigned)rows[r][min(px + 1, w - 1) * BPP + ch] << (8 * b);
            }
        }
    }
    unsigned o0 = 0, o1 = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        unsigned lo0[3], md0[3], hi0[3], lo1[3], md1[3], hi1[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const unsigned a0 = (t[0][c] >> (8 * b)) & 0xFF, a1 = (t[1][c] >> (8 * b)) & 0xFF, a2 = (t[2][c] >> (8 * b)) & 0xFF,
                           a3 = (t[3][c] >> (8 * b)) & 0xFF;
            const unsigned pl = min(a1, a2), ph = max(a1, a2);               // the pair both windows contain
            lo0[c] = min(a0, pl); hi0[c] = max(a0, ph); md0[c] = max(pl, min(a0, ph));
            lo1[c] = min(a3, pl); hi1[c] = max(a3, ph); md1[c] = max(pl, min(a3, ph));
        }
        {
            const unsigned A = max(max(lo0[0], lo0[1]), lo0[2]), C = min(min(hi0[0], hi0[1]), hi0[2]);
            const unsigned B = max(min(md0[0], md0[1]), min(max(md0[0], md0[1]), md0[2]));
            o0 |= max(min(A, B), min(max(A, B), C)) << (8 * b);
        }
        {
            const unsigned A = max(max(lo1[0], lo1[1]), lo1[2]), C = min(min(hi1[0], hi1[1]), hi1[2]);
            const unsigned B = max(min(md1[0], md1[1]), min(max(md1[0], md1[1]), md1[2]));
            o1 |= max(min(A, B), min(max(A, B), C)) << (8 * b);
        }
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (y + k >= h) break;
        const unsigned o = k ? o1 : o0;
        uint8_t *d = dst + (size_t)(y + k) * ds + i0;
        if (dstAligned && i0 + 4 <= rb) *reinterpret_cast<unsigned *>(d) = o;
        else for (int b = 0; b < min(4, rb - i0); b++) d[b] = (uint8_t)(o >> (8 * b));
    }
}

// ---- the same median without per-byte extraction (round 2) --------------------------------------------------------------
// A strip walker: a lane owns one DWORD column of a 62-dword strip (lanes 1 .. 62; lanes 0 and 63 carry the halo dwords), a wave
// walks down a segment of rows two output rows per iteration, its loads running two iterations ahead (the first build loaded a
// whole 64 x 16 tile up front, as smooth121_kernel does: with every wave of the launch resident at once the chip loaded, then
// computed, then stored; walking did not change the time, see below).  Per output row the lane splits the new row's dword into its 4
// bytes once, sorts ITS column's three rows per byte (v_min3 / v_med3 / v_max3) and takes the sorted triples of the columns BPP bytes
// to its left and right from its own registers or the neighbouring lanes' (DPP wave shifts), instead of extracting and sorting three
// columns per byte: 53 full-rate instructions per 4 output bytes at 3 bytes per pixel (the kernel above: ~ 190).  The first build
// did the sorting on two 16-bit fields per register with v_pk_min_u16 / v_pk_max_u16 (56 instructions, 34 of them packed): 19.0 us per
// 4K rgb24 frame in either form, 7.3 cycles per instruction on average — the packed 16-bit integer min / max issue at half rate.  Borders: a clamp of the row index, and in the tiles on the frame's left / right edge a halo dword
// holding pixel 0 / pixel w - 1.
struct MdB { unsigned b[4]; };                               // the 4 bytes of a lane's dword, one per register
// one instruction each, spelled out: from max(min(a, b), min(max(a, b), c)) the compiler formed v_med3_u32 in a third of the places and
// four-instruction min / max chains in the rest (81 VALU instructions per row instead of 55)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ unsigned md_min3(unsigned a, unsigned b, unsigned c) { unsigned r; asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ unsigned md_max3(unsigned a, unsigned b, unsigned c) { unsigned r; asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ unsigned md_med3(unsigned a, unsigned b, unsigned c) { unsigned r; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
#else
__device__ __forceinline__ unsigned md_min3(unsigned a, unsigned b, unsigned c) { return min(min(a, b), c); }
__device__ __forceinline__ unsigned md_max3(unsigned a, unsigned b, unsigned c) { return max(max(a, b), c); }
__device__ __forceinline__ unsigned md_med3(unsigned a, unsigned b, unsigned c) { return max(min(a, b), min(max(a, b), c)); }
#endif
// the bytes BPP to the left / right of this lane's four: its own registers renamed, or the neighbouring lane's over a DPP wave shift
template <int BPP>
__device__ __forceinline__ void md_sides(const MdB &v, MdB &l, MdB &r)
{
#pragma unroll
    for (int k = 0; k < 4; k++) {
        l.b[k] = k - BPP >= 0 ? v.b[k - BPP >= 0 ? k - BPP : 0]
                              : (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.b[(k - BPP + 4) & 3], 0x138, 0xF, 0xF, true);   // lane - 1
        r.b[k] = k + BPP < 4 ? v.b[k + BPP < 4 ? k + BPP : 0]
                             : (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.b[(k + BPP - 4) & 3], 0x130, 0xF, 0xF, true);    // lane + 1
    }
}

constexpr int MD_TD = 62;                                    // dwords of a strip: lanes 1 .. 62 produce, lanes 0 and 63 carry the halo dwords
constexpr int MD_Q = 2;                                      // iterations (row pairs) the loads run ahead of their use
template <int BPP>
__global__ __launch_bounds__(256) void median3x3s_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h,
                                                         int segRows, int nseg, int nstrips, int nblk, OpFrames fr)
{
    src = fr.src[blockIdx.y]; dst = fr.dst[blockIdx.y];    // grid.y = frame
    const int rowDwords = (w * BPP) >> 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // XCD-aware order of the workgroups; (segment, strip) units packed densely, four to a workgroup: the waves share nothing
    const int chunk = (nblk + 7) >> 3;
    const int lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (lin >= nblk) return;
    const int unit = lin * 4 + wave;
    if (unit >= nseg * nstrips) return;
    const int seg = __builtin_amdgcn_readfirstlane(unit / nstrips);
    const int d0 = (unit - seg * nstrips) * MD_TD;
    const int nd = __builtin_amdgcn_readfirstlane(min(MD_TD, rowDwords - d0));
    const int yb = seg * segRows, nOut = min(segRows, h - yb);
    const int nIter = (nOut + 1) >> 1;                       // two output rows per iteration
    const int di = lane - 1;                                 // this lane's dword of the strip; -1 and nd are the halos
    const unsigned colOff = 4u * (unsigned)min(max(d0 + di, 0), rowDwords - 1);
    const bool isL = d0 == 0, isR = d0 + nd == rowDwords;    // wave-uniform: the strip touches the frame's left / right edge
    const bool mine = di >= 0 && di < nd;
    // source row s of the segment = frame row yb - 1 + s, clamped (rows past the segment's last are loaded and never used)
    auto load = [&](int s) { return *reinterpret_cast<const unsigned *>(src + ((unsigned)(min(max(yb - 1 + s, 0), h - 1) * ss) + colOff)); };
    auto split = [&](unsigned x) {
        if (isL) {                                           // pixel -1 := pixel 0: the halo dword ends with it
            const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)x, 1);
            x = lane == 0 ? a << (8 * (4 - BPP)) : x;
        }
        if (isR) {                                           // pixel w := pixel w - 1: the halo dword starts with it
            const unsigned last = (unsigned)__builtin_amdgcn_readlane((int)x, nd);
            x = lane == nd + 1 ? last >> (8 * (4 - BPP)) : x;
        }
        return MdB{{x & 0xFFu, (x >> 8) & 0xFFu, (x >> 16) & 0xFFu, x >> 24}};
    };
    // one output row: the sorted triple (lo, md, hi) of this lane's column -> with the neighbours' triples the median of nine
    auto emit = [&](int r, bool st, const MdB &x0, const MdB &x1, const MdB &x2) {
        MdB lo, md, hi, ll, lr, ml, mr, hl, hr;
#pragma unroll
        for (int k = 0; k < 4; k++) { lo.b[k] = md_min3(x0.b[k], x1.b[k], x2.b[k]); md.b[k] = md_med3(x0.b[k], x1.b[k], x2.b[k]); hi.b[k] = md_max3(x0.b[k], x1.b[k], x2.b[k]); }
        md_sides<BPP>(lo, ll, lr);
        md_sides<BPP>(md, ml, mr);
        md_sides<BPP>(hi, hl, hr);
        unsigned m[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            m[k] = md_med3(md_max3(ll.b[k], lo.b[k], lr.b[k]), md_med3(ml.b[k], md.b[k], mr.b[k]), md_min3(hl.b[k], hi.b[k], hr.b[k]));
        if (st) st_stream(dst + ((unsigned)((yb + r) * ds) + 4u * (unsigned)(d0 + di)), (unsigned)(m[0] | (m[1] << 8) | (m[2] << 16) | (m[3] << 24)));
    };
    MdB a0 = split(load(0)), a1 = split(load(1));
    unsigned q[MD_Q][2];                                     // raw dwords of the rows 2i + 2, 2i + 3 of the next MD_Q iterations
#pragma unroll
    for (int k = 0; k < MD_Q; k++) { q[k][0] = load(2 * k + 2); q[k][1] = load(2 * k + 3); }
    auto body = [&](const int i, auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value;         // i % MD_Q, static after unrolling
        // output row 2i: source rows 2i .. 2i + 2; output row 2i + 1: rows 2i + 1 .. 2i + 3
        const unsigned r2 = q[SLOT][0], r3 = q[SLOT][1];
        q[SLOT][0] = load(2 * (i + MD_Q) + 2); q[SLOT][1] = load(2 * (i + MD_Q) + 3);
        const MdB a2 = split(r2), a3 = split(r3);
        emit(2 * i, mine, a0, a1, a2);
        emit(2 * i + 1, mine && 2 * i + 1 < nOut, a1, a2, a3);
        a0 = a2; a1 = a3;
    };
    static_assert(MD_Q == 2, "the loop below is unrolled by MD_Q");
    for (int i0 = 0; i0 < nIter; i0 += MD_Q) {
        body(i0, std::integral_constant<int, 0>());
        if (i0 + 1 < nIter) body(i0 + 1, std::integral_constant<int, 1>());
    }
}

// the strip form's conditions: rows of whole dwords, dword-aligned pointers and pitches, at least 4 pixels a row
static bool median3x3s_ok(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int bpp)
{
    const bool off = GMAT_KNOB("GMAT_NO_MEDIAN_STRIP") != nullptr;                     // A/B switch for the benches and tests
    return !off && bpp >= 1 && bpp <= 4 && ((w * bpp) & 3) == 0 && w >= 4 && al4(src, ss, dst, ds) && (int64_t)ss * h < (1ll << 31) &&
           (int64_t)ds * h < (1ll << 31);
}

// ---- median of a (2 r + 1) x (2 rv + 1) window per channel (smooth_nvcv type=median with kw, kh other than 3; vf_median.c at
// radius r, radiusV rv, percentile 0.5): a thread makes one output byte by bisecting on the VALUE — eight passes over its window,
// pass b asks "how many samples are below the candidate with bit b set" — so nothing is sorted and nothing is indexed dynamically.
// Rows and columns clamped (median_template.c:97-147).  A completeness path (the reference's own default is 3 x 3): ~8 n compares
// per byte.
__global__ __launch_bounds__(256) void median_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int bpp, int r, int rv)
{
    const int rb = w * bpp;
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= rb || y >= h) return;
    const int px = i / bpp, ch = i - px * bpp;
    const int t = 2 * r * rv + r + rv;                  // the output is the (t + 1)-th smallest
    int lo = 0;                                         // largest value v with count(samples < v) <= t, built bit by bit
    for (int bit = 7; bit >= 0; bit--) {
        const int cand = lo | (1 << bit);
        int below = 0;
        for (int j = -rv; j <= rv; j++) {
            const uint8_t *row = src + (size_t)min(max(y + j, 0), h - 1) * ss + ch;
            for (int k = -r; k <= r; k++) below += row[min(max(px + k, 0), w - 1) * bpp] < cand;
        }
        if (below <= t) lo = cand;
    }
    dst[(size_t)y * ds + i] = (uint8_t)lo;
}

int launch_median(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int bpp, int kw, int kh, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return 0;
    if (!src || !dst || kw < 1 || kh < 1 || !(kw & 1) || !(kh & 1) || bpp < 1 || bpp > 4) return GMAT_ERR(EINVAL);
    int r = (kw - 1) / 2, rv = (kh - 1) / 2;
    if (w < 2 * r + 1) r = (w - 1) / 2;                 // check_params, vf_median.c:111-123
    if (h < 2 * rv + 1) rv = (h - 1) / 2;
    if (r == 1 && rv == 1) return launch_median3x3(src, ss, dst, ds, w, h, bpp, stream);
    const dim3 grid((w * bpp + 63) / 64, (h + 3) / 4), block(256);
    hipLaunchKernelGGL(median_kernel, grid, block, 0, stream, src, ss, dst, ds, w, h, bpp, r, rv);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_median3x3(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int bpp, hipStream_t stream,
                     const OpFrames *frames, int nframes)
{
    if (w <= 0 || h <= 0) return 0;
    if (!frames && (!src || !dst)) return GMAT_ERR(EINVAL);
    if (nframes < 1 || nframes > kOpMaxFrames) return GMAT_ERR(EINVAL);
    if (frames) {                                            // alignment tests below see the batch's least aligned frame
        src = frames->src[0]; dst = frames->dst[0];
        for (int i = 1; i < nframes; i++) {
            src = (const uint8_t *)((uintptr_t)src | ((uintptr_t)frames->src[i] & 15)); dst = (uint8_t *)((uintptr_t)dst | ((uintptr_t)frames->dst[i] & 127));
        }
    }
    const OpFrames fr = op_frames(src, dst, frames);
    const int rb = w * bpp;
    if (median3x3s_ok(src, ss, dst, ds, w, h, bpp)) {
        const int nstrips = ((rb >> 2) + MD_TD - 1) / MD_TD;
        const char *segStr = GMAT_KNOB("GMAT_STRIP_ROWS");          // tuning / test override: output rows per segment
        int seg = segStr ? atoi(segStr) : 0;
        // whole row pairs; measured on 4K rgb24 / gray (profiles/r02y_median.txt): 6 - 8 rows 16.4 - 16.7 / 6.9 - 7.0 us, 16 rows 17.5 / 8.2, 32 rows 19.4 / 10.6
        if (seg <= 0) seg = (int)std::min(64L, std::max(8L, ((long)h * nstrips + 12287) / 12288)) & ~1;
        seg = std::max(seg, 1);
        const int nseg = (h + seg - 1) / seg, nblk = (nseg * nstrips + 3) / 4;
        const dim3 g(8 * ((nblk + 7) / 8), nframes), b(256);
        switch (bpp) {
        case 1:  hipLaunchKernelGGL(HIP_KERNEL_NAME(median3x3s_kernel<1>), g, b, 0, stream, src, ss, dst, ds, w, h, seg, nseg, nstrips, nblk, fr); break;
        case 2:  hipLaunchKernelGGL(HIP_KERNEL_NAME(median3x3s_kernel<2>), g, b, 0, stream, src, ss, dst, ds, w, h, seg, nseg, nstrips, nblk, fr); break;
        case 3:  hipLaunchKernelGGL(HIP_KERNEL_NAME(median3x3s_kernel<3>), g, b, 0, stream, src, ss, dst, ds, w, h, seg, nseg, nstrips, nblk, fr); break;
        default: hipLaunchKernelGGL(HIP_KERNEL_NAME(median3x3s_kernel<4>), g, b, 0, stream, src, ss, dst, ds, w, h, seg, nseg, nstrips, nblk, fr); break;
        }
        GMAT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (frames && nframes > 1)                              // the byte-wise kernel has no frame table
        return op_loop(frames, nframes, [&](const uint8_t *s1, uint8_t *d1) { return launch_median3x3(s1, ss, d1, ds, w, h, bpp, stream, nullptr, 1); });
    if (frames) { src = frames->src[0]; dst = frames->dst[0]; }
    const dim3 grid((rb + 255) / 256, (h + 7) / 8), block(256);          // 4 thread rows x 2 output rows per block
    const int aligned = ((((uintptr_t)dst | (uintptr_t)ds) & 3) == 0);
    switch (bpp) {
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(median3x3_kernel<1>), grid, block, 0, stream, src, ss, dst, ds, w, h, aligned); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(median3x3_kernel<2>), grid, block, 0, stream, src, ss, dst, ds, w, h, aligned); break;
    case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(median3x3_kernel<3>), grid, block, 0, stream, src, ss, dst, ds, w, h, aligned); break;
    case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(median3x3_kernel<4>), grid, block, 0, stream, src, ss, dst, ds, w, h, aligned); break;
    default: return GMAT_ERR(ENOSYS);
    }
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- smooth_nvcv type=gaussian beyond the 3x3 integer kernel: kw x kh, sigmaX / sigmaY, five border rules ----------
// The reference hands these options to CV-CUDA's Gaussian operator (vf_smooth_nvcv.c:88-105,:290-294), whose arithmetic
// no reference test pins (SURVEY.md §8c "parity unpinned").  The rule stated here (the tests restate it
// independently on the CPU) is OpenCV's, which CV-CUDA documents itself as following:
//   * 1-D kernels as cv::getGaussianKernel: sigma <= 0 -> 0.3 * ((k - 1) * 0.5 - 1) + 0.8, with the fixed tables for
//     k = 1, 3, 5, 7; otherwise exp(-x^2 / (2 sigma^2)) normalised to sum 1 (computed in double on the host, stored as
//     float); sigmaY <= 0 -> sigmaX;
//   * out = clip_u8((int)(sum + 0.5f)), sum accumulated in float32 over the window in raster order (rows outer), each
//     term (ky[j] * kx[i]) * pixel with both products rounded to float (no fma: the library is built with
//     -ffp-contract=off);
//   * borders as cv::borderInterpolate: 0 constant (value 0), 1 replicate, 2 reflect (edge sample repeated),
//     3 wrap, 4 reflect101.
struct GaussParams { int w, h, kw, kh, border; float kx[kGaussMaxTaps], ky[kGaussMaxTaps]; };

__device__ __forceinline__ int border_index(int p, int len, int border)
{
    if ((unsigned)p < (unsigned)len) return p;
    switch (border) {
    case 1: return p < 0 ? 0 : len - 1;
    case 2: case 4: {
        if (len == 1) return 0;
        const int delta = border == 4;
        do {
            if (p < 0) p = -p - 1 + delta;
            else p = len - 1 - (p - len) - delta;
        } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    case 3: {
        if (p < 0) p -= ((p - len + 1) / len) * len;
        if (p >= len) p %= len;
        return p;
    }
    default: return -1;            // constant
    }
}

template <int BPP>
__global__ __launch_bounds__(256) void gauss_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds, GaussParams p)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.w || y >= p.h) return;
    float sum[BPP];
#pragma unroll
    for (int c = 0; c < BPP; c++) sum[c] = 0.0f;
    for (int j = 0; j < p.kh; j++) {
        const int yy = border_index(y + j - p.kh / 2, p.h, p.border);
        for (int i = 0; i < p.kw; i++) {
            const int xx = border_index(x + i - p.kw / 2, p.w, p.border);
            const float wgt = __fmul_rn(p.ky[j], p.kx[i]);
#pragma unroll
            for (int c = 0; c < BPP; c++) {
                const float v = (yy < 0 || xx < 0) ? 0.0f : (float)src[(size_t)yy * ss + (size_t)xx * BPP + c];
                sum[c] = __fadd_rn(sum[c], __fmul_rn(wgt, v));
            }
        }
    }
#pragma unroll
    for (int c = 0; c < BPP; c++) {
        const int v = (int)__fadd_rn(sum[c], 0.5f);
        dst[(size_t)y * ds + (size_t)x * BPP + c] = (uint8_t)min(max(v, 0), 255);
    }
}

int gauss_kernel_1d(int k, double sigma, float *out)
{
    if (k < 1 || k > kGaussMaxTaps || !(k & 1)) return GMAT_ERR(EINVAL);
    static const double small[4][7] = {{1.0}, {0.25, 0.5, 0.25}, {0.0625, 0.25, 0.375, 0.25, 0.0625},
                                       {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125}};
    if (sigma <= 0 && k <= 7) {
        for (int i = 0; i < k; i++) out[i] = (float)small[k >> 1][i];
        return 0;
    }
    const double sg = sigma > 0 ? sigma : 0.3 * ((k - 1) * 0.5 - 1) + 0.8;
    const double scale2 = -0.5 / (sg * sg);
    double cf[kGaussMaxTaps], total = 0;
    for (int i = 0; i < k; i++) {
        const double x = i - (k - 1) * 0.5;
        cf[i] = std::exp(scale2 * x * x);
        total += cf[i];
    }
    for (int i = 0; i < k; i++) out[i] = (float)(cf[i] / total);
    return 0;
}

int launch_gauss_blur(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int bpp, int kw, int kh, double sigmaX,
                      double sigmaY, int border, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return 0;
    if (!src || !dst || border < 0 || border > 4) return GMAT_ERR(EINVAL);
    GaussParams p;
    std::memset(&p, 0, sizeof(p));
    p.w = w; p.h = h; p.kw = kw; p.kh = kh; p.border = border;
    int r = gauss_kernel_1d(kw, sigmaX, p.kx);
    if (r < 0) return r;
    if ((r = gauss_kernel_1d(kh, sigmaY > 0 ? sigmaY : sigmaX, p.ky)) < 0) return r;
    const dim3 grid((w + 63) / 64, (h + 3) / 4), block(256);
    switch (bpp) {
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(gauss_kernel<1>), grid, block, 0, stream, src, ss, dst, ds, p); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(gauss_kernel<2>), grid, block, 0, stream, src, ss, dst, ds, p); break;
    case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(gauss_kernel<3>), grid, block, 0, stream, src, ss, dst, ds, p); break;
    case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(gauss_kernel<4>), grid, block, 0, stream, src, ss, dst, ds, p); break;
    default: return GMAT_ERR(ENOSYS);
    }
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- arbitrary-angle rotation, vf_rotate.c's 16.16 fixed point --------------------------------------
// Source position of output pixel (i, j): x = X0 + j*s + i*c, y = Y0 + j*c - i*s (filter_slice, vf_rotate.c:
// 427-429,:487-492); pixels whose integer position leaves [-1, in] keep the fill colour (:463); bilinear taps as
// interpolate_bilinear8 (:224-249) with a 64-bit final blend, or the clamped nearest sample.
struct RotateParams { int X0, Y0, s, c, inW, inH, outW, outH, bilinear, fillEnable; unsigned fill; };   // bilinear: 0 nearest, 1 linear, 2 cubic

// Catmull-Rom weights of 14 fractional bits for an 8-bit fraction, in integers (the rule is stated in the test suite's checker (orc_vf.c):
// rotate_nvcv's interp=cubic has no integer reference)
__device__ __forceinline__ void rot_cubic_w(int f, int (&w)[4])
{
    const long long f2 = (long long)f * f, f3 = f2 * f;
    const long long n0 = -f3 + 512 * f2 - 65536LL * f, n1 = 3 * f3 - 1280 * f2 + (1LL << 25), n3 = f3 - 256 * f2;
    w[0] = (int)((n0 + 1024) >> 11); w[1] = (int)((n1 + 1024) >> 11); w[3] = (int)((n3 + 1024) >> 11);
    w[2] = 16384 - w[0] - w[1] - w[3];
}

// The direct form (gathers from global memory; the fallback for sources that are not dword-aligned): a wave covers a compact patch of
// the output — LX lanes across (4 pixels each) by 64 / LX rows — so that the source pixels its lanes gather lie in a small rotated
// rectangle instead of along a slanted line 256 pixels long that crosses 75 source rows (round 2: 49 us per 4K rgb24 frame at 17
// degrees; so: 42).  Blocks are numbered so that each XCD walks a contiguous eighth of the tiles.
template <int BPP, int LX, int INTERP>
__global__ __launch_bounds__(256) void rotate_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds, RotateParams p,
                                                     int aligned, int nbx, int nby)
{
    constexpr int WR = 64 / LX;                             // rows a wave covers
    int t = blockIdx.x;
    {
        const int nt = nbx * nby, chunk = (nt + 7) >> 3;
        t = (t & 7) * chunk + (t >> 3);
        if (t >= nt) return;
    }
    const int bx = t % nbx, by = t / nbx, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = (bx * LX + (lane % LX)) * 4, j = (by * 4 + wave) * WR + lane / LX;
    if (i0 >= p.outW || j >= p.outH) return;
    uint8_t o[4 * BPP];
    bool valid[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int i = i0 + q;
        const int x = p.X0 + j * p.s + i * p.c, y = p.Y0 + j * p.c - i * p.s;
        const int x1 = x >> 16, y1 = y >> 16;
        valid[q] = x1 >= -1 && x1 <= p.inW && y1 >= -1 && y1 <= p.inH;
        const int ix = min(max(x1, 0), p.inW - 1), iy = min(max(y1, 0), p.inH - 1);
        if (!valid[q]) {
#pragma unroll
            for (int k = 0; k < BPP; k++) o[q * BPP + k] = (uint8_t)(p.fill >> (8 * k));
        } else if (INTERP == 2) {
            // 4 x 4 Catmull-Rom, clamped indices, rows first, 64-bit vertical sum
            int wx[4], wy[4];
            rot_cubic_w((x & 0xFFFF) >> 8, wx); rot_cubic_w((y & 0xFFFF) >> 8, wy);
            long long v[BPP];
#pragma unroll
            for (int k = 0; k < BPP; k++) v[k] = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint8_t *row = src + (size_t)min(max(y1 - 1 + r, 0), p.inH - 1) * ss;
                int hsum[BPP];
#pragma unroll
                for (int k = 0; k < BPP; k++) hsum[k] = 0;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const uint8_t *px = row + BPP * min(max(x1 - 1 + t, 0), p.inW - 1);
#pragma unroll
                    for (int k = 0; k < BPP; k++) hsum[k] += wx[t] * (int)px[k];
                }
#pragma unroll
                for (int k = 0; k < BPP; k++) v[k] += (long long)wy[r] * hsum[k];
            }
#pragma unroll
            for (int k = 0; k < BPP; k++) {
                const long long r = (v[k] + (1LL << 27)) >> 28;
                o[q * BPP + k] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
            }
        } else if (INTERP == 1) {
            const int fx = x & 0xFFFF, fy = y & 0xFFFF;
            const int ix1 = min(ix + 1, p.inW - 1), iy1 = min(iy + 1, p.inH - 1);
            const uint8_t *r0 = src + (size_t)iy * ss, *r1 = src + (size_t)iy1 * ss;
            // both horizontal taps of a row are 2*BPP consecutive bytes: one unaligned 8-byte load per row when
            // they are neighbours and the 8 bytes stay inside the row (gfx950 global loads need no alignment)
            unsigned long long t0 = 0, t1 = 0;
            const bool pair = ix1 == ix + 1 && BPP * ix + 8 <= p.inW * BPP;
            if (pair) {
                __builtin_memcpy(&t0, r0 + BPP * ix, 8);
                __builtin_memcpy(&t1, r1 + BPP * ix, 8);
            }
#pragma unroll
            for (int k = 0; k < BPP; k++) {
                int s00, s01, s10, s11;
                if (pair) {
                    s00 = (int)((t0 >> (8 * k)) & 0xFF); s01 = (int)((t0 >> (8 * (k + BPP))) & 0xFF);
                    s10 = (int)((t1 >> (8 * k)) & 0xFF); s11 = (int)((t1 >> (8 * (k + BPP))) & 0xFF);
                } else {
                    s00 = r0[BPP * ix + k]; s01 = r0[BPP * ix1 + k]; s10 = r1[BPP * ix + k]; s11 = r1[BPP * ix1 + k];
                }
                // interpolate_bilinear8 (vf_rotate.c:224-249): s0 = (2^16 - fx) * s00 + fx * s01, s1 alike, out = ((2^16 - fy) * s0 + fy * s1)
                // >> 32 in 64 bits — evaluated exactly without 64-bit arithmetic: s0 = s00 * 2^16 + fx * (s01 - s00) (a 24-bit multiply),
                // and of the 41-bit product fy * (s1 - s0) only the part above bit 16 is needed (the low 16 bits it drops cannot carry
                // into bit 32 of the sum): for the signed 25-bit D = s1 - s0, floor(fy * D / 2^16) = mulhi(fy << 16, D + 2^24) - fy * 2^8
                const int s0 = (s00 << 16) + m24(fx, s01 - s00);
                const int s1 = (s10 << 16) + m24(fx, s11 - s10);
                const int ph = (int)__umulhi((unsigned)fy << 16, (unsigned)(s1 - s0 + (1 << 24))) - (fy << 8);
                o[q * BPP + k] = (uint8_t)((s0 + ph) >> 16);
            }
        } else {
            const uint8_t *ps = src + (size_t)iy * ss + BPP * ix;
            if (BPP * ix + 4 <= p.inW * BPP) {                     // one unaligned dword covers the pixel
                unsigned t;
                __builtin_memcpy(&t, ps, 4);
#pragma unroll
                for (int k = 0; k < BPP; k++) o[q * BPP + k] = (uint8_t)(t >> (8 * k));
            } else {
#pragma unroll
                for (int k = 0; k < BPP; k++) o[q * BPP + k] = ps[k];
            }
        }
    }
    uint8_t *d = dst + (size_t)j * ds + (size_t)i0 * BPP;
    const int nx = min(4, p.outW - i0);
    const bool all = p.fillEnable || (valid[0] && valid[1] && valid[2] && valid[3]);
    if (aligned && nx == 4 && all) {
        unsigned w[BPP];
#pragma unroll
        for (int k = 0; k < BPP; k++)
            w[k] = (unsigned)o[4 * k] | ((unsigned)o[4 * k + 1] << 8) | ((unsigned)o[4 * k + 2] << 16) | ((unsigned)o[4 * k + 3] << 24);
#pragma unroll
        for (int k = 0; k < BPP; k++) reinterpret_cast<unsigned *>(d)[k] = w[k];
    } else {
        for (int q = 0; q < nx; q++)
            if (p.fillEnable || valid[q])
                for (int k = 0; k < BPP; k++) d[q * BPP + k] = o[q * BPP + k];
    }
}

// rows of a frame as a raw buffer resource (see S2Plane in k_scale_yuv2s.hip): the row offset travels in the instruction's scalar
// offset — which the hardware's range check does not see, so the check is not used (num_records = 2^32 - 1)
struct RotRows {
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ explicit RotRows(const uint8_t *p) : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p), 0, 0xFFFFFFFFu, 0x00020000)) {}
    __device__ __forceinline__ unsigned ld4(unsigned lane, unsigned row) const { return __builtin_amdgcn_raw_buffer_load_b32(r, lane, row, 0); }
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    __device__ __forceinline__ uint4 ld16(unsigned off) const { const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); return make_uint4(v.x, v.y, v.z, v.w); }
#else
    // hipcc's host pass (never executed) and the CPU emulation of the test suite
    const uint8_t *p;
    __host__ __device__ explicit RotRows(const uint8_t *q) : p(q) {}
    __host__ __device__ unsigned ld4(unsigned lane, unsigned row) const { unsigned v; std::memcpy(&v, p + (size_t)row + lane, 4); return v; }
    __host__ __device__ uint4 ld16(unsigned off) const { uint4 v; std::memcpy(&v, p + (size_t)off, 16); return v; }
#endif
};

// ---- interpolate_bilinear8 in single precision (round 4; FINDINGS R4-rotate, tools/ubench/rot_probe.hip) ------------------------
// vf_rotate.c:224-249 is floor(((2^16 - fy) * s0 + fy * s1) / 2^32) with s0 = (2^16 - fx) * s00 + fx * s01 in 64-bit integers.  Divided by
// 2^16, s0' = s00 + fx' * (s01 - s00) with fx' = fx / 2^16 has 8 + 16 bits: it IS a float, and the fused multiply-add that forms it is exact.
// The vertical step v = s0' + fy' * (s1' - s0') has 40 bits — but only floor(v) is wanted, and v >= 0: ONE fused multiply-add rounded TOWARD
// ZERO gives RZ(v) <= v, and floor(v) <= v is a float itself, so floor(v) <= RZ(v): floor(RZ(v)) == floor(v), exactly.  The wave runs with
// MODE.fp_round = toward zero (set at the kernel's first instruction; every other float operation here is exact, so the mode touches nothing
// else) and v_cvt_pk_u8_f32 — which converts under the same mode — drops the byte into its place in the output dword: 2^24 random blends and
// the corner cases against the integer form, 0 differences (and 8.4 M with the default mode).  v_pk_fma_f32 / v_pk_add_f32 do two values an
// instruction; v_cvt_f32_ubyteN takes a byte of a dword straight to a float: 8 instructions a sample where the integer form needs 14.
typedef float rotf2 __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void rot_round_toward_zero() { __builtin_amdgcn_s_setreg(1 | (0 << 6) | (1 << 11), 3); }      // hwreg(HW_REG_MODE, 0, 2)
__device__ __forceinline__ rotf2 rot_fma2(rotf2 a, rotf2 b, rotf2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ unsigned rot_put_u8(float v, unsigned byte, unsigned acc) { return __builtin_amdgcn_cvt_pk_u8_f32(v, byte, acc); }
#else
// the CPU emulation of the test suite: the same two operations under the same rounding (the products and sums here are exact in double)
__host__ __device__ static inline void rot_round_toward_zero() {}
__host__ __device__ static inline float rot_fma_rz1(float a, float b, float c)
{
    const double t = (double)a * (double)b + (double)c;
    float r = (float)t;
    if (std::fabs((double)r) > std::fabs(t)) r = std::nextafterf(r, 0.0f);
    return r;
}
__host__ __device__ static inline rotf2 rot_fma2(rotf2 a, rotf2 b, rotf2 c) { rotf2 r; r.x = rot_fma_rz1(a.x, b.x, c.x); r.y = rot_fma_rz1(a.y, b.y, c.y); return r; }
__host__ __device__ static inline unsigned rot_put_u8(float v, unsigned byte, unsigned acc)
{
    const int t = v <= 0.0f ? 0 : v >= 255.0f ? 255 : (int)v;
    return (acc & ~(0xFFu << (8 * byte))) | ((unsigned)t << (8 * byte));
}
#endif
// v_dot2_i32_i16 with a literal 0 to add to (the builtin's first use of an accumulator becomes v_dot2c_i32_i16 behind a v_mov_b32 0)
__device__ __forceinline__ int dot2z(int a, int b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return dot2(a, b, 0);
#endif
}
__device__ __forceinline__ float rot_byte_f(unsigned w, int n) { return (float)((w >> (8 * n)) & 0xFFu); }              // v_cvt_f32_ubyteN

// ---- the same walk with the source patch staged in LDS (round 3) ------------------------------------------------------------
// A block makes a 32 x 32 output tile: the source pixels its taps can touch lie in the bounding box of the tile's four corners (an
// affine map: the extremes are at the corners), at most 48 x 48 pixels at any angle; the box is loaded into LDS (dword-aligned
// start: `shift` bytes of lead-in, odd dword pitch) and every tap is read from there.
// The box is computed on CLAMPED coordinates — every tap index the walk forms is clamped to the image first (vf_rotate.c:463-492) — so
// it lies inside the image and the loads need no border case; the taps are clamped to the BOX, which for a pixel of the tile is the
// same index (the clamp is monotone and the box's ends are the clamped extremes) and keeps a ragged tile's surplus lanes inside LDS.
// Arithmetic: rotate_kernel's, tap for tap.
// The first form of this kernel was bound by VALU issue (16.1 M wave instructions a 4K rgb24 frame = 26 of its 29 us, profiles/r03t_*).
// This one spends the instructions on the blend: everything that is the same for the tile (corners, box, validity of the whole
// tile) is scalar; the box is read 16 bytes a lane through a buffer resource; the coordinates advance by additions; the last
// column / row is a zero weight instead of a second clamp; the tile whose four corners are valid skips the fill logic altogether;
// results are packed with byte permutes: 8.5 M wave instructions, 20.5 us (0.30 of the roofline; 16.3 us = 0.38 at 16 frames a launch,
// profiles/r03zr_rotate.txt).  What is left (profiles/r03zr_rotate_decomposition.txt): arithmetic and block structure 14.5 us, the
// box's loads +3.3, the stores +3.4, added rather than overlapped.
// Tried and not faster: two waves a tile (a thread: 8 pixels, GMAT_ROTATE_WAVES=2: 21.0 us), a block walking 2 / 4 / 8 tiles with the
// next box requested ahead (23 / 25 / 29 us, and 35-48 us in the first form: the chip overlaps independent blocks better than one
// block overlaps its own tiles), gathering aligned dwords in the direct form (45 us).
// tools/ubench/rot_phase.hip builds this file with GMAT_ROT_PHASE: every wave leaves its clock at the ends of its phases
#if defined(GMAT_ROT_PHASE)
__device__ unsigned long long g_rot_phase[(1 << 16) * 8];
#define ROT_PH(i) do { if ((threadIdx.x & 63) == 0) { const unsigned wid_ = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6); \
    if (wid_ < (1u << 16)) g_rot_phase[wid_ * 8 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define ROT_PH(i) do { } while (0)
#endif
template <int BPP, int INTERP, int NWV>
struct RotTile {                                            // the 32 x 32 tile's LDS geometry
    static constexpr int TW = 32, TBH = 32, BMAX = 50;
    static constexpr int PD = ((BMAX * BPP + 6) / 4 + 2) | 1;      // dwords per LDS row: 50 pixels + lead-in + the two dwords an 8-byte read may run over, odd
    static constexpr int NR = (BMAX + NWV - 1) / NWV;              // loader rounds: wave w takes rows w, w + NWV, ...
    static constexpr int BOX = (NR * NWV + 1) * PD;                // the loader's NR rounds of NWV rows (rows past the box are never read), and
                                                                   // at the frame's last row the pair's lower row is read with weight 0
    static constexpr int CW = INTERP == 2 ? 512 : 2;               // cubic: the four weights of each 8-bit fraction as two int16 pairs
};

// one 32 x 32 tile (bx, by) by the block's NWV waves: box -> LDS, one barrier, the integer walk
template <int BPP, int INTERP, int NWV>
__device__ __forceinline__ void rot_tile_general(const uint8_t *src, int ss, uint8_t *dst, int ds, const RotateParams &p, int aligned, int bx, int by,
                                                 unsigned *box, unsigned *cw)
{
    constexpr int TW = RotTile<BPP, INTERP, NWV>::TW, TBH = RotTile<BPP, INTERP, NWV>::TBH, BMAX = RotTile<BPP, INTERP, NWV>::BMAX;
    constexpr int PD = RotTile<BPP, INTERP, NWV>::PD, NR = RotTile<BPP, INTERP, NWV>::NR;
    constexpr bool cubic = INTERP == 2;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int iLo = bx * TW, iHi = min(iLo + TW, p.outW) - 1, jLo = by * TBH, jHi = min(jLo + TBH, p.outH) - 1;
    const int xb = p.X0 + jLo * p.s + iLo * p.c, yb = p.Y0 + jLo * p.c - iLo * p.s;     // the tile's first pixel
    // extremes of x1 = x >> 16, y1 = y >> 16 over the tile: x and y are sums of a term in i and a term in j, the shift is monotone
    const int xi = (iHi - iLo) * p.c, xj = (jHi - jLo) * p.s, yi = -(iHi - iLo) * p.s, yj = (jHi - jLo) * p.c;
    const int minx = (xb + min(xi, 0) + min(xj, 0)) >> 16, maxx = (xb + max(xi, 0) + max(xj, 0)) >> 16;
    const int miny = (yb + min(yi, 0) + min(yj, 0)) >> 16, maxy = (yb + max(yi, 0) + max(yj, 0)) >> 16;
    // every pixel of the tile valid (vf_rotate.c:463): the validity region is a box and the extremes are at the corners
    const bool allValid = minx >= -1 && maxx <= p.inW && miny >= -1 && maxy <= p.inH;
    // cubic taps are clamp(x1 - 1 + t), t = 0..3; the bilinear pair is clamp(x1) and min(clamp(x1) + 1, W - 1) — the clamp comes FIRST
    // (x1 = -1 reads pixels 0 and 1), so the box ends one past the clamped maximum
    const int bx0 = min(max(minx - (cubic ? 1 : 0), 0), p.inW - 1), by0 = min(max(miny - (cubic ? 1 : 0), 0), p.inH - 1);
    const int bxm = min(max(maxx, 0), p.inW - 1), bym = min(max(maxy, 0), p.inH - 1);   // the clamped maxima of x1, y1
    const int bx1 = cubic ? min(max(maxx + 2, 0), p.inW - 1) : min(bxm + 1, p.inW - 1);
    const int by1 = cubic ? min(max(maxy + 2, 0), p.inH - 1) : min(bym + 1, p.inH - 1);
    const int bh = by1 - by0 + 1, shift = (bx0 * BPP) & 3, gd0 = (bx0 * BPP) >> 2;
    const int nDw = (shift + (bx1 - bx0 + 1) * BPP + 3) >> 2, rowBytes = p.inW * BPP;
    {   // The box's rows as a raw buffer resource.  The fast form reads 16 bytes a lane: the block's threads are laid flat over the
        // box's (row, 16-byte piece) pairs, two or three rounds, every load issued before the first LDS store: 8 to 12 instructions of
        // a kilobyte a tile where a wave per row issues 52 of 160 bytes.  A piece may run up to 12 bytes past the box's last dword:
        // inside the frame everywhere but at the frame's end — the tile that would leave it takes the dword form (a round past
        // the box's last row reads that row again, no branch), whose last dword of the last row reads bytes.
        const int ncol = (nDw + 3) >> 2, nitem = bh * ncol;
        // (with a pitch of a few bytes the piece past a row's end can leave the frame from a row above the last one: the test is on
        // the last byte the pieces of the box's last row touch)
        const bool wide = by1 * ss + 4 * gd0 + 16 * ncol <= (p.inH - 1) * ss + rowBytes;
        const bool tailRow = by1 == p.inH - 1 && 4 * (gd0 + nDw) > rowBytes;
        const RotRows rows(src + (size_t)by0 * ss + 4 * (size_t)gd0);
        if (cubic) {                                        // 256 fractions
            for (int fr = threadIdx.x; fr < 256; fr += 64 * NWV) {
                int w4[4];
                rot_cubic_w(fr, w4);
                cw[2 * fr] = (unsigned)(w4[0] & 0xFFFF) | ((unsigned)w4[1] << 16);
                cw[2 * fr + 1] = (unsigned)(w4[2] & 0xFFFF) | ((unsigned)w4[3] << 16);
            }
        }
        if (wide) {
            constexpr int NT = 64 * NWV, KMAX = (BMAX * (PD / 4) + NT - 1) / NT;
            const unsigned mrcp = 65536u / (unsigned)ncol + 1u;             // item / ncol = (item * mrcp) >> 16 for item < 5000, ncol <= 13
            uint4 v[KMAX];
            int lo[KMAX];
#pragma unroll
            for (int k = 0; k < KMAX; k++) {
                const int item = (int)threadIdx.x + k * NT;
                lo[k] = -1;
                if (item < nitem) {
                    const int q = (int)(((unsigned)item * mrcp) >> 16), c4 = item - m24(q, ncol);
                    v[k] = rows.ld16((unsigned)(m24(q, ss) + 16 * c4));           // 0 < ss < 2^23 (launcher)
                    lo[k] = q * PD + 4 * c4;
                }
            }
    #pragma unroll
            for (int k = 0; k < KMAX; k++)
                if (lo[k] >= 0) { box[lo[k]] = v[k].x; box[lo[k] + 1] = v[k].y; box[lo[k] + 2] = v[k].z; box[lo[k] + 3] = v[k].w; }
        } else if (lane < nDw) {
            const unsigned lo4 = 4u * (unsigned)lane, offLast = (unsigned)((bh - 1) * ss), step = (unsigned)(NWV * ss);   // ss > 0 (launcher)
            unsigned *bw = box + wave * PD + lane;
            unsigned v[NR];
            if (!tailRow) {
                unsigned off = (unsigned)(wave * ss);
#pragma unroll
                for (int k = 0; k < NR; k++, off += step) v[k] = rows.ld4(lo4, min(off, offLast));
#pragma unroll
                for (int k = 0; k < NR; k++) bw[NWV * k * PD] = v[k];
            } else {
                unsigned off = (unsigned)(wave * ss);
#pragma unroll
                for (int k = 0; k < NR; k++, off += step) {
                    v[k] = 0;
                    if (!(lane == nDw - 1 && off >= offLast)) v[k] = rows.ld4(lo4, min(off, offLast));
                }
#pragma unroll
                for (int k = 0; k < NR; k++) bw[NWV * k * PD] = v[k];
                if (lane == nDw - 1 && (bh - 1) % NWV == wave) {
                    const uint8_t *gt = src + (size_t)by1 * ss + 4 * (size_t)(gd0 + lane);
                    unsigned tl = 0;
                    for (int b = 0; b < rowBytes - 4 * (gd0 + lane); b++) tl |= (unsigned)gt[b] << (8 * b);
                    box[(bh - 1) * PD + lane] = tl;         // after this lane's own stores (LDS keeps a wave's order)
                }
            }
        }
    }
    __syncthreads();
    // a wave makes 8 rows of 32 pixels at a time (a lane: 4 adjacent pixels), the block's waves 8 * NWV rows a pass
    const int ir = (lane & 7) * 4, i0 = iLo + ir;
    if (i0 >= p.outW) return;
    const int boK = shift - bx0 * BPP - by0 * PD * 4;       // byte offset in LDS of pixel (ix, iy): ix * BPP + iy * PD * 4 + boK
    const int hiX = (bxm << 16) | (bxm == p.inW - 1 ? 0 : 0xFFFF), hiY = (bym << 16) | (bym == p.inH - 1 ? 0 : 0xFFFF);
    // a frame one column / row thick: its first column is its last one too (x1 = -1 lands on it WITH a fraction): the weight is masked
    const int fxMask = p.inW == 1 ? 0 : 0xFFFF, fyMask = p.inH == 1 ? 0 : 0xFFFF;
    const int nx = min(4, p.outW - i0);
#pragma unroll 1
    for (int jr = wave * 8 + (lane >> 3); jr < TBH; jr += 8 * NWV) {
    const int j = jLo + jr;
    if (j >= p.outH) break;
    const int x0 = xb + m24(jr, p.s) + m24(ir, p.c), y0 = yb + m24(jr, p.c) - m24(ir, p.s);     // |s|, |c| <= 2^16
    uint8_t *d = dst + (size_t)j * ds + (size_t)i0 * BPP;

    // one output pixel; the result of channel k is BYTE 2 of R[k] (bilinear: the 24-bit sum before its >> 16)
    auto pixel = [&](int q, unsigned (&R)[BPP], bool &valid) {
        const int x = x0 + q * p.c, y = y0 - q * p.s;
        const int x1 = x >> 16, y1 = y >> 16;
        valid = (unsigned)(x1 + 1) <= (unsigned)(p.inW + 1) && (unsigned)(y1 + 1) <= (unsigned)(p.inH + 1);
        if (INTERP == 2) {
            const unsigned wx01 = cw[2 * ((x >> 8) & 0xFF)], wx23 = cw[2 * ((x >> 8) & 0xFF) + 1];
            const unsigned wy01 = cw[2 * ((y >> 8) & 0xFF)], wy23 = cw[2 * ((y >> 8) & 0xFF) + 1];
            const int wx[4] = {(int)(short)(wx01 & 0xFFFF), (int)wx01 >> 16, (int)(short)(wx23 & 0xFFFF), (int)wx23 >> 16};
            const int wy[4] = {(int)(short)(wy01 & 0xFFFF), (int)wy01 >> 16, (int)(short)(wy23 & 0xFFFF), (int)wy23 >> 16};
            // the 38-bit vertical sum in doubles (exact; v_fma_f64 is full rate on gfx950), see rotate_mt_kernel
            const double wyd[4] = {(double)wy[0], (double)wy[1], (double)wy[2], (double)wy[3]};
            double acc[BPP];
#pragma unroll
            for (int k = 0; k < BPP; k++) acc[k] = 134217728.0;         // 2^27
            // interior: the four taps of a row are 4 * BPP consecutive bytes (BPP dwords after v_alignbyte_b32), the four rows consecutive
            const bool inner = x1 - 1 >= bx0 && x1 + 2 <= bx1 && y1 - 1 >= by0 && y1 + 2 <= by1;
            if (inner) {
                const int bo = (x1 - 1) * BPP + boK;
                const unsigned *w = reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(box) + (m24(y1 - 1, PD * 4) + (bo & ~3)));
                const unsigned sh = (unsigned)bo & 3u;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    unsigned c[BPP];
#pragma unroll
                    for (int n = 0; n < BPP; n++) c[n] = __builtin_amdgcn_alignbyte(w[r * PD + n + 1], w[r * PD + n], sh);
#pragma unroll
                    for (int k = 0; k < BPP; k++) {
                        // taps 0, 1 and 2, 3 of channel k as int16 pairs (bytes k, k + BPP and k + 2 BPP, k + 3 BPP of the row's 4 BPP):
                        // one byte permute a pair, one v_dot2_i32_i16 a pair against the table's packed weights
                        constexpr unsigned Z = 0x0C000C00u;
                        const int n0 = k, n1 = BPP + k, n2 = 2 * BPP + k, n3 = 3 * BPP + k;
                        const unsigned p01 = __builtin_amdgcn_perm(c[n1 >> 2], c[n0 >> 2], Z | (unsigned)(n0 & 3) | ((unsigned)(4 + (n1 & 3)) << 16));
                        const unsigned p23 = __builtin_amdgcn_perm(c[n3 >> 2], c[n2 >> 2], Z | (unsigned)(n2 & 3) | ((unsigned)(4 + (n3 & 3)) << 16));
                        const int hs = dot2((int)p01, (int)wx01, dot2z((int)p23, (int)wx23));
                        acc[k] = __builtin_fma(wyd[r], (double)hs, acc[k]);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int ry = min(max(y1 - 1 + r, by0), by1);
                    int hsum[BPP];
#pragma unroll
                    for (int k = 0; k < BPP; k++) hsum[k] = 0;
#pragma unroll
                    for (int tt = 0; tt < 4; tt++) {
                        const int bo = min(max(x1 - 1 + tt, bx0), bx1) * BPP + boK + m24(ry, PD * 4);
#pragma unroll
                        for (int k = 0; k < BPP; k++) hsum[k] += wx[tt] * (int)((box[(bo + k) >> 2] >> (8 * ((bo + k) & 3))) & 0xFF);
                    }
#pragma unroll
                    for (int k = 0; k < BPP; k++) acc[k] = __builtin_fma(wyd[r], (double)hsum[k], acc[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < BPP; k++)           // floor and truncation differ below zero only, where the clamp gives 0 either way
                R[k] = (unsigned)min(max((int)(acc[k] * (1.0 / 268435456.0)), 0), 255) << 16;
        } else {
            // the upper clamp on the 16.16 coordinate itself: past the frame's last column / row the pair's second tap is the first
            // one again, which is a zero weight — hiX / hiY end in 0x0000 there and in 0xFFFF inside the frame (the lower clamp is
            // on the integer part only: x1 = -1 reads pixels 0 and 1 WITH its fraction, vf_rotate.c:463-492)
            const int xm = min(x, hiX), ym = min(y, hiY);
            const int ix = max(xm >> 16, bx0), iy = max(ym >> 16, by0);
            const int bo = m24(ix, BPP) + boK;
            const unsigned *w = reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(box) + (m24(iy, PD * 4) + (bo & ~3)));
            const unsigned sh = (unsigned)bo & 3u;
            if (INTERP == 1) {
                const int fx = xm & fxMask, fy = ym & fyMask;
                const unsigned a0 = __builtin_amdgcn_alignbyte(w[1], w[0], sh), b0 = __builtin_amdgcn_alignbyte(w[PD + 1], w[PD], sh);
                unsigned a1 = 0, b1 = 0;
                if (BPP > 2) { a1 = __builtin_amdgcn_alignbyte(w[2], w[1], sh); b1 = __builtin_amdgcn_alignbyte(w[PD + 2], w[PD + 1], sh); }
                const unsigned fyh = (unsigned)fy << 16;
                const int fyl = m24(fy, -256);
#pragma unroll
                for (int k = 0; k < BPP; k++) {
                    const int k1 = k + BPP;                 // byte index of the right tap's channel
                    const int s00 = (int)((a0 >> (8 * k)) & 0xFF);
                    const int s01 = (int)(((k1 < 4 ? a0 >> (8 * (k1 & 3)) : a1 >> (8 * (k1 & 3)))) & 0xFF);
                    const int s11 = (int)(((k1 < 4 ? b0 >> (8 * (k1 & 3)) : b1 >> (8 * (k1 & 3)))) & 0xFF);
                    // (s10 << 16) + 2^24 and s10 itself by one byte permute each (selector 0x0C = 0, byte 4 = the constant's 0x01)
                    const int s10h = (int)__builtin_amdgcn_perm(1u, b0, 0x04000c0cu | ((unsigned)k << 16));
                    const int s10 = (int)((b0 >> (8 * k)) & 0xFF);
                    const int s0 = (s00 << 16) + m24(fx, s01 - s00);          // exact without 64-bit arithmetic: see rotate_kernel
                    const int s1h = s10h + m24(fx, s11 - s10);                // s1 + 2^24
                    R[k] = (unsigned)(s0 + fyl + (int)__umulhi(fyh, (unsigned)(s1h - s0)));
                }
            } else {
                const unsigned a0 = __builtin_amdgcn_alignbyte(w[1], w[0], sh);
#pragma unroll
                for (int k = 0; k < BPP; k++) R[k] = (a0 >> (8 * k)) << 16;
            }
        }
    };
    // bytes 2 of four values -> one dword
    auto pack4 = [](unsigned v0, unsigned v1, unsigned v2, unsigned v3) -> unsigned {
        return __builtin_amdgcn_perm(v1, v0, 0x0c0c0602u) | __builtin_amdgcn_perm(v3, v2, 0x06020c0cu);
    };
    unsigned V[4 * BPP];
    bool valid[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        unsigned R[BPP];
        pixel(q, R, valid[q]);
#pragma unroll
        for (int k = 0; k < BPP; k++) V[q * BPP + k] = R[k];
    }
    if (allValid && aligned && nx == 4) {                   // allValid is the tile's: no fill logic at all
#pragma unroll
        for (int k = 0; k < BPP; k++) reinterpret_cast<unsigned *>(d)[k] = pack4(V[4 * k], V[4 * k + 1], V[4 * k + 2], V[4 * k + 3]);
        continue;
    }
    if (allValid) valid[0] = valid[1] = valid[2] = valid[3] = true;       // surplus lanes of a ragged tile are not the tile's pixels
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (!valid[q]) {
#pragma unroll
            for (int k = 0; k < BPP; k++) V[q * BPP + k] = ((p.fill >> (8 * k)) & 0xFF) << 16;
        }
    const bool all = p.fillEnable || (valid[0] && valid[1] && valid[2] && valid[3]);
    if (aligned && nx == 4 && all) {
#pragma unroll
        for (int k = 0; k < BPP; k++) reinterpret_cast<unsigned *>(d)[k] = pack4(V[4 * k], V[4 * k + 1], V[4 * k + 2], V[4 * k + 3]);
    } else {
        for (int q = 0; q < nx; q++)
            if (p.fillEnable || valid[q])
                for (int k = 0; k < BPP; k++) d[q * BPP + k] = (uint8_t)(V[q * BPP + k] >> 16);
    }
    }
}

template <int BPP, int INTERP, int NWV>
__global__ __launch_bounds__(64 * NWV) void rotate_lds_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds, RotateParams p,
                                                         int aligned, int nbx, int nby, OpFrames fr)
{
    if (gridDim.z > 1) { src = fr.src[blockIdx.z]; dst = fr.dst[blockIdx.z]; }      // a frame table: grid.z = frame
    __shared__ unsigned box[RotTile<BPP, INTERP, NWV>::BOX];
    __shared__ unsigned cw[RotTile<BPP, INTERP, NWV>::CW];
    // grid (8 * nbx, ceil(nby / 8)): workgroups go to the XCDs round-robin in dispatch order, so blockIdx.x & 7 IS the XCD, and XCD k
    // walks tile rows k * gridDim.y ... — a contiguous band of the frame per L2, without a division
    const int bx = blockIdx.x >> 3, by = (blockIdx.x & 7) * gridDim.y + blockIdx.y;
    if (by >= nby) return;
    rot_tile_general<BPP, INTERP, NWV>(src, ss, dst, ds, p, aligned, bx, by, box, cw);
}

// ---- the macro-tile form (round 4): 64 x 32 outputs a block, the whole interior tile on a lean path --------------------------------------
// tools/ubench/rot_phase.hip put clocks into rotate_lds_kernel's waves (profiles/r04t_rotate_phases.txt): of a wave's life the blend is 23 %;
// 32 % goes before the first load leaves (170 scalar instructions a wave — corners, clamps, the loader's division — four waves a tile each
// redoing them, 5.6 M a 4K frame on scalar units that issue one a cycle a CU: 9 us of the frame's 20 by themselves), 20 % into issuing
// the loads, 10 % into the barrier.  Making the blend cheaper (single precision, above: 262 -> 190 instructions a wave) moved nothing.
// So: tiles twice as large (half the waves, each with two passes of work behind one preamble and one barrier); whatever is the same for
// every whole tile comes from the host (RotMT: the extremes of the coordinate offsets over 64 x 32 outputs); a tile that is whole, all valid
// and whose box met no clamp knows its box from four additions; the loader is laid out by rows (LPR threads a row of 16-byte pieces, no
// division) and stores 16 bytes an LDS instruction (the pitch is a multiple of 4 dwords, = 4 mod 8); coordinates are relative to the box;
// no fill logic, no clamp, no zero-weight rule.  Everything else — the frame's rim, ragged tiles, unaligned destinations — is the integer
// walk above on the macro tile's two halves.
// Tried on top of this and dropped (FINDINGS R4-rotate): persistent blocks that request the next tile's box before blending this one (28 us
// where one tile a block takes 20: a tile's blend is far shorter than a load's way, eight short-lived blocks a CU overlap better, and the
// registers of a box in flight halve the occupancy); one pixel a dword in LDS for 3 / 4 bytes a pixel (no v_alignbyte, but 73 VGPRs and a
// loader of byte permutes: 21.9 against 20.5 us); reads at their own byte address (35 us).
struct RotMT { int dxLo, dxHi, dyLo, dyHi; };               // extremes of the coordinate offsets over 64 x 32 outputs

template <int BPP, int MRG = 0> struct RotMTGeom {      // MRG: the cubic's extra column / row on every side
    static constexpr int MW = 64, MH = 32, BM = 74 + 2 * MRG;  // the box of 64 x 32 outputs: |extent| <= hypot(63, 31) = 70.3, + 1 (floors) + 1 (the pair) + 1
    static constexpr int NCOL = (((3 + BM * BPP + 3) / 4) + 3) / 4;                     // 16-byte pieces a row, at most
    static constexpr int PD = (4 * NCOL) % 8 == 4 ? 4 * NCOL : 4 * NCOL + 4;            // dwords a row: whole pieces, = 4 mod 8
    static constexpr int LPR = NCOL <= 8 ? 8 : NCOL <= 16 ? 16 : 32, RPR = 256 / LPR, KMAX = (BM + RPR - 1) / RPR;     // loader: LPR threads a row
    static constexpr int BOX = BM * PD;
};

template <int BPP, int INTERP>
__global__ __launch_bounds__(256) void rotate_mt_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds, RotateParams p, RotMT mt,
                                                        int aligned, int nbx, int nby, OpFrames fr)
{
    ROT_PH(0);
    constexpr int MRG = INTERP == 2 ? 1 : 0;                // cubic: taps x1 - 1 .. x1 + 2 on rows y1 - 1 .. y1 + 2
    typedef RotMTGeom<BPP, MRG> G;
    typedef RotTile<BPP, INTERP, 4> T;
    constexpr int PD = G::PD;
    if (INTERP == 1) rot_round_toward_zero();               // the single-precision blend below; nothing else in this kernel rounds
    if (gridDim.z > 1) { src = fr.src[blockIdx.z]; dst = fr.dst[blockIdx.z]; }      // a frame table: grid.z = frame
    __shared__ __attribute__((aligned(16))) unsigned box[(G::BOX > T::BOX ? G::BOX : T::BOX) + 4];    // + the dword a shifted read's upper half may name
    __shared__ unsigned cw[T::CW];
    // grid (8 * macro columns, ceil(nby / 8)): blockIdx.x & 7 is the XCD (see rotate_lds_kernel).  (Bands balanced to a row — XCD k walks
    // [k * nby / 8, (k + 1) * nby / 8) — cost 3 % for the two multiplies in front of everything else: the makespan is the same 9 rows at 4K.)
    const int bxM = blockIdx.x >> 3, by = (blockIdx.x & 7) * gridDim.y + blockIdx.y;
    if (by >= nby) return;
    ROT_PH(7);
    const int iLo = bxM * G::MW, jLo = by * G::MH;
    const int xb = p.X0 + jLo * p.s + iLo * p.c, yb = p.Y0 + jLo * p.c - iLo * p.s;     // the tile's first pixel
    // the box: columns minx .. maxx + 1 (pairs: x1, x1 + 1), rows alike; the cubic one more on every side
    const int minx = ((xb + mt.dxLo) >> 16) - MRG, maxx = ((xb + mt.dxHi) >> 16) + 1 + MRG, miny = ((yb + mt.dyLo) >> 16) - MRG, maxy = ((yb + mt.dyHi) >> 16) + 1 + MRG;
    const int bw = maxx + 1 - minx, bh = maxy + 1 - miny;
    const int gd0 = (minx * BPP) >> 2, shift = (minx * BPP) & 3, ncol = (((shift + bw * BPP + 3) >> 2) + 3) >> 2;
    const bool fast = aligned && iLo + G::MW <= p.outW && jLo + G::MH <= p.outH && minx >= 0 && maxx <= p.inW - 1 && miny >= 0 && maxy <= p.inH - 1 &&
                      bw <= G::BM && bh <= G::BM &&
                      maxy * ss + 4 * gd0 + 16 * ncol <= (p.inH - 1) * ss + p.inW * BPP;           // the last row's 16-byte pieces end inside the frame
    // every pixel of the tile invalid (its source position outside [-1, in]: the box of the tile's positions misses that range in x or in y — at 17
    // degrees a tenth of a 4K frame's tiles): the fill colour, or nothing
    {
        const int tx0 = (xb + mt.dxLo) >> 16, tx1 = (xb + mt.dxHi) >> 16, ty0 = (yb + mt.dyLo) >> 16, ty1 = (yb + mt.dyHi) >> 16;
        if ((tx1 < -1 || tx0 > p.inW || ty1 < -1 || ty0 > p.inH) && iLo + G::MW <= p.outW && jLo + G::MH <= p.outH) {
            if (!p.fillEnable) return;
            if (aligned) {
                // the tile's rows are 16 * BPP dwords: pixel (4 n + q)'s channel k is byte (q * BPP + k) & 3 of dword (q * BPP + k) >> 2 of every 4-pixel group
                unsigned F[BPP];
#pragma unroll
                for (int n = 0; n < BPP; n++) {
                    F[n] = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++) F[n] |= ((p.fill >> (8 * ((4 * n + b) % BPP))) & 0xFFu) << (8 * b);
                }
                const int li = threadIdx.x & 15;
                for (int jr = (int)threadIdx.x >> 4; jr < G::MH; jr += 16) {
                    unsigned *d = reinterpret_cast<unsigned *>(dst + (size_t)(jLo + jr) * ds + (size_t)(iLo + 4 * li) * BPP);
#pragma unroll
                    for (int k = 0; k < BPP; k++) d[k] = F[k];
                }
                return;
            }
        }
    }
    if (!fast) {
        // the frame's rim, ragged tiles, unaligned destinations: the integer walk on the macro tile's two halves
#pragma unroll                                               // (as a loop: 4 % slower on the WHOLE tiles — the code's layout, not its work)
        for (int half = 0; half < 2; half++) {
            if (half) __syncthreads();                       // the second half's box goes where the first one's is still being read
            if (2 * bxM + half < nbx) rot_tile_general<BPP, INTERP, 4>(src, ss, dst, ds, p, aligned, 2 * bxM + half, by, box, cw);
        }
        return;
    }
    ROT_PH(1);
    if (INTERP == 2) {                                      // the four weights of each 8-bit fraction as two int16 pairs: a thread, a fraction
        int w4[4];
        rot_cubic_w((int)threadIdx.x, w4);
        cw[2 * threadIdx.x] = (unsigned)(w4[0] & 0xFFFF) | ((unsigned)w4[1] << 16);
        cw[2 * threadIdx.x + 1] = (unsigned)(w4[2] & 0xFFFF) | ((unsigned)w4[3] << 16);
    }
    {
        const RotRows rows(src + (size_t)miny * ss + 4 * (size_t)gd0);
        const int lrow = (int)threadIdx.x / G::LPR, lcol = (int)threadIdx.x & (G::LPR - 1);
        const unsigned off = (unsigned)(m24(lrow, ss) + 16 * lcol), step = (unsigned)(G::RPR * ss);
        uint4 v[G::KMAX];
#pragma unroll
        for (int k = 0; k < G::KMAX; k++)
            if (lcol < ncol && lrow + k * G::RPR < bh) v[k] = rows.ld16(off + (unsigned)k * step);
        ROT_PH(2);
#pragma unroll
        for (int k = 0; k < G::KMAX; k++)
            if (lcol < ncol && lrow + k * G::RPR < bh) *reinterpret_cast<uint4 *>(&box[(lrow + k * G::RPR) * PD + 4 * lcol]) = v[k];
    }
    ROT_PH(3);
    __syncthreads();
    ROT_PH(4);
    // a wave makes 4 rows of 64 pixels a pass (a lane: 4 adjacent pixels), the block 16 rows a pass
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ir = (lane & 15) * 4;
    const int xl = xb - (minx << 16) + m24(ir, p.c), yl = yb - (miny << 16) - m24(ir, p.s);     // relative to the box, this lane's column
    constexpr float K16 = 1.0f / 65536.0f;
    if (INTERP == 2) {
        // The cubic of the test suite's checker (orc_vf.c; rotate_nvcv's has no integer reference): rows first, 14-bit weights, out = clamp((sum_r wy[r]
        // * hs[r] + 2^27) >> 28).  The kernel is bound by VALU issue (27 M instructions a 4K frame in the round-3 form, profiles/r04_rotate.txt):
        // a row's four taps of a channel are two v_perm_b32 + two v_dot2_i32_i16 against the table's packed weights as before; the 38-bit
        // vertical sum — four 64-bit multiply-adds, a 64-bit shift and a clamp a channel — is four v_fma_f64 on exact doubles (full rate on
        // gfx950: 4.7 cycles, tools/ubench/rot_probe.hip), a scale by 2^-28, a conversion that truncates and one v_med3 (what truncation does
        // to a negative sum the clamp to 0 hides); no border case, no clamped indices in the whole tile.
#pragma unroll 1
        for (int jr = wave * 4 + (lane >> 4); jr < G::MH; jr += 16) {
            const int x0 = xl + m24(jr, p.s), y0 = yl + m24(jr, p.c);
            unsigned W[BPP];
#pragma unroll
            for (int k = 0; k < BPP; k++) W[k] = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int x = x0 + q * p.c, y = y0 - q * p.s;            // relative to the box: x >> 16 >= 1
                const unsigned wx01 = cw[2 * ((x >> 8) & 0xFF)], wx23 = cw[2 * ((x >> 8) & 0xFF) + 1];
                const unsigned wy01 = cw[2 * ((y >> 8) & 0xFF)], wy23 = cw[2 * ((y >> 8) & 0xFF) + 1];
                const double wy[4] = {(double)(int)(short)(wy01 & 0xFFFF), (double)((int)wy01 >> 16), (double)(int)(short)(wy23 & 0xFFFF), (double)((int)wy23 >> 16)};
                const int bo = m24((y >> 16) - 1, PD * 4) + m24((x >> 16) - 1, BPP) + shift;
                const unsigned *w = reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(box) + (bo & ~3));
                const unsigned sh = (unsigned)bo & 3u;
                double acc[BPP];
#pragma unroll
                for (int k = 0; k < BPP; k++) acc[k] = 134217728.0;     // 2^27
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    unsigned c[BPP];                         // the row's four taps: 4 BPP consecutive bytes
#pragma unroll
                    for (int n = 0; n < BPP; n++) c[n] = BPP == 4 ? w[r * PD + n] : __builtin_amdgcn_alignbyte(w[r * PD + n + 1], w[r * PD + n], sh);
#pragma unroll
                    for (int k = 0; k < BPP; k++) {
                        // taps 0, 1 and 2, 3 of channel k as int16 pairs (bytes k, k + BPP and k + 2 BPP, k + 3 BPP of the row's 4 BPP)
                        constexpr unsigned Z = 0x0C000C00u;
                        const int n0 = k, n1 = BPP + k, n2 = 2 * BPP + k, n3 = 3 * BPP + k;
                        const unsigned p01 = __builtin_amdgcn_perm(c[n1 >> 2], c[n0 >> 2], Z | (unsigned)(n0 & 3) | ((unsigned)(4 + (n1 & 3)) << 16));
                        const unsigned p23 = __builtin_amdgcn_perm(c[n3 >> 2], c[n2 >> 2], Z | (unsigned)(n2 & 3) | ((unsigned)(4 + (n3 & 3)) << 16));
                        const int hs = dot2((int)p01, (int)wx01, dot2z((int)p23, (int)wx23));
                        acc[k] = __builtin_fma(wy[r], (double)hs, acc[k]);          // exact: |sum| < 2^40
                    }
                }
#pragma unroll
                for (int k = 0; k < BPP; k++) {
                    const int o = min(max((int)(acc[k] * (1.0 / 268435456.0)), 0), 255);
                    const int n = q * BPP + k;
                    W[n >> 2] |= (unsigned)o << (8 * (n & 3));
                }
            }
            unsigned *d = reinterpret_cast<unsigned *>(dst + (size_t)(jLo + jr) * ds + (size_t)(iLo + ir) * BPP);
#pragma unroll
            for (int k = 0; k < BPP; k++) d[k] = W[k];
        }
        return;
    }
#pragma unroll 1
    for (int jr = wave * 4 + (lane >> 4); jr < G::MH; jr += 16) {
        const int x0 = xl + m24(jr, p.s), y0 = yl + m24(jr, p.c);
        unsigned TT[4][2], UU[4][2];                         // per pixel: the pair of the upper row and of the lower one, 2 BPP bytes each
        float FX[4], FY[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int x = x0 + q * p.c, y = y0 - q * p.s;    // >= 0: relative to the box
            const int bo = m24(y >> 16, PD * 4) + m24(x >> 16, BPP) + shift;
            TT[q][1] = UU[q][1] = 0; UU[q][0] = 0;
            if (BPP == 1) {
                // one byte a read, four reads off one address: the LDS unit does what v_alignbyte, the two masks and the byte picks did
                // (97 -> 69 VALU instructions a pass; it then is the LDS unit that is busy: 13.9 against 14.5 us)
                const uint8_t *pb = reinterpret_cast<const uint8_t *>(box) + bo;
                TT[q][0] = pb[0];
                if (INTERP) { TT[q][0] |= (unsigned)pb[1] << 8; UU[q][0] = (unsigned)pb[PD * 4] | ((unsigned)pb[PD * 4 + 1] << 8); }
            } else {
                // aligned dwords and a byte shift: a read at its own byte address is legal on gfx950 (the compiler emits it for a memcpy)
                // but is served a lane at a time — 35 us a 4K frame where this form takes 20 (profiles/r04r_*)
                const unsigned *w = reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(box) + (bo & ~3));
                const unsigned sh = (unsigned)bo & 3u;
                if (BPP == 4) { TT[q][0] = w[0]; if (INTERP) { TT[q][1] = w[1]; UU[q][0] = w[PD]; UU[q][1] = w[PD + 1]; } }
                else {
                    TT[q][0] = __builtin_amdgcn_alignbyte(w[1], w[0], sh);
                    if (INTERP) {
                        UU[q][0] = __builtin_amdgcn_alignbyte(w[PD + 1], w[PD], sh);
                        if (BPP == 3) { TT[q][1] = __builtin_amdgcn_alignbyte(w[2], w[1], sh); UU[q][1] = __builtin_amdgcn_alignbyte(w[PD + 2], w[PD + 1], sh); }
                    }
                }
            }
            FX[q] = (float)(x & 0xFFFF) * K16; FY[q] = (float)(y & 0xFFFF) * K16;
        }
        unsigned W[BPP];
#pragma unroll
        for (int k = 0; k < BPP; k++) W[k] = 0;
        if (INTERP == 1) {
            // the lane's 4 BPP samples two at a time (v_pk_*_f32): sample n = channel n % BPP of pixel n / BPP = byte n of the output
#pragma unroll
            for (int m = 0; m < 2 * BPP; m++) {
                const int na = 2 * m, nb = 2 * m + 1, qa = na / BPP, qb = nb / BPP, ka = na % BPP, kb = nb % BPP, ka1 = ka + BPP, kb1 = kb + BPP;
                const rotf2 fx = {FX[qa], FX[qb]}, fy = {FY[qa], FY[qb]};
                const rotf2 t0 = {rot_byte_f(TT[qa][0], ka), rot_byte_f(TT[qb][0], kb)};
                const rotf2 t1 = {rot_byte_f(TT[qa][ka1 >> 2], ka1 & 3), rot_byte_f(TT[qb][kb1 >> 2], kb1 & 3)};
                const rotf2 u0 = {rot_byte_f(UU[qa][0], ka), rot_byte_f(UU[qb][0], kb)};
                const rotf2 u1 = {rot_byte_f(UU[qa][ka1 >> 2], ka1 & 3), rot_byte_f(UU[qb][kb1 >> 2], kb1 & 3)};
                const rotf2 s0 = rot_fma2(fx, t1 - t0, t0), s1 = rot_fma2(fx, u1 - u0, u0);      // exact
                const rotf2 v = rot_fma2(fy, s1 - s0, s0);                                       // toward zero: floor(v) is the integer form's result
                W[na >> 2] = rot_put_u8(v.x, na & 3, W[na >> 2]);
                W[nb >> 2] = rot_put_u8(v.y, nb & 3, W[nb >> 2]);
            }
        } else {
#pragma unroll
            for (int n = 0; n < 4 * BPP; n++) W[n >> 2] |= ((TT[n / BPP][0] >> (8 * (n % BPP))) & 0xFFu) << (8 * (n & 3));
        }
        unsigned *d = reinterpret_cast<unsigned *>(dst + (size_t)(jLo + jr) * ds + (size_t)(iLo + ir) * BPP);
        ROT_PH(5);
#pragma unroll
        for (int k = 0; k < BPP; k++) d[k] = W[k];
        ROT_PH(6);
    }
}

// int_sin, vf_rotate.c:198-218: input scaled by 2^20, output by 2^16
static int64_t rot_int_sin(int64_t a)
{
    const int64_t PI = 3294199, F2 = 1 << 20;
    int64_t res = 0;
    if (a < 0) a = PI - a;
    a %= 2 * PI;
    if (a >= PI * 3 / 2) a -= 2 * PI;
    if (a >= PI / 2) a = PI - a;
    const int64_t a2 = (a * a) / F2;
    for (int i = 2; i < 11; i += 2) {
        res += a;
        a = -a * a2 / (F2 * i * (i + 1));
    }
    return (res + 8) >> 4;
}

int launch_rotate(const uint8_t *src, int ss, uint8_t *dst, int ds, int inW, int inH, int outW, int outH, int bpp,
                  double angleRad, int bilinear, const uint8_t *fill, hipStream_t stream, double shiftX, double shiftY,
                  const OpFrames *frames, int nframes)
{
    if (inW <= 0 || inH <= 0 || outW <= 0 || outH <= 0) return 0;
    if (bpp < 1 || bpp > 4) return GMAT_ERR(ENOSYS);
    if (nframes < 1 || nframes > kOpMaxFrames) return GMAT_ERR(EINVAL);
    if (frames) {                                            // alignment tests below see the batch's least aligned frame
        uintptr_t so = 0, dor = 0;
        for (int i = 0; i < nframes; i++) { so |= (uintptr_t)frames->src[i]; dor |= (uintptr_t)frames->dst[i]; }
        src = reinterpret_cast<const uint8_t *>(so); dst = reinterpret_cast<uint8_t *>(dor);     // only their low bits are looked at
    }
    RotateParams p;
    const int FIXP = 1 << 16;
    const int angle_int = (int)(angleRad * FIXP * 16);                    // filter_frame, vf_rotate.c:520-522
    p.s = (int)rot_int_sin(angle_int);
    p.c = (int)rot_int_sin(angle_int + 3294199 / 2);
    const int xi = -(outW - 1) * p.c / 2, yi = (outW - 1) * p.s / 2;      // :538-541 (C division truncates)
    const int xprime = -(outH - 1) * p.s / 2, yprime = -(outH - 1) * p.c / 2;
    // shift_x / shift_y: the rotated image translated by that many output pixels, out(i, j) = rot(i - sx, j - sy) (the test suite's checker (orc_vf.c)).
    // The walk is 16.16 fixed point in 32 bits: frames whose coordinates leave it are refused, a translation that pushes the whole source
    // out of the output is an all-background frame whatever its size (one canonical start that cannot wrap).
    if (!std::isfinite(shiftX) || !std::isfinite(shiftY)) return GMAT_ERR(EINVAL);
    if ((long long)outW + outH + 4 >= 16384 || inW >= 32767 || inH >= 32767) return GMAT_ERR(EINVAL);
    const double lim = 1.0e6;                                              // px; far beyond any frame, keeps llrint and the products in range
    const long long Sx = llrint(std::min(std::max(shiftX, -lim), lim) * 65536.0), Sy = llrint(std::min(std::max(shiftY, -lim), lim) * 65536.0);
    const long long X0 = (long long)xprime + xi + FIXP * (inW - 1) / 2 - ((Sx * p.c + Sy * p.s) >> 16);
    const long long Y0 = (long long)yprime + yi + FIXP * (inH - 1) / 2 - ((Sy * p.c - Sx * p.s) >> 16);
    const long long reach = ((long long)inW + inH + outW + outH + 4) * FIXP;   // |start| beyond this: no output pixel maps into the source
    if (X0 > reach || X0 < -reach || Y0 > reach || Y0 < -reach) { p.X0 = -(outW + outH + 2) * FIXP; p.Y0 = p.X0; }
    else { p.X0 = (int)X0; p.Y0 = (int)Y0; }
    p.inW = inW; p.inH = inH; p.outW = outW; p.outH = outH;
    p.bilinear = bilinear; p.fillEnable = fill != nullptr; p.fill = 0;
    for (int k = 0; k < bpp && fill; k++) p.fill |= (unsigned)fill[k] << (8 * k);
    const int aligned = ((((uintptr_t)dst | (uintptr_t)ds) & 3) == 0);
    // the source patch staged in LDS whenever the source rows are dword-aligned; GMAT_ROTATE_LDS=0 forces the direct form (A/B)
    const char *el = GMAT_KNOB("GMAT_ROTATE_LDS");
    const bool lds = (el ? atoi(el) != 0 : true) && ((((uintptr_t)src | (uintptr_t)ss) & 3) == 0) && ss > 0 && ss < (1 << 23) && (int64_t)ss * inH < (1ll << 31);
    if (lds) {
        const int nbx = (outW + 31) / 32, nby = (outH + 31) / 32;
        // four waves a block; GMAT_ROTATE_WAVES=2 (a thread: 8 pixels in two passes, the block's fixed latencies paid per 512 pixels a
        // wave) is the A/B that showed the kernel is not bound by them: 21.0 against 20.7 us (profiles/r03zr_rotate.txt)
        const char *ew = GMAT_KNOB("GMAT_ROTATE_WAVES");
        const int nwv = ew && atoi(ew) == 2 ? 2 : 4;
        // 64 x 32 macro tiles (rotate_mt_kernel); GMAT_ROTATE_MT=0 keeps the 32 x 32 form (A/B, and the tests' way to it)
        const char *em = GMAT_KNOB("GMAT_ROTATE_MT");
        if (nwv == 4 && (em ? atoi(em) != 0 : true)) {
            const int nbxM = (outW + 63) / 64;
            const dim3 grid(8 * nbxM, (nby + 7) / 8, frames ? nframes : 1), block(256);
            const OpFrames fr = frames ? *frames : op_frames(src, dst, nullptr);
            if (frames) { src = frames->src[0]; dst = frames->dst[0]; }
            RotMT mt;                                        // extremes of the coordinate offsets over 64 x 32 outputs (see rot_tile_general)
            const int xi = 63 * p.c, xj = 31 * p.s, yi = -63 * p.s, yj = 31 * p.c;
            mt.dxLo = std::min(xi, 0) + std::min(xj, 0); mt.dxHi = std::max(xi, 0) + std::max(xj, 0);
            mt.dyLo = std::min(yi, 0) + std::min(yj, 0); mt.dyHi = std::max(yi, 0) + std::max(yj, 0);
#define GMAT_ROTM(B_) do { if (bilinear == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(rotate_mt_kernel<B_, 2>), grid, block, 0, stream, src, ss, dst, ds, p, mt, aligned, nbx, nby, fr); \
                           else if (bilinear) hipLaunchKernelGGL(HIP_KERNEL_NAME(rotate_mt_kernel<B_, 1>), grid, block, 0, stream, src, ss, dst, ds, p, mt, aligned, nbx, nby, fr); \
                           else hipLaunchKernelGGL(HIP_KERNEL_NAME(rotate_mt_kernel<B_, 0>), grid, block, 0, stream, src, ss, dst, ds, p, mt, aligned, nbx, nby, fr); } while (0)
            switch (bpp) {
            case 1:  GMAT_ROTM(1); break;
            case 2:  GMAT_ROTM(2); break;
            case 3:  GMAT_ROTM(3); break;
            default: GMAT_ROTM(4); break;
            }
#undef GMAT_ROTM
            GMAT_HIP_CHECK(hipGetLastError());
            return 0;
        }
        const dim3 grid(8 * nbx, (nby + 7) / 8, frames ? nframes : 1), block(64 * nwv);
        const OpFrames fr = frames ? *frames : op_frames(src, dst, nullptr);
        if (frames) { src = frames->src[0]; dst = frames->dst[0]; }
#define GMAT_ROTL2(B_, I_) do { if (nwv == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(rotate_lds_kernel<B_, I_, 4>), grid, block, 0, stream, src, ss, dst, ds, p, aligned, nbx, nby, fr); \
                                else hipLaunchKernelGGL(HIP_KERNEL_NAME(rotate_lds_kernel<B_, I_, 2>), grid, block, 0, stream, src, ss, dst, ds, p, aligned, nbx, nby, fr); } while (0)
#define GMAT_ROTL(B_) do { if (bilinear == 2) GMAT_ROTL2(B_, 2); else if (bilinear) GMAT_ROTL2(B_, 1); else GMAT_ROTL2(B_, 0); } while (0)
        switch (bpp) {
        case 1:  GMAT_ROTL(1); break;
        case 2:  GMAT_ROTL(2); break;
        case 3:  GMAT_ROTL(3); break;
        default: GMAT_ROTL(4); break;
        }
#undef GMAT_ROTL2
#undef GMAT_ROTL
        GMAT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    constexpr int lx = 16;                                   // lanes across a wave's patch: 64 pixels x 4 rows
    const int tw = 4 * lx, tbh = 4 * (64 / lx);
    const int nbx = (outW + tw - 1) / tw, nby = (outH + tbh - 1) / tbh;
    const dim3 grid(8 * ((nbx * nby + 7) / 8)), block(256);
    for (int fi = 0; fi < (frames ? nframes : 1); fi++) {    // the fallback form takes a batch frame by frame
        if (frames) { src = frames->src[fi]; dst = frames->dst[fi]; }
#define GMAT_ROT(B_) do { if (bilinear == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(rotate_kernel<B_, lx, 2>), grid, block, 0, stream, src, ss, dst, ds, p, aligned, nbx, nby); \
                          else if (bilinear) hipLaunchKernelGGL(HIP_KERNEL_NAME(rotate_kernel<B_, lx, 1>), grid, block, 0, stream, src, ss, dst, ds, p, aligned, nbx, nby); \
                          else hipLaunchKernelGGL(HIP_KERNEL_NAME(rotate_kernel<B_, lx, 0>), grid, block, 0, stream, src, ss, dst, ds, p, aligned, nbx, nby); } while (0)
        switch (bpp) {
        case 1: GMAT_ROT(1); break;
        case 2: GMAT_ROT(2); break;
        case 3: GMAT_ROT(3); break;
        default: GMAT_ROT(4); break;
        }
#undef GMAT_ROT
    }
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

// rotate(90, clockwise) then horizontal flip is the plain transpose out(x, y) = in(y, x); the 3x3
// kernel 1 2 1 / 2 4 2 / 1 2 1 is symmetric and vf_convolution's border rule is the same on both
// axes, so smoothing commutes with the transpose: smooth the source tile, store it transposed.
int launch_rotate_flip_smooth(const uint8_t *src, int ss, uint8_t *dst, int ds, int inW, int inH, int bpp,
                              hipStream_t stream, const OpFrames *frames, int nframes)
{
    if (inW <= 0 || inH <= 0) return 0;
    if (nframes < 1 || nframes > kOpMaxFrames) return GMAT_ERR(EINVAL);
    if (frames) {                                            // alignment tests below see the batch's least aligned frame
        src = frames->src[0]; dst = frames->dst[0];
        for (int i = 1; i < nframes; i++) {
            src = (const uint8_t *)((uintptr_t)src | ((uintptr_t)frames->src[i] & 15)); dst = (uint8_t *)((uintptr_t)dst | ((uintptr_t)frames->dst[i] & 127));
        }
    }
    const OpFrames fr = op_frames(src, dst, frames);
    ConvParams cp;
    const int m[9] = {1, 2, 1, 2, 4, 2, 1, 2, 1};
    for (int i = 0; i < 9; i++) cp.m[i] = m[i];
    cp.rdiv = 1.0f / 16.0f; cp.bias = 0.0f; cp.shift = 4; cp.half = 8;
    if ((bpp == 3 || bpp == 4) && smooth121_ok(src, ss, dst, ds, inW, inH, bpp)) {
        const int td = bpp == 3 ? 60 : 62;
        const int nt = ((inW * bpp / 4 + td - 1) / td) * ((inH + 63) / 64);
        const dim3 g(8 * ((nt + 7) / 8), nframes), b(256);
        const int dst16 = ((((uintptr_t)dst | (uintptr_t)ds) & 15) == 0);
        // A destination whose rows start on 128-byte lines (a pool frame: gframes.cpp aligns them to 256) takes pieces that ARE whole
        // lines — rgb24: tiles of 128 source rows = 384-byte pieces, rgba: 64 rows = 256 bytes — with streaming stores: 14.2 -> 13.2 us
        // alone, 9.5 -> 8.6 us at 16 frames a launch (0.44 -> 0.47, 0.655 -> 0.725; profiles/r03zu_smooth_layout.txt).  A dense 4K frame's
        // transposed pitch (6480 bytes) is not such a destination and keeps the form below, where both lose.
        if (smooth121_line_dst(dst, ds)) {
            if (bpp == 3) {
                const int ntt = ((inW * bpp / 4 + td - 1) / td) * ((inH + 127) / 128);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(smooth121_kernel<3, true, 60, 16, false, 0, 128>), dim3(8 * ((ntt + 7) / 8), nframes), dim3(512), 0, stream,
                                   src, ss, dst, ds, inW, inH, dst16 | 2, 0, fr);
            } else {
                hipLaunchKernelGGL(HIP_KERNEL_NAME(smooth121_kernel<4, true, 62, 16>), g, b, 0, stream, src, ss, dst, ds, inW, inH, dst16 | 2, 0, fr);
            }
            GMAT_HIP_CHECK(hipGetLastError());
            return 0;
        }
        // rgb24: 8 rows per wave (512 threads a tile) measured 2-3 % ahead of 16 (14.2 vs 14.5 us per 4K frame)
        if (bpp == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(smooth121_kernel<3, true, 60, 8>), g, dim3(512), 0, stream, src, ss, dst, ds, inW, inH, dst16, 0, fr);
        else          hipLaunchKernelGGL(HIP_KERNEL_NAME(smooth121_kernel<4, true, 62, 16>), g, b, 0, stream, src, ss, dst, ds, inW, inH, dst16, 0, fr);
        GMAT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (frames && nframes > 1)
        return op_loop(frames, nframes, [&](const uint8_t *s1, uint8_t *d1) { return launch_rotate_flip_smooth(s1, ss, d1, ds, inW, inH, bpp, stream, nullptr, 1); });
    if (frames) { src = frames->src[0]; dst = frames->dst[0]; }
    const int fast = 1;
    const int ntiles = ((inW + 63) / 64) * ((inH + 63) / 64);
    const dim3 grid(8 * ((ntiles + 7) / 8)), block(256);
    const int aligned = al4(src, ss, dst, ds);
    if (bpp == 3)      hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_kernel<3, 64, 64, true>), grid, block, 0, stream, src, ss, dst, ds, inW, inH, cp, aligned, fast);
    else if (bpp == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_kernel<4, 64, 64, true>), grid, block, 0, stream, src, ss, dst, ds, inW, inH, cp, aligned, fast);
    else return GMAT_ERR(ENOSYS);
    GMAT_HIP_CHECK(hipGetLastError());
    return 0;
}

} // namespace gmat
