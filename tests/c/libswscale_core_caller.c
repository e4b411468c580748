/* tests/c/libswscale_core_caller.c — TEST INFRASTRUCTURE (build container only: needs /root/reference).
 *
 * The reference's REAL libswscale core drives the back-end: this file is compiled against the reference's own headers and linked with
 * the reference's libswscale.a + libavutil.a (built out of tree by tools/build_ref_swscale.sh), with integration/swscale_hip_adapter.c
 * and the library under test standing where the reference's libswscale/cuda objects would (the nine symbols utils.c, swscale.c,
 * swscale_unscaled.c and rgb2rgb.c leave undefined).  It calls what an application calls —
 *     sws_getContext(... | SWS_HWACCEL_CUDA)   libswscale/utils.c:2087-2123 -> sws_init_context_cuda :2026-2060
 *     sws_setCudaStream                        libswscale/swscale.c:1249
 *     sws_scale                                libswscale/swscale.c:1204 -> scale_internal :1042-1044 (ff_swscale_cuda) / :1017 (convert_unscaled)
 *     sws_freeContext_cuda                     libswscale/utils.c:2507-2510
 * — on device frames, and compares every byte with the SAME library's CPU path (a second context without the flag, host frames) in the
 * same process.  Exit code 0: identical.
 *   libswscale_core_caller <srcW> <srcH> <srcFmt name> <dstW> <dstH> <dstFmt name> <gpu flags> <cpu flags> [seed]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libavutil/imgutils.h"
#include "libavutil/pixdesc.h"
#include "libswscale/swscale.h"
#include "gmat_hip.h"

static void fill_lcg(uint8_t *p, long n, uint32_t seed)
{
    uint32_t s = seed;
    for (long i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; p[i] = (uint8_t)(s >> 24); }
}

#define CK(x) do { int r_ = (x); if (r_ < 0) { fprintf(stderr, "%s failed: %d\n", #x, r_); return 2; } } while (0)

/* rows of plane i of a w x h frame of this format */
static int plane_rows(const AVPixFmtDescriptor *d, int i, int h)
{
    return (i == 1 || i == 2) ? AV_CEIL_RSHIFT(h, d->log2_chroma_h) : h;
}

int main(int argc, char **argv)
{
    if (argc < 9) { fprintf(stderr, "usage: libswscale_core_caller srcW srcH srcFmt dstW dstH dstFmt gpuFlags cpuFlags [seed]\n"); return 1; }
    const int sw = atoi(argv[1]), sh = atoi(argv[2]), dw = atoi(argv[4]), dh = atoi(argv[5]);
    const enum AVPixelFormat sf = av_get_pix_fmt(argv[3]), df = av_get_pix_fmt(argv[6]);
    const int gflags = (int)strtol(argv[7], NULL, 0), cflags = (int)strtol(argv[8], NULL, 0);
    const uint32_t seed = argc > 9 ? (uint32_t)atoi(argv[9]) : 4242u;
    if (sf == AV_PIX_FMT_NONE || df == AV_PIX_FMT_NONE) { fprintf(stderr, "unknown pixel format\n"); return 1; }
    const AVPixFmtDescriptor *sd = av_pix_fmt_desc_get(sf), *dd = av_pix_fmt_desc_get(df);

    /* host frames, tightly packed rows (av_image_fill_linesizes, as CSwscale.c:25-28 does) */
    uint8_t *hs[4] = {0}, *hd[4] = {0}, *hg[4] = {0};
    int ss[4] = {0}, ds[4] = {0};
    CK(av_image_fill_linesizes(ss, sf, sw));
    CK(av_image_fill_linesizes(ds, df, dw));
    uint8_t *gsrc[4] = {0}, *gdst[4] = {0};
    for (int i = 0; i < 4 && ss[i]; i++) {
        const long n = (long)ss[i] * plane_rows(sd, i, sh);
        hs[i] = malloc(n);
        fill_lcg(hs[i], n, seed + 17u * i);
        if (sd->comp[0].depth > 8 && sd->comp[0].depth < 16)          /* valid input: `depth` significant bits where the format keeps them (P010: high, planar 10-bit: low) */
            for (long k = 0; k + 1 < n; k += 2) {
                const unsigned v = (hs[i][k] | hs[i][k + 1] << 8) >> (16 - sd->comp[0].depth) << sd->comp[0].shift;
                hs[i][k] = (uint8_t)v; hs[i][k + 1] = (uint8_t)(v >> 8);
            }
        CK(gmat_malloc(&gsrc[i], n)); CK(gmat_memcpy_h2d(gsrc[i], hs[i], n));
    }
    for (int i = 0; i < 4 && ds[i]; i++) {
        const long n = (long)ds[i] * plane_rows(dd, i, dh);
        hd[i] = malloc(n); hg[i] = malloc(n);
        memset(hd[i], 0xCD, n);
        CK(gmat_malloc(&gdst[i], n)); CK(gmat_memset(gdst[i], 0xCD, n));
    }

    /* ---- the GPU path: what an application of the reference writes ---- */
    void *stream = NULL;
    CK(gmat_stream_create(&stream));
    struct SwsContext *g = sws_getContext(sw, sh, sf, dw, dh, df, gflags | SWS_HWACCEL_CUDA, NULL, NULL, NULL);
    if (!g) { fprintf(stderr, "sws_getContext(SWS_HWACCEL_CUDA) failed\n"); return 3; }
    sws_setCudaStream(g, stream);
    for (int rep = 0; rep < 2; rep++) {                                      /* a context serves many frames */
        const int r = sws_scale(g, (const uint8_t *const *)gsrc, ss, 0, sh, gdst, ds);
        if (r < 0) { fprintf(stderr, "sws_scale (GPU context) returned %d\n", r); return 3; }
    }
    CK(gmat_stream_sync(stream));
    for (int i = 0; i < 4 && ds[i]; i++) CK(gmat_memcpy_d2h(hg[i], gdst[i], (long)ds[i] * plane_rows(dd, i, dh)));
    sws_freeContext_cuda(g);

    /* ---- the CPU path of the same library ---- */
    struct SwsContext *c = sws_getContext(sw, sh, sf, dw, dh, df, cflags, NULL, NULL, NULL);
    if (!c) { fprintf(stderr, "sws_getContext (CPU) failed\n"); return 3; }
    if (sws_scale(c, (const uint8_t *const *)hs, ss, 0, sh, hd, ds) != dh) { fprintf(stderr, "sws_scale (CPU context) failed\n"); return 3; }
    sws_freeContext(c);

    long bad = 0, total = 0;
    for (int i = 0; i < 4 && ds[i]; i++) {
        const int rows = plane_rows(dd, i, dh);
        /* bytes a row carries (the last chroma pair of an odd-width semi-planar frame included) */
        const int rb = av_image_get_linesize(df, dw, i);
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < rb; x++) {
                total++;
                if (hg[i][(long)y * ds[i] + x] != hd[i][(long)y * ds[i] + x]) {
                    if (bad < 5) fprintf(stderr, "plane %d (%d, %d): gpu %d cpu %d\n", i, x, y, hg[i][(long)y * ds[i] + x], hd[i][(long)y * ds[i] + x]);
                    bad++;
                }
            }
    }
    printf("%s %dx%d -> %s %dx%d: %ld of %ld bytes differ\n", argv[3], sw, sh, argv[6], dw, dh, bad, total);
    return bad ? 4 : 0;
}
