/* tests/c/adapter_caller.c — TEST INFRASTRUCTURE: a C caller of the libswscale back-end adapter (integration/swscale_hip_adapter.c),
 * the way libswscale's core calls its GPU back-end: a SwsContext filled as sws_init_context_cuda does (libswscale/utils.c:2026-2060:
 * sizes, formats, flags, params and chroma positions are in the context before ff_sws_init_swscale_cuda runs), then
 * ff_yuv2rgb_init_tables_cuda (swscale_unscaled.c:2053), ff_swscale_cuda per frame (swscale.c:1043) and ff_sws_free_swscale_cuda
 * (utils.c:2509).  Frames live in device memory; the source content is the suite's LCG, so that the Python side can feed the same
 * bytes to the oracle.  Writes the destination planes to stdout as raw bytes.
 *   adapter_caller <srcW> <srcH> <srcFmt> <dstW> <dstH> <dstFmt> <flags> <seed>      (formats / flags: the ABI's integers) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libswscale/swscale_internal.h"
#include "gmat_hip.h"

/* swscale_internal.h:842 has this as a static inline over the pixel format descriptors */
int isAnyRGB(enum AVPixelFormat f)
{
    return f == GMAT_PIX_FMT_RGB24 || f == GMAT_PIX_FMT_BGR24 || f == GMAT_PIX_FMT_RGBA || f == GMAT_PIX_FMT_BGRA;
}

static void fill_lcg(uint8_t *p, long n, uint32_t seed)
{
    uint32_t s = seed;
    for (long i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; p[i] = (uint8_t)(s >> 24); }
}

/* planes of a tightly packed frame: nv12, yuv420p, packed rgb */
static int plane_layout(int fmt, int w, int h, int rowbytes[4], int rows[4])
{
    memset(rowbytes, 0, 4 * sizeof(int)); memset(rows, 0, 4 * sizeof(int));
    switch (fmt) {
    case GMAT_PIX_FMT_NV12: rowbytes[0] = w; rows[0] = h; rowbytes[1] = 2 * ((w + 1) / 2); rows[1] = (h + 1) / 2; return 2;
    case GMAT_PIX_FMT_YUV420P: rowbytes[0] = w; rows[0] = h; rowbytes[1] = rowbytes[2] = (w + 1) / 2; rows[1] = rows[2] = (h + 1) / 2; return 3;
    case GMAT_PIX_FMT_RGB24: case GMAT_PIX_FMT_BGR24: rowbytes[0] = 3 * w; rows[0] = h; return 1;
    case GMAT_PIX_FMT_RGBA: case GMAT_PIX_FMT_BGRA: rowbytes[0] = 4 * w; rows[0] = h; return 1;
    default: return 0;
    }
}

#define CK(x) do { int r_ = (x); if (r_ < 0) { fprintf(stderr, "%s failed: %d\n", #x, r_); return 2; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 9) { fprintf(stderr, "usage: adapter_caller srcW srcH srcFmt dstW dstH dstFmt flags seed\n"); return 1; }
    SwsContext c;
    memset(&c, 0, sizeof(c));
    c.srcW = atoi(argv[1]); c.srcH = atoi(argv[2]); c.srcFormat = atoi(argv[3]);
    c.dstW = atoi(argv[4]); c.dstH = atoi(argv[5]); c.dstFormat = atoi(argv[6]);
    c.flags = atoi(argv[7]) | GMAT_SWS_HWACCEL;                              /* SWS_HWACCEL_CUDA, swscale.h:95 */
    c.param[0] = c.param[1] = GMAT_SWS_PARAM_DEFAULT;                        /* SWS_PARAM_DEFAULT, utils.c:1292 */
    c.src_h_chr_pos = c.src_v_chr_pos = c.dst_h_chr_pos = c.dst_v_chr_pos = -513;   /* options.c:67-70 defaults */
    c.cspace = 0; c.srcRange = 0; c.dstRange = 0;                            /* cspace: NOTHING in the core sets it (0 after sws_alloc_context;
                                                                                tests/test_libswscale_core.py, the real core); limited range */
    const uint32_t seed = (uint32_t)atoi(argv[8]);

    int sb[4], sr[4], db[4], dr[4];
    const int nsp = plane_layout(c.srcFormat, c.srcW, c.srcH, sb, sr), ndp = plane_layout(c.dstFormat, c.dstW, c.dstH, db, dr);
    if (!nsp || !ndp) { fprintf(stderr, "format not handled by this caller\n"); return 1; }
    uint8_t *src[4] = {0}, *dst[4] = {0};
    int ss[4] = {0}, ds[4] = {0};
    for (int i = 0; i < nsp; i++) {
        const long n = (long)sb[i] * sr[i];
        uint8_t *h = malloc(n);
        fill_lcg(h, n, seed + 17u * i);
        CK(gmat_malloc(&src[i], n)); CK(gmat_memcpy_h2d(src[i], h, n));
        free(h);
        ss[i] = sb[i];
    }
    for (int i = 0; i < ndp; i++) { CK(gmat_malloc(&dst[i], (long)db[i] * dr[i])); CK(gmat_memset(dst[i], 0xCD, (long)db[i] * dr[i])); ds[i] = db[i]; }
    void *stream = NULL;
    CK(gmat_stream_create(&stream));
    c.cuda_stream = stream;                                                  /* sws_setCudaStream, swscale.h:446-448 */

    CK(ff_sws_init_swscale_cuda(&c));
    ff_yuv2rgb_init_tables_cuda(&c);
    for (int rep = 0; rep < 2; rep++)                                        /* a context serves many frames */
        CK(ff_swscale_cuda(&c, (const uint8_t **)src, ss, 0, c.srcH, dst, ds, 0, c.dstH));
    CK(gmat_stream_sync(stream));
    for (int i = 0; i < ndp; i++) {
        const long n = (long)db[i] * dr[i];
        uint8_t *h = malloc(n);
        CK(gmat_memcpy_d2h(h, dst[i], n));
        fwrite(h, 1, n, stdout);
        free(h);
    }
    CK(ff_sws_free_swscale_cuda(&c));
    if (c.cv_resize_handle) { fprintf(stderr, "the back-end left its handle in the context\n"); return 3; }
    gmat_stream_destroy(stream);
    for (int i = 0; i < 4; i++) { if (src[i]) gmat_free(src[i]); if (dst[i]) gmat_free(dst[i]); }
    return 0;
}
