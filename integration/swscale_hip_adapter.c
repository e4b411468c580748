/*
 * libswscale/hip/swscale_hip_adapter.c — the four back-end entry points of the reference's libswscale that take a
 * SwsContext (declared libswscale/swscale_internal.h:704,975,993,1009; defined by libswscale/cuda/swscale_cuda.c in
 * the reference), over the C ABI of libgmat_hip.so.  The five plain-pointer symbols (yuv2rgb_cuda, rgb2yuv_cuda,
 * yuv2yuv_cuda, rgb24tobgr24_cuda, rgb2rgb_init_cuda) are exported by the library under the reference's own names.
 *
 * The back-end's private state lives where the reference already reserves room for it (swscale_internal.h:682-695):
 * `cuda_stream` is the hipStream_t, one pointer-sized cv_* handle slot holds the GmatSwsContext.
 */
#include <errno.h>
#include "libswscale/swscale_internal.h"
#include "gmat_hip.h"

#define GMAT_CTX(c) ((GmatSwsContext *)(c)->cv_resize_handle)

/* c->cspace is an enum AVColorSpace that NOTHING in the core sets (sws_alloc_context zeroes the context: 0 = AVCOL_SPC_RGB): the
 * reference's get_constants (cuda/yuv2rgb_cuda.cu:782-815) sends every value it does not list — 0 included — to BT.601.  The codes it
 * does list coincide with libswscale's SWS_CS_* rows (swscale.h:99-106). */
static int cs_of_avcol(enum AVColorSpace cs)
{
    switch ((int)cs) {
    case 1:  return GMAT_SWS_CS_ITU709;          /* AVCOL_SPC_BT709 */
    case 4:  return 4;                           /* AVCOL_SPC_FCC */
    case 7:  return 7;                           /* AVCOL_SPC_SMPTE240M */
    case 9: case 10: return GMAT_SWS_CS_BT2020;  /* AVCOL_SPC_BT2020_NCL / _CL */
    default: return GMAT_SWS_CS_DEFAULT;         /* BT470BG, SMPTE170M and everything unlisted: BT.601 */
    }
}

/* utils.c:2057 calls this from sws_init_context_cuda once the formats and sizes are in the context */
int ff_sws_init_swscale_cuda(SwsContext *c)
{
    GmatSwsContext *g = gmat_sws_getContext(c->srcW, c->srcH, c->srcFormat, c->dstW, c->dstH, c->dstFormat,
                                            c->flags, c->param);
    if (!g)
        return AVERROR(ENOSYS);
    /* vf_scale sets the chroma positions as AVOptions of the context before it is initialised (vf_scale.c:563-578) */
    if (!isAnyRGB(c->srcFormat) && (c->srcW != c->dstW || c->srcH != c->dstH || c->srcFormat != c->dstFormat))
        gmat_sws_setChromaPos(g, c->src_h_chr_pos, c->src_v_chr_pos, c->dst_h_chr_pos, c->dst_v_chr_pos);
    c->cv_resize_handle = (void *)g;
    ff_yuv2rgb_init_tables_cuda(c);              /* as the reference's init ends (cuda/swscale_cuda.c:268): the core never calls it for a scaling context */
    return 0;
}

/* utils.c:2509, from sws_freeContext_cuda */
int ff_sws_free_swscale_cuda(SwsContext *c)
{
    gmat_sws_freeContext(GMAT_CTX(c));
    c->cv_resize_handle = NULL;
    return 0;
}

/* swscale.c:1043: the whole frame is converted, slices are ignored exactly as swscale_cuda.c does */
int ff_swscale_cuda(SwsContext *c, const uint8_t *src[], int srcStride[], int srcSliceY, int srcSliceH,
                    uint8_t *dst[], int dstStride[], int dstSliceY, int dstSliceH)
{
    (void)srcSliceY; (void)srcSliceH; (void)dstSliceY; (void)dstSliceH;
    gmat_sws_setStream(GMAT_CTX(c), c->cuda_stream);
    return gmat_sws_scale(GMAT_CTX(c), src, srcStride, 0, c->srcH, dst, dstStride);
}

/* swscale_unscaled.c:2053, and sws_setColorspaceDetails: colour constants are per-context kernel arguments here */
void ff_yuv2rgb_init_tables_cuda(SwsContext *c)
{
    if (!GMAT_CTX(c))
        return;
    gmat_sws_setColorspace(GMAT_CTX(c), cs_of_avcol(c->cspace), c->srcRange);
    if (!isAnyRGB(c->srcFormat) && !isAnyRGB(c->dstFormat))
        gmat_sws_setRange(GMAT_CTX(c), c->srcRange, c->dstRange);      /* lum/chrConvertRange, swscale.c:530-556 */
}
