#pragma once
#include "libavutil/common.h"
/* only the fields of SwsContext the adapter reads (swscale_internal.h:300-700 in the reference) */
typedef struct SwsContext {
    int srcW, srcH, dstW, dstH, flags;
    enum AVPixelFormat srcFormat, dstFormat;
    double param[2];
    int src_h_chr_pos, src_v_chr_pos, dst_h_chr_pos, dst_v_chr_pos;
    int srcRange, dstRange;
    enum AVColorSpace cspace;
    void *cv_resize_handle;
    void *cuda_stream;
} SwsContext;
int isAnyRGB(enum AVPixelFormat pix_fmt);
int ff_sws_init_swscale_cuda(SwsContext *c);
int ff_sws_free_swscale_cuda(SwsContext *c);
int ff_swscale_cuda(SwsContext *c, const uint8_t *src[], int srcStride[], int srcSliceY, int srcSliceH, uint8_t *dst[], int dstStride[], int dstSliceY, int dstSliceH);
void ff_yuv2rgb_init_tables_cuda(SwsContext *c);
