/*
 * libavutil/hwcontext_hip.c — what stands BEHIND the AV_HWDEVICE_TYPE_CUDA / AV_PIX_FMT_CUDA slots on an MI355X: the
 * HWContextType that libavutil/hwcontext.c's table names `ff_hwcontext_type_cuda` (hwcontext.c:36-38 under CONFIG_CUDA),
 * rebuilt on the C ABI of libgmat_hip.so.  The slots themselves stay (edit 3): every filter, the ffmpeg
 * CLI's `-init_hw_device cuda` / `-hwaccel_output_format cuda` and `hwdownload` keep working on the names they know, and
 * AVCUDADeviceContext.stream carries a hipStream_t (CUstream is an opaque pointer: hwcontext_cuda.h:42-46).
 *
 * It replaces libavutil/hwcontext_cuda.c (the reference's file needs the CUDA driver API through ffnvcodec's dynlink loader):
 *   device_create          hwcontext_cuda.c:395-434   device "N" -> gmat_set_device + a stream of the context's own
 *   frames_init / pool     hwcontext_cuda.c:96-172    one gmat_malloc block per frame, rows on 256-byte boundaries
 *   frames_get_buffer      hwcontext_cuda.c:174-201   plane layout incl. the YUV420P special case (chroma pitch = luma pitch / 2, V before U)
 *   transfer_data_to/from  hwcontext_cuda.c:221-279   one 2-D copy per plane on the device context's stream; a download returns with the data
 * and differs from it where the reference is defective or NVIDIA-specific:
 *   - packed RGB and planar float are frame formats too (defect 10: the nvcv filters only take RGB, the hw pool refused it);
 *   - chroma planes of odd-height frames have ceil(h / 2) rows (the reference lays out and copies h / 2);
 *   - the row alignment is this library's 256 bytes, not a texture alignment;
 *   - an upload from pageable memory returns with the source consumed (the pinned ring of vf_hwupload_hip.c is the overlapped path).
 * Exercised by the reference's REAL libavutil + libavfilter in tests/test_libavfilter_core.py (tools/build_ref_avfilter.sh).
 */
#include <stdlib.h>
#include <string.h>

#include "buffer.h"
#include "common.h"
#include "hwcontext.h"
#include "hwcontext_internal.h"
#include "hwcontext_cuda.h"
#include "imgutils.h"
#include "mem.h"
#include "pixdesc.h"
#include "pixfmt.h"

#include "gmat_hip.h"

#define HIP_ROW_ALIGN 256

struct AVCUDADeviceContextInternal {
    int device;                 /* HIP device ordinal, -1: whatever the caller made current (a context filled in by the caller) */
    int own_stream;             /* the stream was created here and goes with the context */
};

typedef struct HipFramesPriv {
    int row_align;
    int chroma_shift_w, chroma_shift_h;
} HipFramesPriv;

static const enum AVPixelFormat hip_sw_formats[] = {
    AV_PIX_FMT_NV12,      AV_PIX_FMT_YUV420P,   AV_PIX_FMT_YUV444P,
    AV_PIX_FMT_P010,      AV_PIX_FMT_P016,      AV_PIX_FMT_YUV444P16, AV_PIX_FMT_YUV420P10, AV_PIX_FMT_YUV420P16,
    AV_PIX_FMT_0RGB32,    AV_PIX_FMT_0BGR32,
    AV_PIX_FMT_RGB24,     AV_PIX_FMT_BGR24,     AV_PIX_FMT_RGBA,      AV_PIX_FMT_BGRA,
    AV_PIX_FMT_RGBA64LE,  AV_PIX_FMT_BGRA64LE,  AV_PIX_FMT_RGBPF32LE,
};

static int hip_sw_format_known(enum AVPixelFormat f)
{
    for (size_t i = 0; i < FF_ARRAY_ELEMS(hip_sw_formats); i++)
        if (hip_sw_formats[i] == f)
            return 1;
    return 0;
}

static int hip_make_current(AVHWDeviceContext *device_ctx)
{
    AVCUDADeviceContext *hwctx = device_ctx->hwctx;
    if (hwctx->internal && hwctx->internal->device >= 0 && gmat_set_device(hwctx->internal->device) < 0)
        return AVERROR_EXTERNAL;
    return 0;
}

/* ---- device ------------------------------------------------------------------------------------------------------ */

static void hip_device_uninit(AVHWDeviceContext *device_ctx)
{
    AVCUDADeviceContext *hwctx = device_ctx->hwctx;
    if (!hwctx->internal)
        return;
    if (hwctx->internal->own_stream && hwctx->stream) {
        hip_make_current(device_ctx);
        gmat_stream_sync(hwctx->stream);
        gmat_stream_destroy(hwctx->stream);
        hwctx->stream = NULL;
    }
    av_freep(&hwctx->internal);
}

/* a context the CALLER filled in (av_hwdevice_ctx_alloc + its own stream + av_hwdevice_ctx_init): nothing to create */
static int hip_device_init(AVHWDeviceContext *device_ctx)
{
    AVCUDADeviceContext *hwctx = device_ctx->hwctx;
    if (!hwctx->internal) {
        hwctx->internal = av_mallocz(sizeof(*hwctx->internal));
        if (!hwctx->internal)
            return AVERROR(ENOMEM);
        hwctx->internal->device = -1;
    }
    return 0;
}

static int hip_device_create(AVHWDeviceContext *device_ctx, const char *device, AVDictionary *opts, int flags)
{
    AVCUDADeviceContext *hwctx = device_ctx->hwctx;
    char *end = NULL;
    long idx = device && *device ? strtol(device, &end, 10) : 0;
    void *stream = NULL;
    int n = gmat_device_count();

    if ((end && *end) || idx < 0) {
        av_log(device_ctx, AV_LOG_ERROR, "Invalid HIP device '%s'\n", device);
        return AVERROR(EINVAL);
    }
    if (n <= 0 || idx >= n) {
        av_log(device_ctx, AV_LOG_ERROR, "HIP device %ld requested, %d visible\n", idx, n < 0 ? 0 : n);
        return AVERROR(ENODEV);
    }
    hwctx->internal = av_mallocz(sizeof(*hwctx->internal));
    if (!hwctx->internal)
        return AVERROR(ENOMEM);
    hwctx->internal->device = (int)idx;
    if (gmat_set_device((int)idx) < 0 || gmat_stream_create(&stream) < 0) {
        av_freep(&hwctx->internal);
        return AVERROR_EXTERNAL;
    }
    hwctx->cuda_ctx = NULL;                                     /* HIP has no context object to push: the device ordinal is the context */
    hwctx->stream = (CUstream)stream;
    hwctx->internal->own_stream = 1;
    return 0;
}

/* ---- frames ------------------------------------------------------------------------------------------------------ */

static int hip_frames_get_constraints(AVHWDeviceContext *device_ctx, const void *hwconfig, AVHWFramesConstraints *constraints)
{
    const size_t n = FF_ARRAY_ELEMS(hip_sw_formats);
    constraints->valid_sw_formats = av_malloc_array(n + 1, sizeof(*constraints->valid_sw_formats));
    constraints->valid_hw_formats = av_malloc_array(2, sizeof(*constraints->valid_hw_formats));
    if (!constraints->valid_sw_formats || !constraints->valid_hw_formats)
        return AVERROR(ENOMEM);
    memcpy(constraints->valid_sw_formats, hip_sw_formats, n * sizeof(hip_sw_formats[0]));
    constraints->valid_sw_formats[n] = AV_PIX_FMT_NONE;
    constraints->valid_hw_formats[0] = AV_PIX_FMT_CUDA;
    constraints->valid_hw_formats[1] = AV_PIX_FMT_NONE;
    return 0;
}

/* pitches and plane sizes of one frame of the pool; returns the block's size */
static int64_t hip_frame_layout(const AVHWFramesContext *ctx, int linesize[4], size_t plane_bytes[4])
{
    const HipFramesPriv *priv = ctx->internal->priv;
    ptrdiff_t ls[4];
    int64_t total = 0;
    int ret = av_image_fill_linesizes(linesize, ctx->sw_format, ctx->width);
    if (ret < 0)
        return ret;
    for (int p = 0; p < 4; p++)
        linesize[p] = FFALIGN(linesize[p], priv->row_align);
    if (ctx->sw_format == AV_PIX_FMT_YUV420P)                  /* the encoder-side convention the reference keeps: chroma pitch = luma pitch / 2 */
        linesize[1] = linesize[2] = linesize[0] / 2;
    for (int p = 0; p < 4; p++)
        ls[p] = linesize[p];
    if ((ret = av_image_fill_plane_sizes(plane_bytes, ctx->sw_format, ctx->height, ls)) < 0)
        return ret;
    for (int p = 0; p < 4; p++)
        total += (int64_t)plane_bytes[p];
    return total;
}

static void hip_block_free(void *opaque, uint8_t *data)
{
    AVHWFramesContext *ctx = opaque;
    hip_make_current(ctx->device_ctx);
    gmat_free(data);
}

static AVBufferRef *hip_block_alloc(void *opaque, size_t size)
{
    AVHWFramesContext *ctx = opaque;
    uint8_t *block = NULL;
    AVBufferRef *ref;

    if (hip_make_current(ctx->device_ctx) < 0 || gmat_malloc(&block, size) < 0 || !block)
        return NULL;
    ref = av_buffer_create(block, size, hip_block_free, ctx, 0);
    if (!ref)
        gmat_free(block);
    return ref;
}

static int hip_frames_init(AVHWFramesContext *ctx)
{
    HipFramesPriv *priv = ctx->internal->priv;
    int linesize[4];
    size_t plane_bytes[4];
    int64_t size;

    if (!hip_sw_format_known(ctx->sw_format)) {
        av_log(ctx, AV_LOG_ERROR, "Pixel format '%s' is not supported\n", av_get_pix_fmt_name(ctx->sw_format));
        return AVERROR(ENOSYS);
    }
    priv->row_align = ctx->sw_format == AV_PIX_FMT_YUV420P ? 2 * HIP_ROW_ALIGN : HIP_ROW_ALIGN;     /* halved chroma pitches stay aligned */
    av_pix_fmt_get_chroma_sub_sample(ctx->sw_format, &priv->chroma_shift_w, &priv->chroma_shift_h);
    if (ctx->pool)
        return 0;                                              /* the caller's own pool of device blocks */
    size = hip_frame_layout(ctx, linesize, plane_bytes);
    if (size <= 0)
        return size < 0 ? (int)size : AVERROR(EINVAL);
    ctx->internal->pool_internal = av_buffer_pool_init2((size_t)size, ctx, hip_block_alloc, NULL);
    return ctx->internal->pool_internal ? 0 : AVERROR(ENOMEM);
}

static int hip_frames_get_buffer(AVHWFramesContext *ctx, AVFrame *frame)
{
    int linesize[4];
    size_t plane_bytes[4];
    uint8_t *at;
    int64_t size = hip_frame_layout(ctx, linesize, plane_bytes);

    if (size <= 0)
        return size < 0 ? (int)size : AVERROR(EINVAL);
    frame->buf[0] = av_buffer_pool_get(ctx->pool);
    if (!frame->buf[0])
        return AVERROR(ENOMEM);
    if ((int64_t)frame->buf[0]->size < size) {
        av_buffer_unref(&frame->buf[0]);
        return AVERROR(EINVAL);
    }
    at = frame->buf[0]->data;
    for (int p = 0; p < 4; p++) {
        frame->data[p] = plane_bytes[p] ? at : NULL;
        frame->linesize[p] = plane_bytes[p] ? linesize[p] : 0;
        at += plane_bytes[p];
    }
    if (ctx->sw_format == AV_PIX_FMT_YUV420P)                  /* V in front of U (hwcontext_cuda.c:188-193) */
        FFSWAP(uint8_t *, frame->data[1], frame->data[2]);
    frame->format = AV_PIX_FMT_CUDA;
    frame->width = ctx->width;
    frame->height = ctx->height;
    return 0;
}

static int hip_transfer_get_formats(AVHWFramesContext *ctx, enum AVHWFrameTransferDirection dir, enum AVPixelFormat **formats)
{
    enum AVPixelFormat *list = av_malloc_array(2, sizeof(*list));
    if (!list)
        return AVERROR(ENOMEM);
    list[0] = ctx->sw_format;
    list[1] = AV_PIX_FMT_NONE;
    *formats = list;
    return 0;
}

static void hip_frame_view(GmatFrame *v, const AVFrame *f, const AVHWFramesContext *ctx, int on_device)
{
    memset(v, 0, sizeof(*v));
    for (int p = 0; p < 4; p++) {
        v->data[p] = f->data[p];
        v->linesize[p] = f->linesize[p];
    }
    v->width = f->width;
    v->height = f->height;
    v->sw_format = ctx->sw_format;
    v->format = on_device ? GMAT_PIX_FMT_HIP : ctx->sw_format;
}

static int hip_transfer_data(AVHWFramesContext *ctx, AVFrame *dst, const AVFrame *src)
{
    AVCUDADeviceContext *hwctx = ctx->device_ctx->hwctx;
    GmatFrame to, from;
    const int dst_dev = dst->hw_frames_ctx != NULL, src_dev = src->hw_frames_ctx != NULL;

    if ((src_dev && ((AVHWFramesContext *)src->hw_frames_ctx->data)->format != AV_PIX_FMT_CUDA) ||
        (dst_dev && ((AVHWFramesContext *)dst->hw_frames_ctx->data)->format != AV_PIX_FMT_CUDA) || (!dst_dev && !src_dev))
        return AVERROR(ENOSYS);
    /* two hardware frames: cuda_transfer_data copies device to device (both sides CU_MEMORYTYPE_DEVICE, hwcontext_cuda.c:239-252) — reached through
     * av_hwframe_transfer_data's transfer_data_from, then transfer_data_to (hwcontext.c:448-467); the two pools hold one sw_format or the copy is refused */
    if (dst_dev && src_dev && ((AVHWFramesContext *)dst->hw_frames_ctx->data)->sw_format != ((AVHWFramesContext *)src->hw_frames_ctx->data)->sw_format)
        return AVERROR(ENOSYS);
    if (hip_make_current(ctx->device_ctx) < 0)
        return AVERROR_EXTERNAL;
    hip_frame_view(&to, dst, ctx, dst_dev);
    hip_frame_view(&from, src, ctx, src_dev);
    to.width = from.width = FFMIN(dst->width, src->width);     /* av_hwframe_transfer_data's temporary may be the pool's (larger) size */
    to.height = from.height = FFMIN(dst->height, src->height);
    if (gmat_hwframe_transfer_data(&to, &from, hwctx->stream) < 0)
        return AVERROR_EXTERNAL;
    /* a download hands host memory back (hwcontext_cuda.c:268-272); an upload's source may be pageable and reused at once */
    if (gmat_stream_sync(hwctx->stream) < 0)
        return AVERROR_EXTERNAL;
    return 0;
}

const HWContextType ff_hwcontext_type_cuda = {
    .type                   = AV_HWDEVICE_TYPE_CUDA,
    .name                   = "CUDA",                           /* the slot's name: `-init_hw_device cuda=...` keeps working */
    .device_hwctx_size      = sizeof(AVCUDADeviceContext),
    .frames_priv_size       = sizeof(HipFramesPriv),
    .device_create          = hip_device_create,
    .device_init            = hip_device_init,
    .device_uninit          = hip_device_uninit,
    .frames_get_constraints = hip_frames_get_constraints,
    .frames_init            = hip_frames_init,
    .frames_get_buffer      = hip_frames_get_buffer,
    .transfer_get_formats   = hip_transfer_get_formats,
    .transfer_data_to       = hip_transfer_data,
    .transfer_data_from     = hip_transfer_data,
    .pix_fmts               = (const enum AVPixelFormat[]){ AV_PIX_FMT_CUDA, AV_PIX_FMT_NONE },
};
