/*
 * libavfilter/vf_hwupload_hip.c — hwupload with copy / compute overlap: the filter_frame of the reference's
 * vf_hwupload_cuda.c:123-150 (av_hwframe_get_buffer + av_hwframe_transfer_data, i.e. one cuMemcpy2DAsync per plane from
 * PAGEABLE memory on the device context's single stream, libavutil/hwcontext_cuda.c:221-279) rebuilt on the C ABI of
 * libgmat_hip.so:
 *   - software frames are first copied into a ring of PINNED staging frames (gmat_host_frame_alloc);
 *   - the DMA runs on the filter's own upload stream (gmat_hwframe_transfer_data), an event per ring slot orders the
 *     device context's compute stream behind it (gmat_stream_wait_event) — so the upload of frame n + 1 overlaps the
 *     kernels of frame n — and tells when the slot may be refilled.
 * The whole upload -> scale -> download chain with its ring exists in the library as gmat_pipeline_* (include/gmat_hip.h
 * section 5) for callers outside libavfilter.
 */
#include <string.h>
#include "libavutil/hwcontext.h"
#include "libavutil/hwcontext_cuda.h"
#include "libavutil/imgutils.h"
#include "libavutil/opt.h"
#include "libavutil/pixdesc.h"
#include "avfilter.h"
#include "formats.h"
#include "internal.h"
#include "video.h"
#include "gmat_hip.h"

#define UP_RING 4

typedef struct HipUploadContext {
    const AVClass *class;
    int device_idx;
    AVBufferRef *hwdevice, *hwframe;
    void *upload_stream, *compute_stream;
    GmatFrame staging[UP_RING];
    void *uploaded[UP_RING];
    int busy[UP_RING];
    int64_t count;
} HipUploadContext;

static av_cold int hipupload_init(AVFilterContext *ctx)
{
    HipUploadContext *s = ctx->priv;
    char buf[64] = { 0 };
    snprintf(buf, sizeof(buf), "%d", s->device_idx);
    return av_hwdevice_ctx_create(&s->hwdevice, AV_HWDEVICE_TYPE_CUDA, buf, NULL, 0);   /* the slot GMAT keeps */
}

static av_cold void hipupload_uninit(AVFilterContext *ctx)
{
    HipUploadContext *s = ctx->priv;
    for (int i = 0; i < UP_RING; i++) {
        if (s->busy[i]) gmat_event_sync(s->uploaded[i]);
        if (s->uploaded[i]) gmat_event_destroy(s->uploaded[i]);
        gmat_host_frame_free(&s->staging[i]);
    }
    if (s->upload_stream) gmat_stream_destroy(s->upload_stream);
    av_buffer_unref(&s->hwframe);
    av_buffer_unref(&s->hwdevice);
}

static int hipupload_query_formats(AVFilterContext *ctx)
{
    static const enum AVPixelFormat in_fmts[] = {
        /* vf_hwupload_cuda.c:59-67's list + the packed RGB the nvcv filters take (defect 10) + the planar 10 / 16-bit and
         * 64-bit RGB formats libgpuscale lists (swscale_cuda.c:34-44): everything integration/hwcontext_hip.c pools */
        AV_PIX_FMT_NV12, AV_PIX_FMT_YUV420P, AV_PIX_FMT_YUV444P, AV_PIX_FMT_P010, AV_PIX_FMT_P016, AV_PIX_FMT_YUV444P16,
        AV_PIX_FMT_YUV420P10, AV_PIX_FMT_YUV420P16,
        AV_PIX_FMT_RGB24, AV_PIX_FMT_BGR24, AV_PIX_FMT_RGBA, AV_PIX_FMT_BGRA, AV_PIX_FMT_0RGB32, AV_PIX_FMT_0BGR32,
        AV_PIX_FMT_RGBA64, AV_PIX_FMT_BGRA64, AV_PIX_FMT_RGBPF32LE, AV_PIX_FMT_NONE,
    };
    static const enum AVPixelFormat out_fmts[] = { AV_PIX_FMT_CUDA, AV_PIX_FMT_NONE };
    int ret = ff_formats_ref(ff_make_format_list((const int *)in_fmts), &ctx->inputs[0]->outcfg.formats);
    if (ret < 0)
        return ret;
    return ff_formats_ref(ff_make_format_list((const int *)out_fmts), &ctx->outputs[0]->incfg.formats);
}

static int hipupload_config_output(AVFilterLink *outlink)
{
    AVFilterContext *ctx = outlink->src;
    AVFilterLink *inlink = ctx->inputs[0];
    HipUploadContext *s = ctx->priv;
    AVHWFramesContext *frames;
    AVCUDADeviceContext *dev;
    int ret;

    av_buffer_unref(&s->hwframe);
    s->hwframe = av_hwframe_ctx_alloc(s->hwdevice);
    if (!s->hwframe)
        return AVERROR(ENOMEM);
    frames = (AVHWFramesContext *)s->hwframe->data;
    frames->format = AV_PIX_FMT_CUDA;
    frames->sw_format = inlink->format;
    frames->width = inlink->w;
    frames->height = inlink->h;
    if ((ret = av_hwframe_ctx_init(s->hwframe)) < 0)
        return ret;
    dev = frames->device_ctx->hwctx;
    s->compute_stream = dev->stream;
    if (!s->upload_stream && gmat_stream_create(&s->upload_stream) < 0)
        return AVERROR_EXTERNAL;
    for (int i = 0; i < UP_RING; i++) {
        gmat_host_frame_free(&s->staging[i]);
        if (gmat_host_frame_alloc(&s->staging[i], inlink->format, inlink->w, inlink->h) < 0)
            return AVERROR(ENOMEM);
        if (!s->uploaded[i] && gmat_event_create(&s->uploaded[i]) < 0)
            return AVERROR_EXTERNAL;
    }
    outlink->hw_frames_ctx = av_buffer_ref(s->hwframe);
    return outlink->hw_frames_ctx ? 0 : AVERROR(ENOMEM);
}

static int hipupload_filter_frame(AVFilterLink *link, AVFrame *in)
{
    AVFilterContext *ctx = link->dst;
    AVFilterLink *outlink = ctx->outputs[0];
    HipUploadContext *s = ctx->priv;
    const int slot = (int)(s->count % UP_RING);
    GmatFrame *host = &s->staging[slot], dev_frame;
    AVFrame *out = NULL;
    int ret;

    if (in->format == outlink->format)
        return ff_filter_frame(outlink, in);
    out = av_frame_alloc();
    if (!out || (ret = av_hwframe_get_buffer(s->hwframe, out, 0)) < 0) {
        ret = out ? ret : AVERROR(ENOMEM);
        goto fail;
    }
    if (s->busy[slot])
        gmat_event_sync(s->uploaded[slot]);                     /* the DMA that last read this pinned frame has finished */
    av_image_copy(host->data, host->linesize, (const uint8_t **)in->data, in->linesize, in->format, in->width, in->height);

    memset(&dev_frame, 0, sizeof(dev_frame));
    for (int i = 0; i < 4; i++) { dev_frame.data[i] = out->data[i]; dev_frame.linesize[i] = out->linesize[i]; }
    dev_frame.width = in->width; dev_frame.height = in->height;
    dev_frame.format = GMAT_PIX_FMT_HIP; dev_frame.sw_format = in->format;
    if (gmat_hwframe_transfer_data(&dev_frame, host, s->upload_stream) < 0 ||
        gmat_event_record(s->uploaded[slot], s->upload_stream) < 0 ||
        gmat_stream_wait_event(s->compute_stream, s->uploaded[slot]) < 0) {
        ret = AVERROR_EXTERNAL;
        goto fail;
    }
    s->busy[slot] = 1;
    s->count++;
    out->width = in->width; out->height = in->height;
    if ((ret = av_frame_copy_props(out, in)) < 0)
        goto fail;
    av_frame_free(&in);
    return ff_filter_frame(outlink, out);
fail:
    av_frame_free(&in);
    av_frame_free(&out);
    return ret;
}

#define OFFSET(x) offsetof(HipUploadContext, x)
#define FLAGS (AV_OPT_FLAG_FILTERING_PARAM | AV_OPT_FLAG_VIDEO_PARAM)
static const AVOption hwupload_hip_options[] = {
    { "device", "Number of the device to use", OFFSET(device_idx), AV_OPT_TYPE_INT, { .i64 = 0 }, 0, INT_MAX, FLAGS },
    { NULL }
};
AVFILTER_DEFINE_CLASS(hwupload_hip);

static const AVFilterPad hipupload_inputs[] = {
    { .name = "default", .type = AVMEDIA_TYPE_VIDEO, .filter_frame = hipupload_filter_frame },
};
static const AVFilterPad hipupload_outputs[] = {
    { .name = "default", .type = AVMEDIA_TYPE_VIDEO, .config_props = hipupload_config_output },
};

const AVFilter ff_vf_hwupload_hip = {
    .name           = "hwupload_hip",
    .description    = NULL_IF_CONFIG_SMALL("Upload a system memory frame to the GPU through a pinned staging ring."),
    .init           = hipupload_init,
    .uninit         = hipupload_uninit,
    .priv_size      = sizeof(HipUploadContext),
    .priv_class     = &hwupload_hip_class,
    FILTER_INPUTS(hipupload_inputs),
    FILTER_OUTPUTS(hipupload_outputs),
    FILTER_QUERY_FUNC(hipupload_query_formats),
    .flags_internal = FF_FILTER_FLAG_HWFRAME_AWARE,
};
