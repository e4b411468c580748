/*
 * libavfilter/vf_gmat_hip.c — the GPU pixel filters of the reference tree (crop_nvcv, flip_nvcv, rotate_nvcv, smooth_nvcv,
 * scale_cuda, format_cuda; libavfilter/vf_*_nvcv.c, vf_scale_cuda.c, vf_format_cuda.c) over libgmat_hip.so, plus
 * transpose.  One glue for all of them: every filter is an AVFilter with the reference's option names and defaults
 * (vf_crop_nvcv.c:80-86, vf_flip_nvcv.c:77-80, vf_rotate_nvcv.c:79-88, vf_smooth_nvcv.c:82-105, vf_scale_cuda.c:586-603,
 * vf_format_cuda.c:69-79), its config_props creates the output AVHWFramesContext on the input's device
 * (doc/FFmpeg_GPU_Filter_Implementation.md:13-23) and its frame callback enqueues ONE library call on the device
 * context's stream.  AVFrame.data[] / linesize[] of an AV_PIX_FMT_CUDA frame are device pointers and byte strides
 * (libavutil/hwcontext_cuda.c:183-193): exactly what the C ABI takes.
 *
 * Every filter but crop_hip implements activate(): with batch=K it collects K frames and processes them with one kernel launch
 * (scale / format: gmat_sws_scale_batch; flip, transpose, rotate by k * 90 degrees, the 3 x 3 smooth and median: gmat_op_batch per
 * plane) — one frame per launch is bounded by the launch boundary on MI355X.
 */
#include <float.h>                      /* DBL_MAX, FLT_MAX in the option tables */
#include <math.h>
#include <string.h>
#include "libavutil/mathematics.h"        /* M_PI */
#include "libavutil/avstring.h"
#include "libavutil/common.h"
#include "libavutil/hwcontext.h"
#include "libavutil/hwcontext_cuda.h"
#include "libavutil/opt.h"
#include "libavutil/pixdesc.h"
#include "avfilter.h"
#include "filters.h"
#include "formats.h"
#include "internal.h"
#include "scale_eval.h"
#include "video.h"
#include "gmat_hip.h"

enum { GH_CROP, GH_FLIP, GH_ROTATE, GH_TRANSPOSE, GH_SMOOTH, GH_SCALE, GH_FORMAT };
#define GH_MAX_BATCH 32

typedef struct GmatHipContext {
    const AVClass *class;
    int kind;
    /* crop */
    int w, h, x, y;
    /* flip */
    int code;
    /* rotate / transpose */
    double angle, shift_x, shift_y;
    int quarter;                /* exact clockwise quarter turns 0..3 when angle is a multiple of 90 and no shift is given, else -1: ONE
                                   predicate for config_props (which swaps w / h for odd quarters) and for the launches */
    char *interp;
    int dir;
    /* smooth */
    int type, kw, kh, border_type;
    double sigma_x, sigma_y;
    /* scale / format */
    char *w_expr, *h_expr;
    int interp_algo, passthrough, force_oar, force_div, batch;
    float param;
    enum AVPixelFormat out_fmt_opt;
    /* state */
    enum AVPixelFormat in_fmt, out_fmt;
    int in_w, in_h, bypass;
    AVBufferRef *frames_ref;
    void *stream;
    GmatSwsContext *sws;
    int cur_cs, cur_full;       /* AVFrame.colorspace / full range the sws context was last set to (-1: its defaults) */
    AVFrame *queue[GH_MAX_BATCH];
    int nqueued;
} GmatHipContext;

/* A frame's own colour description drives the conversion, frame by frame:
 *   format_hip  vf_format_cuda.c:184-217 hands in->colorspace to nv12_to_rgbpf32 / rgbpf32_to_nv12 (SetMatYuv2Rgb / SetMatRgb2Yuv)
 *   scale_hip   vf_scale_cuda.c has no matrix (YUV -> YUV only); where it converts it follows libavfilter's `scale` with its default
 *               in_color_matrix=auto (vf_scale.c:793-824): the matrix of in->colorspace on both sides, the source range of in->color_range,
 *               a limited-range output
 * AVCOL_SPC_* -> the SWS_CS_* row (libswscale/utils.c parse_yuv_type / cuda/yuv2rgb_cuda.cu:782-815 get_constants: unlisted values are BT.601) */
static int gh_sws_cs(enum AVColorSpace cs)
{
    switch ((int)cs) {
    case AVCOL_SPC_BT709:                               return GMAT_SWS_CS_ITU709;
    case AVCOL_SPC_FCC:                                 return 4;
    case AVCOL_SPC_SMPTE240M:                           return 7;
    case AVCOL_SPC_BT2020_NCL: case AVCOL_SPC_BT2020_CL: return GMAT_SWS_CS_BT2020;
    default:                                            return GMAT_SWS_CS_DEFAULT;
    }
}

static int gh_follow_frame_colour(GmatHipContext *s, const AVFrame *in)
{
    const int cs = gh_sws_cs(in->colorspace), full = in->color_range == AVCOL_RANGE_JPEG;
    const AVPixFmtDescriptor *si = av_pix_fmt_desc_get(s->in_fmt), *so = av_pix_fmt_desc_get(s->out_fmt);
    const int src_rgb = !!(si->flags & AV_PIX_FMT_FLAG_RGB), dst_rgb = !!(so->flags & AV_PIX_FMT_FLAG_RGB);
    int ret = 0;

    if (!s->sws || (cs == s->cur_cs && full == s->cur_full))
        return 0;
    if (src_rgb != dst_rgb)                             /* a matrix is involved: YUV -> RGB (with the source's range) or RGB -> YUV */
        ret = gmat_sws_setColorspace(s->sws, cs, src_rgb ? 0 : full);
    else if (!src_rgb && full != (s->cur_full > 0))     /* YUV -> YUV: a full-range source into the limited-range output `scale` makes of it */
        ret = gmat_sws_setRange(s->sws, full, 0);
    if (ret < 0)
        return AVERROR(ENOSYS);
    s->cur_cs = cs; s->cur_full = full;
    return 0;
}

/* ... and the output frame says what its samples are (vf_scale.c:783-786,:831): av_frame_copy_props carried the INPUT's tags over, but a
 * full-range YUV source has just been compressed to limited range and a matrix applied or removed.  Without this a downstream filter sees
 * limited-range samples tagged AVCOL_RANGE_JPEG, and a second scale_hip compresses them again.  The passthrough of an untouched frame
 * (gh_filter_frame's bypass, vf_scale_cuda.c:543-544) converts nothing and keeps its tags: samples and tags agree on both routes. */
static void gh_tag_output(const GmatHipContext *s, const AVFrame *in, AVFrame *out)
{
    if (av_pix_fmt_desc_get(s->out_fmt)->flags & AV_PIX_FMT_FLAG_RGB)
        out->colorspace = AVCOL_SPC_RGB;
    else if (out->colorspace == AVCOL_SPC_RGB)
        out->colorspace = AVCOL_SPC_UNSPECIFIED;
    /* no out_range option here: the destination range is libswscale's default — limited for YUV, and sws_getColorspaceDetails reports an RGB
     * end as full (utils.c:1074-1075), which is what vf_scale.c:831 writes into the frame */
    if (in->color_range != AVCOL_RANGE_UNSPECIFIED)
        out->color_range = (av_pix_fmt_desc_get(s->out_fmt)->flags & AV_PIX_FMT_FLAG_RGB) ? AVCOL_RANGE_JPEG : AVCOL_RANGE_MPEG;
}

static int plane_geometry(enum AVPixelFormat fmt, int plane, int w, int h, int *pw, int *ph, int *bpp)
{
    const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(fmt);
    const int sub = plane ? 1 : 0;
    if (!d)
        return AVERROR(EINVAL);
    if (d->flags & AV_PIX_FMT_FLAG_RGB) {                      /* packed RGB: one plane of whole pixels */
        *pw = w; *ph = h; *bpp = av_get_padded_bits_per_pixel(d) / 8;
        return plane == 0 ? 0 : AVERROR(EINVAL);
    }
    *pw = sub ? AV_CEIL_RSHIFT(w, d->log2_chroma_w) : w;
    *ph = sub ? AV_CEIL_RSHIFT(h, d->log2_chroma_h) : h;
    *bpp = (fmt == AV_PIX_FMT_NV12 && plane == 1) ? 2 : 1;      /* the interleaved chroma plane moves as 2-byte samples */
    return 0;
}

static av_cold int gh_init(AVFilterContext *ctx)
{
    GmatHipContext *s = ctx->priv;
    const char *n = ctx->filter->name;

    s->kind = !strcmp(n, "crop_hip") ? GH_CROP : !strcmp(n, "flip_hip") ? GH_FLIP : !strcmp(n, "rotate_hip") ? GH_ROTATE :
              !strcmp(n, "transpose_hip") ? GH_TRANSPOSE : !strcmp(n, "smooth_hip") ? GH_SMOOTH :
              !strcmp(n, "scale_hip") ? GH_SCALE : GH_FORMAT;
    if (s->kind == GH_CROP && (s->w <= 0 || s->h <= 0)) {
        av_log(ctx, AV_LOG_ERROR, "The width and height of the cropping area cannot be 0\n");
        return AVERROR(EINVAL);
    }
    if (s->kind == GH_ROTATE) {
        /* map_interpolation, vf_rotate_nvcv.c:114-135 */
        if (strcmp(s->interp, "linear") && strcmp(s->interp, "nearest") && strcmp(s->interp, "cubic") && strcmp(s->interp, "area")) {
            av_log(ctx, AV_LOG_ERROR, "Interpolation '%s' not supported (linear, nearest, cubic, area)\n", s->interp);
            return AVERROR(EINVAL);
        }
        {
            const double q = s->angle / 90.0;
            s->quarter = fabs(q - rint(q)) < 1e-9 && s->shift_x == 0 && s->shift_y == 0 ? (((int)lrint(q) % 4) + 4) % 4 : -1;
        }
    }
    /* every window is odd (the median's and the general gaussian's alike): say so at init, not at the first frame */
    if (s->kind == GH_SMOOTH && (!(s->kw & 1) || !(s->kh & 1))) {
        av_log(ctx, AV_LOG_ERROR, "kw and kh must be odd\n");
        return AVERROR(EINVAL);
    }
    if (s->batch < 1 || s->batch > GH_MAX_BATCH)
        s->batch = 1;
    return 0;
}

static av_cold void gh_uninit(AVFilterContext *ctx)
{
    GmatHipContext *s = ctx->priv;
    for (int i = 0; i < s->nqueued; i++)
        av_frame_free(&s->queue[i]);
    s->nqueued = 0;
    if (s->sws)
        gmat_sws_freeContext(s->sws);
    s->sws = NULL;
    av_buffer_unref(&s->frames_ref);
}

static int gh_query_formats(AVFilterContext *ctx)
{
    static const enum AVPixelFormat pix_fmts[] = { AV_PIX_FMT_CUDA, AV_PIX_FMT_NONE };
    return ff_set_common_formats_from_list(ctx, (const int *)pix_fmts);
}

static int gh_config_props(AVFilterLink *outlink)
{
    AVFilterContext *ctx = outlink->src;
    AVFilterLink *inlink = ctx->inputs[0];
    GmatHipContext *s = ctx->priv;
    AVHWFramesContext *in_frames, *out_frames;
    AVCUDADeviceContext *dev;
    int ow, oh, ret;

    if (!inlink->hw_frames_ctx) {
        av_log(ctx, AV_LOG_ERROR, "No hw context provided on input\n");
        return AVERROR(EINVAL);
    }
    in_frames = (AVHWFramesContext *)inlink->hw_frames_ctx->data;
    dev = in_frames->device_ctx->hwctx;
    s->stream = dev->stream;                                   /* a hipStream_t in the AVCUDADeviceContext slot */
    s->in_fmt = in_frames->sw_format;
    s->in_w = inlink->w; s->in_h = inlink->h;
    s->out_fmt = s->in_fmt;
    ow = inlink->w; oh = inlink->h;

    switch (s->kind) {
    case GH_CROP:
        if (s->x == -1) s->x = (inlink->w - s->w) / 2;
        if (s->y == -1) s->y = (inlink->h - s->h) / 2;
        if (s->in_fmt == AV_PIX_FMT_NV12 || s->in_fmt == AV_PIX_FMT_YUV420P) {      /* chroma grid, vf_crop.c:186-187 */
            s->w &= ~1; s->h &= ~1; s->x &= ~1; s->y &= ~1;
        }
        if (s->x < 0 || s->y < 0 || s->w <= 0 || s->h <= 0 || s->w + s->x > inlink->w || s->h + s->y > inlink->h) {
            av_log(ctx, AV_LOG_ERROR, "The cropping area cannot fall out of the image border\n");
            return AVERROR(EINVAL);
        }
        ow = s->w; oh = s->h;
        break;
    case GH_ROTATE: {
        if (s->quarter > 0 && (s->quarter & 1)) { ow = inlink->h; oh = inlink->w; }   /* quarter turns swap (never with a shift) */
        break;
    }
    case GH_TRANSPOSE:
        ow = inlink->h; oh = inlink->w;
        break;
    case GH_SCALE:
        if ((ret = ff_scale_eval_dimensions(s, s->w_expr, s->h_expr, inlink, outlink, &ow, &oh)) < 0)
            return ret;
        ff_scale_adjust_dimensions(inlink, &ow, &oh, s->force_oar, s->force_div);
        if (ow <= 0 || oh <= 0)
            return AVERROR(EINVAL);
        if (s->out_fmt_opt != AV_PIX_FMT_NONE)
            s->out_fmt = s->out_fmt_opt;
        s->bypass = s->passthrough && ow == inlink->w && oh == inlink->h && s->out_fmt == s->in_fmt;
        break;
    case GH_FORMAT:
        if (s->out_fmt_opt == AV_PIX_FMT_NONE)
            return AVERROR(EINVAL);
        s->out_fmt = s->out_fmt_opt;
        break;
    default:
        break;
    }
    outlink->w = ow; outlink->h = oh;

    if (s->kind == GH_SCALE || s->kind == GH_FORMAT) {
        static const int algo[] = { GMAT_SWS_BICUBIC, GMAT_SWS_POINT, GMAT_SWS_BILINEAR, GMAT_SWS_BICUBIC, GMAT_SWS_LANCZOS };
        double param[2] = { GMAT_SWS_PARAM_DEFAULT, GMAT_SWS_PARAM_DEFAULT };
        if (s->kind == GH_SCALE && s->param != 999999.0f)       /* SCALE_CUDA_PARAM_DEFAULT */
            param[0] = s->param;
        if (s->sws)
            gmat_sws_freeContext(s->sws);
        s->sws = gmat_sws_getContext(inlink->w, inlink->h, s->in_fmt, ow, oh, s->out_fmt,
                                     algo[s->kind == GH_SCALE ? s->interp_algo : 0] | GMAT_SWS_HWACCEL, param);
        if (!s->sws) {
            av_log(ctx, AV_LOG_ERROR, "Unsupported conversion: %s -> %s\n", av_get_pix_fmt_name(s->in_fmt), av_get_pix_fmt_name(s->out_fmt));
            return AVERROR(ENOSYS);
        }
        gmat_sws_setStream(s->sws, s->stream);
        s->cur_cs = GMAT_SWS_CS_DEFAULT; s->cur_full = 0;       /* a new context's defaults: BT.601, limited range */
    }

    av_buffer_unref(&s->frames_ref);
    s->frames_ref = av_hwframe_ctx_alloc(in_frames->device_ref);
    if (!s->frames_ref)
        return AVERROR(ENOMEM);
    out_frames = (AVHWFramesContext *)s->frames_ref->data;
    out_frames->format = AV_PIX_FMT_CUDA;
    out_frames->sw_format = s->out_fmt;
    out_frames->width = FFALIGN(ow, 32);
    out_frames->height = FFALIGN(oh, 32);
    if ((ret = av_hwframe_ctx_init(s->frames_ref)) < 0)
        return ret;
    av_buffer_unref(&outlink->hw_frames_ctx);
    outlink->hw_frames_ctx = av_buffer_ref(s->frames_ref);
    return outlink->hw_frames_ctx ? 0 : AVERROR(ENOMEM);
}

static AVFrame *gh_get_output(AVFilterLink *outlink, const AVFrame *in)
{
    GmatHipContext *s = outlink->src->priv;
    AVFrame *out = av_frame_alloc();
    if (!out)
        return NULL;
    if (av_hwframe_get_buffer(s->frames_ref, out, 0) < 0 || av_frame_copy_props(out, in) < 0) {
        av_frame_free(&out);
        return NULL;
    }
    out->width = outlink->w;
    out->height = outlink->h;
    return out;
}

/* one frame through the plane-wise filters */
static int gh_run_planes(GmatHipContext *s, const AVFrame *in, AVFrame *out)
{
    static const int gauss3[9] = { 1, 2, 1, 2, 4, 2, 1, 2, 1 };
    const int general = s->kw != 3 || s->kh != 3 || s->sigma_x > 0 || s->sigma_y > 0 || s->border_type >= 0;
    const int quarter = s->quarter;
    int ret = 0;

    for (int p = 0; p < 3 && in->data[p] && ret >= 0; p++) {
        int pw, ph, bpp;
        if (plane_geometry(s->in_fmt, p, s->in_w, s->in_h, &pw, &ph, &bpp) < 0)
            break;
        const int sub = p ? av_pix_fmt_desc_get(s->in_fmt)->log2_chroma_w : 0;    /* 4:2:0: 1, 4:4:4: 0 */
        switch (s->kind) {
        case GH_CROP:
            ret = gmat_crop(in->data[p], in->linesize[p], out->data[p], out->linesize[p], s->x >> sub, s->y >> sub,
                            (s->w + sub) >> sub, (s->h + sub) >> sub, bpp, s->stream);
            break;
        case GH_FLIP:
            ret = gmat_flip(in->data[p], in->linesize[p], out->data[p], out->linesize[p], pw, ph, bpp, s->code, s->stream);
            break;
        case GH_TRANSPOSE:
            ret = gmat_transpose(in->data[p], in->linesize[p], out->data[p], out->linesize[p], pw, ph, bpp, s->dir, s->stream);
            break;
        case GH_ROTATE:
            if (quarter == 1 || quarter == 3)
                ret = gmat_transpose(in->data[p], in->linesize[p], out->data[p], out->linesize[p], pw, ph, bpp, quarter == 1 ? 1 : 2, s->stream);
            else if (quarter == 2)
                ret = gmat_flip(in->data[p], in->linesize[p], out->data[p], out->linesize[p], pw, ph, bpp, -1, s->stream);
            else if (quarter == 0)
                ret = gmat_crop(in->data[p], in->linesize[p], out->data[p], out->linesize[p], 0, 0, pw, ph, bpp, s->stream);
            else {
                uint8_t fill[4] = { 0, 0, 0, 255 };             /* black: RGB 0,0,0 / limited-range YUV 16,128,128 */
                if (!(av_pix_fmt_desc_get(s->in_fmt)->flags & AV_PIX_FMT_FLAG_RGB)) { fill[0] = p ? 128 : 16; fill[1] = 128; }
                /* interp: 0 nearest, 1 linear (= area, as cv::warpAffine), 2 cubic; the chroma planes of a 4:2:0 frame move by half the shift */
                const int interp = !strcmp(s->interp, "nearest") ? 0 : !strcmp(s->interp, "cubic") ? 2 : 1;
                double tx = 0, ty = 0;                          /* a shift has the reference's meaning: rotation about the origin */
                if (s->shift_x != 0 || s->shift_y != 0)
                    gmat_rotate_shift_translation(s->angle * M_PI / 180.0, s->shift_x / (1 << sub), s->shift_y / (1 << sub), pw, ph, pw, ph, &tx, &ty);
                ret = gmat_rotate2(in->data[p], in->linesize[p], out->data[p], out->linesize[p], pw, ph, pw, ph, bpp,
                                   s->angle * M_PI / 180.0, interp, tx, ty, fill, s->stream);
            }
            break;
        case GH_SMOOTH:
            if (s->type == 2)
                ret = gmat_median(in->data[p], in->linesize[p], out->data[p], out->linesize[p], pw, ph, bpp, s->kw, s->kh, s->stream);
            else if (general)
                ret = gmat_gauss_blur(in->data[p], in->linesize[p], out->data[p], out->linesize[p], pw, ph, bpp, s->kw, s->kh,
                                      s->sigma_x, s->sigma_y, s->border_type < 0 ? 0 : s->border_type, s->stream);
            else
                ret = gmat_smooth3x3(in->data[p], in->linesize[p], out->data[p], out->linesize[p], pw, ph, bpp, gauss3, 1.0f / 16, 0.0f, s->stream);
            break;
        }
    }
    return ret < 0 ? AVERROR_EXTERNAL : 0;
}

static int gh_filter_frame(AVFilterLink *inlink, AVFrame *in)
{
    AVFilterContext *ctx = inlink->dst;
    AVFilterLink *outlink = ctx->outputs[0];
    GmatHipContext *s = ctx->priv;
    AVFrame *out;
    int ret;

    if (s->kind == GH_SCALE && s->bypass)
        return ff_filter_frame(outlink, in);                    /* vf_scale_cuda.c:543-544 */
    out = gh_get_output(outlink, in);
    if (!out) {
        av_frame_free(&in);
        return AVERROR(ENOMEM);
    }
    if (s->kind == GH_SCALE || s->kind == GH_FORMAT) {
        if ((ret = gh_follow_frame_colour(s, in)) >= 0) {
            ret = gmat_sws_scale(s->sws, (const uint8_t *const *)in->data, in->linesize, 0, in->height, out->data, out->linesize);
            ret = ret < 0 ? AVERROR_EXTERNAL : 0;
            gh_tag_output(s, in, out);
        }
    } else {
        ret = gh_run_planes(s, in, out);
    }
    av_frame_free(&in);
    if (ret < 0) {
        av_frame_free(&out);
        return ret;
    }
    return ff_filter_frame(outlink, out);
}

#define GH_OP_ROTATE_ANY 100     /* vf_rotate.c's walk at an arbitrary angle: gmat_rotate2_batch, not one of gmat_op_batch's */

/* the transform of this filter instance as a gmat_op_batch operation (its kernels take a frame table), or -1 */
static int gh_batched_op(const GmatHipContext *s, int *arg)
{
    const int general = s->kw != 3 || s->kh != 3 || s->sigma_x > 0 || s->sigma_y > 0 || s->border_type >= 0;
    const int quarter = s->quarter;
    *arg = 0;
    switch (s->kind) {
    case GH_FLIP:      *arg = s->code; return GMAT_OP_FLIP;
    case GH_TRANSPOSE: *arg = s->dir;  return GMAT_OP_TRANSPOSE;
    case GH_ROTATE:
        if (quarter == 1 || quarter == 3) { *arg = quarter == 1 ? 1 : 2; return GMAT_OP_TRANSPOSE; }
        if (quarter == 2) { *arg = -1; return GMAT_OP_FLIP; }
        return quarter < 0 ? GH_OP_ROTATE_ANY : -1;
    case GH_SMOOTH:
        if (s->type == 2) return s->kw == 3 && s->kh == 3 ? GMAT_OP_MEDIAN3X3 : -1;
        return general ? -1 : GMAT_OP_SMOOTH3X3;
    default: return -1;
    }
}

/* batch > 1: queue frames, process K of them with one launch (a grid dimension = frame): scale_hip / format_hip through
 * gmat_sws_scale_batch, the transform filters plane by plane through gmat_op_batch */
static int gh_flush_queue(AVFilterContext *ctx)
{
    AVFilterLink *outlink = ctx->outputs[0];
    GmatHipContext *s = ctx->priv;
    const uint8_t *sp[4 * GH_MAX_BATCH] = { 0 };
    uint8_t *dp[4 * GH_MAX_BATCH] = { 0 };
    AVFrame *outs[GH_MAX_BATCH] = { 0 };
    void *streams[1] = { s->stream };
    const int n = s->nqueued;
    int ret = 0;

    if (!n)
        return 0;
    for (int i = 0; i < n && ret >= 0; i++) {
        outs[i] = gh_get_output(outlink, s->queue[i]);
        if (!outs[i]) { ret = AVERROR(ENOMEM); break; }
        for (int k = 0; k < 4; k++) { sp[4 * i + k] = s->queue[i]->data[k]; dp[4 * i + k] = outs[i]->data[k]; }
    }
    /* one launch takes ONE stride set: frames of one pool share it, a frame from elsewhere (another pool, a cropped view) may
     * not — those batches go frame by frame instead of being read and written with the first frame's pitch */
    if (ret >= 0) {
        int same = 1;
        for (int i = 1; i < n && same; i++)
            for (int k = 0; k < 4; k++)
                if (s->queue[i]->linesize[k] != s->queue[0]->linesize[k] || outs[i]->linesize[k] != outs[0]->linesize[k])
                    same = 0;
        const int plane_wise = s->kind != GH_SCALE && s->kind != GH_FORMAT;
        int oparg = 0;
        const int op = plane_wise ? gh_batched_op(s, &oparg) : -1;
        for (int i = 1; i < n && same && !plane_wise; i++)      /* ... and ONE colour description (gh_follow_frame_colour) */
            if (s->queue[i]->colorspace != s->queue[0]->colorspace || s->queue[i]->color_range != s->queue[0]->color_range)
                same = 0;
        if (same && !plane_wise) {
            if ((ret = gh_follow_frame_colour(s, s->queue[0])) < 0)
                ;
            else if (gmat_sws_scale_batch(s->sws, n, sp, s->queue[0]->linesize, dp, outs[0]->linesize, streams, 1, 0) < 0)
                ret = AVERROR_EXTERNAL;
        } else if (same && op >= 0) {
            for (int p = 0; p < 3 && s->queue[0]->data[p] && ret >= 0; p++) {
                const uint8_t *ps[GH_MAX_BATCH];
                uint8_t *pd[GH_MAX_BATCH];
                int pw, ph, bpp;
                if (plane_geometry(s->in_fmt, p, s->in_w, s->in_h, &pw, &ph, &bpp) < 0)
                    break;
                for (int i = 0; i < n; i++) { ps[i] = s->queue[i]->data[p]; pd[i] = outs[i]->data[p]; }
                if (op == GH_OP_ROTATE_ANY) {               /* background and shift per plane as in gh_run_planes */
                    const int sub = p ? av_pix_fmt_desc_get(s->in_fmt)->log2_chroma_w : 0;
                    const int interp = !strcmp(s->interp, "nearest") ? 0 : !strcmp(s->interp, "cubic") ? 2 : 1;
                    uint8_t fill[4] = { 0, 0, 0, 255 };
                    if (!(av_pix_fmt_desc_get(s->in_fmt)->flags & AV_PIX_FMT_FLAG_RGB)) { fill[0] = p ? 128 : 16; fill[1] = 128; }
                    double tx = 0, ty = 0;
                    if (s->shift_x != 0 || s->shift_y != 0)
                        gmat_rotate_shift_translation(s->angle * M_PI / 180.0, s->shift_x / (1 << sub), s->shift_y / (1 << sub), pw, ph, pw, ph, &tx, &ty);
                    if (gmat_rotate2_batch(n, ps, s->queue[0]->linesize[p], pd, outs[0]->linesize[p], pw, ph, pw, ph, bpp, s->angle * M_PI / 180.0,
                                           interp, tx, ty, fill, s->stream) < 0)
                        ret = AVERROR_EXTERNAL;
                    continue;
                }
                if (gmat_op_batch(op, n, ps, s->queue[0]->linesize[p], pd, outs[0]->linesize[p], pw, ph, bpp, oparg, s->stream) < 0)
                    ret = AVERROR_EXTERNAL;
            }
        } else if (plane_wise) {                            /* no frame table for this form, or mixed strides: frame by frame */
            for (int i = 0; i < n && ret >= 0; i++)
                ret = gh_run_planes(s, s->queue[i], outs[i]);
        } else {
            for (int i = 0; i < n && ret >= 0; i++)
                if ((ret = gh_follow_frame_colour(s, s->queue[i])) >= 0 &&
                    gmat_sws_scale(s->sws, (const uint8_t *const *)s->queue[i]->data, s->queue[i]->linesize, 0, s->queue[i]->height,
                                   outs[i]->data, outs[i]->linesize) < 0)
                    ret = AVERROR_EXTERNAL;
        }
    }
    if (s->kind == GH_SCALE || s->kind == GH_FORMAT)
        for (int i = 0; i < n; i++)
            if (outs[i])
                gh_tag_output(s, s->queue[i], outs[i]);
    for (int i = 0; i < n; i++)
        av_frame_free(&s->queue[i]);
    s->nqueued = 0;
    for (int i = 0; i < n; i++) {
        if (ret >= 0 && outs[i])
            ret = ff_filter_frame(outlink, outs[i]);
        else
            av_frame_free(&outs[i]);
    }
    return ret;
}

static int gh_activate(AVFilterContext *ctx)
{
    AVFilterLink *inlink = ctx->inputs[0], *outlink = ctx->outputs[0];
    GmatHipContext *s = ctx->priv;
    AVFrame *in = NULL;
    int64_t pts;
    int ret, status;

    FF_FILTER_FORWARD_STATUS_BACK(outlink, inlink);
    while ((ret = ff_inlink_consume_frame(inlink, &in)) > 0) {
        if (s->batch <= 1 || s->bypass) {
            if ((ret = gh_filter_frame(inlink, in)) < 0)
                return ret;
            continue;
        }
        s->queue[s->nqueued++] = in;
        if (s->nqueued >= s->batch && (ret = gh_flush_queue(ctx)) < 0)
            return ret;
    }
    if (ret < 0)
        return ret;
    if (ff_inlink_acknowledge_status(inlink, &status, &pts)) {
        if ((ret = gh_flush_queue(ctx)) < 0)                    /* EOF: the partial batch */
            return ret;
        ff_outlink_set_status(outlink, status, pts);
        return 0;
    }
    FF_FILTER_FORWARD_WANTED(outlink, inlink);
    return FFERROR_NOT_READY;
}

#define OFFSET(x) offsetof(GmatHipContext, x)
#define FLAGS (AV_OPT_FLAG_FILTERING_PARAM | AV_OPT_FLAG_VIDEO_PARAM)

static const AVOption crop_hip_options[] = {
    { "w", "set the width of the cropping area",  OFFSET(w), AV_OPT_TYPE_INT, { .i64 = 0 },  0, INT_MAX, FLAGS },
    { "h", "set the height of the cropping area", OFFSET(h), AV_OPT_TYPE_INT, { .i64 = 0 },  0, INT_MAX, FLAGS },
    { "x", "left edge of the cropping area (-1: centred)", OFFSET(x), AV_OPT_TYPE_INT, { .i64 = -1 }, -1, INT_MAX, FLAGS },
    { "y", "top edge of the cropping area (-1: centred)",  OFFSET(y), AV_OPT_TYPE_INT, { .i64 = -1 }, -1, INT_MAX, FLAGS },
    { NULL }
};
static const AVOption flip_hip_options[] = {
    { "code", "0 vertical, 1 horizontal, -1 both", OFFSET(code), AV_OPT_TYPE_INT, { .i64 = 0 }, -1, 1, FLAGS },
    { "batch", "frames processed by one kernel launch (activate-based queue)", OFFSET(batch), AV_OPT_TYPE_INT, { .i64 = 1 }, 1, GH_MAX_BATCH, FLAGS },
    { NULL }
};
static const AVOption rotate_hip_options[] = {
    { "angle", "rotation angle in degrees", OFFSET(angle), AV_OPT_TYPE_DOUBLE, { .dbl = 0.0 }, -360, 360, FLAGS },
    { "interp", "Interpolation algorithm (linear, nearest, cubic, area)", OFFSET(interp), AV_OPT_TYPE_STRING, { .str = "linear" }, 0, 0, FLAGS },
    { "shift_x", "Shift in x to move the center at the same coord after rotation (rotation about the origin when given)", OFFSET(shift_x), AV_OPT_TYPE_DOUBLE, { .dbl = 0.0 }, -32767, 32767, FLAGS },
    { "shift_y", "Shift in y to move the center at the same coord after rotation (rotation about the origin when given)", OFFSET(shift_y), AV_OPT_TYPE_DOUBLE, { .dbl = 0.0 }, -32767, 32767, FLAGS },
    { "batch", "frames processed by one kernel launch (activate-based queue)", OFFSET(batch), AV_OPT_TYPE_INT, { .i64 = 1 }, 1, GH_MAX_BATCH, FLAGS },
    { NULL }
};
static const AVOption transpose_hip_options[] = {
    { "dir", "0 cclock_flip, 1 clock, 2 cclock, 3 clock_flip", OFFSET(dir), AV_OPT_TYPE_INT, { .i64 = 0 }, 0, 3, FLAGS },
    { "batch", "frames processed by one kernel launch (activate-based queue)", OFFSET(batch), AV_OPT_TYPE_INT, { .i64 = 1 }, 1, GH_MAX_BATCH, FLAGS },
    { NULL }
};
static const AVOption smooth_hip_options[] = {
    { "type", "0 default = 1 gaussian, 2 median", OFFSET(type), AV_OPT_TYPE_INT, { .i64 = 0 }, 0, 2, FLAGS, "type" },
        { "gaussian", "gaussian blur", 0, AV_OPT_TYPE_CONST, { .i64 = 1 }, 0, 0, FLAGS, "type" },
        { "median",   "median blur",   0, AV_OPT_TYPE_CONST, { .i64 = 2 }, 0, 0, FLAGS, "type" },
    { "kw", "kernel width",  OFFSET(kw), AV_OPT_TYPE_INT, { .i64 = 3 }, 1, 255, FLAGS },
    { "kh", "kernel height", OFFSET(kh), AV_OPT_TYPE_INT, { .i64 = 3 }, 1, 255, FLAGS },
    { "border_type", "border rule of the gaussian (-1: the 3x3 integer kernel's own)", OFFSET(border_type), AV_OPT_TYPE_INT, { .i64 = -1 }, -1, 4, FLAGS, "border_type" },
        { "constant",   NULL, 0, AV_OPT_TYPE_CONST, { .i64 = 0 }, 0, 0, FLAGS, "border_type" },
        { "replicate",  NULL, 0, AV_OPT_TYPE_CONST, { .i64 = 1 }, 0, 0, FLAGS, "border_type" },
        { "reflect",    NULL, 0, AV_OPT_TYPE_CONST, { .i64 = 2 }, 0, 0, FLAGS, "border_type" },
        { "warp",       NULL, 0, AV_OPT_TYPE_CONST, { .i64 = 3 }, 0, 0, FLAGS, "border_type" },
        { "reflect101", NULL, 0, AV_OPT_TYPE_CONST, { .i64 = 4 }, 0, 0, FLAGS, "border_type" },
    { "sigmaX", "gaussian sigma in x", OFFSET(sigma_x), AV_OPT_TYPE_DOUBLE, { .dbl = 0 }, 0, DBL_MAX, FLAGS },
    { "sigmaY", "gaussian sigma in y", OFFSET(sigma_y), AV_OPT_TYPE_DOUBLE, { .dbl = 0 }, 0, DBL_MAX, FLAGS },
    { "batch", "frames processed by one kernel launch (activate-based queue)", OFFSET(batch), AV_OPT_TYPE_INT, { .i64 = 1 }, 1, GH_MAX_BATCH, FLAGS },
    { NULL }
};
static const AVOption scale_hip_options[] = {
    { "w", "Output video width",  OFFSET(w_expr), AV_OPT_TYPE_STRING, { .str = "iw" }, .flags = FLAGS },
    { "h", "Output video height", OFFSET(h_expr), AV_OPT_TYPE_STRING, { .str = "ih" }, .flags = FLAGS },
    { "interp_algo", "Interpolation algorithm used for resizing", OFFSET(interp_algo), AV_OPT_TYPE_INT, { .i64 = 0 }, 0, 4, FLAGS, "interp_algo" },
        { "nearest",  NULL, 0, AV_OPT_TYPE_CONST, { .i64 = 1 }, 0, 0, FLAGS, "interp_algo" },
        { "bilinear", NULL, 0, AV_OPT_TYPE_CONST, { .i64 = 2 }, 0, 0, FLAGS, "interp_algo" },
        { "bicubic",  NULL, 0, AV_OPT_TYPE_CONST, { .i64 = 3 }, 0, 0, FLAGS, "interp_algo" },
        { "lanczos",  NULL, 0, AV_OPT_TYPE_CONST, { .i64 = 4 }, 0, 0, FLAGS, "interp_algo" },
    { "format", "Output video pixel format", OFFSET(out_fmt_opt), AV_OPT_TYPE_PIXEL_FMT, { .i64 = AV_PIX_FMT_NONE }, INT_MIN, INT_MAX, .flags = FLAGS },
    { "passthrough", "Do not process frames at all if parameters match", OFFSET(passthrough), AV_OPT_TYPE_BOOL, { .i64 = 1 }, 0, 1, FLAGS },
    { "param", "Algorithm-Specific parameter (libswscale's param0)", OFFSET(param), AV_OPT_TYPE_FLOAT, { .dbl = 999999.0f }, -FLT_MAX, FLT_MAX, FLAGS },
    { "force_original_aspect_ratio", "decrease or increase w/h if necessary to keep the original AR", OFFSET(force_oar), AV_OPT_TYPE_INT, { .i64 = 0 }, 0, 2, FLAGS, "force_oar" },
        { "disable",  NULL, 0, AV_OPT_TYPE_CONST, { .i64 = 0 }, 0, 0, FLAGS, "force_oar" },
        { "decrease", NULL, 0, AV_OPT_TYPE_CONST, { .i64 = 1 }, 0, 0, FLAGS, "force_oar" },
        { "increase", NULL, 0, AV_OPT_TYPE_CONST, { .i64 = 2 }, 0, 0, FLAGS, "force_oar" },
    { "force_divisible_by", "enforce that the output resolution is divisible by an integer when force_original_aspect_ratio is used", OFFSET(force_div), AV_OPT_TYPE_INT, { .i64 = 1 }, 1, 256, FLAGS },
    { "batch", "frames converted by one kernel launch (activate-based queue)", OFFSET(batch), AV_OPT_TYPE_INT, { .i64 = 1 }, 1, GH_MAX_BATCH, FLAGS },
    { NULL }
};
static const AVOption format_hip_options[] = {
    { "pix_fmt", "Output video pixel format", OFFSET(out_fmt_opt), AV_OPT_TYPE_PIXEL_FMT, { .i64 = AV_PIX_FMT_NONE }, INT_MIN, INT_MAX, .flags = FLAGS },
    { "batch", "frames converted by one kernel launch (activate-based queue)", OFFSET(batch), AV_OPT_TYPE_INT, { .i64 = 1 }, 1, GH_MAX_BATCH, FLAGS },
    { NULL }
};

static const AVFilterPad gh_inputs_frame[] = {
    { .name = "default", .type = AVMEDIA_TYPE_VIDEO, .filter_frame = gh_filter_frame },
};
static const AVFilterPad gh_inputs_activate[] = {
    { .name = "default", .type = AVMEDIA_TYPE_VIDEO },
};
static const AVFilterPad gh_outputs[] = {
    { .name = "default", .type = AVMEDIA_TYPE_VIDEO, .config_props = gh_config_props },
};

#define GH_FILTER(name_, desc_, inputs_, activate_)                                                   \
    AVFILTER_DEFINE_CLASS(name_);                                                                     \
    const AVFilter ff_vf_##name_ = {                                                                  \
        .name           = #name_,                                                                     \
        .description    = NULL_IF_CONFIG_SMALL(desc_),                                                \
        .priv_size      = sizeof(GmatHipContext),                                                     \
        .priv_class     = &name_##_class,                                                             \
        .init           = gh_init,                                                                    \
        .uninit         = gh_uninit,                                                                  \
        .activate       = activate_,                                                                  \
        FILTER_INPUTS(inputs_),                                                                       \
        FILTER_OUTPUTS(gh_outputs),                                                                   \
        FILTER_QUERY_FUNC(gh_query_formats),                                                          \
        .flags_internal = FF_FILTER_FLAG_HWFRAME_AWARE,                                               \
    }

GH_FILTER(crop_hip,      "Crop the input video on the GPU (libgmat_hip).",                gh_inputs_frame,    NULL);
GH_FILTER(flip_hip,      "Flip the input video on the GPU (libgmat_hip).",                gh_inputs_activate, gh_activate);
GH_FILTER(rotate_hip,    "Rotate the input video on the GPU (libgmat_hip).",              gh_inputs_activate, gh_activate);
GH_FILTER(transpose_hip, "Transpose the input video on the GPU (libgmat_hip).",           gh_inputs_activate, gh_activate);
GH_FILTER(smooth_hip,    "Smooth the input video on the GPU (libgmat_hip).",              gh_inputs_activate, gh_activate);
GH_FILTER(scale_hip,     "GPU accelerated video resizer / converter (libgmat_hip).",      gh_inputs_activate, gh_activate);
GH_FILTER(format_hip,    "GPU accelerated pixel format converter (libgmat_hip).",         gh_inputs_activate, gh_activate);
