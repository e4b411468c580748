/* tests/c/filter_caller.c — TEST INFRASTRUCTURE: a miniature libavfilter / libavutil around integration/vf_gmat_hip.c, so that the
 * AVFilter glue RUNS (rounds 1-3 only type-checked it): the handful of libav* functions the glue calls are implemented here over the
 * hand-written declarations of integration/compat and the library's device memory, then one filter instance is driven the way
 * libavfilter drives it —
 *     AVOption defaults + "key=value:key=value"  ->  .init  ->  the output pad's .config_props  ->  frames through the input pad's
 *     .filter_frame or through .activate (ff_inlink_consume_frame ... ff_filter_frame, EOF via ff_inlink_acknowledge_status)  ->  .uninit
 * (libavfilter/avfilter.c:init_dict / avfilter_config_links / ff_filter_activate; conventions: doc/FFmpeg_GPU_Filter_Implementation.md:13-23).
 * Frames are device frames of a miniature AVHWFramesContext (linesize aligned to 256, NV12 chroma behind the aligned luma plane:
 * libavutil/hwcontext_cuda.c:145-193).  The source content is the suite's LCG, so that the Python side can feed the same bytes to the
 * oracle.  Output on stdout: per frame  int32 w, h, sw_format, pts_low  then the planes, rows tightly packed.
 *   filter_caller <filter> <options | -> <sw format name> <w> <h> <nframes> <seed> */
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libavutil/common.h"
#include "libavutil/opt.h"
#include "libavutil/pixdesc.h"
#include "libavutil/hwcontext.h"
#include "libavutil/hwcontext_cuda.h"
#include "libavfilter/avfilter.h"
#include "libavfilter/filters.h"
#include "libavfilter/internal.h"
#include "libavfilter/formats.h"
#include "libavfilter/scale_eval.h"
#include "gmat_hip.h"

/* ---- libavutil: logging, pixel formats ------------------------------------------------------------------------------------------ */
void av_log(void *avcl, int level, const char *fmt, ...)
{
    va_list ap;
    (void)avcl;
    if (level > AV_LOG_ERROR) return;
    va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap);
}
const char *av_default_item_name(void *ctx) { (void)ctx; return "gmat_hip"; }

static const struct { enum AVPixelFormat f; AVPixFmtDescriptor d; int bits; } g_fmts[] = {
    { AV_PIX_FMT_YUV420P, { "yuv420p", 3, 1, 1, 0 }, 12 }, { AV_PIX_FMT_NV12, { "nv12", 3, 1, 1, 0 }, 12 },
    { AV_PIX_FMT_YUV444P, { "yuv444p", 3, 0, 0, 0 }, 24 },
    { AV_PIX_FMT_RGB24, { "rgb24", 3, 0, 0, AV_PIX_FMT_FLAG_RGB }, 24 }, { AV_PIX_FMT_BGR24, { "bgr24", 3, 0, 0, AV_PIX_FMT_FLAG_RGB }, 24 },
    { AV_PIX_FMT_RGBA, { "rgba", 4, 0, 0, AV_PIX_FMT_FLAG_RGB }, 32 }, { AV_PIX_FMT_BGRA, { "bgra", 4, 0, 0, AV_PIX_FMT_FLAG_RGB }, 32 },
};
const AVPixFmtDescriptor *av_pix_fmt_desc_get(enum AVPixelFormat f)
{
    for (size_t i = 0; i < sizeof(g_fmts) / sizeof(g_fmts[0]); i++) if (g_fmts[i].f == f) return &g_fmts[i].d;
    return NULL;
}
int av_get_padded_bits_per_pixel(const AVPixFmtDescriptor *d)
{
    for (size_t i = 0; i < sizeof(g_fmts) / sizeof(g_fmts[0]); i++) if (&g_fmts[i].d == d) return g_fmts[i].bits;
    return 0;
}
const char *av_get_pix_fmt_name(enum AVPixelFormat f) { const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(f); return d ? d->name : "?"; }
static enum AVPixelFormat fmt_by_name(const char *n)
{
    for (size_t i = 0; i < sizeof(g_fmts) / sizeof(g_fmts[0]); i++) if (!strcmp(g_fmts[i].d.name, n)) return g_fmts[i].f;
    return AV_PIX_FMT_NONE;
}

/* ---- buffers and frames -------------------------------------------------------------------------------------------------------- */
AVBufferRef *av_buffer_ref(const AVBufferRef *b) { AVBufferRef *r = malloc(sizeof(*r)); *r = *b; return r; }     /* (no counting: a short-lived test process) */
void av_buffer_unref(AVBufferRef **b) { if (b && *b) { free(*b); *b = NULL; } }
AVFrame *av_frame_alloc(void) { return calloc(1, sizeof(AVFrame)); }
void av_frame_free(AVFrame **f)
{
    if (!f || !*f) return;
    if ((*f)->data[0]) gmat_free((*f)->data[0]);
    av_buffer_unref(&(*f)->hw_frames_ctx);
    free(*f); *f = NULL;
}
int av_frame_copy_props(AVFrame *dst, const AVFrame *src) { dst->pts = src->pts; dst->colorspace = src->colorspace; dst->color_range = src->color_range; return 0; }

AVBufferRef *av_hwframe_ctx_alloc(AVBufferRef *device_ref)
{
    AVHWFramesContext *fc = calloc(1, sizeof(*fc));
    AVBufferRef *r = calloc(1, sizeof(*r));
    fc->device_ref = device_ref; fc->device_ctx = (AVHWDeviceContext *)device_ref->data;
    r->data = (uint8_t *)fc; r->size = sizeof(*fc);
    return r;
}
int av_hwframe_ctx_init(AVBufferRef *ref) { AVHWFramesContext *fc = (AVHWFramesContext *)ref->data; return fc->width > 0 && fc->height > 0 ? 0 : AVERROR(EINVAL); }
/* cuda_get_buffer, hwcontext_cuda.c:160-193: one allocation, planes laid out at the CONTEXT's (aligned) size, rows aligned to 256 */
static void plane_rows_bytes(enum AVPixelFormat f, int w, int h, int rb[4], int rows[4])
{
    memset(rb, 0, 4 * sizeof(int)); memset(rows, 0, 4 * sizeof(int));
    const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(f);
    if (d->flags & AV_PIX_FMT_FLAG_RGB) { rb[0] = w * av_get_padded_bits_per_pixel(d) / 8; rows[0] = h; return; }
    const int cw = AV_CEIL_RSHIFT(w, d->log2_chroma_w), ch = AV_CEIL_RSHIFT(h, d->log2_chroma_h);
    rb[0] = w; rows[0] = h;
    if (f == AV_PIX_FMT_NV12) { rb[1] = 2 * cw; rows[1] = ch; }
    else { rb[1] = rb[2] = cw; rows[1] = rows[2] = ch; }
}
int av_hwframe_get_buffer(AVBufferRef *ref, AVFrame *frame, int flags)
{
    AVHWFramesContext *fc = (AVHWFramesContext *)ref->data;
    int rb[4], rows[4];
    size_t off[4] = {0}, total = 0;
    (void)flags;
    plane_rows_bytes(fc->sw_format, fc->width, fc->height, rb, rows);
    for (int i = 0; i < 4 && rb[i]; i++) {
        frame->linesize[i] = FFALIGN(rb[i], 256);
        if (fc->sw_format == AV_PIX_FMT_YUV420P && i) frame->linesize[i] = frame->linesize[0] / 2;      /* hwcontext_cuda.c:188-193 */
        off[i] = total; total += (size_t)frame->linesize[i] * rows[i];
    }
    uint8_t *base = NULL;
    if (gmat_malloc(&base, total) < 0) return AVERROR(ENOMEM);
    gmat_memset(base, 0xCD, total);
    for (int i = 0; i < 4 && rb[i]; i++) frame->data[i] = base + off[i];
    frame->format = AV_PIX_FMT_CUDA; frame->width = fc->width; frame->height = fc->height;
    frame->hw_frames_ctx = av_buffer_ref(ref);
    return 0;
}

/* ---- libavfilter: links, the activate() helpers, format negotiation, size expressions -------------------------------------------- */
#define QMAX 64
static AVFrame *g_in[QMAX], *g_out[QMAX];
static int g_nin, g_in_pos, g_nout, g_eof, g_eof_taken, g_out_status;
int ff_filter_frame(AVFilterLink *link, AVFrame *frame) { (void)link; if (g_nout >= QMAX) return AVERROR(ENOMEM); g_out[g_nout++] = frame; return 0; }
int ff_inlink_consume_frame(AVFilterLink *link, AVFrame **rframe) { (void)link; if (g_in_pos >= g_nin) return 0; *rframe = g_in[g_in_pos++]; return 1; }
int ff_inlink_acknowledge_status(AVFilterLink *link, int *rstatus, int64_t *rpts)
{
    (void)link;
    if (!g_eof || g_eof_taken || g_in_pos < g_nin) return 0;
    g_eof_taken = 1; *rstatus = FFERRTAG('E', 'O', 'F', ' '); *rpts = 0;
    return 1;
}
void ff_outlink_set_status(AVFilterLink *link, int status, int64_t pts) { (void)link; (void)pts; g_out_status = status; }
int ff_outlink_get_status(AVFilterLink *link) { (void)link; return 0; }
int ff_outlink_frame_wanted(AVFilterLink *link) { (void)link; return 1; }
void ff_inlink_set_status(AVFilterLink *link, int status) { (void)link; (void)status; }
void ff_inlink_request_frame(AVFilterLink *link) { (void)link; }
int ff_set_common_formats_from_list(AVFilterContext *ctx, const int *fmts) { (void)ctx; return fmts[0] == AV_PIX_FMT_CUDA ? 0 : AVERROR(EINVAL); }
static int eval_dim(const char *e, int iw, int ih)
{
    if (!strcmp(e, "iw")) return iw;
    if (!strcmp(e, "ih")) return ih;
    if (!strncmp(e, "iw/", 3)) return iw / atoi(e + 3);
    if (!strncmp(e, "ih/", 3)) return ih / atoi(e + 3);
    if (!strncmp(e, "iw*", 3)) return iw * atoi(e + 3);
    if (!strncmp(e, "ih*", 3)) return ih * atoi(e + 3);
    return atoi(e);
}
int ff_scale_eval_dimensions(void *ctx, const char *w_expr, const char *h_expr, AVFilterLink *inlink, AVFilterLink *outlink, int *ret_w, int *ret_h)
{
    (void)ctx; (void)outlink;
    *ret_w = eval_dim(w_expr, inlink->w, inlink->h); *ret_h = eval_dim(h_expr, inlink->w, inlink->h);
    return 0;
}
int ff_scale_adjust_dimensions(AVFilterLink *inlink, int *ret_w, int *ret_h, int oar, int div) { (void)inlink; (void)ret_w; (void)ret_h; (void)oar; (void)div; return 0; }

/* ---- AVOptions: defaults and "key=value:key=value" (libavutil/opt.c av_opt_set_defaults / av_set_options_string) ------------------ */
static void opt_store(void *priv, const AVOption *o, double num, const char *str)
{
    uint8_t *p = (uint8_t *)priv + o->offset;
    switch (o->type) {
    case AV_OPT_TYPE_INT: case AV_OPT_TYPE_BOOL: case AV_OPT_TYPE_FLAGS: case AV_OPT_TYPE_PIXEL_FMT: *(int *)p = (int)num; break;
    case AV_OPT_TYPE_INT64: *(int64_t *)p = (int64_t)num; break;
    case AV_OPT_TYPE_DOUBLE: *(double *)p = num; break;
    case AV_OPT_TYPE_FLOAT: *(float *)p = (float)num; break;
    case AV_OPT_TYPE_STRING: free(*(char **)p); *(char **)p = str ? strdup(str) : NULL; break;
    default: break;
    }
}
static void opt_defaults(void *priv, const AVClass *c)
{
    for (const AVOption *o = c->option; o && o->name; o++) {
        if (o->type == AV_OPT_TYPE_CONST) continue;
        if (o->type == AV_OPT_TYPE_STRING) opt_store(priv, o, 0, o->default_val.str);
        else if (o->type == AV_OPT_TYPE_DOUBLE || o->type == AV_OPT_TYPE_FLOAT) opt_store(priv, o, o->default_val.dbl, NULL);
        else opt_store(priv, o, (double)o->default_val.i64, NULL);
    }
}
static int opt_set(void *priv, const AVClass *c, const char *key, const char *val)
{
    for (const AVOption *o = c->option; o && o->name; o++) {
        if (o->type == AV_OPT_TYPE_CONST || strcmp(o->name, key)) continue;
        if (o->type == AV_OPT_TYPE_STRING) { opt_store(priv, o, 0, val); return 0; }
        if (o->type == AV_OPT_TYPE_PIXEL_FMT) { const enum AVPixelFormat f = fmt_by_name(val); if (f == AV_PIX_FMT_NONE) return AVERROR(EINVAL); opt_store(priv, o, f, NULL); return 0; }
        char *end = NULL;
        double num = strtod(val, &end);
        if (end == val || *end) {                           /* a named constant of the option's unit */
            const AVOption *k = c->option;
            for (; k && k->name; k++) if (k->type == AV_OPT_TYPE_CONST && k->unit && o->unit && !strcmp(k->unit, o->unit) && !strcmp(k->name, val)) break;
            if (!k || !k->name) return AVERROR(EINVAL);
            num = (double)k->default_val.i64;
        }
        if (num < o->min || num > o->max) { fprintf(stderr, "Value %f for parameter '%s' out of range [%g - %g]\n", num, key, o->min, o->max); return AVERROR(ERANGE); }
        opt_store(priv, o, num, NULL);
        return 0;
    }
    fprintf(stderr, "Option '%s' not found\n", key);
    return AVERROR(EINVAL);
}

/* ---- the driver ---------------------------------------------------------------------------------------------------------------- */
extern const AVFilter ff_vf_crop_hip, ff_vf_flip_hip, ff_vf_rotate_hip, ff_vf_transpose_hip, ff_vf_smooth_hip, ff_vf_scale_hip, ff_vf_format_hip;
static void fill_lcg(uint8_t *p, long n, uint32_t seed) { uint32_t s = seed; for (long i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; p[i] = (uint8_t)(s >> 24); } }
#define CK(x) do { int r_ = (x); if (r_ < 0) { fprintf(stderr, "%s failed: %d\n", #x, r_); return 2; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 8) { fprintf(stderr, "usage: filter_caller filter options|- swfmt w h nframes seed\n"); return 1; }
    const AVFilter *all[] = { &ff_vf_crop_hip, &ff_vf_flip_hip, &ff_vf_rotate_hip, &ff_vf_transpose_hip, &ff_vf_smooth_hip, &ff_vf_scale_hip, &ff_vf_format_hip };
    const AVFilter *flt = NULL;
    for (size_t i = 0; i < sizeof(all) / sizeof(all[0]); i++) if (!strcmp(all[i]->name, argv[1])) flt = all[i];
    if (!flt) { fprintf(stderr, "no such filter\n"); return 1; }
    const enum AVPixelFormat sf = fmt_by_name(argv[3]);
    const int w = atoi(argv[4]), h = atoi(argv[5]), n = atoi(argv[6]);
    const uint32_t seed = (uint32_t)atoi(argv[7]);
    if (sf == AV_PIX_FMT_NONE || n < 1 || n > QMAX) return 1;

    /* the device and the input link's frames context (what hwupload_cuda hands a filter) */
    void *stream = NULL;
    CK(gmat_stream_create(&stream));
    AVCUDADeviceContext cu = { NULL, stream, NULL };
    AVHWDeviceContext dev = { NULL, AV_HWDEVICE_TYPE_CUDA, &cu };
    AVBufferRef devref = { (uint8_t *)&dev, sizeof(dev) };
    AVBufferRef *inref = av_hwframe_ctx_alloc(&devref);
    AVHWFramesContext *infc = (AVHWFramesContext *)inref->data;
    infc->format = AV_PIX_FMT_CUDA; infc->sw_format = sf; infc->width = w; infc->height = h;

    AVFilterContext ctx;
    AVFilterLink inlink, outlink, *ins[1] = { &inlink }, *outs[1] = { &outlink };
    memset(&ctx, 0, sizeof(ctx)); memset(&inlink, 0, sizeof(inlink)); memset(&outlink, 0, sizeof(outlink));
    ctx.filter = flt; ctx.av_class = flt->priv_class; ctx.inputs = ins; ctx.outputs = outs;
    ctx.priv = calloc(1, flt->priv_size);
    *(const AVClass **)ctx.priv = flt->priv_class;
    inlink.dst = &ctx; outlink.src = &ctx;
    inlink.w = w; inlink.h = h; inlink.format = AV_PIX_FMT_CUDA; inlink.hw_frames_ctx = inref;
    outlink.format = AV_PIX_FMT_CUDA;

    opt_defaults(ctx.priv, flt->priv_class);
    if (strcmp(argv[2], "-")) {
        char *opts = strdup(argv[2]);
        for (char *kv = strtok(opts, ":"); kv; kv = strtok(NULL, ":")) {
            char *eq = strchr(kv, '=');
            if (!eq) return 1;
            *eq = 0;
            CK(opt_set(ctx.priv, flt->priv_class, kv, eq + 1));
        }
        free(opts);
    }
    CK(flt->formats.query_func(&ctx));
    CK(flt->init(&ctx));
    CK(flt->outputs[0].config_props(&outlink));

    /* input frames */
    int rb[4], rows[4];
    plane_rows_bytes(sf, w, h, rb, rows);
    for (int f = 0; f < n; f++) {
        AVFrame *fr = av_frame_alloc();
        CK(av_hwframe_get_buffer(inref, fr, 0));
        fr->pts = 1000 + f;
        for (int i = 0; i < 4 && rb[i]; i++) {
            uint8_t *host = malloc((size_t)fr->linesize[i] * rows[i]);
            memset(host, 0xCD, (size_t)fr->linesize[i] * rows[i]);
            uint8_t *tight = malloc((size_t)rb[i] * rows[i]);
            fill_lcg(tight, (long)rb[i] * rows[i], seed + 17u * i + 1000u * f);
            for (int y = 0; y < rows[i]; y++) memcpy(host + (size_t)y * fr->linesize[i], tight + (size_t)y * rb[i], rb[i]);
            CK(gmat_memcpy_h2d(fr->data[i], host, (size_t)fr->linesize[i] * rows[i]));
            free(host); free(tight);
        }
        g_in[g_nin++] = fr;
    }
    /* the frames through the filter: the pad's filter_frame, or activate() until it has nothing left to do */
    if (flt->inputs[0].filter_frame) {
        for (int f = 0; f < n; f++) CK(flt->inputs[0].filter_frame(&inlink, g_in[f]));
    } else {
        g_eof = 1;
        for (int it = 0; it < 4 * n + 8 && !g_out_status; it++) {
            const int r = flt->activate(&ctx);
            if (r < 0 && r != FFERROR_NOT_READY) { fprintf(stderr, "activate failed: %d\n", r); return 2; }
        }
    }
    CK(gmat_stream_sync(stream));
    if (g_nout != n) { fprintf(stderr, "%d frames in, %d out\n", n, g_nout); return 3; }
    for (int f = 0; f < g_nout; f++) {
        AVFrame *o = g_out[f];
        AVHWFramesContext *ofc = (AVHWFramesContext *)o->hw_frames_ctx->data;
        const int32_t head[4] = { o->width, o->height, (int32_t)ofc->sw_format, (int32_t)o->pts };
        fwrite(head, sizeof(head), 1, stdout);
        int orb[4], orows[4];
        plane_rows_bytes(ofc->sw_format, o->width, o->height, orb, orows);
        for (int i = 0; i < 4 && orb[i]; i++) {
            int frb[4], frows[4];
            plane_rows_bytes(ofc->sw_format, ofc->width, ofc->height, frb, frows);      /* the allocation's rows (aligned context size) */
            uint8_t *host = malloc((size_t)o->linesize[i] * frows[i]);
            CK(gmat_memcpy_d2h(host, o->data[i], (size_t)o->linesize[i] * frows[i]));
            for (int y = 0; y < orows[i]; y++) fwrite(host + (size_t)y * o->linesize[i], 1, orb[i], stdout);
            free(host);
        }
    }
    flt->uninit(&ctx);
    return 0;
}
