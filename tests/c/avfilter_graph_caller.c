/* tests/c/avfilter_graph_caller.c — TEST INFRASTRUCTURE (build container only: needs /root/reference).
 *
 * The reference's REAL libavfilter + libavutil drive the reference-side sources of this repository: compiled against the reference's own
 * headers and linked with its libavfilter.a / libavutil.a / libswscale.a (tools/build_ref_avfilter.sh), with
 *     integration/hwcontext_hip.c   as libavutil's ff_hwcontext_type_cuda                (hwcontext.c:36-38)
 *     integration/vf_gmat_hip.c     the seven GPU filters, integration/vf_hwupload_hip.c  (const AVFilter ff_vf_*)
 * and the library under test behind them.  It does what an application (fftools/ffmpeg_filter.c) does with libavfilter's public API:
 *     av_hwdevice_ctx_create(AV_HWDEVICE_TYPE_CUDA, "0")        libavutil/hwcontext.c:610 -> HWContextType.device_create
 *     avfilter_graph_alloc_filter / avfilter_init_str / avfilter_link / avfilter_graph_config      (format negotiation, config_props,
 *                                                              av_hwframe_ctx_alloc / _init inside the filters: hwcontext.c:247,333)
 *     av_buffersrc_add_frame -> ... -> av_buffersink_get_frame  (filter_frame / activate() scheduling, frame pools, av_hwframe_transfer_data)
 * on TWO graphs over the same software frames —
 *     buffer -> <gpu chain, e.g. hwupload_hip,scale_hip=w=160:h=90:format=rgb24,hwdownload> -> buffersink
 *     buffer -> <cpu chain, e.g. scale=160:90:flags=bicubic,format=rgb24>                   -> buffersink
 * — and compares every byte of every frame.  Exit code 0: identical.
 *   avfilter_graph_caller <w> <h> <pix_fmt> <nframes> "<gpu chain>" "<cpu chain>" [seed [AVCOL_SPC [AVCOL_RANGE]]]   (tags of the source frames)
 * A chain is `name=args,name=args,...` (no quoting: ',' only between filters); names ending in _hip are this repository's.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libavfilter/avfilter.h"
#include "libavfilter/buffersink.h"
#include "libavfilter/buffersrc.h"
#include "libavutil/frame.h"
#include "libavutil/hwcontext.h"
#include "libavutil/imgutils.h"
#include "libavutil/pixdesc.h"

extern const AVFilter ff_vf_crop_hip, ff_vf_flip_hip, ff_vf_rotate_hip, ff_vf_transpose_hip, ff_vf_smooth_hip, ff_vf_scale_hip,
                      ff_vf_format_hip, ff_vf_hwupload_hip;
static const AVFilter *const ours[] = { &ff_vf_crop_hip, &ff_vf_flip_hip, &ff_vf_rotate_hip, &ff_vf_transpose_hip, &ff_vf_smooth_hip,
                                        &ff_vf_scale_hip, &ff_vf_format_hip, &ff_vf_hwupload_hip };

static const AVFilter *find_filter(const char *name)
{
    for (size_t i = 0; i < sizeof(ours) / sizeof(ours[0]); i++)
        if (!strcmp(ours[i]->name, name))
            return ours[i];
    return avfilter_get_by_name(name);
}

#define CK(x) do { int r_ = (x); if (r_ < 0) { char e_[128]; av_strerror(r_, e_, sizeof(e_)); \
                   fprintf(stderr, "%s failed: %d (%s)\n", #x, r_, e_); return r_; } } while (0)

typedef struct Graph {
    AVFilterGraph *g;
    AVFilterContext *src, *sink;
} Graph;

static int build(Graph *G, const char *chain, int w, int h, enum AVPixelFormat fmt, AVBufferRef *device)
{
    char args[256], *copy = strdup(chain), *save = NULL;
    AVFilterContext *last;
    int n = 0;

    G->g = avfilter_graph_alloc();
    if (!G->g || !copy)
        return AVERROR(ENOMEM);
    /* one thread: libswscale's slice threads give planarCopyWrapper's dither a phase per SLICE (the row counter of DITHER_COPY restarts in every
     * slice, swscale_unscaled.c:1764-1765), so a 10 -> 8 bit copy depends on the thread count; the whole-frame result is the reference */
    G->g->nb_threads = 1;
    snprintf(args, sizeof(args), "video_size=%dx%d:pix_fmt=%d:time_base=1/25:pixel_aspect=1/1", w, h, (int)fmt);
    CK(avfilter_graph_create_filter(&G->src, avfilter_get_by_name("buffer"), "in", args, NULL, G->g));
    last = G->src;
    for (char *tok = strtok_r(copy, ",", &save); tok; tok = strtok_r(NULL, ",", &save)) {
        char *eq = strchr(tok, '='), inst[32];
        const AVFilter *f;
        AVFilterContext *ctx;
        if (eq)
            *eq = 0;
        f = find_filter(tok);
        if (!f) { fprintf(stderr, "no filter '%s'\n", tok); return AVERROR_FILTER_NOT_FOUND; }
        snprintf(inst, sizeof(inst), "f%d_%s", n++, tok);
        ctx = avfilter_graph_alloc_filter(G->g, f, inst);
        if (!ctx)
            return AVERROR(ENOMEM);
        if (device && !strcmp(tok, "hwupload"))                  /* libavfilter's generic uploader takes the device from the application (ffmpeg_filter.c) */
            ctx->hw_device_ctx = av_buffer_ref(device);
        CK(avfilter_init_str(ctx, eq ? eq + 1 : NULL));
        CK(avfilter_link(last, 0, ctx, 0));
        last = ctx;
    }
    CK(avfilter_graph_create_filter(&G->sink, avfilter_get_by_name("buffersink"), "out", NULL, NULL, G->g));
    CK(avfilter_link(last, 0, G->sink, 0));
    CK(avfilter_graph_config(G->g, NULL));
    free(copy);
    return 0;
}

static int g_colorspace = AVCOL_SPC_UNSPECIFIED, g_range = AVCOL_RANGE_UNSPECIFIED;   /* tags of every source frame (argv[8], argv[9]) */

static AVFrame *make_frame(int w, int h, enum AVPixelFormat fmt, uint32_t seed, int64_t pts)
{
    AVFrame *f = av_frame_alloc();
    const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(fmt);
    uint32_t s = seed;
    if (!f)
        return NULL;
    f->format = fmt; f->width = w; f->height = h; f->pts = pts;
    f->colorspace = (enum AVColorSpace)g_colorspace; f->color_range = (enum AVColorRange)g_range;
    if (av_frame_get_buffer(f, 0) < 0) { av_frame_free(&f); return NULL; }
    for (int p = 0; p < 4 && f->data[p]; p++) {
        const int rows = (p == 1 || p == 2) ? AV_CEIL_RSHIFT(h, d->log2_chroma_h) : h;
        const int bytes = av_image_get_linesize(fmt, w, p);
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < bytes; x++) {
                s = s * 1664525u + 1013904223u;
                /* smooth-ish content with noise on top: gradients exercise the filters' rounding, noise their clipping */
                f->data[p][(size_t)y * f->linesize[p] + x] = (uint8_t)(((x * 3 + y * 5) & 0xFF) / 2 + (s >> 25));
            }
        if (d->comp[0].depth > 8 && d->comp[0].depth < 16)     /* valid input: `depth` significant bits where the format keeps them (P010: high, planar 10-bit: low) */
            for (int y = 0; y < rows; y++)
                for (int x = 0; x + 1 < bytes; x += 2) {
                    uint8_t *q = f->data[p] + (size_t)y * f->linesize[p] + x;
                    const unsigned v = (q[0] | q[1] << 8) >> (16 - d->comp[0].depth) << d->comp[0].shift;
                    q[0] = (uint8_t)v; q[1] = (uint8_t)(v >> 8);
                }
    }
    return f;
}

static long compare(const AVFrame *a, const AVFrame *b, int frame_no)
{
    const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(a->format);
    long bad = 0;
    if (a->format != b->format || a->width != b->width || a->height != b->height) {
        fprintf(stderr, "frame %d: %s %dx%d against %s %dx%d\n", frame_no, av_get_pix_fmt_name(a->format), a->width, a->height,
                av_get_pix_fmt_name(b->format), b->width, b->height);
        return 1 << 30;
    }
    for (int p = 0; p < 4 && a->data[p]; p++) {
        const int rows = (p == 1 || p == 2) ? AV_CEIL_RSHIFT(a->height, d->log2_chroma_h) : a->height;
        const int bytes = av_image_get_linesize(a->format, a->width, p);
        for (int y = 0; y < rows; y++) {
            const uint8_t *ra = a->data[p] + (ptrdiff_t)y * a->linesize[p], *rb = b->data[p] + (ptrdiff_t)y * b->linesize[p];
            for (int x = 0; x < bytes; x++)
                if (ra[x] != rb[x]) {
                    if (bad < 6)
                        fprintf(stderr, "frame %d plane %d (%d, %d): gpu %d cpu %d\n", frame_no, p, x, y, ra[x], rb[x]);
                    bad++;
                }
        }
    }
    return bad;
}

/* av_hwframe_transfer_data between two HARDWARE frames of two pools (hwcontext.c:448-467: transfer_data_from of the source's context, then
 * transfer_data_to of the destination's): up, device to device, down — the bytes that went in (ENOSYS where cuda_transfer_data copies) */
static int d2d_check(AVBufferRef *device, int w, int h, enum AVPixelFormat fmt, uint32_t seed)
{
    AVBufferRef *pool[2] = { NULL, NULL };
    AVFrame *hw[2] = { av_frame_alloc(), av_frame_alloc() }, *in = make_frame(w, h, fmt, seed, 0), *back = av_frame_alloc();
    long bad;
    if (!hw[0] || !hw[1] || !in || !back)
        return AVERROR(ENOMEM);
    for (int i = 0; i < 2; i++) {
        AVHWFramesContext *fc;
        pool[i] = av_hwframe_ctx_alloc(device);
        if (!pool[i])
            return AVERROR(ENOMEM);
        fc = (AVHWFramesContext *)pool[i]->data;
        fc->format = AV_PIX_FMT_CUDA; fc->sw_format = fmt; fc->width = w; fc->height = h;
        CK(av_hwframe_ctx_init(pool[i]));
        CK(av_hwframe_get_buffer(pool[i], hw[i], 0));
    }
    CK(av_hwframe_transfer_data(hw[0], in, 0));
    CK(av_hwframe_transfer_data(hw[1], hw[0], 0));
    back->format = fmt;
    CK(av_hwframe_transfer_data(back, hw[1], 0));
    back->width = w; back->height = h;
    bad = compare(back, in, 0);
    printf("%s %dx%d  upload, device to device, download: %ld mismatching bytes\n", av_get_pix_fmt_name(fmt), w, h, bad);
    av_frame_free(&hw[0]); av_frame_free(&hw[1]); av_frame_free(&in); av_frame_free(&back);
    av_buffer_unref(&pool[0]); av_buffer_unref(&pool[1]);
    return bad ? 4 : 0;
}

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: avfilter_graph_caller w h pix_fmt nframes \"gpu chain\" \"cpu chain\" [seed]\n"); return 1; }
    const int w = atoi(argv[1]), h = atoi(argv[2]), nframes = atoi(argv[4]);
    const enum AVPixelFormat fmt = av_get_pix_fmt(argv[3]);
    const uint32_t seed = argc > 7 ? (uint32_t)atoi(argv[7]) : 2026u;
    if (argc > 8) g_colorspace = atoi(argv[8]);             /* AVCOL_SPC_*: 1 bt709, 5 bt470bg, 6 smpte170m, 7 smpte240m, 9 bt2020nc */
    if (argc > 9) g_range = atoi(argv[9]);                  /* AVCOL_RANGE_*: 1 limited (mpeg), 2 full (jpeg) */
    AVBufferRef *device = NULL;
    Graph gpu = {0}, cpu = {0};
    long bad = 0;
    int got = 0;

    if (fmt == AV_PIX_FMT_NONE || nframes < 1 || nframes > 64) { fprintf(stderr, "bad arguments\n"); return 1; }
    av_log_set_level(AV_LOG_WARNING);
    CK(av_hwdevice_ctx_create(&device, AV_HWDEVICE_TYPE_CUDA, "0", NULL, 0));
    if (!strcmp(argv[5], "d2d")) {
        int r = d2d_check(device, w, h, fmt, seed);
        av_buffer_unref(&device);
        return r;
    }
    CK(build(&gpu, argv[5], w, h, fmt, device));
    CK(build(&cpu, argv[6], w, h, fmt, NULL));

    AVFrame *outs_gpu[64] = {0}, *outs_cpu[64] = {0};
    int ng = 0, nc = 0;
    for (int i = 0; i <= nframes; i++) {
        /* frame i into both graphs (i == nframes: EOF, which flushes a partial batch of the queued filters), then whatever has come out */
        for (int k = 0; k < 2; k++) {
            Graph *G = k ? &cpu : &gpu;
            AVFrame *in = i < nframes ? make_frame(w, h, fmt, seed + 977u * i, i) : NULL;
            if (i < nframes && !in)
                return 2;
            CK(av_buffersrc_add_frame_flags(G->src, in, AV_BUFFERSRC_FLAG_PUSH));
            av_frame_free(&in);
            for (;;) {
                AVFrame *out = av_frame_alloc();
                int r = av_buffersink_get_frame(G->sink, out);
                if (r == AVERROR(EAGAIN) || r == AVERROR_EOF) { av_frame_free(&out); break; }
                CK(r);
                if ((k ? nc : ng) >= 64) return 2;
                if (k) outs_cpu[nc++] = out; else outs_gpu[ng++] = out;
            }
        }
    }
    if (ng != nframes || nc != nframes) {
        fprintf(stderr, "%d frames in, %d out of the gpu graph, %d out of the cpu graph\n", nframes, ng, nc);
        return 3;
    }
    for (int i = 0; i < nframes; i++) {
        if (outs_gpu[i]->pts != outs_cpu[i]->pts) { fprintf(stderr, "frame %d: pts %ld against %ld\n", i, (long)outs_gpu[i]->pts, (long)outs_cpu[i]->pts); bad++; }
        /* the tags a downstream filter would act on: a frame must say what its samples are (vf_scale.c:783-786,:831) */
        if (argc > 9 && (outs_gpu[i]->color_range != outs_cpu[i]->color_range || outs_gpu[i]->colorspace != outs_cpu[i]->colorspace)) {   /* (tagged runs: scale_hip / format_hip against `scale`) */
            fprintf(stderr, "frame %d: color_range %d colorspace %d against %d %d\n", i, outs_gpu[i]->color_range, outs_gpu[i]->colorspace,
                    outs_cpu[i]->color_range, outs_cpu[i]->colorspace);
            bad++;
        }
        bad += compare(outs_gpu[i], outs_cpu[i], i);
        got++;
    }
    printf("%s %dx%d x%d  [%s]  ==  [%s]  ->  %s %dx%d: %ld mismatching bytes\n", argv[3], w, h, nframes, argv[5], argv[6],
           av_get_pix_fmt_name(outs_gpu[0]->format), outs_gpu[0]->width, outs_gpu[0]->height, bad);
    for (int i = 0; i < nframes; i++) { av_frame_free(&outs_gpu[i]); av_frame_free(&outs_cpu[i]); }
    avfilter_graph_free(&gpu.g);
    avfilter_graph_free(&cpu.g);
    av_buffer_unref(&device);
    return bad ? 4 : (got == nframes ? 0 : 3);
}
