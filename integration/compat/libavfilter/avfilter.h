#ifndef COMPAT_AVFILTER_H
#define COMPAT_AVFILTER_H
#include "libavutil/common.h"
#include "libavutil/opt.h"
typedef struct AVFilterContext AVFilterContext;
typedef struct AVFilterLink AVFilterLink;
typedef struct AVFilterFormats AVFilterFormats;
typedef struct AVFilterFormatsConfig { AVFilterFormats *formats; } AVFilterFormatsConfig;
typedef struct AVFilterPad {
    const char *name;
    enum AVMediaType type;
    int (*filter_frame)(AVFilterLink *link, AVFrame *frame);
    int (*config_props)(AVFilterLink *link);
} AVFilterPad;
typedef struct AVFilter {
    const char *name, *description;
    const AVFilterPad *inputs, *outputs;
    const AVClass *priv_class;
    int flags;
    uint8_t nb_inputs, nb_outputs, formats_state;
    int (*init)(AVFilterContext *ctx);
    void (*uninit)(AVFilterContext *ctx);
    union { int (*query_func)(AVFilterContext *); } formats;
    int priv_size, flags_internal;
    int (*activate)(AVFilterContext *ctx);
} AVFilter;
struct AVFilterContext { const AVClass *av_class; const AVFilter *filter; char *name; AVFilterLink **inputs, **outputs; void *priv; };
struct AVFilterLink {
    AVFilterContext *src, *dst;
    int w, h, format;
    AVFilterFormatsConfig incfg, outcfg;
    AVBufferRef *hw_frames_ctx;
};
#endif
