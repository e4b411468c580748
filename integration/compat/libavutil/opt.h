#pragma once
#include "common.h"
enum AVOptionType { AV_OPT_TYPE_FLAGS, AV_OPT_TYPE_INT, AV_OPT_TYPE_INT64, AV_OPT_TYPE_DOUBLE, AV_OPT_TYPE_FLOAT, AV_OPT_TYPE_STRING,
                    AV_OPT_TYPE_CONST, AV_OPT_TYPE_PIXEL_FMT, AV_OPT_TYPE_BOOL };
#define AV_OPT_FLAG_VIDEO_PARAM 16
#define AV_OPT_FLAG_FILTERING_PARAM (1 << 16)
typedef struct AVOption {
    const char *name, *help;
    int offset;
    enum AVOptionType type;
    union { int64_t i64; double dbl; const char *str; } default_val;
    double min, max;
    int flags;
    const char *unit;
} AVOption;
struct AVClass { const char *class_name; const char *(*item_name)(void *ctx); const AVOption *option; int version; };
const char *av_default_item_name(void *ctx);
#define LIBAVUTIL_VERSION_INT 0
