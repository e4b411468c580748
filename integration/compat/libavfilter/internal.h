#pragma once
#include "avfilter.h"
#define FF_FILTER_FLAG_HWFRAME_AWARE (1 << 0)
#define NULL_IF_CONFIG_SMALL(x) x
#define FF_ARRAY_ELEMS(a) (sizeof(a) / sizeof((a)[0]))
#define FILTER_INPUTS(array) .inputs = array, .nb_inputs = FF_ARRAY_ELEMS(array)
#define FILTER_OUTPUTS(array) .outputs = array, .nb_outputs = FF_ARRAY_ELEMS(array)
#define FILTER_QUERY_FUNC(func) .formats.query_func = func, .formats_state = 1
#define AVFILTER_DEFINE_CLASS(fname) \
    static const AVClass fname##_class = { .class_name = #fname, .item_name = av_default_item_name, .option = fname##_options, .version = LIBAVUTIL_VERSION_INT }
int ff_filter_frame(AVFilterLink *link, AVFrame *frame);
