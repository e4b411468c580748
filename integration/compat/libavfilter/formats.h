#pragma once
#include "avfilter.h"
AVFilterFormats *ff_make_format_list(const int *fmts);
int ff_formats_ref(AVFilterFormats *formats, AVFilterFormats **ref);
int ff_set_common_formats_from_list(AVFilterContext *ctx, const int *fmts);
