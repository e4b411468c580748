#pragma once
#include "avfilter.h"
#define FFERROR_NOT_READY FFERRTAG('N', 'R', 'D', 'Y')
int ff_inlink_consume_frame(AVFilterLink *link, AVFrame **rframe);
int ff_inlink_acknowledge_status(AVFilterLink *link, int *rstatus, int64_t *rpts);
void ff_outlink_set_status(AVFilterLink *link, int status, int64_t pts);
int ff_outlink_get_status(AVFilterLink *link);
int ff_outlink_frame_wanted(AVFilterLink *link);
void ff_inlink_set_status(AVFilterLink *link, int status);
void ff_inlink_request_frame(AVFilterLink *link);
#define FF_FILTER_FORWARD_STATUS_BACK(outlink, inlink) do { int ret_ = ff_outlink_get_status(outlink); \
    if (ret_) { ff_inlink_set_status(inlink, ret_); return 0; } } while (0)
#define FF_FILTER_FORWARD_WANTED(outlink, inlink) do { if (ff_outlink_frame_wanted(outlink)) { ff_inlink_request_frame(inlink); return 0; } } while (0)
