#pragma once
#include "common.h"
enum AVHWDeviceType { AV_HWDEVICE_TYPE_CUDA = 2 };
typedef struct AVHWDeviceContext { const AVClass *av_class; enum AVHWDeviceType type; void *hwctx; } AVHWDeviceContext;
typedef struct AVHWFramesContext {
    const AVClass *av_class;
    AVBufferRef *device_ref;
    AVHWDeviceContext *device_ctx;
    enum AVPixelFormat format, sw_format;
    int width, height;
} AVHWFramesContext;
AVBufferRef *av_hwframe_ctx_alloc(AVBufferRef *device_ctx);
int av_hwframe_ctx_init(AVBufferRef *ref);
int av_hwframe_get_buffer(AVBufferRef *hwframe_ctx, AVFrame *frame, int flags);
int av_hwdevice_ctx_create(AVBufferRef **device_ctx, enum AVHWDeviceType type, const char *device, void *opts, int flags);
