/* integration/compat — MINIMAL hand-written declarations for `gcc -fsyntax-only` of the integration sources in a container
 * without the reference tree's generated config.h.  Not the libav* headers: only the names those sources touch, with just
 * enough structure to type-check.  Inside the reference tree the real headers are used instead. */
#ifndef COMPAT_AVUTIL_COMMON_H
#define COMPAT_AVUTIL_COMMON_H
#include <errno.h>
#include <float.h>
#include <limits.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#define AVERROR(e) (-(e))
#define AVERROR_EXTERNAL (-0x20545845)
#define FFERRTAG(a, b, c, d) (-(int)((a) | ((b) << 8) | ((c) << 16) | ((unsigned)(d) << 24)))
#define av_cold __attribute__((cold))
#define FFALIGN(x, a) (((x) + (a) - 1) & ~((a) - 1))
#define AV_CEIL_RSHIFT(a, b) (-((-(a)) >> (b)))
#define AV_LOG_ERROR 16
void av_log(void *avcl, int level, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
enum AVMediaType { AVMEDIA_TYPE_VIDEO };
enum AVPixelFormat {
    AV_PIX_FMT_NONE = -1, AV_PIX_FMT_YUV420P = 0, AV_PIX_FMT_RGB24 = 2, AV_PIX_FMT_BGR24 = 3, AV_PIX_FMT_YUV444P = 5,
    AV_PIX_FMT_NV12 = 23, AV_PIX_FMT_RGBA = 26, AV_PIX_FMT_BGRA = 28, AV_PIX_FMT_YUV444P16 = 49, AV_PIX_FMT_CUDA = 117,
    AV_PIX_FMT_0BGR32 = 119, AV_PIX_FMT_0RGB32 = 121, AV_PIX_FMT_P010 = 159, AV_PIX_FMT_P016 = 170,
    AV_PIX_FMT_YUV420P16 = 45, AV_PIX_FMT_YUV420P10 = 62, AV_PIX_FMT_RGBA64 = 105, AV_PIX_FMT_BGRA64 = 107, AV_PIX_FMT_RGBPF32LE = 179,
};
enum AVColorSpace { AVCOL_SPC_RGB = 0, AVCOL_SPC_BT709 = 1, AVCOL_SPC_UNSPECIFIED = 2, AVCOL_SPC_FCC = 4, AVCOL_SPC_BT470BG = 5, AVCOL_SPC_SMPTE170M = 6,
                    AVCOL_SPC_SMPTE240M = 7, AVCOL_SPC_BT2020_NCL = 9, AVCOL_SPC_BT2020_CL = 10 };
enum AVColorRange { AVCOL_RANGE_UNSPECIFIED = 0, AVCOL_RANGE_MPEG = 1, AVCOL_RANGE_JPEG = 2 };
typedef struct AVClass AVClass;
typedef struct AVBufferRef { uint8_t *data; size_t size; } AVBufferRef;
AVBufferRef *av_buffer_ref(const AVBufferRef *buf);
void av_buffer_unref(AVBufferRef **buf);
typedef struct AVFrame {
    uint8_t *data[8];
    int linesize[8];
    int width, height, format;
    int64_t pts;
    enum AVColorRange color_range;
    enum AVColorSpace colorspace;
    AVBufferRef *hw_frames_ctx;
} AVFrame;
AVFrame *av_frame_alloc(void);
void av_frame_free(AVFrame **frame);
int av_frame_copy_props(AVFrame *dst, const AVFrame *src);
#endif
