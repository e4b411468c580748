#pragma once
#include "common.h"
#define AV_PIX_FMT_FLAG_RGB (1 << 5)
typedef struct AVPixFmtDescriptor { const char *name; uint8_t nb_components, log2_chroma_w, log2_chroma_h; uint64_t flags; } AVPixFmtDescriptor;
const AVPixFmtDescriptor *av_pix_fmt_desc_get(enum AVPixelFormat pix_fmt);
int av_get_padded_bits_per_pixel(const AVPixFmtDescriptor *pixdesc);
const char *av_get_pix_fmt_name(enum AVPixelFormat pix_fmt);
