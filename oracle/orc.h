/*
 * oracle/orc.h — CPU restatement of the reference's pixel-transform arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gmat_amd/ (the product) may include,
 * link, dlopen or call anything declared here; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg do.  The product path fails loudly when its HIP
 * library is missing — it never falls back to this code.
 *
 * What is restated (all paths relative to /root/reference/ffmpeg-gpu):
 *   libswscale/yuv2rgb.c      — fixed-point yuv->rgb look-up tables and the
 *                               yuv2rgb_c_24_rgb/_bgr/_32 "fast path" converters
 *   libswscale/utils.c        — initFilter() coefficient generator, the chroma
 *                               sub-sampling / filter-selection logic of
 *                               sws_init_single_context(), fill_rgb2yuv_table()
 *   libswscale/input.c        — rgb24/bgr24 readers (ToY, ToUV, ToUV_half), nv12 reader
 *   libswscale/swscale.c      — hScale8To15_c / hScale16To15_c, the swscale() row schedule
 *   libswscale/output.c       — yuv2rgb_{X,2,1}_c and yuv2rgb_full_{X,2,1}_c templates,
 *                               yuv2planeX_8_c / yuv2plane1_8_c / yuv2nv12cX_c
 *   libavfilter/vf_transpose.c, vf_hflip.c, vf_vflip.c, vf_crop.c, vf_convolution.c, vf_rotate.c
 *
 * PARITY PIN STATUS: the reference's libswscale cannot be built in this
 * image without generated headers (config.h) and stand-ins for cuda.h / CV-CUDA, so no
 * oracle/_ref exists.  The oracle is pinned against the reference's OWN golden values:
 *   (1) FATE checksums the reference tree ships (tests/ref/fate, tests/ref/pixfmt), reproduced
 *       bit-for-bit by tests/test_oracle_fate.py from the restated vsynth1 clip (orc_vsynth.c):
 *         filter-transpose (50 frames)    -> orc_transpose
 *         sws-yuv-range                   -> 1-tap hScale8To15, range conversion, yuv2plane1_8
 *         filter-scalechroma (25 frames)  -> initFilter with chroma positions, hScale8To15,
 *                                            yuv2planeX_8 (2:1 bicubic, the headline geometry)
 *         filter-colorlevels (50 frames)  -> yuv420p->rgb24 generic path: yuv2rgb_X_c and the
 *                                            yuv2rgb.c tables shared with the fast path
 *         pixfmt-rgb24 / -bgr24 / -yuv420p -> rgb24ToY/ToUV, hScale16To15, chroma up-scaling
 *         sws-yuv-colorspace              -> BT.709 yuv2rgb tables, BGR readers + BT.601 rgb2yuv literals,
 *                                            range conversion of an RGB-sourced context (the cascade's two halves)
 *       and, through tests/test_oracle_fate_nut.py (NUT md5s over the restated muxer), filter-pixfmts-scale for rgba64le /
 *       bgra64le -> rgb64To{Y,UV_half}_c, rgba64leToA_c, the alpha leg of the 19-bit lines and of yuv2rgba64_*; for rgba /
 *       bgra -> the 32-bit readers and rgbaToA_c with the 8-bit writers' alpha (opaque alpha in both: the vsynth clip has none);
 *   (2) known-answer values the survey recorded from the reference (SURVEY.md §8a row 7,
 *       §8c item 5) and constants literal in its sources (ff_yuv2rgb_coeffs, BT.601 literals);
 *   (3) the LUT path and the closed form being two independent restatements that must agree
 *       over all 2^24 (Y,U,V) triples.
 * Not covered by a reference golden value (pinned only by construction from pinned parts):
 * the yuv2rgb_full_* (FULL_CHR_H_INT) outputs, the Lanczos branch of initFilter, the nearest-
 * chroma frame walk of yuv2rgb_c_24_* (its tables are pinned by filter-colorlevels), ToUV_half,
 * the P010LE / P016LE readers (p010LEToY_c / ToUV_c, three lines each; the hScale16To15_c they
 * feed is pinned through the RGB sources), the P010LE / P016LE / RGBA64 output stages (yuv2p010*, the 19-bit
 * hScale*To19_c / yuv2planeX_16_c / yuv2rgba64_* family), the 24 <-> 32 bit RGB re-packing (orc_rgb_repack),
 * hflip/vflip/crop/convolution (their FATE references are NUT-container md5s, which would need
 * the muxer restated).
 */
#ifndef GMAT_ORACLE_ORC_H
#define GMAT_ORACLE_ORC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enum AVPixelFormat values, libavutil/pixfmt.h:65-... of the reference tree */
enum {
    ORC_PIX_YUV420P = 0,
    ORC_PIX_RGB24   = 2,
    ORC_PIX_BGR24   = 3,
    ORC_PIX_YUV444P = 5,
    ORC_PIX_NV12    = 23,
    ORC_PIX_RGBA    = 26,
    ORC_PIX_BGRA    = 28,
    ORC_PIX_P010LE  = 159,
    ORC_PIX_P016LE  = 170,
    ORC_PIX_YUV444P16LE = 49,
    ORC_PIX_YUV420P16LE = 45,       /* planar 4:2:0, 16-bit containers: sources only (swscale_cuda.c:34-44 lists both) */
    ORC_PIX_YUV420P10LE = 62,       /* 10 significant bits in the LOW end of each sample */
    ORC_PIX_RGBA64LE = 105,
    ORC_PIX_BGRA64LE = 107,
    ORC_PIX_RGBPF32LE = 179,
};

/* libswscale/swscale.h:65-95 */
#define ORC_SWS_FAST_BILINEAR 1
#define ORC_SWS_BILINEAR      2
#define ORC_SWS_BICUBIC       4
#define ORC_SWS_POINT      0x10
#define ORC_SWS_AREA       0x20
#define ORC_SWS_LANCZOS   0x200
#define ORC_SWS_FULL_CHR_H_INT 0x2000
#define ORC_SWS_FULL_CHR_H_INP 0x4000
#define ORC_SWS_ACCURATE_RND  0x40000
#define ORC_SWS_BITEXACT      0x80000
#define ORC_SWS_PARAM_DEFAULT 123456

#define ORC_TABLE_HEADROOM      512   /* swscale_internal.h:45 */
#define ORC_TABLE_LUMA_HEADROOM 512   /* swscale_internal.h:46 */

/* yuv2rgb.c:774-1030 (24 bpp branch :958-971) */
typedef struct OrcYuv2Rgb {
    int      full_range;
    int64_t  cy, oy, yb0;                 /* luma scale / offset / table origin       */
    int64_t  crv, cbu, cgu, cgv;          /* chroma increments after the /cy rescale  */
    int      yoffs;
    int16_t  y_coeff, y_offset, v2r, v2g, u2g, u2b;   /* yuv2rgb.c:843-848 */
    uint8_t  y_table[1024 + 2 * ORC_TABLE_LUMA_HEADROOM];
    int32_t  off_rV[256 + 2 * ORC_TABLE_HEADROOM];    /* index offsets into y_table   */
    int32_t  off_gU[256 + 2 * ORC_TABLE_HEADROOM];
    int32_t  off_gV[256 + 2 * ORC_TABLE_HEADROOM];
    int32_t  off_bU[256 + 2 * ORC_TABLE_HEADROOM];
} OrcYuv2Rgb;

/* colour matrix of the YUV end of a context (SWS_CS_* index, swscale.h:98-107); -1 for RGB -> RGB / YUV -> YUV */
struct OrcSws;
int  orc_sws_set_colorspace(struct OrcSws *c, int colorspace);
int  orc_yuv2rgb_init(OrcYuv2Rgb *t, int colorspace, int full_range,
                      int brightness, int contrast, int saturation);
/* one pixel through the look-up tables exactly as LOADCHROMA/PUTRGB24 do */
void orc_yuv2rgb_lut_px(const OrcYuv2Rgb *t, int Y, int U, int V, uint8_t rgb[3]);
/* the closed form of SURVEY.md §8a row 1 — independent of the tables */
void orc_yuv2rgb_closed_px(const OrcYuv2Rgb *t, int Y, int U, int V, uint8_t rgb[3]);
/* returns number of mismatching triples over all 2^24 inputs */
long orc_yuv2rgb_selfcheck(const OrcYuv2Rgb *t);

/* yuv2rgb.c:346-405 fast path: nearest chroma, 2x2 quads.  src_fmt NV12 or YUV420P,
 * dst_fmt RGB24/BGR24/RGBA/BGRA.  Handles odd w/h by clamping the chroma index
 * (the reference requires even sizes; the GPU kernels accept odd ones the same way
 * the reference's yuv2rgb_odd_kernel does). */
int  orc_yuv2rgb_frame(const OrcYuv2Rgb *t, const uint8_t *const src[4], const int src_stride[4],
                       uint8_t *dst, int dst_stride, int w, int h, int src_fmt, int dst_fmt);
/* nv12 -> planar float rgb, value = u8 / 255.0f (yuv2rgb_cuda.cu:381-545 semantics with
 * the integer colour stage substituted) */
int  orc_nv12_to_rgbpf32(const OrcYuv2Rgb *t, const uint8_t *const src[4], const int src_stride[4],
                         float *dst, int dst_stride_bytes, int w, int h);

/* utils.c:367-763 */
int  orc_init_filter(int16_t **out_filter, int32_t **filter_pos, int *out_filter_size,
                     int x_inc, int src_w, int dst_w, int filter_align, int one,
                     int flags, const double param[2], int src_pos, int dst_pos);
void orc_free(void *p);

/* generic scaler context (utils.c:1293-2020 + swscale.c:234-520) */
typedef struct OrcSws OrcSws;
OrcSws *orc_sws_create(int src_w, int src_h, int src_fmt, int dst_w, int dst_h, int dst_fmt,
                       int flags, const double param[2]);
OrcSws *orc_sws_create_ex(int src_w, int src_h, int src_fmt, int dst_w, int dst_h, int dst_fmt,
                          int flags, const double param[2], const int chr_pos[4], int src_range, int dst_range);
int   orc_sws_scale(OrcSws *c, const uint8_t *const src[4], const int src_stride[4],
                    uint8_t *const dst[4], const int dst_stride[4]);
/* row-sliced variant for the threaded cpu baseline: computes output rows [y0,y1) */
int   orc_sws_scale_rows(OrcSws *c, const uint8_t *const src[4], const int src_stride[4],
                         uint8_t *const dst[4], const int dst_stride[4], int y0, int y1);
void  orc_sws_free(OrcSws *c);
/* introspection for the known-answer tests: which = 0 hLum, 1 hChr, 2 vLum, 3 vChr */
int   orc_sws_filter(const OrcSws *c, int which, const int16_t **coef, const int32_t **pos,
                     int *size, int *count);
int   orc_sws_info(const OrcSws *c, int *chr_src_w, int *chr_src_h, int *chr_dst_w, int *chr_dst_h,
                   int *flags);

/* filters (libavfilter CPU counterparts) — packed pixels of `bpp` bytes */
void orc_transpose(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride,
                   int in_w, int in_h, int bpp, int dir);           /* vf_transpose.c:267-327 */
void orc_hflip(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride,
               int w, int h, int bpp);                              /* vf_hflip.c:89-117 */
void orc_vflip(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride,
               int w, int h, int bpp);                              /* vf_vflip.c:108-127 */
void orc_crop(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride,
              int x, int y, int w, int h, int bpp);                 /* vf_crop.c */
/* vf_convolution.c:495-512 + setup_3x3 :555-569 applied per channel of a packed frame */
void orc_conv3x3(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride,
                 int w, int h, int bpp, const int matrix[9], float rdiv, float bias);
/* vf_median.c / median_template.c at radius 1, percentile 0.5: per-channel 3x3 median, edges clamped */
void orc_median3x3(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int w, int h, int bpp);
/* vf_rotate.c:198-548 — arbitrary angle (radians, clockwise positive), 16.16 fixed point; fill NULL = leave */
void orc_median(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int w, int h, int bpp, int kw, int kh);
void orc_rotate2(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int inw, int inh, int outw, int outh, int bpp,
                 double angle_rad, int interp, double shift_x, double shift_y, const uint8_t *fill);
void orc_rotate_sincos(double angle_rad, int *s, int *c);
void orc_rotate(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride,
                int inw, int inh, int outw, int outh, int bpp, double angle_rad, int bilinear,
                const uint8_t *fill);
/* planar8ToP01xleWrapper, swscale_unscaled.c:286-324: 8-bit 4:2:0 -> P010LE / P016LE, t -> t | t << 8.
 * src_nv12 != 0: the same rule applied to an NV12 source (src[1] interleaved). */
void orc_yuv420_to_p01x(const uint8_t *const src[4], const int src_stride[4], uint8_t *const dst[4],
                        const int dst_stride[4], int w, int h, int src_nv12);
/* planarCopyWrapper's 8 -> `depth` bit plane copy (swscale_unscaled.c:1844-1862); shiftonly: chroma, and luma of a limited-range source */
void orc_plane_copy_up(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int w, int h, int depth, int shiftonly);
void orc_rgb24_swap_rb(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride,
                       int w, int h);                               /* rgb2rgb_template.c rgb24tobgr24 */
/* rgbToRgbWrapper (swscale_unscaled.c:1579-1640): RGB24/BGR24 <-> RGBA/BGRA and RGBA <-> BGRA byte re-packing at
 * equal size (alpha 255 when created, dropped when removed, kept for 32 -> 32); -1 for other pairs */
int orc_rgb_repack(const uint8_t *src, int src_stride, int src_fmt, uint8_t *dst, int dst_stride, int dst_fmt,
                   int w, int h);

/* tests/videogen.c + tests/utils.c: frames [0,nframes) of the reference's vsynth1 clip, yuv420p */
int orc_vsynth1(uint8_t *out, int w, int h, int nframes);

/* deterministic synthetic planes: s = s*1664525 + 1013904223, byte = s>>24 */
void orc_fill_lcg(uint8_t *p, long n, uint32_t seed);

/* orc_metrans.c — MeTrans kernels with in-tree arithmetic (metrans/include/NvCodec/Resize_bicubic.cu:83-159, BitDepth.cu:15-29);
 * parity unpinned: no reference vector exists, CUDA cannot run here (header of orc_metrans.c) */
int  orc_mt_scale_nv12_bicubic(const uint8_t *src, int src_pitch, int src_w, int src_h, uint8_t *dst, int dst_pitch, int dst_w, int dst_h);
void orc_mt_u8_to_u16(const uint8_t *src, uint16_t *dst, long n);
void orc_mt_u16_to_u8(const uint16_t *src, uint8_t *dst, long n);

/* Gaussian blur, OpenCV / CV-CUDA rule (parity unpinned: vf_smooth_nvcv.c:88-105,:290-294 only names the options) */
int  orc_gauss_blur(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int w, int h, int bpp,
                    int kw, int kh, double sigma_x, double sigma_y, int border);

#ifdef __cplusplus
}
#endif
#endif
