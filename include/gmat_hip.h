/*
 * gmat_hip.h — C ABI of the MI355X-native pixel-transform path (libgpuscale + GPU filters).
 *
 * Plain C, no HIP or torch types in any signature: device pointers are `uint8_t *`, streams
 * are `void *` (a hipStream_t; the reference passes CUstream, also an opaque pointer —
 * SURVEY.md §8b).  All work is ENQUEUED on the given stream and the call returns without
 * synchronising, exactly like the reference (swscale_cuda.c:273-479, vf_crop_nvcv.c:209-291).
 * Return values: >= 0 success, negative errno-style code on failure (GMAT_ERR(EINVAL) ...).
 * Every `path:line` below is relative to /root/reference/ffmpeg-gpu/.
 */
#ifndef GMAT_HIP_H
#define GMAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMAT_API __attribute__((visibility("default")))
#define GMAT_ERR(e) (-(e))

/* ---- pixel formats: the integer values of enum AVPixelFormat, libavutil/pixfmt.h:64-376, as the
 * reference tree's preprocessor evaluates it (libavutil major 57: the FF_API_XVMC entry :263-265 is
 * present).  A caller inside the reference tree passes its AV_PIX_FMT_* values unchanged. ---- */
enum GmatPixelFormat {
    GMAT_PIX_FMT_NONE      = -1,
    GMAT_PIX_FMT_YUV420P   = 0,
    GMAT_PIX_FMT_RGB24     = 2,
    GMAT_PIX_FMT_BGR24     = 3,
    GMAT_PIX_FMT_YUV444P   = 5,     /* scale_cuda's list, vf_scale_cuda.c:45-54: source for every output format;
                                       destination of 8-bit YUV sources (any size) and of RGB24/BGR24 (equal size) */
    GMAT_PIX_FMT_NV12      = 23,
    GMAT_PIX_FMT_RGBA      = 26,
    GMAT_PIX_FMT_BGRA      = 28,
    GMAT_PIX_FMT_YUV420P16LE = 45,  /* } swscale_cuda's planar high-depth 4:2:0 formats (swscale_cuda.c:34-44): sources for every 8-bit,   */
    GMAT_PIX_FMT_YUV420P10LE = 62,  /* } 10-bit and 19-bit-path destination at any size, and destinations of every YUV source (10-bit: on  */
                                    /*   the 15-bit lines, yuv2planeX_10_c, bits in the LOW end; 16-bit: on the 19-bit lines)              */
    GMAT_PIX_FMT_YUV444P16LE = 49,  /* scale_cuda's list: source for every destination at any size; destination of every
                                       YUV source on the 19-bit path (with P016LE, RGBA64LE, BGRA64LE) */
    GMAT_PIX_FMT_RGBA64LE  = 105,   /* destinations of every YUV source at any size (yuv2rgba64_*_c on libswscale's 19-bit */
    GMAT_PIX_FMT_BGRA64LE  = 107,   /* lines; alpha 0xFFFF) — yuv2rgb_cuda's 64-bit outputs, yuv2rgb_cuda.cu:862-907; SOURCES   */
                                    /* of every destination at any size (rgb64ToY_c / ToUV_c / ToUV_half_c, input.c:36-121);   */
                                    /* with an alpha channel at both ends the alpha plane is scaled too (needAlpha)             */
    GMAT_PIX_FMT_HIP       = 117,   /* AV_PIX_FMT_CUDA's slot (pixfmt.h:225): opaque device frame */
    GMAT_PIX_FMT_RGB0      = 119,   /* = AV_PIX_FMT_0BGR32 on little endian  } scale_cuda's 32-bit formats (vf_scale_cuda.c:45-54): */
    GMAT_PIX_FMT_BGR0      = 121,   /* = AV_PIX_FMT_0RGB32 on little endian  } RGBA / BGRA whose 4th byte is padding — libswscale     */
                                    /*   handles them as those (handle_0alpha, utils.c:1121-1144): read ignored, written 255          */
    GMAT_PIX_FMT_P010LE    = 159,   /* pixfmt.h:276 — like NV12, 16-bit containers, data in the high bits */
    GMAT_PIX_FMT_P016LE    = 170,   /* both: SOURCE for every 8-bit destination and for P010LE at any size
                                       (hScale16To15_c semantics); DESTINATION of 8-bit 4:2:0 at equal size
                                       (planar8ToP01xleWrapper); equal format and size: plane copy.  Both are also
                                       scaled destinations of every YUV source: P010LE on the 15-bit lines
                                       (yuv2p010lX_c / cX_c), P016LE on libswscale's 19-bit lines (hScale8To19_c,
                                       yuv2planeX_16_c; a plain two-pass path) */
    GMAT_PIX_FMT_RGBPF32LE = 179,   /* GMAT addition, pixfmt.h:315 */
};

/* ---- scaler flags: libswscale/swscale.h:65-95 ---- */
#define GMAT_SWS_FAST_BILINEAR   1
#define GMAT_SWS_BILINEAR        2
#define GMAT_SWS_BICUBIC         4
#define GMAT_SWS_X               8   /* "experimental" (swscale.h:68) */
#define GMAT_SWS_POINT        0x10
#define GMAT_SWS_AREA         0x20
#define GMAT_SWS_BICUBLIN     0x40   /* bicubic luma, bilinear chroma */
#define GMAT_SWS_GAUSS        0x80
#define GMAT_SWS_SINC        0x100
#define GMAT_SWS_LANCZOS     0x200
#define GMAT_SWS_SPLINE      0x400
#define GMAT_SWS_FULL_CHR_H_INT 0x2000
#define GMAT_SWS_FULL_CHR_H_INP 0x4000
#define GMAT_SWS_ACCURATE_RND  0x40000
#define GMAT_SWS_BITEXACT      0x80000
#define GMAT_SWS_HWACCEL      0x1000000   /* == SWS_HWACCEL_CUDA, swscale.h:95 (accepted, implied) */
#define GMAT_SWS_PARAM_DEFAULT 123456

/* colour spaces: swscale.h:99-106 */
#define GMAT_SWS_CS_ITU709   1
#define GMAT_SWS_CS_ITU601   5
#define GMAT_SWS_CS_DEFAULT  5
#define GMAT_SWS_CS_BT2020   9

/* =====================================================================================
 * 1. libgpuscale — the sws-style context API
 *    replaces sws_getContext (libswscale/utils.c:2087) with SWS_HWACCEL_CUDA
 *    -> sws_init_context_cuda (utils.c:2026-2060), sws_scale (swscale.c:1204),
 *    sws_setCudaStream (swscale.h:446-448), sws_freeContext_cuda (swscale.h:170-188).
 *
 *    Semantics are libswscale's CPU arithmetic (SURVEY.md §0 lists why the reference GPU
 *    arithmetic is not the parity target):
 *      - same size, YUV -> RGB: nearest-chroma fixed-point yuv2rgb (yuv2rgb.c:346-405), bit-exact;
 *        with GMAT_SWS_ACCURATE_RND: libswscale's generic path instead (vertical chroma interpolation), which
 *        is what the CPU runs for NV12 always and for planar sources with that flag (swscale_unscaled.c:2094)
 *      - same size, RGB24 <-> BGR24: byte swap (rgb2rgb_template.c)
 *      - everything else: the generic scaler (swscale.c:234-520) in integer arithmetic, bit-exact
 *        with the portable C build (filterAlign 1); a YUV source with a different output size is
 *        scaled exactly as one CPU libswscale context would (see gmat_sws_setFused for the
 *        convert-then-resize alternative that mirrors the reference GPU back-end's structure).
 * ===================================================================================== */
typedef struct GmatSwsContext GmatSwsContext;

GMAT_API GmatSwsContext *gmat_sws_getContext(int srcW, int srcH, int srcFormat,
                                             int dstW, int dstH, int dstFormat,
                                             int flags, const double *param /* [2] or NULL */);
/* src[]/dst[] are device pointers in AVFrame plane order, strides in bytes.  The whole frame is
 * processed (srcSliceY/srcSliceH are validated then ignored, as swscale_cuda.c does). */
GMAT_API int  gmat_sws_scale(GmatSwsContext *c, const uint8_t *const src[], const int srcStride[],
                             int srcSliceY, int srcSliceH,
                             uint8_t *const dst[], const int dstStride[]);
GMAT_API void gmat_sws_setStream(GmatSwsContext *c, void *stream);
GMAT_API void gmat_sws_freeContext(GmatSwsContext *c);
/* sws_setColorspaceDetails subset (utils.c:902-1030): the matrix (SWS_CS_* index, swscale.h:98-107) of the context's
 * YUV end.  YUV source -> RGB: that source's matrix and range (ff_yuv2rgb_c_init_tables).  RGB source -> YUV: the
 * destination's matrix (fill_rgb2yuv_table, utils.c:765-858), limited range only (srcFullRange must be 0). */
GMAT_API int  gmat_sws_setColorspace(GmatSwsContext *c, int colorspace, int srcFullRange);
/* srcRange / dstRange of sws_setColorspaceDetails for YUV -> YUV contexts (1 = full "jpeg" range): when they
 * differ, the h-scaled lines go through lum/chrRangeToJpeg_c or ...FromJpeg_c (swscale.c:157-188, hooked in
 * hscale.c:60,:193) and a same-size context leaves the plane-copy path for the generic one, exactly as
 * libswscale does (utils.c:1996-2000).  An RGB source has no range of its own; an RGB24 / BGR24 -> 4:2:0 context of
 * equal size accepts dstFullRange = 1 (the limited -> full conversion of its 15-bit lines — the second half of
 * libswscale's YUV -> RGB -> YUV cascade for differing matrices, utils.c:966-1036).  Other contexts with an RGB end
 * return -ENOSYS for a non-zero range. */
GMAT_API int  gmat_sws_setRange(GmatSwsContext *c, int srcFullRange, int dstFullRange);
/* the AVOptions src_h_chr_pos / src_v_chr_pos / dst_h_chr_pos / dst_v_chr_pos (options.c:67-70, used by
 * vf_scale.c:567-578): chroma sample positions in 1/256 of a luma sample, -513 = unset.  Rebuilds the chroma
 * filter banks (get_local_pos + initFilter, utils.c:338-345,:1838-1873).  YUV-source scaling contexts only. */
GMAT_API int  gmat_sws_setChromaPos(GmatSwsContext *c, int src_h_chr_pos, int src_v_chr_pos,
                                    int dst_h_chr_pos, int dst_v_chr_pos);

/* how a scaled YUV->RGB context computes:
 *   2 (default) = what ONE libswscale context does for the same arguments: luma and chroma planes are
 *       scaled separately (swscale.c:234-520), LUT colour stage — bit-exact with sws_scale() on the CPU;
 *   1 = the reference GPU back-end's order of operations (convert at source size, then resize in RGB,
 *       swscale_cuda.c:352-371) in libswscale arithmetic, i.e. sws(NV12->RGB24, POINT) followed by
 *       sws(RGB24->RGB, flags), as ONE fused kernel without the full-size RGB frame in HBM;
 *   0 = the same arithmetic as 1 as two kernels with the HBM RGB24 intermediate owned by the context
 *       (the reference's structure, swscale_cuda.c:248-266). */
GMAT_API int  gmat_sws_setFused(GmatSwsContext *c, int fused);
/* introspection used by tests and bench: which 0 hLum 1 hChr 2 vLum 3 vChr.  Copies up to `cap`
 * int16 coefficients / int32 positions to HOST buffers; returns filter size, *count = rows. */
GMAT_API int  gmat_sws_getFilter(const GmatSwsContext *c, int which, int16_t *coef, int32_t *pos,
                                 int cap, int *count);
/* tuning aid: device buffer of 8 x uint64 per workgroup receiving the scaler's phase timestamps
 * (shader clock); NULL disables.  Only the single-context YUV scaler honours it. */
GMAT_API int  gmat_sws_setProfileBuffer(GmatSwsContext *c, uint8_t *devbuf);
/* name of the kernel the last gmat_sws_scale() launched last (static string) */
GMAT_API const char *gmat_sws_lastKernel(const GmatSwsContext *c);
/* number of frames that launch carried: 1, or up to 32 when gmat_sws_scale_batch / gmat_sws_graph_create put a
 * stream's share of the frames into one launch of the 2:1 kernel */
GMAT_API int  gmat_sws_lastLaunchFrames(const GmatSwsContext *c);
/* A context owns ONE set of intermediates (the two-kernel forms, the 16-bit paths, NV12 <-> YUV420P scaled); like the reference's cv_* buffers
 * (swscale_cuda.c:248-266) they are shared by every call.  A call on a stream other than the one that used them last is ordered behind that use
 * by an event inside the library, so alternating streams (gmat_sws_setStream, gmat_sws_scale_batch) is always safe; contexts without
 * intermediates (the fused kernels, every same-size converter) overlap freely.  A context that only ever sees ONE stream records nothing (an
 * event record per call kept the next call's first launch waiting ~ 4 us); the first call on a second stream synchronises the device once
 * (blocking the host), and from then on every call records its last use.  Returns how often that ordering was needed (tests). */
GMAT_API int  gmat_sws_streamHandoffs(const GmatSwsContext *c);

/* ---- the plain-pointer back-end entry points, under the reference's own names ----------
 * libswscale core calls these (swscale_unscaled.c:1970-2012); CUstream == void*.          */
GMAT_API int  yuv2rgb_cuda(const uint8_t *src[], int srcStride[], uint8_t *dst[], int dstStride[],
                           int w, int h, int srcFormat, int dstFormat, void *stream);   /* :1980 */
GMAT_API int  rgb2yuv_cuda(const uint8_t *src[], int srcStride[], uint8_t *dst[], int dstStride[],
                           int w, int h, int srcFormat, int dstFormat, void *stream);   /* :1984 */
GMAT_API int  yuv2yuv_cuda(const uint8_t *src[], int srcStride[], uint8_t *dst[], int dstStride[],
                           int w, int h, int srcFormat, int dstFormat, void *stream);   /* :1988 */
GMAT_API void rgb24tobgr24_cuda(const uint8_t *src[], uint8_t *dst[], int srcStride[], int dstStride[],
                                int width, int height, void *stream);                   /* :1970 */
GMAT_API void rgb2rgb_init_cuda(void);                                 /* rgb2rgb.h:175 (no-op) */

/* ---- caller of libgpuscale: metrans/app/CSwscale.c:9-40 (bound by metrans/python/swscale.py) -- */
GMAT_API GmatSwsContext *SwscaleCuda_Nv12ToRgbpf32_Init(int w, int h);
GMAT_API int  SwscaleCuda_Nv12ToRgbpf32_Convert(GmatSwsContext *c, uint8_t *src, int srcStride,
                                                uint8_t *dst, int dstStride, int w, int h, void *stream);
GMAT_API void SwscaleCuda_Nv12ToRgbpf32_Delete(GmatSwsContext *c);

/* =====================================================================================
 * 2. Frames — the AVFrame / AVHWFramesContext subset the filters touch
 *    (libavutil/frame.h, libavutil/hwcontext.h:124-229, hwcontext_cuda.c:96-193)
 * ===================================================================================== */
typedef struct GmatHWFramesContext GmatHWFramesContext;

typedef struct GmatFrame {
    uint8_t *data[4];          /* device pointers (format == GMAT_PIX_FMT_HIP) or host pointers   */
    int      linesize[4];
    int      width, height;
    int      format;           /* GMAT_PIX_FMT_HIP for device frames, else the sw format          */
    int      sw_format;        /* layout of the planes                                             */
    int64_t  pts;
    int      colorspace;       /* AVCOL_SPC_* as carried by AVFrame.colorspace (copied by props)   */
    GmatHWFramesContext *hw_frames_ctx;   /* owner pool, NULL for caller-owned memory              */
    void    *buf;              /* pool bookkeeping, opaque                                          */
} GmatFrame;

/* av_hwframe_ctx_alloc + field assignment + av_hwframe_ctx_init (hwcontext.h:361,371) */
GMAT_API GmatHWFramesContext *gmat_hwframe_ctx_create(int device, int sw_format, int width, int height,
                                                      int initial_pool_size);
GMAT_API void gmat_hwframe_ctx_free(GmatHWFramesContext *fc);
GMAT_API int  gmat_hwframe_ctx_info(const GmatHWFramesContext *fc, int *device, int *sw_format,
                                    int *width, int *height);
/* av_hwframe_get_buffer (hwcontext.h:382): pooled hipMalloc block, plane layout per
 * hwcontext_cuda.c:183-193 (linesize aligned to 256 B; NV12 UV directly after Y) */
GMAT_API int  gmat_hwframe_get_buffer(GmatHWFramesContext *fc, GmatFrame *frame);
GMAT_API GmatFrame *gmat_frame_alloc(void);                 /* av_frame_alloc  */
GMAT_API void gmat_frame_free(GmatFrame **frame);           /* av_frame_free: returns the buffer to its pool */
/* av_frame_unref for a caller-owned GmatFrame struct filled by gmat_hwframe_get_buffer: returns the block to its pool
 * and clears the struct.  Lifetime rule (AVBufferRef semantics, hwcontext.c:236-258): a frames context lives until
 * gmat_hwframe_ctx_free has been called AND every frame taken from it has been unref'd / freed; frames returned after
 * gmat_hwframe_ctx_free release their device memory at once.  gmat_hwframe_get_buffer on a freed context fails. */
GMAT_API void gmat_frame_unref(GmatFrame *frame);
/* av_hwframe_transfer_data (hwcontext.h:413, hwcontext_cuda.c:221-279): one 2-D async copy per
 * plane on `stream`; direction from which side has format == GMAT_PIX_FMT_HIP (both: device to device, as cuda_transfer_data
 * copies between two hardware frames) */
GMAT_API int  gmat_hwframe_transfer_data(GmatFrame *dst, const GmatFrame *src, void *stream);
/* pinned host staging (hipHostMalloc) so transfers overlap compute */
GMAT_API int  gmat_host_frame_alloc(GmatFrame *frame, int sw_format, int width, int height);
GMAT_API void gmat_host_frame_free(GmatFrame *frame);

/* =====================================================================================
 * 3. Filters — AVFilter-shaped: init / config_props / filter_frame / uninit
 *    crop_hip    <- vf_crop_nvcv.c    options w,h,x,y            (:80-86)
 *    flip_hip    <- vf_flip_nvcv.c    option  code 0|1|-1        (:77-80)
 *    rotate_hip  <- vf_rotate_nvcv.c  options angle, interp, shift_x, shift_y (:79-88);
 *                   multiples of 90 degrees are exact transposes (vf_transpose.c semantics,
 *                   output w/h swapped); other angles run the CPU rotate filter's fixed-point
 *                   arithmetic (vf_rotate.c) about the centre, interp linear|nearest|cubic|area, same-size output on black;
 *                   a non-zero shift_x / shift_y has the reference's meaning (gmat_rotate_shift_translation below) and
 *                   always takes the arbitrary-angle walk at the input's size, also for multiples of 90 degrees
 *    transpose_hip <- vf_transpose.c  option dir 0..3 (names :374-379)
 *    smooth_hip  <- vf_smooth_nvcv.c  options type, kw, kh, border_type, sigmaX, sigmaY (:82-105);
 *                   the default request (3x3 gaussian, no sigma, no border_type) = integer kernel 1 2 1 / 2 4 2 / 1 2 1,
 *                   rdiv 1/16 (vf_convolution.c:495-512 arithmetic and :555-569 borders); any of kw / kh / sigmaX /
 *                   sigmaY / border_type given -> gmat_gauss_blur's float kernel with that border rule; median:
 *                   any odd kw x kh up to 255 (vf_median.c's rule), gaussian-only options refused with EINVAL; gaussian windows up to
 *                   255 taps an axis (beyond: ENOSYS)
 *    scale_hip   <- vf_scale_cuda.c   options (:586-603) on top of libgpuscale: w / h expressions (iw ih ow oh a sar dar
 *                   hsub vsub ohsub ovsub, + - * / ( ) min max trunc floor ceil round abs; -1 / -n as scale_eval.c:113-175),
 *                   interp_algo, format, passthrough (default 1: a frame whose size and format already match is handed on
 *                   untouched, :254-260,:543), param (-> libswscale's param[0]), force_original_aspect_ratio,
 *                   force_divisible_by
 *    format_hip  <- vf_format_cuda.c  option pix_fmt (:69-79)
 *  Input sw formats accepted by the nvcv-style filters: rgb24 bgr24 rgba bgra (vf_crop_nvcv.c:90-98)
 *  and — the two the reference lists but leaves commented out (:91,:94) — nv12 and yuv420p, filtered
 *  plane by plane like the CPU filters (vf_hflip.c:89-117, vf_transpose.c:267-327; crop aligns
 *  x,y,w,h down to the chroma grid, vf_crop.c:186-187,:223-224; nv12 chroma = 2-byte samples).
 * ===================================================================================== */
typedef struct GmatFilterContext GmatFilterContext;

GMAT_API GmatFilterContext *gmat_filter_alloc(const char *name);
GMAT_API int  gmat_filter_set_option(GmatFilterContext *f, const char *key, const char *value); /* AVOption */
GMAT_API int  gmat_filter_init(GmatFilterContext *f);                                /* AVFilter.init          */
/* config_props(outlink): derives the output size/format from the input frames context and creates
 * the output frames context on the same device (vf_crop_nvcv.c:133-207) */
GMAT_API int  gmat_filter_config_props(GmatFilterContext *f, GmatHWFramesContext *in_frames, void *stream);
GMAT_API GmatHWFramesContext *gmat_filter_out_frames(GmatFilterContext *f);
/* filter_frame(inlink, in): takes ownership of `in` (frees it, also on error), returns a pooled
 * output frame with props copied (vf_crop_nvcv.c:209-291) */
GMAT_API int  gmat_filter_frame(GmatFilterContext *f, GmatFrame *in, GmatFrame **out);
GMAT_API void gmat_filter_free(GmatFilterContext *f);                                /* uninit + free */
/* Queued form — libavfilter's activate() model (ff_inlink_consume_frame ... ff_filter_frame) instead of one kernel launch
 * per filter_frame call (vf_scale_cuda.c:532-573), which is launch-bound here: a 4K frame is a 4-8 us kernel.  With the
 * option batch=K (scale_hip, format_hip; default 1) send_frame takes ownership of `in`, queues it, and every K-th call
 * launches ONE kernel for the K queued frames (grid.y = frame); receive_frame returns finished frames in input order or
 * -EAGAIN when none is ready; flush launches a partial batch (EOF).  Other filters and batch=1 process at once. */
GMAT_API int  gmat_filter_send_frame(GmatFilterContext *f, GmatFrame *in);
GMAT_API int  gmat_filter_receive_frame(GmatFilterContext *f, GmatFrame **out);
GMAT_API int  gmat_filter_flush(GmatFilterContext *f);

/* direct launchers behind the filters (device pointers, packed pixels of bpp bytes; bpp 1..4, where
 * 1 = one plane of planar YUV, 2 = the interleaved chroma plane of NV12).  gmat_rotate_flip_smooth
 * takes bpp 3|4 only. */
GMAT_API int gmat_transpose(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride,
                            int inW, int inH, int bpp, int dir, void *stream);
GMAT_API int gmat_flip(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride,
                       int w, int h, int bpp, int code, void *stream);
GMAT_API int gmat_crop(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride,
                       int x, int y, int w, int h, int bpp, void *stream);
GMAT_API int gmat_smooth3x3(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride,
                            int w, int h, int bpp, const int matrix[9], float rdiv, float bias, void *stream);
/* smooth_nvcv type=gaussian with its options (vf_smooth_nvcv.c:88-105): kw x kh odd and <= 255 (the reference takes any int; beyond 255: -ENOSYS), sigmaX / sigmaY (<= 0:
 * derived from the kernel size; sigmaY <= 0: sigmaX), border_type 0 constant(0) 1 replicate 2 reflect 3 wrap 4 reflect101.
 * CV-CUDA's arithmetic is pinned by nothing in the reference; the rule is OpenCV's (cv::getGaussianKernel,
 * cv::borderInterpolate) with float32 accumulation in raster order and out = clip((int)(sum + 0.5f)) — stated in
 * k_transform.hip and restated by the oracle.  smooth_hip uses it whenever kw, kh, sigma or border_type is given; the
 * default 3x3 request keeps the integer kernel above. */
GMAT_API int gmat_gauss_blur(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride, int w, int h, int bpp,
                             int kw, int kh, double sigmaX, double sigmaY, int border_type, void *stream);
/* smooth_nvcv type=median at kw = kh = 3 (vf_smooth_nvcv.c:82-105): per channel the median of the 3x3 window, rows
 * and columns clamped at the frame edges — the CPU median filter's semantics at radius 1, percentile 0.5
 * (vf_median.c:125, median_template.c:101-147).  bpp 1..4. */
GMAT_API int gmat_median3x3(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride,
                            int w, int h, int bpp, void *stream);
/* Arbitrary-angle rotation about the centre, bit-exact with the CPU rotate filter's 16.16 fixed point
 * (vf_rotate.c:198-249 int_sin + interpolate_bilinear8, :410-548 position walk).  angle_rad > 0 turns
 * clockwise; bilinear 0 = nearest; fill = bpp bytes written where the source position is out of range,
 * NULL = leave those pixels untouched (fillcolor=none).  rotate_hip uses it for angles that are not
 * multiples of 90 degrees, with out size = in size and a black background like rotate_nvcv. */
GMAT_API int gmat_rotate(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride,
                         int inW, int inH, int outW, int outH, int bpp, double angle_rad, int bilinear,
                         const uint8_t *fill, void *stream);
/* The same walk with rotate_nvcv's remaining options (vf_rotate_nvcv.c:79-88): interp 0 nearest, 1 linear (what "area" maps to,
 * as cv::warpAffine does), 2 cubic (Catmull-Rom on the clamped 4 x 4 neighbourhood, integer weights); shift_x / shift_y translate
 * the rotated image by that many output pixels.  CV-CUDA's own arithmetic is not in the reference tree: the rule is stated in
 * the test suite's checker (orc_vf.c, orc_rotate2) and held bit for bit. */
GMAT_API int gmat_rotate2(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride,
                          int inW, int inH, int outW, int outH, int bpp, double angle_rad, int interp,
                          double shift_x, double shift_y, const uint8_t *fill, void *stream);
/* rotate_nvcv's shift_x / shift_y as THE REFERENCE means them (vf_rotate_nvcv.c:85-86,276: handed to cvcudaRotate unchanged, "to move the
 * center at the same coord after rotation"): the rotation is about the ORIGIN and the shift is what re-centres, src = R(angle) * (dst - shift).
 * This helper turns such a shift into gmat_rotate2's translation of the centre-rotated image for one plane:
 *   t = shift - (C_out - R^T * C_in),   C = ((w - 1) / 2, (h - 1) / 2),
 * so the re-centring shift of an angle gives t = 0 — exactly the picture of the shift-free call.  rotate_hip applies it whenever a shift
 * is given; with both shifts 0 (unset) it keeps vf_rotate.c's rotation about the centre, where the reference's picture would mostly leave
 * the frame (INTEGRATION.md section 3 lists this as an incompatibility).  CV-CUDA's source is not in the
 * reference tree: the rule is its documented one in THIS library's (vf_rotate.c's, clockwise-positive) sense of rotation — unpinned. */
GMAT_API void gmat_rotate_shift_translation(double angle_rad, double shift_x, double shift_y, int inW, int inH, int outW, int outH,
                                            double *tx, double *ty);
/* per-channel median of a kw x kh window (odd, <= 255: vf_median.c caps its radius at 127), vf_median.c's rule at radius (kw - 1) / 2, radiusV (kh - 1) / 2 */
GMAT_API int gmat_median(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride, int w, int h, int bpp,
                         int kw, int kh, void *stream);
/* rotate(90 clockwise) + horizontal flip + 3x3 smooth in ONE kernel (cfg4 fused form) */
GMAT_API int gmat_rotate_flip_smooth(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride,
                                     int inW, int inH, int bpp, void *stream);

/* n frames of ONE geometry through one launch of a transform kernel (one more grid dimension = frame; 16 frames a launch, more
 * frames more launches) — no reference counterpart: a launch boundary costs 1.6 us + the ramp of a 50 MB kernel, which bounds one 4K
 * frame per launch at 0.56-0.62 of the HBM roofline whatever the kernel.  op: GMAT_OP_*; arg: the direction of
 * gmat_transpose / the code of gmat_flip, else 0.  The queued filter form (option batch) uses it. */
enum { GMAT_OP_ROTATE_FLIP_SMOOTH = 0, GMAT_OP_SMOOTH3X3 = 1, GMAT_OP_TRANSPOSE = 2, GMAT_OP_FLIP = 3, GMAT_OP_MEDIAN3X3 = 4 };
GMAT_API int gmat_op_batch(int op, int n, const uint8_t *const *src, int srcStride, uint8_t *const *dst, int dstStride,
                           int w, int h, int bpp, int arg, void *stream);
/* gmat_rotate2 over n frames of one geometry, angle and background through one launch (grid.z = frame) */
GMAT_API int gmat_rotate2_batch(int n, const uint8_t *const *src, int srcStride, uint8_t *const *dst, int dstStride,
                                int inW, int inH, int outW, int outH, int bpp, double angle_rad, int interp,
                                double shift_x, double shift_y, const uint8_t *fill, void *stream);

/* =====================================================================================
 * 4. Runtime helpers (logging, device, timing) — no reference counterpart beyond av_log
 * ===================================================================================== */
typedef void (*gmat_log_fn)(int level, const char *msg);
GMAT_API void gmat_set_log_callback(gmat_log_fn fn);     /* default: stderr for level <= 16 (AV_LOG_ERROR) */
GMAT_API int  gmat_device_count(void);
GMAT_API int  gmat_set_device(int device);
/* Host placement for one-process-per-GPU callers (BASELINE configs[4]; the reference picks the device per stream in
 * libavutil/hwcontext_cuda.c:395-434 and has no counterpart): the NUMA node of `device` from sysfs (-1: none reported), and
 * "bind the calling thread to the host cores local to `device`" — never beyond the affinity the process already has.
 * Returns the number of cores bound to, 0 when nothing was changed.  Call before allocating pinned host memory
 * (gmat_host_frame_alloc, gmat_pipeline_create): first touch decides which socket the staging ring lives on. */
GMAT_API int  gmat_device_numa_node(int device);
/* compute units of `device` (hipDeviceProp_t.multiProcessorCount; MI355X: 256) — the launch-size rules scale with it; < 0: error */
GMAT_API int  gmat_device_compute_units(int device);
GMAT_API int  gmat_bind_thread_to_device(int device);
GMAT_API const char *gmat_version(void);
/* tests and A/B measurements only: re-read the GMAT_* environment knobs now.  Contexts read them when they are
 * created (gmat_sws_getContext, gmat_filter_init); the stateless launchers (gmat_transpose ... gmat_rotate2_batch) read them on their
 * first call and after this — never per launch, and a launch never invalidates another thread's or context's cached knobs. */
GMAT_API void gmat_knobs_reload(void);
GMAT_API int  gmat_malloc(uint8_t **ptr, size_t bytes);   /* hipMalloc */
GMAT_API int  gmat_free(uint8_t *ptr);
GMAT_API int  gmat_memcpy_h2d(uint8_t *dst, const uint8_t *src, size_t bytes);
GMAT_API int  gmat_memcpy_d2h(uint8_t *dst, const uint8_t *src, size_t bytes);
GMAT_API int  gmat_memset(uint8_t *dst, int value, size_t bytes);
GMAT_API int  gmat_stream_create(void **stream);
GMAT_API int  gmat_stream_destroy(void *stream);
GMAT_API int  gmat_stream_sync(void *stream);
GMAT_API int  gmat_device_sync(void);
/* events for cross-stream ordering (upload stream -> compute stream -> download stream) */
GMAT_API int  gmat_event_create(void **event);
GMAT_API int  gmat_event_record(void *event, void *stream);
GMAT_API int  gmat_stream_wait_event(void *stream, void *event);
GMAT_API int  gmat_event_sync(void *event);
GMAT_API void gmat_event_destroy(void *event);
/* hipEvent pair timing on `stream`: begin/end bracket a region, elapsed in milliseconds */
GMAT_API int  gmat_timer_create(void **timer);
GMAT_API int  gmat_timer_begin(void *timer, void *stream);
GMAT_API int  gmat_timer_end(void *timer, void *stream);
GMAT_API int  gmat_timer_elapsed_ms(void *timer, float *ms);   /* synchronises on the end event */
GMAT_API void gmat_timer_destroy(void *timer);
/* capture one gmat_sws_scale() per frame set (nframes independent frames) into a hipGraph: the
 * per-frame kernels are a few microseconds long, so replaying a captured batch removes the per-launch
 * host cost.  `nbranches` > 1 forks the capture into that many parallel branches (frame f runs on
 * branch f % nbranches), so the tail of one frame's kernel overlaps the head of the next — frames are
 * independent.  The two-kernel form (setFused 0) shares one intermediate and is forced to 1 branch. */
GMAT_API int  gmat_sws_graph_create(GmatSwsContext *c, int nframes,
                                    const uint8_t *const *src_planes /* [nframes][4] */, const int srcStride[],
                                    uint8_t *const *dst_planes /* [nframes][4] */, const int dstStride[],
                                    void *stream, int nbranches, void **graph_exec);
/* enqueue nframes independent frames of the context's geometry in one call (the context's own stream is restored
 * afterwards).  A context that runs the 2:1 kernel (4:2:0 source, exact 2:1 down-scale, 16-byte aligned rows) gives
 * each stream a contiguous share of the frames as ONE launch (grid.y = frame, at most 32 frames per launch, the
 * plane pointers travel in the kernel arguments); every other context enqueues frame f on streams[f % nstreams].
 * The same holds for the branches of gmat_sws_graph_create.  The two-kernel form uses streams[0] only.
 * flags: GMAT_BATCH_FORK makes streams[1..] wait for the work already queued on streams[0];
 *        GMAT_BATCH_JOIN makes streams[0] wait for the batch on every other stream, so an event recorded
 *        on streams[0] afterwards covers the whole batch. */
#define GMAT_BATCH_FORK 1
#define GMAT_BATCH_JOIN 2
GMAT_API int  gmat_sws_scale_batch(GmatSwsContext *c, int nframes,
                                   const uint8_t *const *src_planes /* [nframes][4] */, const int srcStride[],
                                   uint8_t *const *dst_planes /* [nframes][4] */, const int dstStride[],
                                   void *const *streams, int nstreams, int flags);
GMAT_API int  gmat_graph_launch(void *graph_exec, void *stream);
GMAT_API void gmat_graph_destroy(void *graph_exec);

/* =====================================================================================
 * 5. Host pipeline — hwupload -> libgpuscale -> hwdownload with copy / compute overlap
 *    (vf_hwupload_cuda.c:123-150, hwcontext_cuda.c:221-279 do this with pageable memory on one stream and a wait per
 *    frame).  `depth` ring slots of PINNED host frames; three streams (upload, compute, download) chained per slot by
 *    events.  The caller fills gmat_pipeline_host_input(seq) — the slot the NEXT submit will use, seq = number of
 *    submits so far — calls gmat_pipeline_submit(), and reads gmat_pipeline_host_output(seq) after
 *    gmat_pipeline_wait(seq).  submit blocks only while the slot's frame of `depth` submits ago is still downloading.
 *    One pipeline = one device = one host thread; N GPUs run N pipelines (independent streams, no exchange).
 * ===================================================================================== */
typedef struct GmatPipeline GmatPipeline;
GMAT_API GmatPipeline *gmat_pipeline_create(int device, int srcW, int srcH, int srcFormat,
                                            int dstW, int dstH, int dstFormat, int flags, int depth);
GMAT_API int  gmat_pipeline_host_input(GmatPipeline *p, int64_t seq, GmatFrame *view);    /* a view: do not free */
GMAT_API int  gmat_pipeline_host_output(GmatPipeline *p, int64_t seq, GmatFrame *view);
GMAT_API int64_t gmat_pipeline_submit(GmatPipeline *p);       /* sequence number of the frame, or < 0 */
GMAT_API int  gmat_pipeline_wait(GmatPipeline *p, int64_t seq);
GMAT_API int  gmat_pipeline_drain(GmatPipeline *p);
GMAT_API void gmat_pipeline_free(GmatPipeline *p);

#ifdef __cplusplus
}
#endif
#endif /* GMAT_HIP_H */
