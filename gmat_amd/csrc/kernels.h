// kernels.h — launch interface between the host layer and the HIP kernels (gfx950).
#pragma once
#include <cstdint>
#include <vector>
#include <hip/hip_runtime.h>
#include "sws_tables.h"

namespace gmat {

// ---- colour conversion (k_yuv2rgb.hip) ---------------------------------------------------
struct YuvSrc {
    const uint8_t *y, *u, *v;      // nv12: u = interleaved UV plane, v unused
    int ys, us, vs;                // strides in bytes
    int nv12;                      // 1: interleaved chroma
};

// Frames of one launch (one grid dimension = frame): the plane pointers travel in the kernel-argument segment, so a
// batch needs no device-side pointer table.  Geometry, strides and alignment class are shared by every frame.
constexpr int kYuv2xMaxFrames = 32;
struct Yuv2xFrames {
    const uint8_t *y[kYuv2xMaxFrames], *u[kYuv2xMaxFrames], *v[kYuv2xMaxFrames];
    uint8_t *dst[kYuv2xMaxFrames], *dstU[kYuv2xMaxFrames], *dstV[kYuv2xMaxFrames];
};

// nearest-chroma yuv420 -> packed rgb, libswscale fixed-point arithmetic; frames != nullptr: nframes frames in one
// launch (strides from src / dstStride, pointers from *frames)
int launch_yuv2rgb(const YuvSrc &src, uint8_t *dst, int dstStride, int w, int h, int dstFormat,
                   const Yuv2RgbConsts &k, hipStream_t stream, const Yuv2xFrames *frames = nullptr, int nframes = 1);
// nv12 -> planar float rgb (value = u8 / 255.0f), plane stride = dstStride * h
int launch_nv12_to_rgbpf32(const YuvSrc &src, uint8_t *dst, int dstStride, int w, int h,
                           const Yuv2RgbConsts &k, hipStream_t stream, const Yuv2xFrames *frames = nullptr, int nframes = 1);
int launch_swap_rb24(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride, int w, int h,
                     hipStream_t stream);
// 24 <-> 32 bit and 32 <-> 32 bit packed RGB at equal size (rgbToRgbWrapper's byte moves); swapRB exchanges bytes 0 and 2
int launch_repack_rgb(const uint8_t *src, int srcStride, int srcBpp, uint8_t *dst, int dstStride, int dstBpp, int w, int h,
                      int swapRB, hipStream_t stream);
// planar float rgb (plane stride = srcStride * h) -> packed rgb24 / bgr24, u8 = (int)(clamp(f, 0, 1) * 255 + 0.5)
int launch_rgbpf32_to_rgb24(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride, int w, int h, int bgr,
                            hipStream_t stream);

// ---- MeTrans-only kernels (k_metrans.hip) ---------------------------------------------------
// the reference's own NV12 bicubic (Resize_bicubic.cu:83-159): float 4 x 4, a = -0.5, coordinates clamped to [2, n - 2]; one allocation
// per side, chroma at base + pitch * height
int launch_scale_nv12_bicubic_ref(const uint8_t *src, int ss, int srcW, int srcH, uint8_t *dst, int ds, int dstW, int dstH, hipStream_t stream);
// nv12 -> three stacked planes (plane stride = ds * h) of 8-bit (f32 = 0) or float = u8 / 255 samples, R,G,B or (bgr) B,G,R
int launch_nv12_to_planar(const YuvSrc &s, uint8_t *dst, int ds, int w, int h, const Yuv2RgbConsts &k, int f32, int bgr, hipStream_t stream);
// 32-bit packed pixels -> three stacked planes of their first three bytes, in byte order
int launch_split_packed32(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int f32, hipStream_t stream);
// BitDepth.cu:15-36: dst = src << 8 / dst = src >> 8 over n samples
int launch_widen_shift8(const uint8_t *src, uint16_t *dst, long n, hipStream_t stream);
int launch_narrow_shift8(const uint16_t *src, uint8_t *dst, long n, hipStream_t stream);

// ---- generic scaler, packed-RGB output (k_scale.hip) ---------------------------------------
// Device-resident copy of a FilterBank in the dword-packed form.
struct DevFilter {
    const int32_t *packed;    // count * pairs  (int16 pairs for v_dot2)
    const int32_t *pos_even;  // count
    const int32_t *round;     // count; vertical only: accumulator start value (512, or 0 for the
                              // reference's 2-tap "packed2" form, vscale.c:146-160 / output.c:2118)
    int pairs, taps, count;
};

// Output tiling of one context.  A block produces a TW x TH output tile; the host precomputes,
// per tile column / tile row, the window of source columns / rows the tile's filter taps touch.
struct ScaleTiling {
    int TW = 0, TH = 0, ntx = 0, nty = 0;
    int maxRows = 0, maxCols = 0;        // over all tiles (rows even, cols multiple of 4)
    int ldsBytes = 0;
    int xcdRemap = 1, chromaDirect = 0;
    std::vector<int32_t> colStart, colCount, rowStart, rowCount, colMagic;
};

struct ScaleArgs {
    // source: packed RGB24/BGR24 (srcKind 0) or NV12/YUV420P converted on the fly (srcKind 1)
    const uint8_t *src0, *src1, *src2;
    int ss0, ss1, ss2;
    int srcKind, srcNv12, srcBgr, srcAligned;
    int srcW, srcH, dstW, dstH;
    int chrHalf;              // chroma plane is the pairwise horizontal average (chrSrcHSubSample)
    uint8_t *dst;
    int ds, dstFormat, dstAligned;
    DevFilter hLum, hChr, vLum;   // vertical chroma filter == vLum for a non-subsampled source
    const int32_t *colStart, *colCount, *rowStart, *rowCount, *colMagic;   // device copies of ScaleTiling's
    int TH, ntx, nty, xcdRemap, chromaDirect;
    Rgb2YuvConsts r2y;
    Yuv2RgbConsts y2r;
};

int  scale_pick_tiling(const ScalePlan &p, ScaleTiling &t);
int  launch_scale_rgb(const ScaleArgs &a, const ScaleTiling &t, hipStream_t stream);
const char *scale_kernel_name(const ScaleArgs &a, const ScaleTiling &t);

// ---- generic scaler for 8-bit YUV 4:2:0 sources, packed-RGB output (k_scale_yuv.hip) ---------
// libswscale's single-context semantics: luma and chroma planes are scaled separately
// (hScale8To15_c), vertical chroma has its own filter, output through the LUT form (half chroma,
// yuv2rgb_X_c) or the full-chroma form (yuv2rgb_full_X_c).
struct YuvScaleTiling {
    int TW = 0, TH = 0, ntx = 0, nty = 0, fullChroma = 0;
    int yuvOut = 0;                               // 1: destination NV12 / YUV420P (vChr indexed by chroma row), 2: YUV444P
    int rowsL = 0, colsL = 0, rowsC = 0, colsC = 0, ldsBytes = 0, xcdRemap = 1;
    std::vector<int32_t> colStartL, colCountL, rowStartL, rowCountL, colStartC, colCountC, rowStartC, rowCountC;
    std::vector<int32_t> lumRound, chrRound;      // per output row accumulator start values
    FilterBank vChrEff;                           // vertical chroma filter after the 1-/2-tap special forms
    FilterBank vLumEff;                           // vertical luma filter after them (the 1-tap forms ignore the coefficient)
};

struct YuvScaleArgs {
    const uint8_t *y, *u, *v;
    int ys, us, vs, nv12, srcAligned;
    int srcAligned16;                             // luma and NV12 chroma rows 16-byte, planar chroma rows 8-byte aligned
    int rangeConv;                                // YUV out: 0 none, 1 limited -> full range, 2 full -> limited (swscale.c:157-188)
    int dither8;                                  // 8-bit YUV out of a source deeper than 8 bits: ff_dither_8x8_128 instead of the constant 64 (px_math.h dither_8x8_128)
    int rgbBgr, chrHalf;                          // src16 == 3: packed RGB24 / BGR24 source (y = pixels); rgbBgr: BGR order;
    Rgb2YuvConsts r2y;                            // chrHalf: chroma from pixel pairs (rgb24ToUV_half_c); rgb24ToY_c constants
    int src16, hShift, hBias;                     // P010LE / P016LE source: 10 / 16 (0 = 8-bit); hScale16To15_c's shift
                                                  // (depth - 1) and the accumulator start that undoes the P016 image bias
    int srcW, srcH, chrSrcW, chrSrcH, dstW, dstH, chrDstW;
    uint8_t *dst;                                 // packed RGB, or the Y plane for YUV output
    int ds, dstFormat, dstAligned;
    uint8_t *dstU, *dstV;                         // YUV output: chroma planes (NV12: dstU = interleaved UV)
    int dsU, dsV, dstNv12, chrDstH;
    int dst16;                                    // 10-bit output, 16-bit stores: 1 = P010LE (clip10 << 6, implies dstNv12), 2 = planar YUV420P10LE
    int dstShift;                                 //   6 / 0
    DevFilter hLum, hChr, vLum, vChr;             // vLum.round / vChr.round = lumRound / chrRound
    const int32_t *colStartL, *colCountL, *rowStartL, *rowCountL, *colStartC, *colCountC, *rowStartC, *rowCountC;
    int TH, ntx, nty, xcdRemap, fullChroma;
    int rowsL, colsL, rowsC, colsC;
    unsigned long long *prof;                     // optional per-block phase timestamps (tuning aid)
    Yuv2RgbConsts y2r;
};

int  yuvscale_prepare(const ScalePlan &p, YuvScaleTiling &t);
// frames != nullptr: nframes frames of this geometry in one launch (grid.y = frame)
int  launch_scale_yuv(const YuvScaleArgs &a, const YuvScaleTiling &t, hipStream_t stream,
                      const Yuv2xFrames *frames = nullptr, int nframes = 1);
const char *yuvscale_kernel_name(const YuvScaleTiling &t);

// ---- 16-bit destinations: 19-bit int32 lines, two passes (k_scale16.hip) -------------------------------------
// kind 0: 8-bit samples; 10 / 16: 16-bit samples (P010: >> 6); step = bytes between consecutive samples of the plane
// kind 208: 8-bit alpha samples (a << 6 | a >> 2); to15: hScale16To15_c's shift and clamp instead (the alpha lines of an 8-bit destination)
int launch_hscale19(const uint8_t *src, int srcStride, int kind, int step, int srcW, int srcH, const DevFilter &f, int32_t *dst,
                    int dstW, hipStream_t stream, int to15 = 0, int rangeConv = 0);   // rangeConv: 1 / 2 luma to / from full range, 3 / 4 chroma
// lineB == nullptr: one plane of 16-bit samples; else U / V lines -> interleaved 16-bit pairs
int launch_vscale16(const int32_t *lineA, const int32_t *lineB, int lineW, int lineH, const DevFilter &f, uint8_t *dst, int dstStride,
                    int dstW, int dstH, hipStream_t stream);

// ---- 16-bit YUV destinations in ONE launch (k_scale19.hip, round 6): the two passes above with the 19-bit lines of a tile in LDS ----
// A job = the 64-column tiles of the luma plane (one component) or of the two chroma planes side by side (two components: one staged row
// image when the source interleaves them, one dword a sample pair stored when the destination does).
constexpr int kS19TW = 64;
struct S19Job {
    int ncomp, nraw, ileave;              // components (1: the luma plane, 2: the chroma planes); source row images (1: interleaved chroma); interleaved destination
    int layout;                           // 0: 8-bit planar samples, 1: 8-bit interleaved, 2: 16-bit planar, 3: 16-bit interleaved
    int rawSel[2], rawStride[2];          // source row image i: the frame's source pointer (0 y, 1 u, 2 v); (per call) its pitch
    int rowBytes;                         // valid bytes of a source row
    int kind; unsigned xorv;              // 10: P010's >> 6; what a staged pair of 16-bit samples is XORed with (the bias v_dot2_i32_i16 needs)
    int srcW, srcH, dstW, dstH;
    int dstSel[2], dstOff[2], ds[2];      // component c: the frame's destination pointer (0 dst, 1 dstU, 2 dstV) + a byte offset; (per call) its pitch
    DevFilter h, v;
    int sh, maxv, rc;                     // hScale*To19_c's (To15_c's) shift and clamp; (per call) range conversion of the lines (hscale19_kernel's codes; 5-8: the 15-bit lines')
    int outMode, outShift, dither8;       // 0: 16-bit samples out; 1: 8-bit, 2: 10-bit << outShift (the 15-bit lines' writers); (per call) 8-bit output of a deep source: ff_dither_8x8_128
    int TW, TH, ntx, nty, nblk;           // output columns (64; 32: the half-width chroma of a packed 64-bit destination) and rows a tile, tiles across / down
    int np;                               // horizontal pairs the job keeps in registers: 4 | 8, 0 = any number (read in the loop)
    int lshift, cp2;                      // log2 of the lanes that share a staged row's units; a row has more than 64 units (a lane stages two)
    int nrMax, nrLines, PP, G, vtBytes;   // most source rows a tile's taps span (inside the plane / with the padded taps); dwords (sample pairs) a staged row; rows staged at once; bytes of a tile's vertical tables
    const int32_t *colStart;              // [ntx] first staged sample of a tile column's rows (a multiple of 4)
    const int32_t *rowStart, *rowCount;   // [nty] source rows a tile row's taps span
    int unitCoef, unitRound;              // the unit form (equal size, one-tap identity banks: scale19_unit_kernel): the vertical bank's one coefficient, the sums' start value
};
struct S19Tables {
    int ok = 0, np = 0, ldsBytes = 0;     // np: 4 | 8 horizontal pairs in registers, 0 = any number (coefficients read in the loop)
    int unit = 0;                         // every bank of both jobs a one-tap identity at equal size: the launch takes scale19_unit_kernel (s19_unit_plan)
    int rgb64 = 0, chrShift = 0, linesOff = 0;   // a packed 64-bit destination (1 RGBA64LE, 2 BGRA64LE): chroma columns = pixel columns >> chrShift; byte offset of the lines in LDS
    S19Job job[2];                        // luma, chroma (device pointers left null: the caller uploads col / row tables and fills them in)
    std::vector<int32_t> colStart[2], rowStart[2], rowCount[2];
};
struct S19Args { S19Job job[2]; int srcAl4, dstAl4, xcdRemap; int rgb64, chrShift, linesOff; Yuv2RgbConsts y2r;
                 int unit, srcAl16, dstAl16, unitBlk[2]; unsigned unitMul[2]; int unitShr[2]; };   // the unit form: planes and pitches on 16-byte addresses; blocks a frame's luma / chroma job takes; n / (units a row) = umulhi(n, unitMul) >> unitShr
static_assert(sizeof(S19Args) + sizeof(Yuv2xFrames) <= 4096, "S19Args + Yuv2xFrames exceed the kernel-argument segment");
// hl / hc / vl / vc: the 19-bit path's banks (the vertical ones after the one-tap forms' substitution); srcSemi / dstSemi: interleaved chroma
// outMode / outShift: S19Job's; hsh: the horizontal shift of the 15-bit lines (outMode != 0)
int s19_prepare(const ScalePlan &p, const FilterBank &vl, const FilterBank &vc, int bps, int kind, int srcSemi, int dstSemi, int rgb64, int chrShift, S19Tables &t,
                int outMode = 0, int outShift = 0, int hsh = 0);
int launch_scale19(const S19Args &a, int np, int ldsBytes, hipStream_t stream, const Yuv2xFrames *frames, int nframes);
// after s19_prepare: t.unit = 1 (and the jobs' unitCoef / unitRound) when the plan is an equal-size one whose four banks are one-tap identities with ONE vertical
// coefficient and start value for every row.  lumRound / chrRound: the 15-bit writers' start values a row (YuvScaleTiling's), nullptr for the 16-bit writers' constant
void s19_unit_plan(const ScalePlan &p, const FilterBank &vl, const FilterBank &vc, const int32_t *lumRound, const int32_t *chrRound, S19Tables &t);

// ---- 16-bit 4:2:0 sources into packed 8-bit RGB at equal size: the unit form of the 15-bit lines' RGB writer (unit_rgb_kernel, k_scale19.hip) -------------
// P010 / P016 / planar 10- / 16-bit sources have no unscaled converter in libswscale: identity horizontal banks, an identity vertical luma bank, the chroma's
// vertical filter, yuv2rgb_X_c's table arithmetic in closed form (px_math.h).  Host: unit_rgb_plan; per call: planes and pitches on 16-byte addresses.
struct UnitRgbArgs {
    int w, h, chrH;                       // pixels (w a multiple of 8), chroma rows of the source
    int ys, us, vs, ds;                   // pitches: luma, chroma (us only where interleaved), destination
    int semi, shr6, shl, shr, maxv;       // interleaved chroma; P010's >> 6; a line value = min((s << shl) >> shr, maxv)
    int coefL, roundL;                    // the vertical luma bank's one coefficient, the luma sums' start value
    DevFilter vChr;                       // the chroma's vertical bank (packed pairs, pos_even, round: device pointers)
    int px, bgr;                          // bytes a pixel (3 | 4); blue first
    Yuv2RgbConsts y2r;
    int blocks; unsigned rowMul; int rowShr;     // blocks a frame; idx / (units a row) = umulhi(idx, rowMul) >> rowShr
};
struct UnitRgbPlan { int ok = 0, coefL = 0, roundL = 0; };
// ok = 1 when the plan is an equal-size one with identity horizontal banks and an identity vertical luma bank of ONE start value (vl: the effective bank, lumRound: its rows')
void unit_rgb_plan(const ScalePlan &p, const FilterBank &vl, const int32_t *lumRound, int fullChroma, UnitRgbPlan &u);
int launch_unit_rgb(const UnitRgbArgs &a, hipStream_t stream, const Yuv2xFrames *frames, int nframes);

// RGBA64LE / BGRA64LE from the 19-bit lines; chrShift 1: one chroma sample per pixel pair, 0: per pixel (full chroma)
int launch_vrgba64(const int32_t *ly, const int32_t *lu, const int32_t *lv, int lumW, int lumH, int chrW, int chrH, const DevFilter &fl,
                   const DevFilter &fc, int chrShift, uint8_t *dst, int dstStride, int dstW, int dstH, int bgr, const Yuv2RgbConsts &k,
                   hipStream_t stream, const int32_t *la = nullptr);     // la: the alpha plane's 19-bit lines (lumH x lumW), nullptr = 0xFFFF

// ---- RGBA64LE / BGRA64LE sources, alpha planes (k_rgb64.hip) -------------------------------------------------------------
// packed 64-bit RGB -> Y / U / V planes of 16-bit samples (rgb64ToY_c / ToUV_c / ToUV_half_c); half: chroma from pixel pairs
int launch_rgb64_planes(const uint8_t *src, int srcStride, int w, int h, int chrW, int half, int bgr, const Rgb2YuvConsts &k,
                        uint8_t *py, int ys, uint8_t *pu, int us, uint8_t *pv, int vs, hipStream_t stream);
// 8-bit packed RGB (px = 3 or 4 bytes a pixel) -> the same planes, as rgb24ToY_c / ToUV_c / ToUV_half_c write their 16-bit lines
int launch_rgb8_planes(const uint8_t *src, int srcStride, int w, int h, int chrW, int half, int bgr, int px, const Rgb2YuvConsts &k,
                       uint8_t *py, int ys, uint8_t *pu, int us, uint8_t *pv, int vs, hipStream_t stream);
// byte 3 of every pixel of an RGBA / BGRA frame from the 15-bit alpha lines; form / first: per output row, see alpha8_out_kernel
int launch_alpha8_out(const int32_t *la, int lineW, int lineH, const DevFilter &f, const int32_t *form, const int32_t *first,
                      uint8_t *dst, int dstStride, int dstW, int dstH, hipStream_t stream);

// ---- geometric transforms and smoothing (k_transform.hip) ----------------------------------
// Frames of one launch of a transform kernel (one more grid dimension = frame): the pointers travel in the kernel-argument segment; geometry
// and strides are shared.  frames == nullptr: the one frame given by src / dst.  A launcher whose kernel for the case at hand takes no
// frame table loops over the frames itself, so every launcher accepts one.
constexpr int kOpMaxFrames = 16;
struct OpFrames { const uint8_t *src[kOpMaxFrames]; uint8_t *dst[kOpMaxFrames]; };
int launch_transpose(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride,
                     int inW, int inH, int bpp, int dir, hipStream_t stream, const OpFrames *frames = nullptr, int nframes = 1);
int launch_flip(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride,
                int w, int h, int bpp, int flipH, int flipV, hipStream_t stream, const OpFrames *frames = nullptr, int nframes = 1);
int launch_copy2d(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride,
                  int rowBytes, int h, hipStream_t stream);
int launch_conv3x3(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride,
                   int w, int h, int bpp, const int matrix[9], float rdiv, float bias, hipStream_t stream,
                   const OpFrames *frames = nullptr, int nframes = 1);
// smooth_nvcv type=gaussian in general: kw x kh (odd, <= kGaussMaxTaps), sigmaX / sigmaY (<= 0: OpenCV's default rule),
// border 0 constant, 1 replicate, 2 reflect, 3 wrap, 4 reflect101; float32 accumulation in a stated order
constexpr int kGaussMaxTaps = 255;     // both axes; the weights travel in the kernel-argument segment (2 x 255 floats); vf_median.c caps its radius at 127 too
int launch_gauss_blur(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride, int w, int h, int bpp, int kw, int kh,
                      double sigmaX, double sigmaY, int border, hipStream_t stream);
// per-channel 3x3 median, window rows / columns clamped at the edges (vf_median.c semantics at radius 1)
int launch_median3x3(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride, int w, int h, int bpp, hipStream_t stream,
                     const OpFrames *frames = nullptr, int nframes = 1);
// kw x kh (odd) in general: vf_median.c at radius (kw - 1) / 2, radiusV (kh - 1) / 2; 3 x 3 goes to the kernels above
int launch_median(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride, int w, int h, int bpp, int kw, int kh, hipStream_t stream);
int launch_rotate_flip_smooth(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride,
                              int inW, int inH, int bpp, hipStream_t stream, const OpFrames *frames = nullptr, int nframes = 1);
// arbitrary angle (radians, clockwise positive) in vf_rotate.c's 16.16 fixed point; fill == nullptr leaves
// the pixels whose source position is out of range untouched
// bilinear: 0 nearest, 1 linear, 2 cubic (Catmull-Rom, integer weights); shiftX / shiftY: translation of the rotated image in output pixels
int launch_rotate(const uint8_t *src, int srcStride, uint8_t *dst, int dstStride, int inW, int inH, int outW, int outH,
                  int bpp, double angleRad, int bilinear, const uint8_t *fill, hipStream_t stream, double shiftX = 0.0, double shiftY = 0.0,
                  const OpFrames *frames = nullptr, int nframes = 1);

} // namespace gmat

// ---- packed RGB -> YUV 4:2:0 at equal size, and NV12 <-> YUV420P (k_rgb2yuv.hip) ------------------
namespace gmat {
struct Rgb2YuvPlan {
    int ntx = 0, nty = 0, maxRows = 0;
    std::vector<int32_t> rowStart, rowCount, round;
    int stripOk = 0;                     // the vertical chroma filter is the replicated 8-tap window: rgb2yuv420s_kernel may take the frame
    int32_t vC[4] = {0, 0, 0, 0};
};
struct Rgb2YuvLaunch {
    int toJpeg = 0;                      // full-range YUV output: lum/chrRangeToJpeg_c on the 15-bit values (swscale.c:157-181)
    const uint8_t *src; int ss, bgr;
    uint8_t *y, *u, *v; int ys, us, vs, nv12;
    int w, h;
    DevFilter vChr;
    const int32_t *rowStart, *rowCount;
    int maxRows;
    Rgb2YuvConsts k;
    int stripOk = 0; int32_t vC[4] = {0, 0, 0, 0};      // from Rgb2YuvPlan
    int px = 3;                          // bytes a source pixel: 4 = RGBA / BGRA read as they are (rgb2yuv420s_kernel only)
};
int rgb2yuv_prepare(const ScalePlan &p, Rgb2YuvPlan &t);
int launch_rgb2yuv420(const Rgb2YuvLaunch &L, hipStream_t stream);
bool rgb2yuv420_strip_takes(const Rgb2YuvLaunch &L);        // the launch goes to rgb2yuv420s_kernel
// frames == nullptr: the frame of L; else nframes frames (y[] = packed source, dst / dstU / dstV = planes) of L's geometry and strides
int launch_rgb2yuv420s(const Rgb2YuvLaunch &L, hipStream_t stream, const Yuv2xFrames *frames, int nframes);
// planar float RGB (L.src = the first plane, L.ss = a float row's pitch, planes L.ss * L.h apart) -> 8-bit 4:2:0 in one kernel (round 4)
bool pf32_to_yuv420_strip_takes(const Rgb2YuvLaunch &L);
int launch_pf32_to_yuv420s(const Rgb2YuvLaunch &L, hipStream_t stream, const Yuv2xFrames *frames, int nframes);
// packed RGB24 / BGR24 -> planar YUV 4:4:4 at equal size (one-tap filters everywhere: a per-pixel conversion)
int launch_rgb2yuv444(const uint8_t *src, int ss, int bgr, uint8_t *y, int ys, uint8_t *u, int us, uint8_t *v, int vs, int w, int h,
                      const Rgb2YuvConsts &k, hipStream_t stream);
// toPlanar: (a0 = interleaved UV) -> d0 = U, d1 = V;  else (a0 = U, a1 = V) -> d0 = interleaved UV
int launch_uv_relayout(int toPlanar, const uint8_t *a0, int s0, const uint8_t *a1, int s1, uint8_t *d0, int ds0,
                       uint8_t *d1, int ds1, int cw, int ch, hipStream_t stream);
// 8 -> 16 bit, t -> t | t << 8 (planar8ToP01xleWrapper): n samples per row of plane a (b == nullptr), or n
// (a, b) sample pairs interleaved
// one plane of planarCopyWrapper's 8 -> `depth` bit copy; replicate: the luma of a full-range source (swscale_unscaled.c:1844-1862)
int launch_plane_copy_up(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int depth, int replicate, hipStream_t stream);
int launch_plane_copy_down(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int depth, int shiftonly, hipStream_t stream);
// the three planes of a frame in one launch (luma w x h, chroma cw x ch each); shiftonlyY: the luma's form (the chroma's is always the shift-only one)
int launch_planes_copy_down(const uint8_t *const src[3], const int ss[3], uint8_t *const dst[3], const int ds[3], int w, int h, int cw, int ch, int depth,
                            int shiftonlyY, hipStream_t stream);
// NV12 <-> YUV420P in one launch (luma copy + chroma (de)interleave, streaming both ways; frames: grid.z) where every plane moves in 16 / 8 bytes
bool yuv420_relayout_takes(int toPlanar, const uint8_t *y, int ys, const uint8_t *a0, int s0, const uint8_t *a1, int s1,
                           const uint8_t *dy, int dys, const uint8_t *d0, int ds0, const uint8_t *d1, int ds1);
int launch_yuv420_relayout(int toPlanar, const uint8_t *y, int ys, const uint8_t *a0, int s0, const uint8_t *a1, int s1,
                           uint8_t *dy, int dys, uint8_t *d0, int ds0, uint8_t *d1, int ds1, int w, int h, hipStream_t stream,
                           const Yuv2xFrames *frames = nullptr, int nframes = 1);
// NV12 -> P010LE / P016LE at equal size: every sample of both planes t << 8 (the generic lines' result, round 4)
int launch_nv12_shift8(const uint8_t *y, int ys, const uint8_t *uv, int uvs, uint8_t *dy, int dys, uint8_t *duv, int duvs, int w, int h, hipStream_t stream);
int launch_widen8to16(const uint8_t *a, int sa, const uint8_t *b, int sb, uint8_t *d, int ds, int n, int h,
                      hipStream_t stream);
} // namespace gmat

// ---- 2:1 specialisation of the YUV scaler (k_scale_yuv2x.hip) -------------------------------------
// Every horizontal filter row is re-expressed on the regular window [2x + w0, 2x + w0 + 10) (zero taps
// trimmed; border rows keep their folded coefficients), so the kernel needs no per-output positions:
// 16-byte pixel loads, ds_read_b64 windows shared by 4 adjacent outputs, coefficient tables in LDS.
namespace gmat {
// Interior tiles of an exact 2:1 geometry: every output column has the same regular coefficient row and every output
// row the same vertical one (borders fold differently), so a tile that touches no border needs no tables at all —
// the coefficients travel as kernel arguments (SGPRs).  Tile columns [tcLo, tcHi] x tile rows [trLo, trHi]; empty
// (tcLo > tcHi) when the geometry has no such structure.
struct Yuv2xUniform {
    int tcLo = 1, tcHi = 0, trLo = 1, trHi = 0;
    int hL[8] = {0}, hC[8] = {0};               // horizontal pairs (P used)
    int vL[8] = {0};                            // vertical luma pairs; window row of output row y: 2 * y + aL (even)
    int vC[2][4] = {{0}};                       // RGB output: vertical chroma pairs by row parity; window row: (y + aC) & ~1
    int vCy[8] = {0};                           // 4:2:0 output: vertical chroma pairs; window row of chroma row cy: 2 * cy + aCy
    int aL = 0, aC = 0, aCy = 0, lr = 0, cr = 0;   // lr / cr: accumulator start values
    int r0L = 0, dL = 0, nrL = 0, r0C = 0, dC = 0, nrC = 0;   // source row window of tile row t: r0 + t * d, n rows
};
struct Yuv2xTables {
    int ok = 0;
    Yuv2xUniform uni;
    int w0L = 0, w0C = 0;                       // regular window origins (multiples of 4)
    int ntx = 0, nty = 0;
    std::vector<int32_t> hLreg, hCreg;          // [ntx*64][5], [ntx*32][5] packed int16 pairs
    std::vector<int32_t> vrec;                  // [nty*16][12] per-output-row vertical record
    std::vector<int32_t> vrecC;                 // YUV output: [nty*8][8] per-chroma-row record (5 pairs, pos, round)
    int vLpairs = 0, vCpairs = 0, yuvOut = 0;
    int P = 5;                                  // coefficient pairs per output: window of 2*P samples (5 or 8)
};
struct Yuv2xArgs {
    const uint8_t *y, *u, *v;
    int ys, us, vs, nv12;
    int srcW, srcH, chrSrcW, chrSrcH, dstW, dstH;
    uint8_t *dst;
    int ds, dstFormat, dstAligned;
    uint8_t *dstU, *dstV;                       // YUV output (yuvOut): chroma planes, NV12: dstU = interleaved UV
    int dsU, dsV, dstNv12, yuvOut, chrDstW, chrDstH;
    int P;                                      // 5 or 8 (Yuv2xTables::P)
    const int32_t *vrecC;
    const int32_t *hLreg, *hCreg, *vrec;
    int w0L, w0C, vLpairs, vCpairs;
    const int32_t *rowStartL, *rowCountL, *rowStartC, *rowCountC;
    int ntx, nty, xcdRemap;
    unsigned long long *prof;
    Yuv2RgbConsts y2r;
    Yuv2xUniform uni;
};
int  yuv2x_prepare(const ScalePlan &p, const YuvScaleTiling &generic, Yuv2xTables &t);
// frames == nullptr: the one frame described by `a`
int  launch_scale_yuv2x(const Yuv2xArgs &a, int rowsL, int rowsC, int ldsBytes, hipStream_t stream,
                        const Yuv2xFrames *frames = nullptr, int nframes = 1);
// ---- strip-walking form of the exact 2:1 YUV 4:2:0 -> packed RGB scaler (k_scale_yuv2s.hip) -----------------------
// A wave walks down a strip of 256 output columns with the horizontally filtered rows it still needs in registers; the
// borders are the interior filter on an edge-replicated frame (checked coefficient by coefficient on the host), so
// every coefficient is a kernel argument and there are no tables.
struct Yuv2sTables {
    int ok = 0;
    int np = 4;                                        // coefficient pairs per filter: 4 (8 taps) or 6 (Lanczos-3)
    int32_t hL[6] = {0}, hC[6] = {0}, vL[6] = {0};   // int16 pairs on the odd-aligned window [2x - (np - 1), 2x + np]
    int lr = 0;                                        // vertical luma accumulator start
};
struct Yuv2sArgs {
    int ys, us, vs, nv12;
    int srcW, srcH, chrSrcW, chrSrcH, dstW, dstH;
    int ds, dstFormat;
    int np;                                     // coefficient pairs per filter (4 | 6)
    int32_t hL[6], hC[6], vL[6];
    int lr;
    int segRows, nseg, nsg, xcdRemap;           // filled by the launcher: rows per strip segment, segments, groups of 4 strips per row
    int updown;                                 // filled by the launcher: odd segments walk upward (the 4-pair kernel)
    Yuv2RgbConsts y2r;
};
int  yuv2s_prepare(const ScalePlan &p, const YuvScaleTiling &generic, Yuv2sTables &t);
// plane pointers of the nframes frames in *frames (grid.y = frame)
int  launch_scale_yuv2s(const Yuv2sArgs &a, hipStream_t stream, const Yuv2xFrames *frames, int nframes);
// the form that launch takes: true = the block-cooperative kernel of small launches (scale_yuv2s_blk_kernel), false = the walker
bool yuv2s_block_form(const Yuv2sArgs &a, int nframes);

// ---- the polyphase band walker for ANY ratio (k_scale_yuvg.hip): 8-bit 4:2:0 -> packed RGB, and -> 4:2:0 of the same chroma layout ----
// The vertical program of a plane class (pair) in one walking direction; see build_qprog in k_scale_yuvg.hip.
struct YuvGQProg {
    int K = 0, Q = 0, rows = 0, tapsL = 0, tapsC = 0;
    std::vector<int32_t> prog, qfirst, qdone;                 // [(Q + 1) * (2 + 3 K)] [rows] [rows]
    std::vector<int> yBase, posL, posC;                       // (host only)
    std::vector<int16_t> cfL, cfC;                            // (host only) taps in walking order
};
struct YuvGTables {
    int ok = 0, P = 0, K = 0, yuvOut = 0, roundL = 0, roundC = 0;
    std::vector<int32_t> hL, hC, posL, posC;                  // [dstW][P] / [chrDstW][P] packed coefficient pairs, window starts
    YuvGQProg rgb[2];                                         // RGB destination: luma + chroma; [0] walking down, [1] walking up (mirrored)
    YuvGQProg pl[2], pc[2];                                   // 4:2:0 destination: the luma plane / a chroma plane on its own
    // the block-cooperative form (a launch of few frames): per output row [first row pair, last row pair, 4 n4 coefficient pairs on the
    // row pairs from the first one on]; blkRows: the tallest band (output rows, a multiple of 4) whose row pairs fit a block at ANY start row
    std::vector<int32_t> vtL, vtC;
    int n4L = 0, n4C = 0, blkRows = 0, blkRowsC = 0;
    // packed RGB -> packed RGB, block-cooperative (scale_yuvg_rgbsrc_blk_kernel): the walker's form available at all (K), pixels a lane converts, the
    // tallest band on four row pairs a wave (blkRows: on eight)
    int walkOk = 0, blkPPL = 0, blkRows4 = 0;
    std::vector<int32_t> vtRnd;                               // ... and the sums' start by output row (packed_vscale's one- and two-tap forms)
    // packed RGB -> 4:2:0, block-cooperative and FUSED (scale_yuvg_rgb2p_blk_kernel: luma and chroma of a band behind one load of the pixels): the chroma's
    // coefficient pairs on 8-byte aligned windows of a PLANE's line whatever the destination's layout, pixels a lane converts (0: no instance), and per band
    // height 4 i the most row pairs a band needs (luma and chroma windows together)
    std::vector<int32_t> hCp;
    int f2PPL = 0, f2Pairs[17] = {0};
};
struct YuvGArgs {
    int ys, us, vs, nv12;
    int srcW, srcH, chrSrcW, chrSrcH, dstW, dstH, chrDstW, chrDstH;
    int ds, dsU, dsV, dstFormat, yuvOut;
    int P, K, roundL, roundC;
    const int32_t *hL, *hC, *posL, *posC;
    const int32_t *prog[2], *qfirst[2], *qdone[2];            // RGB destination, or the luma job of a 4:2:0 destination
    const int32_t *progC[2], *qfirstC[2], *qdoneC[2];         // the chroma jobs of a 4:2:0 destination
    // filled by the launcher: rows per band, 4-strip groups per row, blocks (luma | chroma jobs of a 4:2:0 destination)
    int bandRows, nbands, nsg, bandRowsC, nbandsC, nsgC, nblkL, nblkC, nblk, xcdRemap, updown, bandStep, bandStepC;
    // the block-cooperative form: vertical tables by output row, groups of 4 coefficient pairs a row, the tallest bands that fit
    const int32_t *vtL, *vtC;
    int n4L, n4C, blkRows, blkRowsC;
    const int32_t *hCp; int f2PPL, f2Pairs[17];               // scale_yuvg_rgb2p_blk_kernel (YuvGTables')
    int blkPPL, blkRows4, blkSlots; const int32_t *vtRnd;     // scale_yuvg_rgbsrc_blk_kernel: pixels a lane, the tallest band on four pairs a wave; (launcher) LDS pair slots a line
    int srcPx, srcAlpha;                                      // ... (per call) 4: the source is RGBA / BGRA pixels read as they are; its alpha channel is a fourth line
    // (round 5) 16-bit samples in (k_scale_yuvg16.hip: YuvScaleArgs' kind, hScale16To15_c's shift, the sums' start), 10-bit samples out, and the
    // ordered dither of 8-bit planar output of a deeper source (YuvScaleArgs')
    int src16, hShift, hBias, dst16, dstShift, dither8;
    Rgb2YuvConsts r2y; int rgbBgr;                            // src16 == 3: a packed RGB24 / BGR24 source (the walker's own converter, GStream LK)
    Yuv2RgbConsts y2r;
};
// both structs travel by value in one launch's kernel-argument segment (4 KB)
static_assert(sizeof(YuvGArgs) + sizeof(Yuv2xFrames) <= 4096, "YuvGArgs + Yuv2xFrames exceed the kernel-argument segment");
int  yuvg_prepare(const ScalePlan &p, const YuvScaleTiling &generic, YuvGTables &t);
int  launch_scale_yuvg(const YuvGArgs &a, hipStream_t stream, const Yuv2xFrames *frames, int nframes);
bool yuvg_block_form(const YuvGArgs &a, int nframes);         // whether a launch of nframes frames takes scale_yuvg_blk_*_kernel
// the same walker over 16-bit samples (k_scale_yuvg16.hip = k_scale_yuvg.hip compiled with G_BPS = 2): P010LE / P016LE / planar 10- and 16-bit sources
int  yuvg_prepare16(const ScalePlan &p, const YuvScaleTiling &generic, YuvGTables &t);
int  launch_scale_yuvg16(const YuvGArgs &a, hipStream_t stream, const Yuv2xFrames *frames, int nframes);
bool yuvg_block_form16(const YuvGArgs &a, int nframes);
bool yuvg_rgb2p_fused(const YuvGArgs &a, int nframes);        // a packed RGB source into a 4:2:0 frame: whether the launch takes scale_yuvg_rgb2p_blk_kernel
// packed RGB24 / BGR24 -> packed RGB at the walker's ratios (scale_yuvg_rgbsrc_kernel in k_scale_yuvg16.hip): p = the RGB scaler's plan
int  yuvg_rgbsrc_prepare(const ScalePlan &p, YuvGTables &t);
int  launch_scale_yuvg_rgbsrc(const YuvGArgs &a, hipStream_t stream, const Yuv2xFrames *frames, int nframes);
bool yuvg_rgbsrc_block_form(const YuvGArgs &a, int nframes);  // whether a launch of nframes frames takes scale_yuvg_rgbsrc_blk_kernel (a.K == 0: always)

// ---- the quad-lane polyphase walker (k_scale_yuvu.hip, round 4): a lane owns FOUR adjacent outputs of a row, the vertical filter is a
// GATHER over a short register ring of horizontally filtered row pairs (coefficients by output row, relative to the newest pair) — no
// running sums, so the number of output rows open at once does not matter: UP-scales of any factor (and down-scales up to 2 : 1) ------
struct YuvUTables {
    int ok = 0, P = 0, SD = 1, yuvOut = 0, roundL = 0, roundC = 0;
    int RL = 0, RC = 0;                                       // ring depth: 4:2:0 destination: luma job / chroma jobs; RGB: luma / chroma stream
    int lead = 0;                                             // RGB: steps the luma stream runs behind the chroma stream
    std::vector<int32_t> hL, hC, posL, posC;                  // [dstW][P] / [chrDstW][P] packed coefficient pairs, window starts
    // a STEP is a row pair of the plane (4:2:0 destination) or a quad = two luma row pairs + the chroma row pair beside them (RGB).
    // vt*: per output row the coefficient pairs on the ring, newest pair first, as of the step that completes the row (RGB: vtL holds
    // RL luma + RC chroma pairs a row); end*[s]: output rows complete after step s; first* / last*: a row's first / last step
    std::vector<int32_t> vtL, vtC, endL, endC, firstL, firstC, lastL, lastC;
};
struct YuvUArgs {
    int ys, us, vs, nv12;
    int srcW, srcH, chrSrcW, chrSrcH, dstW, dstH, chrDstW, chrDstH;
    int ds, dsU, dsV, dstFormat, yuvOut;
    int P, SD, RL, RC, lead, roundL, roundC;
    const int32_t *hL, *hC, *posL, *posC;
    const int32_t *vtL, *vtC, *endL, *endC, *firstL, *firstC, *lastL, *lastC;
    // filled by the launcher: rows per band, groups of four 256-byte strips per row, blocks (luma | chroma jobs of a 4:2:0 destination)
    int bandRows, nbands, nsg, bandRowsC, nbandsC, nsgC, nblkL, nblkC, nblk, xcdRemap;
    int src16, hShift, hBias, dst16, dstShift, dither8;       // (round 5) 16-bit samples in (k_scale_yuvu16.hip), 10-bit samples out, the dither of a deeper source: as YuvGArgs'
    Yuv2RgbConsts y2r;
};
int  yuvu_prepare(const ScalePlan &p, const YuvScaleTiling &generic, YuvUTables &t);
int  launch_scale_yuvu(const YuvUArgs &a, hipStream_t stream, const Yuv2xFrames *frames, int nframes);
// the same walker over 16-bit samples (k_scale_yuvu16.hip = k_scale_yuvu.hip compiled with U_BPS = 2)
int  yuvu_prepare16(const ScalePlan &p, const YuvScaleTiling &generic, YuvUTables &t);
int  launch_scale_yuvu16(const YuvUArgs &a, hipStream_t stream, const Yuv2xFrames *frames, int nframes);

// ---- the LINES form (k_scale_yuvl.hip, round 4): two launches through a frame of horizontally filtered 15-bit lines in HBM — the down-scales
// no walker takes (beyond 6.1 : 1, range conversion, filters the band walker's tables do not hold), which the tiled kernel served at 0.03 - 0.1 of
// the roofline and refused beyond ~ 20 : 1.  8-bit YUV sources (NV12, YUV420P, YUV444P) -> packed RGB (half / full chroma), 8-bit 4:2:0, YUV444P.
struct YuvLTables {
    int ok = 0, P = 0, yuvOut = 0, fullChroma = 0;
    int nld = 0;                                              // 1 KB pieces of a source row a wave's 64 windows span, at most
    int wide = 0;                                             // 16-bit samples (scale_yuvl_h16_kernel)
    int dot4L = 0, dot4C = 0;                                 // a byte plane's table in the signed-byte form (k_scale_yuvl.hip): [P + 1][pitch]
    int RW = 1;                                               // dwords a lane reads from a byte plane's row image at once (the windows' alignment / 4)
    std::vector<int32_t> hL, hC;                              // [dstW][P] / [chrDstW][P] coefficient pairs on the window that starts at off*
    std::vector<int32_t> offL, offC;                          // byte offset of a column's window in its source row (a multiple of 4)
    int pitchL = 0, pitchC = 0, pairRowsL = 0, pairRowsC = 0; // the lines frame: dwords a row (a dword = rows 2p and 2p + 1 of one column), rows
    size_t baseU = 0, baseV = 0, frameInts = 0;               // dword offsets of the U / V lines, dwords a frame
};
struct YuvLArgs {
    int ys, us, vs, nv12;
    int srcW, srcH, chrSrcW, chrSrcH, dstW, dstH, chrDstW, chrDstH;
    int ds, dsU, dsV, dstFormat, dstAligned, dstNv12, yuvOut, fullChroma, rangeConv;
    int src16, hShift, hBias;                                 // 16-bit samples: YuvScaleArgs' kind (10 | 16 semi-planar, 17 | 18 planar), hScale16To15_c's shift, accumulator start
    int dst16, dstShift, dither8;                             // 10-bit 4:2:0 destinations; 8-bit planar output of a deeper source (YuvScaleArgs')
    int P, nld, RW, dot4L, dot4C;
    const int32_t *hL, *hC, *offL, *offC;
    int pitchL, pitchC, pairRowsL, pairRowsC;
    size_t baseU, baseV, frameInts;
    int32_t *inter;                                           // nframes lines frames
    DevFilter vLum, vChr;                                     // the tiled kernel's: effective taps, per-row start values
    // filled by the launcher
    int nColL, nColC, rp, nItemL, nItemC, nItem, nColV;
    Yuv2RgbConsts y2r;
};
int  yuvl_prepare(const ScalePlan &p, const YuvScaleTiling &generic, YuvLTables &t);
int  launch_scale_yuvl(const YuvLArgs &a, hipStream_t stream, const Yuv2xFrames *frames, int nframes);

// ---- strip-walking form of the exact 2:1 YUV 4:2:0 -> YUV 4:2:0 scaler (k_scale_yuv2p.hip): NV12 -> NV12 and
// YUV420P -> YUV420P, every plane walked on its own -------------------------------------------------------------------
struct Yuv2pTables {
    int ok = 0, srcDepth = 8, dstDepth = 8;                         // 10: P010LE / YUV420P10LE on that side
    int ok444 = 0;                                                  // 8-bit 4:2:0 -> YUV444P at 2:1: the luma walker alone (the chroma filters are the identity: a re-layout)
    int cross = 0, snv = 0;                                         // mixed chroma layouts (interleaved <-> planar); source interleaved
    int np = 4;                                                     // coefficient pairs per filter: 4 (8 taps) or 6 (Lanczos-3)
    int32_t hL[6] = {0}, hC[6] = {0}, vL[6] = {0}, vC[6] = {0};   // int16 pairs on the odd-aligned window [2x - (np - 1), 2x + np]
    int lr = 0, cr = 0;                                             // vertical accumulator start values (dither << 12)
};
struct Yuv2pArgs {
    int ys, us, vs, nv12;                        // nv12: interleaved chroma (NV12, P010LE)
    int srcDepth, dstDepth;                      // 8 or 10 bits per sample on each side
    int cross;                                   // the destination's chroma layout is the other one (nv12: the SOURCE's is interleaved)
    int lumaOnly;                                // no chroma workgroups (YUV444P destinations: the caller re-lays the chroma out)
    int srcW, srcH, chrSrcW, chrSrcH, dstW, dstH, chrDstW, chrDstH;
    int ds, dsU, dsV;
    int np;                                      // coefficient pairs per filter (4 | 6)
    int32_t hL[6], hC[6], vL[6], vC[6];
    int lr, cr;
    int dither8;                                 // see YuvScaleArgs::dither8 (the <10to8> instances only)
    // filled by the launcher: rows per strip segment, segments and groups of 4 strips per plane kind, workgroup counts
    int segRowsL, nsegL, nsgL, segRowsC, nsegC, nsgC, nblkL, nblk, xcdRemap;
    int updown; int32_t vLup[6], vCup[6];        // odd segments walk upward: the vertical pairs of the mirrored plane
};
int  yuv2p_prepare(const ScalePlan &p, const YuvScaleTiling &generic, Yuv2pTables &t);
int  launch_scale_yuv2p(const Yuv2pArgs &a, hipStream_t stream, const Yuv2xFrames *frames, int nframes);

// ---- strip-walking form of the exact 1:2 UP-scale of 8-bit YUV 4:2:0 (k_scale_yuv1x2.hip): NV12 -> NV12, YUV420P -> YUV420P ------
// per axis: A / B = the two phases (even / odd outputs) as 2 int16 pairs, S0 / S2 = the table's own rows of outputs 0 and 2
struct Yuv1x2Tables {
    int ok = 0;
    int32_t hLA[2] = {0}, hLB[2] = {0}, hLS0[2] = {0}, hLS2[2] = {0}, vLA[2] = {0}, vLB[2] = {0}, vLS0[2] = {0}, vLS2[2] = {0};
    int32_t hCA[2] = {0}, hCB[2] = {0}, hCS0[2] = {0}, hCS2[2] = {0}, vCA[2] = {0}, vCB[2] = {0}, vCS0[2] = {0}, vCS2[2] = {0};
    int lr = 0, cr = 0;
};
struct Yuv1x2Args {
    int ys, us, vs, nv12;
    int srcW, srcH, chrSrcW, chrSrcH;            // the destination is exactly twice as large in both directions
    int ds, dsU, dsV;
    int32_t hLA[2], hLB[2], hLS0[2], hLS2[2], vLA[2], vLB[2], vLS0[2], vLS2[2];
    int32_t hCA[2], hCB[2], hCS0[2], hCS2[2], vCA[2], vCB[2], vCS0[2], vCS2[2];
    int lr, cr;
    int segRowsL, nsegL, nsgL, segRowsC, nsegC, nsgC, nblkL, nblk, xcdRemap;     // filled by the launcher (rows = OUTPUT rows)
};
int  yuv1x2_prepare(const ScalePlan &p, const YuvScaleTiling &generic, Yuv1x2Tables &t);
int  launch_scale_yuv1x2(const Yuv1x2Args &a, hipStream_t stream, const Yuv2xFrames *frames, int nframes);

// ---- strip-walking form of the exact 3:1 down-scale of 8-bit YUV 4:2:0 (k_scale_yuv3x1.hip): NV12 -> NV12, YUV420P -> YUV420P -----
// per filter: the 11 taps on the window [3x - 4, 3x + 7] as 6 int16 pairs (the 12th coefficient is 0)
struct Yuv3x1Tables {
    int ok = 0;
    int32_t hL[6] = {0}, hC[6] = {0}, vL[6] = {0}, vC[6] = {0};
    int lr = 0, cr = 0;
};
struct Yuv3x1Args {
    int ys, us, vs, nv12;
    int dstW, dstH, chrDstW, chrDstH;            // the source is exactly three times as large in both directions
    int ds, dsU, dsV;
    int32_t hL[6], hC[6], vL[6], vC[6];
    int lr, cr;
    int segRowsL, nsegL, nsgL, segRowsC, nsegC, nsgC, nblkL, nblk, xcdRemap;     // filled by the launcher (nsg = strips per row here)
    int updown;                                  // 1: odd segments walk upward (shared boundary rows meet in L2)
};
int  yuv3x1_prepare(const ScalePlan &p, const YuvScaleTiling &generic, Yuv3x1Tables &t);

// scale_yuv3r_kernel (k_scale_yuv3x1.hip): NV12 at exactly a third of the size into packed RGB, ONE libswscale context.  Luma 3:1 on
// both axes (11 taps), chroma 3:1 horizontally (RGB destinations keep half-width chroma) and 3:2 vertically (6 taps, two phases,
// output row 1 with its own table row: down32_axis in k_scale_yuv3x2.hip)
struct Yuv3rTables {
    int ok = 0;
    int32_t hL[6] = {0}, hC[6] = {0}, vL[6] = {0};      // 11 taps on [3x - 4, 3x + 7] as int16 pairs
    int32_t vCA[3] = {0}, vCB[3] = {0}, vCS[3] = {0};    // vertical chroma: even / odd output rows, and output row 1
    int lr = 0, cr = 0;
};
struct Yuv3rArgs {
    int ys, us, dstW, dstH, ds, dstFormat;
    int32_t hL[6], hC[6];
    int32_t vP[4], vS[4];                                // vertical luma: pairs (c8 c9) (c5 c6) (c2 c3) (0 c0) for rows (3u-2 | 3u-1), singles c10 c7 c4 c1 for row 3u
    int32_t cA[6], cB[6], cS[6];                         // vertical chroma taps
    int lr, cr;
    int segRows, nseg, nstrips, nblk, xcdRemap;          // filled by the launcher
    Yuv2RgbConsts y2r;
};
bool down32_axis(const FilterBank &fb, int srcLen, int32_t (&A)[3], int32_t (&B)[3], int32_t (&S)[3]);
int  yuv3r_prepare(const ScalePlan &p, const YuvScaleTiling &generic, Yuv3rTables &t);
int  launch_scale_yuv3r(const Yuv3rArgs &a, hipStream_t stream, const Yuv2xFrames *frames, int nframes);
int  launch_scale_yuv3x1(const Yuv3x1Args &a, hipStream_t stream, const Yuv2xFrames *frames, int nframes);

// ---- strip-walking form of the exact 3:2 down-scale of 8-bit YUV 4:2:0 (k_scale_yuv3x2.hip): NV12 -> NV12, YUV420P -> YUV420P -----
// per axis and plane kind: A / B = the two phases (even / odd outputs) as 3 int16 pairs, S = the table's own row of output 1
struct Yuv3x2Tables {
    int ok = 0;
    int32_t hLA[3] = {0}, hLB[3] = {0}, hLS[3] = {0}, vLA[3] = {0}, vLB[3] = {0}, vLS[3] = {0};
    int32_t hCA[3] = {0}, hCB[3] = {0}, hCS[3] = {0}, vCA[3] = {0}, vCB[3] = {0}, vCS[3] = {0};
    int lr = 0, cr = 0;
};
struct Yuv3x2Args {
    int ys, us, vs, nv12;
    int dstW, dstH, chrDstW, chrDstH;            // the source is exactly 3/2 as large in both directions
    int ds, dsU, dsV;
    int32_t hLA[3], hLB[3], hLS[3], vLA[3], vLB[3], vLS[3];
    int32_t hCA[3], hCB[3], hCS[3], vCA[3], vCB[3], vCS[3];
    int lr, cr;
    int segRowsL, nsegL, nsgL, segRowsC, nsegC, nsgC, nblkL, nblk, xcdRemap;     // filled by the launcher (nsg = strips per row)
};
int  yuv3x2_prepare(const ScalePlan &p, const YuvScaleTiling &generic, Yuv3x2Tables &t);
int  launch_scale_yuv3x2(const Yuv3x2Args &a, hipStream_t stream, const Yuv2xFrames *frames, int nframes);

// scale_yuv4r_kernel (k_scale_yuv4r.hip): NV12 at exactly a quarter of the size into packed RGB, ONE libswscale context.  Luma 4:1 on
// both axes and chroma 4:1 horizontally: 16 taps on [4x - 6, 4x + 9] as 8 int16 pairs; chroma 2:1 vertically: 8 taps on [2y - 3, 2y + 4]
struct Yuv4rTables {
    int ok = 0;
    int32_t hL[8] = {0}, hC[8] = {0}, vL[8] = {0}, vC[4] = {0};
    int lr = 0, cr = 0;
};
struct Yuv4rArgs {
    int ys, us, vs, nv12, dstW, dstH, ds, dstFormat;
    int32_t hL[8], hC[8], vL[8], vC[4];
    int lr, cr;
    int segRows, nseg, nstrips, nblk, xcdRemap, updown;   // filled by the launcher; updown: odd segments walk upward
    Yuv2RgbConsts y2r;
};
int  yuv4r_prepare(const ScalePlan &p, const YuvScaleTiling &generic, Yuv4rTables &t);
// scale_yuv4x1_kernel (k_scale_yuv4r.hip): the 4:1 down-scale of 8-bit 4:2:0 into the same layout (4K -> 540p, 1080p -> 270p), every plane
// walked on its own: 16 taps on [4x - 6, 4x + 9] on every axis
struct Yuv4x1Tables {
    int ok = 0;
    int32_t hL[8] = {0}, hC[8] = {0}, vL[8] = {0}, vC[8] = {0};
    int lr = 0, cr = 0;
};
struct Yuv4x1Args {
    int ys, us, vs, nv12;
    int dstW, dstH, chrDstW, chrDstH;
    int ds, dsU, dsV;
    int32_t hL[8], hC[8], vL[8], vC[8];
    int lr, cr;
    int segRows, nsegL, nsgL, nsegC, nsgC, nblkL, nblk, xcdRemap, updown;     // filled by the launcher (nsg = strips per row)
};
int  yuv4x1_prepare(const ScalePlan &p, const YuvScaleTiling &generic, Yuv4x1Tables &t);
int  launch_scale_yuv4x1(const Yuv4x1Args &a, hipStream_t stream, const Yuv2xFrames *frames, int nframes);
int  launch_scale_yuv4r(const Yuv4rArgs &a, hipStream_t stream, const Yuv2xFrames *frames, int nframes);

// scale_yuv32r_kernel (k_scale_yuv3x2.hip): NV12 at exactly two thirds of the size into packed RGB (1080p -> 720p, 4K -> 1440p), ONE
// libswscale context.  Luma 3:2 on both axes, chroma 3:2 horizontally (half-width chroma at the output) and 3:4 UP vertically: 4 taps,
// four phases, output rows 0 and 1 with their own table rows
struct Yuv32rTables {
    int ok = 0;
    int32_t hLA[3] = {0}, hLB[3] = {0}, hLS[3] = {0}, hCA[3] = {0}, hCB[3] = {0}, hCS[3] = {0};   // int16 pairs (down32_axis)
    int32_t vLA[3] = {0}, vLB[3] = {0}, vLS[3] = {0};
    int32_t cP[4][4] = {{0}}, cS0[4] = {0}, cS1[4] = {0};   // vertical chroma taps on the window of output row 4k + 2 + i: rows 3k + (0, 1, 1, 2)[i] .. + 3
    int lr = 0, cr = 0;
};
struct Yuv32rArgs {
    int ys, us, dstW, dstH, ds, dstFormat;
    int32_t hLA[3], hLB[3], hLS[3], hCA[3], hCB[3], hCS[3];
    int32_t lA[6], lB[6], lS[6];                          // vertical luma taps
    int32_t cP[4][4], cS0[4], cS1[4];
    int lr, cr;
    int segRows, nseg, nstrips, nblk, xcdRemap;           // filled by the launcher
    Yuv2RgbConsts y2r;
};
int  yuv32r_prepare(const ScalePlan &p, const YuvScaleTiling &generic, Yuv32rTables &t);
int  launch_scale_yuv32r(const Yuv32rArgs &a, hipStream_t stream, const Yuv2xFrames *frames, int nframes);

// ---- strip-walking form of the exact 2:1 packed RGB -> packed RGB scaler (k_scale_rgb2s.hip) -----------------------
// rgb24 / bgr24 at 2W x 2H -> rgb24 / bgr24 / rgba / bgra at W x H, one libswscale context's arithmetic.
struct Rgb2sTables {
    int ok = 0;
    int32_t hL[4] = {0}, vL[4] = {0};           // int16 pairs on the odd-aligned window [2x - 3, 2x + 4]
};
struct Rgb2sArgs {
    int ss, srcW, srcH, dstW, dstH, ds, dstFormat;
    int srcKind, us, vs;                        // 0 packed RGB (ss = its stride); 1 NV12, 2 YUV420P: ss / us / vs = plane strides (scale_rgb2h_kernel only)
    int32_t hL[4], vL[4];
    int rnd;                                    // vertical accumulator start (1 << 9)
    int32_t cY01, cY2, cU01, cU2, cV01, cV2;    // rgb -> yuv coefficients as (first, second) int16 pair and third, in byte order
    int segRows, nseg, nsg, xcdRemap;           // filled by the launcher
    Yuv2RgbConsts y2r;
};
// scale_rgb2y_kernel: packed RGB24 / BGR24 at 2:1 into an 8-bit 4:2:0 frame (NV12 / YUV420P), one libswscale context
struct Rgb2yTables {
    int ok = 0;
    int32_t hL[4] = {0}, hC[4] = {0}, vL[4] = {0};   // int16 pairs on the odd-aligned window [2x - 3, 2x + 4] (luma; chroma on pixel PAIRS)
    int32_t vE[9] = {0};                             // the 16 vertical chroma taps on rows [4c - 6, 4c + 9] as pairs (row 2m - 1, row 2m), m = 2c - 3 .. 2c + 5
};
struct Rgb2yArgs {
    int ss, srcW, srcH, dstW, dstH, ys, us, vs, nv12;
    int32_t hL[4], hC[4], vL[4], vE[9];
    int rnd;                                         // 64 << 12: the dither of yuv2planeX_8_c / yuv2nv12cX_c
    int32_t cY01, cY2, cU01, cU2, cV01, cV2;         // rgb -> yuv coefficients as (first, second) int16 pair and third, in byte order
    int segRowsC, nseg, nstrips, nblk, xcdRemap;     // filled by the launcher: chroma rows per segment
};
int  rgb2y_prepare(const ScalePlan &p, Rgb2yTables &t);
// frames->y[] = packed source frames, frames->dst / dstU / dstV = the planes (grid.y = frame)
int  launch_scale_rgb2y(const Rgb2yArgs &a, hipStream_t stream, const Yuv2xFrames *frames, int nframes);
bool filter_is_edge_replication(const FilterBank &fb, int srcLen, int32_t (&pairs)[4]);
bool filter_is_edge_replication_np(const FilterBank &fb, int srcLen, int NP, int32_t *pairs);      // NP pairs on [2x - (NP - 1), 2x + NP]
bool filter_is_edge_replication_ratio(const FilterBank &fb, int srcLen, int R, int L, int NP, int32_t *pairs);   // R:1, window from R x - L
int  rgb2s_prepare(const ScalePlan &p, Rgb2sTables &t);
// frames->y[] = source frames, frames->dst[] = destination frames (grid.y = frame)
int  launch_scale_rgb2s(const Rgb2sArgs &a, hipStream_t stream, const Yuv2xFrames *frames, int nframes);
bool rgb2h_takes_yuv();                     // the fused convert-then-scale form rides on scale_rgb2h_kernel<.., yuv>
const char *rgb2s_kernel_name();            // scale_rgb2h_kernel (converted samples shared between lanes) unless GMAT_RGB2_SHARED=0
} // namespace gmat
