/*
 * gmat_metrans.h — the MeTrans colour-space, bit-depth and resize entry points (metrans/include/NvCodec/NvCommon.h:232-255), all 17,
 * served by the kernels behind gmat_hip.h.  Paths below are relative to /root/reference/metrans/include/NvCodec/.
 *
 * C++ linkage on purpose: the reference declares these as plain C++ functions, so an application object file references their
 * MANGLED names.  With cudaStream_t spelled as a pointer to `struct CUstream_st` (its real definition) the symbols exported here
 * mangle identically, e.g. _Z12Nv12ToBgra32PhiS_iiiiP11CUstream_st, and MeTrans links against libgmat_hip.so unchanged.
 *
 * Layout and argument meaning are the reference's: one NV12 / P016 allocation, chroma plane at base + pitch * height
 * (ColorSpace.cu:146-148, Resize.cu:41, Resize_bicubic.cu:150-152); planar outputs are three stacked planes, plane k at
 * base + k * pitch * height (ColorSpace.cu:183-194: pDst += nRgbpPitch * nHeight), in the order the name says; iMatrix = ColorSpaceStandard (ColorSpace.cu:16-27):
 * 1 BT.709, 4 FCC, 5 BT.470, 6 BT.601, 7 SMPTE 240M, 9 / 10 BT.2020, anything else BT.709 (GetConstants' default, ColorSpace.cu:32-64 —
 * app/FrameExtractor.h:225 passes 0).
 * Streams as the reference launches them: the FloatPlanar converters run on `stream` (ColorSpace.cu:273-292); every other entry
 * point runs on the NULL stream whatever `stream` is (ColorSpace.cu:219-271,345-350: launches without a stream argument;
 * Resize.cu:68,159, BitDepth.cu:32,36).
 *
 * Arithmetic:
 *   - colour conversion: libswscale's fixed point (limited-range YUV), not the reference's float matrix with truncation
 *     (YuvToRgbForPixel, ColorSpace.cu:105-135) — BASELINE's parity target for every yuv <-> rgb conversion of this library:
 *       Nv12ToBgra32 / Nv12ToRgba32 and the NV12 planar forms: yuv2rgb.c's nearest-chroma tables (bit-exact with yuv2rgb_c_32);
 *       the planar forms carry the SAME 8-bit values as the packed ones, plane by plane; float = value / 255.0f (ToValue, :157-163);
 *       Nv12ToBgra64, P016ToBgra32 / 64, Bgra64ToP016: what one libswscale context computes for nv12 -> bgra64le, p016le -> bgra /
 *       bgra64le, bgra64le -> p016le at equal size (the generic path: 15- / 19-bit lines);
 *       P016ToBgrPlanar / P016ToBgrFloatPlanar: the planes of P016ToBgra32's pixels;
 *   - ScaleNv12 / ScaleP016: SWS_BILINEAR of one libswscale context (the reference samples a hardware texture unit in linear mode,
 *     Resize.cu:15-74, whose 9-bit weight arithmetic is not in the tree);
 *   - ScaleNv12_Bicubic: THE REFERENCE'S OWN KERNEL (Resize_bicubic.cu:83-159): float 4 x 4, a = -0.5, coordinates x * scale clamped
 *     to [2, n - 2], no anti-alias widening, truncating cast; within +-1 LSB of the reference (fma contraction of nvcc), bit-exact
 *     with the restatement in oracle/orc_metrans.c.  gmat_metrans_bicubic_mode(1) selects libswscale's SWS_BICUBIC instead
 *     (B = 0, C = 0.6, anti-aliased on down-scales: rounds 1-3's behaviour);
 *   - ConvertUInt8ToUInt16 / ConvertUInt16ToUInt8: v << 8 / v >> 8 (BitDepth.cu:15-29), exact.
 * Errors: the reference's functions return void; a refused call logs through gmat_set_log_callback and writes nothing.
 */
#ifndef GMAT_METRANS_H
#define GMAT_METRANS_H
#include <stdint.h>

#ifdef __cplusplus
struct CUstream_st;
typedef struct CUstream_st *cudaStream_t;      /* carries a hipStream_t */

#define GMAT_MT_API __attribute__((visibility("default")))

GMAT_MT_API void Nv12ToBgra32(uint8_t *dpNv12, int nNv12Pitch, uint8_t *dpBgra, int nBgraPitch, int nWidth, int nHeight,
                              int iMatrix, cudaStream_t stream);                                  /* NvCommon.h:232, ColorSpace.cu:219 */
GMAT_MT_API void Nv12ToRgba32(uint8_t *dpNv12, int nNv12Pitch, uint8_t *dpBgra, int nBgraPitch, int nWidth, int nHeight,
                              int iMatrix, cudaStream_t stream);                                  /* :233, ColorSpace.cu:226 */
GMAT_MT_API void Nv12ToBgra64(uint8_t *dpNv12, int nNv12Pitch, uint8_t *dpBgra, int nBgraPitch, int nWidth, int nHeight,
                              int iMatrix, cudaStream_t stream);                                  /* :234, ColorSpace.cu:232 */
GMAT_MT_API void P016ToBgra32(uint8_t *dpP016, int nP016Pitch, uint8_t *dpBgra, int nBgraPitch, int nWidth, int nHeight,
                              int iMatrix, cudaStream_t stream);                                  /* :236, ColorSpace.cu:239 */
GMAT_MT_API void P016ToBgra64(uint8_t *dpP016, int nP016Pitch, uint8_t *dpBgra, int nBgraPitch, int nWidth, int nHeight,
                              int iMatrix, cudaStream_t stream);                                  /* :237, ColorSpace.cu:246 */
GMAT_MT_API void Nv12ToBgrPlanar(uint8_t *dpNv12, int nNv12Pitch, uint8_t *dpBgrp, int nBgrpPitch, int nWidth, int nHeight,
                                 int iMatrix, cudaStream_t stream);                               /* :239, ColorSpace.cu:253 */
GMAT_MT_API void Nv12ToRgbPlanar(uint8_t *dpNv12, int nNv12Pitch, uint8_t *dpBgrp, int nBgrpPitch, int nWidth, int nHeight,
                                 int iMatrix, cudaStream_t stream);                               /* :240, ColorSpace.cu:260 */
GMAT_MT_API void P016ToBgrPlanar(uint8_t *dpP016, int nP016Pitch, uint8_t *dpBgrp, int nBgrpPitch, int nWidth, int nHeight,
                                 int iMatrix, cudaStream_t stream);                               /* :241, ColorSpace.cu:266 */
GMAT_MT_API void Nv12ToBgrFloatPlanar(uint8_t *dpNv12, int nNv12Pitch, float *dpBgrp, int nBgrpPitch, int nWidth, int nHeight,
                                      int iMatrix, cudaStream_t stream);                          /* :243, ColorSpace.cu:273 */
GMAT_MT_API void Nv12ToRgbFloatPlanar(uint8_t *dpNv12, int nNv12Pitch, float *dpBgrp, int nBgrpPitch, int nWidth, int nHeight,
                                      int iMatrix, cudaStream_t stream);                          /* :244, ColorSpace.cu:280 */
GMAT_MT_API void P016ToBgrFloatPlanar(uint8_t *dpNv12, int nNv12Pitch, float *dpBgrp, int nBgrpPitch, int nWidth, int nHeight,
                                      int iMatrix, cudaStream_t stream);                          /* :245, ColorSpace.cu:287 */
GMAT_MT_API void Bgra64ToP016(uint8_t *dpBgra, int nBgraPitch, uint8_t *dpP016, int nP016Pitch, int nWidth, int nHeight,
                              int iMatrix, cudaStream_t stream);                                  /* :247, ColorSpace.cu:345 */
GMAT_MT_API void ConvertUInt8ToUInt16(uint8_t *dpUInt8, uint16_t *dpUInt16, int n);               /* :249, BitDepth.cu:31 */
GMAT_MT_API void ConvertUInt16ToUInt8(uint16_t *dpUInt16, uint8_t *dpUInt8, int n);               /* :250, BitDepth.cu:35 */
GMAT_MT_API void ScaleNv12(unsigned char *dpSrcNv12, int nSrcPitch, int nSrcWidth, int nSrcHeight,
                           unsigned char *dpDstNv12, int nDstPitch, int nDstWidth, int nDstHeight);   /* :252, Resize.cu:75 */
GMAT_MT_API void ScaleP016(unsigned char *dpSrcP016, int nSrcPitch, int nSrcWidth, int nSrcHeight,
                           unsigned char *dpDstP016, int nDstPitch, int nDstWidth, int nDstHeight);   /* :253, Resize.cu:79 */
GMAT_MT_API void ScaleNv12_Bicubic(unsigned char *dpSrcNv12, int nSrcPitch, int nSrcWidth, int nSrcHeight,
                                   unsigned char *dpDstNv12, int nDstPitch, int nDstWidth, int nDstHeight); /* :255, Resize_bicubic.cu:158 */

/* not in the reference: 0 (default) = ScaleNv12_Bicubic computes the reference's kernel; 1 = libswscale's SWS_BICUBIC of one context
 * (process-wide; returns the previous mode) */
extern "C" GMAT_MT_API int gmat_metrans_bicubic_mode(int mode);
#endif
#endif
