/*
 * gmat_metrans.h — MeTrans (metrans/include/NvCodec) colour-space and resize entry points, served by the
 * same kernels as gmat_hip.h.
 *
 * C++ linkage on purpose: the reference declares these as plain C++ functions (NvCommon.h:232-255), so an
 * application object file references their MANGLED names.  With cudaStream_t spelled as a pointer to
 * `struct CUstream_st` (its real definition) the symbols exported here mangle identically, e.g.
 * _Z12Nv12ToBgra32PhiS_iiiiP11CUstream_st, and MeTrans links against libgmat_hip.so unchanged.
 *
 * Layout and argument meaning are the reference's: one NV12 allocation, chroma plane at
 * base + pitch * height (ColorSpace.cu:219-231, Resize.cu:160-200); iMatrix = ColorSpaceStandard
 * (NvCommon.h:16-27), whose codes coincide with libswscale's SWS_CS_*.
 * Arithmetic is libswscale's fixed point (limited-range source), not the reference's float matrix /
 * texture filtering: the colour converters are bit-exact with yuv2rgb.c, ScaleNv12 is SWS_BILINEAR and
 * ScaleNv12_Bicubic is SWS_BICUBIC (B=0, C=0.6, with anti-alias widening on down-scales) rather than the
 * reference's fixed 4x4 a=-0.5 kernel (Resize_bicubic.cu:83-159).
 */
#ifndef GMAT_METRANS_H
#define GMAT_METRANS_H
#include <stdint.h>

#ifdef __cplusplus
struct CUstream_st;
typedef struct CUstream_st *cudaStream_t;      /* carries a hipStream_t */

#define GMAT_MT_API __attribute__((visibility("default")))

GMAT_MT_API void Nv12ToBgra32(uint8_t *dpNv12, int nNv12Pitch, uint8_t *dpBgra, int nBgraPitch, int nWidth, int nHeight,
                              int iMatrix, cudaStream_t stream);                                  /* NvCommon.h:232 */
GMAT_MT_API void Nv12ToRgba32(uint8_t *dpNv12, int nNv12Pitch, uint8_t *dpRgba, int nRgbaPitch, int nWidth, int nHeight,
                              int iMatrix, cudaStream_t stream);                                  /* NvCommon.h:233 */
GMAT_MT_API void ScaleNv12(unsigned char *dpSrcNv12, int nSrcPitch, int nSrcWidth, int nSrcHeight,
                           unsigned char *dpDstNv12, int nDstPitch, int nDstWidth, int nDstHeight);   /* :252 */
GMAT_MT_API void ScaleNv12_Bicubic(unsigned char *dpSrcNv12, int nSrcPitch, int nSrcWidth, int nSrcHeight,
                                   unsigned char *dpDstNv12, int nDstPitch, int nDstWidth, int nDstHeight); /* :255 */
#endif
#endif
