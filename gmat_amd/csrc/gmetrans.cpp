// gmetrans.cpp — MeTrans front-ends (include/gmat_metrans.h) over the libgpuscale layer of this library.
// The reference's versions are stateless free functions (metrans/include/NvCodec/ColorSpace.cu:219-231,
// Resize.cu:160-200, Resize_bicubic.cu:161-175); the context each call implies is cached per geometry.
#include <mutex>
#include <vector>
#include "common.h"
#include "gmat_metrans.h"

namespace {

struct Key { int sw, sh, sf, dw, dh, df, flags, cs; GmatSwsContext *c; };
std::mutex g_lock;
std::vector<Key> g_cache;

// returns a context for the conversion (created on first use, at most 16 kept)
GmatSwsContext *cached(int sw, int sh, int sf, int dw, int dh, int df, int flags, int cs)
{
    for (const Key &k : g_cache)
        if (k.sw == sw && k.sh == sh && k.sf == sf && k.dw == dw && k.dh == dh && k.df == df && k.flags == flags && k.cs == cs)
            return k.c;
    GmatSwsContext *c = gmat_sws_getContext(sw, sh, sf, dw, dh, df, flags | GMAT_SWS_HWACCEL, nullptr);
    if (!c) return nullptr;
    if (cs >= 0) gmat_sws_setColorspace(c, cs, 0);
    if (g_cache.size() >= 16) { gmat_sws_freeContext(g_cache.front().c); g_cache.erase(g_cache.begin()); }
    g_cache.push_back({sw, sh, sf, dw, dh, df, flags, cs, c});
    return c;
}

void nv12_to_packed(uint8_t *nv12, int pitch, uint8_t *dst, int dstPitch, int w, int h, int matrix, int dstFormat, void *stream)
{
    std::lock_guard<std::mutex> g(g_lock);
    GmatSwsContext *c = cached(w, h, GMAT_PIX_FMT_NV12, w, h, dstFormat, 0, matrix);
    if (!c) { gmat::logf(gmat::LOG_ERROR, "metrans: nv12 -> %d %dx%d is not available", dstFormat, w, h); return; }
    const uint8_t *src[4] = {nv12, nv12 + (size_t)pitch * h, nullptr, nullptr};
    const int ss[4] = {pitch, pitch, 0, 0};
    uint8_t *d[4] = {dst, nullptr, nullptr, nullptr};
    const int ds[4] = {dstPitch, 0, 0, 0};
    gmat_sws_setStream(c, stream);
    (void)gmat_sws_scale(c, src, ss, 0, h, d, ds);
}

void scale_nv12(unsigned char *s, int sp, int sw, int sh, unsigned char *dptr, int dp, int dw, int dh, int flags)
{
    std::lock_guard<std::mutex> g(g_lock);
    GmatSwsContext *c = cached(sw, sh, GMAT_PIX_FMT_NV12, dw, dh, GMAT_PIX_FMT_NV12, flags, -1);
    if (!c) { gmat::logf(gmat::LOG_ERROR, "metrans: nv12 %dx%d -> %dx%d is not available", sw, sh, dw, dh); return; }
    const uint8_t *src[4] = {s, s + (size_t)sp * sh, nullptr, nullptr};
    const int ss[4] = {sp, sp, 0, 0};
    uint8_t *d[4] = {dptr, dptr + (size_t)dp * dh, nullptr, nullptr};
    const int ds[4] = {dp, dp, 0, 0};
    gmat_sws_setStream(c, nullptr);                    // the reference launches these on the default stream
    (void)gmat_sws_scale(c, src, ss, 0, sh, d, ds);
}

} // namespace

void Nv12ToBgra32(uint8_t *dpNv12, int nNv12Pitch, uint8_t *dpBgra, int nBgraPitch, int nWidth, int nHeight, int iMatrix,
                  cudaStream_t stream)
{
    nv12_to_packed(dpNv12, nNv12Pitch, dpBgra, nBgraPitch, nWidth, nHeight, iMatrix, GMAT_PIX_FMT_BGRA, (void *)stream);
}

void Nv12ToRgba32(uint8_t *dpNv12, int nNv12Pitch, uint8_t *dpRgba, int nRgbaPitch, int nWidth, int nHeight, int iMatrix,
                  cudaStream_t stream)
{
    nv12_to_packed(dpNv12, nNv12Pitch, dpRgba, nRgbaPitch, nWidth, nHeight, iMatrix, GMAT_PIX_FMT_RGBA, (void *)stream);
}

void ScaleNv12(unsigned char *dpSrcNv12, int nSrcPitch, int nSrcWidth, int nSrcHeight, unsigned char *dpDstNv12, int nDstPitch,
               int nDstWidth, int nDstHeight)
{
    scale_nv12(dpSrcNv12, nSrcPitch, nSrcWidth, nSrcHeight, dpDstNv12, nDstPitch, nDstWidth, nDstHeight, GMAT_SWS_BILINEAR);
}

void ScaleNv12_Bicubic(unsigned char *dpSrcNv12, int nSrcPitch, int nSrcWidth, int nSrcHeight, unsigned char *dpDstNv12,
                       int nDstPitch, int nDstWidth, int nDstHeight)
{
    scale_nv12(dpSrcNv12, nSrcPitch, nSrcWidth, nSrcHeight, dpDstNv12, nDstPitch, nDstWidth, nDstHeight, GMAT_SWS_BICUBIC);
}
