// gmetrans.cpp — the 17 MeTrans entry points (include/gmat_metrans.h; metrans/include/NvCodec/NvCommon.h:232-255) over the libgpuscale
// layer of this library and the kernels of k_metrans.hip.
// The reference's versions are stateless free functions (ColorSpace.cu:219-350, Resize.cu:75-81, Resize_bicubic.cu:158-160,
// BitDepth.cu:31-37); the context or scratch frame a call implies is cached per geometry and device.
#include <atomic>
#include <mutex>
#include <vector>
#include "common.h"
#include "kernels.h"
#include "sws_tables.h"
#include "gmat_metrans.h"

using namespace gmat;

namespace {

struct Key { int sw, sh, sf, dw, dh, df, flags, cs, dev; GmatSwsContext *c; };
std::mutex g_lock;
std::vector<Key> g_cache;
std::atomic<int> g_bicubicMode{0};

int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return dev;
}

// ColorSpaceStandard -> the SWS_CS_* row with the same coefficients; GetConstants (ColorSpace.cu:32-64) sends every code it does not
// list to BT.709
int sws_matrix(int iMatrix)
{
    switch (iMatrix) {
    case 4: return 4;                        // FCC
    case 5: case 6: return GMAT_SWS_CS_ITU601;   // BT.470 / BT.601: wr 0.299, wb 0.114
    case 7: return 7;                        // SMPTE 240M
    case 9: case 10: return GMAT_SWS_CS_BT2020;
    default: return GMAT_SWS_CS_ITU709;
    }
}

// returns a context for the conversion (created on first use, at most 16 kept); g_lock held
GmatSwsContext *cached(int sw, int sh, int sf, int dw, int dh, int df, int flags, int cs)
{
    const int dev = current_device();
    for (const Key &k : g_cache)
        if (k.sw == sw && k.sh == sh && k.sf == sf && k.dw == dw && k.dh == dh && k.df == df && k.flags == flags && k.cs == cs && k.dev == dev)
            return k.c;
    GmatSwsContext *c = gmat_sws_getContext(sw, sh, sf, dw, dh, df, flags | GMAT_SWS_HWACCEL, nullptr);
    if (!c) return nullptr;
    if (cs >= 0 && gmat_sws_setColorspace(c, cs, 0) < 0) { gmat_sws_freeContext(c); return nullptr; }
    if (g_cache.size() >= 16) { gmat_sws_freeContext(g_cache.front().c); g_cache.erase(g_cache.begin()); }
    g_cache.push_back({sw, sh, sf, dw, dh, df, flags, cs, dev, c});
    return c;
}

// a scratch frame per device (the BGRA pixels between a P016 source and its planar outputs); grows, never shrinks; g_lock held
struct Scratch { int dev; uint8_t *p; size_t bytes; };
std::vector<Scratch> g_scratch;
uint8_t *scratch(size_t bytes)
{
    const int dev = current_device();
    for (Scratch &s : g_scratch)
        if (s.dev == dev) {
            if (s.bytes >= bytes) return s.p;
            (void)hipFree(s.p);
            s.p = nullptr; s.bytes = 0;
            if (hipMalloc((void **)&s.p, bytes) != hipSuccess) return nullptr;
            s.bytes = bytes;
            return s.p;
        }
    uint8_t *p = nullptr;
    if (hipMalloc((void **)&p, bytes) != hipSuccess) return nullptr;
    g_scratch.push_back({dev, p, bytes});
    return p;
}

// one semi-planar 4:2:0 allocation (chroma at base + pitch * height) through a context into a packed frame
int semi_to_packed(uint8_t *yuv, int pitch, int srcFormat, uint8_t *dst, int dstPitch, int w, int h, int iMatrix, int dstFormat, void *stream)
{
    GmatSwsContext *c = cached(w, h, srcFormat, w, h, dstFormat, 0, sws_matrix(iMatrix));
    if (!c) { logf(LOG_ERROR, "metrans: %d -> %d %dx%d is not available", srcFormat, dstFormat, w, h); return GMAT_ERR(ENOSYS); }
    const uint8_t *src[4] = {yuv, yuv + (size_t)pitch * h, nullptr, nullptr};
    const int ss[4] = {pitch, pitch, 0, 0};
    uint8_t *d[4] = {dst, nullptr, nullptr, nullptr};
    const int ds[4] = {dstPitch, 0, 0, 0};
    gmat_sws_setStream(c, stream);
    return gmat_sws_scale(c, src, ss, 0, h, d, ds);
}

void nv12_to_planar(uint8_t *nv12, int pitch, uint8_t *dst, int dstPitch, int w, int h, int iMatrix, int f32, int bgr, void *stream)
{
    if (!nv12 || !dst || w < 1 || h < 1) { logf(LOG_ERROR, "metrans: nv12 -> planar: bad arguments"); return; }
    YuvSrc s{};
    s.y = nv12; s.ys = pitch; s.u = nv12 + (size_t)pitch * h; s.us = pitch; s.nv12 = 1;
    (void)launch_nv12_to_planar(s, dst, dstPitch, w, h, make_yuv2rgb_consts(sws_matrix(iMatrix), false), f32, bgr, (hipStream_t)stream);
}

void p016_to_planar(uint8_t *p016, int pitch, uint8_t *dst, int dstPitch, int w, int h, int iMatrix, int f32, void *stream)
{
    if (!p016 || !dst || w < 1 || h < 1) { logf(LOG_ERROR, "metrans: p016 -> planar: bad arguments"); return; }
    std::lock_guard<std::mutex> g(g_lock);
    const int ip = align_up(4 * w, 256);
    uint8_t *bgra = scratch((size_t)ip * h);
    if (!bgra) { logf(LOG_ERROR, "metrans: p016 -> planar: no memory for the %dx%d intermediate", w, h); return; }
    if (semi_to_packed(p016, pitch, GMAT_PIX_FMT_P016LE, bgra, ip, w, h, iMatrix, GMAT_PIX_FMT_BGRA, stream) < 0) return;
    (void)launch_split_packed32(bgra, ip, dst, dstPitch, w, h, f32, (hipStream_t)stream);
}

void to_packed(uint8_t *yuv, int pitch, int srcFormat, uint8_t *dst, int dstPitch, int w, int h, int iMatrix, int dstFormat, void *stream)
{
    if (!yuv || !dst) { logf(LOG_ERROR, "metrans: null frame"); return; }
    std::lock_guard<std::mutex> g(g_lock);
    (void)semi_to_packed(yuv, pitch, srcFormat, dst, dstPitch, w, h, iMatrix, dstFormat, stream);
}

void scale_semi(unsigned char *s, int sp, int sw, int sh, unsigned char *dptr, int dp, int dw, int dh, int format, int flags)
{
    if (!s || !dptr) { logf(LOG_ERROR, "metrans: null frame"); return; }
    std::lock_guard<std::mutex> g(g_lock);
    GmatSwsContext *c = cached(sw, sh, format, dw, dh, format, flags, -1);
    if (!c) { logf(LOG_ERROR, "metrans: %d %dx%d -> %dx%d is not available", format, sw, sh, dw, dh); return; }
    const uint8_t *src[4] = {s, s + (size_t)sp * sh, nullptr, nullptr};
    const int ss[4] = {sp, sp, 0, 0};
    uint8_t *d[4] = {dptr, dptr + (size_t)dp * dh, nullptr, nullptr};
    const int ds[4] = {dp, dp, 0, 0};
    gmat_sws_setStream(c, nullptr);                    // the reference launches these on the default stream
    (void)gmat_sws_scale(c, src, ss, 0, sh, d, ds);
}

} // namespace

// ---- colour conversion (ColorSpace.cu:219-350).  `stream` reaches the kernels only where the reference passes it to its launch ----
void Nv12ToBgra32(uint8_t *dpNv12, int nNv12Pitch, uint8_t *dpBgra, int nBgraPitch, int nWidth, int nHeight, int iMatrix, cudaStream_t)
{
    to_packed(dpNv12, nNv12Pitch, GMAT_PIX_FMT_NV12, dpBgra, nBgraPitch, nWidth, nHeight, iMatrix, GMAT_PIX_FMT_BGRA, nullptr);
}

void Nv12ToRgba32(uint8_t *dpNv12, int nNv12Pitch, uint8_t *dpBgra, int nBgraPitch, int nWidth, int nHeight, int iMatrix, cudaStream_t)
{
    to_packed(dpNv12, nNv12Pitch, GMAT_PIX_FMT_NV12, dpBgra, nBgraPitch, nWidth, nHeight, iMatrix, GMAT_PIX_FMT_RGBA, nullptr);
}

void Nv12ToBgra64(uint8_t *dpNv12, int nNv12Pitch, uint8_t *dpBgra, int nBgraPitch, int nWidth, int nHeight, int iMatrix, cudaStream_t)
{
    to_packed(dpNv12, nNv12Pitch, GMAT_PIX_FMT_NV12, dpBgra, nBgraPitch, nWidth, nHeight, iMatrix, GMAT_PIX_FMT_BGRA64LE, nullptr);
}

void P016ToBgra32(uint8_t *dpP016, int nP016Pitch, uint8_t *dpBgra, int nBgraPitch, int nWidth, int nHeight, int iMatrix, cudaStream_t)
{
    to_packed(dpP016, nP016Pitch, GMAT_PIX_FMT_P016LE, dpBgra, nBgraPitch, nWidth, nHeight, iMatrix, GMAT_PIX_FMT_BGRA, nullptr);
}

void P016ToBgra64(uint8_t *dpP016, int nP016Pitch, uint8_t *dpBgra, int nBgraPitch, int nWidth, int nHeight, int iMatrix, cudaStream_t)
{
    to_packed(dpP016, nP016Pitch, GMAT_PIX_FMT_P016LE, dpBgra, nBgraPitch, nWidth, nHeight, iMatrix, GMAT_PIX_FMT_BGRA64LE, nullptr);
}

void Nv12ToBgrPlanar(uint8_t *dpNv12, int nNv12Pitch, uint8_t *dpBgrp, int nBgrpPitch, int nWidth, int nHeight, int iMatrix, cudaStream_t)
{
    nv12_to_planar(dpNv12, nNv12Pitch, dpBgrp, nBgrpPitch, nWidth, nHeight, iMatrix, 0, 1, nullptr);
}

void Nv12ToRgbPlanar(uint8_t *dpNv12, int nNv12Pitch, uint8_t *dpBgrp, int nBgrpPitch, int nWidth, int nHeight, int iMatrix, cudaStream_t)
{
    nv12_to_planar(dpNv12, nNv12Pitch, dpBgrp, nBgrpPitch, nWidth, nHeight, iMatrix, 0, 0, nullptr);
}

void P016ToBgrPlanar(uint8_t *dpP016, int nP016Pitch, uint8_t *dpBgrp, int nBgrpPitch, int nWidth, int nHeight, int iMatrix, cudaStream_t)
{
    p016_to_planar(dpP016, nP016Pitch, dpBgrp, nBgrpPitch, nWidth, nHeight, iMatrix, 0, nullptr);
}

void Nv12ToBgrFloatPlanar(uint8_t *dpNv12, int nNv12Pitch, float *dpBgrp, int nBgrpPitch, int nWidth, int nHeight, int iMatrix,
                          cudaStream_t stream)
{
    nv12_to_planar(dpNv12, nNv12Pitch, (uint8_t *)dpBgrp, nBgrpPitch, nWidth, nHeight, iMatrix, 1, 1, (void *)stream);
}

void Nv12ToRgbFloatPlanar(uint8_t *dpNv12, int nNv12Pitch, float *dpBgrp, int nBgrpPitch, int nWidth, int nHeight, int iMatrix,
                          cudaStream_t stream)
{
    nv12_to_planar(dpNv12, nNv12Pitch, (uint8_t *)dpBgrp, nBgrpPitch, nWidth, nHeight, iMatrix, 1, 0, (void *)stream);
}

void P016ToBgrFloatPlanar(uint8_t *dpP016, int nP016Pitch, float *dpBgrp, int nBgrpPitch, int nWidth, int nHeight, int iMatrix,
                          cudaStream_t stream)
{
    p016_to_planar(dpP016, nP016Pitch, (uint8_t *)dpBgrp, nBgrpPitch, nWidth, nHeight, iMatrix, 1, (void *)stream);
}

void Bgra64ToP016(uint8_t *dpBgra, int nBgraPitch, uint8_t *dpP016, int nP016Pitch, int nWidth, int nHeight, int iMatrix, cudaStream_t)
{
    if (!dpBgra || !dpP016) { logf(LOG_ERROR, "metrans: null frame"); return; }
    std::lock_guard<std::mutex> g(g_lock);
    GmatSwsContext *c = cached(nWidth, nHeight, GMAT_PIX_FMT_BGRA64LE, nWidth, nHeight, GMAT_PIX_FMT_P016LE, 0, sws_matrix(iMatrix));
    if (!c) { logf(LOG_ERROR, "metrans: bgra64 -> p016 %dx%d is not available", nWidth, nHeight); return; }
    const uint8_t *src[4] = {dpBgra, nullptr, nullptr, nullptr};
    const int ss[4] = {nBgraPitch, 0, 0, 0};
    uint8_t *d[4] = {dpP016, dpP016 + (size_t)nP016Pitch * nHeight, nullptr, nullptr};
    const int ds[4] = {nP016Pitch, nP016Pitch, 0, 0};
    gmat_sws_setStream(c, nullptr);
    (void)gmat_sws_scale(c, src, ss, 0, nHeight, d, ds);
}

// ---- BitDepth.cu:31-37 ----
void ConvertUInt8ToUInt16(uint8_t *dpUInt8, uint16_t *dpUInt16, int n) { (void)launch_widen_shift8(dpUInt8, dpUInt16, n, nullptr); }
void ConvertUInt16ToUInt8(uint16_t *dpUInt16, uint8_t *dpUInt8, int n) { (void)launch_narrow_shift8(dpUInt16, dpUInt8, n, nullptr); }

// ---- Resize.cu:75-81, Resize_bicubic.cu:158-160 ----
void ScaleNv12(unsigned char *dpSrcNv12, int nSrcPitch, int nSrcWidth, int nSrcHeight, unsigned char *dpDstNv12, int nDstPitch,
               int nDstWidth, int nDstHeight)
{
    scale_semi(dpSrcNv12, nSrcPitch, nSrcWidth, nSrcHeight, dpDstNv12, nDstPitch, nDstWidth, nDstHeight, GMAT_PIX_FMT_NV12, GMAT_SWS_BILINEAR);
}

void ScaleP016(unsigned char *dpSrcP016, int nSrcPitch, int nSrcWidth, int nSrcHeight, unsigned char *dpDstP016, int nDstPitch,
               int nDstWidth, int nDstHeight)
{
    scale_semi(dpSrcP016, nSrcPitch, nSrcWidth, nSrcHeight, dpDstP016, nDstPitch, nDstWidth, nDstHeight, GMAT_PIX_FMT_P016LE, GMAT_SWS_BILINEAR);
}

void ScaleNv12_Bicubic(unsigned char *dpSrcNv12, int nSrcPitch, int nSrcWidth, int nSrcHeight, unsigned char *dpDstNv12,
                       int nDstPitch, int nDstWidth, int nDstHeight)
{
    if (g_bicubicMode.load() == 1) {
        scale_semi(dpSrcNv12, nSrcPitch, nSrcWidth, nSrcHeight, dpDstNv12, nDstPitch, nDstWidth, nDstHeight, GMAT_PIX_FMT_NV12, GMAT_SWS_BICUBIC);
        return;
    }
    const int r = launch_scale_nv12_bicubic_ref(dpSrcNv12, nSrcPitch, nSrcWidth, nSrcHeight, dpDstNv12, nDstPitch, nDstWidth, nDstHeight, nullptr);
    if (r < 0) logf(LOG_ERROR, "metrans: ScaleNv12_Bicubic %dx%d -> %dx%d refused (%d): the reference's kernel needs at least 8 x 8 source samples",
                    nSrcWidth, nSrcHeight, nDstWidth, nDstHeight, r);
}

extern "C" int gmat_metrans_bicubic_mode(int mode)
{
    if (mode != 0 && mode != 1) return GMAT_ERR(EINVAL);
    return g_bicubicMode.exchange(mode);
}
