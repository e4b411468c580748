// sws_tables.h — host-side construction of the fixed-point tables the kernels consume.
//
// Product code (not the oracle): an independent C++ implementation of the table maths of
//   libswscale/yuv2rgb.c:774-1030  (ff_yuv2rgb_c_init_tables)  -> Yuv2RgbConsts
//   libswscale/utils.c:765-858     (fill_rgb2yuv_table)        -> Rgb2YuvConsts
//   libswscale/utils.c:367-763     (initFilter)                -> FilterBank
//   libswscale/utils.c:1293-1875   (sub-sampling decisions and filter parameters) -> ScalePlan
// Paths relative to /root/reference/ffmpeg-gpu.
#pragma once
#include <cstdint>
#include <vector>

namespace gmat {

// Constants of the closed form of the yuv2rgb look-up tables:
//   chan = clip_u8((base + (Y + k_chan(U,V)) * cy) >> 16), base = yb0 + 0x8000
//   kR = offR + ((V*crv)>>16), kG = offG + ((U*cgu)>>16) + ((V*cgv)>>16), kB = offB + ((U*cbu)>>16)
// plus the 13-bit coefficients of the full-chroma output stage (yuv2rgb.c:843-848).
struct Yuv2RgbConsts {
    int32_t base, cy;
    int32_t crv, cbu, cgu, cgv;
    int32_t offR, offG, offB;
    int32_t y_coeff, y_offset, v2r, v2g, u2g, u2b;
};

struct Rgb2YuvConsts { int32_t ry, gy, by, ru, gu, bu, rv, gv, bv; };

Yuv2RgbConsts make_yuv2rgb_consts(int colorspace, bool full_range,
                                  int brightness = 0, int contrast = 1 << 16, int saturation = 1 << 16);
Rgb2YuvConsts make_rgb2yuv_consts(int colorspace);

// One resampling axis: `count` output samples, each a window of `taps` int16 coefficients starting
// at source index pos[i].  Coefficients of a row sum to `one` (16384 horizontal, 4096 vertical).
struct FilterBank {
    int taps = 0, count = 0;
    std::vector<int16_t> coef;   // count * taps
    std::vector<int32_t> pos;    // count
    // dword-packed form for v_dot2_i32_i16: for output i the window is re-based to the even index
    // pos_even[i] = pos[i] & ~1 and holds `pairs` (lo,hi) int16 pairs; a row whose pos is odd gets
    // a leading zero tap instead of a data re-alignment in the kernel.
    int pairs = 0;
    std::vector<int32_t> packed;    // count * pairs
    std::vector<int32_t> pos_even;  // count
};

// (re)builds packed / pos_even / pairs from coef / pos
void pack_filter_pairs(FilterBank &fb);

// returns 0 or a negative errno; flags are GMAT_SWS_*; param[] as libswscale's param[2]
int build_filter(FilterBank &out, int inc, int src_len, int dst_len, int one, int flags,
                 const double param[2], int src_pos, int dst_pos);

// Geometry of one generic-scaler context (sws_init_single_context).
struct ScalePlan {
    int srcW, srcH, dstW, dstH, srcFormat, dstFormat, flags;
    int chrSrcHSub, chrSrcVSub, chrDstHSub, chrDstVSub;
    int chrSrcW, chrSrcH, chrDstW, chrDstH;
    int lumXInc, lumYInc, chrXInc, chrYInc;
    FilterBank hLum, hChr, vLum, vChr;
};

// chrPos = { src_h_chr_pos, src_v_chr_pos, dst_h_chr_pos, dst_v_chr_pos } (the AVOptions of those names,
// libswscale/options.c:67-70); nullptr or -513 = unset
int build_scale_plan(ScalePlan &p, int srcW, int srcH, int srcFormat, int dstW, int dstH, int dstFormat,
                     int flags, const double param[2], const int *chrPos = nullptr);

} // namespace gmat
