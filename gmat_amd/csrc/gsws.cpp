// gsws.cpp — libgpuscale for MI355X: the sws-style context API behind include/gmat_hip.h §1.
//
// Mirrors the control flow of the reference's GPU back-end (paths relative to
// /root/reference/ffmpeg-gpu/libswscale):
//   sws_init_context_cuda            utils.c:2026-2060   -> gmat_sws_getContext
//   ff_get_unscaled_swscale_cuda     swscale_unscaled.c:2014-2054 (same-size converters)
//   ff_sws_init_swscale_cuda         cuda/swscale_cuda.c:112-271  (scaled: intermediates + op)
//   ff_swscale_cuda                  cuda/swscale_cuda.c:273-479  -> gmat_sws_scale
//   ff_sws_free_swscale_cuda         cuda/swscale_cuda.c:86-109   -> gmat_sws_freeContext
// Differences by design (SURVEY.md §0 defects): colour constants are per-context kernel arguments
// (no process-global __constant__ upload on the NULL stream), the interpolation follows the flags,
// and every device error is returned.
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <vector>
#include "common.h"
#include "kernels.h"
#include "sws_tables.h"

using namespace gmat;

namespace {

enum Mode { MODE_YUV2RGB, MODE_RGBPF32, MODE_SWAP_RB, MODE_COPY, MODE_SCALE, MODE_RGB2YUV, MODE_YUV2YUV, MODE_DEPTH, MODE_FROM_PF32,
            MODE_RGB2YUV444, MODE_REPACK, MODE_PLANECOPY, MODE_VIA_INNER, MODE_SCALE16, MODE_VIA_PLANES16, MODE_PLANE_UP, MODE_PLANE_DOWN };

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int upload(const void *host, size_t bytes)
    {
        if (p) { (void)hipFree(p); p = nullptr; }
        if (!bytes) return 0;
        GMAT_HIP_CHECK(hipMalloc(&p, bytes));
        GMAT_HIP_CHECK(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
        return 0;
    }
    int reserve(size_t bytes)
    {
        if (p) { (void)hipFree(p); p = nullptr; }
        if (!bytes) return 0;
        GMAT_HIP_CHECK(hipMalloc(&p, bytes));
        return 0;
    }
};

struct DevFilterStore {
    DevBuf packed, pos, round;
    int upload(const FilterBank &fb, const std::vector<int32_t> &rnd, DevFilter &out)
    {
        int r;
        if ((r = packed.upload(fb.packed.data(), fb.packed.size() * sizeof(int32_t))) < 0) return r;
        if ((r = pos.upload(fb.pos_even.data(), fb.pos_even.size() * sizeof(int32_t))) < 0) return r;
        if ((r = round.upload(rnd.data(), rnd.size() * sizeof(int32_t))) < 0) return r;
        out.packed = (const int32_t *)packed.p;
        out.pos_even = (const int32_t *)pos.p;
        out.round = (const int32_t *)round.p;
        out.pairs = fb.pairs; out.taps = fb.taps; out.count = fb.count;
        return 0;
    }
    int upload(const FilterBank &fb, bool vertical, DevFilter &out)
    {
        int r;
        if ((r = packed.upload(fb.packed.data(), fb.packed.size() * sizeof(int32_t))) < 0) return r;
        if ((r = pos.upload(fb.pos_even.data(), fb.pos_even.size() * sizeof(int32_t))) < 0) return r;
        std::vector<int32_t> rnd(fb.count, 1 << 9);
        if (vertical && fb.taps == 1) {
            // packed_vscale's 1-tap form (vscale.c:135-145 -> yuv2rgb_full_1_c, output.c:2164-2200) takes the line as
            // it is: the coefficient is never read, and for degenerate rows it is not 4096
            FilterBank eff = fb;
            std::fill(eff.coef.begin(), eff.coef.end(), (int16_t)4096);
            pack_filter_pairs(eff);
            if ((r = packed.upload(eff.packed.data(), eff.packed.size() * sizeof(int32_t))) < 0) return r;
        }
        if (vertical && fb.taps == 2) {
            // packed_vscale's 2-tap form (vscale.c:146-160 -> yuv2rgb_full_2_c, output.c:2118-2120):
            // no rounding constant when the two taps are a proper blend
            for (int i = 0; i < fb.count; i++) {
                const int f0 = fb.coef[(size_t)i * 2], f1 = fb.coef[(size_t)i * 2 + 1];
                if (f0 + f1 == 4096 && (unsigned)f1 <= 4096u) rnd[i] = 0;
            }
        }
        if ((r = round.upload(rnd.data(), rnd.size() * sizeof(int32_t))) < 0) return r;
        out.packed = (const int32_t *)packed.p;
        out.pos_even = (const int32_t *)pos.p;
        out.round = (const int32_t *)round.p;
        out.pairs = fb.pairs; out.taps = fb.taps; out.count = fb.count;
        return 0;
    }
};

} // namespace

extern "C" void gmat_sws_freeContext(GmatSwsContext *c);

struct GmatSwsContext {
    int srcW, srcH, srcFormat, dstW, dstH, dstFormat, flags;
    double param[2];
    hipStream_t stream = nullptr;
    int device = 0;               // the HIP device current at creation: the context's tables live there (hwcontext_cuda.c:395-434
                                  // makes the stream's device current around every call; a context is used on ITS device only)
    Mode mode;
    Mode unscaledMode = MODE_SCALE;   // the mode chosen at creation: gmat_sws_setRange leaves and re-enters the special converters
    int colorspace = GMAT_SWS_CS_DEFAULT, srcFullRange = 0;
    int chrPos[4] = {-513, -513, -513, -513};   // src_h / src_v / dst_h / dst_v chroma positions (options.c:67-70)
    int rangeConv = 0;            // YUV -> YUV: 1 limited->full (lum/chrRangeToJpeg), 2 full->limited
    bool planeDownFull = false;   // MODE_PLANE_DOWN with both ranges full: the luma's (v - (v >> 8) + d) >> shift
    Yuv2RgbConsts y2r;            // for the same-size converter (honours colourspace / range)
    // scaler
    ScalePlan plan;               // always an RGB24 -> dst plan (the YUV source is converted in front)
    ScaleTiling tiling;
    DevFilterStore dHLum, dHChr, dVLum;
    DevBuf dColStart, dColCount, dRowStart, dRowCount, dColMagic;
    ScaleArgs args;
    bool rgbReady = false;
    // YUV-source scaler with libswscale's single-context semantics (k_scale_yuv.hip)
    ScalePlan planYuv;
    YuvScaleTiling ytiling;
    DevFilterStore yHLum, yHChr, yVLum, yVChr;
    DevBuf yWin[8];
    YuvScaleArgs yargs;
    bool yuvReady = false;
    // same-size RGB -> YUV 4:2:0 (k_rgb2yuv.hip)
    ScalePlan planR2Y;
    Rgb2YuvPlan r2y;
    DevFilterStore dR2YvChr;
    DevBuf dR2YrowStart, dR2YrowCount;
    DevFilter r2yVChr;
    Yuv2xTables y2x;              // 2:1 horizontal specialisation (k_scale_yuv2x.hip), y2x.ok = LDS bytes
    Yuv2sTables y2s;              // strip-walking 2:1 form (k_scale_yuv2s.hip), RGB destinations
    Yuv2pTables y2p;                     // strip-walking 4:2:0 -> 4:2:0 form (same chroma layout on both sides)
    Yuv1x2Tables y1x2;                   // strip-walking 1:2 up-scale, 8-bit 4:2:0 -> 4:2:0
    Yuv3x1Tables y3x1;                   // strip-walking 3:1 down-scale, 8-bit 4:2:0 -> 4:2:0
    Yuv4x1Tables y4x1;                   // strip-walking 4:1 down-scale, 8-bit 4:2:0 -> 4:2:0
    Yuv4rTables y4r;                     // strip-walking 4:1 NV12 -> packed RGB
    Yuv32rTables y32r;                   // strip-walking 3:2 NV12 -> packed RGB
    Yuv3rTables y3r;                     // strip-walking 3:1 NV12 -> packed RGB
    Rgb2yTables r2ys;                    // strip-walking 2:1 packed RGB -> 8-bit 4:2:0
    Yuv3x2Tables y3x2;                   // strip-walking 3:2 down-scale, 8-bit 4:2:0 -> 4:2:0
    Rgb2sTables r2s;              // strip-walking 2:1 form of the packed-RGB source scaler (k_scale_rgb2s.hip)
    YuvGTables rg;                // (round 5) any other ratio the band walker reaches: scale_yuvg_rgbsrc_kernel (k_scale_yuvg16.hip) ...
    YuvGArgs rgargs;              // ... its arguments but the call's pointers and pitches ...
    DevBuf dRG[9];                // ... and device tables: hL, hC, posL, posC, prog, qfirst, qdone, vtL (the block-cooperative form's)
    YuvGTables yg;                // the polyphase band walker for any ratio (k_scale_yuvg.hip)
    DevBuf dG[4 + 2 * 2 * 5];     // its device tables: hL, hC, posL, posC, then per (plane class, direction) coef / first / last / round / yLo
    YuvGArgs gargs;
    YuvUTables yu;                // the quad-lane walker: up-scales of any factor, short filters (k_scale_yuvu.hip)
    DevBuf dU[12];                // its device tables
    YuvUArgs uargs;
    YuvLTables yl;                // the lines form: two launches through a frame of 15-bit lines (k_scale_yuvl.hip)
    DevBuf dL[4];                 // its device tables: hL, hC, offL, offC
    YuvLArgs largs;
    int32_t *linesBuf = nullptr;  // linesFrames lines frames (an intermediate the context owns: stream_handoff_*)
    uint8_t *linesStage = nullptr; size_t linesStageBytes = 0;     // dword-aligned copies of source planes that are not (lines_stage)
    int linesFrames = 0;
    DevBuf dHLreg, dHCreg, dVrec, dVrecC;
    // how a scaled YUV->RGB context runs: 0 two kernels (convert, scale) with an HBM intermediate,
    // 1 the same arithmetic in one fused kernel, 2 one libswscale context (planes scaled separately)
    int fused = 2;
    uint8_t *inter = nullptr;     // RGB24 intermediate at source size for the two-kernel form
    int interStride = 0;
    uint8_t *interBatch = nullptr; // kYuv2xMaxFrames such intermediates for the batched two-kernel form
    int interBatchFrames = 0;
    const char *lastKernel = "";
    int lastLaunchFrames = 1;
    // RGBA / BGRA sources of the scaling / RGB -> YUV paths: alpha dropped into `inter` (RGB24 / BGR24), then `inner`
    GmatSwsContext *inner = nullptr;
    int px4 = 0, px4Alpha = 0;    // (set around ONE call by the RGBA / BGRA context in front of this one: the source pixels are four bytes wide — scale_yuvg_rgbsrc_blk_kernel reads them as they are)
    // RGBA64LE / BGRA64LE sources (MODE_VIA_PLANES16): rgb64ToY / ToUV(_half) into Y / U / V planes of 16-bit samples, then `inner`
    // — the planar-16 context with an RGB source's chroma geometry (k_rgb64.hip)
    DevBuf planes16;
    int p16Stride[2] = {0, 0};            // luma / chroma pitch in bytes
    size_t p16Off[3] = {0, 0, 0};
    // both ends carry alpha (needAlpha, utils.c:1902): the alpha plane through the LUMA filters of the context that makes the colours
    bool needAlpha = false;
    DevBuf alphaLines, aForm, aFirst;     // srcH x dstW int32 lines; per output row: the packed writer's form, the filter's position
    DevFilterStore aH, aV;
    DevFilter daH, daV;
    const int32_t *alpha19 = nullptr;     // set on `inner` by its owner for one call: the alpha lines of an RGBA64 destination
    bool rgbViaPlanes = false;            // RGB24 / BGR24 source scaled to a YUV destination: the plane scaler with its RGB loader
    bool src0 = false, dst0 = false;      // RGB0 / BGR0 ends, handled as RGBA / BGRA (handle_0alpha, utils.c:1121-1144)
    // 16-bit destinations (P016LE): 19-bit int32 lines in HBM between the two passes of k_scale16.hip
    ScalePlan plan16;
    DevFilterStore f16[4];
    DevFilter d16[4];                 // hLum, hChr, vLum, vChr
    DevBuf line16[3];                 // luma, U, V: srcH x dstW / chrSrcH x chrDstW int32
    mutable int shift8Ident = -1;     // shift8_shortcut's answer about the active plan's four banks (-1: not asked since the plan was last built)
    S19Tables s19;                    // (round 6) 16-bit YUV destinations in one launch, the lines of a tile in LDS (k_scale19.hip); s19.ok = 0: the two passes
    DevBuf dS19[6];                   // its tables: per job the tile columns' first bytes, the tile rows' first source rows and their counts
    S19Tables t15;                    // (round 6) the same tile kernel on the 15-bit lines (8- and 10-bit YUV destinations): the pairs no walker serves — a layout change, a 4:4:4 end
    DevBuf dT15[6];
    bool t15Mixed = false;            // ... semi-planar <-> planar, or 4:4:4 at one end: in front of the lines form below 4 : 1
    UnitRgbPlan urgb;                 // (round 6) 16-bit 4:2:0 sources into packed 8-bit RGB at equal size: unit_rgb_kernel (k_scale19.hip)
    unsigned long long *prof = nullptr;
    hipEvent_t batchEv[9] = {nullptr};
    bool batchEvReady = false;
    // NV12 <-> YUV420P SCALED (round 4): no walker but the 2:1 one writes the other chroma layout, so such a context fell to the tiled kernel of
    // round 1 (0.04 - 0.14 of the roofline) — nvdec's NV12 scaled for a planar consumer, the commonest ladder step there is.  `cross` is the same
    // context in the SOURCE's layout (every walker applies), created on first use with this context's positions and ranges and dropped when
    // they change; its output lands in `crossBuf` (a frame of the destination's size) and yuv420_relayout_kernel moves it into the destination
    GmatSwsContext *cross = nullptr;
    uint8_t *crossBuf = nullptr;
    int crossPitch = 0, crossFrames = 0;
    // A context owns ONE set of intermediates (crossBuf, inter / interBatch, the 16-bit lines, an inner context's): a call on a stream other than
    // the one that used them last is ordered behind that use by an event (stream_handoff_*) — gmat_sws_scale_batch hands a context's frames to
    // several streams, and a per-frame caller may alternate streams of its own.  The reference has one set of cv_* buffers per context and leaves
    // the ordering to its caller (swscale_cuda.c:86-109, 248-266).
    hipEvent_t interEv = nullptr;
    hipStream_t interStream = nullptr;
    bool interUsed = false, interTouched = false, interMulti = false;      // interMulti: the context has been used on more than one stream
    int handoffs = 0;
    ~GmatSwsContext()
    {
        if (inter) (void)hipFree(inter);
        if (interBatch) (void)hipFree(interBatch);
        if (cross) gmat_sws_freeContext(cross);
        if (crossBuf) (void)hipFree(crossBuf);
        if (linesBuf) (void)hipFree(linesBuf); if (linesStage) (void)hipFree(linesStage);
        if (interEv) (void)hipEventDestroy(interEv);
        if (inner) gmat_sws_freeContext(inner);
        if (batchEvReady) for (hipEvent_t e : batchEv) if (e) (void)hipEventDestroy(e);
    }
};

// sources of the single-context plane scaler: 8-bit planar / semi-planar YUV and the 16-bit semi-planar P010LE / P016LE
static inline bool is_plane_src(int f) { return is_yuv8_src(f) || is_p01x(f) || pl16_depth(f); }

static int scale16_kind(int srcFormat);
static int init_yuv_scaler(GmatSwsContext *c)
{
    if (c->yuvReady) return 0;
    c->shift8Ident = -1;
    int r = build_scale_plan(c->planYuv, c->srcW, c->srcH, c->srcFormat, c->dstW, c->dstH, c->dstFormat, c->flags, c->param,
                             c->chrPos);
    if (r < 0) return r;
    if ((r = yuvscale_prepare(c->planYuv, c->ytiling)) < 0) return r;
    YuvScaleArgs &a = c->yargs;
    std::memset(&a, 0, sizeof(a));
    const YuvScaleTiling &t = c->ytiling;
    const std::vector<int32_t> none(std::max(c->dstW, c->dstH), 0);
    a.chrDstH = c->planYuv.chrDstH;
    a.dstNv12 = c->dstFormat == GMAT_PIX_FMT_NV12 || c->dstFormat == GMAT_PIX_FMT_P010LE;   // interleaved chroma
    a.dst16 = c->dstFormat == GMAT_PIX_FMT_P010LE ? 1 : c->dstFormat == GMAT_PIX_FMT_YUV420P10LE ? 2 : 0;
    a.dstShift = a.dst16 == 1 ? 6 : 0;
    if ((r = c->yHLum.upload(c->planYuv.hLum, none, a.hLum)) < 0) return r;
    if ((r = c->yHChr.upload(c->planYuv.hChr, none, a.hChr)) < 0) return r;
    if ((r = c->yVLum.upload(t.vLumEff, t.lumRound, a.vLum)) < 0) return r;
    if ((r = c->yVChr.upload(t.vChrEff, t.chrRound, a.vChr)) < 0) return r;
    const std::vector<int32_t> *w[8] = {&t.colStartL, &t.colCountL, &t.rowStartL, &t.rowCountL,
                                        &t.colStartC, &t.colCountC, &t.rowStartC, &t.rowCountC};
    const int32_t **dst[8] = {&a.colStartL, &a.colCountL, &a.rowStartL, &a.rowCountL,
                              &a.colStartC, &a.colCountC, &a.rowStartC, &a.rowCountC};
    for (int i = 0; i < 8; i++) {
        if ((r = c->yWin[i].upload(w[i]->data(), w[i]->size() * 4)) < 0) return r;
        *dst[i] = (const int32_t *)c->yWin[i].p;
    }
    a.TH = t.TH; a.ntx = t.ntx; a.nty = t.nty; a.xcdRemap = t.xcdRemap; a.fullChroma = t.fullChroma;
    a.rowsL = t.rowsL; a.colsL = t.colsL; a.rowsC = t.rowsC; a.colsC = t.colsC;
    a.srcW = c->srcW; a.srcH = c->srcH; a.chrSrcW = c->planYuv.chrSrcW; a.chrSrcH = c->planYuv.chrSrcH;
    a.dstW = c->dstW; a.dstH = c->dstH; a.chrDstW = c->planYuv.chrDstW;
    a.dstFormat = c->dstFormat;
    a.nv12 = c->srcFormat == GMAT_PIX_FMT_NV12;
    if (c->rgbViaPlanes) {
        a.src16 = 3; a.hShift = 13; a.hBias = 0;         // hScale16To15_c: sh = 13 for RGB sources (swscale.c:93-119)
        a.rgbBgr = c->srcFormat == GMAT_PIX_FMT_BGR24; a.chrHalf = c->planYuv.chrSrcHSub;
        a.r2y = make_rgb2yuv_consts(is_packed_rgb(c->dstFormat) ? GMAT_SWS_CS_DEFAULT : c->colorspace);
    }
    if (pl16_depth(c->srcFormat) == 16) { a.src16 = 17; a.hShift = 15; a.hBias = 1 << 29; }     // planar, any chroma subsampling
    if (pl16_depth(c->srcFormat) == 10) { a.src16 = 18; a.hShift = 9; a.hBias = 0; }            // 10 bits in the low end: as they are
    if (is_p01x(c->srcFormat)) {
        a.src16 = c->srcFormat == GMAT_PIX_FMT_P010LE ? 10 : 16;
        a.hShift = a.src16 - 1;                          // hScale16To15_c: sh = depth - 1 (swscale.c:93-119)
        a.hBias = a.src16 == 16 ? (1 << 29) : 0;         // 32768 * 16384: undoes the -32768 of the P016 LDS image
    }
    if ((r = yuv2s_prepare(c->planYuv, c->ytiling, c->y2s)) < 0) return r;
    if ((r = yuv2p_prepare(c->planYuv, c->ytiling, c->y2p)) < 0) return r;
    if ((r = yuv1x2_prepare(c->planYuv, c->ytiling, c->y1x2)) < 0) return r;
    if ((r = yuv3x1_prepare(c->planYuv, c->ytiling, c->y3x1)) < 0) return r;
    if ((r = yuv3x2_prepare(c->planYuv, c->ytiling, c->y3x2)) < 0) return r;
    if ((r = yuv3r_prepare(c->planYuv, c->ytiling, c->y3r)) < 0) return r;
    if ((r = yuv32r_prepare(c->planYuv, c->ytiling, c->y32r)) < 0) return r;
    if ((r = yuv4r_prepare(c->planYuv, c->ytiling, c->y4r)) < 0) return r;
    if ((r = yuv4x1_prepare(c->planYuv, c->ytiling, c->y4x1)) < 0) return r;
    if (c->rgbViaPlanes && (r = rgb2y_prepare(c->planYuv, c->r2ys)) < 0) return r;
    if (!a.src16 && !c->rgbViaPlanes && (r = yuvg_prepare(c->planYuv, c->ytiling, c->yg)) < 0) return r;
    // (round 5) the same walker over 16-bit samples: P010LE / P016LE / planar 10- and 16-bit 4:2:0 sources (k_scale_yuvg16.hip) — one set of tables a context
    if (a.src16 >= 10 && !c->rgbViaPlanes && (r = yuvg_prepare16(c->planYuv, c->ytiling, c->yg)) < 0) return r;
    // ... and over a packed RGB24 / BGR24 source into a 4:2:0 frame (its own converter in front of the same 16-bit lines: hScale16To15_c with sh = 13)
    if (a.src16 == 3 && c->rgbViaPlanes && (r = yuvg_prepare16(c->planYuv, c->ytiling, c->yg)) < 0) return r;
    if (c->yg.ok) {
        YuvGArgs &g = c->gargs;
        std::memset(&g, 0, sizeof(g));
        g.src16 = (a.src16 >= 10 || a.src16 == 3) ? a.src16 : 0; g.hShift = g.src16 ? a.hShift : 7; g.hBias = g.src16 ? a.hBias : 0;
        g.r2y = a.r2y; g.rgbBgr = a.rgbBgr;
        g.dst16 = a.dst16; g.dstShift = a.dstShift;
        const YuvGTables &t = c->yg;
        int k = 0;
        auto up = [&](const std::vector<int32_t> &v, const int32_t *&out) {
            if (v.empty()) { out = nullptr; k++; return 0; }      // (an RGB up-scale has the fused block form only: no walker programs)
            int rr = c->dG[k].upload(v.data(), v.size() * 4);
            out = (const int32_t *)c->dG[k++].p;
            return rr;
        };
        if ((r = up(t.hL, g.hL)) < 0 || (r = up(t.hC, g.hC)) < 0 || (r = up(t.posL, g.posL)) < 0 || (r = up(t.posC, g.posC)) < 0) return r;
        for (int d = 0; d < 2; d++) {
            const YuvGQProg &m = t.yuvOut ? t.pl[d] : t.rgb[d];
            if ((r = up(m.prog, g.prog[d])) < 0 || (r = up(m.qfirst, g.qfirst[d])) < 0 || (r = up(m.qdone, g.qdone[d])) < 0) return r;
            if (t.yuvOut && ((r = up(t.pc[d].prog, g.progC[d])) < 0 || (r = up(t.pc[d].qfirst, g.qfirstC[d])) < 0 || (r = up(t.pc[d].qdone, g.qdoneC[d])) < 0)) return r;
        }
        if ((r = up(t.vtL, g.vtL)) < 0 || (r = up(t.vtC, g.vtC)) < 0) return r;
        if (!t.hCp.empty() && (r = up(t.hCp, g.hCp)) < 0) return r;      // (the fused block form of an RGB source: scale_yuvg_rgb2p_blk_kernel)
        g.f2PPL = t.f2PPL; std::memcpy(g.f2Pairs, t.f2Pairs, sizeof(g.f2Pairs));
        g.n4L = t.n4L; g.n4C = t.n4C; g.blkRows = t.blkRows; g.blkRowsC = t.blkRowsC;
        g.roundL = t.roundL; g.roundC = t.roundC;
        g.P = t.P; g.K = t.K; g.yuvOut = t.yuvOut;
    }
    if (!a.src16 && !c->rgbViaPlanes && (r = yuvu_prepare(c->planYuv, c->ytiling, c->yu)) < 0) return r;
    if (a.src16 >= 10 && !c->rgbViaPlanes && (r = yuvu_prepare16(c->planYuv, c->ytiling, c->yu)) < 0) return r;      // (round 5: k_scale_yuvu16.hip)
    if (c->yu.ok) {
        YuvUArgs &u = c->uargs;
        std::memset(&u, 0, sizeof(u));
        u.src16 = a.src16 >= 10 ? a.src16 : 0; u.hShift = u.src16 ? a.hShift : 7; u.hBias = u.src16 ? a.hBias : 0;
        u.dst16 = a.dst16; u.dstShift = a.dstShift;
        const YuvUTables &t = c->yu;
        int k = 0;
        auto up = [&](const std::vector<int32_t> &v, const int32_t *&out) {
            int rr = c->dU[k].upload(v.data(), v.size() * 4);
            out = (const int32_t *)c->dU[k++].p;
            return rr;
        };
        if ((r = up(t.hL, u.hL)) < 0 || (r = up(t.hC, u.hC)) < 0 || (r = up(t.posL, u.posL)) < 0 || (r = up(t.posC, u.posC)) < 0 ||
            (r = up(t.vtL, u.vtL)) < 0 || (r = up(t.vtC, u.vtC)) < 0 || (r = up(t.endL, u.endL)) < 0 || (r = up(t.endC, u.endC)) < 0 ||
            (r = up(t.firstL, u.firstL)) < 0 || (r = up(t.firstC, u.firstC)) < 0 || (r = up(t.lastL, u.lastL)) < 0 || (r = up(t.lastC, u.lastC)) < 0) return r;
        u.P = t.P; u.SD = t.SD; u.RL = t.RL; u.RC = t.RC; u.lead = t.lead; u.roundL = t.roundL; u.roundC = t.roundC; u.yuvOut = t.yuvOut;
    }
    if (!c->rgbViaPlanes && (r = yuvl_prepare(c->planYuv, c->ytiling, c->yl)) < 0) return r;
    if (c->yl.ok) {
        YuvLArgs &l = c->largs;
        std::memset(&l, 0, sizeof(l));
        const YuvLTables &t = c->yl;
        int k = 0;
        auto up = [&](const std::vector<int32_t> &v, const int32_t *&out) {
            int rr = c->dL[k].upload(v.data(), v.size() * 4);
            out = (const int32_t *)c->dL[k++].p;
            return rr;
        };
        if ((r = up(t.hL, l.hL)) < 0 || (r = up(t.hC, l.hC)) < 0 || (r = up(t.offL, l.offL)) < 0 || (r = up(t.offC, l.offC)) < 0) return r;
        l.P = t.P; l.nld = t.nld; l.RW = t.RW; l.yuvOut = t.yuvOut; l.fullChroma = t.fullChroma; l.dot4L = t.dot4L; l.dot4C = t.dot4C;
        l.pitchL = t.pitchL; l.pitchC = t.pitchC; l.pairRowsL = t.pairRowsL; l.pairRowsC = t.pairRowsC;
        l.baseU = t.baseU; l.baseV = t.baseV; l.frameInts = t.frameInts;
    }
    if (c->ytiling.TW == 0 && !c->yl.ok) return GMAT_ERR(ENOSYS);     // no tiling fits a workgroup's LDS and the lines form does not serve the formats
    if ((r = yuv2x_prepare(c->planYuv, c->ytiling, c->y2x)) < 0) return r;
    if (c->y2x.ok) {
        if ((r = c->dHLreg.upload(c->y2x.hLreg.data(), c->y2x.hLreg.size() * 4)) < 0) return r;
        if ((r = c->dHCreg.upload(c->y2x.hCreg.data(), c->y2x.hCreg.size() * 4)) < 0) return r;
        if ((r = c->dVrec.upload(c->y2x.vrec.data(), c->y2x.vrec.size() * 4)) < 0) return r;
        if ((r = c->dVrecC.upload(c->y2x.vrecC.data(), c->y2x.vrecC.size() * 4)) < 0) return r;
    }
    // (round 6) the tile kernel of k_scale19.hip on the 15-bit lines: any plane layout and depth in, 8- or 10-bit YUV out (kPlaneKernels: behind every walker, in front
    // of the tiled catch-all; in front of the lines form where the layouts differ and the ratio is below 4 : 1).  GMAT_T15=0: never
    c->urgb.ok = 0;
    if ((a.src16 == 10 || a.src16 == 16 || a.src16 == 17 || a.src16 == 18) && is_packed_rgb(c->dstFormat) && !c->rgbViaPlanes && is_plane_src(c->srcFormat) && !is_priv_planes(c->srcFormat))
        unit_rgb_plan(c->planYuv, t.vLumEff, t.lumRound.data(), t.fullChroma, c->urgb);
    c->t15.ok = 0; c->t15Mixed = false;
    {
        const char *k15 = GMAT_KNOB("GMAT_T15");
        const bool yuvDst = is_yuv8_src(c->dstFormat) || is_dst10(c->dstFormat);
        if (!(k15 && atoi(k15) == 0) && !c->rgbViaPlanes && yuvDst && is_plane_src(c->srcFormat) && !is_priv_planes(c->srcFormat)) {
            const bool srcSemi = c->srcFormat == GMAT_PIX_FMT_NV12 || is_p01x(c->srcFormat);
            const bool src444 = c->srcFormat == GMAT_PIX_FMT_YUV444P || c->srcFormat == GMAT_PIX_FMT_YUV444P16LE, dst444 = c->dstFormat == GMAT_PIX_FMT_YUV444P;
            r = s19_prepare(c->planYuv, t.vLumEff, t.vChrEff, a.src16 ? 2 : 1, scale16_kind(c->srcFormat), srcSemi ? 1 : 0, a.dstNv12 ? 1 : 0, 0, 0, c->t15,
                            a.dst16 ? 2 : 1, a.dstShift, a.src16 ? a.hShift : 7);
            if (r < 0 && r != GMAT_ERR(ENOSYS) && r != GMAT_ERR(EINVAL)) return r;
            // (measured, profiles/r06_sweep_tile15.txt: an 8-bit source whose two jobs need different pair counts — YUV444P into 4:2:0: four luma pairs, seven at the chroma's
            // 3 : 1 — runs the larger instance for both and loses to the tiled kernel, 11.3 against 7.7 us a frame; 16-bit sources win either way)
            auto np_of = [](int pairs) { return pairs <= 4 ? 4 : pairs <= 8 ? 8 : 0; };
            if (c->t15.ok && !a.src16 && np_of(c->planYuv.hLum.pairs) != np_of(c->planYuv.hChr.pairs)) c->t15.ok = 0;
            s19_unit_plan(c->planYuv, t.vLumEff, t.vChrEff, t.lumRound.data(), t.chrRound.data(), c->t15);
            if (c->t15.ok) {
                for (int j = 0; j < 2; j++) {
                    S19Job &J = c->t15.job[j];
                    if ((r = c->dT15[3 * j].upload(c->t15.colStart[j].data(), c->t15.colStart[j].size() * 4)) < 0) return r;
                    if ((r = c->dT15[3 * j + 1].upload(c->t15.rowStart[j].data(), c->t15.rowStart[j].size() * 4)) < 0) return r;
                    if ((r = c->dT15[3 * j + 2].upload(c->t15.rowCount[j].data(), c->t15.rowCount[j].size() * 4)) < 0) return r;
                    J.colStart = (const int32_t *)c->dT15[3 * j].p; J.rowStart = (const int32_t *)c->dT15[3 * j + 1].p; J.rowCount = (const int32_t *)c->dT15[3 * j + 2].p;
                    J.h = j ? a.hChr : a.hLum; J.v = j ? a.vChr : a.vLum;
                }
                c->t15Mixed = (srcSemi != (a.dstNv12 != 0)) || src444 || dst444;
            }
        }
    }
    c->yuvReady = true;
    return 0;
}

static int init_scaler(GmatSwsContext *c)
{
    if (c->rgbReady) return 0;
    const bool src_yuv = is_yuv420(c->srcFormat);
    const int planSrc = src_yuv ? GMAT_PIX_FMT_RGB24 : c->srcFormat;
    c->shift8Ident = -1;
    int r = build_scale_plan(c->plan, c->srcW, c->srcH, planSrc, c->dstW, c->dstH, c->dstFormat, c->flags, c->param);
    if (r < 0) return r;
    if ((r = scale_pick_tiling(c->plan, c->tiling)) < 0) return r;
    if ((r = rgb2s_prepare(c->plan, c->r2s)) < 0) return r;
    if (!src_yuv && (r = yuvg_rgbsrc_prepare(c->plan, c->rg)) < 0) return r;
    if (c->rg.ok) {
        YuvGArgs &g = c->rgargs;
        std::memset(&g, 0, sizeof(g));
        const YuvGTables &t = c->rg;
        int k = 0;
        auto up = [&](const std::vector<int32_t> &v, const int32_t *&out) {
            if (v.empty()) { out = nullptr; k++; return 0; }
            int rr = c->dRG[k].upload(v.data(), v.size() * 4);
            out = (const int32_t *)c->dRG[k++].p;
            return rr;
        };
        static const std::vector<int32_t> none;
        if ((r = up(t.hL, g.hL)) < 0 || (r = up(t.hC, g.hC)) < 0 || (r = up(t.posL, g.posL)) < 0 || (r = up(t.posC, g.posC)) < 0 ||
            (r = up(t.walkOk ? t.rgb[0].prog : none, g.prog[0])) < 0 || (r = up(t.walkOk ? t.rgb[0].qfirst : none, g.qfirst[0])) < 0 ||
            (r = up(t.walkOk ? t.rgb[0].qdone : none, g.qdone[0])) < 0 || (r = up(t.blkRows ? t.vtL : none, g.vtL)) < 0 ||
            (r = up(t.blkRows ? t.vtRnd : none, g.vtRnd)) < 0) return r;
        g.P = t.P; g.K = t.K; g.roundL = t.roundL; g.roundC = t.roundC; g.src16 = 3; g.hShift = 13;
        g.n4L = t.n4L; g.blkRows = t.blkRows; g.blkRows4 = t.blkRows4; g.blkPPL = t.blkPPL;
        g.srcW = c->srcW; g.srcH = c->srcH; g.chrSrcW = c->plan.chrSrcW; g.chrSrcH = c->plan.chrSrcH;
        g.dstW = c->dstW; g.dstH = c->dstH; g.chrDstW = c->dstW; g.chrDstH = c->dstH; g.dstFormat = c->dstFormat;
        g.rgbBgr = c->srcFormat == GMAT_PIX_FMT_BGR24;
    }
    ScaleArgs &a = c->args;
    std::memset(&a, 0, sizeof(a));
    if ((r = c->dHLum.upload(c->plan.hLum, false, a.hLum)) < 0) return r;
    if ((r = c->dHChr.upload(c->plan.hChr, false, a.hChr)) < 0) return r;
    if ((r = c->dVLum.upload(c->plan.vLum, true, a.vLum)) < 0) return r;
    const ScaleTiling &t = c->tiling;
    if ((r = c->dColStart.upload(t.colStart.data(), t.colStart.size() * 4)) < 0) return r;
    if ((r = c->dColCount.upload(t.colCount.data(), t.colCount.size() * 4)) < 0) return r;
    if ((r = c->dRowStart.upload(t.rowStart.data(), t.rowStart.size() * 4)) < 0) return r;
    if ((r = c->dRowCount.upload(t.rowCount.data(), t.rowCount.size() * 4)) < 0) return r;
    if ((r = c->dColMagic.upload(t.colMagic.data(), t.colMagic.size() * 4)) < 0) return r;
    a.colMagic = (const int32_t *)c->dColMagic.p; a.chromaDirect = t.chromaDirect;
    a.colStart = (const int32_t *)c->dColStart.p; a.colCount = (const int32_t *)c->dColCount.p;
    a.rowStart = (const int32_t *)c->dRowStart.p; a.rowCount = (const int32_t *)c->dRowCount.p;
    a.TH = t.TH; a.ntx = t.ntx; a.nty = t.nty; a.xcdRemap = t.xcdRemap;
    a.srcW = c->srcW; a.srcH = c->srcH; a.dstW = c->dstW; a.dstH = c->dstH;
    a.chrHalf = c->plan.chrSrcHSub;
    a.dstFormat = c->dstFormat;
    // internal colour model of the generic scaler: BT.601, limited range at both RGB ends
    a.r2y = make_rgb2yuv_consts(GMAT_SWS_CS_DEFAULT);
    a.y2r = make_yuv2rgb_consts(GMAT_SWS_CS_DEFAULT, false);
    if (src_yuv) {
        // the conversion stage in front uses the context's colourspace; the scaler's own output stage
        // always uses BT.601 limited (as an RGB24->RGB libswscale context does)
        c->y2r = make_yuv2rgb_consts(c->colorspace, c->srcFullRange != 0);
    }
    c->rgbReady = true;
    return 0;
}

static int init_rgb2yuv(GmatSwsContext *c)
{
    int r = build_scale_plan(c->planR2Y, c->srcW, c->srcH, c->srcFormat, c->dstW, c->dstH, c->dstFormat, c->flags, c->param);
    if (r < 0) return r;
    if ((r = rgb2yuv_prepare(c->planR2Y, c->r2y)) < 0) return r;
    if ((r = c->dR2YvChr.upload(c->planR2Y.vChr, c->r2y.round, c->r2yVChr)) < 0) return r;
    if ((r = c->dR2YrowStart.upload(c->r2y.rowStart.data(), c->r2y.rowStart.size() * 4)) < 0) return r;
    if ((r = c->dR2YrowCount.upload(c->r2y.rowCount.data(), c->r2y.rowCount.size() * 4)) < 0) return r;
    return 0;
}

// hscale19_kernel's sample kind of a plane source of the 19-bit path
static int scale16_kind(int srcFormat)
{
    const bool s16 = is_p01x(srcFormat) || pl16_depth(srcFormat) != 0;
    return srcFormat == GMAT_PIX_FMT_P010LE ? 10 : pl16_depth(srcFormat) == 10 ? 110 : srcFormat == GMAT_PIX_FMT_PRIV_RGB8_PLANES ? 14 : s16 ? 16 : 0;
}

static int init_scale16(GmatSwsContext *c)
{
    c->shift8Ident = -1;
    int r = build_scale_plan(c->plan16, c->srcW, c->srcH, c->srcFormat, c->dstW, c->dstH, c->dstFormat, c->flags, c->param, c->chrPos);
    if (r < 0) return r;
    ScalePlan &p = c->plan16;
    // a one-tap vertical luma filter goes through yuv2plane1_16_c, which does not read the coefficient: the X form
    // with 4096 gives the same value; semi-planar chroma always takes the X form (vscale.c:30-105)
    FilterBank vl = p.vLum, vc = p.vChr;
    if (is_rgb64(c->dstFormat)) {
        // packed_vscale's 1-tap forms (vscale.c:135-160 -> yuv2rgba64_1_c, output.c:1172-1272) do not read the
        // coefficients: luma as it is, chroma either its first line (uvalpha < 2048) or the mean of its two lines.
        // The 2-tap form (_2_c) and the X form are the same sums.
        const int lfs = vl.taps, cfs = vc.taps;
        for (int y = 0; y < c->dstH; y++) {
            int16_t *lf = &vl.coef[(size_t)y * lfs], *cf = &vc.coef[(size_t)y * cfs];
            const bool chr2 = cfs == 2 && cf[0] + cf[1] == 4096 && (unsigned)cf[1] <= 4096u;
            if (lfs == 1 && cfs == 1) { lf[0] = 4096; cf[0] = 4096; }
            else if (lfs == 1 && chr2) {
                lf[0] = 4096;
                if (cf[1] < 2048) { cf[0] = 4096; cf[1] = 0; } else { cf[0] = 2048; cf[1] = 2048; }
            }
        }
        pack_filter_pairs(vl); pack_filter_pairs(vc);
    } else {
        if (vl.taps == 1) { std::fill(vl.coef.begin(), vl.coef.end(), (int16_t)4096); pack_filter_pairs(vl); }
        // planar chroma goes through yuv2plane1_16_c too when its filter has one tap; interleaved chroma never does
        // (every planar 16-bit destination: round 6's fuzz_unit found YUV420P16LE left out — a one-tap chroma bank whose coefficient is not 4096, which extreme chroma
        // positions produce at a plane's first row, wrote zeros where libswscale writes the line)
        if (vc.taps == 1 && is_pl16_dst(c->dstFormat)) { std::fill(vc.coef.begin(), vc.coef.end(), (int16_t)4096); pack_filter_pairs(vc); }
    }
    const std::vector<int32_t> none(std::max(std::max(c->dstW, c->dstH), 1), 0);
    if ((r = c->f16[0].upload(p.hLum, none, c->d16[0])) < 0) return r;
    if ((r = c->f16[1].upload(p.hChr, none, c->d16[1])) < 0) return r;
    if ((r = c->f16[2].upload(vl, none, c->d16[2])) < 0) return r;
    if ((r = c->f16[3].upload(vc, none, c->d16[3])) < 0) return r;
    // (round 6) YUV destinations: one launch a frame, the 19-bit lines of a tile in LDS (k_scale19.hip; GMAT_S19=0: the two passes through HBM)
    c->s19.ok = 0;
    const char *ks19 = GMAT_KNOB("GMAT_S19");
    if (!(ks19 && atoi(ks19) == 0)) {
        const bool s16 = is_p01x(c->srcFormat) || pl16_depth(c->srcFormat) != 0;
        const bool srcSemi = c->srcFormat == GMAT_PIX_FMT_NV12 || is_p01x(c->srcFormat);
        const int rgb64 = c->dstFormat == GMAT_PIX_FMT_RGBA64LE ? 1 : c->dstFormat == GMAT_PIX_FMT_BGRA64LE ? 2 : 0;
        r = s19_prepare(p, vl, vc, s16 ? 2 : 1, scale16_kind(c->srcFormat), srcSemi ? 1 : 0, c->dstFormat == GMAT_PIX_FMT_P016LE ? 1 : 0,
                        rgb64, p.chrDstW == c->dstW ? 0 : 1, c->s19);
        if (r < 0 && r != GMAT_ERR(ENOSYS)) return r;
        s19_unit_plan(p, vl, vc, nullptr, nullptr, c->s19);              // (equal size, one-tap identity banks: the launch is scale19_unit_kernel's)
        if (c->s19.ok) {
            for (int j = 0; j < 2; j++) {
                S19Job &J = c->s19.job[j];
                if ((r = c->dS19[3 * j].upload(c->s19.colStart[j].data(), c->s19.colStart[j].size() * 4)) < 0) return r;
                if ((r = c->dS19[3 * j + 1].upload(c->s19.rowStart[j].data(), c->s19.rowStart[j].size() * 4)) < 0) return r;
                if ((r = c->dS19[3 * j + 2].upload(c->s19.rowCount[j].data(), c->s19.rowCount[j].size() * 4)) < 0) return r;
                J.colStart = (const int32_t *)c->dS19[3 * j].p; J.rowStart = (const int32_t *)c->dS19[3 * j + 1].p; J.rowCount = (const int32_t *)c->dS19[3 * j + 2].p;
                J.h = c->d16[j]; J.v = c->d16[2 + j];
            }
            if (!rgb64) {                    // (a 64-bit destination keeps the two passes' lines: the alpha plane of an RGBA source rides on them)
                for (int i = 0; i < 3; i++) if ((r = c->line16[i].reserve(0)) < 0) return r;
                return 0;
            }
        }
    }
    if ((r = c->line16[0].reserve((size_t)c->srcH * c->dstW * 4)) < 0) return r;
    if ((r = c->line16[1].reserve((size_t)p.chrSrcH * p.chrDstW * 4)) < 0) return r;
    if ((r = c->line16[2].reserve((size_t)p.chrSrcH * p.chrDstW * 4)) < 0) return r;
    return 0;
}

// prepares whichever scaler the current mode needs
static int ensure_scaler(GmatSwsContext *c)
{
    if ((is_yuv420(c->srcFormat) && (is_yuv8_src(c->dstFormat) || is_dst10(c->dstFormat))) ||
        c->srcFormat == GMAT_PIX_FMT_YUV444P || pl16_depth(c->srcFormat) || is_p01x(c->srcFormat) || c->rgbViaPlanes) {
        c->fused = 2;                        // planes are always scaled separately; there is no RGB stage to fuse
        return init_yuv_scaler(c);           // (a 4:4:4 source has no convert-then-scale form here either)
    }
    if (is_yuv420(c->srcFormat) && c->fused == 2) {
        int r = init_yuv_scaler(c);
        if (r == GMAT_ERR(ENOSYS)) {
            logf(LOG_WARNING, "gmat_sws: single-context YUV scaler unavailable for this geometry; using convert-then-scale");
            c->fused = 1;
            return init_scaler(c);
        }
        return r;
    }
    return init_scaler(c);
}

static YuvSrc yuv_src_of(int fmt, const uint8_t *const src[], const int stride[])
{
    YuvSrc s{};
    s.y = src[0]; s.ys = stride[0];
    s.u = src[1]; s.us = stride[1];
    s.nv12 = fmt == GMAT_PIX_FMT_NV12;
    if (!s.nv12) { s.v = src[2]; s.vs = stride[2]; }
    return s;
}

// ---- per-frame argument blocks of the single-context YUV scaler ------------------------------------------------
static bool al4(const void *p, int s) { return (((uintptr_t)p | (uintptr_t)s) & 3) == 0; }

static int prep_yuv_args(const GmatSwsContext *c, const uint8_t *const src[], const int srcStride[], uint8_t *const dst[],
                         const int dstStride[], YuvScaleArgs &ya)
{
    const bool planarYuv = c->srcFormat == GMAT_PIX_FMT_YUV420P || c->srcFormat == GMAT_PIX_FMT_YUV444P;
    ya = c->yargs;
    ya.y = src[0]; ya.ys = srcStride[0];
    ya.u = src[1]; ya.us = srcStride[1];
    ya.v = planarYuv ? src[2] : nullptr; ya.vs = planarYuv ? srcStride[2] : 0;
    ya.srcAligned = al4(src[0], srcStride[0]) &&
                    (ya.nv12 ? al4(src[1], srcStride[1])
                             : ((((uintptr_t)src[1] | (uintptr_t)src[2] | (uintptr_t)srcStride[1] |
                                  (uintptr_t)srcStride[2]) & 1) == 0));
    ya.srcAligned16 = ((((uintptr_t)src[0] | (uintptr_t)srcStride[0]) & 15) == 0) &&
                      (ya.nv12 ? ((((uintptr_t)src[1] | (uintptr_t)srcStride[1]) & 15) == 0)
                               : ((((uintptr_t)src[1] | (uintptr_t)srcStride[1] | (uintptr_t)src[2] | (uintptr_t)srcStride[2]) & 7) == 0));
    if (ya.src16 == 3) {
        ya.u = ya.v = nullptr; ya.us = ya.vs = 0;
        ya.srcAligned = al4(src[0], srcStride[0]); ya.srcAligned16 = 0;        // 12-byte pixel groups as three dwords
    } else if (ya.src16 >= 17) {
        if (!src[2] || (((uintptr_t)src[0] | (uintptr_t)src[1] | (uintptr_t)src[2] | (uintptr_t)srcStride[0] | (uintptr_t)srcStride[1] | (uintptr_t)srcStride[2]) & 1) != 0)
            return GMAT_ERR(EINVAL);
        ya.v = src[2]; ya.vs = srcStride[2];
        ya.srcAligned = al4(src[0], srcStride[0]); ya.srcAligned16 = 0;
    } else if (ya.src16) {
        // 16-bit samples: rows and planes 2-byte aligned at least; dword loads when 4-byte aligned
        if ((((uintptr_t)src[0] | (uintptr_t)src[1] | (uintptr_t)srcStride[0] | (uintptr_t)srcStride[1]) & 1) != 0) return GMAT_ERR(EINVAL);
        ya.srcAligned = al4(src[0], srcStride[0]) && al4(src[1], srcStride[1]);
        ya.srcAligned16 = 0;
    }
    ya.dst = dst[0]; ya.ds = dstStride[0];
    const int ybpp = bytes_per_pixel(c->dstFormat);
    ya.dstAligned = ybpp == 4 ? ((((uintptr_t)dst[0] | (uintptr_t)dstStride[0]) & 15) == 0) : al4(dst[0], dstStride[0]);
    if (c->dstFormat == GMAT_PIX_FMT_YUV420P10LE) {
        if (!dst[1] || !dst[2]) return GMAT_ERR(EINVAL);
        uintptr_t all = 0;
        for (int i = 0; i < 3; i++) all |= (uintptr_t)dst[i] | (uintptr_t)dstStride[i];
        if (all & 1) return GMAT_ERR(EINVAL);
        ya.dstU = dst[1]; ya.dsU = dstStride[1]; ya.dstV = dst[2]; ya.dsV = dstStride[2];
        ya.dstAligned = (all & 7) == 0;                      // 8-byte stores on every plane
    } else if (c->dstFormat == GMAT_PIX_FMT_P010LE) {
        if (!dst[1]) return GMAT_ERR(EINVAL);
        if ((((uintptr_t)dst[0] | (uintptr_t)dst[1] | (uintptr_t)dstStride[0] | (uintptr_t)dstStride[1]) & 1) != 0) return GMAT_ERR(EINVAL);
        ya.dstU = dst[1]; ya.dsU = dstStride[1]; ya.dstV = nullptr; ya.dsV = 0;
        // 8-byte luma stores, 16-byte chroma stores
        ya.dstAligned = ((((uintptr_t)dst[0] | (uintptr_t)dstStride[0]) & 7) == 0) && ((((uintptr_t)dst[1] | (uintptr_t)dstStride[1]) & 15) == 0);
    } else if (is_yuv8_src(c->dstFormat)) {
        const bool dnv = c->dstFormat == GMAT_PIX_FMT_NV12;
        if (!dst[1] || (!dnv && !dst[2])) return GMAT_ERR(EINVAL);
        ya.dstU = dst[1]; ya.dsU = dstStride[1];
        ya.dstV = dnv ? nullptr : dst[2]; ya.dsV = dnv ? 0 : dstStride[2];
        // one flag for all planes: dword stores for luma and planar chroma, 8-byte stores for NV12 chroma
        ya.dstAligned = al4(dst[0], dstStride[0]) &&
                        (dnv ? ((((uintptr_t)dst[1] | (uintptr_t)dstStride[1]) & 7) == 0)
                             : (al4(dst[1], dstStride[1]) && al4(dst[2], dstStride[2])));
    }
    // (an RGB source written to an RGB destination keeps both stages at the default, as every RGB -> RGB context here)
    ya.y2r = c->rgbViaPlanes ? make_yuv2rgb_consts(GMAT_SWS_CS_DEFAULT, false) : make_yuv2rgb_consts(c->colorspace, c->srcFullRange != 0);
    ya.prof = c->prof;
    ya.rangeConv = c->rangeConv;
    // swscale.c:263-264, 482-485 (should_dither): 8-bit planar output of a source deeper than 8 bits is dithered with ff_dither_8x8_128
    ya.dither8 = is_yuv8_src(c->dstFormat) && c->srcFormat != GMAT_PIX_FMT_PRIV_RGB8_PLANES && (is_p01x(c->srcFormat) || pl16_depth(c->srcFormat) != 0);
    return 0;
}

// the 2:1 kernel reads whole 16-byte chunks: luma and NV12 chroma rows 16-byte, planar chroma rows 8-byte aligned
static bool yuv2x_eligible(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    return c->y2x.ok && !c->rangeConv && ya.srcAligned &&
           ((((uintptr_t)ya.y | (uintptr_t)ya.ys | (uintptr_t)ya.u | (uintptr_t)ya.us) & 15) == 0) &&
           (ya.nv12 || ((((uintptr_t)ya.v | (uintptr_t)ya.vs) & 7) == 0 && (((uintptr_t)ya.u | (uintptr_t)ya.us) & 7) == 0));
}

static Yuv2xArgs make_yuv2x_args(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    Yuv2xArgs xa;
    std::memset(&xa, 0, sizeof(xa));
    xa.y = ya.y; xa.u = ya.u; xa.v = ya.v; xa.ys = ya.ys; xa.us = ya.us; xa.vs = ya.vs; xa.nv12 = ya.nv12;
    xa.srcW = ya.srcW; xa.srcH = ya.srcH; xa.chrSrcW = ya.chrSrcW; xa.chrSrcH = ya.chrSrcH;
    xa.dstW = ya.dstW; xa.dstH = ya.dstH;
    xa.dst = ya.dst; xa.ds = ya.ds; xa.dstFormat = ya.dstFormat; xa.dstAligned = ya.dstAligned;
    xa.dstU = ya.dstU; xa.dstV = ya.dstV; xa.dsU = ya.dsU; xa.dsV = ya.dsV; xa.dstNv12 = ya.dstNv12;
    xa.yuvOut = c->y2x.yuvOut; xa.chrDstW = ya.chrDstW; xa.chrDstH = ya.chrDstH; xa.P = c->y2x.P;
    xa.vrecC = (const int32_t *)c->dVrecC.p;
    xa.hLreg = (const int32_t *)c->dHLreg.p; xa.hCreg = (const int32_t *)c->dHCreg.p;
    xa.w0L = c->y2x.w0L; xa.w0C = c->y2x.w0C;
    xa.vrec = (const int32_t *)c->dVrec.p; xa.vLpairs = c->y2x.vLpairs; xa.vCpairs = c->y2x.vCpairs;
    xa.rowStartL = ya.rowStartL; xa.rowCountL = ya.rowCountL;
    xa.rowStartC = ya.rowStartC; xa.rowCountC = ya.rowCountC;
    xa.ntx = ya.ntx; xa.nty = ya.nty; xa.xcdRemap = ya.xcdRemap;
    xa.prof = ya.prof; xa.y2r = ya.y2r;
    static const bool noUni = GMAT_KNOB("GMAT_SCALE_NO_UNIFORM") != nullptr;
    if (!noUni) xa.uni = c->y2x.uni;
    return xa;
}

// the strip kernel: 4-byte aligned rows on both sides (its loads are dword-aligned at any window position)
static bool yuv2s_eligible(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    return c->y2s.ok && !c->rangeConv && ya.srcAligned && ya.dstAligned && !ya.prof &&
           (ya.nv12 || ((((uintptr_t)ya.u | (uintptr_t)ya.v | (uintptr_t)ya.us | (uintptr_t)ya.vs) & 3) == 0));
}

static Yuv2sArgs make_yuv2s_args(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    Yuv2sArgs sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.ys = ya.ys; sa.us = ya.us; sa.vs = ya.vs; sa.nv12 = ya.nv12;
    sa.srcW = ya.srcW; sa.srcH = ya.srcH; sa.chrSrcW = ya.chrSrcW; sa.chrSrcH = ya.chrSrcH;
    sa.dstW = ya.dstW; sa.dstH = ya.dstH; sa.ds = ya.ds; sa.dstFormat = ya.dstFormat;
    sa.np = c->y2s.np;
    for (int k = 0; k < 6; k++) { sa.hL[k] = c->y2s.hL[k]; sa.hC[k] = c->y2s.hC[k]; sa.vL[k] = c->y2s.vL[k]; }
    sa.lr = c->y2s.lr; sa.xcdRemap = ya.xcdRemap; sa.y2r = ya.y2r;
    return sa;
}

// the polyphase band walker (any ratio): dword-aligned planes on both sides, planar chroma planes of one pitch
static bool yuvg_eligible(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    if (!c->yg.ok || c->rangeConv || ya.prof) return false;
    uintptr_t all = (uintptr_t)ya.y | (uintptr_t)ya.ys | (uintptr_t)ya.u | (uintptr_t)ya.us | (uintptr_t)ya.dst | (uintptr_t)ya.ds;
    const bool semi = c->gargs.src16 ? (c->gargs.src16 == 10 || c->gargs.src16 == 16) : ya.nv12 != 0;
    if (c->gargs.src16 == 3) all = (uintptr_t)ya.y | (uintptr_t)ya.ys | (uintptr_t)ya.dst | (uintptr_t)ya.ds;      // (a packed RGB source: one plane)
    else if (!semi) { all |= (uintptr_t)ya.v | (uintptr_t)ya.vs; if (ya.us != ya.vs) return false; }
    if (c->yg.yuvOut) {
        all |= (uintptr_t)ya.dstU | (uintptr_t)ya.dsU;
        if (!ya.dstNv12) all |= (uintptr_t)ya.dstV | (uintptr_t)ya.dsV;
    }
    return (all & 3) == 0;
}

static YuvGArgs make_yuvg_args(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    YuvGArgs g = c->gargs;
    g.ys = ya.ys; g.us = ya.us; g.vs = ya.vs;
    g.nv12 = g.src16 == 3 ? ya.dstNv12 : g.src16 ? (g.src16 == 10 || g.src16 == 16) : ya.nv12;        // interleaved chroma (P010LE / P016LE: YuvScaleArgs' kinds 10 and 16; an RGB source: the destination's layout)
    g.dither8 = ya.dither8;
    g.srcW = ya.srcW; g.srcH = ya.srcH; g.chrSrcW = ya.chrSrcW; g.chrSrcH = ya.chrSrcH;
    g.dstW = ya.dstW; g.dstH = ya.dstH; g.chrDstW = ya.chrDstW; g.chrDstH = c->planYuv.chrDstH;
    g.ds = ya.ds; g.dsU = ya.dsU; g.dsV = ya.dsV; g.dstFormat = ya.dstFormat;
    g.xcdRemap = ya.xcdRemap; g.y2r = ya.y2r;
    g.srcPx = c->px4 ? 4 : 3;
    return g;
}

// the quad-lane walker (up-scales of any factor, short filters): the band walker's pointer rule (dword-aligned planes on both sides, planar
// chroma planes of one pitch); RGBA destinations store 16 bytes a lane.  Which contexts take it: up-scales on the vertical axis — where
// the band walker keeps 12 - 22 running sums or declines — at every launch size, and the short-filter down-scales it is eligible for (up to
// 1.75 : 1) in launches of more than three frames (measured, profiles/r04v_*: 32 frames a launch, band walker / this kernel: 1440p -> 1080p nv12
// 5.00 / 3.23 us, rgb24 5.09 / 4.33, 1080p -> 900p rgb24 3.67 / 3.58; ONE frame: the band walker's block form 10.0 / 11.6 / 9.5 against 10.5 /
// 14.1 / 10.6) — unless GMAT_QUAD_WALKER says otherwise (0: never, 2: wherever it is eligible)
static bool yuvu_eligible(const GmatSwsContext *c, const YuvScaleArgs &ya, int n)
{
    if (!c->yu.ok || c->rangeConv || ya.prof) return false;
    const char *qw = GMAT_KNOB("GMAT_QUAD_WALKER");
    const int mode = qw ? atoi(qw) : 1;
    // (16-bit samples, round 5: up-scales only — its six-pair instance loses the short-filter down-scales to the band walker: yuv420p10le 1080p -> 720p 4.96
    // against 3.2 us a frame, P010 -> rgb24 6.4 against 2.9: profiles/r05t_quad16.txt)
    if (mode == 0 || (mode == 1 && !(ya.dstH > ya.srcH || (n > 3 && !c->uargs.src16)))) return false;
    uintptr_t all = (uintptr_t)ya.y | (uintptr_t)ya.ys | (uintptr_t)ya.u | (uintptr_t)ya.us | (uintptr_t)ya.dst | (uintptr_t)ya.ds;
    const bool semiU = c->uargs.src16 ? (c->uargs.src16 == 10 || c->uargs.src16 == 16) : ya.nv12 != 0;
    if (!semiU) { all |= (uintptr_t)ya.v | (uintptr_t)ya.vs; if (ya.us != ya.vs) return false; }
    if (c->yu.yuvOut) {
        all |= (uintptr_t)ya.dstU | (uintptr_t)ya.dsU;
        if (!ya.dstNv12) all |= (uintptr_t)ya.dstV | (uintptr_t)ya.dsV;
    }
    return (all & 3) == 0;
}

static YuvUArgs make_yuvu_args(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    YuvUArgs g = c->uargs;
    g.ys = ya.ys; g.us = ya.us; g.vs = ya.vs;
    g.nv12 = g.src16 ? (g.src16 == 10 || g.src16 == 16) : ya.nv12;        // interleaved chroma (P010LE / P016LE: YuvScaleArgs' kinds 10 and 16)
    g.dither8 = ya.dither8;
    g.srcW = ya.srcW; g.srcH = ya.srcH; g.chrSrcW = ya.chrSrcW; g.chrSrcH = ya.chrSrcH;
    g.dstW = ya.dstW; g.dstH = ya.dstH; g.chrDstW = ya.chrDstW; g.chrDstH = c->planYuv.chrDstH;
    g.ds = ya.ds; g.dsU = ya.dsU; g.dsV = ya.dsV; g.dstFormat = ya.dstFormat;
    g.xcdRemap = ya.xcdRemap; g.y2r = ya.y2r;
    return g;
}

// the plane-walking 4:2:0 -> 4:2:0 kernel: dword loads and stores on every plane
static bool yuv2p_eligible(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    // its own destination rule, not ya.dstAligned: that flag carries the tiled kernel's 8-byte NV12 chroma stores, and a
    // frame with 4-byte pitches would be sent to the slower kernel for no reason of this one's
    // (the 10-bit twin stores 8 bytes per lane: 8-byte aligned rows there)
    const uintptr_t dall = (uintptr_t)ya.dst | (uintptr_t)ya.ds | (uintptr_t)ya.dstU | (uintptr_t)ya.dsU | (uintptr_t)ya.dstV | (uintptr_t)ya.dsV;
    const bool dst4 = (dall & (c->y2p.dstDepth == 10 ? 7 : 3)) == 0;
    return c->y2p.ok && !c->rangeConv && ya.srcAligned && dst4 && !ya.prof &&
           (ya.nv12 || ((((uintptr_t)ya.u | (uintptr_t)ya.v | (uintptr_t)ya.us | (uintptr_t)ya.vs) & 3) == 0));
}

// 8-bit 4:2:0 -> YUV444P at exactly 2:1 with identity chroma filters: the luma walker of scale_yuv2p_kernel + a chroma re-layout
static bool yuv2p444_eligible(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    const uintptr_t dl = (uintptr_t)ya.dst | (uintptr_t)ya.ds;
    return c->y2p.ok444 && !c->rangeConv && ya.srcAligned && (dl & 3) == 0 && !ya.prof && ya.dstU && ya.dstV;
}

static Yuv2pArgs make_yuv2p_args(const GmatSwsContext *c, const YuvScaleArgs &ya);
// the chroma of one frame: NV12 -> two planes (uv_deinterleave_kernel) or two plane copies
static int yuv2p444_chroma(const GmatSwsContext *c, const YuvScaleArgs &ya, const uint8_t *u, const uint8_t *v, uint8_t *dU, uint8_t *dV, hipStream_t stream)
{
    (void)c;
    if (ya.nv12) return launch_uv_relayout(1, u, ya.us, nullptr, 0, dU, ya.dsU, dV, ya.dsV, ya.chrSrcW, ya.chrSrcH, stream);
    int r = launch_copy2d(u, ya.us, dU, ya.dsU, ya.chrSrcW, ya.chrSrcH, stream);
    return r < 0 ? r : launch_copy2d(v, ya.vs, dV, ya.dsV, ya.chrSrcW, ya.chrSrcH, stream);
}
static const char *yuv2p444_name(const YuvScaleArgs &ya) { return ya.nv12 ? "scale_yuv2p_kernel<luma>+uv_deinterleave_kernel" : "scale_yuv2p_kernel<luma>+copy2d"; }

static Yuv2pArgs make_yuv2p_args(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    Yuv2pArgs pa;
    std::memset(&pa, 0, sizeof(pa));
    pa.ys = ya.ys; pa.us = ya.us; pa.vs = ya.vs;
    pa.nv12 = ya.nv12 || c->srcFormat == GMAT_PIX_FMT_P010LE;   // interleaved chroma on the SOURCE side
    pa.cross = c->y2p.cross;                                     // ... and the other layout on the destination's
    pa.srcDepth = c->y2p.srcDepth; pa.dstDepth = c->y2p.dstDepth; pa.dither8 = ya.dither8;
    pa.srcW = ya.srcW; pa.srcH = ya.srcH; pa.chrSrcW = ya.chrSrcW; pa.chrSrcH = ya.chrSrcH;
    pa.dstW = ya.dstW; pa.dstH = ya.dstH; pa.chrDstW = ya.chrDstW; pa.chrDstH = ya.chrDstH;
    pa.ds = ya.ds; pa.dsU = ya.dsU; pa.dsV = ya.dsV;
    pa.np = c->y2p.np;
    for (int k = 0; k < 6; k++) { pa.hL[k] = c->y2p.hL[k]; pa.hC[k] = c->y2p.hC[k]; pa.vL[k] = c->y2p.vL[k]; pa.vC[k] = c->y2p.vC[k]; }
    pa.lr = c->y2p.lr; pa.cr = c->y2p.cr; pa.xcdRemap = ya.xcdRemap;
    return pa;
}

// the 1:2 up-scale kernel: dword loads, 8-byte stores on every plane
static bool yuv1x2_eligible(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    const uintptr_t dall = (uintptr_t)ya.dst | (uintptr_t)ya.ds | (uintptr_t)ya.dstU | (uintptr_t)ya.dsU | (uintptr_t)ya.dstV | (uintptr_t)ya.dsV;
    return c->y1x2.ok && !c->rangeConv && ya.srcAligned && (dall & 7) == 0 && !ya.prof &&
           (ya.nv12 || ((((uintptr_t)ya.u | (uintptr_t)ya.v | (uintptr_t)ya.us | (uintptr_t)ya.vs) & 3) == 0));
}

static Yuv1x2Args make_yuv1x2_args(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    Yuv1x2Args ua;
    std::memset(&ua, 0, sizeof(ua));
    const Yuv1x2Tables &t = c->y1x2;
    ua.ys = ya.ys; ua.us = ya.us; ua.vs = ya.vs; ua.nv12 = ya.nv12;
    ua.srcW = ya.srcW; ua.srcH = ya.srcH; ua.chrSrcW = ya.chrSrcW; ua.chrSrcH = ya.chrSrcH;
    ua.ds = ya.ds; ua.dsU = ya.dsU; ua.dsV = ya.dsV;
    for (int k = 0; k < 2; k++) {
        ua.hLA[k] = t.hLA[k]; ua.hLB[k] = t.hLB[k]; ua.hLS0[k] = t.hLS0[k]; ua.hLS2[k] = t.hLS2[k];
        ua.vLA[k] = t.vLA[k]; ua.vLB[k] = t.vLB[k]; ua.vLS0[k] = t.vLS0[k]; ua.vLS2[k] = t.vLS2[k];
        ua.hCA[k] = t.hCA[k]; ua.hCB[k] = t.hCB[k]; ua.hCS0[k] = t.hCS0[k]; ua.hCS2[k] = t.hCS2[k];
        ua.vCA[k] = t.vCA[k]; ua.vCB[k] = t.vCB[k]; ua.vCS0[k] = t.vCS0[k]; ua.vCS2[k] = t.vCS2[k];
    }
    ua.lr = t.lr; ua.cr = t.cr; ua.xcdRemap = ya.xcdRemap;
    return ua;
}

// the 3:1 down-scale kernel: dword loads and stores on every plane
static bool yuv3x1_eligible(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    const uintptr_t dall = (uintptr_t)ya.dst | (uintptr_t)ya.ds | (uintptr_t)ya.dstU | (uintptr_t)ya.dsU | (uintptr_t)ya.dstV | (uintptr_t)ya.dsV;
    return c->y3x1.ok && !c->rangeConv && ya.srcAligned && (dall & 3) == 0 && !ya.prof &&
           (ya.nv12 || ((((uintptr_t)ya.u | (uintptr_t)ya.v | (uintptr_t)ya.us | (uintptr_t)ya.vs) & 3) == 0));
}

static Yuv3x1Args make_yuv3x1_args(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    Yuv3x1Args da;
    std::memset(&da, 0, sizeof(da));
    const Yuv3x1Tables &t = c->y3x1;
    da.ys = ya.ys; da.us = ya.us; da.vs = ya.vs; da.nv12 = ya.nv12;
    da.dstW = ya.srcW / 3; da.dstH = ya.srcH / 3; da.chrDstW = ya.chrSrcW / 3; da.chrDstH = ya.chrSrcH / 3;
    da.ds = ya.ds; da.dsU = ya.dsU; da.dsV = ya.dsV;
    for (int k = 0; k < 6; k++) { da.hL[k] = t.hL[k]; da.hC[k] = t.hC[k]; da.vL[k] = t.vL[k]; da.vC[k] = t.vC[k]; }
    da.lr = t.lr; da.cr = t.cr; da.xcdRemap = ya.xcdRemap;
    return da;
}

// the 3:2 down-scale kernel: dword loads, 8-byte stores on every plane
static bool yuv3x2_eligible(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    const uintptr_t dall = (uintptr_t)ya.dst | (uintptr_t)ya.ds | (uintptr_t)ya.dstU | (uintptr_t)ya.dsU | (uintptr_t)ya.dstV | (uintptr_t)ya.dsV;
    return c->y3x2.ok && !c->rangeConv && ya.srcAligned && (dall & 7) == 0 && !ya.prof &&
           (ya.nv12 || ((((uintptr_t)ya.u | (uintptr_t)ya.v | (uintptr_t)ya.us | (uintptr_t)ya.vs) & 3) == 0));
}

static Yuv3x2Args make_yuv3x2_args(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    Yuv3x2Args ea;
    std::memset(&ea, 0, sizeof(ea));
    const Yuv3x2Tables &t = c->y3x2;
    ea.ys = ya.ys; ea.us = ya.us; ea.vs = ya.vs; ea.nv12 = ya.nv12;
    ea.dstW = 2 * (ya.srcW / 3); ea.dstH = 2 * (ya.srcH / 3); ea.chrDstW = 2 * (ya.chrSrcW / 3); ea.chrDstH = 2 * (ya.chrSrcH / 3);
    ea.ds = ya.ds; ea.dsU = ya.dsU; ea.dsV = ya.dsV;
    for (int k = 0; k < 3; k++) {
        ea.hLA[k] = t.hLA[k]; ea.hLB[k] = t.hLB[k]; ea.hLS[k] = t.hLS[k]; ea.vLA[k] = t.vLA[k]; ea.vLB[k] = t.vLB[k]; ea.vLS[k] = t.vLS[k];
        ea.hCA[k] = t.hCA[k]; ea.hCB[k] = t.hCB[k]; ea.hCS[k] = t.hCS[k]; ea.vCA[k] = t.vCA[k]; ea.vCB[k] = t.vCB[k]; ea.vCS[k] = t.vCS[k];
    }
    ea.lr = t.lr; ea.cr = t.cr; ea.xcdRemap = ya.xcdRemap;
    return ea;
}

// the 3:1 NV12 -> packed RGB kernel: dword loads on both planes, the tiled kernel's destination rule (4- / 16-byte stores)
static bool yuv3r_eligible(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    return c->y3r.ok && !c->rangeConv && ya.nv12 && ya.srcAligned && ya.dstAligned && !ya.prof;
}

static Yuv3rArgs make_yuv3r_args(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    Yuv3rArgs a;
    std::memset(&a, 0, sizeof(a));
    const Yuv3rTables &t = c->y3r;
    a.ys = ya.ys; a.us = ya.us; a.dstW = ya.dstW; a.dstH = ya.dstH; a.ds = ya.ds; a.dstFormat = ya.dstFormat;
    for (int k = 0; k < 6; k++) { a.hL[k] = t.hL[k]; a.hC[k] = t.hC[k]; }
    int cl[12];                                          // the 11 vertical luma taps (slot 11: 0)
    for (int k = 0; k < 6; k++) { cl[2 * k] = (int16_t)(t.vL[k] & 0xFFFF); cl[2 * k + 1] = (int16_t)((uint32_t)t.vL[k] >> 16); }
    auto pk = [](int lo, int hi) { return (int32_t)((uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16)); };
    a.vP[0] = pk(cl[8], cl[9]); a.vP[1] = pk(cl[5], cl[6]); a.vP[2] = pk(cl[2], cl[3]); a.vP[3] = pk(0, cl[0]);
    a.vS[0] = cl[10]; a.vS[1] = cl[7]; a.vS[2] = cl[4]; a.vS[3] = cl[1];
    for (int k = 0; k < 3; k++) {
        a.cA[2 * k] = (int16_t)(t.vCA[k] & 0xFFFF); a.cA[2 * k + 1] = (int16_t)((uint32_t)t.vCA[k] >> 16);
        a.cB[2 * k] = (int16_t)(t.vCB[k] & 0xFFFF); a.cB[2 * k + 1] = (int16_t)((uint32_t)t.vCB[k] >> 16);
        a.cS[2 * k] = (int16_t)(t.vCS[k] & 0xFFFF); a.cS[2 * k + 1] = (int16_t)((uint32_t)t.vCS[k] >> 16);
    }
    a.lr = t.lr; a.cr = t.cr; a.y2r = ya.y2r;
    return a;
}

// the 4:1 4:2:0 -> 4:2:0 kernel: dword loads and stores on every plane
static bool yuv4x1_eligible(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    const uintptr_t dall = (uintptr_t)ya.dst | (uintptr_t)ya.ds | (uintptr_t)ya.dstU | (uintptr_t)ya.dsU | (uintptr_t)ya.dstV | (uintptr_t)ya.dsV;
    return c->y4x1.ok && !c->rangeConv && ya.srcAligned && (dall & 3) == 0 && !ya.prof &&
           (ya.nv12 || ((((uintptr_t)ya.u | (uintptr_t)ya.v | (uintptr_t)ya.us | (uintptr_t)ya.vs) & 3) == 0));
}

static Yuv4x1Args make_yuv4x1_args(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    Yuv4x1Args a;
    std::memset(&a, 0, sizeof(a));
    const Yuv4x1Tables &t = c->y4x1;
    a.ys = ya.ys; a.us = ya.us; a.vs = ya.vs; a.nv12 = ya.nv12;
    a.dstW = ya.srcW / 4; a.dstH = ya.srcH / 4; a.chrDstW = ya.chrSrcW / 4; a.chrDstH = ya.chrSrcH / 4;
    a.ds = ya.ds; a.dsU = ya.dsU; a.dsV = ya.dsV;
    for (int k = 0; k < 8; k++) { a.hL[k] = t.hL[k]; a.hC[k] = t.hC[k]; a.vL[k] = t.vL[k]; a.vC[k] = t.vC[k]; }
    a.lr = t.lr; a.cr = t.cr;
    return a;
}

// the 4:1 NV12 -> packed RGB kernel: 16-byte loads on both planes
static bool yuv4r_eligible(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    uintptr_t sall = (uintptr_t)ya.y | (uintptr_t)ya.ys | (uintptr_t)ya.u | (uintptr_t)ya.us;
    if (!ya.nv12) sall |= (uintptr_t)ya.v | (uintptr_t)ya.vs;
    return c->y4r.ok && !c->rangeConv && ya.src16 == 0 && (ya.nv12 || ya.v) && (sall & 3) == 0 && ya.dstAligned && !ya.prof;
}

static Yuv4rArgs make_yuv4r_args(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    Yuv4rArgs a;
    std::memset(&a, 0, sizeof(a));
    const Yuv4rTables &t = c->y4r;
    a.ys = ya.ys; a.us = ya.us; a.vs = ya.vs; a.nv12 = ya.nv12; a.dstW = ya.dstW; a.dstH = ya.dstH; a.ds = ya.ds; a.dstFormat = ya.dstFormat;
    for (int k = 0; k < 8; k++) { a.hL[k] = t.hL[k]; a.hC[k] = t.hC[k]; a.vL[k] = t.vL[k]; }
    for (int k = 0; k < 4; k++) a.vC[k] = t.vC[k];
    a.lr = t.lr; a.cr = t.cr; a.y2r = ya.y2r;
    return a;
}

// the 3:2 NV12 -> packed RGB kernel
static bool yuv32r_eligible(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    return c->y32r.ok && !c->rangeConv && ya.nv12 && ya.srcAligned && ya.dstAligned && !ya.prof;
}

static Yuv32rArgs make_yuv32r_args(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    Yuv32rArgs a;
    std::memset(&a, 0, sizeof(a));
    const Yuv32rTables &t = c->y32r;
    a.ys = ya.ys; a.us = ya.us; a.dstW = ya.dstW; a.dstH = ya.dstH; a.ds = ya.ds; a.dstFormat = ya.dstFormat;
    for (int k = 0; k < 3; k++) {
        a.hLA[k] = t.hLA[k]; a.hLB[k] = t.hLB[k]; a.hLS[k] = t.hLS[k]; a.hCA[k] = t.hCA[k]; a.hCB[k] = t.hCB[k]; a.hCS[k] = t.hCS[k];
        a.lA[2 * k] = (int16_t)(t.vLA[k] & 0xFFFF); a.lA[2 * k + 1] = (int16_t)((uint32_t)t.vLA[k] >> 16);
        a.lB[2 * k] = (int16_t)(t.vLB[k] & 0xFFFF); a.lB[2 * k + 1] = (int16_t)((uint32_t)t.vLB[k] >> 16);
        a.lS[2 * k] = (int16_t)(t.vLS[k] & 0xFFFF); a.lS[2 * k + 1] = (int16_t)((uint32_t)t.vLS[k] >> 16);
    }
    for (int i = 0; i < 4; i++) {
        for (int k = 0; k < 4; k++) a.cP[i][k] = t.cP[i][k];
        a.cS0[i] = t.cS0[i]; a.cS1[i] = t.cS1[i];
    }
    a.lr = t.lr; a.cr = t.cr; a.y2r = ya.y2r;
    return a;
}

// the 2:1 packed RGB -> 4:2:0 kernel: dword loads of the pixels, dword stores on luma and NV12 chroma (2-byte stores on planar chroma)
static bool rgb2y_eligible(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    const uintptr_t dall = (uintptr_t)ya.dst | (uintptr_t)ya.ds | (uintptr_t)ya.dstU | (uintptr_t)ya.dsU | (uintptr_t)ya.dstV | (uintptr_t)ya.dsV;
    return c->rgbViaPlanes && c->r2ys.ok && !c->rangeConv && ya.src16 == 3 && ya.srcAligned && (dall & 3) == 0 && !ya.prof &&
           !c->px4;           // (four-byte pixels, read as they are: the fused block form behind it in the table)
}

static Rgb2yArgs make_rgb2y_args(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    Rgb2yArgs a;
    std::memset(&a, 0, sizeof(a));
    const Rgb2yTables &t = c->r2ys;
    a.ss = ya.ys; a.srcW = c->srcW; a.srcH = c->srcH; a.dstW = c->dstW; a.dstH = c->dstH;
    a.ys = ya.ds; a.us = ya.dsU; a.vs = ya.dsV; a.nv12 = c->dstFormat == GMAT_PIX_FMT_NV12;
    for (int k = 0; k < 4; k++) { a.hL[k] = t.hL[k]; a.hC[k] = t.hC[k]; a.vL[k] = t.vL[k]; }
    for (int k = 0; k < 9; k++) a.vE[k] = t.vE[k];
    a.rnd = 64 << 12;
    auto pk = [](int lo, int hi) { return (int32_t)((uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16)); };
    const Rgb2YuvConsts &q = ya.r2y;
    if (ya.rgbBgr) { a.cY01 = pk(q.by, q.gy); a.cY2 = q.ry; a.cU01 = pk(q.bu, q.gu); a.cU2 = q.ru; a.cV01 = pk(q.bv, q.gv); a.cV2 = q.rv; }
    else           { a.cY01 = pk(q.ry, q.gy); a.cY2 = q.by; a.cU01 = pk(q.ru, q.gu); a.cU2 = q.bu; a.cV01 = pk(q.rv, q.gv); a.cV2 = q.bv; }
    return a;
}

// the name the plane-walking kernel reports by sample depths (one template, four instantiations per chroma layout)
static const char *yuv2p_name(const GmatSwsContext *c)
{
    const int s = c->y2p.srcDepth, d = c->y2p.dstDepth;
    if (c->y2p.cross) return s == 8 ? (d == 8 ? "scale_yuv2px_kernel" : "scale_yuv2px_kernel<8to10>") : (d == 8 ? "scale_yuv2px_kernel<10to8>" : "scale_yuv2px_kernel<10to10>");
    return s == 8 ? (d == 8 ? "scale_yuv2p_kernel" : "scale_yuv2p_kernel<8to10>") : (d == 8 ? "scale_yuv2p_kernel<10to8>" : "scale_yuv2p16_kernel");
}

// ---- the kernels of the single-context plane scaler, ONE record each, in priority order (round 4: rounds 2-3 spelled this list out
// twice — a 110-line if-chain in gmat_sws_scale and 13 use* flags with their own launch loops in sws_scale_frames_batched) -----------------
// eligible: this context / this frame's pointers and pitches satisfy the kernel's rule (a batch: every frame must; n = the frames of
// the call — only the one-frame form in front of the exact-ratio walkers looks at it).  launch: n frames
// of the table through one launch (n = 1: what sws_scale() issues).  The exact-ratio walkers' geometries are disjoint, so the order
// only matters inside a ratio (2:1: strip walker / 4:4:4 luma walker / plane walker before the tiled kernel) and for the two catch-alls.
// the lines form (k_scale_yuvl.hip).  Context level: which contexts MAY take it — the ones no walker of the table serves (a ratio beyond the
// band walker's 6.1 : 1, range conversion, filters its tables do not hold, a 4:4:4 end, full-chroma RGB), or every context the tiled kernel
// has no tiling for.  GMAT_LINES=0: never, 2: wherever the table reaches it.  Such a context owns its lines frames (stream_handoff_*).
static int lines_mode()
{
    const char *kn = GMAT_KNOB("GMAT_LINES");
    return kn ? atoi(kn) : 1;
}
static bool lines_context(const GmatSwsContext *c)
{
    if (c->mode != MODE_SCALE || !c->yuvReady || !c->yl.ok || c->fused != 2) return false;
    if (c->ytiling.TW == 0) return true;
    const int mode = lines_mode();
    if (mode != 1) return mode == 2;
    const bool walker = !c->rangeConv && (c->y2s.ok || c->y2p.ok || c->y2p.ok444 || c->y1x2.ok || c->y3x1.ok || c->y3x2.ok || c->y3r.ok || c->y32r.ok ||
                                          c->y4r.ok || c->y4x1.ok || c->y2x.ok || c->yg.ok || c->yu.ok);
    // NV12 <-> YUV420P within the band walker's range: the cascade (a sibling context in the source's layout on a walker + the re-layout) keeps it
    const bool crossWalk = is_yuv420(c->srcFormat) && is_yuv420(c->dstFormat) && c->srcFormat != c->dstFormat && !c->rangeConv &&
                           10 * c->srcW <= 61 * c->dstW && 10 * c->srcH <= 61 * c->dstH;
    return !walker && !crossWalk;
}
// frame level: dword-aligned source planes (the rows are read as aligned 16-byte pieces of a buffer resource), 8-bit destinations; and the
// launch's size — measured against the tiled kernel with every walker switched off (profiles/r04_lines.txt, us a frame, lines / tiled): 32 frames
// a launch the lines form wins at EVERY ratio (4K -> 480 x 270 4.8 / 28.1, -> 1600 x 900 12.6 / 15.2, 1080p -> 720p 5.2 / 6.8, 1440p -> 1080p
// 10.8 / 11.8, 720p -> 1080p 6.3 / 9.4); ONE frame a launch pays two launches and a lines frame through memory: from 2 : 1 on (4.5 : 1 20.2 / 25.4,
// 2.4 : 1 24.0 / 23.5, 3 : 2 18.9 / 12.3, 720p -> 1080p 18.5 / 14.0)
static bool yuvl_eligible(const GmatSwsContext *c, const YuvScaleArgs &ya, int n)
{
    if (!lines_context(c) || ya.prof || ya.src16 == 3) return false;
    // 16-bit samples in (scale_yuvl_h16_kernel, a row at a time) or 10-bit samples out: from 2 : 1 on whatever the launch (profiles/r04_lines_deep.txt,
    // lines / tiled: 32 frames a launch P010 4K -> 1600 x 900 18.2 / 28.9, -> 854 x 480 11.5 / 68.7, 3 : 2 and 2 : 3 a tie; one frame 29.5 / 35.8,
    // 24.7 / 70.2, 3 : 2 22.8 / 13.1)
    const bool deep = ya.src16 || ya.dst16 || ya.dither8;
    if (c->ytiling.TW != 0 && lines_mode() == 1 && !((n > 3 && !deep) || c->srcW >= 2 * c->dstW)) return false;
    uintptr_t all = (uintptr_t)ya.y | (uintptr_t)ya.ys | (uintptr_t)ya.u | (uintptr_t)ya.us;
    if (!ya.nv12) all |= (uintptr_t)ya.v | (uintptr_t)ya.vs;
    // planes that are not dword-aligned go to the tiled kernel — where there is one: a context it has no tiling for (a ratio or a filter beyond a tile's
    // LDS) was accepted on the strength of this form and must not fail frame by frame (found by fuzz_ref_core: yuv420p 321 x 432 -> 81 x 108 sinc, chroma
    // rows of 161 bytes, -ENOSYS from sws_scale); the launch then reads dword-aligned copies of the planes (lines_stage)
    return (all & 3) == 0 || c->ytiling.TW == 0;
}
static int lines_prepare(GmatSwsContext *c, int nframes)
{
    if (c->linesFrames < nframes) {
        if (c->linesBuf) { (void)hipFree(c->linesBuf); c->linesBuf = nullptr; c->linesFrames = 0; }
        GMAT_HIP_CHECK(hipMalloc((void **)&c->linesBuf, c->yl.frameInts * sizeof(int32_t) * nframes));
        c->linesFrames = nframes;
    }
    return 0;
}
// source planes that are not dword-aligned, in a context the tiled kernel has no tiling for: the launch reads dword-aligned COPIES (pitch a multiple of 256,
// device-to-device on the call's stream) — pass H reads a row as aligned 16-byte pieces of a buffer resource whose range ends on a dword
static int lines_stage(GmatSwsContext *c, YuvScaleArgs &ya, hipStream_t st, Yuv2xFrames &fr, int n)
{
    const int bps = ya.src16 ? 2 : 1;
    const bool semi = ya.src16 ? ya.src16 < 17 : ya.nv12 != 0;              // interleaved chroma
    const size_t rowL = (size_t)ya.srcW * bps, rowC = (size_t)(semi ? 2 * ya.chrSrcW : ya.chrSrcW) * bps;
    const size_t pitchL = (rowL + 255) & ~(size_t)255, pitchC = (rowC + 255) & ~(size_t)255;
    const size_t bytesL = pitchL * ya.srcH, bytesC = pitchC * ya.chrSrcH, frame = bytesL + (semi ? 1 : 2) * bytesC;
    if (c->linesStageBytes < frame * n) {
        if (c->linesStage) { (void)hipFree(c->linesStage); c->linesStage = nullptr; c->linesStageBytes = 0; }
        GMAT_HIP_CHECK(hipMalloc((void **)&c->linesStage, frame * n));
        c->linesStageBytes = frame * n;
    }
    for (int i = 0; i < n; i++) {
        uint8_t *b = c->linesStage + frame * i;
        GMAT_HIP_CHECK(hipMemcpy2DAsync(b, pitchL, fr.y[i], ya.ys, rowL, ya.srcH, hipMemcpyDeviceToDevice, st));
        GMAT_HIP_CHECK(hipMemcpy2DAsync(b + bytesL, pitchC, fr.u[i], ya.us, rowC, ya.chrSrcH, hipMemcpyDeviceToDevice, st));
        if (!semi) GMAT_HIP_CHECK(hipMemcpy2DAsync(b + bytesL + bytesC, pitchC, fr.v[i], ya.vs, rowC, ya.chrSrcH, hipMemcpyDeviceToDevice, st));
        fr.y[i] = b; fr.u[i] = b + bytesL; fr.v[i] = semi ? nullptr : b + bytesL + bytesC;
    }
    ya.ys = (int)pitchL; ya.us = (int)pitchC; ya.vs = (int)pitchC;
    return 0;
}
static YuvLArgs make_yuvl_args(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    YuvLArgs l = c->largs;
    l.ys = ya.ys; l.us = ya.us; l.vs = ya.vs; l.nv12 = ya.nv12;
    l.srcW = ya.srcW; l.srcH = ya.srcH; l.chrSrcW = ya.chrSrcW; l.chrSrcH = ya.chrSrcH;
    l.dstW = ya.dstW; l.dstH = ya.dstH; l.chrDstW = ya.chrDstW; l.chrDstH = c->planYuv.chrDstH;
    l.ds = ya.ds; l.dsU = ya.dsU; l.dsV = ya.dsV; l.dstFormat = ya.dstFormat; l.dstAligned = ya.dstAligned; l.dstNv12 = ya.dstNv12;
    l.rangeConv = ya.rangeConv;
    l.src16 = ya.src16; l.hShift = ya.hShift; l.hBias = ya.hBias; l.dst16 = ya.dst16; l.dstShift = ya.dstShift; l.dither8 = ya.dither8;
    l.vLum = ya.vLum; l.vChr = ya.vChr; l.y2r = ya.y2r;
    l.inter = c->linesBuf;
    return l;
}

// the tile kernel on the 15-bit lines: n frames of the context's geometry (kPlaneKernels)
static int launch_tile15(const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n)
{
    const bool semiS = c->srcFormat == GMAT_PIX_FMT_NV12 || is_p01x(c->srcFormat), semiD = ya.dstNv12 != 0;
    S19Args a;
    std::memset(&a, 0, sizeof(a));
    a.job[0] = c->t15.job[0]; a.job[1] = c->t15.job[1];
    a.job[0].rawStride[0] = ya.ys; a.job[1].rawStride[0] = ya.us; a.job[1].rawStride[1] = semiS ? 0 : ya.vs;
    a.job[0].ds[0] = ya.ds; a.job[1].ds[0] = ya.dsU; a.job[1].ds[1] = semiD ? ya.dsU : ya.dsV;
    a.job[0].rc = ya.rangeConv ? 4 + ya.rangeConv : 0; a.job[1].rc = ya.rangeConv ? 6 + ya.rangeConv : 0;     // (s19_range: 5 / 6 luma to / from full range, 7 / 8 chroma)
    a.job[0].dither8 = a.job[1].dither8 = ya.dither8;
    uintptr_t sA = (uintptr_t)ya.ys | (uintptr_t)ya.us | (semiS ? 0 : (uintptr_t)ya.vs), dA = (uintptr_t)ya.ds | (uintptr_t)ya.dsU | (semiD ? 0 : (uintptr_t)ya.dsV);
    for (int i = 0; i < n; i++) {
        sA |= (uintptr_t)fr.y[i] | (uintptr_t)fr.u[i] | (semiS ? 0 : (uintptr_t)fr.v[i]);
        dA |= (uintptr_t)fr.dst[i] | (uintptr_t)fr.dstU[i] | (semiD ? 0 : (uintptr_t)fr.dstV[i]);
    }
    if ((ya.src16 && (sA & 1)) || (ya.dst16 && (dA & 1))) return GMAT_ERR(EINVAL);        // 16-bit samples sit on even addresses
    a.srcAl4 = (sA & 3) == 0; a.dstAl4 = (dA & 3) == 0;
    a.unit = c->t15.unit; a.srcAl16 = (sA & 15) == 0; a.dstAl16 = (dA & 15) == 0;
    return launch_scale19(a, c->t15.np, c->t15.ldsBytes, st, &fr, n);
}

struct PlaneKernel {
    bool (*eligible)(const GmatSwsContext *c, const YuvScaleArgs &ya, int n);      // n: frames of the whole call
    const char *(*name)(const GmatSwsContext *c, const YuvScaleArgs &ya, int n);
    int (*launch)(const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t stream, const Yuv2xFrames &fr, int n);
};
// 16-bit 4:2:0 sources into packed 8-bit RGB at equal size: this call's planes and pitches on 16-byte addresses (unit_rgb_kernel's loads and 8-byte pieces)
static bool unit_rgb_eligible(const GmatSwsContext *c, const YuvScaleArgs &ya)
{
    // (src16: 10 P010LE, 16 P016LE — interleaved chroma —, 17 / 18 planar 16- / 10-bit)
    if (!c->urgb.ok || ya.prof || !(ya.src16 == 10 || ya.src16 == 16 || ya.src16 == 17 || ya.src16 == 18)) return false;
    const bool semi = ya.src16 == 10 || ya.src16 == 16;
    uintptr_t all = (uintptr_t)ya.y | (uintptr_t)ya.ys | (uintptr_t)ya.u | (uintptr_t)ya.us | (uintptr_t)ya.dst | (uintptr_t)ya.ds;
    if (!semi) { if (!ya.v) return false; all |= (uintptr_t)ya.v | (uintptr_t)ya.vs; }
    return (all & 15) == 0;
}
static int launch_unit_rgb_frames(const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n)
{
    UnitRgbArgs a;
    std::memset(&a, 0, sizeof(a));
    a.w = ya.dstW; a.h = ya.dstH; a.chrH = ya.chrSrcH;
    a.semi = (ya.src16 == 10 || ya.src16 == 16) ? 1 : 0; a.shr6 = ya.src16 == 10 ? 1 : 0;       // (P010LE: ten bits in the high end)
    a.ys = ya.ys; a.us = ya.us; a.vs = a.semi ? 0 : ya.vs; a.ds = ya.ds;
    a.shl = ya.hShift <= 14 ? 14 - ya.hShift : 0; a.shr = ya.hShift > 14 ? ya.hShift - 14 : 0; a.maxv = 32767;      // hScale16To15_c with one coefficient of 2^14
    a.coefL = c->urgb.coefL; a.roundL = c->urgb.roundL;
    a.vChr = ya.vChr;
    a.px = bytes_per_pixel(c->dstFormat); a.bgr = (c->dstFormat == GMAT_PIX_FMT_BGR24 || c->dstFormat == GMAT_PIX_FMT_BGRA) ? 1 : 0;
    a.y2r = ya.y2r;
    return launch_unit_rgb(a, st, &fr, n);
}

static const PlaneKernel kPlaneKernels[] = {
    // (round 6, last hours) P010 / P016 / planar 10- / 16-bit 4:2:0 into packed 8-bit RGB at equal size — no unscaled converter in libswscale, 7.3-9.5 us a 1080p frame on the
    // 16-bit walker: the unit form of the RGB writer
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int) { return unit_rgb_eligible(c, ya); },
     [](const GmatSwsContext *, const YuvScaleArgs &, int) -> const char * { return "unit_rgb_kernel"; },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) { return launch_unit_rgb_frames(c, ya, st, fr, n); }},
    // (round 6, last third) equal size with one-tap identity banks — yuv2yuv_cuda's same-size conversions between depths and layouts that libswscale runs through its
    // generic scaler: the tile kernel's unit form, a sample a multiply and a shift (in front of every walker: they took 5.2-6.5 us a 1080p frame for what is a copy's bytes)
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int) { return c->t15.ok && c->t15.unit && !ya.prof; },
     [](const GmatSwsContext *, const YuvScaleArgs &, int) -> const char * { return "scale19_unit_kernel"; },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) { return launch_tile15(c, ya, st, fr, n); }},
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int) { return yuv2s_eligible(c, ya); },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, int n) -> const char * {
         return c->y2s.np == 6 ? "scale_yuv2s_np_kernel<6>" : yuv2s_block_form(make_yuv2s_args(c, ya), n) ? "scale_yuv2s_blk_kernel" : "scale_yuv2s_kernel"; },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) { return launch_scale_yuv2s(make_yuv2s_args(c, ya), st, &fr, n); }},
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int) { return yuv2p444_eligible(c, ya); },
     [](const GmatSwsContext *, const YuvScaleArgs &ya, int) -> const char * { return yuv2p444_name(ya); },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) {
         // luma of all frames in one launch, the chroma re-layout frame by frame (its kernels take one frame)
         Yuv2pArgs pa = make_yuv2p_args(c, ya);
         pa.lumaOnly = 1; pa.cross = 0; pa.srcDepth = pa.dstDepth = 8;
         int r = launch_scale_yuv2p(pa, st, &fr, n);
         for (int i = 0; i < n && r >= 0; i++) r = yuv2p444_chroma(c, ya, fr.u[i], ya.nv12 ? nullptr : fr.v[i], fr.dstU[i], fr.dstV[i], st);
         return r; }},
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int) { return yuv2p_eligible(c, ya); },
     [](const GmatSwsContext *c, const YuvScaleArgs &, int) -> const char * { return yuv2p_name(c); },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) { return launch_scale_yuv2p(make_yuv2p_args(c, ya), st, &fr, n); }},
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int) { return yuv1x2_eligible(c, ya); },
     [](const GmatSwsContext *, const YuvScaleArgs &, int) -> const char * { return "scale_yuv1x2_kernel"; },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) { return launch_scale_yuv1x2(make_yuv1x2_args(c, ya), st, &fr, n); }},
    // a call of few frames (what sws_scale() / filter_frame() issue): the band walker's block-cooperative form IN FRONT of the exact-ratio
    // walkers it beats at that launch size (measured, profiles/r04p_blk_vs_ratio_kernels.txt: per-ratio kernel / block form, us a launch) —
    // 4:1 at one to three frames (-> nv12 11.2 / 8.6, 16.6 / 12.6, 20.1 / 16.5; -> rgb24 10.2 / 9.1, 16.0 / 15.4, 21.4 / 18.6), 3:1 and 3:2 to
    // packed RGB at ONE frame (9.9 / 9.1, 8.7 / 7.8; at two and three frames the exact-ratio walkers are level or ahead).  The 3:1 / 3:2 plane
    // walkers keep their frames at every size (7.4 against 8.8, 5.1 against 6.6).  GMAT_BLOCK_FIRST=0: never in front
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int n) {
         if (c->gargs.src16 || !yuvg_eligible(c, ya) || yuv3x1_eligible(c, ya) || yuv3x2_eligible(c, ya)) return false;
         const bool four = yuv4r_eligible(c, ya) || yuv4x1_eligible(c, ya), threeRgb = yuv32r_eligible(c, ya) || yuv3r_eligible(c, ya);
         if (!((four && n <= 3) || (threeRgb && n == 1))) return false;   // (every other context reaches the form in the band walker's own place)
         const char *bf = GMAT_KNOB("GMAT_BLOCK_FIRST");
         return !(bf && !atoi(bf)) && yuvg_block_form(make_yuvg_args(c, ya), n); },
     [](const GmatSwsContext *, const YuvScaleArgs &, int) -> const char * { return "scale_yuvg_blk_kernel"; },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) { return launch_scale_yuvg(make_yuvg_args(c, ya), st, &fr, n); }},
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int) { return yuv4r_eligible(c, ya); },
     [](const GmatSwsContext *, const YuvScaleArgs &, int) -> const char * { return "scale_yuv4r_kernel"; },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) { return launch_scale_yuv4r(make_yuv4r_args(c, ya), st, &fr, n); }},
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int) { return yuv32r_eligible(c, ya); },
     [](const GmatSwsContext *, const YuvScaleArgs &, int) -> const char * { return "scale_yuv32r_kernel"; },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) { return launch_scale_yuv32r(make_yuv32r_args(c, ya), st, &fr, n); }},
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int) { return yuv3r_eligible(c, ya); },
     [](const GmatSwsContext *, const YuvScaleArgs &, int) -> const char * { return "scale_yuv3r_kernel"; },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) { return launch_scale_yuv3r(make_yuv3r_args(c, ya), st, &fr, n); }},
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int) { return rgb2y_eligible(c, ya); },
     [](const GmatSwsContext *, const YuvScaleArgs &, int) -> const char * { return "scale_rgb2y_kernel"; },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) { return launch_scale_rgb2y(make_rgb2y_args(c, ya), st, &fr, n); }},
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int) { return yuv3x1_eligible(c, ya); },
     [](const GmatSwsContext *, const YuvScaleArgs &, int) -> const char * { return "scale_yuv3x1_kernel"; },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) { return launch_scale_yuv3x1(make_yuv3x1_args(c, ya), st, &fr, n); }},
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int) { return yuv3x2_eligible(c, ya); },
     [](const GmatSwsContext *, const YuvScaleArgs &, int) -> const char * { return "scale_yuv3x2_kernel"; },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) { return launch_scale_yuv3x2(make_yuv3x2_args(c, ya), st, &fr, n); }},
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int) { return yuv4x1_eligible(c, ya); },
     [](const GmatSwsContext *, const YuvScaleArgs &, int) -> const char * { return "scale_yuv4x1_kernel"; },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) { return launch_scale_yuv4x1(make_yuv4x1_args(c, ya), st, &fr, n); }},
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int) { return yuv2x_eligible(c, ya); },                                   // the tiled 2:1 kernel of round 1: behind every 2:1 walker
     [](const GmatSwsContext *c, const YuvScaleArgs &, int) -> const char * { return c->y2x.yuvOut ? "scale_yuv2x_kernel<yuv>" : "scale_yuv2x_kernel"; },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) {
         const Yuv2xArgs xa = make_yuv2x_args(c, ya);
         return n == 1 ? launch_scale_yuv2x(xa, c->ytiling.rowsL, c->ytiling.rowsC, c->y2x.ok, st)      // (one frame: the pointers of the argument block)
                       : launch_scale_yuv2x(xa, c->ytiling.rowsL, c->ytiling.rowsC, c->y2x.ok, st, &fr, n); }},
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int n) { return yuvu_eligible(c, ya, n); },                                    // up-scales of any factor: the quad-lane walker
     [](const GmatSwsContext *c, const YuvScaleArgs &, int) -> const char * { return c->uargs.src16 ? "scale_yuvu16_kernel" : "scale_yuvu_kernel"; },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) {
         return c->uargs.src16 ? launch_scale_yuvu16(make_yuvu_args(c, ya), st, &fr, n) : launch_scale_yuvu(make_yuvu_args(c, ya), st, &fr, n); }},
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int) { return yuvg_eligible(c, ya); },                                    // any ratio: the polyphase band walker
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, int n) -> const char * {
         if (c->gargs.src16 == 3 && yuvg_rgb2p_fused(make_yuvg_args(c, ya), n)) return "scale_yuvg_rgb2p_blk_kernel";       // (an RGB source: the fused block form)
         if (c->gargs.src16) return yuvg_block_form16(make_yuvg_args(c, ya), n) ? "scale_yuvg16_blk_kernel" : "scale_yuvg16_kernel";      // (16-bit samples: k_scale_yuvg16.hip)
         return yuvg_block_form(make_yuvg_args(c, ya), n) ? "scale_yuvg_blk_kernel" : "scale_yuvg_kernel"; },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) {
         return c->gargs.src16 ? launch_scale_yuvg16(make_yuvg_args(c, ya), st, &fr, n) : launch_scale_yuvg(make_yuvg_args(c, ya), st, &fr, n); }},
    // (round 6) the tile kernel on the 15-bit lines where the two ends' plane layouts differ (semi-planar <-> planar, a 4:4:4 end) below 4 : 1: in front of the lines form's
    // two launches (profiles/r06_sweep_*.txt: 6.7-9.1 us a 1080p -> 720p frame there, 8-12 on the tiled kernel behind it)
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int) {
         // (8-bit ends keep the lines form: 6.7 against 7.8-11 us a 1080p -> 720p frame batched, profiles/r06_sweep_tile15.txt; the deep ones — 16-bit samples in, 10-bit out,
         // the dithered 8-bit output — ran its untuned 16-bit pass or fell through to the tiled kernel)
         return c->t15.ok && c->t15Mixed && !ya.prof && (ya.src16 || ya.dst16 || ya.dither8) && c->srcW < 4 * c->dstW && c->srcH < 4 * c->dstH; },
     [](const GmatSwsContext *c, const YuvScaleArgs &, int) -> const char * { return c->t15.unit ? "scale19_unit_kernel" : "scale19_kernel"; },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) { return launch_tile15(c, ya, st, fr, n); }},
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int n) { return yuvl_eligible(c, ya, n); },                                    // what no walker takes: the lines form
     [](const GmatSwsContext *, const YuvScaleArgs &ya, int) -> const char * { return ya.src16 ? "scale_yuvl_h16_kernel+scale_yuvl_v_kernel" : "scale_yuvl_h_kernel+scale_yuvl_v_kernel"; },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) {
         GmatSwsContext *m = const_cast<GmatSwsContext *>(c);             // (the lines frame is allocated on first use and grows with the launch)
         if (int r = lines_prepare(m, n); r < 0) return r;
         m->interTouched = true;
         uintptr_t all = (uintptr_t)ya.y | (uintptr_t)ya.ys | (uintptr_t)ya.u | (uintptr_t)ya.us;
         if (!ya.nv12) all |= (uintptr_t)ya.v | (uintptr_t)ya.vs;
         for (int i = 0; i < n; i++) all |= (uintptr_t)fr.y[i] | (uintptr_t)fr.u[i] | (ya.nv12 ? 0 : (uintptr_t)fr.v[i]);
         if (all & 3) {                                                    // (only where the tiled kernel has no tiling: yuvl_eligible)
             YuvScaleArgs ya2 = ya; Yuv2xFrames fr2 = fr;
             if (int r = lines_stage(m, ya2, st, fr2, n); r < 0) return r;
             return launch_scale_yuvl(make_yuvl_args(c, ya2), st, &fr2, n);
         }
         return launch_scale_yuvl(make_yuvl_args(c, ya), st, &fr, n); }},
    {[](const GmatSwsContext *c, const YuvScaleArgs &ya, int) { return c->t15.ok && !ya.prof; },       // ... and in front of the tiled catch-all wherever it has a plan
     [](const GmatSwsContext *c, const YuvScaleArgs &, int) -> const char * { return c->t15.unit ? "scale19_unit_kernel" : "scale19_kernel"; },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) { return launch_tile15(c, ya, st, fr, n); }},
    {[](const GmatSwsContext *, const YuvScaleArgs &, int) { return true; },      // everything else: the tiled plane scaler
     [](const GmatSwsContext *c, const YuvScaleArgs &, int) -> const char * { return yuvscale_kernel_name(c->ytiling); },
     [](const GmatSwsContext *c, const YuvScaleArgs &ya, hipStream_t st, const Yuv2xFrames &fr, int n) {
         if (c->ytiling.TW == 0) return GMAT_ERR(ENOSYS);                 // no tiling fits the LDS and this frame's planes are not the lines form's
         return n == 1 ? launch_scale_yuv(ya, c->ytiling, st) : launch_scale_yuv(ya, c->ytiling, st, &fr, n); }},
};
constexpr int kNumPlaneKernels = (int)(sizeof(kPlaneKernels) / sizeof(kPlaneKernels[0]));
// the records that are not walkers of the context's own (the cascade's rule): the two unit forms in front, the tile kernel on the 15-bit lines second and fourth from the end
static bool plane_record_is_tile(int k) { return k == 0 || k == 1 || k == kNumPlaneKernels - 2 || k == kNumPlaneKernels - 4; }

// argument block of the strip-walking packed-RGB scaler
static Rgb2sArgs make_rgb2s_args(const GmatSwsContext *c, int srcStride, int dstStride, bool srcBgr)
{
    Rgb2sArgs ra;
    std::memset(&ra, 0, sizeof(ra));
    ra.ss = srcStride; ra.srcW = c->srcW; ra.srcH = c->srcH; ra.dstW = c->dstW; ra.dstH = c->dstH; ra.ds = dstStride; ra.dstFormat = c->dstFormat;
    for (int k = 0; k < 4; k++) { ra.hL[k] = c->r2s.hL[k]; ra.vL[k] = c->r2s.vL[k]; }
    ra.rnd = 1 << 9;
    const Rgb2YuvConsts &q = c->args.r2y;
    auto pk = [](int lo, int hi) { return (int32_t)((uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16)); };
    // coefficients in the byte order of the pixels: (first, second) as an int16 pair, third alone
    if (srcBgr) { ra.cY01 = pk(q.by, q.gy); ra.cY2 = q.ry; ra.cU01 = pk(q.bu, q.gu); ra.cU2 = q.ru; ra.cV01 = pk(q.bv, q.gv); ra.cV2 = q.rv; }
    else        { ra.cY01 = pk(q.ry, q.gy); ra.cY2 = q.by; ra.cU01 = pk(q.ru, q.gu); ra.cU2 = q.bu; ra.cV01 = pk(q.rv, q.gv); ra.cV2 = q.bv; }
    ra.xcdRemap = c->args.xcdRemap; ra.y2r = c->args.y2r;
    return ra;
}

// packed RGB -> packed RGB on the band walker (scale_yuvg_rgbsrc_kernel): the context's tables + the call's pitches; the scaler's own colour model at both ends
static YuvGArgs make_rg_args(const GmatSwsContext *c, int srcStride, int dstStride)
{
    YuvGArgs g = c->rgargs;
    g.ys = srcStride; g.ds = dstStride;
    g.r2y = c->args.r2y; g.y2r = c->args.y2r; g.xcdRemap = c->args.xcdRemap;
    g.srcPx = c->px4 ? 4 : 3; g.srcAlpha = c->px4 ? c->px4Alpha : 0;
    return g;
}

// an RGBA / BGRA source in front of this (24-bit) context: whether a call of n frames lands on the block-cooperative RGB -> RGB kernel, which reads four-byte
// pixels as they are (and scales their alpha channel as a fourth line) — the 32 -> 24-bit pass and the alpha passes of MODE_VIA_INNER are then not run
static bool rg_block_takes_px4(GmatSwsContext *in, int n, const uint8_t *src, int ss, const uint8_t *dst, int ds)
{
    if (!in || in->mode != MODE_SCALE || in->prof || in->inner || in->rgbViaPlanes) return false;
    if (!(in->srcFormat == GMAT_PIX_FMT_RGB24 || in->srcFormat == GMAT_PIX_FMT_BGR24) || !is_packed_rgb(in->dstFormat)) return false;
    if (const char *k = GMAT_KNOB("GMAT_RGBSRC_NO_PX4")) if (atoi(k)) return false;
    if (const char *rw = GMAT_KNOB("GMAT_RGBSRC_WALKER")) if (!atoi(rw)) return false;
    if (ensure_scaler(in) < 0 || !in->rg.ok) return false;      // (exactly 2 : 1 included: the strip kernel of the 24-bit context reads three-byte pixels)
    return al4(src, ss) && al4(dst, ds) && yuvg_rgbsrc_block_form(in->rgargs, std::min(n, kYuv2xMaxFrames));
}

// the fused convert-then-scale form on the same kernel: a YUV 4:2:0 source converted lane by lane with the FIRST context's closed-form
// constants (c->y2r: base, off*, c*) — the output stage fields (y_coeff .. u2b) stay those of the scaler's BT.601 model
static Rgb2sArgs make_rgb2h_yuv_args(const GmatSwsContext *c, const int srcStride[], int dstStride)
{
    Rgb2sArgs ra = make_rgb2s_args(c, srcStride[0], dstStride, false);
    const bool nv12 = c->srcFormat == GMAT_PIX_FMT_NV12;
    ra.srcKind = nv12 ? 1 : 2; ra.us = srcStride[1]; ra.vs = nv12 ? 0 : srcStride[2];
    Yuv2RgbConsts k = c->y2r;
    k.y_coeff = c->args.y2r.y_coeff; k.y_offset = c->args.y2r.y_offset;
    k.v2r = c->args.y2r.v2r; k.v2g = c->args.y2r.v2g; k.u2g = c->args.y2r.u2g; k.u2b = c->args.y2r.u2b;
    ra.y2r = k;
    return ra;
}

// dword loads of every plane (the lane offsets are multiples of 8 / 4 bytes), the packed-RGB store side as the RGB source form
static bool rgb2h_yuv_eligible(const GmatSwsContext *c, const uint8_t *const src[], const int srcStride[], const uint8_t *dst, int dstStride)
{
    if (!c->r2s.ok || !rgb2h_takes_yuv() || c->fused != 1 || !is_yuv420(c->srcFormat) || !is_packed_rgb(c->dstFormat)) return false;
    const bool nv12 = c->srcFormat == GMAT_PIX_FMT_NV12;
    if (!src[0] || !src[1] || (!nv12 && !src[2]) || !dst) return false;
    if (!al4(src[0], srcStride[0]) || !al4(src[1], srcStride[1]) || (!nv12 && !al4(src[2], srcStride[2]))) return false;
    return bytes_per_pixel(c->dstFormat) == 4 ? ((((uintptr_t)dst | (uintptr_t)dstStride) & 15) == 0) : al4(dst, dstStride);
}

static bool shift8_shortcut(const GmatSwsContext *c);          // (defined with active_plan below)

namespace gmat {
// A context's tables live on the device that was current when it was created; a launch made while another device is current
// would run there with pointers of this one (hwcontext_cuda.c:395-434 makes the stream's device current around every call).
static int check_device(const GmatSwsContext *c, const char *who)
{
    int dev = c->device;
    if (hipGetDevice(&dev) == hipSuccess && dev != c->device) {
        logf(LOG_ERROR, "%s: context created on device %d used while device %d is current", who, c->device, dev);
        return GMAT_ERR(EINVAL);
    }
    return 0;
}

// ---- one set of intermediates per context: ordering between streams ---------------------------------------------------------------------
static bool stream_is_capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}
// contexts whose calls may write a buffer the context owns: the modes with an intermediate, and NV12 <-> YUV420P scaled (the cascade's crossBuf)
static bool owns_intermediates(const GmatSwsContext *c)
{
    return sws_shares_intermediate(c) || lines_context(c) ||
           (c->mode == MODE_SCALE && is_yuv420(c->srcFormat) && is_yuv420(c->dstFormat) && c->srcFormat != c->dstFormat);
}
// lines_context() reads tables the first call builds: they are built before the ownership question is asked
static void lines_ready(GmatSwsContext *c)
{
    if (c && c->mode == MODE_SCALE && !c->yuvReady && is_yuv8_src(c->srcFormat) && (c->fused == 2 || !is_yuv420(c->srcFormat))) (void)ensure_scaler(c);
}
// before a call's first launch: `s` waits for the last use of the intermediates if that was on another stream.  Captured work is ordered by its
// graph (gmat_sws_graph_create keeps such contexts on one branch and synchronises before it captures).
// A context that has only ever seen ONE stream records nothing: an event record is a marker packet in the queue, and the next launch waited
// ≈ 4 µs behind it (tools/trace_gaps.sh: 5.6 µs between a call's last kernel and the next call's first, 0 without the record; one frame a call of the
// lines form 17.0 -> 13.4 µs).  The first time a second stream shows up the device is synchronised ONCE (the old stream's handle may be gone by then:
// nothing is recorded on it), and from then on every call records its last use.
static int stream_handoff_acquire(GmatSwsContext *c, hipStream_t s)
{
    c->interTouched = false;
    if (!c->interUsed || c->interStream == s || stream_is_capturing(s)) return 0;
    if (!c->interMulti) {
        if (hipDeviceSynchronize() != hipSuccess) {           // (refused while another thread captures in global mode: order the two streams by an event instead)
            (void)hipGetLastError();
            if (!c->interEv) GMAT_HIP_CHECK(hipEventCreateWithFlags(&c->interEv, hipEventDisableTiming));
            GMAT_HIP_CHECK(hipEventRecord(c->interEv, c->interStream));
            GMAT_HIP_CHECK(hipStreamWaitEvent(s, c->interEv, 0));
        }
        // the event exists from here on: a call that does not touch the intermediates records nothing (no release), and the next call on yet another
        // stream waits on it — a wait on a never-recorded event is a no-op, a wait on a null handle an error that stuck
        if (!c->interEv) GMAT_HIP_CHECK(hipEventCreateWithFlags(&c->interEv, hipEventDisableTiming));
        c->interMulti = true;
    } else if (c->interEv) {
        GMAT_HIP_CHECK(hipStreamWaitEvent(s, c->interEv, 0));
    }
    c->handoffs++;
    return 0;
}
// after a call's last launch, when it used the intermediates
static int stream_handoff_release(GmatSwsContext *c, hipStream_t s)
{
    if (stream_is_capturing(s)) return 0;
    if (c->interMulti) {
        if (!c->interEv) GMAT_HIP_CHECK(hipEventCreateWithFlags(&c->interEv, hipEventDisableTiming));
        GMAT_HIP_CHECK(hipEventRecord(c->interEv, s));
    }
    c->interStream = s;
    c->interUsed = true;
    return 0;
}

// Frames [0, n) of one geometry (plane pointers 4 per frame, shared strides) through ONE launch of the 2:1 kernel
// per kYuv2xMaxFrames frames.  Returns 1 when taken, 0 when this context / these frames are not eligible (the
// caller then goes frame by frame), < 0 on error.
// NV12 <-> YUV420P scaled: whether this call takes the cascade (scale in the source's layout, re-layout the result) — a context of the plane
// scaler between the two 8-bit 4:2:0 layouts whose own table would end at the tiled catch-all, destination planes the re-layout kernel can move
static bool cross_layout_cascade(GmatSwsContext *c, const YuvScaleArgs &ya, int n)
{
    if (!is_yuv420(c->srcFormat) || !is_yuv420(c->dstFormat) || c->srcFormat == c->dstFormat || c->prof) return false;
    const char *off = GMAT_KNOB("GMAT_NO_CROSS_CASCADE");
    if (off && atoi(off)) return false;
    // (the tile kernel's two records stand in front of the tiled catch-all since round 6: the cascade — a walker in the source's layout, then a copy — keeps the
    // frames it had, the ones whose pick is the catch-all once those two are set aside)
    int pick = 0;
    while (plane_record_is_tile(pick) || !kPlaneKernels[pick].eligible(c, ya, n)) pick++;
    if (pick != kNumPlaneKernels - 1) return false;              // a walker of this context's own takes the frame (2:1)
    const bool dnv = c->dstFormat == GMAT_PIX_FMT_NV12;
    return yuv420_relayout_takes(!dnv, nullptr, 256, nullptr, 256, nullptr, dnv ? 256 : 0, ya.dst, ya.ds, ya.dstU, ya.dsU, dnv ? nullptr : ya.dstV, dnv ? 0 : ya.dsV);
}

static size_t cross_frame_bytes(const GmatSwsContext *c) { return (size_t)align_up(c->dstW + 1, 256) * (c->dstH + 2 * ceil_rshift(c->dstH, 1)); }

// the sibling context and its frames; 0 or a negative error
static int cross_prepare(GmatSwsContext *c, int nframes)
{
    if (!c->cross) {
        c->cross = gmat_sws_getContext(c->srcW, c->srcH, c->srcFormat, c->dstW, c->dstH, c->srcFormat, c->flags, c->param);
        if (!c->cross) return GMAT_ERR(ENOSYS);
        int r = 0;
        if (c->chrPos[0] != -513 || c->chrPos[1] != -513 || c->chrPos[2] != -513 || c->chrPos[3] != -513)
            r = gmat_sws_setChromaPos(c->cross, c->chrPos[0], c->chrPos[1], c->chrPos[2], c->chrPos[3]);
        if (r >= 0 && c->rangeConv) r = gmat_sws_setRange(c->cross, c->rangeConv == 2, c->rangeConv == 1);
        if (r < 0) { gmat_sws_freeContext(c->cross); c->cross = nullptr; return r; }
    }
    if (c->crossFrames < nframes) {
        if (c->crossBuf) { (void)hipFree(c->crossBuf); c->crossBuf = nullptr; c->crossFrames = 0; }
        c->crossPitch = align_up(c->dstW + 1, 256);
        GMAT_HIP_CHECK(hipMalloc((void **)&c->crossBuf, cross_frame_bytes(c) * nframes));
        c->crossFrames = nframes;
    }
    return 0;
}

// the planes of the sibling's output frame inside crossBuf (the SOURCE's layout at the destination's size)
static void cross_planes(const GmatSwsContext *c, uint8_t *base, uint8_t *pl[4], int st[4])
{
    const int ch = ceil_rshift(c->dstH, 1);
    pl[0] = base; pl[1] = base + (size_t)c->crossPitch * c->dstH; pl[2] = pl[3] = nullptr;
    st[0] = st[1] = c->crossPitch; st[2] = st[3] = 0;
    if (c->srcFormat == GMAT_PIX_FMT_YUV420P) { st[1] = st[2] = c->crossPitch / 2; pl[2] = pl[1] + (size_t)st[1] * ch; }
}

// ... or, into a 4:2:0 frame, on the fused block form of the plane scaler's table (scale_yuvg_rgb2p_blk_kernel: the one entry that reads four-byte pixels)
static bool planes_fused_takes_px4(GmatSwsContext *in, int n, const uint8_t *const src[], const int srcStride[], uint8_t *const dst[], const int dstStride[])
{
    if (!in || in->mode != MODE_SCALE || in->prof || in->inner || !in->rgbViaPlanes || in->fused != 2) return false;
    if (const char *k = GMAT_KNOB("GMAT_RGBSRC_NO_PX4")) if (atoi(k)) return false;
    if (ensure_scaler(in) < 0 || in->fused != 2) return false;
    YuvScaleArgs ya;
    if (prep_yuv_args(in, src, srcStride, dst, dstStride, ya) < 0) return false;
    in->px4 = 1;
    int pick = 0;
    while (!kPlaneKernels[pick].eligible(in, ya, n)) pick++;
    const bool ok = !std::strcmp(kPlaneKernels[pick].name(in, ya, n), "scale_yuvg_rgb2p_blk_kernel");
    in->px4 = 0;
    return ok;
}

// ... and the equal-size converter (round 6): rgb2yuv420s_kernel reads four-byte pixels itself where its rule takes the frame
static bool r2y_strip_takes_px4(const GmatSwsContext *in, const uint8_t *const src[], const int srcStride[], uint8_t *const dst[], const int dstStride[])
{
    if (!in || in->mode != MODE_RGB2YUV || in->inner) return false;
    if (GMAT_KNOB("GMAT_RGBSRC_NO_PX4") && atoi(GMAT_KNOB("GMAT_RGBSRC_NO_PX4"))) return false;
    const bool dnv12 = in->dstFormat == GMAT_PIX_FMT_NV12;
    if (!src[0] || !dst[0] || !dst[1] || (!dnv12 && !dst[2])) return false;
    Rgb2YuvLaunch L;
    L.src = src[0]; L.ss = srcStride[0]; L.bgr = 0;
    L.y = dst[0]; L.ys = dstStride[0]; L.u = dst[1]; L.us = dstStride[1]; L.v = dnv12 ? nullptr : dst[2]; L.vs = dnv12 ? 0 : dstStride[2]; L.nv12 = dnv12;
    L.w = in->srcW; L.h = in->srcH; L.maxRows = 0; L.rowStart = nullptr; L.rowCount = nullptr;
    L.stripOk = in->r2y.stripOk;
    return rgb2yuv420_strip_takes(L);
}

// MODE_SCALE16 with a YUV destination on scale19_kernel: frames [0, n) of the context's geometry, up to 32 a launch
static int scale19_frames(GmatSwsContext *c, int n, const uint8_t *const *src_planes, const int srcStride[], uint8_t *const *dst_planes,
                          const int dstStride[], hipStream_t stream)
{
    const bool semiS = c->srcFormat == GMAT_PIX_FMT_NV12 || is_p01x(c->srcFormat), semiD = c->dstFormat == GMAT_PIX_FMT_P016LE;
    const bool s16 = is_p01x(c->srcFormat) || pl16_depth(c->srcFormat) != 0;
    const bool rgb64 = c->s19.rgb64 != 0;
    S19Args a;
    std::memset(&a, 0, sizeof(a));
    a.job[0] = c->s19.job[0]; a.job[1] = c->s19.job[1];
    a.job[0].rawStride[0] = srcStride[0]; a.job[0].rawStride[1] = 0;
    a.job[1].rawStride[0] = srcStride[1]; a.job[1].rawStride[1] = semiS ? 0 : srcStride[2];
    a.job[0].ds[0] = dstStride[0]; a.job[0].ds[1] = 0;
    a.job[1].ds[0] = rgb64 ? 0 : dstStride[1]; a.job[1].ds[1] = rgb64 ? 0 : semiD ? dstStride[1] : dstStride[2];
    a.job[0].rc = rgb64 ? 0 : c->rangeConv; a.job[1].rc = rgb64 ? 0 : c->rangeConv ? c->rangeConv + 2 : 0;      // (swscale.c:536: not for RGB destinations)
    a.rgb64 = c->s19.rgb64; a.chrShift = c->s19.chrShift; a.linesOff = c->s19.linesOff;
    if (rgb64) a.y2r = make_yuv2rgb_consts(c->colorspace, c->srcFullRange != 0);
    uintptr_t sAl = (uintptr_t)srcStride[0] | (uintptr_t)srcStride[1] | (semiS ? 0 : (uintptr_t)srcStride[2]);
    uintptr_t dAl = (uintptr_t)dstStride[0] | (rgb64 ? 0 : (uintptr_t)dstStride[1] | (semiD ? 0 : (uintptr_t)dstStride[2]));
    for (int f0 = 0; f0 < n; f0 += kYuv2xMaxFrames) {
        const int m = std::min(kYuv2xMaxFrames, n - f0);
        Yuv2xFrames fr;
        std::memset(&fr, 0, sizeof(fr));
        uintptr_t sA = sAl, dA = dAl;
        for (int i = 0; i < m; i++) {
            const uint8_t *const *sp = src_planes + 4 * (f0 + i);
            uint8_t *const *dp = dst_planes + 4 * (f0 + i);
            if (!sp[0] || !sp[1] || (!semiS && !sp[2]) || !dp[0] || (!rgb64 && (!dp[1] || (!semiD && !dp[2])))) return GMAT_ERR(EINVAL);
            fr.y[i] = sp[0]; fr.u[i] = sp[1]; fr.v[i] = semiS ? nullptr : sp[2];
            fr.dst[i] = dp[0]; fr.dstU[i] = rgb64 ? nullptr : dp[1]; fr.dstV[i] = (rgb64 || semiD) ? nullptr : dp[2];
            sA |= (uintptr_t)sp[0] | (uintptr_t)sp[1] | (semiS ? 0 : (uintptr_t)sp[2]);
            dA |= (uintptr_t)dp[0] | (rgb64 ? 0 : (uintptr_t)dp[1] | (semiD ? 0 : (uintptr_t)dp[2]));
        }
        if ((dA & 1) || (s16 && (sA & 1))) return GMAT_ERR(EINVAL);              // 16-bit samples sit on even addresses
        a.srcAl4 = (sA & 3) == 0; a.dstAl4 = (dA & 3) == 0;
        a.unit = c->s19.unit; a.srcAl16 = (sA & 15) == 0; a.dstAl16 = (dA & 15) == 0;
        c->lastKernel = c->s19.unit == 1 ? "scale19_unit_kernel" : (c->s19.unit == 2 && a.srcAl16 && a.dstAl16) ? "scale19_unit64_kernel" : "scale19_kernel";
        if (int r = launch_scale19(a, c->s19.np, c->s19.ldsBytes, stream, &fr, m); r < 0) return r;
        c->lastLaunchFrames = m;
    }
    return 0;
}

static int sws_scale_frames_batched_impl(GmatSwsContext *c, int n, const uint8_t *const *src_planes, const int srcStride[],
                                         uint8_t *const *dst_planes, const int dstStride[], hipStream_t stream);
int sws_scale_frames_batched(GmatSwsContext *c, int n, const uint8_t *const *src_planes, const int srcStride[],
                             uint8_t *const *dst_planes, const int dstStride[], hipStream_t stream)
{
    if (!c || n < 2) return 0;
    lines_ready(c);
    const bool own = owns_intermediates(c);
    if (own)
        if (int r = stream_handoff_acquire(c, stream); r < 0) return r;
    const int t = sws_scale_frames_batched_impl(c, n, src_planes, srcStride, dst_planes, dstStride, stream);
    if (own && t > 0 && (c->interTouched || sws_shares_intermediate(c)))
        if (int r = stream_handoff_release(c, stream); r < 0) return r;
    return t;
}
static int sws_scale_frames_batched_impl(GmatSwsContext *c, int n, const uint8_t *const *src_planes, const int srcStride[],
                                         uint8_t *const *dst_planes, const int dstStride[], hipStream_t stream)
{
    if (!c || n < 2) return 0;
    const bool off = GMAT_KNOB("GMAT_SWS_NO_BATCH_KERNEL") != nullptr;
    // the common precheck of every branch below: profiling contexts (per-launch phase stamps) and cascades go frame by frame
    if (off || c->prof) return 0;
    if (int r = check_device(c, "gmat_sws_scale_batch"); r < 0) return r;
    if (c->mode == MODE_VIA_INNER && c->inner) {
        // an RGBA / BGRA source whose 24-bit context lands on the block-cooperative RGB -> RGB kernel: that context's launches, reading the pixels as they are
        for (int f = 0; f < n; f++) {
            if (!src_planes[4 * f] || !dst_planes[4 * f]) return GMAT_ERR(EINVAL);
            if (!rg_block_takes_px4(c->inner, n, src_planes[4 * f], srcStride[0], dst_planes[4 * f], dstStride[0]) &&
                !planes_fused_takes_px4(c->inner, n, src_planes + 4 * f, srcStride, dst_planes + 4 * f, dstStride) &&
                !r2y_strip_takes_px4(c->inner, src_planes + 4 * f, srcStride, dst_planes + 4 * f, dstStride)) return 0;
        }
        c->inner->px4 = 1; c->inner->px4Alpha = c->needAlpha;
        const int t = sws_scale_frames_batched_impl(c->inner, n, src_planes, srcStride, dst_planes, dstStride, stream);
        c->inner->px4 = 0; c->inner->px4Alpha = 0;
        c->lastKernel = c->inner->lastKernel; c->lastLaunchFrames = c->inner->lastLaunchFrames;
        return t == GMAT_ERR(EINVAL) ? 0 : t;       // (a promise the inner context does not keep is "not taken" — the frames go one by one, through the pass that drops the alpha)
    }
    if (c->mode == MODE_YUV2RGB) {
        // the same-size converter: one launch per 32 frames, grid.z = frame
        const bool planar = c->srcFormat == GMAT_PIX_FMT_YUV420P;
        for (int f0 = 0; f0 < n; f0 += kYuv2xMaxFrames) {
            Yuv2xFrames fr;
            const int m = std::min(kYuv2xMaxFrames, n - f0);
            std::memset(&fr, 0, sizeof(fr));
            for (int i = 0; i < m; i++) {
                const uint8_t *const *sp = src_planes + 4 * (f0 + i);
                if (!sp[0] || !sp[1] || (planar && !sp[2]) || !dst_planes[4 * (f0 + i)]) return GMAT_ERR(EINVAL);
                fr.y[i] = sp[0]; fr.u[i] = sp[1]; fr.v[i] = planar ? sp[2] : nullptr;
                fr.dst[i] = dst_planes[4 * (f0 + i)];
            }
            c->lastKernel = "yuv2rgb_kernel";
            int r = launch_yuv2rgb(yuv_src_of(c->srcFormat, src_planes + 4 * f0, srcStride), fr.dst[0], dstStride[0], c->srcW, c->srcH,
                                   c->dstFormat, c->y2r, stream, &fr, m);
            if (r < 0) return r;
            c->lastLaunchFrames = m;
        }
        return 1;
    }
    if (shift8_shortcut(c)) {
        // NV12 -> P010LE / P016LE at equal size: t << 8 of every sample (gmat_sws_scale's rule), frame by frame — the kernel is a copy
        for (int f = 0; f < n; f++) {
            const uint8_t *const *sp = src_planes + 4 * f;
            uint8_t *const *dp = dst_planes + 4 * f;
            if (!sp[0] || !sp[1] || !dp[0] || !dp[1]) return GMAT_ERR(EINVAL);
            c->lastKernel = "nv12_shift8_kernel";
            int r = launch_nv12_shift8(sp[0], srcStride[0], sp[1], srcStride[1], dp[0], dstStride[0], dp[1], dstStride[1], c->srcW, c->srcH, stream);
            if (r < 0) return r;
        }
        c->lastLaunchFrames = 1;
        return 1;
    }
    if (c->mode == MODE_SCALE16 && c->s19.ok && !c->alpha19) {
        const int r = scale19_frames(c, n, src_planes, srcStride, dst_planes, dstStride, stream);
        return r < 0 ? r : 1;
    }
    if (c->mode == MODE_YUV2YUV && (c->srcFormat == GMAT_PIX_FMT_NV12) != (c->dstFormat == GMAT_PIX_FMT_NV12) &&
        !(GMAT_KNOB("GMAT_NO_RELAYOUT_FUSED") && atoi(GMAT_KNOB("GMAT_NO_RELAYOUT_FUSED")))) {
        // NV12 <-> YUV420P: one launch per 32 frames (grid.z = frame) when every frame's planes move 16 / 8 bytes at a time
        const bool snv = c->srcFormat == GMAT_PIX_FMT_NV12;
        for (int f = 0; f < n; f++) {
            const uint8_t *const *sp = src_planes + 4 * f;
            uint8_t *const *dp = dst_planes + 4 * f;
            if (!sp[0] || !sp[1] || (!snv && !sp[2]) || !dp[0] || !dp[1] || (snv && !dp[2])) return GMAT_ERR(EINVAL);
            if (!yuv420_relayout_takes(snv, sp[0], srcStride[0], sp[1], srcStride[1], snv ? nullptr : sp[2], snv ? 0 : srcStride[2],
                                       dp[0], dstStride[0], dp[1], dstStride[1], snv ? dp[2] : nullptr, snv ? dstStride[2] : 0)) return 0;
        }
        for (int f0 = 0; f0 < n; f0 += kYuv2xMaxFrames) {
            Yuv2xFrames fr;
            const int m = std::min(kYuv2xMaxFrames, n - f0);
            std::memset(&fr, 0, sizeof(fr));
            for (int i = 0; i < m; i++) {
                const uint8_t *const *sp = src_planes + 4 * (f0 + i);
                uint8_t *const *dp = dst_planes + 4 * (f0 + i);
                fr.y[i] = sp[0]; fr.u[i] = sp[1]; fr.v[i] = snv ? nullptr : sp[2];
                fr.dst[i] = dp[0]; fr.dstU[i] = dp[1]; fr.dstV[i] = snv ? dp[2] : nullptr;
            }
            c->lastKernel = "yuv420_relayout_kernel";
            int r = launch_yuv420_relayout(snv, fr.y[0], srcStride[0], fr.u[0], srcStride[1], fr.v[0], snv ? 0 : srcStride[2],
                                           fr.dst[0], dstStride[0], fr.dstU[0], dstStride[1], fr.dstV[0], snv ? dstStride[2] : 0, c->srcW, c->srcH, stream, &fr, m);
            if (r < 0) return r;
            c->lastLaunchFrames = m;
        }
        return 1;
    }
    if (c->mode == MODE_RGBPF32) {
        // nv12 -> planar float RGB (the tensor a network reads): one launch per 32 frames, grid.z = frame
        for (int f0 = 0; f0 < n; f0 += kYuv2xMaxFrames) {
            Yuv2xFrames fr;
            const int m = std::min(kYuv2xMaxFrames, n - f0);
            std::memset(&fr, 0, sizeof(fr));
            for (int i = 0; i < m; i++) {
                const uint8_t *const *sp = src_planes + 4 * (f0 + i);
                if (!sp[0] || !sp[1] || !dst_planes[4 * (f0 + i)]) return GMAT_ERR(EINVAL);
                fr.y[i] = sp[0]; fr.u[i] = sp[1]; fr.dst[i] = dst_planes[4 * (f0 + i)];
            }
            c->lastKernel = "nv12_to_rgbpf32_kernel";
            int r = launch_nv12_to_rgbpf32(yuv_src_of(c->srcFormat, src_planes + 4 * f0, srcStride), fr.dst[0], dstStride[0], c->srcW, c->srcH,
                                           c->y2r, stream, &fr, m);
            if (r < 0) return r;
            c->lastLaunchFrames = m;
        }
        return 1;
    }
    if (c->mode == MODE_FROM_PF32 && is_yuv420(c->dstFormat)) {
        // planar float RGB -> 4:2:0 (a network's output on its way to the encoder): one launch of pf32_to_yuv420s_kernel per 32 frames when every
        // frame passes its rule
        const bool dnv12 = c->dstFormat == GMAT_PIX_FMT_NV12;
        Rgb2YuvLaunch L;
        L.ss = srcStride[0]; L.bgr = 0;
        L.ys = dstStride[0]; L.us = dstStride[1]; L.vs = dnv12 ? 0 : dstStride[2]; L.nv12 = dnv12;
        L.w = c->srcW; L.h = c->srcH; L.maxRows = c->r2y.maxRows; L.rowStart = nullptr; L.rowCount = nullptr;
        L.k = make_rgb2yuv_consts(c->colorspace);
        L.stripOk = c->r2y.stripOk; for (int k = 0; k < 4; k++) L.vC[k] = c->r2y.vC[k];
        for (int f = 0; f < n; f++) {
            const uint8_t *const *sp = src_planes + 4 * f;
            uint8_t *const *dp = dst_planes + 4 * f;
            if (!sp[0] || !dp[0] || !dp[1] || (!dnv12 && !dp[2])) return GMAT_ERR(EINVAL);
            L.src = sp[0]; L.y = dp[0]; L.u = dp[1]; L.v = dnv12 ? nullptr : dp[2];
            if (!pf32_to_yuv420_strip_takes(L)) return 0;
        }
        for (int f0 = 0; f0 < n; f0 += kYuv2xMaxFrames) {
            Yuv2xFrames fr;
            const int m = std::min(kYuv2xMaxFrames, n - f0);
            std::memset(&fr, 0, sizeof(fr));
            for (int i = 0; i < m; i++) {
                fr.y[i] = src_planes[4 * (f0 + i)];
                fr.dst[i] = dst_planes[4 * (f0 + i)]; fr.dstU[i] = dst_planes[4 * (f0 + i) + 1]; fr.dstV[i] = dnv12 ? nullptr : dst_planes[4 * (f0 + i) + 2];
            }
            c->lastKernel = "pf32_to_yuv420s_kernel";
            int r = launch_pf32_to_yuv420s(L, stream, &fr, m);
            if (r < 0) return r;
            c->lastLaunchFrames = m;
        }
        return 1;
    }
    if (c->mode == MODE_RGB2YUV && !c->inner) {
        // the same-size RGB -> 4:2:0 converter: one launch of rgb2yuv420s_kernel per 32 frames when every frame passes its rule
        const bool dnv12 = c->dstFormat == GMAT_PIX_FMT_NV12;
        Rgb2YuvLaunch L;
        L.ss = srcStride[0]; L.bgr = c->srcFormat == GMAT_PIX_FMT_BGR24;
        L.ys = dstStride[0]; L.us = dstStride[1]; L.vs = dnv12 ? 0 : dstStride[2]; L.nv12 = dnv12;
        L.w = c->srcW; L.h = c->srcH; L.maxRows = c->r2y.maxRows; L.rowStart = nullptr; L.rowCount = nullptr;
        L.k = make_rgb2yuv_consts(c->colorspace); L.toJpeg = c->rangeConv == 1;
        L.stripOk = c->r2y.stripOk; for (int k = 0; k < 4; k++) L.vC[k] = c->r2y.vC[k];
        L.px = c->px4 ? 4 : 3;
        for (int f = 0; f < n; f++) {
            const uint8_t *const *sp = src_planes + 4 * f;
            uint8_t *const *dp = dst_planes + 4 * f;
            if (!sp[0] || !dp[0] || !dp[1] || (!dnv12 && !dp[2])) return GMAT_ERR(EINVAL);
            L.src = sp[0]; L.y = dp[0]; L.u = dp[1]; L.v = dnv12 ? nullptr : dp[2];
            if (!rgb2yuv420_strip_takes(L)) return c->px4 ? GMAT_ERR(EINVAL) : 0;       // (four-byte pixels were promised the kernel that reads them)
        }
        c->lastKernel = "rgb2yuv420s_kernel";
        for (int f0 = 0; f0 < n; f0 += kYuv2xMaxFrames) {
            Yuv2xFrames fr;
            const int m = std::min(kYuv2xMaxFrames, n - f0);
            std::memset(&fr, 0, sizeof(fr));
            for (int i = 0; i < m; i++) {
                fr.y[i] = src_planes[4 * (f0 + i)];
                fr.dst[i] = dst_planes[4 * (f0 + i)]; fr.dstU[i] = dst_planes[4 * (f0 + i) + 1]; fr.dstV[i] = dnv12 ? nullptr : dst_planes[4 * (f0 + i) + 2];
            }
            int r = launch_rgb2yuv420s(L, stream, &fr, m);
            if (r < 0) return r;
            c->lastLaunchFrames = m;
        }
        return 1;
    }
    if (c->mode == MODE_SCALE && (c->srcFormat == GMAT_PIX_FMT_RGB24 || c->srcFormat == GMAT_PIX_FMT_BGR24) && !c->rgbViaPlanes &&
        !c->inner && is_packed_rgb(c->dstFormat)) {
        // packed RGB at exactly 2:1: the strip-walking scaler, one launch per 32 frames; at the band walker's other ratios: its RGB-source form
        if (ensure_scaler(c) < 0 || !(c->r2s.ok || c->rg.ok)) return 0;
        if (!c->r2s.ok || (c->px4 && c->rg.ok)) {                 // (four-byte pixels at exactly 2 : 1 too: the strip kernel reads three)
            // (the walker's form, 32 frames a launch: rgb24 1080p -> 720p 14.2 -> 6.6 us a frame, 4K -> 900p 30.1 -> 22.0, 720p -> 1080p 14.8 -> 9.3 — from four frames a
            // launch on, and only where the block-cooperative form, scale_yuvg_rgbsrc_blk_kernel, has no instance: that one is the rule at every launch size
            // (4.2 / 12.6 / 5.5 us a frame batched, 8.2 / 19.4 / 10.4 alone; k_scale_yuvg.hip yuvg_rgbsrc_block_form))
            const char *rw = GMAT_KNOB("GMAT_RGBSRC_WALKER");
            const int mode = rw ? atoi(rw) : 1;
            const bool blk = yuvg_rgbsrc_block_form(c->rgargs, std::min(n, kYuv2xMaxFrames));
            if (mode == 0 || !(blk || (c->rgargs.K && (mode == 2 || n >= 4)))) return 0;
            for (int f = 0; f < n; f++) {
                const uint8_t *sp = src_planes[4 * f];
                uint8_t *dp = dst_planes[4 * f];
                if (!sp || !dp) return GMAT_ERR(EINVAL);
                if (!al4(sp, srcStride[0]) || !al4(dp, dstStride[0])) return 0;
            }
            const YuvGArgs ga = make_rg_args(c, srcStride[0], dstStride[0]);
            c->lastKernel = blk ? "scale_yuvg_rgbsrc_blk_kernel" : "scale_yuvg_rgbsrc_kernel";
            for (int f0 = 0; f0 < n; f0 += kYuv2xMaxFrames) {
                Yuv2xFrames fr;
                const int m = std::min(kYuv2xMaxFrames, n - f0);
                std::memset(&fr, 0, sizeof(fr));
                for (int i = 0; i < m; i++) { fr.y[i] = src_planes[4 * (f0 + i)]; fr.dst[i] = dst_planes[4 * (f0 + i)]; }
                int r = launch_scale_yuvg_rgbsrc(ga, stream, &fr, m);
                if (r < 0) return r;
                c->lastKernel = yuvg_rgbsrc_block_form(ga, m) ? "scale_yuvg_rgbsrc_blk_kernel" : "scale_yuvg_rgbsrc_kernel";   // (the form of THIS chunk — the launcher decides per launch)
                c->lastLaunchFrames = m;
            }
            return 1;
        }
        const int bpp = bytes_per_pixel(c->dstFormat);
        for (int f = 0; f < n; f++) {
            const uint8_t *sp = src_planes[4 * f];
            uint8_t *dp = dst_planes[4 * f];
            if (!sp || !dp) return GMAT_ERR(EINVAL);
            if (!al4(sp, srcStride[0])) return 0;
            if (bpp == 4 ? ((((uintptr_t)dp | (uintptr_t)dstStride[0]) & 15) != 0) : !al4(dp, dstStride[0])) return 0;
        }
        const Rgb2sArgs ra = make_rgb2s_args(c, srcStride[0], dstStride[0], c->srcFormat == GMAT_PIX_FMT_BGR24);
        c->lastKernel = rgb2s_kernel_name();
        for (int f0 = 0; f0 < n; f0 += kYuv2xMaxFrames) {
            Yuv2xFrames fr;
            const int m = std::min(kYuv2xMaxFrames, n - f0);
            std::memset(&fr, 0, sizeof(fr));
            for (int i = 0; i < m; i++) { fr.y[i] = src_planes[4 * (f0 + i)]; fr.dst[i] = dst_planes[4 * (f0 + i)]; }
            int r = launch_scale_rgb2s(ra, stream, &fr, m);
            if (r < 0) return r;
            c->lastLaunchFrames = m;
        }
        return 1;
    }
    if (c->mode == MODE_SCALE && is_yuv420(c->srcFormat) && c->fused == 1 && is_packed_rgb(c->dstFormat)) {
        // the fused convert-then-scale form at exactly 2:1: one launch of scale_rgb2h_kernel<.., yuv> per 32 frames
        if (ensure_scaler(c) < 0 || c->fused != 1) return 0;
        for (int f = 0; f < n; f++)
            if (!rgb2h_yuv_eligible(c, src_planes + 4 * f, srcStride, dst_planes[4 * f], dstStride[0])) return 0;
        const Rgb2sArgs ra = make_rgb2h_yuv_args(c, srcStride, dstStride[0]);
        const bool planar = c->srcFormat == GMAT_PIX_FMT_YUV420P;
        c->lastKernel = "scale_rgb2h_kernel<yuv>";
        for (int f0 = 0; f0 < n; f0 += kYuv2xMaxFrames) {
            Yuv2xFrames fr;
            const int m = std::min(kYuv2xMaxFrames, n - f0);
            std::memset(&fr, 0, sizeof(fr));
            for (int i = 0; i < m; i++) {
                const uint8_t *const *sp = src_planes + 4 * (f0 + i);
                fr.y[i] = sp[0]; fr.u[i] = sp[1]; fr.v[i] = planar ? sp[2] : nullptr; fr.dst[i] = dst_planes[4 * (f0 + i)];
            }
            int r = launch_scale_rgb2s(ra, stream, &fr, m);
            if (r < 0) return r;
            c->lastLaunchFrames = m;
        }
        return 1;
    }
    if (c->mode == MODE_SCALE && is_yuv420(c->srcFormat) && c->fused == 0 && is_packed_rgb(c->dstFormat)) {
        // the two-kernel form (convert at source size, then scale: the reference's structure, swscale_cuda.c:352-371) for
        // n frames: ONE launch of the converter into n context-owned RGB24 intermediates, ONE launch of the strip scaler
        if (ensure_scaler(c) < 0 || !c->r2s.ok) return 0;
        const bool planar = c->srcFormat == GMAT_PIX_FMT_YUV420P;
        const int bpp = bytes_per_pixel(c->dstFormat);
        for (int f = 0; f < n; f++) {
            const uint8_t *const *sp = src_planes + 4 * f;
            uint8_t *dp = dst_planes[4 * f];
            if (!sp[0] || !sp[1] || (planar && !sp[2]) || !dp) return GMAT_ERR(EINVAL);
            if (bpp == 4 ? ((((uintptr_t)dp | (uintptr_t)dstStride[0]) & 15) != 0) : !al4(dp, dstStride[0])) return 0;
        }
        const int per = std::min(n, kYuv2xMaxFrames);
        const int interBatchStride = align_up(c->srcW * 3, 256);     // the batch's own pitch: c->interStride belongs to the single-frame `inter`
        const size_t frameBytes = (size_t)interBatchStride * c->srcH;
        if (c->interBatchFrames < per) {
            if (c->interBatch) (void)hipFree(c->interBatch);
            c->interBatch = nullptr; c->interBatchFrames = 0;
            GMAT_HIP_CHECK(hipMalloc((void **)&c->interBatch, frameBytes * per));
            c->interBatchFrames = per;
        }
        const Rgb2sArgs ra = make_rgb2s_args(c, interBatchStride, dstStride[0], false);
        for (int f0 = 0; f0 < n; f0 += kYuv2xMaxFrames) {
            const int m = std::min(kYuv2xMaxFrames, n - f0);
            Yuv2xFrames cv, sc;
            std::memset(&cv, 0, sizeof(cv)); std::memset(&sc, 0, sizeof(sc));
            for (int i = 0; i < m; i++) {
                const uint8_t *const *sp = src_planes + 4 * (f0 + i);
                cv.y[i] = sp[0]; cv.u[i] = sp[1]; cv.v[i] = planar ? sp[2] : nullptr;
                cv.dst[i] = c->interBatch + frameBytes * i;
                sc.y[i] = cv.dst[i]; sc.dst[i] = dst_planes[4 * (f0 + i)];
            }
            int r = launch_yuv2rgb(yuv_src_of(c->srcFormat, src_planes + 4 * f0, srcStride), cv.dst[0], interBatchStride, c->srcW, c->srcH,
                                   GMAT_PIX_FMT_RGB24, c->y2r, stream, &cv, m);
            if (r < 0) return r;
            if ((r = launch_scale_rgb2s(ra, stream, &sc, m)) < 0) return r;
            c->lastLaunchFrames = m;
        }
        c->lastKernel = rgb2s_kernel_name();
        return 1;
    }
    if (c->mode != MODE_SCALE || !(is_plane_src(c->srcFormat) || c->rgbViaPlanes) || c->fused != 2) return 0;
    if (ensure_scaler(c) < 0 || c->fused != 2) return 0;
    // every frame must fall in the same alignment class (the flags select vector or byte paths for the whole launch), and the
    // kernel is the first of the table that takes EVERY frame
    YuvScaleArgs ya0;
    bool can[kNumPlaneKernels];
    for (int k = 0; k < kNumPlaneKernels; k++) can[k] = true;
    for (int f = 0; f < n; f++) {
        YuvScaleArgs ya;
        if (!src_planes[4 * f] || !dst_planes[4 * f]) return GMAT_ERR(EINVAL);
        int r = prep_yuv_args(c, src_planes + 4 * f, srcStride, dst_planes + 4 * f, dstStride, ya);
        if (r < 0) return r;
        for (int k = 0; k < kNumPlaneKernels; k++) can[k] = can[k] && kPlaneKernels[k].eligible(c, ya, n);
        if (f == 0) ya0 = ya;
        else if (ya.dstAligned != ya0.dstAligned || ya.srcAligned != ya0.srcAligned || ya.srcAligned16 != ya0.srcAligned16) return 0;
    }
    int pick = 0;
    while (!can[pick]) pick++;                         // (the last record takes everything)
    int pickNoTile = 0;
    while (plane_record_is_tile(pickNoTile) || !can[pickNoTile]) pickNoTile++;
    if (pickNoTile == kNumPlaneKernels - 1 && cross_layout_cascade(c, ya0, n)) {
        // NV12 <-> YUV420P scaled (gmat_sws_scale's rule), every frame's destination movable by the re-layout kernel: the sibling context's own
        // batch into crossBuf, then one re-layout launch per 32 frames
        bool all = true;
        for (int f = 1; f < n && all; f++) {
            YuvScaleArgs ya;
            if (prep_yuv_args(c, src_planes + 4 * f, srcStride, dst_planes + 4 * f, dstStride, ya) < 0) return GMAT_ERR(EINVAL);
            all = cross_layout_cascade(c, ya, n);
        }
        if (all) {
            const bool snv = c->srcFormat == GMAT_PIX_FMT_NV12;
            for (int f0 = 0; f0 < n; f0 += kYuv2xMaxFrames) {
                const int m = std::min(kYuv2xMaxFrames, n - f0);
                if (int r = cross_prepare(c, m); r < 0) return r;
                c->interTouched = true;
                std::vector<uint8_t *> ip((size_t)4 * m);
                int st[4] = {0, 0, 0, 0};
                Yuv2xFrames fr;
                std::memset(&fr, 0, sizeof(fr));
                for (int i = 0; i < m; i++) {
                    uint8_t *pl[4];
                    cross_planes(c, c->crossBuf + cross_frame_bytes(c) * i, pl, st);
                    for (int q = 0; q < 4; q++) ip[(size_t)4 * i + q] = pl[q];
                    uint8_t *const *dp = dst_planes + 4 * (f0 + i);
                    fr.y[i] = pl[0]; fr.u[i] = pl[1]; fr.v[i] = pl[2];
                    fr.dst[i] = dp[0]; fr.dstU[i] = dp[1]; fr.dstV[i] = snv ? dp[2] : nullptr;
                }
                gmat_sws_setStream(c->cross, (void *)stream);
                int r = m > 1 ? sws_scale_frames_batched(c->cross, m, src_planes + 4 * f0, srcStride, ip.data(), st, stream) : 0;
                if (r < 0) return r;
                if (r == 0)
                    for (int i = 0; i < m; i++)
                        if ((r = gmat_sws_scale(c->cross, src_planes + 4 * (f0 + i), srcStride, 0, c->srcH, ip.data() + 4 * i, st)) < 0) return r;
                r = launch_yuv420_relayout(snv, fr.y[0], st[0], fr.u[0], st[1], fr.v[0], st[2], fr.dst[0], dstStride[0], fr.dstU[0], dstStride[1],
                                           fr.dstV[0], snv ? dstStride[2] : 0, c->dstW, c->dstH, stream, &fr, m);
                if (r < 0) return r;
                c->lastKernel = c->cross->lastKernel;
                c->lastLaunchFrames = c->cross->lastLaunchFrames;
            }
            return 1;
        }
    }
    const PlaneKernel &K = kPlaneKernels[pick];
    const bool planarSrc = c->srcFormat == GMAT_PIX_FMT_YUV420P || c->srcFormat == GMAT_PIX_FMT_YUV444P || pl16_depth(c->srcFormat);
    const bool yuvDst = is_yuv8_src(c->dstFormat) || is_dst10(c->dstFormat);
    const bool planarDst = c->dstFormat == GMAT_PIX_FMT_YUV420P || c->dstFormat == GMAT_PIX_FMT_YUV444P || c->dstFormat == GMAT_PIX_FMT_YUV420P10LE;
    for (int f0 = 0; f0 < n; f0 += kYuv2xMaxFrames) {
        Yuv2xFrames fr;
        const int m = std::min(kYuv2xMaxFrames, n - f0);
        std::memset(&fr, 0, sizeof(fr));
        for (int i = 0; i < m; i++) {
            const uint8_t *const *sp = src_planes + 4 * (f0 + i);
            uint8_t *const *dp = dst_planes + 4 * (f0 + i);
            fr.y[i] = sp[0]; fr.u[i] = c->rgbViaPlanes ? nullptr : sp[1]; fr.v[i] = planarSrc ? sp[2] : nullptr;
            fr.dst[i] = dp[0]; fr.dstU[i] = yuvDst ? dp[1] : nullptr; fr.dstV[i] = planarDst ? dp[2] : nullptr;
        }
        c->lastKernel = K.name(c, ya0, m);
        if (c->px4 && std::strcmp(c->lastKernel, "scale_yuvg_rgb2p_blk_kernel")) return GMAT_ERR(EINVAL);
        int r = K.launch(c, ya0, stream, fr, m);
        if (r < 0) return r;
        c->lastLaunchFrames = m;
    }
    return 1;
}
} // namespace gmat

namespace gmat {
int sws_src_height(const GmatSwsContext *c) { return c ? c->srcH : 0; }
hipEvent_t *sws_batch_events(GmatSwsContext *c)
{
    if (!c) return nullptr;
    if (!c->batchEvReady) {
        for (hipEvent_t &e : c->batchEv)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        c->batchEvReady = true;
    }
    return c->batchEv;
}
void *sws_current_stream(const GmatSwsContext *c) { return c ? (void *)c->stream : nullptr; }
bool sws_owns_intermediates(const GmatSwsContext *c) { lines_ready(const_cast<GmatSwsContext *>(c)); return c && owns_intermediates(c); }
bool sws_shares_intermediate(const GmatSwsContext *c)
{
    if (!c) return false;
    if (c->mode == MODE_SCALE16 && c->s19.ok && !is_rgb64(c->dstFormat)) return false;                           // (round 6: the lines of a tile live in LDS; a 64-bit destination may still take the two passes — an RGBA source's alpha lines)
    if (c->mode == MODE_VIA_INNER || c->mode == MODE_SCALE16 || c->mode == MODE_VIA_PLANES16) return true;       // one set of intermediates per context
    if (c->mode == MODE_FROM_PF32 && is_yuv420(c->dstFormat)) return true;
    return c->mode == MODE_SCALE && is_yuv420(c->srcFormat) && c->fused == 0;
}
}


// ---- alpha (needAlpha) ----------------------------------------------------------------------------------------------
// the plan whose filters make the colours of this context's destination
static const ScalePlan &active_plan(const GmatSwsContext *c)
{
    if (c->mode == MODE_SCALE16) return c->plan16;
    return ((is_plane_src(c->srcFormat) || c->rgbViaPlanes) && c->fused == 2) ? c->planYuv : c->plan;
}

// every output of the bank is its own source sample: one non-zero coefficient (= the bank's unit, the rows are normalised) at source index i
static bool bank_is_identity(const FilterBank &fb)
{
    if (fb.count <= 0 || fb.taps <= 0) return false;
    for (int i = 0; i < fb.count; i++) {
        const int k = i - fb.pos[i];
        if (k < 0 || k >= fb.taps) return false;
        for (int t = 0; t < fb.taps; t++)
            if ((fb.coef[(size_t)i * fb.taps + t] != 0) != (t == k)) return false;
    }
    return true;
}

// NV12 -> P010LE / P016LE at equal size is t << 8 of every sample only while all four filter banks are one-tap identities: chroma positions that differ
// between the ends (gmat_sws_setChromaPos) make the chroma banks real filters at equal size too
static bool shift8_shortcut(const GmatSwsContext *c)
{
    if (!((c->mode == MODE_SCALE || c->mode == MODE_SCALE16) && c->srcFormat == GMAT_PIX_FMT_NV12 && is_p01x(c->dstFormat) &&
          c->srcW == c->dstW && c->srcH == c->dstH && !c->rangeConv)) return false;
    if (GMAT_KNOB("GMAT_NO_SHIFT8") && atoi(GMAT_KNOB("GMAT_NO_SHIFT8"))) return false;
    // (the four banks are walked once a plan, not once a call — the plan changes where it is built: init, setChromaPos, setRange)
    if (c->shift8Ident < 0) {
        const ScalePlan &p = active_plan(c);
        c->shift8Ident = bank_is_identity(p.hLum) && bank_is_identity(p.vLum) && bank_is_identity(p.hChr) && bank_is_identity(p.vChr);
    }
    return c->shift8Ident != 0;
}

// packed_vscale's choice of writer per output row (vscale.c:135-167) as the alpha kernel's form word (k_rgb64.hip)
static void alpha_forms(const ScalePlan &p, std::vector<int32_t> &form, std::vector<int32_t> &first)
{
    const int lfs = p.vLum.taps, cfs = p.vChr.taps;
    const bool full = (p.flags & GMAT_SWS_FULL_CHR_H_INT) != 0;
    form.assign(p.dstH, 0); first.assign(p.dstH, 0);
    for (int y = 0; y < p.dstH; y++) {
        const int16_t *lf = &p.vLum.coef[(size_t)y * lfs], *cf = &p.vChr.coef[(size_t)(y >> p.chrDstVSub) * cfs];
        const bool chr2 = cfs == 2 && cf[0] + cf[1] == 4096 && (unsigned)cf[1] <= 4096u;
        const bool lum2 = lfs == 2 && lf[0] + lf[1] == 4096 && (unsigned)lf[1] <= 4096u;
        int m, ya = 0;
        if (lfs == 1 && (cfs == 1 || chr2)) m = full ? 6 : ((cfs == 1 ? 0 : cf[1]) < 2048 ? 1 : 2);
        else if (lum2 && chr2) { m = full ? 5 : 3; ya = lf[1]; }
        else m = full ? 4 : 0;
        form[y] = m | (ya << 8);
        first[y] = p.vLum.pos[y];
    }
}

// owner: the context the caller holds; maker: the context whose plan scales the colour channels
static int alpha_prepare(GmatSwsContext *owner, const GmatSwsContext *maker)
{
    const ScalePlan &p = active_plan(maker);
    const std::vector<int32_t> none(std::max(std::max(p.dstW, p.dstH), 1), 0);
    int r;
    if ((r = owner->aH.upload(p.hLum, none, owner->daH)) < 0) return r;
    if ((r = owner->alphaLines.reserve((size_t)p.srcH * p.dstW * 4)) < 0) return r;
    if (is_rgb64(owner->dstFormat)) return 0;             // the 64-bit writer takes the lines as an operand (vrgba64_kernel)
    if ((r = owner->aV.upload(p.vLum, none, owner->daV)) < 0) return r;
    std::vector<int32_t> form, first;
    alpha_forms(p, form, first);
    if ((r = owner->aForm.upload(form.data(), form.size() * 4)) < 0) return r;
    return owner->aFirst.upload(first.data(), first.size() * 4);
}

// the alpha byte of every pixel of an RGBA / BGRA destination, after the colour channels were written
static int alpha_run8(GmatSwsContext *c, const uint8_t *alpha0, int srcStride, int kind, int step, uint8_t *dst, int dstStride)
{
    int32_t *la = (int32_t *)c->alphaLines.p;
    int r = launch_hscale19(alpha0, srcStride, kind, step, c->srcW, c->srcH, c->daH, la, c->dstW, c->stream, 1);
    if (r < 0) return r;
    return launch_alpha8_out(la, c->dstW, c->srcH, c->daV, (const int32_t *)c->aForm.p, (const int32_t *)c->aFirst.p, dst, dstStride,
                             c->dstW, c->dstH, c->stream);
}

static thread_local bool g_privFormatOk = false;          // gmat_sws_getContext accepts the library-internal source format

extern "C" {

GmatSwsContext *gmat_sws_getContext(int srcW, int srcH, int srcFormat, int dstW, int dstH, int dstFormat,
                                    int flags, const double *param)
{
    gmat::knobs_refresh();                       // the environment knobs are read now, not per launch (common.h)
    if (srcW < 1 || srcH < 1 || dstW < 1 || dstH < 1) {
        logf(LOG_ERROR, "gmat_sws_getContext: %dx%d -> %dx%d is an invalid scaling dimension", srcW, srcH, dstW, dstH);
        return nullptr;
    }
    if (is_priv_planes(srcFormat) && !g_privFormatOk) return nullptr;
    GmatSwsContext *c = new (std::nothrow) GmatSwsContext();
    if (!c) return nullptr;
    if (hipGetDevice(&c->device) != hipSuccess) c->device = 0;
    // the 4th byte of RGB0 / BGR0 is padding: libswscale runs them as RGBA / BGRA (handle_0alpha, utils.c:1121-1144);
    // nothing here reads alpha, and every path that creates one writes 255
    auto alpha_twin = [](int f, bool &was0) {
        was0 = f == GMAT_PIX_FMT_RGB0 || f == GMAT_PIX_FMT_BGR0;
        return f == GMAT_PIX_FMT_RGB0 ? (int)GMAT_PIX_FMT_RGBA : f == GMAT_PIX_FMT_BGR0 ? (int)GMAT_PIX_FMT_BGRA : f;
    };
    srcFormat = alpha_twin(srcFormat, c->src0);
    dstFormat = alpha_twin(dstFormat, c->dst0);
    c->srcW = srcW; c->srcH = srcH; c->srcFormat = srcFormat;
    c->dstW = dstW; c->dstH = dstH; c->dstFormat = dstFormat;
    c->flags = flags & ~GMAT_SWS_HWACCEL;
    c->param[0] = param ? param[0] : GMAT_SWS_PARAM_DEFAULT;
    c->param[1] = param ? param[1] : GMAT_SWS_PARAM_DEFAULT;
    c->y2r = make_yuv2rgb_consts(GMAT_SWS_CS_DEFAULT, false);
    if (const char *e = GMAT_KNOB("GMAT_SWS_FUSED")) c->fused = atoi(e);

    const bool same = srcW == dstW && srcH == dstH;
    int r = 0;
    const bool src32 = srcFormat == GMAT_PIX_FMT_RGBA || srcFormat == GMAT_PIX_FMT_BGRA;
    if (is_packed_rgb(srcFormat) && is_dst16(dstFormat)) {
        // 8-bit packed RGB into a 16-bit destination: the 19-bit lines of a 16-bit-lined source (an RGB source's lines are 16 bits wide
        // whatever its depth, utils.c:1561-1570) — rgb24ToY_c / ToUV(_half)_c into planes, then the planar context of the 19-bit path
        g_privFormatOk = true;
        c->inner = gmat_sws_getContext(srcW, srcH, GMAT_PIX_FMT_PRIV_RGB8_PLANES, dstW, dstH, dstFormat, flags, param);
        g_privFormatOk = false;
        if (!c->inner) { delete c; return nullptr; }
        c->mode = MODE_VIA_PLANES16;
        const ScalePlan &p = active_plan(c->inner);
        c->p16Stride[0] = align_up(2 * srcW, 256); c->p16Stride[1] = align_up(2 * p.chrSrcW, 256);
        c->p16Off[1] = (size_t)c->p16Stride[0] * srcH; c->p16Off[2] = c->p16Off[1] + (size_t)c->p16Stride[1] * srcH;
        if (c->planes16.reserve(c->p16Off[2] + (size_t)c->p16Stride[1] * srcH) < 0) { delete c; return nullptr; }
        c->needAlpha = src32 && !c->src0 && is_rgb64(dstFormat);
        if (c->needAlpha && alpha_prepare(c, c->inner) < 0) { delete c; return nullptr; }
        return c;
    }
    if (src32 && !(same && is_packed_rgb(dstFormat))) {
        // 32-bit RGB sources (swscale_cuda.c:34-44 lists RGBA / BGRA): rgb32ToY / ToUV (input.c rgb16_32 templates) read
        // the same three channels with the same coefficients as the 24-bit readers and ignore alpha, so the context is the
        // 24-bit one behind a byte re-pack into an intermediate frame
        c->inner = gmat_sws_getContext(srcW, srcH, srcFormat == GMAT_PIX_FMT_RGBA ? GMAT_PIX_FMT_RGB24 : GMAT_PIX_FMT_BGR24, dstW, dstH,
                                       dstFormat, flags, param);
        if (!c->inner) { delete c; return nullptr; }
        c->mode = MODE_VIA_INNER;
        // ... the colour channels; with an alpha channel at both ends libswscale scales the alpha plane too (needAlpha,
        // utils.c:1902; rgbaToA_c, the luma filters, the packed writer's alpha)
        c->needAlpha = has_alpha(dstFormat) && !c->src0 && !c->dst0;
        if (c->needAlpha && alpha_prepare(c, c->inner) < 0) { delete c; return nullptr; }
        return c;
    }
    if (is_rgb64(srcFormat) && !(same && srcFormat == dstFormat)) {
        // RGBA64LE / BGRA64LE sources (swscale_cuda.c:34-44): no special converter takes them at equal size either (findRgbConvFn,
        // swscale_unscaled.c:1458-1528, pairs them with the 48-bit formats only) — always the generic path on the 16-bit lines
        // rgb64ToY_c / ToUV_c / ToUV_half_c make (input.c:36-121)
        g_privFormatOk = true;
        c->inner = gmat_sws_getContext(srcW, srcH, GMAT_PIX_FMT_PRIV_RGB64_PLANES, dstW, dstH, dstFormat, flags, param);
        g_privFormatOk = false;
        if (!c->inner) { delete c; return nullptr; }
        c->mode = MODE_VIA_PLANES16;
        const ScalePlan &p = active_plan(c->inner);
        c->p16Stride[0] = align_up(2 * srcW, 256); c->p16Stride[1] = align_up(2 * p.chrSrcW, 256);
        c->p16Off[1] = (size_t)c->p16Stride[0] * srcH; c->p16Off[2] = c->p16Off[1] + (size_t)c->p16Stride[1] * srcH;
        if (c->planes16.reserve(c->p16Off[2] + (size_t)c->p16Stride[1] * srcH) < 0) { delete c; return nullptr; }
        c->needAlpha = has_alpha(dstFormat) && !c->dst0;
        if (c->needAlpha && alpha_prepare(c, c->inner) < 0) { delete c; return nullptr; }
        return c;
    }
    if (same && is_rgb64(srcFormat)) {
        c->mode = MODE_COPY;                 // equal format and size: the frame as it is
        c->unscaledMode = c->mode;
        return c;
    }
    if (same && is_yuv420(srcFormat) && is_packed_rgb(dstFormat) && (c->flags & GMAT_SWS_ACCURATE_RND)) {
        // libswscale takes its nearest-chroma special converter only for planar sources without
        // SWS_ACCURATE_RND (swscale_unscaled.c:2094-2100); with the flag (and always for NV12) the CPU runs the
        // generic path, which interpolates chroma vertically.  Opting in with the flag gives exactly that.
        c->mode = MODE_SCALE;
        c->fused = 2;
        r = ensure_scaler(c);
        if (r == 0 && c->fused != 2) r = GMAT_ERR(ENOSYS);
    } else if (same && is_yuv420(srcFormat) && is_packed_rgb(dstFormat)) {
        c->mode = MODE_YUV2RGB;
    } else if (same && srcFormat == GMAT_PIX_FMT_NV12 && dstFormat == GMAT_PIX_FMT_RGBPF32LE) {
        c->mode = MODE_RGBPF32;
    } else if (same && srcFormat == GMAT_PIX_FMT_RGBPF32LE &&
               (dstFormat == GMAT_PIX_FMT_RGB24 || dstFormat == GMAT_PIX_FMT_BGR24 || is_yuv420(dstFormat))) {
        // format_cuda's other direction (vf_format_cuda.c:184-217, rgbpf32_to_nv12): quantise to 8 bits, then — for
        // 4:2:0 outputs — the RGB24 -> YUV path of this library
        c->mode = MODE_FROM_PF32;
        if (is_yuv420(dstFormat)) {
            c->srcFormat = GMAT_PIX_FMT_RGB24;
            r = init_rgb2yuv(c);
            c->srcFormat = GMAT_PIX_FMT_RGBPF32LE;
        }
    } else if (same && ((srcFormat == GMAT_PIX_FMT_RGB24 && dstFormat == GMAT_PIX_FMT_BGR24) ||
                        (srcFormat == GMAT_PIX_FMT_BGR24 && dstFormat == GMAT_PIX_FMT_RGB24))) {
        c->mode = MODE_SWAP_RB;
    } else if (same && srcFormat == dstFormat && is_packed_rgb(srcFormat)) {
        c->mode = (c->src0 && !c->dst0) ? MODE_REPACK : MODE_COPY;      // padding -> alpha: the byte is set to 255 (swscale.c:959-978)
    } else if (same && is_packed_rgb(srcFormat) && is_packed_rgb(dstFormat)) {
        // the remaining packed pairs have a 32-bit end: rgbToRgbWrapper's byte moves (swscale_unscaled.c:1579-1640).
        // With SWS_BITEXACT libswscale does not use its 24 -> 32 converters (:1571-1574) and the context runs the
        // generic scaler; same here.
        if (bytes_per_pixel(srcFormat) == 3 && (c->flags & GMAT_SWS_BITEXACT)) {
            c->mode = MODE_SCALE;
            r = ensure_scaler(c);
        } else {
            c->mode = MODE_REPACK;
        }
    } else if (same && (srcFormat == GMAT_PIX_FMT_RGB24 || srcFormat == GMAT_PIX_FMT_BGR24) && is_yuv420(dstFormat)) {
        c->mode = MODE_RGB2YUV;
        r = init_rgb2yuv(c);
    } else if (same && (srcFormat == GMAT_PIX_FMT_RGB24 || srcFormat == GMAT_PIX_FMT_BGR24) && dstFormat == GMAT_PIX_FMT_YUV444P &&
               !(c->flags & GMAT_SWS_FAST_BILINEAR)) {
        // every filter has one tap: a per-pixel conversion — unless SWS_FAST_BILINEAR halves the source's chroma (utils.c:1529-1545:
        // rgb24ToUV_half_c, then a 1:2 chroma filter), which is the plane scaler's job below (a fuzz find of round 3)
        c->mode = MODE_RGB2YUV444;
    } else if (same && is_yuv420(srcFormat) && is_yuv420(dstFormat)) {
        c->mode = MODE_YUV2YUV;
    } else if (same && srcFormat == GMAT_PIX_FMT_YUV420P && (dstFormat == GMAT_PIX_FMT_P010LE || dstFormat == GMAT_PIX_FMT_P016LE)) {
        // planar8ToP01xleWrapper (swscale_unscaled.c:286-324), which libswscale selects for PLANAR 8-bit sources only (:2108-2112).
        // An NV12 source has no special converter on the CPU: it runs the generic lines below like any other pair (rounds 1-3 gave it
        // the wrapper's t | t << 8 too; the reference's real core against its own CPU path showed the difference, round 4:
        // tests/test_libswscale_core.py)
        c->mode = MODE_DEPTH;
    } else if (!same && (srcFormat == GMAT_PIX_FMT_RGB24 || srcFormat == GMAT_PIX_FMT_BGR24 || is_yuv420(srcFormat)) &&
               is_packed_rgb(dstFormat)) {
        c->mode = MODE_SCALE;
        r = ensure_scaler(c);
        if (r == GMAT_ERR(ENOSYS) && !is_yuv420(srcFormat)) {
            // the RGB scaler writes full chroma only; with SWS_FAST_BILINEAR an RGB source keeps the half-chroma writer
            // (utils.c:1439-1447): the plane scaler with its RGB loader has that writer
            c->rgbViaPlanes = true;
            c->fused = 2;
            r = ensure_scaler(c);
        }
    } else if ((srcFormat == GMAT_PIX_FMT_RGB24 || srcFormat == GMAT_PIX_FMT_BGR24) &&
               ((!same && is_yuv8_src(dstFormat)) || is_dst10(dstFormat) || dstFormat == GMAT_PIX_FMT_YUV444P)) {       // (round 6: planar YUV420P10LE too — swscale_cuda.c:34-44 lists it; the format sweep found the pair refused)
        // packed RGB scaled into a YUV frame (one libswscale context: rgb24ToY / ToUV(_half), hScale16To15_c, planar
        // vertical stage): the plane scaler with its RGB loader
        c->mode = MODE_SCALE;
        c->rgbViaPlanes = true;
        c->fused = 2;
        r = ensure_scaler(c);
    } else if (same && is_p01x(srcFormat) && srcFormat == dstFormat) {
        c->mode = MODE_PLANECOPY;            // equal format and size: libswscale copies the planes verbatim
    } else if (same && pl16_depth(srcFormat) && dstFormat == srcFormat) {
        c->mode = MODE_PLANECOPY;
    } else if (same && ((pl16_depth(srcFormat) && !is_priv_planes(srcFormat) && srcFormat != GMAT_PIX_FMT_YUV444P16LE && dstFormat == GMAT_PIX_FMT_YUV420P) ||
                        (srcFormat == GMAT_PIX_FMT_YUV444P16LE && dstFormat == GMAT_PIX_FMT_YUV444P))) {
        // equal size, a deeper planar format into the 8-bit one of the SAME layout: planarCopyWrapper with its own dither tables
        // (swscale_unscaled.c:1743-1800, 2293-2309) while the two ranges agree; gmat_sws_setRange moves it onto the generic lines when they do not
        c->mode = MODE_PLANE_DOWN;
    } else if (is_plane_src(srcFormat) && is_dst16(dstFormat)) {
        // 16-bit destination: 19-bit intermediates, the two-pass path of k_scale16.hip (equal-size 8-bit 4:2:0 sources were
        // taken above as the depth expansion, equal format as the plane copy)
        c->mode = MODE_SCALE16;
        r = init_scale16(c);
    } else if ((is_plane_src(srcFormat)) && is_dst10(dstFormat)) {
        // scaled (or 16-bit sourced) P010LE output: dstBpc = 10 keeps the 15-bit intermediates; yuv2p010l1_c /
        // yuv2p010lX_c / yuv2p010cX_c (output.c:459-519).  Equal-size 8-bit 4:2:0 sources were taken above (MODE_DEPTH).
        c->mode = MODE_SCALE;
        r = ensure_scaler(c);
    } else if ((is_p01x(srcFormat) || pl16_depth(srcFormat)) && (is_packed_rgb(dstFormat) || is_yuv8_src(dstFormat))) {
        // 16-bit semi-planar sources (scale_cuda's list, vf_scale_cuda.c:45-54) to any 8-bit destination, any size:
        // libswscale has no special converter for them, the generic path's hScale16To15_c brings the samples to the
        // same 15-bit lines an 8-bit source gives
        c->mode = MODE_SCALE;
        r = ensure_scaler(c);
    } else if (srcFormat == GMAT_PIX_FMT_YUV444P && (is_packed_rgb(dstFormat) || is_yuv8_src(dstFormat))) {
        // planar 4:4:4 source (scale_cuda's format list, vf_scale_cuda.c:45-54): always the generic plane scaler —
        // even at the same size the chroma planes are filtered (2:1 for 4:2:0 outputs)
        c->mode = MODE_SCALE;
        r = ensure_scaler(c);
    } else if (is_yuv420(srcFormat) && dstFormat == GMAT_PIX_FMT_YUV444P) {
        // 4:2:0 -> planar 4:4:4 (any size): the chroma planes go through their own 1:2 filters
        c->mode = MODE_SCALE;
        r = ensure_scaler(c);
    } else if (!same && is_yuv420(srcFormat) && is_yuv420(dstFormat)) {
        // scale_cuda's main job (vf_scale_cuda.c:428-501): 4:2:0 in, 4:2:0 out at another size.  Arithmetic of
        // the CPU generic path: hScale8To15_c per plane, yuv2planeX_8_c / yuv2nv12cX_c vertically.
        c->mode = MODE_SCALE;
        r = ensure_scaler(c);
    } else {
        logf(LOG_ERROR, "gmat_sws_getContext: unsupported conversion %d %dx%d -> %d %dx%d", srcFormat, srcW, srcH,
             dstFormat, dstW, dstH);
        r = GMAT_ERR(ENOSYS);
    }
    if (r < 0) {
        delete c;
        return nullptr;
    }
    c->unscaledMode = c->mode;
    return c;
}

void gmat_sws_setStream(GmatSwsContext *c, void *stream)
{
    if (c) c->stream = (hipStream_t)stream;
}

void gmat_sws_freeContext(GmatSwsContext *c) { delete c; }

int gmat_sws_setColorspace(GmatSwsContext *c, int colorspace, int srcFullRange)
{
    if (!c || colorspace < 0 || colorspace > 10) return GMAT_ERR(EINVAL);
    if (c->mode == MODE_VIA_PLANES16) {
        // an RGB source: the matrix belongs to the RGB -> YUV stage of a YUV destination (read per call above); no source range
        if (srcFullRange) return GMAT_ERR(ENOSYS);
        c->colorspace = colorspace;
        return 0;
    }
    if (c->inner) return gmat_sws_setColorspace(c->inner, colorspace, srcFullRange);
    // a YUV source: the matrix (and range) of its YUV -> RGB stage; an RGB source with a YUV destination: the matrix of
    // the RGB -> YUV stage (fill_rgb2yuv_table, utils.c:765-858), limited range only
    if (is_packed_rgb(c->srcFormat) && srcFullRange) return GMAT_ERR(ENOSYS);
    c->colorspace = colorspace;
    c->srcFullRange = srcFullRange;
    c->y2r = make_yuv2rgb_consts(colorspace, srcFullRange != 0);
    if (c->rgbViaPlanes) { c->yuvReady = false; return init_yuv_scaler(c); }      // the loader's rgb2yuv constants
    return 0;
}

int gmat_sws_setRange(GmatSwsContext *c, int srcFullRange, int dstFullRange)
{
    if (!c) return GMAT_ERR(EINVAL);
    if (c->mode == MODE_VIA_PLANES16) {
        // as for the 8-bit RGB sources: no range of its own; a full-range YUV destination is the limited -> full conversion of the
        // 15-bit lines, which the inner context carries (its "source" is the limited-range planes)
        if (srcFullRange) return GMAT_ERR(ENOSYS);
        if (is_packed_rgb(c->dstFormat) || is_rgb64(c->dstFormat)) return dstFullRange ? GMAT_ERR(ENOSYS) : 0;
        return gmat_sws_setRange(c->inner, 0, dstFullRange);
    }
    if (c->inner) return gmat_sws_setRange(c->inner, srcFullRange, dstFullRange);
    if (c->rgbViaPlanes && is_packed_rgb(c->dstFormat)) return (srcFullRange || dstFullRange) ? GMAT_ERR(ENOSYS) : 0;   // RGB ends: no range
    if (c->mode == MODE_RGB2YUV || c->rgbViaPlanes) {
        // an RGB source has no range of its own (forced to 0, utils.c:902-1030): a full-range destination is the
        // limited -> full conversion of the 15-bit lines (lum/chrRangeToJpeg_c), as in the second half of libswscale's
        // YUV -> RGB -> YUV cascade for differing matrices (utils.c:966-1036)
        if (srcFullRange) return GMAT_ERR(ENOSYS);
        c->rangeConv = dstFullRange ? 1 : 0;
        return 0;
    }
    {
        // same-size 8-bit planar -> high-depth planar with the same subsampling is planarCopyWrapper in libswscale
        // (swscale_unscaled.c:1803-1862), taken when the two ranges agree (utils.c:1996-2000).  It SHIFTS chroma and limited-range
        // luma — what the generic lines compute too, so a limited-range context stays on them — and bit-replicates the luma of a
        // full-range source: its own little kernel.
        const bool same = c->srcW == c->dstW && c->srcH == c->dstH;
        const bool pair = (c->srcFormat == GMAT_PIX_FMT_YUV420P && (c->dstFormat == GMAT_PIX_FMT_YUV420P10LE || c->dstFormat == GMAT_PIX_FMT_YUV420P16LE)) ||
                          (c->srcFormat == GMAT_PIX_FMT_YUV444P && c->dstFormat == GMAT_PIX_FMT_YUV444P16LE);
        if (same && pair && srcFullRange && dstFullRange) {
            if (c->mode != MODE_PLANE_UP) c->unscaledMode = c->mode;       // the generic mode to return to
            c->mode = MODE_PLANE_UP;
            c->rangeConv = 0;
            return 0;
        }
        if (c->mode == MODE_PLANE_UP) c->mode = c->unscaledMode;           // leaving it: back on the generic lines
    }
    if (!is_plane_src(c->srcFormat) || !(is_yuv8_src(c->dstFormat) || is_p01x(c->dstFormat) || pl16_depth(c->dstFormat))) {
        // RGB ends have no range of their own (utils.c:902-1030 forces them to 0); the source range of a
        // YUV -> RGB context is part of gmat_sws_setColorspace
        return (srcFullRange || dstFullRange) ? GMAT_ERR(ENOSYS) : 0;
    }
    const int conv = (!!srcFullRange == !!dstFullRange) ? 0 : (dstFullRange ? 1 : 2);
    if (c->unscaledMode == MODE_PLANE_DOWN) c->planeDownFull = srcFullRange && dstFullRange;     // planarCopyWrapper's luma rule follows srcRange
    if (conv == c->rangeConv) return 0;
    if (c->cross) { gmat_sws_freeContext(c->cross); c->cross = nullptr; }
    // a same-size context is a plane copy / depth expansion only while the ranges agree (utils.c:1996-2000: the
    // special converters are skipped when srcRange != dstRange); otherwise it runs the generic path, whose 15-bit
    // lines carry the conversion (8-bit 4:2:0 and P010LE destinations: lum / chrRange{To,From}Jpeg_c on the 15-bit lines; 16-bit
    // destinations: their ...16_c twins on the 19-bit lines, swscale.c:189-226).
    const bool same = c->srcW == c->dstW && c->srcH == c->dstH;
    const bool special = same && (c->unscaledMode == MODE_YUV2YUV || c->unscaledMode == MODE_DEPTH || c->unscaledMode == MODE_PLANECOPY || c->unscaledMode == MODE_PLANE_DOWN);
    if (special) {
        const bool generic15 = is_yuv8_src(c->dstFormat) || is_dst10(c->dstFormat);
        c->rangeConv = conv;
        c->shift8Ident = -1;
        if (conv && !generic15) { c->mode = MODE_SCALE16; return init_scale16(c); }       // 16-bit destination: the 19-bit lines
        if (conv) { c->mode = MODE_SCALE; c->fused = 2; return ensure_scaler(c); }
        c->mode = c->unscaledMode;
        return 0;
    }
    if (c->mode == MODE_SCALE16 && conv && is_rgb64(c->dstFormat)) return GMAT_ERR(ENOSYS);   // an RGB end has no range (swscale.c:536)
    c->rangeConv = conv;
    return 0;
}

int gmat_sws_setChromaPos(GmatSwsContext *c, int src_h_chr_pos, int src_v_chr_pos, int dst_h_chr_pos, int dst_v_chr_pos)
{
    if (!c) return GMAT_ERR(EINVAL);
    if ((c->mode != MODE_SCALE && c->mode != MODE_SCALE16) || !is_plane_src(c->srcFormat) || is_priv_planes(c->srcFormat)) return GMAT_ERR(ENOSYS);
    const int np[4] = {src_h_chr_pos, src_v_chr_pos, dst_h_chr_pos, dst_v_chr_pos};
    for (int i = 0; i < 4; i++)
        if (np[i] < -513 || np[i] > 512) return GMAT_ERR(EINVAL);           // option range, options.c:67-70
    std::memcpy(c->chrPos, np, sizeof(np));
    if (c->cross) { gmat_sws_freeContext(c->cross); c->cross = nullptr; }
    if (c->mode == MODE_SCALE16) return init_scale16(c);                     // the 19-bit path's own filter banks
    c->fused = 2;
    c->yuvReady = false;                     // the chroma filter banks depend on the positions
    return init_yuv_scaler(c);
}

int gmat_sws_setFused(GmatSwsContext *c, int fused)
{
    if (!c || fused < 0 || fused > 2) return GMAT_ERR(EINVAL);
    if (c->inner) return gmat_sws_setFused(c->inner, fused);
    if ((c->srcFormat == GMAT_PIX_FMT_YUV444P || pl16_depth(c->srcFormat) || is_p01x(c->srcFormat) || c->rgbViaPlanes) && fused != 2) return GMAT_ERR(ENOSYS);
    c->fused = fused;
    if (c->mode == MODE_SCALE) return ensure_scaler(c);
    return 0;
}

int gmat_sws_setProfileBuffer(GmatSwsContext *c, uint8_t *devbuf)
{
    if (!c) return GMAT_ERR(EINVAL);
    c->prof = (unsigned long long *)devbuf;
    return 0;
}

const char *gmat_sws_lastKernel(const GmatSwsContext *c) { return c ? c->lastKernel : ""; }
int gmat_sws_lastLaunchFrames(const GmatSwsContext *c) { return c ? c->lastLaunchFrames : 0; }
int gmat_sws_streamHandoffs(const GmatSwsContext *c) { return c ? c->handoffs : 0; }

int gmat_sws_getFilter(const GmatSwsContext *c, int which, int16_t *coef, int32_t *pos, int cap, int *count)
{
    if (c && c->inner) return gmat_sws_getFilter(c->inner, which, coef, pos, cap, count);
    if (!c || c->mode != MODE_SCALE) return GMAT_ERR(EINVAL);
    const FilterBank *fb;
    const ScalePlan &pl = ((is_plane_src(c->srcFormat) || c->rgbViaPlanes) && c->fused == 2) ? c->planYuv : c->plan;
    switch (which) {
    case 0: fb = &pl.hLum; break;
    case 1: fb = &pl.hChr; break;
    case 2: fb = &pl.vLum; break;
    case 3: fb = &pl.vChr; break;
    default: return GMAT_ERR(EINVAL);
    }
    if (count) *count = fb->count;
    const int n = std::min(cap, fb->count);
    if (coef) std::memcpy(coef, fb->coef.data(), (size_t)n * fb->taps * sizeof(int16_t));
    if (pos)  std::memcpy(pos, fb->pos.data(), (size_t)n * sizeof(int32_t));
    return fb->taps;
}


static int sws_scale_impl(GmatSwsContext *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                          int srcSliceH, uint8_t *const dst[], const int dstStride[]);
int gmat_sws_scale(GmatSwsContext *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                   int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    lines_ready(c);
    if (!c || !owns_intermediates(c)) return sws_scale_impl(c, src, srcStride, srcSliceY, srcSliceH, dst, dstStride);
    if (int r = stream_handoff_acquire(c, c->stream); r < 0) return r;
    const int h = sws_scale_impl(c, src, srcStride, srcSliceY, srcSliceH, dst, dstStride);
    if (h > 0 && (c->interTouched || sws_shares_intermediate(c)))
        if (int r = stream_handoff_release(c, c->stream); r < 0) return r;
    return h;
}
static int sws_scale_impl(GmatSwsContext *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                          int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    if (!c || !src || !dst || !srcStride || !dstStride || !src[0] || !dst[0]) {
        logf(LOG_ERROR, "gmat_sws_scale: one of the input parameters to sws_scale() is NULL");
        return GMAT_ERR(EINVAL);
    }
    // the core's slice check (swscale.c:902-907), then — like ff_swscale_cuda, which never reads its slice arguments
    // (swscale_cuda.c:273-479) — the WHOLE frame is converted whatever slice was named; src[] / dst[] are the frame's planes
    if (srcSliceY < 0 || srcSliceH < 0 || srcSliceY + srcSliceH > c->srcH) {
        logf(LOG_ERROR, "gmat_sws_scale: Slice parameters %d, %d are invalid", srcSliceY, srcSliceH);
        return GMAT_ERR(EINVAL);
    }
    if (srcSliceH == 0) return 0;                    // a trailing empty slice (swscale.c:927-928)
    if (int r = check_device(c, "gmat_sws_scale"); r < 0) return r;
    c->lastLaunchFrames = 1;
    const bool planarYuv = c->srcFormat == GMAT_PIX_FMT_YUV420P || c->srcFormat == GMAT_PIX_FMT_YUV444P || pl16_depth(c->srcFormat);
    if (is_plane_src(c->srcFormat) && (!src[1] || (planarYuv && !src[2]))) return GMAT_ERR(EINVAL);

    int r = 0;
    // NV12 -> P010LE / P016LE at equal size: the generic lines with one-tap filters make t << 8 of every sample (see nv12_shift8_kernel); a range
    // conversion, the only thing that changes the lines' values here, keeps the plane scaler
    if (shift8_shortcut(c)) {
        if (!dst[1]) return GMAT_ERR(EINVAL);
        c->lastKernel = "nv12_shift8_kernel";
        r = launch_nv12_shift8(src[0], srcStride[0], src[1], srcStride[1], dst[0], dstStride[0], dst[1], dstStride[1], c->srcW, c->srcH, c->stream);
        return r < 0 ? r : c->dstH;
    }
    switch (c->mode) {
    case MODE_YUV2RGB:
        c->lastKernel = "yuv2rgb_kernel";
        r = launch_yuv2rgb(yuv_src_of(c->srcFormat, src, srcStride), dst[0], dstStride[0], c->srcW, c->srcH,
                           c->dstFormat, c->y2r, c->stream);
        break;
    case MODE_RGBPF32:
        c->lastKernel = "nv12_to_rgbpf32_kernel";
        r = launch_nv12_to_rgbpf32(yuv_src_of(c->srcFormat, src, srcStride), dst[0], dstStride[0], c->srcW, c->srcH,
                                   c->y2r, c->stream);
        break;
    case MODE_SWAP_RB:
        c->lastKernel = "swap_rb24_kernel";
        r = launch_swap_rb24(src[0], srcStride[0], dst[0], dstStride[0], c->srcW, c->srcH, c->stream);
        break;
    case MODE_COPY:
        c->lastKernel = "copy2d";
        r = launch_copy2d(src[0], srcStride[0], dst[0], dstStride[0], c->srcW * (is_rgb64(c->srcFormat) ? 8 : bytes_per_pixel(c->srcFormat)),
                          c->srcH, c->stream);
        break;
    case MODE_RGB2YUV: {
        const bool dnv12 = c->dstFormat == GMAT_PIX_FMT_NV12;
        if (!dst[1] || (!dnv12 && !dst[2])) { r = GMAT_ERR(EINVAL); break; }
        Rgb2YuvLaunch L;
        L.src = src[0]; L.ss = srcStride[0]; L.bgr = c->srcFormat == GMAT_PIX_FMT_BGR24;
        L.y = dst[0]; L.ys = dstStride[0]; L.u = dst[1]; L.us = dstStride[1];
        L.v = dnv12 ? nullptr : dst[2]; L.vs = dnv12 ? 0 : dstStride[2]; L.nv12 = dnv12;
        L.w = c->srcW; L.h = c->srcH;
        L.vChr = c->r2yVChr;
        L.rowStart = (const int32_t *)c->dR2YrowStart.p; L.rowCount = (const int32_t *)c->dR2YrowCount.p;
        L.maxRows = c->r2y.maxRows;
        L.k = make_rgb2yuv_consts(c->colorspace);      // the destination's matrix (fill_rgb2yuv_table, utils.c:765-858)
        L.toJpeg = c->rangeConv == 1;
        L.stripOk = c->r2y.stripOk; for (int k = 0; k < 4; k++) L.vC[k] = c->r2y.vC[k];
        L.px = c->px4 ? 4 : 3;
        if (c->px4 && !rgb2yuv420_strip_takes(L)) { r = GMAT_ERR(EINVAL); break; }      // (four-byte pixels were promised the kernel that reads them)
        c->lastKernel = rgb2yuv420_strip_takes(L) ? "rgb2yuv420s_kernel" : "rgb2yuv420_kernel";
        r = launch_rgb2yuv420(L, c->stream);
        break;
    }
    case MODE_VIA_INNER: {
        if (rg_block_takes_px4(c->inner, 1, src[0], srcStride[0], dst[0], dstStride[0]) || planes_fused_takes_px4(c->inner, 1, src, srcStride, dst, dstStride) ||
            r2y_strip_takes_px4(c->inner, src, srcStride, dst, dstStride)) {
            gmat_sws_setStream(c->inner, (void *)c->stream);
            c->inner->px4 = 1; c->inner->px4Alpha = c->needAlpha;
            r = gmat_sws_scale(c->inner, src, srcStride, 0, c->srcH, dst, dstStride);
            c->inner->px4 = 0; c->inner->px4Alpha = 0;
            c->lastKernel = c->inner->lastKernel;
            if (r >= 0) return r;
            //  the inner context answers EINVAL before it launches anything when the kernel that reads four-byte pixels is not the one
            // its own rules pick after all: the 32 -> 24-bit pass below still serves the call (a caller's own EINVAL comes back from there too)
            if (r != GMAT_ERR(EINVAL)) break;
        }
        if (!c->inter) {
            c->interStride = align_up(c->srcW * 3, 256);
            if (hipMalloc((void **)&c->inter, (size_t)c->interStride * c->srcH) != hipSuccess) { r = GMAT_ERR(ENOMEM); break; }
        }
        if ((r = launch_repack_rgb(src[0], srcStride[0], 4, c->inter, c->interStride, 3, c->srcW, c->srcH, 0, c->stream)) < 0) break;
        gmat_sws_setStream(c->inner, (void *)c->stream);
        const uint8_t *isrc[4] = {c->inter, nullptr, nullptr, nullptr};
        const int istr[4] = {c->interStride, 0, 0, 0};
        r = gmat_sws_scale(c->inner, isrc, istr, 0, c->srcH, dst, dstStride);
        c->lastKernel = c->inner->lastKernel;
        if (r >= 0 && c->needAlpha) {
            const int ra = alpha_run8(c, src[0] + 3, srcStride[0], 208, 4, dst[0], dstStride[0]);
            if (ra < 0) r = ra;
        }
        if (r >= 0) return r;
        break;
    }
    case MODE_VIA_PLANES16: {
        if (is_rgb64(c->srcFormat) && (((uintptr_t)src[0] | (uintptr_t)srcStride[0]) & 1) != 0) { r = GMAT_ERR(EINVAL); break; }
        const ScalePlan &p = active_plan(c->inner);
        uint8_t *pl = (uint8_t *)c->planes16.p;
        // the RGB -> YUV stage takes the DESTINATION's matrix when that is YUV (fill_rgb2yuv_table, utils.c:765-858); an RGB
        // destination leaves both stages at the default (as every RGB -> RGB context of this library)
        const Rgb2YuvConsts k = make_rgb2yuv_consts(is_packed_rgb(c->dstFormat) || is_rgb64(c->dstFormat) ? GMAT_SWS_CS_DEFAULT : c->colorspace);
        const bool s64 = is_rgb64(c->srcFormat);
        if (s64) r = launch_rgb64_planes(src[0], srcStride[0], c->srcW, c->srcH, p.chrSrcW, p.chrSrcHSub, c->srcFormat == GMAT_PIX_FMT_BGRA64LE, k,
                                         pl, c->p16Stride[0], pl + c->p16Off[1], c->p16Stride[1], pl + c->p16Off[2], c->p16Stride[1], c->stream);
        else     r = launch_rgb8_planes(src[0], srcStride[0], c->srcW, c->srcH, p.chrSrcW, p.chrSrcHSub,
                                        c->srcFormat == GMAT_PIX_FMT_BGR24 || c->srcFormat == GMAT_PIX_FMT_BGRA, bytes_per_pixel(c->srcFormat), k,
                                        pl, c->p16Stride[0], pl + c->p16Off[1], c->p16Stride[1], pl + c->p16Off[2], c->p16Stride[1], c->stream);
        if (r < 0) break;
        const bool a64 = c->needAlpha && is_rgb64(c->dstFormat);
        if (a64) {                           // the alpha plane's 19-bit lines, an operand of the inner context's writer
            if (s64) r = launch_hscale19(src[0] + 6, srcStride[0], 16, 8, c->srcW, c->srcH, c->daH, (int32_t *)c->alphaLines.p, c->dstW, c->stream);
            else     r = launch_hscale19(src[0] + 3, srcStride[0], 208, 4, c->srcW, c->srcH, c->daH, (int32_t *)c->alphaLines.p, c->dstW, c->stream);
            if (r < 0) break;
            c->inner->alpha19 = (const int32_t *)c->alphaLines.p;
        }
        gmat_sws_setStream(c->inner, (void *)c->stream);
        const uint8_t *isrc[4] = {pl, pl + c->p16Off[1], pl + c->p16Off[2], nullptr};
        const int istr[4] = {c->p16Stride[0], c->p16Stride[1], c->p16Stride[1], 0};
        r = gmat_sws_scale(c->inner, isrc, istr, 0, c->srcH, dst, dstStride);
        c->inner->alpha19 = nullptr;
        c->lastKernel = c->inner->lastKernel;
        if (r >= 0 && c->needAlpha && !a64) {
            const int ra = alpha_run8(c, src[0] + 6, srcStride[0], 16, 8, dst[0], dstStride[0]);
            if (ra < 0) r = ra;
        }
        if (r >= 0) return r;
        break;
    }
    case MODE_SCALE16: {
        const bool rgb64 = is_rgb64(c->dstFormat);
        if (!rgb64 && !dst[1]) { r = GMAT_ERR(EINVAL); break; }
        const ScalePlan &p = c->plan16;
        const bool pl16 = pl16_depth(c->srcFormat) != 0;
        const bool s16 = is_p01x(c->srcFormat) || pl16;
        const int odd = s16 ? 1 : 0;
        if ((((uintptr_t)dst[0] | (uintptr_t)dstStride[0] | (rgb64 ? 0 : ((uintptr_t)dst[1] | (uintptr_t)dstStride[1]))) & 1) != 0 ||
            (odd && (((uintptr_t)src[0] | (uintptr_t)src[1] | (uintptr_t)srcStride[0] | (uintptr_t)srcStride[1]) & 1) != 0)) { r = GMAT_ERR(EINVAL); break; }
        const int kind = scale16_kind(c->srcFormat);
        const int bps = s16 ? 2 : 1;
        int32_t *ly = (int32_t *)c->line16[0].p, *lu = (int32_t *)c->line16[1].p, *lv = (int32_t *)c->line16[2].p;
        c->lastKernel = "hscale19_kernel+vscale16_kernel";
        const int rcL = rgb64 ? 0 : c->rangeConv, rcC = rgb64 ? 0 : c->rangeConv ? c->rangeConv + 2 : 0;      // (swscale.c:536: not for RGB destinations)
        if (c->s19.ok && !(rgb64 && c->alpha19)) {                            // (the alpha lines of an RGBA source: vrgba64_kernel's operand)
            const bool semiS = c->srcFormat == GMAT_PIX_FMT_NV12 || is_p01x(c->srcFormat);
            const uint8_t *const sp[4] = {src[0], src[1], semiS ? nullptr : src[2], nullptr};
            uint8_t *const dp[4] = {dst[0], rgb64 ? nullptr : dst[1], (rgb64 || c->dstFormat == GMAT_PIX_FMT_P016LE) ? nullptr : dst[2], nullptr};
            r = scale19_frames(c, 1, sp, srcStride, dp, dstStride, c->stream);
            break;
        }
        if ((r = launch_hscale19(src[0], srcStride[0], kind, bps, c->srcW, c->srcH, c->d16[0], ly, c->dstW, c->stream, 0, rcL)) < 0) break;
        const bool semi = c->srcFormat == GMAT_PIX_FMT_NV12 || is_p01x(c->srcFormat);           // interleaved U, V
        if (!semi && !src[2]) { r = GMAT_ERR(EINVAL); break; }
        const uint8_t *pu = src[1], *pv = semi ? src[1] + bps : src[2];
        const int su = srcStride[1], sv = semi ? srcStride[1] : srcStride[2], cstep = semi ? 2 * bps : bps;
        if ((r = launch_hscale19(pu, su, kind, cstep, p.chrSrcW, p.chrSrcH, c->d16[1], lu, p.chrDstW, c->stream, 0, rcC)) < 0) break;
        if ((r = launch_hscale19(pv, sv, kind, cstep, p.chrSrcW, p.chrSrcH, c->d16[1], lv, p.chrDstW, c->stream, 0, rcC)) < 0) break;
        if (rgb64) {
            c->lastKernel = "hscale19_kernel+vrgba64_kernel";
            r = launch_vrgba64(ly, lu, lv, c->dstW, c->srcH, p.chrDstW, p.chrSrcH, c->d16[2], c->d16[3], p.chrDstW == c->dstW ? 0 : 1,
                               dst[0], dstStride[0], c->dstW, c->dstH, c->dstFormat == GMAT_PIX_FMT_BGRA64LE,
                               make_yuv2rgb_consts(c->colorspace, c->srcFullRange != 0), c->stream, c->alpha19);
            break;
        }
        if ((r = launch_vscale16(ly, nullptr, c->dstW, c->srcH, c->d16[2], dst[0], dstStride[0], c->dstW, c->dstH, c->stream)) < 0) break;
        if (is_pl16_dst(c->dstFormat)) {
            if (!dst[2] || (((uintptr_t)dst[2] | (uintptr_t)dstStride[2]) & 1)) { r = GMAT_ERR(EINVAL); break; }
            if ((r = launch_vscale16(lu, nullptr, p.chrDstW, p.chrSrcH, c->d16[3], dst[1], dstStride[1], p.chrDstW, p.chrDstH, c->stream)) < 0) break;
            r = launch_vscale16(lv, nullptr, p.chrDstW, p.chrSrcH, c->d16[3], dst[2], dstStride[2], p.chrDstW, p.chrDstH, c->stream);
            break;
        }
        r = launch_vscale16(lu, lv, p.chrDstW, p.chrSrcH, c->d16[3], dst[1], dstStride[1], p.chrDstW, p.chrDstH, c->stream);
        break;
    }
    case MODE_PLANE_UP: {
        if (!src[1] || !src[2] || !dst[1] || !dst[2]) { r = GMAT_ERR(EINVAL); break; }
        for (int i = 0; i < 3; i++)
            if ((((uintptr_t)dst[i] | (uintptr_t)dstStride[i]) & 1) != 0) r = GMAT_ERR(EINVAL);
        if (r < 0) break;
        const int depth = pl16_depth(c->dstFormat), sub = c->srcFormat == GMAT_PIX_FMT_YUV444P ? 0 : 1;
        const int cw = ceil_rshift(c->srcW, sub), ch = ceil_rshift(c->srcH, sub);
        c->lastKernel = "plane_copy_up_kernel";
        if ((r = launch_plane_copy_up(src[0], srcStride[0], dst[0], dstStride[0], c->srcW, c->srcH, depth, 1, c->stream)) < 0) break;
        for (int i = 1; i < 3 && r >= 0; i++) r = launch_plane_copy_up(src[i], srcStride[i], dst[i], dstStride[i], cw, ch, depth, 0, c->stream);
        break;
    }
    case MODE_PLANE_DOWN: {
        if (!src[1] || !src[2] || !dst[1] || !dst[2]) { r = GMAT_ERR(EINVAL); break; }
        for (int i = 0; i < 3; i++)
            if ((((uintptr_t)src[i] | (uintptr_t)srcStride[i]) & 1) != 0) r = GMAT_ERR(EINVAL);
        if (r < 0) break;
        const int depth = pl16_depth(c->srcFormat), sub = c->srcFormat == GMAT_PIX_FMT_YUV444P16LE ? 0 : 1;
        const int cw = ceil_rshift(c->srcW, sub), ch = ceil_rshift(c->srcH, sub);
        c->lastKernel = "plane_copy_down3_kernel";
        {
            const uint8_t *const sp[3] = {src[0], src[1], src[2]};
            uint8_t *const dp[3] = {dst[0], dst[1], dst[2]};
            r = launch_planes_copy_down(sp, srcStride, dp, dstStride, c->srcW, c->srcH, cw, ch, depth, c->planeDownFull ? 0 : 1, c->stream);
        }
        break;
    }
    case MODE_PLANECOPY: {
        if (!src[1] || !dst[1]) { r = GMAT_ERR(EINVAL); break; }
        c->lastKernel = "copy2d_kernel";
        r = launch_copy2d(src[0], srcStride[0], dst[0], dstStride[0], 2 * c->srcW, c->srcH, c->stream);
        if (pl16_depth(c->srcFormat)) {
            if (!src[2] || !dst[2]) { r = GMAT_ERR(EINVAL); break; }
            const int sub = c->srcFormat == GMAT_PIX_FMT_YUV444P16LE ? 0 : 1;
            const int cw = (c->srcW + sub) >> sub, ch = (c->srcH + sub) >> sub;
            for (int i = 1; i < 3 && r >= 0; i++) r = launch_copy2d(src[i], srcStride[i], dst[i], dstStride[i], 2 * cw, ch, c->stream);
            break;
        }
        if (r >= 0) r = launch_copy2d(src[1], srcStride[1], dst[1], dstStride[1], 4 * ((c->srcW + 1) / 2), (c->srcH + 1) / 2, c->stream);
        break;
    }
    case MODE_REPACK: {
        const bool srcRgbOrder = c->srcFormat == GMAT_PIX_FMT_RGB24 || c->srcFormat == GMAT_PIX_FMT_RGBA;
        const bool dstRgbOrder = c->dstFormat == GMAT_PIX_FMT_RGB24 || c->dstFormat == GMAT_PIX_FMT_RGBA;
        c->lastKernel = "repack_rgb_kernel";
        r = launch_repack_rgb(src[0], srcStride[0], bytes_per_pixel(c->srcFormat), dst[0], dstStride[0], bytes_per_pixel(c->dstFormat),
                              c->srcW, c->srcH, (srcRgbOrder != dstRgbOrder ? 1 : 0) | ((c->src0 && !c->dst0) ? 2 : 0), c->stream);
        break;
    }
    case MODE_RGB2YUV444: {
        if (!dst[1] || !dst[2]) { r = GMAT_ERR(EINVAL); break; }
        c->lastKernel = "rgb2yuv444_kernel";
        r = launch_rgb2yuv444(src[0], srcStride[0], c->srcFormat == GMAT_PIX_FMT_BGR24, dst[0], dstStride[0], dst[1], dstStride[1],
                              dst[2], dstStride[2], c->srcW, c->srcH, make_rgb2yuv_consts(c->colorspace), c->stream);
        break;
    }
    case MODE_YUV2YUV: {
        // nv12ToPlanarWrapper / planarToNv12Wrapper / plane copies (swscale_unscaled.c): lossless re-layout
        const bool snv = c->srcFormat == GMAT_PIX_FMT_NV12, dnv = c->dstFormat == GMAT_PIX_FMT_NV12;
        const int cw = ceil_rshift(c->srcW, 1), ch = ceil_rshift(c->srcH, 1);
        if (!dst[1] || (!dnv && !dst[2])) { r = GMAT_ERR(EINVAL); break; }
        if (snv != dnv && !(GMAT_KNOB("GMAT_NO_RELAYOUT_FUSED") && atoi(GMAT_KNOB("GMAT_NO_RELAYOUT_FUSED"))) &&
            yuv420_relayout_takes(snv, src[0], srcStride[0], src[1], srcStride[1], snv ? nullptr : src[2], snv ? 0 : srcStride[2],
                                  dst[0], dstStride[0], dst[1], dstStride[1], dnv ? nullptr : dst[2], dnv ? 0 : dstStride[2])) {
            // round 4: luma copy and chroma (de)interleave in one launch
            c->lastKernel = "yuv420_relayout_kernel";
            r = launch_yuv420_relayout(snv, src[0], srcStride[0], src[1], srcStride[1], snv ? nullptr : src[2], snv ? 0 : srcStride[2],
                                       dst[0], dstStride[0], dst[1], dstStride[1], dnv ? nullptr : dst[2], dnv ? 0 : dstStride[2], c->srcW, c->srcH, c->stream);
            break;
        }
        c->lastKernel = snv == dnv ? "copy2d" : (snv ? "uv_deinterleave_kernel" : "uv_interleave_kernel");
        if ((r = launch_copy2d(src[0], srcStride[0], dst[0], dstStride[0], c->srcW, c->srcH, c->stream)) < 0) break;
        if (snv && dnv) r = launch_copy2d(src[1], srcStride[1], dst[1], dstStride[1], 2 * cw, ch, c->stream);
        else if (!snv && !dnv) {
            if ((r = launch_copy2d(src[1], srcStride[1], dst[1], dstStride[1], cw, ch, c->stream)) < 0) break;
            r = launch_copy2d(src[2], srcStride[2], dst[2], dstStride[2], cw, ch, c->stream);
        } else if (snv) r = launch_uv_relayout(1, src[1], srcStride[1], nullptr, 0, dst[1], dstStride[1], dst[2], dstStride[2], cw, ch, c->stream);
        else r = launch_uv_relayout(0, src[1], srcStride[1], src[2], srcStride[2], dst[1], dstStride[1], nullptr, 0, cw, ch, c->stream);
        break;
    }
    case MODE_FROM_PF32: {
        c->lastKernel = "rgbpf32_to_rgb24_kernel";
        if (!is_yuv420(c->dstFormat)) {
            r = launch_rgbpf32_to_rgb24(src[0], srcStride[0], dst[0], dstStride[0], c->srcW, c->srcH,
                                        c->dstFormat == GMAT_PIX_FMT_BGR24, c->stream);
            break;
        }
        const bool dnv12 = c->dstFormat == GMAT_PIX_FMT_NV12;
        if (!dst[1] || (!dnv12 && !dst[2])) { r = GMAT_ERR(EINVAL); break; }
        {
            // round 4: one kernel from the floats to the 4:2:0 planes where the strip converter's rule holds (no RGB24 intermediate)
            Rgb2YuvLaunch F;
            F.src = src[0]; F.ss = srcStride[0]; F.bgr = 0;
            F.y = dst[0]; F.ys = dstStride[0]; F.u = dst[1]; F.us = dstStride[1];
            F.v = dnv12 ? nullptr : dst[2]; F.vs = dnv12 ? 0 : dstStride[2]; F.nv12 = dnv12;
            F.w = c->srcW; F.h = c->srcH; F.vChr = c->r2yVChr; F.rowStart = nullptr; F.rowCount = nullptr; F.maxRows = c->r2y.maxRows;
            F.k = make_rgb2yuv_consts(c->colorspace);
            F.stripOk = c->r2y.stripOk; for (int k = 0; k < 4; k++) F.vC[k] = c->r2y.vC[k];
            if (pf32_to_yuv420_strip_takes(F)) {
                c->lastKernel = "pf32_to_yuv420s_kernel";
                r = launch_pf32_to_yuv420s(F, c->stream, nullptr, 1);
                break;
            }
        }
        if (!c->inter) {
            c->interStride = align_up(c->srcW * 3, 256);
            GMAT_HIP_CHECK(hipMalloc((void **)&c->inter, (size_t)c->interStride * c->srcH));
        }
        if ((r = launch_rgbpf32_to_rgb24(src[0], srcStride[0], c->inter, c->interStride, c->srcW, c->srcH, 0, c->stream)) < 0) break;
        Rgb2YuvLaunch L;
        L.src = c->inter; L.ss = c->interStride; L.bgr = 0;
        L.y = dst[0]; L.ys = dstStride[0]; L.u = dst[1]; L.us = dstStride[1];
        L.v = dnv12 ? nullptr : dst[2]; L.vs = dnv12 ? 0 : dstStride[2]; L.nv12 = dnv12;
        L.w = c->srcW; L.h = c->srcH;
        L.vChr = c->r2yVChr;
        L.rowStart = (const int32_t *)c->dR2YrowStart.p; L.rowCount = (const int32_t *)c->dR2YrowCount.p;
        L.maxRows = c->r2y.maxRows;
        L.k = make_rgb2yuv_consts(c->colorspace);      // the destination's matrix (fill_rgb2yuv_table, utils.c:765-858)
        L.stripOk = c->r2y.stripOk; for (int k = 0; k < 4; k++) L.vC[k] = c->r2y.vC[k];
        c->lastKernel = rgb2yuv420_strip_takes(L) ? "rgbpf32_to_rgb24_kernel+rgb2yuv420s_kernel" : "rgbpf32_to_rgb24_kernel+rgb2yuv420_kernel";
        r = launch_rgb2yuv420(L, c->stream);
        break;
    }
    case MODE_DEPTH: {
        if (!dst[1]) { r = GMAT_ERR(EINVAL); break; }
        c->lastKernel = "widen8to16_kernel";
        if ((r = launch_widen8to16(src[0], srcStride[0], nullptr, 0, dst[0], dstStride[0], c->srcW, c->srcH, c->stream)) < 0) break;
        // the CPU wrapper converts srcW / 2 chroma pairs on ceil(srcH / 2) rows (:309-317); planar sources only (see gmat_sws_getContext)
        const int cw = c->srcW / 2, ch = ceil_rshift(c->srcH, 1);
        r = launch_widen8to16(src[1], srcStride[1], src[2], srcStride[2], dst[1], dstStride[1], cw, ch, c->stream);
        break;
    }
    case MODE_SCALE: {
        if ((r = ensure_scaler(c)) < 0) break;
        if ((is_plane_src(c->srcFormat) || c->rgbViaPlanes) && c->fused == 2) {
            YuvScaleArgs ya;
            if ((r = prep_yuv_args(c, src, srcStride, dst, dstStride, ya)) < 0) break;
            if (cross_layout_cascade(c, ya, 1)) {
                // NV12 <-> YUV420P scaled: the sibling context in the source's layout, then the re-layout of its (destination-size) frame
                if ((r = cross_prepare(c, 1)) < 0) break;
                c->interTouched = true;
                uint8_t *pl[4]; int st[4];
                cross_planes(c, c->crossBuf, pl, st);
                gmat_sws_setStream(c->cross, (void *)c->stream);
                if ((r = gmat_sws_scale(c->cross, src, srcStride, 0, c->srcH, pl, st)) < 0) break;
                const bool snv = c->srcFormat == GMAT_PIX_FMT_NV12;
                r = launch_yuv420_relayout(snv, pl[0], st[0], pl[1], st[1], pl[2], st[2], dst[0], dstStride[0], dst[1], dstStride[1],
                                           snv ? dst[2] : nullptr, snv ? dstStride[2] : 0, c->dstW, c->dstH, c->stream);
                c->lastKernel = c->cross->lastKernel;
                break;
            }
            // one frame per call: the first kernel of the table that takes this frame (kPlaneKernels)
            Yuv2xFrames one;
            std::memset(&one, 0, sizeof(one));
            one.y[0] = ya.y; one.u[0] = ya.u; one.v[0] = ya.v; one.dst[0] = ya.dst; one.dstU[0] = ya.dstU; one.dstV[0] = ya.dstV;
            int pick = 0;
            while (!kPlaneKernels[pick].eligible(c, ya, 1)) pick++;
            c->lastKernel = kPlaneKernels[pick].name(c, ya, 1);
            if (c->px4 && std::strcmp(c->lastKernel, "scale_yuvg_rgb2p_blk_kernel")) { r = GMAT_ERR(EINVAL); break; }      // (four-byte pixels were promised the kernel that reads them)
            r = kPlaneKernels[pick].launch(c, ya, c->stream, one, 1);
            break;
        }
        ScaleArgs a = c->args;
        a.dst = dst[0]; a.ds = dstStride[0];
        const int bpp = bytes_per_pixel(c->dstFormat);
        a.dstAligned = bpp == 4 ? ((((uintptr_t)dst[0] | (uintptr_t)dstStride[0]) & 15) == 0) : al4(dst[0], dstStride[0]);
        if (is_yuv420(c->srcFormat) && !c->fused) {
            // two kernels + HBM intermediate, the reference's structure (swscale_cuda.c:352-371)
            if (!c->inter) {
                c->interStride = align_up(c->srcW * 3, 256);
                GMAT_HIP_CHECK(hipMalloc((void **)&c->inter, (size_t)c->interStride * c->srcH));
            }
            r = launch_yuv2rgb(yuv_src_of(c->srcFormat, src, srcStride), c->inter, c->interStride, c->srcW, c->srcH,
                               GMAT_PIX_FMT_RGB24, c->y2r, c->stream);
            if (r < 0) break;
            a.srcKind = 0; a.srcBgr = 0;
            a.src0 = c->inter; a.ss0 = c->interStride;
            a.srcAligned = 1;
        } else if (is_yuv420(c->srcFormat)) {
            a.srcKind = 1;
            a.srcNv12 = c->srcFormat == GMAT_PIX_FMT_NV12;
            a.src0 = src[0]; a.ss0 = srcStride[0];
            a.src1 = src[1]; a.ss1 = srcStride[1];
            a.src2 = planarYuv ? src[2] : nullptr; a.ss2 = planarYuv ? srcStride[2] : 0;
            a.srcAligned = al4(src[0], srcStride[0]) &&
                           (a.srcNv12 ? al4(src[1], srcStride[1])
                                      : ((((uintptr_t)src[1] | (uintptr_t)src[2] | (uintptr_t)srcStride[1] |
                                           (uintptr_t)srcStride[2]) & 1) == 0));
            a.y2r = c->args.y2r;
        } else {
            a.srcKind = 0;
            a.srcBgr = c->srcFormat == GMAT_PIX_FMT_BGR24;
            a.src0 = src[0]; a.ss0 = srcStride[0];
            a.srcAligned = al4(src[0], srcStride[0]);
        }
        if (a.srcKind == 1) {
            // front conversion constants ride in a.y2r's closed-form fields; the output stage fields
            // (y_coeff .. u2b) stay those of the scaler's BT.601 model
            Yuv2RgbConsts k = c->y2r;
            k.y_coeff = c->args.y2r.y_coeff; k.y_offset = c->args.y2r.y_offset;
            k.v2r = c->args.y2r.v2r; k.v2g = c->args.y2r.v2g; k.u2g = c->args.y2r.u2g; k.u2b = c->args.y2r.u2b;
            a.y2r = k;
        }
        if (a.srcKind == 1 && !c->prof && rgb2h_yuv_eligible(c, src, srcStride, dst[0], dstStride[0])) {
            // the fused convert-then-scale form at exactly 2:1 on the strip kernel
            const Rgb2sArgs ra = make_rgb2h_yuv_args(c, srcStride, dstStride[0]);
            Yuv2xFrames one;
            std::memset(&one, 0, sizeof(one));
            one.y[0] = src[0]; one.u[0] = src[1]; one.v[0] = planarYuv ? src[2] : nullptr; one.dst[0] = dst[0];
            c->lastKernel = "scale_rgb2h_kernel<yuv>";
            r = launch_scale_rgb2s(ra, c->stream, &one, 1);
            break;
        }
        if (a.srcKind == 0 && c->r2s.ok && a.srcAligned && a.dstAligned && !c->px4) {
            // exact 2:1 from packed RGB (also the second kernel of the two-kernel form): the strip-walking scaler
            const Rgb2sArgs ra = make_rgb2s_args(c, a.ss0, a.ds, a.srcBgr != 0);
            Yuv2xFrames one;
            std::memset(&one, 0, sizeof(one));
            one.y[0] = a.src0; one.dst[0] = a.dst;
            c->lastKernel = rgb2s_kernel_name();
            r = launch_scale_rgb2s(ra, c->stream, &one, 1);
            break;
        }
        const char *rgw = GMAT_KNOB("GMAT_RGBSRC_WALKER");
        if (a.srcKind == 0 && c->rg.ok && a.srcAligned && al4(dst[0], dstStride[0]) && !c->prof && !(rgw && !atoi(rgw)) &&
            (yuvg_rgbsrc_block_form(c->rgargs, 1) || (c->rgargs.K && rgw && atoi(rgw) == 2))) {
            // away from the strip kernel's exact 2 : 1 (round 5): the block-cooperative form wherever it has an instance — every launch size, up-scales of any factor,
            // RGBA sources read as they are (c->px4) — and the walker's form (one column a lane, running sums) behind it, alone in a launch only with
            // GMAT_RGBSRC_WALKER=2 (tests, A/B: 17.2 against the tiled kernel's 13.0 us for a 1080p -> 720p frame; profiles/r05s_rgbrgb_walker.txt)
            Yuv2xFrames one;
            std::memset(&one, 0, sizeof(one));
            one.y[0] = a.src0; one.dst[0] = a.dst;
            c->lastKernel = yuvg_rgbsrc_block_form(c->rgargs, 1) ? "scale_yuvg_rgbsrc_blk_kernel" : "scale_yuvg_rgbsrc_kernel";
            r = launch_scale_yuvg_rgbsrc(make_rg_args(c, a.ss0, a.ds), c->stream, &one, 1);
            break;
        }
        if (c->px4) { r = GMAT_ERR(EINVAL); break; }              // (four-byte pixels were promised a kernel that reads them)
        c->lastKernel = scale_kernel_name(a, c->tiling);
        r = launch_scale_rgb(a, c->tiling, c->stream);
        break;
    }
    }
    return r < 0 ? r : c->dstH;
}

// ---- plain-pointer back-end entry points under the reference's names --------------------------
static int stateless_convert(const uint8_t *src[], int srcStride[], uint8_t *dst[], int dstStride[], int w, int h,
                             int srcFormat, int dstFormat, void *stream);

// libswscale/cuda/yuv2rgb_cuda.cu:862-907: NV12 -> RGB24 / BGR24 / RGBA / BGRA / RGBA64 / BGRA64 / RGBPF32LE, YUV420P -> the six packed ones
int yuv2rgb_cuda(const uint8_t *src[], int srcStride[], uint8_t *dst[], int dstStride[], int w, int h,
                 int srcFormat, int dstFormat, void *stream)
{
    if (!src || !dst || !srcStride || !dstStride || !is_yuv420(srcFormat)) return GMAT_ERR(EINVAL);
    if (is_rgb64(dstFormat))                       // the 19-bit lines of a context (yuv2rgba64_*_c): the context API's path
        return stateless_convert(src, srcStride, dst, dstStride, w, h, srcFormat, dstFormat, stream);
    const Yuv2RgbConsts k = make_yuv2rgb_consts(GMAT_SWS_CS_DEFAULT, false);
    if (dstFormat == GMAT_PIX_FMT_RGBPF32LE)
        return launch_nv12_to_rgbpf32(yuv_src_of(srcFormat, src, srcStride), dst[0], dstStride[0], w, h, k,
                                      (hipStream_t)stream);
    return launch_yuv2rgb(yuv_src_of(srcFormat, src, srcStride), dst[0], dstStride[0], w, h, dstFormat, k,
                          (hipStream_t)stream);
}

// The reference's entry points are stateless; the table set-up they imply is cached per geometry.
static int stateless_convert(const uint8_t *src[], int srcStride[], uint8_t *dst[], int dstStride[], int w, int h,
                             int srcFormat, int dstFormat, void *stream)
{
    // keyed by geometry AND device (the context's tables live on the device current at creation).  The global lock
    // covers look-up and insertion only; the launch holds the entry's own lock (a context is not re-entrant), so callers
    // on other devices or geometries do not wait for it.  An evicted entry dies with its last user.
    struct Entry {
        int w, h, s, d, dev; GmatSwsContext *c; std::mutex use;
        ~Entry() { if (c) gmat_sws_freeContext(c); }
    };
    static std::mutex lock;
    static std::vector<std::shared_ptr<Entry>> cache;
    if (!src || !dst || !srcStride || !dstStride) return GMAT_ERR(EINVAL);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::shared_ptr<Entry> e;
    {
        std::lock_guard<std::mutex> g(lock);
        for (const auto &k : cache)
            if (k->w == w && k->h == h && k->s == srcFormat && k->d == dstFormat && k->dev == dev) e = k;
        if (!e) {
            GmatSwsContext *c = gmat_sws_getContext(w, h, srcFormat, w, h, dstFormat, GMAT_SWS_HWACCEL, nullptr);
            if (!c) return GMAT_ERR(ENOSYS);
            e = std::make_shared<Entry>();
            e->w = w; e->h = h; e->s = srcFormat; e->d = dstFormat; e->dev = dev; e->c = c;
            if (cache.size() >= 32) cache.erase(cache.begin());
            cache.push_back(e);
        }
    }
    std::lock_guard<std::mutex> u(e->use);
    gmat_sws_setStream(e->c, stream);
    const int r = gmat_sws_scale(e->c, src, srcStride, 0, h, dst, dstStride);
    return r < 0 ? r : 0;
}

// libswscale/cuda/yuv2rgb_cuda.cu:909-947: RGB24 / BGR24 / RGBA / BGRA / RGBA64 / BGRA64 -> NV12 / YUV420P
int rgb2yuv_cuda(const uint8_t *src[], int srcStride[], uint8_t *dst[], int dstStride[], int w, int h, int srcFormat,
                 int dstFormat, void *stream)
{
    const bool srcOk = srcFormat == GMAT_PIX_FMT_RGB24 || srcFormat == GMAT_PIX_FMT_BGR24 || srcFormat == GMAT_PIX_FMT_RGBA ||
                       srcFormat == GMAT_PIX_FMT_BGRA || is_rgb64(srcFormat);
    if (!srcOk || !is_yuv420(dstFormat)) return GMAT_ERR(ENOSYS);
    return stateless_convert(src, srcStride, dst, dstStride, w, h, srcFormat, dstFormat, stream);
}

// libswscale/cuda/yuv2yuv_cuda.cu:324-366: equal formats = a copy of every plane; NV12 / YUV420P -> the other layout, P010, P016,
// YUV420P10, YUV420P16.  (The reference's copy moves `width` bytes of `height` rows of every plane whatever the format — short for
// 16-bit samples, tall for chroma planes; here every plane is copied at its own size.)
int yuv2yuv_cuda(const uint8_t *src[], int srcStride[], uint8_t *dst[], int dstStride[], int w, int h, int srcFormat,
                 int dstFormat, void *stream)
{
    const bool yuvAny = is_yuv420(srcFormat) || is_p01x(srcFormat) || srcFormat == GMAT_PIX_FMT_YUV420P10LE || srcFormat == GMAT_PIX_FMT_YUV420P16LE;
    const bool dstOk = is_yuv420(dstFormat) || is_p01x(dstFormat) || dstFormat == GMAT_PIX_FMT_YUV420P10LE || dstFormat == GMAT_PIX_FMT_YUV420P16LE;
    // The reference's switch (yuv2yuv_cuda.cu:324-366) has arms for equal formats and for NV12 / YUV420P sources only — a P010 / P016 / 10- or 16-bit planar source into
    // another format of the list falls through it and writes NOTHING (found with the reference's own core, round 6: sws_scale() of a same-size P010 -> NV12
    // SWS_HWACCEL_CUDA context leaves the destination untouched).  This symbol serves every pair of the list — what the CPU context of the same libswscale computes
    // (the generic lines with one-tap filters: scale19_unit_kernel) — so that the core's convert_unscaled path has no silent hole here.
    if (!(yuvAny && dstOk)) return GMAT_ERR(ENOSYS);
    return stateless_convert(src, srcStride, dst, dstStride, w, h, srcFormat, dstFormat, stream);
}

void rgb24tobgr24_cuda(const uint8_t *src[], uint8_t *dst[], int srcStride[], int dstStride[], int width, int height,
                       void *stream)
{
    if (!src || !dst) return;
    (void)launch_swap_rb24(src[0], srcStride[0], dst[0], dstStride[0], width, height, (hipStream_t)stream);
}

void rgb2rgb_init_cuda(void) {}

// ---- metrans/app/CSwscale.c:9-40 ------------------------------------------------------------------
GmatSwsContext *SwscaleCuda_Nv12ToRgbpf32_Init(int w, int h)
{
    return gmat_sws_getContext(w, h, GMAT_PIX_FMT_NV12, w, h, GMAT_PIX_FMT_RGBPF32LE, GMAT_SWS_HWACCEL, nullptr);
}

int SwscaleCuda_Nv12ToRgbpf32_Convert(GmatSwsContext *c, uint8_t *src, int srcStride, uint8_t *dst, int dstStride,
                                      int w, int h, void *stream)
{
    // av_image_fill_linesizes / av_image_fill_pointers for tightly packed NV12 and planar float RGB
    // (CSwscale.c:25-28): linesize = {w, w} and {4w, 4w, 4w}; the srcStride/dstStride arguments are
    // accepted and ignored exactly as the reference does.
    (void)srcStride; (void)dstStride;
    if (!c || !src || !dst) return GMAT_ERR(EINVAL);
    const uint8_t *s[4] = {src, src + (size_t)w * h, nullptr, nullptr};
    const int ss[4] = {w, w, 0, 0};
    uint8_t *d[4] = {dst, dst + (size_t)4 * w * h, dst + (size_t)8 * w * h, nullptr};
    const int dd[4] = {4 * w, 4 * w, 4 * w, 0};
    gmat_sws_setStream(c, stream);
    return gmat_sws_scale(c, s, ss, 0, h, d, dd);
}

void SwscaleCuda_Nv12ToRgbpf32_Delete(GmatSwsContext *c) { gmat_sws_freeContext(c); }

} // extern "C"
