// gfilter.cpp — AVFilter-shaped GPU filters (include/gmat_hip.h §3).
//
// Each filter keeps the reference's four callbacks and option names (paths relative to
// /root/reference/ffmpeg-gpu/libavfilter):
//   crop_hip      vf_crop_nvcv.c    init :113-122  config_props :133-207  filter_frame :209-291
//   flip_hip      vf_flip_nvcv.c    options :77-80 (code 0 vertical, 1 horizontal, -1 both)
//   rotate_hip    vf_rotate_nvcv.c  options :79-88
//   smooth_hip    vf_smooth_nvcv.c  options :82-105
//   transpose_hip vf_transpose.c    dir names :374-379
//   scale_hip     vf_scale_cuda.c   options :586-603, on top of libgpuscale (gsws.cpp)
//   format_hip    vf_format_cuda.c  option pix_fmt :69-79
// filter_frame takes ownership of `in` and frees it on every path; errors are negative codes and
// nothing continues after a failed launch (the reference's CK_NVCV only logs, vf_crop_nvcv.c:62-77).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <vector>
#include <new>
#include <string>
#include "common.h"
#include "kernels.h"

using namespace gmat;

namespace {

enum Kind { K_CROP, K_FLIP, K_ROTATE, K_TRANSPOSE, K_SMOOTH, K_SCALE, K_FORMAT };

int parse_pix_fmt(const std::string &s)
{
    static const std::map<std::string, int> m = {
        {"rgb24", GMAT_PIX_FMT_RGB24}, {"bgr24", GMAT_PIX_FMT_BGR24}, {"rgba", GMAT_PIX_FMT_RGBA},
        {"bgra", GMAT_PIX_FMT_BGRA}, {"nv12", GMAT_PIX_FMT_NV12}, {"yuv420p", GMAT_PIX_FMT_YUV420P},
        {"yuv444p", GMAT_PIX_FMT_YUV444P}, {"p010le", GMAT_PIX_FMT_P010LE}, {"p016le", GMAT_PIX_FMT_P016LE},
        {"rgba64le", GMAT_PIX_FMT_RGBA64LE}, {"bgra64le", GMAT_PIX_FMT_BGRA64LE},
        {"yuv444p16le", GMAT_PIX_FMT_YUV444P16LE}, {"yuv420p16le", GMAT_PIX_FMT_YUV420P16LE}, {"yuv420p10le", GMAT_PIX_FMT_YUV420P10LE}, {"rgb0", GMAT_PIX_FMT_RGB0}, {"bgr0", GMAT_PIX_FMT_BGR0}, {"0bgr32", GMAT_PIX_FMT_RGB0}, {"0rgb32", GMAT_PIX_FMT_BGR0},
        {"rgbpf32le", GMAT_PIX_FMT_RGBPF32LE}, {"same", GMAT_PIX_FMT_NONE}};
    auto it = m.find(s);
    if (it != m.end()) return it->second;
    char *end = nullptr;
    long v = strtol(s.c_str(), &end, 10);
    return end && *end == 0 ? (int)v : -2;
}

// Per-plane geometry of a frame.  Packed RGB: one plane of bpp-byte pixels.  Planar / semi-planar 4:2:0
// (an extension: the reference lists NV12 / YUV420P as intended but disabled, vf_crop_nvcv.c:91-94): the
// geometric filters act on each plane like the CPU filters do (vf_hflip.c:89-117 per plane with the chroma
// shifts; vf_transpose.c:267-327; vf_crop.c:300-304), the NV12 chroma plane as 2-byte samples.
struct PlaneGeom { int bpp, w, h, sub; };

int plane_geoms(int fmt, int w, int h, PlaneGeom g[3])
{
    if (fmt == GMAT_PIX_FMT_YUV420P) {
        g[0] = {1, w, h, 0}; g[1] = g[2] = {1, (w + 1) >> 1, (h + 1) >> 1, 1};
        return 3;
    }
    if (fmt == GMAT_PIX_FMT_NV12) {
        g[0] = {1, w, h, 0}; g[1] = {2, (w + 1) >> 1, (h + 1) >> 1, 1};
        return 2;
    }
    if (fmt == GMAT_PIX_FMT_YUV444P) {
        g[0] = g[1] = g[2] = {1, w, h, 0};
        return 3;
    }
    g[0] = {bytes_per_pixel(fmt), w, h, 0};
    return 1;
}


// ---- w / h expressions of scale_cuda (vf_scale_cuda.c:586-587, evaluated by scale_eval.c:57-111 through libavutil's
// expression parser).  The subset an output size needs: numbers, + - * / ( ), unary minus, the variables of
// scale_eval.c:27-41 and min / max / trunc / floor / ceil / round / abs.  Anything else is an error, never a default.
struct ExprVars { double iw, ih, ow, oh, a, sar, dar, hsub, vsub, ohsub, ovsub; };

struct ExprParser {
    const char *p;
    const ExprVars &v;
    bool ok = true;
    void skip() { while (*p == ' ' || *p == '\t') p++; }
    bool eat(char c) { skip(); if (*p == c) { p++; return true; } return false; }
    double ident()
    {
        const char *b = p;
        while ((*p >= 'a' && *p <= 'z') || (*p >= 'A' && *p <= 'Z') || *p == '_' || (p > b && *p >= '0' && *p <= '9')) p++;
        const std::string n(b, p);
        if (eat('(')) {
            double a0 = sum(), a1 = 0;
            const bool two = eat(',');
            if (two) a1 = sum();
            if (!eat(')')) ok = false;
            if (n == "min" && two) return std::min(a0, a1);
            if (n == "max" && two) return std::max(a0, a1);
            if (!two) {
                if (n == "trunc") return std::trunc(a0);
                if (n == "floor") return std::floor(a0);
                if (n == "ceil") return std::ceil(a0);
                if (n == "round") return std::round(a0);
                if (n == "abs") return std::fabs(a0);
            }
            ok = false;
            return 0;
        }
        if (n == "in_w" || n == "iw") return v.iw;
        if (n == "in_h" || n == "ih") return v.ih;
        if (n == "out_w" || n == "ow") return v.ow;
        if (n == "out_h" || n == "oh") return v.oh;
        if (n == "a") return v.a;
        if (n == "sar") return v.sar;
        if (n == "dar") return v.dar;
        if (n == "hsub") return v.hsub;
        if (n == "vsub") return v.vsub;
        if (n == "ohsub") return v.ohsub;
        if (n == "ovsub") return v.ovsub;
        ok = false;
        return 0;
    }
    double atom()
    {
        skip();
        if (eat('(')) { const double r = sum(); if (!eat(')')) ok = false; return r; }
        if (eat('-')) return -atom();
        if (eat('+')) return atom();
        if ((*p >= '0' && *p <= '9') || *p == '.') { char *e = nullptr; const double r = strtod(p, &e); if (e == p) ok = false; p = e; return r; }
        if ((*p >= 'a' && *p <= 'z') || (*p >= 'A' && *p <= 'Z') || *p == '_') return ident();
        ok = false;
        return 0;
    }
    double term()
    {
        double r = atom();
        for (;;) {
            if (eat('*')) r *= atom();
            else if (eat('/')) r /= atom();
            else return r;
        }
    }
    double sum()
    {
        double r = term();
        for (;;) {
            if (eat('+')) r += term();
            else if (eat('-')) r -= term();
            else return r;
        }
    }
};

// av_expr_parse_and_eval for that subset: false when the text does not parse completely
bool eval_expr(const std::string &text, const ExprVars &v, double &res)
{
    ExprParser ps{text.c_str(), v};
    res = ps.sum();
    ps.skip();
    return ps.ok && *ps.p == 0 && !text.empty();
}

// av_rescale(a, b, c): a * b / c rounded to nearest, halves away from zero (mathematics.c AV_ROUND_NEAR_INF)
int64_t rescale_near(int64_t a, int64_t b, int64_t c)
{
    if (c <= 0) return 0;
    const int64_t n = a * b;
    return n >= 0 ? (n + c / 2) / c : -((-n + c / 2) / c);
}

int chroma_log2(int fmt)      // log2_chroma_w == log2_chroma_h for every format handled here
{
    return (fmt == GMAT_PIX_FMT_NV12 || fmt == GMAT_PIX_FMT_YUV420P || fmt == GMAT_PIX_FMT_P010LE || fmt == GMAT_PIX_FMT_P016LE) ? 1 : 0;
}

} // namespace

struct GmatFilterContext {
    Kind kind;
    std::string name;
    std::map<std::string, std::string> opt;
    bool inited = false, configured = false;
    hipStream_t stream = nullptr;
    int in_w = 0, in_h = 0, in_fmt = GMAT_PIX_FMT_NONE, device = 0;
    int out_w = 0, out_h = 0, out_fmt = GMAT_PIX_FMT_NONE;
    GmatHWFramesContext *out_frames = nullptr;
    // crop
    int x = -1, y = -1, w = 0, h = 0;
    // flip
    int code = 0;
    // rotate / transpose
    double angle = 0; int dir = 1, quarter = 0;       // quarter == -1: arbitrary angle (vf_rotate.c arithmetic)
    int rot_bilinear = 1;                 // 0 nearest, 1 linear (and area), 2 cubic
    double rot_shift_x = 0, rot_shift_y = 0;
    // smooth
    int smooth_median = 0, kw = 3, kh = 3;
    int border = -1;                     // -1: not given (the 3x3 integer kernel keeps vf_convolution's borders)
    double sigmaX = 0, sigmaY = 0;
    bool gauss_general = false;          // any of kw / kh / sigma / border_type given: the float kernel of launch_gauss_blur
    // scale / format
    GmatSwsContext *sws = nullptr;
    int cur_cs = GMAT_SWS_CS_DEFAULT;            // the colourspace row the sws context was last set to (follow_frame_colour)
    int sws_flags = GMAT_SWS_BICUBIC;
    int passthrough = 1, force_oar = 0, force_div = 1;
    bool bypass = false;                 // passthrough && nothing to do: filter_frame hands the input frame on
    // queued form (gmat_filter_send_frame / receive_frame / flush)
    int batch = 1;
    std::vector<GmatFrame *> pending;
    std::deque<GmatFrame *> ready;
    double sws_param[2] = {GMAT_SWS_PARAM_DEFAULT, GMAT_SWS_PARAM_DEFAULT};
};

static int opt_int(GmatFilterContext *f, const char *k, int dflt)
{
    auto it = f->opt.find(k);
    return it == f->opt.end() ? dflt : atoi(it->second.c_str());
}

static double opt_double(GmatFilterContext *f, const char *k, double dflt)
{
    auto it = f->opt.find(k);
    return it == f->opt.end() ? dflt : atof(it->second.c_str());
}

extern "C" {

GmatFilterContext *gmat_filter_alloc(const char *name)
{
    static const std::map<std::string, Kind> kinds = {
        {"crop_hip", K_CROP}, {"flip_hip", K_FLIP}, {"rotate_hip", K_ROTATE}, {"transpose_hip", K_TRANSPOSE},
        {"smooth_hip", K_SMOOTH}, {"scale_hip", K_SCALE}, {"format_hip", K_FORMAT}};
    if (!name) return nullptr;
    auto it = kinds.find(name);
    if (it == kinds.end()) {
        logf(LOG_ERROR, "gmat_filter_alloc: no such filter '%s'", name);
        return nullptr;
    }
    GmatFilterContext *f = new (std::nothrow) GmatFilterContext();
    if (!f) return nullptr;
    f->kind = it->second;
    f->name = name;
    return f;
}

int gmat_filter_set_option(GmatFilterContext *f, const char *key, const char *value)
{
    if (!f || !key || !value) return GMAT_ERR(EINVAL);
    static const std::map<Kind, std::string> allowed = {
        {K_CROP, " w h x y "}, {K_FLIP, " code batch "}, {K_ROTATE, " angle interp shift_x shift_y batch "},
        {K_TRANSPOSE, " dir batch "}, {K_SMOOTH, " type kw kh border_type sigmaX sigmaY batch "},
        {K_SCALE, " w h interp_algo format passthrough param force_original_aspect_ratio force_divisible_by batch "},
        {K_FORMAT, " pix_fmt batch "}};
    const std::string needle = std::string(" ") + key + " ";
    if (allowed.at(f->kind).find(needle) == std::string::npos) {
        logf(LOG_ERROR, "%s: option '%s' not found", f->name.c_str(), key);
        return GMAT_ERR(ENOENT);      // AVERROR_OPTION_NOT_FOUND analogue
    }
    f->opt[key] = value;
    return 0;
}

int gmat_filter_init(GmatFilterContext *f)
{
    knobs_refresh();                             // the environment knobs are read now, not per frame (common.h)
    if (!f) return GMAT_ERR(EINVAL);
    switch (f->kind) {
    case K_CROP:
        f->w = opt_int(f, "w", 0); f->h = opt_int(f, "h", 0);
        f->x = opt_int(f, "x", -1); f->y = opt_int(f, "y", -1);
        if (f->w <= 0 || f->h <= 0) {
            logf(LOG_ERROR, "crop_hip: The width and height of the cropping area cannot be 0");
            return GMAT_ERR(EINVAL);
        }
        break;
    case K_FLIP:
        f->code = opt_int(f, "code", 0);
        if (f->code < -1 || f->code > 1) return GMAT_ERR(EINVAL);
        break;
    case K_ROTATE: {
        auto it = f->opt.find("angle");
        f->angle = it == f->opt.end() ? 0.0 : atof(it->second.c_str());
        if (f->angle < -360 || f->angle > 360) return GMAT_ERR(EINVAL);
        const double q = f->angle / 90.0;
        {
            auto ip = f->opt.find("interp");
            const std::string m = ip == f->opt.end() ? "linear" : ip->second;
            // map_interpolation, vf_rotate_nvcv.c:114-135.  cubic: Catmull-Rom in integers; area: linear, as cv::warpAffine
            // treats INTER_AREA (the rules are stated in the test suite's checker (orc_vf.c): CV-CUDA's own arithmetic is not in the reference tree)
            if (m == "linear" || m == "area") f->rot_bilinear = 1;
            else if (m == "nearest") f->rot_bilinear = 0;
            else if (m == "cubic") f->rot_bilinear = 2;
            else {
                logf(LOG_ERROR, "rotate_hip: Interpolation '%s' not supported.", m.c_str());
                return GMAT_ERR(EINVAL);
            }
            // shift_x / shift_y (vf_rotate_nvcv.c:85-86,276): the reference's meaning — rotation about the origin, the shift re-centres
            // (gmat_rotate_shift_translation); unset, the rotation is about the centre (defect 11)
            f->rot_shift_x = opt_double(f, "shift_x", 0.0); f->rot_shift_y = opt_double(f, "shift_y", 0.0);
            if (!std::isfinite(f->rot_shift_x) || !std::isfinite(f->rot_shift_y) || std::fabs(f->rot_shift_x) > 32767 || std::fabs(f->rot_shift_y) > 32767)
                return GMAT_ERR(EINVAL);
        }
        if (std::fabs(q - std::round(q)) > 1e-9 || f->rot_shift_x != 0 || f->rot_shift_y != 0) f->quarter = -1;   // vf_rotate.c's fixed-point walk
        else f->quarter = (((int)std::lround(q)) % 4 + 4) % 4;      // exact clockwise quarter turns
        break;
    }
    case K_TRANSPOSE: {
        auto it = f->opt.find("dir");
        std::string d = it == f->opt.end() ? "0" : it->second;
        static const std::map<std::string, int> names = {{"cclock_flip", 0}, {"clock", 1}, {"cclock", 2}, {"clock_flip", 3}};
        auto n = names.find(d);
        f->dir = n != names.end() ? n->second : atoi(d.c_str());
        if (f->dir < 0 || f->dir > 3) return GMAT_ERR(EINVAL);
        break;
    }
    case K_SMOOTH: {
        auto it = f->opt.find("type");
        const std::string t = it == f->opt.end() ? "gaussian" : it->second;
        if (!(t == "gaussian" || t == "median" || t == "0" || t == "1" || t == "2")) return GMAT_ERR(EINVAL);
        f->smooth_median = (t == "median" || t == "2");          // 0 default = 1 gaussian, 2 median (vf_smooth_nvcv.c:76-80)
        f->kw = opt_int(f, "kw", 3); f->kh = opt_int(f, "kh", 3);
        f->sigmaX = opt_double(f, "sigmaX", 0.0); f->sigmaY = opt_double(f, "sigmaY", 0.0);
        if (f->kw < 1 || f->kh < 1 || f->sigmaX < 0 || f->sigmaY < 0) return GMAT_ERR(EINVAL);
        f->border = -1;
        {
            auto bt = f->opt.find("border_type");
            if (bt != f->opt.end()) {
                static const std::map<std::string, int> names = {{"constant", 0}, {"replicate", 1}, {"reflect", 2}, {"warp", 3},
                                                                 {"wrap", 3}, {"reflect101", 4}, {"0", 0}, {"1", 1}, {"2", 2}, {"3", 3}, {"4", 4}};
                auto n = names.find(bt->second);
                if (n == names.end()) return GMAT_ERR(EINVAL);
                f->border = n->second;
            }
        }
        if (f->smooth_median) {
            // median: any odd kw x kh up to kGaussMaxTaps (vf_median.c's rule at radius (kw - 1) / 2, radiusV (kh - 1) / 2);
            // border_type / sigma are "only for gaussian" (vf_smooth_nvcv.c:93,:100-101) and are refused rather than ignored
            if (!(f->kw & 1) || !(f->kh & 1) || f->kw > kGaussMaxTaps || f->kh > kGaussMaxTaps) {
                logf(LOG_ERROR, "smooth_hip: median wants odd kw, kh <= %d", kGaussMaxTaps);
                return GMAT_ERR(EINVAL);
            }
            if (f->border >= 0 || f->sigmaX > 0 || f->sigmaY > 0) {
                logf(LOG_ERROR, "smooth_hip: border_type / sigmaX / sigmaY apply to type=gaussian only");
                return GMAT_ERR(EINVAL);
            }
        } else {
            // the default 3x3 kernel with no sigma and no border rule given is the integer 1-2-1 kernel with
            // vf_convolution's arithmetic and borders; every other request runs the float kernel
            f->gauss_general = f->kw != 3 || f->kh != 3 || f->sigmaX > 0 || f->sigmaY > 0 || f->border >= 0;
            if (f->gauss_general && (!(f->kw & 1) || !(f->kh & 1) || f->kw > kGaussMaxTaps || f->kh > kGaussMaxTaps)) {
                logf(LOG_ERROR, "smooth_hip: gaussian kernels must be odd and at most %d x %d", kGaussMaxTaps, kGaussMaxTaps);
                return GMAT_ERR(f->kw > kGaussMaxTaps || f->kh > kGaussMaxTaps ? ENOSYS : EINVAL);
            }
        }
        break;
    }
    case K_SCALE: {
        auto it = f->opt.find("interp_algo");
        const std::string a = it == f->opt.end() ? "bicubic" : it->second;
        if (a == "nearest" || a == "1") f->sws_flags = GMAT_SWS_POINT;
        else if (a == "bilinear" || a == "2") f->sws_flags = GMAT_SWS_BILINEAR;
        else if (a == "bicubic" || a == "3" || a == "0") f->sws_flags = GMAT_SWS_BICUBIC;
        else if (a == "lanczos" || a == "4") f->sws_flags = GMAT_SWS_LANCZOS;
        else return GMAT_ERR(EINVAL);
        f->passthrough = opt_int(f, "passthrough", 1) != 0;
        {
            auto fo = f->opt.find("force_original_aspect_ratio");
            const std::string m = fo == f->opt.end() ? "disable" : fo->second;
            if (m == "disable" || m == "0") f->force_oar = 0;
            else if (m == "decrease" || m == "1") f->force_oar = 1;
            else if (m == "increase" || m == "2") f->force_oar = 2;
            else return GMAT_ERR(EINVAL);
        }
        f->force_div = opt_int(f, "force_divisible_by", 1);
        if (f->force_div < 1 || f->force_div > 256) return GMAT_ERR(EINVAL);
        // "param": the algorithm-specific parameter (vf_scale_cuda.c:596, default SCALE_CUDA_PARAM_DEFAULT = unset).  The
        // arithmetic here is libswscale's, so it becomes libswscale's param[0] (bicubic B, lanczos lobes, gauss sharpness:
        // utils.c:455-530), exactly what the CPU scale filter's param0 does (vf_scale.c:897)
        if (f->opt.find("param") != f->opt.end()) f->sws_param[0] = opt_double(f, "param", GMAT_SWS_PARAM_DEFAULT);
        break;
    }
    case K_FORMAT:
        if (f->opt.find("pix_fmt") == f->opt.end()) return GMAT_ERR(EINVAL);
        break;
    }
    // "batch" (no reference counterpart): frames the queued entry points collect before they launch ONE kernel for all of them
    // (gmat_filter_send_frame below) — scale_hip, format_hip, and the transform filters whose kernels take a frame table (flip,
    // transpose, rotate by k * 90 degrees, the 3 x 3 smooth and median; the others process a full queue frame by frame)
    if (f->kind != K_CROP) {
        f->batch = opt_int(f, "batch", 1);
        if (f->batch < 1 || f->batch > 256) return GMAT_ERR(EINVAL);
    }
    f->inited = true;
    return 0;
}

GmatHWFramesContext *gmat_filter_out_frames(GmatFilterContext *f) { return f ? f->out_frames : nullptr; }

int gmat_filter_config_props(GmatFilterContext *f, GmatHWFramesContext *in_frames, void *stream)
{
    if (!f || !f->inited || !in_frames) return GMAT_ERR(EINVAL);
    gmat_hwframe_ctx_info(in_frames, &f->device, &f->in_fmt, &f->in_w, &f->in_h);
    // the link's device becomes current BEFORE anything is created for it (cuCtxPushCurrent in every reference filter,
    // e.g. vf_scale_cuda.c:292-294): the scaler's tables and the output pool must live where the frames do
    DeviceScope onDevice;                        // ... and the caller's device is current again on return (cuCtxPopCurrent)
    if (int e = onDevice.enter(f->device); e < 0) return e;
    f->stream = (hipStream_t)stream;
    f->out_w = f->in_w; f->out_h = f->in_h; f->out_fmt = f->in_fmt;
    const bool nvcv_style = f->kind != K_SCALE && f->kind != K_FORMAT;
    if (nvcv_style && !is_packed_rgb(f->in_fmt) && !is_yuv8_src(f->in_fmt)) {
        logf(LOG_ERROR, "%s: Unsupported input format: %d", f->name.c_str(), f->in_fmt);
        return GMAT_ERR(ENOSYS);
    }
    switch (f->kind) {
    case K_CROP:
        if (f->x == -1) f->x = (f->in_w - f->w) / 2;
        if (f->y == -1) f->y = (f->in_h - f->h) / 2;
        if (is_yuv420(f->in_fmt)) {          // chroma-grid alignment, vf_crop.c:186-187,:223-224
            f->w &= ~1; f->h &= ~1; f->x &= ~1; f->y &= ~1;
            if (f->w <= 0 || f->h <= 0) return GMAT_ERR(EINVAL);
        }
        if (f->x < 0 || f->y < 0 || f->w + f->x > f->in_w || f->h + f->y > f->in_h) {
            logf(LOG_ERROR, "crop_hip: The cropping area cannot fall out of the image border");
            return GMAT_ERR(EINVAL);
        }
        f->out_w = f->w; f->out_h = f->h;
        break;
    case K_ROTATE:
        // quarter turns swap the dimensions, like CPU transpose (vf_transpose.c:216-217); the
        // reference's rotate_nvcv keeps w x h and loses pixels (SURVEY.md §0 defect 11)
        if (f->quarter > 0 && (f->quarter & 1)) { f->out_w = f->in_h; f->out_h = f->in_w; }
        break;
    case K_TRANSPOSE:
        f->out_w = f->in_h; f->out_h = f->in_w;
        break;
    case K_SCALE: {
        auto it = f->opt.find("format");
        int fmt = it == f->opt.end() ? GMAT_PIX_FMT_NONE : parse_pix_fmt(it->second);
        if (fmt == -2) return GMAT_ERR(EINVAL);
        f->out_fmt = fmt == GMAT_PIX_FMT_NONE ? f->in_fmt : fmt;      // "same" (vf_scale_cuda.c:594)
        // ff_scale_eval_dimensions (scale_eval.c:57-111): w, then h, then w again (it may refer to oh); 0 -> input size
        const std::string we = f->opt.count("w") ? f->opt["w"] : "iw", he = f->opt.count("h") ? f->opt["h"] : "ih";
        ExprVars v;
        v.iw = f->in_w; v.ih = f->in_h; v.ow = v.oh = NAN;
        v.a = (double)f->in_w / f->in_h; v.sar = 1; v.dar = v.a;              // GmatFrame carries no sample aspect ratio
        // hsub / vsub / ohsub / ovsub: ff_scale_eval_dimensions takes them from av_pix_fmt_desc_get(inlink->format) (scale_eval.c:
        // 76-83), and the format of a hardware link is the hardware pixel format, whose descriptor has no chroma shift — so under
        // scale_cuda they are all 1 whatever the frames' sw_format is, and integration/vf_gmat_hip.c (which calls the real
        // function) sees the same.  Matched here so that one filter string gives one size through both doors.
        v.hsub = v.vsub = v.ohsub = v.ovsub = 1;
        double res = 0;
        (void)eval_expr(we, v, res);                                         // first pass: ow / oh are still unknown
        int w = (int)res == 0 ? f->in_w : (int)res;
        v.ow = w;
        if (!eval_expr(he, v, res) || std::isnan(res)) {
            logf(LOG_ERROR, "scale_hip: Error when evaluating the expression '%s'.", he.c_str());
            return GMAT_ERR(EINVAL);
        }
        int h = (int)res == 0 ? f->in_h : (int)res;
        v.oh = h;
        if (!eval_expr(we, v, res) || std::isnan(res)) {
            logf(LOG_ERROR, "scale_hip: Error when evaluating the expression '%s'.", we.c_str());
            return GMAT_ERR(EINVAL);
        }
        w = (int)res == 0 ? f->in_w : (int)res;
        // ff_scale_adjust_dimensions (scale_eval.c:113-175): -1 keeps the aspect ratio, -n also rounds to a multiple of n
        int fw = 1, fh = 1;
        if (w < -1) fw = -w;
        if (h < -1) fh = -h;
        if (w < 0 && h < 0) { w = f->in_w; h = f->in_h; }
        if (w < 0) w = (int)(rescale_near(h, f->in_w, (int64_t)f->in_h * fw) * fw);
        if (h < 0) h = (int)(rescale_near(w, f->in_h, (int64_t)f->in_w * fh) * fh);
        if (f->force_oar) {
            const int tw = (int)rescale_near(h, f->in_w, f->in_h), th = (int)rescale_near(w, f->in_h, f->in_w);
            if (f->force_oar == 1) {
                w = std::min(tw, w); h = std::min(th, h);
                if (f->force_div > 1) { w = w / f->force_div * f->force_div; h = h / f->force_div * f->force_div; }
            } else {
                w = std::max(tw, w); h = std::max(th, h);
                if (f->force_div > 1) { w = (w + f->force_div - 1) / f->force_div * f->force_div; h = (h + f->force_div - 1) / f->force_div * f->force_div; }
            }
        }
        if (w <= 0 || h <= 0) {
            logf(LOG_ERROR, "scale_hip: invalid output size %dx%d", w, h);
            return GMAT_ERR(EINVAL);
        }
        f->out_w = w; f->out_h = h;
        // vf_scale_cuda.c:254-260,:543: with passthrough (default on) equal size and format means "do not process"
        f->bypass = f->passthrough && w == f->in_w && h == f->in_h && f->out_fmt == f->in_fmt;
        break;
    }
    case K_FORMAT: {
        int fmt = parse_pix_fmt(f->opt["pix_fmt"]);
        if (fmt < 0) return GMAT_ERR(EINVAL);
        f->out_fmt = fmt;
        break;
    }
    default: break;
    }
    if (f->kind == K_SCALE || f->kind == K_FORMAT) {
        if (f->sws) gmat_sws_freeContext(f->sws);
        f->sws = gmat_sws_getContext(f->in_w, f->in_h, f->in_fmt, f->out_w, f->out_h, f->out_fmt,
                                     f->sws_flags | GMAT_SWS_HWACCEL, f->kind == K_SCALE ? f->sws_param : nullptr);
        if (!f->sws) return GMAT_ERR(ENOSYS);
        gmat_sws_setStream(f->sws, stream);
    }
    if (f->out_frames) gmat_hwframe_ctx_free(f->out_frames);
    f->out_frames = gmat_hwframe_ctx_create(f->device, f->out_fmt, f->out_w, f->out_h, 2);
    if (!f->out_frames) return GMAT_ERR(ENOMEM);
    f->configured = true;
    return 0;
}

// A frame's own colour description drives the conversion (vf_format_cuda.c:184-217: in->colorspace goes to SetMatYuv2Rgb / SetMatRgb2Yuv per frame;
// libavfilter's `scale` does the same with in_color_matrix=auto, vf_scale.c:793-824).  GmatFrame carries the colourspace only (no range: limited).
// AVCOL_SPC_* -> the SWS_CS_* row as cuda/yuv2rgb_cuda.cu:782-815 get_constants maps them: unlisted values are BT.601.
static int follow_frame_colour(GmatFilterContext *f, const GmatFrame *in)
{
    int cs = GMAT_SWS_CS_DEFAULT;
    switch (in->colorspace) {
    case 1: cs = GMAT_SWS_CS_ITU709; break;          // AVCOL_SPC_BT709
    case 4: cs = 4; break;                           // AVCOL_SPC_FCC
    case 7: cs = 7; break;                           // AVCOL_SPC_SMPTE240M
    case 9: case 10: cs = GMAT_SWS_CS_BT2020; break; // AVCOL_SPC_BT2020_NCL / _CL
    default: break;
    }
    if (!f->sws || cs == f->cur_cs) return 0;
    const bool src_rgb = is_packed_rgb(f->in_fmt) || is_rgb64(f->in_fmt) || f->in_fmt == GMAT_PIX_FMT_RGBPF32LE;
    const bool dst_rgb = is_packed_rgb(f->out_fmt) || is_rgb64(f->out_fmt) || f->out_fmt == GMAT_PIX_FMT_RGBPF32LE;
    if (src_rgb != dst_rgb) {
        const int r = gmat_sws_setColorspace(f->sws, cs, 0);
        if (r < 0) return r;
    }
    f->cur_cs = cs;
    return 0;
}

int gmat_filter_frame(GmatFilterContext *f, GmatFrame *in, GmatFrame **out_p)
{
    if (!in) return GMAT_ERR(EINVAL);
    int r = GMAT_ERR(EINVAL);
    GmatFrame *out = nullptr;
    DeviceScope onDevice;                                                            // vf_scale_cuda.c:553 cuCtxPushCurrent ... :571 PopCurrent
    if (!f || !f->configured || !out_p) goto fail;
    if (onDevice.enter(f->device) < 0) { r = GMAT_ERR(EIO); goto fail; }
    if (in->format != GMAT_PIX_FMT_HIP || in->sw_format != f->in_fmt || in->width != f->in_w || in->height != f->in_h) {
        logf(LOG_ERROR, "%s: input frame does not match the configured link (%dx%d fmt %d)", f->name.c_str(),
             in->width, in->height, in->sw_format);
        goto fail;
    }
    if (f->kind == K_SCALE && f->bypass) {          // passthrough: the frame itself goes downstream (vf_scale_cuda.c:543-544)
        *out_p = in;
        return 0;
    }
    out = gmat_frame_alloc();
    if (!out) { r = GMAT_ERR(ENOMEM); goto fail; }
    if ((r = gmat_hwframe_get_buffer(f->out_frames, out)) < 0) goto fail;
    if (f->kind == K_SCALE || f->kind == K_FORMAT) {
        if ((r = follow_frame_colour(f, in)) >= 0)
            r = gmat_sws_scale(f->sws, in->data, in->linesize, 0, f->in_h, out->data, out->linesize);
    } else {
        PlaneGeom g[3];
        const int np = plane_geoms(f->in_fmt, f->in_w, f->in_h, g);
        r = 0;
        for (int i = 0; i < np && r >= 0; i++) {
            const int bpp = g[i].bpp, pw = g[i].w, ph = g[i].h, sub = g[i].sub;
            const uint8_t *s = in->data[i];
            const int ss = in->linesize[i], ds = out->linesize[i];
            uint8_t *d = out->data[i];
            switch (f->kind) {
            case K_CROP:
                r = launch_copy2d(s + (size_t)(f->y >> sub) * ss + (size_t)(f->x >> sub) * bpp, ss, d, ds,
                                  ((f->w + sub) >> sub) * bpp, (f->h + sub) >> sub, f->stream);
                break;
            case K_FLIP:
                r = launch_flip(s, ss, d, ds, pw, ph, bpp, f->code != 0, f->code <= 0, f->stream);
                break;
            case K_TRANSPOSE:
                r = launch_transpose(s, ss, d, ds, pw, ph, bpp, f->dir, f->stream);
                break;
            case K_ROTATE:
                switch (f->quarter) {
                case 0: r = launch_copy2d(s, ss, d, ds, pw * bpp, ph, f->stream); break;
                case 1: r = launch_transpose(s, ss, d, ds, pw, ph, bpp, 1, f->stream); break;   // clock
                case 2: r = launch_flip(s, ss, d, ds, pw, ph, bpp, 1, 1, f->stream); break;
                case 3: r = launch_transpose(s, ss, d, ds, pw, ph, bpp, 2, f->stream); break;   // cclock
                default: {
                    // out size = in size, background "black" (vf_rotate.c:102-107; ff_draw_color: RGB 0,0,0 / alpha 255,
                    // limited-range YUV 16,128,128 — drawutils.c:159-202)
                    uint8_t fill[4] = {0, 0, 0, 255};
                    if (is_yuv8_src(f->in_fmt)) { fill[0] = i == 0 ? 16 : 128; fill[1] = 128; }
                    // chroma planes of a 4:2:0 frame move by half the shift (their samples are twice as far apart)
                    double tx = 0, ty = 0;
                    if (f->rot_shift_x != 0 || f->rot_shift_y != 0)
                        gmat_rotate_shift_translation(f->angle * M_PI / 180.0, f->rot_shift_x / (1 << sub), f->rot_shift_y / (1 << sub), pw, ph, pw, ph, &tx, &ty);
                    r = launch_rotate(s, ss, d, ds, pw, ph, pw, ph, bpp, f->angle * M_PI / 180.0, f->rot_bilinear, fill, f->stream, tx, ty);
                    break;
                }
                }
                break;
            case K_SMOOTH: {
                static const int m[9] = {1, 2, 1, 2, 4, 2, 1, 2, 1};
                r = f->smooth_median ? launch_median(s, ss, d, ds, pw, ph, bpp, f->kw, f->kh, f->stream)
                    : f->gauss_general ? launch_gauss_blur(s, ss, d, ds, pw, ph, bpp, f->kw, f->kh, f->sigmaX, f->sigmaY,
                                                           f->border < 0 ? 0 : f->border, f->stream)
                                       : launch_conv3x3(s, ss, d, ds, pw, ph, bpp, m, 1.0f / 16.0f, 0.0f, f->stream);
                break;
            }
            default: break;
            }
        }
    }
    if (r < 0) goto fail;
    // av_frame_copy_props
    out->pts = in->pts;
    out->colorspace = in->colorspace;
    gmat_frame_free(&in);
    *out_p = out;
    return 0;
fail:
    if (out) gmat_frame_free(&out);
    gmat_frame_free(&in);
    return r;
}

// ---- queued form -------------------------------------------------------------------------------------------------
// filter_frame launches one kernel per frame, and a 4K frame is a 4-8 us kernel behind a ~4 us launch: the per-frame
// entry point of the reference (vf_scale_cuda.c:532-573) is launch-bound on this hardware.  A filter may also collect
// frames and emit them later (libavfilter's activate() model: ff_inlink_consume_frame ... ff_filter_frame): with the
// option batch=K, send_frame queues frames and every K-th call launches ONE kernel for K frames (grid.y = frame,
// gmat_sws_scale_batch); receive_frame hands the finished frames out in order; flush launches a partial batch at EOF.
// Filters without a batched kernel, and batch=1, process each frame at once — the three calls then behave like
// filter_frame.
static int op_batch(int op, int n, const uint8_t *const *src, int ss, uint8_t *const *dst, int ds, int w, int h, int bpp, int arg, hipStream_t stream);
static int rotate_batch(int n, const uint8_t *const *src, int ss, uint8_t *const *dst, int ds, int inW, int inH, int outW, int outH, int bpp,
                        double angle_rad, int interp, double shift_x, double shift_y, const uint8_t *fill, hipStream_t stream);

static int run_pending(GmatFilterContext *f)
{
    const int n = (int)f->pending.size();
    if (!n) return 0;
    int r = 0;
    std::vector<GmatFrame *> outs;
    DeviceScope onDevice;
    if (onDevice.enter(f->device) < 0) {
        for (GmatFrame *&p : f->pending) gmat_frame_free(&p);
        f->pending.clear();
        return GMAT_ERR(EIO);
    }
    // flip / transpose / rotate / the 3 x 3 smooth and median: one launch per PLANE for the whole batch
    constexpr int OP_ROTATE_ANY = 100;              // vf_rotate.c's walk at an arbitrary angle (not one of gmat_op_batch's)
    int top = -1, targ = 0;
    if (f->kind == K_FLIP) { top = GMAT_OP_FLIP; targ = f->code; }
    else if (f->kind == K_TRANSPOSE) { top = GMAT_OP_TRANSPOSE; targ = f->dir; }
    else if (f->kind == K_ROTATE && (f->quarter == 1 || f->quarter == 3)) { top = GMAT_OP_TRANSPOSE; targ = f->quarter == 1 ? 1 : 2; }
    else if (f->kind == K_ROTATE && f->quarter == 2) { top = GMAT_OP_FLIP; targ = -1; }
    else if (f->kind == K_ROTATE && f->quarter < 0) top = OP_ROTATE_ANY;
    else if (f->kind == K_SMOOTH && f->smooth_median && f->kw == 3 && f->kh == 3) top = GMAT_OP_MEDIAN3X3;
    else if (f->kind == K_SMOOTH && !f->smooth_median && !f->gauss_general) top = GMAT_OP_SMOOTH3X3;
    if (top >= 0 && n > 1) {
        std::vector<GmatFrame *> outs(n, nullptr);
        int r = 0;                                  // (the device is current: onDevice above)
        for (int i = 0; i < n && r >= 0; i++) {
            GmatFrame *in = f->pending[i];
            if (in->format != GMAT_PIX_FMT_HIP || in->sw_format != f->in_fmt || in->width != f->in_w || in->height != f->in_h) { r = GMAT_ERR(EINVAL); break; }
            outs[i] = gmat_frame_alloc();
            if (!outs[i]) { r = GMAT_ERR(ENOMEM); break; }
            r = gmat_hwframe_get_buffer(f->out_frames, outs[i]);
        }
        PlaneGeom g[3];
        const int np = plane_geoms(f->in_fmt, f->in_w, f->in_h, g);
        bool same = r >= 0;
        for (int i = 1; i < n && same; i++)
            for (int k = 0; k < np; k++)
                same = same && f->pending[i]->linesize[k] == f->pending[0]->linesize[k] && outs[i]->linesize[k] == outs[0]->linesize[k];
        if (r >= 0 && same) {
            std::vector<const uint8_t *> sp(n);
            std::vector<uint8_t *> dp(n);
            for (int k = 0; k < np && r >= 0; k++) {
                for (int i = 0; i < n; i++) { sp[i] = f->pending[i]->data[k]; dp[i] = outs[i]->data[k]; }
                if (top == OP_ROTATE_ANY) {             // background and shift per plane as in gmat_filter_frame
                    uint8_t fill[4] = {0, 0, 0, 255};
                    if (is_yuv8_src(f->in_fmt)) { fill[0] = k == 0 ? 16 : 128; fill[1] = 128; }
                    double tx = 0, ty = 0;
                    if (f->rot_shift_x != 0 || f->rot_shift_y != 0)
                        gmat_rotate_shift_translation(f->angle * M_PI / 180.0, f->rot_shift_x / (1 << g[k].sub), f->rot_shift_y / (1 << g[k].sub),
                                                      g[k].w, g[k].h, g[k].w, g[k].h, &tx, &ty);
                    r = rotate_batch(n, sp.data(), f->pending[0]->linesize[k], dp.data(), outs[0]->linesize[k], g[k].w, g[k].h, g[k].w, g[k].h,
                                     g[k].bpp, f->angle * M_PI / 180.0, f->rot_bilinear, tx, ty, fill, f->stream);
                    continue;
                }
                r = op_batch(top, n, sp.data(), f->pending[0]->linesize[k], dp.data(), outs[0]->linesize[k], g[k].w, g[k].h, g[k].bpp, targ, f->stream);
            }
            if (r >= 0) {
                for (int i = 0; i < n; i++) {
                    outs[i]->pts = f->pending[i]->pts; outs[i]->colorspace = f->pending[i]->colorspace;    // av_frame_copy_props
                    gmat_frame_free(&f->pending[i]);
                    f->ready.push_back(outs[i]);
                }
                f->pending.clear();
                return 0;
            }
        }
        for (GmatFrame *&o : outs) if (o) gmat_frame_free(&o);
        if (r < 0) {
            for (GmatFrame *&p : f->pending) gmat_frame_free(&p);
            f->pending.clear();
            return r;
        }
        // frames of differing strides: one by one below
    }
    bool one_colour = true;                                    // a launch carries one colour description (follow_frame_colour): mixed queues go frame by frame
    for (int i = 1; i < n; i++) one_colour = one_colour && f->pending[i]->colorspace == f->pending[0]->colorspace;
    const bool batched = (f->kind == K_SCALE || f->kind == K_FORMAT) && !f->bypass && n > 1 && one_colour;
    if (batched) {
        std::vector<const uint8_t *> sp((size_t)n * 4, nullptr);
        std::vector<uint8_t *> dp((size_t)n * 4, nullptr);
        for (int i = 0; i < n && r >= 0; i++) {
            GmatFrame *o = gmat_frame_alloc();
            if (!o) { r = GMAT_ERR(ENOMEM); break; }
            outs.push_back(o);
            if ((r = gmat_hwframe_get_buffer(f->out_frames, o)) < 0) break;
            for (int k = 0; k < 4; k++) { sp[(size_t)i * 4 + k] = f->pending[i]->data[k]; dp[(size_t)i * 4 + k] = o->data[k]; }
            // one stride set per launch: pool frames of one context share it, caller-made frames must too
            for (int k = 0; k < 4; k++)
                if (f->pending[i]->linesize[k] != f->pending[0]->linesize[k] || o->linesize[k] != outs[0]->linesize[k]) r = GMAT_ERR(EINVAL);
        }
        if (r >= 0) r = follow_frame_colour(f, f->pending[0]);
        if (r >= 0) {
            void *streams[1] = {(void *)f->stream};
            r = gmat_sws_scale_batch(f->sws, n, sp.data(), f->pending[0]->linesize, dp.data(), outs[0]->linesize, streams, 1, 0);
        }
        if (r >= 0) {
            for (int i = 0; i < n; i++) {
                outs[i]->pts = f->pending[i]->pts; outs[i]->colorspace = f->pending[i]->colorspace;    // av_frame_copy_props
                gmat_frame_free(&f->pending[i]);
                f->ready.push_back(outs[i]);
            }
            f->pending.clear();
            return 0;
        }
        for (GmatFrame *&o : outs) gmat_frame_free(&o);
    } else {
        for (int i = 0; i < n; i++) {
            GmatFrame *o = nullptr;
            const int ri = gmat_filter_frame(f, f->pending[i], &o);     // frees its input on every path
            f->pending[i] = nullptr;
            if (ri < 0) r = ri; else f->ready.push_back(o);
        }
        f->pending.clear();
        return r;
    }
    for (GmatFrame *&p : f->pending) gmat_frame_free(&p);
    f->pending.clear();
    return r;
}

int gmat_filter_send_frame(GmatFilterContext *f, GmatFrame *in)
{
    if (!in) return GMAT_ERR(EINVAL);
    if (!f || !f->configured) { gmat_frame_free(&in); return GMAT_ERR(EINVAL); }
    if (in->format != GMAT_PIX_FMT_HIP || in->sw_format != f->in_fmt || in->width != f->in_w || in->height != f->in_h) {
        logf(LOG_ERROR, "%s: input frame does not match the configured link (%dx%d fmt %d)", f->name.c_str(), in->width, in->height,
             in->sw_format);
        gmat_frame_free(&in);
        return GMAT_ERR(EINVAL);
    }
    f->pending.push_back(in);
    if ((int)f->pending.size() >= f->batch) return run_pending(f);
    return 0;
}

int gmat_filter_receive_frame(GmatFilterContext *f, GmatFrame **out)
{
    if (!f || !out) return GMAT_ERR(EINVAL);
    if (f->ready.empty()) return GMAT_ERR(EAGAIN);
    *out = f->ready.front();
    f->ready.pop_front();
    return 0;
}

int gmat_filter_flush(GmatFilterContext *f)
{
    if (!f || !f->configured) return GMAT_ERR(EINVAL);
    return run_pending(f);
}

void gmat_filter_free(GmatFilterContext *f)
{
    if (!f) return;
    for (GmatFrame *&p : f->pending) gmat_frame_free(&p);
    while (!f->ready.empty()) { GmatFrame *o = f->ready.front(); f->ready.pop_front(); gmat_frame_free(&o); }
    if (f->sws) gmat_sws_freeContext(f->sws);
    if (f->out_frames) gmat_hwframe_ctx_free(f->out_frames);
    delete f;
}

// ---- direct launchers ------------------------------------------------------------------------------
int gmat_transpose(const uint8_t *src, int ss, uint8_t *dst, int ds, int inW, int inH, int bpp, int dir, void *stream)
{
    return launch_transpose(src, ss, dst, ds, inW, inH, bpp, dir, (hipStream_t)stream);
}

int gmat_flip(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int bpp, int code, void *stream)
{
    if (code < -1 || code > 1) return GMAT_ERR(EINVAL);
    return launch_flip(src, ss, dst, ds, w, h, bpp, code != 0, code <= 0, (hipStream_t)stream);
}

int gmat_crop(const uint8_t *src, int ss, uint8_t *dst, int ds, int x, int y, int w, int h, int bpp, void *stream)
{
    if (x < 0 || y < 0) return GMAT_ERR(EINVAL);
    return launch_copy2d(src + (size_t)y * ss + (size_t)x * bpp, ss, dst, ds, w * bpp, h, (hipStream_t)stream);
}

int gmat_smooth3x3(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int bpp, const int matrix[9],
                   float rdiv, float bias, void *stream)
{
    return launch_conv3x3(src, ss, dst, ds, w, h, bpp, matrix, rdiv, bias, (hipStream_t)stream);
}

int gmat_gauss_blur(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int bpp, int kw, int kh, double sigmaX,
                    double sigmaY, int border_type, void *stream)
{
    return launch_gauss_blur(src, ss, dst, ds, w, h, bpp, kw, kh, sigmaX, sigmaY, border_type, (hipStream_t)stream);
}

int gmat_median3x3(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int bpp, void *stream)
{
    return launch_median3x3(src, ss, dst, ds, w, h, bpp, (hipStream_t)stream);
}

void gmat_rotate_shift_translation(double angle_rad, double shift_x, double shift_y, int inW, int inH, int outW, int outH, double *tx, double *ty)
{
    // the walk's matrix (vf_rotate.c:538-548): src = C_in + M (dst - C_out - t), M = [c s; -s c]; src = M (dst - shift) needs
    // t = shift - C_out + M^T C_in
    const double c = std::cos(angle_rad), s = std::sin(angle_rad);
    const double cix = (inW - 1) / 2.0, ciy = (inH - 1) / 2.0, cox = (outW - 1) / 2.0, coy = (outH - 1) / 2.0;
    if (tx) *tx = shift_x - cox + (c * cix - s * ciy);
    if (ty) *ty = shift_y - coy + (s * cix + c * ciy);
}

int gmat_rotate2(const uint8_t *src, int ss, uint8_t *dst, int ds, int inW, int inH, int outW, int outH, int bpp,
                 double angle_rad, int interp, double shift_x, double shift_y, const uint8_t *fill, void *stream)
{
    if (!src || !dst || interp < 0 || interp > 2) return GMAT_ERR(EINVAL);
    return launch_rotate(src, ss, dst, ds, inW, inH, outW, outH, bpp, angle_rad, interp, fill, (hipStream_t)stream, shift_x, shift_y);
}

static int rotate_batch(int n, const uint8_t *const *src, int ss, uint8_t *const *dst, int ds, int inW, int inH, int outW, int outH, int bpp,
                        double angle_rad, int interp, double shift_x, double shift_y, const uint8_t *fill, hipStream_t stream)
{
    for (int i0 = 0; i0 < n; i0 += kOpMaxFrames) {
        const int m = std::min(kOpMaxFrames, n - i0);
        OpFrames fr;
        std::memset(&fr, 0, sizeof(fr));
        for (int i = 0; i < m; i++) {
            if (!src[i0 + i] || !dst[i0 + i]) return GMAT_ERR(EINVAL);
            fr.src[i] = src[i0 + i]; fr.dst[i] = dst[i0 + i];
        }
        const int r = launch_rotate(nullptr, ss, nullptr, ds, inW, inH, outW, outH, bpp, angle_rad, interp, fill, stream, shift_x, shift_y, &fr, m);
        if (r < 0) return r;
    }
    return 0;
}

int gmat_rotate2_batch(int n, const uint8_t *const *src, int ss, uint8_t *const *dst, int ds, int inW, int inH, int outW, int outH, int bpp,
                       double angle_rad, int interp, double shift_x, double shift_y, const uint8_t *fill, void *stream)
{
    if (n < 0 || !src || !dst || interp < 0 || interp > 2) return GMAT_ERR(EINVAL);
    return rotate_batch(n, src, ss, dst, ds, inW, inH, outW, outH, bpp, angle_rad, interp, shift_x, shift_y, fill, (hipStream_t)stream);
}

int gmat_median(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h, int bpp, int kw, int kh, void *stream)
{
    return launch_median(src, ss, dst, ds, w, h, bpp, kw, kh, (hipStream_t)stream);
}

int gmat_rotate(const uint8_t *src, int ss, uint8_t *dst, int ds, int inW, int inH, int outW, int outH, int bpp,
                double angle_rad, int bilinear, const uint8_t *fill, void *stream)
{
    if (!src || !dst) return GMAT_ERR(EINVAL);
    return launch_rotate(src, ss, dst, ds, inW, inH, outW, outH, bpp, angle_rad, bilinear, fill, (hipStream_t)stream);
}

// one transform over n frames: kOpMaxFrames frames a launch
static int op_batch(int op, int n, const uint8_t *const *src, int ss, uint8_t *const *dst, int ds, int w, int h, int bpp, int arg, hipStream_t stream)
{
    static const int m121[9] = {1, 2, 1, 2, 4, 2, 1, 2, 1};
    for (int i0 = 0; i0 < n; i0 += kOpMaxFrames) {
        const int m = std::min(kOpMaxFrames, n - i0);
        OpFrames fr;
        std::memset(&fr, 0, sizeof(fr));
        for (int i = 0; i < m; i++) {
            if (!src[i0 + i] || !dst[i0 + i]) return GMAT_ERR(EINVAL);
            fr.src[i] = src[i0 + i]; fr.dst[i] = dst[i0 + i];
        }
        int r;
        switch (op) {
        case GMAT_OP_ROTATE_FLIP_SMOOTH: r = launch_rotate_flip_smooth(nullptr, ss, nullptr, ds, w, h, bpp, stream, &fr, m); break;
        case GMAT_OP_SMOOTH3X3:          r = launch_conv3x3(nullptr, ss, nullptr, ds, w, h, bpp, m121, 1.0f / 16.0f, 0.0f, stream, &fr, m); break;
        case GMAT_OP_TRANSPOSE:          r = launch_transpose(nullptr, ss, nullptr, ds, w, h, bpp, arg, stream, &fr, m); break;
        case GMAT_OP_FLIP:               r = launch_flip(nullptr, ss, nullptr, ds, w, h, bpp, arg != 0, arg <= 0, stream, &fr, m); break;
        case GMAT_OP_MEDIAN3X3:          r = launch_median3x3(nullptr, ss, nullptr, ds, w, h, bpp, stream, &fr, m); break;
        default: return GMAT_ERR(EINVAL);
        }
        if (r < 0) return r;
    }
    return 0;
}

int gmat_op_batch(int op, int n, const uint8_t *const *src, int ss, uint8_t *const *dst, int ds, int w, int h, int bpp, int arg, void *stream)
{
    if (n < 0 || !src || !dst) return GMAT_ERR(EINVAL);
    if (op == GMAT_OP_TRANSPOSE && (arg < 0 || arg > 3)) return GMAT_ERR(EINVAL);
    if (op == GMAT_OP_FLIP && (arg < -1 || arg > 1)) return GMAT_ERR(EINVAL);
    return op_batch(op, n, src, ss, dst, ds, w, h, bpp, arg, (hipStream_t)stream);
}

int gmat_rotate_flip_smooth(const uint8_t *src, int ss, uint8_t *dst, int ds, int inW, int inH, int bpp, void *stream)
{
    return launch_rotate_flip_smooth(src, ss, dst, ds, inW, inH, bpp, (hipStream_t)stream);
}

} // extern "C"
