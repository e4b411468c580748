// common.h — shared host-side helpers of the MI355X pixel-transform library.
#pragma once
#include <cerrno>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <hip/hip_runtime.h>
#include "../../include/gmat_hip.h"

namespace gmat {

// av_log levels used by the reference filters (libavutil/log.h)
enum { LOG_ERROR = 16, LOG_WARNING = 24, LOG_INFO = 32, LOG_VERBOSE = 40, LOG_DEBUG = 48 };

void logf(int level, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Unlike the reference's CK_NVCV / CHECK_CU, which log and continue (vf_crop_nvcv.c:62-77,
// swscale_cuda.c:23-25), a failed device call always aborts the operation with an error code.
#define GMAT_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            ::gmat::logf(::gmat::LOG_ERROR, "HIP error %s (%d) at %s:%d: %s",            \
                         hipGetErrorName(_e), (int)_e, __FILE__, __LINE__, #expr);       \
            return GMAT_ERR(EIO);                                                         \
        }                                                                                 \
    } while (0)

// cuCtxPushCurrent / cuCtxPopCurrent as the reference brackets every filter and transfer call with (vf_scale_cuda.c:292-294,:553,
// hwcontext_cuda.c:231-276): the object's device is current inside the call and the CALLER's device is current again after it — a
// thread that works with several devices is not left on the last filter's one.
struct DeviceScope {
    int saved = -1;
    int enter(int device)
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        if (cur == device) return 0;
        if (hipSetDevice(device) != hipSuccess) return GMAT_ERR(EIO);
        if (saved < 0) saved = cur;
        return 0;
    }
    ~DeviceScope() { if (saved >= 0) (void)hipSetDevice(saved); }
};

inline bool is_packed_rgb(int f)
{
    return f == GMAT_PIX_FMT_RGB24 || f == GMAT_PIX_FMT_BGR24 || f == GMAT_PIX_FMT_RGBA || f == GMAT_PIX_FMT_BGRA ||
           f == GMAT_PIX_FMT_RGB0 || f == GMAT_PIX_FMT_BGR0;      // the 0-alpha twins: gsws maps them to RGBA / BGRA on entry
}
// 64-bit packed RGB destinations (16 bits per channel, alpha last)
inline bool is_rgb64(int f) { return f == GMAT_PIX_FMT_RGBA64LE || f == GMAT_PIX_FMT_BGRA64LE; }
// destinations of the 19-bit path (k_scale16.hip)
inline bool is_dst16(int f) { return f == GMAT_PIX_FMT_P016LE || f == GMAT_PIX_FMT_YUV444P16LE || f == GMAT_PIX_FMT_YUV420P16LE || is_rgb64(f); }
// ... of them, the ones with planar chroma (yuv2planeX_16_c per plane)
inline bool is_pl16_dst(int f) { return f == GMAT_PIX_FMT_YUV444P16LE || f == GMAT_PIX_FMT_YUV420P16LE; }
// 10-bit destinations of the 15-bit lines: P010LE (interleaved chroma, sample << 6) and planar YUV420P10LE (yuv2planeX_10_c)
inline bool is_dst10(int f) { return f == GMAT_PIX_FMT_P010LE || f == GMAT_PIX_FMT_YUV420P10LE; }
inline bool is_yuv420(int f) { return f == GMAT_PIX_FMT_NV12 || f == GMAT_PIX_FMT_YUV420P; }
// 8-bit YUV sources of the plane scaler: 4:2:0 and (source only) planar 4:4:4
inline bool is_yuv8_src(int f) { return is_yuv420(f) || f == GMAT_PIX_FMT_YUV444P; }
// 16-bit semi-planar 4:2:0 (interleaved U,V; P010: the 10 significant bits are the high ones)
inline bool is_p01x(int f) { return f == GMAT_PIX_FMT_P010LE || f == GMAT_PIX_FMT_P016LE; }
// planar YUV in 16-bit little-endian containers (sources; YUV444P16LE is a destination too): significant bits, 0 = not one.
// The samples go to hScale16To15_c / hScale16To19_c as they are (no input converter on a little-endian host,
// input.c:1523-1528), so the 10-bit form keeps its bits in the LOW end, unlike P010LE.
// library-internal source format (never accepted from a caller): the Y / U / V planes of 16-bit samples that rgb64ToY_c / ToUV_c /
// ToUV_half_c make of an RGBA64LE / BGRA64LE frame (k_rgb64.hip) — planar 16-bit samples with an RGB source's chroma geometry
constexpr int GMAT_PIX_FMT_PRIV_RGB64_PLANES = 0x47520064;
// ... and the 16-bit lines rgb24ToY_c / ToUV_c / ToUV_half_c make of an 8-bit packed RGB frame, for the 19-bit path only (16-bit
// destinations): hScale16To19_c shifts them by 9, not by depth - 5 (swscale.c:74-76)
constexpr int GMAT_PIX_FMT_PRIV_RGB8_PLANES = 0x47520008;
inline bool is_priv_planes(int f) { return f == GMAT_PIX_FMT_PRIV_RGB64_PLANES || f == GMAT_PIX_FMT_PRIV_RGB8_PLANES; }
inline int  pl16_depth(int f) { return (f == GMAT_PIX_FMT_YUV444P16LE || f == GMAT_PIX_FMT_YUV420P16LE || is_priv_planes(f)) ? 16 : f == GMAT_PIX_FMT_YUV420P10LE ? 10 : 0; }
inline bool has_alpha(int f) { return f == GMAT_PIX_FMT_RGBA || f == GMAT_PIX_FMT_BGRA || is_rgb64(f); }
inline int  bytes_per_pixel(int f)
{
    switch (f) {
    case GMAT_PIX_FMT_RGB24: case GMAT_PIX_FMT_BGR24: return 3;
    case GMAT_PIX_FMT_RGBA:  case GMAT_PIX_FMT_BGRA:  case GMAT_PIX_FMT_RGB0: case GMAT_PIX_FMT_BGR0: return 4;
    default: return 0;
    }
}
// internal accessor (gsws.cpp) used by the graph-capture helper
int sws_src_height(const GmatSwsContext *c);
bool sws_shares_intermediate(const GmatSwsContext *c);
bool sws_owns_intermediates(const GmatSwsContext *c);   // ... or may (NV12 <-> YUV420P scaled: the cascade's frame)
void *sws_current_stream(const GmatSwsContext *c);
// frames [0,n) through one launch per 32 frames when the context runs the 2:1 kernel: 1 taken, 0 not eligible, < 0 error
int sws_scale_frames_batched(GmatSwsContext *c, int n, const uint8_t *const *src_planes, const int srcStride[],
                             uint8_t *const *dst_planes, const int dstStride[], hipStream_t stream);
hipEvent_t *sws_batch_events(GmatSwsContext *c);           // 9 lazily created events owned by the context   // two-kernel form: frames must not overlap

// ---- environment knobs (tests and measurements only) ---------------------------------------------------
// A knob is read from the environment when a context is created — gmat_sws_getContext, gmat_filter_init, and the stateless direct
// launchers (gmat_transpose ...), which are their own one-call contexts — and NOT per launch: round 2's launchers called getenv()
// up to three times per launch, a scan of the whole environment on the per-frame path.  knob() hands out
// the value as of the last such creation; GMAT_KNOB("NAME") adds a per-call-site cache, so that a launch costs a compare.
void knobs_refresh();                                  // a context is being created: the next knob() of every name re-reads it
unsigned long long knobs_epoch();
const char *knob_read(const char *name, unsigned long long *seen, char *buf, unsigned bufsz, int *present);
struct KnobSite { unsigned long long seen = ~0ull; char buf[48]; int present = 0; };
inline const char *knob_at(KnobSite &s, const char *name) { return knob_read(name, &s.seen, s.buf, sizeof(s.buf), &s.present); }
#define GMAT_KNOB(name) (::gmat::knob_at(*([]() -> ::gmat::KnobSite * { static thread_local ::gmat::KnobSite s; return &s; }()), name))
const char *knob(const char *name);                    // by a run-time name (no call-site cache)

inline int ceil_rshift(int a, int b) { return -((-a) >> b); }
inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

} // namespace gmat
