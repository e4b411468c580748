// gframes.cpp — device frame pools and host<->device transfers (include/gmat_hip.h §2).
//
// Restates the part of libavutil the GPU filters rely on (paths relative to
// /root/reference/ffmpeg-gpu/libavutil):
//   AVHWFramesContext {format, sw_format, width, height, pool}     hwcontext.h:124-229
//   cuda_frames_init / cuda_pool_alloc / cuda_get_buffer            hwcontext_cuda.c:96-193
//   cuda_transfer_data (one cuMemcpy2DAsync per plane)              hwcontext_cuda.c:221-279
// Unlike hwcontext_cuda.c:38-52 the packed RGB formats the nvcv filters need are accepted
// (SURVEY.md §0 defect 10), and host staging memory is pinned so copies overlap compute.
#include <cstring>
#include <mutex>
#include <new>
#include <vector>
#include "common.h"

using namespace gmat;

namespace {

const int kLineAlign = 256;   // hwcontext_cuda.c:145-157 aligns rows to the texture alignment

struct PlaneLayout { int planes; int linesize[4]; size_t offset[4]; size_t total; };

int layout_for(int fmt, int w, int h, int align, PlaneLayout &L)
{
    std::memset(&L, 0, sizeof(L));
    switch (fmt) {
    case GMAT_PIX_FMT_RGB0: case GMAT_PIX_FMT_BGR0:
        L.planes = 1;
        L.linesize[0] = align_up(w * 4, align);
        L.total = (size_t)L.linesize[0] * h;
        return 0;
    case GMAT_PIX_FMT_RGB24: case GMAT_PIX_FMT_BGR24: case GMAT_PIX_FMT_RGBA: case GMAT_PIX_FMT_BGRA:
        L.planes = 1;
        L.linesize[0] = align_up(w * bytes_per_pixel(fmt), align);
        L.total = (size_t)L.linesize[0] * h;
        return 0;
    case GMAT_PIX_FMT_RGBA64LE: case GMAT_PIX_FMT_BGRA64LE:
        L.planes = 1;
        L.linesize[0] = align_up(w * 8, align);
        L.total = (size_t)L.linesize[0] * h;
        return 0;
    case GMAT_PIX_FMT_NV12:
        // data[1] = data[0] + linesize[0]*height (the layout yuv2rgb_cuda.cu:226 assumes)
        L.planes = 2;
        L.linesize[0] = L.linesize[1] = align_up(w, align);
        L.offset[1] = (size_t)L.linesize[0] * h;
        L.total = L.offset[1] + (size_t)L.linesize[1] * ceil_rshift(h, 1);
        return 0;
    case GMAT_PIX_FMT_P010LE:
    case GMAT_PIX_FMT_P016LE:
        // NV12's plane structure with 16-bit samples (hwcontext_cuda.c:38-52 lists both)
        L.planes = 2;
        L.linesize[0] = L.linesize[1] = align_up(2 * w, align);
        L.offset[1] = (size_t)L.linesize[0] * h;
        L.total = L.offset[1] + (size_t)L.linesize[1] * ceil_rshift(h, 1);
        return 0;
    case GMAT_PIX_FMT_YUV420P:
        // hwcontext_cuda.c:188-193: linesize[1] = linesize[2] = linesize[0]/2
        L.planes = 3;
        L.linesize[0] = align_up(w, align);
        L.linesize[1] = L.linesize[2] = L.linesize[0] / 2;
        L.offset[1] = (size_t)L.linesize[0] * h;
        L.offset[2] = L.offset[1] + (size_t)L.linesize[1] * ceil_rshift(h, 1);
        L.total = L.offset[2] + (size_t)L.linesize[2] * ceil_rshift(h, 1);
        return 0;
    case GMAT_PIX_FMT_YUV444P:
        L.planes = 3;
        for (int i = 0; i < 3; i++) {
            L.linesize[i] = align_up(w, align);
            L.offset[i] = (size_t)i * L.linesize[0] * h;
        }
        L.total = (size_t)3 * L.linesize[0] * h;
        return 0;
    case GMAT_PIX_FMT_YUV420P16LE:
    case GMAT_PIX_FMT_YUV420P10LE:
        L.planes = 3;
        L.linesize[0] = align_up(2 * w, align);
        L.linesize[1] = L.linesize[2] = align_up(2 * ceil_rshift(w, 1), align);
        L.offset[1] = (size_t)L.linesize[0] * h;
        L.offset[2] = L.offset[1] + (size_t)L.linesize[1] * ceil_rshift(h, 1);
        L.total = L.offset[2] + (size_t)L.linesize[2] * ceil_rshift(h, 1);
        return 0;
    case GMAT_PIX_FMT_YUV444P16LE:
        L.planes = 3;
        for (int i = 0; i < 3; i++) {
            L.linesize[i] = align_up(2 * w, align);
            L.offset[i] = (size_t)i * L.linesize[0] * h;
        }
        L.total = (size_t)3 * L.linesize[0] * h;
        return 0;
    case GMAT_PIX_FMT_RGBPF32LE:
        L.planes = 3;
        for (int i = 0; i < 3; i++) {
            L.linesize[i] = align_up(w * 4, align);
            L.offset[i] = (size_t)i * L.linesize[0] * h;
        }
        L.total = (size_t)3 * L.linesize[0] * h;
        return 0;
    }
    return GMAT_ERR(ENOSYS);
}

int plane_rows(int fmt, int plane, int h)
{
    if ((fmt == GMAT_PIX_FMT_NV12 || fmt == GMAT_PIX_FMT_YUV420P || is_p01x(fmt) || fmt == GMAT_PIX_FMT_YUV420P16LE ||
         fmt == GMAT_PIX_FMT_YUV420P10LE) && plane > 0) return ceil_rshift(h, 1);
    return h;
}

int plane_row_bytes(int fmt, int plane, int w)
{
    switch (fmt) {
    case GMAT_PIX_FMT_NV12:      return plane == 0 ? w : 2 * ceil_rshift(w, 1);
    case GMAT_PIX_FMT_YUV420P:   return plane == 0 ? w : ceil_rshift(w, 1);
    case GMAT_PIX_FMT_YUV444P:   return w;
    case GMAT_PIX_FMT_P010LE:
    case GMAT_PIX_FMT_P016LE:    return plane == 0 ? 2 * w : 4 * ceil_rshift(w, 1);
    case GMAT_PIX_FMT_RGBPF32LE: return 4 * w;
    case GMAT_PIX_FMT_RGBA64LE:
    case GMAT_PIX_FMT_BGRA64LE:  return 8 * w;
    case GMAT_PIX_FMT_RGB0:
    case GMAT_PIX_FMT_BGR0:      return 4 * w;
    case GMAT_PIX_FMT_YUV444P16LE: return 2 * w;
    case GMAT_PIX_FMT_YUV420P16LE:
    case GMAT_PIX_FMT_YUV420P10LE: return plane == 0 ? 2 * w : 2 * ceil_rshift(w, 1);
    default:                     return w * bytes_per_pixel(fmt);
    }
}

} // namespace

// Lifetime (the reference keeps an AVHWFramesContext alive through AVBufferRef counts, hwcontext.c:236-258): the
// context holds one reference for its owner plus one per frame handed out.  gmat_hwframe_ctx_free drops the owner's
// and closes the pool; frames still out keep the object alive, and returning a frame to a closed pool frees its block.
struct GmatHWFramesContext {
    int device, sw_format, width, height;
    PlaneLayout layout;
    std::mutex lock;
    std::vector<uint8_t *> free_list;   // pooled hipMalloc blocks (av_buffer_pool equivalent)
    int outstanding = 0;
    bool closed = false;
};

// returns a frame's block to its pool; true when the context itself must be deleted (closed and last frame back)
static bool pool_release(GmatHWFramesContext *fc, uint8_t *block)
{
    bool last;
    bool free_block;
    {
        std::lock_guard<std::mutex> g(fc->lock);
        free_block = fc->closed;
        if (!free_block) fc->free_list.push_back(block);
        fc->outstanding--;
        last = fc->closed && fc->outstanding == 0;
    }
    if (free_block) (void)hipFree(block);
    return last;
}

extern "C" {

GmatHWFramesContext *gmat_hwframe_ctx_create(int device, int sw_format, int width, int height, int initial_pool_size)
{
    if (width < 1 || height < 1) return nullptr;
    GmatHWFramesContext *fc = new (std::nothrow) GmatHWFramesContext();
    if (!fc) return nullptr;
    fc->device = device; fc->sw_format = sw_format; fc->width = width; fc->height = height;
    if (layout_for(sw_format, width, height, kLineAlign, fc->layout) < 0) {
        logf(LOG_ERROR, "gmat_hwframe_ctx_create: pixel format %d is not supported", sw_format);
        delete fc;
        return nullptr;
    }
    DeviceScope onDevice;
    if (onDevice.enter(device) < 0) { delete fc; return nullptr; }
    for (int i = 0; i < initial_pool_size; i++) {
        uint8_t *p = nullptr;
        if (hipMalloc((void **)&p, fc->layout.total) != hipSuccess) break;
        fc->free_list.push_back(p);
    }
    return fc;
}

void gmat_hwframe_ctx_free(GmatHWFramesContext *fc)
{
    if (!fc) return;
    std::vector<uint8_t *> blocks;
    bool last;
    {
        std::lock_guard<std::mutex> g(fc->lock);
        if (fc->closed) return;                              // the owner's reference is dropped once
        fc->closed = true;
        blocks.swap(fc->free_list);
        last = fc->outstanding == 0;
    }
    for (uint8_t *p : blocks) (void)hipFree(p);
    if (last) delete fc;                                     // else the last gmat_frame_unref / gmat_frame_free deletes it
}

int gmat_hwframe_ctx_info(const GmatHWFramesContext *fc, int *device, int *sw_format, int *width, int *height)
{
    if (!fc) return GMAT_ERR(EINVAL);
    if (device) *device = fc->device;
    if (sw_format) *sw_format = fc->sw_format;
    if (width) *width = fc->width;
    if (height) *height = fc->height;
    return 0;
}

int gmat_hwframe_get_buffer(GmatHWFramesContext *fc, GmatFrame *f)
{
    if (!fc || !f) return GMAT_ERR(EINVAL);
    uint8_t *base = nullptr;
    {
        std::lock_guard<std::mutex> g(fc->lock);
        if (fc->closed) return GMAT_ERR(EINVAL);
        if (!fc->free_list.empty()) { base = fc->free_list.back(); fc->free_list.pop_back(); }
        fc->outstanding++;
    }
    if (!base) {
        DeviceScope onDevice;
        if (int e = onDevice.enter(fc->device); e < 0) return e;
        if (hipMalloc((void **)&base, fc->layout.total) != hipSuccess) {
            bool last;
            {
                std::lock_guard<std::mutex> g(fc->lock);
                fc->outstanding--;
                last = fc->closed && fc->outstanding == 0;
            }
            if (last) delete fc;
            return GMAT_ERR(ENOMEM);
        }
    }
    std::memset(f->data, 0, sizeof(f->data));
    std::memset(f->linesize, 0, sizeof(f->linesize));
    for (int i = 0; i < fc->layout.planes; i++) {
        f->data[i] = base + fc->layout.offset[i];
        f->linesize[i] = fc->layout.linesize[i];
    }
    f->width = fc->width; f->height = fc->height;
    f->format = GMAT_PIX_FMT_HIP; f->sw_format = fc->sw_format;
    f->hw_frames_ctx = fc; f->buf = base;
    return 0;
}

GmatFrame *gmat_frame_alloc(void)
{
    GmatFrame *f = new (std::nothrow) GmatFrame();
    if (f) { std::memset(f, 0, sizeof(*f)); f->format = GMAT_PIX_FMT_NONE; f->sw_format = GMAT_PIX_FMT_NONE; }
    return f;
}

void gmat_frame_unref(GmatFrame *f)
{
    if (!f) return;
    if (f->hw_frames_ctx && f->buf) {
        GmatHWFramesContext *fc = f->hw_frames_ctx;
        if (pool_release(fc, (uint8_t *)f->buf)) delete fc;
    }
    std::memset(f->data, 0, sizeof(f->data));
    std::memset(f->linesize, 0, sizeof(f->linesize));
    f->hw_frames_ctx = nullptr; f->buf = nullptr;
    f->format = GMAT_PIX_FMT_NONE; f->sw_format = GMAT_PIX_FMT_NONE;
}

void gmat_frame_free(GmatFrame **pf)
{
    if (!pf || !*pf) return;
    gmat_frame_unref(*pf);
    delete *pf;
    *pf = nullptr;
}

int gmat_host_frame_alloc(GmatFrame *f, int sw_format, int width, int height)
{
    if (!f) return GMAT_ERR(EINVAL);
    PlaneLayout L;
    int r = layout_for(sw_format, width, height, 64, L);
    if (r < 0) return r;
    uint8_t *base = nullptr;
    GMAT_HIP_CHECK(hipHostMalloc((void **)&base, L.total, hipHostMallocDefault));
    std::memset(f, 0, sizeof(*f));
    for (int i = 0; i < L.planes; i++) { f->data[i] = base + L.offset[i]; f->linesize[i] = L.linesize[i]; }
    f->width = width; f->height = height;
    f->format = sw_format; f->sw_format = sw_format;
    f->buf = base;
    return 0;
}

void gmat_host_frame_free(GmatFrame *f)
{
    if (f && f->buf && !f->hw_frames_ctx) { (void)hipHostFree(f->buf); f->buf = nullptr; f->data[0] = nullptr; }
}

int gmat_hwframe_transfer_data(GmatFrame *dst, const GmatFrame *src, void *stream)
{
    if (!dst || !src) return GMAT_ERR(EINVAL);
    const bool up = dst->format == GMAT_PIX_FMT_HIP && src->format != GMAT_PIX_FMT_HIP;
    const bool down = src->format == GMAT_PIX_FMT_HIP && dst->format != GMAT_PIX_FMT_HIP;
    // both frames on the device: cuda_transfer_data copies whatever memory types the two sides have (hwcontext_cuda.c:239-252: srcMemoryType /
    // dstMemoryType by hw_frames_ctx), and av_hwframe_transfer_data reaches it for two hardware frames (hwcontext.c:448-467)
    const bool d2d = dst->format == GMAT_PIX_FMT_HIP && src->format == GMAT_PIX_FMT_HIP;
    if (!up && !down && !d2d) return GMAT_ERR(EINVAL);
    const int fmt = up ? dst->sw_format : src->sw_format;
    if ((up ? src->sw_format : dst->sw_format) != fmt || src->width != dst->width || src->height != dst->height)
        return GMAT_ERR(EINVAL);
    PlaneLayout L;
    int r = layout_for(fmt, src->width, src->height, 1, L);
    if (r < 0) return r;
    for (int i = 0; i < L.planes; i++) {
        GMAT_HIP_CHECK(hipMemcpy2DAsync(dst->data[i], (size_t)dst->linesize[i], src->data[i], (size_t)src->linesize[i],
                                        (size_t)plane_row_bytes(fmt, i, src->width), (size_t)plane_rows(fmt, i, src->height),
                                        d2d ? hipMemcpyDeviceToDevice : up ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, (hipStream_t)stream));
    }
    dst->pts = src->pts;
    dst->colorspace = src->colorspace;
    return 0;
}

} // extern "C"
