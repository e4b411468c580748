// gpipeline.cpp — host <-> device frame pipeline behind the C ABI (include/gmat_hip.h §5).
//
// What hwupload -> scale -> hwdownload does in the reference (libavfilter/vf_hwupload_cuda.c:123-150 allocates the
// device frame and calls av_hwframe_transfer_data; libavutil/hwcontext_cuda.c:221-279 issues one cuMemcpy2DAsync per
// plane on the device context's single stream from PAGEABLE memory and the caller waits), rebuilt for overlap:
//   * host frames live in PINNED staging rings (hipHostMalloc), `depth` slots;
//   * three HIP streams per pipeline — upload, compute, download — chained per slot by events, so the upload of
//     frame k + 1 and the download of frame k - 1 run beside the kernels of frame k;
//   * slot k mod depth is reused only after its previous download has completed (gmat_pipeline_submit blocks on that
//     event and on nothing else).
// One pipeline belongs to one device (hipSetDevice(device) on entry to every call, like the reference's
// cuCtxPushCurrent, hwcontext_cuda.c:395-434) and one host thread; a node with N GPUs runs N pipelines.
// Software decode / encode stay on the host (VCN is out of scope): frames arrive as host NV12.
#include <new>
#include <vector>
#include "common.h"

using namespace gmat;

struct GmatPipeline {
    int device = 0, depth = 0;
    int srcW = 0, srcH = 0, srcFormat = 0, dstW = 0, dstH = 0, dstFormat = 0;
    GmatSwsContext *ctx = nullptr;
    hipStream_t up = nullptr, comp = nullptr, down = nullptr;
    GmatHWFramesContext *inPool = nullptr, *outPool = nullptr;
    struct Slot {
        GmatFrame hin{}, hout{}, din{}, dout{};
        hipEvent_t uploaded = nullptr, computed = nullptr, downloaded = nullptr;
        bool busy = false;
    };
    std::vector<Slot> slots;
    int64_t next = 0;
};

extern "C" {

void gmat_pipeline_free(GmatPipeline *p)
{
    if (!p) return;
    DeviceScope onDevice;
    (void)onDevice.enter(p->device);
    for (auto &s : p->slots)
        if (s.busy && s.downloaded) (void)hipEventSynchronize(s.downloaded);
    for (auto &s : p->slots) {
        gmat_host_frame_free(&s.hin); gmat_host_frame_free(&s.hout);
        gmat_frame_unref(&s.din); gmat_frame_unref(&s.dout);          // back to the pools before the pools go
        for (hipEvent_t e : {s.uploaded, s.computed, s.downloaded}) if (e) (void)hipEventDestroy(e);
    }
    if (p->ctx) gmat_sws_freeContext(p->ctx);
    for (hipStream_t st : {p->up, p->comp, p->down}) if (st) (void)hipStreamDestroy(st);
    gmat_hwframe_ctx_free(p->inPool); gmat_hwframe_ctx_free(p->outPool);
    delete p;
}

GmatPipeline *gmat_pipeline_create(int device, int srcW, int srcH, int srcFormat, int dstW, int dstH, int dstFormat, int flags, int depth)
{
    if (depth < 1 || depth > 64) return nullptr;
    DeviceScope onDevice;
    if (onDevice.enter(device) < 0) return nullptr;
    GmatPipeline *p = new (std::nothrow) GmatPipeline();
    if (!p) return nullptr;
    p->device = device; p->depth = depth;
    p->srcW = srcW; p->srcH = srcH; p->srcFormat = srcFormat; p->dstW = dstW; p->dstH = dstH; p->dstFormat = dstFormat;
    bool ok = (p->ctx = gmat_sws_getContext(srcW, srcH, srcFormat, dstW, dstH, dstFormat, flags | GMAT_SWS_HWACCEL, nullptr)) != nullptr;
    ok = ok && hipStreamCreateWithFlags(&p->up, hipStreamNonBlocking) == hipSuccess &&
         hipStreamCreateWithFlags(&p->comp, hipStreamNonBlocking) == hipSuccess &&
         hipStreamCreateWithFlags(&p->down, hipStreamNonBlocking) == hipSuccess;
    if (ok) gmat_sws_setStream(p->ctx, p->comp);
    ok = ok && (p->inPool = gmat_hwframe_ctx_create(device, srcFormat, srcW, srcH, depth)) != nullptr &&
         (p->outPool = gmat_hwframe_ctx_create(device, dstFormat, dstW, dstH, depth)) != nullptr;
    p->slots.resize(ok ? depth : 0);
    for (auto &s : p->slots) {
        if (!ok) break;
        ok = gmat_host_frame_alloc(&s.hin, srcFormat, srcW, srcH) == 0 && gmat_host_frame_alloc(&s.hout, dstFormat, dstW, dstH) == 0 &&
             gmat_hwframe_get_buffer(p->inPool, &s.din) == 0 && gmat_hwframe_get_buffer(p->outPool, &s.dout) == 0 &&
             hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&s.computed, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&s.downloaded, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) { gmat_pipeline_free(p); return nullptr; }
    return p;
}

static int slot_view(GmatPipeline *p, int64_t seq, bool input, GmatFrame *view)
{
    if (!p || !view || seq < 0) return GMAT_ERR(EINVAL);
    const GmatPipeline::Slot &s = p->slots[(size_t)(seq % p->depth)];
    *view = input ? s.hin : s.hout;
    view->buf = nullptr;                      // a view: the ring owns the pinned memory
    return 0;
}

int gmat_pipeline_host_input(GmatPipeline *p, int64_t seq, GmatFrame *view) { return slot_view(p, seq, true, view); }
int gmat_pipeline_host_output(GmatPipeline *p, int64_t seq, GmatFrame *view) { return slot_view(p, seq, false, view); }

int64_t gmat_pipeline_submit(GmatPipeline *p)
{
    if (!p) return GMAT_ERR(EINVAL);
    DeviceScope onDevice;
    if (int e = onDevice.enter(p->device); e < 0) return e;
    GmatPipeline::Slot &s = p->slots[(size_t)(p->next % p->depth)];
    if (s.busy) GMAT_HIP_CHECK(hipEventSynchronize(s.downloaded));      // the slot's previous frame has left the device
    int r = gmat_hwframe_transfer_data(&s.din, &s.hin, p->up);
    if (r < 0) return r;
    GMAT_HIP_CHECK(hipEventRecord(s.uploaded, p->up));
    GMAT_HIP_CHECK(hipStreamWaitEvent(p->comp, s.uploaded, 0));
    r = gmat_sws_scale(p->ctx, s.din.data, s.din.linesize, 0, p->srcH, s.dout.data, s.dout.linesize);
    if (r < 0) return r;
    GMAT_HIP_CHECK(hipEventRecord(s.computed, p->comp));
    GMAT_HIP_CHECK(hipStreamWaitEvent(p->down, s.computed, 0));
    r = gmat_hwframe_transfer_data(&s.hout, &s.dout, p->down);
    if (r < 0) return r;
    GMAT_HIP_CHECK(hipEventRecord(s.downloaded, p->down));
    s.busy = true;
    return p->next++;
}

int gmat_pipeline_wait(GmatPipeline *p, int64_t seq)
{
    if (!p || seq < 0 || seq >= p->next) return GMAT_ERR(EINVAL);
    if (seq + p->depth < p->next) return 0;                   // its slot has been reused: that frame left long ago
    GmatPipeline::Slot &s = p->slots[(size_t)(seq % p->depth)];
    if (s.busy) GMAT_HIP_CHECK(hipEventSynchronize(s.downloaded));
    return 0;
}

int gmat_pipeline_drain(GmatPipeline *p)
{
    if (!p) return GMAT_ERR(EINVAL);
    for (auto &s : p->slots)
        if (s.busy) GMAT_HIP_CHECK(hipEventSynchronize(s.downloaded));
    return 0;
}

} // extern "C"
