// gruntime.cpp — logging, device selection, raw memory, HIP-event timers and graph capture
// (include/gmat_hip.h §4).  The reference has av_log and nothing else here.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <new>
#include <string>
#include <vector>
#include <sched.h>
#include "common.h"

namespace gmat {

static gmat_log_fn g_log = nullptr;

void logf(int level, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (g_log) g_log(level, buf);
    else if (level <= LOG_ERROR) fprintf(stderr, "[gmat_hip] %s\n", buf);
}

static std::atomic<unsigned long long> g_knob_epoch{1};
void knobs_refresh() { g_knob_epoch.fetch_add(1, std::memory_order_relaxed); }
unsigned long long knobs_epoch() { return g_knob_epoch.load(std::memory_order_relaxed); }
// the call site's cache (thread-local, so no lock): valid while no context has been created since it was filled
const char *knob_read(const char *name, unsigned long long *seen, char *buf, unsigned bufsz, int *present)
{
    const unsigned long long e = knobs_epoch();
    if (*seen != e) {
        const char *v = getenv(name);
        *present = v != nullptr;
        if (v) { snprintf(buf, bufsz, "%s", v); }
        *seen = e;
    }
    return *present ? buf : nullptr;
}
const char *knob(const char *name)
{
    static thread_local struct { unsigned long long seen = ~0ull; std::string name, val; int present = 0; } slots[16];
    const unsigned long long e = knobs_epoch();
    for (auto &s : slots) {
        if (s.name == name) {
            if (s.seen != e) { const char *v = getenv(name); s.present = v != nullptr; s.val = v ? v : ""; s.seen = e; }
            return s.present ? s.val.c_str() : nullptr;
        }
    }
    for (auto &s : slots) {
        if (s.name.empty()) { s.name = name; const char *v = getenv(name); s.present = v != nullptr; s.val = v ? v : ""; s.seen = e; return s.present ? s.val.c_str() : nullptr; }
    }
    return getenv(name);
}

} // namespace gmat

using namespace gmat;

struct GmatTimer { hipEvent_t e0, e1; };

extern "C" {

void gmat_set_log_callback(gmat_log_fn fn) { g_log = fn; }
const char *gmat_version(void) { return "gmat_hip 0.1 (gfx950)"; }
void gmat_knobs_reload(void) { knobs_refresh(); }

int gmat_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int gmat_set_device(int device)
{
    GMAT_HIP_CHECK(hipSetDevice(device));
    return 0;
}

// ---- host placement ("each GPU gets its own host thread, pinned staging ring") ------------------------------
// The reference selects the device per stream (libavutil/hwcontext_cuda.c:395-434) and leaves host placement to the OS.  With
// eight streams each moving 75 GB/s over PCIe, a rank whose staging ring lives on the other socket pays the socket link on every
// frame: these two calls let the caller put its thread (and, by first touch, the ring it allocates next) next to its GPU.
static bool device_sysfs_dir(int device, char *out, size_t n)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess || !bus[0]) return false;
    for (char *c = bus; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');     // sysfs spells the address in lower case
    snprintf(out, n, "/sys/bus/pci/devices/%s", bus);
    return true;
}

static bool read_small_file(const std::string &path, char *buf, size_t n)
{
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return false;
    const size_t got = fread(buf, 1, n - 1, f);
    fclose(f);
    buf[got] = 0;
    return got > 0;
}

int gmat_device_numa_node(int device)
{
    char dir[128], buf[64];
    if (!device_sysfs_dir(device, dir, sizeof(dir)) || !read_small_file(std::string(dir) + "/numa_node", buf, sizeof(buf))) return -1;
    return atoi(buf);                                           // -1: the platform reports no affinity
}

int gmat_device_compute_units(int device)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return GMAT_ERR(EINVAL);
    return prop.multiProcessorCount;
}

int gmat_bind_thread_to_device(int device)
{
    char dir[128], buf[4096];
    if (!device_sysfs_dir(device, dir, sizeof(dir)) || !read_small_file(std::string(dir) + "/local_cpulist", buf, sizeof(buf))) return 0;
    cpu_set_t want, have, both;
    CPU_ZERO(&want);
    // "0-15,128-143": comma-separated ranges
    for (const char *p = buf; *p;) {
        char *e = nullptr;
        const long a = strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        p = e;
        if (*p == '-') { b = strtol(p + 1, &e, 10); p = e; }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (c >= 0) CPU_SET((int)c, &want);
        while (*p == ',' || *p == ' ' || *p == '\n') p++;
    }
    if (sched_getaffinity(0, sizeof(have), &have) != 0) return 0;
    CPU_AND(&both, &want, &have);                               // never widen what the launcher (a cgroup, taskset) allowed
    const int n = CPU_COUNT(&both);
    if (n < 1 || n == CPU_COUNT(&have)) return 0;               // nothing to narrow (one node, or no overlap): leave it
    if (sched_setaffinity(0, sizeof(both), &both) != 0) return 0;
    return n;
}

int gmat_malloc(uint8_t **ptr, size_t bytes)
{
    if (!ptr) return GMAT_ERR(EINVAL);
    GMAT_HIP_CHECK(hipMalloc((void **)ptr, bytes ? bytes : 1));
    return 0;
}

int gmat_free(uint8_t *ptr)
{
    if (ptr) GMAT_HIP_CHECK(hipFree(ptr));
    return 0;
}

int gmat_memcpy_h2d(uint8_t *dst, const uint8_t *src, size_t bytes)
{
    GMAT_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}

int gmat_memcpy_d2h(uint8_t *dst, const uint8_t *src, size_t bytes)
{
    GMAT_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int gmat_memset(uint8_t *dst, int value, size_t bytes)
{
    // hipMemset on device memory is asynchronous to the host; this helper is for set-up code, so wait
    GMAT_HIP_CHECK(hipMemset(dst, value, bytes));
    GMAT_HIP_CHECK(hipDeviceSynchronize());
    return 0;
}

int gmat_stream_create(void **stream)
{
    hipStream_t s;
    GMAT_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return 0;
}

int gmat_stream_destroy(void *stream)
{
    if (stream) GMAT_HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
    return 0;
}

int gmat_stream_sync(void *stream)
{
    GMAT_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int gmat_device_sync(void)
{
    GMAT_HIP_CHECK(hipDeviceSynchronize());
    return 0;
}

int gmat_event_create(void **event)
{
    hipEvent_t e;
    GMAT_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event = (void *)e;
    return 0;
}

int gmat_event_record(void *event, void *stream)
{
    GMAT_HIP_CHECK(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return 0;
}

int gmat_stream_wait_event(void *stream, void *event)
{
    GMAT_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return 0;
}

int gmat_event_sync(void *event)
{
    GMAT_HIP_CHECK(hipEventSynchronize((hipEvent_t)event));
    return 0;
}

void gmat_event_destroy(void *event)
{
    if (event) (void)hipEventDestroy((hipEvent_t)event);
}

int gmat_timer_create(void **timer)
{
    GmatTimer *t = new (std::nothrow) GmatTimer();
    if (!t) return GMAT_ERR(ENOMEM);
    if (hipEventCreate(&t->e0) != hipSuccess || hipEventCreate(&t->e1) != hipSuccess) {
        delete t;
        return GMAT_ERR(EIO);
    }
    *timer = t;
    return 0;
}

int gmat_timer_begin(void *timer, void *stream)
{
    GMAT_HIP_CHECK(hipEventRecord(((GmatTimer *)timer)->e0, (hipStream_t)stream));
    return 0;
}

int gmat_timer_end(void *timer, void *stream)
{
    GMAT_HIP_CHECK(hipEventRecord(((GmatTimer *)timer)->e1, (hipStream_t)stream));
    return 0;
}

int gmat_timer_elapsed_ms(void *timer, float *ms)
{
    GmatTimer *t = (GmatTimer *)timer;
    GMAT_HIP_CHECK(hipEventSynchronize(t->e1));
    GMAT_HIP_CHECK(hipEventElapsedTime(ms, t->e0, t->e1));
    return 0;
}

void gmat_timer_destroy(void *timer)
{
    GmatTimer *t = (GmatTimer *)timer;
    if (!t) return;
    (void)hipEventDestroy(t->e0);
    (void)hipEventDestroy(t->e1);
    delete t;
}

// Capture one gmat_sws_scale() per frame set into a graph: the per-frame kernels are a few
// microseconds long, so replaying a captured batch removes the per-launch host cost.
int gmat_sws_graph_create(GmatSwsContext *c, int nframes, const uint8_t *const *src_planes, const int srcStride[],
                          uint8_t *const *dst_planes, const int dstStride[], void *stream, int nbranches,
                          void **graph_exec)
{
    if (!c || nframes < 1 || !graph_exec) return GMAT_ERR(EINVAL);
    hipStream_t s = (hipStream_t)stream;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    const int srcH = gmat::sws_src_height(c);
    if (nbranches < 1) nbranches = 1;
    if (nbranches > 8) nbranches = 8;
    if (gmat::sws_owns_intermediates(c)) nbranches = 1;     // captured work is ordered by its graph alone (stream_handoff_*, gsws.cpp)
    gmat_sws_setStream(c, stream);
    // warm launches outside capture so lazy allocations (the intermediate frames of the two-kernel form) are not captured
    {
        int r = gmat_sws_scale(c, src_planes, srcStride, 0, srcH, dst_planes, dstStride);
        if (r < 0) return r;
        if (nframes >= 2 * nbranches) {
            r = gmat::sws_scale_frames_batched(c, nframes / nbranches + (nframes % nbranches ? 1 : 0), src_planes, srcStride, dst_planes, dstStride, s);
            if (r < 0) return r;
        }
        GMAT_HIP_CHECK(hipStreamSynchronize(s));
    }
    // side streams + fork/join events for the parallel branches (only needed while capturing)
    hipStream_t side[8] = {nullptr};
    hipEvent_t fork = nullptr, join[8] = {nullptr};
    auto release = [&]() {                 // every exit path gives the capture-time streams and events back
        for (int b = 1; b < 8; b++) {
            if (side[b]) (void)hipStreamDestroy(side[b]);
            if (join[b]) (void)hipEventDestroy(join[b]);
            side[b] = nullptr; join[b] = nullptr;
        }
        if (fork) (void)hipEventDestroy(fork);
        fork = nullptr;
    };
    hipError_t e = hipSuccess;
    for (int b = 1; b < nbranches && e == hipSuccess; b++) {
        e = hipStreamCreateWithFlags(&side[b], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&join[b], hipEventDisableTiming);
    }
    if (nbranches > 1 && e == hipSuccess) e = hipEventCreateWithFlags(&fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
        gmat::logf(gmat::LOG_ERROR, "gmat_sws_graph_create: %s while preparing the capture", hipGetErrorName(e));
        release();
        return GMAT_ERR(EIO);
    }
    int rr = 0;
    if (nbranches > 1) {
        e = hipEventRecord(fork, s);
        for (int b = 1; b < nbranches && e == hipSuccess; b++) e = hipStreamWaitEvent(side[b], fork, 0);
    }
    // contexts on the 2:1 kernel: each branch is one launch carrying a contiguous share of the frames
    bool batched = nframes >= 2 * nbranches && e == hipSuccess;
    for (int b = 0, f0 = 0; batched && b < nbranches && rr >= 0; b++) {
        const int n = nframes / nbranches + (b < nframes % nbranches ? 1 : 0);
        hipStream_t bs = b ? side[b] : s;
        const int t = gmat::sws_scale_frames_batched(c, n, src_planes + 4 * f0, srcStride, dst_planes + 4 * f0, dstStride, bs);
        if (t < 0) rr = t;
        else if (t == 0) {
            if (b == 0) { batched = false; break; }
            for (int f = f0; f < f0 + n && rr >= 0; f++) {
                gmat_sws_setStream(c, (void *)bs);
                rr = gmat_sws_scale(c, src_planes + 4 * f, srcStride, 0, srcH, dst_planes + 4 * f, dstStride);
            }
        }
        f0 += n;
    }
    for (int f = 0; !batched && f < nframes && rr >= 0 && e == hipSuccess; f++) {
        const int b = f % nbranches;
        gmat_sws_setStream(c, b ? (void *)side[b] : stream);
        rr = gmat_sws_scale(c, src_planes + 4 * f, srcStride, 0, srcH, dst_planes + 4 * f, dstStride);
    }
    gmat_sws_setStream(c, stream);
    for (int b = 1; b < nbranches && e == hipSuccess; b++) {
        e = hipEventRecord(join[b], side[b]);
        if (e == hipSuccess) e = hipStreamWaitEvent(s, join[b], 0);
    }
    hipError_t e2 = hipStreamEndCapture(s, &graph);
    release();
    if (e == hipSuccess) e = e2;
    if (rr < 0 || e != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        return rr < 0 ? rr : GMAT_ERR(EIO);
    }
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return GMAT_ERR(EIO);
    *graph_exec = (void *)exec;
    return 0;
}

int gmat_sws_scale_batch(GmatSwsContext *c, int nframes, const uint8_t *const *src_planes, const int srcStride[],
                         uint8_t *const *dst_planes, const int dstStride[], void *const *streams, int nstreams, int flags)
{
    if (!c || nframes < 0 || !streams || nstreams < 1) return GMAT_ERR(EINVAL);
    const int srcH = gmat::sws_src_height(c);
    if (gmat::sws_shares_intermediate(c)) nstreams = 1;
    if (nstreams > 8) nstreams = 8;
    void *saved = gmat::sws_current_stream(c);
    // fork from streams[0] and join back into it, so that anything ordered on streams[0] (events, later
    // work) is ordered against the whole batch
    const bool fork = (flags & GMAT_BATCH_FORK) && nstreams > 1, join = (flags & GMAT_BATCH_JOIN) && nstreams > 1;
    hipEvent_t *ev = (fork || join) ? gmat::sws_batch_events(c) : nullptr;   // [0] fork, [1..8] joins
    if (fork) {
        if (!ev) return GMAT_ERR(ENOMEM);
        GMAT_HIP_CHECK(hipEventRecord(ev[0], (hipStream_t)streams[0]));
        for (int s = 1; s < nstreams; s++) GMAT_HIP_CHECK(hipStreamWaitEvent((hipStream_t)streams[s], ev[0], 0));
    }
    int r = 0;
    // contexts on the 2:1 kernel: one launch per stream, each carrying a contiguous share of the frames (grid.y =
    // frame) — no launch gaps and no half-empty last wave of blocks between frames
    bool batched = nframes >= 2 * nstreams;
    for (int s = 0, f0 = 0; batched && s < nstreams && r >= 0; s++) {
        const int n = nframes / nstreams + (s < nframes % nstreams ? 1 : 0);
        const int t = gmat::sws_scale_frames_batched(c, n, src_planes + 4 * f0, srcStride, dst_planes + 4 * f0, dstStride,
                                                     (hipStream_t)streams[s]);
        if (t < 0) r = t;
        else if (t == 0) {
            if (s > 0) {            // eligibility is a property of the context and the strides; later shares differ only in
                for (int f = f0; f < f0 + n && r >= 0; f++) {   // pointer alignment: finish those frame by frame
                    gmat_sws_setStream(c, streams[s]);
                    r = gmat_sws_scale(c, src_planes + 4 * f, srcStride, 0, srcH, dst_planes + 4 * f, dstStride);
                }
            } else {
                batched = false;
            }
        }
        f0 += n;
    }
    for (int f = 0; !batched && f < nframes && r >= 0; f++) {
        gmat_sws_setStream(c, streams[f % nstreams]);
        r = gmat_sws_scale(c, src_planes + 4 * f, srcStride, 0, srcH, dst_planes + 4 * f, dstStride);
    }
    gmat_sws_setStream(c, saved);
    if ((fork || join) && !ev) return GMAT_ERR(ENOMEM);
    for (int s = 1; join && s < nstreams; s++) {
        GMAT_HIP_CHECK(hipEventRecord(ev[s], (hipStream_t)streams[s]));
        GMAT_HIP_CHECK(hipStreamWaitEvent((hipStream_t)streams[0], ev[s], 0));
    }
    return r < 0 ? r : nframes;
}

int gmat_graph_launch(void *graph_exec, void *stream)
{
    GMAT_HIP_CHECK(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return 0;
}

void gmat_graph_destroy(void *graph_exec)
{
    if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
}

} // extern "C"
