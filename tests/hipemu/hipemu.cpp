// tests/hipemu/hipemu.cpp — TEST INFRASTRUCTURE: fiber scheduler and runtime stubs for
// tests/hipemu/hip/hip_runtime.h (see the header for scope and rules).
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <ucontext.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <map>
#include <mutex>
#include <vector>

namespace hipemu {

Idx g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

namespace {
enum State { READY, AT_BLOCK, AT_WAVE, DONE };
struct Fiber {
    ucontext_t ctx;
    State state;
    Idx tid;
};
const size_t kStack = 256 * 1024;
std::vector<Fiber> fibers;
std::vector<char *> stacks;
ucontext_t sched_ctx;
int cur = -1;
const std::function<void()> *cur_body = nullptr;
char *lds_ptr = nullptr;
std::mutex big_lock;          // the emulation is single-threaded by design

void trampoline()
{
    (*cur_body)();
    fibers[cur].state = DONE;
    swapcontext(&fibers[cur].ctx, &sched_ctx);
}

void yield_as(State s)
{
    fibers[cur].state = s;
    const int me = cur;
    swapcontext(&fibers[me].ctx, &sched_ctx);
}

// guard-page allocator: the END of the usable range abuts a PROT_NONE page
struct Guarded { void *map; size_t map_len; };
std::map<void *, Guarded> allocs;

void *guarded_alloc(size_t n, size_t align)
{
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    const size_t body = (n + align - 1) / align * align;
    const size_t pages = (body + page - 1) / page;
    const size_t len = (pages + 2) * page;
    char *m = (char *)mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return nullptr;
    mprotect(m, page, PROT_NONE);
    mprotect(m + (pages + 1) * page, page, PROT_NONE);
    char *p = m + (pages + 1) * page - body;     // aligned because body is a multiple of align
    memset(p, 0xA5, body);
    allocs[p] = {m, len};
    return p;
}

void guarded_free(void *p)
{
    auto it = allocs.find(p);
    if (it == allocs.end()) return;
    munmap(it->second.map, it->second.map_len);
    allocs.erase(it);
}
} // namespace

void block_sync() { yield_as(AT_BLOCK); }
void wave_sync() { yield_as(AT_WAVE); }

// v_readfirstlane_b32: the value of the lowest-numbered ACTIVE lane of the wave.  The callers rendezvous at a wave
// barrier (lanes that have left the enclosing loop and are parked at a block barrier do not take part — see the
// scheduler), the lowest participant's value is handed to all of them.  A per-wave round number tells this
// rendezvous' participants from stale slots; the lowest participant advances it after everybody has read.
static int rfl_slot[1024];
static unsigned rfl_tag[1024], rfl_round[16];
static void rfl_reset()
{
    memset(rfl_tag, 0, sizeof(rfl_tag));
    memset(rfl_round, 0, sizeof(rfl_round));
}
int readfirstlane(int v)
{
    const int me = cur, w = me >> 6;
    const unsigned round = rfl_round[w] + 1;
    rfl_slot[me] = v;
    rfl_tag[me] = round;
    wave_sync();
    int first = me;
    for (int l = w * 64; l < w * 64 + 64 && l < (int)fibers.size(); l++)
        if (rfl_tag[l] == round) { first = l; break; }
    const int r = rfl_slot[first];
    wave_sync();
    if (me == first) rfl_round[w] = round;          // fibers resume in lane order: done before anyone's next call
    return r;
}

// cross-lane exchange (ds_bpermute semantics for the lanes of one wave): every live lane of the wave must
// call it (the wave barrier inside aborts on divergence)
static int shfl_slot[1024];
int shfl_xor(int v, int mask)
{
    const int me = cur;
    shfl_slot[me] = v;
    wave_sync();
    const int src = (me & ~63) | ((me ^ mask) & 63);
    const int r = src < (int)fibers.size() ? shfl_slot[src] : 0;
    wave_sync();
    return r;
}
// v_readlane_b32 and the wave-wide DPP shifts (wave_shr:1 = 0x138: lane i takes lane i-1; wave_shl:1 = 0x130: lane i takes
// lane i+1; a lane without a source keeps `old`).  Every live lane of the wave calls it, like shfl_xor.
int lane_read(int v, int srcLane)
{
    const int me = cur;
    shfl_slot[me] = v;
    wave_sync();
    const int src = (me & ~63) | (srcLane & 63);
    const int r = src < (int)fibers.size() ? shfl_slot[src] : 0;
    wave_sync();
    return r;
}
int dpp_wave_shift(int old, int v, int ctrl)
{
    const int me = cur, lane = me & 63;
    shfl_slot[me] = v;
    wave_sync();
    int r = old;
    if (ctrl == 0x138 && lane > 0) r = shfl_slot[me - 1];
    else if (ctrl == 0x130 && lane < 63 && me + 1 < (int)fibers.size()) r = shfl_slot[me + 1];
    else if (ctrl >= 0 && ctrl < 0x100) {                   // quad_perm: lane i of each group of four reads lane (ctrl >> 2i) & 3 of it
        const int srcl = (me & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
        if (srcl < (int)fibers.size()) r = shfl_slot[srcl];
    }
    else if (ctrl != 0x138 && ctrl != 0x130) abort();
    wave_sync();
    return r;
}
void *dyn_lds() { return lds_ptr; }

void launch(const std::function<void()> &body, dim3 grid, dim3 block, size_t shmem)
{
    std::lock_guard<std::mutex> g(big_lock);
    const int nthreads = (int)(block.x * block.y * block.z);
    if ((int)fibers.size() < nthreads) {
        fibers.resize(nthreads);
        while ((int)stacks.size() < nthreads) stacks.push_back((char *)malloc(kStack));
    }
    char *lds = shmem ? (char *)guarded_alloc(shmem, 16) : nullptr;
    lds_ptr = lds;
    cur_body = &body;
    g_blockDim = {block.x, block.y, block.z};
    g_gridDim = {grid.x, grid.y, grid.z};
    const int nwaves = (nthreads + 63) / 64;
    for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
        g_blockIdx = {bx, by, bz};
        rfl_reset();
        for (int t = 0; t < nthreads; t++) {
            Fiber &f = fibers[t];
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = stacks[t];
            f.ctx.uc_stack.ss_size = kStack;
            f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, trampoline, 0);
            f.state = READY;
            f.tid = {(unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y)};
        }
        for (;;) {
            int alive = 0;
            for (int w = 0; w < nwaves; w++) {
                const int lo = w * 64, hi = std::min(nthreads, lo + 64);
                for (;;) {
                    for (int t = lo; t < hi; t++) {
                        if (fibers[t].state != READY) continue;
                        cur = t;
                        g_threadIdx = fibers[t].tid;
                        swapcontext(&sched_ctx, &fibers[t].ctx);
                    }
                    // a wave-level rendezvous completes when every lane that is still running has arrived; lanes
                    // that already wait at a block barrier (they left the enclosing loop / branch) are not
                    // "active" for it, like lanes masked off in a real wave
                    int at_wave = 0, live = 0, parked = 0;
                    for (int t = lo; t < hi; t++) {
                        if (fibers[t].state == DONE) continue;
                        live++;
                        at_wave += fibers[t].state == AT_WAVE;
                        parked += fibers[t].state == AT_BLOCK;
                    }
                    if (live && at_wave && at_wave + parked == live) {
                        for (int t = lo; t < hi; t++) if (fibers[t].state == AT_WAVE) fibers[t].state = READY;
                        continue;
                    }
                    if (at_wave) {
                        fprintf(stderr, "hipemu: divergent wave barrier in block (%u,%u,%u)\n", bx, by, bz);
                        abort();
                    }
                    alive += live;
                    break;
                }
            }
            if (!alive) break;
            for (int t = 0; t < nthreads; t++) if (fibers[t].state == AT_BLOCK) fibers[t].state = READY;
        }
    }
    cur_body = nullptr;
    if (lds) guarded_free(lds);
    lds_ptr = nullptr;
}

} // namespace hipemu

using namespace hipemu;

hipError_t hipMalloc(void **p, size_t n)
{
    *p = guarded_alloc(n ? n : 1, 16);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p) { guarded_free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t)
{
    for (size_t y = 0; y < h; y++) memcpy((char *)d + y * dp, (const char *)s + y * sp, w);
    return hipSuccess;
}
// two emulated devices (one address space): enough to exercise per-device state such as the context's device check
static thread_local int g_device = 0;
hipError_t hipSetDevice(int d) { if (d < 0 || d > 1) return hipErrorInvalidValue; g_device = d; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = g_device; return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = 2; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *prop, int) { memset(prop, 0, sizeof(*prop)); prop->multiProcessorCount = 256; prop->warpSize = 64; snprintf(prop->name, sizeof(prop->name), "emulated gfx950"); return hipSuccess; }
// no PCI device behind an emulated GPU: callers that look for sysfs placement find none
hipError_t hipDeviceGetPCIBusId(char *buf, int len, int) { if (len > 0) buf[0] = 0; return hipErrorInvalidValue; }
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorName(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipError(emulated)"; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(1); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
struct ihipEvent_t { std::chrono::steady_clock::time_point t; };
hipError_t hipEventCreate(hipEvent_t *e) { *e = new ihipEvent_t(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new ihipEvent_t(); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { return e ? hipSuccess : hipErrorInvalidHandle; }   // (the runtime refuses a null event: ADVICE r4)
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus *st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *) { return hipErrorNotSupported; }
hipError_t hipGraphInstantiate(hipGraphExec_t *, hipGraph_t, void *, void *, size_t) { return hipErrorNotSupported; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
