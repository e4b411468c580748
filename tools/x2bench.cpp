// tools/x2bench.cpp — tuning aid (not product code): times the scaled YUV -> RGB / YUV contexts of the C ABI on one GPU
// without Python.  Frames rotate through a set larger than L2 + Infinity Cache; HIP-event timing on one stream.
//   x2bench [frames_per_launch=32] [launches=40] [case substring]
// Prints one line per case: kernel, microseconds per launch and per frame, algorithmic GB/s and the fraction of 8 TB/s.
// It also cross-checks the batched launch against the one-frame-per-call path of the same context (same library, other
// kernel when the strip kernel is active): a CRC of every output frame must agree.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <string>
#include <vector>
#include "../include/gmat_hip.h"

// The set-up of a case (64 hipMallocs, uploads, memsets) leaves the GPU idle for tens of milliseconds and its clocks drop; six
// warm-up launches (under a millisecond) do not bring them back, and the first cases of round 2's tables read 8-10 % slow
// against bench.py's 40 ms pre-warm.  Every case now runs its own launches for this long, untimed, first.
static double prewarm_ms() { const char *e = getenv("X2BENCH_PREWARM_MS"); return e ? atof(e) : 30.0; }
template <class F, class S> static void prewarm(F &&launch, S &&sync)
{
    const auto t0 = std::chrono::steady_clock::now();
    const double want = prewarm_ms();
    int i = 0;
    while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() < want) { for (int k = 0; k < 8; k++) launch(i++); sync(); }
}

#define CK(x) do { int _r = (x); if (_r < 0) { fprintf(stderr, "error %d at %s:%d: %s\n", _r, __FILE__, __LINE__, #x); exit(1); } } while (0)

static uint32_t lcg_state = 12345;
static void fill_lcg(std::vector<uint8_t> &v) { for (auto &b : v) { lcg_state = lcg_state * 1664525u + 1013904223u; b = (uint8_t)(lcg_state >> 24); } }
static uint32_t adler(const uint8_t *p, size_t n)
{
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < n; i++) { a = (a + p[i]) % 65521u; b = (b + a) % 65521u; }
    return (b << 16) | a;
}

struct Fmt { const char *name; int id; };
// row pitch alignment of the frames of a case: 1 = tight (the historical tables), 256 = AVHWFramesContext's
// (libavutil/hwcontext_cuda.c:145-157) — used by the "any:" cases, whose odd widths (1366, 854) have no dword pitch when tight
static int g_align = 1;
static int al(int v) { return (v + g_align - 1) / g_align * g_align; }
static size_t frame_bytes(int fmt, int w, int h)
{
    if (g_align > 1) {
        switch (fmt) {
        case GMAT_PIX_FMT_NV12: return (size_t)al(w) * h * 3 / 2;
        case GMAT_PIX_FMT_YUV420P: return (size_t)al(w) * h + 2 * (size_t)al(w / 2) * (h / 2);
        case GMAT_PIX_FMT_RGB24: case GMAT_PIX_FMT_BGR24: return (size_t)al(3 * w) * h;
        case GMAT_PIX_FMT_RGBA: case GMAT_PIX_FMT_BGRA: return (size_t)al(4 * w) * h;
        default: break;
        }
    }
    switch (fmt) {
    case GMAT_PIX_FMT_NV12: case GMAT_PIX_FMT_YUV420P: return (size_t)w * h * 3 / 2;
    case GMAT_PIX_FMT_P010LE: case GMAT_PIX_FMT_YUV420P10LE: case GMAT_PIX_FMT_YUV420P16LE: return (size_t)w * h * 3;
    case GMAT_PIX_FMT_RGB24: case GMAT_PIX_FMT_BGR24: case GMAT_PIX_FMT_YUV444P: return (size_t)w * h * 3;
    case GMAT_PIX_FMT_RGBA: case GMAT_PIX_FMT_BGRA: return (size_t)w * h * 4;
    case GMAT_PIX_FMT_RGBPF32LE: return (size_t)w * h * 12;        // three stacked planes of floats
    case GMAT_PIX_FMT_YUV444P16LE: return (size_t)w * h * 6;
    case GMAT_PIX_FMT_P016LE: return (size_t)w * h * 3;
    case GMAT_PIX_FMT_RGBA64LE: case GMAT_PIX_FMT_BGRA64LE: return (size_t)w * h * 8;
    default: return 0;
    }
}
static void frame_ptrs(uint8_t *b, int fmt, int w, int h, uint8_t *p[4], int s[4])
{
    p[0] = p[1] = p[2] = p[3] = nullptr; s[0] = s[1] = s[2] = s[3] = 0;
    if (g_align > 1) {
        switch (fmt) {
        case GMAT_PIX_FMT_NV12: p[0] = b; p[1] = b + (size_t)al(w) * h; s[0] = s[1] = al(w); return;
        case GMAT_PIX_FMT_YUV420P: p[0] = b; p[1] = b + (size_t)al(w) * h; p[2] = p[1] + (size_t)al(w / 2) * (h / 2); s[0] = al(w); s[1] = s[2] = al(w / 2); return;
        case GMAT_PIX_FMT_RGB24: case GMAT_PIX_FMT_BGR24: p[0] = b; s[0] = al(3 * w); return;
        case GMAT_PIX_FMT_RGBA: case GMAT_PIX_FMT_BGRA: p[0] = b; s[0] = al(4 * w); return;
        default: break;
        }
    }
    switch (fmt) {
    case GMAT_PIX_FMT_NV12: p[0] = b; p[1] = b + (size_t)w * h; s[0] = w; s[1] = w; break;
    case GMAT_PIX_FMT_P010LE: case GMAT_PIX_FMT_P016LE: p[0] = b; p[1] = b + (size_t)w * h * 2; s[0] = 2 * w; s[1] = 2 * w; break;
    case GMAT_PIX_FMT_YUV444P16LE: p[0] = b; p[1] = b + (size_t)w * h * 2; p[2] = p[1] + (size_t)w * h * 2; s[0] = s[1] = s[2] = 2 * w; break;
    case GMAT_PIX_FMT_YUV420P10LE: case GMAT_PIX_FMT_YUV420P16LE: p[0] = b; p[1] = b + (size_t)w * h * 2; p[2] = p[1] + (size_t)(w / 2) * (h / 2) * 2; s[0] = 2 * w; s[1] = s[2] = w; break;
    case GMAT_PIX_FMT_YUV420P: p[0] = b; p[1] = b + (size_t)w * h; p[2] = p[1] + (size_t)(w / 2) * (h / 2); s[0] = w; s[1] = s[2] = w / 2; break;
    case GMAT_PIX_FMT_YUV444P: p[0] = b; p[1] = b + (size_t)w * h; p[2] = p[1] + (size_t)w * h; s[0] = s[1] = s[2] = w; break;
    case GMAT_PIX_FMT_RGB24: case GMAT_PIX_FMT_BGR24: p[0] = b; s[0] = 3 * w; break;
    case GMAT_PIX_FMT_RGBA64LE: case GMAT_PIX_FMT_BGRA64LE: p[0] = b; s[0] = 8 * w; break;
    default: p[0] = b; s[0] = 4 * w; break;
    }
}

static void run_case(const char *label, int sf, int sw, int sh, int df, int dw, int dh, int flags, int NF, int launches, int verify)
{
    g_align = (strncmp(label, "any:", 4) == 0 || strncmp(label, "thumb:", 6) == 0 || strncmp(label, "deep:", 5) == 0 || strncmp(label, "rgbsrc:", 7) == 0) ? 256 : 1;
    const size_t sb = frame_bytes(sf, sw, sh), db = frame_bytes(df, dw, dh);
    // rotate NSETS frame sets (two: > 256 MiB together at 4K x 32, yet a third of the source set can stay in the 256 MB Infinity Cache between
    // launches; X2BENCH_SETS=8 is bench.py's regime, every frame from HBM — the headline's band-height A/B came out differently in the two, FINDINGS R4-rows)
    const int NSETS = getenv("X2BENCH_SETS") ? std::max(2, std::min(16, atoi(getenv("X2BENCH_SETS")))) : 2;
    const int NSET = NSETS * NF;
    std::vector<uint8_t *> src(NSET), dst(NSET);
    std::vector<uint8_t> host(sb);
    for (int i = 0; i < NSET; i++) {
        CK(gmat_malloc(&src[i], sb)); CK(gmat_malloc(&dst[i], db));
        if (i < 4) { fill_lcg(host); CK(gmat_memcpy_h2d(src[i], host.data(), sb)); }
        else CK(gmat_memcpy_h2d(src[i], host.data(), sb));          // same bytes; content does not matter for timing
        CK(gmat_memset(dst[i], 0, db));
    }
    GmatSwsContext *c = gmat_sws_getContext(sw, sh, sf, dw, dh, df, flags, nullptr);
    if (!c) { fprintf(stderr, "%s: no context\n", label); exit(1); }
    void *stream = nullptr; CK(gmat_stream_create(&stream));
    gmat_sws_setStream(c, stream);
    std::vector<const uint8_t *> sp((size_t)NSET * 4); std::vector<uint8_t *> dp((size_t)NSET * 4);
    int ss[4], ds[4];
    for (int i = 0; i < NSET; i++) {
        uint8_t *p[4];
        frame_ptrs(src[i], sf, sw, sh, p, ss); for (int k = 0; k < 4; k++) sp[(size_t)i * 4 + k] = p[k];
        frame_ptrs(dst[i], df, dw, dh, p, ds); for (int k = 0; k < 4; k++) dp[(size_t)i * 4 + k] = p[k];
    }
    // X2BENCH_CALL_STREAMS=n: calls round-robin over n streams of the CALLER's (one library call per frame, nothing else added) — what
    // a per-frame caller with n frames in flight sees; joined to `stream` by events around the timed region
    const int ncs = getenv("X2BENCH_CALL_STREAMS") ? std::max(1, std::min(8, atoi(getenv("X2BENCH_CALL_STREAMS")))) : 1;
    std::vector<void *> xs(ncs, stream), xev(ncs, nullptr);
    for (int k = 1; k < ncs; k++) { CK(gmat_stream_create(&xs[k])); CK(gmat_event_create(&xev[k])); }
    int callNo = 0;
    auto launch = [&](int set) {
        void *streams[1] = {xs[callNo++ % ncs]};
        CK(gmat_sws_scale_batch(c, NF, sp.data() + (size_t)set * NF * 4, ss, dp.data() + (size_t)set * NF * 4, ds, streams, 1, 0));
    };
    auto sync_all = [&] { for (void *x : xs) CK(gmat_stream_sync(x)); };
    for (int i = 0; i < 3 * NSETS; i++) launch(i % NSETS);
    sync_all();
    prewarm([&](int i) { launch(i % NSETS); }, sync_all);
    const std::string kname = gmat_sws_lastKernel(c);
    void *timer = nullptr; CK(gmat_timer_create(&timer));
    float best = 1e30f, sum = 0;
    const int REPS = 3;
    for (int r = 0; r < REPS; r++) {
        sync_all();
        CK(gmat_timer_begin(timer, stream));
        if (ncs > 1) { void *e0 = nullptr; CK(gmat_event_create(&e0)); CK(gmat_event_record(e0, stream)); for (int k = 1; k < ncs; k++) CK(gmat_stream_wait_event(xs[k], e0)); gmat_event_destroy(e0); }
        for (int i = 0; i < launches; i++) launch(i % NSETS);
        for (int k = 1; k < ncs; k++) { CK(gmat_event_record(xev[k], xs[k])); CK(gmat_stream_wait_event(stream, xev[k])); }
        CK(gmat_timer_end(timer, stream));
        float ms = 0; CK(gmat_timer_elapsed_ms(timer, &ms));
        best = ms < best ? ms : best; sum += ms;
    }
    const double usLaunch = best * 1e3 / launches, usFrame = usLaunch / NF;
    const int keepAlign = g_align; g_align = 1;
    const size_t algBytes = frame_bytes(sf, sw, sh) + frame_bytes(df, dw, dh);      // algorithmic bytes: the tight sizes, whatever the pitch
    g_align = keepAlign;
    const double gbs = (double)algBytes / usFrame / 1e3;
    if (getenv("X2BENCH_JSON"))
        printf("{\"case\": \"%s\", \"kernel\": \"%s\", \"frames_per_launch\": %d, \"us_per_launch\": %.2f, \"us_per_frame\": %.3f, "
               "\"algorithmic_bytes_per_frame\": %zu, \"achieved_GBps\": %.1f, \"frac\": %.4f, \"Gpix/s\": %.1f}\n",
               label, kname.c_str(), NF, usLaunch, usFrame, algBytes, gbs, gbs / 8000.0, (double)sw * sh / usFrame / 1e3);
    else
    printf("%-34s %-26s %8.1f us/launch %7.3f us/frame %8.1f GB/s  frac %.3f  %7.1f Gpix/s  (avg %.1f us/launch)\n", label, kname.c_str(),
           usLaunch, usFrame, gbs, gbs / 8000.0, (double)sw * sh / usFrame / 1e3, sum / REPS * 1e3 / launches);
    fflush(stdout);
    if (verify) {
        // batched result of the first 4 (distinct) frames vs one frame per call
        std::vector<uint8_t> a(db), b(db);
        std::vector<uint32_t> crcBatch(4);
        const int nv = NF < 4 ? NF : 4;
        launch(0); sync_all();
        for (int i = 0; i < nv; i++) { CK(gmat_memcpy_d2h(a.data(), dst[i], db)); crcBatch[i] = adler(a.data(), db); CK(gmat_memset(dst[i], 0, db)); }
        int bad = 0;
        for (int i = 0; i < nv; i++) {
            int r = gmat_sws_scale(c, sp.data() + (size_t)i * 4, ss, 0, sh, dp.data() + (size_t)i * 4, ds);
            if (r < 0) { fprintf(stderr, "single-frame call failed %d\n", r); exit(1); }
            CK(gmat_stream_sync(stream));
            CK(gmat_memcpy_d2h(b.data(), dst[i], db));
            if (adler(b.data(), db) != crcBatch[i]) bad++;
        }
        printf("    verify vs one-frame path (%s): %s\n", gmat_sws_lastKernel(c), bad ? "MISMATCH" : "identical");
        if (bad) exit(2);
    }
    gmat_timer_destroy(timer);
    gmat_sws_freeContext(c);
    for (int k = 1; k < ncs; k++) { gmat_event_destroy(xev[k]); gmat_stream_destroy(xs[k]); }
    gmat_stream_destroy(stream);
    for (int i = 0; i < NSET; i++) { gmat_free(src[i]); gmat_free(dst[i]); }
}

// one frame per launch of a pixel filter on packed 4K frames: BASELINE configs[3] (rotate 90 + hflip + 3x3 smooth as one kernel)
// and its parts.  GMAT_NO_SMOOTH121=1 selects the general 3x3 kernel instead of the separable one.
static int g_op_frames = 1;
static void run_op(const char *label, int op, int w, int h, int bpp, int launches, int poolPitch = 0)
{
    const size_t nb = (size_t)w * h * bpp;
    const int NSET = 16;
    std::vector<uint8_t *> src(NSET), dst(NSET);
    std::vector<uint8_t> host(nb);
    fill_lcg(host);
    for (int i = 0; i < NSET; i++) {
        CK(gmat_malloc(&src[i], nb)); CK(gmat_malloc(&dst[i], nb + (size_t)w * 512));
        CK(gmat_memcpy_h2d(src[i], host.data(), nb)); CK(gmat_memset(dst[i], 0, nb));
    }
    // X2BENCH_OP_DST_ALIGN=n: the destination pitch of the transposing ops (h * bpp bytes: 6480 for a 4K rgb24 frame, not a multiple of a
    // 128-byte line) rounded up to a multiple of n — what a frame of a hardware pool has
    const int tal = poolPitch ? 256 : getenv("X2BENCH_OP_DST_ALIGN") ? std::max(1, atoi(getenv("X2BENCH_OP_DST_ALIGN"))) : 1;      // (a pool frame: rows aligned to 256, gframes.cpp)
    const int tpitch = (h * bpp + tal - 1) / tal * tal;
    void *stream = nullptr; CK(gmat_stream_create(&stream));
    // X2BENCH_OP_STREAMS=n: launches round-robin over n streams (n - 1 extra ones joined to `stream` by events around the timed
    // region) — what a caller with n frames in flight sees; 1 (default) = filter_frame()'s one frame per call on one stream
    const int nstreams = getenv("X2BENCH_OP_STREAMS") ? std::max(1, std::min(8, atoi(getenv("X2BENCH_OP_STREAMS")))) : 1;
    std::vector<void *> xs(nstreams, stream);
    for (int k = 1; k < nstreams; k++) CK(gmat_stream_create(&xs[k]));
    const int m[9] = {1, 2, 1, 2, 4, 2, 1, 2, 1};
    // NFOP frames per launch (x2bench <nf> ...): gmat_op_batch over NFOP consecutive frame sets — what the queued filter form does
    const int NFOP = std::max(1, std::min(g_op_frames, NSET));
    auto launch = [&](int i) {
        void *stream = xs[i % nstreams];
        if (NFOP > 1 && op <= 4) {
            const uint8_t *sp[16]; uint8_t *dp[16];
            for (int k = 0; k < NFOP; k++) { sp[k] = src[(i * NFOP + k) % NSET]; dp[k] = dst[(i * NFOP + k) % NSET]; }
            const int opid = op == 0 ? GMAT_OP_ROTATE_FLIP_SMOOTH : op == 1 ? GMAT_OP_SMOOTH3X3 : op == 2 ? GMAT_OP_TRANSPOSE : op == 3 ? GMAT_OP_FLIP : GMAT_OP_MEDIAN3X3;
            const int ods = (op == 0 || op == 2) ? tpitch : w * bpp;
            CK(gmat_op_batch(opid, NFOP, sp, w * bpp, dp, ods, w, h, bpp, op == 3 ? 1 : 0, stream));
            return;
        }
        if (NFOP > 1 && op <= 7) {                           // the arbitrary-angle rotate over a frame table
            const uint8_t *sp[16]; uint8_t *dp[16];
            for (int k = 0; k < NFOP; k++) { sp[k] = src[(i * NFOP + k) % NSET]; dp[k] = dst[(i * NFOP + k) % NSET]; }
            CK(gmat_rotate2_batch(NFOP, sp, w * bpp, dp, w * bpp, w, h, w, h, bpp, 17.0 * 3.14159265358979323846 / 180.0, op == 5 ? 1 : op == 6 ? 2 : 0, 0.0, 0.0,
                                  nullptr, stream));
            return;
        }
        switch (op) {
        case 0: CK(gmat_rotate_flip_smooth(src[i], w * bpp, dst[i], tpitch, w, h, bpp, stream)); break;
        case 1: CK(gmat_smooth3x3(src[i], w * bpp, dst[i], w * bpp, w, h, bpp, m, 1.0f / 16, 0.0f, stream)); break;
        case 2: CK(gmat_transpose(src[i], w * bpp, dst[i], tpitch, w, h, bpp, 0, stream)); break;
        case 3: CK(gmat_flip(src[i], w * bpp, dst[i], w * bpp, w, h, bpp, 1, stream)); break;
        case 4: CK(gmat_median3x3(src[i], w * bpp, dst[i], w * bpp, w, h, bpp, stream)); break;
        case 5: CK(gmat_rotate(src[i], w * bpp, dst[i], w * bpp, w, h, w, h, bpp, 17.0 * 3.14159265358979323846 / 180.0, 1, nullptr, stream)); break;
        case 6: CK(gmat_rotate2(src[i], w * bpp, dst[i], w * bpp, w, h, w, h, bpp, 17.0 * 3.14159265358979323846 / 180.0, 2, 0.0, 0.0, nullptr, stream)); break;
        case 7: CK(gmat_rotate(src[i], w * bpp, dst[i], w * bpp, w, h, w, h, bpp, 17.0 * 3.14159265358979323846 / 180.0, 0, nullptr, stream)); break;
        case 8: CK(gmat_crop(src[i], w * bpp, dst[i], (w - 64) * bpp, 32, 16, w - 64, h - 32, bpp, stream)); break;   // the frame less a 32 / 16 pixel border
        }
    };
    auto sync_all = [&] { for (void *x : xs) CK(gmat_stream_sync(x)); };
    for (int i = 0; i < NSET; i++) launch(i);
    sync_all();
    prewarm([&](int i) { launch(i % NSET); }, sync_all);
    void *timer = nullptr; CK(gmat_timer_create(&timer));
    std::vector<void *> ev(nstreams, nullptr);
    for (int k = 1; k < nstreams; k++) CK(gmat_event_create(&ev[k]));
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        sync_all();
        CK(gmat_timer_begin(timer, stream));
        if (nstreams > 1) { void *e0 = nullptr; CK(gmat_event_create(&e0)); CK(gmat_event_record(e0, stream)); for (int k = 1; k < nstreams; k++) CK(gmat_stream_wait_event(xs[k], e0)); gmat_event_destroy(e0); }
        for (int i = 0; i < launches; i++) launch(i % NSET);
        for (int k = 1; k < nstreams; k++) { CK(gmat_event_record(ev[k], xs[k])); CK(gmat_stream_wait_event(stream, ev[k])); }
        CK(gmat_timer_end(timer, stream));
        float ms = 0; CK(gmat_timer_elapsed_ms(timer, &ms));
        best = ms < best ? ms : best;
    }
    const int fpl = op <= 7 ? NFOP : 1;                  // (a crop is a copy: one frame a call)
    const double us = best * 1e3 / launches / fpl, gbs = 2.0 * nb / us / 1e3;
    if (getenv("X2BENCH_JSON"))
        printf("{\"case\": \"%s\", \"frames_per_launch\": %d, \"us_per_frame\": %.2f, \"algorithmic_bytes_per_frame\": %zu, "
               "\"achieved_GBps\": %.1f, \"frac\": %.4f}\n", label, fpl, us, 2 * nb, gbs, gbs / 8000.0);
    else
    printf("%-34s %8.2f us/frame %8.1f GB/s  frac %.3f\n", label, us, gbs, gbs / 8000.0);
    fflush(stdout);
    gmat_timer_destroy(timer);
    for (int k = 1; k < nstreams; k++) { gmat_event_destroy(ev[k]); gmat_stream_destroy(xs[k]); }
    gmat_stream_destroy(stream);
    for (int i = 0; i < NSET; i++) { gmat_free(src[i]); gmat_free(dst[i]); }
}

int main(int argc, char **argv)
{
    const int NF = argc > 1 ? atoi(argv[1]) : 32, launches = argc > 2 ? atoi(argv[2]) : 40;
    const char *only = argc > 3 ? argv[3] : "";
    const int verify = getenv("X2BENCH_VERIFY") ? atoi(getenv("X2BENCH_VERIFY")) : 1;
    if (gmat_device_count() < 1) { fprintf(stderr, "no GPU\n"); return 1; }
    CK(gmat_set_device(0));
    struct Case { const char *label; int sf, sw, sh, df, dw, dh, flags; };
    const Case cases[] = {
        {"nv12 4K->1080p rgb24 bicubic", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 1920, 1080, GMAT_SWS_BICUBIC},
        {"yuv420p 4K->1080p rgb24 bicubic", GMAT_PIX_FMT_YUV420P, 3840, 2160, GMAT_PIX_FMT_RGB24, 1920, 1080, GMAT_SWS_BICUBIC},
        {"nv12 4K->1080p rgba bicubic", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGBA, 1920, 1080, GMAT_SWS_BICUBIC},
        {"nv12 4K->1080p rgb24 bilinear", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 1920, 1080, GMAT_SWS_BILINEAR},
        {"nv12 4K->1080p nv12 bicubic", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_SWS_BICUBIC},
        {"nv12 4K->1080p yuv420p bicubic", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_YUV420P, 1920, 1080, GMAT_SWS_BICUBIC},
        {"yuv420p 4K->1080p nv12 bicubic", GMAT_PIX_FMT_YUV420P, 3840, 2160, GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_SWS_BICUBIC},
        {"yuv420p 4K->1080p yuv420p bicubic", GMAT_PIX_FMT_YUV420P, 3840, 2160, GMAT_PIX_FMT_YUV420P, 1920, 1080, GMAT_SWS_BICUBIC},
        {"nv12 4K->720p nv12 bicubic (3:1)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_NV12, 1280, 720, GMAT_SWS_BICUBIC},
        {"nv12 4K->720p rgb24 bicubic (3:1)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 1280, 720, GMAT_SWS_BICUBIC},
        {"nv12 4K->540p nv12 bicubic (4:1)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_NV12, 960, 540, GMAT_SWS_BICUBIC},
        {"nv12 4K->540p rgb24 bicubic (4:1)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 960, 540, GMAT_SWS_BICUBIC},
        {"nv12 1080p->720p nv12 bicubic (3:2)", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_NV12, 1280, 720, GMAT_SWS_BICUBIC},
        {"nv12 1080p->720p rgb24 bicubic (3:2)", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_RGB24, 1280, 720, GMAT_SWS_BICUBIC},
        {"nv12 1080p->540p rgb24 bicubic", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_RGB24, 960, 540, GMAT_SWS_BICUBIC},
        {"rgb24 4K->1080p nv12 bicubic (scale)", GMAT_PIX_FMT_RGB24, 3840, 2160, GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_SWS_BICUBIC},
        {"rgb24 4K->4K nv12 (convert)", GMAT_PIX_FMT_RGB24, 3840, 2160, GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_SWS_BICUBIC},
        {"rgb24 1080p->1080p yuv420p (convert)", GMAT_PIX_FMT_RGB24, 1920, 1080, GMAT_PIX_FMT_YUV420P, 1920, 1080, GMAT_SWS_BICUBIC},
        {"rgb24 4K->1080p rgb24 bicubic", GMAT_PIX_FMT_RGB24, 3840, 2160, GMAT_PIX_FMT_RGB24, 1920, 1080, GMAT_SWS_BICUBIC},
        {"rgb24 4K->1080p bgra bilinear", GMAT_PIX_FMT_RGB24, 3840, 2160, GMAT_PIX_FMT_BGRA, 1920, 1080, GMAT_SWS_BILINEAR},
        {"p010 4K->1080p p010 bicubic", GMAT_PIX_FMT_P010LE, 3840, 2160, GMAT_PIX_FMT_P010LE, 1920, 1080, GMAT_SWS_BICUBIC},
        {"yuv420p10le 4K->1080p yuv420p10le", GMAT_PIX_FMT_YUV420P10LE, 3840, 2160, GMAT_PIX_FMT_YUV420P10LE, 1920, 1080, GMAT_SWS_BICUBIC},
        {"nv12 4K->1080p p010 bicubic", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_P010LE, 1920, 1080, GMAT_SWS_BICUBIC},
        {"p010 4K->1080p nv12 bicubic", GMAT_PIX_FMT_P010LE, 3840, 2160, GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_SWS_BICUBIC},
        {"land: p010 4K->1080p nv12 (dup)", GMAT_PIX_FMT_P010LE, 3840, 2160, GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_SWS_BICUBIC},
        {"land: yuv420p 4K->720p yuv420p bicubic", GMAT_PIX_FMT_YUV420P, 3840, 2160, GMAT_PIX_FMT_YUV420P, 1280, 720, GMAT_SWS_BICUBIC},
        {"land: nv12 1080p->360p nv12 bicubic", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_NV12, 640, 360, GMAT_SWS_BICUBIC},
        {"land: nv12 4K->1080p yuv444p bicubic", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_YUV444P, 1920, 1080, GMAT_SWS_BICUBIC},
        {"land: yuv420p 1080p->720p yuv420p bicubic", GMAT_PIX_FMT_YUV420P, 1920, 1080, GMAT_PIX_FMT_YUV420P, 1280, 720, GMAT_SWS_BICUBIC},
        {"land: yuv420p 4K->540p rgb24 bicubic", GMAT_PIX_FMT_YUV420P, 3840, 2160, GMAT_PIX_FMT_RGB24, 960, 540, GMAT_SWS_BICUBIC},
        {"land: nv12 4K->1440p rgb24 bicubic", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 2560, 1440, GMAT_SWS_BICUBIC},
        {"land: nv12 4K->1440p nv12 bicubic", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_NV12, 2560, 1440, GMAT_SWS_BICUBIC},
        {"nv12 1080p->4K nv12 bicubic (up)", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_SWS_BICUBIC},
        {"land: yuv420p 1080p->4K yuv420p bicubic", GMAT_PIX_FMT_YUV420P, 1920, 1080, GMAT_PIX_FMT_YUV420P, 3840, 2160, GMAT_SWS_BICUBIC},
        {"land: nv12 1080p->4K nv12 bilinear", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_SWS_BILINEAR},
        {"land: (old label) nv12 1080p->4K nv12 bicubic", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_SWS_BICUBIC},
        {"land: p010 4K->1080p p010 lanczos", GMAT_PIX_FMT_P010LE, 3840, 2160, GMAT_PIX_FMT_P010LE, 1920, 1080, GMAT_SWS_LANCZOS},
        {"land: nv12 4K->1080p nv12 lanczos", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_SWS_LANCZOS},
        {"land: nv12 4K->1080p rgb24 lanczos", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 1920, 1080, GMAT_SWS_LANCZOS},
        {"nv12 1080p->1080p rgb24 convert", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_RGB24, 1920, 1080, GMAT_SWS_BICUBIC},
        // format_cuda / CSwscale's pair (the tensor a network reads): nv12 -> planar float RGB, value = u8 / 255
        {"nv12 1080p->1080p rgbpf32 convert", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_RGBPF32LE, 1920, 1080, GMAT_SWS_BICUBIC},
        {"nv12 4K->4K rgbpf32 convert", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGBPF32LE, 3840, 2160, GMAT_SWS_BICUBIC},
        {"rgbpf32 1080p->1080p nv12 convert", GMAT_PIX_FMT_RGBPF32LE, 1920, 1080, GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_SWS_BICUBIC},
        {"rgbpf32 4K->4K nv12 convert", GMAT_PIX_FMT_RGBPF32LE, 3840, 2160, GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_SWS_BICUBIC},
        // the lossless re-layouts of SURVEY 8a row 16 (run on request: "relayout")
        {"relayout: nv12 4K->yuv420p", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_YUV420P, 3840, 2160, GMAT_SWS_BICUBIC},
        {"relayout: yuv420p 4K->nv12", GMAT_PIX_FMT_YUV420P, 3840, 2160, GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_SWS_BICUBIC},
        {"relayout: rgb24 4K->bgr24", GMAT_PIX_FMT_RGB24, 3840, 2160, GMAT_PIX_FMT_BGR24, 3840, 2160, GMAT_SWS_BICUBIC},
        {"relayout: rgb24 4K->rgba", GMAT_PIX_FMT_RGB24, 3840, 2160, GMAT_PIX_FMT_RGBA, 3840, 2160, GMAT_SWS_BICUBIC},
        {"relayout: rgba 4K->rgb24", GMAT_PIX_FMT_RGBA, 3840, 2160, GMAT_PIX_FMT_RGB24, 3840, 2160, GMAT_SWS_BICUBIC},
        {"relayout: nv12 4K->p010", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_P010LE, 3840, 2160, GMAT_SWS_BICUBIC},
        {"relayout: nv12 4K->yuv444p", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_YUV444P, 3840, 2160, GMAT_SWS_BICUBIC},
        // packed-RGB sources away from 2 : 1 (the frames a network writes, scaled for an encoder or a display): "rgbsrc:" cases run when the filter names them
        {"rgbsrc: rgb24 1080p->720p rgb24 bicubic (3:2)", GMAT_PIX_FMT_RGB24, 1920, 1080, GMAT_PIX_FMT_RGB24, 1280, 720, GMAT_SWS_BICUBIC},
        {"rgbsrc: rgb24 1080p->720p nv12 bicubic (3:2)", GMAT_PIX_FMT_RGB24, 1920, 1080, GMAT_PIX_FMT_NV12, 1280, 720, GMAT_SWS_BICUBIC},
        {"rgbsrc: rgb24 4K->720p nv12 bicubic (3:1)", GMAT_PIX_FMT_RGB24, 3840, 2160, GMAT_PIX_FMT_NV12, 1280, 720, GMAT_SWS_BICUBIC},
        {"rgbsrc: rgb24 4K->1600x900 rgb24 bicubic (2.4:1)", GMAT_PIX_FMT_RGB24, 3840, 2160, GMAT_PIX_FMT_RGB24, 1600, 900, GMAT_SWS_BICUBIC},
        {"rgbsrc: rgb24 720p->1080p rgb24 bicubic (2:3 up)", GMAT_PIX_FMT_RGB24, 1280, 720, GMAT_PIX_FMT_RGB24, 1920, 1080, GMAT_SWS_BICUBIC},
        {"rgbsrc: rgb24 720p->4K rgb24 bicubic (1:3 up)", GMAT_PIX_FMT_RGB24, 1280, 720, GMAT_PIX_FMT_RGB24, 3840, 2160, GMAT_SWS_BICUBIC},
        {"rgbsrc: rgb24 1080p->1600x900 rgb24 lanczos (1.2:1)", GMAT_PIX_FMT_RGB24, 1920, 1080, GMAT_PIX_FMT_RGB24, 1600, 900, GMAT_SWS_LANCZOS},
        {"rgbsrc: rgb24 720p->1080p nv12 bicubic (2:3 up)", GMAT_PIX_FMT_RGB24, 1280, 720, GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_SWS_BICUBIC},
        {"rgbsrc: rgb24 1366x768->1080p rgb24 bicubic (a width that is not a multiple of four, up)", GMAT_PIX_FMT_RGB24, 1366, 768, GMAT_PIX_FMT_RGB24, 1920, 1080, GMAT_SWS_BICUBIC},
        {"rgbsrc: rgb24 1366x768->854x480 nv12 bicubic (widths that are not multiples of four)", GMAT_PIX_FMT_RGB24, 1366, 768, GMAT_PIX_FMT_NV12, 854, 480, GMAT_SWS_BICUBIC},
        {"rgbsrc: bgra 4K->1080p bgra bicubic (2:1)", GMAT_PIX_FMT_BGRA, 3840, 2160, GMAT_PIX_FMT_BGRA, 1920, 1080, GMAT_SWS_BICUBIC},
        {"rgbsrc: bgra 4K->1080p nv12 bicubic (2:1)", GMAT_PIX_FMT_BGRA, 3840, 2160, GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_SWS_BICUBIC},
        {"rgbsrc: bgra 1080p->720p nv12 bicubic (3:2)", GMAT_PIX_FMT_BGRA, 1920, 1080, GMAT_PIX_FMT_NV12, 1280, 720, GMAT_SWS_BICUBIC},
        {"rgbsrc: bgra 1080p->720p rgb24 bicubic (3:2)", GMAT_PIX_FMT_BGRA, 1920, 1080, GMAT_PIX_FMT_RGB24, 1280, 720, GMAT_SWS_BICUBIC},
        {"rgbsrc: bgra 1080p->720p bgra bicubic (3:2)", GMAT_PIX_FMT_BGRA, 1920, 1080, GMAT_PIX_FMT_BGRA, 1280, 720, GMAT_SWS_BICUBIC},
        {"rgbsrc: rgb24 640x640->1080p rgb24 bilinear (up)", GMAT_PIX_FMT_RGB24, 640, 640, GMAT_PIX_FMT_RGB24, 1920, 1080, GMAT_SWS_BILINEAR},
        {"dst16: yuv444p16le 1080p->720p yuv444p16le bicubic (16-bit 4:4:4 at both ends)", GMAT_PIX_FMT_YUV444P16LE, 1920, 1080, GMAT_PIX_FMT_YUV444P16LE, 1280, 720, GMAT_SWS_BICUBIC},
        {"dst16: yuv444p16le 4K->1080p yuv444p16le bicubic (16-bit 4:4:4 at both ends)", GMAT_PIX_FMT_YUV444P16LE, 3840, 2160, GMAT_PIX_FMT_YUV444P16LE, 1920, 1080, GMAT_SWS_BICUBIC},
        {"dst16: p016 4K->1080p p016 bicubic", GMAT_PIX_FMT_P016LE, 3840, 2160, GMAT_PIX_FMT_P016LE, 1920, 1080, GMAT_SWS_BICUBIC},
        {"dst16: p016 1080p->720p p016 bicubic", GMAT_PIX_FMT_P016LE, 1920, 1080, GMAT_PIX_FMT_P016LE, 1280, 720, GMAT_SWS_BICUBIC},
        {"dst16: nv12 1080p->720p p016 bicubic", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_P016LE, 1280, 720, GMAT_SWS_BICUBIC},
        {"dst16: nv12 1080p->1080p rgba64 convert (yuv2rgb_cuda's 64-bit output)", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_RGBA64LE, 1920, 1080, GMAT_SWS_BICUBIC},
        {"dst16: nv12 1080p->1080p rgba64 convert, point", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_RGBA64LE, 1920, 1080, GMAT_SWS_POINT},
        {"dst16: nv12 4K->1080p rgba64 bicubic", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGBA64LE, 1920, 1080, GMAT_SWS_BICUBIC},
        {"dst16: p016 1080p->720p bgra64 bicubic", GMAT_PIX_FMT_P016LE, 1920, 1080, GMAT_PIX_FMT_BGRA64LE, 1280, 720, GMAT_SWS_BICUBIC},
        {"dst16: p010 4K->1080p p016 bicubic", GMAT_PIX_FMT_P010LE, 3840, 2160, GMAT_PIX_FMT_P016LE, 1920, 1080, GMAT_SWS_BICUBIC},
        {"dst16: p016 720p->1080p p016 bicubic (up)", GMAT_PIX_FMT_P016LE, 1280, 720, GMAT_PIX_FMT_P016LE, 1920, 1080, GMAT_SWS_BICUBIC},
        {"dst16: p016 4K->720p p016 lanczos (3:1)", GMAT_PIX_FMT_P016LE, 3840, 2160, GMAT_PIX_FMT_P016LE, 1280, 720, GMAT_SWS_LANCZOS},
        // same-size conversions between depths and layouts (yuv2yuv_cuda's space; libswscale: the generic scaler with one-tap filters): the tile kernel's unit form
        {"yuv2yuv: p010 1080p->1080p nv12 (10 -> 8 bits, dithered)", GMAT_PIX_FMT_P010LE, 1920, 1080, GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_SWS_BICUBIC},
        {"yuv2yuv: nv12 1080p->1080p yuv420p10le", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_YUV420P10LE, 1920, 1080, GMAT_SWS_BICUBIC},
        {"yuv2yuv: yuv420p10le 4K->4K p010", GMAT_PIX_FMT_YUV420P10LE, 3840, 2160, GMAT_PIX_FMT_P010LE, 3840, 2160, GMAT_SWS_BICUBIC},
        {"yuv2yuv: p016 4K->4K nv12 (16 -> 8 bits, dithered)", GMAT_PIX_FMT_P016LE, 3840, 2160, GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_SWS_BICUBIC},
        // 16-bit 4:2:0 sources into packed 8-bit RGB at equal size (no unscaled converter in libswscale): unit_rgb_kernel
        {"deep2rgb: p010 1080p->1080p rgb24 convert", GMAT_PIX_FMT_P010LE, 1920, 1080, GMAT_PIX_FMT_RGB24, 1920, 1080, GMAT_SWS_BICUBIC},
        {"deep2rgb: yuv420p10le 1080p->1080p bgra convert", GMAT_PIX_FMT_YUV420P10LE, 1920, 1080, GMAT_PIX_FMT_BGRA, 1920, 1080, GMAT_SWS_BICUBIC},
        {"rgbsrc: yuv444p 1080p->720p yuv444p bicubic (4:4:4 at both ends)", GMAT_PIX_FMT_YUV444P, 1920, 1080, GMAT_PIX_FMT_YUV444P, 1280, 720, GMAT_SWS_BICUBIC},
        {"rgbsrc: yuv444p 4K->1080p yuv444p bicubic (4:4:4 at both ends)", GMAT_PIX_FMT_YUV444P, 3840, 2160, GMAT_PIX_FMT_YUV444P, 1920, 1080, GMAT_SWS_BICUBIC},
        {"rgbsrc: nv12 1080p->640x640 rgb24 bilinear (a network's input)", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_RGB24, 640, 640, GMAT_SWS_BILINEAR},
        {"rgbsrc: nv12 4K->640x640 rgb24 bilinear (a network's input)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 640, 640, GMAT_SWS_BILINEAR},
        {"rgbsrc: nv12 1080p->224x224 rgb24 bilinear (a classifier's input)", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_RGB24, 224, 224, GMAT_SWS_BILINEAR},
        // beyond every walker: the lines form (k_scale_yuvl.hip, round 4); "thumb:" cases run when the filter names them ("thumb")
        {"thumb: nv12 4K->480x270 rgb24 bicubic (8:1)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 480, 270, GMAT_SWS_BICUBIC},
        {"thumb: nv12 4K->480x270 nv12 bicubic (8:1)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_NV12, 480, 270, GMAT_SWS_BICUBIC},
        {"thumb: nv12 4K->320x180 rgb24 bicubic (12:1)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 320, 180, GMAT_SWS_BICUBIC},
        {"thumb: yuv420p 4K->320x180 yuv420p bicubic (12:1)", GMAT_PIX_FMT_YUV420P, 3840, 2160, GMAT_PIX_FMT_YUV420P, 320, 180, GMAT_SWS_BICUBIC},
        {"thumb: nv12 4K->160x90 rgb24 bicubic (24:1)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 160, 90, GMAT_SWS_BICUBIC},
        {"thumb: nv12 4K->640x360 rgb24 bicubic (6:1)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 640, 360, GMAT_SWS_BICUBIC},
        {"thumb: nv12 4K->854x480 rgb24 bicubic (4.5:1)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 854, 480, GMAT_SWS_BICUBIC},
        {"thumb: nv12 4K->1600x900 rgb24 bicubic (2.4:1)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 1600, 900, GMAT_SWS_BICUBIC},
        {"thumb: nv12 4K->1600x900 rgb24 lanczos (2.4:1)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 1600, 900, GMAT_SWS_LANCZOS},
        {"thumb: nv12 4K->854x480 rgb24 lanczos (4.5:1)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 854, 480, GMAT_SWS_LANCZOS},
        {"thumb: nv12 1080p->240x136 rgb24 bicubic (8:1)", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_RGB24, 240, 136, GMAT_SWS_BICUBIC},
        {"thumb: nv12 1080p->320x180 nv12 bicubic (6:1)", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_NV12, 320, 180, GMAT_SWS_BICUBIC},
        {"thumb: nv12 1080p->1280x720 rgb24 bicubic (3:2)", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_RGB24, 1280, 720, GMAT_SWS_BICUBIC},
        // deep samples on the lines form (scale_yuvl_h16_kernel, round 4); "deep:" cases run when the filter names them ("deep")
        {"deep: p010 4K->1600x900 p010 bicubic (2.4:1)", GMAT_PIX_FMT_P010LE, 3840, 2160, GMAT_PIX_FMT_P010LE, 1600, 900, GMAT_SWS_BICUBIC},
        {"deep: p010 4K->1280x720 nv12 bicubic (3:1)", GMAT_PIX_FMT_P010LE, 3840, 2160, GMAT_PIX_FMT_NV12, 1280, 720, GMAT_SWS_BICUBIC},
        {"deep: p010 4K->854x480 nv12 bicubic (4.5:1)", GMAT_PIX_FMT_P010LE, 3840, 2160, GMAT_PIX_FMT_NV12, 854, 480, GMAT_SWS_BICUBIC},
        {"deep: p010 4K->480x270 rgb24 bicubic (8:1)", GMAT_PIX_FMT_P010LE, 3840, 2160, GMAT_PIX_FMT_RGB24, 480, 270, GMAT_SWS_BICUBIC},
        {"deep: yuv420p10le 4K->480x270 yuv420p10le bicubic (8:1)", GMAT_PIX_FMT_YUV420P10LE, 3840, 2160, GMAT_PIX_FMT_YUV420P10LE, 480, 270, GMAT_SWS_BICUBIC},
        {"deep: yuv420p10le 1080p->1280x720 yuv420p10le bicubic (3:2)", GMAT_PIX_FMT_YUV420P10LE, 1920, 1080, GMAT_PIX_FMT_YUV420P10LE, 1280, 720, GMAT_SWS_BICUBIC},
        {"deep: p010 1080p->1280x720 rgb24 bicubic (3:2)", GMAT_PIX_FMT_P010LE, 1920, 1080, GMAT_PIX_FMT_RGB24, 1280, 720, GMAT_SWS_BICUBIC},
        {"deep: nv12 4K->1600x900 p010 bicubic (2.4:1)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_P010LE, 1600, 900, GMAT_SWS_BICUBIC},
        {"deep: nv12 4K->480x270 p010 bicubic (8:1)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_P010LE, 480, 270, GMAT_SWS_BICUBIC},
        {"deep: p010 720p->1080p p010 bicubic (2:3 up)", GMAT_PIX_FMT_P010LE, 1280, 720, GMAT_PIX_FMT_P010LE, 1920, 1080, GMAT_SWS_BICUBIC},
        // any ratio: the polyphase band walker (scale_yuvg_kernel); "any:" cases run when the filter names them or "any"
        {"any: nv12 4K->1600x900 rgb24 bicubic", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 1600, 900, GMAT_SWS_BICUBIC},
        {"any: nv12 4K->1600x900 nv12 bicubic", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_NV12, 1600, 900, GMAT_SWS_BICUBIC},
        {"any: nv12 4K->1366x768 rgb24 bicubic", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 1366, 768, GMAT_SWS_BICUBIC},
        {"any: nv12 4K->1366x768 nv12 bicubic", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_NV12, 1366, 768, GMAT_SWS_BICUBIC},
        {"any: nv12 4K->854x480 rgb24 bicubic", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 854, 480, GMAT_SWS_BICUBIC},
        {"any: nv12 4K->854x480 nv12 bicubic", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_NV12, 854, 480, GMAT_SWS_BICUBIC},
        {"any: nv12 4K->640x360 rgb24 bicubic (6:1)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 640, 360, GMAT_SWS_BICUBIC},
        {"any: nv12 4K->640x360 nv12 bicubic (6:1)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_NV12, 640, 360, GMAT_SWS_BICUBIC},
        {"any: nv12 1080p->768x432 rgb24 bicubic", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_RGB24, 768, 432, GMAT_SWS_BICUBIC},
        {"any: nv12 1080p->768x432 nv12 bicubic", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_NV12, 768, 432, GMAT_SWS_BICUBIC},
        {"any: nv12 1080p->854x480 rgb24 bicubic", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_RGB24, 854, 480, GMAT_SWS_BICUBIC},
        {"any: yuv420p 4K->1600x900 yuv420p bicubic", GMAT_PIX_FMT_YUV420P, 3840, 2160, GMAT_PIX_FMT_YUV420P, 1600, 900, GMAT_SWS_BICUBIC},
        {"any: nv12 4K->1600x900 rgb24 lanczos", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_RGB24, 1600, 900, GMAT_SWS_LANCZOS},
        // up-scales into 4:2:0 (the plane jobs of the band walker hold up to 15 open output rows: to 1 : 2)
        {"any: up nv12 720p->1080p nv12 bicubic", GMAT_PIX_FMT_NV12, 1280, 720, GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_SWS_BICUBIC},
        {"any: up nv12 1080p->1440p nv12 bicubic", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_NV12, 2560, 1440, GMAT_SWS_BICUBIC},
        {"any: up yuv420p 720p->1080p yuv420p bicubic", GMAT_PIX_FMT_YUV420P, 1280, 720, GMAT_PIX_FMT_YUV420P, 1920, 1080, GMAT_SWS_BICUBIC},
        {"any: up nv12 720p->1440p nv12 bicubic", GMAT_PIX_FMT_NV12, 1280, 720, GMAT_PIX_FMT_NV12, 2560, 1440, GMAT_SWS_BICUBIC},
        {"any: up nv12 720p->1080p rgb24 bicubic", GMAT_PIX_FMT_NV12, 1280, 720, GMAT_PIX_FMT_RGB24, 1920, 1080, GMAT_SWS_BICUBIC},
        // (round 4) beyond 1 : 2 — the tiled kernel's until the quad-lane walker — and an RGB destination at exactly 1 : 2
        {"any: up nv12 720p->4K nv12 bicubic (1:3)", GMAT_PIX_FMT_NV12, 1280, 720, GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_SWS_BICUBIC},
        {"any: up nv12 720p->4K rgb24 bicubic (1:3)", GMAT_PIX_FMT_NV12, 1280, 720, GMAT_PIX_FMT_RGB24, 3840, 2160, GMAT_SWS_BICUBIC},
        {"any: up nv12 1080p->4K rgb24 bicubic (1:2)", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_RGB24, 3840, 2160, GMAT_SWS_BICUBIC},
        {"any: up nv12 640x480->1080p nv12 bicubic", GMAT_PIX_FMT_NV12, 640, 480, GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_SWS_BICUBIC},
        {"any: up nv12 720p->1080p rgb24 lanczos", GMAT_PIX_FMT_NV12, 1280, 720, GMAT_PIX_FMT_RGB24, 1920, 1080, GMAT_SWS_LANCZOS},
        // between the chroma layouts, scaled (nvdec's NV12 for a planar consumer): rounds 1-3 the tiled kernel, round 4 the same-layout walker + a re-layout
        {"any: cross nv12 1080p->720p yuv420p bicubic (3:2)", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_YUV420P, 1280, 720, GMAT_SWS_BICUBIC},
        {"any: cross nv12 4K->720p yuv420p bicubic (3:1)", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_YUV420P, 1280, 720, GMAT_SWS_BICUBIC},
        {"any: cross nv12 4K->1600x900 yuv420p bicubic", GMAT_PIX_FMT_NV12, 3840, 2160, GMAT_PIX_FMT_YUV420P, 1600, 900, GMAT_SWS_BICUBIC},
        {"any: cross yuv420p 720p->1080p nv12 bicubic", GMAT_PIX_FMT_YUV420P, 1280, 720, GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_SWS_BICUBIC},
        // short-filter down-scales (8 taps): the band walker's by default, the quad-lane walker's behind GMAT_QUAD_WALKER=2
        {"any: short nv12 1440p->1080p nv12 bicubic (4:3)", GMAT_PIX_FMT_NV12, 2560, 1440, GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_SWS_BICUBIC},
        {"any: short nv12 1440p->1080p rgb24 bicubic (4:3)", GMAT_PIX_FMT_NV12, 2560, 1440, GMAT_PIX_FMT_RGB24, 1920, 1080, GMAT_SWS_BICUBIC},
        {"any: short nv12 1080p->1600x900 rgb24 bicubic (1.2:1)", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_RGB24, 1600, 900, GMAT_SWS_BICUBIC},
        {"any: short nv12 1080p->720p rgb24 bicubic (3:2)", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_RGB24, 1280, 720, GMAT_SWS_BICUBIC},
        {"any: short nv12 1080p->720p nv12 bicubic (3:2)", GMAT_PIX_FMT_NV12, 1920, 1080, GMAT_PIX_FMT_NV12, 1280, 720, GMAT_SWS_BICUBIC},
    };
    struct Op { const char *label; int op, bpp, pool; };
    const Op ops[] = {{"op: rotate+flip+smooth 4K rgb24", 0, 3}, {"op: rotate+flip+smooth 4K rgba", 0, 4}, {"op: smooth3x3 4K rgb24", 1, 3},
                      {"op: smooth3x3 4K gray", 1, 1}, {"op: transpose 4K rgb24", 2, 3}, {"op: hflip 4K rgb24", 3, 3},
                      {"op: transpose 4K gray (a luma plane)", 2, 1}, {"op: transpose 4K 2 bytes per sample", 2, 2},
                      {"op: median3x3 4K rgb24", 4, 3}, {"op: median3x3 4K gray", 4, 1},
                      {"op: rotate 17 deg bilinear 4K rgb24", 5, 3}, {"op: rotate 17 deg bilinear 4K gray", 5, 1},
                      {"op: rotate 17 deg cubic 4K rgb24", 6, 3}, {"op: rotate 17 deg nearest 4K rgb24", 7, 3},
                      {"op: crop 4K rgb24 (less a 32 x 16 border)", 8, 3}, {"op: crop 4K gray (less a 32 x 16 border)", 8, 1},
                      // the transposing ops into a destination with a hardware pool's pitch (rows aligned to 256 bytes instead of the dense 6480 / 8640 / 4320)
                      {"op: rotate+flip+smooth 4K rgb24, pool pitch", 0, 3, 1}, {"op: rotate+flip+smooth 4K rgba, pool pitch", 0, 4, 1},
                      {"op: transpose 4K rgb24, pool pitch", 2, 3, 1}, {"op: transpose 4K 2 bytes per sample, pool pitch", 2, 2, 1}};
    g_op_frames = NF;
    for (const Op &o : ops)
        if (*only && strstr(o.label, only)) run_op(o.label, o.op, 3840, 2160, o.bpp, launches * 4, o.pool);
    // "sweep:<w>x<h>-<w>x<h>": every pair of swscale_cuda's format list (libswscale/cuda/swscale_cuda.c:34-44; RGB32 / BGR32 are BGRA / RGBA on a little-endian host)
    // at that geometry, bicubic — the table that finds the pairs nobody has timed (round 5's method; round 6: profiles/r06_sweep_*.txt)
    if (strncmp(only, "sweep:", 6) == 0) {
        int sw = 1920, sh = 1080, dw = 1280, dh = 720;
        sscanf(only + 6, "%dx%d-%dx%d", &sw, &sh, &dw, &dh);
        struct F { const char *n; int f; };
        const F fm[] = {{"yuv420p", GMAT_PIX_FMT_YUV420P}, {"nv12", GMAT_PIX_FMT_NV12}, {"yuv420p10", GMAT_PIX_FMT_YUV420P10LE}, {"yuv420p16", GMAT_PIX_FMT_YUV420P16LE},
                        {"p010", GMAT_PIX_FMT_P010LE}, {"p016", GMAT_PIX_FMT_P016LE}, {"yuv444p", GMAT_PIX_FMT_YUV444P}, {"rgba", GMAT_PIX_FMT_RGBA}, {"rgb24", GMAT_PIX_FMT_RGB24},
                        {"rgba64", GMAT_PIX_FMT_RGBA64LE}, {"yuv444p16", GMAT_PIX_FMT_YUV444P16LE}};
        static char lab[128][64];
        int nl = 0;
        for (const F &a : fm)
            for (const F &b : fm) {
                if (a.f == b.f && sw == dw && sh == dh) continue;
                GmatSwsContext *probe = gmat_sws_getContext(sw, sh, a.f, dw, dh, b.f, GMAT_SWS_BICUBIC, nullptr);
                if (!probe) { printf("sweep: %s -> %s: no context\n", a.n, b.n); continue; }
                gmat_sws_freeContext(probe);
                snprintf(lab[nl], sizeof(lab[nl]), "sweep: %-9s -> %-9s", a.n, b.n);
                run_case(lab[nl], a.f, sw, sh, b.f, dw, dh, GMAT_SWS_BICUBIC, NF, launches, 0);
                nl++;
            }
        return 0;
    }
    for (const Case &k : cases) {
        if (strstr(k.label, "land:") && !strstr(only, "land")) continue;      // the landscape cases run on request only
        if (strstr(k.label, "any:") && !strstr(only, "any") && !(*only && strstr(k.label, only))) continue;
        if (strstr(k.label, "relayout:") && !(*only && strstr(k.label, only))) continue;
        if (strstr(k.label, "deep:") && !(*only && strstr(k.label, only))) continue;
        if (strstr(k.label, "thumb:") && !(*only && strstr(k.label, only))) continue;      // (or when the filter names them otherwise: "up nv12")
        if (strstr(k.label, "rgbsrc:") && !(*only && strstr(k.label, only))) continue;
        if (strstr(k.label, only)) run_case(k.label, k.sf, k.sw, k.sh, k.df, k.dw, k.dh, k.flags, NF, launches, verify);
    }
    return 0;
}
