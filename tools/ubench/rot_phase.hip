// rot_phase.hip — where a wave of rotate_lds_kernel spends its life: the product's k_transform.hip compiled with GMAT_ROT_PHASE (every wave
// leaves s_memtime at the ends of its phases) around one 4K frame.  FINDINGS R4-rotate.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Igmat_amd/csrc tools/ubench/rot_phase.hip -o tools/bin/rot_phase
//   tools/bin/rot_phase [bpp=3] [interp=1] [deg=17]
#define GMAT_ROT_PHASE 1
#include "../../gmat_amd/csrc/k_transform.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
namespace gmat {
void logf(int, const char *, ...) {}
const char *knob_read(const char *name, unsigned long long *seen, char *buf, unsigned bufsz, int *present)
{
    const char *v = getenv(name);
    *present = v != nullptr;
    if (v) snprintf(buf, bufsz, "%s", v);
    return v ? buf : nullptr;
}
}
int main(int argc, char **argv)
{
    const int bpp = argc > 1 ? atoi(argv[1]) : 3, interp = argc > 2 ? atoi(argv[2]) : 1;
    const double deg = argc > 3 ? atof(argv[3]) : 17.0;
    const int w = 3840, h = 2160, NB = 8;
    uint8_t *src[NB], *dst[NB];
    std::vector<uint8_t> host((size_t)w * h * bpp);
    for (size_t i = 0; i < host.size(); i++) host[i] = (uint8_t)(i * 2654435761u >> 13);
    for (int i = 0; i < NB; i++) {
        (void)hipMalloc(&src[i], host.size()); (void)hipMalloc(&dst[i], host.size());
        (void)hipMemcpy(src[i], host.data(), host.size(), hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 400; i++) gmat::launch_rotate(src[i % NB], w * bpp, dst[i % NB], w * bpp, w, h, w, h, bpp, deg * 3.14159265358979 / 180, interp, nullptr, 0, 0.0, 0.0, nullptr, 1);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 200; i++) gmat::launch_rotate(src[i % NB], w * bpp, dst[i % NB], w * bpp, w, h, w, h, bpp, deg * 3.14159265358979 / 180, interp, nullptr, 0, 0.0, 0.0, nullptr, 1);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("bpp %d interp %d deg %.1f: %.2f us a frame (with the stamps)\n", bpp, interp, deg, ms * 1000 / 200);
    (void)hipDeviceSynchronize();
    const size_t N = (size_t)(1 << 16) * 8;
    std::vector<unsigned long long> ph(N);
    (void)hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(gmat::g_rot_phase), N * 8);
    const int nw = std::min(1 << 16, 8 * ((w + 63) / 64) * ((((h + 31) / 32) + 7) / 8) * 4);
    // s_memtime ticks: find the rate from the kernel's span against the event time
    unsigned long long tmin = ~0ull, tmax = 0;
    double sum[8] = {0}; int cnt = 0, cntFast = 0;
    std::vector<double> life;
    for (int i = 0; i < nw; i++) {
        const unsigned long long *q = &ph[(size_t)i * 8];
        if (!q[0]) continue;
        tmin = std::min(tmin, q[0]);
        const unsigned long long end = q[6] > q[0] ? q[6] : q[4];
        tmax = std::max(tmax, end);
        cnt++;
        if (q[6] > q[0] && q[2] > q[0]) {
            cntFast++;
            sum[0] += (double)(q[7] - q[0]); sum[1] += (double)(q[1] - q[7]); sum[2] += (double)(q[2] - q[1]); sum[3] += (double)(q[3] - q[2]); sum[4] += (double)(q[4] - q[3]);
            sum[5] += (double)(q[5] - q[4]); sum[6] += (double)(q[6] - q[5]);
            life.push_back((double)(q[6] - q[0]));
        }
    }
    const double span = (double)(tmax - tmin);
    printf("waves stamped %d (fast path, 16-byte loader: %d); kernel span %.0f ticks (if 100 MHz: %.2f us)\n", cnt, cntFast, span, span / 100.0);
    const char *nm[] = {"", "scalar preamble (corners, box)", "loads issued", "loads back, LDS stores issued", "barrier", "LDS reads + blend", "stores issued"};
    printf("  %-34s %8.1f ticks\n", "start -> tile row known (kernarg)", sum[0] / std::max(cntFast, 1));
    for (int k = 1; k <= 6; k++) printf("  %-34s %8.1f ticks\n", nm[k], sum[k] / std::max(cntFast, 1));
    std::sort(life.begin(), life.end());
    double ls = 0; for (double v : life) ls += v;
    if (!life.empty()) printf("  wave life: mean %.1f, p10 %.0f, p50 %.0f, p90 %.0f ticks; sum of lives / span = %.1f waves in flight (of %d slots)\n", ls / life.size(),
           life[life.size() / 10], life[life.size() / 2], life[life.size() * 9 / 10], ls / span, 256 * 32);
    // how many distinct (xcc, se, cu) the first 2000 waves saw, and waves per CU at the kernel's middle
    {
        const unsigned long long mid = tmin + (tmax - tmin) / 2;
        int inflight = 0;
        for (int i = 0; i < nw; i++) { const unsigned long long *q = &ph[(size_t)i * 8]; if (q[0] && q[0] <= mid && (q[6] > q[0] ? q[6] : q[4]) >= mid) inflight++; }
        printf("  waves alive at the middle of the kernel: %d\n", inflight);
    }
    return 0;
}
