// tests/hipemu/hip/hip_runtime.h — TEST INFRASTRUCTURE: a minimal CPU emulation of the HIP
// constructs the product sources use, so that the UNMODIFIED kernel sources in gmat_amd/csrc can be
// compiled with a host compiler and their indexing / LDS / barrier logic exercised by the
// `-m "not gpu"` tests in this GPU-less container.  It is never part of the product library
// (gmat_amd/lib/libgmat_hip.so is built by hipcc against the real <hip/hip_runtime.h>) and is not
// a fallback: gmat_amd's loader never loads the emulated build.
//
// Execution model: blocks run one after another; the threads of a block are ucontext fibers
// scheduled round-robin per 64-lane wave.  __syncthreads() and wave-level barriers park a fiber
// until its block / wave has arrived.  hipMalloc places buffers against a PROT_NONE guard page so
// reads or writes past the end of a device buffer fault immediately.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define HIP_DYNAMIC_SHARED(type, var) type *var = reinterpret_cast<type *>(hipemu::dyn_lds());

using std::max;
using std::min;

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct uint2 { unsigned x, y; };
struct int2 { int x, y; };
static inline int2 make_int2(int x, int y) { return {x, y}; }
struct uint3 { unsigned x, y, z; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) float4 { float x, y, z, w; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint3 make_uint3(unsigned x, unsigned y, unsigned z) { return {x, y, z}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }

namespace hipemu {
struct Idx { unsigned x, y, z; };
extern Idx g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
void block_sync();
void wave_sync();
int shfl_xor(int v, int mask);
int readfirstlane(int v);
int lane_read(int v, int srcLane);
int dpp_wave_shift(int old, int v, int ctrl);
void *dyn_lds();
void launch(const std::function<void()> &body, dim3 grid, dim3 block, size_t shmem);
typedef short short2v __attribute__((ext_vector_type(2)));
static inline int sdot2(short2v a, short2v b, int c) { return c + (int)a.x * (int)b.x + (int)a.y * (int)b.y; }
static inline int sdot4(int a, int b, int c) { for (int i = 0; i < 4; i++) c += (int)(signed char)(a >> (8 * i)) * (int)(signed char)(b >> (8 * i)); return c; }
static inline unsigned udot4(unsigned a, unsigned b, unsigned c)
{
    for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xFF) * ((b >> (8 * i)) & 0xFF);
    return c;
}
static inline short2v cvt_pk_i16(int a, int b)
{
    short2v r;
    r.x = (short)(a < -32768 ? -32768 : a > 32767 ? 32767 : a);
    r.y = (short)(b < -32768 ? -32768 : b > 32767 ? 32767 : b);
    return r;
}
static inline unsigned perm(unsigned hi, unsigned lo, unsigned sel)
{
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    unsigned r = 0;
    for (int i = 0; i < 4; i++) {
        const unsigned s = (sel >> (8 * i)) & 0xFF;
        unsigned b = s < 8 ? (unsigned)((v >> (8 * s)) & 0xFF) : (s == 0x0C ? 0u : 0xFFu);
        r |= b << (8 * i);
    }
    return r;
}
} // namespace hipemu

#define threadIdx hipemu::g_threadIdx
#define blockIdx hipemu::g_blockIdx
#define blockDim hipemu::g_blockDim
#define gridDim hipemu::g_gridDim
#define __syncthreads() hipemu::block_sync()
#define __builtin_amdgcn_wave_barrier() hipemu::wave_sync()
#define __shfl_xor(v, m) hipemu::shfl_xor((int)(v), (m))
#define __builtin_amdgcn_readfirstlane(v) hipemu::readfirstlane((int)(v))
#define __builtin_amdgcn_readlane(v, l) hipemu::lane_read((int)(v), (l))
#define __builtin_amdgcn_update_dpp(old, v, ctrl, rm, bm, bc) hipemu::dpp_wave_shift((int)(old), (int)(v), (ctrl))
#define __builtin_amdgcn_sdot2(a, b, c, clamp) hipemu::sdot2(a, b, c)
#define __builtin_amdgcn_sdot4(a, b, c, clamp) hipemu::sdot4(a, b, c)
#define __builtin_amdgcn_perm(hi, lo, sel) hipemu::perm(hi, lo, sel)
#define __builtin_amdgcn_cvt_pk_i16(a, b) hipemu::cvt_pk_i16(a, b)
#define __builtin_amdgcn_udot4(a, b, c, clamp) hipemu::udot4(a, b, c)
static inline int __mul24(int a, int b) { return (int)((unsigned)((a << 8) >> 8) * (unsigned)((b << 8) >> 8)); }
// v_alignbyte_b32: ({hi, lo} >> (8 * (sh & 3))) & 0xFFFFFFFF
static inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)(((((uint64_t)hi) << 32) | lo) >> (8 * (sh & 3))); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
static inline unsigned long long __builtin_amdgcn_s_memrealtime() { return 0; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }

// ---- runtime API subset -------------------------------------------------------------------------
typedef enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidHandle = 400, hipErrorNotSupported = 801 } hipError_t;
typedef struct ihipStream_t *hipStream_t;
typedef struct ihipEvent_t *hipEvent_t;
typedef struct ihipGraph *hipGraph_t;
typedef struct hipGraphExec *hipGraphExec_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0 };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal };

hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k);
hipError_t hipMemset(void *d, int v, size_t n);
hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t st);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipGetDeviceCount(int *n);
struct hipDeviceProp_t { int multiProcessorCount; int warpSize; char name[64]; };
hipError_t hipGetDeviceProperties(hipDeviceProp_t *prop, int device);
hipError_t hipDeviceGetPCIBusId(char *buf, int len, int device);
hipError_t hipGetLastError();
const char *hipGetErrorName(hipError_t e);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t *e);
enum { hipEventDisableTiming = 2 };
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1, hipStreamCaptureStatusInvalidated = 2 };
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus *st);
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode m);
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t *g);
hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t g, void *, void *, size_t);
hipError_t hipGraphDestroy(hipGraph_t g);
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s);
hipError_t hipGraphExecDestroy(hipGraphExec_t e);

template <typename... KArgs, typename... Args>
static inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t,
                                      Args... args)
{
    hipemu::launch([=]() { kernel(args...); }, grid, block, shmem);
}
