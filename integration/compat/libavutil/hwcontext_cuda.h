#pragma once
#include "hwcontext.h"
/* the three-field slot of libavutil/hwcontext_cuda.h:42-46; context and stream are opaque pointers (a hipStream_t here) */
typedef struct AVCUDADeviceContext { void *cuda_ctx; void *stream; void *internal; } AVCUDADeviceContext;
