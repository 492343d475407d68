// Shared helpers for the gfx950 kernels behind the C-ABI in include/phnet_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PHNET_OK 0
#define PHNET_ERR_ARG (-1)        // bad shape / null pointer / unsupported size
#define PHNET_ERR_WORKSPACE (-2)  // caller-provided workspace too small
#define PHNET_ERR_LAUNCH (-3)     // hipGetLastError() after the launch

#define PHNET_API extern "C" __attribute__((visibility("default")))

static inline int phnet_launch_status() {
    return hipGetLastError() == hipSuccess ? PHNET_OK : PHNET_ERR_LAUNCH;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
