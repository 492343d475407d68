// Small streaming kernels of the lane head (HBM-bound, 16-byte accesses where the shape allows).
#include "common.h"

namespace {
constexpr int NT = 256;

__global__ __launch_bounds__(NT) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ dx, long n)
{
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    if (i < n) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
}  // namespace

// dx = dy where y > 0 else 0 (ReLU backward through the saved output); dx may alias dy.
PHNET_API int phnet_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream)
{
    if (n < 0) return PHNET_ERR_ARG;
    if (n == 0) return PHNET_OK;
    if (!dy || !y || !dx) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)ceil_div64(n, NT)), dim3(NT), 0, (hipStream_t)stream, dy, y, dx, (long)n);
    return phnet_launch_status();
}
