// Lane-head kernels of the Router4OLV2 model family (what testOLV3.py runs) for gfx950 - inference only, like the
// reference (its training path cannot run as shipped: Router4OLV2.py:283 vs loss4OL.py:177).
//
//   gate_v2_kernel        AdaptiveRouter4LaneV2.forward (libs/models/Router.py:83-132): Conv1d(k3) + BatchNorm1d + ReLU,
//                         Conv1d(k1) + BatchNorm1d + ReLU, Flatten, Linear(96 -> P), mean, sigmoid - one launch for the
//                         ~10 ATen launches of the reference; BatchNorm in eval form (scale / shift folded by the host).
//   dyn_any_kernel        relu(LayerNorm(x_n @ w_n)) per anchor with run-time (P, K, J): DynamicConvV2's two products
//                         (libs/models/utils/dynamic_head.py:94-104) at the per-level widths 64/32/16 and 24/48/96 sample
//                         points (csrc/dynhead.hip is specialised for the 36-point, 64-channel V1 head and its backward).
//   route_lines_kernel    RouterOL.forward, eval (Router4OLV2.py:508-511): mean of the stage gates per anchor, then the
//                         HARD selection torch.where(mean >= 0.5, branch B, branch A) (or the soft blend of Router4OL.py:538-541).
//
// All three are latency-bound (240 anchors x <= 3072 values): one workgroup per anchor, operands staged in LDS, fp32 FMAs.
#include "common.h"

namespace {

constexpr int NT = 256;

struct GateV2 {
    int C, P, C1, C2;           // input channels, sample points, C/reduction, C/C_last
    const float *w1, *s1, *t1;  // [C1][C][3], BatchNorm scale / shift [C1]
    const float *w2, *s2, *t2;  // [C2][C1],   BatchNorm scale / shift [C2]
    const float *wl, *bl;       // [P][C2*P], [P]
};

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    __syncthreads();
    return t;
}

// x [M][C][P] (the [anchor][channel][point] copy the ROI pooling writes), out [M]
__global__ __launch_bounds__(NT) void gate_v2_kernel(const float* __restrict__ x, GateV2 g, float* __restrict__ out)
{
    extern __shared__ float lds[];
    __shared__ float red[NT / 64];
    const int C = g.C, P = g.P, C1 = g.C1, C2 = g.C2, PP = P + 2;
    float* xs = lds;                    // [C][P+2]: one zero column on either side (Conv1d padding = 1)
    float* h1 = xs + C * PP;            // [C1][P]
    float* h2 = h1 + C1 * P;            // [C2*P]  (= nn.Flatten(1) of [C2][P])
    const float* xm = x + (size_t)blockIdx.x * C * P;
    for (int i = threadIdx.x; i < C * PP; i += NT) {
        const int c = i / PP, p = i - c * PP;
        xs[i] = (p == 0 || p == PP - 1) ? 0.f : xm[c * P + p - 1];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C1 * P; i += NT) {
        const int c1 = i / P, p = i - c1 * P;
        const float* w = g.w1 + (size_t)c1 * C * 3;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int c = 0; c < C; ++c) {
            const float* xr = xs + c * PP + p;
            a0 += w[3 * c] * xr[0]; a1 += w[3 * c + 1] * xr[1]; a2 += w[3 * c + 2] * xr[2];
        }
        h1[i] = fmaxf((a0 + a1 + a2) * g.s1[c1] + g.t1[c1], 0.f);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C2 * P; i += NT) {
        const int c2 = i / P, p = i - c2 * P;
        float a = 0.f;
        for (int c = 0; c < C1; ++c) a += g.w2[c2 * C1 + c] * h1[c * P + p];
        h2[i] = fmaxf(a * g.s2[c2] + g.t2[c2], 0.f);
    }
    __syncthreads();
    const int F = C2 * P;
    float part = 0.f;
    for (int j = threadIdx.x; j < P; j += NT) {
        const float* w = g.wl + (size_t)j * F;
        float a0 = 0.f, a1 = 0.f;
        int i = 0;
        for (; i + 1 < F; i += 2) { a0 += w[i] * h2[i]; a1 += w[i + 1] * h2[i + 1]; }
        if (i < F) a0 += w[i] * h2[i];
        part += (a0 + a1) + g.bl[j];
    }
    const float mean = block_sum(part, red) / (float)P;
    if (threadIdx.x == 0) out[blockIdx.x] = 1.0f / (1.0f + expf(-mean));
}

// y[n] = relu(LayerNorm_J(x[n] @ w[n]) * gamma + beta);  x [N][P][K], w [N][K][J], y [N][P][J]
__global__ __launch_bounds__(NT) void dyn_any_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ y, int P, int K, int J, float eps)
{
    extern __shared__ float lds[];
    const int KP = K + 1;                 // odd pitch: lanes of one wave that walk different rows hit different banks
    float* xs = lds;                      // [P][K+1]
    float* ws = xs + P * KP;              // [K][J]
    float* fs = ws + K * J;               // [P][J]
    const size_t n = blockIdx.x;
    const float4* x4 = reinterpret_cast<const float4*>(x + n * P * K);
    for (int i = threadIdx.x; i < P * K / 4; i += NT) {
        const float4 v = x4[i];
        const int r = (4 * i) / K, c = 4 * i - r * K;
        float* d = xs + r * KP + c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    const float4* w4 = reinterpret_cast<const float4*>(w + n * K * J);
    for (int i = threadIdx.x; i < K * J / 4; i += NT) reinterpret_cast<float4*>(ws)[i] = w4[i];
    __syncthreads();
    for (int i = threadIdx.x; i < P * J; i += NT) {
        const int p = i / J, j = i - p * J;
        const float* xr = xs + p * KP;
        float a0 = 0.f, a1 = 0.f;
        for (int k = 0; k < K; k += 2) { a0 += xr[k] * ws[k * J + j]; a1 += xr[k + 1] * ws[(k + 1) * J + j]; }
        fs[i] = a0 + a1;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int p = wave; p < P; p += NT / 64) {
        const float* f = fs + p * J;
        float s = 0.f;
        for (int j = lane; j < J; j += 64) s += f[j];
        const float mu = wave_sum(s) / (float)J;
        float q = 0.f;
        for (int j = lane; j < J; j += 64) { const float d = f[j] - mu; q += d * d; }
        const float rs = 1.0f / sqrtf(wave_sum(q) / (float)J + eps);
        for (int j = lane; j < J; j += 64) y[(n * P + p) * J + j] = fmaxf((f[j] - mu) * rs * gamma[j] + beta[j], 0.f);
    }
}

// gates [S][M]; a, b, out [M][W]
__global__ __launch_bounds__(NT) void route_lines_kernel(const float* __restrict__ gates, const float* __restrict__ a,
                                                         const float* __restrict__ b, float* __restrict__ out,
                                                         int S, int M, int W, int hard)
{
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    if (i >= (long)M * W) return;
    const int m = (int)(i / W);
    float d = 0.f;
    for (int s = 0; s < S; ++s) d += gates[(size_t)s * M + m];
    d /= (float)S;
    out[i] = hard ? (d >= 0.5f ? b[i] : a[i]) : (b[i] * d + a[i] * (1.0f - d));
}

}  // namespace

// Routing gate of the V2 family, inference form.  x [M][C][P]; w1 [C1][C][3] (Conv1d k=3, pad 1, no bias), s1 / t1 [C1] =
// BatchNorm1d folded to y = conv * s + t (s = gamma / sqrt(var + eps), t = beta - mean * s); w2 [C2][C1] (Conv1d k=1),
// s2 / t2 [C2]; wl [P][C2*P], bl [P] (Linear over the flattened [C2][P] map); out [M] = sigmoid(mean_j linear_j).
PHNET_API int phnet_gate_v2_fwd(const float* x, const float* w1, const float* s1, const float* t1, const float* w2,
                                const float* s2, const float* t2, const float* wl, const float* bl, float* out,
                                int32_t M, int32_t C, int32_t P, int32_t C1, int32_t C2, void* stream)
{
    if (M < 0 || C < 1 || P < 1 || C1 < 1 || C2 < 1) return PHNET_ERR_ARG;
    if (M == 0) return PHNET_OK;
    if (!x || !w1 || !s1 || !t1 || !w2 || !s2 || !t2 || !wl || !bl || !out) return PHNET_ERR_ARG;
    const size_t lds = ((size_t)C * (P + 2) + (size_t)C1 * P + (size_t)C2 * P) * sizeof(float);
    if (lds > 60 * 1024) return PHNET_ERR_ARG;
    GateV2 g{C, P, C1, C2, w1, s1, t1, w2, s2, t2, wl, bl};
    hipLaunchKernelGGL(gate_v2_kernel, dim3(M), dim3(NT), lds, (hipStream_t)stream, x, g, out);
    return phnet_launch_status();
}

// y[n] = relu(LayerNorm_J(x[n] @ w[n]) * gamma + beta) for run-time shapes (forward only).  x [N][P][K], w [N][K][J],
// gamma / beta [J], y [N][P][J]; K, J multiples of 4, K even; (P*(K+1) + K*J + P*J) floats must fit 60 KB of LDS.
PHNET_API int phnet_dyn_bmm_ln_relu_fwd_any(const float* x, const float* w, const float* gamma, const float* beta, float* y,
                                            int32_t N, int32_t P, int32_t K, int32_t J, float eps, void* stream)
{
    if (N < 0 || P < 1 || K < 4 || J < 4 || (K & 3) || (J & 3)) return PHNET_ERR_ARG;
    if (N == 0) return PHNET_OK;
    if (!x || !w || !gamma || !beta || !y) return PHNET_ERR_ARG;
    const size_t lds = ((size_t)P * (K + 1) + (size_t)K * J + (size_t)P * J) * sizeof(float);
    if (lds > 60 * 1024) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(dyn_any_kernel, dim3(N), dim3(NT), lds, (hipStream_t)stream, x, w, gamma, beta, y, P, K, J, eps);
    return phnet_launch_status();
}

// out[m] = hard ? (mean_s gates[s][m] >= 0.5 ? b[m] : a[m]) : b[m] * d + a[m] * (1 - d), d = mean_s gates[s][m].
// gates [S][M]; a (branch A), b (branch B), out [M][W].
PHNET_API int phnet_route_lines(const float* gates, const float* a, const float* b, float* out,
                                int32_t S, int32_t M, int32_t W, int32_t hard, void* stream)
{
    if (S < 1 || M < 0 || W < 1) return PHNET_ERR_ARG;
    if (M == 0) return PHNET_OK;
    if (!gates || !a || !b || !out) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(route_lines_kernel, dim3((unsigned)ceil_div64((int64_t)M * W, NT)), dim3(NT), 0, (hipStream_t)stream,
                       gates, a, b, out, S, M, W, hard);
    return phnet_launch_status();
}
