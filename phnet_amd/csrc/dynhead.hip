// Per-anchor dynamic "convolution" of the lane head for gfx950: y_n = relu(LayerNorm(x_n @ w_n)) for every anchor n,
// forward and backward, one workgroup per anchor.
//
// Replaces, in libs/models/utils/dynamic_head.py:40-51, `torch.bmm(roi_feature, param1)` + norm1 + ReLU and
// `torch.bmm(features, param2)` + norm2 + ReLU (rocBLAS batched GEMM + ATen LayerNorm + ReLU, and in the backward two
// batched GEMMs + the LayerNorm backward kernels) by one launch each way (+ one small reduce for the shared LayerNorm
// affine gradients).  x_n is [P=36][K], w_n is the anchor's own generated [K][J] weight (K,J in {64,128}): 0.3 MFLOP
// per anchor - far too small for MFMA tiles to pay, so plain fp32 FMAs with x_n, w_n and the product staged in LDS.
#include "common.h"

namespace {

constexpr int NT = 1024;                   // 16 wavefronts: 4 per SIMD hide the LDS/global latency of the one block a CU gets
constexpr int NW = NT / 64;
constexpr int PMAX = 36;                    // sample points per anchor (Router4OL.py:97 sample_points)

struct DynShape { int N, P; float eps; };

// Every product phase below gives a thread ONE value of the per-thread operand per step and a register column of
// accumulators fed by wave-uniform (broadcast) LDS reads of the other operand, so the loads of one step are
// independent and pipeline; the accumulator count is a compile-time constant of <K,J>.

template <int K, int J> struct Lay {
    static constexpr int XP = K + 1, WP = J + 1, FP = J + 1;          // odd pitches: column walks are conflict-free
    static constexpr int X = 0, W = X + PMAX * XP, F = W + K * WP, L = F + PMAX * FP, END_FWD = L, END_BWD = L + 2 * NW * J;
};

template <int PITCH, int COLS>
__device__ __forceinline__ void stage(const float* __restrict__ src, float* dst, int rows)
{
    static_assert(COLS % 4 == 0, "float4 staging");
    const float4* s4 = reinterpret_cast<const float4*>(src);          // anchor slices are 16-byte aligned (COLS % 4 == 0)
    for (int i = threadIdx.x; i < rows * (COLS / 4); i += NT) {
        const int r = i / (COLS / 4), c = (i - r * (COLS / 4)) * 4;
        const float4 v = s4[i];
        float* d = dst + r * PITCH + c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}

// F[p][j] = sum_k X[p][k] * W[k][j]
template <int K, int J>
__device__ __forceinline__ void product_xw(const float* Xs, const float* Ws, float* Fs, int P)
{
    using L = Lay<K, J>;
    constexpr int G = NT / J, R = (PMAX + G - 1) / G;
    const int j = threadIdx.x % J, rg = threadIdx.x / J;
    float acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = 0.f;
#pragma unroll 2
    for (int k = 0; k < K; ++k) {
        const float wv = Ws[k * L::WP + j];
#pragma unroll
        for (int i = 0; i < R; ++i) acc[i] += Xs[(rg + i * G) * L::XP + k] * wv;      // rows >= P read staged zeros
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int p = rg + i * G;
        if (p < P) Fs[p * L::FP + j] = acc[i];
    }
}

template <int K, int J>
__device__ __forceinline__ void load_operands(const float* x, const float* w, float* lds, int n, int P)
{
    using L = Lay<K, J>;
    for (int i = P * L::XP + threadIdx.x; i < PMAX * L::XP; i += NT) lds[L::X + i] = 0.f;      // rows >= P
    stage<L::XP, K>(x + (size_t)n * P * K, lds + L::X, P);
    stage<L::WP, J>(w + (size_t)n * K * J, lds + L::W, K);
}

template <int K, int J>
__global__ __launch_bounds__(NT) void dyn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ y, float* __restrict__ stats, DynShape g)
{
    using L = Lay<K, J>;
    extern __shared__ float lds[];
    const int n = blockIdx.x, P = g.P;
    load_operands<K, J>(x, w, lds, n, P);
    __syncthreads();
    float* Fs = lds + L::F;
    product_xw<K, J>(lds + L::X, lds + L::W, Fs, P);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int p = wave; p < P; p += NW) {
        float s = 0.f;
        for (int j = lane; j < J; j += 64) s += Fs[p * L::FP + j];
        const float mu = wave_sum(s) / (float)J;
        float q = 0.f;
        for (int j = lane; j < J; j += 64) { const float d = Fs[p * L::FP + j] - mu; q += d * d; }
        const float rs = 1.0f / sqrtf(wave_sum(q) / (float)J + g.eps);
        if (lane == 0 && stats) { stats[((size_t)n * P + p) * 2] = mu; stats[((size_t)n * P + p) * 2 + 1] = rs; }
        for (int j = lane; j < J; j += 64)
            y[((size_t)n * P + p) * J + j] = fmaxf((Fs[p * L::FP + j] - mu) * rs * gamma[j] + beta[j], 0.f);
    }
}

// lnpart [N][2][J]: per-anchor partial gradients of the shared LayerNorm (weight, bias)
template <int K, int J>
__global__ __launch_bounds__(NT) void dyn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ w, const float* __restrict__ y,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ lnpart,
                                                     DynShape g)
{
    using L = Lay<K, J>;
    extern __shared__ float lds[];
    float* Xs = lds + L::X;
    float* Ws = lds + L::W;
    float* Fs = lds + L::F;                            // product, then dF in place
    float* Ls = lds + L::L;                            // [NW waves][2][J] LayerNorm affine partials
    const int n = blockIdx.x, P = g.P;
    load_operands<K, J>(x, w, lds, n, P);
    for (int i = P * L::FP + threadIdx.x; i < PMAX * L::FP; i += NT) Fs[i] = 0.f;               // dF rows >= P
    __syncthreads();
    product_xw<K, J>(Xs, Ws, Fs, P);
    __syncthreads();
    // ---- relu + LayerNorm backward, one wavefront per row; affine partials per wave ----
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int U = (J + 63) / 64;                   // lane owns columns lane + 64u
    float pw[U], pb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { pw[u] = 0.f; pb[u] = 0.f; }
    for (int p = wave; p < P; p += NW) {
        const float mu = stats[((size_t)n * P + p) * 2], rs = stats[((size_t)n * P + p) * 2 + 1];
        float gv[U], xh[U], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = lane + 64 * u;
            gv[u] = 0.f; xh[u] = 0.f;
            if (j < J) {
                const size_t o = ((size_t)n * P + p) * J + j;
                float gg = dy[o];
                if (!(y[o] > 0.f)) gg = 0.f;
                xh[u] = (Fs[p * L::FP + j] - mu) * rs;
                pw[u] += gg * xh[u];
                pb[u] += gg;
                gv[u] = gg * gamma[j];
                s1 += gv[u];
                s2 += gv[u] * xh[u];
            }
        }
        s1 = wave_sum(s1) / (float)J;
        s2 = wave_sum(s2) / (float)J;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = lane + 64 * u;
            if (j < J) Fs[p * L::FP + j] = rs * (gv[u] - s1 - xh[u] * s2);      // dF
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = lane + 64 * u;
        if (j < J) { Ls[(wave * 2 + 0) * J + j] = pw[u]; Ls[(wave * 2 + 1) * J + j] = pb[u]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * J; i += NT) {
        const int which = i / J, j = i - which * J;
        float t = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) t += Ls[(wv * 2 + which) * J + j];
        lnpart[((size_t)n * 2 + which) * J + j] = t;
    }
    // ---- dX[p][k] = sum_j dF[p][j] * W[k][j] ----
    if (dx) {
        constexpr int G = NT / K, R = (PMAX + G - 1) / G;
        const int k = threadIdx.x % K, rg = threadIdx.x / K;
        float acc[R];
#pragma unroll
        for (int i = 0; i < R; ++i) acc[i] = 0.f;
#pragma unroll 2
        for (int j = 0; j < J; ++j) {
            const float wv = Ws[k * L::WP + j];
#pragma unroll
            for (int i = 0; i < R; ++i) acc[i] += Fs[(rg + i * G) * L::FP + j] * wv;
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int p = rg + i * G;
            if (p < P) dx[((size_t)n * P + p) * K + k] = acc[i];
        }
    }
    // ---- dW[k][j] = sum_p X[p][k] * dF[p][j] ----
    {
        constexpr int G = NT / J, KA = K / G;
        const int j = threadIdx.x % J, kg = threadIdx.x / J;
        float acc[KA];
#pragma unroll
        for (int i = 0; i < KA; ++i) acc[i] = 0.f;
        for (int p = 0; p < P; ++p) {
            const float fv = Fs[p * L::FP + j];
#pragma unroll
            for (int i = 0; i < KA; ++i) acc[i] += Xs[p * L::XP + kg + i * G] * fv;
        }
#pragma unroll
        for (int i = 0; i < KA; ++i) dw[((size_t)n * K + kg + i * G) * J + j] = acc[i];
    }
}

// dgamma/dbeta (+)= sum over anchors of lnpart [N][2][J]: a workgroup owns 32 of the 2J columns, 32 anchor groups walk the
// rows (8 loads per thread for N = 240 instead of 60: the loop is pure load latency) and are folded through LDS
__global__ __launch_bounds__(1024) void dyn_ln_grad_reduce_kernel(const float* __restrict__ lnpart, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, int N, int J, int accumulate)
{
    __shared__ float part[32][33];
    const int cols = 2 * J, cl = threadIdx.x & 31, grp = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    float* dst = (c < J ? dgamma : dbeta) + (c < J ? c : c - J);
    const float old = (accumulate && grp == 0 && c < cols) ? *dst : 0.f;      // cold read first
    float s = 0.f;
    if (c < cols)
#pragma unroll 4
        for (int n = grp; n < N; n += 32) s += lnpart[(size_t)n * cols + c];
    part[grp][cl] = s;
    __syncthreads();
    if (grp != 0 || c >= cols) return;
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) t += part[r][cl];
    *dst = old + t;
}

template <int K, int J>
int launch_fwd(const float* x, const float* w, const float* gamma, const float* beta, float* y, float* stats, DynShape g, hipStream_t st)
{
    constexpr size_t lds = Lay<K, J>::END_FWD * sizeof(float);
    static bool attr = false;
    if (lds > 64 * 1024 && !attr) {
        if (hipFuncSetAttribute((const void*)dyn_fwd_kernel<K, J>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PHNET_ERR_LAUNCH;
        attr = true;
    }
    hipLaunchKernelGGL((dyn_fwd_kernel<K, J>), dim3(g.N), dim3(NT), lds, st, x, w, gamma, beta, y, stats, g);
    return phnet_launch_status();
}

template <int K, int J>
int launch_bwd(const float* dy, const float* x, const float* w, const float* y, const float* stats, const float* gamma,
               float* dx, float* dw, float* lnpart, DynShape g, hipStream_t st)
{
    constexpr size_t lds = Lay<K, J>::END_BWD * sizeof(float);
    static bool attr = false;
    if (lds > 64 * 1024 && !attr) {
        if (hipFuncSetAttribute((const void*)dyn_bwd_kernel<K, J>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PHNET_ERR_LAUNCH;
        attr = true;
    }
    hipLaunchKernelGGL((dyn_bwd_kernel<K, J>), dim3(g.N), dim3(NT), lds, st, dy, x, w, y, stats, gamma, dx, dw, lnpart, g);
    return phnet_launch_status();
}

// supported (K, J): the lane head's two products for hidden widths 64 and 32
#define DYN_DISPATCH(K_, J_, CALL)                           \
    if (K_ == 64 && J_ == 128) { CALL(64, 128); }            \
    else if (K_ == 128 && J_ == 64) { CALL(128, 64); }       \
    else if (K_ == 32 && J_ == 64) { CALL(32, 64); }         \
    else if (K_ == 64 && J_ == 32) { CALL(64, 32); }         \
    else return PHNET_ERR_ARG;

}  // namespace

extern "C" int phnet_dyn_mfma_applies(int32_t P, int32_t K, int32_t J);
extern "C" int phnet_dyn_mfma_fwd(const float* x, const float* w, const float* gamma, const float* beta, float* y, float* stats,
                                  int32_t N, int32_t P, int32_t K, int32_t J, float eps, void* stream);
extern "C" int phnet_dyn_mfma_bwd(const float* dy, const float* x, const float* w, const float* y, const float* stats, const float* gamma,
                                  float* dx, float* dw, float* lnpart, int32_t N, int32_t P, int32_t K, int32_t J, void* stream);
static int g_dyn_mfma = 1;       // matrix-pipe forward (dyn_mfma.hip) where it applies; phnet_tune_dyn_mfma(0) = the LDS / FMA kernels
extern int g_dyn_rows;           // (dyn_mfma.hip) its forward with one wavefront per (anchor, row fragment); bit 1 of the argument switches that off
PHNET_API int phnet_tune_dyn_mfma(int32_t on) { g_dyn_mfma = (on & 1) != 0; g_dyn_rows = !(on & 2); return PHNET_OK; }

// y[n] = relu(LayerNorm_J(x[n] @ w[n]) * gamma + beta);  x [N][P][K], w [N][K][J], y [N][P][J], stats [N][P][2]
// (mean, rstd; may be NULL for inference).  P <= 36; (K, J) in {(64,128), (128,64), (32,64), (64,32)}.
PHNET_API int phnet_dyn_bmm_ln_relu_fwd(const float* x, const float* w, const float* gamma, const float* beta, float* y,
                                        float* stats, int32_t N, int32_t P, int32_t K, int32_t J, float eps, void* stream)
{
    if (N < 1 || P < 1 || P > PMAX || !x || !w || !gamma || !beta || !y) return PHNET_ERR_ARG;
    if (g_dyn_mfma && phnet_dyn_mfma_applies(P, K, J)) return phnet_dyn_mfma_fwd(x, w, gamma, beta, y, stats, N, P, K, J, eps, stream);
    DynShape g{N, P, eps};
#define CALL(K_, J_) return launch_fwd<K_, J_>(x, w, gamma, beta, y, stats, g, (hipStream_t)stream)
    DYN_DISPATCH(K, J, CALL)
#undef CALL
}

// dy [N][P][J] -> dx [N][P][K] (optional), dw [N][K][J]; dgamma/dbeta [J] overwritten or accumulated.
// workspace: N*2*J floats.
PHNET_API int phnet_dyn_bmm_ln_relu_bwd(const float* dy, const float* x, const float* w, const float* y, const float* stats,
                                        const float* gamma, float* dx, float* dw, float* dgamma, float* dbeta,
                                        int32_t N, int32_t P, int32_t K, int32_t J, float eps, int32_t param_accumulate,
                                        void* workspace, uint64_t ws_bytes, void* stream)
{
    if (N < 1 || P < 1 || P > PMAX || !dy || !x || !w || !y || !stats || !gamma || !dw || !dgamma || !dbeta || !workspace)
        return PHNET_ERR_ARG;
    if ((uint64_t)N * 2 * J * sizeof(float) > ws_bytes) return PHNET_ERR_WORKSPACE;
    DynShape g{N, P, eps};
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (g_dyn_mfma && phnet_dyn_mfma_applies(P, K, J)) {
        rc = phnet_dyn_mfma_bwd(dy, x, w, y, stats, gamma, dx, dw, (float*)workspace, N, P, K, J, stream);
    } else {
#define CALL(K_, J_) rc = launch_bwd<K_, J_>(dy, x, w, y, stats, gamma, dx, dw, (float*)workspace, g, st)
    DYN_DISPATCH(K, J, CALL)
#undef CALL
    }
    if (rc != PHNET_OK) return rc;
    hipLaunchKernelGGL(dyn_ln_grad_reduce_kernel, dim3((2 * J + 31) / 32), dim3(1024), 0, st, (const float*)workspace, dgamma, dbeta,
                       N, J, param_accumulate);
    return phnet_launch_status();
}
