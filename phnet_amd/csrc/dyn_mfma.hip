// Per-anchor dynamic "convolution" of the lane head, y_n = relu(LayerNorm(x_n @ w_n)) (libs/models/utils/dynamic_head.py:40-51),
// with ONE WAVEFRONT PER ANCHOR on the f32 matrix pipe (gfx950) - the forward of dynhead.hip's one-workgroup-per-anchor kernels.
//
// dynhead.hip stages x_n, w_n and the product in LDS and multiplies with vector FMAs: every FMA needs an LDS read (6 reads per 5
// FMAs), so a 1024-thread workgroup spends ~5.6 us of LDS bandwidth per anchor and a 1200-anchor launch is five rounds of that
// (49 / 42 us).  Here the product runs on v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: bit-for-bit an fmaf chain, so the
// arithmetic stays f32 like the reference's bmm) with both operands loaded straight from global memory into the registers of the
// lanes that feed them: 36 rows = 3 row fragments, K/4 steps, J/16 column fragments, 384 MFMAs per anchor, no LDS, no barrier.
// Layout tricks (the reduction index and the output columns may be permuted freely as long as both operands agree):
//   * lane (jj = lane & 15, kk = lane >> 4) owns the reduction indices k = (K/4) kk + s for step s: its A values of a row are
//     K/4 CONSECUTIVE floats of x (float4 loads), its B values of 4 steps... come from 4 consecutive ROWS of w;
//   * column fragment g covers the columns 64 (g >> 2) + 4 jj + (g & 3): one float4 of a w row feeds four fragments, and the
//     lane's results of a row are 4 consecutive columns per 64-column group: float4 stores, float4 gamma / beta.
// LayerNorm statistics per row: in-lane sum over the lane's columns, then over the 16 lanes of its group (two-pass: mean, then the
// variance of the centred values, as dynhead.hip).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int DP_MAX = 48;                  // 3 row fragments of 16 (the model: P = 36)

__device__ __forceinline__ float group16_sum(float v) {     // sum over the 16 lanes that share lane >> 4
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

// F[3 row fragments][J/16 column fragments] = x_n @ w_n in the accumulator layout: lane (jj, kk), register e of fragment (f, g)
// = row 16 f + 4 kk + e, column 64 (g >> 2) + 4 jj + (g & 3)
template <int K, int J>
__device__ __forceinline__ void product(const float* __restrict__ xn, const float* __restrict__ wn, int P, int lane, f32x4 (&acc)[3][J / 16])
{
    constexpr int KS = K / 4, FN = J / 16, G2 = J / 64;
    const int jj = lane & 15, kk = lane >> 4;
    float a[3][KS];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const int row = 16 * f + jj;
        const bool ok = row < P;
        const f32x4* src = reinterpret_cast<const f32x4*>(xn + (size_t)(ok ? row : 0) * K + KS * kk);
#pragma unroll
        for (int t = 0; t < KS / 4; ++t) {
            const f32x4 v = src[t];
            a[f][4 * t] = ok ? v.x : 0.f; a[f][4 * t + 1] = ok ? v.y : 0.f; a[f][4 * t + 2] = ok ? v.z : 0.f; a[f][4 * t + 3] = ok ? v.w : 0.f;
        }
    }
#pragma unroll
    for (int f = 0; f < 3; ++f)
#pragma unroll
        for (int g = 0; g < FN; ++g) acc[f][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* wrow = wn + (size_t)(KS * kk) * J + 4 * jj;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        f32x4 b[G2];
#pragma unroll
        for (int g2 = 0; g2 < G2; ++g2) b[g2] = *reinterpret_cast<const f32x4*>(wrow + (size_t)s * J + 64 * g2);
#pragma unroll
        for (int g = 0; g < FN; ++g) {
            const float bv = b[g >> 2][g & 3];
#pragma unroll
            for (int f = 0; f < 3; ++f) acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f][s], bv, acc[f][g], 0, 0, 0);
        }
    }
}

// one row fragment (rows 16 f .. 16 f + 15) of the same product, same order of operations per accumulator
template <int K, int J>
__device__ __forceinline__ void product_frag(const float* __restrict__ xn, const float* __restrict__ wn, int P, int lane, int f, f32x4 (&acc)[J / 16])
{
    constexpr int KS = K / 4, FN = J / 16, G2 = J / 64;
    const int jj = lane & 15, kk = lane >> 4;
    float a[KS];
    {
        const int row = 16 * f + jj;
        const bool ok = row < P;
        const f32x4* src = reinterpret_cast<const f32x4*>(xn + (size_t)(ok ? row : 0) * K + KS * kk);
#pragma unroll
        for (int t = 0; t < KS / 4; ++t) {
            const f32x4 v = src[t];
            a[4 * t] = ok ? v.x : 0.f; a[4 * t + 1] = ok ? v.y : 0.f; a[4 * t + 2] = ok ? v.z : 0.f; a[4 * t + 3] = ok ? v.w : 0.f;
        }
    }
#pragma unroll
    for (int g = 0; g < FN; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* wrow = wn + (size_t)(KS * kk) * J + 4 * jj;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        f32x4 b[G2];
#pragma unroll
        for (int g2 = 0; g2 < G2; ++g2) b[g2] = *reinterpret_cast<const f32x4*>(wrow + (size_t)s * J + 64 * g2);
#pragma unroll
        for (int g = 0; g < FN; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[g >> 2][g & 3], acc[g], 0, 0, 0);
    }
}

// Forward with one wavefront per (anchor, 16-row fragment): a launch of the kernel below is 1200 wavefronts on 1024 SIMDs, each
// streaming its anchor's 32 KB of weights with nothing else on its SIMD to hide the load latency (47 / 40 us for 39 MB: ~1 TB/s).
// The rows of an anchor are independent up to the LayerNorm, which is per row: the three row fragments of an anchor go to three
// wavefronts (the second and third read the anchor's weights from L2), 3600 wavefronts per launch, a third of the registers each.
// Same arithmetic per accumulator, bit for bit.
template <int K, int J>
__global__ __launch_bounds__(256, 4) void dyn_mfma_fwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* __restrict__ y, float* __restrict__ stats, int N, int P, int NFR, float eps)
{
    constexpr int FN = J / 16, G2 = J / 64;
    const int lane = threadIdx.x & 63;
    const int job = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int n = job / NFR, f = job - n * NFR;
    if (n >= N) return;
    const int jj = lane & 15, kk = lane >> 4;
    f32x4 acc[FN];
    product_frag<K, J>(x + (size_t)n * P * K, w + (size_t)n * K * J, P, lane, f, acc);
    f32x4 gam[G2], bet[G2];
#pragma unroll
    for (int g2 = 0; g2 < G2; ++g2) {
        gam[g2] = *reinterpret_cast<const f32x4*>(gamma + 64 * g2 + 4 * jj);
        bet[g2] = *reinterpret_cast<const f32x4*>(beta + 64 * g2 + 4 * jj);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = 16 * f + 4 * kk + e;
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < FN; ++g) s += acc[g][e];
        const float mu = group16_sum(s) * (1.0f / (float)J);
        float q = 0.f;
#pragma unroll
        for (int g = 0; g < FN; ++g) { const float d = acc[g][e] - mu; q += d * d; }
        const float rs = 1.0f / sqrtf(group16_sum(q) * (1.0f / (float)J) + eps);
        if (row < P) {
            if (jj == 0 && stats) { stats[((size_t)n * P + row) * 2] = mu; stats[((size_t)n * P + row) * 2 + 1] = rs; }
            float* dst = y + ((size_t)n * P + row) * J + 4 * jj;
#pragma unroll
            for (int g2 = 0; g2 < G2; ++g2) {
                f32x4 o;
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] = fmaxf((acc[4 * g2 + t][e] - mu) * rs * gam[g2][t] + bet[g2][t], 0.f);
                *reinterpret_cast<f32x4*>(dst + 64 * g2) = o;
            }
        }
    }
}

// 256 threads = 4 anchors per workgroup (no LDS, no barrier)
template <int K, int J>
__global__ __launch_bounds__(256, 2) void dyn_mfma_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ y, float* __restrict__ stats, int N, int P, float eps)
{
    constexpr int FN = J / 16, G2 = J / 64;
    const int lane = threadIdx.x & 63;
    const int n = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (n >= N) return;
    const int jj = lane & 15, kk = lane >> 4;
    f32x4 acc[3][FN];
    product<K, J>(x + (size_t)n * P * K, w + (size_t)n * K * J, P, lane, acc);
    f32x4 gam[G2], bet[G2];
#pragma unroll
    for (int g2 = 0; g2 < G2; ++g2) {
        gam[g2] = *reinterpret_cast<const f32x4*>(gamma + 64 * g2 + 4 * jj);
        bet[g2] = *reinterpret_cast<const f32x4*>(beta + 64 * g2 + 4 * jj);
    }
#pragma unroll
    for (int f = 0; f < 3; ++f)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 16 * f + 4 * kk + e;
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < FN; ++g) s += acc[f][g][e];
            const float mu = group16_sum(s) * (1.0f / (float)J);
            float q = 0.f;
#pragma unroll
            for (int g = 0; g < FN; ++g) { const float d = acc[f][g][e] - mu; q += d * d; }
            const float rs = 1.0f / sqrtf(group16_sum(q) * (1.0f / (float)J) + eps);
            if (row < P) {
                if (jj == 0 && stats) { stats[((size_t)n * P + row) * 2] = mu; stats[((size_t)n * P + row) * 2 + 1] = rs; }
                float* dst = y + ((size_t)n * P + row) * J + 4 * jj;
#pragma unroll
                for (int g2 = 0; g2 < G2; ++g2) {
                    f32x4 o;
#pragma unroll
                    for (int t = 0; t < 4; ++t) o[t] = fmaxf((acc[f][4 * g2 + t][e] - mu) * rs * gam[g2][t] + bet[g2][t], 0.f);
                    *reinterpret_cast<f32x4*>(dst + 64 * g2) = o;
                }
            }
        }
}

// ---- backward: dX = dF @ w^T, dW = x^T @ dF with dF = LayerNorm / ReLU backward of dy; per-anchor LayerNorm affine partials ----
// One wavefront per anchor again.  The product is recomputed on the matrix pipe (its accumulator layout is the layout dy / y are
// read in: float4 per row and 64-column group), dF goes through a wave-private LDS image [P][J + 4] because the two products
// need it in operand layouts the accumulators are not in:
//   dX[p][k] = sum_j dF[p][j] w[k][j]: A = dF rows (J/4 consecutive floats per lane from LDS), B = w rows straight from global
//              (lane (jj, kk) of column fragment g reads J/4 consecutive floats of w row 16 g + jj);
//   dW[k][j] = sum_p x[p][k] dF[p][j]: reduction over the points, p = ceil(P/4) kk + s; A = x columns (4-byte loads, L1-warm: the
//              product just read them), B = dF rows as float4 per 64-column group (four column fragments per read).
template <int K, int J>
__global__ __launch_bounds__(256, 2) void dyn_mfma_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              const float* __restrict__ w, const float* __restrict__ y,
                                                              const float* __restrict__ stats, const float* __restrict__ gamma,
                                                              float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ lnpart,
                                                              int N, int P)
{
    constexpr int FN = J / 16, G2 = J / 64, FK = K / 16, JS = J / 4, PITCH = J + 4, ROWS = 36;
    extern __shared__ float lds[];                              // [4 waves][ROWS][PITCH]
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (n >= N) return;                                         // (wave-uniform; the LDS image is private to the wave: no barrier)
    const int jj = lane & 15, kk = lane >> 4;
    float* Fs = lds + (size_t)wv * ROWS * PITCH;
    const float* xn = x + (size_t)n * P * K;
    const float* wn = w + (size_t)n * K * J;
    // ---- the product again, then relu + LayerNorm backward in the accumulator layout ----
    f32x4 acc[3][FN];
    product<K, J>(xn, wn, P, lane, acc);
    f32x4 gam[G2], pw[G2], pb[G2];
#pragma unroll
    for (int g2 = 0; g2 < G2; ++g2) {
        gam[g2] = *reinterpret_cast<const f32x4*>(gamma + 64 * g2 + 4 * jj);
        pw[g2] = (f32x4){0.f, 0.f, 0.f, 0.f}; pb[g2] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int f = 0; f < 3; ++f)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 16 * f + 4 * kk + e;
            const bool ok = row < P;
            const size_t ro = ((size_t)n * P + (ok ? row : 0)) * J + 4 * jj;
            const float mu = stats[((size_t)n * P + (ok ? row : 0)) * 2], rs = stats[((size_t)n * P + (ok ? row : 0)) * 2 + 1];
            f32x4 gv[G2], xh[G2];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < G2; ++g2) {
                const f32x4 dyv = *reinterpret_cast<const f32x4*>(dy + ro + 64 * g2);
                const f32x4 yv = *reinterpret_cast<const f32x4*>(y + ro + 64 * g2);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float gg = (ok && yv[t] > 0.f) ? dyv[t] : 0.f;
                    const float h = (acc[f][4 * g2 + t][e] - mu) * rs;
                    xh[g2][t] = h;
                    pw[g2][t] += gg * h;
                    pb[g2][t] += gg;
                    const float v = gg * gam[g2][t];
                    gv[g2][t] = v;
                    s1 += v; s2 += v * h;
                }
            }
            s1 = group16_sum(s1) * (1.0f / (float)J);
            s2 = group16_sum(s2) * (1.0f / (float)J);
            if (row < ROWS) {
#pragma unroll
                for (int g2 = 0; g2 < G2; ++g2) {
                    f32x4 d;
#pragma unroll
                    for (int t = 0; t < 4; ++t) d[t] = ok ? rs * (gv[g2][t] - s1 - xh[g2][t] * s2) : 0.f;
                    *reinterpret_cast<f32x4*>(Fs + row * PITCH + 64 * g2 + 4 * jj) = d;       // dF, zero rows past P
                }
            }
        }
    // LayerNorm affine partials of this anchor: the lane's column sums over its 12 rows, then over the four row groups kk
#pragma unroll
    for (int g2 = 0; g2 < G2; ++g2)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float a = pw[g2][t], b = pb[g2][t];
            a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
            pw[g2][t] = a; pb[g2][t] = b;
        }
    if (kk == 0) {
#pragma unroll
        for (int g2 = 0; g2 < G2; ++g2) {
            *reinterpret_cast<f32x4*>(lnpart + ((size_t)n * 2 + 0) * J + 64 * g2 + 4 * jj) = pw[g2];
            *reinterpret_cast<f32x4*>(lnpart + ((size_t)n * 2 + 1) * J + 64 * g2 + 4 * jj) = pb[g2];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the wave's LDS writes have landed (DS operations of a wave run in order)
    __builtin_amdgcn_wave_barrier();
    // ---- dX[p][k] = sum_j dF[p][j] * w[k][j] ----
    if (dx) {
        float a[3][JS];
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            const int row = 16 * f + jj;
            const bool ok = row < P;
            const f32x4* src = reinterpret_cast<const f32x4*>(Fs + (ok ? row : 0) * PITCH + JS * kk);
#pragma unroll
            for (int t = 0; t < JS / 4; ++t) {
                const f32x4 v = src[t];
                a[f][4 * t] = ok ? v.x : 0.f; a[f][4 * t + 1] = ok ? v.y : 0.f; a[f][4 * t + 2] = ok ? v.z : 0.f; a[f][4 * t + 3] = ok ? v.w : 0.f;
            }
        }
#pragma unroll
        for (int g = 0; g < FK; ++g) {
            float b[JS];
            const f32x4* src = reinterpret_cast<const f32x4*>(wn + (size_t)(16 * g + jj) * J + JS * kk);
#pragma unroll
            for (int t = 0; t < JS / 4; ++t) { const f32x4 v = src[t]; b[4 * t] = v.x; b[4 * t + 1] = v.y; b[4 * t + 2] = v.z; b[4 * t + 3] = v.w; }
            f32x4 ax[3];
#pragma unroll
            for (int f = 0; f < 3; ++f) ax[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < JS; ++s)
#pragma unroll
                for (int f = 0; f < 3; ++f) ax[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f][s], b[s], ax[f], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < 3; ++f)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = 16 * f + 4 * kk + e;
                    if (row < P) dx[((size_t)n * P + row) * K + 16 * g + jj] = ax[f][e];
                }
        }
    }
    // ---- dW[k][j] = sum_p x[p][k] * dF[p][j] ----
    {
        const int RS = (P + 3) >> 2;                             // reduction steps: p = RS * kk + s
        f32x4 aw[FK][FN];
#pragma unroll
        for (int fk = 0; fk < FK; ++fk)
#pragma unroll
            for (int g = 0; g < FN; ++g) aw[fk][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < RS; ++s) {
            const int p = RS * kk + s;
            const bool ok = p < P;
            float av[FK];
#pragma unroll
            for (int fk = 0; fk < FK; ++fk) { const float v = xn[(size_t)(ok ? p : 0) * K + 16 * fk + jj]; av[fk] = ok ? v : 0.f; }
            f32x4 bv[G2];
#pragma unroll
            for (int g2 = 0; g2 < G2; ++g2) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(Fs + (ok && p < ROWS ? p : 0) * PITCH + 64 * g2 + 4 * jj);
                bv[g2] = (ok && p < ROWS) ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int g = 0; g < FN; ++g)
#pragma unroll
                for (int fk = 0; fk < FK; ++fk)
                    aw[fk][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[fk], bv[g >> 2][g & 3], aw[fk][g], 0, 0, 0);
        }
#pragma unroll
        for (int fk = 0; fk < FK; ++fk)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float* dst = dw + ((size_t)n * K + 16 * fk + 4 * kk + e) * J + 4 * jj;
#pragma unroll
                for (int g2 = 0; g2 < G2; ++g2)
                    *reinterpret_cast<f32x4*>(dst + 64 * g2) = (f32x4){aw[fk][4 * g2][e], aw[fk][4 * g2 + 1][e], aw[fk][4 * g2 + 2][e], aw[fk][4 * g2 + 3][e]};
            }
    }
}

// ---- the same backward with FOUR wavefronts per anchor (one workgroup = one anchor) ----
// One wavefront per anchor is 1200 wavefronts of 1056 dependent-ish MFMAs each on 1024 SIMDs, with nothing to hide a load behind.
// Here waves 0-2 take one 16-row fragment each: product, LayerNorm / ReLU backward, dF rows into a workgroup-shared LDS image, and -
// after the one barrier - dX of those rows; the weight gradient sums over ALL rows, so its K/16 row blocks go to the four waves
// (one each at K = 64, two at K = 128), each reading the whole dF image; wave 3, idle before the barrier, folds the three
// fragments' LayerNorm affine partials.  4800 wavefronts per launch, 328 MFMAs on the longest.
template <int K, int J>
__global__ __launch_bounds__(256, 4) void dyn_mfma_bwd_split_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                    const float* __restrict__ w, const float* __restrict__ y,
                                                                    const float* __restrict__ stats, const float* __restrict__ gamma,
                                                                    float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ lnpart,
                                                                    int N, int P)
{
    constexpr int FN = J / 16, G2 = J / 64, FK = K / 16, JS = J / 4, PITCH = J + 4, ROWS = 48, FKW = FK / 4;
    extern __shared__ float lds[];                              // dF [ROWS][PITCH], then the affine partials [3][2][J]
    float* Fs = lds;
    float* lp = lds + ROWS * PITCH;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.x;
    const int jj = lane & 15, kk = lane >> 4;
    const float* xn = x + (size_t)n * P * K;
    const float* wn = w + (size_t)n * K * J;
    const int nfr = (P + 15) >> 4;                              // row fragments that hold rows
    if (wv < 3) {
        const int f = wv;
        f32x4 pw[G2], pb[G2];
#pragma unroll
        for (int g2 = 0; g2 < G2; ++g2) { pw[g2] = (f32x4){0.f, 0.f, 0.f, 0.f}; pb[g2] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        if (f < nfr) {                                          // (uniform)
            f32x4 acc[FN];
            product_frag<K, J>(xn, wn, P, lane, f, acc);
            f32x4 gam[G2];
#pragma unroll
            for (int g2 = 0; g2 < G2; ++g2) gam[g2] = *reinterpret_cast<const f32x4*>(gamma + 64 * g2 + 4 * jj);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 16 * f + 4 * kk + e;
                const bool ok = row < P;
                const size_t ro = ((size_t)n * P + (ok ? row : 0)) * J + 4 * jj;
                const float mu = stats[((size_t)n * P + (ok ? row : 0)) * 2], rs = stats[((size_t)n * P + (ok ? row : 0)) * 2 + 1];
                f32x4 gv[G2], xh[G2];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int g2 = 0; g2 < G2; ++g2) {
                    const f32x4 dyv = *reinterpret_cast<const f32x4*>(dy + ro + 64 * g2);
                    const f32x4 yv = *reinterpret_cast<const f32x4*>(y + ro + 64 * g2);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float gg = (ok && yv[t] > 0.f) ? dyv[t] : 0.f;
                        const float h = (acc[4 * g2 + t][e] - mu) * rs;
                        xh[g2][t] = h;
                        pw[g2][t] += gg * h;
                        pb[g2][t] += gg;
                        const float v = gg * gam[g2][t];
                        gv[g2][t] = v;
                        s1 += v; s2 += v * h;
                    }
                }
                s1 = group16_sum(s1) * (1.0f / (float)J);
                s2 = group16_sum(s2) * (1.0f / (float)J);
#pragma unroll
                for (int g2 = 0; g2 < G2; ++g2) {
                    f32x4 d;
#pragma unroll
                    for (int t = 0; t < 4; ++t) d[t] = ok ? rs * (gv[g2][t] - s1 - xh[g2][t] * s2) : 0.f;
                    *reinterpret_cast<f32x4*>(Fs + row * PITCH + 64 * g2 + 4 * jj) = d;           // dF, zero rows past P
                }
            }
        } else {                                                // a fragment without rows: its dF rows are zeros
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int g2 = 0; g2 < G2; ++g2)
                    *reinterpret_cast<f32x4*>(Fs + (16 * f + 4 * kk + e) * PITCH + 64 * g2 + 4 * jj) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // the fragment's LayerNorm affine partials: the lane's column sums over its 4 rows, then over the four row groups kk
#pragma unroll
        for (int g2 = 0; g2 < G2; ++g2)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float a = pw[g2][t], b = pb[g2][t];
                a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
                b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
                pw[g2][t] = a; pb[g2][t] = b;
            }
        if (kk == 0) {
#pragma unroll
            for (int g2 = 0; g2 < G2; ++g2) {
                *reinterpret_cast<f32x4*>(lp + (f * 2 + 0) * J + 64 * g2 + 4 * jj) = pw[g2];
                *reinterpret_cast<f32x4*>(lp + (f * 2 + 1) * J + 64 * g2 + 4 * jj) = pb[g2];
            }
        }
    }
    __syncthreads();
    if (wv == 3) {                                              // fold the three fragments' partials in a fixed order
        for (int i = lane; i < 2 * J; i += 64)
            lnpart[(size_t)n * 2 * J + i] = (lp[i] + lp[2 * J + i]) + lp[4 * J + i];
    } else if (dx && wv < nfr) {
        // ---- dX[p][k] = sum_j dF[p][j] * w[k][j], rows of fragment wv ----
        const int f = wv;
        float a[JS];
        {
            const int row = 16 * f + jj;
            const bool ok = row < P;
            const f32x4* src = reinterpret_cast<const f32x4*>(Fs + (ok ? row : 0) * PITCH + JS * kk);
#pragma unroll
            for (int t = 0; t < JS / 4; ++t) {
                const f32x4 v = src[t];
                a[4 * t] = ok ? v.x : 0.f; a[4 * t + 1] = ok ? v.y : 0.f; a[4 * t + 2] = ok ? v.z : 0.f; a[4 * t + 3] = ok ? v.w : 0.f;
            }
        }
#pragma unroll
        for (int g = 0; g < FK; ++g) {
            float b[JS];
            const f32x4* src = reinterpret_cast<const f32x4*>(wn + (size_t)(16 * g + jj) * J + JS * kk);
#pragma unroll
            for (int t = 0; t < JS / 4; ++t) { const f32x4 v = src[t]; b[4 * t] = v.x; b[4 * t + 1] = v.y; b[4 * t + 2] = v.z; b[4 * t + 3] = v.w; }
            f32x4 ax = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < JS; ++s) ax = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], ax, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 16 * f + 4 * kk + e;
                if (row < P) dx[((size_t)n * P + row) * K + 16 * g + jj] = ax[e];
            }
        }
    }
    // ---- dW[k][j] = sum_p x[p][k] * dF[p][j]: row blocks fk = FKW wv .. of this wave, all points ----
    {
        const int RS = (P + 3) >> 2;                             // reduction steps: p = RS * kk + s
        f32x4 aw[FKW][FN];
#pragma unroll
        for (int i = 0; i < FKW; ++i)
#pragma unroll
            for (int g = 0; g < FN; ++g) aw[i][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < RS; ++s) {
            const int p = RS * kk + s;
            const bool ok = p < P;
            float av[FKW];
#pragma unroll
            for (int i = 0; i < FKW; ++i) { const float v = xn[(size_t)(ok ? p : 0) * K + 16 * (FKW * wv + i) + jj]; av[i] = ok ? v : 0.f; }
            f32x4 bv[G2];
#pragma unroll
            for (int g2 = 0; g2 < G2; ++g2) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(Fs + (ok ? p : 0) * PITCH + 64 * g2 + 4 * jj);
                bv[g2] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int g = 0; g < FN; ++g)
#pragma unroll
                for (int i = 0; i < FKW; ++i)
                    aw[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[g >> 2][g & 3], aw[i][g], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < FKW; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float* dst = dw + ((size_t)n * K + 16 * (FKW * wv + i) + 4 * kk + e) * J + 4 * jj;
#pragma unroll
                for (int g2 = 0; g2 < G2; ++g2)
                    *reinterpret_cast<f32x4*>(dst + 64 * g2) = (f32x4){aw[i][4 * g2][e], aw[i][4 * g2 + 1][e], aw[i][4 * g2 + 2][e], aw[i][4 * g2 + 3][e]};
            }
    }
}

}  // namespace

int g_dyn_rows = 1;             // forward: one wavefront per (anchor, row fragment); backward: four wavefronts per anchor; phnet_tune_dyn_mfma(2 | ...) switches both off

PHNET_API int phnet_dyn_mfma_applies(int32_t P, int32_t K, int32_t J)
{
    return P >= 1 && P <= 36 && ((K == 64 && J == 128) || (K == 128 && J == 64));
}

// the forward of phnet_dyn_bmm_ln_relu_fwd (dynhead.hip documents the arguments) on the matrix pipe, one wavefront per anchor
PHNET_API int phnet_dyn_mfma_fwd(const float* x, const float* w, const float* gamma, const float* beta, float* y, float* stats,
                                 int32_t N, int32_t P, int32_t K, int32_t J, float eps, void* stream)
{
    if (N < 1 || !phnet_dyn_mfma_applies(P, K, J) || !x || !w || !gamma || !beta || !y) return PHNET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (g_dyn_rows) {                                           // one wavefront per (anchor, row fragment)
        const int nfr = (P + 15) / 16;
        const dim3 grid((unsigned)(((long)N * nfr + 3) / 4));
        if (K == 64) hipLaunchKernelGGL((dyn_mfma_fwd_rows_kernel<64, 128>), grid, dim3(256), 0, st, x, w, gamma, beta, y, stats, N, P, nfr, eps);
        else hipLaunchKernelGGL((dyn_mfma_fwd_rows_kernel<128, 64>), grid, dim3(256), 0, st, x, w, gamma, beta, y, stats, N, P, nfr, eps);
        return phnet_launch_status();
    }
    const dim3 grid((unsigned)((N + 3) / 4));
    if (K == 64) hipLaunchKernelGGL((dyn_mfma_fwd_kernel<64, 128>), grid, dim3(256), 0, st, x, w, gamma, beta, y, stats, N, P, eps);
    else hipLaunchKernelGGL((dyn_mfma_fwd_kernel<128, 64>), grid, dim3(256), 0, st, x, w, gamma, beta, y, stats, N, P, eps);
    return phnet_launch_status();
}

// the backward kernel of phnet_dyn_bmm_ln_relu_bwd on the matrix pipe (the caller runs the shared LayerNorm-gradient reduce):
// dx (optional) [N][P][K], dw [N][K][J], lnpart [N][2][J]
PHNET_API int phnet_dyn_mfma_bwd(const float* dy, const float* x, const float* w, const float* y, const float* stats, const float* gamma,
                                 float* dx, float* dw, float* lnpart, int32_t N, int32_t P, int32_t K, int32_t J, void* stream)
{
    if (N < 1 || !phnet_dyn_mfma_applies(P, K, J) || !dy || !x || !w || !y || !stats || !gamma || !dw || !lnpart) return PHNET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (g_dyn_rows) {                                           // one workgroup (four wavefronts) per anchor
        const size_t lds4 = ((size_t)48 * (J + 4) + 6 * J) * sizeof(float);
        if (K == 64) hipLaunchKernelGGL((dyn_mfma_bwd_split_kernel<64, 128>), dim3((unsigned)N), dim3(256), lds4, st, dy, x, w, y, stats, gamma, dx, dw, lnpart, N, P);
        else hipLaunchKernelGGL((dyn_mfma_bwd_split_kernel<128, 64>), dim3((unsigned)N), dim3(256), lds4, st, dy, x, w, y, stats, gamma, dx, dw, lnpart, N, P);
        return phnet_launch_status();
    }
    const dim3 grid((unsigned)((N + 3) / 4));
    const size_t lds = (size_t)4 * 36 * (J + 4) * sizeof(float);
    static bool attr_a = false, attr_b = false;
    if (K == 64) {
        if (!attr_a) { (void)hipFuncSetAttribute((const void*)dyn_mfma_bwd_kernel<64, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_a = true; }
        hipLaunchKernelGGL((dyn_mfma_bwd_kernel<64, 128>), grid, dim3(256), lds, st, dy, x, w, y, stats, gamma, dx, dw, lnpart, N, P);
    } else {
        if (!attr_b) { (void)hipFuncSetAttribute((const void*)dyn_mfma_bwd_kernel<128, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_b = true; }
        hipLaunchKernelGGL((dyn_mfma_bwd_kernel<128, 64>), grid, dim3(256), lds, st, dy, x, w, y, stats, gamma, dx, dw, lnpart, N, P);
    }
    return phnet_launch_status();
}
