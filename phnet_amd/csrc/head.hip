// Lane-head normalisation / gate kernels for gfx950: LayerNorm (+residual)(+ReLU) forward/backward and the
// per-anchor depth-wise 3x3 convolution of the adaptive routing gate, forward/backward.
//
// Replaces the ATen ops behind
//   libs/models/Router.py:72-81      (LayerNorm([C,P]) -> 4 x relu(DWblock(x)+x); Conv2d(N,N,3,padding=1,groups=N))
//   libs/models/utils/dynamic_head.py:42-58  (norm1/norm2 + ReLU, norm3)
//   libs/models/utils/transformer.py:275-298 (pre-norm LayerNorms)
// All HBM/LDS-bound: one wavefront per LayerNorm row (wave shuffles for the two reductions), one workgroup
// per anchor plane for the depth-wise filter (plane staged in LDS once, 9 taps from LDS).
#include "common.h"

namespace {

constexpr int NT = 256;

// ---- LayerNorm over the last L elements: y = relu?( (x-mu)*rstd*w + b (+res) ) ---------------------------
__global__ __launch_bounds__(NT) void layernorm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ res,
    float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, long rows, int L, float eps, int relu)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * L;
    float s = 0.f;
    for (int i = lane; i < L; i += 64) s += xr[i];
    const float mu = wave_sum(s) / (float)L;
    float v = 0.f;
    for (int i = lane; i < L; i += 64) { const float d = xr[i] - mu; v += d * d; }
    const float rs = 1.0f / sqrtf(wave_sum(v) / (float)L + eps);
    if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
    float* yr = y + (size_t)row * L;
    const float* rr = res ? res + (size_t)row * L : nullptr;
    for (int i = lane; i < L; i += 64) {
        float o = (xr[i] - mu) * rs * w[i] + b[i];
        if (rr) o += rr[i];
        if (relu) o = fmaxf(o, 0.f);
        yr[i] = o;
    }
}

// g = dy * (y>0 if relu); dx = rstd*(g*w - mean(g*w) - xhat*mean(g*w*xhat)); dres = g
__global__ __launch_bounds__(NT) void layernorm_bwd_dx_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ w,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx, float* __restrict__ dres,
    long rows, int L, int relu)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const size_t o = (size_t)row * L;
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < L; i += 64) {
        float g = dy[o + i];
        if (relu && !(y[o + i] > 0.f)) g = 0.f;
        const float gw = g * w[i];
        s1 += gw;
        s2 += gw * ((x[o + i] - mu) * rs);
    }
    s1 = wave_sum(s1) / (float)L;
    s2 = wave_sum(s2) / (float)L;
    for (int i = lane; i < L; i += 64) {
        float g = dy[o + i];
        if (relu && !(y[o + i] > 0.f)) g = 0.f;
        const float xh = (x[o + i] - mu) * rs;
        dx[o + i] = rs * (g * w[i] - s1 - xh * s2);
        if (dres) dres[o + i] = g;
    }
}

// ---- long rows (gate: L = C*P = 2304): one 256-thread workgroup per row, row kept in registers ----------------
constexpr int LONG_MAX_PER_THREAD = 16;          // rows up to 4096 elements

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(NT) void layernorm_fwd_long_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ res,
    float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int L, float eps, int relu)
{
    __shared__ float red[4];
    const long row = blockIdx.x;
    const float* xr = x + (size_t)row * L;
    float v[LONG_MAX_PER_THREAD];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LONG_MAX_PER_THREAD; ++k) {
        const int i = threadIdx.x + k * NT;
        v[k] = i < L ? xr[i] : 0.f;
        s += v[k];
    }
    const float mu = block_sum(s, red) / (float)L;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LONG_MAX_PER_THREAD; ++k) {
        const int i = threadIdx.x + k * NT;
        const float d = i < L ? v[k] - mu : 0.f;
        q += d * d;
    }
    const float rs = 1.0f / sqrtf(block_sum(q, red) / (float)L + eps);
    if (threadIdx.x == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
    const float* rr = res ? res + (size_t)row * L : nullptr;
#pragma unroll
    for (int k = 0; k < LONG_MAX_PER_THREAD; ++k) {
        const int i = threadIdx.x + k * NT;
        if (i < L) {
            float o = (v[k] - mu) * rs * w[i] + b[i];
            if (rr) o += rr[i];
            if (relu) o = fmaxf(o, 0.f);
            y[(size_t)row * L + i] = o;
        }
    }
}

__global__ __launch_bounds__(NT) void layernorm_bwd_dx_long_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ w,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx, float* __restrict__ dres,
    int L, int relu)
{
    __shared__ float red[4];
    const long row = blockIdx.x;
    const size_t o = (size_t)row * L;
    const float mu = mean[row], rs = rstd[row];
    float g[LONG_MAX_PER_THREAD], xh[LONG_MAX_PER_THREAD];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < LONG_MAX_PER_THREAD; ++k) {
        const int i = threadIdx.x + k * NT;
        g[k] = 0.f; xh[k] = 0.f;
        if (i < L) {
            float gg = dy[o + i];
            if (relu && !(y[o + i] > 0.f)) gg = 0.f;
            g[k] = gg;
            xh[k] = (x[o + i] - mu) * rs;
            const float gw = gg * w[i];
            s1 += gw;
            s2 += gw * xh[k];
        }
    }
    s1 = block_sum(s1, red) / (float)L;
    s2 = block_sum(s2, red) / (float)L;
#pragma unroll
    for (int k = 0; k < LONG_MAX_PER_THREAD; ++k) {
        const int i = threadIdx.x + k * NT;
        if (i < L) {
            dx[o + i] = rs * (g[k] * w[i] - s1 - xh[k] * s2);
            if (dres) dres[o + i] = g[k];
        }
    }
}

// Short rows (L <= 256: the transformer / dynamic-head LayerNorms): dx AND the slab's affine-gradient partials in one pass.
// One wavefront per row as above; a lane owns columns lane + 64u and carries their partial sums over the wave's rows,
// the four waves of the workgroup are folded through LDS.  partial layout as below.
constexpr int SHORT_MAX_L = 256;
__global__ __launch_bounds__(NT) void layernorm_bwd_short_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ w,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx, float* __restrict__ dres,
    float* __restrict__ partial, long rows, int L, long rows_per_slab, int relu)
{
    __shared__ float red[2][NT / 64][SHORT_MAX_L];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long a = (long)blockIdx.x * rows_per_slab, e = min(rows, a + rows_per_slab);
    float pw[SHORT_MAX_L / 64], pb[SHORT_MAX_L / 64];
#pragma unroll
    for (int u = 0; u < SHORT_MAX_L / 64; ++u) { pw[u] = 0.f; pb[u] = 0.f; }
    for (long row = a + wave; row < e; row += NT / 64) {
        const size_t o = (size_t)row * L;
        const float mu = mean[row], rs = rstd[row];
        float g[SHORT_MAX_L / 64], xh[SHORT_MAX_L / 64], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int u = 0; u < SHORT_MAX_L / 64; ++u) {
            const int i = lane + 64 * u;
            g[u] = 0.f; xh[u] = 0.f;
            if (i < L) {
                float gg = dy[o + i];
                if (relu && !(y[o + i] > 0.f)) gg = 0.f;
                g[u] = gg;
                xh[u] = (x[o + i] - mu) * rs;
                const float gw = gg * w[i];
                s1 += gw;
                s2 += gw * xh[u];
                pw[u] += gg * xh[u];
                pb[u] += gg;
            }
        }
        s1 = wave_sum(s1) / (float)L;
        s2 = wave_sum(s2) / (float)L;
#pragma unroll
        for (int u = 0; u < SHORT_MAX_L / 64; ++u) {
            const int i = lane + 64 * u;
            if (i < L) {
                dx[o + i] = rs * (g[u] * w[i] - s1 - xh[u] * s2);
                if (dres) dres[o + i] = g[u];
            }
        }
    }
    if (!partial) return;
#pragma unroll
    for (int u = 0; u < SHORT_MAX_L / 64; ++u) { red[0][wave][lane + 64 * u] = pw[u]; red[1][wave][lane + 64 * u] = pb[u]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * L; i += NT) {
        const int which = i / L, c = i - which * L;
        partial[((size_t)blockIdx.x * 2 + which) * L + c] = (red[which][0][c] + red[which][1][c]) + (red[which][2][c] + red[which][3][c]);
    }
}

// ---- residual + dropout + LayerNorm of the pre-norm transformer (utils/transformer.py:275-298), L <= 256 ----------------
//   t = res + dropout(x);  h = LayerNorm(t) * w + b       -> both written (t is the running stream, h feeds the next block)
// The dropout bit of element i of the [rows][L] array comes from the counter-based generator of common.h.
__global__ __launch_bounds__(NT) void dropout_add_ln_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ w, const float* __restrict__ b,
    float* __restrict__ t_out, float* __restrict__ h_out, float* __restrict__ mean, float* __restrict__ rstd,
    long rows, int L, float eps, DropRng rng, float scale)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const size_t o = (size_t)row * L;
    const uint64_t seed = phnet_rng_seed(rng);
    float t[SHORT_MAX_L / 64], s = 0.f;
#pragma unroll
    for (int u = 0; u < SHORT_MAX_L / 64; ++u) {
        const int i = lane + 64 * u;
        t[u] = 0.f;
        if (i < L) {
            const bool kept = !rng.thresh || phnet_rng_keep(seed, phnet_rng_index(rng, (uint64_t)(o + i)), rng.thresh);
            t[u] = res[o + i] + (kept ? x[o + i] * scale : 0.f);
            t_out[o + i] = t[u];
            s += t[u];
        }
    }
    const float mu = wave_sum(s) / (float)L;
    float v = 0.f;
#pragma unroll
    for (int u = 0; u < SHORT_MAX_L / 64; ++u) {
        const int i = lane + 64 * u;
        if (i < L) { const float d = t[u] - mu; v += d * d; }
    }
    const float rs = 1.0f / sqrtf(wave_sum(v) / (float)L + eps);
    if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int u = 0; u < SHORT_MAX_L / 64; ++u) {
        const int i = lane + 64 * u;
        if (i < L) h_out[o + i] = (t[u] - mu) * rs * w[i] + b[i];
    }
}

// backward: g = LayerNorm'(dh) + dt (dt optional);  dres = g;  dx = dropout'(g);  affine partials as in the short kernel
__global__ __launch_bounds__(NT) void dropout_add_ln_bwd_kernel(
    const float* __restrict__ dh, const float* __restrict__ dt, const float* __restrict__ t, const float* __restrict__ w,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dres, float* __restrict__ dx,
    float* __restrict__ partial, long rows, int L, long rows_per_slab, DropRng rng, float scale)
{
    __shared__ float red[2][NT / 64][SHORT_MAX_L];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long a = (long)blockIdx.x * rows_per_slab, e = min(rows, a + rows_per_slab);
    const uint64_t seed = phnet_rng_seed(rng);
    float pw[SHORT_MAX_L / 64], pb[SHORT_MAX_L / 64];
#pragma unroll
    for (int u = 0; u < SHORT_MAX_L / 64; ++u) { pw[u] = 0.f; pb[u] = 0.f; }
    for (long row = a + wave; row < e; row += NT / 64) {
        const size_t o = (size_t)row * L;
        const float mu = mean[row], rs = rstd[row];
        float g[SHORT_MAX_L / 64], xh[SHORT_MAX_L / 64], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int u = 0; u < SHORT_MAX_L / 64; ++u) {
            const int i = lane + 64 * u;
            g[u] = 0.f; xh[u] = 0.f;
            if (i < L) {
                g[u] = dh[o + i];
                xh[u] = (t[o + i] - mu) * rs;
                const float gw = g[u] * w[i];
                s1 += gw;
                s2 += gw * xh[u];
                pw[u] += g[u] * xh[u];
                pb[u] += g[u];
            }
        }
        s1 = wave_sum(s1) / (float)L;
        s2 = wave_sum(s2) / (float)L;
#pragma unroll
        for (int u = 0; u < SHORT_MAX_L / 64; ++u) {
            const int i = lane + 64 * u;
            if (i < L) {
                float gt = rs * (g[u] * w[i] - s1 - xh[u] * s2);
                if (dt) gt += dt[o + i];
                dres[o + i] = gt;
                const bool kept = !rng.thresh || phnet_rng_keep(seed, phnet_rng_index(rng, (uint64_t)(o + i)), rng.thresh);
                dx[o + i] = kept ? gt * scale : 0.f;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < SHORT_MAX_L / 64; ++u) { red[0][wave][lane + 64 * u] = pw[u]; red[1][wave][lane + 64 * u] = pb[u]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * L; i += NT) {
        const int which = i / L, c = i - which * L;
        partial[((size_t)blockIdx.x * 2 + which) * L + c] = (red[which][0][c] + red[which][1][c]) + (red[which][2][c] + red[which][3][c]);
    }
}

// partial[slab][0][L] = sum_rows g*xhat, partial[slab][1][L] = sum_rows g   (rows of this slab)
__global__ __launch_bounds__(NT) void layernorm_bwd_param_partial_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ partial,
    long rows, int L, long rows_per_slab, int relu)
{
    __shared__ float r0[NT], r1[NT];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const long a = (long)blockIdx.y * rows_per_slab, e = min(rows, a + rows_per_slab);
    float sw = 0.f, sb = 0.f;
    if (col < L)
        for (long r = a + rl; r < e; r += NT / 64) {
            const size_t o = (size_t)r * L + col;
            float g = dy[o];
            if (relu && !(y[o] > 0.f)) g = 0.f;
            sw += g * ((x[o] - mean[r]) * rstd[r]);
            sb += g;
        }
    r0[threadIdx.x] = sw; r1[threadIdx.x] = sb;
    __syncthreads();
    if (rl == 0 && col < L) {
        const int t = threadIdx.x;
        partial[((size_t)blockIdx.y * 2 + 0) * L + col] = (r0[t] + r0[t + 64]) + (r0[t + 128] + r0[t + 192]);
        partial[((size_t)blockIdx.y * 2 + 1) * L + col] = (r1[t] + r1[t + 64]) + (r1[t + 128] + r1[t + 192]);
    }
}

__global__ void layernorm_bwd_param_finalize_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                    float* __restrict__ db, int slabs, int L, int accumulate)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= L) return;
    const float old_w = accumulate ? dw[c] : 0.f, old_b = accumulate ? db[c] : 0.f;      // cold reads first: they overlap the sums
    double sw = 0.0, sb = 0.0;
#pragma unroll 8
    for (int s = 0; s < slabs; ++s) {                 // independent loads: unrolled so that they are in flight together
        sw += (double)partial[((size_t)s * 2 + 0) * L + c];
        sb += (double)partial[((size_t)s * 2 + 1) * L + c];
    }
    dw[c] = old_w + (float)sw;
    db[c] = old_b + (float)sb;
}

// ---- depth-wise 3x3 over each anchor's (C x P) plane, zero padding 1 --------------------------------------
// plane layout [C][P]: element (c, p) at c*P + p (the reference's [1,N,C,P]; the ROI pooling kernel emits this
// copy for the gate); filter w[n][i][j] with i along c and j along p (PyTorch's [N,1,3,3]), bias[n].
// flip=1 correlates with the 180-degree rotated filter (data gradient).
__global__ __launch_bounds__(NT) void dwconv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       int C, int P, int flip)
{
    extern __shared__ float plane[];
    const int n = blockIdx.x;
    const int CP = C * P;
    const float* xp = x + (size_t)n * CP;
    for (int i = threadIdx.x; i < CP; i += NT) plane[i] = xp[i];
    float f[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) f[k] = w[n * 9 + (flip ? 8 - k : k)];
    const float bv = bias ? bias[n] : 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < CP; i += NT) {
        const int c = i / P, p = i - c * P;
        float acc = bv;
#pragma unroll
        for (int di = 0; di < 3; ++di) {
            const int cc = c + di - 1;
            if (cc < 0 || cc >= C) continue;
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) {
                const int pp = p + dj - 1;
                if (pp < 0 || pp >= P) continue;
                acc += f[di * 3 + dj] * plane[cc * P + pp];
            }
        }
        y[(size_t)n * CP + i] = acc;
    }
}

// dw[n][i][j] = sum_{c,p} dy(c,p) * x(c+i-1, p+j-1);  db[n] = sum dy
__global__ __launch_bounds__(NT) void dwconv3x3_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             float* __restrict__ dw, float* __restrict__ db,
                                                             int C, int P, int accumulate)
{
    extern __shared__ float plane[];
    __shared__ float red[10][NT / 64];
    const int n = blockIdx.x;
    const int CP = C * P;
    for (int i = threadIdx.x; i < CP; i += NT) plane[i] = x[(size_t)n * CP + i];
    __syncthreads();
    float acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.f;
    for (int i = threadIdx.x; i < CP; i += NT) {
        const int c = i / P, p = i - c * P;
        const float g = dy[(size_t)n * CP + i];
        acc[9] += g;
#pragma unroll
        for (int di = 0; di < 3; ++di) {
            const int cc = c + di - 1;
            if (cc < 0 || cc >= C) continue;
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) {
                const int pp = p + dj - 1;
                if (pp < 0 || pp >= P) continue;
                acc[di * 3 + dj] += g * plane[cc * P + pp];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const float v = wave_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        const float v = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
        float* dst = threadIdx.x < 9 ? dw + n * 9 + threadIdx.x : db + n;
        *dst = accumulate ? *dst + v : v;
    }
}

}  // namespace

// y[rows][L] = relu?( LayerNorm(x) * w + b (+ res) ); mean/rstd [rows] saved for the backward (may be NULL).
PHNET_API int phnet_layernorm_fwd(const float* x, const float* w, const float* b, const float* res, float* y,
                                  float* mean, float* rstd, int64_t rows, int32_t L, float eps, int32_t relu, void* stream)
{
    if (rows < 0 || L < 1) return PHNET_ERR_ARG;
    if (rows == 0) return PHNET_OK;
    if (!x || !w || !b || !y || (!!mean != !!rstd)) return PHNET_ERR_ARG;
    if (L >= 1024 && L <= NT * LONG_MAX_PER_THREAD)
        hipLaunchKernelGGL(layernorm_fwd_long_kernel, dim3((unsigned)rows), dim3(NT), 0, (hipStream_t)stream,
                           x, w, b, res, y, mean, rstd, L, eps, relu);
    else
        hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)ceil_div64(rows, NT / 64)), dim3(NT), 0, (hipStream_t)stream,
                           x, w, b, res, y, mean, rstd, (long)rows, L, eps, relu);
    return phnet_launch_status();
}

PHNET_API uint64_t phnet_layernorm_bwd_workspace(int64_t rows, int32_t L)
{
    (void)rows;
    return (uint64_t)((long)128 * 2 * L * sizeof(float));        // at most 128 row slabs of (weight, bias) partials
}

// dy: gradient of the output (after the optional ReLU whose mask is y > 0).  dx overwritten; dres (optional)
// overwritten with the masked gradient (residual branch); dw/db [L] overwritten or accumulated.
PHNET_API int phnet_layernorm_bwd(const float* dy, const float* x, const float* y, const float* w,
                                  const float* mean, const float* rstd, float* dx, float* dres, float* dw, float* db,
                                  int64_t rows, int32_t L, int32_t relu, int32_t param_accumulate,
                                  void* workspace, uint64_t ws_bytes, void* stream)
{
    if (rows < 0 || L < 1) return PHNET_ERR_ARG;
    if (rows == 0) return PHNET_OK;
    if (!dy || !x || !w || !mean || !rstd || !dx || (relu && !y)) return PHNET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (L <= SHORT_MAX_L) {                               // dx + affine partials in one launch, then the finalize
        const bool params = dw && db;
        const long slabs = max((long)1, min((long)128, ceil_div64(rows, 8)));
        if (params && (!workspace || (uint64_t)(slabs * 2 * L * sizeof(float)) > ws_bytes)) return PHNET_ERR_WORKSPACE;
        hipLaunchKernelGGL(layernorm_bwd_short_kernel, dim3((unsigned)slabs), dim3(NT), 0, st, dy, x, y, w, mean, rstd, dx, dres,
                           params ? (float*)workspace : nullptr, (long)rows, L, ceil_div64(rows, slabs), relu);
        if (params)
            hipLaunchKernelGGL(layernorm_bwd_param_finalize_kernel, dim3((L + 255) / 256), dim3(256), 0, st,
                               (const float*)workspace, dw, db, (int)slabs, L, param_accumulate);
        return phnet_launch_status();
    }
    if (L >= 1024 && L <= NT * LONG_MAX_PER_THREAD)
        hipLaunchKernelGGL(layernorm_bwd_dx_long_kernel, dim3((unsigned)rows), dim3(NT), 0, st,
                           dy, x, y, w, mean, rstd, dx, dres, L, relu);
    else
        hipLaunchKernelGGL(layernorm_bwd_dx_kernel, dim3((unsigned)ceil_div64(rows, NT / 64)), dim3(NT), 0, st,
                           dy, x, y, w, mean, rstd, dx, dres, (long)rows, L, relu);
    if (dw && db) {
        const long slabs = max((long)1, min((long)128, (long)rows / 32));
        if (!workspace || (uint64_t)(slabs * 2 * L * sizeof(float)) > ws_bytes) return PHNET_ERR_WORKSPACE;
        const long rps = ceil_div64(rows, slabs);
        hipLaunchKernelGGL(layernorm_bwd_param_partial_kernel, dim3((L + 63) / 64, (unsigned)slabs), dim3(NT), 0, st,
                           dy, x, y, mean, rstd, (float*)workspace, (long)rows, L, rps, relu);
        hipLaunchKernelGGL(layernorm_bwd_param_finalize_kernel, dim3((L + 255) / 256), dim3(256), 0, st,
                           (const float*)workspace, dw, db, (int)slabs, L, param_accumulate);
    }
    return phnet_launch_status();
}

// x,y [N][C][P] planes; w [N][3][3] (i along C, j along P); bias [N] or NULL; flip=1 -> data gradient.
PHNET_API int phnet_dwconv3x3(const float* x, const float* w, const float* bias, float* y,
                              int32_t N, int32_t C, int32_t P, int32_t flip, void* stream)
{
    if (N < 0 || C < 1 || P < 1 || (size_t)C * P * 4 > 64 * 1024) return PHNET_ERR_ARG;
    if (N == 0) return PHNET_OK;
    if (!x || !w || !y) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(dwconv3x3_kernel, dim3(N), dim3(NT), (size_t)C * P * 4, (hipStream_t)stream, x, w, bias, y, C, P, flip);
    return phnet_launch_status();
}

PHNET_API int phnet_dwconv3x3_wgrad(const float* dy, const float* x, float* dw, float* db,
                                    int32_t N, int32_t C, int32_t P, int32_t accumulate, void* stream)
{
    if (N < 0 || C < 1 || P < 1 || (size_t)C * P * 4 > 60 * 1024) return PHNET_ERR_ARG;
    if (N == 0) return PHNET_OK;
    if (!dy || !x || !dw || !db) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(dwconv3x3_wgrad_kernel, dim3(N), dim3(NT), (size_t)C * P * 4, (hipStream_t)stream, dy, x, dw, db, C, P,
                       accumulate);
    return phnet_launch_status();
}

// t = res + dropout(x), h = LayerNorm_L(t) * w + b in one launch (L <= 256); mean/rstd [rows] saved for the backward.
PHNET_API int phnet_dropout_add_ln_fwd(const float* x, const float* res, const float* w, const float* b, float* t, float* h,
                                       float* mean, float* rstd, int64_t rows, int32_t L, float eps,
                                       const uint64_t* rng_state, uint64_t rng_call, float drop_p, void* stream)
{
    if (rows < 0 || L < 1 || L > SHORT_MAX_L || drop_p < 0.f || drop_p >= 1.f) return PHNET_ERR_ARG;
    if (rows == 0) return PHNET_OK;
    if (!x || !res || !w || !b || !t || !h) return PHNET_ERR_ARG;
    const DropRng rng = phnet_make_rng(rng_state, rng_call, drop_p);
    hipLaunchKernelGGL(dropout_add_ln_fwd_kernel, dim3((unsigned)ceil_div64(rows, NT / 64)), dim3(NT), 0, (hipStream_t)stream,
                       x, res, w, b, t, h, mean, rstd, (long)rows, L, eps, rng, rng.thresh ? 1.0f / (1.0f - drop_p) : 1.0f);
    return phnet_launch_status();
}

// Backward of phnet_dropout_add_ln_fwd: dh = dL/dh, dt = dL/dt from the residual stream (may be NULL) ->
// dres [rows][L] (gradient of res), dx (gradient of x, same dropout bits), dw/db [L] overwritten or accumulated.
// workspace: phnet_layernorm_bwd_workspace(rows, L) bytes.
PHNET_API int phnet_dropout_add_ln_bwd(const float* dh, const float* dt, const float* t, const float* w, const float* mean,
                                       const float* rstd, float* dres, float* dx, float* dw, float* db,
                                       int64_t rows, int32_t L, int32_t param_accumulate,
                                       const uint64_t* rng_state, uint64_t rng_call, float drop_p,
                                       void* workspace, uint64_t ws_bytes, void* stream)
{
    if (rows < 0 || L < 1 || L > SHORT_MAX_L || drop_p < 0.f || drop_p >= 1.f) return PHNET_ERR_ARG;
    if (rows == 0) return PHNET_OK;
    if (!dh || !t || !w || !mean || !rstd || !dres || !dx || !dw || !db || !workspace) return PHNET_ERR_ARG;
    const long slabs = max((long)1, min((long)128, ceil_div64(rows, 8)));
    if ((uint64_t)(slabs * 2 * L * sizeof(float)) > ws_bytes) return PHNET_ERR_WORKSPACE;
    const DropRng rng = phnet_make_rng(rng_state, rng_call, drop_p);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dropout_add_ln_bwd_kernel, dim3((unsigned)slabs), dim3(NT), 0, st, dh, dt, t, w, mean, rstd, dres, dx,
                       (float*)workspace, (long)rows, L, ceil_div64(rows, slabs), rng, rng.thresh ? 1.0f / (1.0f - drop_p) : 1.0f);
    hipLaunchKernelGGL(layernorm_bwd_param_finalize_kernel, dim3((L + 255) / 256), dim3(256), 0, st,
                       (const float*)workspace, dw, db, (int)slabs, L, param_accumulate);
    return phnet_launch_status();
}
