// BatchNorm (training + inference), max-pool and FPN top-down kernels on NHWC fp32 maps for gfx950.
// All HBM-bound streaming kernels: 16-byte vector accesses, channel index = fastest dimension.
//
// Replaces the ATen kernels behind  nn.BatchNorm2d / ReLU / residual add (libs/models/resnet.py:79-95,
// 293-297), nn.MaxPool2d(3,2,1) (resnet.py:217,297) and the FPN nearest-upsample add
// (libs/models/fpn.py:127-141), forward and backward.
//
// Batch statistics are reduced in two deterministic steps (per-block partial sums in fp32, final
// reduction in fp64), so repeated runs are bit-identical - no float atomics.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 256;

// ---- per-channel partial sums of (a, a*b) or (a, a*a) over a slab of rows -----------------------------
// mode 0: (sum x, sum x^2)                                  (BN forward statistics)
// mode 1: (sum g, sum g*xhat), g = dy * (y > 0 if relu)     (BN backward reductions)
template <int MODE>
__global__ __launch_bounds__(NT) void channel_partials_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    float* __restrict__ partial, long M, int C, long rows_per_block, int relu)
{
    extern __shared__ f32x4 red[];                         // [2][NT]
    const int lanes_per_row = C >> 2;
    const int rows_at_once = NT / lanes_per_row;
    const int cl = threadIdx.x % lanes_per_row, rl = threadIdx.x / lanes_per_row;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) {
        mu = *reinterpret_cast<const f32x4*>(mean + cl * 4);
        is = *reinterpret_cast<const f32x4*>(invstd + cl * 4);
    }
    for (long r = r0 + rl; r < r1; r += rows_at_once) {
        const size_t o = (size_t)r * C + cl * 4;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + o);
        if (MODE == 0) {
            s0 += xv;
            s1 += xv * xv;
        } else {
            f32x4 g = *reinterpret_cast<const f32x4*>(dy + o);
            if (relu) {
                const f32x4 yv = *reinterpret_cast<const f32x4*>(y + o);
                g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
                g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
            }
            s0 += g;
            s1 += g * ((xv - mu) * is);
        }
    }
    red[threadIdx.x] = s0;
    red[NT + threadIdx.x] = s1;
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < rows_at_once; ++k) {
            s0 += red[k * lanes_per_row + cl];
            s1 += red[NT + k * lanes_per_row + cl];
        }
        float* p = partial + (size_t)blockIdx.x * 2 * C;
        *reinterpret_cast<f32x4*>(p + cl * 4) = s0;
        *reinterpret_cast<f32x4*>(p + C + cl * 4) = s1;
    }
}

// one 64-lane wave per channel: lanes stride over the per-block partial sums (fp64), shuffle-reduce, lane 0 finishes
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Column sums of the per-block partials [nblk][2][C] in fp64, for FIN_CH channels per workgroup: thread (channel cl, row group g)
// walks the rows g, g + FIN_G, ... (a wave reads FIN_CH consecutive channels of 8 rows per step), the FIN_G partial sums of a
// channel are folded through LDS in a fixed tree order (bit-reproducible).  The one-wave-per-channel form this replaces read
// 4 bytes per 512-byte row stride and needed 15-19 us for the 1250-2500 partial rows of the stem / layer1 BatchNorms.
constexpr int FIN_CH = 8, FIN_NT = 1024, FIN_G = FIN_NT / FIN_CH;

__device__ __forceinline__ void finalize_column_sums(const float* __restrict__ partial, int nblk, int C, double* red, double& s0, double& s1)
{
    const int cl = threadIdx.x % FIN_CH, g = threadIdx.x / FIN_CH;
    const int c = blockIdx.x * FIN_CH + cl;
    double a = 0.0, b = 0.0;
    if (c < C) {
        // eight rows per trip, all 16 loads issued before the first add: the loop is a chain of memory latencies otherwise, and
        // the partial rows of a trunk layer (<= 2500) are then covered by one or two trips (rows beyond nblk re-read row g and are
        // weighted 0: branch-free)
        constexpr int U = 8;
        for (int r = g; r < nblk; r += U * FIN_G) {
            float v[U], w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = r + u * FIN_G;
                const size_t o = (size_t)(rr < nblk ? rr : g) * 2 * C + c;
                v[u] = partial[o];
                w[u] = partial[o + C];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool in = r + u * FIN_G < nblk;
                a += in ? (double)v[u] : 0.0;
                b += in ? (double)w[u] : 0.0;
            }
        }
    }
    red[threadIdx.x] = a;
    red[FIN_NT + threadIdx.x] = b;
    __syncthreads();
    for (int half = FIN_G / 2; half > 0; half >>= 1) {
        if (g < half) {
            red[threadIdx.x] += red[threadIdx.x + half * FIN_CH];
            red[FIN_NT + threadIdx.x] += red[FIN_NT + threadIdx.x + half * FIN_CH];
        }
        __syncthreads();
    }
    s0 = red[cl];
    s1 = red[FIN_NT + cl];
}

__global__ __launch_bounds__(FIN_NT) void bn_fwd_finalize_kernel(
    const float* __restrict__ partial, int nblk, long M, int C, float eps,
    float momentum, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ running_mean, float* __restrict__ running_var,
    float* __restrict__ save_mean, float* __restrict__ save_invstd,
    float* __restrict__ scale, float* __restrict__ shift, int training)
{
    __shared__ double red[2 * FIN_NT];
    const int c = blockIdx.x * FIN_CH + threadIdx.x % FIN_CH;
    float mean = 0.f, invstd = 0.f;
    if (training) {
        double s, ss;
        finalize_column_sums(partial, nblk, C, red, s, ss);
        if (threadIdx.x >= FIN_CH || c >= C) return;
        const double mu = s / (double)M;
        double var = ss / (double)M - mu * mu;
        var = var < 0.0 ? 0.0 : var;
        mean = (float)mu;
        invstd = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
        if (save_mean) { save_mean[c] = mean; save_invstd[c] = invstd; }
    } else {
        if (threadIdx.x >= FIN_CH || c >= C) return;
        mean = running_mean[c];
        invstd = 1.0f / sqrtf(running_var[c] + eps);
    }
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
}

// y = x*scale[c] + shift[c] (+ residual) (relu)
__global__ __launch_bounds__(NT) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, const float* __restrict__ res,
                                                      float* __restrict__ y, long total4, int C, int relu)
{
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    if (i >= total4) return;
    const int c = (int)((i * 4) % C);
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    v = v * *reinterpret_cast<const f32x4*>(scale + c) + *reinterpret_cast<const f32x4*>(shift + c);
    if (res) v += reinterpret_cast<const f32x4*>(res)[i];
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    reinterpret_cast<f32x4*>(y)[i] = v;
}

__global__ __launch_bounds__(FIN_NT) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, long M, int C,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                 float* __restrict__ c1, float* __restrict__ c2, int accumulate)
{
    __shared__ double red[2 * FIN_NT];
    const int c = blockIdx.x * FIN_CH + threadIdx.x % FIN_CH;
    double s, sx;
    finalize_column_sums(partial, nblk, C, red, s, sx);
    if (threadIdx.x >= FIN_CH || c >= C) return;
    if (accumulate) { dgamma[c] += (float)sx; dbeta[c] += (float)s; }
    else { dgamma[c] = (float)sx; dbeta[c] = (float)s; }
    c1[c] = (float)(s / (double)M);
    c2[c] = (float)(sx / (double)M);
}

// g = dy * (y>0 if relu);  dx = gamma*invstd*(g - c1 - xhat*c2);  dres (+)= g
__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ c1, const float* __restrict__ c2,
    float* __restrict__ dx, float* __restrict__ dres, long total4, int C, int relu, int dres_accumulate)
{
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    if (i >= total4) return;
    const int c = (int)((i * 4) % C);
    f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
    if (relu) {
        const f32x4 yv = reinterpret_cast<const f32x4*>(y)[i];
        g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
        g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
    }
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
    const f32x4 xh = (reinterpret_cast<const f32x4*>(x)[i] - mu) * is;
    const f32x4 k = *reinterpret_cast<const f32x4*>(gamma + c) * is;
    reinterpret_cast<f32x4*>(dx)[i] = k * (g - *reinterpret_cast<const f32x4*>(c1 + c) - xh * *reinterpret_cast<const f32x4*>(c2 + c));
    if (dres) {
        if (dres_accumulate) g += reinterpret_cast<const f32x4*>(dres)[i];
        reinterpret_cast<f32x4*>(dres)[i] = g;
    }
}

// ---- cross-rank (SyncBatchNorm) pieces: everything stays on the device, the caller only all-reduces two small buffers ----
// sums [2C+1] fp64 = (sum x [C], sum x^2 [C], element count) of THIS rank: additive across ranks
__global__ __launch_bounds__(FIN_NT) void bn_local_sums_kernel(const float* __restrict__ partial, int nblk, long M, int C,
                                                               double* __restrict__ sums)
{
    __shared__ double red[2 * FIN_NT];
    const int c = blockIdx.x * FIN_CH + threadIdx.x % FIN_CH;
    double s, ss;
    finalize_column_sums(partial, nblk, C, red, s, ss);
    if (threadIdx.x >= FIN_CH || c >= C) return;
    sums[c] = s;
    sums[C + c] = ss;
    if (c == 0) sums[2 * C] = (double)M;
}

// (all-reduced) sums -> statistics of the union batch, running statistics (unbiased variance with the GLOBAL count, like
// torch.nn.SyncBatchNorm), scale / shift.  The count is read from device memory: no host round trip.
__global__ __launch_bounds__(NT) void bn_finalize_sums_kernel(
    const double* __restrict__ sums, int C, float eps, float momentum, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
    float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ scale, float* __restrict__ shift)
{
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= C) return;
    const double n = sums[2 * C];
    const double mu = sums[c] / n;
    double var = sums[C + c] / n - mu * mu;
    var = var < 0.0 ? 0.0 : var;
    const float mean = (float)mu, invstd = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
}

// c1 = sum_g / count, c2 = sum_g_xhat / count from the (all-reduced) f32 sums [2][C] = (sum g*xhat, sum g) and the device count
__global__ __launch_bounds__(NT) void bn_bwd_means_kernel(const float* __restrict__ sums, const double* __restrict__ count, int C,
                                                          float* __restrict__ c1, float* __restrict__ c2)
{
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= C) return;
    const double n = *count;
    c1[c] = (float)((double)sums[C + c] / n);
    c2[c] = (float)((double)sums[c] / n);
}

// ---- max pool 3x3 / stride 2 / pad 1, NHWC ---------------------------------------------------------------
__global__ __launch_bounds__(NT) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         unsigned char* __restrict__ arg, int N, int Hi, int Wi, int C,
                                                         int Ho, int Wo)
{
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    const int c4 = C >> 2;
    const long total = (long)N * Ho * Wo * c4;
    if (i >= total) return;
    const int c = (int)(i % c4) * 4;
    long p = i / c4;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho);
    const int n = (int)(p / Ho);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    bool first = true;
    for (int r = 0; r < 3; ++r) {
        const int iy = oy * 2 - 1 + r;
        if (iy < 0 || iy >= Hi) continue;
        for (int q = 0; q < 3; ++q) {
            const int ix = ox * 2 - 1 + q;
            if (ix < 0 || ix >= Wi) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)(n * Hi + iy) * Wi + ix) * C + c);
            const int id = r * 3 + q;
            // first maximum in scan order wins (ATen: val > maxval || isnan(val))
            if (first || v.x > best.x || v.x != v.x) { best.x = v.x; bi[0] = id; }
            if (first || v.y > best.y || v.y != v.y) { best.y = v.y; bi[1] = id; }
            if (first || v.z > best.z || v.z != v.z) { best.z = v.z; bi[2] = id; }
            if (first || v.w > best.w || v.w != v.w) { best.w = v.w; bi[3] = id; }
            first = false;
        }
    }
    reinterpret_cast<f32x4*>(y)[i] = best;
    if (arg) *reinterpret_cast<uchar4*>(arg + i * 4) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
}

__global__ __launch_bounds__(NT) void maxpool_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg,
                                                         float* __restrict__ dx, int N, int Hi, int Wi, int C, int Ho, int Wo)
{
    // gather form: every input pixel looks at the <=4 windows that contain it - deterministic, no atomics
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    const int c4 = C >> 2;
    const long total = (long)N * Hi * Wi * c4;
    if (i >= total) return;
    const int c = (int)(i % c4) * 4;
    long p = i / c4;
    const int ix = (int)(p % Wi); p /= Wi;
    const int iy = (int)(p % Hi);
    const int n = (int)(p / Hi);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    for (int oy = max(0, iy / 2); oy <= min(Ho - 1, (iy + 1) / 2); ++oy) {
        const int r = iy - (oy * 2 - 1);
        if (r < 0 || r > 2) continue;
        for (int ox = max(0, ix / 2); ox <= min(Wo - 1, (ix + 1) / 2); ++ox) {
            const int q = ix - (ox * 2 - 1);
            if (q < 0 || q > 2) continue;
            const size_t o = ((size_t)(n * Ho + oy) * Wo + ox) * C + c;
            const uchar4 a = *reinterpret_cast<const uchar4*>(arg + o);
            const f32x4 d = *reinterpret_cast<const f32x4*>(dy + o);
            const int id = r * 3 + q;
            if (a.x == id) g.x += d.x;
            if (a.y == id) g.y += d.y;
            if (a.z == id) g.z += d.z;
            if (a.w == id) g.w += d.w;
        }
    }
    reinterpret_cast<f32x4*>(dx)[i] = g;
}

// ---- FPN top-down path: fine += nearest_upsample(coarse) and its adjoint -----------------------------------
__global__ __launch_bounds__(NT) void upsample_add_kernel(float* __restrict__ fine, const float* __restrict__ coarse,
                                                          int N, int H, int W, int h, int w, int C)
{
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    const int c4 = C >> 2;
    const long total = (long)N * H * W * c4;
    if (i >= total) return;
    const int c = (int)(i % c4) * 4;
    long p = i / c4;
    const int x = (int)(p % W); p /= W;
    const int y = (int)(p % H);
    const int n = (int)(p / H);
    const int sy = min((int)(((long)y * h) / H), h - 1), sx = min((int)(((long)x * w) / W), w - 1);
    reinterpret_cast<f32x4*>(fine)[i] += *reinterpret_cast<const f32x4*>(coarse + ((size_t)(n * h + sy) * w + sx) * C + c);
}

__global__ __launch_bounds__(NT) void upsample_add_bwd_kernel(const float* __restrict__ dfine, float* __restrict__ dcoarse,
                                                              int N, int H, int W, int h, int w, int C)
{
    // dcoarse += sum of the fine-grid gradients that were fed from this coarse pixel
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    const int c4 = C >> 2;
    const long total = (long)N * h * w * c4;
    if (i >= total) return;
    const int c = (int)(i % c4) * 4;
    long p = i / c4;
    const int sx = (int)(p % w); p /= w;
    const int sy = (int)(p % h);
    const int n = (int)(p / h);
    const int y0 = (int)(((long)sy * H + h - 1) / h), y1 = (int)(((long)(sy + 1) * H + h - 1) / h);
    const int x0 = (int)(((long)sx * W + w - 1) / w), x1 = (int)(((long)(sx + 1) * W + w - 1) / w);
    f32x4 g = reinterpret_cast<const f32x4*>(dcoarse)[i];
    for (int y = y0; y < min(y1, H); ++y)
        for (int x = x0; x < min(x1, W); ++x)
            g += *reinterpret_cast<const f32x4*>(dfine + ((size_t)(n * H + y) * W + x) * C + c);
    reinterpret_cast<f32x4*>(dcoarse)[i] = g;
}

// column sums of a [M][C] matrix (bias gradients): grid (column blocks of 64, row slabs); each slab writes a
// partial row, a second pass adds the slabs in fp64 - deterministic, no atomics.
__global__ __launch_bounds__(NT) void colsum_partial_kernel(const float* __restrict__ a, float* __restrict__ part, long M, int C,
                                                            long rows_per_slab)
{
    __shared__ float red[NT];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.y * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
    float s = 0.f;
    if (col < C)
        for (long r = r0 + rl; r < r1; r += NT / 64) s += a[(size_t)r * C + col];
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && col < C)
        part[(size_t)blockIdx.y * C + col] = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}

__global__ void colsum_finalize_kernel(const float* __restrict__ part, float* __restrict__ out, int slabs, int C, int accumulate)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float old = accumulate ? out[c] : 0.f;         // cold read first
    double t = 0.0;
#pragma unroll 8
    for (int s = 0; s < slabs; ++s) t += (double)part[(size_t)s * C + c];
    out[c] = old + (float)t;
}

bool channels_ok(int C) { return C >= 4 && (C & 3) == 0 && (C >> 2) <= NT && NT % (C >> 2) == 0; }

long stat_blocks(long M, long* rows_per_block)
{
    long rpb = max((long)32, ceil_div64(M, 1024));
    *rows_per_block = rpb;
    return ceil_div64(M, rpb);
}

}  // namespace

// Number of floats the caller must provide as `partial` for M rows of C channels.
PHNET_API uint64_t phnet_channel_partials_size(int64_t M, int32_t C)
{
    long rpb;
    return (uint64_t)(stat_blocks(M, &rpb) * 2 * C);
}

// BatchNorm forward, step 1+2: statistics -> scale/shift (and running-stat update, saved mean/invstd).
// training=0: scale/shift from the running statistics (x, partial, save_* unused).
PHNET_API int phnet_bn_fwd_stats(const float* x, int64_t M, int32_t C, float eps, float momentum,
                                 const float* gamma, const float* beta, float* running_mean, float* running_var,
                                 float* save_mean, float* save_invstd, float* scale, float* shift,
                                 float* partial, int32_t training, void* stream)
{
    if (M < 1 || !channels_ok(C) || !gamma || !beta || !scale || !shift) return PHNET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int nblk = 0;
    if (training) {
        if (!x || !partial) return PHNET_ERR_ARG;
        long rpb;
        nblk = (int)stat_blocks(M, &rpb);
        hipLaunchKernelGGL(channel_partials_kernel<0>, dim3(nblk), dim3(NT), 2 * NT * sizeof(f32x4), st,
                           x, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                           partial, (long)M, C, rpb, 0);
    } else if (!running_mean || !running_var) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_NT), 0, st, partial, nblk, (long)M, C, eps, momentum,
                       gamma, beta, running_mean, running_var, save_mean, save_invstd, scale, shift, training);
    return phnet_launch_status();
}

// Batch statistics from per-block (sum, sum of squares) partials that the convolution's epilogue (or its split-K reduce)
// wrote: phnet_conv2d_fwd_fused(..., stats = partial).  Same finalize kernel, no pass over x.
PHNET_API int phnet_bn_finalize_partials(const float* partial, int64_t nblk, int64_t M, int32_t C, float eps, float momentum,
                                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                                         float* save_mean, float* save_invstd, float* scale, float* shift, void* stream)
{
    if (M < 1 || nblk < 1 || nblk > 0x7fffffff || !channels_ok(C) || !partial || !gamma || !beta || !scale || !shift) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_NT), 0, (hipStream_t)stream, partial, (int)nblk, (long)M, C, eps,
                       momentum, gamma, beta, running_mean, running_var, save_mean, save_invstd, scale, shift, 1);
    return phnet_launch_status();
}

// y = x*scale + shift (+residual) (relu);  y may alias x.
PHNET_API int phnet_bn_apply(const float* x, const float* scale, const float* shift, const float* residual, float* y,
                             int64_t M, int32_t C, int32_t relu, void* stream)
{
    if (M < 1 || !channels_ok(C) || !x || !scale || !shift || !y) return PHNET_ERR_ARG;
    const long total4 = M * C / 4;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)ceil_div64(total4, NT)), dim3(NT), 0, (hipStream_t)stream,
                       x, scale, shift, residual, y, total4, C, relu);
    return phnet_launch_status();
}

// BatchNorm backward (training statistics).  dy: gradient w.r.t. the block output (after the optional ReLU, whose
// mask is y > 0).  dx: gradient w.r.t. the conv output x.  dres (optional): gradient of the residual input,
// overwritten or accumulated.  c1/c2: [C] scratch.  dgamma/dbeta overwritten or accumulated.
PHNET_API int phnet_bn_bwd(const float* dy, const float* x, const float* y, const float* save_mean,
                           const float* save_invstd, const float* gamma, float* dx, float* dres,
                           float* dgamma, float* dbeta, float* partial, float* c1, float* c2,
                           int64_t M, int32_t C, int32_t relu, int32_t dres_accumulate, int32_t param_accumulate, void* stream)
{
    if (M < 1 || !channels_ok(C) || !dy || !x || !save_mean || !save_invstd || !gamma || !dx || !dgamma || !dbeta ||
        !partial || !c1 || !c2 || (relu && !y))
        return PHNET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    long rpb;
    const int nblk = (int)stat_blocks(M, &rpb);
    hipLaunchKernelGGL(channel_partials_kernel<1>, dim3(nblk), dim3(NT), 2 * NT * sizeof(f32x4), st,
                       x, dy, y, save_mean, save_invstd, partial, (long)M, C, rpb, relu);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_NT), 0, st, partial, nblk, (long)M, C,
                       dgamma, dbeta, c1, c2, param_accumulate);
    const long total4 = M * C / 4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)ceil_div64(total4, NT)), dim3(NT), 0, st,
                       dy, x, y, save_mean, save_invstd, gamma, c1, c2, dx, dres, total4, C, relu, dres_accumulate);
    return phnet_launch_status();
}

// Split form of phnet_bn_bwd for cross-rank (SyncBatchNorm) training: step 1 = local per-channel sums
// sums[0][C] = sum g*xhat, sums[1][C] = sum g  (g = dy masked by y>0 when relu; xhat uses the GLOBAL mean/invstd);
// the caller all-reduces `sums` (and the element count) and calls step 2 with c1 = sum_g/M_global, c2 = sum_gx/M_global.
PHNET_API int phnet_bn_bwd_reduce(const float* dy, const float* x, const float* y, const float* mean, const float* invstd,
                                  float* sums, float* partial, float* c1_scratch, float* c2_scratch,
                                  int64_t M, int32_t C, int32_t relu, void* stream)
{
    if (M < 1 || !channels_ok(C) || !dy || !x || !mean || !invstd || !sums || !partial || !c1_scratch || !c2_scratch || (relu && !y))
        return PHNET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    long rpb;
    const int nblk = (int)stat_blocks(M, &rpb);
    hipLaunchKernelGGL(channel_partials_kernel<1>, dim3(nblk), dim3(NT), 2 * NT * sizeof(f32x4), st,
                       x, dy, y, mean, invstd, partial, (long)M, C, rpb, relu);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_NT), 0, st, partial, nblk, (long)M, C,
                       sums, sums + C, c1_scratch, c2_scratch, 0);
    return phnet_launch_status();
}

PHNET_API int phnet_bn_bwd_apply(const float* dy, const float* x, const float* y, const float* mean, const float* invstd,
                                 const float* gamma, const float* c1, const float* c2, float* dx, float* dres,
                                 int64_t M, int32_t C, int32_t relu, int32_t dres_accumulate, void* stream)
{
    if (M < 1 || !channels_ok(C) || !dy || !x || !mean || !invstd || !gamma || !c1 || !c2 || !dx || (relu && !y))
        return PHNET_ERR_ARG;
    const long total4 = M * C / 4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)ceil_div64(total4, NT)), dim3(NT), 0, (hipStream_t)stream,
                       dy, x, y, mean, invstd, gamma, c1, c2, dx, dres, total4, C, relu, dres_accumulate);
    return phnet_launch_status();
}

// ---- SyncBatchNorm forward, device-resident (trainOL.py:141 nn.SyncBatchNorm.convert_sync_batchnorm) -------------------------
// step 1: sums [2C+1] fp64 = (sum x, sum x^2, count M) of this rank -> the caller all-reduces (SUM) the buffer across ranks
PHNET_API int phnet_bn_local_sums(const float* x, int64_t M, int32_t C, float* partial, double* sums, void* stream)
{
    if (M < 1 || !channels_ok(C) || !x || !partial || !sums) return PHNET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    long rpb;
    const int nblk = (int)stat_blocks(M, &rpb);
    hipLaunchKernelGGL(channel_partials_kernel<0>, dim3(nblk), dim3(NT), 2 * NT * sizeof(f32x4), st,
                       x, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       partial, (long)M, C, rpb, 0);
    hipLaunchKernelGGL(bn_local_sums_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_NT), 0, st, (const float*)partial, nblk, (long)M, C, sums);
    return phnet_launch_status();
}

// step 1 from per-block (sum, sum of squares) partials that the convolution's epilogue (or its split-K reduce) wrote
// (phnet_conv2d_fwd_fused(..., stats = partial)): no pass over x.
PHNET_API int phnet_bn_partials_to_sums(const float* partial, int64_t nblk, int64_t M, int32_t C, double* sums, void* stream)
{
    if (M < 1 || nblk < 1 || nblk > 0x7fffffff || !channels_ok(C) || !partial || !sums) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(bn_local_sums_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_NT), 0, (hipStream_t)stream, partial, (int)nblk, (long)M, C,
                       sums);
    return phnet_launch_status();
}

// step 2: reduced sums -> save_mean / save_invstd / scale / shift (+ running statistics).  Then phnet_bn_apply as usual.
PHNET_API int phnet_bn_finalize_sums(const double* sums, int32_t C, float eps, float momentum, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                                     float* scale, float* shift, void* stream)
{
    if (!channels_ok(C) || !sums || !gamma || !beta || !save_mean || !save_invstd || !scale || !shift ||
        (!running_mean != !running_var))
        return PHNET_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize_sums_kernel, dim3((C + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, sums, C, eps, momentum,
                       gamma, beta, running_mean, running_var, save_mean, save_invstd, scale, shift);
    return phnet_launch_status();
}

// SyncBatchNorm backward, step 2 with the element count in device memory: sums [2][C] f32 = all-reduced (sum g*xhat, sum g)
// from phnet_bn_bwd_reduce, count = the all-reduced sums[2C] of the forward.  c1 / c2: [C] scratch.
PHNET_API int phnet_bn_bwd_apply_sums(const float* dy, const float* x, const float* y, const float* mean, const float* invstd,
                                      const float* gamma, const float* sums, const double* count, float* c1, float* c2,
                                      float* dx, float* dres, int64_t M, int32_t C, int32_t relu, int32_t dres_accumulate,
                                      void* stream)
{
    if (M < 1 || !channels_ok(C) || !dy || !x || !mean || !invstd || !gamma || !sums || !count || !c1 || !c2 || !dx || (relu && !y))
        return PHNET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_bwd_means_kernel, dim3((C + NT - 1) / NT), dim3(NT), 0, st, sums, count, C, c1, c2);
    const long total4 = M * C / 4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)ceil_div64(total4, NT)), dim3(NT), 0, st,
                       dy, x, y, mean, invstd, gamma, (const float*)c1, (const float*)c2, dx, dres, total4, C, relu, dres_accumulate);
    return phnet_launch_status();
}

PHNET_API int phnet_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* argmax, int32_t N, int32_t Hi, int32_t Wi, int32_t C,
                                     void* stream)
{
    if (N < 1 || Hi < 1 || Wi < 1 || C < 4 || (C & 3) || !x || !y) return PHNET_ERR_ARG;
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    const long total = (long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)ceil_div64(total, NT)), dim3(NT), 0, (hipStream_t)stream,
                       x, y, argmax, N, Hi, Wi, C, Ho, Wo);
    return phnet_launch_status();
}

PHNET_API int phnet_maxpool3x3s2_bwd(const float* dy, const uint8_t* argmax, float* dx, int32_t N, int32_t Hi, int32_t Wi,
                                     int32_t C, void* stream)
{
    if (N < 1 || Hi < 1 || Wi < 1 || C < 4 || (C & 3) || !dy || !argmax || !dx) return PHNET_ERR_ARG;
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    const long total = (long)N * Hi * Wi * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)ceil_div64(total, NT)), dim3(NT), 0, (hipStream_t)stream,
                       dy, argmax, dx, N, Hi, Wi, C, Ho, Wo);
    return phnet_launch_status();
}

// fine[N][H][W][C] += coarse[N][h][w][C] nearest-upsampled to (H,W).
PHNET_API int phnet_upsample_add(float* fine, const float* coarse, int32_t N, int32_t H, int32_t W, int32_t h, int32_t w,
                                 int32_t C, void* stream)
{
    if (N < 1 || H < 1 || W < 1 || h < 1 || w < 1 || C < 4 || (C & 3) || !fine || !coarse) return PHNET_ERR_ARG;
    const long total = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(upsample_add_kernel, dim3((unsigned)ceil_div64(total, NT)), dim3(NT), 0, (hipStream_t)stream,
                       fine, coarse, N, H, W, h, w, C);
    return phnet_launch_status();
}

// dcoarse += adjoint of the nearest upsample applied to dfine.
PHNET_API int phnet_upsample_add_bwd(const float* dfine, float* dcoarse, int32_t N, int32_t H, int32_t W, int32_t h, int32_t w,
                                     int32_t C, void* stream)
{
    if (N < 1 || H < 1 || W < 1 || h < 1 || w < 1 || C < 4 || (C & 3) || !dfine || !dcoarse) return PHNET_ERR_ARG;
    const long total = (long)N * h * w * (C / 4);
    hipLaunchKernelGGL(upsample_add_bwd_kernel, dim3((unsigned)ceil_div64(total, NT)), dim3(NT), 0, (hipStream_t)stream,
                       dfine, dcoarse, N, H, W, h, w, C);
    return phnet_launch_status();
}

// out[C] (+)= column sums of a[M][C]  (bias gradients).  workspace: >= phnet_colsum_workspace(M, C) bytes.
PHNET_API uint64_t phnet_colsum_workspace(int64_t M, int32_t C)
{
    const long slabs = max((long)1, min((long)256, M / 64));
    return (uint64_t)(slabs * C * sizeof(float));
}

PHNET_API int phnet_colsum(const float* a, float* out, int64_t M, int32_t C, int32_t accumulate,
                           void* workspace, uint64_t ws_bytes, void* stream)
{
    if (M < 0 || C < 1 || !a || !out || !workspace) return PHNET_ERR_ARG;
    const long slabs = max((long)1, min((long)256, (long)M / 64));
    if ((uint64_t)(slabs * C * sizeof(float)) > ws_bytes) return PHNET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long rps = ceil_div64(max((long)M, (long)1), slabs);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((C + 63) / 64, (unsigned)slabs), dim3(NT), 0, st,
                       a, (float*)workspace, (long)M, C, rps);
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st,
                       (const float*)workspace, out, (int)slabs, C, accumulate);
    return phnet_launch_status();
}
