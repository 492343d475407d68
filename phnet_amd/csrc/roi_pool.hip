// Lane-anchor ROI pooling (bilinear grid sampling along each anchor) for gfx950.
// Replaces DetNetV2.pool_prior_features = F.grid_sample(map, grid, align_corners=True) + permutes
// (libs/models/Router4OL.py:132-150, 269-272) and its autograd backward.
//
// Layout: feature maps are NHWC [B][h][w][64] (the trunk/FPN kernels produce NHWC), so one
// wavefront = the 64 channels of one sample point: every corner fetch is one coalesced 256-B
// line instead of ATen's 64 strided 4-B reads in NCHW.  Output [B][N][P][64] is already the
// [anchor][point][channel] layout the dynamic head's bmm wants (Router4OL.py:279).
// HBM-bound: algorithmic bytes per sample = 4 corner lines read + 1 line written = 1280 B.
#include "common.h"

namespace {

constexpr int ROI_C = 64;
constexpr int ROI_THREADS = 256;

struct Corners {
    int x0, y0;                 // north-west corner (floor)
    float nw, ne, sw, se;       // bilinear weights
    float ix, iy;
};

__device__ __forceinline__ Corners corners_of(float xn, float yn, int h, int w) {
    // same operation order as ATen's grid_sampler_unnormalize(align_corners=True) on g = v*2-1
    Corners c;
    const float gx = xn * 2.0f - 1.0f, gy = yn * 2.0f - 1.0f;
    c.ix = ((gx + 1.0f) / 2.0f) * (float)(w - 1);
    c.iy = ((gy + 1.0f) / 2.0f) * (float)(h - 1);
    const float fx = floorf(c.ix), fy = floorf(c.iy);
    // clamp the float before the int conversion: anchors may leave the map by a lot (tan blow-up)
    c.x0 = (int)fminf(fmaxf(fx, -2.0f), (float)w + 1.0f);
    c.y0 = (int)fminf(fmaxf(fy, -2.0f), (float)h + 1.0f);
    const float xe = fx + 1.0f, ye = fy + 1.0f;
    c.nw = (xe - c.ix) * (ye - c.iy);
    c.ne = (c.ix - fx) * (ye - c.iy);
    c.sw = (xe - c.ix) * (c.iy - fy);
    c.se = (c.ix - fx) * (c.iy - fy);
    return c;
}

__global__ __launch_bounds__(ROI_THREADS) void roi_pool_fwd_kernel(
    const float* __restrict__ fmap, const float* __restrict__ xs, const float* __restrict__ ys,
    float* __restrict__ out, float* __restrict__ out_cp, int B, int N, int P, int h, int w)
{
    const int lane = threadIdx.x & 63;
    const long sample = (long)blockIdx.x * (ROI_THREADS / 64) + (threadIdx.x >> 6);
    const long total = (long)B * N * P;
    if (sample >= total) return;
    const int k = (int)(sample % P);
    const long bn = sample / P;
    const int b = (int)(bn / N);
    // x_k pairs with y_k after the flip of the anchor's x list (Router4OL.py:269)
    const float xn = xs[bn * P + (P - 1 - k)];
    const float yn = ys[k];
    float acc = 0.0f;
    if (isfinite(xn)) {
        const Corners c = corners_of(xn, yn, h, w);
        const float* base = fmap + (size_t)b * h * w * ROI_C + lane;
        const bool x0in = c.x0 >= 0 && c.x0 < w, x1in = c.x0 + 1 >= 0 && c.x0 + 1 < w;
        const bool y0in = c.y0 >= 0 && c.y0 < h, y1in = c.y0 + 1 >= 0 && c.y0 + 1 < h;
        float vnw = 0.f, vne = 0.f, vsw = 0.f, vse = 0.f;
        if (y0in && x0in) vnw = base[((size_t)c.y0 * w + c.x0) * ROI_C];
        if (y0in && x1in) vne = base[((size_t)c.y0 * w + c.x0 + 1) * ROI_C];
        if (y1in && x0in) vsw = base[((size_t)(c.y0 + 1) * w + c.x0) * ROI_C];
        if (y1in && x1in) vse = base[((size_t)(c.y0 + 1) * w + c.x0 + 1) * ROI_C];
        acc = vnw * c.nw;
        acc += vne * c.ne;
        acc += vsw * c.sw;
        acc += vse * c.se;
    } else {
        acc = xn * 0.0f;        // NaN in -> NaN out, like ATen
    }
    out[sample * ROI_C + lane] = acc;
    if (out_cp) out_cp[(bn * ROI_C + lane) * P + k] = acc;       // [B][N][C][P] copy for the routing gate
}

// Any channel count C <= 64 (the Router4OLV2 family pools 64 / 32 / 16 channels per pyramid level, Router4OLV2.py:143-161):
// one thread per (sample, channel); C consecutive lanes read one contiguous C*4-byte corner line.  Forward only.
__global__ __launch_bounds__(ROI_THREADS) void roi_pool_fwd_any_kernel(
    const float* __restrict__ fmap, const float* __restrict__ xs, const float* __restrict__ ys,
    float* __restrict__ out, float* __restrict__ out_cp, int B, int N, int P, int h, int w, int C)
{
    const long id = (long)blockIdx.x * ROI_THREADS + threadIdx.x;
    if (id >= (long)B * N * P * C) return;
    const int ch = (int)(id % C);
    const long sample = id / C;
    const int k = (int)(sample % P);
    const long bn = sample / P;
    const int b = (int)(bn / N);
    const float xn = xs[bn * P + (P - 1 - k)];
    const float yn = ys[k];
    float acc = 0.0f;
    if (isfinite(xn)) {
        const Corners c = corners_of(xn, yn, h, w);
        const float* base = fmap + (size_t)b * h * w * C + ch;
        const bool x0in = c.x0 >= 0 && c.x0 < w, x1in = c.x0 + 1 >= 0 && c.x0 + 1 < w;
        const bool y0in = c.y0 >= 0 && c.y0 < h, y1in = c.y0 + 1 >= 0 && c.y0 + 1 < h;
        float vnw = 0.f, vne = 0.f, vsw = 0.f, vse = 0.f;
        if (y0in && x0in) vnw = base[((size_t)c.y0 * w + c.x0) * C];
        if (y0in && x1in) vne = base[((size_t)c.y0 * w + c.x0 + 1) * C];
        if (y1in && x0in) vsw = base[((size_t)(c.y0 + 1) * w + c.x0) * C];
        if (y1in && x1in) vse = base[((size_t)(c.y0 + 1) * w + c.x0 + 1) * C];
        acc = vnw * c.nw;
        acc += vne * c.ne;
        acc += vsw * c.sw;
        acc += vse * c.se;
    } else {
        acc = xn * 0.0f;
    }
    out[sample * C + ch] = acc;
    if (out_cp) out_cp[(bn * C + ch) * P + k] = acc;
}

__global__ __launch_bounds__(ROI_THREADS) void roi_pool_bwd_kernel(
    const float* __restrict__ dout, const float* __restrict__ fmap, const float* __restrict__ xs,
    const float* __restrict__ ys, float* __restrict__ dmap, float* __restrict__ dxs,
    int B, int N, int P, int h, int w)
{
    const int lane = threadIdx.x & 63;
    const long sample = (long)blockIdx.x * (ROI_THREADS / 64) + (threadIdx.x >> 6);
    const long total = (long)B * N * P;
    if (sample >= total) return;
    const int k = (int)(sample % P);
    const long bn = sample / P;
    const int b = (int)(bn / N);
    const float xn = xs[bn * P + (P - 1 - k)];
    const float yn = ys[k];
    const float g = dout[sample * ROI_C + lane];
    float gix = 0.0f;
    if (isfinite(xn)) {
        const Corners c = corners_of(xn, yn, h, w);
        const size_t plane = (size_t)b * h * w * ROI_C + lane;
        const bool x0in = c.x0 >= 0 && c.x0 < w, x1in = c.x0 + 1 >= 0 && c.x0 + 1 < w;
        const bool y0in = c.y0 >= 0 && c.y0 < h, y1in = c.y0 + 1 >= 0 && c.y0 + 1 < h;
        const float fy = floorf(c.iy);
        const float ye = fy + 1.0f;
        if (y0in && x0in) {
            const size_t o = plane + ((size_t)c.y0 * w + c.x0) * ROI_C;
            if (dmap) atomicAdd(dmap + o, g * c.nw);
            if (dxs) gix -= fmap[o] * (ye - c.iy) * g;
        }
        if (y0in && x1in) {
            const size_t o = plane + ((size_t)c.y0 * w + c.x0 + 1) * ROI_C;
            if (dmap) atomicAdd(dmap + o, g * c.ne);
            if (dxs) gix += fmap[o] * (ye - c.iy) * g;
        }
        if (y1in && x0in) {
            const size_t o = plane + ((size_t)(c.y0 + 1) * w + c.x0) * ROI_C;
            if (dmap) atomicAdd(dmap + o, g * c.sw);
            if (dxs) gix -= fmap[o] * (c.iy - fy) * g;
        }
        if (y1in && x1in) {
            const size_t o = plane + ((size_t)(c.y0 + 1) * w + c.x0 + 1) * ROI_C;
            if (dmap) atomicAdd(dmap + o, g * c.se);
            if (dxs) gix += fmap[o] * (c.iy - fy) * g;
        }
    }
    if (dxs) {
        gix = wave_sum(gix);
        // d(ix)/d(g) = (w-1)/2 and g = 2*x-1
        if (lane == 0) dxs[bn * P + (P - 1 - k)] = gix * ((float)(w - 1) / 2.0f) * 2.0f;
    }
}

}  // namespace

// fmap [B][h][w][C] f32 NHWC; xs [B][N][P] normalised anchor x per sample row (un-flipped
// priors_on_featmap); ys [P] = prior_feat_ys; out [B][N][P][C].  C = 64: one wavefront per sample (the V1 head, has a
// backward); other C <= 64: one thread per (sample, channel), forward only.
// out_cp (optional): the same samples as [B][N][64][P] (the layout the routing gate consumes).
PHNET_API int phnet_roi_pool_fwd(const float* fmap, const float* xs, const float* ys, float* out, float* out_cp,
                                 int32_t B, int32_t N, int32_t P, int32_t h, int32_t w, int32_t C, void* stream)
{
    if (C < 1 || C > ROI_C || B < 0 || N < 0 || P < 0 || h < 1 || w < 1) return PHNET_ERR_ARG;
    const long total = (long)B * N * P;
    if (total == 0) return PHNET_OK;
    if (!fmap || !xs || !ys || !out) return PHNET_ERR_ARG;
    if (C != ROI_C) {                                   // per-level widths of the Router4OLV2 family (forward only)
        hipLaunchKernelGGL(roi_pool_fwd_any_kernel, dim3((unsigned)ceil_div64(total * C, ROI_THREADS)), dim3(ROI_THREADS), 0,
                           (hipStream_t)stream, fmap, xs, ys, out, out_cp, B, N, P, h, w, C);
        return phnet_launch_status();
    }
    const unsigned blocks = (unsigned)ceil_div64(total, ROI_THREADS / 64);
    hipLaunchKernelGGL(roi_pool_fwd_kernel, dim3(blocks), dim3(ROI_THREADS), 0, (hipStream_t)stream,
                       fmap, xs, ys, out, out_cp, B, N, P, h, w);
    return phnet_launch_status();
}

// dmap [B][h][w][64] is ACCUMULATED into (float atomics; caller zero-fills or passes an existing
// gradient buffer; may be NULL); dxs [B][N][P] is overwritten (may be NULL when the anchors are detached).
PHNET_API int phnet_roi_pool_bwd(const float* dout, const float* fmap, const float* xs, const float* ys,
                                 float* dmap, float* dxs,
                                 int32_t B, int32_t N, int32_t P, int32_t h, int32_t w, int32_t C, void* stream)
{
    if (C != ROI_C || B < 0 || N < 0 || P < 0 || h < 1 || w < 1) return PHNET_ERR_ARG;
    const long total = (long)B * N * P;
    if (total == 0 || (!dmap && !dxs)) return PHNET_OK;
    if (!dout || !fmap || !xs || !ys) return PHNET_ERR_ARG;
    const unsigned blocks = (unsigned)ceil_div64(total, ROI_THREADS / 64);
    hipLaunchKernelGGL(roi_pool_bwd_kernel, dim3(blocks), dim3(ROI_THREADS), 0, (hipStream_t)stream,
                       dout, fmap, xs, ys, dmap, dxs, B, N, P, h, w);
    return phnet_launch_status();
}
