// Input pre-processing of a clip on the GPU: decoded uint8 RGB frames -> the float tensor the
// model consumes, one launch per clip.  Replaces, for the image half of the reference's data pipeline,
//   libs/dataset/openlane/datasetOL.py:40-52  (crop the top `crop_size` rows, optional left-right flip)
//   libs/dataset/openlane/transforms.py:150-156 (iaa.Resize -> cv2 INTER_CUBIC on the uint8 image)
//   libs/dataset/openlane/datasetOL.py:63-75, 11-17 (ToTensor /255, Normalize(mean, std), stacking of the frames)
// which the reference runs per frame on CPU data-loader workers; at the inference rates of this implementation (1 500
// frames/s) that is the next bottleneck.
//
// Arithmetic: OpenCV's 8-bit bicubic resize (half-pixel centres, a = -0.75, 11-bit fixed-point taps stored per tap without renormalisation,
// replicated borders, rounding 22-bit shift, saturation) followed by x/255 and (x - mean)/std in f32.  The tap tables are
// built on the host side of the C-ABI once per geometry and passed in (idx [n][4] int32, coef [n][4] int16 per axis).
// HBM-bound: 16 source bytes x 3 channels per output pixel (L2-served re-reads), 12-16 bytes written.
#include "common.h"

namespace {

constexpr int NT = 256;

// one thread per output pixel: 4x4 taps x 3 channels
template <bool NHWC4>
__global__ __launch_bounds__(NT) void preprocess_kernel(
    const uint8_t* __restrict__ src, float* __restrict__ dst, uint8_t* __restrict__ dst_u8,
    const int32_t* __restrict__ xi, const int16_t* __restrict__ xc, const int32_t* __restrict__ yi, const int16_t* __restrict__ yc,
    int T, int H0, int W0, int crop_top, int out_h, int out_w, int flip,
    float m0, float m1, float m2, float s0, float s1, float s2)
{
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    const long per = (long)out_h * out_w;
    if (i >= (long)T * per) return;
    const int t = (int)(i / per);
    const int rem = (int)(i - (long)t * per);
    const int oy = rem / out_w, ox = rem - oy * out_w;
    const uint8_t* frame = src + (size_t)t * H0 * W0 * 3 + (size_t)crop_top * W0 * 3;
    const int Wc = W0;
    int acc[3] = {0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint8_t* row = frame + (size_t)yi[oy * 4 + r] * Wc * 3;
        int h[3] = {0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int x = xi[ox * 4 + q];
            if (flip) x = Wc - 1 - x;
            const uint8_t* px = row + (size_t)x * 3;
            const int c = xc[ox * 4 + q];
            h[0] += px[0] * c; h[1] += px[1] * c; h[2] += px[2] * c;
        }
        const int cy = yc[oy * 4 + r];
        acc[0] += h[0] * cy; acc[1] += h[1] * cy; acc[2] += h[2] * cy;
    }
    float v[3];
    uint8_t u[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int q = (acc[c] + (1 << 21)) >> 22;
        q = q < 0 ? 0 : (q > 255 ? 255 : q);
        u[c] = (uint8_t)q;
        v[c] = (float)q / 255.0f;
    }
    v[0] = (v[0] - m0) / s0; v[1] = (v[1] - m1) / s1; v[2] = (v[2] - m2) / s2;
    if (dst_u8) { uint8_t* o = dst_u8 + (size_t)i * 3; o[0] = u[0]; o[1] = u[1]; o[2] = u[2]; }
    if (NHWC4) {
        reinterpret_cast<float4*>(dst)[i] = make_float4(v[0], v[1], v[2], 0.f);
    } else {
        float* o = dst + (size_t)t * 3 * per + rem;
        o[0] = v[0]; o[per] = v[1]; o[2 * per] = v[2];
    }
}

}  // namespace

// frames u8 [T][H0][W0][3] RGB (device) -> out f32: layout 0 = NCHW [T][3][out_h][out_w] (the reference's tensor), layout 1 =
// NHWC padded to 4 channels [T][out_h][out_w][4] (what the stem convolution of this implementation stages: skips the
// NCHW -> NHWC4 pass).  The top crop_top rows are dropped, flip != 0 mirrors left-right before resampling.
// xi/xc: [out_w][4] source columns (int32, clamped) and 11-bit taps (int16, each saturate_cast<short>(c * 2048): sum 2047..2049); yi/yc the same for rows of the CROPPED
// image.  out_u8 (optional) [T][out_h][out_w][3]: the resized 8-bit image (what the reference's `img_rgb` holds, x255).
PHNET_API int phnet_preprocess_u8(const uint8_t* frames, float* out, uint8_t* out_u8,
                                  const int32_t* xi, const int16_t* xc, const int32_t* yi, const int16_t* yc,
                                  int32_t T, int32_t H0, int32_t W0, int32_t crop_top, int32_t out_h, int32_t out_w, int32_t flip,
                                  int32_t layout, const float* mean3_host, const float* std3_host, void* stream)
{
    if (T < 0 || H0 < 1 || W0 < 1 || crop_top < 0 || crop_top >= H0 || out_h < 1 || out_w < 1 || (layout != 0 && layout != 1))
        return PHNET_ERR_ARG;
    if (T == 0) return PHNET_OK;
    if (!frames || !out || !xi || !xc || !yi || !yc || !mean3_host || !std3_host) return PHNET_ERR_ARG;
    if (std3_host[0] == 0.f || std3_host[1] == 0.f || std3_host[2] == 0.f) return PHNET_ERR_ARG;
    const long total = (long)T * out_h * out_w;
    const dim3 grid((unsigned)ceil_div64(total, NT));
    if (layout == 1)
        hipLaunchKernelGGL(preprocess_kernel<true>, grid, dim3(NT), 0, (hipStream_t)stream, frames, out, out_u8, xi, xc, yi, yc, T, H0, W0,
                           crop_top, out_h, out_w, flip, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
    else
        hipLaunchKernelGGL(preprocess_kernel<false>, grid, dim3(NT), 0, (hipStream_t)stream, frames, out, out_u8, xi, xc, yi, yc, T, H0, W0,
                           crop_top, out_h, out_w, flip, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
    return phnet_launch_status();
}
