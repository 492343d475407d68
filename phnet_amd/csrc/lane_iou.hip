// Lane-IoU kernels of the CULane-style evaluator (reference: evaluation/culane/src/lane_compare.cpp:11-57 - draw both lanes
// as thick poly-lines, count pixels, IoU = |A & B| / |A | B|).  A lane is a bit mask [height][ceil(width / 32)] words in HBM:
// `lane_raster_kernel` ORs one thick segment per workgroup into its lane's mask, `lane_mask_stats_kernel` counts the bits of
// every lane and of every requested pair's intersection.  Pixel rule (oracle/culane_cpu.py raster_lane, exact integers): the
// pixel centre lies within lane_width / 2 of the segment between the integer end points - the ideal shape of cv::line's
// quadrilateral + end circles (parity against OpenCV's scan conversion unpinned: OpenCV is not in the reference tree).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void lane_raster_kernel(const int* __restrict__ segs, long n_segs, unsigned* __restrict__ masks,
                                                          int n_lanes, int height, int width, int lane_width)
{
    const long s = blockIdx.x;
    if (s >= n_segs) return;
    const int x0 = segs[s * 5 + 0], y0 = segs[s * 5 + 1], x1 = segs[s * 5 + 2], y1 = segs[s * 5 + 3], lane = segs[s * 5 + 4];
    if ((unsigned)lane >= (unsigned)n_lanes) return;
    const int r = (lane_width + 1) / 2;
    const int xa = max(0, min(x0, x1) - r), xb = min(width - 1, max(x0, x1) + r);
    const int ya = max(0, min(y0, y1) - r), yb = min(height - 1, max(y0, y1) + r);
    if (xa > xb || ya > yb) return;
    const int wpr = (width + 31) >> 5;                               // words per mask row
    const int wa = xa >> 5, wb = xb >> 5, nw = wb - wa + 1;
    const long dx = x1 - x0, dy = y1 - y0, L2 = dx * dx + dy * dy, w2 = (long)lane_width * lane_width;
    unsigned* m = masks + (size_t)lane * height * wpr;
    for (int item = threadIdx.x; item < (yb - ya + 1) * nw; item += blockDim.x) {
        const int y = ya + item / nw, w = wa + item % nw;
        const long qy = y - y0, ey = y - y1;
        unsigned bits = 0;
        for (int b = 0; b < 32; ++b) {
            const int x = w * 32 + b;
            if (x < xa || x > xb) continue;
            const long qx = x - x0, ex = x - x1;
            const long dot = qx * dx + qy * dy;
            bool in;
            if (dot <= 0) in = 4 * (qx * qx + qy * qy) <= w2;
            else if (dot >= L2) in = 4 * (ex * ex + ey * ey) <= w2;
            else { const long c = qx * dy - qy * dx; in = 4 * c * c <= w2 * L2; }
            bits |= in ? 1u << b : 0u;
        }
        if (bits) atomicOr(m + (size_t)y * wpr + w, bits);
    }
}

// blockIdx.y < n_lanes: area of that lane; else pair blockIdx.y - n_lanes: bits of the intersection
__global__ __launch_bounds__(256) void lane_mask_stats_kernel(const unsigned* __restrict__ masks, int n_lanes, long words,
                                                              const int* __restrict__ pairs, int n_pairs,
                                                              unsigned long long* __restrict__ area, unsigned long long* __restrict__ inter)
{
    const int job = blockIdx.y;
    const unsigned *a, *b;
    unsigned long long* dst;
    if (job < n_lanes) { a = b = masks + (size_t)job * words; dst = area + job; }
    else {
        const int p = job - n_lanes, i = pairs[2 * p], j = pairs[2 * p + 1];
        if ((unsigned)i >= (unsigned)n_lanes || (unsigned)j >= (unsigned)n_lanes) return;
        a = masks + (size_t)i * words; b = masks + (size_t)j * words; dst = inter + p;
    }
    unsigned cnt = 0;
    for (long w = (long)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (long)gridDim.x * blockDim.x) cnt += __popc(a[w] & b[w]);
    float f = (float)cnt;                                            // < 2^24 per wave: exact
    f = wave_sum(f);
    __shared__ unsigned part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = (unsigned)f;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(dst, (unsigned long long)t);
    }
}

}  // namespace

// segs [n_segs][5] int32 (x0, y0, x1, y1, lane); masks [n_lanes][height][ceil(width / 32)] uint32, zeroed by the caller (bits are ORed in)
PHNET_API int phnet_lane_raster(const int32_t* segs, int64_t n_segs, uint32_t* masks, int32_t n_lanes, int32_t height, int32_t width,
                                int32_t lane_width, void* stream)
{
    if (n_segs < 0 || n_lanes < 0 || height < 1 || width < 1 || height > 4096 || width > 4096 || lane_width < 1 || lane_width > 256)
        return PHNET_ERR_ARG;                       // with end points inside +-2^13 the 64-bit distance tests cannot overflow
    if (n_segs == 0 || n_lanes == 0) return PHNET_OK;
    if (!segs || !masks || n_segs > 0x7fffffffL) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(lane_raster_kernel, dim3((unsigned)n_segs), dim3(256), 0, (hipStream_t)stream,
                       segs, (long)n_segs, masks, n_lanes, height, width, lane_width);
    return phnet_launch_status();
}

// area [n_lanes] and inter [n_pairs] int64, zeroed by the caller: set bits of every lane mask / of masks[pairs[p][0]] & masks[pairs[p][1]]
PHNET_API int phnet_lane_mask_stats(const uint32_t* masks, int32_t n_lanes, int32_t height, int32_t width, const int32_t* pairs,
                                    int32_t n_pairs, int64_t* area, int64_t* inter, void* stream)
{
    if (n_lanes < 0 || n_pairs < 0 || height < 1 || width < 1) return PHNET_ERR_ARG;
    if (n_lanes + n_pairs == 0) return PHNET_OK;
    if (!masks || !area || (n_pairs && (!pairs || !inter)) || n_lanes + n_pairs > 65535) return PHNET_ERR_ARG;
    const long words = (long)height * ((width + 31) >> 5);
    const unsigned bx = (unsigned)min((long)64, ceil_div64(words, 256 * 8));
    hipLaunchKernelGGL(lane_mask_stats_kernel, dim3(bx, (unsigned)(n_lanes + n_pairs)), dim3(256), 0, (hipStream_t)stream,
                       masks, n_lanes, words, pairs, n_pairs, (unsigned long long*)area, (unsigned long long*)inter);
    return phnet_launch_status();
}
