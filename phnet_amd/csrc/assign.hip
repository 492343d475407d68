// On-device label assignment for the lane criterion (gfx950): cost matrix + exact minimum-cost matching of
// <= 4 ground-truth lanes to 240 anchors in ONE single-workgroup launch, no host round trip.
//
// Replaces  libs/utils/dynamic_assign.py:128-190 `assign` (cost terms :44-80, :5-36) together with its
// `C.cpu(); scipy.optimize.linear_sum_assignment(C)` (:186-188) - six device->host syncs per frame in the reference
// (SURVEY.md 8(f) rank 2).  The matching is exact: an optimal assignment of m columns always exists inside the
// m cheapest rows of every column, so the m^m <= 256 candidate combinations are enumerated (one per thread) and the
// minimum total cost wins (ties: lowest combination index; scipy's tie order is unspecified too).
#include "assign_device.h"

namespace {

using namespace phassign;

__global__ __launch_bounds__(4 * NT) void lane_assign_kernel(
    const float* __restrict__ pred, const float* __restrict__ tgt, int N, int L, int S, float img_w, float img_h,
    int64_t* __restrict__ rows_by_col, int64_t* __restrict__ rows_sorted, int32_t* __restrict__ n_valid_out,
    float* __restrict__ cost_out)
{
    extern __shared__ float cost[];                  // [N][MAXL]
    lane_assign_block<false, 4>(pred, tgt, N, L, S, img_w, img_h, rows_by_col, rows_sorted, n_valid_out, cost_out, cost);
}

// Assignment + the memory tokens it selects (Router4OL.py:563-584) in one launch: tokens [L+1][E] = the features of the matched
// anchors in ascending anchor order (unused slots zero, valid = 0), then the mean of all other anchors; E <= 256.
__global__ __launch_bounds__(4 * NT) void lane_assign_tokens_kernel(
    const float* __restrict__ pred, const float* __restrict__ tgt, int N, int L, int S, float img_w, float img_h,
    int64_t* __restrict__ rows_by_col, int64_t* __restrict__ rows_sorted, const float* __restrict__ feat, int E,
    float* __restrict__ tokens, unsigned char* __restrict__ valid)
{
    extern __shared__ float cost[];                  // [N][MAXL], then [groups][E] column partial sums
    float* part = cost + N * MAXL;
    // column sums of the features do not depend on the assignment: their loads are in flight while it runs
    const int e = threadIdx.x % E, grp = threadIdx.x / E, groups = (4 * NT) / E;
    float s = 0.f;
    if (grp < groups)
        for (int n = grp; n < N; n += groups) s += feat[(size_t)n * E + e];
    lane_assign_block<false, 4>(pred, tgt, N, L, S, img_w, img_h, rows_by_col, rows_sorted, nullptr, nullptr, cost);
    if (grp < groups) part[grp * E + e] = s;
    __syncthreads();                                 // also orders rows_sorted (written by thread 0) before the reads below
    if (grp != 0) return;
    float total = 0.f;
    for (int g2 = 0; g2 < groups; ++g2) total += part[g2 * E + e];
    float possum = 0.f;
    int cnt = 0;
    for (int l = 0; l < L; ++l) {
        const long long r = rows_sorted[l];
        const bool ok = r >= 0 && r < N;
        const float v = ok ? feat[(size_t)r * E + e] : 0.f;
        tokens[(size_t)l * E + e] = v;
        possum += v;
        cnt += ok;
        if (e == 0) valid[l] = ok;
    }
    tokens[(size_t)L * E + e] = (total - possum) / (float)(N - cnt);
    if (e == 0) valid[L] = 1;
}

__global__ __launch_bounds__(4 * NT) void lane_assign_many_kernel(
    const float* __restrict__ pred, const float* __restrict__ tgt, int N, int L, int S, float img_w, float img_h,
    int64_t* __restrict__ rows, int64_t* __restrict__ cols, int32_t* __restrict__ n_pairs)
{
    extern __shared__ float cost[];                  // [2][N][MAXL]: cost, clamped IoU
    lane_assign_block<true, 4>(pred, tgt, N, L, S, img_w, img_h, nullptr, nullptr, nullptr, nullptr, cost, 0.5f, rows, cols, n_pairs);
}

}  // namespace
// pred [N][6+S] predictions of one head/stage, tgt [L][6+S] label rows (col 1 == 1 marks a valid lane; L <= 4,
// N <= 256, S <= 250).  rows_by_col [L] int64: matched anchor of label j or -1; rows_sorted [L] int64: the matched
// anchors ascending, padded with -1; n_valid (optional) int32 number of matches; cost (optional) [N][L] the matrix
// handed to the solver (+inf in invalid columns).
PHNET_API int phnet_lane_assign(const float* pred, const float* tgt, int32_t N, int32_t L, int32_t S,
                                float img_w, float img_h, int64_t* rows_by_col, int64_t* rows_sorted,
                                int32_t* n_valid, float* cost, void* stream)
{
    if (N < 1 || N > NT || L < 0 || L > MAXL || S < 1 || S > 250) return PHNET_ERR_ARG;
    if (L == 0) return PHNET_OK;
    if (!pred || !tgt || !rows_by_col || !rows_sorted) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(lane_assign_kernel, dim3(1), dim3(4 * NT), (size_t)N * MAXL * sizeof(float), (hipStream_t)stream,
                       pred, tgt, N, L, S, img_w, img_h, rows_by_col, rows_sorted, n_valid, cost);
    return phnet_launch_status();
}

// One-to-many assignment, replaces libs/utils/dynamic_assign.py:292-357 `assignOne2Many` (its loop of
// `C.cpu(); scipy.optimize.linear_sum_assignment(C)` rounds) in one single-workgroup launch.  pred / tgt as phnet_lane_assign.
// rows / cols [16] int64: the (anchor, label row) pairs in the reference's order (round by round, each round ascending by
// anchor), padded with -1; n_pairs (optional) their number.
PHNET_API int phnet_lane_assign_one2many(const float* pred, const float* tgt, int32_t N, int32_t L, int32_t S,
                                         float img_w, float img_h, int64_t* rows, int64_t* cols, int32_t* n_pairs, void* stream)
{
    if (N < 1 || N > NT || L < 1 || L > MAXL || S < 1 || S > 250) return PHNET_ERR_ARG;
    if (!pred || !tgt || !rows || !cols) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(lane_assign_many_kernel, dim3(1), dim3(4 * NT), (size_t)2 * N * MAXL * sizeof(float), (hipStream_t)stream,
                       pred, tgt, N, L, S, img_w, img_h, rows, cols, n_pairs);
    return phnet_launch_status();
}

// phnet_lane_assign followed by phnet_memory_tokens on its result, as one launch: feat [N][E] (E <= 256, 1024 % E == 0),
// tokens [L+1][E], valid u8 [L+1].  L >= 1.
PHNET_API int phnet_lane_assign_tokens(const float* pred, const float* tgt, int32_t N, int32_t L, int32_t S, float img_w, float img_h,
                                       int64_t* rows_by_col, int64_t* rows_sorted, const float* feat, int32_t E,
                                       float* tokens, uint8_t* valid, void* stream)
{
    if (N < 1 || N > NT || L < 1 || L > MAXL || S < 1 || S > 250 || E < 1 || E > 256 || (4 * NT) % E) return PHNET_ERR_ARG;
    if (!pred || !tgt || !rows_by_col || !rows_sorted || !feat || !tokens || !valid) return PHNET_ERR_ARG;
    const size_t lds = ((size_t)N * MAXL + (size_t)4 * NT) * sizeof(float);
    hipLaunchKernelGGL(lane_assign_tokens_kernel, dim3(1), dim3(4 * NT), lds, (hipStream_t)stream,
                       pred, tgt, N, L, S, img_w, img_h, rows_by_col, rows_sorted, feat, E, tokens, valid);
    return phnet_launch_status();
}
