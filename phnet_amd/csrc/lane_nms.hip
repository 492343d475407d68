// Lane NMS for gfx950: one workgroup per frame does score ranking, the pairwise
// similarity bit-matrix and the greedy sweep in a single launch (the reference needs a
// device sort + 2 launches, the second one <<<1,1>>>: libs/ops/csrc/nms.cpp:51,
// nms_kernel.cu:50-96, 99-143).  Results are bit-identical to that arithmetic
// (nms_kernel.cu:26-48): float multiply / double add / truncation for the row extents,
// float accumulation of |dx| in ascending row order, no fused multiply-adds.
#include "common.h"

namespace {

constexpr int NMS_THREADS = 256;

struct Extent { int start, end; };

__device__ __forceinline__ Extent lane_extent(const float* a, int n_strips) {
#pragma clang fp contract(off)
    Extent e;
    float scaled = a[2] * (float)n_strips;
    e.start = (int)((double)scaled + 0.5);
    float f = (float)e.start + a[4];
    f = f - 1.0f;
    double d = (double)f + 0.5;
    d -= (double)((a[4] - 1.0f) < 0.0f ? 1 : 0);
    e.end = (int)d;
    return e;
}

__device__ __forceinline__ bool lanes_similar(const float* a, const float* b, Extent ea, Extent eb,
                                              int n_offsets, float thr) {
#pragma clang fp contract(off)
    const int start = max(ea.start, eb.start);
    const int end = min(min(ea.end, eb.end), n_offsets - 1);
    if (end < start) return false;
    float dist = 0.0f;
    // the reference's loop counter is an unsigned char that starts at 5 + start
    for (unsigned char i = (unsigned char)(5 + start); (int)i <= 5 + end; ++i) {
        const float x = a[i], y = b[i];
        dist += (x < y) ? (y - x) : (x - y);
    }
    return dist < thr * (float)(end - start + 1);
}

// Score ranking, pairwise similarity bits and greedy sweep on K rows `R` (LDS or global), shared by the plain NMS
// entry point and the fused decode.  Leaves order[] (rank -> row), keepers[] (ranks kept, in order) and mask[] filled
// and returns the number of keepers (block-uniform).
__device__ __forceinline__ int nms_core(const float* R, const float* scores, int K, int prop, int n_offsets, float thr,
                                        int64_t top_k, int* order, Extent* ext, int* keepers, unsigned long long* mask,
                                        int* s_kept)
{
    const int tid = threadIdx.x;
    const int words = (K + 63) >> 6;
    // ---- rank by descending score (ties: lower index first).  The order must be TOTAL or two rows share a rank and a slot
    // of order[] stays unwritten: a NaN score ranks above every number, as in the reference's scores.sort(0, descending=True)
    // (csrc/nms.cpp:51: ATen orders NaN as the largest value), NaNs among themselves by index ----
    for (int i = tid; i < K; i += NMS_THREADS) {
        const float si = scores[i];
        const bool ni = si != si;
        int rank = 0;
        for (int j = 0; j < K; ++j) {
            const float sj = scores[j];
            const bool nj = sj != sj;
            const bool above = (nj && !ni) || (sj > si);
            const bool tied = (nj && ni) || (sj == si);
            rank += above || (tied && j < i);
        }
        order[rank] = i;
    }
    __syncthreads();
    for (int i = tid; i < K; i += NMS_THREADS) ext[i] = lane_extent(R + (size_t)order[i] * prop, n_offsets - 1);
    __syncthreads();
    // ---- similarity bits for i<j in score order; one (row i, 64-column word) per work item ----
    for (int item = tid; item < K * words; item += NMS_THREADS) {
        const int i = item / words, w = item - i * words;
        unsigned long long bits = 0;
        const int j0 = max(w << 6, i + 1), j1 = min((w << 6) + 64, K);
        if (j0 < j1) {
            const float* a = R + (size_t)order[i] * prop;
            const Extent ea = ext[i];
            for (int j = j0; j < j1; ++j)
                if (lanes_similar(a, R + (size_t)order[j] * prop, ea, ext[j], n_offsets, thr)) bits |= 1ull << (j & 63);
        }
        mask[item] = bits;
    }
    __syncthreads();
    // ---- greedy sweep by wave 0; lane w owns suppression word w (K <= 4096 by the LDS bound) ----
    if (tid < 64) {
        unsigned long long remv = 0;
        int kept = 0;
        for (int i = 0; i < K; ++i) {
            const unsigned long long word = __shfl(remv, i >> 6, 64);
            if (!((word >> (i & 63)) & 1ull)) {
                if (tid == 0) keepers[kept] = i;
                if (tid < words) remv |= mask[(size_t)i * words + tid];
                ++kept;
                if ((int64_t)kept == top_k) break;
            }
        }
        if (tid == 0) *s_kept = kept;
    }
    __syncthreads();
    return *s_kept;
}

// Dynamic LDS carve-up (bytes): order[K] i32 | ext[K] 2xi32 | keepers[K] i32 | mask[K*words] u64 | rows (optional)
__global__ __launch_bounds__(NMS_THREADS) void lane_nms_kernel(
    const float* __restrict__ rows_all, const float* __restrict__ scores_all, const int32_t* __restrict__ counts,
    int64_t k_max, int n_offsets, float thr, int64_t top_k,
    int64_t* __restrict__ keep_all, int64_t* __restrict__ num_all, int64_t* __restrict__ parent_all, int stage_rows)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int frame = blockIdx.x;
    const int K = counts ? min((int)counts[frame], (int)k_max) : (int)k_max;
    const int prop = 5 + n_offsets;
    const int words = (K + 63) >> 6;
    const float* rows = rows_all + (size_t)frame * k_max * prop;
    const float* scores = scores_all + (size_t)frame * k_max;
    int64_t* keep = keep_all + (size_t)frame * k_max;
    int64_t* parent = parent_all + (size_t)frame * k_max;
    const int tid = threadIdx.x;

    const int kpad = (int)((k_max + 1) & ~(int64_t)1);
    const int wmax = (int)((k_max + 63) >> 6);
    int* order = (int*)smem;
    Extent* ext = (Extent*)(order + kpad);
    int* keepers = (int*)(ext + kpad);
    unsigned long long* mask = (unsigned long long*)(keepers + kpad);
    float* lrows = (float*)(mask + (size_t)kpad * wmax);
    __shared__ int s_kept;

    if (stage_rows)
        for (int i = tid; i < K * prop; i += NMS_THREADS) lrows[i] = rows[i];
    __syncthreads();
    const float* R = stage_rows ? lrows : rows;
    const int kept = nms_core(R, scores, K, prop, n_offsets, thr, top_k, order, ext, keepers, mask, &s_kept);

    // ---- outputs exactly as nms_collect leaves them ----
    for (int i = tid; i < K; i += NMS_THREADS) keep[i] = i < kept ? (int64_t)order[keepers[i]] : 0;
    for (int j = tid; j < K; j += NMS_THREADS) {
        int64_t p = 0;
        for (int r = 0; r < kept; ++r) {               // later keepers overwrite earlier ones
            const int i = keepers[r];
            if (i == j || (i < j && ((mask[(size_t)i * words + (j >> 6)] >> (j & 63)) & 1ull))) p = r + 1;
        }
        parent[order[j]] = p;
    }
    if (tid == 0) num_all[frame] = top_k < (int64_t)kept ? top_k : (int64_t)kept;
}

// Fused eval decode of one frame (Router4OL.py:437-471): softmax score + confidence threshold + candidate compaction +
// NMS row construction (theta column dropped, pixel / strip units) + lane NMS + gather of the kept rows, no host sync.
// One workgroup per frame; N <= 256 anchors.
__global__ __launch_bounds__(NMS_THREADS) void lane_decode_kernel(
    const float* __restrict__ lines_all, int N, int n_offsets, float conf_thr, float nms_thr, int64_t top_k, float img_w,
    unsigned char* __restrict__ keep_mask_all, int64_t* __restrict__ num_all, int64_t* __restrict__ keep_c_all,
    int64_t* __restrict__ anchors_all, int64_t* __restrict__ anchors_sorted_all, float* __restrict__ kept_rows_all)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int frame = blockIdx.x, tid = threadIdx.x;
    const int W = 6 + n_offsets, prop = 5 + n_offsets, TK = (int)top_k;
    const float* lines = lines_all + (size_t)frame * N * W;
    const int kpad = (N + 1) & ~1, wmax = (N + 63) >> 6;
    int* order = (int*)smem;
    Extent* ext = (Extent*)(order + kpad);
    int* keepers = (int*)(ext + kpad);
    unsigned long long* mask = (unsigned long long*)(keepers + kpad);
    float* lrows = (float*)(mask + (size_t)kpad * wmax);
    float* cscore = lrows + (size_t)N * prop;
    int* canchor = (int*)(cscore + N);
    __shared__ int s_kept, s_count;
    __shared__ unsigned char flag[NMS_THREADS];
    __shared__ float score_all[NMS_THREADS];

    // ---- score, threshold (softmax(cls)[1] >= conf_thr) ----
    bool ok = false;
    if (tid < N) {
        const float z0 = lines[(size_t)tid * W], z1 = lines[(size_t)tid * W + 1];
        const float zm = fmaxf(z0, z1);
        const float e0 = expf(z0 - zm), e1 = expf(z1 - zm);
        const float sc = e1 / (e0 + e1);
        score_all[tid] = sc;
        ok = sc >= conf_thr;
        keep_mask_all[(size_t)frame * N + tid] = ok ? 1 : 0;
    }
    flag[tid] = ok ? 1 : 0;
    __syncthreads();
    // ---- compaction in anchor order + NMS rows ----
    if (ok) {
        int r = 0;
        for (int j = 0; j < tid; ++j) r += flag[j];
        canchor[r] = tid;
        cscore[r] = score_all[tid];
        const float* src = lines + (size_t)tid * W;
        float* dst = lrows + (size_t)r * prop;
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
        dst[3] = src[3] * (img_w - 1.0f);
        dst[4] = src[5] * (float)(n_offsets - 1);
        for (int k = 0; k < n_offsets; ++k) dst[5 + k] = src[6 + k] * (img_w - 1.0f);
    }
    if (tid == 0) { int c = 0; for (int j = 0; j < N; ++j) c += flag[j]; s_count = c; }
    __syncthreads();
    const int K = s_count;
    int kept = 0;
    if (K > 0) kept = nms_core(lrows, cscore, K, prop, n_offsets, nms_thr, top_k, order, ext, keepers, mask, &s_kept);
    __syncthreads();
    // ---- outputs: kept candidates in NMS order, their anchors (also ascending), the gathered rows ----
    if (tid == 0) {
        num_all[frame] = kept;
        int sorted[64];
        for (int r = 0; r < TK; ++r) {
            const int c = r < kept ? order[keepers[r]] : -1;
            keep_c_all[(size_t)frame * TK + r] = c;
            anchors_all[(size_t)frame * TK + r] = c >= 0 ? canchor[c] : -1;
            if (r < 64) sorted[r] = c >= 0 ? canchor[c] : 0x7fffffff;
        }
        const int n = TK < 64 ? TK : 64;
        for (int a = 0; a < n; ++a)
            for (int b = a + 1; b < n; ++b)
                if (sorted[b] < sorted[a]) { const int t = sorted[a]; sorted[a] = sorted[b]; sorted[b] = t; }
        for (int r = 0; r < TK; ++r)
            anchors_sorted_all[(size_t)frame * TK + r] = (r < n && sorted[r] != 0x7fffffff) ? sorted[r] : -1;
    }
    for (int i = tid; i < TK * W; i += NMS_THREADS) {
        const int r = i / W, c = i - r * W;
        float v = 0.f;
        if (r < kept) {
            const int a = canchor[order[keepers[r]]];
            v = lines[(size_t)a * W + c];
            if (c == 5) v = rintf(v * (float)(n_offsets - 1));       // torch.round: half to even
        }
        kept_rows_all[((size_t)frame * TK + r) * W + c] = v;
    }
}

size_t nms_lds_bytes(int64_t k_max, int n_offsets, bool stage_rows) {
    const size_t kpad = (size_t)((k_max + 1) & ~(int64_t)1);
    const size_t wmax = (size_t)((k_max + 63) >> 6);
    size_t b = kpad * 4 + kpad * 8 + kpad * 4 + kpad * wmax * 8;
    if (stage_rows) b += (size_t)k_max * (5 + n_offsets) * 4;
    return b;
}

}  // namespace

// Replaces nms_forward / nms_cuda_forward (libs/ops/csrc/nms.cpp:44-57, nms_kernel.cu:147-192).
// rows [frames][k_max][5+n_offsets] f32, scores [frames][k_max] f32, counts [frames] i32 or NULL (= k_max each).
// keep [frames][k_max] i64, num_to_keep [frames] i64, parent [frames][k_max] i64: caller-allocated device memory.
PHNET_API int phnet_lane_nms(const float* rows, const float* scores, const int32_t* counts, int64_t frames,
                             int64_t k_max, int32_t n_offsets, float thresh, int64_t top_k,
                             int64_t* keep, int64_t* num_to_keep, int64_t* parent, void* stream)
{
    if (frames < 0 || k_max < 0 || n_offsets < 1 || n_offsets > 250) return PHNET_ERR_ARG;
    if (frames == 0) return PHNET_OK;
    if (!num_to_keep) return PHNET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (k_max == 0) {
        return hipMemsetAsync(num_to_keep, 0, sizeof(int64_t) * frames, st) == hipSuccess ? PHNET_OK : PHNET_ERR_LAUNCH;
    }
    if (!rows || !scores || !keep || !parent) return PHNET_ERR_ARG;
    if (k_max >= 64 * 1000) return PHNET_ERR_ARG;      // the reference's MAX_COL_BLOCKS bound (nms_kernel.cu:10,157)
    bool stage = true;
    size_t lds = nms_lds_bytes(k_max, n_offsets, true);
    if (lds > 150 * 1024) { stage = false; lds = nms_lds_bytes(k_max, n_offsets, false); }
    if (lds > 150 * 1024) return PHNET_ERR_ARG;        // K too large for the single-workgroup design (K <~ 1000)
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)lane_nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PHNET_ERR_LAUNCH;
    hipLaunchKernelGGL(lane_nms_kernel, dim3((unsigned)frames), dim3(NMS_THREADS), lds, st,
                       rows, scores, counts, k_max, (int)n_offsets, thresh, top_k, keep, num_to_keep, parent, stage ? 1 : 0);
    return phnet_launch_status();
}

// Fused eval decode: replaces DetNetV2.get_lanes up to the kept rows (Router4OL.py:441-471: softmax, confidence mask,
// boolean-mask compaction (a host sync in the reference), NMS row build, libs.ops.nms, keep[:num] slicing (another
// sync), gather, length rounding).  lines [frames][N][6+S] (N <= 256, top_k <= 64).  Outputs (caller-allocated):
// keep_mask u8 [frames][N]; num i64 [frames]; keep_c i64 [frames][top_k] indices into the compacted candidate list in
// NMS order (-1 padded; what the reference calls `keep`); anchors / anchors_sorted i64 [frames][top_k] the same lanes
// as anchor indices (NMS order / ascending, -1 padded); kept_rows [frames][top_k][6+S] (column 5 rounded to strips,
// zero rows beyond num).
PHNET_API int phnet_lane_decode(const float* lines, int64_t frames, int32_t N, int32_t n_offsets, float conf_thresh,
                                float nms_thresh, int64_t top_k, float img_w, uint8_t* keep_mask, int64_t* num,
                                int64_t* keep_c, int64_t* anchors, int64_t* anchors_sorted, float* kept_rows, void* stream)
{
    if (frames < 0 || N < 1 || N > NMS_THREADS || n_offsets < 1 || n_offsets > 250 || top_k < 1 || top_k > 64) return PHNET_ERR_ARG;
    if (frames == 0) return PHNET_OK;
    if (!lines || !keep_mask || !num || !keep_c || !anchors || !anchors_sorted || !kept_rows) return PHNET_ERR_ARG;
    const size_t lds = nms_lds_bytes(N, n_offsets, true) + (size_t)N * 8;
    if (lds > 150 * 1024) return PHNET_ERR_ARG;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)lane_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PHNET_ERR_LAUNCH;
    hipLaunchKernelGGL(lane_decode_kernel, dim3((unsigned)frames), dim3(NMS_THREADS), lds, (hipStream_t)stream,
                       lines, N, (int)n_offsets, conf_thresh, nms_thresh, top_k, img_w, keep_mask, num, keep_c, anchors,
                       anchors_sorted, kept_rows);
    return phnet_launch_status();
}
