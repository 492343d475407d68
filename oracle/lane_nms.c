/* CPU oracle for the lane NMS  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar restatement of the arithmetic of the reference CUDA extension
 *   libs/ops/csrc/nms_kernel.cu:26-48   (pair test "devIoU")
 *   libs/ops/csrc/nms_kernel.cu:50-96   (which pairs are tested: i<j in score order)
 *   libs/ops/csrc/nms_kernel.cu:99-143  (greedy sweep "nms_collect")
 * with the compile-time N_OFFSETS (:12) turned into a run-time argument.  The score
 * sort of libs/ops/csrc/nms.cpp:51 is done by the caller (oracle/lane_nms.py).
 *
 * The reference extension is CUDA-only and cannot be built in this image, and the
 * reference ships no test vectors for it: the parity of this file is pinned by
 * source review only ("parity unpinned" in the sense of DESIGN.md) plus the
 * hand-derived known-answer cases in tests/test_oracle_nms.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Row layout (5 + n_offsets floats): cls0, cls1, start_y, start_x_px, length_strips, x_0..x_{S-1}. */

static int lane_extent_start(const float *a, int n_strips)
{
    /* (int)(a[2] * N_STRIPS - 0 + 0.5): float multiply, double add, truncation  (:28) */
    float scaled = a[2] * (float)n_strips;
    return (int)((double)scaled + 0.5);
}

static int lane_extent_end(const float *a, int start)
{
    /* start_a + a[4] - 1 + 0.5 - ((a[4] - 1) < 0): float adds, then double, truncation  (:31) */
    float f = (float)start + a[4];
    f = f - 1.0f;
    double d = (double)f + 0.5;
    d -= (double)((a[4] - 1.0f) < 0.0f ? 1 : 0);
    return (int)d;
}

int phnet_oracle_lane_similar(const float *a, const float *b, int n_offsets, float threshold)
{
    const int n_strips = n_offsets - 1;
    const int start_a = lane_extent_start(a, n_strips);
    const int start_b = lane_extent_start(b, n_strips);
    const int start = start_a > start_b ? start_a : start_b;
    const int end_a = lane_extent_end(a, start_a);
    const int end_b = lane_extent_end(b, start_b);
    int end = end_a < end_b ? end_a : end_b;
    if (end > n_offsets - 1) end = n_offsets - 1;
    if (end < start) return 0;
    float dist = 0.0f;
    /* the reference loop counter is an unsigned char starting at 5 + start  (:38) */
    for (unsigned char i = (unsigned char)(5 + start); (int)i <= 5 + end; ++i) {
        if (a[i] < b[i]) dist += b[i] - a[i];
        else             dist += a[i] - b[i];
    }
    return dist < threshold * (float)(end - start + 1);
}

/* rows [K][5+n_offsets]; order[K] = row indices by descending score.
 * keep[K], parent[K], *num_to_keep as nms_collect writes them. */
int phnet_oracle_lane_nms(const float *rows, const int64_t *order, int64_t K, int n_offsets,
                          float threshold, int64_t top_k,
                          int64_t *keep, int64_t *num_to_keep, int64_t *parent)
{
    const int prop = 5 + n_offsets;
    unsigned char *removed = (unsigned char *)calloc((size_t)(K > 0 ? K : 1), 1);
    if (!removed) return -1;
    int64_t kept = 0;
    for (int64_t i = 0; i < K; ++i) parent[i] = 0;
    for (int64_t i = 0; i < K; ++i) {
        if (removed[i]) continue;
        const float *cur = rows + order[i] * prop;
        keep[kept] = order[i];
        for (int64_t j = i + 1; j < K; ++j) {
            if (phnet_oracle_lane_similar(cur, rows + order[j] * prop, n_offsets, threshold)) {
                removed[j] = 1;
                parent[order[j]] = kept + 1;
            }
        }
        parent[order[i]] = kept + 1;
        ++kept;
        if (kept == top_k) break;
    }
    for (int64_t i = kept; i < K; ++i) keep[i] = 0;
    *num_to_keep = top_k < kept ? top_k : kept;
    free(removed);
    return 0;
}
