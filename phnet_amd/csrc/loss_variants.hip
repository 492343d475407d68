// The reference's other two training criteria as fused kernels (gfx950), two launches per frame each - like phnet_frame_loss
// (loss.hip) for the V3 criterion:
//   variant 1  libs/utils/loss4OL.py:88-232 (trainOLV2.py / trainOLV3.py): one-to-one assignment per (branch, stage); focal
//              classification vector; per matched PAIR smooth-L1 (mean of the four start / angle / length values) and
//              1 - line IoU (fixed 15 px radius, dynamic_assign.py:5-36), each divided by the number of pairs, summed over the
//              stages BY POSITION in the row-sorted pair list and placed on the anchors matched at the LAST stage
//              (CalculateInstLoss :168-175); branch balance (lower median shift, gate-weighted sum) on that per-anchor vector.
//   variant 2  libs/utils/loss4OLV2.py:12-186: one-to-many assignment (dynamic_assign.py:292-357, up to 16 pairs); focal vector
//              balanced between the branches as in V3; regression / IoU are means over all pairs of a (branch, stage).
// Value AND every input gradient (six prediction tensors, three gate tensors) come out of the two launches; the ATen
// formulation they replace issues ~450 launches per frame.  Latency-bound: 6 workgroups + 1.
#include "assign_device.h"

namespace {

using namespace phassign;

constexpr int MAXP = MAXL * MAXL;              // pairs per (branch, stage): <= 4 (variant 1) or <= 16 (variant 2)

struct VarParams {
    const float* pred[6];      // [N][6+S] : branch A stages 0..2, branch B stages 0..2
    const float* gate[3];      // [N]
    float* dpred[6];           // [N][6+S] gradients (unit upstream)
    int N, L, S, variant;      // variant 1 | 2
    float img_w, img_h;
    float cls_w, reg_w, iou_w;
    float alpha0, alpha1;
};

// scratch layout (floats): focal [6][N] | pairs: reg [6][MAXP], iou [6][MAXP] | pair rows (as float bits are avoided: int64 arrays apart)
__global__ __launch_bounds__(4 * NT) void variant_terms_kernel(
    VarParams p, const float* __restrict__ tgt, int64_t* __restrict__ pair_rows_all /* [6][MAXP] */,
    int64_t* __restrict__ pair_cols_all /* [6][MAXP] */, int64_t* __restrict__ rows_sorted_all /* [6][L] (variant 1) */,
    float* __restrict__ focal_all, float* __restrict__ pair_reg /* [6][MAXP] */, float* __restrict__ pair_iou /* [6][MAXP] */)
{
    extern __shared__ float cost[];
    __shared__ int s_row[MAXP], s_col[MAXP], s_np;
    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = p.N, L = p.L, S = p.S, W = 6 + S;
    const float* pred = p.pred[q];
    float* dpred = p.dpred[q];
    int64_t* prow = pair_rows_all + q * MAXP;
    int64_t* pcol = pair_cols_all + q * MAXP;
    if (p.variant == 2) {
        lane_assign_block<true, 4>(pred, tgt, N, L, S, p.img_w, p.img_h, nullptr, nullptr, nullptr, nullptr, cost, 0.5f, prow, pcol, nullptr);
    } else {
        // one-to-one: pairs = (rows_by_col[j], j) for the matched labels; rows_sorted is what the caller gets as `matched`
        lane_assign_block<false, 4>(pred, tgt, N, L, S, p.img_w, p.img_h, prow /* rows_by_col[L] */, rows_sorted_all + q * L, nullptr, nullptr, cost);
    }
    __syncthreads();
    if (tid == 0) {
        int np = 0;
        if (p.variant == 2) {
            for (int i = 0; i < MAXP; ++i) { s_row[i] = (int)prow[i]; s_col[i] = (int)pcol[i]; np += s_row[i] >= 0; }
        } else {
            for (int j = 0; j < MAXP; ++j) {
                const int r = j < L ? (int)prow[j] : -1;
                s_row[j] = r; s_col[j] = r >= 0 ? j : -1; np += r >= 0;
            }
            for (int j = 0; j < MAXP; ++j) { pcol[j] = s_col[j]; if (j >= L) prow[j] = -1; }
        }
        s_np = np;
    }
    __syncthreads();
    const int m = max(s_np, 1);

    // ---- focal term per anchor + gradient w.r.t. the two logits (focal_loss.py:78-136, alpha = (0.1, 0.9), gamma = 2) ----
    if (tid < N) {
        const int i = tid;
        bool pos = false;
#pragma unroll
        for (int j = 0; j < MAXP; ++j) pos |= (s_row[j] == i);
        const float z0 = pred[(size_t)i * W], z1 = pred[(size_t)i * W + 1];
        const float zm = fmaxf(z0, z1);
        const float e0 = expf(z0 - zm), e1 = expf(z1 - zm);
        const float q0 = e0 / (e0 + e1), q1 = e1 / (e0 + e1);
        const float p0 = q0 + 1e-6f, p1 = q1 + 1e-6f;
        const float o0 = (pos ? 0.f : 1.f) + 1e-6f, o1 = (pos ? 1.f : 0.f) + 1e-6f;
        const float l0 = logf(p0), l1 = logf(p1);
        const float f0 = -p.alpha0 * (1.f - p0) * (1.f - p0) * l0, f1 = -p.alpha1 * (1.f - p1) * (1.f - p1) * l1;
        focal_all[(size_t)q * N + i] = o0 * f0 + o1 * f1;
        const float g0 = o0 * p.alpha0 * (2.f * (1.f - p0) * l0 - (1.f - p0) * (1.f - p0) / p0);
        const float g1 = o1 * p.alpha1 * (2.f * (1.f - p1) * l1 - (1.f - p1) * (1.f - p1) / p1);
        const float gq = g0 * q0 + g1 * q1;
        const float d = (p.gate[0][i] + p.gate[1][i] + p.gate[2][i]) / 3.0f;
        const float wgt = (p.cls_w / 3.0f) * (q < 3 ? (1.f - d) : d);
        float* dr = dpred + (size_t)i * W;
        dr[0] = wgt * q0 * (g0 - gq);
        dr[1] = wgt * q1 * (g1 - gq);
        for (int c = 2; c < W; ++c) dr[c] = 0.f;
    }
    __syncthreads();

    // ---- per pair: smooth-L1 on (start_y, start_x, theta, length) and 1 - line IoU (radius 15 px): wave <-> pair ----
    // variant 2: the terms enter the total with a fixed weight, the gradient rows are final here.
    // variant 1: the weight of a pair is the gate factor of the anchor that the LAST stage matched at the pair's position - the
    //            finalize kernel scales the rows written here (columns 2..) by it.
    float reg_v = 0.f, iou_v = 0.f;
    if (wave < MAXP && s_row[wave] >= 0) {
        const int r = s_row[wave], j = s_col[wave];
        const float* pr = pred + (size_t)r * W;
        const float* tr = tgt + (size_t)j * W;
        float* dr = dpred + (size_t)r * W;
        const float wr = p.variant == 2 ? p.reg_w / 6.0f : p.reg_w / 3.0f;      // (reg_a + reg_b) * w / 2, each a mean over 3 stages | w / 3
        const float wi = p.variant == 2 ? p.iou_w / 6.0f : p.iou_w / 3.0f;
        float part = 0.f;
        if (lane < 4) {
            const float scale = lane == 0 ? (float)(S - 1) : lane == 1 ? (p.img_w - 1.0f) : lane == 2 ? 180.0f : (float)(S - 1);
            const float x = (pr[2 + lane] - tr[2 + lane]) * scale;
            const float ax = fabsf(x);
            part = ax < 1.0f ? 0.5f * x * x : ax - 0.5f;
            const float gx = ax < 1.0f ? x : (x > 0.f ? 1.f : -1.f);
            dr[2 + lane] = wr / ((float)m * 4.0f) * gx * scale;
        }
        reg_v = wave_sum(part) / ((float)m * 4.0f);
        const float R = 15.0f, sx = p.img_w - 1.0f;
        float O = 0.f, U = 0.f;
        for (int k = lane; k < S; k += 64) {
            const float x = pr[6 + k] * sx, t = tr[6 + k];
            if (!((t < 0.f) || (t >= p.img_w))) {
                O += fminf(x + R, t + R) - fmaxf(x - R, t - R);
                U += fmaxf(x + R, t + R) - fminf(x - R, t - R);
            }
        }
        O = wave_sum(O); U = wave_sum(U);
        const float Ue = U + 1e-9f;
        iou_v = (1.0f - O / Ue) / (float)m;
        for (int k = lane; k < S; k += 64) {
            const float x = pr[6 + k] * sx, t = tr[6 + k];
            float g = 0.f;
            if (!((t < 0.f) || (t >= p.img_w))) {
                const float dO = (x < t ? 1.f : (x == t ? 0.5f : 0.f)) - (x > t ? 1.f : (x == t ? 0.5f : 0.f));
                const float dU = -dO;
                g = -(dO * Ue - O * dU) / (Ue * Ue);             // d(1 - O/U)/dx
            }
            dr[6 + k] = wi / (float)m * g * sx;
        }
    }
    if (wave < MAXP && lane == 0) {
        pair_reg[q * MAXP + wave] = reg_v;
        pair_iou[q * MAXP + wave] = iou_v;
    }
}

__global__ __launch_bounds__(NT) void variant_finalize_kernel(
    VarParams p, const float* __restrict__ focal_all, const float* __restrict__ pair_reg, const float* __restrict__ pair_iou,
    const int64_t* __restrict__ pair_rows_all, const int64_t* __restrict__ rows_sorted_all,
    float* __restrict__ loss_out, float* __restrict__ dgate /* [3][N] */)
{
    __shared__ float diff[NT], s_d[NT];
    __shared__ float red[4], s_delta;
    __shared__ float pos_reg[2][MAXL], pos_iou[2][MAXL];        // variant 1: pair terms summed over the stages by position
    __shared__ int last_row[2][MAXL];                            // variant 1: anchor matched by the LAST stage at position
    __shared__ int pos_of[6][MAXL];                              // variant 1: position of label j's pair in (branch, stage) q
    const int tid = threadIdx.x, N = p.N, L = p.L, W = 6 + p.S;
    float ca = 0.f, cb = 0.f, d = 0.f;
    if (tid < N) {
        ca = (focal_all[0 * N + tid] + focal_all[1 * N + tid] + focal_all[2 * N + tid]) / 3.0f;
        cb = (focal_all[3 * N + tid] + focal_all[4 * N + tid] + focal_all[5 * N + tid]) / 3.0f;
        d = (p.gate[0][tid] + p.gate[1][tid] + p.gate[2][tid]) / 3.0f;
    }
    s_d[tid] = d;
    float la = ca, lb = cb;                                      // variant 2: the balanced vector is the focal vector
    if (p.variant == 1) {
        if (tid < 2 * MAXL) { pos_reg[tid / MAXL][tid % MAXL] = 0.f; pos_iou[tid / MAXL][tid % MAXL] = 0.f; }
        __syncthreads();
        if (tid == 0) {
            for (int q = 0; q < 6; ++q) {
                const int br = q / 3;
                for (int j = 0; j < MAXL; ++j) {
                    const long long r = j < L ? pair_rows_all[q * MAXP + j] : -1;
                    int pos = -1;
                    if (r >= 0) {                                // position = number of matched anchors below this one
                        pos = 0;
                        for (int k = 0; k < L; ++k) { const long long rk = pair_rows_all[q * MAXP + k]; pos += (rk >= 0 && rk < r); }
                        pos_reg[br][pos] += pair_reg[q * MAXP + j];
                        pos_iou[br][pos] += pair_iou[q * MAXP + j];
                    }
                    pos_of[q][j] = pos;
                }
            }
            for (int br = 0; br < 2; ++br)
                for (int k = 0; k < MAXL; ++k) last_row[br][k] = k < L ? (int)rows_sorted_all[(br * 3 + 2) * L + k] : -1;
        }
        __syncthreads();
        la = ca * p.cls_w; lb = cb * p.cls_w;
        for (int k = 0; k < MAXL; ++k) {
            if (last_row[0][k] == tid) la += (pos_reg[0][k] / 3.0f) * p.reg_w + (pos_iou[0][k] / 3.0f) * p.iou_w;
            if (last_row[1][k] == tid) lb += (pos_reg[1][k] / 3.0f) * p.reg_w + (pos_iou[1][k] / 3.0f) * p.iou_w;
        }
    }
    if (tid < N) diff[tid] = la - lb;
    __syncthreads();
    if (tid < N) {                                               // torch.median = lower median = sorted[(N-1)/2]
        const float v = diff[tid];
        int rank = 0;
        for (int j = 0; j < N; ++j) { const float u = diff[j]; rank += (u < v) || (u == v && j < tid); }
        if (rank == (N - 1) / 2) s_delta = v;
    }
    __syncthreads();
    const float delta = s_delta;
    float term = 0.f;
    if (tid < N) {
        term = (1.f - d) * (la - delta * 0.5f) + d * (lb + delta * 0.5f);
        const float g = (p.variant == 1 ? 1.0f : p.cls_w) / 3.0f * (lb - la + delta);
        dgate[0 * N + tid] = g; dgate[1 * N + tid] = g; dgate[2 * N + tid] = g;
    }
    term = wave_sum(term);
    if ((tid & 63) == 0) red[tid >> 6] = term;
    __syncthreads();
    if (tid == 0) {
        const float bal = (red[0] + red[1]) + (red[2] + red[3]);
        if (p.variant == 1) {
            loss_out[0] = bal;
        } else {
            float reg = 0.f, iou = 0.f;
            for (int i = 0; i < 6 * MAXP; ++i) { reg += pair_reg[i]; iou += pair_iou[i]; }
            loss_out[0] = (reg / 3.0f) * p.reg_w * 0.5f + (iou / 3.0f) * p.iou_w * 0.5f + bal * p.cls_w;
        }
    }
    // variant 1: the pair gradient rows take the gate factor of the anchor the last stage matched at the pair's position
    if (p.variant == 1) {
        for (int idx = tid; idx < 6 * MAXL * (W - 2); idx += NT) {
            const int c = 2 + idx % (W - 2), j = (idx / (W - 2)) % MAXL, q = idx / ((W - 2) * MAXL);
            const long long r = j < L ? pair_rows_all[q * MAXP + j] : -1;
            if (r < 0) continue;
            const int pos = pos_of[q][j], br = q / 3;
            const int a = pos >= 0 ? last_row[br][pos] : -1;
            const float f = a >= 0 ? (br == 0 ? 1.f - s_d[a] : s_d[a]) : 0.f;
            p.dpred[q][(size_t)r * W + c] *= f;
        }
    }
}

}  // namespace

// One frame of libs.utils.loss4OL.Criterion4OL (variant 1) or libs.utils.loss4OLV2.Criterion4OL (variant 2).
//   pred[6] : [N][6+S] (branch A stages 0,1,2 then branch B stages 0,1,2); gate[3] : [N]; tgt : [L][6+S], L <= 4, N <= 256
// Outputs (caller-allocated): loss [1]; dpred[6] [N][6+S], dgate [3][N] = d loss / d input;
//   pair_rows / pair_cols [6][16] int64: the (anchor, label row) pairs per (branch, stage), -1 padded (variant 1: entry j =
//   label j; variant 2: the reference's round order); rows_sorted [6][L] int64 (variant 1: matched anchors ascending);
//   scratch: 6*N + 2*6*16 floats.
PHNET_API int phnet_frame_loss_variant(int32_t variant, const float* const* pred, const float* const* gate, const float* tgt,
                                       int32_t N, int32_t L, int32_t S, float img_w, float img_h,
                                       float cls_w, float reg_w, float iou_w,
                                       float* loss, float* const* dpred, float* dgate,
                                       int64_t* pair_rows, int64_t* pair_cols, int64_t* rows_sorted, float* scratch, void* stream)
{
    if ((variant != 1 && variant != 2) || N < 1 || N > NT || L < 1 || L > MAXL || S < 3 || S > 250) return PHNET_ERR_ARG;
    if (!pred || !gate || !tgt || !loss || !dpred || !dgate || !pair_rows || !pair_cols || !rows_sorted || !scratch) return PHNET_ERR_ARG;
    VarParams p{};
    for (int i = 0; i < 6; ++i) { p.pred[i] = pred[i]; p.dpred[i] = dpred[i]; if (!pred[i] || !dpred[i]) return PHNET_ERR_ARG; }
    for (int i = 0; i < 3; ++i) { p.gate[i] = gate[i]; if (!gate[i]) return PHNET_ERR_ARG; }
    p.N = N; p.L = L; p.S = S; p.variant = variant; p.img_w = img_w; p.img_h = img_h;
    p.cls_w = cls_w; p.reg_w = reg_w; p.iou_w = iou_w; p.alpha0 = 0.1f; p.alpha1 = 0.9f;
    float* focal = scratch;
    float* preg = scratch + (size_t)6 * N;
    float* piou = preg + 6 * MAXP;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(variant_terms_kernel, dim3(6), dim3(4 * NT), (size_t)2 * N * MAXL * sizeof(float), st,
                       p, tgt, pair_rows, pair_cols, rows_sorted, focal, preg, piou);
    hipLaunchKernelGGL(variant_finalize_kernel, dim3(1), dim3(NT), 0, st, p, (const float*)focal, (const float*)preg, (const float*)piou,
                       (const int64_t*)pair_rows, (const int64_t*)rows_sorted, loss, dgate);
    return phnet_launch_status();
}
