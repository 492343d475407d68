// Fused per-frame criterion of the two-branch lane head for gfx950: label assignment, focal / smooth-L1 / LaneIoU
// terms, the gate-weighted combination AND the gradients w.r.t. all six prediction tensors and three gate tensors,
// in two launches per frame (the reference issues ~250 tiny ATen kernels per (branch, stage) plus a host sync).
//
// Replaces libs/utils/loss4OLV3.py:34-82 (line_loss_diff) and :100-123 (loss4OneStep), with
//   dynamic_assign.py:128-190 (assign + Hungarian), focal_loss.py:78-136 (softmax focal, per-class alpha),
//   F.smooth_l1_loss, dynamic_assignV2.py:55-98 (LaneIoU with detached prediction widths, class-default geometry).
// Latency-bound: 6 workgroups (one per branch x stage) + 1 finalize workgroup; everything lives in registers/LDS.
#include "assign_device.h"

namespace {

using namespace phassign;

struct LossParams {
    const float* pred[6];      // [N][6+S] : branch A stages 0..2, branch B stages 0..2
    const float* gate[3];      // [N]
    float* dpred[6];           // [N][6+S] gradients (unit upstream)
    int N, L, S;
    float img_w, img_h;
    float cls_w, reg_w, iou_w;
    float alpha0, alpha1;
    float liou_hw, liou_h, liou_w;
};

constexpr int MAXT = 8;                    // frames per launch (phnet_clip_loss)
struct ClipParams { LossParams f[MAXT]; };

__device__ __forceinline__ float wave_sum_all(float v) { return wave_sum(v); }

// grid (6, T): blockIdx.y = frame; the per-frame outputs follow each other (tgt [T][L][6+S], rows [T][6][L], focal [T][6][N],
// scalars [T][12])
__global__ __launch_bounds__(NT) void frame_loss_terms_kernel(
    ClipParams cp, const float* __restrict__ tgt, int64_t* __restrict__ rows_by_col_all, int64_t* __restrict__ rows_sorted_all,
    float* __restrict__ focal_all, float* __restrict__ scalars /* [6][2] reg, iou */)
{
    extern __shared__ float cost[];
    __shared__ int s_rows[MAXL];
    __shared__ int s_nvalid;
    const LossParams& p = cp.f[blockIdx.y];
    const int q = blockIdx.x;                     // pair index: branch*3 + stage
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = p.N, L = p.L, S = p.S, W = 6 + S;
    {
        const size_t f = blockIdx.y;
        tgt += f * L * W; rows_by_col_all += f * 6 * L; rows_sorted_all += f * 6 * L; focal_all += f * 6 * N; scalars += f * 12;
    }
    const float* pred = p.pred[q];
    float* dpred = p.dpred[q];
    int64_t* rows_by_col = rows_by_col_all + q * L;
    lane_assign_block(pred, tgt, N, L, S, p.img_w, p.img_h, rows_by_col, rows_sorted_all + q * L, nullptr, nullptr, cost);
    __syncthreads();
    if (tid == 0) {
        int nv = 0;
        for (int j = 0; j < MAXL; ++j) { s_rows[j] = j < L ? (int)rows_by_col[j] : -1; nv += s_rows[j] >= 0; }
        s_nvalid = nv;
    }
    __syncthreads();
    const int m = s_nvalid;

    // ---- focal term per anchor + gradient w.r.t. the two logits ------------------------------------------------
    if (tid < N) {
        const int i = tid;
        bool pos = false;
#pragma unroll
        for (int j = 0; j < MAXL; ++j) pos |= (s_rows[j] == i);
        const float z0 = pred[(size_t)i * W], z1 = pred[(size_t)i * W + 1];
        const float zm = fmaxf(z0, z1);
        const float e0 = expf(z0 - zm), e1 = expf(z1 - zm);
        const float q0 = e0 / (e0 + e1), q1 = e1 / (e0 + e1);
        const float p0 = q0 + 1e-6f, p1 = q1 + 1e-6f;
        const float o0 = (pos ? 0.f : 1.f) + 1e-6f, o1 = (pos ? 1.f : 0.f) + 1e-6f;
        const float l0 = logf(p0), l1 = logf(p1);
        const float f0 = -p.alpha0 * (1.f - p0) * (1.f - p0) * l0, f1 = -p.alpha1 * (1.f - p1) * (1.f - p1) * l1;
        focal_all[(size_t)q * N + i] = o0 * f0 + o1 * f1;
        // dF/dp_c = o_c * alpha_c * (2 (1-p_c) log p_c - (1-p_c)^2 / p_c)
        const float g0 = o0 * p.alpha0 * (2.f * (1.f - p0) * l0 - (1.f - p0) * (1.f - p0) / p0);
        const float g1 = o1 * p.alpha1 * (2.f * (1.f - p1) * l1 - (1.f - p1) * (1.f - p1) / p1);
        const float gq = g0 * q0 + g1 * q1;
        float d = (p.gate[0][i] + p.gate[1][i] + p.gate[2][i]) / 3.0f;
        const float wgt = (p.cls_w / 3.0f) * (q < 3 ? (1.f - d) : d);
        float* dr = dpred + (size_t)i * W;
        dr[0] = wgt * q0 * (g0 - gq);
        dr[1] = wgt * q1 * (g1 - gq);
        for (int c = 2; c < W; ++c) dr[c] = 0.f;
    }
    __syncthreads();

    // ---- regression + LaneIoU on the matched anchors: wave j <-> label row j -------------------------------------
    float reg_part = 0.f, iou_part = 0.f;
    if (wave < L && s_rows[wave] >= 0) {
        const int j = wave, r = s_rows[j];
        const float* pr = pred + (size_t)r * W;
        const float* tr = tgt + (size_t)j * W;
        float* dr = dpred + (size_t)r * W;
        if (lane < 4) {
            const float scale = lane == 0 ? (float)(S - 1) : lane == 1 ? (p.img_w - 1.0f) : lane == 2 ? 180.0f : (float)(S - 1);
            const float x = (pr[2 + lane] - tr[2 + lane]) * scale;
            const float ax = fabsf(x);
            reg_part = ax < 1.0f ? 0.5f * x * x : ax - 0.5f;
            const float gx = ax < 1.0f ? x : (x > 0.f ? 1.f : -1.f);
            dr[2 + lane] = (p.reg_w / 3.0f) / ((float)m * 4.0f) * gx * scale;
        }
        reg_part = wave_sum_all(reg_part) / ((float)m * 4.0f);
        // LaneIoU: pred xs * (W-1)/W vs label xs / W, virtual widths from the local slope (prediction side detached)
        const float sx = (p.img_w - 1.0f) / p.img_w, dy = p.liou_h / (float)(S - 1) * 2.0f;
        float O = 0.f, U = 0.f;
        for (int k = lane; k < S; k += 64) {
            const int c = min(max(k, 1), S - 2);                  // widths are replicated at both ends
            const float pd = (pr[6 + c + 1] * sx - pr[6 + c - 1] * sx) * p.liou_w;
            float td = (tr[6 + c + 1] / p.img_w - tr[6 + c - 1] / p.img_w) * p.liou_w;
            if (fabsf(td) > 1e4f) td = 0.f;
            const float pw = p.liou_hw * sqrtf(pd * pd + dy * dy) / dy, tw = p.liou_hw * sqrtf(td * td + dy * dy) / dy;
            const float x = pr[6 + k] * sx, t = tr[6 + k] / p.img_w;
            if (!((t < 0.f) || (t >= 1.0f))) {
                O += fminf(x + pw, t + tw) - fmaxf(x - pw, t - tw);
                U += fmaxf(x + pw, t + tw) - fminf(x - pw, t - tw);
            }
        }
        O = wave_sum_all(O); U = wave_sum_all(U);
        const float Ue = U + 1e-9f;
        iou_part = (1.0f - O / Ue) / (float)m;
        const float gscale = (p.iou_w / 3.0f) / (float)m;
        for (int k = lane; k < S; k += 64) {
            const int c = min(max(k, 1), S - 2);
            const float pd = (pr[6 + c + 1] * sx - pr[6 + c - 1] * sx) * p.liou_w;
            float td = (tr[6 + c + 1] / p.img_w - tr[6 + c - 1] / p.img_w) * p.liou_w;
            if (fabsf(td) > 1e4f) td = 0.f;
            const float pw = p.liou_hw * sqrtf(pd * pd + dy * dy) / dy, tw = p.liou_hw * sqrtf(td * td + dy * dy) / dy;
            const float x = pr[6 + k] * sx, t = tr[6 + k] / p.img_w;
            float g = 0.f;
            if (!((t < 0.f) || (t >= 1.0f))) {
                const float a = x + pw, b = t + tw, c2 = x - pw, d2 = t - tw;
                const float dO = (a < b ? 1.f : (a == b ? 0.5f : 0.f)) - (c2 > d2 ? 1.f : (c2 == d2 ? 0.5f : 0.f));
                const float dU = (a > b ? 1.f : (a == b ? 0.5f : 0.f)) - (c2 < d2 ? 1.f : (c2 == d2 ? 0.5f : 0.f));
                g = -(dO * Ue - O * dU) / (Ue * Ue);             // d(1 - O/U)/dx
            }
            dr[6 + k] = gscale * g * sx;
        }
    }
    // one value per wave lives in lane 0 after the shuffles
    __shared__ float s_reg[4], s_iou[4];
    if (lane == 0) { s_reg[wave] = reg_part; s_iou[wave] = iou_part; }
    __syncthreads();
    if (tid == 0) {
        scalars[q * 2 + 0] = (s_reg[0] + s_reg[1]) + (s_reg[2] + s_reg[3]);
        scalars[q * 2 + 1] = (s_iou[0] + s_iou[1]) + (s_iou[2] + s_iou[3]);
    }
}

// grid T: blockIdx.x = frame (loss [T], dgate [T][3][N])
__global__ __launch_bounds__(NT) void frame_loss_finalize_kernel(
    ClipParams cp, const float* __restrict__ focal_all, const float* __restrict__ scalars,
    float* __restrict__ loss_out, float* __restrict__ dgate /* [3][N] */)
{
    __shared__ float diff[NT];
    __shared__ float red[4], s_delta;
    const LossParams& p = cp.f[blockIdx.x];
    const int tid = threadIdx.x, N = p.N;
    {
        const size_t f = blockIdx.x;
        focal_all += f * 6 * N; scalars += f * 12; loss_out += f; dgate += f * 3 * N;
    }
    float ca = 0.f, cb = 0.f, d = 0.f;
    if (tid < N) {
        ca = (focal_all[0 * N + tid] + focal_all[1 * N + tid] + focal_all[2 * N + tid]) / 3.0f;
        cb = (focal_all[3 * N + tid] + focal_all[4 * N + tid] + focal_all[5 * N + tid]) / 3.0f;
        d = (p.gate[0][tid] + p.gate[1][tid] + p.gate[2][tid]) / 3.0f;
        diff[tid] = ca - cb;
    }
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
        term = (1.f - d) * (ca - delta * 0.5f) + d * (cb + delta * 0.5f);
        const float g = (p.cls_w / 3.0f) * (cb - ca + delta);
        dgate[0 * N + tid] = g; dgate[1 * N + tid] = g; dgate[2 * N + tid] = g;
    }
    term = wave_sum(term);
    if ((tid & 63) == 0) red[tid >> 6] = term;
    __syncthreads();
    if (tid == 0) {
        const float cls = (red[0] + red[1]) + (red[2] + red[3]);
        float reg = 0.f, iou = 0.f;
        for (int q = 0; q < 6; ++q) { reg += scalars[q * 2]; iou += scalars[q * 2 + 1]; }
        loss_out[0] = (reg / 3.0f) * p.reg_w + (iou / 3.0f) * p.iou_w + cls * p.cls_w;
    }
}

}  // namespace

// One frame of Criterion4OL (loss4OLV3.py:100-123) with both branches and all three stages.
//   pred[6]  : [N][6+S] predictions (branch A stages 0,1,2 then branch B stages 0,1,2); gate[3] : [N] gate scores
//   tgt      : [L][6+S] label rows (col 1 == 1 marks a valid lane), L <= 4, N <= 256
//   weights  : {cls_weight, reg_weight, iou_weight};  focal alpha = {0.1, 0.9}, gamma = 2
// Outputs (caller-allocated): loss [1]; dpred[6] [N][6+S] and dgate [3][N] = d loss / d input (unit upstream);
//   rows_by_col / rows_sorted [6][L] int64 (matched anchors per branch x stage, -1 padded);
//   scratch: focal [6][N] f32, scalars [12] f32.
static int fill_frame(LossParams& p, const float* const* pred, const float* const* gate, float* const* dpred, int32_t N, int32_t L, int32_t S,
                      float img_w, float img_h, float cls_w, float reg_w, float iou_w, float liou_half_width, float liou_img_h, float liou_img_w)
{
    for (int i = 0; i < 6; ++i) { p.pred[i] = pred[i]; p.dpred[i] = dpred[i]; if (!pred[i] || !dpred[i]) return PHNET_ERR_ARG; }
    for (int i = 0; i < 3; ++i) { p.gate[i] = gate[i]; if (!gate[i]) return PHNET_ERR_ARG; }
    p.N = N; p.L = L; p.S = S; p.img_w = img_w; p.img_h = img_h;
    p.cls_w = cls_w; p.reg_w = reg_w; p.iou_w = iou_w; p.alpha0 = 0.1f; p.alpha1 = 0.9f;
    p.liou_hw = liou_half_width; p.liou_h = liou_img_h; p.liou_w = liou_img_w;
    return PHNET_OK;
}

// The frames of a clip in the same two launches: frame t's loss needs nothing from frame t' (the criterion is called per frame,
// trainOLV3.py:150-171, and the losses are added up).  pred [T*6], gate [T*3], dpred [T*6] pointers (frame-major); tgt [T][L][6+S];
// loss [T]; dgate [T][3][N]; rows_by_col / rows_sorted [T][6][L]; scratch focal [T][6][N], scalars [T][12].  T <= 8.
PHNET_API int phnet_clip_loss(const float* const* pred, const float* const* gate, const float* tgt, int32_t T,
                              int32_t N, int32_t L, int32_t S, float img_w, float img_h,
                              float cls_w, float reg_w, float iou_w,
                              float liou_half_width, float liou_img_h, float liou_img_w,
                              float* loss, float* const* dpred, float* dgate,
                              int64_t* rows_by_col, int64_t* rows_sorted, float* focal, float* scalars, void* stream)
{
    if (T < 1 || T > MAXT || N < 1 || N > NT || L < 1 || L > MAXL || S < 3 || S > 250) return PHNET_ERR_ARG;
    if (!pred || !gate || !tgt || !loss || !dpred || !dgate || !rows_by_col || !rows_sorted || !focal || !scalars) return PHNET_ERR_ARG;
    ClipParams cp{};
    for (int t = 0; t < T; ++t) {
        const int rc = fill_frame(cp.f[t], pred + 6 * t, gate + 3 * t, dpred + 6 * t, N, L, S, img_w, img_h, cls_w, reg_w, iou_w,
                                  liou_half_width, liou_img_h, liou_img_w);
        if (rc != PHNET_OK) return rc;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(frame_loss_terms_kernel, dim3(6, T), dim3(NT), (size_t)N * MAXL * sizeof(float), st,
                       cp, tgt, rows_by_col, rows_sorted, focal, scalars);
    hipLaunchKernelGGL(frame_loss_finalize_kernel, dim3(T), dim3(NT), 0, st, cp, (const float*)focal, (const float*)scalars, loss, dgate);
    return phnet_launch_status();
}

PHNET_API int phnet_frame_loss(const float* const* pred, const float* const* gate, const float* tgt,
                               int32_t N, int32_t L, int32_t S, float img_w, float img_h,
                               float cls_w, float reg_w, float iou_w,
                               float liou_half_width, float liou_img_h, float liou_img_w,
                               float* loss, float* const* dpred, float* dgate,
                               int64_t* rows_by_col, int64_t* rows_sorted, float* focal, float* scalars, void* stream)
{
    return phnet_clip_loss(pred, gate, tgt, 1, N, L, S, img_w, img_h, cls_w, reg_w, iou_w, liou_half_width, liou_img_h, liou_img_w, loss,
                           dpred, dgate, rows_by_col, rows_sorted, focal, scalars, stream);
}

