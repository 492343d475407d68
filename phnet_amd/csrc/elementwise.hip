// Small streaming kernels of the lane head (HBM-bound, 16-byte accesses where the shape allows).
#include "common.h"

namespace {
constexpr int NT = 256;
typedef float f32x4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(NT) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ dx, long n)
{
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    if (i < n) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// ---- transformer glue (utils/transformer.py:275-298): residual + dropout, GELU + dropout ------------------------------
// y = res + drop(x);   MODE 1: dx = drop'(dy) (the residual's gradient is dy itself)
__global__ __launch_bounds__(NT) void dropout_add_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                         float* __restrict__ y, long n, DropRng rng, float scale)
{
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    const bool kept = !rng.thresh || phnet_rng_keep(phnet_rng_seed(rng), phnet_rng_index(rng, (uint64_t)i), rng.thresh);
    const float v = kept ? x[i] * scale : 0.f;
    y[i] = res ? res[i] + v : v;
}

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float v) {
    return 0.5f * (1.0f + erff(v * 0.70710678118654752f)) + v * 0.39894228040143268f * expf(-0.5f * v * v);
}

// y = drop(gelu(x));  backward: dx = drop'(dy) * gelu'(x)
__global__ __launch_bounds__(NT) void gelu_dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n,
                                                              DropRng rng, float scale)
{
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    const bool kept = !rng.thresh || phnet_rng_keep(phnet_rng_seed(rng), phnet_rng_index(rng, (uint64_t)i), rng.thresh);
    y[i] = kept ? gelu_erf(x[i]) * scale : 0.f;
}
__global__ __launch_bounds__(NT) void gelu_dropout_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              float* __restrict__ dx, long n, DropRng rng, float scale)
{
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    const bool kept = !rng.thresh || phnet_rng_keep(phnet_rng_seed(rng), phnet_rng_index(rng, (uint64_t)i), rng.thresh);
    dx[i] = kept ? dy[i] * scale * gelu_erf_grad(x[i]) : 0.f;
}

// ---- AdamW over the flat parameter / gradient arenas (libs/utils/optimizer.py:33-35: optim.AdamW) -------------------------------
// decoupled weight decay on elements [0, n_decay); same update order as torch's single-tensor AdamW:
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(NT) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n4, long n_decay, const long long* __restrict__ step,
                                                   float lr, const float* __restrict__ lr_dev, float b1, float b2, float eps, float wd)
{
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    if (i >= n4) return;
    if (lr_dev) lr = lr_dev[0];                      // learning rate from device memory: a replayed hipGraph follows the schedule
    const float t = (float)step[0];
    const float c1 = 1.0f - powf(b1, t), c2s = sqrtf(1.0f - powf(b2, t));
    const float step_size = lr / c1;
    f32x4v pp = reinterpret_cast<f32x4v*>(p)[i], mm = reinterpret_cast<f32x4v*>(m)[i], vv = reinterpret_cast<f32x4v*>(v)[i];
    const f32x4v gg = reinterpret_cast<const f32x4v*>(g)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float pe = pp[e];
        if (i * 4 + e < n_decay) pe *= 1.0f - lr * wd;
        const float me = b1 * mm[e] + (1.0f - b1) * gg[e];
        const float ve = b2 * vv[e] + (1.0f - b2) * gg[e] * gg[e];
        pe -= step_size * (me / (sqrtf(ve) / c2s + eps));
        pp[e] = pe; mm[e] = me; vv[e] = ve;
    }
    reinterpret_cast<f32x4v*>(p)[i] = pp; reinterpret_cast<f32x4v*>(m)[i] = mm; reinterpret_cast<f32x4v*>(v)[i] = vv;
}

// ---- cross-frame memory tokens (Router4OL.py:563-584): the positives' features in prior-index order, then the mean of
// all other anchors.  rows i64[L] ascending anchor ids, -1 padded; tokens [L+1][E]; valid u8[L+1] ------------------------
__global__ __launch_bounds__(1024) void memory_tokens_kernel(const float* __restrict__ feat, const long long* __restrict__ rows,
                                                             float* __restrict__ tokens, unsigned char* __restrict__ valid,
                                                             int N, int E, int L)
{
    extern __shared__ float part[];                  // [groups][E]
    {   // clip blockIdx.x of a batch
        const size_t b = blockIdx.x;
        feat += b * N * E; rows += b * L; tokens += b * (L + 1) * E; valid += b * (L + 1);
    }
    const int e = threadIdx.x % E, grp = threadIdx.x / E, groups = blockDim.x / E;
    float s = 0.f;
    if (grp < groups)
        for (int n = grp; n < N; n += groups) s += feat[(size_t)n * E + e];
    if (grp < groups) part[grp * E + e] = s;
    __syncthreads();
    if (grp != 0) return;
    float total = 0.f;
    for (int g2 = 0; g2 < groups; ++g2) total += part[g2 * E + e];
    float possum = 0.f;
    int cnt = 0;
    for (int l = 0; l < L; ++l) {
        const long long r = rows[l];
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

// ---- routing-gate tail (Router.py:76-80): gate = sigmoid(relu(h . w + b)), one wavefront per anchor ------------------------
__global__ __launch_bounds__(NT) void gate_tail_fwd_kernel(const float* __restrict__ h, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ out, int N, int K)
{
    const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += h[(size_t)row * K + k] * w[k];
    s = wave_sum(s) + b[0];
    if (lane == 0) out[row] = 1.0f / (1.0f + expf(-fmaxf(s, 0.f)));
}

// dpre_n = dout_n * out_n (1 - out_n) where the ReLU was open (out_n > 0.5);  dh = dpre (x) w,  dw += sum_n dpre_n h_n,
// db += sum_n dpre_n.  A 1024-thread workgroup owns 64 columns k: 16 row groups walk the anchors (coalesced along k) and
// their partial column sums are folded through LDS in a fixed order.
__global__ __launch_bounds__(1024) void gate_tail_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                            const float* __restrict__ h, const float* __restrict__ w,
                                                            float* __restrict__ dh, float* __restrict__ dw, float* __restrict__ db,
                                                            int N, int K, int accumulate)
{
    constexpr int CH = 2048;                            // anchors per pass (stage 0 batches all frames of a clip: N = T * 240)
    __shared__ float g[CH];
    __shared__ float part[16][64];
    const int kc = threadIdx.x & 63, grp = threadIdx.x >> 6, k = blockIdx.x * 64 + kc;
    const float wk = k < K ? w[k] : 0.f;
    float acc = 0.f, bsum = 0.f;
    for (int base = 0; base < N; base += CH) {
        const int cnt = min(CH, N - base);
        __syncthreads();
        for (int n = threadIdx.x; n < cnt; n += 1024) {
            const float o = out[base + n];
            g[n] = o > 0.5f ? dout[base + n] * o * (1.0f - o) : 0.f;
        }
        __syncthreads();
        if (k < K) {
#pragma unroll 4
            for (int n = grp; n < cnt; n += 16) {
                const float gn = g[n];
                acc += gn * h[(size_t)(base + n) * K + k];
                if (dh) dh[(size_t)(base + n) * K + k] = gn * wk;
            }
        }
        if (blockIdx.x == 0 && grp == 1)
            for (int n = kc; n < cnt; n += 64) bsum += g[n];
    }
    part[grp][kc] = acc;
    __syncthreads();
    if (grp == 0 && k < K) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += part[r][kc];
        dw[k] = accumulate ? dw[k] + t : t;
    }
    if (blockIdx.x == 0 && grp == 1) {                   // one wavefront: the bias gradient
        const float s = wave_sum(bsum);
        if (kc == 0) db[0] = accumulate ? db[0] + s : s;
    }
}

// ---- stage hand-over (Router4OL.py:298-302): priors' = (1-g) lines_a + g lines_b,  on_map = priors'[:, 6 + idx] ---------------
__global__ __launch_bounds__(NT) void blend_priors_kernel(const float* __restrict__ gate, const float* __restrict__ a,
                                                          const float* __restrict__ b, const long long* __restrict__ idx,
                                                          float* __restrict__ priors, float* __restrict__ on_map, int N, int W, int P)
{
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i < N * W) {
        const float gt = gate[i / W];
        priors[i] = (1.0f - gt) * a[i] + gt * b[i];
    }
    if (i < N * P) {
        const int n = i / P, c = 6 + (int)idx[i - n * P];
        const float gt = gate[n];
        on_map[i] = (1.0f - gt) * a[(size_t)n * W + c] + gt * b[(size_t)n * W + c];
    }
}

// ---- lane prior update (Router4OL.py:328-345): one wavefront per anchor, lanes over the S x-columns --------------------
// head [N][HW] = (cls 2 | reg 4 | offsets S | pad); priors [N][6+S]; ys [S] = prior_ys
// lines = (cls, start_y/start_x/theta + tanh(reg[:3]), reg[3], xs(line));  preds = lines with xs + offsets
__global__ __launch_bounds__(NT) void lane_update_fwd_kernel(const float* __restrict__ priors, const float* __restrict__ head,
                                                             const float* __restrict__ ys, float* __restrict__ preds,
                                                             float* __restrict__ lines, int N, int S, int HW, float img_w, float img_h)
{
    // no fused multiply-adds here: theta*pi + 1e-5 sits next to tan's poles for near-horizontal anchors, where one
    // rounding more or less in the angle moves x by 1e-3; the reference rounds after the multiply and after the add
#pragma clang fp contract(off)
    const int i = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= N) return;
    const int W = 6 + S;
    const float* pr = priors + (size_t)i * W;
    const float* hd = head + (size_t)i * HW;
    float* po = preds + (size_t)i * W;
    float* lo = lines + (size_t)i * W;
    const float sy = pr[2] + tanhf(hd[2]), sx = pr[3] + tanhf(hd[3]), th = pr[4] + tanhf(hd[4]);
    const float tn = tanf(th * 3.14159265358979323846f + 1e-5f);
    if (lane == 0) {
        po[0] = lo[0] = hd[0]; po[1] = lo[1] = hd[1];
        po[2] = lo[2] = sy; po[3] = lo[3] = sx; po[4] = lo[4] = th; po[5] = lo[5] = hd[5];
    }
    for (int k = lane; k < S; k += 64) {
        const float x = (sx * (img_w - 1.0f) + ((1.0f - ys[k] - sy) * img_h / tn)) / (img_w - 1.0f);
        lo[6 + k] = x;
        po[6 + k] = x + hd[6 + k];
    }
}

__global__ __launch_bounds__(NT) void lane_update_bwd_kernel(const float* __restrict__ dpreds, const float* __restrict__ dlines,
                                                             const float* __restrict__ lines, const float* __restrict__ head,
                                                             const float* __restrict__ ys, float* __restrict__ dhead,
                                                             float* __restrict__ dpriors, int N, int S, int HW, float img_w, float img_h)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= N) return;
    const int W = 6 + S;
    const float* dp = dpreds ? dpreds + (size_t)i * W : nullptr;
    const float* dl = dlines ? dlines + (size_t)i * W : nullptr;
    const float* ln = lines + (size_t)i * W;
    const float* hd = head + (size_t)i * HW;
    float* dh = dhead + (size_t)i * HW;
    float* d = dpriors ? dpriors + (size_t)i * W : nullptr;
    auto G = [&](int c) { return (dp ? dp[c] : 0.f) + (dl ? dl[c] : 0.f); };
    const float sy = ln[2], th = ln[4];
    const float ang = th * 3.14159265358979323846f + 1e-5f;
    const float tn = tanf(ang), sn = sinf(ang);
    const float kx = img_h / (img_w - 1.0f);
    float gsy = 0.f, gsx = 0.f, gth = 0.f;
    for (int k = lane; k < S; k += 64) {
        const float gx = G(6 + k);
        gsx += gx;
        gsy += gx * (-kx / tn);
        gth += gx * (-(1.0f - ys[k] - sy) * kx * 3.14159265358979323846f / (sn * sn));
        dh[6 + k] = dp ? dp[6 + k] : 0.f;
        if (d) d[6 + k] = 0.f;
    }
    for (int c = W + lane; c < HW; c += 64) dh[c] = 0.f;
    gsy = wave_sum(gsy); gsx = wave_sum(gsx); gth = wave_sum(gth);
    if (lane != 0) return;
    gsy += G(2); gsx += G(3); gth += G(4);
    dh[0] = G(0); dh[1] = G(1);
    const float t2 = tanhf(hd[2]), t3 = tanhf(hd[3]), t4 = tanhf(hd[4]);
    dh[2] = gsy * (1.0f - t2 * t2);
    dh[3] = gsx * (1.0f - t3 * t3);
    dh[4] = gth * (1.0f - t4 * t4);
    dh[5] = G(5);
    if (d) { d[0] = d[1] = 0.f; d[2] = gsy; d[3] = gsx; d[4] = gth; d[5] = 0.f; }
}
}  // namespace

// Lane prior update: replaces the tanh / tan / repeat / cat chain of DetNetV2.forward_first/second (Router4OL.py:328-345).
// priors [N][6+S], head [N][HW] (cls 2, reg 4, offsets S, zero pad; HW >= 6+S), ys [S] -> preds, lines [N][6+S].
PHNET_API int phnet_lane_update_fwd(const float* priors, const float* head, const float* ys, float* preds, float* lines,
                                    int32_t N, int32_t S, int32_t HW, float img_w, float img_h, void* stream)
{
    if (N < 0 || S < 1 || HW < 6 + S) return PHNET_ERR_ARG;
    if (N == 0) return PHNET_OK;
    if (!priors || !head || !ys || !preds || !lines) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(lane_update_fwd_kernel, dim3((N + NT / 64 - 1) / (NT / 64)), dim3(NT), 0, (hipStream_t)stream,
                       priors, head, ys, preds, lines, N, S, HW, img_w, img_h);
    return phnet_launch_status();
}

// dpreds / dlines [N][6+S] (either may be NULL) -> dhead [N][HW]; dpriors [N][6+S] optional (only cols 2..4 non-zero).
PHNET_API int phnet_lane_update_bwd(const float* dpreds, const float* dlines, const float* lines, const float* head,
                                    const float* ys, float* dhead, float* dpriors,
                                    int32_t N, int32_t S, int32_t HW, float img_w, float img_h, void* stream)
{
    if (N < 0 || S < 1 || HW < 6 + S) return PHNET_ERR_ARG;
    if (N == 0) return PHNET_OK;
    if (!lines || !head || !ys || !dhead || (!dpreds && !dlines)) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(lane_update_bwd_kernel, dim3((N + NT / 64 - 1) / (NT / 64)), dim3(NT), 0, (hipStream_t)stream,
                       dpreds, dlines, lines, head, ys, dhead, dpriors, N, S, HW, img_w, img_h);
    return phnet_launch_status();
}

// dx = dy where y > 0 else 0 (ReLU backward through the saved output); dx may alias dy.
PHNET_API int phnet_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream)
{
    if (n < 0) return PHNET_ERR_ARG;
    if (n == 0) return PHNET_OK;
    if (!dy || !y || !dx) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)ceil_div64(n, NT)), dim3(NT), 0, (hipStream_t)stream, dy, y, dx, (long)n);
    return phnet_launch_status();
}

// y = res + dropout(x) (res may be NULL: plain dropout; drop_p = 0: plain residual add).  The mask of element i is the
// counter-based bit of common.h for site rng_call; calling again with x = dy, res = NULL gives the backward.
PHNET_API int phnet_dropout_add(const float* x, const float* res, float* y, int64_t n,
                                const uint64_t* rng_state, uint64_t rng_call, float drop_p, void* stream)
{
    if (n < 0 || drop_p < 0.f || drop_p >= 1.f) return PHNET_ERR_ARG;
    if (n == 0) return PHNET_OK;
    if (!x || !y) return PHNET_ERR_ARG;
    const DropRng rng = phnet_make_rng(rng_state, rng_call, drop_p);
    hipLaunchKernelGGL(dropout_add_kernel, dim3((unsigned)ceil_div64(n, NT)), dim3(NT), 0, (hipStream_t)stream, x, res, y, (long)n, rng,
                       rng.thresh ? 1.0f / (1.0f - drop_p) : 1.0f);
    return phnet_launch_status();
}

// y = dropout(gelu(x)) with the exact (erf) GELU of F.gelu; backward dx = dropout'(dy) * gelu'(x) with the same mask bits.
PHNET_API int phnet_gelu_dropout_fwd(const float* x, float* y, int64_t n, const uint64_t* rng_state, uint64_t rng_call, float drop_p,
                                     void* stream)
{
    if (n < 0 || drop_p < 0.f || drop_p >= 1.f) return PHNET_ERR_ARG;
    if (n == 0) return PHNET_OK;
    if (!x || !y) return PHNET_ERR_ARG;
    const DropRng rng = phnet_make_rng(rng_state, rng_call, drop_p);
    hipLaunchKernelGGL(gelu_dropout_fwd_kernel, dim3((unsigned)ceil_div64(n, NT)), dim3(NT), 0, (hipStream_t)stream, x, y, (long)n, rng,
                       rng.thresh ? 1.0f / (1.0f - drop_p) : 1.0f);
    return phnet_launch_status();
}

PHNET_API int phnet_gelu_dropout_bwd(const float* dy, const float* x, float* dx, int64_t n, const uint64_t* rng_state,
                                     uint64_t rng_call, float drop_p, void* stream)
{
    if (n < 0 || drop_p < 0.f || drop_p >= 1.f) return PHNET_ERR_ARG;
    if (n == 0) return PHNET_OK;
    if (!dy || !x || !dx) return PHNET_ERR_ARG;
    const DropRng rng = phnet_make_rng(rng_state, rng_call, drop_p);
    hipLaunchKernelGGL(gelu_dropout_bwd_kernel, dim3((unsigned)ceil_div64(n, NT)), dim3(NT), 0, (hipStream_t)stream, dy, x, dx, (long)n,
                       rng, rng.thresh ? 1.0f / (1.0f - drop_p) : 1.0f);
    return phnet_launch_status();
}

// Memory tokens of one frame and stage (Router4OL.py:563-584) for B clips.  feat [B][N][E] (E <= 1024), rows i64[B][L]
// (-1 padded), tokens [B][L+1][E], valid u8[B][L+1].
PHNET_API int phnet_memory_tokens(const float* feat, const int64_t* rows, float* tokens, uint8_t* valid,
                                  int32_t B, int32_t N, int32_t E, int32_t L, void* stream)
{
    if (B < 1 || N < 1 || E < 1 || E > 1024 || L < 0 || !feat || (L && !rows) || !tokens || !valid) return PHNET_ERR_ARG;
    const int groups = 1024 / E;
    hipLaunchKernelGGL(memory_tokens_kernel, dim3(B), dim3(groups * E), (size_t)groups * E * sizeof(float), (hipStream_t)stream,
                       feat, (const long long*)rows, tokens, valid, N, E, L);
    return phnet_launch_status();
}

// gate[n] = sigmoid(relu(h[n] . w + b)); h [N][K], w [K], b [1], out [N].
PHNET_API int phnet_gate_tail_fwd(const float* h, const float* w, const float* b, float* out, int32_t N, int32_t K, void* stream)
{
    if (N < 1 || K < 1 || !h || !w || !b || !out) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(gate_tail_fwd_kernel, dim3((N + NT / 64 - 1) / (NT / 64)), dim3(NT), 0, (hipStream_t)stream, h, w, b, out, N, K);
    return phnet_launch_status();
}

// Backward of phnet_gate_tail_fwd from its output: dh [N][K] (optional), dw [K], db [1] overwritten or accumulated.
PHNET_API int phnet_gate_tail_bwd(const float* dout, const float* out, const float* h, const float* w, float* dh, float* dw, float* db,
                                  int32_t N, int32_t K, int32_t accumulate, void* stream)
{
    if (N < 1 || K < 1 || !dout || !out || !h || !w || !dw || !db) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(gate_tail_bwd_kernel, dim3((K + 63) / 64), dim3(1024), 0, (hipStream_t)stream, dout, out, h, w, dh, dw, db,
                       N, K, accumulate);
    return phnet_launch_status();
}

// priors [N][W] = (1 - gate[n]) a + gate[n] b and on_map [N][P] = priors[:, 6 + idx[p]] in one launch.
PHNET_API int phnet_blend_priors(const float* gate, const float* a, const float* b, const int64_t* idx, float* priors, float* on_map,
                                 int32_t N, int32_t W, int32_t P, void* stream)
{
    if (N < 1 || W < 7 || P < 1 || !gate || !a || !b || !idx || !priors || !on_map) return PHNET_ERR_ARG;
    const int cols = W > P ? W : P;                     // P > W: the V2 family samples 96 positions out of 72 offsets (repeats)
    hipLaunchKernelGGL(blend_priors_kernel, dim3((N * cols + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, gate, a, b,
                       (const long long*)idx, priors, on_map, N, W, P);
    return phnet_launch_status();
}

// One AdamW step over flat fp32 arrays (n % 4 == 0, 16-byte aligned); elements [0, n_decay) get decoupled weight decay.
// step: DEVICE pointer to the 1-based step count as int64 (the caller increments it before the call - part of the captured
// step, so replays advance the bias correction).  lr_dev (optional DEVICE pointer to one float): when given, the learning
// rate is read from it and `lr` is ignored - the host updates that scalar between replays of a captured step.
PHNET_API int phnet_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_decay, const int64_t* step,
                               float lr, const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, void* stream)
{
    if (n < 0 || (n & 3) || n_decay < 0 || n_decay > n) return PHNET_ERR_ARG;
    if (n == 0) return PHNET_OK;
    if (!p || !g || !m || !v || !step) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)ceil_div64(n / 4, NT)), dim3(NT), 0, (hipStream_t)stream, p, g, m, v, (long)(n / 4),
                       (long)n_decay, (const long long*)step, lr, lr_dev, beta1, beta2, eps, weight_decay);
    return phnet_launch_status();
}
