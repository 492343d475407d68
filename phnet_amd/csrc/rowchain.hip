// Row-local chain of a pre-norm transformer layer in ONE launch (gfx950), forward only.
//
// The decoder of branch B (libs/models/utils/transformer.py:275-298) is, between its two attention cores, a chain of ROW-LOCAL
// operations on 240 x 128 activations: projection, residual + dropout, LayerNorm, feed-forward with GELU, the next projection.
// As separate launches (GEMM, fused dropout-add-LayerNorm, GELU, GEMM ...) every link costs a dependent launch of 5-7 us for
// microseconds of work, 13 per layer.  Here a workgroup owns 16 rows and walks the whole chain with its rows resident in LDS:
//
//   v  = in @ Wa^T + ba                      (optional: attention out-projection)
//   t  = resid + dropout_A(v)                (t = in when there is no stage A)
//   h  = LayerNorm_1(t)
//   f  = dropout_F(gelu(h @ W1^T + b1));  t = t + dropout_3(f @ W2^T + b2);  h = LayerNorm_2(t)        (optional: feed-forward block)
//   y  = h @ Wg^T + bg                       (optional: the NEXT projection - q of the cross-attention, or q|k|v of the next layer)
//
// so a layer is 3 such launches + its two attention cores instead of 13.  Used where no autograd graph is needed: inference, and
// the forward passes of branch B in training (their backward is recomputed as one batch through the unfused kernels,
// libs/models/Router4OL.py::_BranchBDeferred) - the dropout bits come from the same counter-based generator with the same site
// ids / element indices as the unfused kernels (csrc/common.h), so that recomputation sees the masks drawn here.
//
// GEMMs: rows are few (16 per workgroup), weights stream from L2: v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate), each of the
// 4 waves takes 16-column chunks of the output; lane (c = lane % 16, q = lane / 16) holds W[col c][16j + 4q .. +3] as one float4 per
// 16 k's and reads the matching float4 of its activation row from LDS - the k order inside a 16-group is permuted identically for
// both operands, which a dot product does not see.
#include "common.h"

namespace {

constexpr int NT = 512, ROWS = 16;          // 8 waves: each takes every 8th 16-column chunk of a GEMM
typedef float f4 __attribute__((ext_vector_type(4)));

struct RowChain {
    const float *in, *resid;                       // [R][E]
    const float *Wa, *ba;                          // [E][E], [E] or null
    const float *ln1w, *ln1b;                      // [E]
    const float *W1, *b1, *W2, *b2;                // [FF][E], [FF], [E][FF], [E] or null
    const float *ln2w, *ln2b;                      // [E] (with the feed-forward block)
    const float *Wg, *bg;                          // [NG][E], [NG] or null
    float *t_out, *h_out, *y_out;                  // [R][E], [R][E], [R][NG]; each may be null
    int R, NG;
    float eps;
    DropRng rngA, rngF, rng3;
    float scale;                                   // 1 / (1 - p) of the three dropout sites (same p)
};

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

// out(row, col, value) for value = sum_k xs[row][k] * W[col][k] (+ bias[col]), rows 0..15, cols 0..N-1.
// The weights of chunk i+1 are loaded (into the other register set) before the MFMAs of chunk i are issued: with 15 workgroups on
// the chip nothing else hides the L2 latency of a chunk's loads (12.7 -> 11.2 us median per launch).
// `first` (optional): the weights of this wave's first chunk, loaded by the caller ahead of time (rowchain_kernel<.., PRE = true>
// issues the first chunk of EVERY stage at its very start).  Measured: no gain (12.2 vs 11.2 us median) - the chain is bound by the
// f32 MFMA rate of the 15 CUs it runs on (16 rows per workgroup: 256 dependent 16x16x4 MFMAs per wave in the feed-forward form),
// not by the weight loads; PRE stays off.
template <int K>
__device__ __forceinline__ void load_chunk(f4 (&b)[K / 16], const float* __restrict__ W, int N, int c0)
{
    const int lane = threadIdx.x & 63, n = c0 + (lane & 15);
    const float* wr = W + (size_t)(n < N ? n : 0) * K + 4 * (lane >> 4);
#pragma unroll
    for (int j = 0; j < K / 16; ++j) b[j] = *reinterpret_cast<const f4*>(wr + 16 * j);
}

template <int K, typename OUT>
__device__ __forceinline__ void gemm16(const float* xs, int XP, const float* __restrict__ W, const float* __restrict__ bias, int N, OUT&& out,
                                       const f4 (*first)[K / 16] = nullptr)
{
    constexpr int KB = K / 16, STEP = (NT / 64) * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const float* xr = xs + c * XP + 4 * q;                           // A operand: row = lane % 16
    auto load = [&](f4 (&b)[KB], int c0) {
        const int n = c0 + c;
        const float* wr = W + (size_t)(n < N ? n : 0) * K + 4 * q;
#pragma unroll
        for (int j = 0; j < KB; ++j) b[j] = *reinterpret_cast<const f4*>(wr + 16 * j);
    };
    auto compute = [&](const f4 (&b)[KB], int c0) {
        const int n = c0 + c;
        const bool live = n < N;
        const float bv = (bias && live) ? bias[n] : 0.f;
        f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const f4 a = *reinterpret_cast<const f4*>(xr + 16 * j);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[j].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[j].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[j].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[j].w, acc, 0, 0, 0);
        }
        if (live) {                                                  // D[row = 4q + i][col = c]
            out(4 * q + 0, n, acc.x + bv); out(4 * q + 1, n, acc.y + bv);
            out(4 * q + 2, n, acc.z + bv); out(4 * q + 3, n, acc.w + bv);
        }
    };
    if constexpr (KB > 16) {                                         // K = 512: two register sets would spill - one set, no prefetch
        f4 b[KB];
        for (int c0 = wave * 16; c0 < N; c0 += STEP) { load(b, c0); compute(b, c0); }
        return;
    }
    f4 b0[KB], b1[KB];
    int c0 = wave * 16;
    if (first) {
#pragma unroll
        for (int j = 0; j < KB; ++j) b0[j] = (*first)[j];
    } else if (c0 < N) load(b0, c0);
    while (c0 < N) {
        const int c1 = c0 + STEP;
        if (c1 < N) load(b1, c1);
        compute(b0, c0);
        if (c1 >= N) break;
        const int c2 = c1 + STEP;
        if (c2 < N) load(b0, c2);
        compute(b1, c1);
        c0 = c2;
    }
}

// h[r][:] = LayerNorm(t[r][:]) * w + b for the 16 rows (4 per wave); E <= 256
template <int E>
__device__ __forceinline__ void layernorm16(const float* ts, int TP, const float* __restrict__ w, const float* __restrict__ b, float eps,
                                            float* hs, int HP, float* __restrict__ h_glob, long row0, int R)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < ROWS; r += NT / 64) {
        float v[E / 64], s = 0.f;
#pragma unroll
        for (int u = 0; u < E / 64; ++u) { v[u] = ts[r * TP + lane + 64 * u]; s += v[u]; }
        const float mu = wave_sum(s) / (float)E;
        float d2 = 0.f;
#pragma unroll
        for (int u = 0; u < E / 64; ++u) { const float d = v[u] - mu; d2 += d * d; }
        const float rs = 1.0f / sqrtf(wave_sum(d2) / (float)E + eps);
#pragma unroll
        for (int u = 0; u < E / 64; ++u) {
            const int i = lane + 64 * u;
            const float h = (v[u] - mu) * rs * w[i] + b[i];
            hs[r * HP + i] = h;
            if (h_glob && row0 + r < R) h_glob[(size_t)(row0 + r) * E + i] = h;
        }
    }
}

template <bool PRE, int KB, int M>
__device__ __forceinline__ const f4 (*first_of(const f4 (&a)[M]))[KB]
{
    if constexpr (PRE) return &a; else return nullptr;
}

template <int E, int FF, bool PRE>
__global__ __launch_bounds__(NT) void rowchain_kernel(RowChain p)
{
    constexpr int EP = E + 4, FP = FF + 4;
    extern __shared__ __attribute__((aligned(16))) float rc_lds[];
    float* xs = rc_lds;                      // [ROWS][EP] stage input, later the normalised rows h
    float* ts = xs + ROWS * EP;              // [ROWS][EP] residual stream t
    float* fs = ts + ROWS * EP;              // [ROWS][FP] feed-forward hidden
    const long row0 = (long)blockIdx.x * ROWS;
    const int R = p.R;
    const bool ffn = p.W1 != nullptr;
    // ---- first weight chunk of every stage, before anything else (PRE) ----
    f4 pa[PRE ? E / 16 : 1], p1[PRE ? E / 16 : 1], p2[PRE ? FF / 16 : 1], pg[PRE ? E / 16 : 1];
    if constexpr (PRE) {
        const int c0 = (threadIdx.x >> 6) * 16;
        if (p.Wa) load_chunk<E>(pa, p.Wa, E, c0);
        if (ffn) { load_chunk<E>(p1, p.W1, FF, c0); load_chunk<FF>(p2, p.W2, E, c0); }
        if (p.Wg) load_chunk<E>(pg, p.Wg, p.NG, c0);
    }
    // ---- the 16 input rows (rows past R: zeros) ----
    for (int i = threadIdx.x; i < ROWS * (E / 4); i += NT) {
        const int r = i / (E / 4), c4 = (i - r * (E / 4)) * 4;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (row0 + r < R) v = *reinterpret_cast<const f4*>(p.in + (size_t)(row0 + r) * E + c4);
        *reinterpret_cast<f4*>(xs + r * EP + c4) = v;
    }
    __syncthreads();
    // ---- t = resid + dropout(in @ Wa^T + ba)   |   t = in ----
    if (p.Wa) {
        const uint64_t seed = phnet_rng_seed(p.rngA);
        gemm16<E>(xs, EP, p.Wa, p.ba, E, [&](int r, int n, float v) {
            const long row = row0 + r;
            float t = 0.f;
            if (row < R) {
                const bool kept = !p.rngA.thresh || phnet_rng_keep(seed, phnet_rng_index(p.rngA, (uint64_t)row * E + n), p.rngA.thresh);
                t = p.resid[(size_t)row * E + n] + (kept ? v * p.scale : 0.f);
            }
            ts[r * EP + n] = t;
        }, first_of<PRE, E / 16>(pa));
    } else {
        for (int i = threadIdx.x; i < ROWS * E; i += NT) { const int r = i / E, c = i - r * E; ts[r * EP + c] = xs[r * EP + c]; }
    }
    __syncthreads();
    // ---- h = LayerNorm_1(t) (into xs) ----
    layernorm16<E>(ts, EP, p.ln1w, p.ln1b, p.eps, xs, EP, ffn ? nullptr : p.h_out, row0, R);
    __syncthreads();
    if (ffn) {
        {   // f = dropout(gelu(h @ W1^T + b1))
            const uint64_t seed = phnet_rng_seed(p.rngF);
            gemm16<E>(xs, EP, p.W1, p.b1, FF, [&](int r, int n, float v) {
                const long row = row0 + r;
                const bool kept = !p.rngF.thresh || phnet_rng_keep(seed, phnet_rng_index(p.rngF, (uint64_t)row * FF + n), p.rngF.thresh);
                fs[r * FP + n] = (row < R && kept) ? gelu_erf(v) * p.scale : 0.f;
            }, first_of<PRE, E / 16>(p1));
        }
        __syncthreads();
        {   // t = t + dropout(f @ W2^T + b2)
            const uint64_t seed = phnet_rng_seed(p.rng3);
            gemm16<FF>(fs, FP, p.W2, p.b2, E, [&](int r, int n, float v) {
                const long row = row0 + r;
                const bool kept = !p.rng3.thresh || phnet_rng_keep(seed, phnet_rng_index(p.rng3, (uint64_t)row * E + n), p.rng3.thresh);
                ts[r * EP + n] += (row < R && kept) ? v * p.scale : 0.f;
            }, first_of<PRE, FF / 16>(p2));
        }
        __syncthreads();
        layernorm16<E>(ts, EP, p.ln2w, p.ln2b, p.eps, xs, EP, p.h_out, row0, R);
        __syncthreads();
    }
    if (p.t_out)
        for (int i = threadIdx.x; i < ROWS * (E / 4); i += NT) {
            const int r = i / (E / 4), c4 = (i - r * (E / 4)) * 4;
            if (row0 + r < R) *reinterpret_cast<f4*>(p.t_out + (size_t)(row0 + r) * E + c4) = *reinterpret_cast<const f4*>(ts + r * EP + c4);
        }
    // ---- y = h @ Wg^T + bg ----
    if (p.Wg)
        gemm16<E>(xs, EP, p.Wg, p.bg, p.NG, [&](int r, int n, float v) {
            if (row0 + r < R) p.y_out[(size_t)(row0 + r) * p.NG + n] = v;
        }, first_of<PRE, E / 16>(pg));
}

// ---- the towers of a branch + the lane prior update in one launch (forward only) ----------------------------------------------
// Per tower t (cls / reg / offsets): h1 = relu(x @ W1_t^T + b1_t); h2 = relu(h1 @ W2_t^T + b2_t); head_t = h2 @ Wh_t^T + bh_t;
// then the prior update of Router4OL.py:328-345 on (cls 2 | reg 4 | offsets S) - the arithmetic of lane_update_fwd_kernel
// (elementwise.hip).  The per-tower parameters are used where they are: no concatenated / block-diagonal copies.
struct TowerChain {
    const float* p[18];                            // per tower: W1 [C][C], b1, W2 [C][C], b2, Wh [o][C], bh
    int out[3];
    const float *x, *priors, *ys;                  // [R][C], [R][6+S], [S]
    float *preds, *lines;                          // [R][6+S]
    int R, T, S;
    float img_w, img_h;
};

template <int C>
__global__ __launch_bounds__(NT) void tower_chain_kernel(TowerChain p)
{
    constexpr int CP = C + 4, HP = 3 * C + 4, DP = 96;
    extern __shared__ __attribute__((aligned(16))) float rc_lds[];
    float* xs = rc_lds;                       // [ROWS][CP]
    float* h1 = xs + ROWS * CP;               // [ROWS][HP]
    float* h2 = h1 + ROWS * HP;               // [ROWS][HP]
    float* hd = h2 + ROWS * HP;               // [ROWS][DP]: cls 2 | reg 4 | offsets S
    const long row0 = (long)blockIdx.x * ROWS;
    const int R = p.R, S = p.S, W = 6 + S;
    for (int i = threadIdx.x; i < ROWS * (C / 4); i += NT) {
        const int r = i / (C / 4), c4 = (i - r * (C / 4)) * 4;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (row0 + r < R) v = *reinterpret_cast<const f4*>(p.x + (size_t)(row0 + r) * C + c4);
        *reinterpret_cast<f4*>(xs + r * CP + c4) = v;
    }
    __syncthreads();
    for (int t = 0; t < p.T; ++t)
        gemm16<C>(xs, CP, p.p[6 * t], p.p[6 * t + 1], C, [&](int r, int n, float v) { h1[r * HP + t * C + n] = fmaxf(v, 0.f); });
    __syncthreads();
    for (int t = 0; t < p.T; ++t)
        gemm16<C>(h1 + t * C, HP, p.p[6 * t + 2], p.p[6 * t + 3], C, [&](int r, int n, float v) { h2[r * HP + t * C + n] = fmaxf(v, 0.f); });
    __syncthreads();
    int off = 0;
    for (int t = 0; t < p.T; ++t) {
        gemm16<C>(h2 + t * C, HP, p.p[6 * t + 4], p.p[6 * t + 5], p.out[t], [&](int r, int n, float v) { hd[r * DP + off + n] = v; });
        off += p.out[t];
    }
    __syncthreads();
    {
#pragma clang fp contract(off)
        const int lane = threadIdx.x & 63;
        for (int r = threadIdx.x >> 6; r < ROWS; r += NT / 64) {
            const long row = row0 + r;
            if (row >= R) continue;
            const float* pr = p.priors + (size_t)row * W;
            const float* h = hd + r * DP;
            float* po = p.preds + (size_t)row * W;
            float* lo = p.lines + (size_t)row * W;
            const float sy = pr[2] + tanhf(h[2]), sx = pr[3] + tanhf(h[3]), th = pr[4] + tanhf(h[4]);
            const float tn = tanf(th * 3.14159265358979323846f + 1e-5f);
            if (lane == 0) {
                po[0] = lo[0] = h[0]; po[1] = lo[1] = h[1];
                po[2] = lo[2] = sy; po[3] = lo[3] = sx; po[4] = lo[4] = th; po[5] = lo[5] = h[5];
            }
            for (int k = lane; k < S; k += 64) {
                const float x = (sx * (p.img_w - 1.0f) + ((1.0f - p.ys[k] - sy) * p.img_h / tn)) / (p.img_w - 1.0f);
                lo[6 + k] = x;
                po[6 + k] = x + h[6 + k];
            }
        }
    }
}

}  // namespace

// One launch for the row-local chain described at the top of this file.  All matrices row-major with the reduction dimension
// contiguous (nn.Linear layout [out][in]); E = 128 with FF = 256 (Router4OL.py:97-99) or E = 256 with FF = 512 (Router4OLV2.py:98-101).
// in [R][E]; Wa / ba: optional out-projection, then t = resid + dropout_A(...), else t = in; h = LayerNorm(t; ln1w, ln1b);
// W1 / b1 / W2 / b2 (optional, all four): feed-forward block with dropout_F on the GELU and dropout_3 on its output, then
// h = LayerNorm(t; ln2w, ln2b); Wg / bg [NG][E] (optional): y = h @ Wg^T + bg.  t_out / h_out / y_out: optional outputs.
// Dropout: drop_p and the three `rng_call` values follow the convention of phnet_dropout_add (0 / NULL state = no dropout).
PHNET_API int phnet_rowchain_fwd(const float* in, const float* resid, const float* Wa, const float* ba, const float* ln1w, const float* ln1b,
                                 const float* W1, const float* b1, const float* W2, const float* b2, const float* ln2w, const float* ln2b,
                                 const float* Wg, const float* bg, float* t_out, float* h_out, float* y_out,
                                 int32_t R, int32_t E, int32_t FF, int32_t NG, float eps,
                                 const uint64_t* rng_state, uint64_t call_a, uint64_t call_f, uint64_t call_3, float drop_p, void* stream)
{
    if (R < 0 || NG < 0 || drop_p < 0.f || drop_p >= 1.f) return PHNET_ERR_ARG;
    if (R == 0) return PHNET_OK;
    if (!in || !ln1w || !ln1b || (Wa && (!ba || !resid)) || (Wg && (!bg || !y_out || NG < 1))) return PHNET_ERR_ARG;
    const bool ffn = W1 || b1 || W2 || b2;
    if (ffn && (!W1 || !b1 || !W2 || !b2 || !ln2w || !ln2b)) return PHNET_ERR_ARG;
    RowChain p{in, resid, Wa, ba, ln1w, ln1b, W1, b1, W2, b2, ln2w, ln2b, Wg, bg, t_out, h_out, y_out, R, NG, eps,
               phnet_make_rng(rng_state, call_a, drop_p), phnet_make_rng(rng_state, call_f, drop_p), phnet_make_rng(rng_state, call_3, drop_p),
               (rng_state && drop_p > 0.f) ? 1.0f / (1.0f - drop_p) : 1.0f};
    const dim3 grid((unsigned)ceil_div64(R, ROWS));
    if (E == 128 && (!ffn || FF == 256)) {
        const size_t lds = (size_t)ROWS * (2 * (128 + 4) + (256 + 4)) * sizeof(float);
        hipLaunchKernelGGL((rowchain_kernel<128, 256, false>), grid, dim3(NT), lds, (hipStream_t)stream, p);
    } else if (E == 256 && (!ffn || FF == 512)) {
        const size_t lds = (size_t)ROWS * (2 * (256 + 4) + (512 + 4)) * sizeof(float);          // 66 KB
        static bool attr = false;
        if (!attr) {
            if (hipFuncSetAttribute((const void*)rowchain_kernel<256, 512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return PHNET_ERR_LAUNCH;
            attr = true;
        }
        hipLaunchKernelGGL((rowchain_kernel<256, 512, false>), grid, dim3(NT), lds, (hipStream_t)stream, p);
    } else return PHNET_ERR_ARG;
    return phnet_launch_status();
}

// Towers + heads + lane prior update of a branch in one launch, forward only (libs/models/Router4OL.py:308-345).
// params: HOST array of 6*T device pointers (per tower: layer-1 weight [C][C], bias, layer-2 weight, bias, head weight [o_t][C],
// head bias); the head outputs concatenated must be (cls 2 | reg 4 | offsets S), i.e. sum o_t = 6 + S <= 96.  x [R][C], C = 64 or
// 128; priors [R][6+S]; ys [S]; preds / lines [R][6+S].
PHNET_API int phnet_tower_chain_fwd(const float* x, const float* const* params, int32_t T, int32_t C, const int32_t* head_out,
                                    const float* priors, const float* ys, float* preds, float* lines, int32_t R, int32_t S,
                                    float img_w, float img_h, void* stream)
{
    if (R < 0 || T < 1 || T > 3 || S < 1 || !head_out || !params) return PHNET_ERR_ARG;
    if (R == 0) return PHNET_OK;
    if (!x || !priors || !ys || !preds || !lines) return PHNET_ERR_ARG;
    TowerChain p{};
    int sum = 0;
    for (int t = 0; t < T; ++t) { if (head_out[t] < 1) return PHNET_ERR_ARG; p.out[t] = head_out[t]; sum += head_out[t]; }
    if (sum != 6 + S || sum > 96) return PHNET_ERR_ARG;
    for (int i = 0; i < 6 * T; ++i) { if (!params[i]) return PHNET_ERR_ARG; p.p[i] = params[i]; }
    p.x = x; p.priors = priors; p.ys = ys; p.preds = preds; p.lines = lines; p.R = R; p.T = T; p.S = S; p.img_w = img_w; p.img_h = img_h;
    const dim3 grid((unsigned)ceil_div64(R, ROWS));
    if (C == 64) {
        const size_t lds = (size_t)ROWS * ((64 + 4) + 2 * (3 * 64 + 4) + 96) * sizeof(float);
        hipLaunchKernelGGL((tower_chain_kernel<64>), grid, dim3(NT), lds, (hipStream_t)stream, p);
    } else if (C == 128) {
        const size_t lds = (size_t)ROWS * ((128 + 4) + 2 * (3 * 128 + 4) + 96) * sizeof(float);
        hipLaunchKernelGGL((tower_chain_kernel<128>), grid, dim3(NT), lds, (hipStream_t)stream, p);
    } else return PHNET_ERR_ARG;
    return phnet_launch_status();
}
