// Depth-wise stack of the adaptive routing gate (libs/models/Router.py:39-81: LayerNorm([C,P]) + four residual blocks
//   relu( LN(dw3x3( relu(LN(dw3x3(x))) )) + x )  with per-anchor 3x3 filters) for the geometry the model runs, C = 64 channels
// x P = 36 points, with ONE WAVEFRONT PER PLANE: lane c owns channel row c, its 36 points live in registers.
//
// gate.hip gives a plane a 1024-thread workgroup: every LayerNorm is two block reductions, every depth-wise convolution a trip
// through LDS, ~42 barriers of 16 wavefronts per plane and direction - 35 us (forward) / 56 us (backward) of pure latency per
// plane, five rounds of planes per launch, and 12 saved intermediate planes per plane through HBM (140 + 195 MB per launch at
// 18 % of the HBM rate).  Here nothing crosses a wavefront: the LayerNorm sums are per-lane sums + one DPP wave reduction, the
// p +/- 1 taps of the filter are the lane's own registers, the c +/- 1 taps come from the neighbour lanes by DPP wave shifts
// (zero-filled at lanes 0 / 63 = the zero padding), the per-anchor filters are wave-uniform scalars.  No barrier, no LDS in
// the forward.  The backward RECOMPUTES a block from its input (the only planes the training forward saves: 4 instead of
// 12) and parks two planes in wave-private LDS; 8 planes per CU are in flight, a 1200-plane launch is a single round.
// LayerNorm statistics: two-pass (mean, then the variance of the centred values), as gate.hip.
#include "common.h"

namespace {

constexpr int GC = 64, GP = 36, GCP = GC * GP;
constexpr int NPARAM = 34;                  // ln0_w, ln0_b, then per block: c1_w c1_b ln1_w ln1_b c2_w c2_b ln2_w ln2_b
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GateParams { const float* p[NPARAM]; };
struct GateGrads { float* p[NPARAM]; };

__device__ __forceinline__ float lane_up(float v) {        // the value of lane - 1 (0 in lane 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_dn(float v) {        // the value of lane + 1 (0 in lane 63)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// the lane's row of a [C][P] plane: 36 consecutive floats
__device__ __forceinline__ void load_row(const float* __restrict__ src, float (&v)[GP]) {
#pragma unroll
    for (int k = 0; k < GP / 4; ++k) {
        const f32x4 t = reinterpret_cast<const f32x4*>(src)[k];
        v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
    }
}
__device__ __forceinline__ void store_row(float* __restrict__ dst, const float (&v)[GP]) {
#pragma unroll
    for (int k = 0; k < GP / 4; ++k) reinterpret_cast<f32x4*>(dst)[k] = (f32x4){v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
}

__device__ __forceinline__ void plane_stats(const float (&v)[GP], float eps, float& mu, float& rs) {
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < GP; ++p) s += v[p];
    mu = wave_sum(s) / (float)GCP;
    float q = 0.f;
#pragma unroll
    for (int p = 0; p < GP; ++p) { const float d = v[p] - mu; q += d * d; }
    rs = 1.0f / sqrtf(wave_sum(q) / (float)GCP + eps);
}

// out[c][p] = bias + sum f[di][dj] * in[c + di - 1][p + dj - 1] (zero padding); flip: the 180-degree rotated filter (conv backward)
__device__ __forceinline__ void dwconv(const float (&in)[GP], const float* __restrict__ f9, float bias, bool flip, float (&out)[GP]) {
    float f[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) f[t] = f9[flip ? 8 - t : t];
    float up_m = 0.f, up_0 = lane_up(in[0]), dn_m = 0.f, dn_0 = lane_dn(in[0]);     // (c -+ 1) rows at p - 1 and p
#pragma unroll
    for (int p = 0; p < GP; ++p) {
        const float up_p = p + 1 < GP ? lane_up(in[p + 1]) : 0.f, dn_p = p + 1 < GP ? lane_dn(in[p + 1]) : 0.f;
        const float c_m = p > 0 ? in[p - 1] : 0.f, c_p = p + 1 < GP ? in[p + 1] : 0.f;
        float acc = bias;
        acc += f[0] * up_m; acc += f[1] * up_0; acc += f[2] * up_p;
        acc += f[3] * c_m;  acc += f[4] * in[p]; acc += f[5] * c_p;
        acc += f[6] * dn_m; acc += f[7] * dn_0; acc += f[8] * dn_p;
        out[p] = acc;
        up_m = up_0; up_0 = up_p; dn_m = dn_0; dn_0 = dn_p;
    }
}

// x = relu?((x - mu) * rs * w + b (+ res))
__device__ __forceinline__ void ln_apply(float (&x)[GP], float mu, float rs, const float* __restrict__ w, const float* __restrict__ b,
                                         const float* res, bool relu) {
    float wv[GP], bv[GP];
    load_row(w, wv); load_row(b, bv);
#pragma unroll
    for (int p = 0; p < GP; ++p) {
        float y = (x[p] - mu) * rs * wv[p] + bv[p];
        if (res) y += res[p];
        x[p] = relu ? fmaxf(y, 0.f) : y;
    }
}

// blocks: 256 threads = 4 wavefronts = 4 planes.  sblk (training, may be NULL): [4][N][CP] the block inputs s_0..s_3.
__global__ __launch_bounds__(256) void gate_wave_fwd_kernel(const float* __restrict__ x, GateParams w, float* __restrict__ out,
                                                            float* __restrict__ sblk, int N, int A, float eps)
{
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;                                          // (wave-uniform; no barrier in this kernel)
    const int an = n % A;
    const size_t row = (size_t)n * GCP + lane * GP, slab = (size_t)N * GCP;
    float s[GP], a[GP];
    load_row(x + row, s);
    float mu, rs;
    plane_stats(s, eps, mu, rs);
    ln_apply(s, mu, rs, w.p[0] + lane * GP, w.p[1] + lane * GP, nullptr, false);
    for (int b = 0; b < 4; ++b) {
        const float* const* q = w.p + 2 + 8 * b;
        if (sblk) store_row(sblk + (size_t)b * slab + row, s);
        dwconv(s, q[0] + an * 9, q[1][an], false, a);
        plane_stats(a, eps, mu, rs);
        ln_apply(a, mu, rs, q[2] + lane * GP, q[3] + lane * GP, nullptr, true);
        float t[GP];
        dwconv(a, q[4] + an * 9, q[5][an], false, t);
        plane_stats(t, eps, mu, rs);
        ln_apply(t, mu, rs, q[6] + lane * GP, q[7] + lane * GP, s, true);
#pragma unroll
        for (int p = 0; p < GP; ++p) s[p] = t[p];
    }
    store_row(out + row, s);
}

// LayerNorm backward: g = upstream gradient of the LN output, xin = its input; writes the affine partials (g * xhat, g) of this plane
// and returns dx in g
__device__ __forceinline__ void ln_backward(float (&g)[GP], const float (&xin)[GP], float mu, float rs, const float* __restrict__ gamma,
                                            float* __restrict__ part_w, float* __restrict__ part_b) {
    float gm[GP], pw[GP];
    load_row(gamma, gm);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int p = 0; p < GP; ++p) {
        const float xh = (xin[p] - mu) * rs;
        pw[p] = g[p] * xh;
        const float gw = g[p] * gm[p];
        s1 += gw; s2 += gw * xh;
        gm[p] = gw;
    }
    store_row(part_w, pw);
    store_row(part_b, g);
    s1 = wave_sum(s1) / (float)GCP; s2 = wave_sum(s2) / (float)GCP;
#pragma unroll
    for (int p = 0; p < GP; ++p) g[p] = rs * (gm[p] - s1 - (xin[p] - mu) * rs * s2);
}

// dW[tap] = sum g[c][p] * src[c + di - 1][p + dj - 1], db = sum g: per-lane partials, nine + one wave sums; lane 0 writes
__device__ __forceinline__ void filter_grad(const float (&src)[GP], const float (&g)[GP], float* __restrict__ dw, float* __restrict__ db,
                                            int accumulate, int lane) {
    float acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = 0.f;
    float up_m = 0.f, up_0 = lane_up(src[0]), dn_m = 0.f, dn_0 = lane_dn(src[0]);
#pragma unroll
    for (int p = 0; p < GP; ++p) {
        const float up_p = p + 1 < GP ? lane_up(src[p + 1]) : 0.f, dn_p = p + 1 < GP ? lane_dn(src[p + 1]) : 0.f;
        const float c_m = p > 0 ? src[p - 1] : 0.f, c_p = p + 1 < GP ? src[p + 1] : 0.f;
        const float gv = g[p];
        acc[0] += gv * up_m; acc[1] += gv * up_0; acc[2] += gv * up_p;
        acc[3] += gv * c_m;  acc[4] += gv * src[p]; acc[5] += gv * c_p;
        acc[6] += gv * dn_m; acc[7] += gv * dn_0; acc[8] += gv * dn_p;
        acc[9] += gv;
        up_m = up_0; up_0 = up_p; dn_m = dn_0; dn_0 = dn_p;
    }
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = wave_sum(acc[t]);
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t) dw[t] = accumulate ? dw[t] + acc[t] : acc[t];
        *db = accumulate ? *db + acc[9] : acc[9];
    }
}

// 128 threads = 2 planes per workgroup, two wave-private LDS planes each (4 x 9 KB per workgroup).
// lnpart: [N][18][CP] per-plane partial gradients of the 9 LayerNorms' (weight, bias); fpart [N][8][10] filter partials (N > A)
__global__ __launch_bounds__(128, 2) void gate_wave_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ x,
                                                               const float* __restrict__ out, GateParams w,
                                                               const float* __restrict__ sblk, GateGrads dg, float* __restrict__ lnpart,
                                                               float* __restrict__ fpart, int N, int A, float eps, int accumulate)
{
    __shared__ float park[2][2][GCP];                            // [wave][slot]: the lane's rows, private to the wave
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = blockIdx.x * 2 + wv;
    if (n >= N) return;                                          // (wave-uniform; no barrier in this kernel)
    const int an = n % A;
    const size_t row = (size_t)n * GCP + lane * GP, slab = (size_t)N * GCP;
    float* lp = lnpart + (size_t)n * 18 * GCP + lane * GP;
    const bool direct_f = N == A;                                // this wave is the only one that touches its anchor's filter gradients
    float* fp = fpart + (size_t)n * 80;
    float* park_c1 = &park[wv][0][lane * GP];
    float* park_gr = &park[wv][1][lane * GP];
    float g[GP];
    load_row(gout + row, g);
    for (int b = 3; b >= 0; --b) {
        const float* const* q = w.p + 2 + 8 * b;
        float* const* dq = dg.p + 2 + 8 * b;
        // ---- recompute the block from its input s_b ----
        float u[GP], c2[GP];
        float mu1, rs1, mu2, rs2;
        {
            float s[GP], c1[GP];
            load_row(sblk + (size_t)b * slab + row, s);
            dwconv(s, q[0] + an * 9, q[1][an], false, c1);
            store_row(park_c1, c1);                              // LN1's input: needed again for its backward
            plane_stats(c1, eps, mu1, rs1);
#pragma unroll
            for (int p = 0; p < GP; ++p) u[p] = c1[p];
            ln_apply(u, mu1, rs1, q[2] + lane * GP, q[3] + lane * GP, nullptr, true);            // u = relu(LN1(c1))
            dwconv(u, q[4] + an * 9, q[5][an], false, c2);
            plane_stats(c2, eps, mu2, rs2);
            // through the block's output relu: the mask is the forward's own output (the next block's saved input, or `out`)
            float o[GP];
            load_row(b == 3 ? out + row : sblk + (size_t)(b + 1) * slab + row, o);
#pragma unroll
            for (int p = 0; p < GP; ++p) g[p] = o[p] > 0.f ? g[p] : 0.f;
        }
        store_row(park_gr, g);                                   // the residual path's gradient
        // ---- LN2 backward (input c2), conv2 backward (input u) ----
        ln_backward(g, c2, mu2, rs2, q[6] + lane * GP, lp + (size_t)(2 + 4 * b + 2) * GCP, lp + (size_t)(2 + 4 * b + 3) * GCP);
        if (direct_f) filter_grad(u, g, dq[4] + an * 9, dq[5] + an, accumulate, lane);
        else filter_grad(u, g, fp + (2 * b + 1) * 10, fp + (2 * b + 1) * 10 + 9, 0, lane);
        {
            float dv[GP];
            dwconv(g, q[4] + an * 9, 0.f, true, dv);
#pragma unroll
            for (int p = 0; p < GP; ++p) g[p] = u[p] > 0.f ? dv[p] : 0.f;                         // through the inner relu
        }
        // ---- LN1 backward (input c1, parked), conv1 backward (input s_b, re-read) ----
        {
            float c1[GP];
            load_row(park_c1, c1);
            ln_backward(g, c1, mu1, rs1, q[2] + lane * GP, lp + (size_t)(2 + 4 * b + 0) * GCP, lp + (size_t)(2 + 4 * b + 1) * GCP);
        }
        {
            float s[GP];
            load_row(sblk + (size_t)b * slab + row, s);
            if (direct_f) filter_grad(s, g, dq[0] + an * 9, dq[1] + an, accumulate, lane);
            else filter_grad(s, g, fp + (2 * b) * 10, fp + (2 * b) * 10 + 9, 0, lane);
        }
        {
            float dv[GP], gr[GP];
            dwconv(g, q[0] + an * 9, 0.f, true, dv);
            load_row(park_gr, gr);
#pragma unroll
            for (int p = 0; p < GP; ++p) g[p] = dv[p] + gr[p];                                   // + residual path
        }
    }
    // ---- pre-norm: only its affine gradients are needed (the gate input is detached, Router4OL.py:275) ----
    {
        float a[GP], pw[GP];
        load_row(x + row, a);
        float mu, rs;
        plane_stats(a, eps, mu, rs);
#pragma unroll
        for (int p = 0; p < GP; ++p) pw[p] = g[p] * ((a[p] - mu) * rs);
        store_row(lp, pw);
        store_row(lp + GCP, g);
    }
}

}  // namespace

// The wave-per-plane forms of phnet_gate_stack_fwd / _bwd (gate.hip documents arguments and layouts); C = 64, P = 36 only.
// saved (training): the first 4 planes-slabs [4][N][C*P] receive the block inputs s_0..s_3 - all the backward needs.
PHNET_API int phnet_gate_wave_applies(int32_t C, int32_t P) { return C == GC && P == GP; }

PHNET_API int phnet_gate_wave_fwd(const float* x, const float* const* params, float* out, float* saved,
                                  int32_t N, int32_t anchors, float eps, void* stream)
{
    if (N < 1 || anchors < 1 || N % anchors || !x || !params || !out) return PHNET_ERR_ARG;
    GateParams w;
    for (int i = 0; i < NPARAM; ++i) { w.p[i] = params[i]; if (!params[i]) return PHNET_ERR_ARG; }
    hipLaunchKernelGGL(gate_wave_fwd_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, out, saved, N, anchors, eps);
    return phnet_launch_status();
}

// workspace: [N][18][C*P] LayerNorm affine partials + [N][80] filter partials (phnet_gate_stack_bwd_workspace bytes); the
// caller folds them with the reduce kernels of gate.hip (phnet_gate_stack_reduce).
PHNET_API int phnet_gate_wave_bwd(const float* gout, const float* x, const float* out, const float* const* params, const float* saved,
                                  float* const* grads, int32_t N, int32_t anchors, float eps, int32_t accumulate,
                                  void* workspace, void* stream)
{
    if (N < 1 || anchors < 1 || N % anchors || !gout || !x || !out || !params || !saved || !grads || !workspace) return PHNET_ERR_ARG;
    GateParams w; GateGrads dg;
    for (int i = 0; i < NPARAM; ++i) { w.p[i] = params[i]; dg.p[i] = grads[i]; if (!params[i] || !grads[i]) return PHNET_ERR_ARG; }
    float* fpart = (float*)workspace + (size_t)18 * N * GCP;
    hipLaunchKernelGGL(gate_wave_bwd_kernel, dim3((N + 1) / 2), dim3(128), 0, (hipStream_t)stream,
                       gout, x, out, w, saved, dg, (float*)workspace, fpart, N, anchors, eps, accumulate);
    return phnet_launch_status();
}
