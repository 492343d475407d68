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
// 12), parks one plane per wave in LDS and folds the LayerNorm affine partials of a workgroup's four planes before they leave
// the CU; 8 planes per CU are in flight, a 1200-plane launch is a single round.
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
    float s[4] = {0.f, 0.f, 0.f, 0.f};                            // four chains: a lone wave issues dependent adds at their latency
#pragma unroll
    for (int p = 0; p < GP; ++p) s[p & 3] += v[p];
    mu = wave_sum((s[0] + s[1]) + (s[2] + s[3])) * (1.0f / (float)GCP);
    float q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < GP; ++p) { const float d = v[p] - mu; q[p & 3] += d * d; }
    rs = 1.0f / sqrtf(wave_sum((q[0] + q[1]) + (q[2] + q[3])) * (1.0f / (float)GCP) + eps);
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

// x = relu?((x - mu) * rs * w + b (+ res)) on rows the caller loaded EARLY (a lone wave per SIMD has nothing to hide a load behind
// except its own arithmetic: the affine rows are requested before the convolution whose output they scale)
template <bool RES, bool RELU>
__device__ __forceinline__ void ln_apply(float (&x)[GP], float mu, float rs, const float (&wv)[GP], const float (&bv)[GP],
                                         const float (&res)[GP]) {
#pragma unroll
    for (int p = 0; p < GP; ++p) {
        float y = (x[p] - mu) * rs * wv[p] + bv[p];
        if (RES) y += res[p];
        x[p] = RELU ? fmaxf(y, 0.f) : y;
    }
}

// blocks: 256 threads = 4 wavefronts = 4 planes.  sblk (training, may be NULL): [4][N][CP] the block inputs s_0..s_3.
__global__ __launch_bounds__(256) void gate_wave_fwd_kernel(const float* __restrict__ x, GateParams w, float* __restrict__ out,
                                                            float* __restrict__ sblk, int N, int A, float eps)
{
    const int lane = threadIdx.x & 63;
    const int n = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));      // wave-uniform: scalar addressing
    if (n >= N) return;                                          // (no barrier in this kernel)
    const int an = n % A;
    const size_t row = (size_t)n * GCP + lane * GP, slab = (size_t)N * GCP;
    float s[GP], a[GP], wv[GP], bv[GP];
    load_row(w.p[0] + lane * GP, wv); load_row(w.p[1] + lane * GP, bv);
    load_row(x + row, s);
    float mu, rs;
    plane_stats(s, eps, mu, rs);
    ln_apply<false, false>(s, mu, rs, wv, bv, s);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const float* const* q = w.p + 2 + 8 * b;
        if (sblk) store_row(sblk + (size_t)b * slab + row, s);
        load_row(q[2] + lane * GP, wv); load_row(q[3] + lane * GP, bv);                   // LN1's rows: in flight during conv1
        __builtin_amdgcn_sched_barrier(0);
        dwconv(s, q[0] + an * 9, q[1][an], false, a);
        plane_stats(a, eps, mu, rs);
        ln_apply<false, true>(a, mu, rs, wv, bv, s);
        load_row(q[6] + lane * GP, wv); load_row(q[7] + lane * GP, bv);                   // LN2's rows: in flight during conv2
        __builtin_amdgcn_sched_barrier(0);
        float t[GP];
        dwconv(a, q[4] + an * 9, q[5][an], false, t);
        plane_stats(t, eps, mu, rs);
        ln_apply<true, true>(t, mu, rs, wv, bv, s);
#pragma unroll
        for (int p = 0; p < GP; ++p) s[p] = t[p];
    }
    store_row(out + row, s);
}

constexpr int BW = 4;                        // planes (= wavefronts) per workgroup of the backward kernel

// The LayerNorm affine gradients are sums over ALL planes: every plane's (g * xhat, g) rows used to travel to HBM (18 rows of
// 9 KB per plane: 199 MB per launch written, and read again by the column reduce - more than everything else the backward
// moves).  The BW planes of a workgroup are folded first, in a fixed order (deterministic): each wave parks its row in LDS,
// wave k then adds the BW copies of elements [576 k, 576 k + 576) and writes that quarter of the workgroup's partial row.
__device__ __forceinline__ void fold_store(const float (&v)[GP], bool live, float* __restrict__ red, float* __restrict__ dst, int wave, int lane) {
    float z[GP];
#pragma unroll
    for (int p = 0; p < GP; ++p) z[p] = live ? v[p] : 0.f;        // planes past N contribute zeros
    store_row(red + wave * GCP + lane * GP, z);
    __syncthreads();
    const int base = wave * (GCP / BW) + lane * (GP / BW);        // 9 consecutive elements
#pragma unroll
    for (int i = 0; i < GP / BW; ++i) {
        float acc = red[base + i];
#pragma unroll
        for (int j = 1; j < BW; ++j) acc += red[j * GCP + base + i];
        dst[base + i] = acc;
    }
    __syncthreads();
}

// LayerNorm backward: g = upstream gradient of the LN output, xin = its input; folds the affine partials (g * xhat, g) of the
// workgroup's planes into part_w / part_b and returns dx in g
__device__ __forceinline__ void ln_backward(float (&g)[GP], const float (&xin)[GP], float mu, float rs, const float (&gamma)[GP],
                                            bool live, float* __restrict__ red, float* __restrict__ part_w, float* __restrict__ part_b,
                                            int wave, int lane) {
    float gm[GP];
    {
        float pw[GP];
#pragma unroll
        for (int p = 0; p < GP; ++p) pw[p] = g[p] * ((xin[p] - mu) * rs);
        fold_store(pw, live, red, part_w, wave, lane);
    }
    fold_store(g, live, red, part_b, wave, lane);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int p = 0; p < GP; ++p) {
        const float gw = g[p] * gamma[p];
        s1 += gw; s2 += gw * ((xin[p] - mu) * rs);
        gm[p] = gw;
    }
    s1 = wave_sum(s1) * (1.0f / (float)GCP); s2 = wave_sum(s2) * (1.0f / (float)GCP);
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

// 256 threads = 4 planes per workgroup; LDS: one wave-private parked plane each (the residual gradient) + the fold buffer.
// lnpart: [ceil(N/4)][18][CP] per-WORKGROUP partial gradients of the 9 LayerNorms' (weight, bias); fpart [N][8][10] filter partials (N > A)
__global__ __launch_bounds__(256, 2) void gate_wave_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ x,
                                                               const float* __restrict__ out, GateParams w,
                                                               const float* __restrict__ sblk, GateGrads dg, float* __restrict__ lnpart,
                                                               float* __restrict__ fpart, int N, int A, float eps, int accumulate)
{
    __shared__ float park[BW][GCP];                              // the lane's rows, private to the wave
    __shared__ float red[BW * GCP];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_raw = __builtin_amdgcn_readfirstlane(blockIdx.x * BW + (threadIdx.x >> 6));
    const bool live = n_raw < N;                                 // a wave past the last plane runs along (barriers) on plane N-1, writes nothing
    const int n = live ? n_raw : N - 1;
    const int an = n % A;
    const size_t row = (size_t)n * GCP + lane * GP, slab = (size_t)N * GCP;
    float* lp = lnpart + (size_t)blockIdx.x * 18 * GCP;          // the workgroup's folded partial rows
    const bool direct_f = N == A;                                // this wave is the only one that touches its anchor's filter gradients
    float* fp = fpart + (size_t)n * 80;
    float* park_gr = &park[wv][lane * GP];
    float g[GP];
    load_row(gout + row, g);
#pragma unroll
    for (int b = 3; b >= 0; --b) {
        const float* const* q = w.p + 2 + 8 * b;
        float* const* dq = dg.p + 2 + 8 * b;
        // ---- recompute the block from its input s_b ----
        float u[GP], c2[GP];
        float mu1, rs1, mu2, rs2;
        {
            float s[GP], c1[GP], wv1[GP], bv1[GP];
            load_row(sblk + (size_t)b * slab + row, s);
            load_row(q[2] + lane * GP, wv1); load_row(q[3] + lane * GP, bv1);                    // LN1's rows: in flight during conv1
            __builtin_amdgcn_sched_barrier(0);
            dwconv(s, q[0] + an * 9, q[1][an], false, c1);
            plane_stats(c1, eps, mu1, rs1);
#pragma unroll
            for (int p = 0; p < GP; ++p) u[p] = c1[p];
            ln_apply<false, true>(u, mu1, rs1, wv1, bv1, s);                                     // u = relu(LN1(c1))
        }
        float gam[GP];
        {
            float o[GP];
            // through the block's output relu: the mask is the forward's own output (the next block's saved input, or `out`)
            load_row(b == 3 ? out + row : sblk + (size_t)(b + 1) * slab + row, o);
            load_row(q[6] + lane * GP, gam);                                                     // LN2's weight: in flight during conv2
            __builtin_amdgcn_sched_barrier(0);
            dwconv(u, q[4] + an * 9, q[5][an], false, c2);
            plane_stats(c2, eps, mu2, rs2);
#pragma unroll
            for (int p = 0; p < GP; ++p) g[p] = o[p] > 0.f ? g[p] : 0.f;
        }
        store_row(park_gr, g);                                   // the residual path's gradient
        // ---- LN2 backward (input c2), conv2 backward (input u) ----
        ln_backward(g, c2, mu2, rs2, gam, live, red, lp + (size_t)(2 + 4 * b + 2) * GCP, lp + (size_t)(2 + 4 * b + 3) * GCP, wv, lane);
        if (live) {
            if (direct_f) filter_grad(u, g, dq[4] + an * 9, dq[5] + an, accumulate, lane);
            else filter_grad(u, g, fp + (2 * b + 1) * 10, fp + (2 * b + 1) * 10 + 9, 0, lane);
        }
        {
            float dv[GP];
            dwconv(g, q[4] + an * 9, 0.f, true, dv);
#pragma unroll
            for (int p = 0; p < GP; ++p) g[p] = u[p] > 0.f ? dv[p] : 0.f;                         // through the inner relu
        }
        // ---- LN1 backward (input c1: conv1 of the re-read s_b once more - arithmetic is cheap, a parked plane is not),
        //      conv1 backward (input s_b) ----
        {
            float s[GP], c1[GP];
            load_row(sblk + (size_t)b * slab + row, s);
            load_row(q[2] + lane * GP, gam);
            __builtin_amdgcn_sched_barrier(0);
            dwconv(s, q[0] + an * 9, q[1][an], false, c1);
            ln_backward(g, c1, mu1, rs1, gam, live, red, lp + (size_t)(2 + 4 * b + 0) * GCP, lp + (size_t)(2 + 4 * b + 1) * GCP, wv, lane);
            if (live) {
                if (direct_f) filter_grad(s, g, dq[0] + an * 9, dq[1] + an, accumulate, lane);
                else filter_grad(s, g, fp + (2 * b) * 10, fp + (2 * b) * 10 + 9, 0, lane);
            }
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
        fold_store(pw, live, red, lp, wv, lane);
        fold_store(g, live, red, lp + GCP, wv, lane);
    }
}

}  // namespace

// The wave-per-plane forms of phnet_gate_stack_fwd / _bwd (gate.hip documents arguments and layouts); C = 64, P = 36 only.
// saved (training): the first 4 planes-slabs [4][N][C*P] receive the block inputs s_0..s_3 - all the backward needs.
PHNET_API int phnet_gate_wave_applies(int32_t C, int32_t P) { return C == GC && P == GP; }
PHNET_API int phnet_gate_wave_partial_planes(int32_t N) { return (N + BW - 1) / BW; }

PHNET_API int phnet_gate_wave_fwd(const float* x, const float* const* params, float* out, float* saved,
                                  int32_t N, int32_t anchors, float eps, void* stream)
{
    if (N < 1 || anchors < 1 || N % anchors || !x || !params || !out) return PHNET_ERR_ARG;
    GateParams w;
    for (int i = 0; i < NPARAM; ++i) { w.p[i] = params[i]; if (!params[i]) return PHNET_ERR_ARG; }
    hipLaunchKernelGGL(gate_wave_fwd_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, out, saved, N, anchors, eps);
    return phnet_launch_status();
}

// workspace (phnet_gate_stack_bwd_workspace bytes): [ceil(N/4)][18][C*P] LayerNorm affine partials, folded per workgroup of four
// planes, then - at the SAME offset as the generic kernel's, 18*N*C*P floats in - [N][80] filter partials; phnet_gate_stack_bwd
// folds them with the reduce kernels of gate.hip (phnet_gate_wave_partial_planes = how many partial planes the column reduce reads).
PHNET_API int phnet_gate_wave_bwd(const float* gout, const float* x, const float* out, const float* const* params, const float* saved,
                                  float* const* grads, int32_t N, int32_t anchors, float eps, int32_t accumulate,
                                  void* workspace, void* stream)
{
    if (N < 1 || anchors < 1 || N % anchors || !gout || !x || !out || !params || !saved || !grads || !workspace) return PHNET_ERR_ARG;
    GateParams w; GateGrads dg;
    for (int i = 0; i < NPARAM; ++i) { w.p[i] = params[i]; dg.p[i] = grads[i]; if (!params[i] || !grads[i]) return PHNET_ERR_ARG; }
    float* fpart = (float*)workspace + (size_t)18 * N * GCP;
    hipLaunchKernelGGL(gate_wave_bwd_kernel, dim3((N + BW - 1) / BW), dim3(64 * BW), 0, (hipStream_t)stream,
                       gout, x, out, w, saved, dg, (float*)workspace, fpart, N, anchors, eps, accumulate);
    return phnet_launch_status();
}
