// Fused depth-wise stack of the adaptive routing gate for gfx950: LayerNorm([C,P]) followed by four residual blocks
//   relu( LN(dw3x3( relu(LN(dw3x3(x))) )) + x )
// as ONE forward launch and ONE backward launch (+ one reduce launch for the shared LayerNorm affine gradients).
// The ATen formulation is 9 LayerNorm + 8 grouped-conv launches forward and ~50 backward per call, 15 calls per clip.
//
// Replaces libs/models/Router.py:72-75 (pre_norm + DWNets loop) forward and autograd backward; the Linear->ReLU->
// Linear->ReLU->sigmoid tail (Router.py:76-80) stays on the GEMM kernels.
// One workgroup per anchor: its (C x P) plane (2304 floats) lives in registers (9 per thread) and two LDS planes
// (the depth-wise filter reads neighbours from LDS); every anchor has its own 3x3 filters (groups = N), the LayerNorm
// affine parameters are shared by all anchors (-> per-anchor partial gradients + column reduce, deterministic).
#include "common.h"

namespace {

constexpr int NT = 1024;                    // 16 wavefronts per anchor: the chain is latency-bound, more waves hide it
constexpr int NW = NT / 64;
constexpr int EPT = 3;                      // elements per thread: planes up to 3072 floats
constexpr int NPARAM = 34;                  // ln0_w, ln0_b, then per block: c1_w c1_b ln1_w ln1_b c2_w c2_b ln2_w ln2_b

struct GateParams { const float* p[NPARAM]; };
struct GateGrads { float* p[NPARAM]; };

__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
    a = wave_sum(a); b = wave_sum(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[NW + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) { sa += red[k]; sb += red[NW + k]; }
    a = sa; b = sb;
}

// LayerNorm statistics of the per-thread values v[] (elements idx < CP)
__device__ __forceinline__ void plane_stats(const float (&v)[EPT], int CP, float eps, float& mu, float& rs, float* red) {
    float s = 0.f, dummy = 0.f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) s += (threadIdx.x + k * NT < CP) ? v[k] : 0.f;
    block_sum2(s, dummy, red);
    mu = s / (float)CP;
    float q = 0.f; dummy = 0.f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) { const float d = (threadIdx.x + k * NT < CP) ? v[k] - mu : 0.f; q += d * d; }
    block_sum2(q, dummy, red);
    rs = 1.0f / sqrtf(q / (float)CP + eps);
}

// out[k] = bias + sum_taps f[tap] * plane(c+di-1, p+dj-1)   (zero padding); flip -> 180-degree rotated filter
__device__ __forceinline__ void load_filter(const float* f9, bool flip, float (&f)[9]) {
#pragma unroll
    for (int t = 0; t < 9; ++t) f[t] = f9[flip ? 8 - t : t];
}
__device__ __forceinline__ void dwconv_apply(const float* plane, const float (&f)[9], float bias, int C, int P, int CP, float (&out)[EPT]);
__device__ __forceinline__ void dwconv_plane(const float* plane, const float* f9, float bias, int C, int P, int CP, bool flip,
                                             float (&out)[EPT]) {
    float f[9];
    load_filter(f9, flip, f);
    dwconv_apply(plane, f, bias, C, P, CP, out);
}
__device__ __forceinline__ void dwconv_apply(const float* plane, const float (&f)[9], float bias, int C, int P, int CP, float (&out)[EPT]) {
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int i = threadIdx.x + k * NT;
        float acc = bias;
        if (i < CP) {
            const int c = i / P, p = i - c * P;
#pragma unroll
            for (int di = 0; di < 3; ++di) {
                const int cc = c + di - 1;
                if (cc < 0 || cc >= C) continue;
#pragma unroll
                for (int dj = 0; dj < 3; ++dj) {
                    const int pp = p + dj - 1;
                    if (pp < 0 || pp >= P) continue;
                    acc += f[di * 3 + dj] * plane[cc * P + pp];
                }
            }
        }
        out[k] = acc;
    }
}

__device__ __forceinline__ void load_plane(const float* src, int CP, float (&v)[EPT]) {
#pragma unroll
    for (int k = 0; k < EPT; ++k) {                   // branch-free: the loads of several planes stay in flight together
        const int i = threadIdx.x + k * NT;
        const float t = src[i < CP ? i : 0];
        v[k] = i < CP ? t : 0.f;
    }
}
__device__ __forceinline__ void store_plane(float* dst, int CP, const float (&v)[EPT]) {
#pragma unroll
    for (int k = 0; k < EPT; ++k) { const int i = threadIdx.x + k * NT; if (i < CP) dst[i] = v[k]; }
}

// planes: [13][N][CP] = s0..s3 | u0..u3 | t0..t3 | (12 unused)   stats: [N][18] = (mu, rs) of LN0, then per block LN1, LN2
__global__ __launch_bounds__(NT) void gate_stack_fwd_kernel(const float* __restrict__ x, GateParams w, float* __restrict__ out,
                                                            float* __restrict__ planes, float* __restrict__ stats,
                                                            int N, int A, int C, int P, float eps)
{
    extern __shared__ float lds[];                  // 2 planes
    __shared__ float red[2 * NW];
    const int n = blockIdx.x, CP = C * P;
    const int an = n % A;                            // anchor of this plane: planes of several frames share the A per-anchor filters
    float* P0 = lds;
    float* P1 = lds + CP;
    const size_t plane_off = (size_t)n * CP, slab = (size_t)N * CP;
    float s[EPT], a[EPT];
    load_plane(x + plane_off, CP, s);
    float mu, rs;
    plane_stats(s, CP, eps, mu, rs, red);
    if (stats && threadIdx.x == 0) { stats[n * 18 + 0] = mu; stats[n * 18 + 1] = rs; }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int i = threadIdx.x + k * NT;
        if (i < CP) s[k] = (s[k] - mu) * rs * w.p[0][i] + w.p[1][i];
    }
    for (int b = 0; b < 4; ++b) {
        const float* const* q = w.p + 2 + 8 * b;
        if (planes) store_plane(planes + (size_t)b * slab + plane_off, CP, s);            // s_b
        store_plane(P0, CP, s);
        __syncthreads();
        dwconv_plane(P0, q[0] + an * 9, q[1][an], C, P, CP, false, a);                     // u_b
        if (planes) store_plane(planes + (size_t)(4 + b) * slab + plane_off, CP, a);
        plane_stats(a, CP, eps, mu, rs, red);
        if (stats && threadIdx.x == 0) { stats[n * 18 + 2 + 4 * b] = mu; stats[n * 18 + 3 + 4 * b] = rs; }
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int i = threadIdx.x + k * NT;
            if (i < CP) a[k] = fmaxf((a[k] - mu) * rs * q[2][i] + q[3][i], 0.f);           // v_b
        }
        store_plane(P1, CP, a);
        __syncthreads();
        dwconv_plane(P1, q[4] + an * 9, q[5][an], C, P, CP, false, a);                     // t_b
        if (planes) store_plane(planes + (size_t)(8 + b) * slab + plane_off, CP, a);
        plane_stats(a, CP, eps, mu, rs, red);
        if (stats && threadIdx.x == 0) { stats[n * 18 + 4 + 4 * b] = mu; stats[n * 18 + 5 + 4 * b] = rs; }
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int i = threadIdx.x + k * NT;
            if (i < CP) s[k] = fmaxf((a[k] - mu) * rs * q[6][i] + q[7][i] + s[k], 0.f);    // s_{b+1}
        }
        __syncthreads();                            // everyone is done reading P0/P1 before the next block overwrites
    }
    store_plane(out + plane_off, CP, s);
}

// block-reduce the 10 depth-wise filter gradient partials (9 taps + bias) and write / accumulate them
__device__ __forceinline__ void reduce_filter_grad(float (&acc)[10], float* red10, float* dw, float* db, int accumulate) {
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = wave_sum(acc[t]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int t = 0; t < 10; ++t) red10[t * NW + (threadIdx.x >> 6)] = acc[t];
    __syncthreads();
    if (threadIdx.x < 10) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) v += red10[threadIdx.x * NW + k];
        float* dst = threadIdx.x < 9 ? dw + threadIdx.x : db;
        *dst = accumulate ? *dst + v : v;
    }
}

// acc[tap] += g(c,p) * src(c+di-1, p+dj-1), acc[9] += g
__device__ __forceinline__ void filter_grad_partials(const float* src_plane, const float (&g)[EPT], int C, int P, int CP,
                                                     float (&acc)[10]) {
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = 0.f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int i = threadIdx.x + k * NT;
        if (i >= CP) continue;
        const int c = i / P, p = i - c * P;
        acc[9] += g[k];
#pragma unroll
        for (int di = 0; di < 3; ++di) {
            const int cc = c + di - 1;
            if (cc < 0 || cc >= C) continue;
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) {
                const int pp = p + dj - 1;
                if (pp < 0 || pp >= P) continue;
                acc[di * 3 + dj] += g[k] * src_plane[cc * P + pp];
            }
        }
    }
}

// LayerNorm backward on register planes: g = upstream (already masked), xin = LN input; writes the affine partials
// (g*xhat, g) to lnpart and returns dx in g.
__device__ __forceinline__ void ln_backward(float (&g)[EPT], const float (&xin)[EPT], float mu, float rs, const float (&gamma)[EPT],
                                            float* part_w, float* part_b, int CP, float* red) {
    float xh[EPT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int i = threadIdx.x + k * NT;
        xh[k] = 0.f;
        if (i < CP) {
            xh[k] = (xin[k] - mu) * rs;
            part_w[i] = g[k] * xh[k];
            part_b[i] = g[k];
            const float gw = g[k] * gamma[k];
            g[k] = gw;
            s1 += gw;
            s2 += gw * xh[k];
        }
    }
    block_sum2(s1, s2, red);
    s1 /= (float)CP; s2 /= (float)CP;
#pragma unroll
    for (int k = 0; k < EPT; ++k) g[k] = rs * (g[k] - s1 - xh[k] * s2);
}

// lnpart: [N][18][CP] per-anchor partial gradients of the 9 LayerNorms' (weight, bias)
__global__ __launch_bounds__(NT) void gate_stack_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ x,
                                                            const float* __restrict__ out, GateParams w,
                                                            const float* __restrict__ planes, const float* __restrict__ stats,
                                                            GateGrads dg, float* __restrict__ lnpart, float* __restrict__ fpart,
                                                            int N, int A, int C, int P, float eps, int accumulate)
{
    extern __shared__ float lds[];                  // 2 planes
    __shared__ float red[2 * NW], red10[10 * NW];
    const int n = blockIdx.x, CP = C * P;
    float* P0 = lds;
    float* P1 = lds + CP;
    const size_t plane_off = (size_t)n * CP, slab = (size_t)N * CP;
    float* lp = lnpart + (size_t)n * 18 * CP;
    const int an = n % A;
    // N == A: this workgroup is the only one that touches its anchor's filter gradients - write them directly;
    // N > A (planes of several frames): per-plane partials [N][8][10] in fpart, folded by gate_filter_grad_reduce_kernel
    const bool direct_f = N == A;
    float* fp = fpart + (size_t)n * 80;
    float g[EPT], v[EPT], acc[10];
    load_plane(gout + plane_off, CP, g);
    for (int b = 3; b >= 0; --b) {
        const float* const* q = w.p + 2 + 8 * b;
        float* const* dq = dg.p + 2 + 8 * b;
        // every global read of this block is issued up front (saved planes, LayerNorm affine, statistics): one exposed
        // memory latency per block instead of one per phase - the kernel is a chain of latencies, not of FLOPs
        float a_o[EPT], a_t[EPT], a[EPT], a_s[EPT], ln2w[EPT], ln1w[EPT], ln1b[EPT];
        load_plane(b == 3 ? out + plane_off : planes + (size_t)(b + 1) * slab + plane_off, CP, a_o);     // block output
        load_plane(planes + (size_t)(8 + b) * slab + plane_off, CP, a_t);                                // t_b
        load_plane(planes + (size_t)(4 + b) * slab + plane_off, CP, a);                                  // u_b
        load_plane(planes + (size_t)b * slab + plane_off, CP, a_s);                                      // s_b
        load_plane(q[6], CP, ln2w);
        load_plane(q[2], CP, ln1w);
        load_plane(q[3], CP, ln1b);
        const float mu2 = stats[n * 18 + 4 + 4 * b], rs2 = stats[n * 18 + 5 + 4 * b];
        const float mu1 = stats[n * 18 + 2 + 4 * b], rs1 = stats[n * 18 + 3 + 4 * b];
        // ---- relu of the block output, then LN2 backward (input t_b) ----
#pragma unroll
        for (int k = 0; k < EPT; ++k) g[k] = a_o[k] > 0.f ? g[k] : 0.f;
        float gres[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) gres[k] = g[k];
        ln_backward(g, a_t, mu2, rs2, ln2w, lp + (size_t)(2 + 4 * b + 2) * CP, lp + (size_t)(2 + 4 * b + 3) * CP, CP, red);   // g = dt
        // ---- conv2 backward: needs v_b = relu(LN1(u_b)) and dt as planes ----
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int i = threadIdx.x + k * NT;
            v[k] = i < CP ? fmaxf((a[k] - mu1) * rs1 * ln1w[k] + ln1b[k], 0.f) : 0.f;
        }
        __syncthreads();
        store_plane(P0, CP, v);
        store_plane(P1, CP, g);
        __syncthreads();
        filter_grad_partials(P0, g, C, P, CP, acc);
        if (direct_f) reduce_filter_grad(acc, red10, dq[4] + an * 9, dq[5] + an, accumulate);
        else reduce_filter_grad(acc, red10, fp + (2 * b + 1) * 10, fp + (2 * b + 1) * 10 + 9, 0);
        float dv[EPT];
        dwconv_plane(P1, q[4] + an * 9, 0.f, C, P, CP, true, dv);
#pragma unroll
        for (int k = 0; k < EPT; ++k) g[k] = v[k] > 0.f ? dv[k] : 0.f;                         // through the inner relu
        // ---- LN1 backward (input u_b, still in a[]) ----
        ln_backward(g, a, mu1, rs1, ln1w, lp + (size_t)(2 + 4 * b + 0) * CP, lp + (size_t)(2 + 4 * b + 1) * CP, CP, red);     // g = du
        // ---- conv1 backward: input s_b ----
        __syncthreads();
        store_plane(P0, CP, a_s);
        store_plane(P1, CP, g);
        __syncthreads();
        filter_grad_partials(P0, g, C, P, CP, acc);
        if (direct_f) reduce_filter_grad(acc, red10, dq[0] + an * 9, dq[1] + an, accumulate);
        else reduce_filter_grad(acc, red10, fp + (2 * b) * 10, fp + (2 * b) * 10 + 9, 0);
        dwconv_plane(P1, q[0] + an * 9, 0.f, C, P, CP, true, dv);
#pragma unroll
        for (int k = 0; k < EPT; ++k) g[k] = dv[k] + gres[k];                                  // + residual path
    }
    // ---- pre-norm: only its affine gradients are needed (the gate input is detached, Router4OL.py:275) ----
    float a[EPT];
    load_plane(x + plane_off, CP, a);
    {
        const float mu = stats[n * 18 + 0], rs = stats[n * 18 + 1];
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int i = threadIdx.x + k * NT;
            if (i < CP) { lp[i] = g[k] * ((a[k] - mu) * rs); lp[CP + i] = g[k]; }
        }
    }
}

// dst_j[i] (+)= sum_n lnpart[n][j][i]   for the 18 LayerNorm parameter tensors
// 256 threads = 64 columns x 4 row groups (group g sums the planes n = g, g+4, ...; fp64; the four sums are folded through LDS
// in a fixed order): a quarter of the serial chain per thread and four times the workgroups of the one-thread-per-column form
// (20 -> 7 us at 240 planes, and it scales with the plane count of the batched modes).
__global__ __launch_bounds__(256) void gate_ln_grad_reduce_kernel(const float* __restrict__ lnpart, GateGrads dg, int N, int CP,
                                                                  int accumulate)
{
    __shared__ double red[3][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long col = (long)blockIdx.x * 64 + lane;
    const bool ok = col < 18L * CP;
    const long cc = ok ? col : 0;
    const int j = (int)(cc / CP), i = (int)(cc - (long)j * CP);
    // parameter slot of LayerNorm j: 0,1 -> ln0 (w,b); then block b: ln1 (w,b) at 2+8b+2, ln2 (w,b) at 2+8b+6
    int slot;
    if (j < 2) slot = j;
    else { const int b = (j - 2) / 4, r = (j - 2) % 4; slot = 2 + 8 * b + (r < 2 ? 2 + r : 6 + (r - 2)); }
    float* dst = dg.p[slot] + i;
    const float old = (accumulate && ok && g == 0) ? *dst : 0.f;           // cold read first: it overlaps the sum over the planes
    double s = 0.0;
#pragma unroll 8
    for (int n = g; n < N; n += 4) s += (double)lnpart[((size_t)n * 18 + j) * CP + i];
    if (g) red[g - 1][lane] = s;
    __syncthreads();
    if (g == 0 && ok) *dst = old + (float)((s + red[0][lane]) + (red[1][lane] + red[2][lane]));
}

// filter gradients of a batch of frames: dst (+)= sum over the N/A planes of anchor a;  fpart [N][8 convs][9 taps + bias]
// conv c of block b sits at slot 2b + (0: conv1, 1: conv2);  parameter slots: conv1.weight 2+8b, conv1.bias 3+8b,
// conv2.weight 6+8b, conv2.bias 7+8b
__global__ __launch_bounds__(256) void gate_filter_grad_reduce_kernel(const float* __restrict__ fpart, GateGrads dg, int N, int A,
                                                                      int accumulate)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A * 80) return;
    const int a = i / 80, r = i - a * 80, conv = r / 10, t = r - conv * 10;
    const int b = conv >> 1, second = conv & 1;
    float* dst = t < 9 ? dg.p[2 + 8 * b + (second ? 4 : 0)] + a * 9 + t : dg.p[3 + 8 * b + (second ? 4 : 0)] + a;
    const float old = accumulate ? *dst : 0.f;
    float s = 0.f;
    for (int n = a; n < N; n += A) s += fpart[(size_t)n * 80 + r];
    *dst = old + s;
}

int g_gate_wave = 1;            // wave-per-plane kernels (gate_wave.hip) where they apply; phnet_tune_gate_wave(0) = the generic ones

bool gate_args_ok(int N, int C, int P) { return N >= 1 && C >= 1 && P >= 1 && (long)C * P <= (long)NT * EPT; }

}  // namespace

extern "C" int phnet_gate_wave_applies(int32_t C, int32_t P);
extern "C" int phnet_gate_wave_partial_planes(int32_t N);
extern "C" int phnet_gate_wave_fwd(const float* x, const float* const* params, float* out, float* saved, int32_t N, int32_t anchors, float eps,
                                   void* stream);
extern "C" int phnet_gate_wave_bwd(const float* gout, const float* x, const float* out, const float* const* params, const float* saved, float* const* grads,
                                   int32_t N, int32_t anchors, float eps, int32_t accumulate, void* workspace, void* stream);

PHNET_API int phnet_tune_gate_wave(int32_t on) { g_gate_wave = on != 0; return PHNET_OK; }

PHNET_API uint64_t phnet_gate_stack_saved_floats(int32_t N, int32_t C, int32_t P) { return (uint64_t)12 * N * C * P + (uint64_t)18 * N; }
PHNET_API uint64_t phnet_gate_stack_bwd_workspace(int32_t N, int32_t C, int32_t P) { return ((uint64_t)18 * N * C * P + (uint64_t)80 * N) * sizeof(float); }

// x [N][C][P] gate input; params: HOST array of 34 device pointers in the order
//   pre_norm.weight, pre_norm.bias, then for block 0..3: conv1.weight [N][9], conv1.bias [N], ln1.weight [C*P], ln1.bias,
//   conv2.weight, conv2.bias, ln2.weight, ln2.bias.
// out [N][C][P]; saved (training, may be NULL): phnet_gate_stack_saved_floats floats = 12 planes [N][C][P] + stats [N][18].
PHNET_API int phnet_gate_stack_fwd(const float* x, const float* const* params, float* out, float* saved,
                                   int32_t N, int32_t anchors, int32_t C, int32_t P, float eps, void* stream)
{
    if (!gate_args_ok(N, C, P) || anchors < 1 || N % anchors || !x || !params || !out) return PHNET_ERR_ARG;
    GateParams w;
    for (int i = 0; i < NPARAM; ++i) { w.p[i] = params[i]; if (!params[i]) return PHNET_ERR_ARG; }
    if (g_gate_wave && phnet_gate_wave_applies(C, P)) return phnet_gate_wave_fwd(x, params, out, saved, N, anchors, eps, stream);
    const size_t CP = (size_t)C * P;
    float* stats = saved ? saved + (size_t)12 * N * CP : nullptr;
    hipLaunchKernelGGL(gate_stack_fwd_kernel, dim3(N), dim3(NT), 2 * CP * sizeof(float), (hipStream_t)stream,
                       x, w, out, saved, stats, N, anchors, C, P, eps);
    return phnet_launch_status();
}

// gout [N][C][P] = d loss / d out.  grads: HOST array of 34 device pointers (same order as params) that are
// overwritten (accumulate=0) or added to (accumulate=1).  workspace >= phnet_gate_stack_bwd_workspace bytes.
PHNET_API int phnet_gate_stack_bwd(const float* gout, const float* x, const float* out, const float* const* params,
                                   const float* saved, float* const* grads, int32_t N, int32_t anchors, int32_t C, int32_t P,
                                   float eps, int32_t accumulate, void* workspace, uint64_t ws_bytes, void* stream)
{
    if (!gate_args_ok(N, C, P) || anchors < 1 || N % anchors || !gout || !x || !out || !params || !saved || !grads || !workspace)
        return PHNET_ERR_ARG;
    const size_t CP = (size_t)C * P;
    if (phnet_gate_stack_bwd_workspace(N, C, P) > ws_bytes) return PHNET_ERR_WORKSPACE;
    float* fpart = (float*)workspace + (size_t)18 * N * CP;
    GateParams w; GateGrads dg;
    for (int i = 0; i < NPARAM; ++i) { w.p[i] = params[i]; dg.p[i] = grads[i]; if (!params[i] || !grads[i]) return PHNET_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    int part_planes = N;                                           // partial planes of the LayerNorm affine gradients the column reduce reads
    if (g_gate_wave && phnet_gate_wave_applies(C, P)) {
        part_planes = phnet_gate_wave_partial_planes(N);
        // (the forward of the same switch position saved the four block inputs; partial-gradient layouts are shared with the generic kernel)
        const int rc = phnet_gate_wave_bwd(gout, x, out, params, saved, grads, N, anchors, eps, accumulate, workspace, stream);
        if (rc != PHNET_OK) return rc;
    } else
    hipLaunchKernelGGL(gate_stack_bwd_kernel, dim3(N), dim3(NT), 2 * CP * sizeof(float), st,
                       gout, x, out, w, saved, saved + (size_t)12 * N * CP, dg, (float*)workspace, fpart, N, anchors, C, P, eps, accumulate);
    if (N != anchors)
        hipLaunchKernelGGL(gate_filter_grad_reduce_kernel, dim3((anchors * 80 + 255) / 256), dim3(256), 0, st, fpart, dg, N, anchors,
                           accumulate);
    hipLaunchKernelGGL(gate_ln_grad_reduce_kernel, dim3((unsigned)ceil_div64(18L * CP, 64)), dim3(256), 0, st,
                       (const float*)workspace, dg, part_planes, (int)CP, accumulate);
    return phnet_launch_status();
}
