// Tower weights of a lane-head branch assembled for the 3-GEMM chain, and their gradients scattered back (gfx950).
//
// A branch of the lane head is T towers (cls / reg / offsets; libs/models/Router4OL.py:68-99, 308-326: two Linear + ReLU layers
// each) followed by one Linear head per tower.  The host runs them as ONE chain of three GEMMs on concatenated (layer 1),
// block-diagonal (layer 2) and block-structured (heads) weights.  Building those with torch.cat / torch.block_diag costs ~12 tiny
// launches per branch and clip, and their autograd backward (slices + one add_ into the gradient arena per parameter) ~40 more.
// Here: one launch writes all six assembled tensors, one launch adds the gradients gathered for them into the parameters' own
// gradient buffers.
//
// Parameter table of a branch, 6 device pointers per tower t: layer-1 weight [C][C], bias [C], layer-2 weight [C][C], bias [C],
// head weight [o_t][C], head bias [o_t].  Assembled: w1 [TC][C], b1 [TC], w2 [TC][TC] (block t on the diagonal), b2 [TC],
// wh [HW][TC] (rows of tower t's head in columns tC..tC+C-1; HW = sum o_t rounded up to a multiple of 4), bh [HW].
#include "common.h"

namespace {

constexpr int NT = 256, MAXT = 3;

struct TowerTable { const float* p[MAXT * 6]; float* g[MAXT * 6]; int out[MAXT]; };

struct Layout { long w1, b1, w2, b2, wh, bh, total; int TC, HW; };

__host__ __device__ inline Layout layout_of(int T, int C, const int* out) {
    Layout l;
    int hw = 0;
    for (int t = 0; t < T; ++t) hw += out[t];
    l.TC = T * C; l.HW = (hw + 3) & ~3;
    l.w1 = 0; l.b1 = l.w1 + (long)l.TC * C; l.w2 = l.b1 + l.TC; l.b2 = l.w2 + (long)l.TC * l.TC;
    l.wh = l.b2 + l.TC; l.bh = l.wh + (long)l.HW * l.TC; l.total = l.bh + l.HW;
    return l;
}

// For flat index i of the assembled concatenation (w1 | b1 | w2 | b2 | wh | bh): which parameter and element it mirrors (-1: a
// structural zero).
__device__ __forceinline__ void locate(long i, const Layout& l, int T, int C, const int* out, int& slot, long& elem)
{
    slot = -1; elem = 0;
    if (i < l.b1) { const int r = (int)(i / C), c = (int)(i - (long)r * C); slot = (r / C) * 6 + 0; elem = (long)(r % C) * C + c; return; }
    if (i < l.w2) { const int r = (int)(i - l.b1); slot = (r / C) * 6 + 1; elem = r % C; return; }
    if (i < l.b2) {
        const long j = i - l.w2; const int r = (int)(j / l.TC), c = (int)(j - (long)r * l.TC);
        if (r / C == c / C) { slot = (r / C) * 6 + 2; elem = (long)(r % C) * C + (c % C); }
        return;
    }
    if (i < l.wh) { const int r = (int)(i - l.b2); slot = (r / C) * 6 + 3; elem = r % C; return; }
    if (i < l.bh) {
        const long j = i - l.wh; int r = (int)(j / l.TC); const int c = (int)(j - (long)r * l.TC);
        for (int t = 0; t < T; ++t) { if (r < out[t]) { if (c / C == t) { slot = t * 6 + 4; elem = (long)r * C + (c % C); } return; } r -= out[t]; }
        return;
    }
    int r = (int)(i - l.bh);
    for (int t = 0; t < T; ++t) { if (r < out[t]) { slot = t * 6 + 5; elem = r; return; } r -= out[t]; }
}

__global__ __launch_bounds__(NT) void assemble_towers_kernel(TowerTable tb, int T, int C, float* __restrict__ dst)
{
    const Layout l = layout_of(T, C, tb.out);
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    if (i >= l.total) return;
    int slot; long e;
    locate(i, l, T, C, tb.out, slot, e);
    dst[i] = slot < 0 ? 0.f : tb.p[slot][e];
}

__global__ __launch_bounds__(NT) void scatter_tower_grads_kernel(TowerTable tb, int T, int C, const float* __restrict__ src, int accumulate)
{
    const Layout l = layout_of(T, C, tb.out);
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    if (i >= l.total) return;
    int slot; long e;
    locate(i, l, T, C, tb.out, slot, e);
    if (slot < 0 || !tb.g[slot]) return;
    float* d = tb.g[slot] + e;                         // every parameter element mirrors exactly one assembled element: no race
    *d = accumulate ? *d + src[i] : src[i];
}

bool args_ok(int T, int C, const int32_t* out) {
    if (T < 1 || T > MAXT || C < 1 || !out) return false;
    for (int t = 0; t < T; ++t) if (out[t] < 1) return false;
    return true;
}

}  // namespace

// Number of floats of the assembled concatenation (w1 | b1 | w2 | b2 | wh | bh) and, through offsets[6] (optional), where each
// part starts; HW through hw (optional).
PHNET_API uint64_t phnet_tower_layout(int32_t T, int32_t C, const int32_t* head_out, int64_t* offsets, int32_t* hw)
{
    if (!args_ok(T, C, head_out)) return 0;
    int o[MAXT] = {0, 0, 0};
    for (int t = 0; t < T; ++t) o[t] = head_out[t];
    const Layout l = layout_of(T, C, o);
    if (offsets) { offsets[0] = l.w1; offsets[1] = l.b1; offsets[2] = l.w2; offsets[3] = l.b2; offsets[4] = l.wh; offsets[5] = l.bh; }
    if (hw) *hw = l.HW;
    return (uint64_t)l.total;
}

// params: HOST array of 6*T device pointers (see the top of this file); dst: phnet_tower_layout floats.
PHNET_API int phnet_assemble_towers(const float* const* params, int32_t T, int32_t C, const int32_t* head_out, float* dst, void* stream)
{
    if (!args_ok(T, C, head_out) || !params || !dst) return PHNET_ERR_ARG;
    TowerTable tb{};
    for (int t = 0; t < T; ++t) tb.out[t] = head_out[t];
    for (int i = 0; i < 6 * T; ++i) { if (!params[i]) return PHNET_ERR_ARG; tb.p[i] = params[i]; }
    const Layout l = layout_of(T, C, tb.out);
    hipLaunchKernelGGL(assemble_towers_kernel, dim3((unsigned)ceil_div64(l.total, NT)), dim3(NT), 0, (hipStream_t)stream, tb, T, C, dst);
    return phnet_launch_status();
}

// src: gradients of the assembled concatenation; grads: HOST array of 6*T device pointers (NULL entries are skipped) that are
// overwritten (accumulate = 0) or added to (accumulate = 1).
PHNET_API int phnet_scatter_tower_grads(const float* src, float* const* grads, int32_t T, int32_t C, const int32_t* head_out,
                                        int32_t accumulate, void* stream)
{
    if (!args_ok(T, C, head_out) || !src || !grads) return PHNET_ERR_ARG;
    TowerTable tb{};
    for (int t = 0; t < T; ++t) tb.out[t] = head_out[t];
    for (int i = 0; i < 6 * T; ++i) tb.g[i] = grads[i];
    const Layout l = layout_of(T, C, tb.out);
    hipLaunchKernelGGL(scatter_tower_grads_kernel, dim3((unsigned)ceil_div64(l.total, NT)), dim3(NT), 0, (hipStream_t)stream, tb, T, C, src,
                       accumulate);
    return phnet_launch_status();
}
