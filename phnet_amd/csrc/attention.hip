// Fused multi-head attention core for the cross-frame transformer of branch B (gfx950): softmax(Q K^T / sqrt(d)) V with
// optional key-validity mask and an externally drawn dropout keep-mask, forward and backward, heads of width 16.
//
// Replaces the core of nn.MultiheadAttention as used by libs/models/utils/transformer.py:275-298 (self-attention over
// the 240 anchor queries, cross-attention to <= 40 memory tokens; 8 heads x 16): per attention, ATen issues
// ~12 launches forward (scale, bmm, mask, softmax, dropout, bmm, transposes) and ~20 backward; here 1 + 2.
// The problems are tiny (240 x 240 x 16 per head) and latency-bound: no MFMA, fp32 FMAs, K/V of one head staged in
// LDS, 16 lanes per query row (online softmax per lane, merged with shuffles).
//
// Tensors are addressed with row strides, so q/k/v may be column slices of a packed [L,3E] projection output and the
// gradients are written straight into the packed gradient buffer.
#include "common.h"

namespace {

constexpr int D = 16;                 // head width
constexpr int NT = 256;
constexpr int LPR = 16;               // lanes per row: each walks every 16th key (or query) - the per-lane loop is the critical path
constexpr int ROWS = NT / LPR;        // 16 rows (queries or keys) per workgroup -> 8 heads x 15 tiles = 120+ workgroups per launch
constexpr int MAXK = 256;             // keys per head held in LDS

struct AttnShape { int Lq, Lk, H; long sq, sk, sv, so; float scale, keep_scale; };      // gridDim.z = batch: clip b owns rows [b*Lq, (b+1)*Lq) / [b*Lk, (b+1)*Lk)

constexpr int DP = D + 4;              // LDS row pitch: 16-byte aligned rows (b128 access), 4 consecutive rows on distinct banks
typedef float f4 __attribute__((ext_vector_type(4)));

// Stage head h of `rows` rows (<= MAXK) of two row-strided arrays into LDS.  All global loads are issued before the first
// LDS store (branch-free, clamped addresses): one memory latency for the whole tile instead of one per loop trip.
__device__ __forceinline__ void stage_pair(const float* __restrict__ a, long sa, const float* __restrict__ b, long sb, int h, int rows,
                                           float (*As)[DP], float (*Bs)[DP], float a_scale)
{
    constexpr int N = MAXK * (D / 4) / NT;      // float4 chunks per thread and array
    f4 ra[N], rb[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int idx = threadIdx.x + NT * i, r = idx >> 2, c = (idx & 3) * 4;
        const int rr = r < rows ? r : 0;
        ra[i] = *reinterpret_cast<const f4*>(a + (size_t)rr * sa + h * D + c);
        rb[i] = *reinterpret_cast<const f4*>(b + (size_t)rr * sb + h * D + c);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int idx = threadIdx.x + NT * i, r = idx >> 2, c = (idx & 3) * 4;
        if (r < rows) {
            *reinterpret_cast<f4*>(&As[r][c]) = ra[i] * a_scale;
            *reinterpret_cast<f4*>(&Bs[r][c]) = rb[i];
        }
    }
}

// 16-wide dot products with four independent partial sums (a single chain of 16 dependent FMAs is pure latency here)
__device__ __forceinline__ float dot16(const float (&a)[D], const float* b) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int d = 0; d < D; d += 4) { s0 += a[d] * b[d]; s1 += a[d + 1] * b[d + 1]; s2 += a[d + 2] * b[d + 2]; s3 += a[d + 3] * b[d + 3]; }
    return (s0 + s1) + (s2 + s3);
}

// LPR adjacent lanes -> one row: reduce across them
__device__ __forceinline__ float quad_sum(float v) {
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float quad_max(float v) {
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(NT) void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                      const unsigned char* __restrict__ key_valid, const unsigned char* __restrict__ keep,
                                                      float* __restrict__ o, float* __restrict__ lse, AttnShape g, DropRng rng)
{
    __shared__ __attribute__((aligned(16))) float Ks[MAXK][DP], Vs[MAXK][DP];
    __shared__ unsigned char valid[MAXK];
    const int h = blockIdx.x, row = blockIdx.y * ROWS + threadIdx.x / LPR, part = threadIdx.x % LPR;
    {   // batch element blockIdx.z: contiguous row blocks of q/o and of k/v
        const size_t b = blockIdx.z;
        q += b * g.Lq * g.sq; k += b * g.Lk * g.sk; v += b * g.Lk * g.sv; o += b * g.Lq * g.so; lse += b * g.H * g.Lq;
        if (key_valid) key_valid += b * g.Lk;
        if (keep) keep += b * g.H * g.Lq * g.Lk;
    }
    // the query row first (branch-free float4 loads): its latency overlaps the K/V staging instead of following the barrier
    const bool live = row < g.Lq;
    float qr[D];
    {
        const f4* qp = reinterpret_cast<const f4*>(q + (size_t)(live ? row : 0) * g.sq + h * D);
#pragma unroll
        for (int c = 0; c < D / 4; ++c) {
            const f4 t = qp[c];
            qr[4 * c] = t.x * g.scale; qr[4 * c + 1] = t.y * g.scale; qr[4 * c + 2] = t.z * g.scale; qr[4 * c + 3] = t.w * g.scale;
        }
    }
    stage_pair(k, g.sk, v, g.sv, h, g.Lk, Ks, Vs, 1.0f);
    for (int i = threadIdx.x; i < g.Lk; i += NT) valid[i] = key_valid ? key_valid[i] : 1;
    __syncthreads();
    float m = -INFINITY, l = 0.f, acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.f;
    const unsigned char* kp = keep ? keep + ((size_t)h * g.Lq + (live ? row : 0)) * g.Lk : nullptr;
    const uint64_t seed = phnet_rng_seed(rng), rbase = ((uint64_t)h * g.Lq + (live ? row : 0)) * g.Lk;      // within batch entry blockIdx.z
    for (int kk = part; kk < g.Lk; kk += LPR) {
        if (!valid[kk]) continue;
        const float s = dot16(qr, Ks[kk]);
        const float mn = fmaxf(m, s);
        const float c = expf(m - mn), e = expf(s - mn);
        l = l * c + e;
        const bool kept = kp ? kp[kk] != 0 : (!rng.thresh || phnet_rng_keep(seed, phnet_rng_index_b(rng, blockIdx.z, rbase + kk), rng.thresh));
        const float ed = kept ? e * g.keep_scale : 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = acc[d] * c + ed * Vs[kk][d];
        m = mn;
    }
    // merge the LPR partial (m, l, acc) of the row
    const float mt = quad_max(m);
    const float c = (m == -INFINITY) ? 0.f : expf(m - mt);
    l = quad_sum(l * c);
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = quad_sum(acc[d] * c);
    if (live && part == 0) {
        const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) o[(size_t)row * g.so + h * D + d] = acc[d] * inv;
        lse[(size_t)h * g.Lq + row] = mt + logf(l);
    }
}

// role 0 (blockIdx.y < q_tiles): dQ for a tile of 16 queries;  role 1: dK, dV for a tile of 16 keys;  blockIdx.z = batch entry.
__global__ __launch_bounds__(NT) void attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                      const float* __restrict__ o, const float* __restrict__ dout,
                                                      const float* __restrict__ lse, const unsigned char* __restrict__ key_valid,
                                                      const unsigned char* __restrict__ keep,
                                                      float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
                                                      AttnShape g, long sdq, long sdk, long sdv, int q_tiles, DropRng rng)
{
    const uint64_t seed = phnet_rng_seed(rng);
    __shared__ __attribute__((aligned(16))) float As[MAXK][DP], Bs[MAXK][DP];     // role 0: K, V of the head; role 1: Q*scale, dO
    __shared__ float Dl[MAXK], Ls[MAXK];                   // role 1: rowsum(dO*O), lse per query
    __shared__ unsigned char valid[MAXK];
    const int h = blockIdx.x;
    {
        const size_t b = blockIdx.z;
        q += b * g.Lq * g.sq; k += b * g.Lk * g.sk; v += b * g.Lk * g.sv; o += b * g.Lq * g.so; dout += b * g.Lq * g.so;
        lse += b * g.H * g.Lq; dq += b * g.Lq * sdq; dk += b * g.Lk * sdk; dv += b * g.Lk * sdv;
        if (key_valid) key_valid += b * g.Lk;
        if (keep) keep += b * g.H * g.Lq * g.Lk;
    }
    const bool dq_role = (int)blockIdx.y < q_tiles;
    const int tile = dq_role ? blockIdx.y : blockIdx.y - q_tiles;
    const int row = tile * ROWS + threadIdx.x / LPR, part = threadIdx.x % LPR;
    for (int i = threadIdx.x; i < g.Lk; i += NT) valid[i] = key_valid ? key_valid[i] : 1;
    if (dq_role) {
        // this row's q, dO, O and lse first: their latency overlaps the K/V staging
        const bool live = row < g.Lq;
        const int r = live ? row : 0;
        float qr[D], dor[D], delta = 0.f;
        {
            const f4* qp = reinterpret_cast<const f4*>(q + (size_t)r * g.sq + h * D);
            const f4* gp = reinterpret_cast<const f4*>(dout + (size_t)r * g.so + h * D);
            const f4* op = reinterpret_cast<const f4*>(o + (size_t)r * g.so + h * D);
#pragma unroll
            for (int c = 0; c < D / 4; ++c) {
                const f4 tq = qp[c], tg = gp[c], to = op[c];
                qr[4 * c] = tq.x * g.scale; qr[4 * c + 1] = tq.y * g.scale; qr[4 * c + 2] = tq.z * g.scale; qr[4 * c + 3] = tq.w * g.scale;
                dor[4 * c] = tg.x; dor[4 * c + 1] = tg.y; dor[4 * c + 2] = tg.z; dor[4 * c + 3] = tg.w;
                delta += (tg.x * to.x + tg.y * to.y) + (tg.z * to.z + tg.w * to.w);
            }
        }
        const float L = lse[(size_t)h * g.Lq + r];
        stage_pair(k, g.sk, v, g.sv, h, g.Lk, As, Bs, 1.0f);
        __syncthreads();
        const unsigned char* kp = keep ? keep + ((size_t)h * g.Lq + r) * g.Lk : nullptr;
        const uint64_t rbase = ((uint64_t)h * g.Lq + r) * g.Lk;
        float acc[D];
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = 0.f;
        for (int kk = part; kk < g.Lk; kk += LPR) {
            if (!valid[kk]) continue;
            const float s = dot16(qr, As[kk]);
            float dp = dot16(dor, Bs[kk]);
            const float p = expf(s - L);
            const bool kept = kp ? kp[kk] != 0 : (!rng.thresh || phnet_rng_keep(seed, phnet_rng_index_b(rng, blockIdx.z, rbase + kk), rng.thresh));
            dp = kept ? dp * g.keep_scale : 0.f;
            const float ds = p * (dp - delta);
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] += ds * As[kk][d];
        }
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = quad_sum(acc[d]);
        if (live && part == 0)
#pragma unroll
            for (int d = 0; d < D; ++d) dq[(size_t)row * sdq + h * D + d] = acc[d] * g.scale;
    } else {
        // stage Q*scale, dO, delta, lse of ALL queries of this head (Lq <= MAXK)
        stage_pair(q, g.sq, dout, g.so, h, g.Lq, As, Bs, g.scale);
        __syncthreads();
        for (int qq = threadIdx.x; qq < g.Lq; qq += NT) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) s += Bs[qq][d] * o[(size_t)qq * g.so + h * D + d];
            Dl[qq] = s;
            Ls[qq] = lse[(size_t)h * g.Lq + qq];
        }
        __syncthreads();
        const bool live = row < g.Lk;
        const int r = live ? row : 0;
        float kr[D], vr[D], dkr[D], dvr[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            kr[d] = k[(size_t)r * g.sk + h * D + d];
            vr[d] = v[(size_t)r * g.sv + h * D + d];
            dkr[d] = 0.f; dvr[d] = 0.f;
        }
        const bool kvalid = live && valid[r];
        if (kvalid)
            for (int qq = part; qq < g.Lq; qq += LPR) {
                const float s = dot16(kr, As[qq]), dp = dot16(vr, Bs[qq]);
                const float p = expf(s - Ls[qq]);
                const uint64_t ei = ((uint64_t)h * g.Lq + qq) * g.Lk + r;
                const bool kept = keep ? keep[ei] != 0 : (!rng.thresh || phnet_rng_keep(seed, phnet_rng_index_b(rng, blockIdx.z, ei), rng.thresh));
                const float pd = kept ? p * g.keep_scale : 0.f;       // dropped attention weight
                const float ds = p * ((kept ? dp * g.keep_scale : 0.f) - Dl[qq]);
#pragma unroll
                for (int d = 0; d < D; ++d) { dvr[d] += pd * Bs[qq][d]; dkr[d] += ds * As[qq][d]; }
            }
#pragma unroll
        for (int d = 0; d < D; ++d) { dkr[d] = quad_sum(dkr[d]); dvr[d] = quad_sum(dvr[d]); }
        if (live && part == 0)
#pragma unroll
            for (int d = 0; d < D; ++d) {
                dk[(size_t)row * sdk + h * D + d] = dkr[d];           // As already carries the 1/sqrt(d) scale
                dv[(size_t)row * sdv + h * D + d] = dvr[d];
            }
    }
}

// Head width 32 (the Router4OLV2 family: d_model 256, 8 heads; Router4OLV2.py:98-103), forward only and without dropout -
// that family is inference-only.  Same structure as attn_fwd_kernel; K/V tiles in dynamic LDS (2 x Lk x 36 floats > 64 KB at
// Lk = 240: the self-attention fallback of frames without memory, Router4OLV2.py:320-325).
constexpr int DW = 32, DWP = DW + 4;
__global__ __launch_bounds__(NT) void attn_fwd_wide_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                           const unsigned char* __restrict__ key_valid, float* __restrict__ o,
                                                           float* __restrict__ lse, AttnShape g)
{
    extern __shared__ __attribute__((aligned(16))) float wide_lds[];
    __shared__ unsigned char valid[MAXK];
    float (*Ks)[DWP] = reinterpret_cast<float (*)[DWP]>(wide_lds);
    float (*Vs)[DWP] = reinterpret_cast<float (*)[DWP]>(wide_lds + (size_t)g.Lk * DWP);
    const int h = blockIdx.x, row = blockIdx.y * ROWS + threadIdx.x / LPR, part = threadIdx.x % LPR;
    {
        const size_t b = blockIdx.z;
        q += b * g.Lq * g.sq; k += b * g.Lk * g.sk; v += b * g.Lk * g.sv; o += b * g.Lq * g.so; lse += b * g.H * g.Lq;
        if (key_valid) key_valid += b * g.Lk;
    }
    const bool live = row < g.Lq;
    float qr[DW];
    {
        const f4* qp = reinterpret_cast<const f4*>(q + (size_t)(live ? row : 0) * g.sq + h * DW);
#pragma unroll
        for (int c = 0; c < DW / 4; ++c) {
            const f4 t = qp[c];
            qr[4 * c] = t.x * g.scale; qr[4 * c + 1] = t.y * g.scale; qr[4 * c + 2] = t.z * g.scale; qr[4 * c + 3] = t.w * g.scale;
        }
    }
    for (int idx = threadIdx.x; idx < g.Lk * (DW / 4); idx += NT) {
        const int r = idx / (DW / 4), c = (idx - r * (DW / 4)) * 4;
        *reinterpret_cast<f4*>(&Ks[r][c]) = *reinterpret_cast<const f4*>(k + (size_t)r * g.sk + h * DW + c);
        *reinterpret_cast<f4*>(&Vs[r][c]) = *reinterpret_cast<const f4*>(v + (size_t)r * g.sv + h * DW + c);
    }
    for (int i = threadIdx.x; i < g.Lk; i += NT) valid[i] = key_valid ? key_valid[i] : 1;
    __syncthreads();
    float m = -INFINITY, l = 0.f, acc[DW];
#pragma unroll
    for (int d = 0; d < DW; ++d) acc[d] = 0.f;
    for (int kk = part; kk < g.Lk; kk += LPR) {
        if (!valid[kk]) continue;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int d = 0; d < DW; d += 4) {
            s0 += qr[d] * Ks[kk][d]; s1 += qr[d + 1] * Ks[kk][d + 1]; s2 += qr[d + 2] * Ks[kk][d + 2]; s3 += qr[d + 3] * Ks[kk][d + 3];
        }
        const float s = (s0 + s1) + (s2 + s3);
        const float mn = fmaxf(m, s);
        const float c = expf(m - mn), e = expf(s - mn);
        l = l * c + e;
#pragma unroll
        for (int d = 0; d < DW; ++d) acc[d] = acc[d] * c + e * Vs[kk][d];
        m = mn;
    }
    const float mt = quad_max(m);
    const float c = (m == -INFINITY) ? 0.f : expf(m - mt);
    l = quad_sum(l * c);
#pragma unroll
    for (int d = 0; d < DW; ++d) acc[d] = quad_sum(acc[d] * c);
    if (live && part == 0) {
        const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
        for (int d = 0; d < DW; ++d) o[(size_t)row * g.so + h * DW + d] = acc[d] * inv;
        lse[(size_t)h * g.Lq + row] = mt + logf(l);
    }
}

bool attn_ok(int Lq, int Lk, int H, int E) { return Lq >= 1 && Lk >= 1 && Lq <= MAXK && Lk <= MAXK && H >= 1 && E == H * D; }
// float4 staging: rows 16-byte aligned
bool aligned16(const void* p, int64_t stride) { return ((uintptr_t)p & 15) == 0 && (stride & 3) == 0; }

}  // namespace

// B clips in one launch: clip b owns rows [b*Lq, (b+1)*Lq) of q / o / dq and [b*Lk, (b+1)*Lk) of k / v / dk / dv; key_valid
// u8[B][Lk], keep u8[B][H][Lq][Lk], lse [B][H][Lq].  Per clip:
// q [Lq][.] row stride sq, k/v [Lk][.] row strides sk/sv (heads packed along the row: column h*16+d); o [Lq][.] stride so;
// key_valid (optional) u8[Lk]; dropout of the attention weights either by an explicit mask keep u8[H][Lq][Lk] (kept weights
// scaled by keep_scale) or, when keep is NULL and rng_state/drop_p are given, by the counter-based mask of common.h
// (site id rng_call, scale 1/(1-p)); lse [H][Lq] saved for the backward.  Lq, Lk <= 256, head width 16 (E = 16 H); the forward
// also takes head width 32 (E = 32 H) without dropout (keep = NULL, drop_p = 0): the inference-only Router4OLV2 family.
PHNET_API int phnet_attention_fwd(const float* q, const float* k, const float* v, const uint8_t* key_valid, const uint8_t* keep,
                                  float* o, float* lse, int32_t B, int32_t Lq, int32_t Lk, int32_t H, int32_t E,
                                  int64_t sq, int64_t sk, int64_t sv, int64_t so, float keep_scale,
                                  const uint64_t* rng_state, uint64_t rng_call, float drop_p, void* stream)
{
    if (E == H * DW) {                                   // head width 32: forward only, no dropout (Router4OLV2 family, inference)
        if (B < 1 || Lq < 1 || Lk < 1 || Lq > MAXK || Lk > MAXK || H < 1 || !q || !k || !v || !o || !lse || keep ||
            (rng_state && drop_p > 0.f)) return PHNET_ERR_ARG;
        if (!aligned16(k, sk) || !aligned16(v, sv) || !aligned16(q, sq)) return PHNET_ERR_ARG;
        const size_t lds = (size_t)2 * Lk * DWP * sizeof(float);
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute((const void*)attn_fwd_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PHNET_ERR_LAUNCH;
        AttnShape gw{Lq, Lk, H, sq, sk, sv, so, 1.0f / sqrtf((float)DW), 1.0f};
        hipLaunchKernelGGL(attn_fwd_wide_kernel, dim3(H, (Lq + ROWS - 1) / ROWS, B), dim3(NT), lds, (hipStream_t)stream,
                           q, k, v, key_valid, o, lse, gw);
        return phnet_launch_status();
    }
    if (B < 1 || !attn_ok(Lq, Lk, H, E) || !q || !k || !v || !o || !lse || drop_p < 0.f || drop_p >= 1.f) return PHNET_ERR_ARG;
    if (!aligned16(k, sk) || !aligned16(v, sv) || !aligned16(q, sq)) return PHNET_ERR_ARG;
    const DropRng rng = keep ? DropRng{nullptr, 0, 0u, 0u, 0u} : phnet_make_rng(rng_state, rng_call, drop_p);
    AttnShape g{Lq, Lk, H, sq, sk, sv, so, 1.0f / sqrtf((float)D), keep ? keep_scale : (rng.thresh ? 1.0f / (1.0f - drop_p) : 1.0f)};
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(H, (Lq + ROWS - 1) / ROWS, B), dim3(NT), 0, (hipStream_t)stream,
                       q, k, v, key_valid, keep, o, lse, g, rng);
    return phnet_launch_status();
}

// dq [Lq][.] stride sdq, dk/dv [Lk][.] strides sdk/sdv are overwritten (rows of masked keys get zeros).
PHNET_API int phnet_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* dout,
                                  const float* lse, const uint8_t* key_valid, const uint8_t* keep,
                                  float* dq, float* dk, float* dv, int32_t B, int32_t Lq, int32_t Lk, int32_t H, int32_t E,
                                  int64_t sq, int64_t sk, int64_t sv, int64_t so, int64_t sdq, int64_t sdk, int64_t sdv,
                                  float keep_scale, const uint64_t* rng_state, uint64_t rng_call, float drop_p, void* stream)
{
    if (B < 1 || !attn_ok(Lq, Lk, H, E) || !q || !k || !v || !o || !dout || !lse || !dq || !dk || !dv || drop_p < 0.f || drop_p >= 1.f)
        return PHNET_ERR_ARG;
    if (!aligned16(k, sk) || !aligned16(v, sv) || !aligned16(q, sq) || !aligned16(dout, so) || !aligned16(o, so)) return PHNET_ERR_ARG;
    const DropRng rng = keep ? DropRng{nullptr, 0, 0u, 0u, 0u} : phnet_make_rng(rng_state, rng_call, drop_p);
    AttnShape g{Lq, Lk, H, sq, sk, sv, so, 1.0f / sqrtf((float)D), keep ? keep_scale : (rng.thresh ? 1.0f / (1.0f - drop_p) : 1.0f)};
    const int qt = (Lq + ROWS - 1) / ROWS, kt = (Lk + ROWS - 1) / ROWS;
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(H, qt + kt, B), dim3(NT), 0, (hipStream_t)stream,
                       q, k, v, o, dout, lse, key_valid, keep, dq, dk, dv, g, (long)sdq, (long)sdk, (long)sdv, qt, rng);
    return phnet_launch_status();
}
