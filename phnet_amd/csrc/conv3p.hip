// 3x3 / stride 1 / pad 1 convolution, forward and data gradient, on PACKED weights (gfx950).
//
// Replaces conv3x3s1_kernel (conv.hip) for the trunk / FPN 3x3 convolutions (libs/models/resnet.py:79-95 BasicBlock.conv1 /
// conv2, libs/models/fpn.py:156-160 fpn_convs): same arithmetic (f32 storage and accumulation, every product as the exact
// three-term bf16 split = 6 x v_mfma_f32_32x32x16_bf16, igemm.h), same epilogues, different data flow:
//
//  * the WEIGHT operand never touches the LDS or the vector ALU.  conv3p_pack_kernel splits every weight ONCE per step into
//    its three bf16 terms and lays them out in MFMA fragment order: the 16 bytes lane (r, h) feeds to a 16-deep step of a
//    32-column fragment are contiguous, so a wave fetches its B operand with three coalesced 1-KB buffer loads per step,
//    straight into registers, three steps ahead.  (conv3x3s1_kernel re-split the same 64x16 weight tile in all 1250
//    workgroups of a layer, every step: 22 VALU instructions + 3 LDS writes + 3 LDS reads per thread and step.)  The data
//    gradient is the same kernel on the other packing (flipped taps, transposed channels): no K-strided operand path.
//  * 128 x 64 workgroup tile, 64 x 32 per wave (two accumulator fragments): 12 MFMAs per step and wave against 6 LDS
//    fragment reads, 36 MFMAs per barrier (a unit = filter row x 16-channel chunk: the 130 pixels m0-1 .. m0+128 of the
//    shifted image row staged once - split into bf16 planes on the way - and multiplied by the three taps dx).
//
// Roofline: MFMA (bf16 pipe, 6 MFMAs per f32-exact product: 416.7 TFLOP/s algorithmic); algorithmic FLOP = 2 * M * Co * 9Ci.
#include "igemm.h"

using namespace igemm;

namespace {

struct P3Shape {
    int N, H, W;             // image (input and output have the same size: stride 1, pad 1)
    int Ca, Nn;              // A-side channels (fwd: Ci, dgrad: Co), GEMM columns (fwd: Co, dgrad: Ci)
    int splits, units_per_split;   // split-K over units (filter row, 16-channel chunk), blockIdx.z
};

constexpr int P3_BM = 128, P3_ROWS = P3_BM + 2, P3_THREADS = 256;      // columns per workgroup: 64 * FN (template)
constexpr int P3_APITCH = 48;                                 // bytes: 16 bf16 + 16 pad (conflict-free ds_read_b128, igemm.h KContigPlanes)
constexpr int P3_ZROW = P3_ROWS;                              // one more row per plane that stays zero: where border lanes read
constexpr int P3_APLANE = (P3_ROWS + 1) * P3_APITCH, P3_AIMG = 3 * P3_APLANE;
constexpr int P3_PFB = 3;                                     // weight fragments in flight, in steps (= one unit ahead)
constexpr size_t P3_LDS = 2 * P3_AIMG + 3 * 512;

// packed weights: [step q = (dy*CC + cc)*3 + dx][fragment j = n / 32][plane][lane] x 16 bytes
// lane (r = lane & 31, h = lane >> 5) holds k = 8h .. 8h+7 of the step's 16 channels for column n = 32j + r.
struct P3PackJob { const float* w; unsigned char* dst; int Co, Ci, dgrad, pad_; long first; };   // `first`: index of the job's first item

__device__ __forceinline__ void conv3p_pack_item(const float* __restrict__ w, unsigned char* __restrict__ packed, int Co, int Ci,
                                                 int dgrad, long i)
{
    const int Ca = dgrad ? Co : Ci, Nn = dgrad ? Ci : Co;
    const int CC = Ca >> 4, NF = Nn >> 5;
    const int lane = (int)(i & 63);
    const long qj = i >> 6;
    const int j = (int)(qj % NF);
    const int q = (int)(qj / NF);
    const int dx = q % 3, cc = (q / 3) % CC, dy = q / (3 * CC);
    const int r = lane & 31, h = lane >> 5;
    const int n = 32 * j + r, k0 = cc * 16 + 8 * h;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        // OHWI: w[co][r][s][ci]
        v[e] = !dgrad ? w[(((long)n * 3 + dy) * 3 + dx) * Ci + k0 + e]
                      : w[(((long)(k0 + e) * 3 + (2 - dy)) * 3 + (2 - dx)) * Ci + n];
    }
    unsigned hi[4], mid[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split3_pair(v[2 * e], v[2 * e + 1], hi[e], mid[e], lo[e]);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    unsigned char* dst = packed + ((qj * 3) * 64 + lane) * 16;
    *reinterpret_cast<u32x4*>(dst) = (u32x4){hi[0], hi[1], hi[2], hi[3]};
    *reinterpret_cast<u32x4*>(dst + 1024) = (u32x4){mid[0], mid[1], mid[2], mid[3]};
    *reinterpret_cast<u32x4*>(dst + 2048) = (u32x4){lo[0], lo[1], lo[2], lo[3]};
}

__global__ __launch_bounds__(256) void conv3p_pack_kernel(const float* __restrict__ w, unsigned char* __restrict__ packed,
                                                          int Co, int Ci, int dgrad)
{
    const int Ca = dgrad ? Co : Ci, Nn = dgrad ? Ci : Co;
    const long total = (long)9 * (Ca >> 4) * (Nn >> 5) * 64;          // (q, j, lane) triples
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) conv3p_pack_item(w, packed, Co, Ci, dgrad, i);
}

// every 3x3 weight of the model in ONE launch: item i belongs to the job with the largest `first` <= i (jobs sorted by `first`)
__global__ __launch_bounds__(256) void conv3p_pack_jobs_kernel(const P3PackJob* __restrict__ jobs, int njobs, long total)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {                                          // (a workgroup's 256 items almost always share one job: uniform)
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first <= i) lo = mid; else hi = mid - 1;
    }
    const P3PackJob jb = jobs[lo];
    conv3p_pack_item(jb.w, jb.dst, jb.Co, jb.Ci, jb.dgrad, i - jb.first);
}

template <int FN>
__global__ __launch_bounds__(P3_THREADS, FN == 1 ? 3 : 2) void conv3p_kernel(
    const float* __restrict__ X, const unsigned char* __restrict__ Bp, const float* __restrict__ bias,
    const float* __restrict__ addend, float* __restrict__ out, P3Shape g, int relu, float* __restrict__ stats)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned char* A3 = reinterpret_cast<unsigned char*>(lds);       // [2][3 planes][130 rows][48]
    unsigned char* dump = A3 + 2 * P3_AIMG;                          // 3 x 512 bytes: where lanes without a third A chunk write

    const int W = g.W, H = g.H, Ca = g.Ca;
    const int M = g.N * H * W;
    const int CC = Ca >> 4, NF = g.Nn >> 5;
    constexpr int BN = 64 * FN;
    const int tiles_n = g.Nn / BN;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (int)(tile / tiles_n) * P3_BM, n0 = (int)(tile % tiles_n) * BN;
    const int u_begin = blockIdx.z * g.units_per_split, u_end = min(3 * CC, u_begin + g.units_per_split);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 32 * FN;

    constexpr unsigned OOB = 0x80000000u;
    __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)min((long)M * Ca * 4, (long)0x7fffffff), 0x00020000);
    __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, (int)min((long)9 * Ca * g.Nn * 6, (long)0x7fffffff), 0x00020000);

    // ---- A staging: 130 rows x 4 chunks = 520 float4: thread tid takes idx = tid, tid + 256 and (tid < 8) 512 + tid ----
    int a_off[3], a_ok[3], a_lds[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int idx = tid + 256 * i, row = idx >> 2, chunk = idx & 3;
        const bool rowok = idx < 4 * P3_ROWS;
        const int t = m0 - 1 + row;                          // aligned pixel of the row
        const bool tok = rowok && (unsigned)t < (unsigned)M;
        const int tt = tok ? t : 0;
        const int y = (tt / W) % H;
        a_ok[i] = (tok && y >= 1 ? 1 : 0) | (tok ? 2 : 0) | (tok && y + 1 < H ? 4 : 0);          // bit dyi: image row y + dyi - 1 exists
        a_off[i] = (t * Ca + chunk * 4) * 4;
        a_lds[i] = rowok ? row * P3_APITCH + chunk * 8 : -1;
    }
    const bool wave0 = __builtin_amdgcn_readfirstlane(wave) == 0;
    int la_dy = u_begin / CC, la_cc = u_begin - la_dy * CC, la_u = u_begin;           // next A block to load
    auto load_a = [&](f32x4 (&reg)[3]) {
        const bool uok = la_u < u_end;
        const int s_a = ((la_dy - 1) * W * Ca + la_cc * 16) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = uok && ((a_ok[i] >> la_dy) & 1);
            reg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, ok ? a_off[i] + s_a : (int)OOB, 0, 0));
        }
        if (wave0) {
            const bool ok = uok && ((a_ok[2] >> la_dy) & 1);
            reg[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, ok ? a_off[2] + s_a : (int)OOB, 0, 0));
        }
        ++la_u; ++la_cc;
        const int wrap = la_cc == CC;
        la_cc = wrap ? 0 : la_cc;
        la_dy += wrap;
    };
    auto store_a = [&](int aoff, const f32x4 (&reg)[3]) {
        store_split3<P3_APLANE>(A3 + aoff, a_lds[0], reg[0]);
        store_split3<P3_APLANE>(A3 + aoff, a_lds[1], reg[1]);
        if (wave0) {
            unsigned h0, m0_, l0, h1, m1, l1;
            split3_pair(reg[2].x, reg[2].y, h0, m0_, l0);
            split3_pair(reg[2].z, reg[2].w, h1, m1, l1);
            const bool has = a_lds[2] >= 0;
            unsigned char* q = has ? A3 + aoff + a_lds[2] : dump + lane * 8;
            const int ps = has ? P3_APLANE : 512;
            *reinterpret_cast<u32x2*>(q) = (u32x2){h0, h1};
            *reinterpret_cast<u32x2*>(q + ps) = (u32x2){m0_, m1};
            *reinterpret_cast<u32x2*>(q + 2 * ps) = (u32x2){l0, l1};
        }
    };
    // ---- B: the wave's fragment j of step q, three planes, straight into registers ----
    const int jfrag = (n0 + wn) >> 5;
    const int b_lane = (jfrag * 3 * 64 + lane) * 16;
    const int b_step = NF * 3 * 64 * 16;                     // bytes per step
    int lb_q = u_begin * 3;                                   // next step to load
    const int q_end = u_end * 3;
    auto load_b = [&](Frag3 (&f)[FN]) {
        const bool ok = lb_q < q_end;
        const int off = b_lane + lb_q * b_step;
#pragma unroll
        for (int jn = 0; jn < FN; ++jn) {
            f[jn].hi = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, ok ? off + jn * 3072 : (int)OOB, 0, 0));
            f[jn].mid = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, ok ? off + jn * 3072 + 1024 : (int)OOB, 0, 0));
            f[jn].lo = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, ok ? off + jn * 3072 + 2048 : (int)OOB, 0, 0));
        }
        ++lb_q;
    };

    // accumulators start from bias (+ addend) when this launch is the final pass
    const bool final_pass = g.splits == 1;
    f32x16 acc[2][FN];
    {
        // (branch-free: out-of-range rows and a missing addend read zeros through the buffer descriptor)
        const bool use_add = final_pass && addend != nullptr;
        __amdgpu_buffer_rsrc_t add_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)addend, 0, use_add ? (int)min((long)M * g.Nn * 4, (long)0x7fffffff) : 0, 0x00020000);
#pragma unroll
        for (int jn = 0; jn < FN; ++jn) {
            const int n = n0 + wn + 32 * jn + frag_col(lane);
            const float bv = (final_pass && bias) ? bias[n] : 0.f;
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm + 32 * f + frag_row(lane, e);
                    const int off = (use_add && m < M) ? (m * g.Nn + n) * 4 : (int)OOB;
                    acc[f][jn][e] = bv + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(add_rsrc, off, 0, 0));
                }
        }
    }
    // A fragment addresses per (fragment, tap), relative to the image: row wm + 32 f + r + dx of the lane's pixel - or, where the tap
    // leaves the image row on that pixel (x = 0 for dx = 0, x = W-1 for dx = 2), the image's zero row: no masking in the loop
    int a_rel[2][3];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int rr = wm + 32 * f + (lane & 31);
        const int pr = m0 + rr;
        const int px = (pr < M ? pr : 0) % W;
        const int hb = (lane >> 5) * 16;
        a_rel[f][0] = (px == 0 ? P3_ZROW : rr) * P3_APITCH + hb;
        a_rel[f][1] = (rr + 1) * P3_APITCH + hb;
        a_rel[f][2] = (px == W - 1 ? P3_ZROW : rr + 2) * P3_APITCH + hb;
    }
    for (int i = tid; i < 2 * 3 * (P3_APITCH / 4); i += P3_THREADS) {          // the zero rows of both images, all planes
        const int img = i / (3 * (P3_APITCH / 4)), rem = i % (3 * (P3_APITCH / 4)), pl = rem / (P3_APITCH / 4), w4 = rem % (P3_APITCH / 4);
        *reinterpret_cast<unsigned*>(A3 + img * P3_AIMG + pl * P3_APLANE + P3_ZROW * P3_APITCH + w4 * 4) = 0u;
    }

    if (u_begin < u_end) {
        f32x4 a_reg[3];
        Frag3 bq[P3_PFB][FN];
        load_a(a_reg);
#pragma unroll
        for (int d = 0; d < P3_PFB; ++d) load_b(bq[d]);
        store_a(0, a_reg);
        load_a(a_reg);
        __syncthreads();
        int aoff = 0;
        auto step = [&](auto DX) {
            constexpr int dxi = decltype(DX)::value;
            Frag3 a[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const unsigned char* p = A3 + aoff + a_rel[f][dxi];
                a[f].hi = *reinterpret_cast<const bf16x8*>(p);
                a[f].mid = *reinterpret_cast<const bf16x8*>(p + P3_APLANE);
                a[f].lo = *reinterpret_cast<const bf16x8*>(p + 2 * P3_APLANE);
            }
            mma3_step<2, FN>(a, bq[dxi], acc);
            load_b(bq[dxi]);                                  // the slot takes the fragment three steps ahead
        };
        for (int u = u_begin; u < u_end; ++u) {
            step(std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            step(std::integral_constant<int, 1>{});
            __builtin_amdgcn_sched_barrier(0);                // (keeps the split of the NEXT unit's A block - and the wait for its loads - out of steps 0 and 1)
            store_a(aoff ^ P3_AIMG, a_reg);                   // the next unit's A block: split + LDS fill beside the MFMAs of step 2
            load_a(a_reg);
            step(std::integral_constant<int, 2>{});
            __syncthreads();
            aoff ^= P3_AIMG;
        }
    }

    // ---- per-channel statistics of the result for a following BatchNorm: one (sum, sum of squares) row per 64-row wave tile ----
    if (stats != nullptr && final_pass) {
#pragma unroll
        for (int jn = 0; jn < FN; ++jn) {
            const int n = n0 + wn + 32 * jn + frag_col(lane);
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float v = acc[f][jn][e]; sm += v; sq += v * v; }
            sm += __shfl_xor(sm, 32, 64);
            sq += __shfl_xor(sq, 32, 64);
            if (lane < 32) {
                float* p = stats + (size_t)((m0 + wm) >> 6) * 2 * g.Nn;
                p[n] = sm;
                p[g.Nn + n] = sq;
            }
        }
    }
    float* dst = out + (size_t)blockIdx.z * M * g.Nn;
#pragma unroll
    for (int jn = 0; jn < FN; ++jn) {
        const int n = n0 + wn + 32 * jn + frag_col(lane);
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm + 32 * f + frag_row(lane, e);
                if (m < M) dst[(size_t)m * g.Nn + n] = (final_pass && relu) ? fmaxf(acc[f][jn][e], 0.f) : acc[f][jn][e];
            }
    }
}

// out[i] = sum_z part[z][i] (+bias) (+addend) (relu); optionally the BatchNorm statistics rows of the result (conv.hip's
// splitk_reduce_kernel, restated here so that this translation unit stands alone)
constexpr int P3_RCH = 4;        // 1024-element chunks per workgroup of the statistics-producing reduce: 4x fewer partial rows to finalize

__device__ __forceinline__ f32x4 conv3p_reduce_one(const float* __restrict__ part, float* __restrict__ out, const float* __restrict__ bias,
                                                   const float* __restrict__ addend, long i, long total4, int ncols, int splits, int relu)
{
    const f32x4* p = reinterpret_cast<const f32x4*>(part) + i;
    f32x4 v[8];
#pragma unroll
    for (int z = 0; z < 8; ++z) v[z] = p[(long)min(z, splits - 1) * total4];
    f32x4 s = v[0];
#pragma unroll
    for (int z = 1; z < 8; ++z)
        if (z < splits) s += v[z];
    for (int z = 8; z < splits; ++z) s += p[(long)z * total4];
    if (bias) s += *reinterpret_cast<const f32x4*>(bias + (int)((i * 4) % ncols));
    if (addend) s += reinterpret_cast<const f32x4*>(addend)[i];
    if (relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
    reinterpret_cast<f32x4*>(out)[i] = s;
    return s;
}

__global__ __launch_bounds__(256) void conv3p_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                            const float* __restrict__ bias, const float* __restrict__ addend,
                                                            long total4, int ncols, int splits, int relu, float* __restrict__ stats)
{
    __shared__ f32x4 red_s[2][256];
    if (stats == nullptr) {                                 // (uniform: every thread of the grid takes the same side)
        const long i = (long)blockIdx.x * 256 + threadIdx.x;
        if (i < total4) conv3p_reduce_one(part, out, bias, addend, i, total4, ncols, splits, relu);
        return;
    }
    // with statistics: P3_RCH chunks of 256 float4 per workgroup; a thread's column (4 i mod ncols) is the same in every chunk
    // because 1024 is a multiple of ncols (a power of two <= 1024)
    f32x4 a{0.f, 0.f, 0.f, 0.f}, b{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < P3_RCH; ++c) {
        const long i = ((long)blockIdx.x * P3_RCH + c) * 256 + threadIdx.x;
        if (i < total4) {
            const f32x4 s = conv3p_reduce_one(part, out, bias, addend, i, total4, ncols, splits, relu);
            a += s; b += s * s;
        }
    }
    red_s[0][threadIdx.x] = a;
    red_s[1][threadIdx.x] = b;
    __syncthreads();
    const int c4 = ncols >> 2;                              // float4 columns; rows per chunk = 256 / c4
    if ((int)threadIdx.x < c4) {
        f32x4 sa = red_s[0][threadIdx.x], sb = red_s[1][threadIdx.x];
        for (int r = c4; r < 256; r += c4) { sa += red_s[0][r + threadIdx.x]; sb += red_s[1][r + threadIdx.x]; }
        float* q = stats + (size_t)blockIdx.x * 2 * ncols + threadIdx.x * 4;
        *reinterpret_cast<f32x4*>(q) = sa;
        *reinterpret_cast<f32x4*>(q + ncols) = sb;
    }
}

inline long cdiv(long a, long b) { return (a + b - 1) / b; }

int g_p3_target = 512;          // workgroups a launch is topped up to by split-K (tuning aid: phnet_conv3p_tune): two per CU, evenly
int g_p3_wide = 1;              // 128-column workgroup tile (64 x 64 per wave) where the output has >= 128 channels

struct P3Plan { long tiles; int fn, splits, units_per_split; };

// Measured on MI355X (tests/tools/bench_conv3p.py, one 5-frame clip): what matters at these sizes is that the workgroups of a
// launch spread EVENLY over the 256 CUs - 480 workgroups (two per CU, all resident) beat 640 (three on some CUs, two on others)
// by 8-15 % on layer3 / layer4 - so split-K tops the grid up to at most `target`, never beyond it.
P3Plan p3_plan(long M, int Ca, int Nn, size_t ws_bytes)
{
    P3Plan p;
    // the 128-column tile halves the A-side LDS traffic per MFMA but doubles the weight bytes a wave streams per step: measured
    // +4 % on layer2 (20000 pixels x 128), nothing on layer3, -15 % on layer4 (1250 x 512: few tiles, 14 MB of weights)
    p.fn = (g_p3_wide && Nn % 128 == 0 && M >= 16384) ? 2 : 1;
    p.tiles = cdiv(M, P3_BM) * (Nn / (64 * p.fn));
    const int units = 3 * (Ca / 16);
    int splits = 1;
    if (ws_bytes > 0 && p.tiles * 2 <= g_p3_target) {
        splits = (int)min((long)8, (long)g_p3_target / p.tiles);
        while (splits > 1 && units / splits < 4) --splits;                     // >= 4 units (192 of K) per split
        while (splits > 1 && (size_t)splits * M * Nn * sizeof(float) > ws_bytes) --splits;
    }
    p.units_per_split = (units + splits - 1) / splits;
    p.splits = (units + p.units_per_split - 1) / p.units_per_split;            // no empty split
    return p;
}

bool p3_applies(long M, int Ca, int Nn)
{
    return M >= 1 && Ca >= 16 && (Ca % 16) == 0 && Nn >= 64 && (Nn % 64) == 0 &&
           M * (long)Ca * 4 < 0x7fffffffL && (long)9 * Ca * Nn * 6 < 0x7fffffffL && M * (long)Nn * 4 < 0x7fffffffL * 4;
}

}  // namespace

// ---- C-ABI ---------------------------------------------------------------------------------------------------------
// Does the packed-weight 3x3 kernel take this shape?  (M = N*H*W pixels, Ca = A-side channels, Nn = output channels)
PHNET_API int phnet_conv3p_applies(int64_t M, int32_t Ca, int32_t Nn) { return p3_applies((long)M, Ca, Nn) ? 1 : 0; }

// bytes of the packed image of one 3x3 weight [Co][3][3][Ci]
PHNET_API uint64_t phnet_conv3p_packed_bytes(int32_t Co, int32_t Ci) { return (uint64_t)9 * Co * Ci * 6; }

// w OHWI [Co][3][3][Ci] f32 -> packed bf16 planes in fragment order; dgrad = 0: the forward's operand, 1: the data gradient's
PHNET_API int phnet_conv3p_pack(const float* w, void* packed, int32_t Co, int32_t Ci, int32_t dgrad, void* stream)
{
    if (!w || !packed || Co < 16 || Ci < 16) return PHNET_ERR_ARG;
    const int Ca = dgrad ? Co : Ci, Nn = dgrad ? Ci : Co;
    if ((Ca % 16) || (Nn % 32)) return PHNET_ERR_ARG;
    const long total = (long)9 * (Ca / 16) * (Nn / 32) * 64;
    hipLaunchKernelGGL(conv3p_pack_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, (unsigned char*)packed, Co, Ci, dgrad);
    return phnet_launch_status();
}

// jobs: DEVICE array of njobs records {w, dst, Co, Ci, dgrad, 0, first} (8 + 8 + 4 x 4 + 8 bytes; `first` = running sum of the
// jobs' item counts 9 * (Ca/16) * (Nn/32) * 64, total = their sum): every listed weight packed by one launch.
PHNET_API int phnet_conv3p_pack_jobs(const void* jobs, int32_t njobs, int64_t total, void* stream)
{
    if (!jobs || njobs < 1 || total < 1) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(conv3p_pack_jobs_kernel, dim3((unsigned)cdiv((long)total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const P3PackJob*)jobs, njobs, (long)total);
    return phnet_launch_status();
}

// rows of 2*Nn floats the `stats` output of phnet_conv3p_fwd needs (the `partial` layout of phnet_bn_finalize_partials)
PHNET_API uint64_t phnet_conv3p_stats_blocks(int64_t M, int32_t Ca, int32_t Nn, uint64_t ws_bytes)
{
    if (!p3_applies((long)M, Ca, Nn)) return 0;
    const P3Plan p = p3_plan((long)M, Ca, Nn, (size_t)ws_bytes);
    return (uint64_t)(p.splits > 1 ? cdiv((long)M * Nn / 4, 256 * P3_RCH) : cdiv((long)M, P3_BM) * (P3_BM / 64));
}

PHNET_API int phnet_conv3p_splits(int64_t M, int32_t Ca, int32_t Nn, uint64_t ws_bytes)
{
    if (!p3_applies((long)M, Ca, Nn)) return 0;
    return p3_plan((long)M, Ca, Nn, (size_t)ws_bytes).splits;
}

PHNET_API int phnet_conv3p_tune(int32_t target_workgroups)
{
    if (target_workgroups == -1 || target_workgroups == -2) { g_p3_wide = target_workgroups == -2; return PHNET_OK; }   // -1: 64-column tiles only
    if (target_workgroups < 1) return PHNET_ERR_ARG;
    g_p3_target = target_workgroups;
    return PHNET_OK;
}

// y = conv3x3(x, w) (+ bias) (+ addend) (ReLU) with w given PACKED (phnet_conv3p_pack, dgrad = 0); or, on the dgrad packing,
// dx = conv3x3_dgrad(dy, w) (+ addend).  x NHWC [N][H][W][Ca], y NHWC [N][H][W][Nn].  stats as phnet_conv2d_fwd_fused.
PHNET_API int phnet_conv3p_fwd(const float* x, const void* packed, const float* bias, const float* addend, float* y, float* stats,
                               int32_t N, int32_t H, int32_t W, int32_t Ca, int32_t Nn, int32_t relu,
                               void* workspace, uint64_t ws_bytes, void* stream)
{
    if (N < 0 || H < 1 || W < 2 || Ca < 16 || Nn < 64) return PHNET_ERR_ARG;
    if (N == 0) return PHNET_OK;
    const long M = (long)N * H * W;
    if (!x || !packed || !y || !p3_applies(M, Ca, Nn)) return PHNET_ERR_ARG;
    if (stats && (bias || addend || relu || (Nn & (Nn - 1)) || Nn > 1024)) return PHNET_ERR_ARG;
    const P3Plan p = p3_plan(M, Ca, Nn, workspace ? (size_t)ws_bytes : 0);
    P3Shape g{N, H, W, Ca, Nn, p.splits, p.units_per_split};
    float* dst = p.splits > 1 ? (float*)workspace : y;
    hipStream_t st = (hipStream_t)stream;
    if (p.fn == 2)
        hipLaunchKernelGGL(conv3p_kernel<2>, dim3((unsigned)p.tiles, 1, (unsigned)p.splits), dim3(P3_THREADS), P3_LDS, st,
                           x, (const unsigned char*)packed, bias, addend, dst, g, relu, p.splits > 1 ? (float*)nullptr : stats);
    else
        hipLaunchKernelGGL(conv3p_kernel<1>, dim3((unsigned)p.tiles, 1, (unsigned)p.splits), dim3(P3_THREADS), P3_LDS, st,
                           x, (const unsigned char*)packed, bias, addend, dst, g, relu, p.splits > 1 ? (float*)nullptr : stats);
    if (p.splits > 1) {
        const long total4 = M * Nn / 4;
        hipLaunchKernelGGL(conv3p_reduce_kernel, dim3((unsigned)cdiv(total4, stats ? 256 * P3_RCH : 256)), dim3(256), 0, st,
                           (const float*)workspace, y, bias, addend, total4, Nn, p.splits, relu, stats);
    }
    return phnet_launch_status();
}
