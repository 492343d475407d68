// Weight gradient of a Linear layer / 1x1 convolution over MANY rows with 128 x 128 tiles and producer / consumer wavefronts
// (gfx950, bf16x3 arithmetic):   dW[co][ci] (+)= sum over rows p of dY[p][co] * X[p][ci],   dbias[co] (+)= sum_p dY[p][co]
// (the backward of the lane head's hyper-network layers, utils/dynamic_head.py:31-59 - 1200 rows x 8192 x 1024 and 1200 x 1024 x 4608
// per stage with the frames of a clip batched).
//
// The generic kernel (conv_wgrad_kernel, 64 x 64 tiles) re-reads both operands once per 64 output columns: 1.26 GB of L2 traffic for
// the 8192 x 1024 layer, and its 114 TF/s ARE that bandwidth.  A 128 x 128 tile halves the traffic; what made the
// generic kernel's larger tiles slower - every wave splits and stages between its own MFMAs, few waves per CU - is answered the
// way csrc/wgrad3s.hip answers it: 8 consumer waves (64 x 32 each: two A fragments, one B fragment, 12 MFMAs per 16-row step) only read
// fragments and multiply, 4 producer waves (one per SIMD) only load, split into the three bf16 planes and fill the LDS stage two
// steps ahead, through a branch-free ring of four steps of global loads.  One barrier per step for all 12 waves.
// The bias gradient is summed by the producers of the first column tile on the dY blocks they stage anyway.
#include <type_traits>
#include "igemm.h"
#include "wgrad3s.h"

using namespace igemm;

namespace {

constexpr int L_NT = 768, L_CONSUMERS = 8;                   // 8 consumer + 4 producer waves
constexpr int L_PF = 4;                                      // producer register ring: K steps in flight per thread
constexpr int L_BKW = BK;                                    // rows per K step
constexpr int L_PITCH = KStridedPlanes<128, BK>::PITCH;      // 320 bytes: 128 bf16 + pad (igemm.h)
constexpr int L_PLANE = L_BKW * L_PITCH, L_IMG = 3 * L_PLANE, L_STAGE = 2 * L_IMG;      // dY image | X image
constexpr int L_BYTES = 3 * L_STAGE;

__global__ __launch_bounds__(L_NT) void wgrad1s_kernel(const float* __restrict__ dY, const float* __restrict__ X, float* __restrict__ out,
                                                       float* __restrict__ dbias, Wgrad1sShape g, int want_bias, int accumulate)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];               // [stage 0..2][dY | X][plane][row][col]
    const int P = g.P, Ci = g.Ci, Co = g.Co;
    const int tiles_n = Ci >> 7;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (int)(tile / tiles_n) * 128, c0 = (int)(tile % tiles_n) * 128;
    const int p_begin = blockIdx.z * g.rows_per_split, p_end = min(P, p_begin + g.rows_per_split);
    const int nsteps = p_begin < p_end ? (p_end - p_begin + L_BKW - 1) / L_BKW : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool direct = g.splits == 1;
    const bool bias_wg = want_bias && c0 == 0;               // (uniform) this workgroup also owns dbias[m0 .. m0 + 127]
    float* dst = direct ? out : out + (size_t)blockIdx.z * ((size_t)Co * Ci + Co);

    if (wave >= L_CONSUMERS) {
        // =============================== producers: global -> split -> LDS, two steps ahead ===============================
        const int p = tid - L_CONSUMERS * 64;                // 0 .. 255
        const int r0 = p >> 5, col = (p & 31) * 4;           // rows r0 and r0 + 8 of a step, 4 columns
        constexpr unsigned OOB = 0x80000000u;
        __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, (int)min((long)P * Co * 4, (long)0x7fffffff), 0x00020000);
        __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)min((long)P * Ci * 4, (long)0x7fffffff), 0x00020000);
        int t_row[2], a_off[2], b_off[2], st[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = r0 + 8 * i;
            t_row[i] = p_begin + row;
            a_off[i] = (t_row[i] * Co + m0 + col) * 4;
            b_off[i] = (t_row[i] * Ci + c0 + col) * 4;
            st[i] = row * L_PITCH + col * 2;
        }
        f32x4 ring[L_PF][4];
        f32x4 bsum[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        auto load_step = [&](f32x4 (&reg)[4]) {              // the next K step (calls go through the steps in order)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool ok = t_row[i] < p_end;
                reg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? a_off[i] : (int)OOB, 0, 0));
                reg[2 + i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, ok ? b_off[i] : (int)OOB, 0, 0));
                t_row[i] += L_BKW;
                a_off[i] += L_BKW * Co * 4;
                b_off[i] += L_BKW * Ci * 4;
            }
        };
        auto store_step = [&](int stage_off, const f32x4 (&reg)[4]) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bsum[i] += reg[i];
                store_split3<L_PLANE>(lds_raw, stage_off + st[i], reg[i]);
                store_split3<L_PLANE>(lds_raw, stage_off + L_IMG + st[i], reg[2 + i]);
            }
        };
        if (nsteps > 0) {
#pragma unroll
            for (int d = 0; d < L_PF; ++d) load_step(ring[d]);
            store_step(0, ring[0]);
            load_step(ring[0]);
            store_step(L_STAGE, ring[1]);
            load_step(ring[1]);
            __syncthreads();
            int o_st = 2 * L_STAGE;
            // iteration t stages step t + 2 out of ring[(t + 2) % PF] and reloads that entry with step t + 2 + PF; the main loop is
            // free of conditions (a branch around a load makes hipcc drain the ring: s_waitcnt vmcnt(0), csrc/wgrad3s.hip)
            auto body = [&](auto U, bool reload) {
                constexpr int slot = (decltype(U)::value + 2) % L_PF;
                store_step(o_st, ring[slot]);
                if (reload) load_step(ring[slot]);          // past the end: masked
                __syncthreads();
                o_st = o_st == 2 * L_STAGE ? 0 : o_st + L_STAGE;
            };
            int t = 0;
            for (; t + L_PF <= nsteps; t += L_PF) {
                body(std::integral_constant<int, 0>{}, true);
                body(std::integral_constant<int, 1>{}, true);
                body(std::integral_constant<int, 2>{}, true);
                body(std::integral_constant<int, 3>{}, true);
            }
            static_assert(L_PF == 4, "the unrolled ring walk above");
            if (t < nsteps) body(std::integral_constant<int, 0>{}, false);
            if (t + 1 < nsteps) body(std::integral_constant<int, 1>{}, false);
            if (t + 2 < nsteps) body(std::integral_constant<int, 2>{}, false);
        }
        if (bias_wg) {
            // the two stores past the last step added rows that no step multiplies (zeros or the next split's rows): the sums of
            // rows >= p_end are masked at the load, the rows of steps nsteps, nsteps + 1 were staged but belong to nobody - take them out
            // (they were loaded with the mask `row < p_end`, i.e. they ARE zeros: nothing to correct)
            float* fold = reinterpret_cast<float*>(lds_raw);          // [16 row classes][128]
            __syncthreads();                                          // every consumer has read its last fragments
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(fold + (r0 + 8 * i) * 128 + col) = bsum[i];
            __syncthreads();
            if (p < 128) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) s += fold[k * 128 + p];
                if (direct) dbias[m0 + p] = accumulate ? dbias[m0 + p] + s : s;
                else dst[(size_t)Co * Ci + m0 + p] = s;
            }
        }
        return;
    }

    // ======================================= consumers: fragments -> MFMAs =======================================
    const int wm = (wave >> 2) * 64, wn = (wave & 3) * 32;
    const bool from_old = direct && accumulate;
    f32x16 acc[2][1];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = c0 + wn + frag_col(lane), m = m0 + wm + 32 * f + frag_row(lane, e);
            const float v = out[from_old ? (size_t)m * Ci + n : 0];
            acc[f][0][e] = from_old ? v : 0.f;
        }
    if (nsteps > 0) {
        Frag3 fa[2][2], fb[2][1];
        const unsigned char* a_src = lds_raw + wm * 2;
        const unsigned char* b_src = lds_raw + L_IMG + wn * 2;
        auto read_frags = [&](int stage_off, Frag3 (&a)[2], Frag3 (&b)[1]) {
            read_kstrided3<2, L_PITCH, L_PLANE>(a_src + stage_off, lane, 0, a);
            read_kstrided3<1, L_PITCH, L_PLANE>(b_src + stage_off, lane, 0, b);
        };
        int o_nxt = L_STAGE;
        __syncthreads();                                     // steps 0 and 1 are staged
        read_frags(0, fa[0], fb[0]);
        auto step = [&](auto U) {
            constexpr int cur = decltype(U)::value, nxt = cur ^ 1;
            read_frags(o_nxt, fa[nxt], fb[nxt]);
            mma3_step<2, 1>(fa[cur], fb[cur], acc);
            __syncthreads();
            o_nxt = o_nxt == 2 * L_STAGE ? 0 : o_nxt + L_STAGE;
        };
        int t = 0;
        for (; t + 2 <= nsteps; t += 2) {
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
        }
        if (t < nsteps) step(std::integral_constant<int, 0>{});
    }
    if (bias_wg) {                                           // the producers' fold of the bias sums: same two barriers for every wave
        __syncthreads();
        __syncthreads();
    }
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            dst[(size_t)(m0 + wm + 32 * f + frag_row(lane, e)) * Ci + c0 + wn + frag_col(lane)] = acc[f][0][e];
}

}  // namespace

int phnet_wgrad1s_kstep() { return L_BKW; }

int phnet_wgrad1s_launch(const float* dy, const float* x, float* out, float* dbias, Wgrad1sShape g, int accumulate, hipStream_t st)
{
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)wgrad1s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, L_BYTES) != hipSuccess)
            return PHNET_ERR_LAUNCH;
        attr = true;
    }
    dim3 grid((unsigned)((g.Co / 128) * (g.Ci / 128)), 1, (unsigned)g.splits);
    hipLaunchKernelGGL(wgrad1s_kernel, grid, dim3(L_NT), L_BYTES, st, dy, x, out, dbias, g, dbias != nullptr, accumulate);
    return phnet_launch_status();
}
