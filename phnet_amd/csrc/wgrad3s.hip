// Weight gradient of a 3x3 / stride 1 / pad 1 convolution with PRODUCER and CONSUMER wavefronts (gfx950, bf16x3 arithmetic).
//
//   dW[co][ky][kx][ci] = sum over pixels p of dY[p][co] * X[p + (ky-1) W + (kx-1)][ci]          (libs/models/resnet.py:79-95 backward)
//
// Same tile program as conv_wgrad3x3_kernel (conv.hip): one workgroup = (64 co) x (64 ci) x (the three taps of filter row ky),
// per K step it stages dY[16 px][64 co] and X[18 px][64 ci] as three bf16 planes each (x = hi + mid + lo exactly) and twelve
// wavefronts - tap group dx = wave / 4, quadrant wave % 4 - multiply 32x32 blocks out of them with transposed LDS reads.
//
// What changed is WHO stages.  There all twelve waves loaded, split and stored their share of the next blocks between their
// six MFMAs of a step, and the step was bound by vector-instruction ISSUE: per SIMD and step 18 MFMAs (576 cycles of matrix
// pipe) against ~250 other instructions of 4 cycles each - address / validity arithmetic and the 22-instruction split per
// float4, four waves of the twelve running all of it on dummy data to keep the loop one basic block - with every wave in the same
// phase at the same time (one barrier per step), so that the phases added up instead of overlapping: ~1600 cycles per step, matrix
// pipe 23-29 % busy (profiles/r03_pmc_mfma_trunk.json).
// Here four MORE waves (12-15, one per SIMD) do nothing but stage: they run two steps ahead of the consumers through a register
// ring of global loads (4 steps of 32 pixels in flight), and the twelve consumer waves only read fragments and issue MFMAs.
// Producers and consumers meet at the same one barrier per step, but a SIMD now has MFMAs of three waves and the staging stream of
// a fourth to pick from.  Measured on the four trunk shapes of a clip, reduce included (tests/tools/bench_wgrad3.py, one box):
// 49-55 us -> 37-46 us per launch (108-120 -> 127-158 TF/s).  What the what-if builds of this kernel showed on the way:
//   * no split arithmetic AND no MFMAs still costs 28-29 us: ~10 us of launch / prologue / partial sums / reduce and a loop that
//     is bound by the LDS pipe (twelve waves x 24 transposed reads per 32-pixel step); the split adds ~6 us and the MFMAs ~8 us ON
//     TOP of that floor - the three only partly overlap;
//   * a producer loop with a condition around its loads makes the compiler drain the whole ring before every split
//     (s_waitcnt vmcnt(0)): the main loop below is branch-free, the last < 4 steps run without reloading;
//   * 32-pixel steps (half the barriers) -4 us; this source is built with -fno-slp-vectorize (phnet_amd/build.py): the packed
//     v_pk_add_f32 the vectoriser makes of the split costs the MFMA streams of the same SIMD more than two plain adds, -6 us;
//   * consumers passing the barrier with their fragment reads still in flight (s_barrier without the LDS wait): no change.
// The X rows 32, 33 of a step (its right halo) are rows 0, 1 of the next step: the lanes that hold those store them twice.
// Image rows y + dy outside the frame are staged as zeros; the pixel whose tap dx leaves its image row (x = 0 for dx = -1,
// x = W-1 for dx = +1: at most one per 16-pixel block, W >= 16) is cleared in the dY fragment of that tap group.
#include <type_traits>
#include "igemm.h"
#include "wgrad3s.h"

using namespace igemm;

namespace {

constexpr int S_NT = 1024, S_CONSUMERS = 12;                 // 12 consumer + 4 producer waves
constexpr int S_PF = 4;                                      // producer register ring: K steps in flight per thread (what bounds the step is
                                                             // load latency / steps in flight: 4 -> 8 steps, see wgrad3s_kernel)

template <int NSUB> struct S3Lds {
    static constexpr int BKW = BK * NSUB;                    // pixels per K step
    static constexpr int ROWS = BKW + 2;                     // X rows of a step: pixels pt-1 .. pt+BKW of the shifted image row
    static constexpr int PITCH = KStridedPlanes<64, BK>::PITCH;      // 192 bytes: 64 bf16 + pad (igemm.h)
    static constexpr int PLANE = ROWS * PITCH, IMG = 3 * PLANE;      // both operands use the ROWS-row image (dY leaves two rows unused)
    static constexpr int STAGE = 2 * IMG;                    // dY image | X image
    static constexpr int DUMP = 3 * STAGE;                   // where the idle lanes of the halo wave store
    static constexpr int BYTES = 3 * STAGE + 2 * PLANE + 512;
};

template <int NSUB>
__global__ __launch_bounds__(S_NT) void wgrad3s_kernel(const float* __restrict__ dY, const float* __restrict__ X, float* __restrict__ out,
                                                       Wgrad3sShape g, int accumulate)
{
    typedef S3Lds<NSUB> L;
    constexpr int BKW = L::BKW, PITCH = L::PITCH, PLANE = L::PLANE, IMG = L::IMG, STAGE = L::STAGE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];               // [stage 0..2][dY | X][plane][row][col], dump

    const int NC = 9 * g.Ci, W = g.W, HW = g.H * g.W;
    const int P = g.N * HW;
    const int ctiles = g.Ci >> 6;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int dyi = (int)(tile % 3), ct = (int)((tile / 3) % ctiles), mt = (int)(tile / (3 * ctiles));
    const int m0 = mt * 64, c0 = ct * 64, dy = dyi - 1;
    const int p_begin = blockIdx.z * g.pix_per_split, p_end = min(P, p_begin + g.pix_per_split);
    const int nsteps = p_begin < p_end ? (p_end - p_begin + BKW - 1) / BKW : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    if (wave >= S_CONSUMERS) {
        // =============================== producers: global -> split -> LDS, two steps ahead ===============================
        const int p = tid - S_CONSUMERS * 64;                // 0 .. 255
        const int r0 = p >> 4, col = (p & 15) * 4;
        const bool halo_wave = wave == S_CONSUMERS;          // its first 32 lanes also stage the X rows BKW, BKW + 1
        constexpr unsigned OOB = 0x80000000u;
        __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, (int)min((long)P * g.Co * 4, (long)0x7fffffff), 0x00020000);
        __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)min((long)P * g.Ci * 4, (long)0x7fffffff), 0x00020000);
        const int lo = dy < 0 ? W : 0;                       // aligned pixel t (image-local index timg) has its row y + dy inside the
        const unsigned span = (unsigned)(dy != 0 ? HW - W : HW);       // frame iff lo <= timg < lo + span
        // dY slots: pixel pt + r0 + 16 i, channels m0 + col ..;  X slots: aligned pixel t = pt - 1 + row, source pixel t + dy W
        int a_off[NSUB], a_t[NSUB], b_off[NSUB], b_t[NSUB], b_img[NSUB];
        int a_st[NSUB], b_st[NSUB];
#pragma unroll
        for (int i = 0; i < NSUB; ++i) {
            const int row = r0 + 16 * i;
            a_t[i] = p_begin + row;
            a_off[i] = (a_t[i] * g.Co + m0 + col) * 4;
            a_st[i] = row * PITCH + col * 2;
            b_t[i] = p_begin - 1 + row;
            b_off[i] = ((b_t[i] + dy * W) * g.Ci + c0 + col) * 4;
            b_img[i] = (b_t[i] + HW) % HW;
            b_st[i] = IMG + row * PITCH + col * 2;
        }
        // X rows BKW, BKW + 1 of a step are rows 0, 1 of the NEXT step: the lanes that hold those (p < 32) store them a second time
        const bool halo_lane = p < 32;
        const int halo_st = halo_lane ? IMG + (BKW + r0) * PITCH + col * 2 : L::DUMP + lane * 8;

        f32x4 ring[S_PF][2 * NSUB];
        // loads the next K step (calls go through the steps in order)
        auto load_step = [&](f32x4 (&reg)[2 * NSUB]) {
#pragma unroll
            for (int i = 0; i < NSUB; ++i) {
                reg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, a_t[i] < p_end ? a_off[i] : (int)OOB, 0, 0));
                a_t[i] += BKW;
                a_off[i] += BKW * g.Co * 4;
            }
#pragma unroll
            for (int i = 0; i < NSUB; ++i) {
                const bool ok = (unsigned)b_t[i] < (unsigned)P && (unsigned)(b_img[i] - lo) < span;
                reg[NSUB + i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, ok ? b_off[i] : (int)OOB, 0, 0));
                b_t[i] += BKW;
                b_off[i] += BKW * g.Ci * 4;
                b_img[i] += BKW;
                b_img[i] -= b_img[i] >= HW ? HW : 0;           // HW >= BKW: one wrap at most
            }
        };
        auto store_step = [&](int stage_off, const f32x4 (&reg)[2 * NSUB], const f32x4 (&next)[2 * NSUB]) {
#pragma unroll
            for (int i = 0; i < NSUB; ++i) store_split3<PLANE>(lds_raw, stage_off + a_st[i], reg[i]);
#pragma unroll
            for (int i = 0; i < NSUB; ++i) store_split3<PLANE>(lds_raw, stage_off + b_st[i], reg[NSUB + i]);
            if (halo_wave) store_split3<PLANE>(lds_raw, halo_st + (halo_lane ? stage_off : 0), next[NSUB]);
        };
        if (nsteps > 0) {
#pragma unroll
            for (int d = 0; d < S_PF; ++d) load_step(ring[d]);
            store_step(0, ring[0], ring[1]);
            load_step(ring[0]);
            store_step(STAGE, ring[1], ring[2]);
            load_step(ring[1]);
            __syncthreads();
            int o_st = 2 * STAGE;
            // iteration t stages step t + 2 out of ring[(t + 2) % PF] and reloads that entry with step t + 2 + PF.  The main loop is
            // free of conditions: with a branch around a load the compiler can no longer count the loads in flight and drains
            // the whole ring before every split (s_waitcnt vmcnt(0): measured, the ring then hides nothing)
            auto body = [&](auto U, bool reload) {
                constexpr int slot = (decltype(U)::value + 2) % S_PF;
                store_step(o_st, ring[slot], ring[(slot + 1) % S_PF]);
                if (reload) load_step(ring[slot]);          // past the end: dY masked
                __syncthreads();
                o_st = o_st == 2 * STAGE ? 0 : o_st + STAGE;
            };
            int t = 0;
            for (; t + S_PF <= nsteps; t += S_PF) {
                body(std::integral_constant<int, 0>{}, true);
                body(std::integral_constant<int, 1>{}, true);
                body(std::integral_constant<int, 2>{}, true);
                body(std::integral_constant<int, 3>{}, true);
            }
            static_assert(S_PF == 4, "the unrolled ring walk above");
            if (t < nsteps) body(std::integral_constant<int, 0>{}, false);           // the steps these would load are never used
            if (t + 1 < nsteps) body(std::integral_constant<int, 1>{}, false);
            if (t + 2 < nsteps) body(std::integral_constant<int, 2>{}, false);
        }
        return;
    }

    // ======================================= consumers: fragments -> MFMAs =======================================
    const int tg = wave >> 2, dx = tg - 1;                   // tap group of this wave
    const int wm = ((wave >> 1) & 1) * 32, wn = (wave & 1) * 32;
    const bool from_old = g.splits == 1 && accumulate;
    const int n_base = (dyi * 3 + tg) * g.Ci + c0 + wn;      // first column of this wave's block in [Co][9 Ci]
    f32x16 acc[1][1];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int n = n_base + frag_col(lane), m = m0 + wm + frag_row(lane, e);
        const float v = out[from_old ? (size_t)m * NC + n : 0];
        acc[0][0][e] = from_old ? v : 0.f;
    }
    if (nsteps > 0) {
        Frag3 fa[2], fb[2];
        const unsigned char* a_src = lds_raw + wm * 2;
        const unsigned char* b_src = lds_raw + IMG + (dx + 1) * PITCH + wn * 2;
        auto read_frags = [&](int stage_off, int ks, Frag3& a, Frag3& b) {
            Frag3 (&a1)[1] = *reinterpret_cast<Frag3 (*)[1]>(&a);
            Frag3 (&b1)[1] = *reinterpret_cast<Frag3 (*)[1]>(&b);
            read_kstrided3<1, PITCH, PLANE>(a_src + stage_off, lane, ks, a1);
            read_kstrided3<1, PITCH, PLANE>(b_src + stage_off, lane, ks, b1);
        };
        const int h8 = (lane >> 5) * 8;
        // clears, in a dY fragment, pixel ke of its 16-pixel block (this lane holds k = h8 .. h8 + 7)
        auto mask_edge = [&](Frag3& a, int ke) {
            const int j = ke - h8;
            const int ji = j >> 1;                            // register of the element (other half wave: none)
            const unsigned wmask = (j & 1) ? 0x0000ffffu : 0xffff0000u;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 m;
            m.x = ji == 0 ? wmask : 0xffffffffu; m.y = ji == 1 ? wmask : 0xffffffffu;
            m.z = ji == 2 ? wmask : 0xffffffffu; m.w = ji == 3 ? wmask : 0xffffffffu;
            a.hi = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, a.hi) & m);
            a.mid = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, a.mid) & m);
            a.lo = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, a.lo) & m);
        };
        int mx = p_begin % W;                                // image column of the first pixel of the sub-step whose fragment is masked next
        int o_cur = 0, o_nxt = STAGE;
        __syncthreads();                                     // steps 0 and 1 are staged
        read_frags(0, 0, fa[0], fb[0]);
        auto step = [&](auto U) {
            constexpr int u = decltype(U)::value;             // parity of the step
#pragma unroll
            for (int ks = 0; ks < NSUB; ++ks) {
                const int cur = (u * NSUB + ks) & 1, nxt = cur ^ 1;
                if (ks + 1 < NSUB) read_frags(o_cur, ks + 1, fa[nxt], fb[nxt]);
                else read_frags(o_nxt, 0, fa[nxt], fb[nxt]);
                {
                    const int ke = dx < 0 ? (mx == 0 ? 0 : W - mx) : W - 1 - mx;
                    if (dx != 0 && ke < BK) mask_edge(fa[cur], ke);
                    mx += BK;
                    mx -= mx >= W ? W : 0;
                }
                Frag3 (&a)[1] = *reinterpret_cast<Frag3 (*)[1]>(&fa[cur]);
                Frag3 (&b)[1] = *reinterpret_cast<Frag3 (*)[1]>(&fb[cur]);
                mma3_step<1, 1>(a, b, acc);
            }
            __syncthreads();
            o_cur = o_nxt;
            o_nxt = o_nxt == 2 * STAGE ? 0 : o_nxt + STAGE;
        };
        int t = 0;
        for (; t + 2 <= nsteps; t += 2) {
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, (NSUB & 1)>{});
        }
        if (t < nsteps) step(std::integral_constant<int, 0>{});
    }
    const bool direct = g.splits == 1;
    float* dst = direct ? out : out + (size_t)blockIdx.z * ((size_t)g.Co * NC + g.Co);
#pragma unroll
    for (int e = 0; e < 16; ++e)
        dst[(size_t)(m0 + wm + frag_row(lane, e)) * NC + n_base + frag_col(lane)] = acc[0][0][e];
}

constexpr int S_NSUB = 2;

}  // namespace

int phnet_wgrad3s_kstep() { return BK * S_NSUB; }

int phnet_wgrad3s_launch(const float* dy, const float* x, float* out, Wgrad3sShape g, int accumulate, hipStream_t st)
{
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)wgrad3s_kernel<S_NSUB>, hipFuncAttributeMaxDynamicSharedMemorySize, S3Lds<S_NSUB>::BYTES) != hipSuccess)
            return PHNET_ERR_LAUNCH;
        attr = true;
    }
    dim3 grid((unsigned)((g.Co / 64) * (g.Ci / 64) * 3), 1, (unsigned)g.splits);
    hipLaunchKernelGGL((wgrad3s_kernel<S_NSUB>), grid, dim3(S_NT), S3Lds<S_NSUB>::BYTES, st, dy, x, out, g, accumulate);
    return phnet_launch_status();
}
