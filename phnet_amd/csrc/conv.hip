// Implicit-GEMM convolution / linear kernels on fp32 MFMA for gfx950 (NHWC activations, OHWI weights).
//
// Replaces, for the hot path of SURVEY.md 8(a): the cuDNN convolutions behind
//   libs/models/resnet.py:79-95,293-307 (trunk), libs/models/fpn.py:109-163 (neck)
// and the cuBLAS addmm behind every nn.Linear of the head (Router4OL.py:308-392,
// utils/dynamic_head.py:31-59, Router.py:72-81), forward, data-gradient and weight-gradient.
//
//   forward : out[m][co] = sum_{r,q,c} X[n, oy*s-p+r, ox*s-p+q, c] * W[co][r][q][c]      (+bias)(+relu)
//             GEMM  M = N*Ho*Wo pixels, N = Co, K = R*S*Ci;  A gathered on the fly (K-contiguous),
//             B = W as stored (K-contiguous).
//   dgrad   : dX[m][c] = sum_{r,q,co} dY[n, (y+p-r)/s, (x+p-q)/s, co] * W[co][r][q][c]
//             same kernel: A gathered from dY with "input dilation" s, B read K-strided straight out
//             of the OHWI weights (row (r,q,co) -> W[co][R-1-r'][S-1-q'][:]) - no weight transpose pass.
//   wgrad   : dW[co][r][q][c] = sum_pixels dY[pix][co] * X[pix shifted by (r,q)][c]
//             GEMM  M = Co, N = R*S*Ci, K = pixels (split over blockIdx.z, deterministic 2-pass reduce);
//             both operands K-strided.
// A Linear layer is the R=S=1 case on an [M,1,1,K] image.
//
// Roofline: MFMA-bound (f32-input MFMA, 157.3 TF/s dense).  Algorithmic FLOPs = 2*M*N*K.
#include <type_traits>
#include "igemm.h"
#include "wgrad3s.h"
static constexpr bool g_interleave = true;                   // MFMA / VALU interleave hints of the staged-split loops

using namespace igemm;

namespace {

struct ConvShape {
    int N, Hi, Wi, Ci;       // A-side image (input for fwd, dY for dgrad)
    int Ho, Wo, Co;          // GEMM output image
    int R, S;
    int stride, pad, in_dil; // coordinate: t = o*stride - pad + r ; valid iff t % in_dil == 0 ; i = t / in_dil
    int splits;              // split-K factor (blockIdx.z)
    int k_per_split;         // multiple of BK
};

struct RowCoord { int base, iy0, ix0; bool ok; };

// f(integral_constant<0>, t), f(integral_constant<1>, t + 1), ... : a loop body that needs its index at compile time
template <int N, int I = 0, typename F>
__device__ __forceinline__ void unroll_iterations(F& f, int t) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{}, t + I);
        unroll_iterations<N, I + 1>(f, t);
    }
}
template <int N, int I = 0, typename F>
__device__ __forceinline__ void tail_iterations(F& f, int t, int end) {
    if constexpr (I < N) {
        if (t + I < end) {
            f(std::integral_constant<int, I>{}, t + I);
            tail_iterations<N, I + 1>(f, t, end);
        }
    }
}

// The tile program is a device function so that one launch can run tiles of different GEMMs (linear_bwd_fused_kernel);
// (block, nblocks, split) are what blockIdx.x / gridDim.x / blockIdx.z are for the plain kernel below.
// AMASK: the A operand is a gradient that still has to pass a ReLU - element (row, k) counts only where amask (the saved
// forward output of that ReLU, same layout as X) is positive; applied when the tile is written to LDS.
// PF: register prefetch depth in K tiles.  The loads of tile t + PF are issued while tile t is multiplied, so PF tiles of a
// workgroup are in flight at any time: at one clip the trunk GEMMs are bound by the latency of their operand loads times
// the little that is in flight per CU (measured: 9 B/clk/CU at PF = 1 against the ~28 B/clk/CU an L2-resident gather
// sustains, MI355X_MICROARCH.md "Indexed rows"), not by the MFMA pipe - which is why the 2.7x cheaper bf16x3 arithmetic
// alone changed nothing.  A tile costs 2-4 float4 registers per thread, the LDS double buffer stays.
// BUF: operand loads through buffer instructions with 32-bit offsets (uniform-tap, undilated, unmasked-A problems only): an
// out-of-range element is simply given an offset past the end of the tensor and the hardware returns zeros - no 64-bit
// address arithmetic, no validity select when the tile is written to LDS (the loop is bound by its vector instruction
// count: 90 per 6 MFMAs before, of which 44 are the split).
template <int BM, int BN, bool B_DGRAD, int BKT, bool UNI, bool AMASK = false, int MMA = 0, int PF = 1, bool BUF = false>
__device__ __forceinline__ void igemm_tile(
    const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ bias,
    const float* __restrict__ addend, float* __restrict__ out, const ConvShape& g, int relu,
    float* lds, unsigned block, unsigned nblocks, unsigned split, const float* __restrict__ amask = nullptr,
    float* __restrict__ stats = nullptr)
{
    constexpr int TM = BM / 2, TN = BN / 2, FM = TM / 32, FN = TN / 32;
    constexpr int A_PITCH = KContigTile<BM, BKT>::PITCH;
    constexpr int A_FLOATS = KContigTile<BM, BKT>::FLOATS;
    constexpr int B_PITCH = B_DGRAD ? KStridedTile<BN, BKT>::PITCH : KContigTile<BN, BKT>::PITCH;
    constexpr int B_FLOATS = B_DGRAD ? KStridedTile<BN, BKT>::FLOATS : KContigTile<BN, BKT>::FLOATS;
    constexpr int CHUNKS = BKT / 4;                         // float4 chunks along K per row
    constexpr int ROWS_PER_PASS = THREADS / CHUNKS;         // 64 (BKT 16) or 16 (BKT 64)
    constexpr int A_LOADS = BM / ROWS_PER_PASS;             // float4 loads per thread per K tile
    constexpr int B_LOADS = B_DGRAD ? (BKT * BN / 4) / THREADS : BN / ROWS_PER_PASS;
    float* As = lds;
    float* Bs = lds + 2 * A_FLOATS;
    // MMA = 3: bf16 planes instead of f32 images (igemm.h)
    constexpr bool S3 = MMA == 3;
    typedef KContigPlanes<BM, BKT> AP3;
    typedef KContigPlanes<BN, BKT> BP3C;
    typedef KStridedPlanes<BN, BKT> BP3S;
    constexpr int B3_BYTES = B_DGRAD ? BP3S::BYTES : BP3C::BYTES;
    unsigned char* A3 = reinterpret_cast<unsigned char*>(lds);
    unsigned char* B3 = A3 + 2 * AP3::BYTES;

    const int M = g.N * g.Ho * g.Wo;
    const int K = g.R * g.S * g.Ci;
    const int tiles_n = (g.Co + BN - 1) / BN;
    const unsigned tile = xcd_remap(block, nblocks);
    const int m0 = (int)(tile / tiles_n) * BM, n0 = (int)(tile % tiles_n) * BN;
    const int k_begin = split * g.k_per_split;
    const int k_end = min(K, k_begin + g.k_per_split);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * TM, wn = (wave & 1) * TN;

    // ---- per-thread gather coordinates of the A rows this thread stages (fixed over the K loop) ----
    RowCoord rc[A_LOADS];
    const int a_chunk = tid % CHUNKS, a_row = tid / CHUNKS;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int m = m0 + a_row + ROWS_PER_PASS * i;
        rc[i].ok = m < M;
        const int mm = rc[i].ok ? m : 0;
        const int n = mm / (g.Ho * g.Wo), rem = mm - n * (g.Ho * g.Wo);
        const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
        rc[i].base = n * g.Hi * g.Wi;
        rc[i].iy0 = oy * g.stride - g.pad;
        rc[i].ix0 = ox * g.stride - g.pad;
    }

    // ---- running (tap, channel) decomposition of this thread's k index: one division at start-up, then
    // carry-propagating adds per K tile (per-step integer divisions made the loop VALU-bound) ----------
    struct KPos { int c, r, q; };
    auto kpos_init = [&](int k, int ci) {
        KPos p;
        const int rs = k / ci;
        p.c = k - rs * ci;
        p.r = rs / g.S;
        p.q = rs - p.r * g.S;
        return p;
    };
    auto kpos_advance = [&](KPos& p, int ci) {
        p.c += BKT;
        while (p.c >= ci) {
            p.c -= ci;
            if (++p.q == g.S) { p.q = 0; ++p.r; }
        }
    };
    KPos ka = kpos_init(k_begin + a_chunk * 4, g.Ci);
    KPos kb[B_LOADS];
    int b_kk[B_LOADS], b_ch[B_LOADS];
    if (B_DGRAD) {
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int idx = tid + THREADS * i;                     // BKT rows x BN/4 chunks
            b_kk[i] = idx / (BN / 4);
            b_ch[i] = idx - b_kk[i] * (BN / 4);
            kb[i] = kpos_init(k_begin + b_kk[i], g.Ci);
        }
    }

    // Uniform-tap fast path: when the A-side channel count is a multiple of the K tile, every K tile lies inside ONE filter
    // tap (r,q), the same for the whole workgroup, so the tap walks in scalar registers and a thread's address math per
    // tile is a handful of VALU ops (the general path below carries a per-thread (c,r,q) decomposition and made the
    // trunk loops VALU-bound: ~190 instructions per 8 MFMAs).
    constexpr bool uni = UNI;                            // host guarantees g.Ci % BKT == 0
    int ur = 0, uq = 0, uc0 = 0;
    if (uni) {
        const int rs = k_begin / g.Ci;
        uc0 = k_begin - rs * g.Ci;
        ur = rs / g.S;
        uq = rs - ur * g.S;
    }

    // The loads are BRANCH-FREE: an out-of-range element reads element 0 of its array and is zeroed when it is written to
    // LDS (validity bits travel in `mask`: A loads in bits 0.., B loads in bits 16..), so the loop body is straight-line
    // code the compiler can schedule and count (s_waitcnt) exactly.  (A second register set prefetching two tiles ahead
    // was measured too: +4 % on the trunk forward, -3 % on the skinny head GEMMs, no gain on the step - not kept.)
    f32x4 a_set[PF][A_LOADS], b_set[PF][B_LOADS];
    f32x4 r_set[PF][AMASK ? A_LOADS : 1];
    unsigned mask_set[PF];
    // loads the K tile that starts at kt; MUST be called with kt = k_begin, k_begin+BKT, ... in order
    // ---- BUF: per-thread byte offsets fixed over the K loop + one scalar offset per tile ----
    constexpr unsigned OOB = 0x80000000u;                 // past the end of every tensor of this path (< 2 GB each)
    int a_off[BUF ? A_LOADS : 1], b_off[BUF ? B_LOADS : 1];
    __amdgpu_buffer_rsrc_t x_rsrc, w_rsrc;
    if constexpr (BUF) {
        static_assert(UNI && !AMASK, "buffer-load path: uniform tap, no ReLU mask");
        x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)min((long)g.N * g.Hi * g.Wi * g.Ci * 4, (long)0x7fffffff), 0x00020000);
        w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)min((long)g.Co * K * 4, (long)0x7fffffff), 0x00020000);
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            a_off[i] = ((rc[i].base + rc[i].iy0 * g.Wi + rc[i].ix0) * g.Ci + a_chunk * 4) * 4;
            if (!rc[i].ok) rc[i].iy0 = -(1 << 28);                       // fails the row test below for every tap
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            if (!B_DGRAD) {
                const int n = n0 + a_row + ROWS_PER_PASS * i;
                b_off[i] = n < g.Co ? (n * K + a_chunk * 4) * 4 : (int)OOB;
            } else {
                const int n = n0 + b_ch[i] * 4;
                b_off[i] = n < g.Co ? (b_kk[i] * g.R * g.S * g.Co + n) * 4 : (int)OOB;
            }
        }
    }
    auto load_global = [&](int kt, f32x4 (&a_reg)[A_LOADS], f32x4 (&b_reg)[B_LOADS], f32x4 (&a_relu)[AMASK ? A_LOADS : 1], unsigned& mask) {
        mask = 0;
        if constexpr (BUF) {
            const bool tile_ok = kt < k_end;                             // uniform: K and the split bounds are multiples of BKT
            const int s_tap = ((ur * g.Wi + uq) * g.Ci + uc0) * 4;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const bool ok = tile_ok && (unsigned)(rc[i].iy0 + ur) < (unsigned)g.Hi && (unsigned)(rc[i].ix0 + uq) < (unsigned)g.Wi;
                a_reg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, ok ? a_off[i] + s_tap : (int)OOB, 0, 0));
            }
            // B: scalar part of the offset
            int s_b;
            if (!B_DGRAD) s_b = kt * 4;
            else s_b = ((uc0 * g.R + (g.R - 1 - ur)) * g.S + (g.S - 1 - uq)) * g.Co * 4;
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i)            // (an invalid column already sits at OOB; adding the small s_b keeps it there)
                b_reg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, tile_ok ? b_off[i] + s_b : (int)OOB, 0, 0));
            uc0 += BKT;
            const int wc = uc0 >= g.Ci;
            uc0 = wc ? 0 : uc0;
            uq += wc;
            const int wq = uq == g.S;
            uq = wq ? 0 : uq;
            ur += wq;
            return;
        }
        // A: CHUNKS consecutive lanes fetch BKT*4 contiguous bytes of one pixel tap
        const int k0 = kt + a_chunk * 4;
        const bool kok = k0 < k_end;
        const int tap_r = uni ? ur : ka.r, tap_q = uni ? uq : ka.q, tap_c = uni ? uc0 + a_chunk * 4 : ka.c;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            int ty = rc[i].iy0 + tap_r, tx = rc[i].ix0 + tap_q;
            bool ok = kok && rc[i].ok && ty >= 0 && tx >= 0;
            if (g.in_dil == 2) {                        // dgrad of a stride-2 conv (strides > 2 are rejected on the host)
                ok = ok && !((ty | tx) & 1);
                ty >>= 1; tx >>= 1;
            }
            ok = ok && ty < g.Hi && tx < g.Wi;
            const int pix = ok ? rc[i].base + ty * g.Wi + tx : 0;         // masked: pixel 0 (valid memory, value discarded)
            a_reg[i] = *reinterpret_cast<const f32x4*>(X + (size_t)pix * g.Ci + (kok ? tap_c : 0));
            if (AMASK) a_relu[i] = *reinterpret_cast<const f32x4*>(amask + (size_t)pix * g.Ci + (kok ? tap_c : 0));
            mask |= (unsigned)ok << i;
        }
        if (!uni) kpos_advance(ka, g.Ci);
        if (!B_DGRAD) {
            // B rows = output channels, K contiguous in the OHWI weight
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) {
                const int n = n0 + a_row + ROWS_PER_PASS * i;
                const bool ok = kok && n < g.Co;
                b_reg[i] = *reinterpret_cast<const f32x4*>(W + (size_t)(ok ? n : 0) * K + (kok ? k0 : 0));
                mask |= (unsigned)ok << (16 + i);
            }
        } else {
            // B rows = k = (r,q,co) of the dgrad sum; columns = ci, contiguous in W[co][R-1-r][S-1-q][:]
            // here g.Ci is the dY channel count (= weight Co) and g.Co the weight Ci
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) {
                const int k = kt + b_kk[i], n = n0 + b_ch[i] * 4;
                const bool ok = k < k_end && n < g.Co;
                const int wc = uni ? uc0 + b_kk[i] : kb[i].c, wr = uni ? ur : kb[i].r, wq = uni ? uq : kb[i].q;
                const int wrow = ok ? (wc * g.R + (g.R - 1 - wr)) * g.S + (g.S - 1 - wq) : 0;
                b_reg[i] = *reinterpret_cast<const f32x4*>(W + (size_t)wrow * g.Co + (ok ? n : 0));
                mask |= (unsigned)ok << (16 + i);
                if (!uni) kpos_advance(kb[i], g.Ci);
            }
        }
        if (uni) {                                       // scalar tap walk, branch-free (a branch here costs a vmcnt(0))
            uc0 += BKT;
            const int wc = uc0 >= g.Ci;
            uc0 = wc ? 0 : uc0;
            uq += wc;
            const int wq = uq == g.S;
            uq = wq ? 0 : uq;
            ur += wq;
        }
    };
    auto store_lds = [&](int buf, const f32x4 (&a_reg)[A_LOADS], const f32x4 (&b_reg)[B_LOADS], const f32x4 (&a_relu)[AMASK ? A_LOADS : 1],
                         unsigned mask) {
        float* a = As + buf * A_FLOATS;
        float* b = Bs + buf * B_FLOATS;
        unsigned char* a3 = A3 + buf * AP3::BYTES;
        unsigned char* b3 = B3 + buf * B3_BYTES;
        const f32x4 zero{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            f32x4 v = BUF ? a_reg[i] : ((mask >> i) & 1u ? a_reg[i] : zero);
            if (AMASK) {
                v.x = a_relu[i].x > 0.f ? v.x : 0.f; v.y = a_relu[i].y > 0.f ? v.y : 0.f;
                v.z = a_relu[i].z > 0.f ? v.z : 0.f; v.w = a_relu[i].w > 0.f ? v.w : 0.f;
            }
            if (S3) store_split3<AP3::PLANE>(a3, (a_row + ROWS_PER_PASS * i) * AP3::PITCH + a_chunk * 8, v);
            else
            *reinterpret_cast<f32x4*>(a + (a_row + ROWS_PER_PASS * i) * A_PITCH + a_chunk * 4) = v;
        }
        if (!B_DGRAD) {
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) {
                const f32x4 v = BUF ? b_reg[i] : ((mask >> (16 + i)) & 1u ? b_reg[i] : zero);
                if (S3) store_split3<BP3C::PLANE>(b3, (a_row + ROWS_PER_PASS * i) * BP3C::PITCH + a_chunk * 8, v);
                else *reinterpret_cast<f32x4*>(b + (a_row + ROWS_PER_PASS * i) * B_PITCH + a_chunk * 4) = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) {
                const f32x4 v = BUF ? b_reg[i] : ((mask >> (16 + i)) & 1u ? b_reg[i] : zero);
                if (S3) store_split3<BP3S::PLANE>(b3, b_kk[i] * BP3S::PITCH + b_ch[i] * 8, v);
                else *reinterpret_cast<f32x4*>(b + b_kk[i] * B_PITCH + b_ch[i] * 4) = v;
            }
        }
    };

    // unsplit launches start their accumulators from bias (+ addend): those reads overlap the first K tile's loads and the
    // epilogue has no loads left (16 dependent global reads per fragment, each waited for, when the addend sat there)
    const bool final_pass = g.splits == 1;
    f32x16 acc[FM][FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn + j * 32 + frag_col(lane);
        const bool nok = n < g.Co;
        const float bv = (final_pass && bias && nok) ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm + i * 32 + frag_row(lane, e);
                const bool ok = final_pass && addend && nok && m < M;
                acc[i][j][e] = bv + (ok ? addend[(size_t)(ok ? m : 0) * g.Co + (ok ? n : 0)] : 0.f);
            }
    }

    auto multiply_tile = [&](int buf, int kt) {
        if (S3) {
            // no early exit on a ragged K tail here: ds_read_b64_tr_b16 needs every lane (the tail of the tile is zeros)
#pragma unroll
            for (int ks = 0; ks < BKT / BK; ++ks) {
                Frag3 a[FM], b[FN];
                read_kcontig3<FM, AP3::PITCH, AP3::PLANE>(A3 + buf * AP3::BYTES + wm * AP3::PITCH, lane, ks, a);
                if (!B_DGRAD) read_kcontig3<FN, BP3C::PITCH, BP3C::PLANE>(B3 + buf * B3_BYTES + wn * BP3C::PITCH, lane, ks, b);
                else read_kstrided3<FN, BP3S::PITCH, BP3S::PLANE>(B3 + buf * B3_BYTES + wn * 2, lane, ks, b);
                mma3_step<FM, FN>(a, b, acc);
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < BKT / BK; ++ks) {
            if (BKT > BK && kt + ks * BK >= k_end) break;              // ragged K tail of a deep tile
            float a[FM][8], b[FN][8];
            read_kcontig<FM, A_PITCH>(As + buf * A_FLOATS + wm * A_PITCH, lane, ks, a);
            if (!B_DGRAD) read_kcontig<FN, B_PITCH>(Bs + buf * B_FLOATS + wn * B_PITCH, lane, ks, b);
            else read_kstrided<FN, B_PITCH>(Bs + buf * B_FLOATS + wn, lane, ks, b);
            mma_any<S3 ? 0 : MMA, FM, FN>(a, b, acc);
        }
    };

    if (k_begin < k_end) {
        const int ntiles = (k_end - k_begin + BKT - 1) / BKT;
        // ring of PF register sets: slot (t mod PF) holds tile t until it has been written to LDS (one iteration before it is
        // multiplied), then takes tile t + PF.  Loads past the end are fully masked (they read element 0).
#pragma unroll
        for (int d = 0; d < PF; ++d) load_global(k_begin + d * BKT, a_set[d], b_set[d], r_set[d], mask_set[d]);
        store_lds(0, a_set[0], b_set[0], r_set[0], mask_set[0]);
        __syncthreads();
        int buf = 0;
        // one iteration: slot U was emptied an iteration ago and takes tile tt + PF; tile tt (in LDS) is multiplied; tile
        // tt + 1 moves from its slot to the other LDS buffer.  Straight-line code (no branch between the loads and their use:
        // hipcc answers a branch with s_waitcnt vmcnt(0), which would drain the whole ring every iteration).
        auto iteration = [&](auto U, int tt) {
            constexpr int u = decltype(U)::value;
            load_global(k_begin + (tt + PF) * BKT, a_set[u], b_set[u], r_set[u], mask_set[u]);
            multiply_tile(buf, k_begin + tt * BKT);
            // keep the LDS fill (and the wait for the global loads in front of it) BEHIND the MFMAs: left alone, the
            // scheduler hoists it above them - the operands are already in registers - and every wave then sits out its
            // full load latency before it issues a single MFMA
            // (with a deep ring the loads are long since complete: the scheduler is free to interleave the split / LDS fill of
            // the next tile with this tile's MFMAs, which would otherwise leave the vector ALU idle while the matrix pipe drains)
            if (PF == 1) __builtin_amdgcn_sched_barrier(0);
            constexpr int v = (u + 1) % PF;
            store_lds(buf ^ 1, a_set[v], b_set[v], r_set[v], mask_set[v]);
            if (PF > 1 && S3 && g_interleave) {
                // one MFMA, then a handful of the next tile's split / address instructions, ...: the matrix pipe takes an MFMA
                // every 32 cycles and blocks the vector issue for 8 of them
#pragma unroll
                for (int i = 0; i < 6 * FM * FN * (BKT / BK); ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, (14 + FM * FN - 1) / (FM * FN), 0);
                }
            }
            __syncthreads();
            buf ^= 1;
        };
        int t = 0;
        for (; t + PF <= ntiles; t += PF) unroll_iterations<PF>(iteration, t);
        if (PF > 1) tail_iterations<PF - 1>(iteration, t, ntiles);
    }

    // ---- per-channel statistics of the result (BatchNorm's batch statistics without a second pass over the tensor):
    // every 32-row slab of the output writes its (sum, sum of squares) per column into stats[slab][2*Co]; rows past M are
    // exact zeros (their A rows were zero and a convolution in front of a BatchNorm has no bias) -------------------
    if (stats != nullptr && final_pass) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn + j * 32 + frag_col(lane);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float v = acc[i][j][e]; sm += v; sq += v * v; }
                sm += __shfl_xor(sm, 32, 64);
                sq += __shfl_xor(sq, 32, 64);
                if (lane < 32 && n < g.Co) {
                    float* p = stats + (size_t)((m0 + wm + i * 32) >> 5) * 2 * g.Co;
                    p[n] = sm;
                    p[g.Co + n] = sq;
                }
            }
        }
    }

    // ---- epilogue: bias / relu, or raw partial sums when split-K --------------------------------
    float* dst = out + (size_t)split * M * g.Co;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn + j * 32 + frag_col(lane);
        if (n >= g.Co) continue;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm + i * 32 + frag_row(lane, e);
                if (m < M) dst[(size_t)m * g.Co + n] = (final_pass && relu) ? fmaxf(acc[i][j][e], 0.f) : acc[i][j][e];
            }
    }
}

template <int BM, int BN, bool B_DGRAD, int BKT, bool UNI, int MMA = 0, int PF = 1, bool BUF = false>
__global__ __launch_bounds__(THREADS) void conv_igemm_kernel(
    const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ bias,
    const float* __restrict__ addend, float* __restrict__ out, ConvShape g, int relu, float* __restrict__ stats)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    igemm_tile<BM, BN, B_DGRAD, BKT, UNI, false, MMA, PF, BUF>(X, W, bias, addend, out, g, relu, lds, blockIdx.x, gridDim.x, blockIdx.z,
                                                               nullptr, stats);
}

// ---- 3x3 / stride 1 / pad 1 forward and data gradient: the three taps of a filter row from ONE staged pixel block ----------
// In the generic tile program every K step gathers its own A block: the steps of the taps (dy,-1), (dy,0), (dy,+1) fetch, split
// and stage the SAME pixels shifted by one.  Here a "unit" = (filter row dy, 16-channel chunk) stages the 66 pixels m0-1 ..
// m0+64 of the shifted image row ONCE (bf16 planes, K-contiguous) and runs three K steps on it: step dx reads its A fragments
// dx rows further down the image and only the 64x16 weight tile is staged per step.  A-side loads, split arithmetic and LDS
// fills fall 2.9x - a third of the loop's operand traffic and vector instructions.  Measured: -2..-4 % (forward) and -6 % (dgrad)
// per layer, step -0.19 ms: what remains is the loop skeleton (one barrier and one 6-MFMA burst per wave and step).
// Rows whose tap leaves the image row (x = 0 for dx = -1, x = W-1 for dx = +1) are zeroed in
// the fragment (a per-lane select, the flags are fixed over the K loop); block rows whose image row y + dy is outside the frame -
// the block may span image rows and frames - are staged as zeros.  64x64 tile, 4 waves, bf16x3 arithmetic, buffer loads,
// prefetch ring of one unit (three weight tiles + the next A block); epilogue (bias / addend / ReLU / BatchNorm statistics /
// split-K partials) as in igemm_tile.  DGRAD: the same gather on dY with the flipped, transposed weight as B (K-strided planes).
constexpr int T3_AROWS = 66;
int g_taps3 = 1;                                             // tuning aid: phnet_tune_force_k_tile(-5 / -6) switches this kernel off / on
template <bool DGRAD>
__global__ __launch_bounds__(THREADS) void conv3x3s1_kernel(
    const float* __restrict__ X, const float* __restrict__ Wt, const float* __restrict__ bias,
    const float* __restrict__ addend, float* __restrict__ out, ConvShape g, int relu, float* __restrict__ stats)
{
    constexpr int APITCH = KContigPlanes<64, 16>::PITCH;     // 48 bytes: 16 bf16 + pad
    constexpr int APLANE = T3_AROWS * APITCH, AIMG = 3 * APLANE;
    typedef KContigPlanes<64, 16> BP3C;
    typedef KStridedPlanes<64, 16> BP3S;
    constexpr int BIMG = DGRAD ? BP3S::BYTES : BP3C::BYTES;
    constexpr int BPLANE = DGRAD ? BP3S::PLANE : BP3C::PLANE;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned char* A3 = reinterpret_cast<unsigned char*>(lds);       // [2][3 planes][66 rows][48]
    unsigned char* B3 = A3 + 2 * AIMG;                               // [2][3 planes]...
    unsigned char* dump = B3 + 2 * BIMG;                             // 3 x 512 bytes: where lanes without a second A row write

    const int W = g.Wi, H = g.Hi, Ca = g.Ci;                 // A-side image and channel count
    const int M = g.N * H * W, K = 9 * Ca;
    const int CC = Ca >> 4;                                  // 16-channel chunks
    const int tiles_n = (g.Co + 63) >> 6;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (int)(tile / tiles_n) * 64, n0 = (int)(tile % tiles_n) * 64;
    const int u_begin = blockIdx.z * g.k_per_split, u_end = min(3 * CC, u_begin + g.k_per_split);      // units of this split
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;

    // ---- A staging: 66 rows x 4 chunks = 264 float4, thread tid takes chunk idx = tid and (tid < 8) idx = 256 + tid ----
    constexpr unsigned OOB = 0x80000000u;
    __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)min((long)M * Ca * 4, (long)0x7fffffff), 0x00020000);
    __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, (int)min((long)g.Co * K * 4, (long)0x7fffffff), 0x00020000);
    int a_off[2], a_ok[2], a_lds[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i, row = idx >> 2, chunk = idx & 3;
        const bool rowok = idx < 4 * T3_AROWS;
        const int t = m0 - 1 + row;                          // aligned pixel of the row
        const bool tok = rowok && (unsigned)t < (unsigned)M;
        const int tt = tok ? t : 0;
        const int y = (tt / W) % H;
        a_ok[i] = (tok && y >= 1 ? 1 : 0) | (tok ? 2 : 0) | (tok && y + 1 < H ? 4 : 0);          // bit dyi: image row y + dyi - 1 exists
        a_off[i] = (t * Ca + chunk * 4) * 4;
        a_lds[i] = rowok ? row * APITCH + chunk * 8 : -1;
    }
    // ---- B staging (one float4 per thread and step), as in igemm_tile ----
    const int a_row = tid >> 2, a_chunk = tid & 3;           // forward: weight row n0 + a_row, k chunk a_chunk
    const int b_kk = tid >> 4, b_ch = tid & 15;              // dgrad: k row b_kk, columns n0 + 4 b_ch ..
    int b_off, b_lds;
    if (!DGRAD) {
        const int n = n0 + a_row;
        b_off = n < g.Co ? (n * K + a_chunk * 4) * 4 : (int)OOB;
        b_lds = a_row * BP3C::PITCH + a_chunk * 8;
    } else {
        const int n = n0 + b_ch * 4;
        b_off = n < g.Co ? (b_kk * 9 * g.Co + n) * 4 : (int)OOB;
        b_lds = b_kk * BP3S::PITCH + b_ch * 8;
    }
    // running unit decomposition of the LOADS (they run ahead of the multiplies): unit -> (dyi, cc)
    int la_dy = u_begin / CC, la_cc = u_begin - la_dy * CC, la_u = u_begin;           // next A block to load
    int lb_dy = la_dy, lb_cc = la_cc, lb_u = u_begin, lb_dx = 0;                    // next weight tile to load
    // (the 8 chunks past the first 256 belong to wave 0: the other waves skip them under a wave-uniform branch - done branch-free
    // with a dump slot, every thread paid a second split per unit and the kernel gained nothing over the generic one)
    const bool wave0 = __builtin_amdgcn_readfirstlane(wave) == 0;
    auto load_a = [&](f32x4 (&reg)[2]) {
        const bool uok = la_u < u_end;
        const int s_a = ((la_dy - 1) * W * Ca + la_cc * 16) * 4;
        {
            const bool ok = uok && ((a_ok[0] >> la_dy) & 1);
            reg[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, ok ? a_off[0] + s_a : (int)OOB, 0, 0));
        }
        if (wave0) {
            const bool ok = uok && ((a_ok[1] >> la_dy) & 1);
            reg[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, ok ? a_off[1] + s_a : (int)OOB, 0, 0));
        }
        ++la_u; ++la_cc;
        const int wrap = la_cc == CC;
        la_cc = wrap ? 0 : la_cc;
        la_dy += wrap;
    };
    auto load_b = [&](f32x4& reg) {
        const bool uok = lb_u < u_end;
        const int s_b = !DGRAD ? ((lb_dy * 3 + lb_dx) * Ca + lb_cc * 16) * 4
                               : ((lb_cc * 16 * 3 + (2 - lb_dy)) * 3 + (2 - lb_dx)) * g.Co * 4;
        reg = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, uok ? b_off + s_b : (int)OOB, 0, 0));
        ++lb_dx;
        const int w3 = lb_dx == 3;
        lb_dx = w3 ? 0 : lb_dx;
        lb_u += w3; lb_cc += w3;
        const int wrap = lb_cc == CC;
        lb_cc = wrap ? 0 : lb_cc;
        lb_dy += wrap;
    };
    auto store_a = [&](int aoff, const f32x4 (&reg)[2]) {
        store_split3<APLANE>(A3 + aoff, a_lds[0], reg[0]);
        if (wave0) {
            // second chunk: the 8 lanes that have one write it into the image, the others (an out-of-range load: zeros) into the dump
            unsigned h0, m0_, l0, h1, m1, l1;
            split3_pair(reg[1].x, reg[1].y, h0, m0_, l0);
            split3_pair(reg[1].z, reg[1].w, h1, m1, l1);
            const bool has = a_lds[1] >= 0;
            unsigned char* q = has ? A3 + aoff + a_lds[1] : dump + lane * 8;
            const int ps = has ? APLANE : 512;
            *reinterpret_cast<u32x2*>(q) = (u32x2){h0, h1};
            *reinterpret_cast<u32x2*>(q + ps) = (u32x2){m0_, m1};
            *reinterpret_cast<u32x2*>(q + 2 * ps) = (u32x2){l0, l1};
        }
    };
    auto store_b = [&](int boff, const f32x4& reg) { store_split3<BPLANE>(B3 + boff, b_lds, reg); };

    // accumulators start from bias (+ addend) when this launch is the final pass (igemm_tile)
    const bool final_pass = g.splits == 1;
    f32x16 acc[1][1];
    {
        const int n = n0 + wn + frag_col(lane);
        const bool nok = n < g.Co;
        const float bv = (final_pass && bias && nok) ? bias[n] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + wm + frag_row(lane, e);
            const bool ok = final_pass && addend && nok && m < M;
            acc[0][0][e] = bv + (ok ? addend[(size_t)(ok ? m : 0) * g.Co + (ok ? n : 0)] : 0.f);
        }
    }
    // per-lane border flags of this lane's A row (pixel m0 + wm + (lane & 31)): fixed over the K loop
    const int pr = m0 + wm + (lane & 31);
    const int px = (pr < M ? pr : 0) % W;
    const bool edge_l = px == 0, edge_r = px == W - 1;
    const bool any_l = __builtin_amdgcn_ballot_w64(edge_l) != 0, any_r = __builtin_amdgcn_ballot_w64(edge_r) != 0;

    if (u_begin < u_end) {
        f32x4 a_reg[2], b_set[3];
        load_a(a_reg);
#pragma unroll
        for (int d = 0; d < 3; ++d) load_b(b_set[d]);
        store_a(0, a_reg);
        load_a(a_reg);
        store_b(0, b_set[0]);
        load_b(b_set[0]);
        __syncthreads();
        int aoff = 0, boff = 0;                              // byte offsets of the A image / weight tile being multiplied
        const int r = lane & 31, hb = (lane >> 5) * 16;
        auto step = [&](auto DX) {
            constexpr int dxi = decltype(DX)::value;
            {
                Frag3 a[1], b[1];
                {
                    const unsigned char* p = A3 + aoff + (wm + r + dxi) * APITCH + hb;
                    a[0].hi = *reinterpret_cast<const bf16x8*>(p);
                    a[0].mid = *reinterpret_cast<const bf16x8*>(p + APLANE);
                    a[0].lo = *reinterpret_cast<const bf16x8*>(p + 2 * APLANE);
                }
                if (!DGRAD) read_kcontig3<1, BP3C::PITCH, BP3C::PLANE>(B3 + boff + wn * BP3C::PITCH, lane, 0, b);
                else read_kstrided3<1, BP3S::PITCH, BP3S::PLANE>(B3 + boff + wn * 2, lane, 0, b);
                if (dxi != 1 && (dxi == 0 ? any_l : any_r)) {  // the tap leaves the image row on this lane's pixel: a zero row
                    const bool z = dxi == 0 ? edge_l : edge_r;    // (wave-uniform skip: most waves of a wide image hold no border pixel)
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 keep = z ? (u32x4){0u, 0u, 0u, 0u} : (u32x4){~0u, ~0u, ~0u, ~0u};
                    a[0].hi = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, a[0].hi) & keep);
                    a[0].mid = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, a[0].mid) & keep);
                    a[0].lo = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, a[0].lo) & keep);
                }
                mma3_step<1, 1>(a, b, acc);
                // the next weight tile moves from its ring slot into the other LDS buffer; the slot takes the tile a unit ahead
                store_b(boff ^ BIMG, b_set[(dxi + 1) % 3]);
                load_b(b_set[(dxi + 1) % 3]);
                if (dxi == 2) {                               // ... and, on a unit's last step, the next A block
                    store_a(aoff ^ AIMG, a_reg);
                    load_a(a_reg);
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) {                 // an MFMA, then a few of the split / address instructions, ...
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, dxi == 2 ? 14 : 8, 0);
                }
                __syncthreads();
                boff ^= BIMG;
            }
        };
        for (int u = u_begin; u < u_end; ++u) {
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            aoff ^= AIMG;
        }
    }

    // ---- per-channel statistics of the result for a following BatchNorm (igemm_tile) ----
    if (stats != nullptr && final_pass) {
        const int n = n0 + wn + frag_col(lane);
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float v = acc[0][0][e]; sm += v; sq += v * v; }
        sm += __shfl_xor(sm, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        if (lane < 32 && n < g.Co) {
            float* p = stats + (size_t)((m0 + wm) >> 5) * 2 * g.Co;
            p[n] = sm;
            p[g.Co + n] = sq;
        }
    }
    float* dst = out + (size_t)blockIdx.z * M * g.Co;
    const int n = n0 + wn + frag_col(lane);
    if (n < g.Co) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + wm + frag_row(lane, e);
            if (m < M) dst[(size_t)m * g.Co + n] = (final_pass && relu) ? fmaxf(acc[0][0][e], 0.f) : acc[0][0][e];
        }
    }
}

// out[i] = sum_z part[z][i] (+bias[i % ncols]) (relu)
// Every operand of an output element is fetched before the first add (the first eight partial sums branch-free - a
// split index past the end re-reads the last one - plus bias, addend and the old value): one memory latency per launch
// instead of one per split.  The additions keep their order (z ascending, then bias, addend, relu, old value).
// stats (optional; ncols a power of two <= 1024, no bias / relu): workgroup b also writes the per-column (sum, sum of squares)
// of its 1024 / ncols rows into stats[b][2 * ncols] - BatchNorm's statistics pass folded into the reduce.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                            const float* __restrict__ bias, const float* __restrict__ addend,
                                                            long total4, int ncols, int splits, int relu, int accumulate,
                                                            float* __restrict__ stats)
{
    __shared__ f32x4 red_s[2][256];
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (stats != nullptr) {                               // (every thread reaches the barrier below)
        const bool ok = i < total4;
        f32x4 s{0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const f32x4* p = reinterpret_cast<const f32x4*>(part) + i;
            s = p[0];
            for (int z = 1; z < splits; ++z) s += p[(long)z * total4];
            reinterpret_cast<f32x4*>(out)[i] = s;
        }
        red_s[0][threadIdx.x] = s;
        red_s[1][threadIdx.x] = s * s;
        __syncthreads();
        const int c4 = ncols >> 2;                        // float4 columns; rows per workgroup = 256 / c4
        if ((int)threadIdx.x < c4) {
            f32x4 a = red_s[0][threadIdx.x], b = red_s[1][threadIdx.x];
            for (int r = c4; r < 256; r += c4) { a += red_s[0][r + threadIdx.x]; b += red_s[1][r + threadIdx.x]; }
            float* q = stats + (size_t)blockIdx.x * 2 * ncols + threadIdx.x * 4;
            *reinterpret_cast<f32x4*>(q) = a;
            *reinterpret_cast<f32x4*>(q + ncols) = b;
        }
        return;
    }
    if (i >= total4) return;
    const f32x4* p = reinterpret_cast<const f32x4*>(part) + i;
    const f32x4 zero{0.f, 0.f, 0.f, 0.f};
    f32x4 v[8];
#pragma unroll
    for (int z = 0; z < 8; ++z) v[z] = p[(long)min(z, splits - 1) * total4];
    const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + (int)((i * 4) % ncols)) : zero;
    const f32x4 av = addend ? reinterpret_cast<const f32x4*>(addend)[i] : zero;
    const f32x4 ov = accumulate ? reinterpret_cast<const f32x4*>(out)[i] : zero;
    f32x4 s = v[0];
#pragma unroll
    for (int z = 1; z < 8; ++z)
        if (z < splits) s += v[z];
    for (int z = 8; z < splits; ++z) s += p[(long)z * total4];
    if (bias) s += bv;
    if (addend) s += av;
    if (relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
    if (accumulate) s += ov;
    reinterpret_cast<f32x4*>(out)[i] = s;
}

// ---- weight gradient -----------------------------------------------------------------------------
struct WgradShape {
    int N, Hi, Wi, Ci;       // forward input image X
    int Ho, Wo, Co;          // forward output image (dY)
    int R, S, stride, pad;
    int splits, pix_per_split;   // multiple of BK
};

// out: dW itself when g.splits == 1 (written or accumulated in the epilogue) else the split-K partial buffer
// [splits][Co*NC + Co] (the trailing Co floats of every split hold its bias-gradient partial).
// PF / BUF: register prefetch ring and buffer loads as in igemm_tile (used by the staged-split arithmetic)
template <int BM, int BN, int MMA = 0, int BKW = BK, int PF = 1, bool BUF = false>
__global__ __launch_bounds__(THREADS) void conv_wgrad_kernel(
    const float* __restrict__ dY, const float* __restrict__ X, float* __restrict__ out, float* __restrict__ dbias,
    WgradShape g, int want_bias, int accumulate)
{
    constexpr int TM = BM / 2, TN = BN / 2, FM = TM / 32, FN = TN / 32;
    // BKW = pixels per K step (16; 32 for the staged-split arithmetic, whose MFMA burst per step is 2.7x shorter)
    constexpr int A_PITCH = KStridedTile<BM, BKW>::PITCH, B_PITCH = KStridedTile<BN, BKW>::PITCH;
    constexpr int A_FLOATS = KStridedTile<BM, BKW>::FLOATS, B_FLOATS = KStridedTile<BN, BKW>::FLOATS;
    constexpr int A_LOADS = BKW * BM / 4 / THREADS, B_LOADS = BKW * BN / 4 / THREADS;      // float4 loads per thread and K step
    constexpr bool S3 = MMA == 3;                            // bf16 planes (igemm.h), both operands K-strided: transposed reads
    typedef KStridedPlanes<BM, BKW> AP3;
    typedef KStridedPlanes<BN, BKW> BP3;
    constexpr int LDS_BYTES = S3 ? 2 * (AP3::BYTES + BP3::BYTES) : 2 * (A_FLOATS + B_FLOATS) * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    float* lds = reinterpret_cast<float*>(lds_raw);
    float* As = lds;
    float* Bs = lds + 2 * A_FLOATS;
    unsigned char* A3 = lds_raw;
    unsigned char* B3 = lds_raw + 2 * AP3::BYTES;

    const int NC = g.R * g.S * g.Ci;                         // GEMM N
    const int P = g.N * g.Ho * g.Wo;                         // GEMM K (pixels)
    const int tiles_n = (NC + BN - 1) / BN;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (int)(tile / tiles_n) * BM, n0 = (int)(tile % tiles_n) * BN;
    const int p_begin = blockIdx.z * g.pix_per_split, p_end = min(P, p_begin + g.pix_per_split);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * TM, wn = (wave & 1) * TN;

    // column (r,q,c) of each B chunk this thread stages is fixed over the K loop
    int b_kk[B_LOADS], b_col[B_LOADS], b_r[B_LOADS], b_q[B_LOADS], b_c[B_LOADS];
    bool b_ok[B_LOADS];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const int idx = tid + THREADS * i;
        b_kk[i] = idx / (BN / 4);
        b_col[i] = (idx - b_kk[i] * (BN / 4)) * 4;
        const int col = n0 + b_col[i];
        b_ok[i] = col < NC;
        const int rs = b_ok[i] ? col / g.Ci : 0;
        b_c[i] = col - rs * g.Ci;
        b_r[i] = rs / g.S;
        b_q[i] = rs - b_r[i] * g.S;
    }

    // running (image, row, col) decomposition of the pixel each B chunk reads: divisions once, then carries
    int b_n[B_LOADS], b_oy[B_LOADS], b_ox[B_LOADS];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const int pp = min(p_begin + b_kk[i], max(P - 1, 0));
        b_n[i] = pp / (g.Ho * g.Wo);
        const int rem = pp - b_n[i] * (g.Ho * g.Wo);
        b_oy[i] = rem / g.Wo;
        b_ox[i] = rem - b_oy[i] * g.Wo;
    }
    int a_kk[A_LOADS], a_ch[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int idx = tid + THREADS * i;
        a_kk[i] = idx / (BM / 4);
        a_ch[i] = (idx - a_kk[i] * (BM / 4)) * 4;
    }

    f32x4 a_set[PF][A_LOADS], b_set[PF][B_LOADS], bsum[A_LOADS];
    unsigned m_set[PF];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) bsum[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- BUF: 32-bit byte offsets, out-of-range elements pushed past the end of the tensor (the hardware returns zeros) ----
    constexpr unsigned OOB = 0x80000000u;
    __amdgpu_buffer_rsrc_t y_rsrc, x_rsrc;
    int a_off[BUF ? A_LOADS : 1];
    float inv_wo = 0.f, inv_ho = 0.f;
    if constexpr (BUF) {
        y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, (int)min((long)P * g.Co * 4, (long)0x7fffffff), 0x00020000);
        x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)min((long)g.N * g.Hi * g.Wi * g.Ci * 4, (long)0x7fffffff), 0x00020000);
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) a_off[i] = m0 + a_ch[i] < g.Co ? (a_kk[i] * g.Co + m0 + a_ch[i]) * 4 : (int)OOB;
        inv_wo = 1.0f / (float)g.Wo; inv_ho = 1.0f / (float)g.Ho;
    }
    const bool bias_block = want_bias && n0 == 0;            // the first column tile also sums dY over the pixels
    // loads the K step that starts at pixel pt; MUST be called with pt = p_begin, p_begin+BK, ... in order
    // branch-free loads (masked chunks read element 0 and are zeroed on the LDS write): straight-line code, exact s_waitcnt
    auto load_global = [&](int pt, bool advance, f32x4 (&a_reg)[A_LOADS], f32x4 (&b_reg)[B_LOADS], unsigned& lmask) {
        lmask = 0;
        if (advance) {                                   // pixel carry of the B gather: BEFORE the loads, so that nothing but
#pragma unroll                                           // straight-line code sits between them and the MFMAs
            for (int i = 0; i < B_LOADS; ++i) {
                if constexpr (BUF) {
                    // branch-free (a branch costs the ring an s_waitcnt vmcnt(0)): wraps = floor(ox / Wo) by a float reciprocal
                    // (exact after one correction step for ox < 2^22), then the same for the rows
                    int ox = b_ox[i] + BKW;
                    int w = (int)((float)ox * inv_wo);
                    ox -= w * g.Wo;
                    const int lo = ox < 0, hi = ox >= g.Wo;
                    ox += lo ? g.Wo : 0; ox -= hi ? g.Wo : 0; w += hi - lo;
                    int oy = b_oy[i] + w;
                    int v = (int)((float)oy * inv_ho);
                    oy -= v * g.Ho;
                    const int lo2 = oy < 0, hi2 = oy >= g.Ho;
                    oy += lo2 ? g.Ho : 0; oy -= hi2 ? g.Ho : 0; v += hi2 - lo2;
                    b_ox[i] = ox; b_oy[i] = oy; b_n[i] += v;
                } else {
                    b_ox[i] += BKW;
                    while (b_ox[i] >= g.Wo) {
                        b_ox[i] -= g.Wo;
                        if (++b_oy[i] == g.Ho) { b_oy[i] = 0; ++b_n[i]; }
                    }
                }
            }
        }
        if constexpr (BUF) {
            const int s_a = pt * g.Co * 4;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const bool ok = pt + a_kk[i] < p_end;
                a_reg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(y_rsrc, ok ? a_off[i] + s_a : (int)OOB, 0, 0));
            }
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) {
                const int iy = b_oy[i] * g.stride - g.pad + b_r[i], ix = b_ox[i] * g.stride - g.pad + b_q[i];
                const bool ok = b_ok[i] && pt + b_kk[i] < p_end && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
                const int off = (((b_n[i] * g.Hi + iy) * g.Wi + ix) * g.Ci + b_c[i]) * 4;
                b_reg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, ok ? off : (int)OOB, 0, 0));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int p = pt + a_kk[i], co = m0 + a_ch[i];
            const bool ok = p < p_end && co < g.Co;
            a_reg[i] = *reinterpret_cast<const f32x4*>(dY + (ok ? (size_t)p * g.Co + co : 0));
            lmask |= (unsigned)ok << i;
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int p = pt + b_kk[i];
            const int iy = b_oy[i] * g.stride - g.pad + b_r[i], ix = b_ox[i] * g.stride - g.pad + b_q[i];
            const bool ok = b_ok[i] && p < p_end && iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi;
            const int pix = ok ? (b_n[i] * g.Hi + iy) * g.Wi + ix : 0;
            b_reg[i] = *reinterpret_cast<const f32x4*>(X + (size_t)pix * g.Ci + (ok ? b_c[i] : 0));
            lmask |= (unsigned)ok << (16 + i);
        }
    };
    auto store_lds = [&](int buf, const f32x4 (&a_reg)[A_LOADS], const f32x4 (&b_reg)[B_LOADS], unsigned lmask) {
        const f32x4 zero{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const f32x4 v = BUF ? a_reg[i] : ((lmask >> i) & 1u ? a_reg[i] : zero);
            if (bias_block) bsum[i] += v;
            if (S3) store_split3<AP3::PLANE>(A3 + buf * AP3::BYTES, a_kk[i] * AP3::PITCH + a_ch[i] * 2, v);
            else *reinterpret_cast<f32x4*>(As + buf * A_FLOATS + a_kk[i] * A_PITCH + a_ch[i]) = v;
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const f32x4 v = BUF ? b_reg[i] : ((lmask >> (16 + i)) & 1u ? b_reg[i] : zero);
            if (S3) store_split3<BP3::PLANE>(B3 + buf * BP3::BYTES, b_kk[i] * BP3::PITCH + b_col[i] * 2, v);
            else *reinterpret_cast<f32x4*>(Bs + buf * B_FLOATS + b_kk[i] * B_PITCH + b_col[i]) = v;
        }
    };

    // unsplit + accumulate: start the accumulators from the old gradient (its HBM latency overlaps the first tile's loads;
    // the epilogue is then a plain store instead of a read-modify-write)
    const bool from_old = g.splits == 1 && accumulate;
    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wn + j * 32 + frag_col(lane), m = m0 + wm + i * 32 + frag_row(lane, e);
                const bool ok = from_old && n < NC && m < g.Co;
                const float v = out[ok ? (size_t)m * NC + n : 0];                 // branch-free: all 16 reads in flight together
                acc[i][j][e] = ok ? v : 0.f;
            }

    if (p_begin < p_end) {
        const int nsteps = (p_end - p_begin + BKW - 1) / BKW;
#pragma unroll
        for (int d = 0; d < PF; ++d) load_global(p_begin + d * BKW, d > 0, a_set[d], b_set[d], m_set[d]);
        store_lds(0, a_set[0], b_set[0], m_set[0]);
        __syncthreads();
        int buf = 0;
        auto iteration = [&](auto U, int tt) {
            constexpr int u = decltype(U)::value;
            load_global(p_begin + (tt + PF) * BKW, true, a_set[u], b_set[u], m_set[u]);       // past the end: fully masked
#pragma unroll
            for (int ks = 0; ks < BKW / BK; ++ks) {
                if (S3) {
                    Frag3 a[FM], b[FN];
                    read_kstrided3<FM, AP3::PITCH, AP3::PLANE>(A3 + buf * AP3::BYTES + wm * 2, lane, ks, a);
                    read_kstrided3<FN, BP3::PITCH, BP3::PLANE>(B3 + buf * BP3::BYTES + wn * 2, lane, ks, b);
                    mma3_step<FM, FN>(a, b, acc);
                } else {
                    float a[FM][8], b[FN][8];
                    read_kstrided<FM, A_PITCH>(As + buf * A_FLOATS + wm, lane, ks, a);
                    read_kstrided<FN, B_PITCH>(Bs + buf * B_FLOATS + wn, lane, ks, b);
                    mma_any<S3 ? 0 : MMA, FM, FN>(a, b, acc);
                }
            }
            if (PF == 1) __builtin_amdgcn_sched_barrier(0);           // the LDS fill (and its wait for the loads) stays behind the MFMAs
            constexpr int v = (u + 1) % PF;
            store_lds(buf ^ 1, a_set[v], b_set[v], m_set[v]);
            if (PF > 1 && S3 && g_interleave) {
#pragma unroll
                for (int i = 0; i < 6 * FM * FN * (BKW / BK); ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, (14 + FM * FN - 1) / (FM * FN), 0);
                }
            }
            __syncthreads();
            buf ^= 1;
        };
        int t = 0;
        for (; t + PF <= nsteps; t += PF) unroll_iterations<PF>(iteration, t);
        if (PF > 1) tail_iterations<PF - 1>(iteration, t, nsteps);
    }
    const bool direct = g.splits == 1;
    float* dst = direct ? out : out + (size_t)blockIdx.z * ((size_t)g.Co * NC + g.Co);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn + j * 32 + frag_col(lane);
        if (n >= NC) continue;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm + i * 32 + frag_row(lane, e);
                if (m < g.Co) {
                    float* q = dst + (size_t)m * NC + n;
                    *q = acc[i][j][e];
                }
            }
    }
    if (bias_block) {
        // per-thread sums cover rows a_kk of every K step; fold the BK rows through LDS (reusing the A image)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i)
            *reinterpret_cast<f32x4*>(As + a_kk[i] * A_PITCH + a_ch[i]) = bsum[i];
        __syncthreads();
        if (tid < BM && m0 + tid < g.Co) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < BKW; ++k) t += As[k * A_PITCH + tid];
            if (direct) dbias[m0 + tid] = accumulate ? dbias[m0 + tid] + t : t;
            else dst[(size_t)g.Co * NC + m0 + tid] = t;
        }
    }
}

// ---- weight gradient of a 3x3 / stride 1 / pad 1 convolution: the three taps of a filter row from ONE staged pixel block ----
// The generic kernel above treats the 9 taps as 9 column tiles: every (co tile, tap, ci tile) workgroup loads, splits and
// stages its own dY block and its own (shifted) X block.  The taps (dy, -1), (dy, 0), (dy, +1) read the SAME 16 dY pixels and
// X rows that differ by one pixel, so here one workgroup of 12 waves stages dY[16 px][64 co] and X[18 px][64 ci] once per K
// step and three groups of 4 waves multiply them - group dx reads the X planes one row further down and, where a pixel of
// the block sits on the image border its tap would cross (x = 0 for dx = -1, x = W-1 for dx = +1: at most one pixel per 16,
// W >= 16), clears that pixel's element of its dY fragment.  Loads, split arithmetic and LDS fills per MFMA fall 2.8x; the
// output tile (3 x 64x64) and the partial-sum traffic per MFMA stay what they were.  X rows whose image row y + dy falls
// outside the frame are staged as zeros (the pixel block may span image rows and frames: rows are tested one by one).
// bf16x3 arithmetic, buffer loads; Ci, Co multiples of 64.
constexpr int W3_THREADS = 768;
int g_wgrad3 = 1, g_wgrad3_target = 256;                     // tuning aids (phnet_tune_wgrad: bit 3 of arg 0 switches it off; a negative
int g_wgrad3_bkw = 16;                                       // second argument sets its workgroup target, bit 4 selects 32-pixel steps)
int g_wgrad3s = 1;                                           // producer / consumer variant (csrc/wgrad3s.hip) where no bias gradient is asked for; bit 5 switches it off
int g_wgrad1s = 1;                                           // 128 x 128 producer / consumer kernel for many-row Linear layers (csrc/wgrad1s.hip); bit 6 switches it off
template <int BKW> struct Wgrad3Lds {
    static constexpr int ROWS = BKW + 2;                     // X rows of a step: pixels pt-1 .. pt+BKW of the shifted image row
    static constexpr int PITCH = KStridedPlanes<64, BK>::PITCH;      // 192 bytes: 64 bf16 + pad (igemm.h)
    static constexpr int PLANE = ROWS * PITCH, IMG = 3 * PLANE;      // both operands use the ROWS-row image (dY leaves two rows unused)
    static constexpr int BYTES = 6 * IMG + 2 * PLANE + 512;          // 3 buffers x 2 operands + a dump area for the idle staging lanes
};
template <int PF, int BKW>
__global__ __launch_bounds__(W3_THREADS) void conv_wgrad3x3_kernel(
    const float* __restrict__ dY, const float* __restrict__ X, float* __restrict__ out, float* __restrict__ dbias,
    WgradShape g, int want_bias, int accumulate)
{
    constexpr int NSUB = BKW / BK;                           // 16-pixel MFMA sub-steps per K step
    typedef Wgrad3Lds<BKW> L;
    constexpr int PITCH = L::PITCH, PLANE = L::PLANE, IMG = L::IMG;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];               // [buf 0..2][dY | X][plane][row][col], dump
    static_assert(PF % 2 == 0 && (NSUB == 1 || NSUB == 2), "fragment parity is read off the ring slot");

    const int NC = 9 * g.Ci, W = g.Wi, H = g.Hi;
    const int P = g.N * H * W;
    const int ctiles = g.Ci >> 6;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int dyi = (int)(tile % 3), ct = (int)((tile / 3) % ctiles), mt = (int)(tile / (3 * ctiles));
    const int m0 = mt * 64, c0 = ct * 64, dy = dyi - 1;
    const int p_begin = blockIdx.z * g.pix_per_split, p_end = min(P, p_begin + g.pix_per_split);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tg = wave >> 2, dx = tg - 1;                   // tap group of this wave
    const int wm = ((wave >> 1) & 1) * 32, wn = (wave & 1) * 32;

    // ---- staging roles (wave-uniform): waves 0-3 dY, waves 4-7 X rows 0..BKW-1 (row kk + 16 i each), wave 8 (first half) the
    // two X rows BKW, BKW+1 ----
    const bool role_a = wave < 4, halo = wave == 8;
    const int st = tid - (role_a ? 0 : (halo ? 512 : 256));
    const int kk = (st >> 4) + (halo ? BKW : 0), col = (st & 15) * 4;                     // first row of the image, first of 4 columns
    const float* src = role_a ? dY : X;
    const int cs = role_a ? g.Co : g.Ci;                     // channels of the source tensor
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)min((long)P * cs * 4, (long)0x7fffffff), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    bool active[NSUB];
    int off_c[NSUB], t_cur[NSUB], t_x[NSUB], t_y[NSUB], st_off[NSUB];
#pragma unroll
    for (int i = 0; i < NSUB; ++i) {
        const int row = kk + 16 * i;
        active[i] = wave < 8 || (halo && lane < 32 && i == 0);
        // dY: element (pt + row, m0 + col); X: aligned pixel t = pt - 1 + row, source pixel t + dy * W, channel c0 + col
        off_c[i] = role_a ? (row * cs + m0 + col) * 4 : ((row - 1 + dy * W) * cs + c0 + col) * 4;
        t_cur[i] = p_begin + row - (role_a ? 0 : 1);         // dY: the pixel; X: the aligned pixel
        const int tt = t_cur[i] + W * H;                     // >= 0; same (x, y) as t_cur
        const int rowi = tt / W;
        t_x[i] = tt - rowi * W;
        t_y[i] = rowi % H;
        // lanes without a staging job write their (zero) chunk into the dump area behind the images: the loop body stays one
        // basic block, so that the scheduler can interleave the split with the MFMAs
        st_off[i] = active[i] ? (role_a ? 0 : IMG) + row * PITCH + col * 2 : 6 * IMG + lane * 8;
    }
    f32x4 set[PF][NSUB];
    f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool bias_block = want_bias && ct == 0 && dyi == 0;
    const float bflag = (role_a && bias_block) ? 1.f : 0.f;
    // loads the K step that starts at pixel pt; MUST be called with pt = p_begin, p_begin + BKW, ... in order (branch-free)
    auto load_global = [&](int pt, f32x4 (&reg)[NSUB]) {
#pragma unroll
        for (int i = 0; i < NSUB; ++i) {
            const bool ok_a = t_cur[i] < p_end;
            const bool ok_b = (unsigned)t_cur[i] < (unsigned)P && (unsigned)(t_y[i] + dy) < (unsigned)H;
            const bool ok = active[i] && (role_a ? ok_a : ok_b);
            reg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? off_c[i] + pt * cs * 4 : (int)OOB, 0, 0));
            t_cur[i] += BKW;
            t_x[i] += BKW;
#pragma unroll
            for (int r = 0; r < NSUB; ++r) {                 // W >= 16: at most NSUB row wraps per step
                const int wx = t_x[i] >= W;
                t_x[i] -= wx ? W : 0;
                t_y[i] += wx;
                t_y[i] = t_y[i] == H ? 0 : t_y[i];
            }
        }
    };
    auto store_lds = [&](int buf_off, const f32x4 (&v)[NSUB]) {
#pragma unroll
        for (int i = 0; i < NSUB; ++i) {
            bsum += v[i] * bflag;
            store_split3<PLANE>(lds_raw, st_off[i] + (active[i] ? buf_off : 0), v[i]);
        }
    };

    const bool from_old = g.splits == 1 && accumulate;
    const int n_base = (dyi * 3 + tg) * g.Ci + c0 + wn;      // first column of this wave's block in [Co][9 Ci]
    f32x16 acc[1][1];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int n = n_base + frag_col(lane), m = m0 + wm + frag_row(lane, e);
        const float v = out[from_old ? (size_t)m * NC + n : 0];
        acc[0][0][e] = from_old ? v : 0.f;
    }

    if (p_begin < p_end) {
        // Software pipeline over the 16-pixel sub-steps (all 12 waves of the CU's one workgroup run in lock-step, so nothing
        // else hides an LDS round trip): while sub-step s is multiplied, the fragments of sub-step s+1 - the next one of this
        // K step, or the first one of the next K step - travel from LDS into the other register set.  Three LDS buffers: in
        // iteration t step t is read, step t+1 is read (its first sub-step) and step t+2 is written (into the buffer step t-1
        // left before the last barrier); the ring slot it came from takes step t+2+PF.  One barrier per K step.
        const int nsteps = (p_end - p_begin + BKW - 1) / BKW;
        int mx = p_begin % W;                                // image column of the first pixel of the sub-step whose fragment is masked next
#pragma unroll
        for (int d = 0; d < PF; ++d) load_global(p_begin + d * BKW, set[d]);
        store_lds(0, set[0]);
        load_global(p_begin + PF * BKW, set[0]);
        store_lds(2 * IMG, set[1 % PF]);
        load_global(p_begin + (PF + 1) * BKW, set[1 % PF]);
        __syncthreads();
        Frag3 fa[2], fb[2];
        const unsigned char* a_src = lds_raw + wm * 2;
        const unsigned char* b_src = lds_raw + IMG + (dx + 1) * PITCH + wn * 2;
        auto read_frags = [&](int buf_off, int ks, Frag3& a, Frag3& b) {
            Frag3 (&a1)[1] = *reinterpret_cast<Frag3 (*)[1]>(&a);
            Frag3 (&b1)[1] = *reinterpret_cast<Frag3 (*)[1]>(&b);
            read_kstrided3<1, PITCH, PLANE>(a_src + buf_off, lane, ks, a1);
            read_kstrided3<1, PITCH, PLANE>(b_src + buf_off, lane, ks, b1);
        };
        const int h8 = (lane >> 5) * 8;
        // clears, in a dY fragment, pixel ke of its 16-pixel block - the one whose tap dx leaves the image row (this lane holds
        // k = h8 .. h8 + 7).  Under a wave-uniform branch: the dx = 0 group never needs it, the others once per image row
        // (built unconditionally, these ~26 vector instructions per sub-step ate what the shared staging saves)
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
        read_frags(0, 0, fa[0], fb[0]);
        int o_cur = 0, o_nxt = 2 * IMG, o_st = 4 * IMG;       // byte offsets of the buffers of steps t, t+1, t+2
        auto iteration = [&](auto U, int tt) {
            constexpr int u = decltype(U)::value;             // tt % PF
            constexpr int slot = (u + 2) % PF;
#pragma unroll
            for (int ks = 0; ks < NSUB; ++ks) {
                const int cur = (u * NSUB + ks) & 1, nxt = cur ^ 1;
                {
                    const int ke = dx < 0 ? (mx == 0 ? 0 : W - mx) : W - 1 - mx;          // mx: image column of this sub-step's first pixel
                    if (dx != 0 && ke < BK) mask_edge(fa[cur], ke);
                    mx += BK;
                    mx -= mx >= W ? W : 0;
                }
                if (ks + 1 < NSUB) read_frags(o_cur, ks + 1, fa[nxt], fb[nxt]);
                else read_frags(o_nxt, 0, fa[nxt], fb[nxt]);
                Frag3 (&a)[1] = *reinterpret_cast<Frag3 (*)[1]>(&fa[cur]);
                Frag3 (&b)[1] = *reinterpret_cast<Frag3 (*)[1]>(&fb[cur]);
                mma3_step<1, 1>(a, b, acc);
            }
            store_lds(o_st, set[slot]);                       // step tt + 2
            load_global(p_begin + (tt + 2 + PF) * BKW, set[slot]);                        // past the end: fully masked
#pragma unroll
            for (int i = 0; i < 6; ++i) {                     // an MFMA, then a few of the split / address instructions, ...
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            }
            __syncthreads();
            const int o = o_cur; o_cur = o_nxt; o_nxt = o_st; o_st = o;
        };
        int t = 0;
        for (; t + PF <= nsteps; t += PF) unroll_iterations<PF>(iteration, t);
        if (PF > 1) tail_iterations<PF - 1>(iteration, t, nsteps);
    }
    const bool direct = g.splits == 1;
    float* dst = direct ? out : out + (size_t)blockIdx.z * ((size_t)g.Co * NC + g.Co);
#pragma unroll
    for (int e = 0; e < 16; ++e)
        dst[(size_t)(m0 + wm + frag_row(lane, e)) * NC + n_base + frag_col(lane)] = acc[0][0][e];
    if (bias_block) {
        // the dY staging threads hold the sums of rows kk + 16 i of every step: fold the 16 row classes through LDS
        float* fold = reinterpret_cast<float*>(lds_raw);
        __syncthreads();
        if (role_a) *reinterpret_cast<f32x4*>(fold + kk * 68 + col) = bsum;
        __syncthreads();
        if (tid < 64) {
            float tsum = 0.f;
#pragma unroll
            for (int k = 0; k < BK; ++k) tsum += fold[k * 68 + tid];
            if (direct) dbias[m0 + tid] = accumulate ? dbias[m0 + tid] + tsum : tsum;
            else dst[(size_t)g.Co * NC + m0 + tid] = tsum;
        }
    }
}

// ---- weight gradient of a Linear layer over few rows (P <= 256: the M = 240 layers of the lane head) ------------------
// dW[Co][Ci] (+)= dY^T X with the WHOLE reduction dimension staged in LDS at once: every global load of the workgroup is
// in flight together (one memory latency instead of one per 16-row step - the generic loop needs ~2 us per step, and
// splitting its 15 steps over workgroups costs a second launch), then 15 x 8 MFMAs back to back.  No split, no reduce:
// the result goes (or accumulates) straight into the gradient arena.  The bias gradient is the column sum of the staged
// dY tile.
constexpr int SMALLP_MAX = 256;
template <int BM, int BN, bool AMASK = false, int MMA = 0>
__device__ __forceinline__ void smallp_tile(
    const float* __restrict__ dY, const float* __restrict__ X, float* __restrict__ dw, float* __restrict__ dbias,
    int P, int Co, int Ci, int want_bias, int accumulate, float* lds, unsigned block, unsigned nblocks,
    const float* __restrict__ ymask = nullptr)
{
    constexpr int TM = BM / 2, TN = BN / 2, FM = TM / 32, FN = TN / 32;
    constexpr int AP = BM + 4, BP = BN + 4;
    constexpr int A_CH = BM / 4, B_CH = BN / 4;                     // float4 chunks per staged row
    constexpr int A_LOADS = SMALLP_MAX * A_CH / THREADS, B_LOADS = SMALLP_MAX * B_CH / THREADS;
    const int P16 = (P + 15) & ~15;
    float* As = lds;
    float* Bs = lds + P16 * AP;
    const int tiles_n = (Ci + BN - 1) / BN;
    const unsigned tile = xcd_remap(block, nblocks);
    const int m0 = (int)(tile / tiles_n) * BM, n0 = (int)(tile % tiles_n) * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * TM, wn = (wave & 1) * TN;

    f32x4 ar[A_LOADS], br[B_LOADS], ay[AMASK ? A_LOADS : 1];
    unsigned am = 0, bm = 0;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {                            // branch-free: masked chunks read element 0
        const int idx = tid + THREADS * i, p = idx / A_CH, m = m0 + (idx - p * A_CH) * 4;
        const bool ok = p < P && m < Co;
        ar[i] = *reinterpret_cast<const f32x4*>(dY + (ok ? (size_t)p * Co + m : 0));
        if (AMASK) ay[i] = *reinterpret_cast<const f32x4*>(ymask + (ok ? (size_t)p * Co + m : 0));
        am |= (unsigned)ok << i;
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const int idx = tid + THREADS * i, p = idx / B_CH, n = n0 + (idx - p * B_CH) * 4;
        const bool ok = p < P && n < Ci;
        br[i] = *reinterpret_cast<const f32x4*>(X + (ok ? (size_t)p * Ci + n : 0));
        bm |= (unsigned)ok << i;
    }
    // accumulate mode: the accumulators START from the old gradient values, fetched now so that their (cold, HBM) latency
    // overlaps the staging instead of sitting in a read-modify-write epilogue
    f32x16 acc[FM][FN];
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wn + j * 32 + frag_col(lane), m = m0 + wm + i * 32 + frag_row(lane, e);
                acc[i][j][e] = (accumulate && n < Ci && m < Co) ? dw[(size_t)m * Ci + n] : 0.f;
            }
    const f32x4 zero{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int idx = tid + THREADS * i, p = idx / A_CH;
        f32x4 v = (am >> i) & 1u ? ar[i] : zero;
        if (AMASK) {
            v.x = ay[i].x > 0.f ? v.x : 0.f; v.y = ay[i].y > 0.f ? v.y : 0.f;
            v.z = ay[i].z > 0.f ? v.z : 0.f; v.w = ay[i].w > 0.f ? v.w : 0.f;
        }
        if (p < P16) *reinterpret_cast<f32x4*>(As + p * AP + (idx - p * A_CH) * 4) = v;
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const int idx = tid + THREADS * i, p = idx / B_CH;
        if (p < P16) *reinterpret_cast<f32x4*>(Bs + p * BP + (idx - p * B_CH) * 4) = (bm >> i) & 1u ? br[i] : zero;
    }
    __syncthreads();

    for (int ks = 0; ks < P16 / BK; ++ks) {
        float a[FM][8], b[FN][8];
        read_kstrided<FM, AP>(As + wm, lane, ks, a);
        read_kstrided<FN, BP>(Bs + wn, lane, ks, b);
        mma_any<MMA, FM, FN>(a, b, acc);
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn + j * 32 + frag_col(lane);
        if (n >= Ci) continue;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm + i * 32 + frag_row(lane, e);
                if (m < Co) dw[(size_t)m * Ci + n] = acc[i][j][e];
            }
    }
    if (want_bias && n0 == 0) {                                      // column sums of the staged dY tile: 4 row groups, then fold
        static_assert(BM == 64, "bias fold assumes 64 columns x 4 row groups");
        const int c = tid & 63, grp = tid >> 6;
        float t = 0.f;
#pragma unroll 4
        for (int p = grp; p < P16; p += 4) t += As[p * AP + c];
        __syncthreads();                                            // every wave is done with the MFMA reads of Bs
        Bs[grp * 64 + c] = t;
        __syncthreads();
        if (tid < 64 && m0 + tid < Co) {
            const float v = (Bs[tid] + Bs[64 + tid]) + (Bs[128 + tid] + Bs[192 + tid]);
            dbias[m0 + tid] = accumulate ? dbias[m0 + tid] + v : v;
        }
    }
}

template <int BM, int BN, int MMA = 0>
__global__ __launch_bounds__(THREADS) void linear_wgrad_smallp_kernel(
    const float* __restrict__ dY, const float* __restrict__ X, float* __restrict__ dw, float* __restrict__ dbias,
    int P, int Co, int Ci, int want_bias, int accumulate)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    smallp_tile<BM, BN, false, MMA>(dY, X, dw, dbias, P, Co, Ci, want_bias, accumulate, lds, blockIdx.x, gridDim.x);
}

// Backward of a Linear layer over few rows in ONE launch: the first `dgrad_tiles` workgroups compute the data gradient
// dX = dY W (tiles of the implicit-GEMM program, K tile 64, unsplit), the others the weight (and bias) gradient with the
// few-rows program above.  The two are independent - both only read dY - and each occupies a handful of CUs, so sharing a
// launch removes one ~8 us dependent launch per layer (there are ~270 such layers in a step).
// AMASK: dY is the gradient of relu(x w^T + b); ymask is that layer's saved output and both roles apply the ReLU mask while
// they stage dY (no separate relu-backward launch, no masked copy of dY in memory).
template <bool UNI, bool AMASK, int MMA = 0>
__global__ __launch_bounds__(THREADS) void linear_bwd_fused_kernel(
    const float* __restrict__ dY, const float* __restrict__ W, const float* __restrict__ X, const float* __restrict__ ymask,
    float* __restrict__ dX, float* __restrict__ dw, float* __restrict__ dbias,
    ConvShape gd, unsigned dgrad_tiles, int P, int Co, int Ci, int want_bias, int accumulate)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (blockIdx.x < dgrad_tiles)
        igemm_tile<64, 64, true, 64, UNI, AMASK, MMA>(dY, W, nullptr, nullptr, dX, gd, 0, lds, blockIdx.x, dgrad_tiles, 0, ymask);
    else
        smallp_tile<64, 64, AMASK, MMA>(dY, X, dw, dbias, P, Co, Ci, want_bias, accumulate, lds, blockIdx.x - dgrad_tiles,
                                   gridDim.x - dgrad_tiles, ymask);
}

// dW (+)= sum_z part[z][0:nw],  dbias (+)= sum_z part[z][nw:nw+nb]   (part rows are nw+nb floats long; nw, nb multiples of 4)
// 256 threads = 64 float4 columns x 4 split groups: group g sums the partials z = g, g+4, ... with four loads in flight, the
// four group sums are folded through LDS in a fixed order (deterministic).  The small trunk layers run with 40-190 splits;
// one thread per element walking all of them paid one L2 latency per split (22-62 us per launch, now a few).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                           float* __restrict__ dbias, long nw, long nb, int splits, int accumulate)
{
    __shared__ f32x4 red[3][64];
    const int lane = threadIdx.x & 63, zg = threadIdx.x >> 6;
    const long total4 = (nw + nb) >> 2;
    const long i4 = (long)blockIdx.x * 64 + lane;
    const bool ok = i4 < total4;
    const long i = i4 * 4;
    const bool is_w = i < nw;
    const bool writer = zg == 0 && ok && (is_w || dbias);
    const f32x4 zero{0.f, 0.f, 0.f, 0.f};
    f32x4 old = zero;
    if (writer && accumulate) {                           // cold read first: it overlaps the partial sums
        if (is_w) old = *reinterpret_cast<const f32x4*>(dw + i);
        else { const float* b = dbias + (i - nw); old = f32x4{b[0], b[1], b[2], b[3]}; }
    }
    const f32x4* p = reinterpret_cast<const f32x4*>(part) + (ok ? i4 : 0);
    f32x4 s = zero;
    int z = zg;
    for (; z + 12 < splits; z += 16) {
        const f32x4 v0 = p[(long)z * total4], v1 = p[(long)(z + 4) * total4], v2 = p[(long)(z + 8) * total4],
                    v3 = p[(long)(z + 12) * total4];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; z < splits; z += 4) s += p[(long)z * total4];
    if (zg) red[zg - 1][lane] = s;
    __syncthreads();
    if (writer) {
        const f32x4 t = old + ((s + red[0][lane]) + (red[1][lane] + red[2][lane]));
        if (is_w) *reinterpret_cast<f32x4*>(dw + i) = t;
        else { float* b = dbias + (i - nw); b[0] = t.x; b[1] = t.y; b[2] = t.z; b[3] = t.w; }
    }
}

// ---- stem helpers: NCHW (3 ch) -> NHWC padded to 4 channels; OHWI weight pad 3->4 and back -----------
__global__ void nchw3_to_nhwc4_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int HW)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * HW) return;
    const long n = i / HW, p = i - n * HW;
    const float* s = x + n * 3 * (long)HW + p;
    reinterpret_cast<f32x4*>(y)[i] = f32x4{s[0], s[HW], s[2 * (long)HW], 0.f};
}

__global__ void pad_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows, int cs, int cd)
{
    // dst[row][0..cd) = src[row][0..cs) zero-extended (cd > cs) or truncated (cd < cs)
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cd) return;
    const long r = i / cd;
    const int c = (int)(i - r * cd);
    dst[i] = c < cs ? src[r * cs + c] : 0.f;
}

struct TileChoice { int bm, bn; };

int g_mma_mode = 3;                                          // 3 (default): exact 3-term bf16 split at staging; 0: f32-input MFMA; 1 / 2: bf16 splits in registers (phnet_tune_mma)
int g_buf_loads = 1;                                         // buffer-load operand path where it applies (tuning: phnet_tune_force_k_tile(-200 / -201))
int g_pf = 4;                                                // register prefetch depth in K tiles (tuning: phnet_tune_force_k_tile(-101 / -102 / -104))
int g_deep_kt3 = 64;                                         // K tile of the few-rows GEMMs in mode 3 (tuning: phnet_tune_force_k_tile(-32 / -64))

TileChoice pick_tile(long M, long N)
{
    // Measured on MI355X (tests/tools/bench_conv.py): the problems of this path are small (a 5-frame clip), so what
    // matters is the number of co-resident workgroups per CU, not the tile's arithmetic intensity: 64x64 tiles with
    // the block count topped up to ~1250 by split-K beat the larger tiles on every trunk layer (72-80 us vs 85-130 us).
    if (g_mma_mode == 3 && M <= 2048) return {64, 64};          // (64x64 is the fastest tile up to the 1200 rows of a batched clip)
    if (g_mma_mode >= 1) {
        // split-bf16: the loop is bound by the operand split (VALU) and the LDS reads per MFMA, both of which shrink with
        // the wave tile - problems with enough tiles take the larger ones (bench_conv.py --mma --clips 8: 128x128 is
        // 15-20 % faster than 64x64 on the 8-clip trunk layers, and slower on every 1-clip layer)
        if (N >= 128 && ceil_div64(M, 128) * ceil_div64(N, 128) >= 300) return {128, 128};
        if (ceil_div64(M, 128) * ceil_div64(N, 64) >= 1250) return {128, 64};
    }
    return {64, 64};
}

struct ConvPlan { int bm, bn, splits; long tiles; };

// K-tile depth: 64 for the skinny (latency-bound) GEMMs of the lane head, 16 otherwise; split-bf16 on 64x64 tiles: 32
// (the MFMA work per barrier is 5x shorter there; measured 44-54 us vs 51-58 us on the trunk layers)
int k_tile_for(long M, int K, int ci, int bm, int bn)
{
    // few-rows GEMMs (one frame of the lane head: 240 rows): deep K tiles.  With the frames of a clip batched (1200 rows) the
    // trunk plan wins in the staged-split arithmetic: 1024 -> 8192 174 vs 260 us, 4608 -> 1024 96 vs 146 us (bench_conv --only clipB)
    if (M <= (g_mma_mode == 3 ? 256 : 2048) && K >= 64) return (g_mma_mode == 3 && g_deep_kt3 == 32) ? 32 : 64;
    if (g_mma_mode == 3) return BK;                                   // 37 KB of LDS per 64x64 workgroup: four per CU
    if (g_mma_mode >= 1 && bm == 64 && bn == 64 && ci % 32 == 0) return 32;
    return BK;
}

int g_force_bm = 0, g_force_bn = 0, g_force_splits = 0;      // tuning aid (phnet_tune_force_conv_tile)
int g_force_kt = 0;                                          // tuning aid (phnet_tune_force_k_tile)
int g_uniform_tap = 1;                                       // tuning aid (phnet_tune_force_k_tile(-1) switches the uniform-tap variant off)
int g_wgrad_bm128 = 1, g_wgrad_target = 768;                // tuning aids (phnet_tune_wgrad)
int g_wgrad_bkw = 16;                                        // tuning aid: pixels per K step of the staged-split wgrad kernel (phnet_tune_wgrad bit 2 of arg 0 -> 32)
int g_wgrad_smallp = 1;                                      // bit 1 of phnet_tune_wgrad's first argument switches the few-rows kernel off
int g_smallp_max_tiles = 400;                                 // measured: 64 -> 400 tiles saves 0.75 ms per step, 1300 nothing more

ConvPlan plan_conv(long M, int Co, int K, bool has_ws, size_t ws_bytes)
{
    TileChoice t = pick_tile(M, Co);
    if (g_force_bm) {
        t = {g_force_bm, g_force_bn};
        ConvPlan f{t.bm, t.bn, 1, ceil_div64(M, t.bm) * ceil_div64(Co, t.bn)};
        int splits = g_force_splits > 0 ? g_force_splits : 1;
        while (splits > 1 && (!has_ws || (size_t)splits * M * Co * sizeof(float) > ws_bytes)) --splits;
        f.splits = splits;
        return f;
    }
    ConvPlan p{t.bm, t.bn, 1, ceil_div64(M, t.bm) * ceil_div64(Co, t.bn)};
    // split K until ~1250 workgroups are in flight (about 5 per CU), keeping >= 256 of K per split
    // K-tile-64 plans (M <= 2048: the head GEMMs) hold 70 KB of LDS per workgroup = 2 workgroups per CU, so ~512 tiles already
    // fill the chip in one round and a split only adds the reduce pass (hyper-net 1024 -> 8192 at 240 rows: 54 vs 62 us)
    const bool deep = M <= (g_mma_mode == 3 ? 256 : 2048) && K >= 64;
    if (has_ws && p.tiles < (deep ? 400 : 900)) {
        int splits = (int)min((long)8, max((long)1, (1250 + p.tiles / 2) / p.tiles));
        while (splits > 1 && K / splits < 256) --splits;
        while (splits > 1 && (size_t)splits * M * Co * sizeof(float) > ws_bytes) --splits;
        p.splits = splits;
    }
    return p;
}

template <bool DGRAD>
int launch_conv(const float* X, const float* W, const float* bias, const float* addend, float* out, float* workspace,
                size_t ws_bytes, ConvShape g, int relu, hipStream_t st, float* stats = nullptr)
{
    const long M = (long)g.N * g.Ho * g.Wo;
    const int K = g.R * g.S * g.Ci;
    const ConvPlan t = plan_conv(M, g.Co, K, workspace != nullptr, ws_bytes);
    const long tiles = t.tiles;
    const int splits = t.splits;
    g.splits = splits;
    float* dst = splits > 1 ? workspace : out;
    dim3 grid((unsigned)tiles, 1, (unsigned)splits);
    // deep K tiles for skinny (latency-bound) problems: few row tiles and K long enough to fill them
    const int bkt = g_force_kt ? g_force_kt : k_tile_for(M, K, g.Ci, t.bm, t.bn);
    const int ksteps = (K + bkt - 1) / bkt;
    g.k_per_split = ((ksteps + splits - 1) / splits) * bkt;
#define PHNET_LAUNCH_CONV___(BM_, BN_, BKT_, UNI_, MMA_, PF_)                                                           \
    do {                                                                                                                \
        if (UNI_ && g.in_dil == 1 && g_buf_loads && MMA_ == 3) PHNET_LAUNCH_CONV____(BM_, BN_, BKT_, UNI_, MMA_, PF_, UNI_); \
        else PHNET_LAUNCH_CONV____(BM_, BN_, BKT_, UNI_, MMA_, PF_, false);                                             \
    } while (0)
#define PHNET_LAUNCH_CONV____(BM_, BN_, BKT_, UNI_, MMA_, PF_, BUF_)                                                    \
    do {                                                                                                                \
        constexpr size_t lds_ = MMA_ == 3                                                                               \
            ? 2 * (size_t)(KContigPlanes<BM_, BKT_>::BYTES +                                                            \
                           (DGRAD ? KStridedPlanes<BN_, BKT_>::BYTES : KContigPlanes<BN_, BKT_>::BYTES))                \
            : 2 * (size_t)(KContigTile<BM_, BKT_>::FLOATS +                                                             \
                           (DGRAD ? KStridedTile<BN_, BKT_>::FLOATS : KContigTile<BN_, BKT_>::FLOATS)) * 4;             \
        static bool attr_set_ = false;                                                                                  \
        if (lds_ > 64 * 1024 && !attr_set_) {                                                                           \
            (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<BM_, BN_, DGRAD, BKT_, UNI_, MMA_, PF_, BUF_>,     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_);                                 \
            attr_set_ = true;                                                                                           \
        }                                                                                                               \
        hipLaunchKernelGGL((conv_igemm_kernel<BM_, BN_, DGRAD, BKT_, UNI_, MMA_, PF_, BUF_>), grid, dim3(THREADS), lds_, st, \
                           X, W, bias, addend, dst, g, relu, splits > 1 ? (float*)nullptr : stats);                     \
    } while (0)
#define PHNET_LAUNCH_CONV__(BM_, BN_, BKT_, UNI_, MMA_) PHNET_LAUNCH_CONV___(BM_, BN_, BKT_, UNI_, MMA_, 1)
#define PHNET_LAUNCH_CONV_(BM_, BN_, BKT_, UNI_)                                                                        \
    do {                                                                                                                \
        if (g_mma_mode == 1) PHNET_LAUNCH_CONV__(BM_, BN_, BKT_, UNI_, 1);                                              \
        else if (g_mma_mode == 2) PHNET_LAUNCH_CONV__(BM_, BN_, BKT_, UNI_, 2);                                         \
        else if (g_mma_mode == 3 && g_pf == 4) PHNET_LAUNCH_CONV___(BM_, BN_, BKT_, UNI_, 3, 4);                        \
        else if (g_mma_mode == 3 && g_pf == 2) PHNET_LAUNCH_CONV___(BM_, BN_, BKT_, UNI_, 3, 2);                        \
        else if (g_mma_mode == 3) PHNET_LAUNCH_CONV___(BM_, BN_, BKT_, UNI_, 3, 1);                                     \
        else if (g_pf == 4) PHNET_LAUNCH_CONV___(BM_, BN_, BKT_, UNI_, 0, 4);                                           \
        else PHNET_LAUNCH_CONV__(BM_, BN_, BKT_, UNI_, 0);                                                              \
    } while (0)
#define PHNET_LAUNCH_CONV(BM_, BN_, UNI_)                                                                               \
    do {                                                                                                                \
        if (bkt == 64) PHNET_LAUNCH_CONV_(BM_, BN_, 64, UNI_);                                                          \
        else if (bkt == 32) PHNET_LAUNCH_CONV_(BM_, BN_, 32, UNI_);                                                     \
        else PHNET_LAUNCH_CONV_(BM_, BN_, 16, UNI_);                                                                    \
    } while (0)
    // uniform-tap variant (the production tile only): the A-side channel count is a multiple of the K tile
    const bool uni = (g.Ci % bkt) == 0 && g_uniform_tap;
    if (g_taps3 && uni && g.R == 3 && g.S == 3 && g.stride == 1 && g.pad == 1 && g.in_dil == 1 && g.Ho == g.Hi && g.Wo == g.Wi &&
        t.bm == 64 && t.bn == 64 && bkt == 16 && g_mma_mode == 3 && g_buf_loads && g_pf == 4 && g.Wi >= 2 &&
        M * (long)g.Ci * 4 < 0x7fffffffL && (long)g.Co * K * 4 < 0x7fffffffL) {
        const int units = 3 * (g.Ci / 16);
        g.k_per_split = (units + splits - 1) / splits;                   // in units for this kernel
        constexpr size_t lds_ = 2 * 3 * T3_AROWS * KContigPlanes<64, 16>::PITCH +
                                2 * (size_t)(DGRAD ? KStridedPlanes<64, 16>::BYTES : KContigPlanes<64, 16>::BYTES) + 3 * 512;
        hipLaunchKernelGGL((conv3x3s1_kernel<DGRAD>), grid, dim3(THREADS), lds_, st,
                           X, W, bias, addend, dst, g, relu, splits > 1 ? (float*)nullptr : stats);
    } else
    {
    if (t.bm == 128 && t.bn == 128) PHNET_LAUNCH_CONV(128, 128, false);
    else if (t.bm == 128 && t.bn == 64) PHNET_LAUNCH_CONV(128, 64, false);
    else if (t.bm == 64 && t.bn == 128) PHNET_LAUNCH_CONV(64, 128, false);
    else if (uni) PHNET_LAUNCH_CONV(64, 64, true);
    else PHNET_LAUNCH_CONV(64, 64, false);
    }
#undef PHNET_LAUNCH_CONV____
#undef PHNET_LAUNCH_CONV___
#undef PHNET_LAUNCH_CONV__
#undef PHNET_LAUNCH_CONV_
#undef PHNET_LAUNCH_CONV
    if (splits > 1) {
        const long total4 = M * g.Co / 4;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)ceil_div64(total4, 256)), dim3(256), 0, st,
                           workspace, out, bias, addend, total4, g.Co, splits, relu, 0, stats);
    }
    return phnet_launch_status();
}

}  // namespace

// Which kernel instantiation / split-K factor phnet_conv2d_fwd / _dgrad will use for a GEMM of M x Co x K
// (profiling aid for bench.py: lets the host attribute event timings to one kernel symbol).
PHNET_API int phnet_conv2d_plan(int64_t M, int32_t Co, int32_t K, uint64_t ws_bytes, int32_t* bm, int32_t* bn, int32_t* splits,
                                int32_t* k_tile)
{
    if (M < 1 || Co < 1 || K < 1 || !bm || !bn || !splits || !k_tile) return PHNET_ERR_ARG;
    const ConvPlan p = plan_conv((long)M, Co, K, ws_bytes > 0, (size_t)ws_bytes);
    *bm = p.bm; *bn = p.bn; *splits = p.splits;
    *k_tile = g_force_kt ? g_force_kt : k_tile_for((long)M, K, K, p.bm, p.bn);   // (K stands in for the channel count)
    return PHNET_OK;
}

// Tuning aid (process-global, not thread-safe, never used by the product path): force the tile (64|128 each) and
// split-K factor of the next phnet_conv2d_fwd/_dgrad calls; bm = 0 restores the built-in heuristic.
PHNET_API int phnet_tune_force_conv_tile(int32_t bm, int32_t bn, int32_t splits)
{
    if (bm != 0 && !((bm == 64 || bm == 128) && (bn == 64 || bn == 128))) return PHNET_ERR_ARG;
    g_force_bm = bm; g_force_bn = bn; g_force_splits = splits;
    return PHNET_OK;
}

PHNET_API int phnet_tune_wgrad(int32_t allow_bm128, int32_t target_blocks)
{
    if (target_blocks == 0 || target_blocks < -1024) return PHNET_ERR_ARG;
    g_wgrad_bm128 = allow_bm128 & 1; g_wgrad_smallp = !(allow_bm128 & 2); g_wgrad_bkw = (allow_bm128 & 4) ? 32 : 16;
    g_wgrad3 = !(allow_bm128 & 8);
    g_wgrad3_bkw = (allow_bm128 & 16) ? 32 : 16;
    g_wgrad3s = !(allow_bm128 & 32);
    g_wgrad1s = !(allow_bm128 & 64);
    if (target_blocks < 0) g_wgrad3_target = -target_blocks;      // workgroup target of the three-taps 3x3 kernel
    else g_wgrad_target = target_blocks;
    return PHNET_OK;
}


// Tuning aid (process-global): arithmetic of the GEMM kernels.  0 = f32-input MFMA (default), 1 = split-bf16 (igemm.h).
PHNET_API int phnet_tune_mma(int32_t mode)
{
    if (mode < 0 || mode > 3) return PHNET_ERR_ARG;
    g_mma_mode = mode;
    g_pf = mode == 3 ? 4 : 1;                                  // the staged-split loop runs with a 4-tile register ring
    return PHNET_OK;
}

PHNET_API int phnet_tune_force_k_tile(int32_t kt)
{
    if (kt == -1 || kt == -2) { g_uniform_tap = kt == -2; return PHNET_OK; }     // -1: uniform-tap variant off, -2: on again
    if (kt == -5 || kt == -6) { g_taps3 = kt == -6; return PHNET_OK; }          // -5: three-taps 3x3 forward / dgrad kernel off, -6: on again
    if (kt == -32 || kt == -64) { g_deep_kt3 = -kt; return PHNET_OK; }
    if (kt == -101 || kt == -102 || kt == -104) { g_pf = -kt - 100; return PHNET_OK; }
    if (kt == -200 || kt == -201) { g_buf_loads = -kt - 200; return PHNET_OK; }
    if (kt != 0 && kt != 16 && kt != 32 && kt != 64) return PHNET_ERR_ARG;
    g_force_kt = kt;
    return PHNET_OK;
}

// Forward convolution / linear.  x NHWC [N][Hi][Wi][Ci], w OHWI [Co][R][S][Ci], bias [Co] or NULL,
// y NHWC [N][Ho][Wo][Co].  Ci % 4 == 0, Co % 4 == 0.  workspace (optional, for split-K) is caller-allocated.
PHNET_API int phnet_conv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                               int32_t N, int32_t Hi, int32_t Wi, int32_t Ci, int32_t Co, int32_t R, int32_t S,
                               int32_t stride, int32_t pad, int32_t relu,
                               void* workspace, uint64_t ws_bytes, void* stream)
{
    if (N < 0 || Hi < 1 || Wi < 1 || Ci < 4 || Co < 4 || (Ci & 3) || (Co & 3) || R < 1 || S < 1 || stride < 1 || pad < 0)
        return PHNET_ERR_ARG;
    if (N == 0) return PHNET_OK;
    if (!x || !w || !y) return PHNET_ERR_ARG;
    ConvShape g{};
    g.N = N; g.Hi = Hi; g.Wi = Wi; g.Ci = Ci; g.Co = Co; g.R = R; g.S = S;
    g.Ho = (Hi + 2 * pad - R) / stride + 1;
    g.Wo = (Wi + 2 * pad - S) / stride + 1;
    if (g.Ho < 1 || g.Wo < 1) return PHNET_ERR_ARG;
    g.stride = stride; g.pad = pad; g.in_dil = 1;
    return launch_conv<false>(x, w, bias, nullptr, y, (float*)workspace, ws_bytes, g, relu, (hipStream_t)stream);
}

// Forward convolution with a fused epilogue: y = conv(x, w) + bias + addend, then ReLU (each part optional; addend shaped
// like y - the residual branch of a folded BatchNorm block in eval mode), and / or the per-channel batch statistics of y
// for a following BatchNorm: stats = phnet_conv2d_stats_blocks() rows of 2*Co floats (sum | sum of squares), the `partial`
// layout phnet_bn_finalize_partials reads.  stats needs bias == addend == NULL, relu == 0 and Co a power of two <= 1024.
static long conv_stats_blocks(long M, int Co, int K, bool has_ws, size_t ws_bytes)
{
    const ConvPlan t = plan_conv(M, Co, K, has_ws, ws_bytes);
    return t.splits > 1 ? ceil_div64(M * Co / 4, 256) : t.tiles / ceil_div64(Co, t.bn) * (t.bm / 32);
}

PHNET_API uint64_t phnet_conv2d_stats_blocks(int64_t M, int32_t Co, int32_t K, uint64_t ws_bytes)
{
    if (M < 1 || Co < 4 || K < 1) return 0;
    return (uint64_t)conv_stats_blocks((long)M, Co, K, ws_bytes > 0, (size_t)ws_bytes);
}

PHNET_API int phnet_conv2d_fwd_fused(const float* x, const float* w, const float* bias, const float* addend, float* y, float* stats,
                                     int32_t N, int32_t Hi, int32_t Wi, int32_t Ci, int32_t Co, int32_t R, int32_t S,
                                     int32_t stride, int32_t pad, int32_t relu,
                                     void* workspace, uint64_t ws_bytes, void* stream)
{
    if (N < 0 || Hi < 1 || Wi < 1 || Ci < 4 || Co < 4 || (Ci & 3) || (Co & 3) || R < 1 || S < 1 || stride < 1 || pad < 0)
        return PHNET_ERR_ARG;
    if (stats && (bias || addend || relu || (Co & (Co - 1)) || Co > 1024)) return PHNET_ERR_ARG;
    if (N == 0) return PHNET_OK;
    if (!x || !w || !y) return PHNET_ERR_ARG;
    ConvShape g{};
    g.N = N; g.Hi = Hi; g.Wi = Wi; g.Ci = Ci; g.Co = Co; g.R = R; g.S = S;
    g.Ho = (Hi + 2 * pad - R) / stride + 1;
    g.Wo = (Wi + 2 * pad - S) / stride + 1;
    if (g.Ho < 1 || g.Wo < 1) return PHNET_ERR_ARG;
    g.stride = stride; g.pad = pad; g.in_dil = 1;
    return launch_conv<false>(x, w, bias, addend, y, (float*)workspace, ws_bytes, g, relu, (hipStream_t)stream, stats);
}

// Data gradient.  dy NHWC [N][Ho][Wo][Co], w OHWI [Co][R][S][Ci] (as used by the forward), dx NHWC [N][Hi][Wi][Ci].
// addend (optional, same shape as dx, may alias dx): dx = dgrad + addend  (residual-branch gradient).
PHNET_API int phnet_conv2d_dgrad(const float* dy, const float* w, const float* addend, float* dx,
                                 int32_t N, int32_t Hi, int32_t Wi, int32_t Ci, int32_t Co, int32_t R, int32_t S,
                                 int32_t stride, int32_t pad, void* workspace, uint64_t ws_bytes, void* stream)
{
    if (N < 0 || Hi < 1 || Wi < 1 || Ci < 4 || Co < 4 || (Ci & 3) || (Co & 3) || R < 1 || S < 1 || stride < 1 || pad < 0)
        return PHNET_ERR_ARG;
    if (stride > 2) return PHNET_ERR_ARG;          // the data gradient supports strides 1 and 2 (all the model has)
    if (N == 0) return PHNET_OK;
    if (!dy || !w || !dx) return PHNET_ERR_ARG;
    ConvShape g{};
    g.N = N;
    g.Hi = (Hi + 2 * pad - R) / stride + 1;      // A-side image = dY
    g.Wi = (Wi + 2 * pad - S) / stride + 1;
    g.Ci = Co;
    g.Ho = Hi; g.Wo = Wi; g.Co = Ci;              // GEMM output = dX
    g.R = R; g.S = S;
    g.stride = 1; g.pad = R - 1 - pad; g.in_dil = stride;
    if (R != S && (S - 1 - pad) != g.pad) return PHNET_ERR_ARG;   // square padding only
    if (g.pad < 0) return PHNET_ERR_ARG;
    return launch_conv<true>(dy, w, nullptr, addend, dx, (float*)workspace, ws_bytes, g, 0, (hipStream_t)stream);
}

// Weight gradient.  dw OHWI [Co][R][S][Ci] is overwritten (accumulate=0) or added to (accumulate=1).
// workspace must hold splits*Co*R*S*Ci floats; query with phnet_conv2d_wgrad_workspace.
static long wgrad_splits(long P, long Co, long NC, int* bm_out)
{
    // measured (bench_conv.py --trunk --wgrad, reduce included): on a 5-frame clip 128-row tiles win at Co = 128 only (73 vs 79 us);
    // at Co = 256 / 512 (5000 / 1250 pixels) the 64x64 tile with more splits is 5 / 3 us faster; with 8 clips per step
    // (40000 / 10000 pixels) the 128-row tile wins everywhere (step 129.7 vs 132.4 ms)
    const int bm = (Co >= 128 && (Co < 256 || P > 8192) && g_wgrad_bm128) ? 128 : 64;
    const long tiles = ceil_div64(Co, bm) * ceil_div64(NC, 64);
    const long target = tiles >= 64 ? max((long)g_wgrad_target, (long)1250) : (long)g_wgrad_target;   // measured: bench_conv --wgrad
    long splits = max((long)1, min((long)256, target / max((long)1, tiles)));
    splits = max((long)1, min(splits, P / 64));
    // the M=240 linears of the lane head (15 K steps in all): with >= 32 tiles a split only adds a reduce launch;
    // unsplit launches also accumulate straight into the gradient arena
    if (P <= 1024 && tiles >= 32) splits = 1;
    if (bm_out) *bm_out = bm;
    return splits;
}

// the three-taps kernel (conv_wgrad3x3_kernel): 3x3 / stride 1 / pad 1, channel counts in whole 64-tiles, bf16x3 arithmetic
static bool wgrad3_applies(long P, int Hi, int Wi, int Ci, int Co, int R, int S, int stride, int pad)
{
    return g_wgrad3 && g_mma_mode == 3 && R == 3 && S == 3 && stride == 1 && pad == 1 && (Ci & 63) == 0 && (Co & 63) == 0 &&
           Wi >= BK && P >= 64 && P * (long)max(Ci, Co) * 4 < 0x7fffffffL;
}
static long wgrad3_splits(long P, long Co, long Ci)
{
    const long tiles = (Co / 64) * (Ci / 64) * 3;
    return max((long)1, min(min((long)256, P / 64), (g_wgrad3_target + tiles / 2) / tiles));
}

PHNET_API uint64_t phnet_conv2d_wgrad_workspace(int32_t N, int32_t Hi, int32_t Wi, int32_t Ci, int32_t Co,
                                                int32_t R, int32_t S, int32_t stride, int32_t pad)
{
    const long Ho = (Hi + 2 * pad - R) / stride + 1, Wo = (Wi + 2 * pad - S) / stride + 1;
    const long P = (long)N * Ho * Wo, NC = (long)R * S * Ci;
    long splits = wgrad_splits(P, Co, NC, nullptr);
    if (R == 3 && S == 3 && stride == 1 && pad == 1 && (Ci & 63) == 0 && (Co & 63) == 0)     // whichever kernel the call picks
        splits = max(splits, min(min((long)256, max((long)1, P / 64)), (long)(1024 / ((Co / 64) * (Ci / 64) * 3) + 1)));
    return (uint64_t)(splits * (Co * NC + Co) * sizeof(float));
}

// dw OHWI [Co][R][S][Ci] and (optionally) dbias [Co] = sum of dy over all pixels are overwritten (accumulate=0) or
// added to (accumulate=1).  workspace: phnet_conv2d_wgrad_workspace bytes (split-K partial sums; unused when the
// problem runs unsplit).
PHNET_API int phnet_conv2d_wgrad(const float* dy, const float* x, float* dw, float* dbias,
                                 int32_t N, int32_t Hi, int32_t Wi, int32_t Ci, int32_t Co, int32_t R, int32_t S,
                                 int32_t stride, int32_t pad, int32_t accumulate,
                                 void* workspace, uint64_t ws_bytes, void* stream)
{
    if (N < 0 || Hi < 1 || Wi < 1 || Ci < 4 || Co < 4 || (Ci & 3) || (Co & 3) || R < 1 || S < 1 || stride < 1 || pad < 0)
        return PHNET_ERR_ARG;
    if (!dy || !x || !dw) return PHNET_ERR_ARG;
    WgradShape g{};
    g.N = N; g.Hi = Hi; g.Wi = Wi; g.Ci = Ci; g.Co = Co; g.R = R; g.S = S; g.stride = stride; g.pad = pad;
    g.Ho = (Hi + 2 * pad - R) / stride + 1;
    g.Wo = (Wi + 2 * pad - S) / stride + 1;
    const long P = (long)N * g.Ho * g.Wo, NC = (long)R * S * Ci;
    hipStream_t st = (hipStream_t)stream;
    // Linear over few rows with up to a few hundred output tiles (with thousands of tiles the generic kernel keeps several
    // workgroups per CU in flight, which hides the same latency; this one needs 139 KB of LDS = one workgroup per CU)
    if (R == 1 && S == 1 && stride == 1 && pad == 0 && P <= SMALLP_MAX && g_wgrad_smallp &&
        ceil_div64(Co, 64) * ceil_div64(Ci, 64) < g_smallp_max_tiles) {
        const int P16 = ((int)P + 15) & ~15;
        const size_t lds = (size_t)P16 * (64 + 4) * 2 * sizeof(float);
        static bool attr = false;
        if (!attr) {
            const int cap = (int)((size_t)SMALLP_MAX * 68 * 2 * sizeof(float));
            if (hipFuncSetAttribute((const void*)linear_wgrad_smallp_kernel<64, 64, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
                hipFuncSetAttribute((const void*)linear_wgrad_smallp_kernel<64, 64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess)
                return PHNET_ERR_LAUNCH;
            attr = true;
        }
        const long tiles = ceil_div64(Co, 64) * ceil_div64(Ci, 64);
        if (g_mma_mode == 1)
            hipLaunchKernelGGL((linear_wgrad_smallp_kernel<64, 64, 1>), dim3((unsigned)tiles), dim3(THREADS), lds, st, dy, x, dw, dbias,
                               (int)P, Co, Ci, dbias != nullptr, accumulate);
        else
        hipLaunchKernelGGL((linear_wgrad_smallp_kernel<64, 64, 0>), dim3((unsigned)tiles), dim3(THREADS), lds, st, dy, x, dw, dbias,
                           (int)P, Co, Ci, dbias != nullptr, accumulate);
        return phnet_launch_status();
    }
    if (wgrad3_applies(P, Hi, Wi, Ci, Co, R, S, stride, pad)) {
        long splits = wgrad3_splits(P, Co, Ci);
        const long row = (long)Co * NC + Co;
        while (splits > 1 && (!workspace || (uint64_t)(splits * row * sizeof(float)) > ws_bytes)) --splits;
        const bool pc = g_wgrad3s && !dbias && (long)Hi * Wi >= phnet_wgrad3s_kstep();      // producer / consumer kernel (an image holds >= one K step)
        const int bkw = pc ? phnet_wgrad3s_kstep() : g_wgrad3_bkw == 16 ? 16 : 32;
        const long psteps = ceil_div64(P, bkw);
        g.splits = (int)splits;
        g.pix_per_split = (int)(ceil_div64(psteps, splits) * bkw);
        float* out = splits > 1 ? (float*)workspace : dw;
        if (pc) {
            Wgrad3sShape s{N, Hi, Wi, Ci, Co, g.splits, g.pix_per_split};
            const int rc = phnet_wgrad3s_launch(dy, x, out, s, accumulate, st);
            if (rc != PHNET_OK) return rc;
            if (splits > 1) {
                const long nw = (long)Co * NC, nb = Co;
                hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div64((nw + nb) >> 2, 64)), dim3(256), 0, st,
                                   (const float*)workspace, dw, (float*)nullptr, nw, nb, (int)splits, accumulate);
            }
            return phnet_launch_status();
        }
        dim3 grid((unsigned)((Co / 64) * (Ci / 64) * 3), 1, (unsigned)splits);
        static bool attr = false;
        if (!attr) {
            if (hipFuncSetAttribute((const void*)conv_wgrad3x3_kernel<4, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, Wgrad3Lds<16>::BYTES) != hipSuccess ||
                hipFuncSetAttribute((const void*)conv_wgrad3x3_kernel<4, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, Wgrad3Lds<32>::BYTES) != hipSuccess)
                return PHNET_ERR_LAUNCH;
            attr = true;
        }
        if (bkw == 16)
            hipLaunchKernelGGL((conv_wgrad3x3_kernel<4, 16>), grid, dim3(W3_THREADS), Wgrad3Lds<16>::BYTES, st, dy, x, out, dbias, g, dbias != nullptr, accumulate);
        else
            hipLaunchKernelGGL((conv_wgrad3x3_kernel<4, 32>), grid, dim3(W3_THREADS), Wgrad3Lds<32>::BYTES, st, dy, x, out, dbias, g, dbias != nullptr, accumulate);
        if (splits > 1) {
            const long nw = (long)Co * NC, nb = Co;
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div64((nw + nb) >> 2, 64)), dim3(256), 0, st,
                               (const float*)workspace, dw, dbias, nw, nb, (int)splits, accumulate);
        }
        return phnet_launch_status();
    }
    // Linear / 1x1 layers over many rows with >= 64 tiles of 128 x 128: the producer / consumer kernel (csrc/wgrad1s.hip)
    if (g_wgrad1s && g_mma_mode == 3 && R == 1 && S == 1 && stride == 1 && pad == 0 && (Co & 127) == 0 && (Ci & 127) == 0 && P >= 256 &&
        (long)(Co / 128) * (Ci / 128) >= 64 && P * (long)max(Ci, Co) * 4 < 0x7fffffffL) {
        const long tiles1 = (long)(Co / 128) * (Ci / 128);
        long splits1 = max((long)1, min((long)(P / 128), (long)256 / tiles1));
        const long row = (long)Co * NC + Co;
        while (splits1 > 1 && (!workspace || (uint64_t)(splits1 * row * sizeof(float)) > ws_bytes)) --splits1;
        const int ks = phnet_wgrad1s_kstep();
        Wgrad1sShape s1{(int)P, Ci, Co, (int)splits1, (int)(ceil_div64(ceil_div64(P, ks), splits1) * ks)};
        const int rc = phnet_wgrad1s_launch(dy, x, splits1 > 1 ? (float*)workspace : dw, dbias, s1, accumulate, st);
        if (rc != PHNET_OK) return rc;
        if (splits1 > 1) {
            const long nw = (long)Co * NC, nb = Co;
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div64((nw + nb) >> 2, 64)), dim3(256), 0, st,
                               (const float*)workspace, dw, dbias, nw, nb, (int)splits1, accumulate);
        }
        return phnet_launch_status();
    }
    int bm = 64;
    const int bn = 64;
    long splits = wgrad_splits(P, Co, NC, &bm);
    const long tiles = ceil_div64(Co, bm) * ceil_div64(NC, bn);
    const long row = (long)Co * NC + Co;
    while (splits > 1 && (!workspace || (uint64_t)(splits * row * sizeof(float)) > ws_bytes)) --splits;
    const int bkw = (g_mma_mode == 3 && g_wgrad_bkw == 32) ? 32 : BK;      // pixels per K step of the kernel picked below
    const long psteps = ceil_div64(max(P, (long)1), bkw);
    g.splits = (int)splits;
    g.pix_per_split = (int)(ceil_div64(psteps, splits) * bkw);
    float* out = splits > 1 ? (float*)workspace : dw;
    dim3 grid((unsigned)tiles, 1, (unsigned)splits);
    const int want_bias = dbias != nullptr;
    if (g_mma_mode == 3 && bkw == 32 && bm == 128)
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 64, 3, 32, 2, true>), grid, dim3(THREADS), 0, st, dy, x, out, dbias, g, want_bias, accumulate);
    else if (g_mma_mode == 3 && bkw == 32)
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 3, 32, 2, true>), grid, dim3(THREADS), 0, st, dy, x, out, dbias, g, want_bias, accumulate);
    else if (g_mma_mode == 3 && bm == 128)
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 64, 3, 16, 4, true>), grid, dim3(THREADS), 0, st, dy, x, out, dbias, g, want_bias, accumulate);
    else if (g_mma_mode == 3)
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 3, 16, 4, true>), grid, dim3(THREADS), 0, st, dy, x, out, dbias, g, want_bias, accumulate);
    else if (bm == 128 && g_mma_mode == 2)
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 64, 2>), grid, dim3(THREADS), 0, st, dy, x, out, dbias, g, want_bias, accumulate);
    else if (g_mma_mode == 2)
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 2>), grid, dim3(THREADS), 0, st, dy, x, out, dbias, g, want_bias, accumulate);
    else if (bm == 128 && g_mma_mode == 1)
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 64, 1>), grid, dim3(THREADS), 0, st, dy, x, out, dbias, g, want_bias, accumulate);
    else if (g_mma_mode == 1)
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 1>), grid, dim3(THREADS), 0, st, dy, x, out, dbias, g, want_bias, accumulate);
    else if (bm == 128) hipLaunchKernelGGL((conv_wgrad_kernel<128, 64>), grid, dim3(THREADS), 0, st, dy, x, out, dbias, g, want_bias, accumulate);
    else hipLaunchKernelGGL((conv_wgrad_kernel<64, 64>), grid, dim3(THREADS), 0, st, dy, x, out, dbias, g, want_bias, accumulate);
    if (splits > 1) {
        const long nw = (long)Co * NC, nb = Co;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div64((nw + nb) >> 2, 64)), dim3(256), 0, st,
                           (const float*)workspace, dw, dbias, nw, nb, (int)splits, accumulate);
    }
    return phnet_launch_status();
}

// x NCHW [N][3][H][W] -> y NHWC [N][H][W][4] (4th channel zero): feeds the 7x7 stem.
PHNET_API int phnet_nchw3_to_nhwc4(const float* x, float* y, int32_t N, int32_t H, int32_t W, void* stream)
{
    if (N < 0 || H < 1 || W < 1) return PHNET_ERR_ARG;
    if (N == 0) return PHNET_OK;
    if (!x || !y) return PHNET_ERR_ARG;
    const long total = (long)N * H * W;
    hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       x, y, N, H * W);
    return phnet_launch_status();
}

// dst[rows][cd] <- src[rows][cs], zero-extending or truncating the innermost dimension.
PHNET_API int phnet_pad_channels(const float* src, float* dst, int64_t rows, int32_t cs, int32_t cd, void* stream)
{
    if (rows < 0 || cs < 1 || cd < 1) return PHNET_ERR_ARG;
    if (rows == 0) return PHNET_OK;
    if (!src || !dst) return PHNET_ERR_ARG;
    hipLaunchKernelGGL(pad_channels_kernel, dim3((unsigned)ceil_div64(rows * cd, 256)), dim3(256), 0, (hipStream_t)stream,
                       src, dst, (long)rows, cs, cd);
    return phnet_launch_status();
}

// ---- fused backward of a Linear layer over few rows ------------------------------------------------------------------
static bool linear_bwd_fusable(long M, long K, long N)
{
    if (!g_wgrad_smallp || M < 1 || M > SMALLP_MAX || (K & 3) || (N & 3) || K < 4 || N < 4) return false;
    const long wt = ceil_div64(N, 64) * ceil_div64(K, 64), dt = ceil_div64(M, 64) * ceil_div64(K, 64);
    return wt < g_smallp_max_tiles && N <= 640 && dt <= 256;          // dgrad stays unsplit: reduction length N <= 640 (gate MLP: 576)
}

// 1 when phnet_linear_bwd runs (M, K, N) as ONE launch; 0: use phnet_conv2d_dgrad + phnet_conv2d_wgrad instead.
PHNET_API int phnet_linear_bwd_fusable(int64_t M, int64_t K, int64_t N) { return linear_bwd_fusable(M, K, N) ? 1 : 0; }

// Backward of y = x w^T (+ b) for few rows, one launch: dx [M][K] = dy w;  dw [N][K] and dbias [N] (optional) = dy^T x,
// column sums of dy - overwritten or, with accumulate = 1, added to.  dy [M][N], x [M][K], w [N][K], all row-major.
// relu_y (optional, [M][N]): the layer's saved output when it ended in a ReLU - dy is then masked by relu_y > 0 on the fly.
// Only for shapes with phnet_linear_bwd_fusable(M, K, N) == 1 (PHNET_ERR_ARG otherwise).
PHNET_API int phnet_linear_bwd(const float* dy, const float* x, const float* w, const float* relu_y, float* dx, float* dw,
                               float* dbias, int32_t M, int32_t K, int32_t N, int32_t accumulate, void* stream)
{
    if (!linear_bwd_fusable(M, K, N) || !dy || !x || !w || !dx || !dw) return PHNET_ERR_ARG;
    ConvShape g{};
    g.N = M; g.Hi = 1; g.Wi = 1; g.Ci = N;              // A side = dY rows, N channels
    g.Ho = 1; g.Wo = 1; g.Co = K;                       // output = dX
    g.R = 1; g.S = 1; g.stride = 1; g.pad = 0; g.in_dil = 1;
    g.splits = 1;
    g.k_per_split = (int)(ceil_div64(N, 64) * 64);
    const unsigned dt = (unsigned)(ceil_div64(M, 64) * ceil_div64(K, 64)), wt = (unsigned)(ceil_div64(N, 64) * ceil_div64(K, 64));
    const int P16 = (M + 15) & ~15;
    const size_t lds_dgrad = 2 * (size_t)(KContigTile<64, 64>::FLOATS + KStridedTile<64, 64>::FLOATS) * sizeof(float);
    const size_t lds_wgrad = (size_t)P16 * 68 * 2 * sizeof(float);
    const size_t lds = lds_dgrad > lds_wgrad ? lds_dgrad : lds_wgrad;
    static bool attr = false;
    if (!attr) {
        const int cap = (int)((size_t)SMALLP_MAX * 68 * 2 * sizeof(float));
        const void* fns[8] = {(const void*)linear_bwd_fused_kernel<true, true, 0>, (const void*)linear_bwd_fused_kernel<true, false, 0>,
                              (const void*)linear_bwd_fused_kernel<false, true, 0>, (const void*)linear_bwd_fused_kernel<false, false, 0>,
                              (const void*)linear_bwd_fused_kernel<true, true, 1>, (const void*)linear_bwd_fused_kernel<true, false, 1>,
                              (const void*)linear_bwd_fused_kernel<false, true, 1>, (const void*)linear_bwd_fused_kernel<false, false, 1>};
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess) return PHNET_ERR_LAUNCH;
        attr = true;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool uni = N % 64 == 0 && g_uniform_tap;
#define PHNET_LAUNCH_BWD_(U_, A_, MMA_)                                                                                 \
    hipLaunchKernelGGL((linear_bwd_fused_kernel<U_, A_, MMA_>), dim3(dt + wt), dim3(THREADS), lds, st, dy, w, x, relu_y, dx, \
                       dw, dbias, g, dt, (int)M, (int)N, (int)K, dbias != nullptr, accumulate)
#define PHNET_LAUNCH_BWD(U_, A_)                                                                                        \
    do {                                                                                                                \
        if (g_mma_mode == 1) PHNET_LAUNCH_BWD_(U_, A_, 1);                                                              \
        else PHNET_LAUNCH_BWD_(U_, A_, 0);                                                                              \
    } while (0)
    if (uni && relu_y) PHNET_LAUNCH_BWD(true, true);
    else if (uni) PHNET_LAUNCH_BWD(true, false);
    else if (relu_y) PHNET_LAUNCH_BWD(false, true);
    else PHNET_LAUNCH_BWD(false, false);
#undef PHNET_LAUNCH_BWD
#undef PHNET_LAUNCH_BWD_
    return phnet_launch_status();
}
