// fp32 MFMA implicit-GEMM building blocks for gfx950 (CDNA4), shared by conv.hip and gemm.hip.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-for-bit an fmaf chain; dense peak
// 157.3 TF/s = 64 FLOP/clk/SIMD).  The reference runs this path in fp32 end to end
// (trainOL.py:21-22,225: GradScaler without autocast), so f32-input MFMA keeps its numerics.
//
// Tiling: 256 threads = 4 wavefronts in a 2x2 arrangement; block tile BM x BN (64 or 128 each), wave
// tile (BM/2) x (BN/2) made of 32x32 MFMA fragments; K tile 16 or 64, double-buffered in LDS with register
// prefetch of the next step (one barrier per step).
//
// LDS images (conflict-free by construction, see MI355X_MICROARCH.md "LDS"):
//   K-contiguous operand ([row][k]: im2col rows, [N][K] weights): [rows][BKT+4] floats; the fill is one
//     ds_write_b128 per 4 k's, the fragment read is 2 x ds_read_b128 per 32-row fragment and 16-deep sub-step
//     (80- or 272-byte row pitch -> the 16 lanes of a b128 lane group hit 16 distinct 4-bank slots).
//   K-strided operand ([k][col]: dY for wgrad, [K][N] right-hand sides): [BKT][cols+4] floats; fill is
//     ds_write_b128 along the columns, fragment read is ds_read_b32 (32 consecutive floats per half wave).
// The MFMA k index of (lane half h, step s) is 8h+s for BOTH operands, which is what lets the
// K-contiguous image be read with 128-bit loads.
#pragma once
#include "common.h"

namespace igemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 16;           // MFMA sub-step depth: 8 x v_mfma_f32_32x32x2 (k = 8h+s per lane half h)
constexpr int THREADS = 256;

// A K tile is BKT = 16 or 64 deep.  BKT = 64 is for the latency-bound skinny problems of the lane head (M = 240):
// four times fewer load -> barrier -> MFMA round trips, four times more bytes in flight per round.
template <int ROWS, int BKT> struct KContigTile { static constexpr int PITCH = BKT + 4; static constexpr int FLOATS = ROWS * PITCH; };
template <int COLS, int BKT> struct KStridedTile { static constexpr int PITCH = COLS + 4; static constexpr int FLOATS = BKT * PITCH; };

// ---- fragment reads (sub-step ks of the tile covers k in [16 ks, 16 ks + 16)) ---------------------------------
// K-contiguous image [rows][PITCH]: fragment f covers rows [32f, 32f+32) of the wave's sub-tile.
template <int F, int PITCH>
__device__ __forceinline__ void read_kcontig(const float* tile, int lane, int ks, float (&frag)[F][8]) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const f32x4* p = reinterpret_cast<const f32x4*>(tile + (f * 32 + r) * PITCH + 16 * ks + 8 * h);
        const f32x4 lo = p[0], hi = p[1];
        frag[f][0] = lo.x; frag[f][1] = lo.y; frag[f][2] = lo.z; frag[f][3] = lo.w;
        frag[f][4] = hi.x; frag[f][5] = hi.y; frag[f][6] = hi.z; frag[f][7] = hi.w;
    }
}

// K-strided image [BKT][PITCH]: column c of fragment f, k = 16 ks + 8h + s.
template <int F, int PITCH>
__device__ __forceinline__ void read_kstrided(const float* tile, int lane, int ks, float (&frag)[F][8]) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int s = 0; s < 8; ++s) frag[f][s] = tile[(16 * ks + 8 * h + s) * PITCH + f * 32 + c];
}

template <int FM, int FN>
__device__ __forceinline__ void mma_step(const float (&a)[FM][8], const float (&b)[FN][8], f32x16 (&acc)[FM][FN]) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
}

// ---- split-bf16 arithmetic (MMA = 1, opt-in through phnet_tune_mma; the product default stays f32-input MFMA) --------
// Every f32 operand is split in registers into two bf16 terms x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 mantissa
// bits) and a product becomes hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with f32 accumulation (the dropped
// lo*lo term and the split residual are ~2^-16 relative per product).  The fragment registers are the SAME as for the
// f32 MFMA - lane (r, h) holds k = 8h .. 8h+7 of its row / column for both instructions - so the LDS images, the staging
// and the epilogues do not change: 3 MFMAs of 32 cycles replace 8 of 64 per 16-deep sub-step.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split_bf16(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const __bf16 h = (__bf16)x[s];
        hi[s] = h;
        lo[s] = (__bf16)(x[s] - (float)h);
    }
}

template <int FM, int FN>
__device__ __forceinline__ void mma_step_split(const float (&a)[FM][8], const float (&b)[FN][8], f32x16 (&acc)[FM][FN]) {
    bf16x8 ah[FM], al[FM], bh[FN], bl[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) split_bf16(a[i], ah[i], al[i]);
#pragma unroll
    for (int j = 0; j < FN; ++j) split_bf16(b[j], bh[j], bl[j]);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
}

// MMA = 2: three bf16 terms per operand (x = hi + mid + lo EXACTLY: 3 x 8 mantissa bits, every difference is exact in f32)
// and the six products down to 2^-16 (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi); what is dropped (mid*lo, lo*mid, lo*lo)
// is <= 2^-24 relative per product - the rounding level of an f32 fma chain.  6 MFMAs of 32 cycles per 16-deep sub-step.
__device__ __forceinline__ void split3_bf16(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const __bf16 h = (__bf16)x[s];
        const float r1 = x[s] - (float)h;
        const __bf16 m = (__bf16)r1;
        hi[s] = h;
        mid[s] = m;
        lo[s] = (__bf16)(r1 - (float)m);
    }
}

template <int FM, int FN>
__device__ __forceinline__ void mma_step_split3(const float (&a)[FM][8], const float (&b)[FN][8], f32x16 (&acc)[FM][FN]) {
    bf16x8 ah[FM], am[FM], al[FM], bh[FN], bm[FN], bl[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) split3_bf16(a[i], ah[i], am[i], al[i]);
#pragma unroll
    for (int j = 0; j < FN; ++j) split3_bf16(b[j], bh[j], bm[j], bl[j]);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
}

// ---- MMA = 3: the exact three-term split done ONCE, while a tile is STAGED ------------------------------------------------
// The LDS images hold bf16: three planes (hi, mid, lo) per operand tile, written by the staging code from the f32 values it
// loaded (11 VALU instructions per PAIR of elements, paid once per element instead of once per fragment read and wave);
// the fragment reads are then plain 16-byte bf16x8 reads - one ds_read_b128 per plane for a K-contiguous image, two
// ds_read_b64_tr_b16 (hardware transpose, cdna_hip_programming.md T10) per plane for a K-strided one - and a 16-deep
// sub-step is 6 x v_mfma_f32_32x32x16_bf16 = 192 MFMA cycles against 512 for 8 x v_mfma_f32_32x32x2_f32, at the accuracy
// of the f32 instruction (what is dropped is <= 2^-24 per product, see MMA = 2).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// K-contiguous bf16 plane [rows][BKT] with a 16-byte pad per row: the 16 lanes of a ds_read_b128 lane group sit on 16
// distinct 16-byte slots of the 256-byte bank row for BKT = 16, 32 and 64 (row pitches 48, 80, 144 bytes: odd multiples of 16)
template <int ROWS, int BKT> struct KContigPlanes {
    static constexpr int PITCH = BKT * 2 + 16;                 // bytes
    static constexpr int PLANE = ROWS * PITCH;                 // bytes per plane
    static constexpr int BYTES = 3 * PLANE;
};
// K-strided bf16 plane [BKT][cols] with 64 bytes of pad per row: the 4 k-rows x 64 bytes that one half wave fetches with
// ds_read_b64_tr_b16 then fall on four distinct quarters of the bank row (pitch = 64 mod 256 for 128 columns, 192 for 64)
template <int COLS, int BKT> struct KStridedPlanes {
    static constexpr int PITCH = COLS * 2 + 64;
    static constexpr int PLANE = BKT * PITCH;
    static constexpr int BYTES = 3 * PLANE;
};

// two f32 -> three packed bf16 pairs (hi, mid, lo) with x = hi + mid + lo exactly
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x0, x1}, bf16x2));
    const float r0 = x0 - __builtin_bit_cast(float, hb << 16), r1 = x1 - __builtin_bit_cast(float, hb & 0xffff0000u);
    const unsigned mb = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){r0, r1}, bf16x2));
    const float q0 = r0 - __builtin_bit_cast(float, mb << 16), q1 = r1 - __builtin_bit_cast(float, mb & 0xffff0000u);
    hi = hb; mid = mb;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){q0, q1}, bf16x2));
}

// four consecutive elements (one staged float4) -> 8 bytes into each of the three planes at byte offset `off`
template <int PLANE>
__device__ __forceinline__ void store_split3(unsigned char* planes, int off, const f32x4& v) {
    unsigned h0, m0, l0, h1, m1, l1;
    split3_pair(v.x, v.y, h0, m0, l0);
    split3_pair(v.z, v.w, h1, m1, l1);
    *reinterpret_cast<u32x2*>(planes + off) = (u32x2){h0, h1};
    *reinterpret_cast<u32x2*>(planes + PLANE + off) = (u32x2){m0, m1};
    *reinterpret_cast<u32x2*>(planes + 2 * PLANE + off) = (u32x2){l0, l1};
}

struct Frag3 { bf16x8 hi, mid, lo; };

// fragment f = rows [32f, 32f+32) of the wave's sub-tile, sub-step ks = k in [16 ks, 16 ks + 16): lane (r, h) takes k = 8h..8h+7
template <int F, int PITCH, int PLANE>
__device__ __forceinline__ void read_kcontig3(const unsigned char* planes, int lane, int ks, Frag3 (&frag)[F]) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const unsigned char* p = planes + (f * 32 + r) * PITCH + 32 * ks + 16 * h;
        frag[f].hi = *reinterpret_cast<const bf16x8*>(p);
        frag[f].mid = *reinterpret_cast<const bf16x8*>(p + PLANE);
        frag[f].lo = *reinterpret_cast<const bf16x8*>(p + 2 * PLANE);
    }
}

__device__ __forceinline__ bf16x8 tr_read8(const unsigned char* p0, const unsigned char* p1) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return __builtin_bit_cast(bf16x8, v);
}

// K-strided planes [k][col]: the 16-lane group g = lane >> 4 fetches the 4 x 16 block of k rows 8(g>>1) + 4s .. +3 (s = 0, 1:
// two reads) and columns 32f + 16(g&1) .. +15; lane 4q+p of the group supplies the address of row q, columns 4p..4p+3 and
// receives column (lane & 15) of the four rows - i.e. lane (r = lane & 31, h = lane >> 5) ends up with k = 8h..8h+7 of column
// r, the operand layout of the MFMA.  EXEC must be all ones (no divergence around this call).
template <int F, int PITCH, int PLANE>
__device__ __forceinline__ void read_kstrided3(const unsigned char* planes, int lane, int ks, Frag3 (&frag)[F]) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const int row = 16 * ks + 8 * (g >> 1) + q;
    const int colb = (16 * (g & 1) + 4 * pp) * 2;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const unsigned char* p = planes + row * PITCH + f * 64 + colb;
        frag[f].hi = tr_read8(p, p + 4 * PITCH);
        frag[f].mid = tr_read8(p + PLANE, p + PLANE + 4 * PITCH);
        frag[f].lo = tr_read8(p + 2 * PLANE, p + 2 * PLANE + 4 * PITCH);
    }
}

template <int FM, int FN>
__device__ __forceinline__ void mma3_step(const Frag3 (&a)[FM], const Frag3 (&b)[FN], f32x16 (&acc)[FM][FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].lo, b[j].hi, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].hi, b[j].lo, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].mid, b[j].mid, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].mid, b[j].hi, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].hi, b[j].mid, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].hi, b[j].hi, acc[i][j], 0, 0, 0);
        }
}

template <int MMA, int FM, int FN>
__device__ __forceinline__ void mma_any(const float (&a)[FM][8], const float (&b)[FN][8], f32x16 (&acc)[FM][FN]) {
    if (MMA == 2) mma_step_split3<FM, FN>(a, b, acc);
    else if (MMA == 1) mma_step_split<FM, FN>(a, b, acc);
    else mma_step<FM, FN>(a, b, acc);
}

// C/D fragment element `reg` of lane `lane` sits at (row, col) of the 32x32 tile:
__device__ __forceinline__ int frag_row(int lane, int reg) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ int frag_col(int lane) { return lane & 31; }

// XCD-aware bijective remap (cdna_hip_programming.md 5, "XCD swizzle must be bijective"): blocks that
// are neighbours in the logical tile order land on the same XCD (= same L2).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
    const unsigned xcd = bid & 7u, q = nblocks >> 3, r = nblocks & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

}  // namespace igemm
