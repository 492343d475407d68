// Device-side body of the label assignment (see assign.hip): one 256-thread workgroup, callable from other kernels.
#pragma once
#include "common.h"

namespace phassign {

constexpr int NT = 256;
constexpr int MAXL = 4;

template <int WAVES = 4>
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = red[0];
#pragma unroll
    for (int i = 1; i < WAVES; ++i) m = fmaxf(m, red[i]);
    return m;
}

// QUAD = 4: the workgroup has 4 x NT threads and the cost phase gives every (anchor, label) pair its own thread (with one thread
// per anchor the 4 waves of the single workgroup sit alone on their SIMDs and the 4 S-long accumulation chains of a thread run at
// the latency of dependent VALU issue: 14 of the kernel's 23 us); the matching phases use the first NT threads as before.
// MANY = false: the one-to-one assignment of `assign` (dynamic_assign.py:128-190; focal_alpha 0.25).
// MANY = true: `assignOne2Many` (dynamic_assign.py:292-357; focal_alpha 0.5): label j wants k_j = max(1, int(sum of its 4 largest
// line IoUs)) anchors; rounds of the exact matching over ALL valid labels, each round keeps the pairs whose POSITION p in the
// row-sorted pair list has k_p > 0 (as shipped: the per-label mask indexes the pair list, :351), retires their rows, decrements
// the positive k's.  many_rows / many_cols [MAXL*MAXL] (-1 padded; cols = original label rows), many_n = number of pairs.
template <bool MANY = false, int QUAD = 1>
__device__ __forceinline__ void lane_assign_block(
    const float* __restrict__ pred, const float* __restrict__ tgt, int N, int L, int S, float img_w, float img_h,
    int64_t* __restrict__ rows_by_col, int64_t* __restrict__ rows_sorted, int32_t* __restrict__ n_valid_out,
    float* __restrict__ cost_out, float* cost /* LDS [N][MAXL] (MANY: [2][N][MAXL], the second half holds the IoUs) */,
    float focal_alpha = 0.25f, int64_t* __restrict__ many_rows = nullptr, int64_t* __restrict__ many_cols = nullptr,
    int32_t* __restrict__ many_n = nullptr)
{
    __shared__ float t_x[MAXL][256];                 // target xs (S <= 250)
    __shared__ float t_len[MAXL], red[4 * QUAD];
    __shared__ int t_valid[MAXL], top_rows[MAXL][MAXL], best_combo;
    __shared__ float combo_cost[NT];
    __shared__ unsigned char gone[NT];               // MANY: rows retired by earlier rounds
    __shared__ int ks[MAXL], comp[MAXL], many_count, more;
    const int tid = threadIdx.x;
    const int W = 6 + S;
    if (tid < NT) gone[tid] = 0;

    if (tid < MAXL) {
        t_valid[tid] = (tid < L) && (tgt[tid * W + 1] == 1.0f);
        t_len[tid] = 0.f;
    }
    for (int i = tid; i < L * S; i += NT * QUAD) t_x[i / S][i % S] = tgt[(i / S) * W + 6 + (i % S)];
    __syncthreads();
    if (tid < L) {
        int n = 0;
        for (int k = 0; k < S; ++k) { const float t = t_x[tid][k]; n += !((t < 0.f) || (t >= img_w)); }
        t_len[tid] = (float)n;
    }
    __syncthreads();

    if constexpr (QUAD == 4) {
        // ---- one thread per (anchor, label): the label index is wave-uniform ------------------------------------------
        const int a = tid & (NT - 1), jq = tid / NT;
        const bool arow = a < N, on = arow && jq < L && t_valid[jq];
        const float* p = pred + (size_t)(arow ? a : 0) * W;
        float cls = 0.f, dist = 0.f, start = 0.f, theta = 0.f, iou = 0.f;
        float mx_d = -INFINITY, mx_s = -INFINITY, mx_t = -INFINITY;
        if (arow) {
            const float pr = 1.0f / (1.0f + expf(-p[1]));
            const float negc = -logf(1.0f - pr + 1e-12f) * (1.0f - focal_alpha) * (pr * pr);
            const float posc = -logf(pr + 1e-12f) * focal_alpha * ((1.0f - pr) * (1.0f - pr));
            cls = posc - negc;
        }
        if (on) {
            float d = 0.f, ovr = 0.f, uni = 0.f;
            const float* tx = t_x[jq];
            auto term = [&](int k, float xr) {
                const float x = xr * (img_w - 1.0f), t = tx[k];
                const bool ok = !(t < 0.f) && !(t >= img_w);
                d += ok ? fabsf(t - x) : 0.f;
                ovr += ok ? fminf(x + 15.f, t + 15.f) - fmaxf(x - 15.f, t - 15.f) : 0.f;
                uni += ok ? fmaxf(x + 15.f, t + 15.f) - fminf(x - 15.f, t - 15.f) : 0.f;
            };
            int k = 0;
            if ((W & 1) == 0) {
                for (; k + 1 < S; k += 16) {
                    float2 buf[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int kk = k + 2 * u;
                        buf[u] = *reinterpret_cast<const float2*>(p + 6 + (kk + 1 < S ? kk : 0));
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int kk = k + 2 * u;
                        if (kk + 1 < S) { term(kk, buf[u].x); term(kk + 1, buf[u].y); }
                    }
                }
                k = S & ~1;
            }
            for (; k < S; ++k) term(k, p[6 + k]);
            dist = d / (t_len[jq] + 1e-9f);
            iou = ovr / (uni + 1e-9f);
            const float* tr = tgt + (size_t)jq * W;
            const float dy = p[2] * (img_h - 1.0f) - tr[2] * (img_h - 1.0f), dx = p[3] * (img_w - 1.0f) - tr[3] * (img_w - 1.0f);
            start = sqrtf(dy * dy + dx * dx);
            theta = fabsf(p[4] - tr[4]) * 180.f;
            mx_d = dist; mx_s = start; mx_t = theta;
        }
        mx_d = block_max<16>(mx_d, red);
        mx_s = block_max<16>(mx_s, red);
        mx_t = block_max<16>(mx_t, red);
        if (arow) {
            float c = INFINITY;
            if (on) {
                const float a_ = 1.0f - dist / (mx_d + 1e-4f);
                const float b_ = 1.0f - start / (mx_s + 1e-4f);
                const float t_ = 1.0f - theta / (mx_t + 1e-4f);
                const float prod = a_ * b_ * t_;
                c = -(prod * prod) * 3.0f + cls - iou;
            }
            cost[a * MAXL + jq] = c;
            if (MANY) cost[(N + a) * MAXL + jq] = on ? fmaxf(iou, 0.f) : -1.f;
            if (cost_out && jq < L) cost_out[(size_t)a * L + jq] = c;
        }
    } else {
    // ---- per-anchor raw terms ------------------------------------------------------------------------------
    float dist[MAXL], start[MAXL], theta[MAXL], iou[MAXL], cls = 0.f;
    float mx_d = -INFINITY, mx_s = -INFINITY, mx_t = -INFINITY;
    const bool arow = tid < N;
    if (arow) {
        const float* p = pred + (size_t)tid * W;
        // focal cost of the positive class (focal_cost: alpha .25, gamma 2, eps 1e-12), label column 1
        const float pr = 1.0f / (1.0f + expf(-p[1]));
        const float negc = -logf(1.0f - pr + 1e-12f) * (1.0f - focal_alpha) * (pr * pr);
        const float posc = -logf(pr + 1e-12f) * focal_alpha * ((1.0f - pr) * (1.0f - pr));
        cls = posc - negc;
        const float psy = p[2] * (img_h - 1.0f), psx = p[3] * (img_w - 1.0f), pth = p[4];
        // one pass over the anchor's S x-columns for all label columns at once (each x is read once, as a float2: the row
        // starts at an even element) - the four per-column sums keep their ascending-k order
        float d[MAXL], ovr[MAXL], uni[MAXL];
#pragma unroll
        for (int j = 0; j < MAXL; ++j) d[j] = ovr[j] = uni[j] = 0.f;
        // branch-free: a branch on the freshly read label value (or on the LDS validity flag) makes every one of the 4 S trips wait
        // for its LDS read - the loop then is a chain of ~150 LDS latencies, most of this kernel's time; adding 0 leaves a sum as it is
        bool col_on[MAXL];
#pragma unroll
        for (int j = 0; j < MAXL; ++j) col_on[j] = j < L && t_valid[j];
        auto term = [&](int k, float xr) {
            const float x = xr * (img_w - 1.0f);
#pragma unroll
            for (int j = 0; j < MAXL; ++j) {
                const float t = t_x[j][k];
                const bool ok = col_on[j] && !(t < 0.f) && !(t >= img_w);
                d[j] += ok ? fabsf(t - x) : 0.f;
                ovr[j] += ok ? fminf(x + 15.f, t + 15.f) - fmaxf(x - 15.f, t - 15.f) : 0.f;
                uni[j] += ok ? fmaxf(x + 15.f, t + 15.f) - fminf(x - 15.f, t - 15.f) : 0.f;
            }
        };
        int k = 0;
        if ((W & 1) == 0)
            // 16 columns per trip, their 8 loads issued together (clamped addresses, branch-free): with one load per trip the loop
            // was a chain of S/2 memory latencies (18 x ~0.5 us at S = 36 - most of this kernel's 26 us)
            for (; k + 1 < S; k += 16) {
                float2 buf[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int kk = k + 2 * u;
                    buf[u] = *reinterpret_cast<const float2*>(p + 6 + (kk + 1 < S ? kk : 0));
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int kk = k + 2 * u;
                    if (kk + 1 < S) { term(kk, buf[u].x); term(kk + 1, buf[u].y); }
                }
            }
        k = ((W & 1) == 0) ? (S & ~1) : 0;
        for (; k < S; ++k) term(k, p[6 + k]);
#pragma unroll
        for (int j = 0; j < MAXL; ++j) {
            dist[j] = start[j] = theta[j] = iou[j] = 0.f;
            if (j >= L || !t_valid[j]) continue;
            dist[j] = d[j] / (t_len[j] + 1e-9f);
            iou[j] = ovr[j] / (uni[j] + 1e-9f);
            const float* tr = tgt + (size_t)j * W;
            const float dy = psy - tr[2] * (img_h - 1.0f), dx = psx - tr[3] * (img_w - 1.0f);
            start[j] = sqrtf(dy * dy + dx * dx);
            theta[j] = fabsf(pth - tr[4]) * 180.f;
            mx_d = fmaxf(mx_d, dist[j]); mx_s = fmaxf(mx_s, start[j]); mx_t = fmaxf(mx_t, theta[j]);
        }
    }
    mx_d = block_max(mx_d, red);
    mx_s = block_max(mx_s, red);
    mx_t = block_max(mx_t, red);
    if (arow) {
#pragma unroll
        for (int j = 0; j < MAXL; ++j) {
            float c = INFINITY;
            if (j < L && t_valid[j]) {
                const float a = 1.0f - dist[j] / (mx_d + 1e-4f);
                const float b = 1.0f - start[j] / (mx_s + 1e-4f);
                const float t = 1.0f - theta[j] / (mx_t + 1e-4f);
                const float prod = a * b * t;
                c = -(prod * prod) * 3.0f + cls - iou[j];
            }
            cost[tid * MAXL + j] = c;
            if (MANY) cost[(N + tid) * MAXL + j] = (j < L && t_valid[j]) ? fmaxf(iou[j], 0.f) : -1.f;
            if (cost_out && j < L) cost_out[(size_t)tid * L + j] = c;
        }
    }
    }
    __syncthreads();

    if (MANY) {
        // k_j per valid label (compact order = ascending j): the 4 largest clamped IoUs of the column, summed in descending order
        const int j = tid >> 6, lane = tid & 63;
        if (j < MAXL) {
            const bool valid = j < L && t_valid[j];
            float sum = 0.f;
            int taken[MAXL];
            for (int r = 0; r < MAXL; ++r) {
                float bv = -INFINITY; int bi = 0x7fffffff;
                if (valid)
                    for (int i = lane; i < N; i += 64) {
                        bool tk = false;
                        for (int q = 0; q < r; ++q) tk |= (taken[q] == i);
                        const float v = cost[(N + i) * MAXL + j];
                        if (!tk && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
                    }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float ov = __shfl_xor(bv, off, 64);
                    const int oi = __shfl_xor(bi, off, 64);
                    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                }
                taken[r] = bi;
                if (valid && bi != 0x7fffffff) sum += bv;
            }
            if (lane == 0) {
                const int k = (int)sum;
                ks[j] = valid ? (k < 1 ? 1 : k) : 0;
            }
        }
        if (tid == 0) many_count = 0;
        __syncthreads();
        if (tid == 0) {
            int c = 0;
            for (int j = 0; j < MAXL; ++j) { comp[j] = c; c += (j < L && t_valid[j]); }
            // ks in COMPACT order (the reference filters the valid labels first)
            int kc[MAXL] = {0, 0, 0, 0};
            for (int j = 0; j < MAXL; ++j) if (j < L && t_valid[j]) kc[comp[j]] = ks[j];
            for (int j = 0; j < MAXL; ++j) ks[j] = kc[j];
            more = (kc[0] + kc[1] + kc[2] + kc[3]) > 0;
        }
        __syncthreads();
    }
    for (int round = 0; round < (MANY ? MAXL + 1 : 1); ++round) {
    if (MANY && !more) break;
    // ---- the MAXL cheapest rows of every valid column (wave j <-> column j) -----------------------------------
    {
        const int j = tid >> 6, lane = tid & 63;
        if (j < MAXL) {
            const bool valid = j < L && t_valid[j];
            for (int r = 0; r < MAXL; ++r) {
                float bv = INFINITY; int bi = 0x7fffffff;
                if (valid)
                    for (int i = lane; i < N; i += 64) {
                        bool taken = gone[i] != 0;
                        for (int q = 0; q < r; ++q) taken |= (top_rows[j][q] == i);
                        const float c = cost[i * MAXL + j];
                        if (!taken && (c < bv || (c == bv && i < bi))) { bv = c; bi = i; }
                    }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float ov = __shfl_xor(bv, off, 64);
                    const int oi = __shfl_xor(bi, off, 64);
                    if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                }
                if (lane == 0) top_rows[j][r] = valid ? bi : -1;
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();

    // ---- enumerate candidate combinations: thread t <-> (t&3, (t>>2)&3, (t>>4)&3, t>>6) -----------------------
    {
        int rows[MAXL];
        float total = 0.f;
        bool ok = true;
#pragma unroll
        for (int j = 0; j < MAXL; ++j) {
            const int pick = (tid >> (2 * j)) & 3;
            const bool valid = j < L && t_valid[j];
            rows[j] = valid ? top_rows[j][pick] : -1;
            if (!valid) { ok = ok && pick == 0; continue; }          // one representative per invalid column
            if (rows[j] < 0 || rows[j] == 0x7fffffff) { ok = false; continue; }
            total += cost[rows[j] * MAXL + j];
#pragma unroll
            for (int q = 0; q < j; ++q) ok = ok && !(rows[q] >= 0 && rows[q] == rows[j]);
        }
        if (tid < NT) combo_cost[tid] = ok ? total : INFINITY;
    }
    __syncthreads();
    if (tid < 64) {
        float bv = INFINITY; int bi = 0x7fffffff;
        for (int c = tid; c < NT; c += 64) {
            const float v = combo_cost[c];
            if (v < bv || (v == bv && c < bi)) { bv = v; bi = c; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (tid == 0) best_combo = bi;
    }
    __syncthreads();
    if (tid == 0) {
        int rows[MAXL], nv = 0;
        for (int j = 0; j < MAXL; ++j) {
            const bool valid = j < L && t_valid[j];
            rows[j] = (valid && best_combo != 0x7fffffff) ? top_rows[j][(best_combo >> (2 * j)) & 3] : -1;
            if (j < L && rows_by_col) rows_by_col[j] = rows[j];
            nv += rows[j] >= 0;
        }
        // ascending valid rows first (prior-index order, as scipy returns them), then -1
        for (int a = 0; a < MAXL; ++a)
            for (int b = a + 1; b < MAXL; ++b) {
                const bool swap = (rows[b] >= 0) && (rows[a] < 0 || rows[b] < rows[a]);
                if (swap) { const int t = rows[a]; rows[a] = rows[b]; rows[b] = t; }
            }
        if (!MANY) {
            for (int j = 0; j < L; ++j) rows_sorted[j] = rows[j];
            if (n_valid_out) *n_valid_out = nv;
        } else {
            // pairs in row-sorted order: position p carries (row, label); keep where k_p > 0, retire the kept rows
            int lab[MAXL];
            for (int p_ = 0; p_ < MAXL; ++p_) {
                lab[p_] = -1;
                if (rows[p_] < 0) continue;
                for (int j = 0; j < MAXL; ++j)
                    if (j < L && t_valid[j] && best_combo != 0x7fffffff && top_rows[j][(best_combo >> (2 * j)) & 3] == rows[p_]) lab[p_] = j;
            }
            for (int p_ = 0; p_ < MAXL; ++p_) {
                if (rows[p_] < 0 || ks[p_] <= 0) continue;
                many_rows[many_count] = rows[p_];
                many_cols[many_count] = lab[p_];
                ++many_count;
                gone[rows[p_]] = 1;
            }
            int any = 0;
            for (int j = 0; j < MAXL; ++j) { if (ks[j] > 0) --ks[j]; any += ks[j]; }
            more = any > 0 && nv > 0;
        }
    }
    __syncthreads();
    }   // rounds
    if (MANY && tid == 0) {
        for (int i = many_count; i < MAXL * MAXL; ++i) { many_rows[i] = -1; many_cols[i] = -1; }
        if (many_n) *many_n = many_count;
    }
}


}  // namespace phassign
