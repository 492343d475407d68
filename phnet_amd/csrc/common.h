// Shared helpers for the gfx950 kernels behind the C-ABI in include/phnet_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PHNET_OK 0
#define PHNET_ERR_ARG (-1)        // bad shape / null pointer / unsupported size
#define PHNET_ERR_WORKSPACE (-2)  // caller-provided workspace too small
#define PHNET_ERR_LAUNCH (-3)     // hipGetLastError() after the launch

#define PHNET_API extern "C" __attribute__((visibility("default")))

static inline int phnet_launch_status() {
    return hipGetLastError() == hipSuccess ? PHNET_OK : PHNET_ERR_LAUNCH;
}

// Wave-wide reductions on the DPP data path (quad swaps, row rotations, row broadcasts - the rocPRIM sequence): six
// VALU instructions with a DPP modifier and one v_readlane instead of six ds_bpermute round trips through the LDS pipeline
// (~100 cycles each, and the LayerNorm-heavy kernels chain dozens of them).  All 64 lanes must be active; the total is
// returned to every lane.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, false));
}

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<0xb1>(v);        // quad_perm [1,0,3,2]
    v += dpp_move<0x4e>(v);        // quad_perm [2,3,0,1]
    v += dpp_move<0x124>(v);       // row_ror 4
    v += dpp_move<0x128>(v);       // row_ror 8
    v += dpp_move<0x142>(v);       // row_bcast 15
    v += dpp_move<0x143>(v);       // row_bcast 31: lane 63 now holds the sum of all 64 lanes
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_move<0xb1>(v));
    v = fmaxf(v, dpp_move<0x4e>(v));
    v = fmaxf(v, dpp_move<0x124>(v));
    v = fmaxf(v, dpp_move<0x128>(v));
    v = fmaxf(v, dpp_move<0x142>(v));
    v = fmaxf(v, dpp_move<0x143>(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---- counter-based dropout masks --------------------------------------------------------------------------------
// A training step owns one 64-bit device counter (bumped once per step, so a replayed hipGraph draws fresh masks); each
// dropout site of the step has a host-side call id.  Element idx of site `call` is kept iff the splitmix64 finaliser of
// (state, call, idx) clears the threshold p * 2^32 - the backward kernels recompute the same bit instead of reading a
// stored mask.
// Items: a launch may cover several independent ITEMS (the (frame, stage) passes of branch B whose backward runs as one batch,
// or the clips of a batched step); element `local` of item `item` draws bit (item << 32 | local), so a launch over ONE item
// (item0 = its number, item_elems = 0) and a launch over a batch of items (item0 = 0, item_elems = elements per item) see the
// same masks.  Both numbers travel inside the 64-bit `rng_call` argument of the C-ABI: bits 0-19 site id, bits 20-31 item0,
// bits 32-63 item_elems (0 = the whole launch is one item).
struct DropRng { const uint64_t* state; uint64_t call; uint32_t thresh; uint32_t item0; uint32_t item_elems; };

__device__ __forceinline__ uint64_t phnet_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t phnet_rng_seed(const DropRng& r) {
    return r.thresh ? phnet_mix64(r.state[0] * 0xD1342543DE82EF95ull + r.call) : 0ull;       // one stream per (step, site)
}
__device__ __forceinline__ bool phnet_rng_keep(uint64_t seed, uint64_t idx, uint32_t thresh) {
    return (uint32_t)(phnet_mix64(seed + (idx + 1) * 0x9E3779B97F4A7C15ull) >> 32) >= thresh;
}
// generator index of flat element idx of an elementwise launch
__device__ __forceinline__ uint64_t phnet_rng_index(const DropRng& r, uint64_t idx) {
    if (r.item_elems) return ((uint64_t)(r.item0 + (uint32_t)(idx / r.item_elems)) << 32) | (idx % r.item_elems);
    return ((uint64_t)r.item0 << 32) | idx;
}
// generator index of element `local` of batch entry b (kernels whose grid carries the batch index)
__device__ __forceinline__ uint64_t phnet_rng_index_b(const DropRng& r, uint32_t b, uint64_t local) {
    return ((uint64_t)(r.item0 + b) << 32) | local;
}
static inline DropRng phnet_make_rng(const uint64_t* state, uint64_t call, float p) {
    DropRng r{state, call & 0xFFFFFull, 0u, (uint32_t)((call >> 20) & 0xFFFu), (uint32_t)(call >> 32)};
    if (state && p > 0.f) r.thresh = (uint32_t)fmin(4294967295.0, (double)p * 4294967296.0);
    return r;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
