// One-shot all-reduce for the SMALL messages of the data-parallel step (gfx950, xGMI): the SyncBatchNorm statistic exchanges
// of trainOL.py:141 (nn.SyncBatchNorm: 2C+1 doubles forward, 2C floats backward, C <= 512 -> at most 8 KB), 72 of them per
// step, each on the step's critical path.
//
// A stock RCCL all-reduce of 8 KB is a kernel launch + a ring / tree protocol over the process group's channels: tens of
// microseconds on 8 ranks, 72 times per step.  Here every rank owns an exchange buffer that all peers have mapped (hipIpc
// handles, exchanged once); ONE launch per rank writes the rank's payload straight into its slot of EVERY peer's buffer over
// xGMI (point-to-point links: all peers at once), then polls its own buffer until every peer's payload has arrived, and adds
// the contributions in RANK ORDER - every rank forms the same sum bit for bit, whatever the arrival order.
//
// Transport = 8-byte granules {32 data bits, 32-bit sequence tag} written by ONE 8-byte store each (the LL protocol of NCCL):
// a granule is either entirely old or entirely new, so no flag / fence ordering between payload and flag is needed.  A double
// travels as two granules.  Stores and polls are system-scope relaxed atomics (they bypass the caches); the buffers are
// allocated uncached / fine-grained.  Two slots, selected by the parity of the sequence number: a rank that has received
// sequence n+1 from every peer knows that every peer has finished reading sequence n, so slot (n+2) & 1 is free again.
// The sequence number lives in device memory and is advanced by the kernel itself (one workgroup per call): a hipGraph replay of
// the step keeps counting.
// Every spin is bounded: a peer that never arrives raises `*err` and the wave leaves (no hang, the result is then invalid).
#include "common.h"

namespace {

constexpr int IPC_NT = 1024;                                       // ONE workgroup per call: the messages are <= 1025 elements
constexpr unsigned long long IPC_MAX_POLLS = 1ull << 21;          // x ~1 us per poll: seconds, then give up

// buf layout (per rank): [2 slots][world][cap] granules of 8 bytes; ctrl: {uint32 seq, uint32 err}
template <typename T>
__global__ __launch_bounds__(IPC_NT) void oneshot_allreduce_kernel(T* __restrict__ data, int count, unsigned long long* const* __restrict__ peers,
                                                                   int rank, int world, int cap, unsigned* __restrict__ ctrl)
{
    constexpr int G = sizeof(T) / 4;                               // granules per element
    const unsigned seq = ctrl[0];
    __syncthreads();                                               // everybody holds the sequence number before thread 0 advances it
    const size_t slot = (size_t)(seq & 1u) * world * cap;
    bool timed_out = false;
    for (int k = threadIdx.x; k < count; k += IPC_NT) {
        unsigned w[G];
        __builtin_memcpy(w, &data[k], sizeof(T));
        // 1. my payload into my row of every peer's buffer (my own included: one code path)
        for (int p = 0; p < world; ++p) {
            unsigned long long* dst = peers[p] + slot + (size_t)rank * cap + (size_t)k * G;
#pragma unroll
            for (int g = 0; g < G; ++g)
                __hip_atomic_store(dst + g, ((unsigned long long)seq << 32) | w[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    for (int k = threadIdx.x; k < count; k += IPC_NT) {
        // 2. every peer's payload out of MY buffer, summed in rank order
        T sum = (T)0;
        const unsigned long long* mine = peers[rank] + slot + (size_t)k * G;
        for (int p = 0; p < world; ++p) {
            unsigned r[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                unsigned long long v = 0;
                unsigned long long polls = 0;
                for (;;) {
                    v = __hip_atomic_load(mine + (size_t)p * cap + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if ((unsigned)(v >> 32) == seq) break;
                    if (timed_out || ++polls > IPC_MAX_POLLS) { timed_out = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                r[g] = (unsigned)v;
            }
            T x;
            __builtin_memcpy(&x, r, sizeof(T));
            sum += x;
        }
        data[k] = sum;
    }
    if (timed_out) __hip_atomic_store(&ctrl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) ctrl[0] = seq + 1u == 0u ? 1u : seq + 1u;      // 0 = "never written": skipped
}

}  // namespace

// bytes of one rank's exchange buffer for payloads of up to max_bytes per message
PHNET_API uint64_t phnet_ipc_buffer_bytes(int32_t world, uint64_t max_bytes)
{
    if (world < 1 || max_bytes < 4) return 0;
    return (uint64_t)2 * world * (max_bytes / 4) * 8;
}

// uncached device memory that peers may map (falls back to fine-grained, then to plain device memory); zero-filled
PHNET_API int phnet_ipc_alloc(uint64_t bytes, void** ptr)
{
    if (!ptr || bytes == 0) return PHNET_ERR_ARG;
    *ptr = nullptr;
    if (hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        if (hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            if (hipMalloc(ptr, bytes) != hipSuccess) { (void)hipGetLastError(); return PHNET_ERR_WORKSPACE; }
        }
    }
    if (hipMemset(*ptr, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return PHNET_ERR_LAUNCH;
    return PHNET_OK;
}

PHNET_API int phnet_ipc_free(void* ptr) { return (ptr && hipFree(ptr) == hipSuccess) ? PHNET_OK : PHNET_ERR_ARG; }

// handle64: 64 bytes (hipIpcMemHandle_t) the owner hands to its peers through any host channel
PHNET_API int phnet_ipc_get_handle(void* ptr, void* handle64)
{
    if (!ptr || !handle64) return PHNET_ERR_ARG;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    return hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, ptr) == hipSuccess ? PHNET_OK : PHNET_ERR_LAUNCH;
}

PHNET_API int phnet_ipc_open_handle(const void* handle64, void** ptr)
{
    if (!handle64 || !ptr) return PHNET_ERR_ARG;
    hipIpcMemHandle_t h;
    __builtin_memcpy(&h, handle64, sizeof(h));
    return hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess ? PHNET_OK : PHNET_ERR_LAUNCH;
}

PHNET_API int phnet_ipc_close_handle(void* ptr) { return (ptr && hipIpcCloseMemHandle(ptr) == hipSuccess) ? PHNET_OK : PHNET_ERR_ARG; }

// In-place SUM over the ranks of data[count] (dtype 0 = float32, 1 = float64).  peers: DEVICE array of `world` pointers to the
// ranks' exchange buffers as mapped in THIS process (entry `rank` = the local buffer); cap = granules per rank row =
// max_bytes / 4 of phnet_ipc_buffer_bytes; ctrl: DEVICE {uint32 sequence (start at 1), uint32 error}.  One launch on `stream`.
PHNET_API int phnet_oneshot_allreduce(void* data, int32_t count, int32_t dtype, const void* peers, int32_t rank, int32_t world,
                                      int32_t cap, void* ctrl, void* stream)
{
    if (!data || !peers || !ctrl || count < 1 || world < 1 || rank < 0 || rank >= world || (dtype != 0 && dtype != 1)) return PHNET_ERR_ARG;
    if ((long)count * (dtype ? 2 : 1) > cap) return PHNET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 1)
        hipLaunchKernelGGL(oneshot_allreduce_kernel<double>, dim3(1), dim3(IPC_NT), 0, st, (double*)data, count,
                           (unsigned long long* const*)peers, rank, world, cap, (unsigned*)ctrl);
    else
        hipLaunchKernelGGL(oneshot_allreduce_kernel<float>, dim3(1), dim3(IPC_NT), 0, st, (float*)data, count,
                           (unsigned long long* const*)peers, rank, world, cap, (unsigned*)ctrl);
    return phnet_launch_status();
}
