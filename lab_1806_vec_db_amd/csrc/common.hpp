// common.hpp -- shared host/device helpers for libvdbhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>

namespace vdb {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define VDB_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            throw ::vdb::Error(2, std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + \
                                      ":" + std::to_string(__LINE__) + ")");                            \
    } while (0)

// hipLaunchKernelGGL reports a rejected launch (bad grid, LDS over the limit ...) only through the sticky last-error
// slot.  Every synchronisation point of the library goes through here, so a rejected launch surfaces as an error
// of the call that issued it instead of as stale output buffers.
#define VDB_SYNC(stream)                            \
    do {                                            \
        VDB_HIP(hipStreamSynchronize(stream));      \
        VDB_HIP(hipGetLastError());                 \
    } while (0)

#define VDB_REQUIRE(cond, msg)                          \
    do {                                                \
        if (!(cond)) throw ::vdb::Error(1, (msg));      \
    } while (0)

// Opt a kernel in to more dynamic LDS than the default limit, once per (kernel, device).  Read-side entry points are
// re-entrant and vdb_ctx drives one host thread per GPU, so the "done" set is guarded and keyed by the current device (a
// plain `static bool` raced between threads and only ever covered the device that happened to launch first).
inline void func_max_lds(const void *fn, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    VDB_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(mu);
    if (done.count({fn, dev})) return;
    VDB_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({fn, dev});
}

// ---- total order of CandidatePair (candidate_pair.rs:36-41) as one u64 ---------------------
// OrderedFloat<f32>: NaN greatest, all NaN equal, -0 == +0.  Canonicalise, then the usual
// sign-flip makes unsigned integer order equal to the float order; the index breaks ties.
__host__ __device__ inline uint32_t f32_orderable(float f) {
    uint32_t u;
    f = f + 0.0f;  // -0 -> +0
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(f);
#else
    __builtin_memcpy(&u, &f, 4);
#endif
    if ((u & 0x7fffffffu) > 0x7f800000u) u = 0x7fc00000u;  // any NaN -> canonical +NaN
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float f32_from_orderable(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
#endif
}
__host__ __device__ inline uint64_t pair_key(float d, uint32_t idx) {
    return (uint64_t(f32_orderable(d)) << 32) | idx;
}
constexpr uint64_t PAIR_NONE = ~0ull;  // sorts after every real pair


#if defined(__HIPCC__)
// ---- 64-key bitonic networks across the lanes of a wave (shuffles, no LDS) ----------------------------------------
__device__ __forceinline__ uint64_t cmpx64(uint64_t v, uint32_t lane, uint32_t j, bool up) {
    const uint64_t o = __shfl_xor(v, j);
    const bool take_min = ((lane & j) == 0) == up;
    return ((v < o) == take_min) ? v : o;
}
// ascending sort of one key per lane (21 compare-exchange stages)
__device__ __forceinline__ uint64_t sort64(uint64_t r, uint32_t lane) {
#pragma unroll
    for (uint32_t k = 2; k <= 64; k <<= 1)
#pragma unroll
        for (uint32_t j = k >> 1; j > 0; j >>= 1) r = cmpx64(r, lane, j, (lane & k) == 0);
    return r;
}
// a, b ascending across the lanes -> the 64 smallest of both, ascending
__device__ __forceinline__ uint64_t merge64(uint64_t a, uint64_t b, uint32_t lane) {
    const uint64_t rev = __shfl(b, 63 - lane);
    uint64_t m = a < rev ? a : rev;
#pragma unroll
    for (uint32_t j = 32; j > 0; j >>= 1) m = cmpx64(m, lane, j, true);
    return m;
}
// The 64 smallest of `total` keys at src, ascending across the lanes of wave 0 (other waves: unspecified).  Block of
// 256 threads; a wave takes every fourth batch of 4 x 64 keys, sorts the four runs (interleaved, so that the shuffle
// latencies overlap), merges them pairwise and into its running best-64; the four lists meet in LDS.  A batch without
// a key below the wave's current kth smallest is skipped (kth = 64: the full list; a caller that only needs the kth smallest
// passes its k -- lanes >= kth of the result are then unspecified -- and skips more).  Contains one __syncthreads().
// AGENT_LOADS: the keys were written by other workgroups of the SAME launch with agent-scope atomic stores; they are read with
// agent-scope atomic loads (sc1: served from the coherence point, not from a possibly stale line of this XCD's L2), so that
// the caller needs no acquire fence (which would invalidate the whole L2).
template <bool AGENT_LOADS = false>
__device__ __forceinline__ uint64_t block_top64(const uint64_t *__restrict__ src, uint32_t total, uint64_t (*sbest)[64], uint32_t kth = 64) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t batches = (total + 255) / 256;
    uint64_t best = PAIR_NONE;
    for (uint32_t bt = wave; bt < batches; bt += 4) {
        uint64_t r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = (bt * 4 + u) * 64 + lane;
            if (AGENT_LOADS)
                r[u] = i < total ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : PAIR_NONE;
            else
                r[u] = i < total ? src[i] : PAIR_NONE;
        }
        const uint64_t tau = __shfl(best, kth - 1);
        uint64_t lo = r[0] < r[1] ? r[0] : r[1], lo2 = r[2] < r[3] ? r[2] : r[3];
        lo = lo < lo2 ? lo : lo2;
        if (__ballot(lo < tau) == 0) continue;  // wave-uniform
#pragma unroll
        for (uint32_t k = 2; k <= 64; k <<= 1)
#pragma unroll
            for (uint32_t j = k >> 1; j > 0; j >>= 1)
#pragma unroll
                for (int u = 0; u < 4; u++) r[u] = cmpx64(r[u], lane, j, (lane & k) == 0);
        const uint64_t a = merge64(r[0], r[1], lane), b = merge64(r[2], r[3], lane);
        best = merge64(best, merge64(a, b, lane), lane);
    }
    sbest[wave][lane] = best;
    __syncthreads();
    if (wave != 0) return PAIR_NONE;
    const uint64_t a = merge64(sbest[0][lane], sbest[1][lane], lane), b = merge64(sbest[2][lane], sbest[3][lane], lane);
    return merge64(a, b, lane);
}
#endif

}  // namespace vdb
