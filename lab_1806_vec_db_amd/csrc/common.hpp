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

// a device allocation that failed: the one failure a search can route around (a mirror that cannot be built leaves the tier to the next)
struct AllocError : Error {
    explicit AllocError(const std::string &m) : Error(2, m) {}
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
// ---- 64-key bitonic networks across the lanes of a wave (no LDS) ---------------------------------------------------
// The partner of a compare-exchange stage is lane ^ j with j a power of two known at compile time (the networks are fully
// unrolled).  __shfl_xor is a ds_bpermute_b32 per 32-bit word -- an LDS-crossbar round trip of ~100 cycles that every stage of
// the chain waits for.  The same exchange inside the vector ALU: j = 1, 2 one quad_perm DPP move, j = 4 two row-shift DPP moves
// under complementary bank masks, j = 8 a row rotate by 8 (= ^ 8 in a ring of 16), j = 16 / 32 gfx950's v_permlane16_swap /
// v_permlane32_swap (a copy of the word swapped against itself holds the partner rows in the other operand).  All lanes of the
// wave must be active (as for the shuffle forms: a disabled lane supplies no data either way).
template <uint32_t J>
__device__ __forceinline__ uint32_t lane_xor32(uint32_t v, uint32_t lane) {
    static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "partner distance");
    if constexpr (J == 1) {
        return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);  // quad_perm:[1,0,3,2]
    } else if constexpr (J == 2) {
        return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);  // quad_perm:[2,3,0,1]
    } else if constexpr (J == 4) {
        int t = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xF, 0x5, false);  // row_shl:4 -> banks 0, 2 read lane + 4
        t = __builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xF, 0xA, false);          // row_shr:4 -> banks 1, 3 read lane - 4
        return (uint32_t)t;
    } else if constexpr (J == 8) {
        return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xF, 0xF, false);  // row_ror:8
    } else if constexpr (J == 16) {
        // odd rows of the first operand <-> even rows of the second: first = [r0 r0 r2 r2], second = [r1 r1 r3 r3]
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? r[0] : r[1];
    } else {
        // upper half of the first operand <-> lower half of the second: first = [lo lo], second = [hi hi]
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (lane & 32) ? r[0] : r[1];
    }
}
template <uint32_t J>
__device__ __forceinline__ uint64_t lane_xor64(uint64_t v, uint32_t lane) {
    return (uint64_t(lane_xor32<J>(uint32_t(v >> 32), lane)) << 32) | lane_xor32<J>(uint32_t(v), lane);
}
// value of lane 63 - l (the reverse a bitonic merge starts with): mirror inside the rows of 16, then swap the rows
__device__ __forceinline__ uint32_t lane_reverse32(uint32_t v, uint32_t lane) {
    const uint32_t m = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false);  // row_mirror
    return lane_xor32<32>(lane_xor32<16>(m, lane), lane);
}
__device__ __forceinline__ uint64_t lane_reverse64(uint64_t v, uint32_t lane) {
    return (uint64_t(lane_reverse32(uint32_t(v >> 32), lane)) << 32) | lane_reverse32(uint32_t(v), lane);
}
// compare-exchange with lane ^ J, keeping the smaller key where ((lane & J) == 0) == up
template <uint32_t J>
__device__ __forceinline__ uint64_t cmpx64_t(uint64_t v, uint32_t lane, bool up) {
#ifndef VDB_SHUFFLE_NETWORKS  // (make EXTRA=-DVDB_SHUFFLE_NETWORKS: the ds_bpermute forms, for A/B runs)
    const uint64_t o = lane_xor64<J>(v, lane);
#else
    const uint64_t o = __shfl_xor(v, J);
#endif
    const bool take_min = ((lane & J) == 0) == up;
    return ((v < o) == take_min) ? v : o;
}
// stages J, J/2, .. 1 of the bitonic network's step K over U independent key registers (interleaved: their latencies overlap)
template <uint32_t K, uint32_t J, int U>
__device__ __forceinline__ void bitonic_stages(uint64_t (&r)[U], uint32_t lane) {
#pragma unroll
    for (int u = 0; u < U; u++) r[u] = cmpx64_t<J>(r[u], lane, (lane & K) == 0);
    if constexpr (J > 1) bitonic_stages<K, J / 2, U>(r, lane);
}
// steps K, 2K, .. 64: ascending sort of every register across the lanes (K = 2: the 21 compare-exchange stages)
template <uint32_t K, int U>
__device__ __forceinline__ void bitonic_sort_from(uint64_t (&r)[U], uint32_t lane) {
    bitonic_stages<K, K / 2, U>(r, lane);
    if constexpr (K < 64) bitonic_sort_from<2 * K, U>(r, lane);
}
// ascending sort of one key per lane
__device__ __forceinline__ uint64_t sort64(uint64_t r, uint32_t lane) {
    uint64_t a[1] = {r};
    bitonic_sort_from<2, 1>(a, lane);
    return a[0];
}
// a, b ascending across the lanes -> the 64 smallest of both, ascending
__device__ __forceinline__ uint64_t merge64(uint64_t a, uint64_t b, uint32_t lane) {
#ifndef VDB_SHUFFLE_NETWORKS
    const uint64_t rev = lane_reverse64(b, lane);
#else
    const uint64_t rev = __shfl(b, 63 - lane);
#endif
    uint64_t m[1] = {a < rev ? a : rev};
    bitonic_stages<64, 32, 1>(m, lane);  // (lane & 64) == 0 everywhere: ascending
    return m[0];
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
        bitonic_sort_from<2, 4>(r, lane);
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
