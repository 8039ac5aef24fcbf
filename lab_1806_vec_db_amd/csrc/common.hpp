// common.hpp -- shared host/device helpers for libvdbhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

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

}  // namespace vdb
