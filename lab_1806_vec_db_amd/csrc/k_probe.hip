// k_probe.hip -- attainable HBM read bandwidth of THIS box ("measure the attainable peak on the box with a
// streaming-read microkernel and report both").  Measurement hook only: the search paths never call it.
// Two access patterns, both pure reads with a dead reduction:
//   chunks: every wave streams its own contiguous 120-KB chunk, 10 x 1 KB in flight per wave (the pattern of the Flat
//           filter kernels: a wave's row tiles are one contiguous stream);
//   linear: grid-stride, adjacent waves read adjacent KBs, 4 loads in flight.
// The best of the two is reported; the buffer is larger than the 256-MB Infinity Cache and read non-temporally is not
// needed: every pass touches `bytes` of distinct lines.
#include <algorithm>

#include "common.hpp"

namespace vdb {

template <int INFLIGHT>
__global__ __launch_bounds__(512) void k_probe_chunks(const float4 *__restrict__ src, uint64_t n_kb, uint32_t chunk_kb,
                                                      float *out) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint64_t stride = uint64_t(gridDim.x) * nw;
    float acc = 0.f;
    const uint64_t n_chunks = n_kb / chunk_kb;
    for (uint64_t c = blockIdx.x * nw + wave; c < n_chunks; c += stride) {
        const float4 *p = src + c * chunk_kb * 64 + lane;
        for (uint32_t i = 0; i + INFLIGHT <= chunk_kb; i += INFLIGHT) {
            float4 v[INFLIGHT];
#pragma unroll
            for (int j = 0; j < INFLIGHT; j++) v[j] = p[(i + j) * 64];
#pragma unroll
            for (int j = 0; j < INFLIGHT; j++) acc += v[j].x + v[j].y + v[j].z + v[j].w;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int INFLIGHT>
__global__ __launch_bounds__(256) void k_probe_linear(const float4 *__restrict__ src, uint64_t n_kb, float *out) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint64_t stride = uint64_t(gridDim.x) * nw;
    float acc = 0.f;
    for (uint64_t kb = blockIdx.x * nw + wave; kb + (INFLIGHT - 1) * stride < n_kb; kb += stride * INFLIGHT) {
        float4 v[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++) v[j] = src[(kb + j * stride) * 64 + lane];
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++) acc += v[j].x + v[j].y + v[j].z + v[j].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// returns GB/s (1e9 bytes per second) of the better pattern; bytes is rounded down to whole 120-KB chunks
double stream_probe(int device, uint64_t bytes, int iters) {
    VDB_HIP(hipSetDevice(device));
    VDB_REQUIRE(bytes >= (64ull << 20) && bytes <= (64ull << 30), "stream probe: bytes must be in 64 MiB .. 64 GiB");
    VDB_REQUIRE(iters >= 1 && iters <= 1000, "stream probe: iters must be in 1..1000");
    const uint64_t n_kb = bytes / (120 * 1024) * 120;
    void *src = nullptr, *out = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    double best = 0.0;
    try {
        VDB_HIP(hipMalloc(&src, n_kb * 1024));
        VDB_HIP(hipMalloc(&out, 64));
        VDB_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        VDB_HIP(hipEventCreate(&a));
        VDB_HIP(hipEventCreate(&b));
        VDB_HIP(hipMemsetAsync(src, 1, n_kb * 1024, s));
        for (int pattern = 0; pattern < 2; pattern++) {
            auto launch = [&]() {
                if (pattern == 0)
                    hipLaunchKernelGGL(k_probe_chunks<10>, dim3(256), dim3(512), 0, s, static_cast<const float4 *>(src), n_kb,
                                       120u, static_cast<float *>(out));
                else
                    hipLaunchKernelGGL(k_probe_linear<4>, dim3(2048), dim3(256), 0, s, static_cast<const float4 *>(src), n_kb,
                                       static_cast<float *>(out));
            };
            launch();
            launch();
            VDB_HIP(hipEventRecord(a, s));
            for (int r = 0; r < iters; r++) launch();
            VDB_HIP(hipEventRecord(b, s));
            VDB_HIP(hipEventSynchronize(b));
            VDB_HIP(hipGetLastError());
            float ms = 0;
            VDB_HIP(hipEventElapsedTime(&ms, a, b));
            best = std::max(best, double(n_kb) * 1024.0 * iters / (double(ms) * 1e-3) / 1e9);
        }
    } catch (...) {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
        if (s) (void)hipStreamDestroy(s);
        if (src) (void)hipFree(src);
        if (out) (void)hipFree(out);
        throw;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipStreamDestroy(s);
    (void)hipFree(src);
    (void)hipFree(out);
    return best;
}

}  // namespace vdb
