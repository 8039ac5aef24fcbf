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

// ---- matrix-pipe rate of THIS box under sustained load ------------------------------------------------------------------
// v_mfma_f32_16x16x32_f16 (the Flat filter's instruction) issued back to back from `waves_per_simd` waves per SIMD on
// every CU, 8 independent accumulator tiles per wave: TFLOP/s over the whole launch and the shader clock the chip held
// meanwhile (s_memtime cycles of one wave over the event time).  Measurement hook only.
typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
typedef float pf32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_probe_mfma(uint32_t iters, float *out, unsigned long long *cycles) {
    pf32x4 acc[8];
    pf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a[i] = (_Float16)(float(threadIdx.x & 7) * 0.125f + float(i));
        b[i] = (_Float16)(float(threadIdx.x & 3) * 0.25f - float(i));
    }
#pragma unroll
    for (int t = 0; t < 8; t++) acc[t] = (pf32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = clock64();
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[t], 0, 0, 0);
    }
    const unsigned long long c1 = clock64();
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 8; t++) sum += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    if (sum == 12345.678f) out[0] = sum;
    if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = c1 - c0;
}
void mfma_probe(int device, int waves_per_simd, int iters, double *tflops, double *clock_ghz) {
    VDB_HIP(hipSetDevice(device));
    VDB_REQUIRE(waves_per_simd >= 1 && waves_per_simd <= 8 && iters >= 1 && iters <= (1 << 24), "mfma probe: waves_per_simd in 1..8, iters in 1..2^24");
    hipDeviceProp_t prop;
    VDB_HIP(hipGetDeviceProperties(&prop, device));
    const unsigned grid = (unsigned)prop.multiProcessorCount * (unsigned)waves_per_simd;  // 4 waves per block: one per SIMD
    void *out = nullptr, *cyc = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t ea = nullptr, eb = nullptr;
    try {
        VDB_HIP(hipMalloc(&out, 64));
        VDB_HIP(hipMalloc(&cyc, 64));
        VDB_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        VDB_HIP(hipEventCreate(&ea));
        VDB_HIP(hipEventCreate(&eb));
        hipLaunchKernelGGL(k_probe_mfma, dim3(grid), dim3(256), 0, s, (uint32_t)iters, static_cast<float *>(out), static_cast<unsigned long long *>(cyc));
        VDB_HIP(hipEventRecord(ea, s));
        hipLaunchKernelGGL(k_probe_mfma, dim3(grid), dim3(256), 0, s, (uint32_t)iters, static_cast<float *>(out), static_cast<unsigned long long *>(cyc));
        VDB_HIP(hipEventRecord(eb, s));
        VDB_HIP(hipEventSynchronize(eb));
        VDB_HIP(hipGetLastError());
        float ms = 0;
        VDB_HIP(hipEventElapsedTime(&ms, ea, eb));
        unsigned long long c = 0;
        VDB_HIP(hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost));
        const double flops = double(grid) * 4.0 * double(iters) * 8.0 * (2.0 * 16 * 16 * 32);
        *tflops = flops / (double(ms) * 1e-3) / 1e12;
        *clock_ghz = double(c) / (double(ms) * 1e-3) / 1e9;  // (the timed wave runs for all but the launch overhead of the event window)
    } catch (...) {
        if (ea) (void)hipEventDestroy(ea);
        if (eb) (void)hipEventDestroy(eb);
        if (s) (void)hipStreamDestroy(s);
        if (out) (void)hipFree(out);
        if (cyc) (void)hipFree(cyc);
        throw;
    }
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    (void)hipStreamDestroy(s);
    (void)hipFree(out);
    (void)hipFree(cyc);
}

}  // namespace vdb
