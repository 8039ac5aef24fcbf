// k_probe.hip -- attainable HBM read bandwidth of THIS box ("measure the attainable peak on the box with a
// streaming-read microkernel and report both").  Measurement hook only: the search paths never call it.
// Two access patterns, both pure reads with a dead reduction:
//   chunks: every wave streams its own contiguous 120-KB chunk, 10 x 1 KB in flight per wave (the pattern of the Flat
//           filter kernels: a wave's row tiles are one contiguous stream);
//   linear: grid-stride, adjacent waves read adjacent KBs, 4 loads in flight.
// Each with default and with NON-TEMPORAL loads (the filter kernels' row streams are non-temporal beyond the Infinity Cache); the
// best of the four is reported.  Until the last sessions of round 3 only the default loads were probed and 6.1 - 6.2 TB/s passed for
// "what the box attains"; non-temporal streaming reads reach 7.0 TB/s on the same boxes (tools/inflight_probe.cpp), and that is the
// ceiling the filter kernels are held against.
#include <algorithm>
#include <numeric>
#include <vector>

#include "common.hpp"

namespace vdb {

double stream_probe_pattern(int device, uint64_t bytes, int iters, int pattern_sel, uint32_t row_bytes);

typedef float probe_f4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 probe_load(const float4 *p) {
    if constexpr (NT) {
        const probe_f4 v = __builtin_nontemporal_load(reinterpret_cast<const probe_f4 *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return *p;
    }
}
template <int INFLIGHT, bool NT = false>
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
            for (int j = 0; j < INFLIGHT; j++) v[j] = probe_load<NT>(p + (i + j) * 64);
#pragma unroll
            for (int j = 0; j < INFLIGHT; j++) acc += v[j].x + v[j].y + v[j].z + v[j].w;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int INFLIGHT, bool NT = false>
__global__ __launch_bounds__(256) void k_probe_linear(const float4 *__restrict__ src, uint64_t n_kb, float *out) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint64_t stride = uint64_t(gridDim.x) * nw;
    float acc = 0.f;
    for (uint64_t kb = blockIdx.x * nw + wave; kb + (INFLIGHT - 1) * stride < n_kb; kb += stride * INFLIGHT) {
        float4 v[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++) v[j] = probe_load<NT>(src + (kb + j * stride) * 64 + lane);
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++) acc += v[j].x + v[j].y + v[j].z + v[j].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// The Flat filter's operand loads if its fp16 mirror were the ROW-MAJOR image the HNSW / IVF gathers use (one fp16 copy instead
// of two): a wave owns 48 rows (three 16-row tiles) of row_bytes each and walks them in 64-B k-blocks; one A-fragment load of
// v_mfma_f32_16x16x32_f16 is then lane l -> row (l & 15), bytes 16 (l >> 4) of the k-block: 16 half lines 1 920 B apart per
// instruction instead of one contiguous KB.  INFLIGHT k-block pairs (= whole lines) per tile in flight.
template <int INFLIGHT>
__global__ __launch_bounds__(512) void k_probe_rowfrag(const float4 *__restrict__ src, uint64_t n_rows, uint32_t row_f4, float *out) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint64_t stride = uint64_t(gridDim.x) * nw, n_units = n_rows / 48;
    float acc = 0.f;
    for (uint64_t u = blockIdx.x * nw + wave; u < n_units; u += stride) {
        const float4 *p0 = src + (u * 48 + (lane & 15)) * row_f4 + (lane >> 4);
        for (uint32_t kb = 0; kb + 2 * INFLIGHT <= row_f4 / 4; kb += 2 * INFLIGHT) {  // a 64-B k-block = 4 pieces of 16 B
            float4 v[INFLIGHT][3][2];
#pragma unroll
            for (int j = 0; j < INFLIGHT; j++)
#pragma unroll
                for (int t = 0; t < 3; t++)
#pragma unroll
                    for (int h = 0; h < 2; h++) v[j][t][h] = p0[uint64_t(t) * 16 * row_f4 + (kb + 2 * j + h) * 4];
#pragma unroll
            for (int j = 0; j < INFLIGHT; j++)
#pragma unroll
                for (int t = 0; t < 3; t++)
#pragma unroll
                    for (int h = 0; h < 2; h++) acc += v[j][t][h].x + v[j][t][h].y + v[j][t][h].z + v[j][t][h].w;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

// pattern 0: best of chunks / linear (the attainable streaming rate); pattern 1: k_probe_rowfrag over rows of row_bytes
// returns GB/s (1e9 bytes per second) of the better pattern; bytes is rounded down to whole 120-KB chunks
double stream_probe(int device, uint64_t bytes, int iters) { return stream_probe_pattern(device, bytes, iters, 0, 0); }
double stream_probe_pattern(int device, uint64_t bytes, int iters, int pattern_sel, uint32_t row_bytes) {
    VDB_HIP(hipSetDevice(device));
    VDB_REQUIRE(pattern_sel == 0 || (pattern_sel == 1 && row_bytes >= 512 && row_bytes % 128 == 0 && row_bytes <= 65536),
                "stream probe: pattern 0, or 1 with rows of a multiple of 128 bytes (>= 512)");
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
        for (int pattern = pattern_sel ? 2 : 0; pattern < (pattern_sel ? 3 : 5); pattern++) {
            if (!pattern_sel && pattern == 2) continue;  // (the row-fragment pattern is probed on request only)
            auto launch = [&]() {
                if (pattern == 3)
                    hipLaunchKernelGGL((k_probe_chunks<10, true>), dim3(256), dim3(512), 0, s, static_cast<const float4 *>(src), n_kb, 120u,
                                       static_cast<float *>(out));
                else if (pattern == 4)
                    hipLaunchKernelGGL((k_probe_linear<4, true>), dim3(2048), dim3(256), 0, s, static_cast<const float4 *>(src), n_kb,
                                       static_cast<float *>(out));
                else if (pattern == 2)
                    hipLaunchKernelGGL(k_probe_rowfrag<4>, dim3(256), dim3(512), 0, s, static_cast<const float4 *>(src), n_kb * 1024 / row_bytes,
                                       row_bytes / 16, static_cast<float *>(out));
                else if (pattern == 0)
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
            const double moved = pattern == 2 ? double(n_kb * 1024 / row_bytes / 48 * 48) * double(row_bytes / 512 * 512) : double(n_kb) * 1024.0;
            best = std::max(best, moved * iters / (double(ms) * 1e-3) / 1e9);
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
// the 8-bit filter's instruction (k_gemm8.hip): v_mfma_i32_16x16x64_i8, 2 x 16 x 16 x 64 integer operations each
typedef int pi32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_probe_mfma_i8(uint32_t iters, float *out, unsigned long long *cycles) {
    pi32x4 acc[8], a, b;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a[i] = int(threadIdx.x * 0x01010101u + i * 0x00010203u);
        b[i] = int(threadIdx.x * 0x01020304u - i * 0x01000100u);
    }
#pragma unroll
    for (int t = 0; t < 8; t++) acc[t] = (pi32x4){0, 0, 0, 0};
    const unsigned long long c0 = clock64();
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[t], 0, 0, 0);
    }
    const unsigned long long c1 = clock64();
    int sum = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) sum += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    if (sum == 0x12345678) out[0] = float(sum);
    if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = c1 - c0;
}
void mfma_probe(int device, int waves_per_simd, int iters, double *tflops, double *clock_ghz, int i8) {
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
        auto kern = i8 ? k_probe_mfma_i8 : k_probe_mfma;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, (uint32_t)iters, static_cast<float *>(out), static_cast<unsigned long long *>(cyc));
        VDB_HIP(hipEventRecord(ea, s));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, (uint32_t)iters, static_cast<float *>(out), static_cast<unsigned long long *>(cyc));
        VDB_HIP(hipEventRecord(eb, s));
        VDB_HIP(hipEventSynchronize(eb));
        VDB_HIP(hipGetLastError());
        float ms = 0;
        VDB_HIP(hipEventElapsedTime(&ms, ea, eb));
        unsigned long long c = 0;
        VDB_HIP(hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost));
        const double flops = double(grid) * 4.0 * double(iters) * 8.0 * (2.0 * 16 * 16 * (i8 ? 64 : 32));
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

// ---- latency of ONE dependent HBM access on this box ------------------------------------------------------------------------
// The graph walks (k_hnsw_search) are chains of dependent accesses -- pop -> link row -> visited words -> first row lines -- so
// their floor is a latency, not a bandwidth.  One lane follows a random cyclic permutation (Sattolo) over 128-B lines of a buffer
// larger than every cache: nanoseconds per hop = the round trip of a dependent global load that misses L2 and the Infinity
// Cache.  Measurement hook only.
__global__ __launch_bounds__(64) void k_probe_chase(const uint64_t *__restrict__ buf, uint32_t hops, uint64_t *out) {
    if (threadIdx.x != 0) return;
    uint64_t i = 0;
    for (uint32_t h = 0; h < hops; h++) i = buf[i * 16];  // 16 x 8 B = one 128-B line per node
    out[0] = i;
}
double latency_probe(int device, uint64_t bytes, uint32_t hops) {
    VDB_HIP(hipSetDevice(device));
    VDB_REQUIRE(bytes >= (1ull << 20) && bytes <= (16ull << 30), "latency probe: bytes must be in 1 MiB .. 16 GiB");
    VDB_REQUIRE(hops >= 16 && hops <= (1u << 24), "latency probe: hops must be in 16..2^24");
    const uint64_t lines = bytes / 128;
    std::vector<uint32_t> perm(lines);
    std::iota(perm.begin(), perm.end(), 0u);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto next = [&]() {
        st += 0x9E3779B97F4A7C15ull;
        uint64_t z = st;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    for (uint64_t i = lines - 1; i > 0; i--) std::swap(perm[i], perm[next() % i]);  // Sattolo: one cycle through every line
    std::vector<uint64_t> host(lines * 16, 0);
    for (uint64_t i = 0; i < lines; i++) host[i * 16] = perm[i];
    void *buf = nullptr, *out = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    double ns = 0.0;
    try {
        VDB_HIP(hipMalloc(&buf, lines * 128));
        VDB_HIP(hipMalloc(&out, 64));
        VDB_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        VDB_HIP(hipEventCreate(&a));
        VDB_HIP(hipEventCreate(&b));
        VDB_HIP(hipMemcpy(buf, host.data(), lines * 128, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_probe_chase, dim3(1), dim3(64), 0, s, static_cast<const uint64_t *>(buf), 64u, static_cast<uint64_t *>(out));
        VDB_HIP(hipEventRecord(a, s));
        hipLaunchKernelGGL(k_probe_chase, dim3(1), dim3(64), 0, s, static_cast<const uint64_t *>(buf), hops, static_cast<uint64_t *>(out));
        VDB_HIP(hipEventRecord(b, s));
        VDB_HIP(hipEventSynchronize(b));
        VDB_HIP(hipGetLastError());
        float ms = 0;
        VDB_HIP(hipEventElapsedTime(&ms, a, b));
        ns = double(ms) * 1e6 / double(hops);
    } catch (...) {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
        if (s) (void)hipStreamDestroy(s);
        if (buf) (void)hipFree(buf);
        if (out) (void)hipFree(out);
        throw;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipStreamDestroy(s);
    (void)hipFree(buf);
    (void)hipFree(out);
    return ns;
}

// ---- latency of ONE dependent f32 add: the other half of a walk's floor -------------------------------------------------------
// The reference's distances are strict left folds (distance/mod.rs:72-77): a row of d columns is a chain of d dependent adds
// whatever the lane count.  One wave, `adds` dependent v_add_f32 (the compiler cannot re-associate: -fno-fast-math): ns per add.
__global__ __launch_bounds__(64) void k_probe_fold(uint32_t adds, float x, float *out) {
    float acc = 0.0f;
    for (uint32_t i = 0; i < adds; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            acc = acc + x;
            asm volatile("" : "+v"(acc));  // (keeps the eight adds eight instructions)
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}
double fold_probe(int device, uint32_t adds) {
    VDB_HIP(hipSetDevice(device));
    VDB_REQUIRE(adds >= 1024 && adds <= (1u << 28), "fold probe: adds must be in 1024..2^28");
    void *out = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    double ns = 0.0;
    try {
        VDB_HIP(hipMalloc(&out, 64));
        VDB_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        VDB_HIP(hipEventCreate(&a));
        VDB_HIP(hipEventCreate(&b));
        hipLaunchKernelGGL(k_probe_fold, dim3(1), dim3(64), 0, s, 1024u, 1.0f, static_cast<float *>(out));
        VDB_HIP(hipEventRecord(a, s));
        hipLaunchKernelGGL(k_probe_fold, dim3(1), dim3(64), 0, s, adds, 1.0f, static_cast<float *>(out));
        VDB_HIP(hipEventRecord(b, s));
        VDB_HIP(hipEventSynchronize(b));
        VDB_HIP(hipGetLastError());
        float ms = 0;
        VDB_HIP(hipEventElapsedTime(&ms, a, b));
        ns = double(ms) * 1e6 / double(adds);
    } catch (...) {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
        if (s) (void)hipStreamDestroy(s);
        if (out) (void)hipFree(out);
        throw;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipStreamDestroy(s);
    (void)hipFree(out);
    return ns;
}

}  // namespace vdb
