// stream_probe.hip -- read-bandwidth ceilings for the access patterns used by k_flat_mfma (tooling, not product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// pattern A: each wave streams its own contiguous chunk of `chunk_kb` KB, then jumps by stride (like items)
template <int INFLIGHT>
__global__ __launch_bounds__(512) void k_chunks(const float4 *__restrict__ src, uint64_t n_kb, uint32_t chunk_kb, float *out) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint64_t stride = uint64_t(gridDim.x) * nw;
    float acc = 0.f;
    uint64_t n_chunks = n_kb / chunk_kb;
    for (uint64_t c = blockIdx.x * nw + wave; c < n_chunks; c += stride) {
        const float4 *p = src + c * chunk_kb * 64 + lane;
        for (uint32_t i = 0; i < chunk_kb; i += INFLIGHT) {
            float4 v[INFLIGHT];
#pragma unroll
            for (int j = 0; j < INFLIGHT; j++) v[j] = p[(i + j) * 64];
#pragma unroll
            for (int j = 0; j < INFLIGHT; j++) acc += v[j].x + v[j].y + v[j].z + v[j].w;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}
// pattern C: like A, but the grid is split into `share` groups that read the SAME chunk sequence concurrently
// (do concurrent readers of one stream share L2 / Infinity Cache, i.e. can two query batches ride one HBM pass?)
template <int INFLIGHT>
__global__ __launch_bounds__(512) void k_shared(const float4 *__restrict__ src, uint64_t n_kb, uint32_t chunk_kb, uint32_t share, float *out) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t per = gridDim.x / share;            // workgroups per group
    const uint32_t bid = blockIdx.x / share;           // adjacent block ids (different XCDs) share a chunk ...
    const uint64_t stride = uint64_t(per) * nw;
    float acc = 0.f;
    uint64_t n_chunks = n_kb / chunk_kb;
    for (uint64_t c = bid * nw + wave; c < n_chunks; c += stride) {
        const float4 *p = src + c * chunk_kb * 64 + lane;
        for (uint32_t i = 0; i < chunk_kb; i += INFLIGHT) {
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
__global__ __launch_bounds__(512) void k_shared_xcd(const float4 *__restrict__ src, uint64_t n_kb, uint32_t chunk_kb, uint32_t share, float *out) {
    // ... or block ids b and b + 8*k (same XCD under round-robin placement) share a chunk
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t per = gridDim.x / share;
    const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;       // slot within the XCD
    const uint32_t bid = (slot / share) * 8 + xcd;                    // group id
    const uint64_t stride = uint64_t(per) * nw;
    float acc = 0.f;
    uint64_t n_chunks = n_kb / chunk_kb;
    for (uint64_t c = bid * nw + wave; c < n_chunks; c += stride) {
        const float4 *p = src + c * chunk_kb * 64 + lane;
        for (uint32_t i = 0; i < chunk_kb; i += INFLIGHT) {
            float4 v[INFLIGHT];
#pragma unroll
            for (int j = 0; j < INFLIGHT; j++) v[j] = p[(i + j) * 64];
#pragma unroll
            for (int j = 0; j < INFLIGHT; j++) acc += v[j].x + v[j].y + v[j].z + v[j].w;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}
// pattern B: grid-stride linear (adjacent waves read adjacent KBs)
template <int INFLIGHT>
__global__ __launch_bounds__(512) void k_linear(const float4 *__restrict__ src, uint64_t n_kb, float *out) {
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

int main() {
    const uint64_t bytes = 3840000000ull, n_kb = bytes / 1024;
    float4 *src; float *out;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(src, 1, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char *name, auto launch) {
        for (int w = 0; w < 2; w++) launch();
        hipEventRecord(a);
        for (int r = 0; r < 10; r++) launch();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-40s %.3f ms  %.0f GB/s\n", name, ms / 10, bytes / (ms / 10 * 1e-3) / 1e9);
        return 0;
    };
    for (int grid : {256, 512, 1024}) {
        for (int nt : {256, 512}) {
            char nm[128];
            snprintf(nm, 128, "chunks120KB inflight10 grid%d nt%d", grid, nt);
            timeit(nm, [&]() { hipLaunchKernelGGL(k_chunks<10>, dim3(grid), dim3(nt), 0, 0, src, n_kb, 120, out); });
            snprintf(nm, 128, "chunks60KB inflight10 grid%d nt%d", grid, nt);
            timeit(nm, [&]() { hipLaunchKernelGGL(k_chunks<10>, dim3(grid), dim3(nt), 0, 0, src, n_kb, 60, out); });
            snprintf(nm, 128, "linear inflight10 grid%d nt%d", grid, nt);
            timeit(nm, [&]() { hipLaunchKernelGGL(k_linear<10>, dim3(grid), dim3(nt), 0, 0, src, n_kb, out); });
            snprintf(nm, 128, "linear inflight4 grid%d nt%d", grid, nt);
            timeit(nm, [&]() { hipLaunchKernelGGL(k_linear<4>, dim3(grid), dim3(nt), 0, 0, src, n_kb, out); });
        }
    }
    for (int share : {1, 2, 4}) {
        char nm[128];
        snprintf(nm, 128, "shared x%d (adjacent ids) grid256 nt512 [unique bytes]", share);
        timeit(nm, [&]() { hipLaunchKernelGGL(k_shared<10>, dim3(256), dim3(512), 0, 0, src, n_kb, 120, share, out); });
        snprintf(nm, 128, "shared x%d (same XCD) grid256 nt512 [unique bytes]", share);
        timeit(nm, [&]() { hipLaunchKernelGGL(k_shared_xcd<10>, dim3(256), dim3(512), 0, 0, src, n_kb, 120, share, out); });
        snprintf(nm, 128, "shared x%d (same XCD) grid512 nt256 [unique bytes]", share);
        timeit(nm, [&]() { hipLaunchKernelGGL(k_shared_xcd<10>, dim3(512), dim3(256), 0, 0, src, n_kb, 120, share, out); });
    }
    // high-occupancy linear
    timeit("linear inflight4 grid2048 nt256", [&]() { hipLaunchKernelGGL(k_linear<4>, dim3(2048), dim3(256), 0, 0, src, n_kb, out); });
    timeit("linear inflight2 grid4096 nt256", [&]() { hipLaunchKernelGGL(k_linear<2>, dim3(4096), dim3(256), 0, 0, src, n_kb, out); });
    return 0;
}
