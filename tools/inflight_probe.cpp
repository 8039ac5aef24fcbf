// tools/inflight_probe.cpp -- streaming-read rate of the chip as a function of the bytes each CU keeps in flight: one 512-thread workgroup
// per CU (pinned by its LDS request), every wave streams its own contiguous region in 1-KB pieces (64 lanes x 16 B, the Flat filter's
// fragment loads) with D pieces in flight (a register ring, non-temporal or default loads).  The Flat 8-bit filter keeps 9 KB per wave
// = 72 KB per CU in flight; this says what that depth can reach and what a deeper ring would.  Measurement only.
//   hipcc --offload-arch=gfx950 -O2 -o .probe/inflight_probe tools/inflight_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int D, bool NT>
__global__ __launch_bounds__(512) void stream(const u32x4 *__restrict__ src, uint64_t kb_per_wave, uint32_t *out) {
    extern __shared__ char pin[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4 *p = src + (uint64_t(blockIdx.x) * 8 + wave) * kb_per_wave * 64 + lane;
    u32x4 ring[D];
    auto ld = [&](uint64_t i) -> u32x4 {
        if constexpr (NT) return __builtin_nontemporal_load(p + i * 64);
        else return p[i * 64];
    };
#pragma unroll
    for (int d = 0; d < D; d++) ring[d] = ld(d);
    u32x4 acc = {0, 0, 0, 0};
    uint64_t i = 0;
    for (; i + 2 * D <= kb_per_wave; i += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            acc ^= ring[d];
            ring[d] = ld(i + D + d);
        }
    }
#pragma unroll
    for (int d = 0; d < D; d++) acc ^= ring[d];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
    if (threadIdx.x == 0) pin[0] = 0;
}
template <int D, bool NT>
static int run(const u32x4 *buf, uint64_t bytes, int ncu, int wg_per_cu, uint32_t *out) {
    const int grid = ncu * wg_per_cu;
    const uint64_t kb_per_wave = bytes / 1024 / (uint64_t(grid) * 8) / D * D;
    const size_t lds = wg_per_cu == 1 ? 100 * 1024 : (wg_per_cu == 2 ? 70 * 1024 : 36 * 1024);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream<D, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((stream<D, NT>), dim3(grid), dim3(512), lds, 0, buf, kb_per_wave, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double moved = double(kb_per_wave) * 1024 * grid * 8;
    printf("  D %2d (%3d KB per CU in flight) %s: %.3f ms, %.2f TB/s\n", D, D * 8 * wg_per_cu, NT ? "nt" : "  ", best, moved / best / 1e9);
    return 0;
}
int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    const uint64_t bytes = 4ull << 30;
    u32x4 *buf;
    uint32_t *out;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, bytes));
    for (int w = 1; w <= 2; w++) {
        printf("%d workgroup(s) of 512 threads per CU, %d CUs, 4 GiB read per launch:\n", w, ncu);
        if (run<6, true>(buf, bytes, ncu, w, out)) return 1;
        if (run<9, true>(buf, bytes, ncu, w, out)) return 1;
        if (run<9, false>(buf, bytes, ncu, w, out)) return 1;
        if (run<12, true>(buf, bytes, ncu, w, out)) return 1;
        if (run<18, true>(buf, bytes, ncu, w, out)) return 1;
        if (run<27, true>(buf, bytes, ncu, w, out)) return 1;
    }
    return 0;
}
