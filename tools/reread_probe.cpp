// tools/reread_probe.cpp -- what a unit-major Flat filter could count on: every wave reads its own 45-KB unit (45 x 1 KB) `passes` times in a row
// before it moves to the next unit, 8 waves per CU (one 512-thread workgroup pinned by its LDS request), units dealt round-robin over the
// chip's waves as the filter deals them.  With passes = 1 this is the plain stream; with passes = 8 seven of eight reads of a line come
// from wherever the first left it (L2: 4 MB per XCD against 360 KB per CU and round; Infinity Cache behind it).  Reported: bytes READ per
// second (all passes) and the time per unit-pass.  Measurement only.
//   hipcc --offload-arch=gfx950 -O2 -o .probe/reread_probe tools/reread_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int UNIT_KB = 45, D = 9;  // 5 x 9 KB per unit-pass

template <bool NT_FIRST>
__global__ __launch_bounds__(512) void reread(const u32x4 *__restrict__ src, uint32_t n_units, int passes, uint32_t *out) {
    extern __shared__ char pin[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = gridDim.x * 8, gw = blockIdx.x * 8 + wave;
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t u = gw; u < n_units; u += nwaves) {
        const u32x4 *p = src + uint64_t(u) * UNIT_KB * 64 + lane;
        for (int ps = 0; ps < passes; ps++) {
            asm volatile("" : "+v"(p));  // (the passes read the same addresses: keep the compiler from hoisting them)
            u32x4 ring[D];
#pragma unroll
            for (int d = 0; d < D; d++) ring[d] = (NT_FIRST && ps == 0) ? __builtin_nontemporal_load(p + d * 64) : p[d * 64];
#pragma unroll
            for (int i = D; i < UNIT_KB; i += D) {
#pragma unroll
                for (int d = 0; d < D; d++) {
                    acc ^= ring[d];
                    ring[d] = (NT_FIRST && ps == 0) ? __builtin_nontemporal_load(p + (i + d) * 64) : p[(i + d) * 64];
                }
            }
#pragma unroll
            for (int d = 0; d < D; d++) acc ^= ring[d];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
    if (threadIdx.x == 0) pin[0] = 0;
}
int main() {
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int ncu = pr.multiProcessorCount;
    const uint32_t n_units = 20834;  // 1M rows / 48
    const uint64_t bytes = uint64_t(n_units) * UNIT_KB * 1024;
    u32x4 *buf;
    uint32_t *out;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, bytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(reread<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(reread<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%d CUs, %u units of %d KB (%.0f MB), one 512-thread workgroup per CU\n", ncu, n_units, UNIT_KB, bytes / 1e6);
    for (int nt = 0; nt < 2; nt++)
        for (int passes : {1, 2, 4, 8}) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0, 0));
                if (nt) hipLaunchKernelGGL(reread<true>, dim3(ncu), dim3(512), 100 * 1024, 0, buf, n_units, passes, out);
                else hipLaunchKernelGGL(reread<false>, dim3(ncu), dim3(512), 100 * 1024, 0, buf, n_units, passes, out);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("  first pass %s, %d pass(es) per unit: %.3f ms, %.2f TB/s read (%.3f ms per pass over the whole buffer)\n", nt ? "non-temporal" : "default     ",
                   passes, best, double(bytes) * passes / best / 1e9, best / passes);
        }
    return 0;
}
