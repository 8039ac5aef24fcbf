// tools/l2share_probe.cpp -- can the workgroups of one XCD share a row stream through their 4-MB L2?  256 workgroups of 512 threads (one per CU);
// workgroup b is taken to run on XCD b % 8 (round-robin dispatch); of the 32 workgroups of an XCD, G form a set that reads the SAME units in
// the same order at the same time (as G query groups of a Flat filter pass would), 32 / G sets split the XCD's share of the units.  G = 1 is
// the plain stream (every unit read once, 0.96 GB); G = 8 reads every unit 8 times (7.68 GB) of which 7 can hit the L2 if the set stays
// together.  Reported: bytes read per second over all reads.  Measurement only.
//   hipcc --offload-arch=gfx950 -O2 -o .probe/l2share_probe tools/l2share_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int UNIT_KB = 45, D = 9;

__global__ __launch_bounds__(512) void share(const u32x4 *__restrict__ src, uint32_t n_units, uint32_t G, uint32_t *out, uint32_t *xcc_seen) {
    extern __shared__ char pin[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t xcd = blockIdx.x & 7, li = blockIdx.x >> 3;  // 32 workgroups per XCD
    const uint32_t slices = 32 / G, s = li / G;                 // (li % G = the set member: which "query group")
    if (threadIdx.x == 0) {
        uint32_t x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcc_seen[blockIdx.x] = x & 0xF;
    }
    u32x4 acc = {0, 0, 0, 0};
    // the units of (xcd, slice s, wave): u = ((k * slices + s) * 8 + wave) * 8 + xcd
    for (uint32_t k = 0;; k++) {
        const uint32_t u = ((k * slices + s) * 8 + wave) * 8 + xcd;
        if (u >= n_units) break;
        const u32x4 *p = src + uint64_t(u) * UNIT_KB * 64 + lane;
        u32x4 ring[D];
#pragma unroll
        for (int d = 0; d < D; d++) ring[d] = p[d * 64];
#pragma unroll
        for (int i = D; i < UNIT_KB; i += D) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                acc ^= ring[d];
                ring[d] = p[(i + d) * 64];
            }
        }
#pragma unroll
        for (int d = 0; d < D; d++) acc ^= ring[d];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
    if (threadIdx.x == 0) pin[0] = 0;
}
int main() {
    const uint32_t n_units = 20834;
    const uint64_t bytes = uint64_t(n_units) * UNIT_KB * 1024;
    u32x4 *buf;
    uint32_t *out, *xs;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMalloc(&xs, 256 * 4));
    CK(hipMemset(buf, 1, bytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(share), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (uint32_t G : {1u, 2u, 4u, 8u, 16u, 32u}) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(share, dim3(256), dim3(512), 100 * 1024, 0, buf, n_units, G, out, xs);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("  sets of %2u workgroups read the same units: %.3f ms for %u x 0.96 GB = %.2f TB/s read\n", G, best, G, double(bytes) * G / best / 1e9);
    }
    uint32_t h[256];
    CK(hipMemcpy(h, xs, sizeof(h), hipMemcpyDeviceToHost));
    int ok = 0;
    for (int b = 0; b < 256; b++) ok += h[b] == uint32_t(b & 7);
    printf("  workgroups that ran on XCD blockIdx %% 8: %d of 256\n", ok);
    return 0;
}
