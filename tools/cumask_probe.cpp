// tools/cumask_probe.cpp -- does a CU-masked HIP stream confine a kernel to a subset of the chip's CUs on this box, which bits are which
// CUs, and does a kernel on the complementary mask run beside a chip-filling one?  Measurement only.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/cumask_probe tools/cumask_probe.cpp && /tmp/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

__global__ void where(uint32_t *out, int spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(32);
}
// a chip-filling hog: one 512-thread workgroup per CU, all of the CU's LDS, spins for `ticks` of s_memtime
__global__ __launch_bounds__(512) void hog(unsigned long long ticks, uint32_t *out) {
    extern __shared__ char lds[];
    lds[threadIdx.x] = 1;
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0) out[blockIdx.x] = lds[5];
}
__global__ void small(uint32_t *out, int iters) {
    float a = threadIdx.x;
    for (int i = 0; i < iters; i++) a = a * 1.0001f + 0.5f;
    if (a == 123.f) out[0] = 1;
}
static void report(const char *tag, const std::vector<uint32_t> &h, int n) {
    std::map<uint32_t, std::set<uint32_t>> cus;  // xcc -> {se/sh/cu field}
    for (int i = 0; i < n; i++) cus[h[2 * i + 1] & 0xF].insert((h[2 * i] >> 8) & 0xFF);
    size_t tot = 0;
    printf("%s:", tag);
    for (auto &kv : cus) { printf(" xcc%u:%zu", kv.first, kv.second.size()); tot += kv.second.size(); }
    printf("  total %zu distinct (xcc, se/sh/cu)\n", tot);
}
int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("device %s, %d CUs\n", p.name, ncu);
    const int nwg = 8192;
    uint32_t *d;
    CK(hipMalloc(&d, nwg * 8));
    std::vector<uint32_t> h(nwg * 2);
    auto run = [&](hipStream_t s, const char *tag) -> int {
        CK(hipMemsetAsync(d, 0, nwg * 8, s));
        hipLaunchKernelGGL(where, dim3(nwg), dim3(64), 0, s, d, 20);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost));
        report(tag, h, nwg);
        return 0;
    };
    hipStream_t s0;
    CK(hipStreamCreate(&s0));
    if (run(s0, "no mask")) return 1;
    const int words = (ncu + 31) / 32;
    struct { const char *tag; int lo, hi, stride; } masks[] = {{"bits 0..15", 0, 16, 1}, {"bits 0..31", 0, 32, 1}, {"bits 240..255", 240, 256, 1},
                                                             {"every 16th bit", 0, 256, 16}, {"bits 0..239", 0, 240, 1}, {"bits 16..255", 16, 256, 1}};
    for (auto &m : masks) {
        std::vector<uint32_t> mask(words, 0);
        for (int b = m.lo; b < m.hi && b < ncu; b += m.stride) mask[b / 32] |= 1u << (b % 32);
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, words, mask.data());
        if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", m.tag, hipGetErrorString(e)); continue; }
        if (run(s, m.tag)) return 1;
        CK(hipStreamDestroy(s));
    }
    // concurrency: hog on bits 16..255 (240 workgroups), small kernel on bits 0..15, against the small kernel alone and against an unmasked hog
    std::vector<uint32_t> mbig(words, 0), msmall(words, 0);
    for (int b = 0; b < ncu; b++) (b < 16 ? msmall : mbig)[b / 32] |= 1u << (b % 32);
    hipStream_t sb, ss;
    CK(hipExtStreamCreateWithCUMask(&sb, words, mbig.data()));
    CK(hipExtStreamCreateWithCUMask(&ss, words, msmall.data()));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(hog), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_small = [&](hipStream_t s, float *ms) -> int {
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(small, dim3(64), dim3(256), 0, s, d, 20000);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(ms, e0, e1));
        return 0;
    };
    float t_alone = 0, t_masked_hog = 0, t_full_hog = 0;
    if (time_small(ss, &t_alone)) return 1;
    if (time_small(ss, &t_alone)) return 1;
    const unsigned long long ticks = 4000000ull;  // ~2 ms
    hipLaunchKernelGGL(hog, dim3(240), dim3(512), 150 * 1024, sb, ticks, d + 4096);
    if (time_small(ss, &t_masked_hog)) return 1;
    CK(hipStreamSynchronize(sb));
    hipLaunchKernelGGL(hog, dim3(256), dim3(512), 150 * 1024, s0, ticks, d + 4096);
    if (time_small(ss, &t_full_hog)) return 1;
    CK(hipDeviceSynchronize());
    printf("small kernel (64 x 256 threads) on the 16-CU mask: alone %.3f ms, beside a hog on the other 240 CUs %.3f ms, beside an unmasked 256-workgroup hog %.3f ms\n",
           t_alone, t_masked_hog, t_full_hog);
    return 0;
}
