// tools/fold_chain_probe.cpp -- what ONE wave pays per element of a strict-order dot-product fold (distance/mod.rs:72-74: acc = acc + x*q,
// product and sum separately rounded), in shader cycles (s_memtime) and ns: (a) a pure chain of dependent v_add_f32, (b) mul and add
// alternating as hipcc emits the walk's fold (every add waits for the mul in front of it AND the add before), (c) the products of a
// 4-vector formed first (2 x v_pk_mul_f32), then 4 dependent adds, (d) products one chunk ahead of their adds.  Operands from LDS-like
// registers; nothing else runs.   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o /tmp/fold_chain_probe tools/fold_chain_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
#pragma clang fp contract(off)
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t iters, const v4f *src, float *out, unsigned long long *cyc) {
    v4f x[8], q[8];
    for (int c = 0; c < 8; c++) { x[c] = src[c * 64 + threadIdx.x]; q[c] = src[(8 + c) * 64 + threadIdx.x]; }
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (uint32_t it = 0; it < iters; it++) {
        if (MODE == 3) {
            v4f pr[8];
#pragma unroll
            for (int c = 0; c < 8; c++) pr[c] = x[c] * q[c];
#pragma unroll
            for (int c = 0; c < 8; c++) { acc = acc + pr[c].x; acc = acc + pr[c].y; acc = acc + pr[c].z; acc = acc + pr[c].w; asm volatile("" : "+v"(acc)); }
        } else {
#pragma unroll
        for (int c = 0; c < 8; c++) {
            if (MODE == 0) {
                acc = acc + x[c].x; asm volatile("" : "+v"(acc)); acc = acc + x[c].y; asm volatile("" : "+v"(acc));
                acc = acc + x[c].z; asm volatile("" : "+v"(acc)); acc = acc + x[c].w; asm volatile("" : "+v"(acc));
            } else if (MODE == 1) {
                float p;
                p = x[c].x * q[c].x; asm volatile("" : "+v"(p)); acc = acc + p; asm volatile("" : "+v"(acc));
                p = x[c].y * q[c].y; asm volatile("" : "+v"(p)); acc = acc + p; asm volatile("" : "+v"(acc));
                p = x[c].z * q[c].z; asm volatile("" : "+v"(p)); acc = acc + p; asm volatile("" : "+v"(acc));
                p = x[c].w * q[c].w; asm volatile("" : "+v"(p)); acc = acc + p; asm volatile("" : "+v"(acc));
            } else {
                const v4f pr = x[c] * q[c];
                acc = acc + pr.x; acc = acc + pr.y; acc = acc + pr.z; acc = acc + pr.w;
                asm volatile("" : "+v"(acc));
            }
        }
        }
        x[it & 7].x += acc * 1e-30f;  // (keeps the loop body from being hoisted)
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = acc; cyc[0] = t1 - t0; }
}
int main() {
    v4f *src; float *out; unsigned long long *cyc;
    hipMalloc(&src, 16 * 64 * 16); hipMalloc(&out, 64); hipMalloc(&cyc, 64);
    float h[16 * 64 * 4]; for (int i = 0; i < 16 * 64 * 4; i++) h[i] = 1.0f + 1e-3f * (i % 97);
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    const uint32_t iters = 200000;  // x 32 elements
    const char *names[4] = {"(a) dependent adds only", "(b) mul, add alternating", "(c) 2 x pk_mul then 4 adds per 4-vector", "(d) 16 x pk_mul (a whole line) then 32 adds"};
    for (int mode = 0; mode < 4; mode++) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, iters, src, out, cyc);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, iters, src, out, cyc);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, iters, src, out, cyc);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, iters, src, out, cyc);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-48s %.2f cycles (s_memtime) and %.2f ns per element -> a 960-element row: %.2f us\n", names[mode], double(c) / (iters * 32.0), ms * 1e6 / (iters * 32.0), ms * 1e3 / (iters * 32.0) * 960);
    }
    return 0;
}
