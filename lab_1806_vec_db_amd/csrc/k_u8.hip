// k_u8.hip -- VecSet<u8> (scalar.rs:117-119): rows stored at ONE byte per element in HBM.
// DistanceScalar for u8 converts every element with `as f32` and then runs the f32 folds (distance/mod.rs:79-95), so a
// u8 corpus needs no arithmetic of its own -- only loads that widen on the fly.  Search-time kernels that read the
// row-major rows (exact scan, re-rank of a shortlist) have the native-u8 variants below; build-time kernels (row norms,
// the MFMA mirrors) run on widened chunks (k_widen_u8) -- both mirrors hold a u8 value exactly (8 significant bits fit
// bf16's hi plane and fp16), so the shortlist pass of a u8 index has zero operand rounding error.
#include "common.hpp"
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace vdb {

__global__ void k_widen_u8(const uint8_t *__restrict__ in, uint64_t count, float *__restrict__ out) {
    const uint64_t i = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) * 4;
    if (i + 4 <= count) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(in + i);  // (callers keep chunk starts 4-byte aligned)
        *reinterpret_cast<float4 *>(out + i) = make_float4(float(w & 0xff), float((w >> 8) & 0xff), float((w >> 16) & 0xff), float(w >> 24));
    } else {
        for (uint64_t j = i; j < count; j++) out[j] = float(in[j]);
    }
}
void launch_widen_u8(const uint8_t *in, uint64_t count, float *out, hipStream_t s) {
    if (count == 0) return;
    hipLaunchKernelGGL(k_widen_u8, dim3((unsigned)((count / 4 + 256) / 256)), dim3(256), 0, s, in, count, out);
}

template <int FOLD>
__device__ __forceinline__ float fold_u8(float acc, float x, float q) {
    if (FOLD == 0) {  // l2: (x - q)^2, three roundings (distance/mod.rs:75-77, :86-94 after the widening)
        float df = x - q;
        float sq = df * df;
        return acc + sq;
    }
    float p = x * q;  // dot (:72-74, :79-85)
    return acc + p;
}
__device__ __forceinline__ float epilogue_u8(int metric, float acc, float xsq, float qsq) {
    if (metric == MET_L2_DIRECT) return acc;
    if (metric == MET_COSINE) {
        float den = fmaxf(sqrtf(qsq) * sqrtf(xsq), 1e-10f);
        float r = acc / den;
        return 1.0f - r;
    }
    float s = xsq + qsq;
    float t = 2.0f * acc;
    return s - t;
}

// strict-order fold of one u8 row against a f32 query (LDS or global); 16-B loads when the row allows
template <int FOLD>
__device__ __forceinline__ float row_fold_u8(const uint8_t *x, uint32_t dim, const float *q) {
    float acc = 0.0f;
    uint32_t j = 0;
    if ((dim & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const uint4 *x16 = reinterpret_cast<const uint4 *>(x);
        for (uint32_t t = 0; t < dim / 16; t++) {
            const uint4 v = x16[t];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc = fold_u8<FOLD>(acc, float((w[a] >> (8 * b)) & 0xff), q[t * 16 + a * 4 + b]);
        }
        return acc;
    }
    for (; j < dim; j++) acc = fold_u8<FOLD>(acc, float(x[j]), q[j]);
    return acc;
}

// dense exact distances out[b*ld + r] of every row to nq (<= 8) queries: one thread per row
template <int FOLD>
__global__ __launch_bounds__(256) void k_scan_exact_u8(const uint8_t *__restrict__ X, uint64_t n, uint32_t dim, const float *__restrict__ Q,
                                                       uint32_t nq, int metric, const float *__restrict__ xsq,
                                                       const float *__restrict__ qsq, float *__restrict__ out, uint64_t ld) {
    const uint64_t r = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (r >= n) return;
    const float xs = (metric == MET_L2_DIRECT) ? 0.0f : xsq[r];
    for (uint32_t b = 0; b < nq; b++) {
        const float acc = row_fold_u8<FOLD>(X + r * dim, dim, Q + size_t(b) * dim);
        const float qs = (metric == MET_L2_DIRECT) ? 0.0f : qsq[b];
        out[uint64_t(b) * ld + r] = epilogue_u8(metric, acc, xs, qs);
    }
}
void launch_scan_exact_u8(const uint8_t *X, uint64_t n, uint32_t dim, const float *Q, uint32_t nq, int metric, const float *xsq,
                          const float *qsq, float *out, uint64_t ld, hipStream_t s) {
    if (n == 0 || nq == 0) return;
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    if (metric == MET_L2_DIRECT)
        hipLaunchKernelGGL(k_scan_exact_u8<0>, g, b, 0, s, X, n, dim, Q, nq, metric, xsq, qsq, out, ld);
    else
        hipLaunchKernelGGL(k_scan_exact_u8<1>, g, b, 0, s, X, n, dim, Q, nq, metric, xsq, qsq, out, ld);
}

// exact distance of every candidate (pair keys in: only the row id is read; PAIR_NONE passes through), query in LDS
template <int FOLD>
__global__ __launch_bounds__(64) void k_rerank_u8(const uint8_t *__restrict__ X, uint32_t dim, const float *__restrict__ Q, int metric,
                                                  const float *__restrict__ xsq, const float *__restrict__ qsq,
                                                  const uint64_t *__restrict__ cand, uint64_t *__restrict__ out, uint32_t ncand,
                                                  uint32_t ldc) {
    extern __shared__ float qs_u8[];
    const uint32_t q = blockIdx.y, j = blockIdx.x * 64 + threadIdx.x;
    for (uint32_t i = threadIdx.x; i < dim; i += 64) qs_u8[i] = Q[uint64_t(q) * dim + i];
    __syncthreads();
    if (j >= ldc) return;
    uint64_t r = PAIR_NONE;
    if (j < ncand) {
        const uint64_t c = cand[uint64_t(q) * ldc + j];
        if (c != PAIR_NONE) {
            const uint32_t idx = uint32_t(c);
            const float acc = row_fold_u8<FOLD>(X + uint64_t(idx) * dim, dim, qs_u8);
            const float xs = (metric == MET_L2_DIRECT) ? 0.0f : xsq[idx];
            const float qq = (metric == MET_L2_DIRECT) ? 0.0f : qsq[q];
            r = pair_key(epilogue_u8(metric, acc, xs, qq), idx);
        }
    }
    out[uint64_t(q) * ldc + j] = r;
}
void launch_rerank_u8(const uint8_t *X, uint32_t dim, const float *Q, uint32_t nq, int metric, const float *xsq, const float *qsq,
                      const uint64_t *cand, uint64_t *out, uint32_t ncand, uint32_t ldc, hipStream_t s) {
    if (nq == 0 || ncand == 0) return;
    const dim3 g((std::max(ncand, ldc) + 63) / 64, nq), b(64);
    const size_t lds = size_t(dim) * sizeof(float);
    if (metric == MET_L2_DIRECT)
        hipLaunchKernelGGL(k_rerank_u8<0>, g, b, lds, s, X, dim, Q, metric, xsq, qsq, cand, out, ncand, ldc);
    else
        hipLaunchKernelGGL(k_rerank_u8<1>, g, b, lds, s, X, dim, Q, metric, xsq, qsq, cand, out, ncand, ldc);
}

}  // namespace vdb
