// k_redo.hip -- the small kernels around a Flat call's second attempts (Index::flat_knn_finish): gather the queries a tier could not
// certify, scatter the next tier's answers back, and -- for the 8-bit pass -- turn the k-th exact distance a query already has into the
// threshold of a SECOND 8-bit pass that is certified by construction.
//
// Why a second 8-bit pass (round 4).  The first pass sizes its hit list from a row sample (~1000 expected hits) and its exact stage walks
// at most flat_i8_kprime rows.  Where the k-th neighbour sits in a tight cluster -- hundreds or thousands of rows within the bound's gap of
// each other -- neither is enough, and before this tier such a query fell through the fp16 pass and the split-bf16 pass (whose error bars
// are just as blind to margins of 1e-3) down to the exact scan at 8 queries per corpus pass: 1M rows in 1024 tight clusters took 166 ms
// per 1000 queries where separable data take 1.0.  But the failed walk leaves something behind: D_k', the k-th smallest EXACT distance among
// the rows it did evaluate -- an upper bound of the true k-th distance.  Every row that can still matter has key <= tau' with
//     tau' = the smallest threshold for which flat_certify_lb(D_k', kappa = tau') holds,
// so the second pass runs the filter with tau' instead of a sampled threshold (no sample, no select) and lets the exact stage walk the whole
// list: when the list is exhausted kappa = tau' certifies, because the k-th distance can only have gone down.  What does not fit the
// candidate list (more than 8192 rows within the gap) is flagged as before and goes on to the fp16 tier.
#include "common.hpp"
#include "kernels.hpp"
#include "half_rows.hpp"

namespace vdb {

__global__ void k_gather_rows_f32(const float *__restrict__ src, const uint64_t *__restrict__ rows, uint32_t width, float *__restrict__ dst) {
    const uint64_t r = rows[blockIdx.x];
    for (uint32_t j = threadIdx.x; j < width; j += blockDim.x) dst[uint64_t(blockIdx.x) * width + j] = src[r * width + j];
}
void launch_gather_rows_f32(const float *src, const uint64_t *rows, uint64_t nr, uint32_t width, float *dst, hipStream_t s) {
    if (nr == 0 || width == 0) return;
    hipLaunchKernelGGL(k_gather_rows_f32, dim3((unsigned)nr), dim3(width >= 256 ? 256 : 64), 0, s, src, rows, width, dst);
}

// answers of the redone queries back into the call's outputs: out[rows[j]] = redo[j]
__global__ void k_scatter_results(const uint64_t *__restrict__ ri, const float *__restrict__ rd, const uint64_t *__restrict__ rc,
                                  const uint64_t *__restrict__ rows, uint32_t k, uint64_t *__restrict__ o_idx, float *__restrict__ o_dist,
                                  uint64_t *__restrict__ o_cnt) {
    const uint64_t r = rows[blockIdx.x];
    for (uint32_t j = threadIdx.x; j < k; j += blockDim.x) {
        o_idx[r * k + j] = ri[uint64_t(blockIdx.x) * k + j];
        o_dist[r * k + j] = rd[uint64_t(blockIdx.x) * k + j];
    }
    if (threadIdx.x == 0) o_cnt[r] = rc[blockIdx.x];
}
void launch_scatter_results(const uint64_t *ri, const float *rd, const uint64_t *rc, const uint64_t *rows, uint64_t nr, uint32_t k,
                            uint64_t *o_idx, float *o_dist, uint64_t *o_cnt, hipStream_t s) {
    if (nr == 0) return;
    hipLaunchKernelGGL(k_scatter_results, dim3((unsigned)nr), dim3(64), 0, s, ri, rd, rc, rows, k, o_idx, o_dist, o_cnt);
}

// dk[j] = the ksel-th exact distance query rows[j] already has (+inf when its first walk produced fewer than ksel rows: such a query
// overflows the second pass's list on purpose and goes to the next tier)
__global__ void k_gather_dk(const float *__restrict__ o_dist, const uint64_t *__restrict__ o_cnt, const uint64_t *__restrict__ rows, uint64_t nr,
                            uint32_t k, uint32_t ksel, float *__restrict__ dk) {
    const uint64_t j = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
    if (j >= nr) return;
    const uint64_t r = rows[j];
    dk[j] = o_cnt[r] >= ksel ? o_dist[r * k + (ksel - 1)] : INFINITY;
}
void launch_gather_dk(const float *o_dist, const uint64_t *o_cnt, const uint64_t *rows, uint64_t nr, uint32_t k, uint32_t ksel, float *dk,
                      hipStream_t s) {
    if (nr == 0) return;
    hipLaunchKernelGGL(k_gather_dk, dim3((unsigned)((nr + 63) / 64)), dim3(64), 0, s, o_dist, o_cnt, rows, nr, k, ksel, dk);
}

// tau[q] for the second 8-bit pass: the inverse of flat_certify_lb (k_exact.hip) at D_k = dk[q], a little above it so that the
// certification's own roundings cannot undo it (too small a tau costs a query its certification, never its correctness: the exact stage
// decides).  Padding queries (q >= nq) get -inf.  A NaN offset (query the pass cannot describe) or an infinite dk gives +inf / NaN: the
// query overflows or passes nothing, is flagged and goes on.
__global__ void k_i8_tau_from_dk(const float *__restrict__ dk, uint32_t nq, uint32_t nq_pad, const float *__restrict__ qoff,
                                 const float *__restrict__ qsq, float xsq_max, float mu_norm, uint32_t dim, int cosine,
                                 float *__restrict__ tau) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq_pad) return;
    if (q >= nq) {
        tau[q] = -INFINITY;
        return;
    }
    const float u = 5.9604645e-8f, d = dk[q];
    float L;  // the value kappa + O_q has to exceed
    if (cosine) {
        const float nr = 2.0f + 2.0f * mu_norm;
        L = 2.0f * (d + float(2 * dim + 16) * u * 1.01f + 2.0f * u * nr * nr) / (1.0f - 4.0f * u);
    } else {
        const float qn = sqrtf(qsq[q]);
        const float rx = fminf(sqrtf(xsq_max), (qn + sqrtf(fmaxf(d, 0.0f))) * 1.001f);
        const float nr = rx + qn + 2.0f * mu_norm;
        L = (d + 4.0f * u * nr * nr) / (1.0f - float(dim + 8) * u * 1.01f);
    }
    L = L + fabsf(L) * 1e-6f + 1e-30f;
    tau[q] = L - qoff[q] + fabsf(qoff[q]) * 2.0f * u;
}
void launch_i8_tau_from_dk(const float *dk, uint32_t nq, uint32_t nq_pad, const float *qoff, const float *qsq, float xsq_max, float mu_norm,
                           uint32_t dim, int cosine, float *tau, hipStream_t s) {
    if (nq_pad == 0) return;
    hipLaunchKernelGGL(k_i8_tau_from_dk, dim3((nq_pad + 63) / 64), dim3(64), 0, s, dk, nq, nq_pad, qoff, qsq, xsq_max, mu_norm, dim, cosine,
                       tau);
}


// ---- tighter keys for the hits of the 8-bit pass from the row-major fp16 image (round 4) --------------------------------------------
// The 8-bit keys are worst-case bounds of a DOT product: their slack is ~2 |dx||q|, and against a tight cluster -- a thousand rows whose
// distances to the query are a hundredth of |x|^2 and differ by a few per cent -- that is wider than the spread: the exact stage has to
// evaluate hundreds of rows in key order before the k-th exact distance drops below the next key (1M rows in 1024 clusters of 0.15
// sigma: 8+ rounds of 63 rows for 98 % of the queries, 2.5 ms per step against 1.0 on separable data).  Any bound built on x.q has that
// problem, the fp16 tier's (half_approx: a -/+ E with E ~ 2 |dx||q| + gamma_d (|x| + |q|)^2) included -- measured: refining with it
// changed nothing.  What survives the cancellation is the DIFFERENCE form on the image row x~ = h / sx:
//     A = |x~ - q|^2,   sqrt(D) = |x - q| >= |x~ - q| - |x - x~| = sqrt(A) - |dx_r|,   |dx_r| <= min(dx_abs, dx_rel |x_r|) (measured),
// whose slack is 2 |dx_r| sqrt(D): relative to D it is 2 |dx_r| / sqrt(D), ~0.6 % where the dot form has ~15 %.  half_diffs32 returns
// a = fl(A) within (d + 4) u relative (all terms non-negative), so  D >= lb := max(0, sqrt(a (1 - (d + 4) u)) - |dx_r|)^2  and the key
// becomes the larger of the 8-bit key and lb - O_q (the keys' terms: D >= key + O_q); the walk (k_flat_tail_lb) and its certification run
// unchanged on the tightened list.  The f32 evaluation of lb rounds a handful of times: every step is pushed down by 4 u relative.
// Keys that are not finite keep the 8-bit value.  A wave per 64 hits (32 rows at a time).  Cosine: the keys bound the distance of the
// UNIT vectors, 2 (1 - cos) = |x^ - q^|^2 >= key + O_q, and the same inequality holds for them with the image row scaled by 1 / |x_r|
// (cached norm) on the fly, the unit query in LDS and the slack |dx_r| / |x_r| + (d + 16) u (the norms' own rounding); rows and queries
// whose cached norms are not plain numbers keep their keys.  Costs half the f32 bytes of EVERY hit, so it only runs where the walks are
// long (Index::flat_i8_refine: auto by the rounds walked).
template <bool COS>
__global__ __launch_bounds__(64) void k_flat_refine_half(const uint16_t *__restrict__ rows_h, uint32_t dim, float sx, float dx_abs, float dx_rel,
                                                         const float *__restrict__ Q, const float *__restrict__ xsq, const float *__restrict__ qsq,
                                                         const float *__restrict__ qoff, uint64_t *__restrict__ cand, uint32_t cap,
                                                         const uint32_t *__restrict__ cnt) {
    extern __shared__ __attribute__((aligned(16))) float rf_q[];  // [dim]: q * sx (Cosine: q / |q|)
    const uint32_t q = blockIdx.y, lane = threadIdx.x;
    const uint32_t total = cnt[q];
    if (total > cap || blockIdx.x * 64 >= total) return;  // (an overflowed list is redone anyway)
    const float qs = COS ? qsq[q] : 1.0f;
    if (COS && !(qs >= 1e-30f && qs <= 1e30f)) return;  // (block-uniform: such a query keeps its 8-bit keys)
    const float qmul = COS ? 1.0f / sqrtf(qs) : sx;      // (sx is a power of two: exact)
    for (uint32_t i = lane; i < dim; i += 64) rf_q[i] = Q[uint64_t(q) * dim + i] * qmul;
    __syncthreads();
    constexpr float u = 0x1p-24f;
    const float oq = qoff[q], inv_sx = 1.0f / sx;
#pragma unroll 1
    for (uint32_t h = 0; h < 2; h++) {  // 32 rows per call, one per lane 0..31
        const uint32_t j = blockIdx.x * 64 + h * 32 + (lane & 31);
        const bool live = lane < 32 && j < total;
        const uint64_t c = live ? cand[uint64_t(q) * cap + j] : PAIR_NONE;
        const uint32_t nb = live ? uint32_t(c) : 0u;
        if (__ballot(live) == 0) continue;  // wave-uniform
        const float a = half_diffs32<COS>(rows_h, dim, inv_sx, rf_q, nb, live, lane, xsq);
        if (live) {
            const float xs = xsq[nb];
            const float nx = sqrtf(xs) * 1.001f;  // (cached strict fold: within gamma_d of |x|^2)
            float slack = fminf(dx_abs, dx_rel * nx) * 1.001f;  // |x - x~|
            bool okrow = true;
            if (COS) {
                // unit vectors: |x^ - x^~| <= |dx| / |x| + the relative error of the cached norm and of 1 / sqrt; |q^ - q^c| likewise; the
                // per-element roundings of h * rs and q * qmul: together below (d + 16) u
                okrow = xs >= 1e-30f && xs <= 1e30f;
                slack = slack / (sqrtf(xs) * 0.999f) * 1.002f + float(dim + 16) * u;
            }
            float sr = sqrtf(a * (1.0f - float(dim + 8) * u)) * (1.0f - 4.0f * u) - slack;
            sr = sr > 0.0f ? sr * (1.0f - 4.0f * u) : 0.0f;
            const float lb = sr * sr * (1.0f - 4.0f * u);  // <= |x - q|^2 (Cosine: <= |x^ - q^|^2 = 2 (1 - cos))
            const float kn = (lb - oq) - 4.0f * u * (fabsf(lb) + fabsf(oq));
            const float ko = f32_from_orderable(uint32_t(c >> 32));
            const bool fin = okrow && a - a == 0.0f && slack - slack == 0.0f && kn - kn == 0.0f;  // (NaN / inf anywhere: the 8-bit key stays)
            if (fin && kn > ko) cand[uint64_t(q) * cap + j] = pair_key(kn, nb);
        }
    }
}
void launch_flat_refine_half(const uint16_t *rows_h, uint32_t dim, float sx, float dx_abs, float dx_rel, int cosine, const float *Q, const float *xsq,
                             const float *qsq, const float *qoff, uint64_t *cand, uint32_t cap, const uint32_t *cnt, uint32_t nq, uint32_t max_hits,
                             hipStream_t s) {
    if (nq == 0 || max_hits == 0) return;
    if (cosine)
        hipLaunchKernelGGL(k_flat_refine_half<true>, dim3((max_hits + 63) / 64, nq), dim3(64), dim * sizeof(float), s, rows_h, dim, sx, dx_abs, dx_rel, Q,
                           xsq, qsq, qoff, cand, cap, cnt);
    else
        hipLaunchKernelGGL(k_flat_refine_half<false>, dim3((max_hits + 63) / 64, nq), dim3(64), dim * sizeof(float), s, rows_h, dim, sx, dx_abs, dx_rel, Q,
                           xsq, qsq, qoff, cand, cap, cnt);
    VDB_HIP(hipGetLastError());
}

}  // namespace vdb
