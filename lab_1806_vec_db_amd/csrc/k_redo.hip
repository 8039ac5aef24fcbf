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

}  // namespace vdb
