// k_sort.hip -- FlatIndex::knn for k > 1024: full sort of the pair keys of one query (rocPRIM radix sort through
// hipCUB).  The wave-level select of k_topk.hip keeps at most 1024 pairs in registers; beyond that the whole
// (distance, index) order is materialised.  Exact-scan distances only, so the output is the reference's order.
#include <hipcub/hipcub.hpp>

#include "common.hpp"
#include "heap.hpp"
#include "kernels.hpp"

namespace vdb {

__global__ void k_make_pair_keys(const float *__restrict__ dist, uint64_t n, uint64_t *__restrict__ keys) {
    uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = pair_key(dist[i], uint32_t(i));
}

size_t sort_pairs_temp_bytes(uint64_t n) {
    size_t bytes = 0;
    uint64_t *p = nullptr;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, p, p, (int)n);
    return bytes;
}

// dist[0..n) -> sorted pair keys out[0..n); tmp_keys holds n u64, temp is sort_pairs_temp_bytes(n)
void launch_sort_pairs(const float *dist, uint64_t n, uint64_t *tmp_keys, uint64_t *out, void *temp, size_t temp_bytes,
                       hipStream_t s) {
    if (n == 0) return;
    VDB_REQUIRE(n < (1ull << 31), "sort path: too many rows");
    hipLaunchKernelGGL(k_make_pair_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dist, n, tmp_keys);
    VDB_HIP(hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, tmp_keys, out, (int)n, 0, 64, s));
}

// ---- any-size fallbacks of the register-resident selects (k, ef, n_probes > 1024) -------------------------------------
// rows of pair keys [nq][ld] -> every row sorted ascending (PAIR_NONE pads sort last): one segmented radix sort
size_t sort_rows_temp_bytes(uint64_t nq, uint64_t ld) {
    size_t bytes = 0;
    uint64_t *p = nullptr;
    int *o = nullptr;
    (void)hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, bytes, p, p, (int)(nq * ld), (int)nq, o, o + 1);
    return bytes + (nq + 1) * sizeof(int) + 256;
}
__global__ void k_row_offsets(int *off, uint32_t nq, uint32_t ld) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= nq) off[i] = (int)(i * ld);
}
void launch_sort_rows(const uint64_t *in, uint64_t *out, uint64_t nq, uint64_t ld, void *temp, size_t temp_bytes, hipStream_t s) {
    if (nq == 0 || ld == 0) return;
    VDB_REQUIRE(nq * ld < (1ull << 31), "sort path: too many keys in one call");
    int *off = static_cast<int *>(temp);  // offsets first, the sort's scratch behind them
    const size_t off_bytes = ((nq + 1) * sizeof(int) + 255) & ~size_t(255);
    hipLaunchKernelGGL(k_row_offsets, dim3((unsigned)((nq + 256) / 256)), dim3(256), 0, s, off, (uint32_t)nq, (uint32_t)ld);
    size_t tb = temp_bytes - off_bytes;
    VDB_HIP(hipcub::DeviceSegmentedRadixSort::SortKeys(static_cast<char *>(temp) + off_bytes, tb, in, out, (int)(nq * ld), (int)nq,
                                                        off, off + 1, 0, 64, s));
}

// dense f32 values [nq][ldd] -> pair keys (value, column) [nq][ldk]; columns >= n become PAIR_NONE
__global__ void k_pair_keys_rows(const float *__restrict__ dist, uint64_t ldd, uint64_t n, uint64_t *__restrict__ keys, uint64_t ldk) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, q = blockIdx.y;
    if (i < ldk) keys[q * ldk + i] = i < n ? pair_key(dist[q * ldd + i], uint32_t(i)) : PAIR_NONE;
}
void launch_pair_keys_rows(const float *dist, uint64_t ldd, uint64_t n, uint32_t nq, uint64_t *keys, uint64_t ldk, hipStream_t s) {
    if (nq == 0 || ldk == 0) return;
    hipLaunchKernelGGL(k_pair_keys_rows, dim3((unsigned)((ldk + 255) / 256), nq), dim3(256), 0, s, dist, ldd, n, keys, ldk);
}
// out[q][0..ld_out) = in[q][0..count) then PAIR_NONE
__global__ void k_copy_prefix(const uint64_t *__restrict__ in, uint64_t ld_in, uint64_t *__restrict__ out, uint64_t ld_out, uint64_t count) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, q = blockIdx.y;
    if (i < ld_out) out[q * ld_out + i] = i < count ? in[q * ld_in + i] : PAIR_NONE;
}
void launch_copy_prefix(const uint64_t *in, uint64_t ld_in, uint64_t *out, uint64_t ld_out, uint64_t count, uint32_t nq, hipStream_t s) {
    if (nq == 0 || ld_out == 0) return;
    hipLaunchKernelGGL(k_copy_prefix, dim3((unsigned)((ld_out + 255) / 256), nq), dim3(256), 0, s, in, ld_in, out, ld_out, count);
}

// ResultSet::add replayed over `ncand` offers per query in the given order, any k (candidate_pair.rs:61-74,102-108):
// the set is a max-heap in global memory driven by lane 0; out[q][0..k) = the set, unsorted, PAIR_NONE padded
__global__ __launch_bounds__(64) void k_resort_big(const uint64_t *__restrict__ offers, uint32_t ncand, uint32_t ldc, uint32_t k,
                                                   uint64_t *__restrict__ out, uint32_t ldo) {
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    uint64_t *h = out + uint64_t(q) * ldo;
    for (uint32_t j = lane; j < ldo; j += 64) h[j] = PAIR_NONE;
    __syncthreads();
    if (lane != 0) return;
    uint32_t n = 0;
    const uint64_t *src = offers + uint64_t(q) * ldc;
    for (uint32_t j = 0; j < ncand; j++) {
        const uint64_t e = src[j];
        if (e != PAIR_NONE) (void)result_heap_add(h, n, k, e);
    }
}
void launch_resort_big(const uint64_t *offers, uint32_t ncand, uint32_t ldc, uint32_t nq, uint32_t k, uint64_t *out, uint32_t ldo,
                       hipStream_t s) {
    if (nq == 0) return;
    hipLaunchKernelGGL(k_resort_big, dim3(nq), dim3(64), 0, s, offers, ncand, ldc, k, out, ldo);
}

}  // namespace vdb
