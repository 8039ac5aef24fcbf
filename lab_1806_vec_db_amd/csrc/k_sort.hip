// k_sort.hip -- FlatIndex::knn for k > 1024: full sort of the pair keys of one query (rocPRIM radix sort through
// hipCUB).  The wave-level select of k_topk.hip keeps at most 1024 pairs in registers; beyond that the whole
// (distance, index) order is materialised.  Exact-scan distances only, so the output is the reference's order.
#include <hipcub/hipcub.hpp>

#include "common.hpp"
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

}  // namespace vdb
