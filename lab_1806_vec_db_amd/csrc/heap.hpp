// heap.hpp -- binary heaps of pair keys in global memory, driven by ONE lane: the any-size fallbacks of the bounded
// register / LDS structures (k > 1024 result sets, HNSW candidate pools beyond the LDS pool).  The reference keeps
// these sets in BTreeSets of unbounded size (candidate_pair.rs:43-82, hnsw_index.rs:266-268); a heap gives the two
// operations the replays need -- "largest element" for ResultSet, "smallest element" for the candidate queue -- in
// O(log n) without a size limit.
#pragma once
#include "common.hpp"

namespace vdb {
#if defined(__HIPCC__)

// ---- max-heap: ResultSet with `results.iter().last()` at the root --------------------------------------------------
__device__ __forceinline__ void heap_max_push(uint64_t *h, uint32_t n /* size before */, uint64_t e) {
    uint32_t i = n;
    while (i > 0) {
        const uint32_t p = (i - 1) >> 1;
        const uint64_t hp = h[p];
        if (hp >= e) break;
        h[i] = hp;
        i = p;
    }
    h[i] = e;
}
// pop_last + insert(e) in one sift (candidate_pair.rs:67-70)
__device__ __forceinline__ void heap_max_replace_top(uint64_t *h, uint32_t n, uint64_t e) {
    uint32_t i = 0;
    for (;;) {
        const uint32_t l = 2 * i + 1;
        if (l >= n) break;
        uint32_t c = l;
        uint64_t hc = h[l];
        if (l + 1 < n) {
            const uint64_t hr = h[l + 1];
            if (hr > hc) {
                hc = hr;
                c = l + 1;
            }
        }
        if (hc <= e) break;
        h[i] = hc;
        i = c;
    }
    h[i] = e;
}
// ResultSet::add (candidate_pair.rs:61-74) on a heap of capacity k: admitted when not full, or when the DISTANCE is
// strictly smaller than the worst pair's (the index does not take part in the test)
__device__ __forceinline__ bool result_heap_add(uint64_t *h, uint32_t &n, uint32_t k, uint64_t e) {
    if (n < k) {
        heap_max_push(h, n, e);
        n++;
        return true;
    }
    if (k == 0 || uint32_t(e >> 32) >= uint32_t(h[0] >> 32)) return false;
    heap_max_replace_top(h, n, e);
    return true;
}

// ---- min-heap: the candidate queue of search_on_level_fn (pop_first, hnsw_index.rs:273) --------------------------
__device__ __forceinline__ void heap_min_push(uint64_t *h, uint32_t n /* size before */, uint64_t e) {
    uint32_t i = n;
    while (i > 0) {
        const uint32_t p = (i - 1) >> 1;
        const uint64_t hp = h[p];
        if (hp <= e) break;
        h[i] = hp;
        i = p;
    }
    h[i] = e;
}
__device__ __forceinline__ uint64_t heap_min_pop(uint64_t *h, uint32_t &n) {
    const uint64_t top = h[0];
    n--;
    if (n == 0) return top;
    const uint64_t e = h[n];
    uint32_t i = 0;
    for (;;) {
        const uint32_t l = 2 * i + 1;
        if (l >= n) break;
        uint32_t c = l;
        uint64_t hc = h[l];
        if (l + 1 < n) {
            const uint64_t hr = h[l + 1];
            if (hr < hc) {
                hc = hr;
                c = l + 1;
            }
        }
        if (hc >= e) break;
        h[i] = hc;
        i = c;
    }
    h[i] = e;
    return top;
}

#endif
}  // namespace vdb
