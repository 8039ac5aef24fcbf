// k_sort.hip -- the any-size paths: full (distance, index) order of one query's rows (FlatIndex::knn for k > 1024) and
// row-wise sorts of pair keys (ef, k, n_probes > 1024; the heap walk's result sets).  The wave-level select of k_topk.hip
// keeps at most 1024 pairs in registers; beyond that the order is materialised by a segmented LSD radix sort of the u64
// pair keys: 8 passes of 8 bits, per pass a per-block digit histogram, one scan per segment, a stable scatter.  These
// paths are off the hot path (the reference's own tests reach them through `search(k = len, ef = len)`); the sort is
// written for simplicity -- three small kernels per pass, no look-back, no digit skipping.
#include <algorithm>

#include "common.hpp"
#include "heap.hpp"
#include "kernels.hpp"

namespace vdb {

__global__ void k_make_pair_keys(const float *__restrict__ dist, uint64_t n, uint64_t *__restrict__ keys) {
    uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = pair_key(dist[i], uint32_t(i));
}

// ---- segmented LSD radix sort of u64 keys: nq segments of ld keys each (row stride ld) ------------------------------
constexpr uint32_t RS_WAVE_KEYS = 1024;             // contiguous keys a wave ranks in order (16 steps of 64)
constexpr uint32_t RS_CH = 4 * RS_WAVE_KEYS;        // keys per 256-thread block and pass
static uint32_t rs_blocks(uint64_t ld) { return (uint32_t)((ld + RS_CH - 1) / RS_CH); }

// hist[(seg * 256 + bin) * nblk + blk] = keys of block blk of segment seg whose digit is bin
__global__ __launch_bounds__(256) void k_rs_hist(const uint64_t *__restrict__ in, uint64_t ld, uint32_t nblk, uint32_t shift,
                                                 uint32_t *__restrict__ hist) {
    __shared__ uint32_t hs[256];
    const uint32_t seg = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    hs[tid] = 0;
    __syncthreads();
    const uint64_t base = uint64_t(blk) * RS_CH;
    const uint64_t *row = in + uint64_t(seg) * ld;
    for (uint32_t i = tid; i < RS_CH; i += 256)
        if (base + i < ld) atomicAdd(&hs[(row[base + i] >> shift) & 255u], 1u);
    __syncthreads();
    hist[(uint64_t(seg) * 256 + tid) * nblk + blk] = hs[tid];
}
// per segment: exclusive prefix of its histogram in (bin, block) order = first output position of every (bin, block)
__global__ __launch_bounds__(256) void k_rs_scan(uint32_t *__restrict__ hist, uint32_t nblk) {
    __shared__ uint32_t sc[256];
    const uint32_t tid = threadIdx.x;
    uint32_t *h = hist + (uint64_t(blockIdx.x) * 256 + tid) * nblk;
    uint32_t sum = 0;
    for (uint32_t b = 0; b < nblk; b++) sum += h[b];
    sc[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {  // inclusive scan of the bin totals
        const uint32_t v = tid >= off ? sc[tid - off] : 0u;
        __syncthreads();
        sc[tid] += v;
        __syncthreads();
    }
    uint32_t run = sc[tid] - sum;
    for (uint32_t b = 0; b < nblk; b++) {
        const uint32_t v = h[b];
        h[b] = run;
        run += v;
    }
}
// stable scatter: a wave owns 1024 contiguous keys and places them 64 at a time, in order; the position of a key is
// (first position of its (bin, block)) + (keys of that bin in earlier waves of the block, in earlier steps of this wave,
// in lower lanes of this step)
__global__ __launch_bounds__(256) void k_rs_scatter(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, uint64_t ld,
                                                    uint32_t nblk, uint32_t shift, const uint32_t *__restrict__ hist) {
    __shared__ uint32_t wh[4][256];
    const uint32_t seg = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (uint32_t i = tid; i < 4 * 256; i += 256) (&wh[0][0])[i] = 0;
    __syncthreads();
    const uint64_t wbase = uint64_t(blk) * RS_CH + uint64_t(wave) * RS_WAVE_KEYS;
    const uint64_t *row = in + uint64_t(seg) * ld;
    uint64_t *orow = out + uint64_t(seg) * ld;
    uint64_t keys[16];
#pragma unroll
    for (int st = 0; st < 16; st++) {
        const uint64_t idx = wbase + st * 64 + lane;
        keys[st] = idx < ld ? row[idx] : 0ull;
        if (idx < ld) atomicAdd(&wh[wave][(keys[st] >> shift) & 255u], 1u);
    }
    __syncthreads();
    {  // counts -> running bases
        uint32_t run = hist[(uint64_t(seg) * 256 + tid) * nblk + blk];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const uint32_t c = wh[w][tid];
            wh[w][tid] = run;
            run += c;
        }
    }
    __syncthreads();
    const uint64_t lt = (1ull << lane) - 1;
#pragma unroll
    for (int st = 0; st < 16; st++) {
        const uint64_t idx = wbase + st * 64 + lane;
        const bool valid = idx < ld;
        const uint32_t d = uint32_t(keys[st] >> shift) & 255u;
        uint64_t m = __ballot(valid);  // lanes of this step with the same digit
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const bool one = (d >> bit) & 1u;
            const uint64_t bm = __ballot(one);
            m &= one ? bm : ~bm;
        }
        const uint32_t rank = (uint32_t)__builtin_popcountll(m & lt), cnt = (uint32_t)__builtin_popcountll(m);
        uint32_t pos = 0;
        if (valid) pos = wh[wave][d] + rank;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank + 1 == cnt) wh[wave][d] += cnt;  // one lane per digit moves the base on
        __builtin_amdgcn_wave_barrier();
        if (valid) orow[pos] = keys[st];
    }
}
static size_t rs_temp_bytes(uint64_t nq, uint64_t ld) {
    return ((nq * ld * sizeof(uint64_t) + 255) & ~size_t(255)) + nq * 256 * size_t(rs_blocks(ld)) * sizeof(uint32_t) + 256;
}
// in [nq][ld] -> out [nq][ld], every row ascending; `in` is left intact; temp >= rs_temp_bytes(nq, ld)
static void rs_sort_rows(const uint64_t *in, uint64_t *out, uint64_t nq, uint64_t ld, void *temp, hipStream_t s) {
    if (nq == 0 || ld == 0) return;
    VDB_REQUIRE(ld < (1ull << 32), "sort path: segment too long");
    const uint32_t nblk = rs_blocks(ld);
    uint64_t *pong = static_cast<uint64_t *>(temp);
    uint32_t *hist = reinterpret_cast<uint32_t *>(static_cast<char *>(temp) + ((nq * ld * sizeof(uint64_t) + 255) & ~size_t(255)));
    constexpr uint64_t SEG_PER_LAUNCH = 32768;  // grid.y
    for (uint64_t c0 = 0; c0 < nq; c0 += SEG_PER_LAUNCH) {
        const uint64_t ns = std::min<uint64_t>(SEG_PER_LAUNCH, nq - c0);
        const uint64_t *cin = in + c0 * ld;
        uint64_t *cout = out + c0 * ld, *cpong = pong + c0 * ld;
        uint32_t *chist = hist + c0 * 256 * nblk;
        const dim3 grid(nblk, (unsigned)ns);
        for (uint32_t pass = 0; pass < 8; pass++) {  // in -> pong -> out -> pong -> ... -> out
            const uint64_t *src = pass == 0 ? cin : (pass & 1 ? cpong : cout);
            uint64_t *dst = pass & 1 ? cout : cpong;
            hipLaunchKernelGGL(k_rs_hist, grid, dim3(256), 0, s, src, ld, nblk, 8 * pass, chist);
            hipLaunchKernelGGL(k_rs_scan, dim3((unsigned)ns), dim3(256), 0, s, chist, nblk);
            hipLaunchKernelGGL(k_rs_scatter, grid, dim3(256), 0, s, src, dst, ld, nblk, 8 * pass, chist);
        }
    }
}

size_t sort_pairs_temp_bytes(uint64_t n) { return rs_temp_bytes(1, n); }

// dist[0..n) -> sorted pair keys out[0..n); tmp_keys holds n u64, temp is sort_pairs_temp_bytes(n)
void launch_sort_pairs(const float *dist, uint64_t n, uint64_t *tmp_keys, uint64_t *out, void *temp, size_t temp_bytes,
                       hipStream_t s) {
    if (n == 0) return;
    VDB_REQUIRE(temp_bytes >= rs_temp_bytes(1, n), "sort path: scratch too small");
    hipLaunchKernelGGL(k_make_pair_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dist, n, tmp_keys);
    rs_sort_rows(tmp_keys, out, 1, n, temp, s);
}

// ---- any-size fallbacks of the register-resident selects (k, ef, n_probes > 1024) -------------------------------------
// rows of pair keys [nq][ld] -> every row sorted ascending (PAIR_NONE pads sort last)
size_t sort_rows_temp_bytes(uint64_t nq, uint64_t ld) { return rs_temp_bytes(nq, ld); }
void launch_sort_rows(const uint64_t *in, uint64_t *out, uint64_t nq, uint64_t ld, void *temp, size_t temp_bytes, hipStream_t s) {
    if (nq == 0 || ld == 0) return;
    VDB_REQUIRE(temp_bytes >= rs_temp_bytes(nq, ld), "sort path: scratch too small");
    rs_sort_rows(in, out, nq, ld, temp, s);
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
