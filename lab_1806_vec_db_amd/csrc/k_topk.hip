// k_topk.hip -- wavefront-level top-k select under the CandidatePair total order.
//
// ResultSet (candidate_pair.rs:43-82) keeps the k smallest pairs by (distance, index).  For an
// ascending-index scan (FlatIndex::knn, flat_index.rs:48-57; the ADC scan of knn_pq :98-101) the
// BTreeSet's strict-`<` replacement rule (candidate_pair.rs:66-71) yields exactly the k smallest
// under the lexicographic order, so selection here is order-free: each pair becomes one u64
// (common.hpp pair_key) and unsigned comparison is the reference's Ord.
//
// One 64-lane wave owns a sorted list of 64*R keys spread over R registers per lane
// (position p lives in register p/64, lane p%64).  A candidate is admitted only when it is
// smaller than the current k-th element (tau), found with a ballot; insertion is a lane shift
// (DPP/ds_bpermute via __shfl_up) per register.  With k << chunk the admit rate is
// ~k*ln(chunk/k)/chunk, so the stream side (one float4 per lane per step) dominates.
#include "common.hpp"
#include "kernels.hpp"

namespace vdb {

uint32_t topk_chunk(uint64_t) { return 8192u; }  // finer chunks cost more fill-phase insertions than they gain in parallelism (measured)
uint32_t topk_num_lists(uint64_t n) { return n == 0 ? 1u : (uint32_t)((n + topk_chunk(n) - 1) / topk_chunk(n)); }
uint32_t topk_capacity(uint32_t k) {
    uint32_t r = 1;
    while (64u * r < k) r *= 2;
    return 64u * r;
}

template <int R>
struct WaveList {
    uint64_t v[R];
    uint64_t tau;  // element at position k-1 (wave-uniform)
    uint32_t k;

    __device__ void init(uint32_t kk) {
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = PAIR_NONE;
        tau = PAIR_NONE;
        k = kk;
    }
    // insert a wave-uniform key e (e < tau guaranteed by the caller)
    __device__ void insert(uint64_t e) {
        const uint32_t lane = threadIdx.x & 63;
        bool placed = false;  // once placed, e is the carry pushed out of the previous register
#pragma unroll
        for (int r = 0; r < R; r++) {
            uint64_t cur = v[r];
            uint64_t mask = placed ? ~0ull : __ballot(cur > e);
            if (mask != 0) {
                uint32_t pos = placed ? 0u : (uint32_t)__builtin_ctzll(mask);
                uint64_t carry = __shfl(cur, 63);
                uint64_t up = __shfl_up(cur, 1);
                v[r] = lane < pos ? cur : (lane == pos ? e : up);
                e = carry;
                placed = true;
            }
        }
        uint32_t p = k - 1;
        uint64_t t = PAIR_NONE;
#pragma unroll
        for (int r = 0; r < R; r++)
            if ((p >> 6) == (uint32_t)r) t = __shfl(v[r], p & 63);
        tau = t;
    }
    // offer one key per lane (PAIR_NONE = nothing)
    __device__ void offer(uint64_t c) {
        uint64_t m = __ballot(c < tau);
        while (m) {
            uint32_t src = (uint32_t)__builtin_ctzll(m);
            uint64_t e = __shfl(c, src);
            if (e < tau) insert(e);  // tau may have tightened since the ballot
            m &= m - 1;
        }
    }
    __device__ void store(uint64_t *dst) const {  // dst: 64*R entries
        const uint32_t lane = threadIdx.x & 63;
#pragma unroll
        for (int r = 0; r < R; r++) dst[r * 64 + lane] = v[r];
    }
};

// level 1: grid = (ceil(nlists/4), nq), block = 256 (4 waves, one list each)
template <int R>
__global__ __launch_bounds__(256) void k_topk_dense(const float *__restrict__ keys, uint64_t ld, uint64_t n,
                                                    uint32_t k, uint32_t nlists, uint32_t chunk,
                                                    uint64_t *__restrict__ lists) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t list = blockIdx.x * 4 + wave;
    const uint32_t q = blockIdx.y;
    if (list >= nlists) return;
    WaveList<R> wl;
    wl.init(k);
    const float *kq = keys + uint64_t(q) * ld;
    uint64_t begin = uint64_t(list) * chunk;
    uint64_t end = begin + chunk < n ? begin + chunk : n;
    const bool vec_ok = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(keys) & 15) == 0);
    for (uint64_t base = begin; base < end; base += 256) {
        uint64_t i0 = base + uint64_t(lane) * 4;
        float f[4];
        if (vec_ok && i0 + 3 < end) {
            float4 t = *reinterpret_cast<const float4 *>(kq + i0);
            f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) f[e] = (i0 + e < end) ? kq[i0 + e] : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            uint64_t c = (i0 + e < end) ? pair_key(f[e], uint32_t(i0 + e)) : PAIR_NONE;
            wl.offer(c);
        }
    }
    wl.store(lists + (uint64_t(q) * nlists + list) * (64 * R));
}

// ---------------------------------------------------------------------------------------------------
// tau[q] = the kth smallest of n keys of row q (as a float value), by a range-adaptive radix select on the
// order-preserving u32 image of the keys held in LDS.  This is all the sampled-threshold step of the Flat path needs
// from its key sample -- a sorted shortlist of the sample (k_topk_dense + k_topk_merge, 120 us at nq = 1000) is not.
// Distances of one query sit in a narrow band, so fixed bit fields put the whole sample into one or two histogram
// bins (measured: 95 us of serialised LDS atomics); instead every pass spreads the CURRENT value range [lo, hi]
// over up to 2048 equal power-of-two bins and narrows it to the bin holding the wanted rank, until the range is at most 2048 values
// wide and the histogram is exact (<= 4 passes for 32-bit keys).  +inf (rows past n) and NaN keys sort last and only
// count.  One workgroup per query; n <= SELECT_MAX_N.  Fewer than kth finite keys -> +inf.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t SELECT_MAX_N = 32768;
uint32_t select_tau_max_n() { return SELECT_MAX_N; }

__global__ __launch_bounds__(256) void k_select_tau(const float *__restrict__ keys, uint64_t ld, uint32_t n, uint32_t kth,
                                                    uint32_t nq_real, float *__restrict__ tau) {
    extern __shared__ uint32_t sel_smem[];
    uint32_t *o = sel_smem;        // [n] orderable keys
    uint32_t *hist = o + n;        // [2048]
    uint32_t *part = hist + 2048;  // [256] partial sums / reduction scratch; [256..257] results
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    if (q >= nq_real) {  // padding query of a batch: let nothing through (see Index::flat_knn_device)
        if (tid == 0) tau[q] = -INFINITY;
        return;
    }
    const float *kq = keys + uint64_t(q) * ld;
    const uint32_t O_INF = 0xFF800000u;  // f32_orderable(+inf); NaN is above it
    uint32_t mn = 0xFFFFFFFFu, mx = 0, fin = 0;
    for (uint32_t i = tid; i < n; i += 256) {
        uint32_t v = f32_orderable(kq[i]);
        o[i] = v;
        if (v < O_INF) {
            mn = v < mn ? v : mn;
            mx = v > mx ? v : mx;
            fin++;
        }
    }
    // block reduction of (min, max, count): wave shuffles, then the 4 wave results through LDS
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        uint32_t a = __shfl_xor(mn, d), b = __shfl_xor(mx, d), c = __shfl_xor(fin, d);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
        fin += c;
    }
    if ((tid & 63) == 0) {
        part[(tid >> 6) * 3 + 0] = mn;
        part[(tid >> 6) * 3 + 1] = mx;
        part[(tid >> 6) * 3 + 2] = fin;
    }
    __syncthreads();
    uint32_t lo = part[0], hi = part[1], n_fin = part[2];
#pragma unroll
    for (int w = 1; w < 4; w++) {
        lo = part[w * 3] < lo ? part[w * 3] : lo;
        hi = part[w * 3 + 1] > hi ? part[w * 3 + 1] : hi;
        n_fin += part[w * 3 + 2];
    }
    __syncthreads();
    if (kth == 0 || kth > n_fin) {  // uniform
        if (tid == 0) tau[q] = INFINITY;
        return;
    }
    uint32_t want = kth;
    for (int pass = 0; pass < 5; pass++) {  // the range shrinks >= 1024x per pass
        // bins of 2^sh values: the smallest shift that maps [lo, hi] into 2048 bins (no division in the key loop)
        const uint32_t span = hi - lo;  // width - 1
        const int sh = span < 2048 ? 0 : (32 - __builtin_clz(span)) - 11;
        const bool exact = sh == 0;
        for (uint32_t i = tid; i < 2048; i += 256) hist[i] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += 256) {
            uint32_t v = o[i];
            if (v >= lo && v <= hi) {
                atomicAdd(&hist[(v - lo) >> sh], 1u);
            }
        }
        __syncthreads();
        uint32_t sum = 0;
        for (uint32_t j = 0; j < 8; j++) sum += hist[tid * 8 + j];
        part[tid] = sum;
        __syncthreads();
        if (tid < 64) {  // first wave: scan of the 256 partial sums (4 per lane), then the 8 bins of the crossing one
            uint32_t p4[4], mine = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                p4[j] = part[tid * 4 + j];
                mine += p4[j];
            }
            uint32_t incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t up = __shfl_up(incl, d);
                if ((int)tid >= d) incl += up;
            }
            const uint64_t reach = __ballot(incl >= want);  // want <= keys in range, so some lane reaches it
            const uint32_t first = reach ? (uint32_t)__builtin_ctzll(reach) : 63u;
            if (tid == first) {
                uint32_t cum = incl - mine, t = 0;
                while (t < 3 && cum + p4[t] < want) cum += p4[t++];
                uint32_t b = (tid * 4 + t) * 8;
                const uint32_t bend = b + 7;
                while (b < bend && cum + hist[b] < want) cum += hist[b++];
                part[256] = b;
                part[257] = cum;
            }
        }
        __syncthreads();
        const uint32_t b = part[256];
        want -= part[257];
        __syncthreads();
        if (exact) {
            if (tid == 0) tau[q] = f32_from_orderable(lo + b);
            return;
        }
        const uint32_t nlo = lo + (b << sh);  // values of bin b
        const uint32_t nhi = nlo + ((1u << sh) - 1);
        lo = nlo;
        hi = nhi < hi && nhi >= nlo ? nhi : hi;
    }
    if (tid == 0) tau[q] = f32_from_orderable(hi);  // not reached: 2^32 / 1024^3 < 2048
}

// Cheap upper bound for small ranks: tau only has to be >= the kth smallest sampled key (a looser threshold lets a few
// more rows through the filter, nothing else changes).  Every thread keeps the 2 smallest keys of its strided share;
// the kth smallest of those 512 is the kth smallest of a SUBSET, hence >= the true one, and equal to it unless one
// thread happened to own 3 of the kth smallest (kth = 32 of ~10^4 keys over 256 threads: rarely, and then off by one
// rank).  One pass over the keys, no LDS image of them, no histograms: ~10 us instead of ~50-95 us at nq = 1000.
__global__ __launch_bounds__(256) void k_select_tau_small(const float *__restrict__ keys, uint64_t ld, uint32_t n, uint32_t kth,
                                                          uint32_t nq_real, float *__restrict__ tau) {
    __shared__ uint32_t cand[512];
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    if (q >= nq_real) {
        if (tid == 0) tau[q] = -INFINITY;
        return;
    }
    const float *kq = keys + uint64_t(q) * ld;
    uint32_t m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;  // m1 <= m2
    for (uint32_t i0 = tid; i0 < n; i0 += 256 * 8) {  // 8 independent loads in flight per thread
        float f[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            uint32_t i = i0 + u * 256;
            f[u] = i < n ? kq[i] : INFINITY;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            uint32_t v = f32_orderable(f[u]);
            if (v < m2) {
                m2 = v < m1 ? m1 : v;
                m1 = v < m1 ? v : m1;
            }
        }
    }
    cand[2 * tid] = m1;
    cand[2 * tid + 1] = m2;
    __syncthreads();
    // bitonic sort of the 512 candidates (256 compare-exchanges per stage, 45 stages), then element kth-1
    for (uint32_t k = 2; k <= 512; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            const uint32_t i = ((tid & ~(j - 1)) << 1) | (tid & (j - 1));  // lower index of this thread's pair
            const uint32_t p = i | j;
            const uint32_t a = cand[i], b = cand[p];
            const bool up = (i & k) == 0;
            if ((a > b) == up) {
                cand[i] = b;
                cand[p] = a;
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        const uint32_t v = cand[kth - 1];
        tau[q] = v >= 0xFF800000u ? INFINITY : f32_from_orderable(v);  // +inf / NaN / fewer than kth keys -> +inf
    }
}

// A short sample and a small rank (the unit minima of the 8-bit pass: ~160 values, rank 8): one wave per query holds the keys in
// registers (up to 8 per lane, as (value, position) pairs so that equal values are distinct) and takes the smallest kth times.
__global__ __launch_bounds__(256) void k_select_tau_tiny(const float *__restrict__ keys, uint64_t ld, uint32_t n, uint32_t kth,
                                                         uint32_t nq_real, uint32_t nq, float *__restrict__ tau) {
    const uint32_t lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    if (q >= nq_real) {  // padding query of a batch: let nothing through (see Index::flat_knn_device)
        if (lane == 0) tau[q] = -INFINITY;
        return;
    }
    uint64_t v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t i = j * 64 + lane;
        v[j] = i < n ? pair_key(keys[uint64_t(q) * ld + i], i) : PAIR_NONE;  // (NaN keys order last, as in the other selections)
    }
    uint64_t last = 0, cur = PAIR_NONE;
    for (uint32_t it = 0; it < kth; it++) {  // the smallest pair above the previous one
        uint64_t m = PAIR_NONE;
#pragma unroll
        for (int j = 0; j < 8; j++) m = (v[j] > last || it == 0) && v[j] < m ? v[j] : m;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const uint64_t o = __shfl_xor(m, off);
            m = o < m ? o : m;
        }
        cur = m;
        if (m == PAIR_NONE) break;
        last = m;
    }
    if (lane == 0) {
        const float t = cur == PAIR_NONE ? INFINITY : f32_from_orderable(uint32_t(cur >> 32));
        tau[q] = t == t ? t : INFINITY;  // fewer than kth finite keys -> +inf (a NaN key in rank: likewise)
    }
}

void launch_select_tau(const float *keys, uint64_t ld, uint32_t n, uint32_t nq, uint32_t nq_real, uint32_t kth, float *tau,
                       hipStream_t s) {
    if (nq == 0) return;
    if (kth >= 1 && kth <= 64 && n <= 512) {
        hipLaunchKernelGGL(k_select_tau_tiny, dim3((nq + 3) / 4), dim3(256), 0, s, keys, ld, n, kth, nq_real, nq, tau);
        VDB_HIP(hipGetLastError());
        return;
    }
    if (kth >= 1 && kth <= 64 && n >= 4 * 512) {  // small rank in a long sample: the subset bound
        hipLaunchKernelGGL(k_select_tau_small, dim3(nq), dim3(256), 0, s, keys, ld, n, kth, nq_real, tau);
        VDB_HIP(hipGetLastError());
        return;
    }
    VDB_REQUIRE(n <= SELECT_MAX_N, "select_tau: sample too long");
    size_t lds = (size_t(n) + 2048 + 258) * sizeof(uint32_t);
    func_max_lds(reinterpret_cast<const void *>(&k_select_tau), int(144 * 1024));
    hipLaunchKernelGGL(k_select_tau, dim3(nq), dim3(256), lds, s, keys, ld, n, kth, nq_real, tau);
    VDB_HIP(hipGetLastError());
}

// level 2: one wave per query merges nlists*cap_in keys
template <int R>
__global__ __launch_bounds__(64) void k_topk_merge(const uint64_t *__restrict__ lists, uint32_t nlists,
                                                   uint32_t cap_in, uint32_t k, uint64_t *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t q = blockIdx.x;
    WaveList<R> wl;
    wl.init(k);
    const uint64_t *src = lists + uint64_t(q) * nlists * cap_in;
    uint64_t total = uint64_t(nlists) * cap_in;
    uint64_t rounds = (total + 63) / 64;  // same trip count in every lane: all take part in the ballots
    for (uint64_t it = 0; it < rounds; it++) {
        uint64_t i = it * 64 + lane;
        uint64_t c = i < total ? src[i] : PAIR_NONE;
        wl.offer(c);
    }
    wl.store(out + uint64_t(q) * (64 * R));
}

// level 2 over one candidate list per query whose valid length is min(cnt[q], cap) (the tail is PAIR_NONE)
template <int R>
__global__ __launch_bounds__(64) void k_topk_merge_counted(const uint64_t *__restrict__ lists, uint32_t cap,
                                                           const uint32_t *__restrict__ cnt, uint32_t k,
                                                           uint64_t *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t q = blockIdx.x;
    WaveList<R> wl;
    wl.init(k);
    const uint64_t *src = lists + uint64_t(q) * cap;
    // cnt > cap: candidates were dropped (or a workgroup hit buffer filled) and the slots are not all written; the query
    // is redone by its caller, so an empty list is the safe output (no stale slot is ever interpreted as a row id)
    uint32_t total = cnt[q] <= cap ? cnt[q] : 0;
    uint32_t rounds = (total + 63) / 64;
    // the offers are a serial chain, the loads are not: keep 4 rounds of keys in flight (one wave per query, so the
    // load latency is otherwise paid once per round: 63 us for ~2000 candidates x 1000 queries)
    for (uint32_t it = 0; it < rounds; it += 4) {
        uint64_t c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            uint32_t i = (it + u) * 64 + lane;
            c[u] = i < total ? src[i] : PAIR_NONE;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (it + u < rounds) wl.offer(c[u]);  // wave-uniform
    }
    wl.store(out + uint64_t(q) * (64 * R));
}

template <int R>
static void topk_dense_r(const float *keys, uint64_t ld, uint64_t n, uint32_t nq, uint32_t k, uint64_t *lists,
                         hipStream_t s) {
    uint32_t nl = topk_num_lists(n);
    hipLaunchKernelGGL((k_topk_dense<R>), dim3((nl + 3) / 4, nq), dim3(256), 0, s, keys, ld, n, k, nl,
                       topk_chunk(n), lists);
}
template <int R>
static void topk_merge_r(const uint64_t *lists, uint32_t nlists, uint32_t cap_in, uint32_t nq, uint32_t k,
                         uint64_t *out, hipStream_t s) {
    hipLaunchKernelGGL((k_topk_merge<R>), dim3(nq), dim3(64), 0, s, lists, nlists, cap_in, k, out);
}

// shard merge input: per-shard results (f32 distance, u64 global id < 2^32) -> pair-key lists [q][S][cap_in]
// shard s's arrays start shard_stride BYTES after shard s-1's (the per-rank blocks of one all-gather buffer, or plain
// [S][nq][k] arrays when the stride is nq*k elements)
__global__ void k_pack_pairs(const char *__restrict__ dists, const char *__restrict__ ids,
                             const char *__restrict__ counts, uint64_t stride_d, uint64_t stride_i, uint64_t stride_c,
                             uint32_t S, uint32_t nq, uint32_t k, uint32_t cap_in, uint64_t *__restrict__ lists) {
    uint32_t q = blockIdx.x, s = blockIdx.y;
    const float *d = reinterpret_cast<const float *>(dists + s * stride_d);
    const uint64_t *id = reinterpret_cast<const uint64_t *>(ids + s * stride_i);
    uint64_t c = reinterpret_cast<const uint64_t *>(counts + s * stride_c)[q];
    for (uint32_t j = threadIdx.x; j < cap_in; j += blockDim.x) {
        uint64_t key = PAIR_NONE;
        if (j < k && j < c) {
            uint64_t at = uint64_t(q) * k + j;
            key = pair_key(d[at], uint32_t(id[at]));
        }
        lists[(uint64_t(q) * S + s) * cap_in + j] = key;
    }
}

void launch_pack_pairs(const float *dists, const uint64_t *ids, const uint64_t *counts, uint64_t stride_d,
                       uint64_t stride_i, uint64_t stride_c, uint32_t S, uint32_t nq, uint32_t k, uint32_t cap_in,
                       uint64_t *lists, hipStream_t s) {
    if (nq == 0 || S == 0) return;
    hipLaunchKernelGGL(k_pack_pairs, dim3(nq, S), dim3(64), 0, s, reinterpret_cast<const char *>(dists),
                       reinterpret_cast<const char *>(ids), reinterpret_cast<const char *>(counts), stride_d, stride_i,
                       stride_c, S, nq, k, cap_in, lists);
}

// The whole shard merge for k <= 64 in one launch: a wave per query reads the S per-shard results (distance, global id,
// count) in rounds of 64 pairs, sorts each round across its lanes and merges it into the running best-64
// (CandidatePair order: distance, then id), then writes the first k.  Replaces k_pack_pairs + k_topk_merge + k_finalize
// and their scratch lists.
__global__ __launch_bounds__(64) void k_merge_shards64(const char *__restrict__ dists, const char *__restrict__ ids,
                                                       const char *__restrict__ counts, uint64_t stride_d, uint64_t stride_i,
                                                       uint64_t stride_c, uint32_t S, uint32_t nq, uint32_t k,
                                                       uint64_t *__restrict__ out_idx, float *__restrict__ out_dist,
                                                       uint64_t *__restrict__ out_count) {
    const uint32_t q = blockIdx.x, lane = threadIdx.x & 63;
    const uint32_t total = S * k;
    uint64_t best = PAIR_NONE;
    for (uint32_t base = 0; base < total; base += 64) {
        const uint32_t i = base + lane;
        uint64_t c = PAIR_NONE;
        if (i < total) {
            const uint32_t s = i / k, j = i - s * k;
            const uint64_t cnt = reinterpret_cast<const uint64_t *>(counts + s * stride_c)[q];
            if (j < cnt) {
                const uint64_t at = uint64_t(q) * k + j;
                c = pair_key(reinterpret_cast<const float *>(dists + s * stride_d)[at],
                             uint32_t(reinterpret_cast<const uint64_t *>(ids + s * stride_i)[at]));
            }
        }
        best = merge64(best, sort64(c, lane), lane);
    }
    const bool ok = lane < k && best != PAIR_NONE;
    if (lane < k) {
        out_idx[uint64_t(q) * k + lane] = ok ? uint64_t(uint32_t(best)) : 0;
        out_dist[uint64_t(q) * k + lane] = ok ? f32_from_orderable(uint32_t(best >> 32)) : 0.0f;
    }
    const uint32_t n_ok = __builtin_popcountll(__ballot(ok));
    if (lane == 0) out_count[q] = n_ok;
}
void launch_merge_shards64(const float *dists, const uint64_t *ids, const uint64_t *counts, uint64_t stride_d,
                           uint64_t stride_i, uint64_t stride_c, uint32_t S, uint32_t nq, uint32_t k, uint64_t *out_idx,
                           float *out_dist, uint64_t *out_count, hipStream_t s) {
    if (nq == 0 || S == 0) return;
    VDB_REQUIRE(k >= 1 && k <= 64, "merge_shards64: k must be in 1..64");
    hipLaunchKernelGGL(k_merge_shards64, dim3(nq), dim3(64), 0, s, reinterpret_cast<const char *>(dists),
                       reinterpret_cast<const char *>(ids), reinterpret_cast<const char *>(counts), stride_d, stride_i,
                       stride_c, S, nq, k, out_idx, out_dist, out_count);
}

#define VDB_DISPATCH_R(cap, CALL)                                   \
    switch ((cap) / 64) {                                           \
        case 1: CALL(1); break;                                     \
        case 2: CALL(2); break;                                     \
        case 4: CALL(4); break;                                     \
        case 8: CALL(8); break;                                     \
        case 16: CALL(16); break;                                   \
        default: throw Error(1, "top-k: k must be <= 1024");        \
    }

void launch_topk_dense(const float *keys, uint64_t ld, uint64_t n, uint32_t nq, uint32_t k, uint64_t *lists,
                       hipStream_t s) {
    if (nq == 0) return;
    VDB_REQUIRE(k >= 1, "top-k: k must be >= 1");
    uint32_t cap = topk_capacity(k);
#define CALL(R) topk_dense_r<R>(keys, ld, n, nq, k, lists, s)
    VDB_DISPATCH_R(cap, CALL)
#undef CALL
}

// The same selection for k <= 64 without the serial insertions (~k (1 + ln(cnt / k)) dependent ballot / shuffle
// sequences in ONE wave per query: 71 us for 1000 queries x ~1000 candidates).  Four waves per query; a wave takes
// every fourth batch of 4 x 64 candidates, sorts the four runs across its lanes (bitonic: 21 compare-exchange stages
// through shuffles, the four runs interleaved so that the shuffle latencies overlap), merges them pairwise (minimum
// against the reversed partner = the 64 smallest of the 128 as a bitonic sequence, 6 more stages) and into its
// running best-64; the four best lists meet in LDS and wave 0 merges them.  A batch without a key below the wave's
// current 64th smallest is skipped.
__global__ __launch_bounds__(256) void k_top64_counted(const uint64_t *__restrict__ lists, uint32_t cap,
                                                       const uint32_t *__restrict__ cnt, uint64_t *__restrict__ out) {
    __shared__ uint64_t sbest[4][64];
    const uint32_t q = blockIdx.x;
    const uint32_t total = cnt[q] <= cap ? cnt[q] : 0;  // cnt > cap: see k_topk_merge_counted
    const uint64_t best = block_top64(lists + uint64_t(q) * cap, total, sbest);
    if (threadIdx.x < 64) out[uint64_t(q) * 64 + threadIdx.x] = best;
}

void launch_topk_merge_counted(const uint64_t *lists, uint32_t cap_in, const uint32_t *cnt, uint32_t nq, uint32_t k,
                               uint64_t *out, hipStream_t s) {
    if (nq == 0) return;
    VDB_REQUIRE(k >= 1, "top-k: k must be >= 1");
    uint32_t cap = topk_capacity(k);
    if (cap == 64) {
        hipLaunchKernelGGL(k_top64_counted, dim3(nq), dim3(256), 0, s, lists, cap_in, cnt, out);
        return;
    }
#define CALL(R) hipLaunchKernelGGL((k_topk_merge_counted<R>), dim3(nq), dim3(64), 0, s, lists, cap_in, cnt, k, out)
    VDB_DISPATCH_R(cap, CALL)
#undef CALL
}

void launch_topk_merge(const uint64_t *lists, uint32_t nlists, uint32_t cap_in, uint32_t nq, uint32_t k,
                       uint64_t *out, hipStream_t s) {
    if (nq == 0) return;
    VDB_REQUIRE(k >= 1, "top-k: k must be >= 1");
    uint32_t cap = topk_capacity(k);
#define CALL(R) topk_merge_r<R>(lists, nlists, cap_in, nq, k, out, s)
    VDB_DISPATCH_R(cap, CALL)
#undef CALL
}

}  // namespace vdb
