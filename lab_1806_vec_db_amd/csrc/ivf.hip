// ivf.hip -- IVFIndex (index_algorithm/ivf_index.rs) on the GPU: SURVEY 8(f-4), "a trivially related probe-list scan".
//
// Reference semantics restated:
//  * build (ivf_index.rs:66-118): k-means over all columns (optionally on a random sample), then every row goes to
//    its nearest centroid under the CandidatePair order (k_means.rs:40-57,166-170); cluster c lists its rows in
//    ascending id (:95-100).  The assignment is integer work -> bit-exact given the centroids: it runs as an exact
//    top-1 Flat search of the rows against the centroid set (strict-order distances, ties -> lower centroid).
//  * search (:143-154): probes = find_n_nearest(query, n_probes) = ResultSet over all centroids in index order, i.e.
//    Flat knn over the centroids; the members of the probed clusters are offered to ResultSet::add cluster by
//    cluster in probe order.  `add` replaces only on a strictly smaller DISTANCE (candidate_pair.rs:61-74), so on
//    exact ties at the cut the earlier-offered row stays: the scan is an ordered replay, not a lexicographic top-k.
//    Here: (1) probes by the Flat path of the centroid index, (2) the candidate ids of a query are laid out in offer
//    order, (3) k_rerank computes their exact distances, (4) k_pq_resort replays ResultSet::add over that order
//    (the same replay FlatIndex::knn_pq uses).
#include <algorithm>
#include <atomic>
#include <cstring>
#include <numeric>

#include "half_rows.hpp"
#include "pq_hnsw.hpp"

namespace vdb {

void ivf_clear(Index &ix) {
    ix.ivf.present = false;
    ix.ivf.cent.reset();
    ix.ivf.assign.clear();
    ix.ivf.offsets.clear();
    ix.ivf.sizes_desc.clear();
    ix.ivf.d_offsets.release();
    ix.ivf.d_members.release();
}

// Exact (distance, centroid) pair keys of EVERY centroid for nq device-resident queries, unsorted, in out[q][0..ld):
// one thread per (query, centroid) pair folds in reference order (k_rerank with the identity candidate list).  The
// centroid set is small and L2-resident, the pairs are many: this keeps every CU busy where a Flat scan of a
// 1000-row "corpus" launches 4 workgroups per 8 queries (measured: 15 ms of a 24 ms IVF step, 16 s of a 23 s build).
static void all_centroid_keys(Index &cent, const float *d_q, const float *d_qsq, uint64_t nq, uint32_t ld, uint64_t *ids,
                              uint64_t *out, hipStream_t s) {
    const uint32_t k = (uint32_t)cent.n;
    launch_iota_keys(ids, (uint32_t)nq, k, ld, s);
    launch_rerank(cent.d_rows.as<float>(), (uint32_t)cent.dim, d_q, (uint32_t)nq, cent.dist == 0 ? MET_L2_DIRECT : MET_COSINE,
                  cent.d_sq.as<float>(), d_qsq, ids, out, k, ld, s);
}

// nearest centroid of n device-resident rows (k_means.rs:40-57: minimum under the CandidatePair order): all exact
// distances, then a top-1 select per row
static void assign_nearest(Index &cent, const float *d_rows, const float *d_rows_sq, uint64_t n, uint64_t dim, uint64_t *out) {
    WsLease ws(cent);
    hipStream_t s = ws->stream;
    constexpr uint64_t CHUNK = 8192;
    const uint32_t ld = (uint32_t)((cent.n + 63) & ~63ull);
    ws->keys_a.reserve(CHUNK * ld * sizeof(uint64_t));
    ws->keys_b.reserve(CHUNK * ld * sizeof(uint64_t));
    ws->keys_c.reserve(CHUNK * 64 * sizeof(uint64_t));
    ws->qsq.reserve(CHUNK * sizeof(float));
    std::vector<uint64_t> top(CHUNK * 64);
    for (uint64_t r0 = 0; r0 < n; r0 += CHUNK) {
        const uint64_t nb = std::min<uint64_t>(CHUNK, n - r0);
        const float *qsq = d_rows_sq ? d_rows_sq + r0 : ws->qsq.as<float>();
        if (!d_rows_sq) launch_row_sqnorm(d_rows + r0 * dim, nb, (uint32_t)dim, ws->qsq.as<float>(), s);
        all_centroid_keys(cent, d_rows + r0 * dim, qsq, nb, ld, ws->keys_a.as<uint64_t>(), ws->keys_b.as<uint64_t>(), s);
        launch_topk_merge(ws->keys_b.as<uint64_t>(), 1, ld, (uint32_t)nb, 1, ws->keys_c.as<uint64_t>(), s);
        VDB_HIP(hipMemcpyAsync(top.data(), ws->keys_c.p, nb * 64 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        VDB_SYNC(s);
        for (uint64_t i = 0; i < nb; i++) out[r0 + i] = uint32_t(top[i * 64]);  // smallest pair key: (distance, index)
    }
}
static void ivf_assign_rows(Index &ix, std::vector<uint64_t> &assign) {
    assign.resize(ix.n);
    assign_nearest(*ix.ivf.cent, ix.d_rows.as<float>(), ix.d_sq.as<float>(), ix.n, ix.dim, assign.data());
}

static void ivf_install(Index &ix, uint64_t k, const float *centroids, const uint64_t *assign) {
    ix.ivf.half_overflows = 0;
    ix.ivf.q8_overflows = 0;
    VDB_REQUIRE(k >= 1, "The number of centroids should be greater than 0.");  // k_means.rs:45-48
    VDB_REQUIRE(ix.n < (1ull << 32), "ivf: too many rows");
    ivf_clear(ix);
    IVFState &iv = ix.ivf;
    iv.k = k;
    iv.cent = std::make_shared<Index>(ix.device, ix.dim, ix.dist);
    iv.cent->flat_mode = 1;  // exact scan: the centroid set is small and ties must resolve like ResultSet::add
    iv.cent->add_rows(centroids, k, false);
    if (assign) {
        iv.assign.assign(assign, assign + ix.n);
        for (uint64_t i = 0; i < ix.n; i++) VDB_REQUIRE(iv.assign[i] < k, "ivf: cluster id out of range");
    } else {
        ivf_assign_rows(ix, iv.assign);
    }
    // CSR: counting sort keeps ids ascending inside a cluster (ivf_index.rs:98-100 pushes rows in id order)
    iv.offsets.assign(k + 1, 0);
    for (uint64_t i = 0; i < ix.n; i++) iv.offsets[iv.assign[i] + 1]++;
    iv.sizes_desc.assign(iv.offsets.begin() + 1, iv.offsets.end());
    std::sort(iv.sizes_desc.begin(), iv.sizes_desc.end(), std::greater<uint32_t>());
    for (uint64_t c = 0; c < k; c++) iv.offsets[c + 1] += iv.offsets[c];
    std::vector<uint32_t> members(ix.n), cursor(iv.offsets.begin(), iv.offsets.end() - 1);
    for (uint64_t i = 0; i < ix.n; i++) members[cursor[iv.assign[i]]++] = (uint32_t)i;
    iv.d_offsets.reserve((k + 1) * sizeof(uint32_t));
    iv.d_members.reserve(std::max<uint64_t>(ix.n, 1) * sizeof(uint32_t));
    VDB_HIP(hipMemcpy(iv.d_offsets.p, iv.offsets.data(), (k + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (ix.n) VDB_HIP(hipMemcpy(iv.d_members.p, members.data(), ix.n * sizeof(uint32_t), hipMemcpyHostToDevice));
    iv.present = true;
}

void ivf_attach(Index &ix, uint64_t k_clusters, const float *centroids, const uint64_t *assign) {
    VDB_REQUIRE(!ix.elem_u8, "PQ / HNSW / IVF are built over f32 tables (DynamicIndex, dynamic_index.rs:11-14): a VecSet<u8> index serves Flat search");
    VDB_REQUIRE(centroids, "null centroids");
    ix.use_device();
    ivf_install(ix, k_clusters, centroids, assign);
}

// IVFIndex::from_vec_set (ivf_index.rs:66-118): sample (vec_set.rs:154-163), k-means, assignment
void ivf_build(Index &ix, uint64_t k_clusters, uint64_t train_n, uint64_t max_iter, float tol, uint64_t seed) {
    VDB_REQUIRE(!ix.elem_u8, "PQ / HNSW / IVF are built over f32 tables (DynamicIndex, dynamic_index.rs:11-14): a VecSet<u8> index serves Flat search");
    VDB_REQUIRE(ix.n > 0, "Cannot build an IVF index for an empty table");
    VDB_REQUIRE(k_clusters >= 1, "The number of centroids should be greater than 0.");
    ix.use_device();
    const float *rows = ix.host_rows();
    const size_t n = ix.n, dim = ix.dim;
    std::vector<float> sample;
    const float *train = rows;
    size_t nt = n;
    uint64_t rng = seed;
    if (train_n && train_n < n) {
        std::vector<size_t> perm(n);
        std::iota(perm.begin(), perm.end(), size_t(0));
        sample.resize(train_n * dim);
        for (size_t i = 0; i < train_n; i++) {
            size_t j = i + host_splitmix64(rng) % (n - i);
            std::swap(perm[i], perm[j]);
            std::memcpy(&sample[i * dim], rows + perm[i] * dim, dim * sizeof(float));
        }
        train = sample.data();
        nt = train_n;
    }
    std::vector<float> cent(k_clusters * dim);
    // Lloyd's assignment step on the GPU (k_means.rs:117-120): nt x k x dim strict-order distances per
    // iteration; seeding, the centroid update and the convergence test stay on the host
    DevBuf d_train;
    d_train.reserve(nt * dim * sizeof(float));
    VDB_HIP(hipMemcpy(d_train.p, train, nt * dim * sizeof(float), hipMemcpyHostToDevice));
    std::vector<uint64_t> a64(nt);
    KMeansAssignFn on_gpu = [&](const float *c, uint32_t *assign) {
        Index tmp(ix.device, ix.dim, ix.dist);
        tmp.flat_mode = 1;
        tmp.add_rows(c, k_clusters, false);
        assign_nearest(tmp, d_train.as<float>(), nullptr, nt, dim, a64.data());
        for (size_t i = 0; i < nt; i++) assign[i] = (uint32_t)a64[i];
    };
    host_kmeans(train, nt, dim, 0, dim, k_clusters, max_iter, tol, ix.dist, rng, cent.data(), on_gpu);
    ivf_install(ix, k_clusters, cent.data(), nullptr);
}

void ivf_export(Index &ix, float *centroids, uint64_t *assign) {
    VDB_REQUIRE(ix.ivf.present, "no IVF index");
    if (centroids) std::memcpy(centroids, ix.ivf.cent->host_rows(), ix.ivf.k * ix.dim * sizeof(float));
    if (assign) std::memcpy(assign, ix.ivf.assign.data(), ix.n * sizeof(uint64_t));
}

// candidate ids of query q in offer order: clusters in probe order, rows ascending inside a cluster; PAIR_NONE pads
__global__ __launch_bounds__(256) void k_ivf_candidates(const uint64_t *__restrict__ probe_keys, uint32_t ldp,
                                                        uint32_t n_probes,
                                                        const uint32_t *__restrict__ offsets,
                                                        const uint32_t *__restrict__ members, uint32_t ld,
                                                        uint64_t *__restrict__ cand,
                                                        unsigned long long *__restrict__ n_cand_total) {
    const uint32_t q = blockIdx.x;
    uint64_t *row = cand + uint64_t(q) * ld;
    uint32_t base = 0;
    for (uint32_t p = 0; p < n_probes; p++) {  // block-uniform
        const uint64_t pk = probe_keys[uint64_t(q) * ldp + p];  // sorted (distance, centroid) keys, PAIR_NONE pads
        if (pk == PAIR_NONE) break;
        const uint32_t c = uint32_t(pk);
        const uint32_t b = offsets[c], e = offsets[c + 1];
        for (uint32_t j = threadIdx.x; j < e - b; j += blockDim.x) row[base + j] = members[b + j];
        base += e - b;
    }
    for (uint32_t j = base + threadIdx.x; j < ld; j += blockDim.x) row[j] = PAIR_NONE;
    if (threadIdx.x == 0) atomicAdd(n_cand_total, (unsigned long long)base);  // rows scanned for this query (SURVEY 8d bytes)
}

// ---- certified half-precision pre-pass of the scan (half_rows.hpp) ------------------------------------------------------
// The scan keeps k of the thousands of rows in the probed clusters, and the replay of ResultSet::add over the offers gives
// the same set whether or not rows whose distance is strictly above D_k (the k-th smallest distance among the offers) are
// offered at all: such a row only ever fills a free slot or replaces a worse one, it is evicted before any row at or below
// D_k is (the evicted pair is the maximum of the set), and whether a row at D_k is admitted depends only on how many rows at
// or below D_k the set holds at that moment.  So: (1) lo / hi = a -/+ E for every offer from the fp16 image (half the
// bytes of the f32 rows); (2) T = k-th smallest hi >= D_k (k offers are at or below their own hi <= T); (3) offers with
// lo > T are certainly above D_k and are dropped, the others keep their order; (4) exact distances and the replay for
// what is left.  Offers whose approximation is not finite are kept.  The kept list has a fixed capacity; a query that
// overflows it sends the call back to the plain scan.
__global__ __launch_bounds__(64) void k_ivf_half_bounds(const uint16_t *__restrict__ rows_h, uint32_t dim, float inv_sx, float dx_abs,
                                                        float dx_rel, int metric, const float *__restrict__ Q,
                                                        const float *__restrict__ xsq, const float *__restrict__ qsq,
                                                        const uint64_t *__restrict__ cand, uint32_t ld, float *__restrict__ lo,
                                                        float *__restrict__ hi) {
    extern __shared__ __attribute__((aligned(16))) float ivf_q[];  // [dim]
    const uint32_t q = blockIdx.y, lane = threadIdx.x;
    for (uint32_t i = lane; i < dim; i += 64) ivf_q[i] = Q[uint64_t(q) * dim + i];
    __syncthreads();
    const float qs = qsq[q];
#pragma unroll 1
    for (uint32_t h = 0; h < 2; h++) {  // half_dots32 scores 32 rows per call, one per lane 0..31
        const uint32_t j = blockIdx.x * 64 + h * 32 + (lane & 31);
        const bool mine = lane < 32 && j < ld;
        const uint64_t c = mine ? cand[uint64_t(q) * ld + j] : PAIR_NONE;
        const bool live = c != PAIR_NONE;
        const uint32_t nb = live ? uint32_t(c) : 0u;
        float l = INFINITY, u = INFINITY;  // empty slot: never kept, never counted
        if (__ballot(live) != 0) {         // wave-uniform
            const float S = half_dots32(rows_h, dim, inv_sx, ivf_q, nb, live, lane);
            float a, E;
            half_approx(metric, dim, S, xsq[nb], qs, dx_abs, dx_rel, a, E);
            if (live) {
                const bool fin = E < INFINITY && a - a == 0.0f;
                l = fin ? a - E : -INFINITY;
                u = fin ? a + E : INFINITY;
            }
        }
        if (mine) {
            lo[uint64_t(q) * ld + j] = l;
            hi[uint64_t(q) * ld + j] = u;
        }
    }
}
// the same bounds from the 8-bit image (half_rows.hpp, "8-bit tier"): a quarter of the bytes of the f32 rows; the wider bound lets
// a few dozen offers per query through to the fp16 tier instead of a dozen
__global__ __launch_bounds__(64) void k_ivf_q8_bounds(const int8_t *__restrict__ rows_q8, const float *__restrict__ row_scale,
                                                      const float *__restrict__ row_err, uint32_t dim, int metric,
                                                      const int8_t *__restrict__ Q8, const float *__restrict__ q_scale,
                                                      const float *__restrict__ q_err, const float *__restrict__ xsq,
                                                      const float *__restrict__ qsq, const uint64_t *__restrict__ cand, uint32_t ld,
                                                      float *__restrict__ lo, float *__restrict__ hi) {
    extern __shared__ __attribute__((aligned(16))) int8_t ivf_q8[];  // [dim rounded up to 128], zero past dim
    const uint32_t q = blockIdx.y, lane = threadIdx.x;
    for (uint32_t i = lane; i < ((dim + 127) & ~127u) / 4; i += 64)
        reinterpret_cast<uint32_t *>(ivf_q8)[i] = i < dim / 4 ? reinterpret_cast<const uint32_t *>(Q8 + uint64_t(q) * dim)[i] : 0u;
    __syncthreads();
    const float qs = qsq[q], sq = q_scale[q], dq = q_err[q];
#pragma unroll 1
    for (uint32_t h = 0; h < 2; h++) {
        const uint32_t j = blockIdx.x * 64 + h * 32 + (lane & 31);
        const bool mine = lane < 32 && j < ld;
        const uint64_t c = mine ? cand[uint64_t(q) * ld + j] : PAIR_NONE;
        const bool live = c != PAIR_NONE;
        const uint32_t nb = live ? uint32_t(c) : 0u;
        float l = INFINITY, u = INFINITY;
        if (__ballot(live) != 0) {  // wave-uniform
            const int32_t isum = q8_dots32(rows_q8, dim, ivf_q8, nb, live, lane);
            float a, E;
            q8_approx(metric, dim, isum, xsq[nb], qs, row_scale[nb], row_err[nb], sq, dq, a, E);
            if (live) {
                const bool fin = E < INFINITY && a - a == 0.0f;
                l = fin ? a - E : -INFINITY;
                u = fin ? a + E : INFINITY;
            }
        }
        if (mine) {
            lo[uint64_t(q) * ld + j] = l;
            hi[uint64_t(q) * ld + j] = u;
        }
    }
}
// ---- the 8-bit tier, cluster-major ------------------------------------------------------------------------------------
// A batch of queries visits every cluster several times (1000 queries x 4 probes over 1000 clusters: 7.8 visits per row on
// average, popular clusters more), and query-major the visits of one cluster are far apart in time: every visit fetches the
// cluster's rows from HBM again.  Cluster-major they are adjacent: the (query, probe) pairs of the call are sorted by cluster
// (k_ivf_pairs + one LDS sort), and a wavefront takes 32 rows of a cluster and scores them against EVERY query that probes
// the cluster, one after the other, from registers.  The bounds land at
// the same [query][offer position] the query-major kernel writes (pos = offer position of the probe's first row).
__global__ __launch_bounds__(256) void k_ivf_pairs(const uint64_t *__restrict__ probe_keys, uint32_t ldp, uint32_t n_probes, uint32_t nq,
                                                   const uint32_t *__restrict__ offsets, uint64_t *__restrict__ pairs /* [nq*n_probes] */,
                                                   uint32_t *__restrict__ pos /* [nq*n_probes] */) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    uint32_t base = 0;
    bool ended = false;
    for (uint32_t p = 0; p < n_probes; p++) {
        const uint64_t pk = probe_keys[uint64_t(q) * ldp + p];
        ended |= pk == PAIR_NONE;
        const uint32_t e = q * n_probes + p;
        if (ended) {
            pairs[e] = PAIR_NONE;  // sorts last
            pos[e] = 0;
            continue;
        }
        const uint32_t c = uint32_t(pk);
        pairs[e] = (uint64_t(c) << 32) | e;
        pos[e] = base;
        base += offsets[c + 1] - offsets[c];
    }
}
// ascending sort of n <= 16384 keys by one workgroup in LDS (bitonic), then the runs of equal upper words: first[c] / count[c]
__global__ __launch_bounds__(1024) void k_ivf_sort_pairs(uint64_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ first,
                                                         uint32_t *__restrict__ count) {
    extern __shared__ uint64_t sp[];
    uint32_t m = 1;
    while (m < n) m <<= 1;
    for (uint32_t i = threadIdx.x; i < m; i += 1024) sp[i] = i < n ? keys[i] : PAIR_NONE;
    __syncthreads();
    for (uint32_t k = 2; k <= m; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < m / 2; t += 1024) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), pth = i | j;
                const uint64_t a = sp[i], b = sp[pth];
                if ((a > b) == ((i & k) == 0)) {
                    sp[i] = b;
                    sp[pth] = a;
                }
            }
            __syncthreads();
        }
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const uint64_t v = sp[i];
        keys[i] = v;
        if (v == PAIR_NONE) continue;
        const uint32_t c = uint32_t(v >> 32);
        if (i == 0 || uint32_t(sp[i - 1] >> 32) != c) first[c] = i;
        atomicAdd(&count[c], 1u);
    }
}
// A wavefront owns 32 rows of a cluster and KEEPS them in registers for all visits: lane 8g + j holds the pieces j, j + 8, ... of
// rows 8k + g (k = 0..3): 4 x ceil(dim/128) x 16 B = 128 registers at dim <= 1024.  Per visiting query: the query's 8-bit
// image goes to LDS (1 KB), every lane multiplies its pieces (v_dot4_i32_i8), the 8 partial sums of a row are added
// across the lanes of its group, and lanes 8g + k (k < 4) turn the sum of row 8k + g into its bounds.  The rows are fetched once.
__global__ __launch_bounds__(64) void k_ivf_q8_bounds_cm(const int8_t *__restrict__ rows_q8, const float *__restrict__ row_scale,
                                                         const float *__restrict__ row_err, uint32_t dim, int metric,
                                                         const int8_t *__restrict__ Q8, const float *__restrict__ q_scale,
                                                         const float *__restrict__ q_err, const float *__restrict__ xsq,
                                                         const float *__restrict__ qsq, const uint32_t *__restrict__ offsets,
                                                         const uint32_t *__restrict__ members, const uint64_t *__restrict__ pairs,
                                                         const uint32_t *__restrict__ pos, const uint32_t *__restrict__ first,
                                                         const uint32_t *__restrict__ count, uint32_t n_probes, uint32_t ld,
                                                         float *__restrict__ lo, float *__restrict__ hi) {
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) int8_t ivf_q8[];  // [1024]: the query image, zero past dim
    const uint32_t c = blockIdx.y, lane = threadIdx.x;
    const uint32_t nv = count[c];  // visits of this cluster in the call
    const uint32_t b = offsets[c], size = offsets[c + 1] - b, r0 = blockIdx.x * 32;
    if (nv == 0 || r0 >= size) return;  // block-uniform
    const uint32_t f = first[c];
    const uint32_t gg = lane >> 3, jj = lane & 7, npieces = dim / 16, nlines = (dim + 127) / 128;
    v4u rows[4][8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t j = r0 + 8 * k + gg;
        const uint32_t nbk = j < size ? members[b + j] : members[b + r0];  // (rows past the cluster's end: a copy of its first row here, never written out)
        const v4u *rp = reinterpret_cast<const v4u *>(rows_q8 + uint64_t(nbk) * dim);
        static_for<8>([&](auto ic) {
            constexpr int L = decltype(ic)::value;
            const uint32_t pth = L * 8 + jj;
            rows[k][L] = rp[pth < npieces ? pth : npieces - 1];  // (past the row's end: any piece, it meets zero query bytes)
        });
    }
    // epilogue lanes: lane 8g + k, k < 4, finishes row 8k + g
    const uint32_t jm = r0 + 8 * jj + gg;
    const bool fin_lane = jj < 4 && jm < size;
    const uint32_t nbm = fin_lane ? members[b + jm] : 0u;
    const float xs_m = xsq[nbm], sx_m = row_scale[nbm], dx_m = row_err[nbm];
    for (uint32_t t = 0; t < nv; t++) {
        const uint32_t e = uint32_t(pairs[f + t]);  // q * n_probes + p
        const uint32_t q = e / n_probes, at = pos[e];
        __syncthreads();
        reinterpret_cast<v4u *>(ivf_q8)[lane] = lane < npieces ? reinterpret_cast<const v4u *>(Q8 + uint64_t(q) * dim)[lane] : (v4u){0u, 0u, 0u, 0u};
        __syncthreads();
        int32_t acc[4] = {0, 0, 0, 0};
        static_for<8>([&](auto ic) {
            constexpr int L = decltype(ic)::value;
            if ((uint32_t)L < nlines) {  // uniform
                const v4u qq = reinterpret_cast<const v4u *>(ivf_q8)[L * 8 + jj];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int32_t a = acc[k];
                    a = __builtin_amdgcn_sdot4((int)rows[k][L].x, (int)qq.x, a, false);
                    a = __builtin_amdgcn_sdot4((int)rows[k][L].y, (int)qq.y, a, false);
                    a = __builtin_amdgcn_sdot4((int)rows[k][L].z, (int)qq.z, a, false);
                    a = __builtin_amdgcn_sdot4((int)rows[k][L].w, (int)qq.w, a, false);
                    acc[k] = a;
                }
            }
        });
#pragma unroll
        for (int k = 0; k < 4; k++) {
            acc[k] += __shfl_xor(acc[k], 1);
            acc[k] += __shfl_xor(acc[k], 2);
            acc[k] += __shfl_xor(acc[k], 4);
        }
        const int32_t isum = jj == 0 ? acc[0] : (jj == 1 ? acc[1] : (jj == 2 ? acc[2] : acc[3]));
        if (fin_lane) {
            float a, E;
            q8_approx(metric, dim, isum, xs_m, qsq[q], sx_m, dx_m, q_scale[q], q_err[q], a, E);
            const bool fin = E < INFINITY && a - a == 0.0f;
            const uint64_t o = uint64_t(q) * ld + at + jm;
            lo[o] = fin ? a - E : -INFINITY;
            hi[o] = fin ? a + E : INFINITY;
        }
    }
}
// out[q][0..ld2) = the offers of query q with lo <= T[q], in offer order, PAIR_NONE padded; flags[q] = 1 when they do not fit
__global__ __launch_bounds__(256) void k_ivf_keep(const uint64_t *__restrict__ cand, uint32_t ld, const float *__restrict__ lo,
                                                  const float *__restrict__ T, uint64_t *__restrict__ out, uint32_t ld2,
                                                  uint8_t *__restrict__ flags, unsigned long long *__restrict__ n_kept_total) {
    __shared__ uint32_t wcnt[4];
    const uint32_t q = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float t = T[q];
    uint32_t base = 0;
    for (uint32_t j0 = 0; j0 < ld; j0 += 256) {  // block-uniform
        const uint32_t j = j0 + tid;
        const uint64_t c = j < ld ? cand[uint64_t(q) * ld + j] : PAIR_NONE;
        const bool keep = c != PAIR_NONE && lo[uint64_t(q) * ld + j] <= t;
        const uint64_t m = __ballot(keep);
        if (lane == 0) wcnt[wave] = (uint32_t)__builtin_popcountll(m);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; w++) {
            before += w < wave ? wcnt[w] : 0u;
            total += wcnt[w];
        }
        const uint32_t pos = base + before + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1));
        if (keep && pos < ld2) out[uint64_t(q) * ld2 + pos] = c;
        base += total;
        __syncthreads();
    }
    for (uint32_t j = base + tid; j < ld2; j += 256) out[uint64_t(q) * ld2 + j] = PAIR_NONE;
    if (tid == 0 && base > ld2) flags[q] = 1;
    if (tid == 0) atomicAdd(n_kept_total, (unsigned long long)(base < ld2 ? base : ld2));  // rows the exact stage fetches
}
static std::atomic<int> g_ivf_half{1};  // 1 auto, 0 off
void ivf_set_half(int v) { g_ivf_half = v; }
static std::atomic<int> g_ivf_q8{1};  // the 8-bit tier in front of the fp16 tier: 1 auto, 0 off
void ivf_set_q8(int v) { g_ivf_q8 = v; }

void ivf_knn_device(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, uint64_t k, uint64_t n_probes,
                    uint64_t *d_idx, float *d_dist, uint64_t *d_cnt, bool use_half, bool use_q8) {
    hipStream_t s = ws.stream;
    IVFState &iv = ix.ivf;
    if (nq == 0) return;
    VDB_REQUIRE(n_probes > 0, "The number of probes should be greater than 0.");  // k_means.rs:175-178
    if (k == 0 || ix.n == 0) {
        VDB_HIP(hipMemsetAsync(d_cnt, 0, nq * sizeof(uint64_t), s));
        return;
    }
    const uint64_t np = std::min<uint64_t>(n_probes, iv.k);
    {   // the candidate rows of a call are nq x (rows of the np largest clusters) pair keys, twice: bound them
        uint64_t b = 0;
        for (uint64_t j = 0; j < np; j++) b += iv.sizes_desc[j];
        uint64_t qs = std::max<uint64_t>(1, (size_t(2) << 30) / (std::max<uint64_t>(b, 64) * 16));
        // The cluster-major 8-bit tier sorts the (query, probe) pairs of a call in one workgroup's LDS: 16 384 pairs.  A call with
        // more pairs used to take the query-major tier, which fetches a cluster once per visit (1000 queries x 64 probes: 61 GB
        // per step instead of ~1 GB per sub-batch) -- sub-batches of 16 384 / np queries keep it cluster-major.
        if (use_q8 && g_ivf_q8 == 1 && np <= 4096 && nq * np > 16384) qs = std::min<uint64_t>(qs, std::max<uint64_t>(1, 16384 / np));
        if (nq > qs) {
            uint64_t tot[4] = {0, 0, 0, 0};
            for (uint64_t q0 = 0; q0 < nq; q0 += qs) {
                ivf_knn_device(ix, ws, d_q + q0 * ix.dim, std::min(qs, nq - q0), k, n_probes, d_idx + q0 * k, d_dist + q0 * k, d_cnt + q0, use_half, use_q8);
                tot[0] += iv.last_offers.load();
                tot[1] += iv.last_kept.load();
                tot[2] += iv.last_kept_q8.load();
                tot[3] += iv.last_rows_fetched_q8.load();
            }
            iv.last_offers = tot[0];  // (the statistics of a call are those of all its sub-batches)
            iv.last_kept = tot[1];
            iv.last_kept_q8 = tot[2];
            iv.last_rows_fetched_q8 = tot[3];
            return;
        }
    }
    // (1) probes = find_n_nearest (k_means.rs:174-190): a ResultSet over all centroids in index order keeps, on equal
    // distances, the lower index -- the np smallest (distance, index) pairs.  All centroid distances, then a select.
    ws.qsq.reserve(nq * sizeof(float));
    launch_row_sqnorm(d_q, nq, (uint32_t)ix.dim, ws.qsq.as<float>(), s);
    const uint32_t ldc = (uint32_t)((iv.k + 63) & ~63ull), capp = np <= 1024 ? topk_capacity((uint32_t)np) : ldc;
    ws.keys_a.reserve(nq * ldc * sizeof(uint64_t));
    ws.keys_b.reserve(nq * ldc * sizeof(uint64_t));
    ws.lut.reserve(nq * capp * sizeof(uint64_t));
    uint64_t *d_probes = ws.lut.as<uint64_t>();
    all_centroid_keys(*iv.cent, d_q, ws.qsq.as<float>(), nq, ldc, ws.keys_a.as<uint64_t>(), ws.keys_b.as<uint64_t>(), s);
    if (np <= 1024) {
        launch_topk_merge(ws.keys_b.as<uint64_t>(), 1, ldc, (uint32_t)nq, (uint32_t)np, d_probes, s);
    } else {  // more probes than the register-resident select holds: every centroid key sorted, the first np taken
        const size_t tb = sort_rows_temp_bytes(nq, ldc);
        ws.dense.reserve(tb);
        launch_sort_rows(ws.keys_b.as<uint64_t>(), d_probes, nq, ldc, ws.dense.p, tb, s);
    }
    // (2) candidate lists; the np largest clusters bound every query's candidate count
    uint64_t bound = 0;
    for (uint64_t j = 0; j < np; j++) bound += iv.sizes_desc[j];
    if (bound == 0) {
        VDB_HIP(hipMemsetAsync(d_cnt, 0, nq * sizeof(uint64_t), s));
        return;
    }
    const uint32_t ld = (uint32_t)((bound + 63) & ~63ull);
    const uint32_t ksel = (uint32_t)std::min<uint64_t>(k, bound);
    ws.keys_a.reserve(nq * ld * sizeof(uint64_t));
    ws.keys_b.reserve(nq * ld * sizeof(uint64_t));
    ws.misc.reserve(64 + nq * np * (sizeof(uint64_t) + sizeof(uint32_t)) + 2 * (iv.k + 1) * sizeof(uint32_t) + 64);  // counters | cluster-major pair tables
    unsigned long long *d_ncand = ws.misc.as<unsigned long long>();
    VDB_HIP(hipMemsetAsync(d_ncand, 0, 4 * sizeof(unsigned long long), s));  // [0] offers, [1] kept for the exact stage, [2] kept by the 8-bit tier, [3] rows the cluster-major tier fetched
    hipLaunchKernelGGL(k_ivf_candidates, dim3((unsigned)nq), dim3(256), 0, s, d_probes, capp, (uint32_t)np,
                       iv.d_offsets.as<uint32_t>(), iv.d_members.as<uint32_t>(), ld, ws.keys_a.as<uint64_t>(), d_ncand);
    // (2b) the certified half-precision pre-pass: only the offers that may be among the k nearest go on
    const uint64_t *cand_keys = ws.keys_a.as<uint64_t>();
    uint64_t *exact_keys = ws.keys_b.as<uint64_t>();
    uint32_t ldx = ld;
    const uint32_t ld2 = (uint32_t)((std::min<uint64_t>(ld, std::max<uint64_t>(256, 16ull * ksel)) + 63) & ~63ull);
    const bool half = use_half && g_ivf_half && iv.half_overflows < 4 && (ld <= select_tau_max_n() || ksel <= 64) && ld2 < ld && ksel >= 1 &&  // (k_select_tau_small takes any length)
                      ix.ensure_rows_h(ws);
    double scan_bytes_per_row = double(ix.dim) * sizeof(float) + sizeof(float);
    uint8_t *flags = nullptr;
    bool q8 = false, cluster_major = false;
    if (half) {
        const int metric = ix.dist == 0 ? MET_L2_DIRECT : MET_COSINE;
        ws.dense.reserve(2 * nq * ld * sizeof(float));
        ws.qaux.reserve(nq * sizeof(float));
        ws.lists.reserve(2 * nq * ld2 * sizeof(uint64_t));
        float *d_lo = ws.dense.as<float>(), *d_hi = d_lo + nq * ld, *d_T = ws.qaux.as<float>();
        uint64_t *kept = ws.lists.as<uint64_t>();
        flags = static_cast<uint8_t *>(ws.pinned(2 * nq));
        std::memset(flags, 0, 2 * nq);
        // first tier, when the lists are long enough to pay for it: bounds from the 8-bit image (1 B/element) keep at most ldA
        // offers per query for the fp16 tier
        uint32_t ldh = ld;  // length of the lists the fp16 tier reads
        const uint32_t ldA = (uint32_t)((std::min<uint64_t>(ld, std::max<uint64_t>(std::max<uint64_t>(1024, 64ull * ksel), ld / 8)) + 63) & ~63ull);
        if (use_q8 && g_ivf_q8 && iv.q8_overflows < 4 && ld >= 4 * ldA && ix.ensure_rows_q8(ws)) {
            ws.keys_c.reserve(nq * ldA * sizeof(uint64_t));
            ws.qfrag.reserve(nq * (ix.dim + 2 * sizeof(float)) + 256);  // the queries' 8-bit images, scales and errors
            int8_t *d_q8 = ws.qfrag.as<int8_t>();
            float *d_qsc = reinterpret_cast<float *>(d_q8 + ((nq * ix.dim + 15) & ~uint64_t(15))), *d_qer = d_qsc + nq;
            launch_rows_to_q8(d_q, nq, (uint32_t)ix.dim, d_q8, d_qsc, d_qer, s);
            const uint64_t npairs = nq * np;
            cluster_major = g_ivf_q8 != 2 && npairs >= 256 && npairs <= 16384 && np <= 65535 && iv.k <= 65535;
            if (cluster_major) {
                // (query, probe) pairs by cluster; empty offer slots are never written below: +inf = not an offer
                uint64_t *d_pairs = reinterpret_cast<uint64_t *>(ws.misc.as<uint8_t>() + 64);
                uint32_t *d_pos = reinterpret_cast<uint32_t *>(d_pairs + npairs), *d_first = d_pos + npairs, *d_count = d_first + iv.k + 1;
                VDB_HIP(hipMemsetAsync(d_first, 0, 2 * (iv.k + 1) * sizeof(uint32_t), s));
                VDB_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_lo), 0x7f800000, 2 * nq * ld, s));
                hipLaunchKernelGGL(k_ivf_pairs, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, d_probes, capp, (uint32_t)np, (uint32_t)nq,
                                   iv.d_offsets.as<uint32_t>(), d_pairs, d_pos);
                func_max_lds(reinterpret_cast<const void *>(&k_ivf_sort_pairs), int(16384 * sizeof(uint64_t)));
                hipLaunchKernelGGL(k_ivf_sort_pairs, dim3(1), dim3(1024), (npairs <= 8192 ? 8192 : 16384) * sizeof(uint64_t), s, d_pairs,
                                   (uint32_t)npairs, d_first, d_count);
            }
            ix.prof_begin(ws, "ivf_q8", 0.0);
            if (cluster_major) {
                uint64_t *d_pairs = reinterpret_cast<uint64_t *>(ws.misc.as<uint8_t>() + 64);
                uint32_t *d_pos = reinterpret_cast<uint32_t *>(d_pairs + npairs), *d_first = d_pos + npairs, *d_count = d_first + iv.k + 1;
                const unsigned chunks = (unsigned)((iv.sizes_desc.empty() ? 0u : iv.sizes_desc[0]) + 31) / 32;
                hipLaunchKernelGGL(k_ivf_q8_bounds_cm, dim3(std::max(chunks, 1u), (unsigned)iv.k), dim3(64), 1024, s, ix.d_rows_q8.as<int8_t>(), ix.d_q8_scale.as<float>(),
                                   ix.d_q8_err.as<float>(), (uint32_t)ix.dim, metric, d_q8, d_qsc, d_qer, ix.d_sq.as<float>(), ws.qsq.as<float>(),
                                   iv.d_offsets.as<uint32_t>(), iv.d_members.as<uint32_t>(), d_pairs, d_pos, d_first, d_count, (uint32_t)np, ld, d_lo, d_hi);
            } else {
                hipLaunchKernelGGL(k_ivf_q8_bounds, dim3(ld / 64, (unsigned)nq), dim3(64), (ix.dim + 127) & ~uint64_t(127), s, ix.d_rows_q8.as<int8_t>(),
                                   ix.d_q8_scale.as<float>(), ix.d_q8_err.as<float>(), (uint32_t)ix.dim, metric, d_q8, d_qsc, d_qer,
                                   ix.d_sq.as<float>(), ws.qsq.as<float>(), cand_keys, ld, d_lo, d_hi);
            }
            ix.prof_end(ws);
            launch_select_tau(d_hi, ld, ld, (uint32_t)nq, (uint32_t)nq, ksel, d_T, s);
            uint8_t *flagsA = flags + nq;
            hipLaunchKernelGGL(k_ivf_keep, dim3((unsigned)nq), dim3(256), 0, s, cand_keys, ld, d_lo, d_T, ws.keys_c.as<uint64_t>(), ldA, flagsA,
                               d_ncand + 2);
            cand_keys = ws.keys_c.as<uint64_t>();
            ldh = ldA;
            q8 = true;
        }
        ix.prof_begin(ws, "ivf_half", 0.0);
        hipLaunchKernelGGL(k_ivf_half_bounds, dim3(ldh / 64, (unsigned)nq), dim3(64), ix.dim * sizeof(float), s, ix.d_rows_h.as<uint16_t>(),
                           (uint32_t)ix.dim, 1.0f / ix.half_sx(), ix.half_dx_abs, ix.half_dx_rel, metric, d_q, ix.d_sq.as<float>(),
                           ws.qsq.as<float>(), cand_keys, ldh, d_lo, d_hi);
        ix.prof_end(ws);
        launch_select_tau(d_hi, ldh, ldh, (uint32_t)nq, (uint32_t)nq, ksel, d_T, s);
        hipLaunchKernelGGL(k_ivf_keep, dim3((unsigned)nq), dim3(256), 0, s, cand_keys, ldh, d_lo, d_T, kept, ld2, flags, d_ncand + 1);
        cand_keys = kept;
        exact_keys = kept + nq * ld2;
        ldx = ld2;
    }
    ix.prof_begin(ws, "ivf_rerank", 0.0);
    // (3) exact distances in offer order, (4) ResultSet::add replay, sorted output (into_sorted_vec, :153)
    launch_rerank(ix.d_rows.as<float>(), (uint32_t)ix.dim, d_q, (uint32_t)nq, ix.dist == 0 ? MET_L2_DIRECT : MET_COSINE,
                  ix.d_sq.as<float>(), ws.qsq.as<float>(), cand_keys, exact_keys, ldx, ldx, s);
    ix.prof_end(ws);
    if (k > ksel) {
        VDB_HIP(hipMemsetAsync(d_idx, 0, nq * k * sizeof(uint64_t), s));
        VDB_HIP(hipMemsetAsync(d_dist, 0, nq * k * sizeof(float), s));
    }
    pq_resort_finalize(ix, ws, exact_keys, ldx, ldx, nq, ksel, k, ix.id_offset, d_idx, d_dist, d_cnt);
    if (half) {  // a query whose kept offers did not fit: the whole call again with the plain scan (rare; the index stops trying after four)
        VDB_SYNC(s);
        bool over = false, overA = false;
        for (uint64_t q = 0; q < nq; q++) over |= flags[q] != 0;
        for (uint64_t q = 0; q8 && q < nq; q++) overA |= flags[nq + q] != 0;
        if (overA) iv.q8_overflows++;
        if (over || overA) {  // (the 8-bit tier's lists overflowed: again with the fp16 tier alone; the fp16 tier's: the plain scan)
            if (over) iv.half_overflows++;
            ivf_knn_device(ix, ws, d_q, nq, k, n_probes, d_idx, d_dist, d_cnt, !over, false);
            return;
        }
    }
    if (!ws.pending.empty()) {  // measurement on: the scan's algorithmic bytes = scanned rows x (dim*4 + 4), known only now
        unsigned long long total[4] = {0, 0, 0, 0};
        VDB_HIP(hipMemcpyAsync(total, d_ncand, sizeof(total), hipMemcpyDeviceToHost, s));
        VDB_SYNC(s);
        if (q8 && cluster_major) {  // rows the cluster-major tier read = the sizes of the clusters with at least one visit (a same-address
            // atomic per block inside the kernel costs more than the kernel: the visit counts are read back instead)
            const uint64_t npairs = nq * np;
            const uint32_t *d_count = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint64_t *>(ws.misc.as<uint8_t>() + 64) + npairs) + npairs + iv.k + 1;
            std::vector<uint32_t> cnt(iv.k);
            VDB_HIP(hipMemcpyAsync(cnt.data(), d_count, iv.k * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            VDB_SYNC(s);
            for (uint64_t c = 0; c < iv.k; c++)
                if (cnt[c]) total[3] += iv.offsets[c + 1] - iv.offsets[c];
        }
        iv.last_rows_fetched_q8 = total[3];
        if (q8 && ws.pending.size() >= 3) {  // 8-bit rows + scale, error, norm: for every offer (query-major) or once per row of a
            // visited cluster (cluster-major) + the bounds written per offer; fp16 rows for what that tier kept
            ws.pending[ws.pending.size() - 3].bytes += total[3] ? double(total[3]) * (double(ix.dim) + 3 * sizeof(float)) + double(total[0]) * 2 * sizeof(float)
                                                                  : double(total[0]) * (double(ix.dim) + 3 * sizeof(float));
            ws.pending[ws.pending.size() - 2].bytes += double(total[2]) * (double(ix.dim) * sizeof(uint16_t) + sizeof(float));
            ws.pending.back().bytes += double(total[1]) * scan_bytes_per_row;
            iv.last_offers = total[0];
            iv.last_kept = total[1];
            iv.last_kept_q8 = total[2];
            return;
        }
        iv.last_kept_q8 = 0;
        // (with the pre-pass the two records split the bytes asked for: fp16 rows + norms for every offer, f32 rows for the kept
        // ones; SURVEY 8(d)'s figure stays offers x (dim*4 + 4))
        if (half && ws.pending.size() >= 2) {
            ws.pending[ws.pending.size() - 2].bytes += double(total[0]) * (double(ix.dim) * sizeof(uint16_t) + sizeof(float));
            ws.pending.back().bytes += double(total[1]) * scan_bytes_per_row;
        } else {
            ws.pending.back().bytes += double(total[0]) * scan_bytes_per_row;
        }
        iv.last_offers = total[0];
        iv.last_kept = half ? total[1] : total[0];
    }
}

}  // namespace vdb
