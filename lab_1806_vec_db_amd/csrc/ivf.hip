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
#include <cstring>
#include <numeric>

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

void ivf_knn_device(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, uint64_t k, uint64_t n_probes,
                    uint64_t *d_idx, float *d_dist, uint64_t *d_cnt) {
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
        const uint64_t qs = std::max<uint64_t>(1, (size_t(2) << 30) / (std::max<uint64_t>(b, 64) * 16));
        if (nq > qs) {
            for (uint64_t q0 = 0; q0 < nq; q0 += qs)
                ivf_knn_device(ix, ws, d_q + q0 * ix.dim, std::min(qs, nq - q0), k, n_probes, d_idx + q0 * k, d_dist + q0 * k, d_cnt + q0);
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
    ws.misc.reserve(64);
    unsigned long long *d_ncand = ws.misc.as<unsigned long long>();
    VDB_HIP(hipMemsetAsync(d_ncand, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_ivf_candidates, dim3((unsigned)nq), dim3(256), 0, s, d_probes, capp, (uint32_t)np,
                       iv.d_offsets.as<uint32_t>(), iv.d_members.as<uint32_t>(), ld, ws.keys_a.as<uint64_t>(), d_ncand);
    ix.prof_begin(ws, "ivf_rerank", 0.0);
    // (3) exact distances in offer order, (4) ResultSet::add replay, sorted output (into_sorted_vec, :153)
    launch_rerank(ix.d_rows.as<float>(), (uint32_t)ix.dim, d_q, (uint32_t)nq, ix.dist == 0 ? MET_L2_DIRECT : MET_COSINE,
                  ix.d_sq.as<float>(), ws.qsq.as<float>(), ws.keys_a.as<uint64_t>(), ws.keys_b.as<uint64_t>(), ld, ld, s);
    ix.prof_end(ws);
    if (k > ksel) {
        VDB_HIP(hipMemsetAsync(d_idx, 0, nq * k * sizeof(uint64_t), s));
        VDB_HIP(hipMemsetAsync(d_dist, 0, nq * k * sizeof(float), s));
    }
    pq_resort_finalize(ix, ws, ws.keys_b.as<uint64_t>(), ld, ld, nq, ksel, k, ix.id_offset, d_idx, d_dist, d_cnt);
    if (!ws.pending.empty()) {  // measurement on: the scan's algorithmic bytes = scanned rows x (dim*4 + 4), known only now
        unsigned long long total = 0;
        VDB_HIP(hipMemcpyAsync(&total, d_ncand, sizeof(total), hipMemcpyDeviceToHost, s));
        VDB_SYNC(s);
        ws.pending.back().bytes += double(total) * (double(ix.dim) * sizeof(float) + sizeof(float));
    }
}

}  // namespace vdb
