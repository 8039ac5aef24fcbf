// vdbhip.hpp -- C++ host-side mirror of the reference's `DynamicIndex` (src/database/dynamic_index.rs:11-94) and of
// the index traits it forwards to (src/index_algorithm/mod.rs:35-154), header-only over the C ABI of vdbhip.h.
//
// The reference's host code is Rust; where a Rust toolchain is present the binding is the `extern "C"` block of
// INTEGRATION.md.  This header is the same seam for a C++ host: same method names, argument meaning and error
// behaviour (recoverable errors -> an exception carrying vdb_last_error(), the counterpart of anyhow::Result ->
// PyRuntimeError, pyo3/mod.rs:65,85; dimension mismatch is an error, database/mod.rs:427-429).
// There is no CPU path behind it: without a GPU every call throws.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "vdbhip.h"

namespace vdbhip {

enum class DistanceAlgorithm : int { L2Sqr = VDB_L2SQR, Cosine = VDB_COSINE };  // distance/mod.rs:17-28

// candidate_pair.rs:9-16
struct CandidatePair {
    uint64_t index;
    float distance;
};

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
inline void check(int status) {
    if (status != VDB_OK) throw Error(status, vdb_last_error());
}

// DynamicIndex: starts as the Flat arm; build_hnsw() switches to the HNSW arm like
// MetadataVecTable::build_hnsw_index (metadata_vec_table.rs:112-135), clear_hnsw() back (:137-152).
// The PQ table hangs off the index handle (the reference passes &PQTable into knn_pq; here it is attached state).
class DynamicIndex {
public:
    DynamicIndex(uint64_t dim, DistanceAlgorithm dist, int device = 0) : dim_(dim), dist_(dist) {
        check(vdb_index_create(device, dim, (int)dist, &h_));
    }
    ~DynamicIndex() {
        if (h_) vdb_index_destroy(h_);
    }
    DynamicIndex(const DynamicIndex &) = delete;
    DynamicIndex &operator=(const DynamicIndex &) = delete;
    DynamicIndex(DynamicIndex &&o) noexcept : h_(o.h_), dim_(o.dim_), dist_(o.dist_) { o.h_ = nullptr; }

    uint64_t len() const {
        uint64_t n = 0;
        check(vdb_index_len(h_, &n));
        return n;
    }
    bool is_empty() const { return len() == 0; }
    uint64_t dim() const { return dim_; }
    DistanceAlgorithm dist() const { return dist_; }
    // Index<usize> for VecSet (vec_set.rs:22-30)
    std::vector<float> row(uint64_t i) const {
        std::vector<float> v(dim_);
        check(vdb_index_row(h_, i, v.data()));
        return v;
    }

    // add / batch_add (dynamic_index.rs:44-58): returns the index of the (first) new vector
    uint64_t add(const std::vector<float> &vec) {
        require_dim(vec.size());
        uint64_t first = 0;
        check(vdb_index_add(h_, vec.data(), 1, &first));
        return first;
    }
    uint64_t batch_add(const float *rows, uint64_t n) {
        uint64_t first = 0;
        check(vdb_index_add(h_, rows, n, &first));
        return first;
    }
    void swap_remove(uint64_t i) { check(vdb_index_swap_remove(h_, i)); }  // vec_set.rs:131-137

    // knn (dynamic_index.rs:66-73): Flat -> FlatIndex::knn, HNSW -> HNSWIndex::knn (default ef)
    std::vector<CandidatePair> knn(const std::vector<float> &query, uint64_t k) const {
        return has_hnsw() ? search(vdb_hnsw_knn, query, k, 0) : search_flat(query, k);
    }
    // knn_with_ef (:74-80): Flat ignores ef
    std::vector<CandidatePair> knn_with_ef(const std::vector<float> &query, uint64_t k, uint64_t ef) const {
        return has_hnsw() ? search(vdb_hnsw_knn, query, k, ef) : search_flat(query, k);
    }
    // knn_pq (:82-93)
    std::vector<CandidatePair> knn_pq(const std::vector<float> &query, uint64_t k, uint64_t ef) const {
        return has_hnsw() ? search(vdb_hnsw_knn_pq, query, k, ef) : search(vdb_flat_knn_pq, query, k, ef);
    }
    // batched form of knn for hosts that collect their pending queries (one corpus pass serves up to 128 of them)
    std::vector<std::vector<CandidatePair>> knn_batch(const float *queries, uint64_t nq, uint64_t k) const {
        std::vector<uint64_t> idx(nq * (k ? k : 1)), cnt(nq);
        std::vector<float> d(nq * (k ? k : 1));
        check(vdb_flat_knn(h_, queries, nq, dim_, k, idx.data(), d.data(), cnt.data()));
        std::vector<std::vector<CandidatePair>> out(nq);
        for (uint64_t q = 0; q < nq; q++)
            for (uint64_t j = 0; j < cnt[q]; j++) out[q].push_back({idx[q * k + j], d[q * k + j]});
        return out;
    }

    // MetadataVecTable::build_hnsw_index / clear_hnsw_index, build_pq_table / clear_pq_table
    void build_hnsw(uint64_t M = 16, uint64_t ef_construction = 200, uint64_t seed = 42, uint64_t batch = 64,
                    int nthreads = 16) {
        check(vdb_hnsw_build(h_, M, ef_construction, seed, batch, nthreads));
    }
    void clear_hnsw() { check(vdb_hnsw_clear(h_)); }
    bool has_hnsw() const {
        int v = 0;
        check(vdb_hnsw_has(h_, &v));
        return v != 0;
    }
    void build_pq(uint64_t n_bits, uint64_t m, uint64_t train_n = 0, uint64_t max_iter = 20, float tol = 1e-6f,
                  uint64_t seed = 42) {
        check(vdb_pq_build(h_, n_bits, m, train_n, max_iter, tol, seed));
    }
    void clear_pq() { check(vdb_pq_clear(h_)); }
    bool has_pq() const {
        int v = 0;
        check(vdb_pq_has(h_, &v));
        return v != 0;
    }
    vdb_index *handle() const { return h_; }

private:
    typedef int (*search_ef_fn)(vdb_index *, const float *, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t *, float *,
                                uint64_t *);
    void require_dim(size_t got) const {
        if (got != dim_)
            throw Error(VDB_ERR_INVALID, "dimension mismatch: index dim " + std::to_string(dim_) + ", got " + std::to_string(got));
    }
    std::vector<CandidatePair> collect(const std::vector<uint64_t> &idx, const std::vector<float> &d, uint64_t cnt) const {
        std::vector<CandidatePair> out;
        for (uint64_t j = 0; j < cnt; j++) out.push_back({idx[j], d[j]});
        return out;
    }
    std::vector<CandidatePair> search_flat(const std::vector<float> &query, uint64_t k) const {
        std::vector<uint64_t> idx(k ? k : 1);
        std::vector<float> d(k ? k : 1);
        uint64_t cnt = 0;
        check(vdb_flat_knn(h_, query.data(), 1, query.size(), k, idx.data(), d.data(), &cnt));
        return collect(idx, d, cnt);
    }
    std::vector<CandidatePair> search(search_ef_fn fn, const std::vector<float> &query, uint64_t k, uint64_t ef) const {
        std::vector<uint64_t> idx(k ? k : 1);
        std::vector<float> d(k ? k : 1);
        uint64_t cnt = 0;
        check(fn(h_, query.data(), 1, query.size(), k, ef, idx.data(), d.data(), &cnt));
        return collect(idx, d, cnt);
    }
    vdb_index *h_ = nullptr;
    uint64_t dim_;
    DistanceAlgorithm dist_;
};

// The same index over ALL GPUs of one host process (vdb_ctx_create + vdb_sharded_*): rows in contiguous blocks, per-shard
// top-k, ONE RCCL all-gather inside the library, exact merge.  What a multi-GPU `DynamicIndex::Gpu`
// arm forwards to; answers equal the single-GPU index's.
class ShardedIndex {
public:
    ShardedIndex(uint64_t dim, DistanceAlgorithm dist, const std::vector<int> &devices) : dim_(dim) {
        check(vdb_ctx_create(devices.data(), (int)devices.size(), &ctx_));
        int rc = vdb_sharded_create(ctx_, dim, (int)dist, &h_);
        if (rc != VDB_OK) {
            std::string msg = vdb_last_error();
            vdb_ctx_destroy(ctx_);
            throw Error(rc, msg);
        }
    }
    ~ShardedIndex() {
        if (h_) vdb_sharded_destroy(h_);
        if (ctx_) vdb_ctx_destroy(ctx_);
    }
    ShardedIndex(const ShardedIndex &) = delete;
    ShardedIndex &operator=(const ShardedIndex &) = delete;
    void set_rows(const float *rows, uint64_t n) { check(vdb_sharded_set_rows(h_, rows, n)); }
    // REPLICA layout: every GPU keeps all rows and a search splits the queries -- the layout the HNSW searches need
    // ("replicas only"); knn_batch / knn_pq_batch work in it as well
    void set_rows_replica(const float *rows, uint64_t n) { check(vdb_sharded_set_rows_replica(h_, rows, n)); }
    void build_hnsw(uint64_t ef_construction = 200, uint64_t M = 16, uint64_t seed = 42, uint64_t batch = 1, int nthreads = 0) {
        check(vdb_sharded_hnsw_build(h_, M, ef_construction, seed, batch, nthreads));  // metadata_vec_table.rs:84-98
    }
    // IndexKNNWithEf::knn_with_ef / HNSWIndex::knn_pq over the replicas (dynamic_index.rs:76-93)
    std::vector<std::vector<CandidatePair>> knn_with_ef_batch(const float *queries, uint64_t nq, uint64_t k, uint64_t ef) const {
        return run(nq, k, [&](uint64_t *i, float *d, uint64_t *c) { return vdb_sharded_hnsw_knn(h_, queries, nq, dim_, k, ef, i, d, c); });
    }
    std::vector<std::vector<CandidatePair>> hnsw_knn_pq_batch(const float *queries, uint64_t nq, uint64_t k, uint64_t ef) const {
        return run(nq, k, [&](uint64_t *i, float *d, uint64_t *c) { return vdb_sharded_hnsw_knn_pq(h_, queries, nq, dim_, k, ef, i, d, c); });
    }
    bool poisoned() const {
        int v = 0;
        check(vdb_sharded_poisoned(h_, &v));
        return v != 0;
    }
    uint64_t len() const {
        uint64_t n = 0;
        check(vdb_sharded_len(h_, &n));
        return n;
    }
    std::vector<std::vector<CandidatePair>> knn_batch(const float *queries, uint64_t nq, uint64_t k) const {
        return run(nq, k, [&](uint64_t *i, float *d, uint64_t *c) { return vdb_sharded_flat_knn(h_, queries, nq, dim_, k, i, d, c); });
    }
    void attach_pq(uint64_t n_bits, uint64_t m, const std::vector<float> &centroids) {
        check(vdb_sharded_pq_attach(h_, n_bits, m, centroids.data()));
    }
    std::vector<std::vector<CandidatePair>> knn_pq_batch(const float *queries, uint64_t nq, uint64_t k, uint64_t ef) const {
        return run(nq, k, [&](uint64_t *i, float *d, uint64_t *c) { return vdb_sharded_knn_pq(h_, queries, nq, dim_, k, ef, i, d, c); });
    }

private:
    template <class F>
    std::vector<std::vector<CandidatePair>> run(uint64_t nq, uint64_t k, F fn) const {
        std::vector<uint64_t> idx(nq * (k ? k : 1)), cnt(nq);
        std::vector<float> d(nq * (k ? k : 1));
        check(fn(idx.data(), d.data(), cnt.data()));
        std::vector<std::vector<CandidatePair>> out(nq);
        for (uint64_t q = 0; q < nq; q++)
            for (uint64_t j = 0; j < cnt[q]; j++) out[q].push_back({idx[q * k + j], d[q * k + j]});
        return out;
    }
    vdb_ctx *ctx_ = nullptr;
    vdb_sharded *h_ = nullptr;
    uint64_t dim_;
};

// calc_dist (pyo3/mod.rs:43-48)
inline float calc_dist(const std::vector<float> &a, const std::vector<float> &b, DistanceAlgorithm dist, int device = 0) {
    float out = 0;
    check(vdb_calc_dist(device, a.data(), b.data(), a.size() < b.size() ? a.size() : b.size(), (int)dist, &out));
    return out;
}

}  // namespace vdbhip
