// pq_hnsw.hpp -- PQ (pq.hip) and HNSW (hnsw.hip, hnsw_build.cpp) entry points used by api.hip.
#pragma once
#include <functional>

#include "index.hpp"

namespace vdb {

// ---- PQ (distance/pq_table.rs) ----
void pq_attach(Index &ix, uint64_t n_bits, uint64_t m, const float *centroids, const uint8_t *codes);
void pq_build(Index &ix, uint64_t n_bits, uint64_t m, uint64_t train_n, uint64_t max_iter, float tol, uint64_t seed);
void pq_clear(Index &ix);
void pq_set_adc_fast(int v);
void pq_set_adc16(int v);
void pq_set_adc8_sliced(int v);  // 8-bit codes: 0 = eight queries per pass on sliced 16-bit tables, 1 = one query per pass (byte table)
void pq_set_adc16_sample(int v);  // threshold sample of the quantised scan on the quantised tables too: 0 auto, 1 off (f32 sample)  // quantised first pass of the ADC scan: 0 auto, 1 off
void flat_knn_pq_device(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, uint64_t k, uint64_t ef,
                        uint64_t *d_idx, float *d_dist, uint64_t *d_cnt);

// create_lookup / ADC over all rows, as the search kernels compute them (host outputs; parity tests of a11 / a12)
void pq_export_lookup(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, float *h_lut, float *h_qcache);
void pq_export_adc_all(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, float *h_out);

// row-sharded knn_pq: per-shard export of the ADC shortlist, and the merge + pq_resort replay
void flat_knn_pq_shard_device(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, uint64_t k, uint64_t ef,
                              uint64_t *d_adc_keys, uint64_t *d_exact_keys);
void pq_merge_resort_device(Index &ix, Workspace &ws, const uint64_t *d_adc, const uint64_t *d_exact,
                            uint64_t n_shards, uint64_t nq, uint64_t efg, uint64_t k, uint64_t *d_idx, float *d_dist,
                            uint64_t *d_cnt);

// host k-means over columns [c0, c1) (k_means.rs:61-162; RNG = splitmix64, parity unpinned), k x (c1-c0) centroids out
// assign_fn (optional): cluster of every training row for the given k centroids (the Lloyd assignment step)
using KMeansAssignFn = std::function<void(const float *cent, uint32_t *assign)>;
void host_kmeans(const float *train, size_t nt, size_t dim, size_t c0, size_t c1, size_t k, size_t max_iter, float tol,
                 int dist, uint64_t seed, float *cent, const KMeansAssignFn &assign_fn = nullptr);
uint64_t host_splitmix64(uint64_t &s);
// ResultSet::add replay over offers [nq][ldc] (first ncand per row) with set capacity ksel (any size), then the outputs
void pq_resort_finalize(Index &ix, Workspace &ws, const uint64_t *exact_keys, uint32_t ncand, uint32_t ldc, uint64_t nq,
                        uint32_t ksel, uint64_t k, uint64_t id_offset, uint64_t *d_idx, float *d_dist, uint64_t *d_cnt);
void pq_resort_launch(const uint64_t *exact_keys, uint32_t ncand, uint32_t ldc, uint32_t nq, uint32_t k,
                      uint64_t *out, hipStream_t s);

// ---- IVF (index_algorithm/ivf_index.rs) ----
void ivf_attach(Index &ix, uint64_t k_clusters, const float *centroids, const uint64_t *assign);
void ivf_build(Index &ix, uint64_t k_clusters, uint64_t train_n, uint64_t max_iter, float tol, uint64_t seed);
void ivf_clear(Index &ix);
void ivf_export(Index &ix, float *centroids, uint64_t *assign);
void ivf_knn_device(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, uint64_t k, uint64_t n_probes,
                    uint64_t *d_idx, float *d_dist, uint64_t *d_cnt, bool use_half = true, bool use_q8 = true);
void ivf_set_half(int v);
void ivf_set_q8(int v);

// ---- HNSW (index_algorithm/hnsw_index.rs) ----
void hnsw_build(Index &ix, uint64_t M, uint64_t ef_construction, uint64_t seed, uint64_t batch, int nthreads);
void hnsw_attach(Index &ix, uint64_t M, uint64_t ef_construction, const uint32_t *level0, const uint64_t *len0,
                 const uint64_t *vec_level, const uint32_t *upper, const uint64_t *upper_len, int has_enter,
                 uint64_t enter_point, uint64_t enter_level);
void hnsw_clear(Index &ix);
void hnsw_set_dma(int v);
void hnsw_set_half(int v);
void hnsw_set_build_gpu(int v);  // candidate phase of batched builds on the GPU: 0 auto (batch >= 256), 1 off
void hnsw_set_pool_cap(int v);  // test hook: live candidates the fast walk's LDS pool holds before a query moves to the heap walk
void hnsw_insert_rows(Index &ix, const float *rows, uint64_t n);
void hnsw_knn_device(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, uint64_t k, uint64_t ef, bool use_pq,
                     uint64_t *d_idx, float *d_dist, uint64_t *d_cnt);

}  // namespace vdb
