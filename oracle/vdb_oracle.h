/*
 * vdb_oracle.h -- CPU restatement of the lab-1806-vec-db distance / top-k hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under lab_1806_vec_db_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / CPU baseline.
 *
 * Parity status: the reference is 100 % Rust and cannot be compiled or imported
 * in the build container (no cargo/rustc, no wheel, no network).  This oracle is
 * pinned by (i) every known-answer / property test the reference holds for this
 * path, restated in tests/test_oracle_kat.py, and (ii) an
 * independent numpy emulation of the strict f32 left fold (oracle/np_ref.py) on
 * the reference's own data fixtures (tests/golden/).  RNG-dependent artefacts
 * (k-means centroids, HNSW levels/graphs) are "parity unpinned": the reference
 * uses ChaCha12 StdRng streams that no reference test pins; they are inputs here.
 *
 * All citations are file:line under /root/reference/src.
 */
#ifndef VDB_ORACLE_H
#define VDB_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_L2SQR = 0, ORC_COSINE = 1 }; /* distance/mod.rs:17-28 (bincode variant order) */

/* ---- distance/mod.rs ------------------------------------------------- */
float orc_dot(const float *a, const float *b, size_t n);                 /* :72-74 */
float orc_l2(const float *a, const float *b, size_t n);                  /* :75-77 */
float orc_norm(const float *a, size_t n);                                /* :46-48 */
float orc_cosine(const float *a, const float *b, size_t n);              /* :60-64 */
float orc_dist(int dist, const float *a, const float *b, size_t n);      /* :106-113 */
float orc_dist_cache(int dist, const float *a, size_t n);                /* :31-36 */
float orc_dist_cached(int dist, const float *a, const float *b, size_t n,
                      float cache_a, float cache_b);                     /* :54-57,66-69,120-129 */
float orc_dot_u8(const uint8_t *a, const uint8_t *b, size_t n);          /* :79-85 */
float orc_l2_u8(const uint8_t *a, const uint8_t *b, size_t n);           /* :86-94 */
float orc_dist_u8(int dist, const uint8_t *a, const uint8_t *b, size_t n);

/* ---- candidate_pair.rs ----------------------------------------------- */
/* total order on (OrderedFloat<f32>, usize): <0, 0, >0        :36-41 */
int orc_pair_cmp(float da, uint64_t ia, float db, uint64_t ib);
/* recall@k of one row                                         :127-140 */
float orc_recall(const uint64_t *gt, size_t n_gt, const uint64_t *pred, size_t n_pred);

/* ---- flat_index.rs ---------------------------------------------------- */
/* FlatIndex::knn :48-57.  Returns number of results = min(k, n). */
size_t orc_flat_knn(const float *base, size_t n, size_t dim, int dist,
                    const float *query, size_t k, uint64_t *out_idx, float *out_dist);
/* same, `nthreads` queries in flight (mirrors rayon par_iter over queries,
 * examples/bench.rs:414-416, src/bin/gen_gnd.rs:65-68).  out_* are [nq][k]. */
void orc_flat_knn_batch(const float *base, size_t n, size_t dim, int dist,
                        const float *queries, size_t nq, size_t k,
                        uint64_t *out_idx, float *out_dist, uint64_t *out_count, int nthreads);

/* ---- pq_table.rs ------------------------------------------------------ */
typedef struct {
    uint64_t dim, m, n_bits, k, enc_dim; /* pq_table.rs:116-137 */
    int dist;
    uint64_t *gstart;     /* m+1 group boundaries, pq_groups :38-53 */
    float *centroids;     /* group g, centroid c at k*gstart[g] + c*(gstart[g+1]-gstart[g]) */
    float *cent_cache;    /* m*k: 0 (L2) or dot(c,c) (Cosine)          :160-165 */
    uint8_t *codes;       /* n * enc_dim                                :178-181 */
    uint64_t n;
} orc_pq;

size_t orc_pq_groups(size_t dim, size_t m, uint64_t *gstart /* m+1 */);
orc_pq *orc_pq_new(size_t dim, size_t m, size_t n_bits, int dist, const float *centroids);
void orc_pq_free(orc_pq *pq);
/* pq_encode :66-91 + find_nearest_base k_means.rs:40-57; appends nothing, pure */
void orc_pq_encode_row(const orc_pq *pq, const float *v, uint8_t *out);
void orc_pq_encode_all(orc_pq *pq, const float *base, size_t n); /* fills pq->codes */
void orc_pq_set_codes(orc_pq *pq, const uint8_t *codes, size_t n);
/* create_lookup :195-224.  lut is m*k floats; returns dist_cache (0 or |q|) */
float orc_pq_lookup(const orc_pq *pq, const float *query, float *lut);
/* ADC :239-301 */
float orc_pq_adc(const orc_pq *pq, const uint8_t *code, const float *lut, float q_cache);
/* FlatIndex::knn_pq flat_index.rs:84-104 */
void orc_pq_adc_all(const orc_pq *pq, size_t n, const float *query, float *out);
size_t orc_flat_knn_pq(const float *base, size_t n, size_t dim, int dist, const orc_pq *pq,
                       const float *query, size_t k, size_t ef, uint64_t *out_idx, float *out_dist);
/* k-means (k_means.rs:61-162) with an explicit splitmix64 stream -- RNG unpinned.
 * rows: n x dim, uses columns [c0,c1).  out: k x (c1-c0). */
void orc_kmeans(const float *rows, size_t n, size_t dim, size_t c0, size_t c1, size_t k,
                size_t max_iter, float tol, int dist, uint64_t *rng_state, float *out);
/* PQTable::from_vec_set :141-191 (sampling by partial Fisher-Yates on the same stream) */
orc_pq *orc_pq_train(const float *base, size_t n, size_t dim, size_t m, size_t n_bits, int dist,
                     size_t k_means_size /*0 = all*/, size_t max_iter, float tol, uint64_t seed);

/* ---- ivf_index.rs ------------------------------------------------------ */
void orc_assign_nearest(const float *base, size_t n, size_t dim, int dist, const float *cents, size_t k, uint64_t *out);
size_t orc_ivf_knn(const float *base, size_t dim, int dist, const float *cents, size_t k_clusters,
                   const uint64_t *offsets, const uint64_t *members, const float *query, size_t k, size_t n_probes,
                   uint64_t *out_idx, float *out_dist);

/* ---- hnsw_index.rs ---------------------------------------------------- */
typedef struct {
    uint64_t dim, m, max_m0, ef_construction, default_ef; /* :75-96, :493-506 */
    float inv_log_m;
    int dist;
    uint64_t n, cap;
    float *rows;          /* n x dim (owned copy, :244-245) */
    float *cache;         /* dist_cache :251-254 */
    uint32_t *level0;     /* n x max_m0 :112 */
    uint64_t *len0;       /* links_len[v][0] */
    uint64_t *vec_level;  /* :131 */
    uint64_t *upper_off;  /* number of upper levels of all nodes before v (CSR) */
    uint32_t *upper;      /* (upper_off[v] + L-1)*m + j */
    uint64_t *upper_len;  /* upper_off[v] + L-1 */
    uint64_t upper_cap, upper_total;
    int has_enter;
    uint64_t enter_point, enter_level;
    /* instrumentation for SURVEY 8(d) bytes/query */
    uint64_t stat_n_dist, stat_n_expanded;
} orc_hnsw;

orc_hnsw *orc_hnsw_new(size_t dim, int dist, size_t M, size_t ef_construction); /* :493-536 */
void orc_hnsw_free(orc_hnsw *h);
/* HNSWIndex::add :538-572 with an explicit level (rand_level :144-147 is RNG-dependent) */
uint64_t orc_hnsw_add(orc_hnsw *h, const float *vec, uint64_t level);
/* add_parallel :399-457 executed serially with an explicit batch; levels[nb] */
void orc_hnsw_add_batch(orc_hnsw *h, const float *vecs, size_t nb, const uint64_t *levels);
/* level from a uniform in (0,1): floor(-ln(u)*inv_log_m) in f32 :144-147 */
uint64_t orc_hnsw_level_from_uniform(const orc_hnsw *h, float u);
/* build_on_vec_set :595-611 with levels drawn from splitmix64(seed); batch = explicit
 * batch size used once n >= 1000 (:391-397), 1 = fully serial */
orc_hnsw *orc_hnsw_build(const float *base, size_t n, size_t dim, int dist, size_t M,
                         size_t ef_construction, uint64_t seed, size_t batch);
/* attach an externally built graph (same CSR as the C ABI) */
orc_hnsw *orc_hnsw_from_graph(const float *base, size_t n, size_t dim, int dist, size_t M,
                              size_t ef_construction, const uint32_t *level0, const uint64_t *len0,
                              const uint64_t *vec_level, const uint32_t *upper,
                              const uint64_t *upper_len, int has_enter, uint64_t enter_point,
                              uint64_t enter_level);
/* knn_with_ef :619-634 */
size_t orc_hnsw_knn(orc_hnsw *h, const float *query, size_t k, size_t ef,
                    uint64_t *out_idx, float *out_dist);
/* knn_pq :672-697 */
size_t orc_hnsw_knn_pq(orc_hnsw *h, const orc_pq *pq, const float *query, size_t k, size_t ef,
                       uint64_t *out_idx, float *out_dist);
void orc_hnsw_knn_batch(orc_hnsw *h, const float *queries, size_t nq, size_t k, size_t ef,
                        uint64_t *out_idx, float *out_dist, uint64_t *out_count, int nthreads,
                        uint64_t *stat_n_dist, uint64_t *stat_n_expanded);

/* splitmix64 (own stream; the reference's ChaCha12 StdRng is not reproduced) */
uint64_t orc_splitmix64(uint64_t *state);
float orc_uniform_open01(uint64_t *state);

#ifdef __cplusplus
}
#endif
#endif
