/*
 * vdbhip.h -- C ABI of the MI355X-native distance / top-k engine (libvdbhip.so).
 *
 * This is the drop-in boundary for the one hot path of lab-1806-vec-db:
 * src/distance, src/vec_set, src/index_algorithm::{flat,hnsw,pq}.  The reference has no
 * FFI on this path (everything is generic Rust, monomorphised); the narrowest seam is
 * `DynamicIndex` (src/database/dynamic_index.rs:11-94) which forwards to the index traits
 * (src/index_algorithm/mod.rs:35-154).  Every entry point below names the reference
 * interface it replaces.  A Rust `extern "C"` block binds this 1:1.
 *
 * Conventions
 *  - plain pointers and sizes only; `usize` is uint64_t; DistanceAlgorithm is an int with
 *    the reference's bincode variant order (distance/mod.rs:17-28): 0 = L2Sqr, 1 = Cosine.
 *  - every function returns 0 on success, non-zero on failure; the message is available
 *    through vdb_last_error() (thread-local).  Nothing aborts the process.
 *  - read-side calls (knn*, row, len...) are re-entrant on one handle, like `&self` methods
 *    under the reference's RwLock read guard (database/mod.rs:248-256); write-side calls
 *    (add, swap_remove, *_build, *_attach, *_clear) need external exclusion, like `&mut self`.
 *  - results for query q are written at out_idx[q*k .. q*k+out_count[q]) ascending by
 *    (distance, index) -- the order of `Vec<CandidatePair>` (candidate_pair.rs:36-41);
 *    out_count[q] = min(k, len) for Flat (flat_index.rs:163).
 *  - `*_device` variants take device pointers valid on the index's GPU and a hipStream_t
 *    (passed as void*); they are asynchronous unless noted.
 *  - there is NO CPU fallback: without a usable GPU every compute entry point fails.
 */
#ifndef VDBHIP_H
#define VDBHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VDB_L2SQR 0
#define VDB_COSINE 1

#define VDB_OK 0
#define VDB_ERR_INVALID 1   /* bad argument (reference: bail!/assert!, e.g. database/mod.rs:427-429) */
#define VDB_ERR_HIP 2       /* HIP runtime error */
#define VDB_ERR_STATE 3     /* e.g. knn_pq without a PQ table */
#define VDB_ERR_NOGPU 4     /* no usable gfx950 device */

typedef struct vdb_index vdb_index;

/* thread-local message of the last failing call on this thread */
const char *vdb_last_error(void);
int vdb_version(void);
int vdb_device_count(int *out);

/* ---- VecSet<f32> + DynamicIndex::new ---------------------------------------------------
 * DynamicIndex::new(dim, dist) (dynamic_index.rs:17-19) -> an empty Flat index whose VecSet
 * (vec_set.rs:15-20) lives row-major in HBM on `device_id`. */
int vdb_index_create(int device_id, uint64_t dim, int dist, vdb_index **out);
int vdb_index_destroy(vdb_index *idx);
/* IndexIter::len / dim (index_algorithm/mod.rs:35-52), DynamicIndex::dist (:37-42) */
int vdb_index_len(const vdb_index *idx, uint64_t *out);
int vdb_index_dim(const vdb_index *idx, uint64_t *out);
int vdb_index_dist(const vdb_index *idx, int *out);
/* Index<usize> (vec_set.rs:22-30): copy row i to out[dim] */
int vdb_index_row(const vdb_index *idx, uint64_t i, float *out);
/* DynamicIndex::add / batch_add for Flat = VecSet::push (dynamic_index.rs:44-58, vec_set.rs:113-118);
 * for an index that currently has an HNSW graph = HNSWIndex::add (hnsw_index.rs:538-572) with
 * levels drawn from the index's own RNG stream.  rows is n x dim row-major; *first_id = id of rows[0]. */
int vdb_index_add(vdb_index *idx, const float *rows, uint64_t n, uint64_t *first_id);
/* same, rows already resident in HBM (no host copy is kept until one is needed) */
int vdb_index_add_device(vdb_index *idx, const void *d_rows, uint64_t n, uint64_t *first_id);
/* VecSet::swap_remove (vec_set.rs:131-137); Flat only (metadata_vec_table.rs:163-187) */
int vdb_index_swap_remove(vdb_index *idx, uint64_t i);
/* row sharding: ids reported by knn* are local_row + offset */
int vdb_index_set_id_offset(vdb_index *idx, uint64_t offset);
/* calc_dist (pyo3/mod.rs:43-48): one distance, evaluated on the GPU in reference order */
int vdb_calc_dist(int device_id, const float *a, const float *b, uint64_t n, int dist, float *out);

/* ---- u8 scalar (DistanceScalar for u8, distance/mod.rs:79-95) ---------------------------
 * Every u8 element is converted with `as f32` (exact) before the f32 folds, so a VecSet<u8> index equals the f32
 * index of the converted rows bit for bit.  On an f32 index vdb_index_add_u8 converts and forwards (f32 rows in HBM); on an
 * index made by vdb_index_create_u8 it stores the bytes as they are. */
/* VecSet<u8> index (scalar.rs:117-119): rows stay at ONE byte per element in HBM (a quarter of the f32 bytes); the kernels
 * widen on the fly, the MFMA mirrors hold u8 values exactly.  Serves Flat search (vdb_flat_knn / vdb_flat_knn_u8, swap_remove,
 * row); PQ / HNSW / IVF need an f32 table, as in the reference (DynamicIndex instantiates f32 only, dynamic_index.rs:11-14).
 * Rows are added with vdb_index_add_u8; vdb_index_row returns them widened, vdb_index_row_u8 as stored. */
int vdb_index_create_u8(int device_id, uint64_t dim, int dist, vdb_index **out);
int vdb_index_row_u8(const vdb_index *idx, uint64_t i, uint8_t *out);
int vdb_index_is_u8(const vdb_index *idx, int *out);
int vdb_calc_dist_u8(int device_id, const uint8_t *a, const uint8_t *b, uint64_t n, int dist, float *out);
int vdb_index_add_u8(vdb_index *idx, const uint8_t *rows, uint64_t n, uint64_t *first_id);
int vdb_flat_knn_u8(vdb_index *idx, const uint8_t *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t *out_idx,
                    float *out_dist, uint64_t *out_count);

/* ---- FlatIndex::knn (flat_index.rs:48-57), batched over nq queries --------------------
 * Any k: up to min(k, len) = 1024 results come from the register-resident select, beyond that from a full
 * (distance, index) radix sort per query.  More than 64 queries in one call are served 128 per corpus pass
 * (k_flat_gemm), fewer 2 x 32 per pass (k_flat_mfma); results do not depend on the batching. */
int vdb_flat_knn(vdb_index *idx, const float *queries, uint64_t nq, uint64_t dim, uint64_t k,
                 uint64_t *out_idx, float *out_dist, uint64_t *out_count);
/* device-resident queries and outputs (out_idx u64[nq*k], out_dist f32[nq*k], out_count u64[nq]);
 * synchronises `stream` once at the end (certification read-back). */
int vdb_flat_knn_device(vdb_index *idx, const void *d_queries, uint64_t nq, uint64_t dim, uint64_t k,
                        void *d_out_idx, void *d_out_dist, void *d_out_count, void *stream);
/* The same call in two halves, for hosts that keep several independent query batches in flight (bench.rs:410-418 loops over
 * queries; a serving host loops over batches): _begin enqueues the whole search on one of the index's streams, ordered behind
 * `stream` by an event (no host wait), and returns a pending-call handle; _end waits for it, redoes uncertified queries and
 * frees the handle (also when it reports an error).  With begin(i+1) issued before end(i) the corpus passes of consecutive
 * batches run back to back and the exact stage of one batch overlaps the query preparation of the next.  Every begun call must
 * be ended before the index is destroyed or written to; outputs and query buffers of a pending call must stay untouched. */
typedef struct vdb_pending vdb_pending;
int vdb_flat_knn_device_begin(vdb_index *idx, const void *d_queries, uint64_t nq, uint64_t dim, uint64_t k, void *d_out_idx,
                              void *d_out_dist, void *d_out_count, void *stream, vdb_pending **out);
int vdb_flat_knn_device_end(vdb_pending *pending);
/* the approximate keys the Flat shortlist pass compares with its threshold, for EVERY row, from the same kernel in its dense
 * mode (test / measurement entry point behind the certification-bound tests): out_keys [nq][len];
 * L2Sqr: key = |x|^2 - 2 S~, approximate distance = key + |q|^2;  Cosine: key = -S~ / |x|, approximate distance = 1 + key / |q|.
 * tier 0 = scaled fp16 operands, 1 = split-bf16 operands.  out_qsq [nq] = |q|^2 (strict fold), out_qerr [nq] = measured
 * |q - q~| of the fp16 query images (0 for tier 1), out_dx4 = {max |dx_r|, max |dx_r| / |x_r|, max |x|^2, min positive |x|^2}.
 * nq <= 1024; any of the last three outputs may be NULL. */
int vdb_flat_shortlist_keys(vdb_index *idx, const float *queries, uint64_t nq, uint64_t dim, int tier, float *out_keys,
                            float *out_qsq, float *out_qerr, float *out_dx4);
/* tuning / test hooks for the Flat path:
 *   mode 0 = auto (MFMA shortlist + exact re-rank + certification, exact scan for small inputs),
 *   mode 1 = exact scan only (strict-order f32 fold for every row),
 *   mode 2 = MFMA path forced (still certified, still falls back per query). */
int vdb_flat_set_mode(vdb_index *idx, int mode);
/* developer tuning knobs (kernel variants); results never depend on them.  "flat_half", "flat_half_kmul", "flat_i8",
 * "flat_i8_rows", "flat_gemm", "flat_gemm_debug" and "flat_tail" are per index; ALL OTHER names set process-wide switches (the handle only routes the
 * call) and are not synchronised: set them before concurrent searches start, never while one is running.  Names:
 *   "flat_half"        fp16 first pass of the Flat pipeline: 0 auto (off once > 1/8 of its queries needed the redo), 1 off, 2 on
 *   "flat_half_kmul"   its shortlist = max(64, kmul * k) rows per query (default 4)
 *   "flat_i8"          8-bit first pass in FRONT of it (L2Sqr over f32 rows, k <= 64, dim % 4 == 0 with a 64-column block count divisible by
 *                      2, 3 or 5): a centred int8 mirror at 1 B/element whose keys are lower bounds of the distances; the exact stage walks the
 *                      hit list until the k-th exact distance is below the next bound; what it cannot close goes on to the fp16 pass.
 *                      0 auto (off once > 1/8 of its queries were handed on), 1 off, 2 on.  While it is the first tier the fp16 mirror is
 *                      built by its first use instead of at add time (set "flat_i8" = 1 BEFORE adding rows to have it built at add time)
 *   "flat_i8_rows"     rows its exact stage may walk per query before handing the query on (multiple of 64, default 256)
 *   "flat_gemm_coop"   the same sets for the fp16 / split-bf16 filter kernel (0 auto = on, 1 off)
 *   "flat_gemm8_coop"  cooperative sets of the resident form (calls whose group count is even, tables of >= 98 304 rows, 256-CU chips): the 32
 *                      workgroups of an XCD in sets of 8 / 4 / 2 that take different query groups and walk the same rows at the same time, so
 *                      a row comes from HBM once per set and from the XCD's L2 for the other members; 0 auto = on, 1 off
 *   "flat_gemm8_res"   its kernel keeps the query group's whole 1-B/element image in LDS (dimensions up to 960: no workgroup barrier per chunk;
 *                      0 auto = on when the image fits, 1 off = chunked staging through two buffers)
 *   "flat_gemm8_kc", "flat_gemm8_burst", "flat_gemm8_nt"   variants of its kernel: k-blocks per Q chunk (0 auto, 5 / 3 / 2), staging of
 *                      the next chunk (0 auto, 1 per k-block, 2 one burst per chunk), cache policy of the row stream (as "flat_gemm_nt")
 *   "flat_gemm"        128-queries-per-pass kernel: 0 auto, 1 off (small-batch kernel), 2 forced
 *   "flat_gemm_tw"     row tiles per wave (3 default, 2);  "flat_gemm_stagger" workgroup start delays (0 default)
 *   "flat_gemm_block_rows"  filter pass in row blocks, one launch per block over all query groups (measurement switch): 0 off, n rows
 *   "flat_gemm_nt"     cache policy of the row stream: 0 auto (non-temporal when the mirror exceeds the Infinity Cache), 1 default, 2 non-temporal
 *   "flat_tail"        exact stage: 0 fused launch when the shortlist fits 64 rows, 1 separate kernels
 *   "flat_tail_lb_nw"  form of the 8-bit pass's exact stage (k_flat_tail_lb): 0 auto (= 40), 40 / 41 four waves per query with ONE fold chain
 *                      per lane (products by all four waves, adds by one; row loads 3 / 5 chunks deep), 8 / 4 / 2 / 1 that many waves with the
 *                      chains on the lanes that fetched the rows (process-wide)
 *   "flat_share", "mfma_variant", "flat_sample_thin", "flat_gemm_debug"   small-batch kernel / sample plan / measurement hooks
 *   "pq_adc_fast", "hnsw_dma"   inner-loop variants of the ADC scan and of the HNSW walk (hnsw_dma: 1 rows staged through
 *                      registers (default), 2 through LDS by DMA, 0 plain per-lane loads)
 *   "hnsw_half"        certified half-precision pre-pass of the exact HNSW walk (a row-major fp16 image of the rows, 2 B per
 *                      element, built on first use; rows it cannot rule out are scored exactly as always): 1 auto (default: calls of >= 768
 *                      queries, where the walk is bound by bytes rather than by latency), 0 off, 2 always
 *   "ivf_half"         the same pre-pass for the IVF probe-list scan (offers that cannot be among the k nearest are dropped before the
 *                      f32 rows are fetched; same results): 1 auto (default), 0 off
 *   "ivf_q8"           an 8-bit tier in front of it for long probe lists (1 B/element image with a scale and a measured error per row,
 *                      exact integer dot products; cluster-major for calls of 256 .. 16 384 (query, probe) pairs): 1 auto (default), 0 off, 2 query-major only
 *   "hnsw_build_gpu"   candidate phase of batched HNSW builds (vdb_hnsw_build with batch >= 256): 0 auto = the level-0 searches of a
 *                      batch and the distances between its members run on the GPU (same graph as the all-host builder), 1 off
 *   "hnsw_pool_cap"    most live candidates the fast HNSW walk keeps in LDS (it uses min(this, ef + max_m0 + 64); maximum 2048) before a query is handed to
 *                      the heap walk; tests lower it to exercise that hand-over
 *   "pq_adc16"         quantised first pass of the threshold-filter ADC scan (16-bit tables, 8 queries per pass; exact f32
 *                      sums for its candidates): 0 auto (4-bit codes of any m: L2Sqr 8 and Cosine 7 queries per pass; 8-bit codes, L2Sqr: one
 *                      query per pass on a one-byte table), 1 off
 *   "pq_sample16"      the threshold sample of that scan on the quantised tables as well (L2Sqr): 0 auto (on), 1 off (exact f32 sample).
 *                      The threshold only decides how many rows the scan keeps; the count is checked and short lists are redone
 *   "flat_small"       FlatIndex::knn of a few queries over a small table in ONE launch (the db.search() shape; dim % 4 == 0, k <= 64):
 *                      0 auto (flat mode 0; tables of at most "flat_small_max_rows" = 16 384 rows: calls of fewer than 32 queries; larger
 *                      tables: while rows x (0.9 queries - 0.3) < 75 000, i.e. one query up to ~125k rows, where the MFMA pipeline's
 *                      fixed ~0.1 ms of dependent launches costs more than re-reading the rows per query), 1 off, 2 whenever the shape allows */
int vdb_set_param(vdb_index *idx, const char *name, int64_t value);
/* Builds NOW the operand mirrors the first Flat search of this index would otherwise build inside that search (the centred 8-bit
 * mirror, or the fp16 / split-bf16 mirror where the 8-bit pass does not apply); all_tiers != 0: the mirrors of the tiers behind it too
 * (what a query the first pass cannot certify needs).  A latency-sensitive caller calls it after add / batch_add
 * (metadata_vec_table.rs:63-86 has no counterpart: the reference's Flat index has nothing to build).  A mirror that does not fit the
 * device memory is skipped -- searches then use the next tier, down to the exact scan over the rows -- and is not an error. */
int vdb_index_prepare(vdb_index *idx, int all_tiers);
/* number of queries whose MFMA shortlist failed certification and were redone by the exact scan */
int vdb_flat_fallback_count(const vdb_index *idx, uint64_t *out);
/* counters of the Flat pipeline (diagnostics; results never depend on them):
 *   "flat_fallback"      = vdb_flat_fallback_count,
 *   "flat_half_queries"  queries that went through the fp16 first pass,
 *   "flat_half_redo"     of those, the ones it could not certify (redone with the split-bf16 pass),
 *   "pq_adc16_queries"   queries whose ADC scan ran on the quantised 16-bit tables (k_pq_adc16) since the table was attached,
 *   "flat_half_valid"    1 when the index holds the fp16 mirror,
 *   "flat_i8_queries", "flat_i8_redo", "flat_i8_valid"  the same three for the 8-bit first pass (queries through it, queries it handed
 *                        on to the next tier, mirror present),
 *   "flat_bf16_mirror"   1 once the split-bf16 mirror (4 B/element) has been built -- lazily, by the first search that
 *                        needs it (redo tier, flat_half = 1, calls without an fp16 mirror),
 *   "hnsw_heap_walk_queries"  HNSW queries answered by the any-size heap walk (max(ef, k) > 1024, or the LDS candidate
 *                        pool of the fast walk overflowed),
 *   "hnsw_half_dropped"  of the last HNSW call's distance evaluations (vdb_hnsw_last_stats), the rows its certified
 *                        half-precision pre-pass ruled out without fetching the f32 row,
 *   "ivf_last_offers", "ivf_last_kept_q8", "ivf_last_kept"  (while vdb_prof_enable is on) rows the last IVF call offered to its result
 *                        sets, the ones the 8-bit tier passed on (0: tier not run) and the ones that reached the exact stage; "ivf_last_rows_fetched_q8":
 *                        rows the cluster-major 8-bit tier read, once each (0: query-major),
 *   "hbm_bytes_per_row"  resident HBM bytes per row over all per-row buffers (rows, norms, mirrors, PQ codes, level-0 links). */
int vdb_get_stat(const vdb_index *idx, const char *name, uint64_t *out);

/* ---- PQTable (distance/pq_table.rs) ----------------------------------------------------
 * centroids: group g, centroid c at k_c*gstart[g] + c*len(g) (pq_groups, pq_table.rs:38-53),
 * k_c = 1<<n_bits, total k_c*dim floats.  codes: n x ceil(m*n_bits/8), 4-bit low nibble = even
 * group (pq_table.rs:66-91); NULL -> encoded on the GPU (pq_encode + find_nearest_base). */
int vdb_pq_attach(vdb_index *idx, uint64_t n_bits, uint64_t m, const float *centroids, const uint8_t *codes);
/* PQTable::from_vec_set (pq_table.rs:141-191): sample train_n rows (0 = all), per-group k-means
 * (k_means.rs:61-162) on the host, encode on the GPU.  RNG = splitmix64(seed). */
int vdb_pq_build(vdb_index *idx, uint64_t n_bits, uint64_t m, uint64_t train_n, uint64_t max_iter, float tol,
                 uint64_t seed);
int vdb_pq_clear(vdb_index *idx);                 /* MetadataVecTable::clear_pq_table :154-156 */
int vdb_pq_has(const vdb_index *idx, int *out);
int vdb_pq_info(const vdb_index *idx, uint64_t *n_bits, uint64_t *m, uint64_t *enc_dim);
int vdb_pq_export(const vdb_index *idx, float *centroids, uint8_t *codes);
/* PQTable::create_lookup (pq_table.rs:195-224) for nq queries, as the search kernels build it: out_lut [nq][m * k_c]
 * row-major [group][centroid] (l2 of the query slice to the centroid for L2Sqr, their dot product for Cosine),
 * out_qcache [nq] = PQLookupTable::dist_cache (0 for L2Sqr, |q| for Cosine).  Either output may be NULL. */
int vdb_pq_create_lookup(vdb_index *idx, const float *queries, uint64_t nq, uint64_t dim, float *out_lut, float *out_qcache);
/* the ADC adapter DistanceAdapter<[u8], PQLookupTable> (pq_table.rs:239-301) of every code row: out [nq][len], the values
 * the knn_pq scan ranks (strict group-order f32 sums; Cosine: 1 - sum / max(sqrt(sum cent_cache) * |q|, 1e-10)) */
int vdb_pq_adc_all(vdb_index *idx, const float *queries, uint64_t nq, uint64_t dim, float *out);
/* FlatIndex::knn_pq (flat_index.rs:84-104) */
int vdb_flat_knn_pq(vdb_index *idx, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef,
                    uint64_t *out_idx, float *out_dist, uint64_t *out_count);

/* device-pointer variant (queries / outputs on the index's GPU); returns synchronised */
int vdb_flat_knn_pq_device(vdb_index *idx, const void *d_queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef,
                           void *d_out_idx, void *d_out_dist, void *d_out_count, void *stream);

/* ---- HNSWIndex (index_algorithm/hnsw_index.rs) -----------------------------------------
 * Graph layout (fields :98-141): level0 = n x max_m0 u32 (max_m0 = 2*min(M,10000)), len0[n];
 * vec_level[n]; upper = CSR over nodes, node v level L>=1 at ((sum_{u<v} vec_level[u]) + L-1)*m,
 * upper_len likewise. */
int vdb_hnsw_build(vdb_index *idx, uint64_t M, uint64_t ef_construction, uint64_t seed, uint64_t batch,
                   int nthreads);                  /* build_on_vec_set :595-611 (host builder) */
int vdb_hnsw_attach(vdb_index *idx, uint64_t M, uint64_t ef_construction, const uint32_t *level0,
                    const uint64_t *len0, const uint64_t *vec_level, const uint32_t *upper,
                    const uint64_t *upper_len, int has_enter, uint64_t enter_point, uint64_t enter_level);
int vdb_hnsw_clear(vdb_index *idx);               /* MetadataVecTable::clear_hnsw_index :100-106 */
int vdb_hnsw_has(const vdb_index *idx, int *out);
int vdb_hnsw_info(const vdb_index *idx, uint64_t *m, uint64_t *max_m0, uint64_t *upper_total, int *has_enter,
                  uint64_t *enter_point, uint64_t *enter_level, uint64_t *default_ef);
int vdb_hnsw_export(const vdb_index *idx, uint32_t *level0, uint64_t *len0, uint64_t *vec_level, uint32_t *upper,
                    uint64_t *upper_len);
/* HNSWIndex::knn_with_ef (:619-634); ef = 0 -> default_ef (knn, :614-618).
 * A walk is one wavefront and the chip keeps 2048 of them resident (8 per CU): throughput is flat from ~4096 queries per call
 * on (584k QPS at 1M x 960, ef = 128), a call of ~1000 queries is one round of walks and bounded by the longest of them
 * (2.6 ms), smaller calls take the time of one walk.  Any k / ef (beyond 1024 the heap walk answers). */
int vdb_hnsw_knn(vdb_index *idx, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef,
                 uint64_t *out_idx, float *out_dist, uint64_t *out_count);
/* HNSWIndex::knn_pq (:672-697) */
int vdb_hnsw_knn_pq(vdb_index *idx, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef,
                    uint64_t *out_idx, float *out_dist, uint64_t *out_count);
/* device-pointer variant of knn_with_ef (use_pq = 0) / knn_pq (use_pq = 1); returns synchronised */
int vdb_hnsw_knn_device(vdb_index *idx, const void *d_queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef,
                        int use_pq, void *d_out_idx, void *d_out_dist, void *d_out_count, void *stream);
/* per-call work counters of the last vdb_hnsw_knn on this index (SURVEY 8d bytes/query) */
int vdb_hnsw_last_stats(const vdb_index *idx, uint64_t *n_dist, uint64_t *n_expanded);

/* ---- shard merge ------------------------------------------------------------
 * Merge S per-shard result lists (each [nq][k], ascending, counts[s][q] valid entries, global ids)
 * into the global top-k by (distance, index).  Host utility used after the RCCL all-gather. */
int vdb_merge_topk(const float *dists, const uint64_t *ids, const uint64_t *counts, uint64_t n_shards,
                   uint64_t nq, uint64_t k, uint64_t *out_idx, float *out_dist, uint64_t *out_count);

/* same merge on the index's GPU (inputs = the all-gathered tensors, ids < 2^32); returns synchronised */
int vdb_merge_topk_device(vdb_index *idx, const void *d_dists, const void *d_ids, const void *d_counts,
                          uint64_t n_shards, uint64_t nq, uint64_t k, void *d_out_idx, void *d_out_dist,
                          void *d_out_count, void *stream);

/* same merge reading the S per-rank blocks of ONE all-gather buffer in place (block s at s*block_bytes; ids, distances
 * and counts at the given byte offsets inside a block): no repacking between the collective and the merge */
int vdb_merge_topk_gathered(vdb_index *idx, const void *d_gathered, uint64_t block_bytes, uint64_t off_ids,
                            uint64_t off_dists, uint64_t off_counts, uint64_t n_shards, uint64_t nq, uint64_t k,
                            void *d_out_idx, void *d_out_dist, void *d_out_count, void *stream);

/* the same, ENQUEUED on `stream` (the stream the all-gather was issued on) without any host synchronisation, k <= 64: the
 * exchange of one step can then run under the next step's search (outputs are valid once `stream` has passed this point) */
int vdb_merge_topk_gathered_async(vdb_index *idx, const void *d_gathered, uint64_t block_bytes, uint64_t off_ids,
                                  uint64_t off_dists, uint64_t off_counts, uint64_t n_shards, uint64_t nq, uint64_t k,
                                  void *d_out_idx, void *d_out_dist, void *d_out_count, void *stream);

/* ---- IVFIndex (index_algorithm/ivf_index.rs) ------------------------------
 * from_vec_set (:66-118): k-means over all columns on train_n sampled rows (0 = all; host, RNG = splitmix64(seed),
 * parity unpinned), then every row joins its nearest centroid (k_means.rs:40-57: CandidatePair order) -- computed on
 * the GPU in reference arithmetic, bit-exact given the centroids.  Writes to the index (add) drop the clusters. */
int vdb_ivf_build(vdb_index *idx, uint64_t k_clusters, uint64_t train_n, uint64_t max_iter, float tol, uint64_t seed);
/* centroids: k_clusters x dim; assign: cluster of every row (n values) or NULL -> computed as above */
int vdb_ivf_attach(vdb_index *idx, uint64_t k_clusters, const float *centroids, const uint64_t *assign);
int vdb_ivf_clear(vdb_index *idx);
int vdb_ivf_info(const vdb_index *idx, int *present, uint64_t *k_clusters, uint64_t *default_n_probes);
int vdb_ivf_export(vdb_index *idx, float *centroids, uint64_t *assign);
/* IVFIndex::knn_with_ef (:143-154), ef = n_probes (0 -> default_n_probes = 4, :108): probes by find_n_nearest
 * (k_means.rs:174-190), then ResultSet::add over the probed clusters in probe order, rows ascending */
int vdb_ivf_knn(vdb_index *idx, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t n_probes,
                uint64_t *out_idx, float *out_dist, uint64_t *out_count);

/* device-pointer variant; returns synchronised */
int vdb_ivf_knn_device(vdb_index *idx, const void *d_queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t n_probes,
                       void *d_out_idx, void *d_out_dist, void *d_out_count, void *stream);

/* ---- row-sharded FlatIndex::knn_pq -------------------------------------------
 * pq_resort (candidate_pair.rs:102-108) replays ResultSet::add in the GLOBAL (ADC distance, id) order, so the
 * exchange carries, per shard and query, max(ef,k) pair keys twice: the ADC key row (ascending) and the exact-distance
 * key of the same row at the same position.  A pair key is orderable(f32) << 32 | global row id (id_offset + local
 * row, < 2^32); ~0 pads rows shorter than max(ef,k).  Layout [nq][max(ef,k)] u64 each. */
int vdb_flat_knn_pq_shard(vdb_index *idx, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef,
                          uint64_t *out_adc_keys, uint64_t *out_exact_keys);
int vdb_flat_knn_pq_shard_device(vdb_index *idx, const void *d_queries, uint64_t nq, uint64_t dim, uint64_t k,
                                 uint64_t ef, void *d_out_adc_keys, void *d_out_exact_keys, void *stream);
/* merge of the all-gathered rows ([n_shards][nq][efk], efk = max(ef,k)) to the global ADC top-efk, then the
 * reference's re-sort (flat_index.rs:100-103): out [nq][k] ascending exact distances, global ids.  Host utility. */
int vdb_pq_merge_resort(const uint64_t *adc_keys, const uint64_t *exact_keys, uint64_t n_shards, uint64_t nq,
                        uint64_t efk, uint64_t k, uint64_t *out_idx, float *out_dist, uint64_t *out_count);
/* same on the index's GPU; returns synchronised */
int vdb_pq_merge_resort_device(vdb_index *idx, const void *d_adc_keys, const void *d_exact_keys, uint64_t n_shards,
                               uint64_t nq, uint64_t efk, uint64_t k, void *d_out_idx, void *d_out_dist,
                               void *d_out_count, void *stream);

/* ---- multi-GPU context (vdb_ctx_create; SURVEY 8e: row shards + one RCCL all-gather) ------------------------
 * The library owns the RCCL communicator and the exchange: a host needs no collective library of its own.
 *   vdb_ctx_create        ONE process drives n_dev GPUs (ncclCommInitAll): the layout of a Rust host, whose DynamicIndex
 *                         (dynamic_index.rs:11-94) lives in one process;
 *   vdb_ctx_create_rank   one process per GPU (ncclCommInitRank): rank 0 obtains a 128-byte id from vdb_ctx_unique_id, the
 *                         host distributes it (MPI, a file, torch.distributed's store ...), every rank passes it in.
 * RCCL is loaded at run time (dlopen) by the first context that needs a communicator (world > 1). */
typedef struct vdb_ctx vdb_ctx;
typedef struct vdb_sharded vdb_sharded;
int vdb_ctx_create(const int *device_ids, int n_dev, vdb_ctx **out);
int vdb_ctx_unique_id(void *out_id, uint64_t out_bytes /* >= 128 */);
int vdb_ctx_create_rank(int device_id, const void *id, int rank, int world, vdb_ctx **out);
int vdb_ctx_destroy(vdb_ctx *ctx);
int vdb_ctx_info(const vdb_ctx *ctx, int *world, int *n_local, int *first_rank, int *has_comm);
/* DynamicIndex::new(dim, dist) over all GPUs of the context: rank r of S holds rows [r * ceil(N/S), (r+1) * ceil(N/S))
 * of the corpus handed to vdb_sharded_set_rows (every process passes the same N x dim view) and reports global ids. */
int vdb_sharded_create(vdb_ctx *ctx, uint64_t dim, int dist, vdb_sharded **out);
int vdb_sharded_destroy(vdb_sharded *sh);
int vdb_sharded_set_rows(vdb_sharded *sh, const float *rows, uint64_t n_total);
int vdb_sharded_len(const vdb_sharded *sh, uint64_t *out);
/* borrowed handle of this process's i-th shard (statistics, tuning switches, exports); owned by the sharded index */
int vdb_sharded_local(vdb_sharded *sh, int i, vdb_index **out);
/* FlatIndex::knn (flat_index.rs:48-57) over the whole corpus: per-shard top-k, ONE all-gather of [nq,k] ids / distances /
 * counts per rank, exact merge by (distance, index) -- equal to the unsharded answer.  k <= 1024. */
int vdb_sharded_flat_knn(vdb_sharded *sh, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t *out_idx,
                         float *out_dist, uint64_t *out_count);
/* PQTable replicated on every shard (same centroids; codes encoded per shard on its GPU) and FlatIndex::knn_pq
 * (flat_index.rs:84-104) over the whole corpus: ADC key rows + exact key rows gathered, merged in (adc, idx) order, then
 * pq_resort (candidate_pair.rs:102-108) -- equal to the unsharded answer. */
int vdb_sharded_pq_attach(vdb_sharded *sh, uint64_t n_bits, uint64_t m, const float *centroids);
int vdb_sharded_knn_pq(vdb_sharded *sh, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef,
                       uint64_t *out_idx, float *out_dist, uint64_t *out_count);
/* REPLICA layout ("HNSW: replicas only"): every GPU of the context keeps ALL rows (ids are global as they are);
 * a search splits the QUERIES instead -- rank r of S answers the block [r * ceil(nq/S), min(nq, (r+1) * ceil(nq/S)))
 * (vdb_replica_query_block) and ONE fixed-size all-gather concatenates the blocks on every rank; no merge.  All of
 * DynamicIndex's searches (dynamic_index.rs:68-93) run in this layout: vdb_sharded_flat_knn and vdb_sharded_knn_pq accept it
 * as well as the row layout (any k), the HNSW searches below exist only here (a graph's edges cross any row partition).
 * vdb_sharded_layout: 0 = no rows yet, 1 = row blocks (vdb_sharded_set_rows), 2 = replicas. */
int vdb_sharded_set_rows_replica(vdb_sharded *sh, const float *rows, uint64_t n_total);
int vdb_sharded_layout(const vdb_sharded *sh, int *out);
int vdb_replica_query_block(uint64_t nq, uint64_t world, uint64_t rank, uint64_t *q0, uint64_t *q1);
/* HNSWIndex over the replicas: built once per process on its first GPU (hnsw_index.rs:391-457 through vdb_hnsw_build's host
 * builder; deterministic for a given seed / batch / thread count, so the processes of a multi-process job hold equal graphs)
 * or attached from arrays (vdb_hnsw_attach's layout), then mirrored to the process's other GPUs. */
int vdb_sharded_hnsw_build(vdb_sharded *sh, uint64_t M, uint64_t ef_construction, uint64_t seed, uint64_t batch, int nthreads);
int vdb_sharded_hnsw_attach(vdb_sharded *sh, uint64_t M, uint64_t ef_construction, const uint32_t *level0, const uint64_t *len0,
                            const uint64_t *vec_level, const uint32_t *upper, const uint64_t *upper_len, int has_enter,
                            uint64_t enter_point, uint64_t enter_level);
/* HNSWIndex::knn_with_ef (hnsw_index.rs:619-634; ef = 0 -> the graph's default) / knn_pq (:672-697) with the queries split
 * over the replicas -- equal to the single-GPU answer row by row. */
int vdb_sharded_hnsw_knn(vdb_sharded *sh, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef, uint64_t *out_idx,
                         float *out_dist, uint64_t *out_count);
int vdb_sharded_hnsw_knn_pq(vdb_sharded *sh, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef, uint64_t *out_idx,
                            float *out_dist, uint64_t *out_count);
/* IVFIndex over the replicas (ivf_index.rs:66-154): clusters built once per process (seeded, so equal in every process) and
 * mirrored to its other GPUs; knn_with_ef(ef = n_probes; 0 -> the default 4) with the queries split. */
int vdb_sharded_ivf_build(vdb_sharded *sh, uint64_t k_clusters, uint64_t train_n, uint64_t max_iter, float tol, uint64_t seed);
int vdb_sharded_ivf_knn(vdb_sharded *sh, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t n_probes, uint64_t *out_idx,
                        float *out_dist, uint64_t *out_count);
/* Failed sharded calls.  vdb_sharded_set_rows[_replica] is all-or-nothing: when one GPU fails (out of memory, say) every
 * shard is rolled back to an empty index and the call may be repeated.  A SEARCH that fails on one rank of a multi-PROCESS
 * context (vdb_ctx_create_rank, world > 1) may leave the other ranks inside the all-gather; the object is then "poisoned"
 * (vdb_sharded_poisoned -> 1) and refuses further searches with VDB_ERR_STATE-style errors: destroy it on every rank.
 * Exercised on hardware with ONE rank only (the test boxes have one GPU): the N > 1 paths -- ncclCommInitAll, grouped
 * all-gathers over several local GPUs, cross-process ordering -- are correct by construction, not by test. */
int vdb_sharded_poisoned(const vdb_sharded *sh, int *out);

/* ---- measurement hooks -------------------------------------------------------------------
 * When enabled, the dominant kernels are bracketed by HIP events on their own stream and the
 * elapsed time is accumulated per kernel name ("flat_mfma", "flat_exact", "pq_adc", "hnsw"). */
int vdb_prof_enable(vdb_index *idx, int on);
/* attainable HBM read bandwidth of this box: a pure streaming read of `bytes` (> the 256-MB Infinity Cache)
 * repeated `iters` times, best of two access patterns, in GB/s (1e9 B/s).  Allocates and frees its own buffer. */
int vdb_stream_probe(int device_id, uint64_t bytes, int iters, double *out_gbps);
/* the same bytes read the way the Flat filter would have to read them if its fp16 operand were the ROW-MAJOR image the graph
 * walks gather from (one fp16 copy of the rows instead of two): MFMA A-fragment loads, 16 rows x 64 B per instruction, rows of
 * row_bytes (a multiple of 128, e.g. 1920 = a 960-d fp16 row).  The A/B behind DESIGN's "two fp16 images" note.  Measurement hook. */
int vdb_stream_probe_rows(int device_id, uint64_t bytes, int iters, uint32_t row_bytes, double *out_gbps);
/* matrix-pipe rate of this box under sustained load: v_mfma_f32_16x16x32_f16 (the Flat filter's instruction) issued back to back
 * by `waves_per_simd` waves on every SIMD, `iters` x 8 independent tiles per wave; dense TFLOP/s of the launch and the shader
 * clock (GHz) the chip held meanwhile.  Measurement hook. */
int vdb_mfma_probe(int device_id, int waves_per_simd, int iters, double *out_tflops, double *out_clock_ghz);
/* the same for v_mfma_i32_16x16x64_i8, the instruction of the 8-bit Flat filter (k_gemm8.hip): dense integer TOP/s.  The filter's
 * matrix-pipe roofline is quoted against the chip's nominal int8 peak AND against this sustained rate.  Measurement hook. */
int vdb_mfma_probe_i8(int device_id, int waves_per_simd, int iters, double *out_tops, double *out_clock_ghz);
/* latency of one DEPENDENT HBM access on this box: a single lane follows a random cycle over the 128-B lines of a `bytes`-sized
 * buffer (larger than L2 and the Infinity Cache) for `hops` loads; nanoseconds per load.  The floor of the graph walks
 * (hnsw_index.rs:258-291 is a chain of dependent accesses per expansion) is quoted on it.  Measurement hook. */
int vdb_latency_probe(int device_id, uint64_t bytes, uint32_t hops, double *out_ns_per_load);
/* latency of one DEPENDENT f32 add: the reference's distances are strict left folds (distance/mod.rs:72-77), so a d-column row
 * is a chain of d of them on whatever hardware; nanoseconds per add over a chain of `adds`.  Measurement hook. */
int vdb_fold_probe(int device_id, uint32_t adds, double *out_ns_per_add);
int vdb_prof_reset(vdb_index *idx);
int vdb_prof_get(vdb_index *idx, const char *kernel, double *total_ms, uint64_t *launches, double *bytes);

#ifdef __cplusplus
}
#endif
#endif /* VDBHIP_H */
