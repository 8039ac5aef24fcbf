// kernels.hpp -- launch wrappers of the gfx950 kernels (definitions in k_*.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace vdb {

// how a (row, query) pair turns into a distance; all folds are strict left-to-right f32
enum Metric : int {
    MET_L2_DIRECT = 0,  // sum (x-q)^2                     distance/mod.rs:75-77   (Flat, re-sorts)
    MET_COSINE = 1,     // 1 - dot/max(|x||q|,1e-10)       distance/mod.rs:60-69
    MET_L2_CACHED = 2,  // (xx + qq) - 2*dot(x,q)          distance/mod.rs:54-57   (HNSW)
};

// ---- k_exact.hip ---------------------------------------------------------------------------
// sq[i] = dot(x_i, x_i) in reference order (DistanceAlgorithm::dist_cache, distance/mod.rs:31-36)
void launch_row_sqnorm(const float *X, uint64_t n, uint32_t dim, float *sq, hipStream_t s);
// dense exact distances out[b*ld + i] for nq (<= 8) queries against rows [0,n)
void launch_scan_exact(const float *X, uint64_t n, uint32_t dim, const float *Q, uint32_t nq, int metric,
                       const float *xsq, const float *qsq, float *out, uint64_t ld, bool use_lds, hipStream_t s);
// exact distance of every candidate: in/out are pair keys [nq][ldc]; PAIR_NONE entries pass through
void launch_rerank(const float *X, uint32_t dim, const float *Q, uint32_t nq, int metric, const float *xsq,
                   const float *qsq, const uint64_t *cand, uint64_t *out, uint32_t ncand, uint32_t ldc,
                   hipStream_t s, const uint32_t *cnt = nullptr);  // cnt: counted lists (first cnt[q] slots of a row)
// first `ksel` pair keys of each query -> (u64 id + id_offset, f32 distance) at out[q*kstride + j],
// out_count[q] = number of valid pairs
void launch_finalize(const uint64_t *keys, uint32_t ldk, uint32_t nq, uint32_t ksel, uint32_t kstride,
                     uint64_t id_offset, uint64_t *out_idx, float *out_dist, uint64_t *out_count, hipStream_t s);
// certification of an MFMA shortlist (see k_exact.hip): flags[q] = 1 when the exact top-k might
// not be contained in the shortlist
void launch_iota_keys(uint64_t *rows, uint32_t nq, uint32_t n, uint32_t ld, hipStream_t s);
// rounding of the shortlist keys' operands: qerr == nullptr -> split-bf16 (format constants); otherwise the measured
// errors of the fp16 operands (k_half.hip): dx_abs = max |dx_r|, dx_rel = max |dx_r| / |x_r|, qerr[q] = |dq|
struct SplitErr {
    const float *qerr = nullptr;
    float dx_abs = 0.0f, dx_rel = 0.0f;
    // 8-bit pass (k_gemm8.hip / k_i8.hip): the keys are lower bounds, D(r, q) >= key + qoff[q]; |mu| of the centring vector
    const float *qoff = nullptr;
    float mu_norm = 0.0f;
};
void launch_flat_finish(const uint64_t *exact_sorted, uint32_t lde, const uint64_t *approx_sorted, uint32_t lda,
                        uint32_t nq, uint32_t ksel, uint32_t kstride, uint32_t kprime, uint64_t n_rows, const float *qsq,
                        float xsq_max, float xsq_min_pos, int cosine, uint32_t dim, SplitErr se, const uint32_t *cnt, uint32_t cap,
                        uint64_t id_offset, uint8_t *flags, uint64_t *out_idx, float *out_dist, uint64_t *out_count,
                        hipStream_t s);
// the exact stage of the Flat pipeline fused into one launch (k' <= 64, k <= 64, dim % 4 == 0): counted select of the hit
// list + re-rank + sort + certification + outputs (k_exact.hip)
struct FlatTailArgs {
    const uint64_t *cand;  // [nq][cap] hit lists of the filter pass
    uint32_t cap;
    const uint32_t *cnt;   // [nq]
    uint32_t kprime, ksel, kstride;
    const float *X;        // row-major rows
    uint32_t dim;
    const float *Q;
    int metric;            // MET_L2_DIRECT or MET_COSINE
    const float *xsq, *qsq;
    uint64_t n_rows;
    float xsq_max, xsq_min_pos;
    int cosine;
    SplitErr se;
    uint64_t id_offset;
    uint8_t *flags;
    uint64_t *out_idx;
    float *out_dist;
    uint64_t *out_count;
    unsigned long long *stamps = nullptr;  // k_flat_tail_lb, builds with -DVDB_TAIL_STAMPS: [nq][32] phase stamps (s_memtime)
    const float *tau = nullptr;  // k_flat_tail_lb: the filter pass's thresholds (the bound of every row outside the hit list)
    uint32_t *qstat = nullptr;   // k_flat_tail_lb, optional (flat_i8_stats): per query rounds walked | hit count << 8
};
// the exact stage behind the 8-bit pass: walks the hit list in key order, 64 keys per round (kprime / 64 rounds at most)
bool flat_tail_lb_supported(uint32_t dim, uint32_t kprime, uint32_t ksel);
void launch_flat_tail_lb(const FlatTailArgs &a, uint32_t nq, hipStream_t s);
// all candidates of a few queries evaluated at once (the second 8-bit attempt of a handful of queries): exact_keys nq x a.cap, topk nq x topk_capacity(ksel)
void launch_flat_full_lb(const FlatTailArgs &a, uint32_t nq, uint64_t *exact_keys, uint64_t *topk, hipStream_t s);
void flat_tail_lb_set_nw(int v);  // waves per query: 0 auto, 8 / 4 / 2 / 1
bool flat_tail64_supported(uint32_t dim, uint32_t kprime, uint32_t ksel);
void launch_flat_tail64(const FlatTailArgs &a, uint32_t nq, hipStream_t s);
// k_small.hip: FlatIndex::knn of a few queries over a small table in ONE launch (the db.search() shape): coalesced rows ->
// products -> strict fold per lane -> in-kernel (distance, index) top-k -> outputs.  dim % 4 == 0, k <= 64, nq <= 64.
struct FlatSmallArgs {
    const float *X;        // row-major rows
    uint64_t n;
    uint32_t dim;
    const float *Q;        // [nq][dim]; device memory or device-visible pinned host memory
    int metric;            // MET_L2_DIRECT or MET_COSINE
    const float *xsq;      // row norms (Cosine)
    uint64_t *part;        // flat_small_part_keys() pair keys of scratch
    uint32_t *counter;     // [nq] arrival counters, zero on entry, zero again on exit
    uint32_t ksel, kstride;
    uint64_t id_offset;
    uint64_t *out_idx;     // device memory or device-visible pinned host memory
    float *out_dist;
    uint64_t *out_count;
};
bool flat_small_supported(uint64_t n, uint32_t dim, uint64_t nq, uint64_t k);
uint32_t flat_small_rows_per_wg(uint64_t n, int num_cu);
size_t flat_small_part_keys(uint64_t n, uint64_t nq, uint32_t ksel, int num_cu);
void launch_flat_small(const FlatSmallArgs &a, uint32_t nq, int num_cu, hipStream_t s);
void launch_certify(const uint64_t *exact_sorted, uint32_t lde, const uint64_t *approx_sorted, uint32_t lda,
                    uint32_t nq, uint32_t k, uint32_t kprime, uint64_t n_rows, const float *qsq, float xsq_max,
                    float xsq_min_pos, int cosine, uint32_t dim, uint8_t *flags, hipStream_t s);

void launch_extract_tau(const uint64_t *sorted, uint32_t ld, uint32_t nq, uint32_t kprime, float *tau, hipStream_t s);
void launch_flag_overflow(const uint32_t *cnt, uint32_t cap, uint32_t min_hits, uint32_t nq, uint8_t *flags,
                          hipStream_t s);

// ---- k_topk.hip ----------------------------------------------------------------------------
// rows per level-1 list
// keys one wave of k_topk_dense selects from: short key rows (threshold samples) are cut finer so that nq x lists
// waves fill the chip; long rows keep the level-2 merge small
uint32_t topk_chunk(uint64_t n);
uint32_t topk_num_lists(uint64_t n);
// tau[q] = kth smallest of keys[q][0..n) (k_select_tau, n <= select_tau_max_n())
// queries >= nq_real (batch padding) get tau = -inf
void launch_select_tau(const float *keys, uint64_t ld, uint32_t n, uint32_t nq, uint32_t nq_real, uint32_t kth, float *tau,
                       hipStream_t s);
uint32_t select_tau_max_n();
uint32_t topk_capacity(uint32_t k);  // entries per list (multiple of 64, >= k)
// level 1: dense f32 keys[q*ld + i], i in [0,n) -> lists[q][list][cap] sorted ascending pair keys
void launch_topk_dense(const float *keys, uint64_t ld, uint64_t n, uint32_t nq, uint32_t k, uint64_t *lists,
                       hipStream_t s);
// level 2: merge `nlists` lists of `cap_in` pair keys per query -> out[q][cap] sorted ascending
void launch_topk_merge(const uint64_t *lists, uint32_t nlists, uint32_t cap_in, uint32_t nq, uint32_t k,
                       uint64_t *out, hipStream_t s);

// level 2 over ONE list per query of which only the first min(cnt[q], cap_in) entries are scanned
void launch_topk_merge_counted(const uint64_t *lists, uint32_t cap_in, const uint32_t *cnt, uint32_t nq, uint32_t k,
                               uint64_t *out, hipStream_t s);
// shard merge for k <= 64 in one launch (wave per query); strides in BYTES between consecutive shards' arrays
void launch_merge_shards64(const float *dists, const uint64_t *ids, const uint64_t *counts, uint64_t stride_d,
                           uint64_t stride_i, uint64_t stride_c, uint32_t S, uint32_t nq, uint32_t k, uint64_t *out_idx,
                           float *out_dist, uint64_t *out_count, hipStream_t s);
void launch_pack_pairs(const float *dists, const uint64_t *ids, const uint64_t *counts, uint64_t stride_d,
                       uint64_t stride_i, uint64_t stride_c /* bytes between shards */, uint32_t S, uint32_t nq, uint32_t k,
                       uint32_t cap_in, uint64_t *lists, hipStream_t s);

// ---- k_u8.hip: VecSet<u8> rows at one byte per element (scalar.rs:117-119, distance/mod.rs:79-95) -------------------
void launch_widen_u8(const uint8_t *in, uint64_t count, float *out, hipStream_t s);
void launch_scan_exact_u8(const uint8_t *X, uint64_t n, uint32_t dim, const float *Q, uint32_t nq, int metric, const float *xsq,
                          const float *qsq, float *out, uint64_t ld, hipStream_t s);
void launch_rerank_u8(const uint8_t *X, uint32_t dim, const float *Q, uint32_t nq, int metric, const float *xsq, const float *qsq,
                      const uint64_t *cand, uint64_t *out, uint32_t ncand, uint32_t ldc, hipStream_t s);

// ---- k_probe.hip ---------------------------------------------------------------------------
// attainable HBM read bandwidth in GB/s: pure streaming read of `bytes`, `iters` timed passes, best of two patterns
double stream_probe(int device, uint64_t bytes, int iters);
double stream_probe_pattern(int device, uint64_t bytes, int iters, int pattern, uint32_t row_bytes);  // 1: MFMA-fragment loads from row-major rows
void mfma_probe(int device, int waves_per_simd, int iters, double *tflops, double *clock_ghz, int i8 = 0);
double latency_probe(int device, uint64_t bytes, uint32_t hops);
double fold_probe(int device, uint32_t adds);  // ns per dependent f32 add (the strict fold's chain)  // ns per dependent HBM load (pointer chase over 128-B lines)

// ---- k_sort.hip (k > 1024) -----------------------------------------------------------------
size_t sort_pairs_temp_bytes(uint64_t n);
void launch_sort_pairs(const float *dist, uint64_t n, uint64_t *tmp_keys, uint64_t *out, void *temp, size_t temp_bytes,
                       hipStream_t s);

// any-size fallbacks (k, ef, n_probes beyond the 1024 pairs the register-resident selects hold):
// every row of pair keys [nq][ld] sorted ascending (in != out); temp = sort_rows_temp_bytes(nq, ld) bytes
size_t sort_rows_temp_bytes(uint64_t nq, uint64_t ld);
void launch_sort_rows(const uint64_t *in, uint64_t *out, uint64_t nq, uint64_t ld, void *temp, size_t temp_bytes, hipStream_t s);
void launch_pair_keys_rows(const float *dist, uint64_t ldd, uint64_t n, uint32_t nq, uint64_t *keys, uint64_t ldk, hipStream_t s);
void launch_copy_prefix(const uint64_t *in, uint64_t ld_in, uint64_t *out, uint64_t ld_out, uint64_t count, uint32_t nq, hipStream_t s);
// ResultSet::add replayed over the offers [nq][ldc] (first ncand of every row, in order; PAIR_NONE skipped), set capacity
// k: out [nq][ldo >= k] = the set, unsorted, PAIR_NONE padded
void launch_resort_big(const uint64_t *offers, uint32_t ncand, uint32_t ldc, uint32_t nq, uint32_t k, uint64_t *out, uint32_t ldo,
                       hipStream_t s);

// ---- k_mfma.hip ----------------------------------------------------------------------------
uint32_t mfma_batch(uint32_t dim);  // queries per workgroup batch: 32 (dim <= 1024), 16 (dim <= 2048), 0 = unsupported
// Q [nq][dim] -> ceil(nq/32) fragment-ordered split-bf16 images of mfma_qfrag_floats(dim) floats each
void launch_mfma_pack_queries(const float *Q, uint32_t nq, uint32_t nq_cover, uint32_t dim, float *qfrag, hipStream_t s);
// fragment-ordered mirror of rows: tiles [tile0, tile1) of 16 rows each; T holds ceil(n/16) tiles rounded up to 4
void launch_tile_rows(const float *X, uint64_t n, uint32_t dim, uint64_t tile0, uint64_t tile1, float *T,
                      hipStream_t s);
// approximate keys key(i,q) = xsq[i] - 2*dot(x_i, q); XT = fragment-ordered mirror; qfrag = nbatch images.
// sample: keys of a strided sample of rows, dense: out[q*ld + j], j < mfma_sample_rows(n, step) (+inf past n)
// cosine != 0: keys are -dot(x_i,q)/|x_i| (0 for zero rows), which rank like the cosine distance
void launch_flat_mfma_sample(const float *XT, uint64_t n, uint32_t dim, const float *qfrag, uint32_t nbatch,
                             const float *xsq, int cosine, uint32_t step, float *out, uint64_t ld, int num_cu,
                             hipStream_t s);
// filter: ONE launch walks all nbatch passes; pair keys of all rows with key <= tau[q] land in cand[q][0..cap)
// (cnt[q] counts every hit, so cnt[q] > cap means candidates were dropped)
void launch_flat_mfma_filter(const float *XT, uint64_t n, uint32_t dim, const float *qfrag, uint32_t nbatch,
                             const float *xsq, int cosine, const float *tau, uint64_t *cand, uint32_t *cnt,
                             uint32_t cap, uint32_t *sync /* mfma_sync_words() zeroed words */, int num_cu, hipStream_t s);
size_t mfma_sync_words(uint32_t nbatch, int num_cu);
// k_gemm.hip: the same filter for groups of gemm_group() = 128 queries per corpus pass; qfrag = images packed with
// launch_mfma_pack_queries_nh(.., 8, ..), tau / cand / cnt indexed by the global query number as above
// qmul == nullptr: split-bf16 operands (launch_tile_rows / launch_mfma_pack_queries_nh); otherwise the scaled fp16
// operands of k_half.hip and qmul[q] = 1 / (row scale * query scale).  cnt[ngroups * 128 .. + 127] are the arrival counters of the
// cooperative sets: ZERO on entry (launch_query_prep_h / the caller's memset), part of the same allocation as cnt
void launch_flat_gemm_filter(const float *XT, uint64_t n, uint32_t dim, const float *qfrag, const float *qmul, uint32_t ngroups,
                             const float *xsq, int cosine, const float *tau, uint64_t *cand, uint32_t *cnt,
                             uint32_t cap, int debug, int num_cu, hipStream_t s);
uint32_t gemm_group();
// threshold sample with the same kernel: every unit_step-th unit of rows, dense keys out[q*ld + j]
uint64_t gemm_sample_rows(uint64_t n, uint32_t unit_step);
void launch_flat_gemm_sample(const float *XT, uint64_t n, uint32_t dim, const float *qfrag, const float *qmul, uint32_t ngroups,
                             const float *xsq, int cosine, uint32_t unit_step, float *out, uint64_t ld, int num_cu,
                             hipStream_t s);
void gemm_set_tw(int v);
void gemm_set_zigzag(int v);   // odd query groups walk the corpus backwards: 0 auto, 1 off, 2 on
void gemm_set_nt(int v);       // X-stream cache policy of the filter pass: 0 auto (non-temporal when the mirror exceeds the Infinity Cache), 1 default policy, 2 non-temporal
void gemm_set_block_rows(uint64_t v);  // filter pass in row blocks (one launch per block, all query groups): 0 off (default), n rows
void gemm_set_stagger(int v);  // 0 (default): workgroups start together; n: start delays of up to n unit steps
bool gemm_f16_supported(uint32_t dim);
// k_half.hip: scaled fp16 mirror / query images for the GEMM_F16 variant and their measured rounding errors
void launch_tile_rows_h(const float *X, uint64_t n, uint32_t dim, uint64_t tile0, uint64_t tile1, float sx, void *T,
                        hipStream_t s);
void launch_row_split_err(const float *X, const float *xsq, uint64_t row0, uint64_t row1, uint32_t dim, float sx,
                          uint32_t *out2 /* [2] float bits: max |dx|^2, max |dx|^2/|x|^2 (atomicMax) */, hipStream_t s);
void launch_rows_to_half(const float *X, uint64_t count, float sx, uint16_t *H, hipStream_t s);
void launch_rows_to_q8(const float *X, uint64_t n, uint32_t dim, int8_t *Q8, float *scale, float *err, hipStream_t s);
void launch_query_prep_h(const float *Q, uint32_t nq, uint32_t nq_pad, uint32_t dim, float sx, float *qsq /* out: |q|^2, strict order */,
                         float *qscale, float *qmul, float *qerr, uint32_t *hits /* zeroed */,
                         void *qfrag /* non-null: also write the query image of launch_pack_queries_h(.., NH = 8, ..) */, hipStream_t s);
void launch_pack_queries_h(const float *Q, uint32_t nq, uint32_t nq_cover, uint32_t dim, uint32_t NH, const float *qscale,
                           void *qfrag, hipStream_t s);
void launch_mfma_pack_queries_nh(const float *Q, uint32_t nq, uint32_t nq_cover, uint32_t dim, uint32_t NH, float *qfrag,
                                 hipStream_t s);
// k_gemm8.hip / k_i8.hip: the 8-bit pass (centred int8 mirror, lower-bound keys)
constexpr uint32_t I8_MEAN_CHUNKS = 64;
bool gemm8_supported(uint32_t dim);
void gemm8_set_nt(int v);
void gemm8_set_kc(int v);
void gemm8_set_burst(int v);
uint32_t gemm8_last_coop();   // set size of the most recent 8-bit filter launch (0: no cooperative sets)
uint32_t gemm_last_coop();
void gemm_set_coop(int v);    // the same for the fp16 / split-bf16 filter kernel
void gemm8_set_grid(int v);   // measurement: workgroups of the cooperative 8-bit filter (0 = one per CU)
void gemm8_set_coop(int v);   // 0 auto (the workgroups of an XCD share one row stream through its L2 when the shape allows), 1 off
void gemm8_set_res(int v);
void gemm8_set_sample_res(int v);    // 0 auto (the query group's whole image resident in LDS when it fits), 1 off (chunked staging)
uint64_t gemm8_sample_rows(uint64_t n, uint32_t unit_step);
// (cnt[ngroups * 128 .. + 127]: the arrival counters of the cooperative sets, zero on entry -- launch_query_prep_i8 clears them)
void launch_flat_gemm8_filter(const void *XT, uint64_t n, uint32_t dim, const void *qfrag, const float *qscale, uint32_t ngroups,
                              const float *rowc, const float *tau, uint64_t *cand, uint32_t *cnt, uint32_t cap, int debug, int num_cu,
                              hipStream_t s, uint32_t hits_expected = 0);
void launch_flat_gemm8_sample(const void *XT, uint64_t n, uint32_t dim, const void *qfrag, const float *qscale, uint32_t ngroups,
                              const float *rowc, uint32_t unit_step, float *out, uint64_t ld, int num_cu, hipStream_t s, int unit_min = 0);
uint64_t gemm8_sample_units(uint64_t n, uint32_t unit_step);
// xsq_cos != null / cosine != 0: the Cosine form -- unit rows / unit queries (k_i8.hip); xsq_cos = the rows' cached strict-fold |x|^2
void launch_i8_col_mean(const float *X, uint64_t n, uint32_t dim, float *part /* I8_MEAN_CHUNKS * dim */, float *mu, hipStream_t s,
                        const float *xsq_cos = nullptr);
void launch_i8_row_stats(const float *X, uint64_t n, uint32_t dim, const float *mu, uint64_t n_s, uint64_t stride,
                         float *stats /* 2 * round_up(n_s, 16) */, hipStream_t s, const float *xsq_cos = nullptr);
void launch_tile_rows_i8(const float *X, uint64_t n, uint32_t dim, uint64_t tile0, uint64_t tile1, const float *mu, float l1, float l2,
                         void *T, float *rowc, hipStream_t s, const float *xsq_cos = nullptr);
void launch_query_prep_i8(const float *Q, uint32_t nq, uint32_t nq_pad, uint32_t dim, const float *mu, float l1, float l2, float *qsq,
                          float *qscale, float *qoff, uint32_t *hits, void *qfrag, hipStream_t s, int cosine = 0);
// k_redo.hip: second attempts of a Flat call
void launch_gather_rows_f32(const float *src, const uint64_t *rows, uint64_t nr, uint32_t width, float *dst, hipStream_t s);
void launch_scatter_results(const uint64_t *ri, const float *rd, const uint64_t *rc, const uint64_t *rows, uint64_t nr, uint32_t k,
                            uint64_t *o_idx, float *o_dist, uint64_t *o_cnt, hipStream_t s);
void launch_gather_dk(const float *o_dist, const uint64_t *o_cnt, const uint64_t *rows, uint64_t nr, uint32_t k, uint32_t ksel, float *dk,
                      hipStream_t s);
// tighter lower-bound keys for the hit lists of the 8-bit pass from the row-major fp16 image (k_redo.hip); max_hits = longest list to cover
void launch_flat_refine_half(const uint16_t *rows_h, uint32_t dim, float sx, float dx_abs, float dx_rel, int cosine, const float *Q, const float *xsq,
                             const float *qsq, const float *qoff, uint64_t *cand, uint32_t cap, const uint32_t *cnt, uint32_t nq, uint32_t max_hits,
                             hipStream_t s);
void launch_i8_tau_from_dk(const float *dk, uint32_t nq, uint32_t nq_pad, const float *qoff, const float *qsq, float xsq_max, float mu_norm,
                           uint32_t dim, int cosine, float *tau, hipStream_t s);
void mfma_set_sample_thin(int v);
void mfma_sample_plan(uint64_t n, uint32_t kprime, uint32_t *step, uint32_t *rank, uint32_t target_floor = 1024);  // target_floor: expected hits per query
uint64_t mfma_sample_rows(uint64_t n, uint32_t step);
size_t mfma_qfrag_floats(uint32_t dim);
uint32_t mfma_dim_pad(uint32_t dim);  // columns of the mirror / Q images (dim rounded up to 64, zero filled)
void mfma_set_variant(int v);  // tuning hook (0 = default)
void mfma_set_share(int v);    // query batches (of 32) that ride one HBM pass through XCD-local L2 sharing (1, 2, 4, 8)
uint32_t mfma_share();
uint64_t mfma_row_pad();
bool mfma_supported(uint32_t dim);

}  // namespace vdb
