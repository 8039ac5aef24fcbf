// index.hpp -- the HBM-resident index object behind the C ABI (include/vdbhip.h).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace vdb {

// growable device buffer (amortised doubling, like Vec<T>: vec_set.rs:113-118)
// VDB_EFENCE=1 (debugging aid, read once): every buffer is placed so that it ENDS where its allocation ends, and allocations are
// whole 2-MiB granules -- a kernel that reads or writes more than the 256-B alignment slack past a buffer then leaves the
// mapping and faults at once instead of silently touching whatever the allocator put there (how the over-read of
// k_pq_adc16's idle lanes stayed unseen for a round).  Used for soaks of the test suite, never in production.
inline bool devbuf_efence() {
    static const bool on = [] {
        const char *e = std::getenv("VDB_EFENCE");
        return e && e[0] == '1';
    }();
    return on;
}
// (testing aid) every DevBuf allocation of at least this many bytes fails as if the device were out of memory; 0 = off
inline std::atomic<size_t> &devbuf_fail_over() {
    static std::atomic<size_t> v{0};
    return v;
}
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    void *base = nullptr;  // what hipMalloc returned (== p unless VDB_EFENCE)
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (base) (void)hipFree(base);
        p = base = nullptr;
        cap = 0;
    }
    static void dev_malloc(void **out, size_t bytes) {
        const size_t lim = devbuf_fail_over().load();
        hipError_t e = (lim && bytes >= lim) ? hipErrorOutOfMemory : hipMalloc(out, bytes);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();  // (not sticky, but the slot is read by the next VDB_SYNC)
            throw AllocError("device allocation of " + std::to_string(bytes) + " bytes failed: out of memory");
        }
        VDB_HIP(e);
    }
    static void alloc(size_t want, void **base_out, void **p_out) {
        if (!devbuf_efence()) {
            dev_malloc(base_out, want);
            *p_out = *base_out;
            return;
        }
        constexpr size_t G = size_t(2) << 20;
        const size_t total = (want + G - 1) / G * G;
        dev_malloc(base_out, total);
        *p_out = static_cast<char *>(*base_out) + ((total - want) & ~size_t(255));
    }
    // contents are NOT preserved
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        release();
        size_t want = bytes < 256 ? 256 : bytes;
        alloc(want, &base, &p);
        cap = want;
    }
    // contents preserved up to `keep` bytes
    void grow(size_t bytes, size_t keep, hipStream_t s) {
        if (bytes <= cap) return;
        size_t want = cap ? cap : 256;
        while (want < bytes) want *= 2;
        if (devbuf_efence()) want = bytes;  // (no slack capacity to hide an over-read in)
        void *nb = nullptr, *np = nullptr;
        alloc(want, &nb, &np);
        if (keep) {
            VDB_HIP(hipMemcpyAsync(np, p, keep, hipMemcpyDeviceToDevice, s));
            VDB_SYNC(s);
        }
        if (base) (void)hipFree(base);
        base = nb;
        p = np;
        cap = want;
    }
    template <class T>
    T *as() const { return reinterpret_cast<T *>(p); }
};

struct ProfEntry {
    double ms = 0;
    uint64_t launches = 0;
    double bytes = 0;
};

// per-call scratch: one per concurrent reader (read-side calls are re-entrant, see vdbhip.h)
struct Workspace {
    hipStream_t stream = nullptr;
    DevBuf q, qsq, qfrag, qfrag_g, qaux, dense, lists, keys_a, keys_b, keys_c, flags, out_idx, out_dist, out_cnt, lut, misc;
    DevBuf pq_img16, pq_aux16;     // 8-bit PQ codes: the sliced 16-bit table images of a round of queries, their minima / offsets / steps (pq.hip)
    DevBuf small_part, small_cnt;  // k_flat_small: per-workgroup key lists, arrival counters (zero between launches)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    struct Pending {
        std::string name;
        size_t ev;
        double bytes;
    };
    std::vector<Pending> pending;
    size_t ev_used = 0;
    // pinned host staging for the small per-call read-backs (flags, counters): a D2H copy into pageable memory goes
    // through the runtime's own staging buffer and costs an extra hop before the stream sync returns
    void *h_pinned = nullptr;
    size_t h_pinned_cap = 0;
    hipEvent_t order_ev = nullptr;  // orders this workspace's stream behind a caller's stream without a host wait (vdb_*_begin)
    void *pinned(size_t bytes) {
        if (bytes > h_pinned_cap) {
            if (h_pinned) (void)hipHostFree(h_pinned);
            h_pinned = nullptr;
            size_t want = bytes < (64u << 10) ? (64u << 10) : bytes;
            VDB_HIP(hipHostMalloc(&h_pinned, want, hipHostMallocDefault));
            h_pinned_cap = want;
        }
        return h_pinned;
    }
    Workspace() {
#ifndef VDB_HOST_SANITIZER_BUILD  // (tests/cpp/tsan_host.cpp drives the pool and the host builders with no device present)
        VDB_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
#endif
    }
    ~Workspace() {
        if (h_pinned) (void)hipHostFree(h_pinned);
        if (order_ev) (void)hipEventDestroy(order_ev);
        for (auto &e : ev_pool) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
        if (stream) (void)hipStreamDestroy(stream);
    }
};

struct PQState {
    bool present = false;
    uint64_t n_bits = 0, m = 0, kc = 0, enc_dim = 0;
    uint64_t n_coded = 0;            // rows d_codes covers; every PQ search requires n_coded == Index::n
    std::vector<uint64_t> gstart;    // m+1
    std::vector<float> h_centroids;  // kc*dim
    std::vector<float> h_cent_cache; // m*kc
    DevBuf d_centroids, d_cent_cache, d_codes, d_gstart;
    DevBuf d_codes_t;                // word-major mirror of d_codes for the quantised ADC scan (pq.hip: k_pq_tile_codes)
    bool codes_t_valid = false;
    std::atomic<uint64_t> adc16_queries{0};  // queries whose ADC scan ran on the quantised tables (k_pq_adc16)
    std::atomic<uint64_t> q8_overflow{0}, q8_short{0};  // 8-bit codes: queries the quantised pass handed to the f32 scan (list overflowed | fewer than ef sums at or below tau)
    std::atomic<uint64_t> q8_hits_sum{0}, q8_hits_max{0};  // candidates of the quantised pass (sum over queries, largest list)
};

struct HNSWState {
    bool present = false;
    uint64_t m = 0, max_m0 = 0, ef_construction = 0, default_ef = 0;
    float inv_log_m = 0;
    std::vector<uint32_t> level0;
    std::vector<uint64_t> len0, vec_level, upper_off, upper_len;
    std::vector<uint32_t> upper;
    bool has_enter = false;
    uint64_t enter_point = 0, enter_level = 0;
    uint64_t rng_state = 0;
    // device mirror (uploaded lazily when dirty)
    bool dev_dirty = true;
    DevBuf d_level0, d_len0, d_upper, d_upper_len, d_upper_off;
    std::atomic<uint64_t> last_n_dist{0}, last_n_expanded{0};
    std::atomic<uint64_t> last_half_dropped{0};  // of last_n_dist: rows the certified half-precision pre-pass ruled out (last call)
    std::atomic<uint64_t> heap_walk_queries{0};  // queries answered by k_hnsw_search_big (ef > 1024 or LDS pool overflow)
};

// a Flat call between its two halves (Index::flat_knn_enqueue / flat_knn_finish)
struct FlatPending {
    bool active = false;  // the MFMA pipeline is on the stream, its certification flags are still to be read
    bool half = false;
    bool i8 = false;      // first pass on the 8-bit mirror (its uncertified queries go to the fp16 / split-bf16 tiers)
    bool i8_second = false;  // ... its second attempt: thresholds from the first walk's k-th distances (k_redo.hip)
    bool stats = false;      // the exact stage wrote one word of statistics per query behind the flags
    bool refined = false;    // the hit keys were tightened from the fp16 image before the walk
    uint32_t kprime = 0, ksel = 0;
    uint64_t nq = 0, k = 0;
    const float *d_q = nullptr;
    uint64_t *d_idx = nullptr;
    float *d_dist = nullptr;
    uint64_t *d_cnt = nullptr;
};

struct Index;
// IVFIndex (ivf_index.rs:34-47): centroids as a small Flat index of their own, clusters as CSR over row ids
struct IVFState {
    bool present = false;
    uint64_t k = 0, default_n_probes = 4;  // :108
    std::shared_ptr<Index> cent;           // k x dim centroids (find_n_nearest = Flat knn over them)
    std::vector<uint64_t> assign;          // cluster of every row
    std::vector<uint32_t> offsets;         // k+1
    std::vector<uint32_t> sizes_desc;      // cluster sizes, descending (bounds the candidates of n probes)
    DevBuf d_offsets, d_members;           // u32 [k+1], u32 [n] (ascending id inside a cluster)
    std::atomic<uint64_t> last_rows_fetched_q8{0};  // cluster-major 8-bit tier: rows read (once each); 0: query-major or tier not run
    std::atomic<uint64_t> last_kept_q8{0};  // offers the 8-bit tier passed on to the fp16 tier (0: that tier did not run)
    std::atomic<uint32_t> q8_overflows{0};
    std::atomic<uint64_t> last_offers{0}, last_kept{0};  // measurement on: offers of the last call / those its pre-pass kept
    std::atomic<uint32_t> half_overflows{0};  // calls whose half-precision pre-pass kept more offers than its lists hold (4: stop trying)
};

struct Index {
    int device = 0;
    int num_cu = 256;
    uint64_t dim = 0;
    int dist = 0;
    uint64_t n = 0;
    uint64_t id_offset = 0;
    // VecSet<T>: T = f32 (the DynamicIndex case) or u8 (scalar.rs:117-119): d_rows holds n * dim elements of elem_size()
    // bytes.  A u8 index serves Flat search (exact scan, MFMA shortlist over mirrors that hold u8 values exactly, native
    // u8 re-rank); PQ / HNSW / IVF are built over f32 tables only, as DynamicIndex instantiates them (dynamic_index.rs:11-14).
    bool elem_u8 = false;
    size_t elem_size() const { return elem_u8 ? 1 : sizeof(float); }
    // f32 view of rows [r0, r1) for the build-time kernels: base pointer B with B + r * dim = row r (the rows themselves,
    // or a widened copy of the chunk in `scratch` for a u8 index)
    const float *rows_f32_view(DevBuf &scratch, uint64_t r0, uint64_t r1, hipStream_t s) const;
    // fn(view, tile_a, tile_b, row_a, row_b) over the 16-row tiles [t_begin, t_end), rows clipped to n_rows: one call on
    // the rows themselves (f32), or one per widened chunk (u8); view + r * dim addresses row r for row_a <= r < row_b
    template <class F>
    void for_tile_chunks(Workspace &ws, uint64_t t_begin, uint64_t t_end, uint64_t n_rows, F fn) const {
        if (!elem_u8) {
            fn(d_rows.as<float>(), t_begin, t_end, std::min(t_begin * 16, n_rows), n_rows);
            return;
        }
        constexpr uint64_t CH = 12 * 64;  // tiles per chunk (whole mirror units): 12 288 rows
        ws.dense.reserve(CH * 16 * dim * sizeof(float));  // before the loop: a later, larger reserve would free a buffer in use
        for (uint64_t t = t_begin; t < t_end; t += CH) {
            const uint64_t tb = std::min(t_end, t + CH), ra = std::min(t * 16, n_rows), rb = std::min(tb * 16, n_rows);
            fn(rows_f32_view(ws.dense, ra, rb, ws.stream), t, tb, ra, rb);
        }
    }
    DevBuf d_rows, d_sq;
    // MFMA-fragment-ordered split-bf16 mirror of d_rows (k_mfma.hip; 4 B/element), only when mfma_supported(dim).  Built
    // LAZILY by the first search that needs it (the redo tier of the fp16 pass, calls without an fp16 mirror,
    // flat_half = 1): an index whose queries all certify on the fp16 pass never pays its N*d*4 bytes of HBM.
    DevBuf d_tiled;
    std::atomic<bool> tiled_built{false};  // d_tiled covers rows [0, n); kept in step by add_rows / swap_remove once built
    std::mutex tiled_mu;            // read-side calls are re-entrant: one of them builds, the others wait
    bool ensure_tiled(Workspace &ws);  // false: the split-bf16 mirror could not be allocated (the exact scan answers)
    uint64_t tiled_failed_n = ~0ull, half_failed_n = ~0ull, i8_failed_n = ~0ull;  // row count at which a mirror's allocation failed (not retried until it changes)
    std::atomic<uint64_t> mirror_alloc_failures{0};
    void prepare_flat(bool all_tiers);  // builds now what the first Flat search would build (vdb_index_prepare)
    uint64_t hbm_bytes_per_row() const;  // resident bytes per row over all per-row buffers (rows, norms, mirrors, codes, links)
    // scaled fp16 mirror for k_flat_gemm<GEMM_F16> (k_half.hip): rows stored as fp16(x * 2^(13 - half_exp)), every row
    // norm < 2^half_exp; half_dx_* = measured rounding error of the mirror (max |dx_r|, max |dx_r| / |x_r|)
    DevBuf d_tiled_h, d_half_err;
    bool half_valid = false;
    int half_exp = 0;
    float half_dx_abs = 0.0f, half_dx_rel = 0.0f;
    int flat_half_mode = 0;        // 0 auto, 1 off, 2 on even after many uncertified queries
    uint32_t flat_half_kmul = 4;   // shortlist of the fp16 pass: max(64, kmul * k) rows per query
    std::atomic<uint64_t> half_queries{0}, half_redo{0};  // queries through the fp16 pass / redone with split-bf16
    void half_refresh(Workspace &ws, uint64_t n_old, uint64_t n_new);  // after rows [n_old, n_new) changed
    // An index the 8-bit pass serves (i8_defers_half) builds this mirror on FIRST NEED instead of at add time: a query the
    // 8-bit pass hands on, a call it does not take (k > 64), the walks' / IVF scan's row-major fp16 image (which shares its
    // scale and measured error).  half_n = rows the mirror covers; ensure_half brings it up to n.
    uint64_t half_n = 0;
    std::mutex half_mu;
    bool i8_defers_half() const;
    bool ensure_half(Workspace &ws);  // false: this index has no fp16 mirror (dimension, extreme norms)
    float half_sx() const { return std::ldexp(1.0f, 13 - half_exp); }
    // Centred 8-bit mirror for k_flat_gemm8 (k_gemm8.hip, k_i8.hip; L2Sqr over f32 rows): 1 B/element in fragment order +
    // {C_r, M_r} per row.  Built by the first search that wants it (ensure_i8), extended after add_rows, kept in step by
    // swap_remove; mu / lambda are re-chosen (and everything rewritten) when the table has doubled since they were measured.
    DevBuf d_tiled_i8, d_rowc_i8, d_mu_i8;
    std::atomic<bool> i8_valid{false};
    uint64_t i8_n = 0;          // rows the mirror covers
    uint64_t i8_mu_rows = 0;    // table size when mu / lambda were measured
    float i8_l1 = 0.0f, i8_l2 = 0.0f, i8_mu_norm = 0.0f;
    int flat_i8_mode = 0;       // 0 auto, 1 off, 2 on even after many uncertified queries
    // rows the exact stage of the FIRST attempt may walk per query (64 per round); the second attempt walks the whole candidate list.
    // 2048 = 32 rounds, i.e. the whole hit list of the sample plan's ~1000 hits: a query that exhausts its list is certified by the threshold
    // itself, and only what then is still open pays a second pass over the mirror.  Measured at 1M x 960 (bench.py --data ..., same box):
    // separable data never walk more than 4 rounds whatever the limit; loose clusters 1.51 ms per step at 512 / 1024 / 2048 (their
    // stragglers need the second attempt's longer lists either way); tight clusters 3.03 / 2.89 / 2.47 ms
    uint32_t flat_i8_kprime = 2048;
    int flat_i8_full = 0;           // second attempt of <= 96 queries: all candidates evaluated at once (0 on, 1 off: the walk)
    int flat_i8_refine = 0;         // hit keys tightened from the fp16 row image before the walk (k_flat_refine_half): 0 auto (on while the walks are long), 1 off, 2 always
    std::atomic<int> i8_refine_on{0};        // auto state
    std::atomic<uint32_t> i8_refine_calls{0};  // first-attempt calls since the state changed (every 32nd call of the on state runs without: the probe)
    std::atomic<uint64_t> i8_refine_queries{0};  // queries whose hit lists were refined
    int flat_i8_unit_min = 0;       // threshold sample of the 8-bit pass by unit minima when the sampled units are many: 0 auto, 1 off
    int flat_i8_second = 0;         // second 8-bit attempt with thresholds from the first walk (k_redo.hip): 0 on, 1 off
    std::atomic<uint64_t> i8_second_queries{0}, i8_second_redo{0};
    uint32_t flat_i8_hits = 1024;   // expected hits per query the threshold sample of the 8-bit pass aims at (>= 256 = 4 x the 64 guaranteed)
    std::atomic<uint64_t> i8_queries{0}, i8_redo{0};  // queries through the 8-bit pass / passed on to the next tier
    // (measurement, flat_i8_stats = 1) per-query work of the 8-bit pass's exact stage: queries by rounds walked, hits per query
    int flat_i8_stats = 0;
    std::atomic<uint64_t> i8_rounds_hist[9] = {};
    std::atomic<uint64_t> i8_hits_sum{0}, i8_hits_max{0}, i8_stat_queries{0};
    std::atomic<uint64_t> i8_rows_walked{0};          // (measurement) not maintained in production
    std::mutex i8_mu;
    bool i8_applicable(uint32_t ksel) const;
    bool ensure_i8(Workspace &ws);  // false: the mirror could not be allocated (the next tier answers)
    void ensure_i8_locked(Workspace &ws);
    // Row-major fp16 image of the rows, same scale and rounding as d_tiled_h (so half_dx_* bound its error as well): the
    // operand of the HNSW walk's certified pre-pass (hnsw.hip, hnsw_half_dots).  Built on the first walk that wants it,
    // extended when rows were added since, rebuilt when the scale changed.
    DevBuf d_rows_h;
    uint64_t rows_h_n = 0;
    uint64_t rows_h_failed_n = ~0ull;  // row count at which the image's allocation failed (not retried until it changes)
    int rows_h_exp = 0;
    std::mutex rows_h_mu;
    bool ensure_rows_h(Workspace &ws);  // false: this index has no fp16 image (dim, element type, extreme norms)
    // 8-bit image of the rows with one scale and one measured error per row (half_rows.hpp, "8-bit tier"): the first tier of
    // the IVF scan's pre-pass; built on first use, extended after add, dropped by swap_remove
    DevBuf d_rows_q8, d_q8_scale, d_q8_err;
    uint64_t rows_q8_n = 0;
    std::mutex rows_q8_mu;
    bool ensure_rows_q8(Workspace &ws);
    std::vector<float> h_sq;  // host mirror of d_sq (4 B/row), kept in step by add_rows / swap_remove
    float xsq_max = 0.0f;
    float xsq_min_pos = 3.4e38f;  // smallest positive row |x|^2 seen (cosine certification: clamp check)
    // lazily materialised host mirror of the rows (needed by the host-side builders and vdb_index_row)
    mutable std::vector<float> h_rows;
    mutable bool host_valid = true;
    mutable std::mutex host_mu;
    int flat_mode = 0;
    int flat_gemm_mode = 0;   // 0 auto (more than 64 queries per call), 1 off, 2 forced (k_gemm.hip)
    int flat_gemm_debug = 0;
    int flat_tail_mode = 0;  // 0 auto (fused exact stage when k' <= 64), 1 separate kernels
    int flat_small_mode = 0;          // one-launch search of small tables (k_small.hip): 0 auto, 1 off, 2 whenever the shape allows
    uint64_t flat_small_max_rows = 16384;  // auto: tables up to this many rows (where the MFMA shortlist takes over), calls of < 32 queries
    bool flat_small_applies(uint64_t nq, uint64_t k) const;
    // q / outputs: device memory or device-visible pinned host memory
    void flat_small_device(Workspace &ws, const float *q, uint64_t nq, uint64_t k, uint64_t *o_idx, float *o_dist, uint64_t *o_cnt);
    std::atomic<uint64_t> fallback_count{0};
    PQState pq;
    HNSWState hnsw;
    IVFState ivf;

    std::mutex ws_mu;
    std::vector<std::unique_ptr<Workspace>> ws_free;
    // end of the most recent Flat corpus pass enqueued on any of this index's streams (flat_knn_enqueue orders passes by it)
    std::mutex pass_mu;
    hipEvent_t pass_ev = nullptr;
    bool pass_ev_valid = false;
    bool prof_on = false;
    std::mutex prof_mu;
    std::map<std::string, ProfEntry> prof;

    Index(int dev, uint64_t d, int ds, bool u8 = false);
    ~Index() {
        if (pass_ev) (void)hipEventDestroy(pass_ev);
    }
    void use_device() const {
#ifndef VDB_HOST_SANITIZER_BUILD
        VDB_HIP(hipSetDevice(device));
#endif
    }
    std::unique_ptr<Workspace> acquire_ws();
    void release_ws(std::unique_ptr<Workspace> ws);
    const float *host_rows() const;  // materialise the host mirror if needed

    void add_rows(const void *rows, uint64_t count, bool on_device);  // elements of elem_size() bytes
    void swap_remove(uint64_t i);

    // timing hooks
    void prof_begin(Workspace &ws, const char *name, double bytes);
    void prof_end(Workspace &ws);
    void prof_collect(Workspace &ws);  // after a stream sync

    // search entry points; d_* are device pointers, results [nq][k]
    // d_dk_hint (8-bit pass only): per query an upper bound of its k-th distance -> the pass runs with thresholds derived from it
    // instead of sampled ones and its exact stage may walk the whole candidate list (the second attempt of k_redo.hip)
    void flat_knn_device(Workspace &ws, const float *d_q, uint64_t nq, uint64_t k, uint64_t *d_idx, float *d_dist,
                         uint64_t *d_cnt, bool allow_half = true, uint32_t kprime_min = 0, bool allow_i8 = true,
                         const float *d_dk_hint = nullptr);
    void flat_knn_enqueue(Workspace &ws, const float *d_q, uint64_t nq, uint64_t k, uint64_t *d_idx, float *d_dist, uint64_t *d_cnt,
                          bool allow_half, uint32_t kprime_min, FlatPending &p, bool allow_i8 = true, const float *d_dk_hint = nullptr);
    void flat_knn_finish(Workspace &ws, FlatPending &p);
    void flat_sorted_device(Workspace &ws, const float *d_q, uint64_t nq, uint64_t ksel, uint64_t k, uint64_t *d_idx,
                            float *d_dist, uint64_t *d_cnt);
    void scan_rows(uint64_t nrows, uint32_t d, const float *Q, uint32_t nq, int metric, const float *xsq, const float *qsq, float *out,
                   uint64_t ld, bool use_lds, hipStream_t s) const;
    void rerank_rows(uint32_t d, const float *Q, uint32_t nq, int metric, const float *xsq, const float *qsq, const uint64_t *cand,
                     uint64_t *out, uint32_t ncand, uint32_t ldc, hipStream_t s) const;
    void flat_debug_keys(Workspace &ws, const float *d_q, uint64_t nq, int tier, float *h_keys, float *h_qsq, float *h_qerr,
                         float *h_dx /* [4]: dx_abs, dx_rel, xsq_max, xsq_min_pos */);
    void flat_exact_device(Workspace &ws, const float *d_q, const float *d_qsq, uint64_t nq, uint32_t ksel,
                           uint64_t k, uint64_t *d_idx, float *d_dist, uint64_t *d_cnt);
};

struct WsLease {
    Index &ix;
    std::unique_ptr<Workspace> ws;
    explicit WsLease(Index &i) : ix(i), ws(i.acquire_ws()) {}
    ~WsLease() { ix.release_ws(std::move(ws)); }
    Workspace &operator*() { return *ws; }
    Workspace *operator->() { return ws.get(); }
};

}  // namespace vdb
