// api.hip -- extern "C" boundary of libvdbhip.so (declarations + reference citations: include/vdbhip.h).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "ctx.hpp"
#include "pq_hnsw.hpp"

using namespace vdb;

static thread_local std::string g_last_error;
void vdb::set_last_error(const std::string &m) { g_last_error = m; }
void vdb::require_gpu() {
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0)
        throw Error(VDB_ERR_NOGPU, std::string("no usable HIP device (libvdbhip has no CPU fallback): ") +
                                       hipGetErrorString(e));
}

// device-resident variants: inputs/outputs on the index's GPU, ids < 2^32; synchronous on return
void vdb::merge_topk_dev(Index &ix, const void *d_dists, const void *d_ids, const void *d_counts, uint64_t stride_d,
                           uint64_t stride_i, uint64_t stride_c, uint64_t n_shards, uint64_t nq, uint64_t k,
                           void *d_out_idx, void *d_out_dist, void *d_out_count, void *stream) {
    VDB_REQUIRE(d_dists && d_ids && d_counts && d_out_idx && d_out_dist && d_out_count, "null argument");
    VDB_REQUIRE(k >= 1 && k <= 1024, "k must be in 1..1024");
    VDB_REQUIRE(nq <= 65535 && n_shards <= 65535, "too many queries or shards for one call");
    ix.use_device();
    WsLease ws(ix);
    VDB_SYNC(static_cast<hipStream_t>(stream));
    if (k <= 64) {  // one launch, no scratch lists
        launch_merge_shards64(static_cast<const float *>(d_dists), static_cast<const uint64_t *>(d_ids),
                              static_cast<const uint64_t *>(d_counts), stride_d, stride_i, stride_c, (uint32_t)n_shards,
                              (uint32_t)nq, (uint32_t)k, static_cast<uint64_t *>(d_out_idx), static_cast<float *>(d_out_dist),
                              static_cast<uint64_t *>(d_out_count), ws->stream);
        VDB_SYNC(ws->stream);
        return;
    }
    uint32_t cap = topk_capacity((uint32_t)k);
    ws->lists.reserve(nq * n_shards * cap * sizeof(uint64_t));
    ws->keys_c.reserve(nq * cap * sizeof(uint64_t));
    launch_pack_pairs(static_cast<const float *>(d_dists), static_cast<const uint64_t *>(d_ids),
                      static_cast<const uint64_t *>(d_counts), stride_d, stride_i, stride_c, (uint32_t)n_shards,
                      (uint32_t)nq, (uint32_t)k, cap, ws->lists.as<uint64_t>(), ws->stream);
    launch_topk_merge(ws->lists.as<uint64_t>(), (uint32_t)n_shards, cap, (uint32_t)nq, (uint32_t)k,
                      ws->keys_c.as<uint64_t>(), ws->stream);
    launch_finalize(ws->keys_c.as<uint64_t>(), cap, (uint32_t)nq, (uint32_t)k, (uint32_t)k, 0,
                    static_cast<uint64_t *>(d_out_idx), static_cast<float *>(d_out_dist),
                    static_cast<uint64_t *>(d_out_count), ws->stream);
    VDB_SYNC(ws->stream);
}

extern "C" {

const char *vdb_last_error(void) { return g_last_error.c_str(); }
int vdb_version(void) { return 100; }

int vdb_device_count(int *out) {
    VDB_API_BEGIN
    VDB_REQUIRE(out, "null out");
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    *out = (e == hipSuccess) ? cnt : 0;
    VDB_API_END
}

int vdb_index_create(int device_id, uint64_t dim, int dist, vdb_index **out) {
    VDB_API_BEGIN
    VDB_REQUIRE(out, "null out");
    VDB_REQUIRE(dim > 0 && dim < (1u << 24), "dim must be in 1..2^24");
    VDB_REQUIRE(dist == VDB_L2SQR || dist == VDB_COSINE, "dist must be 0 (L2Sqr) or 1 (Cosine)");
    require_gpu();
    *out = new vdb_index(device_id, dim, dist);
    VDB_API_END
}
// VecSet<u8> (scalar.rs:117-119): rows stored at one byte per element; Flat search only
int vdb_index_create_u8(int device_id, uint64_t dim, int dist, vdb_index **out) {
    VDB_API_BEGIN
    VDB_REQUIRE(out, "null out");
    VDB_REQUIRE(dim > 0 && dim < (1u << 24), "dim must be in 1..2^24");
    VDB_REQUIRE(dist == VDB_L2SQR || dist == VDB_COSINE, "dist must be 0 (L2Sqr) or 1 (Cosine)");
    require_gpu();
    *out = new vdb_index(device_id, dim, dist, true);
    VDB_API_END
}
int vdb_index_destroy(vdb_index *idx) {
    VDB_API_BEGIN
    if (idx) {
        idx->ix.use_device();
        delete idx;
    }
    VDB_API_END
}
int vdb_index_len(const vdb_index *idx, uint64_t *out) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx && out, "null argument");
    *out = idx->ix.n;
    VDB_API_END
}
int vdb_index_dim(const vdb_index *idx, uint64_t *out) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx && out, "null argument");
    *out = idx->ix.dim;
    VDB_API_END
}
int vdb_index_dist(const vdb_index *idx, int *out) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx && out, "null argument");
    *out = idx->ix.dist;
    VDB_API_END
}
int vdb_index_row(const vdb_index *idx, uint64_t i, float *out) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx && out, "null argument");
    const Index &ix = idx->ix;
    VDB_REQUIRE(i < ix.n, "row index out of bounds");
    ix.use_device();
    if (ix.elem_u8) {  // widened with `as f32` (exact)
        std::vector<uint8_t> b(ix.dim);
        VDB_HIP(hipMemcpy(b.data(), ix.d_rows.as<uint8_t>() + i * ix.dim, ix.dim, hipMemcpyDeviceToHost));
        for (uint64_t j = 0; j < ix.dim; j++) out[j] = (float)b[j];
        return VDB_OK;
    }
    VDB_HIP(hipMemcpy(out, ix.d_rows.as<float>() + i * ix.dim, ix.dim * sizeof(float), hipMemcpyDeviceToHost));
    VDB_API_END
}
int vdb_index_row_u8(const vdb_index *idx, uint64_t i, uint8_t *out) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx && out, "null argument");
    const Index &ix = idx->ix;
    VDB_REQUIRE(ix.elem_u8, "not a VecSet<u8> index");
    VDB_REQUIRE(i < ix.n, "row index out of bounds");
    ix.use_device();
    VDB_HIP(hipMemcpy(out, ix.d_rows.as<uint8_t>() + i * ix.dim, ix.dim, hipMemcpyDeviceToHost));
    VDB_API_END
}
int vdb_index_is_u8(const vdb_index *idx, int *out) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx && out, "null argument");
    *out = idx->ix.elem_u8 ? 1 : 0;
    VDB_API_END
}

static void add_common(Index &ix, const float *rows, uint64_t n, uint64_t *first_id, bool on_device) {
    VDB_REQUIRE(rows || n == 0, "null rows");
    VDB_REQUIRE(!ix.elem_u8, "this index stores VecSet<u8> rows: add them with vdb_index_add_u8");
    if (ix.ivf.present && n) ivf_clear(ix);  // IVFIndex has no add (built from_vec_set only): the clusters go stale
    // MetadataVecTable::add / batch_add clear the PQ table before they touch the index (metadata_vec_table.rs:65,77):
    // the codes cover the old rows only, a later knn_pq must fail with "needs a PQ table", not scan past d_codes
    if (ix.pq.present && n) pq_clear(ix);
    if (first_id) *first_id = ix.n;
    if (ix.hnsw.present) {
        // DynamicIndex::add on the HNSW arm (dynamic_index.rs:47-52): HNSWIndex::add per row
        std::vector<float> tmp;
        const float *h = rows;
        if (on_device) {
            tmp.resize(n * ix.dim);
            ix.use_device();
            VDB_HIP(hipMemcpy(tmp.data(), rows, n * ix.dim * sizeof(float), hipMemcpyDeviceToHost));
            h = tmp.data();
        }
        hnsw_insert_rows(ix, h, n);
        return;
    }
    ix.add_rows(rows, n, on_device);
}
int vdb_index_add(vdb_index *idx, const float *rows, uint64_t n, uint64_t *first_id) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    add_common(idx->ix, rows, n, first_id, false);
    VDB_API_END
}
int vdb_index_add_device(vdb_index *idx, const void *d_rows, uint64_t n, uint64_t *first_id) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    add_common(idx->ix, static_cast<const float *>(d_rows), n, first_id, true);
    VDB_API_END
}
int vdb_index_swap_remove(vdb_index *idx, uint64_t i) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    VDB_REQUIRE(!idx->ix.hnsw.present, "swap_remove needs a Flat index (clear the HNSW graph first)");
    VDB_REQUIRE(!idx->ix.pq.present, "swap_remove invalidates the PQ table: clear it first");
    VDB_REQUIRE(!idx->ix.ivf.present, "swap_remove invalidates the IVF clusters: clear them first");
    idx->ix.swap_remove(i);
    VDB_API_END
}
int vdb_index_set_id_offset(vdb_index *idx, uint64_t offset) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    idx->ix.id_offset = offset;
    VDB_API_END
}

int vdb_calc_dist(int device_id, const float *a, const float *b, uint64_t n, int dist, float *out) {
    VDB_API_BEGIN
    VDB_REQUIRE(a && b && out, "null argument");
    VDB_REQUIRE(n > 0 && n < (1u << 24), "bad length");
    VDB_REQUIRE(dist == VDB_L2SQR || dist == VDB_COSINE, "dist must be 0 or 1");
    require_gpu();
    VDB_HIP(hipSetDevice(device_id));
    DevBuf buf;
    buf.reserve((2 * n + 4) * sizeof(float));
    float *d = buf.as<float>();
    VDB_HIP(hipMemcpy(d, a, n * sizeof(float), hipMemcpyHostToDevice));       // "row"
    VDB_HIP(hipMemcpy(d + n, b, n * sizeof(float), hipMemcpyHostToDevice));   // "query"
    float *sq = d + 2 * n;  // [0]=|a|^2 [1]=|b|^2 [2]=out
    launch_row_sqnorm(d, 1, (uint32_t)n, sq, nullptr);
    launch_row_sqnorm(d + n, 1, (uint32_t)n, sq + 1, nullptr);
    launch_scan_exact(d, 1, (uint32_t)n, d + n, 1, dist == VDB_L2SQR ? MET_L2_DIRECT : MET_COSINE, sq, sq + 1, sq + 2,
                      4, false, nullptr);
    VDB_HIP(hipMemcpy(out, sq + 2, sizeof(float), hipMemcpyDeviceToHost));
    VDB_API_END
}

// ---- u8 scalar (distance/mod.rs:79-95) -----------------------------------------------------------------
// DistanceScalar for u8 converts every element with `as f32` (exact for 0..255) and then runs the f32 folds, so a
// VecSet<u8> index behaves exactly like the f32 index of the converted rows: the u8 entry points convert and forward.
static std::vector<float> widen_u8(const uint8_t *p, uint64_t count) {
    std::vector<float> out(count);
    for (uint64_t i = 0; i < count; i++) out[i] = (float)p[i];
    return out;
}
int vdb_calc_dist_u8(int device_id, const uint8_t *a, const uint8_t *b, uint64_t n, int dist, float *out) {
    VDB_API_BEGIN
    VDB_REQUIRE(a && b && out, "null argument");
    VDB_REQUIRE(n > 0 && n < (1u << 24), "bad length");
    std::vector<float> fa = widen_u8(a, n), fb = widen_u8(b, n);
    int rc = vdb_calc_dist(device_id, fa.data(), fb.data(), n, dist, out);
    if (rc != VDB_OK) return rc;
    VDB_API_END
}
int vdb_index_add_u8(vdb_index *idx, const uint8_t *rows, uint64_t n, uint64_t *first_id) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    VDB_REQUIRE(rows || n == 0, "null rows");
    if (idx->ix.elem_u8) {  // native: one byte per element in HBM
        if (first_id) *first_id = idx->ix.n;
        idx->ix.add_rows(rows, n, false);
        return VDB_OK;
    }
    std::vector<float> f = widen_u8(rows, n * idx->ix.dim);
    add_common(idx->ix, f.data(), n, first_id, false);
    VDB_API_END
}

// ---- Flat ----------------------------------------------------------------------------------------
static void check_query_args(const Index &ix, const void *queries, uint64_t nq, uint64_t dim, const void *out_idx,
                             const void *out_dist) {
    VDB_REQUIRE(dim == ix.dim, "query dimension mismatch: index dim " + std::to_string(ix.dim) + ", got " +
                                   std::to_string(dim));
    VDB_REQUIRE(nq == 0 || (queries && out_idx && out_dist), "null argument");
}

typedef void (*dev_search_fn)(Index &, Workspace &, const float *, uint64_t, uint64_t, uint64_t, uint64_t *, float *,
                              uint64_t *);

// host-pointer front end shared by all searches: stage queries, run, copy results back
static void host_search(Index &ix, const float *queries, uint64_t nq, uint64_t k, uint64_t ef, uint64_t *out_idx,
                        float *out_dist, uint64_t *out_count, dev_search_fn fn) {
    if (nq == 0) return;
    ix.use_device();
    WsLease ws(ix);
    hipStream_t s = ws->stream;
    constexpr uint64_t CHUNK = 16384;
    for (uint64_t q0 = 0; q0 < nq; q0 += CHUNK) {
        uint64_t nb = std::min<uint64_t>(CHUNK, nq - q0);
        uint64_t kk = std::max<uint64_t>(k, 1);
        // ONE device block for the three outputs [ids | distances | counts] and ONE pinned host block [queries | that block]:
        // a call moves two async copies through page-locked memory instead of four pageable ones (each of which the runtime
        // stages and waits for on its own) -- what a db.search()-shaped call of one query pays for besides its kernels
        const size_t qb = nb * ix.dim * sizeof(float), ib = nb * kk * sizeof(uint64_t), db = nb * kk * sizeof(float), cb = nb * sizeof(uint64_t);
        const size_t off_d = ib, off_c = (ib + db + 7) & ~size_t(7), ob = off_c + cb, q_pad = (qb + 15) & ~size_t(15);
        const bool staged = q_pad + ob <= (size_t(8) << 20);
        ws->q.reserve(qb);
        ws->out_idx.reserve(ob);
        char *d_out = ws->out_idx.as<char>();
        char *h = staged ? static_cast<char *>(ws->pinned(q_pad + ob)) : nullptr;
        if (staged) {
            std::memcpy(h, queries + q0 * ix.dim, qb);
            VDB_HIP(hipMemcpyAsync(ws->q.p, h, qb, hipMemcpyHostToDevice, s));
        } else {
            VDB_HIP(hipMemcpyAsync(ws->q.p, queries + q0 * ix.dim, qb, hipMemcpyHostToDevice, s));
        }
        fn(ix, *ws, ws->q.as<float>(), nb, k, ef, reinterpret_cast<uint64_t *>(d_out), reinterpret_cast<float *>(d_out + off_d),
           reinterpret_cast<uint64_t *>(d_out + off_c));
        // (fn may have used the pinned block for its own flags and has synchronised after reading them: the block is free again)
        if (staged) {
            h = static_cast<char *>(ws->pinned(q_pad + ob));
            VDB_HIP(hipMemcpyAsync(h + q_pad, d_out, ob, hipMemcpyDeviceToHost, s));
            VDB_SYNC(s);
            if (k) {
                std::memcpy(out_idx + q0 * k, h + q_pad, nb * k * sizeof(uint64_t));
                std::memcpy(out_dist + q0 * k, h + q_pad + off_d, nb * k * sizeof(float));
            }
            if (out_count) std::memcpy(out_count + q0, h + q_pad + off_c, cb);
        } else {
            if (k) {
                VDB_HIP(hipMemcpyAsync(out_idx + q0 * k, d_out, nb * k * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
                VDB_HIP(hipMemcpyAsync(out_dist + q0 * k, d_out + off_d, nb * k * sizeof(float), hipMemcpyDeviceToHost, s));
            }
            std::vector<uint64_t> cnt(nb);
            VDB_HIP(hipMemcpyAsync(cnt.data(), d_out + off_c, cb, hipMemcpyDeviceToHost, s));
            VDB_SYNC(s);
            if (out_count) std::memcpy(out_count + q0, cnt.data(), cb);
        }
        ix.prof_collect(*ws);
    }
}

static void flat_dev(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, uint64_t k, uint64_t, uint64_t *d_idx,
                     float *d_dist, uint64_t *d_cnt) {
    ix.flat_knn_device(ws, d_q, nq, k, d_idx, d_dist, d_cnt);
}

int vdb_flat_knn(vdb_index *idx, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t *out_idx,
                 float *out_dist, uint64_t *out_count) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    check_query_args(idx->ix, queries, nq, dim, out_idx, out_dist);
    Index &ix = idx->ix;
    if (nq > 0 && k > 0 && ix.flat_small_applies(nq, k)) {
        // The db.search() shape (pyo3/mod.rs:199-214): a few queries, a small table.  One launch, and no copy engine on either
        // side of it: the queries go through ONE block of pinned host memory (read by the kernel over PCIe when only a
        // handful of workgroups want them, else one async copy), the kernel writes ids / distances / counts straight back
        // into that block, and the stream synchronisation that ends the call is the only wait.
        ix.use_device();
        WsLease ws(ix);
        const size_t qb = nq * dim * sizeof(float), ib = nq * k * sizeof(uint64_t), db = nq * k * sizeof(float);
        const size_t off_i = (qb + 15) & ~size_t(15), off_d = off_i + ib, off_c = (off_d + db + 7) & ~size_t(7);
        char *h = static_cast<char *>(ws->pinned(off_c + nq * sizeof(uint64_t)));
        std::memcpy(h, queries, qb);
        const float *q = reinterpret_cast<const float *>(h);
        const uint64_t n_wg = (ix.n + flat_small_rows_per_wg(ix.n, ix.num_cu) - 1) / flat_small_rows_per_wg(ix.n, ix.num_cu);
        if (n_wg * nq > 64) {  // many readers: stage the queries in HBM once
            ws->q.reserve(qb);
            VDB_HIP(hipMemcpyAsync(ws->q.p, h, qb, hipMemcpyHostToDevice, ws->stream));
            q = ws->q.as<float>();
        }
        ix.flat_small_device(*ws, q, nq, k, reinterpret_cast<uint64_t *>(h + off_i), reinterpret_cast<float *>(h + off_d),
                             reinterpret_cast<uint64_t *>(h + off_c));
        VDB_SYNC(ws->stream);
        ix.prof_collect(*ws);
        std::memcpy(out_idx, h + off_i, ib);
        std::memcpy(out_dist, h + off_d, db);
        if (out_count) std::memcpy(out_count, h + off_c, nq * sizeof(uint64_t));
        return VDB_OK;
    }
    host_search(idx->ix, queries, nq, k, 0, out_idx, out_dist, out_count, flat_dev);
    VDB_API_END
}

// device-pointer front end shared by all searches (queries and outputs already on the index's GPU)
static void device_search(Index &ix, const void *d_queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef,
                          void *d_out_idx, void *d_out_dist, void *d_out_count, void *stream, dev_search_fn fn) {
    check_query_args(ix, d_queries, nq, dim, d_out_idx, d_out_dist);
    VDB_REQUIRE(nq == 0 || d_out_count, "null out_count");
    VDB_REQUIRE(nq <= 32768, "at most 32768 queries per device call");
    if (nq == 0) return;
    ix.use_device();
    WsLease ws(ix);
    // order after whatever produced the queries on the caller's stream
    VDB_SYNC(static_cast<hipStream_t>(stream));
    fn(ix, *ws, static_cast<const float *>(d_queries), nq, k, ef, static_cast<uint64_t *>(d_out_idx),
       static_cast<float *>(d_out_dist), static_cast<uint64_t *>(d_out_count));
    VDB_SYNC(ws->stream);
    ix.prof_collect(*ws);
}

int vdb_flat_knn_u8(vdb_index *idx, const uint8_t *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t *out_idx,
                    float *out_dist, uint64_t *out_count) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    VDB_REQUIRE(nq == 0 || queries, "null argument");
    std::vector<float> f = widen_u8(queries, nq * dim);
    check_query_args(idx->ix, f.data(), nq, dim, out_idx, out_dist);
    host_search(idx->ix, f.data(), nq, k, 0, out_idx, out_dist, out_count, flat_dev);
    VDB_API_END
}

int vdb_flat_knn_device(vdb_index *idx, const void *d_queries, uint64_t nq, uint64_t dim, uint64_t k,
                        void *d_out_idx, void *d_out_dist, void *d_out_count, void *stream) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    device_search(idx->ix, d_queries, nq, dim, k, 0, d_out_idx, d_out_dist, d_out_count, stream, flat_dev);
    VDB_API_END
}

// ---- a Flat call in two halves: software pipelining of independent query batches ----------------------------------------------
// vdb_flat_knn_device returns when the batch is answered: the host reads the certification flags (one byte per query) to
// decide whether anything must be redone.  Between that read and the first kernel of the next call the GPU idles, and the
// per-query stages at both ends of a call (query preparation, threshold sample and selection in front of the corpus pass;
// the exact stage behind it) cannot overlap each other.  begin() enqueues the whole pipeline on a workspace stream -- ordered
// behind the caller's stream through an event, not a host wait -- and returns; end() waits, reads the flags, redoes what was
// not certified.  Two batches in flight (begin(i+1) before end(i)) keep the corpus passes back to back and let the exact
// stage of batch i run beside the front stages of batch i+1.  Results are the synchronous call's, bit for bit.
struct vdb_pending {
    vdb_index *idx = nullptr;
    std::unique_ptr<Workspace> ws;
    FlatPending p;
};

int vdb_flat_knn_device_begin(vdb_index *idx, const void *d_queries, uint64_t nq, uint64_t dim, uint64_t k, void *d_out_idx,
                              void *d_out_dist, void *d_out_count, void *stream, vdb_pending **out) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx && out, "null argument");
    Index &ix = idx->ix;
    check_query_args(ix, d_queries, nq, dim, d_out_idx, d_out_dist);
    VDB_REQUIRE(nq == 0 || d_out_count, "null out_count");
    VDB_REQUIRE(nq <= 32768, "at most 32768 queries per device call");
    ix.use_device();
    std::unique_ptr<vdb_pending> pd(new vdb_pending);
    pd->idx = idx;
    pd->ws = ix.acquire_ws();
    try {
        Workspace &ws = *pd->ws;
        if (!ws.order_ev) VDB_HIP(hipEventCreateWithFlags(&ws.order_ev, hipEventDisableTiming));
        VDB_HIP(hipEventRecord(ws.order_ev, static_cast<hipStream_t>(stream)));  // whatever produced the queries / last read the outputs
        VDB_HIP(hipStreamWaitEvent(ws.stream, ws.order_ev, 0));
        ix.flat_knn_enqueue(ws, static_cast<const float *>(d_queries), nq, k, static_cast<uint64_t *>(d_out_idx),
                            static_cast<float *>(d_out_dist), static_cast<uint64_t *>(d_out_count), true, 0, pd->p);
    } catch (...) {
        (void)hipStreamSynchronize(pd->ws->stream);
        ix.release_ws(std::move(pd->ws));
        throw;
    }
    *out = pd.release();
    VDB_API_END
}

// completes the call begun with `pending` and releases it (also on error); the outputs are valid when this returns
int vdb_flat_knn_device_end(vdb_pending *pending) {
    VDB_API_BEGIN
    VDB_REQUIRE(pending, "null pending call");
    std::unique_ptr<vdb_pending> pd(pending);
    Index &ix = pd->idx->ix;
    ix.use_device();
    try {
        ix.flat_knn_finish(*pd->ws, pd->p);
        VDB_SYNC(pd->ws->stream);
        ix.prof_collect(*pd->ws);
    } catch (...) {
        (void)hipStreamSynchronize(pd->ws->stream);
        ix.release_ws(std::move(pd->ws));
        throw;
    }
    ix.release_ws(std::move(pd->ws));
    VDB_API_END
}

int vdb_flat_shortlist_keys(vdb_index *idx, const float *queries, uint64_t nq, uint64_t dim, int tier, float *out_keys,
                            float *out_qsq, float *out_qerr, float *out_dx4) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx && queries && out_keys, "null argument");
    Index &ix = idx->ix;
    VDB_REQUIRE(dim == ix.dim, "query dimension mismatch");
    VDB_REQUIRE(tier >= 0 && tier <= 2, "tier must be 0 (fp16 operands), 1 (split-bf16 operands) or 2 (8-bit operands, lower-bound keys)");
    ix.use_device();
    WsLease ws(ix);
    ws->q.reserve(nq * ix.dim * sizeof(float));
    VDB_HIP(hipMemcpyAsync(ws->q.p, queries, nq * ix.dim * sizeof(float), hipMemcpyHostToDevice, ws->stream));
    ix.flat_debug_keys(*ws, ws->q.as<float>(), nq, tier, out_keys, out_qsq, out_qerr, out_dx4);
    VDB_API_END
}

int vdb_flat_set_mode(vdb_index *idx, int mode) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    VDB_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0, 1 or 2");
    idx->ix.flat_mode = mode;
    VDB_API_END
}
int vdb_set_param(vdb_index *idx, const char *name, int64_t value) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx && name, "null argument");
    std::string n(name);
    if (n == "mfma_variant")
        mfma_set_variant((int)value);
    else if (n == "flat_share")
        mfma_set_share((int)value);
    else if (n == "flat_gemm")  // 0 auto (more than 64 queries), 1 off, 2 forced
        idx->ix.flat_gemm_mode = (int)value;
    else if (n == "hnsw_dma")
        hnsw_set_dma((int)value);
    else if (n == "hnsw_half")
        hnsw_set_half((int)value);
    else if (n == "ivf_half")
        ivf_set_half((int)value);
    else if (n == "ivf_q8")
        ivf_set_q8((int)value);
    else if (n == "hnsw_build_gpu")
        hnsw_set_build_gpu((int)value);
    else if (n == "hnsw_pool_cap")
        hnsw_set_pool_cap((int)value);
    else if (n == "pq_adc_fast")
        pq_set_adc_fast((int)value);
    else if (n == "pq_sample16")
        pq_set_adc16_sample((int)value);
    else if (n == "pq_adc16")
        pq_set_adc16((int)value);
    else if (n == "pq_adc8_sliced")  // 8-bit codes: 0 = 16 queries per pass on sliced one-byte tables (k_pq_adc8x16), 1 = one query per pass (k_pq_adc8), 2 = 8 per pass on sliced 16-bit tables
        pq_set_adc8_sliced((int)value);
    else if (n == "flat_sample_thin")
        mfma_set_sample_thin((int)value);
    else if (n == "flat_gemm_tw")
        gemm_set_tw((int)value);
    else if (n == "flat_gemm_nt")
        gemm_set_nt((int)value);
    else if (n == "flat_gemm_zigzag")
        gemm_set_zigzag((int)value);
    else if (n == "flat_gemm_block_rows") {
        VDB_REQUIRE(value >= 0, "flat_gemm_block_rows must be >= 0");
        gemm_set_block_rows((uint64_t)value);
    }
    else if (n == "flat_gemm_stagger")
        gemm_set_stagger((int)value);
    else if (n == "flat_gemm_debug")
        idx->ix.flat_gemm_debug = (int)value;
    else if (n == "flat_tail")  // exact stage of the Flat pipeline: 0 fused launch when the shortlist fits 64 rows, 1 separate kernels
        idx->ix.flat_tail_mode = (int)value;
    else if (n == "flat_tail_lb_nw")  // exact stage of the 8-bit pass: waves per query (0 auto, 8 / 4 / 2 / 1)
        flat_tail_lb_set_nw((int)value);
    else if (n == "flat_small")  // one-launch search of small tables (k_small.hip): 0 auto, 1 off, 2 whenever the shape allows
        idx->ix.flat_small_mode = (int)value;
    else if (n == "flat_small_max_rows")
        idx->ix.flat_small_max_rows = (uint64_t)value;
    else if (n == "flat_half")  // fp16 first pass of large query batches: 0 auto, 1 off, 2 on regardless of the redo rate
        idx->ix.flat_half_mode = (int)value;
    else if (n == "flat_i8")  // 8-bit first pass (L2Sqr): 0 auto, 1 off, 2 on regardless of the redo rate
        idx->ix.flat_i8_mode = (int)value;
    else if (n == "flat_i8_rows") {  // rows its exact stage may walk per query before giving up (multiple of 64)
        VDB_REQUIRE(value >= 64 && value <= 8192 && value % 64 == 0, "flat_i8_rows must be a multiple of 64 in [64, 8192]");
        idx->ix.flat_i8_kprime = (uint32_t)value;
    }
    else if (n == "debug_alloc_fail_over")  // (testing aid, process-wide) device allocations of at least this many bytes fail; 0 = off
        devbuf_fail_over() = (size_t)value;
    else if (n == "flat_i8_unit_min")  // threshold sample of the 8-bit pass: one value per sampled unit when the units are many (0 auto, 1 off, 2 on from 2 x rank units: tests)
        idx->ix.flat_i8_unit_min = (int)value;
    else if (n == "flat_i8_full")  // second 8-bit attempt of a handful of queries: all candidates at once (0 on, 1 off: rounds of 63 rows)
        idx->ix.flat_i8_full = (int)value;
    else if (n == "flat_i8_refine") {  // hit keys of the 8-bit pass tightened from the fp16 row image before the walk: 0 auto, 1 off, 2 always
        idx->ix.flat_i8_refine = (int)value;
        idx->ix.i8_refine_on = 0;
        idx->ix.i8_refine_calls = 0;
    }
    else if (n == "flat_i8_second")  // second 8-bit attempt with thresholds from the first walk's k-th distances: 0 on, 1 off
        idx->ix.flat_i8_second = (int)value;
    else if (n == "flat_i8_stats") {  // (measurement) collect per-query rounds / hits of the 8-bit pass's exact stage; setting it resets them
        idx->ix.flat_i8_stats = (int)value;
        for (auto &h : idx->ix.i8_rounds_hist) h = 0;
        idx->ix.i8_hits_sum = 0;
        idx->ix.i8_hits_max = 0;
        idx->ix.i8_stat_queries = 0;
    }
    else if (n == "flat_i8_hits") {  // expected hits per query its threshold sample aims at
        VDB_REQUIRE(value >= 256 && value <= 4096, "flat_i8_hits must be in [256, 4096]");
        idx->ix.flat_i8_hits = (uint32_t)value;
    }
    else if (n == "flat_gemm8_sample_res")  // threshold sample of the 8-bit pass: 0 = the filter's kernel form, 1 = chunked staging (no whole-image load first)
        gemm8_set_sample_res((int)value);
    else if (n == "flat_gemm8_nt")
        gemm8_set_nt((int)value);
    else if (n == "flat_gemm8_kc")
        gemm8_set_kc((int)value);
    else if (n == "flat_gemm8_burst")
        gemm8_set_burst((int)value);
    else if (n == "flat_gemm8_res")
        gemm8_set_res((int)value);
    else if (n == "flat_gemm8_coop")
        gemm8_set_coop((int)value);
    else if (n == "flat_gemm8_grid")  // measurement: workgroups of the cooperative filter (64 / 128 / 192 / 256; 0 = one per CU)
        gemm8_set_grid((int)value);
    else if (n == "flat_gemm_coop")
        gemm_set_coop((int)value);
    else if (n == "flat_half_kmul") {  // its shortlist: max(64, kmul * k) rows per query
        VDB_REQUIRE(value >= 1 && value <= 64, "flat_half_kmul must be in [1, 64]");
        idx->ix.flat_half_kmul = (uint32_t)value;
    }
    else
        throw Error(VDB_ERR_INVALID, "unknown parameter " + n);
    VDB_API_END
}
int vdb_index_prepare(vdb_index *idx, int all_tiers) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    require_gpu();
    idx->ix.prepare_flat(all_tiers != 0);
    VDB_API_END
}
int vdb_flat_fallback_count(const vdb_index *idx, uint64_t *out) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx && out, "null argument");
    *out = idx->ix.fallback_count.load();
    VDB_API_END
}

int vdb_get_stat(const vdb_index *idx, const char *name, uint64_t *out) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx && name && out, "null argument");
    std::string n(name);
    if (n == "flat_fallback")
        *out = idx->ix.fallback_count.load();
    else if (n == "flat_half_queries")
        *out = idx->ix.half_queries.load();
    else if (n == "flat_half_redo")
        *out = idx->ix.half_redo.load();
    else if (n == "flat_half_valid")
        *out = idx->ix.half_valid ? 1 : 0;
    else if (n == "flat_gemm_coop_sets")  // the same of the fp16 / split-bf16 filter kernel
        *out = gemm_last_coop();
    else if (n == "flat_gemm8_coop_sets")  // workgroups per cooperative set of the most recent 8-bit filter launch of the process (0: none)
        *out = gemm8_last_coop();
    else if (n == "flat_i8_queries")
        *out = idx->ix.i8_queries.load();
    else if (n == "flat_i8_redo")
        *out = idx->ix.i8_redo.load();
    else if (n == "flat_i8_valid")
        *out = idx->ix.i8_valid.load() ? 1 : 0;
    else if (n.rfind("flat_i8_rounds_", 0) == 0 && n.size() == 16 && n[15] >= '0' && n[15] <= '8')  // queries whose exact stage walked N rounds (8: 8 or more)
        *out = idx->ix.i8_rounds_hist[n[15] - '0'].load();
    else if (n == "mirror_alloc_failures")  // mirrors of this index whose allocation failed (the tier was left to the next one)
        *out = idx->ix.mirror_alloc_failures.load();
    else if (n == "flat_i8_second_queries")  // queries that took the second 8-bit attempt / that it passed on to the fp16 tier
        *out = idx->ix.i8_second_queries.load();
    else if (n == "flat_i8_second_redo")
        *out = idx->ix.i8_second_redo.load();
    else if (n == "flat_i8_hits_sum")
        *out = idx->ix.i8_hits_sum.load();
    else if (n == "flat_i8_hits_max")
        *out = idx->ix.i8_hits_max.load();
    else if (n == "flat_i8_stat_queries")
        *out = idx->ix.i8_stat_queries.load();
    else if (n == "flat_i8_rows_walked")
        *out = idx->ix.i8_rows_walked.load();
    else if (n == "flat_bf16_mirror")
        *out = idx->ix.tiled_built ? 1 : 0;
    else if (n == "hnsw_heap_walk_queries")
        *out = idx->ix.hnsw.heap_walk_queries.load();
    else if (n == "hnsw_half_dropped")
        *out = idx->ix.hnsw.last_half_dropped.load();
    else if (n == "ivf_last_offers")
        *out = idx->ix.ivf.last_offers.load();
    else if (n == "ivf_last_rows_fetched_q8")
        *out = idx->ix.ivf.last_rows_fetched_q8.load();
    else if (n == "ivf_last_kept_q8")
        *out = idx->ix.ivf.last_kept_q8.load();
    else if (n == "ivf_last_kept")
        *out = idx->ix.ivf.last_kept.load();
    else if (n == "pq_adc16_queries")
        *out = idx->ix.pq.adc16_queries.load();
    else if (n == "flat_i8_refine_queries")
        *out = idx->ix.i8_refine_queries.load();
    else if (n == "flat_i8_refine_on")
        *out = (uint64_t)idx->ix.i8_refine_on.load();
    else if (n == "pq_q8_overflow")
        *out = idx->ix.pq.q8_overflow.load();
    else if (n == "pq_q8_short")
        *out = idx->ix.pq.q8_short.load();
    else if (n == "pq_q8_hits_sum")
        *out = idx->ix.pq.q8_hits_sum.load();
    else if (n == "pq_q8_hits_max")
        *out = idx->ix.pq.q8_hits_max.load();
    else if (n == "hbm_bytes_per_row")
        *out = idx->ix.hbm_bytes_per_row();
    else
        throw Error(VDB_ERR_INVALID, "unknown statistic " + n);
    VDB_API_END
}

// ---- PQ -------------------------------------------------------------------------------------------
int vdb_pq_attach(vdb_index *idx, uint64_t n_bits, uint64_t m, const float *centroids, const uint8_t *codes) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx && centroids, "null argument");
    pq_attach(idx->ix, n_bits, m, centroids, codes);
    VDB_API_END
}
int vdb_pq_build(vdb_index *idx, uint64_t n_bits, uint64_t m, uint64_t train_n, uint64_t max_iter, float tol,
                 uint64_t seed) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    pq_build(idx->ix, n_bits, m, train_n, max_iter, tol, seed);
    VDB_API_END
}
int vdb_pq_clear(vdb_index *idx) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    pq_clear(idx->ix);
    VDB_API_END
}
int vdb_pq_has(const vdb_index *idx, int *out) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx && out, "null argument");
    *out = idx->ix.pq.present ? 1 : 0;
    VDB_API_END
}
int vdb_pq_info(const vdb_index *idx, uint64_t *n_bits, uint64_t *m, uint64_t *enc_dim) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    VDB_REQUIRE(idx->ix.pq.present, "no PQ table");
    if (n_bits) *n_bits = idx->ix.pq.n_bits;
    if (m) *m = idx->ix.pq.m;
    if (enc_dim) *enc_dim = idx->ix.pq.enc_dim;
    VDB_API_END
}
int vdb_pq_export(const vdb_index *idx, float *centroids, uint8_t *codes) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    const Index &ix = idx->ix;
    VDB_REQUIRE(ix.pq.present, "no PQ table");
    if (centroids) std::memcpy(centroids, ix.pq.h_centroids.data(), ix.pq.h_centroids.size() * sizeof(float));
    if (codes && ix.n) {
        ix.use_device();
        VDB_HIP(hipMemcpy(codes, ix.pq.d_codes.p, ix.n * ix.pq.enc_dim, hipMemcpyDeviceToHost));
    }
    VDB_API_END
}

static void flat_pq_dev(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, uint64_t k, uint64_t ef,
                        uint64_t *d_idx, float *d_dist, uint64_t *d_cnt) {
    flat_knn_pq_device(ix, ws, d_q, nq, k, ef, d_idx, d_dist, d_cnt);
}
int vdb_flat_knn_pq(vdb_index *idx, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef,
                    uint64_t *out_idx, float *out_dist, uint64_t *out_count) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    check_query_args(idx->ix, queries, nq, dim, out_idx, out_dist);
    VDB_REQUIRE(idx->ix.pq.present, "knn_pq needs a PQ table (vdb_pq_build / vdb_pq_attach)");
    host_search(idx->ix, queries, nq, k, ef, out_idx, out_dist, out_count, flat_pq_dev);
    VDB_API_END
}

int vdb_flat_knn_pq_device(vdb_index *idx, const void *d_queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef,
                           void *d_out_idx, void *d_out_dist, void *d_out_count, void *stream) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    VDB_REQUIRE(idx->ix.pq.present, "knn_pq needs a PQ table (vdb_pq_build / vdb_pq_attach)");
    device_search(idx->ix, d_queries, nq, dim, k, ef, d_out_idx, d_out_dist, d_out_count, stream, flat_pq_dev);
    VDB_API_END
}

// PQTable::create_lookup (pq_table.rs:195-224) / the ADC adapter over every row (pq_table.rs:239-301)
static const float *stage_queries(Index &ix, Workspace &ws, const float *queries, uint64_t nq) {
    ws.q.reserve(std::max<uint64_t>(nq, 1) * ix.dim * sizeof(float));
    VDB_HIP(hipMemcpyAsync(ws.q.p, queries, nq * ix.dim * sizeof(float), hipMemcpyHostToDevice, ws.stream));
    return ws.q.as<float>();
}
int vdb_pq_create_lookup(vdb_index *idx, const float *queries, uint64_t nq, uint64_t dim, float *out_lut, float *out_qcache) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    Index &ix = idx->ix;
    VDB_REQUIRE(dim == ix.dim, "query dimension mismatch");  // assert_eq!(query.len(), self.dim) pq_table.rs:196
    VDB_REQUIRE(nq == 0 || queries, "null argument");
    VDB_REQUIRE(ix.pq.present, "create_lookup needs a PQ table (vdb_pq_build / vdb_pq_attach)");
    VDB_REQUIRE(nq <= 65535, "at most 65535 queries per call");
    if (nq == 0) return VDB_OK;
    ix.use_device();
    WsLease ws(ix);
    pq_export_lookup(ix, *ws, stage_queries(ix, *ws, queries, nq), nq, out_lut, out_qcache);
    VDB_API_END
}
int vdb_pq_adc_all(vdb_index *idx, const float *queries, uint64_t nq, uint64_t dim, float *out) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    Index &ix = idx->ix;
    VDB_REQUIRE(dim == ix.dim, "query dimension mismatch");
    VDB_REQUIRE(nq == 0 || (queries && out), "null argument");
    VDB_REQUIRE(ix.pq.present, "ADC needs a PQ table (vdb_pq_build / vdb_pq_attach)");
    VDB_REQUIRE(nq <= 65535, "at most 65535 queries per call");
    if (nq == 0) return VDB_OK;
    ix.use_device();
    WsLease ws(ix);
    pq_export_adc_all(ix, *ws, stage_queries(ix, *ws, queries, nq), nq, out);
    VDB_API_END
}

// ---- row-sharded knn_pq ------------------------------------------------------------------
int vdb_flat_knn_pq_shard(vdb_index *idx, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef,
                          uint64_t *out_adc_keys, uint64_t *out_exact_keys) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    Index &ix = idx->ix;
    check_query_args(ix, queries, nq, dim, out_adc_keys, out_exact_keys);
    VDB_REQUIRE(ix.pq.present, "knn_pq needs a PQ table (vdb_pq_build / vdb_pq_attach)");
    VDB_REQUIRE(nq <= 32768, "at most 32768 queries per shard call");
    if (nq == 0) return VDB_OK;
    const uint64_t efg = std::max(ef, k);
    ix.use_device();
    WsLease ws(ix);
    hipStream_t s = ws->stream;
    ws->q.reserve(nq * ix.dim * sizeof(float));
    ws->out_idx.reserve(2 * nq * std::max<uint64_t>(efg, 1) * sizeof(uint64_t));
    uint64_t *d_adc = ws->out_idx.as<uint64_t>(), *d_exact = d_adc + nq * efg;
    VDB_HIP(hipMemcpyAsync(ws->q.p, queries, nq * ix.dim * sizeof(float), hipMemcpyHostToDevice, s));
    flat_knn_pq_shard_device(ix, *ws, ws->q.as<float>(), nq, k, ef, d_adc, d_exact);
    VDB_HIP(hipMemcpyAsync(out_adc_keys, d_adc, nq * efg * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    VDB_HIP(hipMemcpyAsync(out_exact_keys, d_exact, nq * efg * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    VDB_SYNC(s);
    ix.prof_collect(*ws);
    VDB_API_END
}

int vdb_flat_knn_pq_shard_device(vdb_index *idx, const void *d_queries, uint64_t nq, uint64_t dim, uint64_t k,
                                 uint64_t ef, void *d_out_adc_keys, void *d_out_exact_keys, void *stream) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    Index &ix = idx->ix;
    check_query_args(ix, d_queries, nq, dim, d_out_adc_keys, d_out_exact_keys);
    VDB_REQUIRE(ix.pq.present, "knn_pq needs a PQ table (vdb_pq_build / vdb_pq_attach)");
    VDB_REQUIRE(nq <= 32768, "at most 32768 queries per shard call");
    if (nq == 0) return VDB_OK;
    ix.use_device();
    WsLease ws(ix);
    VDB_SYNC(static_cast<hipStream_t>(stream));
    flat_knn_pq_shard_device(ix, *ws, static_cast<const float *>(d_queries), nq, k, ef,
                             static_cast<uint64_t *>(d_out_adc_keys), static_cast<uint64_t *>(d_out_exact_keys));
    VDB_SYNC(ws->stream);
    ix.prof_collect(*ws);
    VDB_API_END
}

// host merge (no GPU needed): S rows per query sorted by ADC key -> global ADC top-efk -> ResultSet::add replay over
// the exact keys in that order (candidate_pair.rs:61-74,102-108)
int vdb_pq_merge_resort(const uint64_t *adc_keys, const uint64_t *exact_keys, uint64_t n_shards, uint64_t nq,
                        uint64_t efk, uint64_t k, uint64_t *out_idx, float *out_dist, uint64_t *out_count) {
    VDB_API_BEGIN
    VDB_REQUIRE(nq == 0 || (adc_keys && exact_keys && out_idx && out_dist), "null argument");
    VDB_REQUIRE(efk >= k, "pq merge: efk = max(ef, k) must be >= k");
    std::vector<std::pair<uint64_t, uint64_t>> all;
    std::vector<uint64_t> set;  // the ResultSet, ascending pair keys
    for (uint64_t q = 0; q < nq; q++) {
        all.clear();
        for (uint64_t s = 0; s < n_shards; s++)
            for (uint64_t j = 0; j < efk; j++) {
                uint64_t a = adc_keys[(s * nq + q) * efk + j];
                if (a != PAIR_NONE) all.push_back({a, exact_keys[(s * nq + q) * efk + j]});
            }
        std::sort(all.begin(), all.end());
        if (all.size() > efk) all.resize(efk);
        set.clear();
        for (auto &pr : all) {
            uint64_t e = pr.second;
            if (k == 0) break;
            if (set.size() >= k) {
                if (uint32_t(e >> 32) >= uint32_t(set.back() >> 32)) continue;  // not strictly closer than the worst
                set.pop_back();
            }
            set.insert(std::lower_bound(set.begin(), set.end(), e), e);
        }
        for (uint64_t j = 0; j < k; j++) {
            bool ok = j < set.size();
            out_idx[q * k + j] = ok ? uint64_t(uint32_t(set[j])) : 0;
            out_dist[q * k + j] = ok ? f32_from_orderable(uint32_t(set[j] >> 32)) : 0.0f;
        }
        if (out_count) out_count[q] = set.size();
    }
    VDB_API_END
}

int vdb_pq_merge_resort_device(vdb_index *idx, const void *d_adc_keys, const void *d_exact_keys, uint64_t n_shards,
                               uint64_t nq, uint64_t efk, uint64_t k, void *d_out_idx, void *d_out_dist,
                               void *d_out_count, void *stream) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx && d_adc_keys && d_exact_keys && d_out_idx && d_out_dist && d_out_count, "null argument");
    VDB_REQUIRE(nq <= 65535 && n_shards >= 1 && n_shards <= 1024, "too many queries or shards for one call");
    Index &ix = idx->ix;
    ix.use_device();
    WsLease ws(ix);
    VDB_SYNC(static_cast<hipStream_t>(stream));
    pq_merge_resort_device(ix, *ws, static_cast<const uint64_t *>(d_adc_keys),
                           static_cast<const uint64_t *>(d_exact_keys), n_shards, nq, efk, k,
                           static_cast<uint64_t *>(d_out_idx), static_cast<float *>(d_out_dist),
                           static_cast<uint64_t *>(d_out_count));
    VDB_SYNC(ws->stream);
    VDB_API_END
}

// ---- IVF (index_algorithm/ivf_index.rs) --------------------------------------------------------------
int vdb_ivf_build(vdb_index *idx, uint64_t k_clusters, uint64_t train_n, uint64_t max_iter, float tol, uint64_t seed) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    ivf_build(idx->ix, k_clusters, train_n, max_iter, tol, seed);
    VDB_API_END
}
int vdb_ivf_attach(vdb_index *idx, uint64_t k_clusters, const float *centroids, const uint64_t *assign) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    ivf_attach(idx->ix, k_clusters, centroids, assign);
    VDB_API_END
}
int vdb_ivf_clear(vdb_index *idx) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    ivf_clear(idx->ix);
    VDB_API_END
}
int vdb_ivf_info(const vdb_index *idx, int *present, uint64_t *k_clusters, uint64_t *default_n_probes) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    if (present) *present = idx->ix.ivf.present ? 1 : 0;
    if (k_clusters) *k_clusters = idx->ix.ivf.k;
    if (default_n_probes) *default_n_probes = idx->ix.ivf.default_n_probes;
    VDB_API_END
}
int vdb_ivf_export(vdb_index *idx, float *centroids, uint64_t *assign) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    ivf_export(idx->ix, centroids, assign);
    VDB_API_END
}
static void ivf_dev(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, uint64_t k, uint64_t n_probes,
                    uint64_t *d_idx, float *d_dist, uint64_t *d_cnt) {
    ivf_knn_device(ix, ws, d_q, nq, k, n_probes, d_idx, d_dist, d_cnt);
}
int vdb_ivf_knn(vdb_index *idx, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t n_probes,
                uint64_t *out_idx, float *out_dist, uint64_t *out_count) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    check_query_args(idx->ix, queries, nq, dim, out_idx, out_dist);
    VDB_REQUIRE(idx->ix.ivf.present, "knn needs an IVF index (vdb_ivf_build / vdb_ivf_attach)");
    if (n_probes == 0) n_probes = idx->ix.ivf.default_n_probes;
    host_search(idx->ix, queries, nq, k, n_probes, out_idx, out_dist, out_count, ivf_dev);
    VDB_API_END
}

int vdb_ivf_knn_device(vdb_index *idx, const void *d_queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t n_probes,
                       void *d_out_idx, void *d_out_dist, void *d_out_count, void *stream) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    VDB_REQUIRE(idx->ix.ivf.present, "knn needs an IVF index (vdb_ivf_build / vdb_ivf_attach)");
    if (n_probes == 0) n_probes = idx->ix.ivf.default_n_probes;
    device_search(idx->ix, d_queries, nq, dim, k, n_probes, d_out_idx, d_out_dist, d_out_count, stream, ivf_dev);
    VDB_API_END
}

// ---- HNSW -----------------------------------------------------------------------------------------
int vdb_hnsw_build(vdb_index *idx, uint64_t M, uint64_t ef_construction, uint64_t seed, uint64_t batch,
                   int nthreads) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    hnsw_build(idx->ix, M, ef_construction, seed, batch, nthreads);
    VDB_API_END
}
int vdb_hnsw_attach(vdb_index *idx, uint64_t M, uint64_t ef_construction, const uint32_t *level0,
                    const uint64_t *len0, const uint64_t *vec_level, const uint32_t *upper,
                    const uint64_t *upper_len, int has_enter, uint64_t enter_point, uint64_t enter_level) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    hnsw_attach(idx->ix, M, ef_construction, level0, len0, vec_level, upper, upper_len, has_enter, enter_point,
                enter_level);
    VDB_API_END
}
int vdb_hnsw_clear(vdb_index *idx) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    hnsw_clear(idx->ix);
    VDB_API_END
}
int vdb_hnsw_has(const vdb_index *idx, int *out) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx && out, "null argument");
    *out = idx->ix.hnsw.present ? 1 : 0;
    VDB_API_END
}
int vdb_hnsw_info(const vdb_index *idx, uint64_t *m, uint64_t *max_m0, uint64_t *upper_total, int *has_enter,
                  uint64_t *enter_point, uint64_t *enter_level, uint64_t *default_ef) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    const HNSWState &h = idx->ix.hnsw;
    VDB_REQUIRE(h.present, "no HNSW graph");
    if (m) *m = h.m;
    if (max_m0) *max_m0 = h.max_m0;
    if (upper_total) *upper_total = h.upper_len.size();
    if (has_enter) *has_enter = h.has_enter ? 1 : 0;
    if (enter_point) *enter_point = h.enter_point;
    if (enter_level) *enter_level = h.enter_level;
    if (default_ef) *default_ef = h.default_ef;
    VDB_API_END
}
int vdb_hnsw_export(const vdb_index *idx, uint32_t *level0, uint64_t *len0, uint64_t *vec_level, uint32_t *upper,
                    uint64_t *upper_len) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    const HNSWState &h = idx->ix.hnsw;
    VDB_REQUIRE(h.present, "no HNSW graph");
    if (level0) std::memcpy(level0, h.level0.data(), h.level0.size() * sizeof(uint32_t));
    if (len0) std::memcpy(len0, h.len0.data(), h.len0.size() * sizeof(uint64_t));
    if (vec_level) std::memcpy(vec_level, h.vec_level.data(), h.vec_level.size() * sizeof(uint64_t));
    if (upper) std::memcpy(upper, h.upper.data(), h.upper.size() * sizeof(uint32_t));
    if (upper_len) std::memcpy(upper_len, h.upper_len.data(), h.upper_len.size() * sizeof(uint64_t));
    VDB_API_END
}

static void hnsw_dev(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, uint64_t k, uint64_t ef,
                     uint64_t *d_idx, float *d_dist, uint64_t *d_cnt) {
    hnsw_knn_device(ix, ws, d_q, nq, k, ef, false, d_idx, d_dist, d_cnt);
}
static void hnsw_pq_dev(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, uint64_t k, uint64_t ef,
                        uint64_t *d_idx, float *d_dist, uint64_t *d_cnt) {
    hnsw_knn_device(ix, ws, d_q, nq, k, ef, true, d_idx, d_dist, d_cnt);
}
int vdb_hnsw_knn(vdb_index *idx, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef,
                 uint64_t *out_idx, float *out_dist, uint64_t *out_count) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    check_query_args(idx->ix, queries, nq, dim, out_idx, out_dist);
    VDB_REQUIRE(idx->ix.hnsw.present, "knn_with_ef needs an HNSW graph (vdb_hnsw_build / vdb_hnsw_attach)");
    if (ef == 0) ef = idx->ix.hnsw.default_ef;
    host_search(idx->ix, queries, nq, k, ef, out_idx, out_dist, out_count, hnsw_dev);
    VDB_API_END
}
int vdb_hnsw_knn_pq(vdb_index *idx, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef,
                    uint64_t *out_idx, float *out_dist, uint64_t *out_count) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    check_query_args(idx->ix, queries, nq, dim, out_idx, out_dist);
    VDB_REQUIRE(idx->ix.hnsw.present, "knn_pq needs an HNSW graph");
    VDB_REQUIRE(idx->ix.pq.present, "knn_pq needs a PQ table");
    host_search(idx->ix, queries, nq, k, ef, out_idx, out_dist, out_count, hnsw_pq_dev);
    VDB_API_END
}
int vdb_hnsw_knn_device(vdb_index *idx, const void *d_queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef,
                        int use_pq, void *d_out_idx, void *d_out_dist, void *d_out_count, void *stream) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    VDB_REQUIRE(idx->ix.hnsw.present, "knn_with_ef needs an HNSW graph (vdb_hnsw_build / vdb_hnsw_attach)");
    VDB_REQUIRE(!use_pq || idx->ix.pq.present, "knn_pq needs a PQ table");
    if (ef == 0) ef = idx->ix.hnsw.default_ef;
    device_search(idx->ix, d_queries, nq, dim, k, ef, d_out_idx, d_out_dist, d_out_count, stream,
                  use_pq ? hnsw_pq_dev : hnsw_dev);
    VDB_API_END
}
int vdb_hnsw_last_stats(const vdb_index *idx, uint64_t *n_dist, uint64_t *n_expanded) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    if (n_dist) *n_dist = idx->ix.hnsw.last_n_dist.load();
    if (n_expanded) *n_expanded = idx->ix.hnsw.last_n_expanded.load();
    VDB_API_END
}

// ---- shard merge ------------------------------------------------------------------------------------
int vdb_merge_topk(const float *dists, const uint64_t *ids, const uint64_t *counts, uint64_t n_shards, uint64_t nq,
                   uint64_t k, uint64_t *out_idx, float *out_dist, uint64_t *out_count) {
    VDB_API_BEGIN
    VDB_REQUIRE(dists && ids && counts && out_idx && out_dist, "null argument");
    struct P {
        uint32_t o;
        uint64_t i;
    };
    std::vector<P> all;
    for (uint64_t q = 0; q < nq; q++) {
        all.clear();
        for (uint64_t s = 0; s < n_shards; s++) {
            uint64_t c = std::min<uint64_t>(counts[s * nq + q], k);
            for (uint64_t j = 0; j < c; j++)
                all.push_back({f32_orderable(dists[(s * nq + q) * k + j]), ids[(s * nq + q) * k + j]});
        }
        std::sort(all.begin(), all.end(), [](const P &a, const P &b) { return a.o != b.o ? a.o < b.o : a.i < b.i; });
        uint64_t c = std::min<uint64_t>(all.size(), k);
        for (uint64_t j = 0; j < c; j++) {
            out_idx[q * k + j] = all[j].i;
            out_dist[q * k + j] = f32_from_orderable(all[j].o);
        }
        for (uint64_t j = c; j < k; j++) {
            out_idx[q * k + j] = 0;
            out_dist[q * k + j] = 0.0f;
        }
        if (out_count) out_count[q] = c;
    }
    VDB_API_END
}

int vdb_merge_topk_device(vdb_index *idx, const void *d_dists, const void *d_ids, const void *d_counts,
                          uint64_t n_shards, uint64_t nq, uint64_t k, void *d_out_idx, void *d_out_dist,
                          void *d_out_count, void *stream) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    merge_topk_dev(idx->ix, d_dists, d_ids, d_counts, nq * k * sizeof(float), nq * k * sizeof(uint64_t),
                   nq * sizeof(uint64_t), n_shards, nq, k, d_out_idx, d_out_dist, d_out_count, stream);
    VDB_API_END
}
// same, reading the S per-rank blocks of ONE all-gather buffer in place: block s starts s*block_bytes after block 0 and
// holds ids at off_ids, distances at off_dists, counts at off_counts (bytes; 8-byte aligned ids / counts)
int vdb_merge_topk_gathered(vdb_index *idx, const void *d_gathered, uint64_t block_bytes, uint64_t off_ids,
                            uint64_t off_dists, uint64_t off_counts, uint64_t n_shards, uint64_t nq, uint64_t k,
                            void *d_out_idx, void *d_out_dist, void *d_out_count, void *stream) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx && d_gathered, "null argument");
    VDB_REQUIRE((off_ids & 7) == 0 && (off_counts & 7) == 0 && (off_dists & 3) == 0 && (block_bytes & 7) == 0,
                "misaligned block layout");
    const char *g = static_cast<const char *>(d_gathered);
    merge_topk_dev(idx->ix, g + off_dists, g + off_ids, g + off_counts, block_bytes, block_bytes, block_bytes, n_shards,
                   nq, k, d_out_idx, d_out_dist, d_out_count, stream);
    VDB_API_END
}

int vdb_merge_topk_gathered_async(vdb_index *idx, const void *d_gathered, uint64_t block_bytes, uint64_t off_ids,
                                  uint64_t off_dists, uint64_t off_counts, uint64_t n_shards, uint64_t nq, uint64_t k,
                                  void *d_out_idx, void *d_out_dist, void *d_out_count, void *stream) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx && d_gathered && d_out_idx && d_out_dist && d_out_count, "null argument");
    VDB_REQUIRE((off_ids & 7) == 0 && (off_counts & 7) == 0 && (off_dists & 3) == 0 && (block_bytes & 7) == 0,
                "misaligned block layout");
    VDB_REQUIRE(k >= 1 && k <= 64, "the enqueued merge serves k in 1..64 (one launch, no scratch); use vdb_merge_topk_gathered beyond");
    VDB_REQUIRE(nq <= 65535 && n_shards <= 65535, "too many queries or shards for one call");
    idx->ix.use_device();
    const char *g = static_cast<const char *>(d_gathered);
    launch_merge_shards64(reinterpret_cast<const float *>(g + off_dists), reinterpret_cast<const uint64_t *>(g + off_ids),
                          reinterpret_cast<const uint64_t *>(g + off_counts), block_bytes, block_bytes, block_bytes, (uint32_t)n_shards,
                          (uint32_t)nq, (uint32_t)k, static_cast<uint64_t *>(d_out_idx), static_cast<float *>(d_out_dist),
                          static_cast<uint64_t *>(d_out_count), static_cast<hipStream_t>(stream));
    VDB_HIP(hipGetLastError());
    VDB_API_END
}

// ---- measurement hooks --------------------------------------------------------------------------------
int vdb_stream_probe(int device_id, uint64_t bytes, int iters, double *out_gbps) {
    VDB_API_BEGIN
    VDB_REQUIRE(out_gbps, "null out");
    require_gpu();
    *out_gbps = stream_probe(device_id, bytes, iters);
    VDB_API_END
}
int vdb_stream_probe_rows(int device_id, uint64_t bytes, int iters, uint32_t row_bytes, double *out_gbps) {
    VDB_API_BEGIN
    VDB_REQUIRE(out_gbps, "null out");
    require_gpu();
    *out_gbps = stream_probe_pattern(device_id, bytes, iters, 1, row_bytes);
    VDB_API_END
}
int vdb_mfma_probe(int device_id, int waves_per_simd, int iters, double *out_tflops, double *out_clock_ghz) {
    VDB_API_BEGIN
    VDB_REQUIRE(out_tflops && out_clock_ghz, "null out");
    require_gpu();
    mfma_probe(device_id, waves_per_simd, iters, out_tflops, out_clock_ghz);
    VDB_API_END
}
int vdb_mfma_probe_i8(int device_id, int waves_per_simd, int iters, double *out_tops, double *out_clock_ghz) {
    VDB_API_BEGIN
    VDB_REQUIRE(out_tops && out_clock_ghz, "null out");
    require_gpu();
    mfma_probe(device_id, waves_per_simd, iters, out_tops, out_clock_ghz, 1);
    VDB_API_END
}
int vdb_latency_probe(int device_id, uint64_t bytes, uint32_t hops, double *out_ns_per_load) {
    VDB_API_BEGIN
    VDB_REQUIRE(out_ns_per_load, "null out");
    require_gpu();
    *out_ns_per_load = latency_probe(device_id, bytes, hops);
    VDB_API_END
}
int vdb_fold_probe(int device_id, uint32_t adds, double *out_ns_per_add) {
    VDB_API_BEGIN
    VDB_REQUIRE(out_ns_per_add, "null out");
    require_gpu();
    *out_ns_per_add = fold_probe(device_id, adds);
    VDB_API_END
}
int vdb_prof_enable(vdb_index *idx, int on) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    idx->ix.prof_on = on != 0;
    VDB_API_END
}
int vdb_prof_reset(vdb_index *idx) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx, "null index");
    std::lock_guard<std::mutex> g(idx->ix.prof_mu);
    idx->ix.prof.clear();
    VDB_API_END
}
int vdb_prof_get(vdb_index *idx, const char *kernel, double *total_ms, uint64_t *launches, double *bytes) {
    VDB_API_BEGIN
    VDB_REQUIRE(idx && kernel, "null argument");
    std::lock_guard<std::mutex> g(idx->ix.prof_mu);
    auto it = idx->ix.prof.find(kernel);
    ProfEntry e = it == idx->ix.prof.end() ? ProfEntry{} : it->second;
    if (total_ms) *total_ms = e.ms;
    if (launches) *launches = e.launches;
    if (bytes) *bytes = e.bytes;
    VDB_API_END
}

}  // extern "C"
