// ctx.hip -- multi-GPU behind the C ABI: a context owns the RCCL communicator, a sharded index owns
// one row block per GPU, and a search = local top-k on every shard + ONE all-gather over xGMI + the exact merge by
// (distance, index).  No torch, no Python: this is what the `DynamicIndex::Gpu` arm of INTEGRATION.md calls when the
// host has more than one GPU.
//
// Two ways to form the communicator, both ending in the same search code:
//   vdb_ctx_create(device_ids, n_dev)           one host process drives n_dev GPUs (ncclCommInitAll); a Rust host is
//                                               one process, so this is the default for the reference's layout;
//   vdb_ctx_create_rank(device, id, rank, world) one process per GPU (ncclCommInitRank with an id from
//                                               vdb_ctx_unique_id that the host distributes), the layout of bench.py.
// RCCL is bound at run time (dlopen of librccl.so.1, the library of this ROCm image / of the PyTorch wheel already in
// the process): single-GPU users of libvdbhip.so never load it.
//
// Row partition: rank r of S holds rows [r * ceil(N/S), min(N, (r+1) * ceil(N/S))) and reports global ids
// (id_offset).  Exchange block of a rank: [nq*k ids u64 | nq*k distances f32 | pad to 8 | nq counts u64]; the merge
// reads the S received blocks in place (launch_merge_shards64 / pack + merge).  PQ-Flat: the ADC key rows and the
// exact-distance key rows of every shard travel in two all-gathers of one group, then k_pq_shard_merge + pq_resort.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <functional>
#include <thread>

#include "ctx.hpp"
#include "pq_hnsw.hpp"

using namespace vdb;

namespace {

struct Rccl {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    static Rccl &get() {
        static Rccl r;
        static std::once_flag once;
        std::call_once(once, [] {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (r.h) break;
            }
            if (!r.h) return;
#define VDB_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.h, sym))
            VDB_SYM(GetUniqueId, "ncclGetUniqueId");
            VDB_SYM(CommInitRank, "ncclCommInitRank");
            VDB_SYM(CommInitAll, "ncclCommInitAll");
            VDB_SYM(CommDestroy, "ncclCommDestroy");
            VDB_SYM(AllGather, "ncclAllGather");
            VDB_SYM(GroupStart, "ncclGroupStart");
            VDB_SYM(GroupEnd, "ncclGroupEnd");
            VDB_SYM(GetErrorString, "ncclGetErrorString");
#undef VDB_SYM
        });
        if (!r.h || !r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GroupStart ||
            !r.GroupEnd || !r.GetErrorString)
            throw Error(VDB_ERR_STATE, "RCCL is not available (dlopen librccl.so.1 failed): multi-GPU contexts need it");
        return r;
    }
};

#define VDB_NCCL(expr)                                                                                         \
    do {                                                                                                       \
        ncclResult_t _r = (expr);                                                                              \
        if (_r != ncclSuccess)                                                                                 \
            throw ::vdb::Error(2, std::string(#expr) + ": " + Rccl::get().GetErrorString(_r) + " (" + __FILE__ + \
                                      ":" + std::to_string(__LINE__) + ")");                                   \
    } while (0)

}  // namespace

struct vdb_ctx {
    int world = 1;               // ranks of the communicator
    std::vector<int> devices;    // this process's GPUs ...
    std::vector<int> ranks;      // ... and their ranks
    std::vector<ncclComm_t> comms;  // one per local GPU; empty when no communicator exists (world == 1 without force)
    ~vdb_ctx() {
        for (size_t i = 0; i < comms.size(); i++) {
            (void)hipSetDevice(devices[i]);
            if (comms[i]) (void)Rccl::get().CommDestroy(comms[i]);
        }
    }
};

namespace {

struct Shard {
    std::unique_ptr<vdb_index> handle;
    hipStream_t stream = nullptr;  // collectives + merge of this GPU
    DevBuf q, send, recv, adc, exact, g_adc, g_exact, o_idx, o_dist, o_cnt;
    uint64_t r0 = 0, r1 = 0;       // global row range
};

// run fn(i) for every local shard: on the calling thread when there is one, else one host thread per GPU (the
// search entry points synchronise their own stream, so the GPUs only overlap when each has its own thread)
template <class F>
void for_each_shard(size_t n, F fn) {
    if (n == 1) {
        fn(0);
        return;
    }
    std::vector<std::thread> th;
    std::vector<std::string> err(n);
    std::vector<int> code(n, 0);
    for (size_t i = 0; i < n; i++)
        th.emplace_back([&, i] {
            try {
                fn(i);
            } catch (const Error &e) {
                err[i] = e.what();
                code[i] = e.code;
            } catch (const std::exception &e) {
                err[i] = e.what();
                code[i] = VDB_ERR_INVALID;
            }
        });
    for (auto &t : th) t.join();
    for (size_t i = 0; i < n; i++)
        if (code[i]) throw Error(code[i], "shard " + std::to_string(i) + ": " + err[i]);
}

}  // namespace

enum { LAYOUT_UNSET = 0, LAYOUT_ROWS = 1, LAYOUT_REPLICA = 2 };

struct vdb_sharded {
    vdb_ctx *ctx = nullptr;
    uint64_t dim = 0;
    int dist = 0;
    uint64_t n_total = 0;
    int layout = LAYOUT_UNSET;  // ROWS: contiguous row blocks (Flat, PQ-Flat); REPLICA: every GPU holds all rows, queries are split
    // a collective call that failed on one rank of a multi-PROCESS job leaves its peers inside the all-gather: the object
    // cannot know whether they got out, so every later search is refused (vdbhip.h, "failed sharded calls")
    bool poisoned = false;
    std::vector<std::unique_ptr<Shard>> shards;
    ~vdb_sharded() {
        for (size_t i = 0; i < shards.size(); i++) {
            (void)hipSetDevice(ctx->devices[i]);
            shards[i]->handle.reset();
            shards[i]->q.release();
            shards[i]->send.release();
            shards[i]->recv.release();
            shards[i]->adc.release();
            shards[i]->exact.release();
            shards[i]->g_adc.release();
            shards[i]->g_exact.release();
            shards[i]->o_idx.release();
            shards[i]->o_dist.release();
            shards[i]->o_cnt.release();
            if (shards[i]->stream) (void)hipStreamDestroy(shards[i]->stream);
        }
    }
};

namespace {

bool multi_process(const vdb_sharded &sh) { return (size_t)sh.ctx->world > sh.shards.size(); }
// body of a sharded search: refuses a poisoned object; an error inside poisons it when other processes take part
#define VDB_SHARDED_SEARCH_BEGIN(sh)                                                                                           \
    VDB_REQUIRE(!(sh)->poisoned, "this sharded index is poisoned: an earlier collective call failed on this rank while other " \
                                 "processes were inside the exchange; destroy it on every rank and build a new one");          \
    try {
#define VDB_SHARDED_SEARCH_END(sh)             \
    }                                          \
    catch (...) {                              \
        if (multi_process(*(sh))) (sh)->poisoned = true; \
        throw;                                 \
    }

// one all-gather (or two, for the PQ key rows) over the communicator: every local GPU contributes `bytes` from send[i]
// and receives world * bytes into recv[i], ordered on the shard's stream; without a communicator (world == 1) the
// receive buffer IS the send buffer
void all_gather(vdb_sharded &sh, const std::vector<std::pair<const void *, void *>> &bufs_per_shard, size_t n_ops, size_t bytes) {
    vdb_ctx &c = *sh.ctx;
    if (c.comms.empty()) return;
    Rccl &r = Rccl::get();
    VDB_NCCL(r.GroupStart());
    for (size_t i = 0; i < sh.shards.size(); i++) {
        VDB_HIP(hipSetDevice(c.devices[i]));
        for (size_t o = 0; o < n_ops; o++) {
            const auto &b = bufs_per_shard[i * n_ops + o];
            VDB_NCCL(r.AllGather(b.first, b.second, bytes, ncclUint8, c.comms[i], sh.shards[i]->stream));
        }
    }
    VDB_NCCL(r.GroupEnd());
}

// ---- REPLICA layout: every GPU holds all rows, rank r answers the contiguous query block [r * per, min(nq, (r+1) * per)) with
// per = ceil(nq / S) (shard.replica_query_slice), ONE fixed-size all-gather of [per*k ids | per*k distances | pad | per counts]
// concatenates the blocks on every rank.  No merge: the blocks are disjoint rows of the answer.  HNSW searches exist only in
// this layout (a graph's edges cross any row partition); Flat and PQ-Flat accept it too -- when the corpus fits one
// GPU a replica reads the same bytes per step as a row shard and runs the per-query stages on 1/S of the queries.
using LocalSearch = std::function<void(Index &, Workspace &, const float *d_q, uint64_t nq, uint64_t *d_idx, float *d_dist, uint64_t *d_cnt)>;

// query block of rank r
inline void replica_block(uint64_t nq, uint64_t S, uint64_t r, uint64_t &q0, uint64_t &q1) {
    const uint64_t per = (nq + S - 1) / S;
    q0 = std::min(nq, r * per);
    q1 = std::min(nq, (r + 1) * per);
}

void replica_search(vdb_sharded &sh, const float *queries, uint64_t nq, uint64_t k, uint64_t *out_idx, float *out_dist,
                    uint64_t *out_count, const LocalSearch &local) {
    const uint64_t S = (uint64_t)sh.ctx->world, dim = sh.dim;
    if (k == 0) {
        if (out_count) std::memset(out_count, 0, nq * 8);
        return;
    }
    constexpr uint64_t CHUNK = 65536;  // queries per exchange (bounds the staging buffers)
    std::vector<char> host;
    for (uint64_t c0 = 0; c0 < nq; c0 += CHUNK) {
        const uint64_t nb = std::min(CHUNK, nq - c0), per = (nb + S - 1) / S;
        const uint64_t off_d = per * k * 8, off_c = (per * k * 12 + 7) / 8 * 8, block = off_c + per * 8;
        for_each_shard(sh.shards.size(), [&](size_t i) {
            Shard &s = *sh.shards[i];
            Index &ix = s.handle->ix;
            ix.use_device();
            uint64_t q0, q1;
            replica_block(nb, S, (uint64_t)sh.ctx->ranks[i], q0, q1);
            s.q.reserve(std::max<uint64_t>(per, 1) * dim * sizeof(float));
            s.send.reserve(block);
            if (!sh.ctx->comms.empty()) s.recv.reserve(S * block);
            VDB_HIP(hipMemsetAsync(s.send.p, 0, block, s.stream));
            if (q1 > q0)
                VDB_HIP(hipMemcpyAsync(s.q.p, queries + (c0 + q0) * dim, (q1 - q0) * dim * sizeof(float), hipMemcpyHostToDevice, s.stream));
            VDB_SYNC(s.stream);
            if (q1 > q0) {
                WsLease ws(ix);
                char *b = s.send.as<char>();
                local(ix, *ws, s.q.as<float>(), q1 - q0, reinterpret_cast<uint64_t *>(b), reinterpret_cast<float *>(b + off_d),
                      reinterpret_cast<uint64_t *>(b + off_c));
                VDB_SYNC(ws->stream);
                ix.prof_collect(*ws);
            }
        });
        std::vector<std::pair<const void *, void *>> bufs;
        for (auto &s : sh.shards) bufs.push_back({s->send.p, s->recv.p});
        all_gather(sh, bufs, 1, block);
        Shard &s0 = *sh.shards[0];
        s0.handle->ix.use_device();
        VDB_SYNC(s0.stream);  // the collective
        const bool comm = !sh.ctx->comms.empty();
        const uint64_t ns = comm ? S : 1;
        host.resize(ns * block);
        VDB_HIP(hipMemcpy(host.data(), comm ? s0.recv.p : s0.send.p, ns * block, hipMemcpyDeviceToHost));
        for (uint64_t r = 0; r < ns; r++) {
            uint64_t q0, q1;
            replica_block(nb, S, comm ? r : (uint64_t)sh.ctx->ranks[0], q0, q1);
            if (q1 == q0) continue;
            const char *b = host.data() + r * block;
            std::memcpy(out_idx + (c0 + q0) * k, b, (q1 - q0) * k * 8);
            std::memcpy(out_dist + (c0 + q0) * k, b + off_d, (q1 - q0) * k * 4);
            if (out_count) std::memcpy(out_count + c0 + q0, b + off_c, (q1 - q0) * 8);
        }
    }
}

void ctx_validate_devices(const int *device_ids, int n_dev) {
    VDB_REQUIRE(device_ids && n_dev >= 1 && n_dev <= 64, "device list must hold 1..64 device ids");
    require_gpu();
    int cnt = 0;
    VDB_HIP(hipGetDeviceCount(&cnt));
    for (int i = 0; i < n_dev; i++) {
        VDB_REQUIRE(device_ids[i] >= 0 && device_ids[i] < cnt, "device id out of range");
        for (int j = 0; j < i; j++) VDB_REQUIRE(device_ids[j] != device_ids[i], "a device may appear once in a context");
    }
}

bool force_rccl() {
    const char *e = std::getenv("VDB_CTX_FORCE_RCCL");  // build the communicator even for one rank (exercises the RCCL path on a 1-GPU box)
    return e && e[0] == '1';
}

}  // namespace

extern "C" {

int vdb_ctx_create(const int *device_ids, int n_dev, vdb_ctx **out) {
    VDB_API_BEGIN
    VDB_REQUIRE(out, "null out");
    ctx_validate_devices(device_ids, n_dev);
    std::unique_ptr<vdb_ctx> c(new vdb_ctx);
    c->world = n_dev;
    c->devices.assign(device_ids, device_ids + n_dev);
    for (int i = 0; i < n_dev; i++) c->ranks.push_back(i);
    if (n_dev > 1 || force_rccl()) {
        c->comms.assign(n_dev, nullptr);
        VDB_NCCL(Rccl::get().CommInitAll(c->comms.data(), n_dev, c->devices.data()));
    }
    *out = c.release();
    VDB_API_END
}

int vdb_ctx_unique_id(void *out_id, uint64_t out_bytes) {
    VDB_API_BEGIN
    VDB_REQUIRE(out_id && out_bytes >= sizeof(ncclUniqueId), "the id buffer must hold 128 bytes");
    ncclUniqueId id;
    VDB_NCCL(Rccl::get().GetUniqueId(&id));
    std::memcpy(out_id, &id, sizeof(id));
    VDB_API_END
}

int vdb_ctx_create_rank(int device_id, const void *id, int rank, int world, vdb_ctx **out) {
    VDB_API_BEGIN
    VDB_REQUIRE(out, "null out");
    VDB_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank must be in 0..world");
    ctx_validate_devices(&device_id, 1);
    std::unique_ptr<vdb_ctx> c(new vdb_ctx);
    c->world = world;
    c->devices.push_back(device_id);
    c->ranks.push_back(rank);
    if (world > 1 || force_rccl()) {
        VDB_REQUIRE(id, "null communicator id (vdb_ctx_unique_id on rank 0, distributed by the host)");
        ncclUniqueId uid;
        std::memcpy(&uid, id, sizeof(uid));
        VDB_HIP(hipSetDevice(device_id));
        c->comms.assign(1, nullptr);
        VDB_NCCL(Rccl::get().CommInitRank(&c->comms[0], world, uid, rank));
    }
    *out = c.release();
    VDB_API_END
}

int vdb_ctx_destroy(vdb_ctx *ctx) {
    VDB_API_BEGIN
    delete ctx;
    VDB_API_END
}

int vdb_ctx_info(const vdb_ctx *ctx, int *world, int *n_local, int *first_rank, int *has_comm) {
    VDB_API_BEGIN
    VDB_REQUIRE(ctx, "null context");
    if (world) *world = ctx->world;
    if (n_local) *n_local = (int)ctx->devices.size();
    if (first_rank) *first_rank = ctx->ranks[0];
    if (has_comm) *has_comm = ctx->comms.empty() ? 0 : 1;
    VDB_API_END
}

int vdb_sharded_create(vdb_ctx *ctx, uint64_t dim, int dist, vdb_sharded **out) {
    VDB_API_BEGIN
    VDB_REQUIRE(ctx && out, "null argument");
    VDB_REQUIRE(dim > 0 && dim < (1u << 24), "dim must be in 1..2^24");
    VDB_REQUIRE(dist == VDB_L2SQR || dist == VDB_COSINE, "dist must be 0 (L2Sqr) or 1 (Cosine)");
    std::unique_ptr<vdb_sharded> sh(new vdb_sharded);
    sh->ctx = ctx;
    sh->dim = dim;
    sh->dist = dist;
    for (size_t i = 0; i < ctx->devices.size(); i++) sh->shards.emplace_back(new Shard);
    for (size_t i = 0; i < sh->shards.size(); i++) {
        VDB_HIP(hipSetDevice(ctx->devices[i]));
        sh->shards[i]->handle.reset(new vdb_index(ctx->devices[i], dim, dist));
        VDB_HIP(hipStreamCreateWithFlags(&sh->shards[i]->stream, hipStreamNonBlocking));
    }
    *out = sh.release();
    VDB_API_END
}

int vdb_sharded_destroy(vdb_sharded *sh) {
    VDB_API_BEGIN
    delete sh;
    VDB_API_END
}

// every process passes the SAME corpus view.  ROWS layout: each local GPU keeps its contiguous block;
// REPLICA layout: each keeps all rows (HNSW does not shard: "replicas only", queries are split instead).
// A failure on any shard (an allocation on one GPU, say) rolls EVERY shard back to an empty index, so that a retry cannot
// append a block twice on the GPUs that had succeeded.
static void sharded_set_rows(vdb_sharded *sh, const float *rows, uint64_t n_total, int layout) {
    VDB_REQUIRE(sh && (rows || n_total == 0), "null argument");
    VDB_REQUIRE(sh->layout == LAYOUT_UNSET, "the corpus of a sharded index is set once (a different partition would move rows between GPUs)");
    const uint64_t S = (uint64_t)sh->ctx->world, per = (n_total + S - 1) / S;
    try {
        for_each_shard(sh->shards.size(), [&](size_t i) {
            Shard &s = *sh->shards[i];
            const uint64_t r = (uint64_t)sh->ctx->ranks[i];
            s.r0 = layout == LAYOUT_ROWS ? std::min(n_total, r * per) : 0;
            s.r1 = layout == LAYOUT_ROWS ? std::min(n_total, (r + 1) * per) : n_total;
            Index &ix = s.handle->ix;
            ix.use_device();
            if (s.r1 > s.r0) ix.add_rows(rows + s.r0 * sh->dim, s.r1 - s.r0, false);
            ix.id_offset = s.r0;
        });
    } catch (...) {
        // Other processes may have kept their blocks: this rank would refuse the next search before its collective while they enter
        // it and wait for ever.  The object is poisoned on this rank (every later call fails loudly; vdb_sharded_poisoned tells the
        // launcher to tear the job down) -- a multi-process job cannot be rolled back from one side.
        if (multi_process(*sh)) sh->poisoned = true;
        for (size_t i = 0; i < sh->shards.size(); i++) {
            (void)hipSetDevice(sh->ctx->devices[i]);
            try {
                sh->shards[i]->handle.reset(new vdb_index(sh->ctx->devices[i], sh->dim, sh->dist));
            } catch (...) {
                sh->poisoned = true;  // not even an empty index could be created on that GPU
            }
            sh->shards[i]->r0 = sh->shards[i]->r1 = 0;
        }
        throw;
    }
    sh->n_total = n_total;
    sh->layout = layout;
}
int vdb_sharded_set_rows(vdb_sharded *sh, const float *rows, uint64_t n_total) {
    VDB_API_BEGIN
    sharded_set_rows(sh, rows, n_total, LAYOUT_ROWS);
    VDB_API_END
}
int vdb_sharded_set_rows_replica(vdb_sharded *sh, const float *rows, uint64_t n_total) {
    VDB_API_BEGIN
    sharded_set_rows(sh, rows, n_total, LAYOUT_REPLICA);
    VDB_API_END
}
int vdb_sharded_layout(const vdb_sharded *sh, int *out) {
    VDB_API_BEGIN
    VDB_REQUIRE(sh && out, "null argument");
    *out = sh->layout;
    VDB_API_END
}

int vdb_sharded_len(const vdb_sharded *sh, uint64_t *out) {
    VDB_API_BEGIN
    VDB_REQUIRE(sh && out, "null argument");
    *out = sh->n_total;
    VDB_API_END
}

// borrowed handle of local shard i (tuning switches, statistics, PQ export); owned by the sharded index
int vdb_sharded_local(vdb_sharded *sh, int i, vdb_index **out) {
    VDB_API_BEGIN
    VDB_REQUIRE(sh && out && i >= 0 && (size_t)i < sh->shards.size(), "bad local shard number");
    *out = sh->shards[i]->handle.get();
    VDB_API_END
}

// FlatIndex::knn over the whole corpus: local top-k on every shard, one all-gather, exact merge (identical to the
// unsharded result: top-k under a total order is decomposable)
int vdb_sharded_flat_knn(vdb_sharded *sh, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t *out_idx,
                         float *out_dist, uint64_t *out_count) {
    VDB_API_BEGIN
    VDB_REQUIRE(sh, "null index");
    VDB_REQUIRE(dim == sh->dim, "query dimension mismatch");
    VDB_REQUIRE(nq == 0 || (queries && out_idx && out_dist), "null argument");
    VDB_REQUIRE(sh->layout != LAYOUT_UNSET || nq == 0, "the sharded index holds no rows (vdb_sharded_set_rows / _set_rows_replica)");
    VDB_SHARDED_SEARCH_BEGIN(sh)
    if (sh->layout == LAYOUT_REPLICA) {
        replica_search(*sh, queries, nq, k, out_idx, out_dist, out_count,
                       [&](Index &ix, Workspace &ws, const float *d_q, uint64_t n, uint64_t *di, float *dd, uint64_t *dc) {
                           ix.flat_knn_device(ws, d_q, n, k, di, dd, dc);
                       });
        return VDB_OK;
    }
    VDB_REQUIRE(k <= 1024, "sharded knn: k must be <= 1024 (the merge of the gathered lists)");
    const uint64_t S = (uint64_t)sh->ctx->world;
    constexpr uint64_t CHUNK = 8192;
    for (uint64_t q0 = 0; q0 < nq; q0 += CHUNK) {
        const uint64_t nb = std::min(CHUNK, nq - q0), kk = std::max<uint64_t>(k, 1);
        const uint64_t off_d = nb * kk * 8, off_c = (nb * kk * 12 + 7) / 8 * 8, block = off_c + nb * 8;
        for_each_shard(sh->shards.size(), [&](size_t i) {
            Shard &s = *sh->shards[i];
            Index &ix = s.handle->ix;
            ix.use_device();
            s.q.reserve(nb * dim * sizeof(float));
            s.send.reserve(block);
            if (!sh->ctx->comms.empty()) s.recv.reserve(S * block);
            VDB_HIP(hipMemcpyAsync(s.q.p, queries + q0 * dim, nb * dim * sizeof(float), hipMemcpyHostToDevice, s.stream));
            VDB_HIP(hipMemsetAsync(s.send.p, 0, block, s.stream));
            VDB_SYNC(s.stream);
            WsLease ws(ix);
            char *b = s.send.as<char>();
            ix.flat_knn_device(*ws, s.q.as<float>(), nb, k, reinterpret_cast<uint64_t *>(b), reinterpret_cast<float *>(b + off_d),
                               reinterpret_cast<uint64_t *>(b + off_c));
            VDB_SYNC(ws->stream);
            ix.prof_collect(*ws);
        });
        std::vector<std::pair<const void *, void *>> bufs;
        for (auto &s : sh->shards) bufs.push_back({s->send.p, s->recv.p});
        all_gather(*sh, bufs, 1, block);
        // merge on the first local GPU (every rank of a multi-process job ends with the full answer)
        Shard &s0 = *sh->shards[0];
        Index &ix0 = s0.handle->ix;
        ix0.use_device();
        const char *g = sh->ctx->comms.empty() ? s0.send.as<char>() : s0.recv.as<char>();
        const uint64_t ns = sh->ctx->comms.empty() ? 1 : S;
        if (k > 0) {
            s0.o_idx.reserve(nb * kk * 8);
            s0.o_dist.reserve(nb * kk * 4);
            s0.o_cnt.reserve(nb * 8);
            merge_topk_dev(ix0, g + off_d, g, g + off_c, block, block, block, ns, nb, k, s0.o_idx.p, s0.o_dist.p, s0.o_cnt.p, s0.stream);
            VDB_HIP(hipMemcpy(out_idx + q0 * k, s0.o_idx.p, nb * k * 8, hipMemcpyDeviceToHost));
            VDB_HIP(hipMemcpy(out_dist + q0 * k, s0.o_dist.p, nb * k * 4, hipMemcpyDeviceToHost));
            if (out_count) VDB_HIP(hipMemcpy(out_count + q0, s0.o_cnt.p, nb * 8, hipMemcpyDeviceToHost));
        } else if (out_count) {
            std::memset(out_count + q0, 0, nb * 8);
        }
    }
    VDB_SHARDED_SEARCH_END(sh)
    VDB_API_END
}

// PQTable for every shard from the same centroids (replicated); the codes of a shard's rows are encoded on
// its own GPU
int vdb_sharded_pq_attach(vdb_sharded *sh, uint64_t n_bits, uint64_t m, const float *centroids) {
    VDB_API_BEGIN
    VDB_REQUIRE(sh && centroids, "null argument");
    for_each_shard(sh->shards.size(), [&](size_t i) {
        sh->shards[i]->handle->ix.use_device();
        pq_attach(sh->shards[i]->handle->ix, n_bits, m, centroids, nullptr);
    });
    VDB_API_END
}

// FlatIndex::knn_pq over the whole corpus: per-shard ADC top-max(ef,k) as pair-key rows (ADC key + exact key), two
// all-gathers in one group, merge in (adc, idx) order, pq_resort replay -- equal to the unsharded knn_pq
int vdb_sharded_knn_pq(vdb_sharded *sh, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef,
                       uint64_t *out_idx, float *out_dist, uint64_t *out_count) {
    VDB_API_BEGIN
    VDB_REQUIRE(sh, "null index");
    VDB_REQUIRE(dim == sh->dim, "query dimension mismatch");
    VDB_REQUIRE(nq == 0 || (queries && out_idx && out_dist), "null argument");
    for (auto &s : sh->shards) VDB_REQUIRE(s->handle->ix.pq.present, "knn_pq needs a PQ table (vdb_sharded_pq_attach)");
    const uint64_t S = (uint64_t)sh->ctx->world, efg = std::max(ef, k);
    if (k == 0) {
        if (out_count) std::memset(out_count, 0, nq * 8);
        return VDB_OK;
    }
    VDB_SHARDED_SEARCH_BEGIN(sh)
    if (sh->layout == LAYOUT_REPLICA) {
        replica_search(*sh, queries, nq, k, out_idx, out_dist, out_count,
                       [&](Index &ix, Workspace &ws, const float *d_q, uint64_t n, uint64_t *di, float *dd, uint64_t *dc) {
                           flat_knn_pq_device(ix, ws, d_q, n, k, ef, di, dd, dc);
                       });
        return VDB_OK;
    }
    const uint64_t CHUNK = std::max<uint64_t>(1, std::min<uint64_t>(8192, (size_t(256) << 20) / (efg * 16 * S)));
    for (uint64_t q0 = 0; q0 < nq; q0 += CHUNK) {
        const uint64_t nb = std::min(CHUNK, nq - q0), bytes = nb * efg * 8;
        for_each_shard(sh->shards.size(), [&](size_t i) {
            Shard &s = *sh->shards[i];
            Index &ix = s.handle->ix;
            ix.use_device();
            s.q.reserve(nb * dim * sizeof(float));
            s.adc.reserve(bytes);
            s.exact.reserve(bytes);
            if (!sh->ctx->comms.empty()) {
                s.g_adc.reserve(S * bytes);
                s.g_exact.reserve(S * bytes);
            }
            VDB_HIP(hipMemcpyAsync(s.q.p, queries + q0 * dim, nb * dim * sizeof(float), hipMemcpyHostToDevice, s.stream));
            VDB_SYNC(s.stream);
            WsLease ws(ix);
            flat_knn_pq_shard_device(ix, *ws, s.q.as<float>(), nb, k, ef, s.adc.as<uint64_t>(), s.exact.as<uint64_t>());
            VDB_SYNC(ws->stream);
            ix.prof_collect(*ws);
        });
        std::vector<std::pair<const void *, void *>> bufs;
        for (auto &s : sh->shards) {
            bufs.push_back({s->adc.p, s->g_adc.p});
            bufs.push_back({s->exact.p, s->g_exact.p});
        }
        all_gather(*sh, bufs, 2, bytes);
        Shard &s0 = *sh->shards[0];
        Index &ix0 = s0.handle->ix;
        ix0.use_device();
        const bool comm = !sh->ctx->comms.empty();
        s0.o_idx.reserve(nb * k * 8);
        s0.o_dist.reserve(nb * k * 4);
        s0.o_cnt.reserve(nb * 8);
        VDB_SYNC(s0.stream);  // the collectives
        {
            WsLease ws(ix0);
            pq_merge_resort_device(ix0, *ws, comm ? s0.g_adc.as<uint64_t>() : s0.adc.as<uint64_t>(),
                                   comm ? s0.g_exact.as<uint64_t>() : s0.exact.as<uint64_t>(), comm ? S : 1, nb, efg, k,
                                   s0.o_idx.as<uint64_t>(), s0.o_dist.as<float>(), s0.o_cnt.as<uint64_t>());
            VDB_SYNC(ws->stream);
        }
        VDB_HIP(hipMemcpy(out_idx + q0 * k, s0.o_idx.p, nb * k * 8, hipMemcpyDeviceToHost));
        VDB_HIP(hipMemcpy(out_dist + q0 * k, s0.o_dist.p, nb * k * 4, hipMemcpyDeviceToHost));
        if (out_count) VDB_HIP(hipMemcpy(out_count + q0, s0.o_cnt.p, nb * 8, hipMemcpyDeviceToHost));
    }
    VDB_SHARDED_SEARCH_END(sh)
    VDB_API_END
}

// ---- HNSW behind the context: replicas, the third search DynamicIndex dispatches (dynamic_index.rs:68-93) --------
// The graph is built ONCE per process on the first local GPU and attached to the other local replicas.  The host builder is
// deterministic for a given seed, batch size AND thread count, so a multi-process job must name its thread count (the default, the
// hardware concurrency, differs between hosts) -- and because a query's answer depends on the graph of the rank its block landed on,
// the processes compare a hash of what they built (level-0 links and lengths, upper links, levels, entry) in one all-gather: a
// mismatch poisons the object instead of serving answers that differ from the single-process result.
static uint64_t hnsw_graph_hash(const HNSWState &h) {
    uint64_t x = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t bytes) {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i < bytes; i++) x = (x ^ b[i]) * 1099511628211ull;
    };
    mix(h.level0.data(), h.level0.size() * sizeof(h.level0[0]));
    mix(h.len0.data(), h.len0.size() * sizeof(h.len0[0]));
    mix(h.vec_level.data(), h.vec_level.size() * sizeof(h.vec_level[0]));
    mix(h.upper.data(), h.upper.size() * sizeof(h.upper[0]));
    mix(h.upper_len.data(), h.upper_len.size() * sizeof(h.upper_len[0]));
    const uint64_t tail[3] = {h.has_enter ? 1ull : 0ull, h.enter_point, h.enter_level};
    mix(tail, sizeof(tail));
    return x;
}
int vdb_sharded_hnsw_build(vdb_sharded *sh, uint64_t M, uint64_t ef_construction, uint64_t seed, uint64_t batch, int nthreads) {
    VDB_API_BEGIN
    VDB_REQUIRE(sh, "null index");
    VDB_REQUIRE(sh->layout == LAYOUT_REPLICA, "HNSW does not shard by rows (edges cross any partition): use vdb_sharded_set_rows_replica");
    VDB_REQUIRE(!multi_process(*sh) || nthreads > 0,
                "vdb_sharded_hnsw_build in a multi-process job needs an explicit nthreads (the graph depends on it; every process must build the same)");
    VDB_SHARDED_SEARCH_BEGIN(sh)
    Index &ix0 = sh->shards[0]->handle->ix;
    ix0.use_device();
    hnsw_build(ix0, M, ef_construction, seed, batch, nthreads);
    const HNSWState &h = ix0.hnsw;
    for (size_t i = 1; i < sh->shards.size(); i++) {
        Index &ix = sh->shards[i]->handle->ix;
        ix.use_device();
        hnsw_attach(ix, h.m, h.ef_construction, h.level0.data(), h.len0.data(), h.vec_level.data(), h.upper.data(), h.upper_len.data(),
                    h.has_enter ? 1 : 0, h.enter_point, h.enter_level);
    }
    if (!sh->ctx->comms.empty() && sh->ctx->world > 1) {
        const uint64_t mine = hnsw_graph_hash(h), S = (uint64_t)sh->ctx->world;
        std::vector<std::pair<const void *, void *>> bufs;
        for (size_t i = 0; i < sh->shards.size(); i++) {
            Shard &s = *sh->shards[i];
            s.handle->ix.use_device();
            s.send.reserve(8);
            s.recv.reserve(S * 8);
            VDB_HIP(hipMemcpyAsync(s.send.p, &mine, 8, hipMemcpyHostToDevice, s.stream));
            bufs.push_back({s.send.p, s.recv.p});
        }
        all_gather(*sh, bufs, 1, 8);
        Shard &s0 = *sh->shards[0];
        s0.handle->ix.use_device();
        std::vector<uint64_t> all(S);
        VDB_HIP(hipMemcpyAsync(all.data(), s0.recv.p, S * 8, hipMemcpyDeviceToHost, s0.stream));
        for (auto &sp : sh->shards) {
            sp->handle->ix.use_device();
            VDB_SYNC(sp->stream);
        }
        for (uint64_t r = 0; r < S; r++)
            VDB_REQUIRE(all[r] == mine, "vdb_sharded_hnsw_build: rank " + std::to_string(r) + " built a different graph (thread count, batch size or rows differ between the processes)");
    }
    VDB_SHARDED_SEARCH_END(sh)
    VDB_API_END
}
int vdb_sharded_hnsw_attach(vdb_sharded *sh, uint64_t M, uint64_t ef_construction, const uint32_t *level0, const uint64_t *len0,
                            const uint64_t *vec_level, const uint32_t *upper, const uint64_t *upper_len, int has_enter,
                            uint64_t enter_point, uint64_t enter_level) {
    VDB_API_BEGIN
    VDB_REQUIRE(sh, "null index");
    VDB_REQUIRE(sh->layout == LAYOUT_REPLICA, "HNSW does not shard by rows (edges cross any partition): use vdb_sharded_set_rows_replica");
    for (size_t i = 0; i < sh->shards.size(); i++) {
        Index &ix = sh->shards[i]->handle->ix;
        ix.use_device();
        hnsw_attach(ix, M, ef_construction, level0, len0, vec_level, upper, upper_len, has_enter, enter_point, enter_level);
    }
    VDB_API_END
}
static int sharded_hnsw_search(vdb_sharded *sh, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef, bool use_pq,
                               uint64_t *out_idx, float *out_dist, uint64_t *out_count) {
    VDB_API_BEGIN
    VDB_REQUIRE(sh, "null index");
    VDB_REQUIRE(dim == sh->dim, "query dimension mismatch");
    VDB_REQUIRE(nq == 0 || (queries && out_idx && out_dist), "null argument");
    VDB_REQUIRE(sh->layout == LAYOUT_REPLICA, "HNSW does not shard by rows (edges cross any partition): use vdb_sharded_set_rows_replica");
    for (auto &s : sh->shards) {
        VDB_REQUIRE(s->handle->ix.hnsw.present, "knn_with_ef needs an HNSW graph (vdb_sharded_hnsw_build / _attach)");
        VDB_REQUIRE(!use_pq || s->handle->ix.pq.present, "knn_pq needs a PQ table (vdb_sharded_pq_attach)");
    }
    if (ef == 0 && !use_pq) ef = sh->shards[0]->handle->ix.hnsw.default_ef;
    VDB_SHARDED_SEARCH_BEGIN(sh)
    replica_search(*sh, queries, nq, k, out_idx, out_dist, out_count,
                   [&](Index &ix, Workspace &ws, const float *d_q, uint64_t n, uint64_t *di, float *dd, uint64_t *dc) {
                       hnsw_knn_device(ix, ws, d_q, n, k, ef, use_pq, di, dd, dc);
                   });
    VDB_SHARDED_SEARCH_END(sh)
    VDB_API_END
}
int vdb_sharded_hnsw_knn(vdb_sharded *sh, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef, uint64_t *out_idx,
                         float *out_dist, uint64_t *out_count) {
    return sharded_hnsw_search(sh, queries, nq, dim, k, ef, false, out_idx, out_dist, out_count);
}
int vdb_sharded_hnsw_knn_pq(vdb_sharded *sh, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t ef, uint64_t *out_idx,
                            float *out_dist, uint64_t *out_count) {
    return sharded_hnsw_search(sh, queries, nq, dim, k, ef, true, out_idx, out_dist, out_count);
}
// IVFIndex over the replicas (ivf_index.rs:34-154): clusters built once per process on its first GPU (seeded k-means: equal in
// every process) and mirrored to the others; knn with the queries split
int vdb_sharded_ivf_build(vdb_sharded *sh, uint64_t k_clusters, uint64_t train_n, uint64_t max_iter, float tol, uint64_t seed) {
    VDB_API_BEGIN
    VDB_REQUIRE(sh, "null index");
    VDB_REQUIRE(sh->layout == LAYOUT_REPLICA, "IVF behind the context runs on replicas: use vdb_sharded_set_rows_replica");
    Index &ix0 = sh->shards[0]->handle->ix;
    ix0.use_device();
    ivf_build(ix0, k_clusters, train_n, max_iter, tol, seed);
    if (sh->shards.size() > 1) {
        std::vector<float> cent(ix0.ivf.k * ix0.dim);
        std::vector<uint64_t> assign(ix0.n);
        ivf_export(ix0, cent.data(), assign.data());
        for (size_t i = 1; i < sh->shards.size(); i++) {
            Index &ix = sh->shards[i]->handle->ix;
            ix.use_device();
            ivf_attach(ix, ix0.ivf.k, cent.data(), assign.data());
        }
    }
    VDB_API_END
}
int vdb_sharded_ivf_knn(vdb_sharded *sh, const float *queries, uint64_t nq, uint64_t dim, uint64_t k, uint64_t n_probes, uint64_t *out_idx,
                        float *out_dist, uint64_t *out_count) {
    VDB_API_BEGIN
    VDB_REQUIRE(sh, "null index");
    VDB_REQUIRE(dim == sh->dim, "query dimension mismatch");
    VDB_REQUIRE(nq == 0 || (queries && out_idx && out_dist), "null argument");
    VDB_REQUIRE(sh->layout == LAYOUT_REPLICA, "IVF behind the context runs on replicas: use vdb_sharded_set_rows_replica");
    for (auto &s : sh->shards) VDB_REQUIRE(s->handle->ix.ivf.present, "no IVF index (vdb_sharded_ivf_build)");
    if (n_probes == 0) n_probes = sh->shards[0]->handle->ix.ivf.default_n_probes;
    VDB_SHARDED_SEARCH_BEGIN(sh)
    replica_search(*sh, queries, nq, k, out_idx, out_dist, out_count,
                   [&](Index &ix, Workspace &ws, const float *d_q, uint64_t n, uint64_t *di, float *dd, uint64_t *dc) {
                       ivf_knn_device(ix, ws, d_q, n, k, n_probes, di, dd, dc);
                   });
    VDB_SHARDED_SEARCH_END(sh)
    VDB_API_END
}
// partition arithmetic of the REPLICA layout, exported for the host's tests: the query block of `rank` among `world`
int vdb_replica_query_block(uint64_t nq, uint64_t world, uint64_t rank, uint64_t *q0, uint64_t *q1) {
    VDB_API_BEGIN
    VDB_REQUIRE(q0 && q1 && world >= 1 && rank < world, "bad argument");
    replica_block(nq, world, rank, *q0, *q1);
    VDB_API_END
}
int vdb_sharded_poisoned(const vdb_sharded *sh, int *out) {
    VDB_API_BEGIN
    VDB_REQUIRE(sh && out, "null argument");
    *out = sh->poisoned ? 1 : 0;
    VDB_API_END
}

}  // extern "C"
