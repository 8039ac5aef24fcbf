// tsan_host.cpp -- thread-sanitizer run of the library's HOST side ("test with TSAN on the host shim").
// Built by tests/test_tsan_host_cpu.py: every csrc/*.hip compiled host-only (hipcc --cuda-host-only -fsanitize=thread
// -DVDB_HOST_SANITIZER_BUILD) + this file; no device is present or needed -- nothing here launches a kernel.
//   1. HNSWIndex::add_parallel as the host builder runs it (hnsw_index.rs:399-457): 16 threads search the candidates of a
//      batch against the pre-batch graph, the linking phase runs per link list in parallel; the graph must equal the
//      1-thread build's list by list (L2Sqr and Cosine).
//   2. PQ training's per-group k-means (k_means.rs:61-162) from 8 threads over disjoint column groups, as pq_build's
//      par_groups runs it; centroids must equal the serial run's.
//   3. The per-index Workspace pool (read-side calls are re-entrant: one workspace per concurrent reader) hammered by 16
//      threads while two more flip the process-wide tuning switches of vdb_set_param (atomics) and a third polls the
//      lazily-built-mirror flag / statistics the way vdb_get_stat does.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../lab_1806_vec_db_amd/csrc/index.hpp"
#include "../../lab_1806_vec_db_amd/csrc/pq_hnsw.hpp"

namespace vdb {
void hnsw_build_host_only(HNSWState &h, const float *rows, const float *sq, uint64_t n, uint64_t dim, int dist, uint64_t M,
                          uint64_t ef_construction, uint64_t seed, uint64_t batch, int nthreads);
void gemm_set_nt(int v);
void mfma_set_share(int v);
}  // namespace vdb

using namespace vdb;

static uint64_t sm(uint64_t &s) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static std::vector<float> random_rows(uint64_t n, uint64_t dim, uint64_t seed) {
    std::vector<float> r(n * dim);
    for (auto &v : r) v = float(sm(seed) >> 40) * (1.0f / 16777216.0f) - 0.5f;
    return r;
}
static bool same_graph(const HNSWState &a, const HNSWState &b) {
    return a.level0 == b.level0 && a.len0 == b.len0 && a.vec_level == b.vec_level && a.upper == b.upper && a.upper_len == b.upper_len &&
           a.enter_point == b.enter_point && a.enter_level == b.enter_level;
}

int main() {
    int bad = 0;
    // -- 1 ---------------------------------------------------------------------------------------------------------------
    {
        const uint64_t n = 3000, dim = 24;
        auto rows = random_rows(n, dim, 7);
        for (uint64_t i = 0; i < 5 * dim; i++) rows[(n - 5) * dim + i] = rows[i];  // duplicates: distance ties in the link lists
        std::vector<float> sq(n);
        for (uint64_t i = 0; i < n; i++) {
            float acc = 0.0f;
            for (uint64_t j = 0; j < dim; j++) acc = acc + rows[i * dim + j] * rows[i * dim + j];
            sq[i] = acc;
        }
        for (int dist = 0; dist < 2; dist++) {
            HNSWState serial, par;
            hnsw_build_host_only(serial, rows.data(), sq.data(), n, dim, dist, 8, 60, 42, 64, 1);
            hnsw_build_host_only(par, rows.data(), sq.data(), n, dim, dist, 8, 60, 42, 64, 16);
            const bool ok = same_graph(serial, par);
            std::printf("hnsw build, dist %d: 16-thread graph %s the 1-thread graph\n", dist, ok ? "equals" : "DIFFERS FROM");
            bad += ok ? 0 : 1;
        }
    }
    // -- 2 ---------------------------------------------------------------------------------------------------------------
    {
        const uint64_t nt = 2000, dim = 32, groups = 8, gd = dim / groups, kc = 16;
        auto train = random_rows(nt, dim, 11);
        std::vector<float> a(kc * dim), b(kc * dim);
        for (uint64_t g = 0; g < groups; g++) host_kmeans(train.data(), nt, dim, g * gd, (g + 1) * gd, kc, 5, 1e-6f, 0, 100 + g, a.data() + kc * g * gd);
        std::vector<std::thread> th;
        for (uint64_t g = 0; g < groups; g++)
            th.emplace_back([&, g] { host_kmeans(train.data(), nt, dim, g * gd, (g + 1) * gd, kc, 5, 1e-6f, 0, 100 + g, b.data() + kc * g * gd); });
        for (auto &t : th) t.join();
        const bool ok = std::memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0;
        std::printf("k-means, 8 groups in parallel: centroids %s the serial run's\n", ok ? "equal" : "DIFFER FROM");
        bad += ok ? 0 : 1;
    }
    // -- 3 ---------------------------------------------------------------------------------------------------------------
    {
        Index ix(0, 16, 0);
        std::atomic<bool> stop{false};
        std::atomic<uint64_t> leases{0};
        std::vector<std::thread> th;
        for (int t = 0; t < 16; t++)
            th.emplace_back([&] {
                for (int i = 0; i < 2000; i++) {
                    WsLease ws(ix);
                    ws->pending.clear();
                    ix.half_queries += 1;
                    ix.fallback_count += 1;
                    leases++;
                }
            });
        std::thread flip_a([&] {
            for (int i = 0; !stop; i++) {
                hnsw_set_half(i & 1);
                ivf_set_q8(i & 1);
                pq_set_adc16(i & 1);
            }
        });
        std::thread flip_b([&] {
            for (int i = 0; !stop; i++) {
                hnsw_set_half(2);
                gemm_set_nt(i % 3);
                mfma_set_share(1 << (i & 1));
            }
        });
        std::thread poll([&] {
            uint64_t acc = 0;
            while (!stop) acc += ix.hbm_bytes_per_row() + ix.half_queries.load() + (ix.tiled_built ? 1 : 0);
            if (acc == 1) std::printf("-\n");
        });
        for (auto &t : th) t.join();
        stop = true;
        flip_a.join();
        flip_b.join();
        poll.join();
        const bool ok = leases == 32000 && ix.half_queries == 32000 && ix.ws_free.size() >= 1 && ix.ws_free.size() <= 16;
        std::printf("workspace pool: %llu leases over %zu pooled workspaces\n", (unsigned long long)leases.load(), ix.ws_free.size());
        bad += ok ? 0 : 1;
    }
    std::printf(bad ? "tsan_host: FAILED\n" : "tsan_host: ok\n");
    return bad ? 1 : 0;
}
