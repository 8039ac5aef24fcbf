// C++ host test of include/vdbhip.hpp: the scenario of the reference's own DB test (database/mod.rs:551-607) and of
// examples/test_pyo3.py:13-32, driven through the DynamicIndex mirror.  `--no-gpu` only checks that construction
// fails loudly without a device (there is no CPU path).
#include <cmath>
#include <cstdio>
#include <cstring>

#include "vdbhip.hpp"

using namespace vdbhip;

#define EXPECT(cond)                                                       \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(int argc, char **argv) {
    if (argc > 1 && std::strcmp(argv[1], "--no-gpu") == 0) {
        try {
            DynamicIndex ix(4, DistanceAlgorithm::Cosine);
        } catch (const Error &e) {
            std::printf("no gpu: %s\n", e.what());
            return e.code == VDB_ERR_NOGPU ? 0 : 1;
        }
        std::printf("a device is present\n");
        return 0;
    }
    // 27.0 KAT (distance/mod.rs:137-143)
    EXPECT(calc_dist({1, 2, 3}, {4, 5, 6}, DistanceAlgorithm::L2Sqr) == 27.0f);
    DynamicIndex ix(4, DistanceAlgorithm::Cosine);
    EXPECT(ix.is_empty() && ix.dim() == 4);
    EXPECT(ix.add({1, 0, 0, 0}) == 0);
    EXPECT(ix.add({0, 1, 0, 0}) == 1);
    ix.build_hnsw();
    EXPECT(ix.add({0, 0, 1, 0}) == 2);  // HNSWIndex::add keeps the graph (test_pyo3.py:19)
    EXPECT(ix.add({0, 0, 1, 1}) == 3);
    EXPECT(ix.len() == 4 && ix.has_hnsw());
    auto r = ix.knn({0, 0, 1, 0}, 3);
    EXPECT(r.size() == 3 && r[0].index == 2 && r[0].distance == 0.0f);
    EXPECT(r[1].index == 3 && std::fabs(r[1].distance - (1.0f - 1.0f / std::sqrt(2.0f))) < 1e-6f);
    ix.clear_hnsw();
    ix.swap_remove(3);  // delete -> Flat arm only (metadata_vec_table.rs:163-187)
    EXPECT(ix.len() == 3);
    ix.build_pq(4, 2);
    auto p = ix.knn_pq({0, 0, 1, 0}, 3, 3);
    EXPECT(p.size() == 3 && p[0].index == 2);
    // upper_bound = 0.5 keeps exactly one hit (database/mod.rs:596-606)
    size_t within = 0;
    for (auto &c : p) within += c.distance <= 0.5f;
    EXPECT(within == 1);
    bool threw = false;
    try {
        ix.add({1, 2, 3});
    } catch (const Error &) {
        threw = true;
    }
    EXPECT(threw);
    auto b = ix.knn_batch(ix.row(0).data(), 1, 2);
    EXPECT(b.size() == 1 && b[0].size() == 2 && b[0][0].index == 0);
    {   // the same corpus behind the multi-GPU context (one GPU here: no communicator needed, same code path otherwise)
        ShardedIndex sh(4, DistanceAlgorithm::Cosine, {0});
        const float rows[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        sh.set_rows(rows, 3);
        EXPECT(sh.len() == 3);
        const float q[4] = {0, 0, 1, 0};
        auto s1 = sh.knn_batch(q, 1, 2);
        EXPECT(s1.size() == 1 && s1[0].size() == 2 && s1[0][0].index == 2 && s1[0][0].distance == 0.0f);
        // REPLICA layout: all three DynamicIndex searches behind the context (the HNSW ones only here)
        ShardedIndex rp(4, DistanceAlgorithm::Cosine, {0});
        rp.set_rows_replica(rows, 3);
        rp.build_hnsw();
        auto h1 = rp.knn_with_ef_batch(q, 1, 3, 3);
        EXPECT(h1.size() == 1 && h1[0].size() == 3 && h1[0][0].index == 2 && h1[0][0].distance == 0.0f && !rp.poisoned());
        auto f1 = rp.knn_batch(q, 1, 2);
        EXPECT(f1[0].size() == 2 && f1[0][0].index == 2);
        bool refused = false;
        try {
            sh.build_hnsw();  // a graph does not shard by rows
        } catch (const Error &) {
            refused = true;
        }
        EXPECT(refused);
    }
    std::printf("cpp host ok\n");
    return 0;
}
