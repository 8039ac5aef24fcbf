// mt_small_calls.cpp -- db.search()-shaped calls (one query, the reference's gist_1000 table) from T host threads at once, straight
// through the C ABI: how the library scales with concurrent readers when no Python sits in between (tooling).
// build: g++ -O2 -std=c++17 -Iinclude tools/mt_small_calls.cpp -o /tmp/mt_small -Llab_1806_vec_db_amd -lvdbhip -Wl,-rpath,$PWD/lab_1806_vec_db_amd -pthread
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "vdbhip.h"

static std::vector<float> load(const char *path, size_t n) {
    std::vector<float> v(n);
    FILE *f = std::fopen(path, "rb");
    if (!f || std::fread(v.data(), 4, n, f) != n) {
        std::fprintf(stderr, "cannot read %s\n", path);
        std::exit(1);
    }
    std::fclose(f);
    return v;
}

int main(int argc, char **argv) {
    const int rows_mult = argc > 1 ? std::atoi(argv[1]) : 1;
    auto base = load("tests/golden/gist_1000.bin", 960000), test = load("tests/golden/gist_test.bin", 960000);
    vdb_index *ix = nullptr;
    if (vdb_index_create(0, 960, VDB_L2SQR, &ix) != VDB_OK) {
        std::fprintf(stderr, "%s\n", vdb_last_error());
        return 1;
    }
    for (int r = 0; r < rows_mult; r++) vdb_index_add(ix, base.data(), 1000, nullptr);
    for (int T : {1, 2, 4, 8, 16, 32}) {
        const int per = 2000;
        auto run = [&](int t, bool timed) {
            uint64_t idx[10], cnt;
            float d[10];
            for (int i = 0; i < (timed ? per : 50); i++) {
                const float *q = test.data() + size_t((t * 131 + i) % 1000) * 960;
                if (vdb_flat_knn(ix, q, 1, 960, 10, idx, d, &cnt) != VDB_OK) std::abort();
            }
        };
        {
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back(run, t, false);
            for (auto &x : th) x.join();
        }
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back(run, t, true);
        for (auto &x : th) x.join();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("rows %d, %2d threads: %8.0f queries/s (%.1f us per call per thread)\n", 1000 * rows_mult, T, T * per / s, s / per * 1e6);
    }
    vdb_index_destroy(ix);
    return 0;
}
