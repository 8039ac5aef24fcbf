// temporary stubs (replaced by pq.hip / hnsw.hip)
#include "pq_hnsw.hpp"
namespace vdb {
#define NI throw Error(3, "not implemented yet")
void hnsw_build(Index &, uint64_t, uint64_t, uint64_t, uint64_t, int) { NI; }
void hnsw_attach(Index &, uint64_t, uint64_t, const uint32_t *, const uint64_t *, const uint64_t *, const uint32_t *, const uint64_t *, int, uint64_t, uint64_t) { NI; }
void hnsw_clear(Index &ix) { ix.hnsw.present = false; }
void hnsw_insert_rows(Index &, const float *, uint64_t) { NI; }
void hnsw_knn_device(Index &, Workspace &, const float *, uint64_t, uint64_t, uint64_t, bool, uint64_t *, float *, uint64_t *) { NI; }
}
