// ctx.hpp -- pieces shared by api.hip and ctx.hip: the handle type behind vdb_index, error plumbing, the shard merge.
#pragma once
#include <string>

#include "../../include/vdbhip.h"
#include "index.hpp"

struct vdb_index {
    vdb::Index ix;
    vdb_index(int dev, uint64_t dim, int dist, bool u8 = false) : ix(dev, dim, dist, u8) {}
};

namespace vdb {
void set_last_error(const std::string &m);
void require_gpu();  // throws VDB_ERR_NOGPU when no HIP device is usable
// merge of S per-shard result lists on the index's GPU (arrays S x [nq][k], byte strides between shards); returns
// synchronised.  k <= 1024.
void merge_topk_dev(Index &ix, const void *d_dists, const void *d_ids, const void *d_counts, uint64_t stride_d,
                    uint64_t stride_i, uint64_t stride_c, uint64_t n_shards, uint64_t nq, uint64_t k, void *d_out_idx,
                    void *d_out_dist, void *d_out_count, void *stream);
}  // namespace vdb

#define VDB_API_BEGIN try {
#define VDB_API_END                                   \
    return VDB_OK;                                    \
    }                                                 \
    catch (const vdb::Error &e) {                     \
        vdb::set_last_error(e.what());                \
        return e.code;                                \
    }                                                 \
    catch (const std::exception &e) {                 \
        vdb::set_last_error(e.what());                \
        return VDB_ERR_INVALID;                       \
    }                                                 \
    catch (...) {                                     \
        vdb::set_last_error("unknown error");         \
        return VDB_ERR_INVALID;                       \
    }
