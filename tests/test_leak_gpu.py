"""GPU: no device memory is left behind -- indexes of every kind created, searched through every front end (host pointers, the
pinned one-launch path, device pointers, calls in two halves, the sharded context) and destroyed, many times; free HBM before and
after must agree (hipMemGetInfo through torch)."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_create_search_destroy_cycles_return_all_memory():
    torch = pytest.importorskip("torch")
    import lab_1806_vec_db_amd as vdb
    from lab_1806_vec_db_amd.sharded import ShardedIndex

    rng = np.random.default_rng(1)
    base = rng.standard_normal((20000, 64)).astype(np.float32)
    qs = rng.standard_normal((40, 64)).astype(np.float32)
    dq = torch.from_numpy(qs).cuda()
    o_i = torch.zeros((40, 10), dtype=torch.int64, device="cuda")
    o_d = torch.zeros((40, 10), dtype=torch.float32, device="cuda")
    o_c = torch.zeros((40,), dtype=torch.int64, device="cuda")

    def cycle():
        ix = vdb.GpuIndex(64, "l2sqr")
        ix.batch_add(base)
        ix.flat_knn(qs[:1], 10)       # one-launch kernel beyond 16 384 rows (one query)
        ix.flat_knn(qs, 10)           # MFMA pipeline
        ix.set_flat_mode(1)
        ix.flat_knn(qs[:3], 10)       # exact scan
        ix.set_flat_mode(0)
        hs = [ix.flat_knn_device_begin(dq.data_ptr(), 40, 10, o_i.data_ptr(), o_d.data_ptr(), o_c.data_ptr()) for _ in range(3)]
        for h in hs:
            ix.flat_knn_device_end(h)
        ix.pq_build(n_bits=4, m=16, train_n=1000, max_iter=2, seed=1)
        ix.knn_pq(qs[:5], 10, 50)
        ix.ivf_build(12, train_n=500, max_iter=2, seed=1)
        ix.ivf_knn(qs[:5], 10, 4)
        ix.ivf_clear()
        ix.pq_clear()  # (swap_remove refuses to run under a PQ table or IVF clusters: metadata_vec_table.rs:170-171 clears them first)
        ix.swap_remove(5)
        small = vdb.GpuIndex(64, "cosine")
        small.batch_add(base[:2000])
        small.hnsw_build(M=8, ef_construction=40, seed=1, batch=1, nthreads=2)
        small.knn_with_ef(qs[:5], 10, 40)
        small.flat_knn(qs[:2], 5)
        small.close()
        ix.close()
        sh = ShardedIndex(64, "l2sqr", devices=[0])
        sh.set_rows_replica(base[:3000])
        sh.flat_knn(qs[:4], 5)
        sh.close()

    cycle()  # first use: the runtime's own pools, the library's code objects
    gc.collect()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(12):
        cycle()
    gc.collect()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (64 << 20), (free0, free1)  # (the allocator keeps granules around; a leak of 12 cycles would be hundreds of MB)
