"""GPU: the multi-GPU context of the C ABI (vdb_ctx_create / vdb_sharded_*).  A test box has ONE GPU, so the
communicator is exercised with one rank: once without it (world = 1 needs none) and once with VDB_CTX_FORCE_RCCL=1,
which makes the library dlopen RCCL, build a 1-rank communicator (ncclCommInitAll / ncclCommInitRank) and run the real
grouped ncclAllGather calls on the shard's stream before the merge.  Answers must equal the plain index's and the oracle's."""
import os

import numpy as np
import pytest

from conftest import gist_like

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["nocomm", "comm_all", "comm_rank"])
def test_sharded_context_one_gpu(mode, monkeypatch):
    import lab_1806_vec_db_amd as vdb
    from lab_1806_vec_db_amd.sharded import ShardedIndex
    from oracle import oracle as O

    if mode != "nocomm":
        monkeypatch.setenv("VDB_CTX_FORCE_RCCL", "1")
    n, dim = 40000, 96
    base = gist_like(n, dim=dim, seed=91)
    base[20000:20010] = base[:10]
    qs = gist_like(130, dim=dim, seed=92)
    if mode == "comm_rank":
        sh = ShardedIndex(dim, "l2sqr", device=0, rank=0, world=1, uid=ShardedIndex.unique_id())
    else:
        sh = ShardedIndex(dim, "l2sqr", devices=[0])
    info = sh.info()
    assert info["world"] == 1 and info["n_local"] == 1 and info["has_comm"] == (mode != "nocomm")
    sh.set_rows(base)
    assert len(sh) == n
    ref = vdb.GpuIndex(dim, "l2sqr")
    ref.batch_add(base)
    for k in (10, 100):
        a, b = sh.flat_knn(qs, k), ref.flat_knn(qs, k)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    for q in (0, 64, 129):
        oi, od = O.flat_knn(base, qs[q], 10)
        gi, gd, _ = sh.flat_knn(qs[q], 10)
        assert gi[0].tolist() == oi.tolist() and np.array_equal(gd[0], od)
    ref.pq_build(n_bits=4, m=32, train_n=2000, max_iter=4, seed=3)
    cent = ref.pq_export()["centroids"]
    sh.pq_attach(4, 32, cent)
    for k, ef in ((10, 64), (10, 1500)):
        a = sh.knn_pq(qs[:20], k, ef)
        b = ref._search(ref._lib.vdb_flat_knn_pq, qs[:20], k, ef)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert sh.local_stat(0, "flat_i8_queries") + sh.local_stat(0, "flat_half_queries") > 0  # (a shortlist pass answered, not the exact scan)
    sh.close()


def test_context_argument_errors():
    import lab_1806_vec_db_amd as vdb
    from lab_1806_vec_db_amd.sharded import ShardedIndex

    with pytest.raises(vdb.VdbError):
        ShardedIndex(8, "l2sqr", devices=[0, 0])       # a device twice
    with pytest.raises(vdb.VdbError):
        ShardedIndex(8, "l2sqr", devices=[99])         # no such device
    with pytest.raises(vdb.VdbError):
        ShardedIndex(8, "l2sqr", device=0, rank=2, world=2)
    sh = ShardedIndex(8, "l2sqr", devices=[0])
    sh.set_rows(np.zeros((4, 8), dtype=np.float32))
    with pytest.raises(vdb.VdbError):
        sh.set_rows(np.zeros((4, 8), dtype=np.float32))  # the partition is fixed once
    with pytest.raises(vdb.VdbError):
        sh.knn_pq(np.zeros(8, dtype=np.float32), 1, 4)   # no PQ table


@pytest.mark.parametrize("mode", ["nocomm", "comm_all"])
def test_replica_layout_all_three_searches(mode, monkeypatch):
    """REPLICA layout behind the C ABI (vdb_sharded_set_rows_replica): Flat, PQ-Flat, HNSW and HNSW+PQ with the queries split
    over the replicas and one fixed-size all-gather -- with one rank the block is the whole call, and with VDB_CTX_FORCE_RCCL=1
    the real ncclAllGather runs.  Answers equal the plain index's and the oracle's; HNSW refuses the row layout."""
    import lab_1806_vec_db_amd as vdb
    from lab_1806_vec_db_amd.sharded import ShardedIndex
    from oracle import oracle as O

    if mode != "nocomm":
        monkeypatch.setenv("VDB_CTX_FORCE_RCCL", "1")
    n, dim = 6000, 64
    base = gist_like(n, dim=dim, seed=71)
    qs = gist_like(77, dim=dim, seed=72)
    sh = ShardedIndex(dim, "l2sqr", devices=[0])
    assert sh.layout == "unset"
    sh.set_rows_replica(base)
    assert sh.layout == "replica" and len(sh) == n and not sh.poisoned
    ref = vdb.GpuIndex(dim, "l2sqr")
    ref.batch_add(base)
    a, b = sh.flat_knn(qs, 10), ref.flat_knn(qs, 10)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    a, b = sh.flat_knn(qs[:3], 1500), ref.flat_knn(qs[:3], 1500)  # no merge in this layout: any k
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    sh.hnsw_build(M=8, ef_construction=60, seed=5, batch=1, nthreads=1)
    g = sh.local_index(0).hnsw_export()
    oh = O.HNSW.from_graph(base, 0, 8, 60, g)
    gi, gd, gc = sh.knn_with_ef(qs, 10, 50)
    oi, od, oc, _, _ = oh.knn_batch(qs, 10, 50)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od) and gc.tolist() == oc.tolist()
    walk_idx = gi.copy()
    ref.pq_build(n_bits=4, m=16, train_n=1000, max_iter=3, seed=3)
    pq = ref.pq_export()
    sh.pq_attach(4, 16, pq["centroids"])
    a = sh.knn_pq(qs[:9], 5, 40)
    b = ref._search(ref._lib.vdb_flat_knn_pq, qs[:9], 5, 40)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    opq = O.PQ.from_centroids(dim, 16, 4, 0, pq["centroids"])
    opq.set_codes(pq["codes"])
    gi, gd, gc = sh.hnsw_knn_pq(qs[:9], 5, 40)
    for q in range(9):
        oi, od = oh.knn_pq(opq, qs[q], 5, 40)
        assert gi[q, :len(oi)].tolist() == oi.tolist() and np.array_equal(gd[q, :len(od)], od)
    # IVF over the replicas: same clusters, same answers as the plain index with the same seed
    sh.ivf_build(9, train_n=500, max_iter=3, seed=4)
    ref.ivf_build(9, train_n=500, max_iter=3, seed=4)
    a, b = sh.ivf_knn(qs[:11], 10, 3), ref.ivf_knn(qs[:11], 10, 3)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # a second graph attached from arrays replaces the first
    sh.hnsw_attach(8, 60, g)
    gi2, _, _ = sh.knn_with_ef(qs[:5], 10, 50)
    assert np.array_equal(gi2, walk_idx[:5])
    sh.close()
    rows = ShardedIndex(dim, "l2sqr", devices=[0])
    rows.set_rows(base)
    assert rows.layout == "rows"
    with pytest.raises(vdb.VdbError):
        rows.hnsw_build(M=8, ef_construction=60)
    with pytest.raises(vdb.VdbError):
        rows.knn_with_ef(qs[:2], 3, 10)
    rows.close()
