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
    assert sh.local_stat(0, "flat_half_queries") > 0
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
