"""CPU rehearsal of the multi-GPU path with the gloo backend, world_size 2 and 3.

Each rank holds a contiguous row shard, produces its local top-k with GLOBAL ids (here by the CPU oracle,
standing in for the per-GPU kernels), then runs the product's all-gather + exact merge
(lab_1806_vec_db_amd.shard.allgather_merge).  The merged result must equal the unsharded result exactly.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, dim, nq, k, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lab_1806_vec_db_amd.shard import allgather_merge, shard_bounds
        from oracle import oracle as O

        rng = np.random.default_rng(123)
        base = rng.standard_normal((n, dim)).astype(np.float32)
        base[n // 2] = base[3]  # a duplicate row across shards -> exact distance tie across ranks
        qs = rng.standard_normal((nq, dim)).astype(np.float32)
        r0, r1 = shard_bounds(n, world, rank)
        li = np.zeros((nq, k), dtype=np.int64)
        ld = np.zeros((nq, k), dtype=np.float32)
        lc = np.zeros(nq, dtype=np.int64)
        for q in range(nq):
            i, d = O.flat_knn(base[r0:r1], qs[q], k) if r1 > r0 else (np.zeros(0, np.uint64), np.zeros(0, np.float32))
            c = len(i)
            li[q, :c] = i.astype(np.int64) + r0  # vdb_index_set_id_offset
            ld[q, :c] = d
            lc[q] = c
        mi, md, mc = allgather_merge(torch.from_numpy(li), torch.from_numpy(ld), torch.from_numpy(lc), k)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), idx=mi.numpy(), dist=md.numpy(), cnt=mc.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 501), (3, 100), (2, 7)])
def test_row_shard_allgather_merge_equals_unsharded(tmp_path, world, n):
    from oracle import oracle as O

    dim, nq, k = 24, 6, 10
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, dim, nq, k, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(123)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    base[n // 2] = base[3]
    qs = rng.standard_normal((nq, dim)).astype(np.float32)
    outs = [np.load(os.path.join(tmp_path, f"r{r}.npz")) for r in range(world)]
    for q in range(nq):
        oi, od = O.flat_knn(base, qs[q], k)
        for o in outs:  # every rank holds the full merged answer
            c = int(o["cnt"][q])
            assert c == len(oi)
            assert o["idx"][q, :c].tolist() == oi.tolist()
            assert np.array_equal(o["dist"][q, :c], od)


def _pq_shard_rows(O, base, opq, q, r0, r1, efk, kind):
    """What vdb_flat_knn_pq_shard exports for rows [r0, r1): the shard's ADC top-efk as pair-key rows."""
    adc = opq.adc_all(q, base.shape[0])[r0:r1]
    ids = np.arange(r0, r1, dtype=np.uint64)
    ak = np.sort(O.pair_keys(adc, ids))[:efk]
    sel = (ak & np.uint64(0xFFFFFFFF)).astype(np.int64)
    ex = np.array([O.dist(kind, base[i], q) for i in sel], dtype=np.float32)
    a = np.full(efk, np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    e = a.copy()
    a[:len(ak)] = ak
    e[:len(ak)] = O.pair_keys(ex, sel)
    return a, e


def _pq_corpus(n, dim):
    rng = np.random.default_rng(321)
    uniq = rng.standard_normal((n, dim)).astype(np.float32)
    uniq[n // 2:n // 2 + 5] = uniq[:5]  # duplicates across shards: equal codes -> ADC ties, exact-distance ties
    qs = rng.standard_normal((5, dim)).astype(np.float32)
    return uniq, qs


def _pq_worker(rank, world, port, n, dim, k, ef, kind, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lab_1806_vec_db_amd.shard import allgather_merge_pq, shard_bounds
        from oracle import oracle as O

        base, qs = _pq_corpus(n, dim)
        opq = O.PQ.train(base, 8, 4, kind, max_iter=4, seed=7)  # deterministic: every rank trains the same table
        r0, r1 = shard_bounds(n, world, rank)
        efk = max(ef, k)
        rows = [_pq_shard_rows(O, base, opq, q, r0, r1, efk, kind) for q in qs]
        a = np.stack([r[0] for r in rows]).view(np.int64)
        e = np.stack([r[1] for r in rows]).view(np.int64)
        mi, md, mc = allgather_merge_pq(torch.from_numpy(a), torch.from_numpy(e), k)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), idx=mi.numpy(), dist=md.numpy(), cnt=mc.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,k,ef,kind", [(2, 301, 10, 40, 0), (3, 120, 7, 7, 1), (2, 9, 10, 30, 0)])
def test_row_shard_knn_pq_equals_unsharded(tmp_path, world, n, k, ef, kind):
    """PQ-Flat shards: per-shard ADC top-ef gathered, merged in (adc, id) order, THEN re-sorted."""
    from oracle import oracle as O

    dim = 24
    port = _free_port()
    mp.spawn(_pq_worker, args=(world, port, n, dim, k, ef, kind, str(tmp_path)), nprocs=world, join=True)
    base, qs = _pq_corpus(n, dim)
    opq = O.PQ.train(base, 8, 4, kind, max_iter=4, seed=7)
    outs = [np.load(os.path.join(tmp_path, f"r{r}.npz")) for r in range(world)]
    for q in range(qs.shape[0]):
        oi, od = O.flat_knn_pq(base, opq, qs[q], k, ef, kind)
        for o in outs:
            c = int(o["cnt"][q])
            assert c == len(oi)
            assert o["idx"][q, :c].tolist() == oi.tolist()
            assert np.array_equal(o["dist"][q, :c], od)


def _replica_worker(rank, world, port, nq, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lab_1806_vec_db_amd.shard import allgather_concat, replica_query_slice
        from oracle import oracle as O

        rng = np.random.default_rng(99)
        base = rng.standard_normal((300, 16)).astype(np.float32)
        qs = rng.standard_normal((nq, 16)).astype(np.float32)
        h = O.HNSW.build(base, 0, M=8, ef_construction=40, seed=3)  # deterministic: identical replica on every rank
        q0, q1 = replica_query_slice(nq, world, rank)
        k = 5
        li = np.zeros((q1 - q0, k), dtype=np.int64)
        ld = np.zeros((q1 - q0, k), dtype=np.float32)
        lc = np.zeros(q1 - q0, dtype=np.int64)
        for j, q in enumerate(range(q0, q1)):
            i, d = h.knn(qs[q], k, 32)
            li[j, :len(i)], ld[j, :len(i)], lc[j] = i.astype(np.int64), d, len(i)
        gi, gd, gc = allgather_concat(torch.from_numpy(li), torch.from_numpy(ld), torch.from_numpy(lc), nq)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), idx=gi.numpy(), dist=gd.numpy(), cnt=gc.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nq", [(2, 7), (3, 3), (2, 1)])
def test_hnsw_replicas_split_queries(tmp_path, world, nq):
    """HNSW = replicas only: queries are dealt in contiguous blocks, answers concatenated."""
    from oracle import oracle as O

    port = _free_port()
    mp.spawn(_replica_worker, args=(world, port, nq, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(99)
    base = rng.standard_normal((300, 16)).astype(np.float32)
    qs = rng.standard_normal((nq, 16)).astype(np.float32)
    h = O.HNSW.build(base, 0, M=8, ef_construction=40, seed=3)
    for r in range(world):
        o = np.load(os.path.join(tmp_path, f"r{r}.npz"))
        assert o["idx"].shape == (nq, 5)
        for q in range(nq):
            i, d = h.knn(qs[q], 5, 32)
            assert o["idx"][q, :len(i)].tolist() == i.tolist() and np.array_equal(o["dist"][q, :len(i)], d)


def test_shard_bounds():
    from lab_1806_vec_db_amd.shard import shard_bounds

    assert [shard_bounds(10, 3, r) for r in range(3)] == [(0, 4), (4, 8), (8, 10)]
    assert [shard_bounds(2, 4, r) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert shard_bounds(1_000_000, 8, 7) == (875000, 1000000)
