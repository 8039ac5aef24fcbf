"""GPU: read-side entry points are re-entrant on one handle (the reference serves concurrent readers under an
RwLock read guard with the GIL released: database/mod.rs:248-256, pyo3/mod.rs:209, examples/test_multi_threads.py)."""
import threading

import numpy as np
import pytest

from conftest import gist_like

pytestmark = pytest.mark.gpu


def test_concurrent_readers_same_results():
    import lab_1806_vec_db_amd as vdb

    base = gist_like(30000, dim=128, seed=5)
    qs = gist_like(64, dim=128, seed=6)
    ix = vdb.GpuIndex(128, "l2sqr")
    ix.batch_add(base)
    ix.pq_build(n_bits=4, m=32, train_n=2000, max_iter=5, seed=1)
    ix.hnsw_build(M=8, ef_construction=40, seed=2, batch=16, nthreads=8)
    ix.hnsw_clear()
    ref_flat = ix.flat_knn(qs, 10)
    ref_pq = ix.knn_pq(qs, 10, 64)
    ix.hnsw_build(M=8, ef_construction=40, seed=2, batch=16, nthreads=8)
    ref_hnsw = ix.knn_with_ef(qs, 10, 64)
    errors = []

    def worker(kind):
        try:
            for _ in range(6):
                if kind == 0:
                    r, ref = ix.flat_knn(qs, 10), ref_flat
                elif kind == 1:
                    r, ref = ix.knn_with_ef(qs, 10, 64), ref_hnsw
                else:
                    r, ref = ix._search(ix._lib.vdb_flat_knn_pq, qs, 10, 64), ref_pq
                assert np.array_equal(r[0], ref[0]) and np.array_equal(r[1], ref[1])
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(i % 3,)) for i in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_vecdb_search_interleaved_with_writes():
    """VecDB holds a reader/writer lock per table like the reference's RwLock (read guard in search, database/mod.rs:255;
    write guard in add / delete, thread_save.rs:108-113): searches running while another thread adds and deletes rows must
    neither fault (the writes reallocate HBM buffers) nor see metadata and rows out of step."""
    import lab_1806_vec_db_amd as vdb

    rng = np.random.default_rng(3)
    db = vdb.VecDB("")
    db.create_table_if_not_exists("t", 32, "l2sqr")
    fixed = rng.standard_normal((64, 32)).astype(np.float32)
    db.batch_add("t", fixed, [{"kind": "fixed", "id": str(i)} for i in range(64)])
    stop = threading.Event()
    errors = []

    def reader(seed):
        r = np.random.default_rng(seed)
        try:
            while not stop.is_set():
                j = int(r.integers(0, 64))
                hits = db.search("t", fixed[j], 1)
                # the fixed rows are never deleted: their nearest neighbour is themselves at distance 0
                assert len(hits) == 1 and hits[0][1] == 0.0 and hits[0][0] == {"kind": "fixed", "id": str(j)}, hits
                assert db.get_len("t") >= 64
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def writer():
        try:
            for it in range(30):
                extra = (rng.standard_normal((500, 32)) * 3 + 50).astype(np.float32)  # grows d_rows past its capacity
                db.batch_add("t", extra, [{"kind": "tmp"}] * 500)
                if it % 3 == 2:
                    assert db.delete("t", {"kind": "tmp"}) == 1500
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    rs = [threading.Thread(target=reader, args=(100 + i,)) for i in range(3)]
    w = threading.Thread(target=writer)
    for t in rs:
        t.start()
    w.start()
    w.join()
    stop.set()
    for t in rs:
        t.join()
    assert not errors, errors
    assert db.get_len("t") == 64


def test_two_indexes_searching_at_once_with_cooperative_sets():
    """Two tables on one GPU searched from two threads, both through the 8-bit pass with cooperative sets: the sets' start rendezvous
    (k_gemm8.hip) is bounded (~0.1 ms), so a pass whose members are held up by the OTHER index's grid runs unshared instead of
    waiting -- answers unchanged (oracle on a sample), and a call never takes longer than the two calls back to back plus the bound"""
    import time

    import lab_1806_vec_db_amd as vdb
    from oracle import oracle as O

    n, dim, nq = 110_000, 128, 1024
    rng = np.random.default_rng(42)
    tables = []
    for t in range(2):
        base = rng.standard_normal((n, dim)).astype(np.float32)
        qs = rng.standard_normal((nq, dim)).astype(np.float32)
        ix = vdb.GpuIndex(dim, "l2sqr" if t == 0 else "cosine")
        ix.batch_add(base)
        ix.set_flat_mode(2)
        ref = ix.flat_knn(qs, 10)  # alone (also builds the mirror)
        assert ix.get_stat("flat_gemm8_coop_sets") == 8 and ix.get_stat("flat_i8_queries") == nq
        sel = rng.choice(nq, 24, replace=False)
        oi, od, oc = O.flat_knn_batch(base, qs[sel], 10, O.L2SQR if t == 0 else O.COSINE, nthreads=8)
        assert np.array_equal(ref[0][sel], oi) and np.array_equal(ref[1][sel], od)
        tables.append((ix, qs, ref))
    alone = []
    for ix, qs, _ in tables:
        t0 = time.perf_counter()
        for _ in range(5):
            ix.flat_knn(qs, 10)
        alone.append((time.perf_counter() - t0) / 5)
    errors, worst = [], [0.0, 0.0]

    def worker(t):
        ix, qs, ref = tables[t]
        try:
            for _ in range(20):
                t0 = time.perf_counter()
                r = ix.flat_knn(qs, 10)
                worst[t] = max(worst[t], time.perf_counter() - t0)
                assert np.array_equal(r[0], ref[0]) and np.array_equal(r[1], ref[1])
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    print(f"alone {alone[0] * 1e3:.3f} / {alone[1] * 1e3:.3f} ms per call; worst concurrent call {worst[0] * 1e3:.3f} / {worst[1] * 1e3:.3f} ms")
    # two grids that each want every CU take turns at worst: the sum of both calls, plus the rendezvous bound and host jitter
    for t in range(2):
        assert worst[t] <= 3.0 * (alone[0] + alone[1]) + 2e-3, (worst, alone)
    for ix, _, _ in tables:
        ix.close()
