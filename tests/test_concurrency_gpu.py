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
