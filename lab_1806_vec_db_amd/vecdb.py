"""VecDB-shaped host surface over the C ABI (mirror of lab_1806_vec_db.pyi / src/pyo3/mod.rs:55-296).

Scope: this mirrors the reference's table / search semantics so the hot path can sit behind
`db.search() / add() / build_*_index()`; tables live in memory (HBM + host metadata).  Persistence, the
directory lock and the autosave thread (src/database/{mod,thread_save}.rs) are the reference's control
plane and are out of scope -- `dir` is accepted for signature compatibility only.

Semantics reproduced from src/database/metadata_vec_table.rs:
  * add / batch_add clear the PQ table (:64-81); an HNSW graph survives add (examples/test_pyo3.py:19);
  * delete(pattern) clears HNSW and PQ, then swap-removes matches in descending order (:163-187);
  * build_hnsw_index only when currently Flat, M=16, ef_construction=200 unless given (:84-98);
  * build_pq_table: proportion 0.1 default, m = ceil(dim/3) default, n_bits is validated but the table is
    always built with 4 bits (:112-152 -- a reference quirk, kept);
  * search dispatch (:194-212): (ef, pq) -> knn_pq; (ef, no pq) -> knn_with_ef; else knn; then the
    `distance <= upper_bound` filter.
"""
from __future__ import annotations

import threading
from contextlib import contextmanager

import numpy as np

from ._lib import VdbError
from .index import GpuIndex, parse_dist

_DIST_STR = {0: "l2sqr", 1: "cosine"}


class _RwLock:
    """Reader/writer exclusion of one table: the reference wraps every table in an RwLock -- read guard in search /
    extract_data / len (database/mod.rs:248-256), write guard in add / delete / build_* / clear_* (thread_save.rs:108-113).
    The library's read-side calls are re-entrant on one handle, its write-side calls reallocate HBM buffers under running
    kernels, so a search must never overlap a write on the same table.  Writers are preferred (a waiting writer holds
    back new readers), like std's RwLock on Linux."""

    def __init__(self):
        self._cv = threading.Condition(threading.Lock())
        self._readers = 0
        self._writer = False
        self._writers_waiting = 0

    @contextmanager
    def read(self):
        with self._cv:
            while self._writer or self._writers_waiting:
                self._cv.wait()
            self._readers += 1
        try:
            yield
        finally:
            with self._cv:
                self._readers -= 1
                if self._readers == 0:
                    self._cv.notify_all()

    @contextmanager
    def write(self):
        with self._cv:
            self._writers_waiting += 1
            while self._writer or self._readers:
                self._cv.wait()
            self._writers_waiting -= 1
            self._writer = True
        try:
            yield
        finally:
            with self._cv:
                self._writer = False
                self._cv.notify_all()


class _Table:
    def __init__(self, dim: int, dist: str, device: int):
        self.index = GpuIndex(dim, dist, device)
        self.metadata: list[dict[str, str]] = []
        self.lock = _RwLock()
        self.seed = 0x1806

    def next_seed(self) -> int:
        self.seed = (self.seed * 6364136223846793005 + 1442695040888963407) & ((1 << 64) - 1)
        return self.seed


class VecDB:
    def __init__(self, dir: str = "", device: int = 0) -> None:
        self.dir = dir
        self.device = device
        self._tables: dict[str, _Table] = {}
        self._mu = threading.Lock()

    # ---- table management (in memory) -----------------------------------------------------------------
    def create_table_if_not_exists(self, key: str, dim: int, dist: str = "cosine") -> bool:
        parse_dist(dist)  # ValueError on a bad name (pyo3/mod.rs:15-22)
        with self._mu:
            if key in self._tables:
                return False
            self._tables[key] = _Table(dim, dist, self.device)
            return True

    def _t(self, key: str) -> _Table:
        try:
            return self._tables[key]
        except KeyError:
            raise RuntimeError(f"Table {key} not found") from None

    def get_len(self, key: str) -> int:
        t = self._t(key)
        with t.lock.read():
            return len(t.index)

    def get_dim(self, key: str) -> int:
        return self._t(key).index.dim

    def get_dist(self, key: str) -> str:
        return _DIST_STR[self._t(key).index.dist]

    def delete_table(self, key: str) -> bool:
        with self._mu:
            t = self._tables.pop(key, None)
        if t is None:
            return False
        with t.lock.write():  # wait for searches still running on the table
            t.index.close()
        return True

    def get_all_keys(self) -> list[str]:
        return sorted(self._tables)

    def contains_key(self, key: str) -> bool:
        return key in self._tables

    def get_cached_tables(self) -> list[str]:
        return self.get_all_keys()

    def contains_cached(self, key: str) -> bool:
        return key in self._tables

    def remove_cached_table(self, key: str) -> None:
        return None  # nothing is spilled to disk in this mirror

    def force_save(self) -> None:
        return None

    # ---- writes ---------------------------------------------------------------------------------------------
    def add(self, key: str, vec, metadata: dict[str, str]) -> None:
        self.batch_add(key, [vec], [metadata])

    def batch_add(self, key: str, vec_list, metadata_list) -> None:
        t = self._t(key)
        rows = np.asarray(vec_list, dtype=np.float32)
        if rows.ndim != 2 or rows.shape[1] != t.index.dim:  # database/mod.rs:427-429,445-447
            raise RuntimeError(f"Dimension mismatch: table dim {t.index.dim}, got {rows.shape}")
        if len(metadata_list) != rows.shape[0]:
            raise RuntimeError("vec_list and metadata_list differ in length")
        with t.lock.write():
            t.index.pq_clear()  # metadata_vec_table.rs:65,77
            t.metadata.extend(dict(m) for m in metadata_list)
            t.index.batch_add(rows)

    def delete(self, key: str, pattern: dict[str, str]) -> int:
        t = self._t(key)
        with t.lock.write():
            t.index.hnsw_clear()  # :170
            t.index.pq_clear()    # :171
            matches = [i for i, m in enumerate(t.metadata) if all(m.get(k) == v for k, v in pattern.items())]
            for i in reversed(matches):
                last = len(t.metadata) - 1
                t.metadata[i] = t.metadata[last]
                t.metadata.pop()
                t.index.swap_remove(i)
            return len(matches)

    def build_hnsw_index(self, key: str, ef_construction: int | None = None) -> None:
        t = self._t(key)
        with t.lock.write():
            if t.index.has_hnsw():
                return
            t.index.hnsw_build(M=16, ef_construction=200 if ef_construction is None else ef_construction,
                               seed=t.next_seed(), batch=1, nthreads=1)

    def clear_hnsw_index(self, key: str) -> None:
        t = self._t(key)
        with t.lock.write():
            t.index.hnsw_clear()

    def has_hnsw_index(self, key: str) -> bool:
        t = self._t(key)
        with t.lock.read():
            return t.index.has_hnsw()

    def build_pq_table(self, key: str, train_proportion: float | None = None, n_bits: int | None = None,
                       m: int | None = None) -> None:
        t = self._t(key)
        with t.lock.write():
            if t.index.has_pq():
                return
            n = len(t.index)
            if n == 0:
                raise RuntimeError("Cannot build PQ table for an empty table")
            prop = 0.1 if train_proportion is None else train_proportion
            if prop <= 0.0 or prop >= 1.0:
                raise RuntimeError("Train proportion must be in (0, 1)")
            train = int(max(np.float32(n) * np.float32(prop), np.float32(1.0)))
            nb = 4 if n_bits is None else n_bits
            if nb not in (4, 8):
                raise RuntimeError("n_bits must be 4 or 8")
            dim = t.index.dim
            mm = -(-dim // 3) if m is None else m
            if mm == 0 or mm > dim:
                raise RuntimeError("m must be in 1..=dim")
            # the reference validates n_bits but hard-codes 4 in the PQConfig (metadata_vec_table.rs:140)
            t.index.pq_build(n_bits=4, m=mm, train_n=train, max_iter=20, tol=1e-6, seed=t.next_seed())

    def clear_pq_table(self, key: str) -> None:
        t = self._t(key)
        with t.lock.write():
            t.index.pq_clear()

    def has_pq_table(self, key: str) -> bool:
        t = self._t(key)
        with t.lock.read():
            return t.index.has_pq()

    # ---- reads ------------------------------------------------------------------------------------------------
    def search(self, key: str, query, k: int, ef: int | None = None, upper_bound: float | None = None):
        t = self._t(key)
        ix = t.index
        q = np.asarray(query, dtype=np.float32).ravel()
        with t.lock.read():  # database/mod.rs:255: read guard for the whole search, metadata lookup included
            if ef is not None and ix.has_pq():
                idx, dist = ix.knn_pq(q, k, ef)
            elif ef is not None:
                idx, dist = ix.knn_with_ef(q, k, ef)
            else:
                idx, dist = ix.knn(q, k)
            ub = np.float32(np.inf) if upper_bound is None else np.float32(upper_bound)
            return [(dict(t.metadata[int(i)]), float(d)) for i, d in zip(idx, dist) if d <= ub]

    def extract_data(self, key: str):
        t = self._t(key)
        with t.lock.read():
            return [(t.index[i].tolist(), dict(t.metadata[i])) for i in range(len(t.index))]


__all__ = ["VecDB", "VdbError"]
