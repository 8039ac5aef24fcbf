"""ShardedIndex -- the multi-GPU context of the C ABI (vdb_ctx_* / vdb_sharded_*, include/vdbhip.h) from Python.

The library owns the RCCL communicator and the exchange: per-shard local top-k, ONE all-gather, exact
merge by (distance, index).  No torch involved; `shard.py` remains the torch.distributed path bench.py uses.

  ShardedIndex(dim, dist, devices=[0, 1, ...])                one process driving several GPUs (ncclCommInitAll)
  ShardedIndex(dim, dist, device=d, rank=r, world=S, uid=id)  one process per GPU; `id = ShardedIndex.unique_id()` on
                                                              rank 0, distributed by the caller
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .index import _f32, _ptr, parse_dist


class ShardedIndex:
    def __init__(self, dim: int, dist="cosine", devices=None, device: int | None = None, rank: int = 0, world: int = 1,
                 uid: bytes | None = None):
        self._lib = L.load()
        self._ctx = L.vp()
        self._h = L.vp()
        if device is not None:
            buf = C.create_string_buffer(uid, 128) if uid is not None else None
            L.check(self._lib.vdb_ctx_create_rank(int(device), buf, int(rank), int(world), C.byref(self._ctx)))
        else:
            devs = np.ascontiguousarray(devices if devices is not None else [0], dtype=np.int32)
            L.check(self._lib.vdb_ctx_create(_ptr(devs, L.intp), len(devs), C.byref(self._ctx)))
        L.check(self._lib.vdb_sharded_create(self._ctx, int(dim), parse_dist(dist), C.byref(self._h)))
        self.dim = int(dim)

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        L.check(L.load().vdb_ctx_unique_id(buf, 128))
        return buf.raw

    def info(self) -> dict:
        w, nl, fr, hc = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        L.check(self._lib.vdb_ctx_info(self._ctx, C.byref(w), C.byref(nl), C.byref(fr), C.byref(hc)))
        return {"world": w.value, "n_local": nl.value, "first_rank": fr.value, "has_comm": bool(hc.value)}

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vdb_sharded_destroy(self._h)
            self._h = None
        if getattr(self, "_ctx", None):
            self._lib.vdb_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        v = C.c_uint64()
        L.check(self._lib.vdb_sharded_len(self._h, C.byref(v)))
        return int(v.value)

    def set_rows(self, rows):
        """every process passes the same N x dim corpus; each GPU keeps its contiguous row block"""
        r = _f32(rows)
        if r.ndim != 2 or r.shape[1] != self.dim:
            raise L.VdbError(f"dimension mismatch: index dim {self.dim}, got shape {r.shape}")
        L.check(self._lib.vdb_sharded_set_rows(self._h, _ptr(r, L.f32p), r.shape[0]))

    def set_rows_replica(self, rows):
        """REPLICA layout: every GPU keeps all rows, searches split the queries (the only layout HNSW runs in)"""
        r = _f32(rows)
        if r.ndim != 2 or r.shape[1] != self.dim:
            raise L.VdbError(f"dimension mismatch: index dim {self.dim}, got shape {r.shape}")
        L.check(self._lib.vdb_sharded_set_rows_replica(self._h, _ptr(r, L.f32p), r.shape[0]))

    @property
    def layout(self) -> str:
        v = C.c_int()
        L.check(self._lib.vdb_sharded_layout(self._h, C.byref(v)))
        return {0: "unset", 1: "rows", 2: "replica"}[v.value]

    @property
    def poisoned(self) -> bool:
        v = C.c_int()
        L.check(self._lib.vdb_sharded_poisoned(self._h, C.byref(v)))
        return bool(v.value)

    def local_index(self, i: int = 0):
        """borrowed GpuIndex view of this process's i-th shard / replica (exports, statistics); do not close it"""
        from .index import GpuIndex

        h = L.vp()
        L.check(self._lib.vdb_sharded_local(self._h, int(i), C.byref(h)))
        g = GpuIndex.__new__(GpuIndex)
        g._lib, g._h, g.device, g._borrowed = self._lib, h, -1, True
        return g

    def local_stat(self, i: int, name: str) -> int:
        h = L.vp()
        L.check(self._lib.vdb_sharded_local(self._h, int(i), C.byref(h)))
        v = C.c_uint64()
        L.check(self._lib.vdb_get_stat(h, name.encode(), C.byref(v)))
        return int(v.value)

    def _search(self, fn, queries, k, ef=None):
        q = _f32(queries)
        q = q.reshape(1, -1) if q.ndim == 1 else q
        nq, dim = q.shape
        kk = max(int(k), 1)
        idx = np.zeros((nq, kk), dtype=np.uint64)
        dist = np.zeros((nq, kk), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint64)
        args = [self._h, _ptr(q, L.f32p), nq, dim, int(k)] + ([int(ef)] if ef is not None else [])
        L.check(fn(*args, _ptr(idx, L.u64p), _ptr(dist, L.f32p), _ptr(cnt, L.u64p)))
        return idx[:, :int(k)], dist[:, :int(k)], cnt

    def flat_knn(self, queries, k: int):
        return self._search(self._lib.vdb_sharded_flat_knn, queries, k)

    def pq_attach(self, n_bits: int, m: int, centroids):
        c = _f32(centroids).ravel()
        if c.size != (1 << n_bits) * self.dim:
            raise L.VdbError(f"pq_attach: centroids hold {c.size} floats, expected (1 << n_bits) * dim")
        L.check(self._lib.vdb_sharded_pq_attach(self._h, n_bits, m, _ptr(c, L.f32p)))

    def knn_pq(self, queries, k: int, ef: int):
        return self._search(self._lib.vdb_sharded_knn_pq, queries, k, ef)

    # -- HNSW over the replicas (vdb_sharded_hnsw_*) ---------------------------------------------------------------------------
    def hnsw_build(self, M: int = 16, ef_construction: int = 200, seed: int = 42, batch: int = 1, nthreads: int = 0):
        L.check(self._lib.vdb_sharded_hnsw_build(self._h, int(M), int(ef_construction), int(seed), int(batch), int(nthreads)))

    def hnsw_attach(self, M: int, ef_construction: int, graph: dict):
        """graph = GpuIndex.hnsw_export() (level0, len0, vec_level, upper, upper_len, has_enter, enter_point, enter_level)"""
        l0 = np.ascontiguousarray(graph["level0"], dtype=np.uint32)
        n0 = np.ascontiguousarray(graph["len0"], dtype=np.uint64)
        vl = np.ascontiguousarray(graph["vec_level"], dtype=np.uint64)
        up = np.ascontiguousarray(graph["upper"], dtype=np.uint32)
        ul = np.ascontiguousarray(graph["upper_len"], dtype=np.uint64)
        L.check(self._lib.vdb_sharded_hnsw_attach(self._h, int(M), int(ef_construction), _ptr(l0, L.u32p), _ptr(n0, L.u64p), _ptr(vl, L.u64p),
                                                  _ptr(up, L.u32p), _ptr(ul, L.u64p), int(bool(graph["has_enter"])), int(graph["enter_point"]),
                                                  int(graph["enter_level"])))

    def knn_with_ef(self, queries, k: int, ef: int = 0):
        return self._search(self._lib.vdb_sharded_hnsw_knn, queries, k, ef)

    def hnsw_knn_pq(self, queries, k: int, ef: int):
        return self._search(self._lib.vdb_sharded_hnsw_knn_pq, queries, k, ef)


    # -- IVF over the replicas -------------------------------------------------------------------------------------------------
    def ivf_build(self, k: int, train_n: int = 0, max_iter: int = 20, tol: float = 1e-6, seed: int = 42):
        L.check(self._lib.vdb_sharded_ivf_build(self._h, int(k), int(train_n), int(max_iter), float(tol), int(seed)))

    def ivf_knn(self, queries, k: int, n_probes: int = 0):
        return self._search(self._lib.vdb_sharded_ivf_knn, queries, k, n_probes)


def replica_query_block(nq: int, world: int, rank: int) -> tuple[int, int]:
    """the library's partition arithmetic of the REPLICA layout (vdb_replica_query_block); equals shard.replica_query_slice"""
    a, b = C.c_uint64(), C.c_uint64()
    L.check(L.load().vdb_replica_query_block(int(nq), int(world), int(rank), C.byref(a), C.byref(b)))
    return int(a.value), int(b.value)
