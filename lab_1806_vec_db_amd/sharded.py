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
