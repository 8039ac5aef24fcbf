"""GpuIndex -- host-side mirror of the reference's index traits over the C ABI.

Method names and argument meaning follow src/index_algorithm/mod.rs (IndexIter, IndexKNN, IndexKNNWithEf,
IndexPQ, IndexBuilder) and src/database/dynamic_index.rs:11-94 (DynamicIndex).  All arithmetic runs in
libvdbhip.so on the GPU; this file only marshals numpy arrays.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

DIST_NAMES = {"l2sqr": L.L2SQR, "cosine": L.COSINE}  # pyo3/mod.rs:15-22


def parse_dist(dist) -> int:
    if isinstance(dist, str):
        try:
            return DIST_NAMES[dist.lower()]
        except KeyError:
            raise ValueError(f"Invalid distance function: {dist}") from None  # PyValueError, pyo3/mod.rs:20
    if dist in (0, 1):
        return int(dist)
    raise ValueError(f"Invalid distance function: {dist}")


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def calc_dist(a, b, dist="cosine", device: int = 0) -> float:
    """calc_dist (pyo3/mod.rs:43-48), evaluated on the GPU in reference order."""
    kind = parse_dist(dist)
    a, b = _f32(a).ravel(), _f32(b).ravel()
    n = min(a.size, b.size)  # zip truncates (distance/mod.rs:73)
    out = C.c_float(0)
    L.check(L.load().vdb_calc_dist(device, _ptr(a, L.f32p), _ptr(b, L.f32p), n, kind, C.byref(out)))
    return float(out.value)


def calc_dist_u8(a, b, dist="cosine", device: int = 0) -> float:
    """DistanceAdapter<[u8],[u8]> (distance/mod.rs:79-95,106-113): u8 elements widened exactly, f32 folds."""
    kind = parse_dist(dist)
    a = np.ascontiguousarray(a, dtype=np.uint8).ravel()
    b = np.ascontiguousarray(b, dtype=np.uint8).ravel()
    n = min(a.size, b.size)
    out = C.c_float(0)
    L.check(L.load().vdb_calc_dist_u8(device, _ptr(a, L.u8p), _ptr(b, L.u8p), n, kind, C.byref(out)))
    return float(out.value)


class GpuIndex:
    """One HBM-resident VecSet<f32> with optional PQ table and HNSW graph (DynamicIndex + PQTable)."""

    def __init__(self, dim: int, dist="cosine", device: int = 0, scalar: str = "f32"):
        """scalar = "f32" (DynamicIndex) or "u8": a VecSet<u8> held at one byte per element (Flat search only)"""
        self._lib = L.load()
        self._h = L.vp()
        self.device = device
        if scalar not in ("f32", "u8"):
            raise ValueError(f"Invalid scalar type: {scalar}")
        create = self._lib.vdb_index_create_u8 if scalar == "u8" else self._lib.vdb_index_create
        L.check(create(device, int(dim), parse_dist(dist), C.byref(self._h)))

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            if not getattr(self, "_borrowed", False):  # (ShardedIndex.local_index: the sharded index owns the handle)
                self._lib.vdb_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- IndexIter ----------------------------------------------------------------------------------
    def __len__(self) -> int:
        v = C.c_uint64()
        L.check(self._lib.vdb_index_len(self._h, C.byref(v)))
        return int(v.value)

    @property
    def dim(self) -> int:
        v = C.c_uint64()
        L.check(self._lib.vdb_index_dim(self._h, C.byref(v)))
        return int(v.value)

    @property
    def dist(self) -> int:
        v = C.c_int()
        L.check(self._lib.vdb_index_dist(self._h, C.byref(v)))
        return int(v.value)

    def __getitem__(self, i: int) -> np.ndarray:
        out = np.empty(self.dim, dtype=np.float32)
        L.check(self._lib.vdb_index_row(self._h, int(i), _ptr(out, L.f32p)))
        return out

    # -- IndexBuilder ----------------------------------------------------------------------------------
    def add(self, vec) -> int:
        return self.batch_add(_f32(vec).reshape(1, -1))[0]

    def batch_add(self, rows) -> list[int]:
        rows = _f32(rows)
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise L.VdbError(f"dimension mismatch: index dim {self.dim}, got shape {rows.shape}")
        first = C.c_uint64()
        L.check(self._lib.vdb_index_add(self._h, _ptr(rows, L.f32p), rows.shape[0], C.byref(first)))
        return list(range(int(first.value), int(first.value) + rows.shape[0]))

    def add_device(self, data_ptr: int, n: int) -> int:
        """Append n rows already resident on this GPU (e.g. torch tensor .data_ptr())."""
        first = C.c_uint64()
        L.check(self._lib.vdb_index_add_device(self._h, L.vp(data_ptr), int(n), C.byref(first)))
        return int(first.value)

    def swap_remove(self, i: int):
        L.check(self._lib.vdb_index_swap_remove(self._h, int(i)))

    def batch_add_u8(self, rows) -> int:
        """VecSet<u8> rows (widened exactly on the way in, distance/mod.rs:79-95)."""
        r = np.ascontiguousarray(rows, dtype=np.uint8)
        r = r.reshape(1, -1) if r.ndim == 1 else r
        if r.shape[1] != self.dim:
            raise L.VdbError(f"dimension mismatch: index dim {self.dim}, got {r.shape[1]}")
        first = C.c_uint64()
        L.check(self._lib.vdb_index_add_u8(self._h, _ptr(r, L.u8p), r.shape[0], C.byref(first)))
        return int(first.value)

    def row_u8(self, i: int) -> np.ndarray:
        out = np.empty(self.dim, dtype=np.uint8)
        L.check(self._lib.vdb_index_row_u8(self._h, int(i), _ptr(out, L.u8p)))
        return out

    def flat_knn_u8(self, queries, k: int):
        q = np.ascontiguousarray(queries, dtype=np.uint8)
        q = q.reshape(1, -1) if q.ndim == 1 else q
        nq, dim = q.shape
        idx = np.zeros((nq, max(k, 1)), dtype=np.uint64)
        dist = np.zeros((nq, max(k, 1)), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint64)
        L.check(self._lib.vdb_flat_knn_u8(self._h, _ptr(q, L.u8p), nq, dim, int(k), _ptr(idx, L.u64p),
                                          _ptr(dist, L.f32p), _ptr(cnt, L.u64p)))
        return idx[:, :k], dist[:, :k], cnt

    def set_id_offset(self, off: int):
        L.check(self._lib.vdb_index_set_id_offset(self._h, int(off)))

    # -- searches ------------------------------------------------------------------------------------------
    def _search(self, fn, queries, k, ef=None):
        q = _f32(queries)
        single = q.ndim == 1
        q = q.reshape(1, -1) if single else q
        nq, dim = q.shape
        kk = max(int(k), 1)
        idx = np.zeros((nq, kk), dtype=np.uint64)
        dist = np.zeros((nq, kk), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint64)
        args = [self._h, _ptr(q, L.f32p), nq, dim, int(k)]
        if ef is not None:
            args.append(int(ef))
        args += [_ptr(idx, L.u64p), _ptr(dist, L.f32p), _ptr(cnt, L.u64p)]
        L.check(fn(*args))
        if single:
            c = int(cnt[0])
            return idx[0, :c].copy(), dist[0, :c].copy()
        return idx[:, :int(k)], dist[:, :int(k)], cnt

    def knn(self, queries, k: int):
        """IndexKNN::knn: Flat -> FlatIndex::knn; with an HNSW graph -> HNSWIndex::knn (default ef)
        (dynamic_index.rs:68-73)."""
        if self.has_hnsw():
            return self._search(self._lib.vdb_hnsw_knn, queries, k, 0)
        return self._search(self._lib.vdb_flat_knn, queries, k)

    def flat_knn(self, queries, k: int):
        return self._search(self._lib.vdb_flat_knn, queries, k)

    def knn_with_ef(self, queries, k: int, ef: int):
        """IndexKNNWithEf::knn_with_ef; Flat ignores ef (dynamic_index.rs:75-80)."""
        if self.has_hnsw():
            return self._search(self._lib.vdb_hnsw_knn, queries, k, ef)
        return self._search(self._lib.vdb_flat_knn, queries, k)

    def knn_pq(self, queries, k: int, ef: int):
        """IndexPQ::knn_pq (dynamic_index.rs:82-93)."""
        if self.has_hnsw():
            return self._search(self._lib.vdb_hnsw_knn_pq, queries, k, ef)
        return self._search(self._lib.vdb_flat_knn_pq, queries, k, ef)

    def flat_knn_device(self, q_ptr: int, nq: int, k: int, out_idx_ptr: int, out_dist_ptr: int, out_cnt_ptr: int,
                        stream: int = 0):
        L.check(self._lib.vdb_flat_knn_device(self._h, L.vp(q_ptr), int(nq), self.dim, int(k), L.vp(out_idx_ptr),
                                              L.vp(out_dist_ptr), L.vp(out_cnt_ptr), L.vp(stream)))

    def flat_knn_device_begin(self, q_ptr: int, nq: int, k: int, out_idx_ptr: int, out_dist_ptr: int, out_cnt_ptr: int,
                              stream: int = 0):
        """first half of flat_knn_device (vdb_flat_knn_device_begin): enqueues the search behind `stream` and returns a pending
        handle for flat_knn_device_end; keep two batches in flight to pipeline independent query batches"""
        h = L.vp()
        L.check(self._lib.vdb_flat_knn_device_begin(self._h, L.vp(q_ptr), int(nq), self.dim, int(k), L.vp(out_idx_ptr),
                                                    L.vp(out_dist_ptr), L.vp(out_cnt_ptr), L.vp(stream), C.byref(h)))
        return h

    def flat_knn_device_end(self, pending):
        """second half: waits for the call, redoes uncertified queries, frees the handle; outputs valid on return"""
        L.check(self._lib.vdb_flat_knn_device_end(pending))

    def knn_pq_device(self, q_ptr: int, nq: int, k: int, ef: int, out_idx_ptr: int, out_dist_ptr: int,
                      out_cnt_ptr: int, stream: int = 0):
        L.check(self._lib.vdb_flat_knn_pq_device(self._h, L.vp(q_ptr), int(nq), self.dim, int(k), int(ef),
                                                 L.vp(out_idx_ptr), L.vp(out_dist_ptr), L.vp(out_cnt_ptr),
                                                 L.vp(stream)))

    def hnsw_knn_device(self, q_ptr: int, nq: int, k: int, ef: int, out_idx_ptr: int, out_dist_ptr: int,
                        out_cnt_ptr: int, use_pq: bool = False, stream: int = 0):
        L.check(self._lib.vdb_hnsw_knn_device(self._h, L.vp(q_ptr), int(nq), self.dim, int(k), int(ef), int(use_pq),
                                              L.vp(out_idx_ptr), L.vp(out_dist_ptr), L.vp(out_cnt_ptr), L.vp(stream)))

    def merge_topk_device(self, d_dists: int, d_ids: int, d_counts: int, n_shards: int, nq: int, k: int,
                          out_idx: int, out_dist: int, out_cnt: int, stream: int = 0):
        L.check(self._lib.vdb_merge_topk_device(self._h, L.vp(d_dists), L.vp(d_ids), L.vp(d_counts), n_shards, nq, k,
                                                L.vp(out_idx), L.vp(out_dist), L.vp(out_cnt), L.vp(stream)))

    # -- row-sharded knn_pq: local ADC shortlist as pair-key rows, merged after the all-gather ------
    def knn_pq_shard(self, queries, k: int, ef: int):
        """This shard's ADC top-max(ef,k): (adc_keys, exact_keys), each [nq, max(ef,k)] uint64 with GLOBAL ids."""
        q = _f32(queries)
        q = q.reshape(1, -1) if q.ndim == 1 else q
        efk = max(int(ef), int(k))
        adc = np.full((q.shape[0], efk), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
        ex = np.full((q.shape[0], efk), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
        L.check(self._lib.vdb_flat_knn_pq_shard(self._h, _ptr(q, L.f32p), q.shape[0], q.shape[1], int(k), int(ef),
                                                _ptr(adc, L.u64p), _ptr(ex, L.u64p)))
        return adc, ex

    def knn_pq_shard_device(self, q_ptr: int, nq: int, k: int, ef: int, out_adc_ptr: int, out_exact_ptr: int,
                            stream: int = 0):
        L.check(self._lib.vdb_flat_knn_pq_shard_device(self._h, L.vp(q_ptr), int(nq), self.dim, int(k), int(ef),
                                                       L.vp(out_adc_ptr), L.vp(out_exact_ptr), L.vp(stream)))

    def pq_merge_resort_device(self, d_adc: int, d_exact: int, n_shards: int, nq: int, efk: int, k: int,
                               out_idx: int, out_dist: int, out_cnt: int, stream: int = 0):
        L.check(self._lib.vdb_pq_merge_resort_device(self._h, L.vp(d_adc), L.vp(d_exact), n_shards, nq, efk, k,
                                                     L.vp(out_idx), L.vp(out_dist), L.vp(out_cnt), L.vp(stream)))

    def merge_topk_gathered(self, d_gathered: int, block_bytes: int, off_ids: int, off_dists: int, off_counts: int,
                            n_shards: int, nq: int, k: int, out_idx: int, out_dist: int, out_cnt: int, stream: int = 0):
        L.check(self._lib.vdb_merge_topk_gathered(self._h, L.vp(d_gathered), block_bytes, off_ids, off_dists, off_counts,
                                                  n_shards, nq, k, L.vp(out_idx), L.vp(out_dist), L.vp(out_cnt),
                                                  L.vp(stream)))

    def merge_topk_gathered_async(self, d_gathered: int, block_bytes: int, off_ids: int, off_dists: int, off_counts: int,
                                  n_shards: int, nq: int, k: int, out_idx: int, out_dist: int, out_cnt: int, stream: int):
        """vdb_merge_topk_gathered_async: the merge enqueued on `stream`, no host synchronisation (k <= 64)"""
        L.check(self._lib.vdb_merge_topk_gathered_async(self._h, L.vp(d_gathered), block_bytes, off_ids, off_dists, off_counts,
                                                        n_shards, nq, k, L.vp(out_idx), L.vp(out_dist), L.vp(out_cnt),
                                                        L.vp(stream)))

    def flat_shortlist_keys(self, queries, tier: int):
        """approximate keys of the Flat shortlist pass for every row (vdb_flat_shortlist_keys): (keys [nq, len], qsq [nq],
        qerr [nq], dict(dx_abs, dx_rel, xsq_max, xsq_min_pos)); tier 0 = fp16 operands, 1 = split-bf16 operands"""
        q = _f32(queries)
        q = q.reshape(1, -1) if q.ndim == 1 else q
        keys = np.zeros((q.shape[0], len(self)), dtype=np.float32)
        qsq = np.zeros(q.shape[0], dtype=np.float32)
        qerr = np.zeros(q.shape[0], dtype=np.float32)
        dx = np.zeros(4, dtype=np.float32)
        L.check(self._lib.vdb_flat_shortlist_keys(self._h, _ptr(q, L.f32p), q.shape[0], q.shape[1], int(tier), _ptr(keys, L.f32p),
                                                  _ptr(qsq, L.f32p), _ptr(qerr, L.f32p), _ptr(dx, L.f32p)))
        return keys, qsq, qerr, {"dx_abs": float(dx[0]), "dx_rel": float(dx[1]), "xsq_max": float(dx[2]), "xsq_min_pos": float(dx[3])}

    def prepare(self, all_tiers: bool = False):
        """build now the operand mirrors the first Flat search would build (vdb_index_prepare); all_tiers: those of the redo tiers too"""
        L.check(self._lib.vdb_index_prepare(self._h, 1 if all_tiers else 0))

    def set_flat_mode(self, mode: int):
        L.check(self._lib.vdb_flat_set_mode(self._h, int(mode)))

    def set_param(self, name: str, value: int):
        L.check(self._lib.vdb_set_param(self._h, name.encode(), int(value)))

    def flat_fallback_count(self) -> int:
        v = C.c_uint64()
        L.check(self._lib.vdb_flat_fallback_count(self._h, C.byref(v)))
        return int(v.value)

    def get_stat(self, name: str) -> int:
        v = C.c_uint64()
        L.check(self._lib.vdb_get_stat(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    # -- PQ ------------------------------------------------------------------------------------------------
    def pq_attach(self, n_bits: int, m: int, centroids, codes=None):
        """PQTable fields (pq_table.rs:116-137) as arrays: centroids = (1 << n_bits) * dim floats in group order,
        codes = len x ceil(m * n_bits / 8) bytes or None (encoded on the GPU).  Sizes are checked here: the C side
        copies exactly that many elements."""
        c = _f32(centroids).ravel()
        cd = None if codes is None else np.ascontiguousarray(codes, dtype=np.uint8)
        if n_bits not in (4, 8):
            raise L.VdbError("n_bits must be 4 or 8 in PQTable.")
        if not 0 < int(m) <= self.dim:
            raise L.VdbError("m must be in 1..=dim")
        if c.size != (1 << n_bits) * self.dim:
            raise L.VdbError(f"pq_attach: centroids hold {c.size} floats, expected (1 << n_bits) * dim = {(1 << n_bits) * self.dim}")
        enc = (int(m) + 1) // 2 if n_bits == 4 else int(m)
        if cd is not None and cd.size != len(self) * enc:
            raise L.VdbError(f"pq_attach: codes hold {cd.size} bytes, expected len * encoded_dim = {len(self) * enc}")
        L.check(self._lib.vdb_pq_attach(self._h, n_bits, m, _ptr(c, L.f32p), _ptr(cd, L.u8p)))

    def pq_build(self, n_bits: int = 4, m: int | None = None, train_n: int = 0, max_iter: int = 20,
                 tol: float = 1e-6, seed: int = 42):
        m = -(-self.dim // 3) if m is None else m
        L.check(self._lib.vdb_pq_build(self._h, n_bits, m, train_n, max_iter, tol, seed))

    def pq_clear(self):
        L.check(self._lib.vdb_pq_clear(self._h))

    def has_pq(self) -> bool:
        v = C.c_int()
        L.check(self._lib.vdb_pq_has(self._h, C.byref(v)))
        return bool(v.value)

    def pq_export(self):
        nb, m, ed = C.c_uint64(), C.c_uint64(), C.c_uint64()
        L.check(self._lib.vdb_pq_info(self._h, C.byref(nb), C.byref(m), C.byref(ed)))
        cent = np.zeros((1 << nb.value) * self.dim, dtype=np.float32)
        codes = np.zeros((len(self), ed.value), dtype=np.uint8)
        L.check(self._lib.vdb_pq_export(self._h, _ptr(cent, L.f32p), _ptr(codes, L.u8p)))
        return {"n_bits": int(nb.value), "m": int(m.value), "centroids": cent, "codes": codes}

    def pq_create_lookup(self, queries):
        """PQTable::create_lookup (pq_table.rs:195-224): (lut [nq, m * k_c], dist_cache [nq]) as the search kernels build them."""
        q = _f32(queries)
        q = q.reshape(1, -1) if q.ndim == 1 else q
        nb, m, ed = C.c_uint64(), C.c_uint64(), C.c_uint64()
        L.check(self._lib.vdb_pq_info(self._h, C.byref(nb), C.byref(m), C.byref(ed)))
        lut = np.zeros((q.shape[0], m.value << nb.value), dtype=np.float32)
        qc = np.zeros(q.shape[0], dtype=np.float32)
        L.check(self._lib.vdb_pq_create_lookup(self._h, _ptr(q, L.f32p), q.shape[0], q.shape[1], _ptr(lut, L.f32p), _ptr(qc, L.f32p)))
        return lut, qc

    def pq_adc_all(self, queries):
        """ADC distance of every code row to every query (pq_table.rs:239-301): [nq, len] float32."""
        q = _f32(queries)
        q = q.reshape(1, -1) if q.ndim == 1 else q
        out = np.zeros((q.shape[0], len(self)), dtype=np.float32)
        L.check(self._lib.vdb_pq_adc_all(self._h, _ptr(q, L.f32p), q.shape[0], q.shape[1], _ptr(out, L.f32p)))
        return out

    # -- HNSW ------------------------------------------------------------------------------------------------
    def hnsw_build(self, M: int = 16, ef_construction: int = 200, seed: int = 42, batch: int = 1,
                   nthreads: int = 1):
        L.check(self._lib.vdb_hnsw_build(self._h, M, ef_construction, seed, batch, nthreads))

    def hnsw_attach(self, M: int, ef_construction: int, g: dict):
        l0 = np.ascontiguousarray(g["level0"], dtype=np.uint32)
        len0 = np.ascontiguousarray(g["len0"], dtype=np.uint64)
        vl = np.ascontiguousarray(g["vec_level"], dtype=np.uint64)
        up = np.ascontiguousarray(g["upper"], dtype=np.uint32)
        ul = np.ascontiguousarray(g["upper_len"], dtype=np.uint64)
        # the C side reads n * max_m0 / n / n / sum(vec_level) * m / sum(vec_level) elements: check before it does
        n, mm = len(self), min(int(M), 10000)
        tot = int(vl.sum()) if vl.size == n else -1
        if l0.size != n * 2 * mm or len0.size != n or vl.size != n or up.size != tot * mm or ul.size != tot:
            raise L.VdbError(f"hnsw_attach: array sizes do not match len={n}, M={M}: level0 {l0.size}, len0 {len0.size}, "
                             f"vec_level {vl.size}, upper {up.size}, upper_len {ul.size}")
        L.check(self._lib.vdb_hnsw_attach(self._h, M, ef_construction, _ptr(l0, L.u32p), _ptr(len0, L.u64p),
                                          _ptr(vl, L.u64p), _ptr(up, L.u32p), _ptr(ul, L.u64p),
                                          int(g["has_enter"]), int(g["enter_point"]), int(g["enter_level"])))

    def hnsw_clear(self):
        L.check(self._lib.vdb_hnsw_clear(self._h))

    def has_hnsw(self) -> bool:
        v = C.c_int()
        L.check(self._lib.vdb_hnsw_has(self._h, C.byref(v)))
        return bool(v.value)

    def hnsw_export(self) -> dict:
        m, mm0, tot, ep, el, de = (C.c_uint64() for _ in range(6))
        he = C.c_int()
        L.check(self._lib.vdb_hnsw_info(self._h, C.byref(m), C.byref(mm0), C.byref(tot), C.byref(he), C.byref(ep),
                                        C.byref(el), C.byref(de)))
        n = len(self)
        g = {
            "n": n, "m": int(m.value), "max_m0": int(mm0.value),
            "level0": np.zeros(n * mm0.value, dtype=np.uint32), "len0": np.zeros(n, dtype=np.uint64),
            "vec_level": np.zeros(n, dtype=np.uint64), "upper": np.zeros(tot.value * m.value, dtype=np.uint32),
            "upper_len": np.zeros(tot.value, dtype=np.uint64), "has_enter": int(he.value),
            "enter_point": int(ep.value), "enter_level": int(el.value), "default_ef": int(de.value),
        }
        L.check(self._lib.vdb_hnsw_export(self._h, _ptr(g["level0"], L.u32p), _ptr(g["len0"], L.u64p),
                                          _ptr(g["vec_level"], L.u64p), _ptr(g["upper"], L.u32p),
                                          _ptr(g["upper_len"], L.u64p)))
        return g

    def hnsw_last_stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        L.check(self._lib.vdb_hnsw_last_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    # -- IVF (index_algorithm/ivf_index.rs) ---------------------------------------------------------------------
    def ivf_build(self, k: int, train_n: int = 0, max_iter: int = 20, tol: float = 1e-6, seed: int = 42):
        """IVFIndex::from_vec_set with IVFConfig{k, k_means_size, k_means_max_iter, k_means_tol} (ivf_index.rs:19-31)."""
        L.check(self._lib.vdb_ivf_build(self._h, int(k), int(train_n), int(max_iter), float(tol), int(seed)))

    def ivf_attach(self, centroids, assign=None):
        c = _f32(centroids)
        a = None if assign is None else np.ascontiguousarray(assign, dtype=np.uint64)
        if c.ndim != 2 or c.shape[1] != self.dim:
            raise L.VdbError(f"ivf_attach: centroids must be k x dim = ? x {self.dim}, got {c.shape}")
        if a is not None and a.size != len(self):
            raise L.VdbError(f"ivf_attach: assign holds {a.size} entries, expected len = {len(self)}")
        L.check(self._lib.vdb_ivf_attach(self._h, c.shape[0], _ptr(c.ravel(), L.f32p), _ptr(a, L.u64p)))

    def ivf_clear(self):
        L.check(self._lib.vdb_ivf_clear(self._h))

    def ivf_info(self):
        p, k, d = C.c_int(), C.c_uint64(), C.c_uint64()
        L.check(self._lib.vdb_ivf_info(self._h, C.byref(p), C.byref(k), C.byref(d)))
        return {"present": bool(p.value), "k": int(k.value), "default_n_probes": int(d.value)}

    def has_ivf(self) -> bool:
        return self.ivf_info()["present"]

    def ivf_export(self):
        info = self.ivf_info()
        cent = np.zeros((info["k"], self.dim), dtype=np.float32)
        assign = np.zeros(len(self), dtype=np.uint64)
        L.check(self._lib.vdb_ivf_export(self._h, _ptr(cent, L.f32p), _ptr(assign, L.u64p)))
        return {"centroids": cent, "assign": assign}

    def ivf_knn(self, queries, k: int, n_probes: int = 0):
        """IVFIndex::knn_with_ef (ef = n_probes; 0 -> default 4)."""
        return self._search(self._lib.vdb_ivf_knn, queries, k, n_probes)

    def ivf_knn_device(self, q_ptr: int, nq: int, k: int, n_probes: int, out_idx_ptr: int, out_dist_ptr: int,
                       out_cnt_ptr: int, stream: int = 0):
        L.check(self._lib.vdb_ivf_knn_device(self._h, L.vp(q_ptr), int(nq), self.dim, int(k), int(n_probes),
                                             L.vp(out_idx_ptr), L.vp(out_dist_ptr), L.vp(out_cnt_ptr), L.vp(stream)))

    # -- measurement ---------------------------------------------------------------------------------------------
    def prof_enable(self, on: bool = True):
        L.check(self._lib.vdb_prof_enable(self._h, 1 if on else 0))

    def prof_reset(self):
        L.check(self._lib.vdb_prof_reset(self._h))

    def prof_get(self, kernel: str):
        ms, n, by = C.c_double(), C.c_uint64(), C.c_double()
        L.check(self._lib.vdb_prof_get(self._h, kernel.encode(), C.byref(ms), C.byref(n), C.byref(by)))
        return {"ms": float(ms.value), "launches": int(n.value), "bytes": float(by.value)}


def stream_probe(device: int = 0, nbytes: int = 3_840_000_000, iters: int = 5) -> float:
    """Attainable HBM read bandwidth of this box in GB/s: pure streaming read, best of two patterns."""
    v = C.c_double()
    L.check(L.load().vdb_stream_probe(int(device), int(nbytes), int(iters), C.byref(v)))
    return float(v.value)


def stream_probe_rows(device: int = 0, nbytes: int = 1_920_000_000, iters: int = 5, row_bytes: int = 1920) -> float:
    """GB/s of MFMA A-fragment loads (16 rows x 64 B per instruction) from a row-major image of rows of row_bytes (multiple of 128; 1920 = a 960-d fp16 row)."""
    v = C.c_double()
    L.check(L.load().vdb_stream_probe_rows(int(device), int(nbytes), int(iters), int(row_bytes), C.byref(v)))
    return float(v.value)


def mfma_probe(device: int = 0, waves_per_simd: int = 2, iters: int = 200_000, i8: bool = False):
    """(dense fp16 TFLOP/s, shader clock in GHz) of back-to-back v_mfma_f32_16x16x32_f16 on every SIMD of this box;
    i8=True: (dense int8 TOP/s, clock) of v_mfma_i32_16x16x64_i8, the 8-bit Flat filter's instruction."""
    t, c = C.c_double(), C.c_double()
    fn = L.load().vdb_mfma_probe_i8 if i8 else L.load().vdb_mfma_probe
    L.check(fn(int(device), int(waves_per_simd), int(iters), C.byref(t), C.byref(c)))
    return float(t.value), float(c.value)


def latency_probe(device: int = 0, nbytes: int = 1 << 30, hops: int = 20000) -> float:
    """Nanoseconds per DEPENDENT HBM load on this box (one lane chasing a random cycle over 128-B lines of an nbytes buffer)."""
    v = C.c_double()
    L.check(L.load().vdb_latency_probe(int(device), int(nbytes), int(hops), C.byref(v)))
    return float(v.value)


def fold_probe(device: int = 0, adds: int = 1 << 22) -> float:
    """Nanoseconds per DEPENDENT f32 add on this box (the chain a strict left fold is made of)."""
    v = C.c_double()
    L.check(L.load().vdb_fold_probe(int(device), int(adds), C.byref(v)))
    return float(v.value)


def merge_topk(dists: np.ndarray, ids: np.ndarray, counts: np.ndarray, k: int):
    """Merge per-shard sorted lists [S][nq][k] into the global top-k by (distance, index)."""
    d = _f32(dists)
    i = np.ascontiguousarray(ids, dtype=np.uint64)
    c = np.ascontiguousarray(counts, dtype=np.uint64)
    S, nq = d.shape[0], d.shape[1]
    oi = np.zeros((nq, k), dtype=np.uint64)
    od = np.zeros((nq, k), dtype=np.float32)
    oc = np.zeros(nq, dtype=np.uint64)
    L.check(L.load().vdb_merge_topk(_ptr(d, L.f32p), _ptr(i, L.u64p), _ptr(c, L.u64p), S, nq, k, _ptr(oi, L.u64p),
                                    _ptr(od, L.f32p), _ptr(oc, L.u64p)))
    return oi, od, oc


def pq_merge_resort(adc_keys: np.ndarray, exact_keys: np.ndarray, k: int):
    """Merge per-shard ADC shortlists [S][nq][efk] (pair keys, see vdbhip.h) and replay pq_resort."""
    a = np.ascontiguousarray(adc_keys, dtype=np.uint64)
    e = np.ascontiguousarray(exact_keys, dtype=np.uint64)
    S, nq, efk = a.shape
    oi = np.zeros((nq, k), dtype=np.uint64)
    od = np.zeros((nq, k), dtype=np.float32)
    oc = np.zeros(nq, dtype=np.uint64)
    L.check(L.load().vdb_pq_merge_resort(_ptr(a, L.u64p), _ptr(e, L.u64p), S, nq, efk, k, _ptr(oi, L.u64p),
                                         _ptr(od, L.f32p), _ptr(oc, L.u64p)))
    return oi, od, oc
