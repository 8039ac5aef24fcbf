"""ctypes wrapper around oracle/libvdb_oracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never by lab_1806_vec_db_amd (the product).
Parity status and citations: see oracle/vdb_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvdb_oracle.so")

L2SQR, COSINE = 0, 1


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "vdb_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None

_f32p = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    sz = C.c_size_t
    L.orc_dot.restype = C.c_float
    L.orc_dot.argtypes = [_f32p, _f32p, sz]
    L.orc_l2.restype = C.c_float
    L.orc_l2.argtypes = [_f32p, _f32p, sz]
    L.orc_cosine.restype = C.c_float
    L.orc_cosine.argtypes = [_f32p, _f32p, sz]
    L.orc_dist.restype = C.c_float
    L.orc_dist.argtypes = [C.c_int, _f32p, _f32p, sz]
    L.orc_dist_cache.restype = C.c_float
    L.orc_dist_cache.argtypes = [C.c_int, _f32p, sz]
    L.orc_dist_cached.restype = C.c_float
    L.orc_dist_cached.argtypes = [C.c_int, _f32p, _f32p, sz, C.c_float, C.c_float]
    L.orc_dist_u8.restype = C.c_float
    L.orc_dist_u8.argtypes = [C.c_int, _u8p, _u8p, sz]
    L.orc_pair_cmp.restype = C.c_int
    L.orc_pair_cmp.argtypes = [C.c_float, C.c_uint64, C.c_float, C.c_uint64]
    L.orc_recall.restype = C.c_float
    L.orc_recall.argtypes = [_u64p, sz, _u64p, sz]
    L.orc_flat_knn.restype = sz
    L.orc_flat_knn.argtypes = [_f32p, sz, sz, C.c_int, _f32p, sz, _u64p, _f32p]
    L.orc_flat_knn_batch.restype = None
    L.orc_flat_knn_batch.argtypes = [_f32p, sz, sz, C.c_int, _f32p, sz, sz, _u64p, _f32p, _u64p, C.c_int]
    L.orc_pq_groups.restype = sz
    L.orc_pq_groups.argtypes = [sz, sz, _u64p]
    L.orc_pq_new.restype = C.c_void_p
    L.orc_pq_new.argtypes = [sz, sz, sz, C.c_int, _f32p]
    L.orc_pq_free.restype = None
    L.orc_pq_free.argtypes = [C.c_void_p]
    L.orc_pq_encode_row.restype = None
    L.orc_pq_encode_row.argtypes = [C.c_void_p, _f32p, _u8p]
    L.orc_pq_encode_all.restype = None
    L.orc_pq_encode_all.argtypes = [C.c_void_p, _f32p, sz]
    L.orc_pq_set_codes.restype = None
    L.orc_pq_set_codes.argtypes = [C.c_void_p, _u8p, sz]
    L.orc_pq_lookup.restype = C.c_float
    L.orc_pq_lookup.argtypes = [C.c_void_p, _f32p, _f32p]
    L.orc_pq_adc.restype = C.c_float
    L.orc_pq_adc.argtypes = [C.c_void_p, _u8p, _f32p, C.c_float]
    L.orc_pq_adc_all.restype = None
    L.orc_pq_adc_all.argtypes = [C.c_void_p, sz, _f32p, _f32p]
    L.orc_assign_nearest.restype = None
    L.orc_assign_nearest.argtypes = [_f32p, sz, sz, C.c_int, _f32p, sz, _u64p]
    L.orc_ivf_knn.restype = sz
    L.orc_ivf_knn.argtypes = [_f32p, sz, C.c_int, _f32p, sz, _u64p, _u64p, _f32p, sz, sz, _u64p, _f32p]
    L.orc_flat_knn_pq.restype = sz
    L.orc_flat_knn_pq.argtypes = [_f32p, sz, sz, C.c_int, C.c_void_p, _f32p, sz, sz, _u64p, _f32p]
    L.orc_kmeans.restype = None
    L.orc_kmeans.argtypes = [_f32p, sz, sz, sz, sz, sz, sz, C.c_float, C.c_int, _u64p, _f32p]
    L.orc_pq_train.restype = C.c_void_p
    L.orc_pq_train.argtypes = [_f32p, sz, sz, sz, sz, C.c_int, sz, sz, C.c_float, C.c_uint64]
    L.orc_hnsw_new.restype = C.c_void_p
    L.orc_hnsw_new.argtypes = [sz, C.c_int, sz, sz]
    L.orc_hnsw_free.restype = None
    L.orc_hnsw_free.argtypes = [C.c_void_p]
    L.orc_hnsw_add.restype = C.c_uint64
    L.orc_hnsw_add.argtypes = [C.c_void_p, _f32p, C.c_uint64]
    L.orc_hnsw_add_batch.restype = None
    L.orc_hnsw_add_batch.argtypes = [C.c_void_p, _f32p, sz, _u64p]
    L.orc_hnsw_build.restype = C.c_void_p
    L.orc_hnsw_build.argtypes = [_f32p, sz, sz, C.c_int, sz, sz, C.c_uint64, sz]
    L.orc_hnsw_from_graph.restype = C.c_void_p
    L.orc_hnsw_from_graph.argtypes = [_f32p, sz, sz, C.c_int, sz, sz, _u32p, _u64p, _u64p, _u32p, _u64p,
                                      C.c_int, C.c_uint64, C.c_uint64]
    L.orc_hnsw_knn.restype = sz
    L.orc_hnsw_knn.argtypes = [C.c_void_p, _f32p, sz, sz, _u64p, _f32p]
    L.orc_hnsw_knn_pq.restype = sz
    L.orc_hnsw_knn_pq.argtypes = [C.c_void_p, C.c_void_p, _f32p, sz, sz, _u64p, _f32p]
    L.orc_hnsw_knn_batch.restype = None
    L.orc_hnsw_knn_batch.argtypes = [C.c_void_p, _f32p, sz, sz, sz, _u64p, _f32p, _u64p, C.c_int, _u64p, _u64p]
    L.orc_hnsw_level_from_uniform.restype = C.c_uint64
    L.orc_hnsw_level_from_uniform.argtypes = [C.c_void_p, C.c_float]
    _lib = L
    return L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- distances ---------------------------------------------------------
def dot(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().orc_dot(_p(a, _f32p), _p(b, _f32p), min(a.size, b.size)))


def dist(kind, a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().orc_dist(kind, _p(a, _f32p), _p(b, _f32p), min(a.size, b.size)))


def dist_u8(kind, a, b):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    return float(lib().orc_dist_u8(kind, _p(a, _u8p), _p(b, _u8p), min(a.size, b.size)))


def dist_cache(kind, a):
    a = _f32(a)
    return float(lib().orc_dist_cache(kind, _p(a, _f32p), a.size))


def dist_cached(kind, a, b, ca, cb):
    a, b = _f32(a), _f32(b)
    return float(lib().orc_dist_cached(kind, _p(a, _f32p), _p(b, _f32p), a.size, ca, cb))


def recall(gt, pred):
    gt = np.ascontiguousarray(gt, dtype=np.uint64)
    pred = np.ascontiguousarray(pred, dtype=np.uint64)
    return float(lib().orc_recall(_p(gt, _u64p), gt.size, _p(pred, _u64p), pred.size))


# ---- Flat ---------------------------------------------------------------
def flat_knn(base, query, k, kind=L2SQR):
    base, query = _f32(base), _f32(query)
    n, dim = base.shape
    idx = np.zeros(k, dtype=np.uint64)
    d = np.zeros(k, dtype=np.float32)
    c = lib().orc_flat_knn(_p(base, _f32p), n, dim, kind, _p(query, _f32p), k, _p(idx, _u64p), _p(d, _f32p))
    return idx[:c].copy(), d[:c].copy()


def flat_knn_batch(base, queries, k, kind=L2SQR, nthreads=1):
    base, queries = _f32(base), _f32(queries)
    n, dim = base.shape
    nq = queries.shape[0]
    idx = np.zeros((nq, k), dtype=np.uint64)
    d = np.zeros((nq, k), dtype=np.float32)
    cnt = np.zeros(nq, dtype=np.uint64)
    lib().orc_flat_knn_batch(_p(base, _f32p), n, dim, kind, _p(queries, _f32p), nq, k, _p(idx, _u64p),
                             _p(d, _f32p), _p(cnt, _u64p), nthreads)
    return idx, d, cnt


# ---- PQ -----------------------------------------------------------------
def pq_groups(dim, m):
    g = np.zeros(m + 1, dtype=np.uint64)
    cnt = lib().orc_pq_groups(dim, m, _p(g, _u64p))
    return [(int(g[i]), int(g[i + 1])) for i in range(cnt)]


class _PQStruct(C.Structure):
    _fields_ = [("dim", C.c_uint64), ("m", C.c_uint64), ("n_bits", C.c_uint64), ("k", C.c_uint64),
                ("enc_dim", C.c_uint64), ("dist", C.c_int), ("gstart", _u64p), ("centroids", _f32p),
                ("cent_cache", _f32p), ("codes", _u8p), ("n", C.c_uint64)]


class PQ:
    """oracle PQTable (pq_table.rs:116-137)."""

    def __init__(self, handle):
        self.h = handle
        s = C.cast(handle, C.POINTER(_PQStruct)).contents
        self.dim, self.m, self.n_bits, self.k, self.enc_dim, self.dist = (int(s.dim), int(s.m), int(s.n_bits),
                                                                         int(s.k), int(s.enc_dim), int(s.dist))

    @classmethod
    def from_centroids(cls, dim, m, n_bits, kind, centroids):
        c = _f32(centroids).ravel()
        assert c.size == (1 << n_bits) * dim
        return cls(lib().orc_pq_new(dim, m, n_bits, kind, _p(c, _f32p)))

    @classmethod
    def train(cls, base, m, n_bits=4, kind=L2SQR, k_means_size=0, max_iter=20, tol=1e-6, seed=42):
        base = _f32(base)
        n, dim = base.shape
        return cls(lib().orc_pq_train(_p(base, _f32p), n, dim, m, n_bits, kind, k_means_size, max_iter, tol, seed))

    def _s(self):
        return C.cast(self.h, C.POINTER(_PQStruct)).contents

    @property
    def centroids(self):
        return np.ctypeslib.as_array(self._s().centroids, shape=(self.k * self.dim,)).copy()

    @property
    def cent_cache(self):
        return np.ctypeslib.as_array(self._s().cent_cache, shape=(self.m * self.k,)).copy()

    @property
    def codes(self):
        n = int(self._s().n)
        return np.ctypeslib.as_array(self._s().codes, shape=(n, self.enc_dim)).copy()

    def encode_all(self, base):
        base = _f32(base)
        lib().orc_pq_encode_all(self.h, _p(base, _f32p), base.shape[0])

    def set_codes(self, codes):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        lib().orc_pq_set_codes(self.h, _p(codes, _u8p), codes.shape[0])

    def encode_row(self, v):
        v = _f32(v)
        out = np.zeros(self.enc_dim, dtype=np.uint8)
        lib().orc_pq_encode_row(self.h, _p(v, _f32p), _p(out, _u8p))
        return out

    def lookup(self, q):
        q = _f32(q)
        lut = np.zeros(self.m * self.k, dtype=np.float32)
        qc = lib().orc_pq_lookup(self.h, _p(q, _f32p), _p(lut, _f32p))
        return lut, float(qc)

    def adc(self, code, lut, qc):
        code = np.ascontiguousarray(code, dtype=np.uint8)
        return float(lib().orc_pq_adc(self.h, _p(code, _u8p), _p(lut, _f32p), qc))

    def adc_all(self, q, n):
        """ADC distance of each of the n encoded rows (pq_table.rs:239-301), in row order."""
        q = _f32(q)
        out = np.zeros(n, dtype=np.float32)
        lib().orc_pq_adc_all(self.h, n, _p(q, _f32p), _p(out, _f32p))
        return out

    def __del__(self):
        try:
            lib().orc_pq_free(self.h)
        except Exception:
            pass


def flat_knn_pq(base, pq: PQ, query, k, ef, kind=L2SQR):
    base, query = _f32(base), _f32(query)
    n, dim = base.shape
    idx = np.zeros(max(k, 1), dtype=np.uint64)
    d = np.zeros(max(k, 1), dtype=np.float32)
    c = lib().orc_flat_knn_pq(_p(base, _f32p), n, dim, kind, pq.h, _p(query, _f32p), k, ef, _p(idx, _u64p),
                              _p(d, _f32p))
    return idx[:c].copy(), d[:c].copy()


def pair_keys(d, ids):
    """CandidatePair total order (candidate_pair.rs:36-41; ordered-float: NaN greatest, -0 == +0) as one u64 per
    pair: order-preserving u32 image of the f32 distance in the high word, the id in the low word."""
    d = (np.asarray(d, dtype=np.float32) + np.float32(0.0)).astype(np.float32)
    u = d.view(np.uint32).copy()
    u[np.isnan(d)] = 0x7FC00000
    o = np.where(u & 0x80000000, ~u, u | np.uint32(0x80000000)).astype(np.uint64)
    return (o << np.uint64(32)) | np.asarray(ids, dtype=np.uint64)


def kmeans(rows, c0, c1, k, max_iter=20, tol=1e-6, kind=L2SQR, seed=42):
    rows = _f32(rows)
    n, dim = rows.shape
    out = np.zeros((k, c1 - c0), dtype=np.float32)
    st = np.array([seed], dtype=np.uint64)
    lib().orc_kmeans(_p(rows, _f32p), n, dim, c0, c1, k, max_iter, tol, kind, _p(st, _u64p), _p(out, _f32p))
    return out


# ---- IVF ----------------------------------------------------------------
class IVF:
    """oracle IVFIndex (ivf_index.rs:34-47) for given centroids: clusters by find_nearest, search by probes."""

    def __init__(self, base, centroids, kind=L2SQR, assign=None):
        self.base = _f32(base)
        self.cents = _f32(centroids)
        self.kind = kind
        n, dim = self.base.shape
        k = self.cents.shape[0]
        if assign is not None:  # clusters supplied (a serialized IVFIndex carries them, ivf_index.rs:44)
            self.assign = np.ascontiguousarray(assign, dtype=np.uint64)
        else:
            self.assign = np.zeros(n, dtype=np.uint64)
            lib().orc_assign_nearest(_p(self.base, _f32p), n, dim, kind, _p(self.cents, _f32p), k, _p(self.assign, _u64p))
        order = np.argsort(self.assign, kind="stable")  # ascending id inside a cluster (ivf_index.rs:98-100)
        self.members = order.astype(np.uint64)
        counts = np.bincount(self.assign.astype(np.int64), minlength=k)
        self.offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)

    def knn(self, query, k, n_probes=4):
        query = _f32(query)
        idx = np.zeros(max(k, 1), dtype=np.uint64)
        d = np.zeros(max(k, 1), dtype=np.float32)
        c = lib().orc_ivf_knn(_p(self.base, _f32p), self.base.shape[1], self.kind, _p(self.cents, _f32p),
                              self.cents.shape[0], _p(self.offsets, _u64p), _p(self.members, _u64p),
                              _p(query, _f32p), k, n_probes, _p(idx, _u64p), _p(d, _f32p))
        return idx[:c].copy(), d[:c].copy()


# ---- HNSW ---------------------------------------------------------------
class _HNSWStruct(C.Structure):
    _fields_ = [("dim", C.c_uint64), ("m", C.c_uint64), ("max_m0", C.c_uint64), ("ef_construction", C.c_uint64),
                ("default_ef", C.c_uint64), ("inv_log_m", C.c_float), ("dist", C.c_int), ("n", C.c_uint64),
                ("cap", C.c_uint64), ("rows", _f32p), ("cache", _f32p), ("level0", _u32p), ("len0", _u64p),
                ("vec_level", _u64p), ("upper_off", _u64p), ("upper", _u32p), ("upper_len", _u64p),
                ("upper_cap", C.c_uint64), ("upper_total", C.c_uint64), ("has_enter", C.c_int),
                ("enter_point", C.c_uint64), ("enter_level", C.c_uint64), ("stat_n_dist", C.c_uint64),
                ("stat_n_expanded", C.c_uint64)]


class HNSW:
    """oracle HNSWIndex (hnsw_index.rs:98-141)."""

    def __init__(self, handle):
        self.h = handle

    def _s(self):
        return C.cast(self.h, C.POINTER(_HNSWStruct)).contents

    @classmethod
    def new(cls, dim, kind=L2SQR, M=16, ef_construction=200):
        return cls(lib().orc_hnsw_new(dim, kind, M, ef_construction))

    @classmethod
    def build(cls, base, kind=L2SQR, M=16, ef_construction=200, seed=42, batch=1):
        base = _f32(base)
        n, dim = base.shape
        return cls(lib().orc_hnsw_build(_p(base, _f32p), n, dim, kind, M, ef_construction, seed, batch))

    @classmethod
    def from_graph(cls, base, kind, M, ef_construction, g):
        base = _f32(base)
        n, dim = base.shape
        l0 = np.ascontiguousarray(g["level0"], dtype=np.uint32)
        len0 = np.ascontiguousarray(g["len0"], dtype=np.uint64)
        vl = np.ascontiguousarray(g["vec_level"], dtype=np.uint64)
        up = np.ascontiguousarray(g["upper"], dtype=np.uint32)
        ul = np.ascontiguousarray(g["upper_len"], dtype=np.uint64)
        return cls(lib().orc_hnsw_from_graph(_p(base, _f32p), n, dim, kind, M, ef_construction, _p(l0, _u32p),
                                             _p(len0, _u64p), _p(vl, _u64p), _p(up, _u32p), _p(ul, _u64p),
                                             int(g["has_enter"]), int(g["enter_point"]), int(g["enter_level"])))

    def add(self, vec, level):
        vec = _f32(vec)
        return int(lib().orc_hnsw_add(self.h, _p(vec, _f32p), level))

    def add_batch(self, vecs, levels):
        vecs = _f32(vecs)
        levels = np.ascontiguousarray(levels, dtype=np.uint64)
        lib().orc_hnsw_add_batch(self.h, _p(vecs, _f32p), vecs.shape[0], _p(levels, _u64p))

    def level_from_uniform(self, u):
        return int(lib().orc_hnsw_level_from_uniform(self.h, u))

    def graph(self):
        s = self._s()
        n, m, mm0 = int(s.n), int(s.m), int(s.max_m0)
        tot = int(s.upper_total)
        return {
            "n": n, "m": m, "max_m0": mm0,
            "level0": np.ctypeslib.as_array(s.level0, shape=(n * mm0,)).copy() if n else np.zeros(0, np.uint32),
            "len0": np.ctypeslib.as_array(s.len0, shape=(n,)).copy() if n else np.zeros(0, np.uint64),
            "vec_level": np.ctypeslib.as_array(s.vec_level, shape=(n,)).copy() if n else np.zeros(0, np.uint64),
            "upper": np.ctypeslib.as_array(s.upper, shape=(tot * m,)).copy() if tot else np.zeros(0, np.uint32),
            "upper_len": np.ctypeslib.as_array(s.upper_len, shape=(tot,)).copy() if tot else np.zeros(0, np.uint64),
            "has_enter": int(s.has_enter), "enter_point": int(s.enter_point), "enter_level": int(s.enter_level),
        }

    @property
    def default_ef(self):
        return int(self._s().default_ef)

    def knn(self, query, k, ef=None):
        query = _f32(query)
        ef = self.default_ef if ef is None else ef
        idx = np.zeros(max(k, 1), dtype=np.uint64)
        d = np.zeros(max(k, 1), dtype=np.float32)
        c = lib().orc_hnsw_knn(self.h, _p(query, _f32p), k, ef, _p(idx, _u64p), _p(d, _f32p))
        return idx[:c].copy(), d[:c].copy()

    def knn_pq(self, pq: PQ, query, k, ef):
        query = _f32(query)
        idx = np.zeros(max(k, 1), dtype=np.uint64)
        d = np.zeros(max(k, 1), dtype=np.float32)
        c = lib().orc_hnsw_knn_pq(self.h, pq.h, _p(query, _f32p), k, ef, _p(idx, _u64p), _p(d, _f32p))
        return idx[:c].copy(), d[:c].copy()

    def knn_batch(self, queries, k, ef, nthreads=1):
        queries = _f32(queries)
        nq = queries.shape[0]
        idx = np.zeros((nq, k), dtype=np.uint64)
        d = np.zeros((nq, k), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint64)
        nd = np.zeros(1, dtype=np.uint64)
        ne = np.zeros(1, dtype=np.uint64)
        lib().orc_hnsw_knn_batch(self.h, _p(queries, _f32p), nq, k, ef, _p(idx, _u64p), _p(d, _f32p),
                                 _p(cnt, _u64p), nthreads, _p(nd, _u64p), _p(ne, _u64p))
        return idx, d, cnt, int(nd[0]), int(ne[0])

    def __del__(self):
        try:
            lib().orc_hnsw_free(self.h)
        except Exception:
            pass
