"""Reader / writer for the reference's bincode 1.3.3 files (SURVEY.md Appendix B, row (f-2) of section 8).

bincode default options: little-endian, fixed width; usize/u64 = 8 B, u32 = 4 B, f32 = 4 B, bool / Option tag = 1 B;
Vec / String / BTreeMap = u64 count + elements; enum = u32 variant index + payload; struct = fields in declaration
order; Box is transparent; #[serde(skip)] fields are absent.  Field orders are taken from the reference sources:

  VecSet<T>           vec_set.rs:15-20          {dim, data: Vec<T>}
  DistanceAlgorithm   distance/mod.rs:17-28     u32: 0 L2Sqr, 1 Cosine
  KMeansConfig/KMeans k_means.rs:15-37          {k, max_iter, tol, dist, selected: Option<Range>} / {config, centroids}
  PQConfig/PQTable    pq_table.rs:19-34,116-137
  HNSWInnerConfig     hnsw_index.rs:75-96
  HNSWIndex<T>        hnsw_index.rs:99-141      (dist_cache skipped; "without vec_set" = same with empty data :645-656)
  FlatIndex<T>        flat_index.rs:18-23
  DynamicIndex        dynamic_index.rs:11-14    u32: 0 Flat, 1 HNSW
  MetadataVecTable    metadata_vec_table.rs:14-20 (rng skipped) -- the per-table *.db file
  GroundTruth         candidate_pair.rs:111-149

This lets reference-built PQ tables, HNSW graphs, ground truth and whole tables be attached to a GpuIndex
(`GpuIndex.pq_attach`, `GpuIndex.hnsw_attach`) instead of being rebuilt, which is how RNG-dependent artefacts of
the reference (parity unpinned otherwise) can be consumed when someone supplies them.  Host-side I/O only.
"""
from __future__ import annotations

import struct

import numpy as np

L2SQR, COSINE = 0, 1


class _R:
    def __init__(self, buf: bytes):
        self.b = memoryview(buf)
        self.o = 0

    def _take(self, n):
        if self.o + n > len(self.b):
            raise ValueError("bincode: unexpected end of data")
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def u8(self):
        return self._take(1)[0]

    def u32(self):
        return struct.unpack("<I", self._take(4))[0]

    def u64(self):
        return struct.unpack("<Q", self._take(8))[0]

    def f32(self):
        return struct.unpack("<f", self._take(4))[0]

    def arr(self, dtype, count=None):
        n = self.u64() if count is None else count
        dt = np.dtype(dtype).newbyteorder("<")
        return np.frombuffer(self._take(n * dt.itemsize), dtype=dt).copy()

    def opt(self, fn):
        tag = self.u8()
        if tag == 0:
            return None
        if tag != 1:
            raise ValueError("bincode: bad Option tag")
        return fn()

    def string(self):
        n = self.u64()
        return bytes(self._take(n)).decode("utf-8")

    def done(self):
        return self.o == len(self.b)


class _W:
    def __init__(self):
        self.parts = []

    def u8(self, v):
        self.parts.append(struct.pack("<B", v))

    def u32(self, v):
        self.parts.append(struct.pack("<I", v))

    def u64(self, v):
        self.parts.append(struct.pack("<Q", int(v)))

    def f32(self, v):
        self.parts.append(struct.pack("<f", v))

    def arr(self, a, dtype, with_len=True):
        a = np.ascontiguousarray(a, dtype=np.dtype(dtype).newbyteorder("<"))
        if with_len:
            self.u64(a.size)
        self.parts.append(a.tobytes())

    def opt(self, v, fn):
        if v is None:
            self.u8(0)
        else:
            self.u8(1)
            fn(v)

    def string(self, s):
        b = s.encode("utf-8")
        self.u64(len(b))
        self.parts.append(b)

    def bytes(self):
        return b"".join(self.parts)


# ---- VecSet ------------------------------------------------------------------------------------------------
def _read_vec_set(r: _R, dtype):
    dim = r.u64()
    data = r.arr(dtype)
    if dim == 0 or data.size % dim:
        raise ValueError("bincode: VecSet data length is not a multiple of dim")
    return data.reshape(-1, dim)


def _write_vec_set(w: _W, rows, dtype, dim=None):
    rows = np.asarray(rows)
    w.u64(rows.shape[1] if dim is None else dim)
    w.arr(rows.reshape(-1), dtype)


def read_raw_vectors(path, dim, dtype=np.float32, limit=None):
    """Headerless row-major file (scalar.rs:89-105, vec_set.rs:168-181), e.g. data/gist_1000.bin."""
    a = np.fromfile(path, dtype=np.dtype(dtype).newbyteorder("<"), count=-1 if limit is None else limit * dim)
    return a.reshape(-1, dim)


# ---- GroundTruth --------------------------------------------------------------------------------------------
def loads_ground_truth(buf) -> list[np.ndarray]:
    r = _R(buf)
    rows = [r.arr(np.uint64) for _ in range(r.u64())]
    if not r.done():
        raise ValueError("bincode: trailing bytes")
    return rows


def dumps_ground_truth(rows) -> bytes:
    w = _W()
    w.u64(len(rows))
    for row in rows:
        w.arr(row, np.uint64)
    return w.bytes()


# ---- PQTable<f32> ---------------------------------------------------------------------------------------------
def _read_kmeans(r: _R):
    cfg = {"k": r.u64(), "max_iter": r.u64(), "tol": r.f32(), "dist": r.u32(),
           "selected": r.opt(lambda: (r.u64(), r.u64()))}
    return cfg, _read_vec_set(r, np.float32)


def _read_pq(r: _R) -> dict:
    cfg = {"n_bits": r.u64(), "m": r.u64(), "dist": r.u32(), "k_means_size": r.opt(r.u64),
           "k_means_max_iter": r.u64(), "k_means_tol": r.f32()}
    dim, k, enc_dim = r.u64(), r.u64(), r.u64()
    codes = _read_vec_set(r, np.uint8) if True else None
    groups = [_read_kmeans(r) for _ in range(r.u64())]
    dist_cache = r.arr(np.float32)
    if len(groups) != cfg["m"] or k != 1 << cfg["n_bits"] or codes.shape[1] != enc_dim:
        raise ValueError("bincode: inconsistent PQTable")
    # C-ABI centroid layout: group g at k*gstart[g], centroid c at + c*len(g) == concatenation in group order
    centroids = np.concatenate([c.reshape(-1) for _, c in groups]).astype(np.float32)
    if centroids.size != k * dim:
        raise ValueError("bincode: PQ centroids do not cover dim")
    return {"config": cfg, "dim": dim, "k": k, "encoded_dim": enc_dim, "n_bits": cfg["n_bits"], "m": cfg["m"],
            "dist": cfg["dist"], "codes": codes, "centroids": centroids, "dist_cache": dist_cache,
            "selected": [g[0]["selected"] for g in groups]}


def loads_pq_table(buf) -> dict:
    r = _R(buf)
    pq = _read_pq(r)
    if not r.done():
        raise ValueError("bincode: trailing bytes")
    return pq


def _pq_groups(dim, m):
    g, cur = [0], 0
    while cur < dim:
        rem = m - (len(g) - 1)
        cur += -(-(dim - cur) // rem)
        g.append(cur)
    return g


def _write_pq(w: _W, dim, n_bits, m, dist, centroids, codes, k_means_size=None, max_iter=20, tol=1e-6):
    k = 1 << n_bits
    gs = _pq_groups(dim, m)
    centroids = np.asarray(centroids, np.float32).reshape(-1)
    codes = np.asarray(codes, np.uint8)
    w.u64(n_bits); w.u64(m); w.u32(dist); w.opt(k_means_size, w.u64); w.u64(max_iter); w.f32(tol)
    w.u64(dim); w.u64(k); w.u64(codes.shape[1])
    _write_vec_set(w, codes, np.uint8)
    w.u64(m)
    cache = []
    for g in range(m):
        gd = gs[g + 1] - gs[g]
        w.u64(k); w.u64(max_iter); w.f32(tol); w.u32(dist); w.opt((gs[g], gs[g + 1]), lambda t: (w.u64(t[0]), w.u64(t[1])))
        c = centroids[k * gs[g]:k * gs[g + 1]].reshape(k, gd)
        _write_vec_set(w, c, np.float32)
        for row in c:  # dot(c,c) in reference order for Cosine, 0 for L2Sqr (pq_table.rs:160-165)
            acc = np.float32(0)
            if dist == COSINE:
                for v in row:
                    acc = np.float32(acc + np.float32(v * v))
            cache.append(acc)
    w.arr(np.array(cache, np.float32), np.float32)


def dumps_pq_table(dim, n_bits, m, dist, centroids, codes, **kw) -> bytes:
    w = _W()
    _write_pq(w, dim, n_bits, m, dist, centroids, codes, **kw)
    return w.bytes()


# ---- HNSWIndex<f32> --------------------------------------------------------------------------------------------
def _read_hnsw(r: _R) -> dict:
    cfg = {"dim": r.u64(), "dist": r.u32(), "max_elements": r.u64(), "m": r.u64(), "max_m0": r.u64(),
           "ef_construction": r.u64(), "default_ef": r.u64(), "inv_log_m": r.f32(), "start_batch_since": r.u64()}
    dim = r.u64()
    data = r.arr(np.float32)
    rows = data.reshape(-1, dim) if dim else data.reshape(0, 0)
    level0 = r.arr(np.uint32)
    other = [r.arr(np.uint32) for _ in range(r.u64())]
    links_len = [r.arr(np.uint64) for _ in range(r.u64())]
    vec_level = r.arr(np.uint64)
    num_deleted = r.u64()
    enter_level = r.opt(r.u64)
    enter_point = r.opt(r.u64)
    n = len(vec_level)
    m, mm0 = cfg["m"], cfg["max_m0"]
    if level0.size != n * mm0 or len(other) != n or len(links_len) != n:
        raise ValueError("bincode: inconsistent HNSWIndex")
    for v in range(n):
        if other[v].size != vec_level[v] * m or links_len[v].size != vec_level[v] + 1:
            raise ValueError("bincode: inconsistent HNSW node")
    graph = {
        "n": n, "m": m, "max_m0": mm0, "level0": level0,
        "len0": np.array([ll[0] for ll in links_len], dtype=np.uint64),
        "vec_level": vec_level.astype(np.uint64),
        "upper": np.concatenate(other) if n and sum(o.size for o in other) else np.zeros(0, np.uint32),
        "upper_len": (np.concatenate([ll[1:] for ll in links_len]) if n else np.zeros(0, np.uint64)).astype(np.uint64),
        "has_enter": int(enter_point is not None and enter_level is not None),
        "enter_point": enter_point or 0, "enter_level": enter_level or 0,
    }
    return {"config": cfg, "rows": rows, "graph": graph, "num_deleted": num_deleted}


def loads_hnsw_index(buf) -> dict:
    r = _R(buf)
    h = _read_hnsw(r)
    if not r.done():
        raise ValueError("bincode: trailing bytes")
    return h


def _write_hnsw(w: _W, dim, dist, rows, g, ef_construction, default_ef=None, max_elements=None):
    m, mm0, n = int(g["m"]), int(g["max_m0"]), int(g["n"])
    w.u64(dim); w.u32(dist); w.u64(n if max_elements is None else max_elements); w.u64(m); w.u64(mm0)
    w.u64(ef_construction); w.u64(ef_construction // 2 if default_ef is None else default_ef)
    w.f32(np.float32(1.0) / np.log(np.float32(m))); w.u64(1000)
    rows = np.zeros((0, dim), np.float32) if rows is None else np.asarray(rows, np.float32)
    _write_vec_set(w, rows, np.float32, dim=dim)
    w.arr(g["level0"], np.uint32)
    vl = np.asarray(g["vec_level"], np.uint64)
    off = np.concatenate([[0], np.cumsum(vl)]).astype(np.int64)
    up, ul = np.asarray(g["upper"], np.uint32), np.asarray(g["upper_len"], np.uint64)
    w.u64(n)
    for v in range(n):
        w.arr(up[off[v] * m:off[v + 1] * m], np.uint32)
    w.u64(n)
    for v in range(n):
        w.arr(np.concatenate([[g["len0"][v]], ul[off[v]:off[v + 1]]]), np.uint64)
    w.arr(vl, np.uint64)
    w.u64(0)
    w.opt(int(g["enter_level"]) if g["has_enter"] else None, w.u64)
    w.opt(int(g["enter_point"]) if g["has_enter"] else None, w.u64)


def dumps_hnsw_index(dim, dist, rows, graph, ef_construction, **kw) -> bytes:
    w = _W()
    _write_hnsw(w, dim, dist, rows, graph, ef_construction, **kw)
    return w.bytes()


# ---- MetadataVecTable (*.db) ------------------------------------------------------------------------------------
def loads_table(buf) -> dict:
    r = _R(buf)
    metadata = []
    for _ in range(r.u64()):
        metadata.append({r.string(): r.string() for _ in range(r.u64())})
    variant = r.u32()
    if variant == 0:  # DynamicIndex::Flat(FlatIndex{dist, vec_set})
        dist = r.u32()
        rows = _read_vec_set(r, np.float32) if True else None
        inner = {"kind": "flat", "dist": dist, "rows": rows}
    elif variant == 1:
        h = _read_hnsw(r)
        inner = {"kind": "hnsw", "dist": h["config"]["dist"], "rows": h["rows"], "hnsw": h}
    else:
        raise ValueError("bincode: bad DynamicIndex variant")
    pq = r.opt(lambda: _read_pq(r))
    if not r.done():
        raise ValueError("bincode: trailing bytes")
    return {"metadata": metadata, "inner": inner, "pq_table": pq}


def dumps_table(metadata, dist, rows, hnsw_graph=None, ef_construction=200, pq=None) -> bytes:
    w = _W()
    w.u64(len(metadata))
    for m in metadata:
        w.u64(len(m))
        for k in sorted(m):  # BTreeMap iterates in key order
            w.string(k); w.string(m[k])
    rows = np.asarray(rows, np.float32)
    dim = rows.shape[1]
    if hnsw_graph is None:
        w.u32(0); w.u32(dist); _write_vec_set(w, rows, np.float32)
    else:
        w.u32(1); _write_hnsw(w, dim, dist, rows, hnsw_graph, ef_construction)
    w.opt(pq, lambda p: _write_pq(w, dim, p["n_bits"], p["m"], dist, p["centroids"], p["codes"]))
    return w.bytes()


def load(path, kind):
    buf = open(path, "rb").read()
    return {"ground_truth": loads_ground_truth, "pq_table": loads_pq_table, "hnsw_index": loads_hnsw_index,
            "table": loads_table}[kind](buf)
