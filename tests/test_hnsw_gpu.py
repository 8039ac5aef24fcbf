"""GPU parity: HNSW search kernel (knn_with_ef, knn_pq) and the host graph builder vs the CPU oracle.

The graph is RNG- and thread-count-dependent in the reference (parity unpinned), so it is an INPUT to
search parity: built by the library's host builder, exported, and attached to the oracle.  Given the
same graph the walk is deterministic and must match exactly: indices identical, distances bit-exact,
and even the work counters (distance evaluations, expansions) equal.
"""
import numpy as np
import pytest

from conftest import gist_like

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import lab_1806_vec_db_amd as vdb
    from oracle import oracle as O
    return vdb, O


def _graphs_equal(a, b):
    for key in ("level0", "len0", "vec_level", "upper", "upper_len"):
        if not np.array_equal(np.asarray(a[key]), np.asarray(b[key])):
            return False, key
    for key in ("has_enter", "enter_point", "enter_level"):
        if int(a[key]) != int(b[key]):
            return False, key
    return True, ""


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
def test_hnsw_index_test_restated(mods, gist_base, dist, kind):
    """hnsw_index.rs:713-790: default HNSW on gist_1000 x 12 dims, k=6, query = row 200 -> Flat's indices."""
    vdb, O = mods
    b12 = np.ascontiguousarray(gist_base[:, :12])
    ix = vdb.GpuIndex(12, dist)
    ix.batch_add(b12)
    flat_i, _ = ix.flat_knn(b12[200], 6)
    ix.hnsw_build(M=16, ef_construction=200, seed=42)
    assert ix.has_hnsw()
    hi, hd = ix.knn(b12[200], 6)  # IndexKNN::knn -> default ef = ef_construction/2
    assert hi.tolist() == flat_i.tolist()
    assert all(hd[i] <= hd[i + 1] for i in range(5))
    # and the oracle on the same graph agrees bit for bit
    oh = O.HNSW.from_graph(b12, kind, 16, 200, ix.hnsw_export())
    oi, od = oh.knn(b12[200], 6)
    assert hi.tolist() == oi.tolist() and np.array_equal(hd, od)


@pytest.mark.parametrize("batch,nthreads", [(1, 1), (6, 4)])
def test_builder_matches_oracle_builder(mods, gist_base, batch, nthreads):
    """Same level stream + same batch size -> the host builder and the oracle's builder give the same graph."""
    vdb, O = mods
    base = np.ascontiguousarray(gist_base[:, :48]) if batch == 1 else np.concatenate(
        [gist_base[:, :32], gist_base[:, 32:64], gist_base[:, 64:96]])  # 3000 rows so batching kicks in
    ix = vdb.GpuIndex(base.shape[1], "l2sqr")
    ix.batch_add(base)
    ix.hnsw_build(M=8, ef_construction=40, seed=7, batch=batch, nthreads=nthreads)
    oh = O.HNSW.build(base, 0, M=8, ef_construction=40, seed=7, batch=batch)
    ok, key = _graphs_equal(ix.hnsw_export(), oh.graph())
    assert ok, f"graphs differ in {key}"


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
def test_gist1000_knn_with_ef(mods, gist_base, gist_test, dist, kind):
    vdb, O = mods
    ix = vdb.GpuIndex(960, dist)
    ix.batch_add(gist_base)
    ix.hnsw_build(M=16, ef_construction=100, seed=42)
    oh = O.HNSW.from_graph(gist_base, kind, 16, 100, ix.hnsw_export())
    nq = 50
    for k, ef in ((10, 10), (10, 64), (10, 128), (1, 200), (100, 30)):
        idx, d, cnt = ix.knn_with_ef(gist_test[:nq], k, ef)
        oi, od, oc, nd, ne = oh.knn_batch(gist_test[:nq], k, ef)
        assert cnt.tolist() == oc.tolist()
        for q in range(nq):
            c = int(cnt[q])
            assert idx[q, :c].tolist() == oi[q, :c].tolist(), (k, ef, q)
            assert np.array_equal(d[q, :c], od[q, :c]), (k, ef, q)
        assert ix.hnsw_last_stats() == (nd, ne), "distance-evaluation / expansion counts differ from the oracle"


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
def test_gist1000_hnsw_knn_pq(mods, gist_base, gist_test, dist, kind):
    vdb, O = mods
    ix = vdb.GpuIndex(960, dist)
    ix.batch_add(gist_base)
    ix.hnsw_build(M=16, ef_construction=100, seed=1)
    ix.pq_build(n_bits=4, m=320, train_n=300, max_iter=5, seed=2)
    pq = ix.pq_export()
    opq = O.PQ.from_centroids(960, 320, 4, kind, pq["centroids"])
    opq.set_codes(pq["codes"])
    oh = O.HNSW.from_graph(gist_base, kind, 16, 100, ix.hnsw_export())
    for k, ef in ((10, 40), (10, 180), (5, 5)):
        idx, d, cnt = ix.knn_pq(gist_test[:20], k, ef)
        for q in range(20):
            oi, od = oh.knn_pq(opq, gist_test[q], k, ef)
            c = int(cnt[q])
            assert c == len(oi)
            assert idx[q, :c].tolist() == oi.tolist(), (k, ef, q)
            assert np.array_equal(d[q, :c], od), (k, ef, q)


def test_duplicates_and_ties(mods):
    """Many exactly equal distances: exercises check_candidate's (distance, index) tie rule and
    ResultSet::add's distance-only rule in the walk."""
    vdb, O = mods
    rng = np.random.default_rng(9)
    uniq = rng.standard_normal((60, 16)).astype(np.float32)
    base = np.concatenate([uniq] * 20)[rng.permutation(1200)]
    ix = vdb.GpuIndex(16, "l2sqr")
    ix.batch_add(base)
    ix.hnsw_build(M=6, ef_construction=30, seed=3)
    oh = O.HNSW.from_graph(base, 0, 6, 30, ix.hnsw_export())
    qs = uniq[:12] + 0.01
    for k, ef in ((5, 8), (10, 25), (30, 60)):
        idx, d, cnt = ix.knn_with_ef(qs, k, ef)
        for q in range(qs.shape[0]):
            oi, od = oh.knn(qs[q], k, ef)
            c = int(cnt[q])
            assert idx[q, :c].tolist() == oi.tolist(), (k, ef, q)
            assert np.array_equal(d[q, :c], od)


@pytest.mark.parametrize("M,dim,dma", [(24, 64, 1), (16, 50, 1), (16, 64, 0), (8, 960, 1)])
def test_walk_variants_agree_with_oracle(mods, M, dim, dma):
    """The level-0 distance evaluation has two forms: rows staged through LDS by DMA (max_m0 <= 32 and dim % 32 == 0)
    and one register-fed fold per lane (wide graphs, odd dims, or hnsw_dma = 0).  Both must reproduce the oracle's
    results and work counters."""
    vdb, O = mods
    rng = np.random.default_rng(M * 1000 + dim)
    base = rng.standard_normal((2500, dim)).astype(np.float32)
    qs = rng.standard_normal((20, dim)).astype(np.float32)
    for dist, kind in (("l2sqr", 0), ("cosine", 1)):
        ix = vdb.GpuIndex(dim, dist)
        ix.batch_add(base)
        ix.hnsw_build(M=M, ef_construction=50, seed=11, batch=16, nthreads=4)
        oh = O.HNSW.from_graph(base, kind, M, 50, ix.hnsw_export())
        ix.set_param("hnsw_dma", dma)
        try:
            idx, d, cnt = ix.knn_with_ef(qs, 10, 64)
        finally:
            ix.set_param("hnsw_dma", 1)
        for q in range(qs.shape[0]):
            oi, od = oh.knn(qs[q], 10, 64)
            c = int(cnt[q])
            assert idx[q, :c].tolist() == oi.tolist() and np.array_equal(d[q, :c], od), (dist, q)


def test_add_after_build_keeps_graph_valid(mods, gist_base):
    """DynamicIndex::add on the HNSW arm (dynamic_index.rs:47-52): HNSWIndex::add per new row."""
    vdb, O = mods
    b = np.ascontiguousarray(gist_base[:, :24])
    ix = vdb.GpuIndex(24, "l2sqr")
    ix.batch_add(b[:300])
    ix.hnsw_build(M=8, ef_construction=40, seed=5)
    g0 = ix.hnsw_export()
    oh = O.HNSW.from_graph(b[:300], 0, 8, 40, g0)
    ix.batch_add(b[300:340])
    assert ix.has_hnsw() and len(ix) == 340
    g1 = ix.hnsw_export()
    for i in range(300, 340):  # replay the same inserts in the oracle with the levels the library drew
        oh.add(b[i], int(g1["vec_level"][i]))
    ok, key = _graphs_equal(g1, oh.graph())
    assert ok, f"graphs differ in {key}"
    idx, d = ix.knn_with_ef(b[320], 5, 20)
    oi, od = oh.knn(b[320], 5, 20)
    assert idx.tolist() == oi.tolist() and np.array_equal(d, od)
    assert idx[0] == 320


def test_gistlike_batch_built_graph(mods):
    vdb, O = mods
    base = gist_like(6000, dim=96, seed=1806)
    qs = gist_like(64, dim=96, seed=1807)
    ix = vdb.GpuIndex(96, "l2sqr")
    ix.batch_add(base)
    ix.hnsw_build(M=16, ef_construction=60, seed=42, batch=32, nthreads=8)
    oh = O.HNSW.from_graph(base, 0, 16, 60, ix.hnsw_export())
    idx, d, cnt = ix.knn_with_ef(qs, 10, 128)
    oi, od, oc, nd, ne = oh.knn_batch(qs, 10, 128, nthreads=8)
    assert np.array_equal(idx.astype(np.uint64), oi) and np.array_equal(d, od)
    assert ix.hnsw_last_stats() == (nd, ne)
    # recall vs exact Flat on the same data (GroundTruthRow::recall, candidate_pair.rs:127-140)
    ix.hnsw_clear()
    fi, _, _ = ix.flat_knn(qs, 10)
    rec = np.mean([O.recall(fi[q], idx[q]) for q in range(qs.shape[0])])
    assert rec > 0.9


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
def test_gpu_assisted_builder_matches_oracle_builder(mods, dist, kind):
    """Batches of >= 256 points run their candidate phase on the GPU (the level-0 search of every batch member against the
    pre-batch graph = k_hnsw_search with k = ef = ef_construction over the device mirror of the graph; distances between
    batch members = one all-pairs launch).  The graph must equal the oracle's all-host builder with the same level stream
    and batch size, list by list -- and the all-host run of this library's builder (hnsw_build_gpu = 1)."""
    vdb, O = mods
    from conftest import gist_like
    base = gist_like(6000, dim=64, seed=77)
    base[3000:3010] = base[:10]  # exact duplicates: distance ties inside the candidate sets
    ix = vdb.GpuIndex(64, dist)
    ix.batch_add(base)
    ix.hnsw_build(M=8, ef_construction=48, seed=11, batch=300, nthreads=8)   # batches reach 300 once 2 400 rows are in
    g_gpu = ix.hnsw_export()
    oh = O.HNSW.build(base, kind, M=8, ef_construction=48, seed=11, batch=300)
    ok, key = _graphs_equal(g_gpu, oh.graph())
    assert ok, f"GPU-assisted graph differs from the oracle's in {key}"
    try:
        ix.set_param("hnsw_build_gpu", 1)
        ix.hnsw_build(M=8, ef_construction=48, seed=11, batch=300, nthreads=8)
    finally:
        ix.set_param("hnsw_build_gpu", 0)
    ok, key = _graphs_equal(g_gpu, ix.hnsw_export())
    assert ok, f"GPU-assisted graph differs from the all-host graph in {key}"
    # and the graph is searchable
    qs = gist_like(8, dim=64, seed=78)
    idx, d, cnt = ix.knn_with_ef(qs, 5, 40)
    for q in range(8):
        oi, od = oh.knn(qs[q], 5, 40)
        assert idx[q, :len(oi)].tolist() == oi.tolist() and np.array_equal(d[q, :len(od)], od)
