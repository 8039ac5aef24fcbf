"""GPU parity beyond the sizes the register / LDS resident structures hold: the reference has no limit on k, ef or
n_probes (BTreeSet-backed ResultSet, candidate_pair.rs:43-82; HNSWIndex::knn_with_ef, hnsw_index.rs:619-634;
MetadataVecTable::search is called with k = ef = len in the reference's own test, database/mod.rs:551-607).  The library
answers such calls through heap- and sort-based paths; results must still equal the oracle's bit for bit."""
import numpy as np
import pytest

from conftest import gist_like

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import lab_1806_vec_db_amd as vdb
    from oracle import oracle as O
    return vdb, O


def _check(ix_fn, o_fn, qs, k):
    idx, d, cnt = ix_fn(qs)
    for q in range(qs.shape[0]):
        oi, od = o_fn(qs[q])
        c = int(cnt[q])
        assert c == len(oi), (q, c, len(oi))
        assert idx[q, :c].tolist() == oi.tolist(), q
        assert np.array_equal(d[q, :c], od), q


@pytest.fixture(scope="module")
def graph5000(mods):
    vdb, O = mods
    base = gist_like(5000, dim=48, seed=41)
    base[2500:2600] = base[:100]  # exact duplicates: distance ties inside large result sets
    ix = vdb.GpuIndex(48, "l2sqr")
    ix.batch_add(base)
    ix.hnsw_build(M=8, ef_construction=60, seed=4, batch=8, nthreads=8)
    oh = O.HNSW.from_graph(base, 0, 8, 60, ix.hnsw_export())
    return base, ix, oh


@pytest.mark.parametrize("k,ef", [(10, 2000), (1500, 100), (5000, 5000), (7000, 3), (64, 1025)])
def test_hnsw_large_ef_and_k(mods, graph5000, k, ef):
    """ef = 2000 and k = len on a 5 000-row graph (and k > len): the heap walk replays the same visit order, so the
    answers AND the work counters equal the oracle's."""
    vdb, O = mods
    base, ix, oh = graph5000
    qs = gist_like(6, dim=48, seed=42)
    qs[5] = base[17]
    _check(lambda q: ix.knn_with_ef(q, k, ef), lambda q: oh.knn(q, k, ef), qs, k)
    oi, od, oc, nd, ne = oh.knn_batch(qs, k, ef)
    ix.knn_with_ef(qs, k, ef)
    assert ix.hnsw_last_stats() == (nd, ne), "distance-evaluation / expansion counts differ from the oracle"


def test_hnsw_pq_large_ef(mods, graph5000):
    vdb, O = mods
    base, ix, oh = graph5000
    ix.pq_build(n_bits=4, m=16, train_n=800, max_iter=4, seed=6)
    pq = ix.pq_export()
    opq = O.PQ.from_centroids(48, 16, 4, 0, pq["centroids"])
    opq.set_codes(pq["codes"])
    qs = gist_like(5, dim=48, seed=43)
    for k, ef in ((10, 1500), (1200, 1300), (5000, 10)):
        _check(lambda q: ix.knn_pq(q, k, ef), lambda q: oh.knn_pq(opq, q, k, ef), qs, k)
    ix.pq_clear()


def test_hnsw_degenerate_duplicates_pool_overflow(mods):
    """4 000 copies of one point: once the result set is full of copies at distance d, every further copy with a smaller
    index passes check_candidate (full order) without being admitted (distance not strictly smaller), so the live
    candidates outgrow any fixed pool.  Round 1 reported an error here; now the overflowed queries are answered by the
    heap walk, with the oracle's answers and counters."""
    vdb, O = mods
    rng = np.random.default_rng(5)
    uniq = rng.standard_normal((400, 24)).astype(np.float32)
    base = np.concatenate([np.repeat(uniq[:1], 4000, axis=0), uniq, np.concatenate([uniq[1:41]] * 40)])
    base = base[rng.permutation(len(base))]
    ix = vdb.GpuIndex(24, "l2sqr")
    ix.batch_add(base)
    ix.hnsw_build(M=24, ef_construction=400, seed=8, batch=16, nthreads=8)
    oh = O.HNSW.from_graph(base, 0, 24, 400, ix.hnsw_export())
    qs = np.concatenate([uniq[:3] + 0.001, uniq[:2]]).astype(np.float32)
    for k, ef in ((10, 1000), (50, 600)):
        _check(lambda q: ix.knn_with_ef(q, k, ef), lambda q: oh.knn(q, k, ef), qs, k)
        oi, od, oc, nd, ne = oh.knn_batch(qs, k, ef)
        ix.knn_with_ef(qs, k, ef)
        assert ix.hnsw_last_stats() == (nd, ne), "counters of the call (fast walk + heap walk of the overflowed queries)"


def test_hnsw_pool_handover_to_heap_walk(mods, graph5000):
    """the same hand-over forced on an ordinary graph by lowering the LDS pool's capacity (test hook): some queries of
    the call finish in the fast walk, the others are repeated by the heap walk; answers and counters stay the oracle's"""
    vdb, O = mods
    base, ix, oh = graph5000
    qs = gist_like(40, dim=48, seed=44)
    try:
        ix.set_param("hnsw_pool_cap", 24)
        before = ix.get_stat("hnsw_heap_walk_queries")
        for k, ef in ((10, 128), (5, 16)):
            _check(lambda q: ix.knn_with_ef(q, k, ef), lambda q: oh.knn(q, k, ef), qs, k)
            oi, od, oc, nd, ne = oh.knn_batch(qs, k, ef)
            ix.knn_with_ef(qs, k, ef)
            assert ix.hnsw_last_stats() == (nd, ne)
        moved = ix.get_stat("hnsw_heap_walk_queries") - before
        assert moved > 0, "no query overflowed a 24-entry pool: the hook is not effective"
    finally:
        ix.set_param("hnsw_pool_cap", 2048)


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
def test_flat_pq_large_ef_and_k(mods, dist, kind):
    """FlatIndex::knn_pq with ef and k beyond 1024 (full ADC sort; heap replay of pq_resort for k > 1024), including
    k = ef = len as MetadataVecTable::search is driven by the reference's database test."""
    vdb, O = mods
    n = 4000
    base = gist_like(n, dim=32, seed=51)
    base[2000:2040] = base[:40]  # equal codes and equal exact distances across the cut
    ix = vdb.GpuIndex(32, dist)
    ix.batch_add(base)
    ix.pq_build(n_bits=4, m=8, train_n=600, max_iter=4, seed=2)
    pq = ix.pq_export()
    opq = O.PQ.from_centroids(32, 8, 4, kind, pq["centroids"])
    opq.set_codes(pq["codes"])
    qs = gist_like(4, dim=32, seed=52)
    qs[3] = base[5]
    for k, ef in ((10, 1500), (1100, 1100), (1100, 3000), (n, n), (n + 50, 7)):
        _check(lambda q: ix.knn_pq(q, k, ef), lambda q: O.flat_knn_pq(base, opq, q, k, ef, kind), qs, k)
    # the row-sharded export of the same shortlist for ef > 1024
    a, e = ix.knn_pq_shard(qs, 10, 1500)
    for q in range(len(qs)):
        oadc = opq.adc_all(qs[q], n)
        ok = np.sort(O.pair_keys(oadc, np.arange(n, dtype=np.uint64)))[:1500]
        assert np.array_equal(a[q], ok)


def test_ivf_large_k_and_probes(mods):
    vdb, O = mods
    n = 6000
    base = gist_like(n, dim=24, seed=61)
    base[3000:3030] = base[:30]
    ix = vdb.GpuIndex(24, "l2sqr")
    ix.batch_add(base)
    ix.ivf_build(1500, train_n=0, max_iter=3, seed=7)  # more clusters than the register-resident probe select holds
    ex = ix.ivf_export()
    oiv = O.IVF(base, ex["centroids"], 0, assign=ex["assign"])
    qs = gist_like(4, dim=24, seed=62)
    for k, npb in ((10, 1200), (2000, 1500), (1100, 40), (n + 5, 1500)):
        _check(lambda q: ix.ivf_knn(q, k, npb), lambda q: oiv.knn(q, k, npb), qs, k)


def test_vecdb_search_k_ef_len(mods):
    """database/mod.rs:551-607 drives search(k = len, ef = len); on a table larger than 1024 rows that used to fail."""
    vdb, O = mods
    n = 1500
    base = gist_like(n, dim=16, seed=71)
    db = vdb.VecDB("")
    db.create_table_if_not_exists("t", 16, "l2sqr")
    db.batch_add("t", base, [{"i": str(i)} for i in range(n)])
    db.build_hnsw_index("t", ef_construction=40)
    hits = db.search("t", base[3], n, ef=n)
    oh = O.HNSW.from_graph(base, 0, 16, 40, db._t("t").index.hnsw_export())
    oi, od = oh.knn(base[3], n, n)
    assert [int(h[0]["i"]) for h in hits] == oi.tolist()
    assert np.array_equal(np.array([h[1] for h in hits], dtype=np.float32), od)
    db.build_pq_table("t", train_proportion=0.2)
    hits = db.search("t", base[3], n, ef=n)
    assert len(hits) == len(oi)


def test_flat_full_order_many_blocks(mods):
    """FlatIndex::knn with k = len (flat_index.rs:48-57: the whole (distance, index) order) on 30 000 rows: the radix sort
    of k_sort.hip runs 8 blocks per pass; duplicated rows give equal distances that the index must order, rows with
    inf / NaN coordinates sort last (candidate_pair.rs:36-41)."""
    vdb, O = mods
    rng = np.random.default_rng(77)
    n, dim = 30000, 20
    base = rng.standard_normal((n, dim)).astype(np.float32)
    base[5000:5200] = base[100:300]          # exact duplicates: ties on the distance
    base[29990, 3] = np.inf
    base[29995, 0] = np.nan
    qs = rng.standard_normal((3, dim)).astype(np.float32)
    for dist, kind in (("l2sqr", 0), ("cosine", 1)):
        ix = vdb.GpuIndex(dim, dist)
        ix.batch_add(base)
        for k in (n, 2500):
            idx, d, cnt = ix.flat_knn(qs, k)
            for q in range(qs.shape[0]):
                oi, od = O.flat_knn(base, qs[q], k, kind)
                c = int(cnt[q])
                assert c == len(oi) == k
                assert idx[q, :c].tolist() == oi.tolist(), (dist, k, q)
                assert np.array_equal(d[q, :c], od, equal_nan=True), (dist, k, q)
