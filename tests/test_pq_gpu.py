"""GPU parity: PQ encode, ADC scan and FlatIndex::knn_pq through the C ABI vs the CPU oracle.

Integer work (codes, neighbour indices) must be bit-exact; the returned distances are the exact
re-sorted ones (flat_index.rs:102) and are asserted bit-exact as well.  Centroids are RNG-dependent in
the reference (parity unpinned), so they are an INPUT: trained once by the library and handed to the oracle.
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


def _oracle_pq(O, ix, base, kind):
    pq = ix.pq_export()
    opq = O.PQ.from_centroids(ix.dim, pq["m"], pq["n_bits"], kind, pq["centroids"])
    opq.encode_all(base)
    return pq, opq


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
@pytest.mark.parametrize("n_bits,m", [(4, 320), (4, 240), (8, 96), (8, 320)])
def test_gist1000_knn_pq(mods, gist_base, gist_test, dist, kind, n_bits, m):
    vdb, O = mods
    ix = vdb.GpuIndex(960, dist)
    ix.batch_add(gist_base)
    ix.pq_build(n_bits=n_bits, m=m, train_n=300, max_iter=5, seed=3)
    assert ix.has_pq()
    pq, opq = _oracle_pq(O, ix, gist_base, kind)
    assert np.array_equal(pq["codes"], opq.codes), "GPU pq_encode differs from the oracle"
    for ef in (10, 100, 200):
        idx, d, cnt = ix.knn_pq(gist_test[:24], 10, ef)
        for q in range(24):
            oi, od = O.flat_knn_pq(gist_base, opq, gist_test[q], 10, ef, kind)
            assert idx[q].tolist() == oi.tolist(), (ef, q)
            assert np.array_equal(d[q], od), (ef, q)
    ix.pq_clear()
    assert not ix.has_pq()
    with pytest.raises(vdb.VdbError):
        ix.knn_pq(gist_test[0], 10, 100)


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
def test_small_odd_m_and_n_lt_k(mods, dist, kind):
    """pq_table.rs:324-372 setting (5 vectors, dim 8, m 2, 16 centroids > 5 points) and an odd m."""
    vdb, O = mods
    rng = np.random.default_rng(42)
    for dim, m, n in ((8, 2, 5), (13, 7, 64), (13, 5, 64)):
        base = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
        ix = vdb.GpuIndex(dim, dist)
        ix.batch_add(base)
        ix.pq_build(n_bits=4, m=m, train_n=0, seed=42)
        pq, opq = _oracle_pq(O, ix, base, kind)
        assert np.array_equal(pq["codes"], opq.codes)
        for q in range(n if n < 8 else 8):
            for k, ef in ((3, 3), (n, n), (2, 50)):
                gi, gd = ix.knn_pq(base[q], k, ef)
                oi, od = O.flat_knn_pq(base, opq, base[q], k, ef, kind)
                assert gi.tolist() == oi.tolist(), (dim, m, q, k, ef)
                assert np.array_equal(gd, od)


def test_attach_external_codes(mods, gist_base, gist_test):
    """vdb_pq_attach with centroids AND codes supplied (reference-built tables)."""
    vdb, O = mods
    opq = O.PQ.train(gist_base, 320, 4, 0, k_means_size=200, max_iter=3, seed=9)
    ix = vdb.GpuIndex(960, "l2sqr")
    ix.batch_add(gist_base)
    ix.pq_attach(4, 320, opq.centroids, opq.codes)
    idx, d, cnt = ix.knn_pq(gist_test[:16], 10, 128)
    for q in range(16):
        oi, od = O.flat_knn_pq(gist_base, opq, gist_test[q], 10, 128)
        assert idx[q].tolist() == oi.tolist() and np.array_equal(d[q], od)


def test_pq_resort_ties(mods):
    """Duplicated rows give exact-distance ties at the cut; pq_resort keeps the earlier-in-ADC-order pair
    (candidate_pair.rs:61-74,102-108), which is NOT the lexicographic rule."""
    vdb, O = mods
    rng = np.random.default_rng(5)
    uniq = rng.standard_normal((40, 24)).astype(np.float32)
    base = np.concatenate([uniq, uniq, uniq])[rng.permutation(120)]
    ix = vdb.GpuIndex(24, "l2sqr")
    ix.batch_add(base)
    ix.pq_build(n_bits=4, m=8, train_n=0, max_iter=4, seed=1)
    pq, opq = _oracle_pq(O, ix, base, 0)
    for q in range(10):
        for k, ef in ((4, 30), (5, 7), (7, 120)):
            gi, gd = ix.knn_pq(uniq[q] + 0.05, k, ef)
            oi, od = O.flat_knn_pq(base, opq, uniq[q] + 0.05, k, ef)
            assert gi.tolist() == oi.tolist(), (q, k, ef)
            assert np.array_equal(gd, od)


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
def test_fused_threshold_path(mods, dist, kind):
    """n >= 65536 takes the sampled-threshold + in-kernel filter path of the ADC scan (pq.hip); duplicated rows
    put exact ADC ties at the cut."""
    vdb, O = mods
    base = gist_like(70000, dim=96, seed=11)
    base[60000:60050] = base[100:150]  # equal codes -> equal ADC distances
    qs = gist_like(10, dim=96, seed=12)
    ix = vdb.GpuIndex(96, dist)
    ix.batch_add(base)
    ix.pq_build(n_bits=4, m=32, train_n=2000, max_iter=4, seed=5)
    pq, opq = _oracle_pq(O, ix, base, kind)
    assert np.array_equal(pq["codes"], opq.codes)
    for k, ef in ((10, 10), (10, 128), (5, 1000)):
        idx, d, cnt = ix.knn_pq(qs, k, ef)
        for q in range(qs.shape[0]):
            oi, od = O.flat_knn_pq(base, opq, qs[q], k, ef, kind)
            assert idx[q].tolist() == oi.tolist(), (k, ef, q)
            assert np.array_equal(d[q], od), (k, ef, q)


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
@pytest.mark.parametrize("n,shards", [(70000 * 3, 3), (5000, 2), (37, 4)])
def test_row_sharded_knn_pq(mods, dist, kind, n, shards):
    """PQ codes shard by rows, centroids replicated; the per-shard ADC top-ef rows are merged in (adc, id)
    order and only then re-sorted (candidate_pair.rs:102-108).  Host merge and GPU merge must both equal the
    unsharded search."""
    import torch
    vdb, O = mods
    from lab_1806_vec_db_amd.index import pq_merge_resort
    from lab_1806_vec_db_amd.shard import shard_bounds
    dim = 48
    base = gist_like(n, dim=dim, seed=21)
    base[n - 3:] = base[:3]  # equal codes on different shards: ADC ties across the shard boundary
    qs = gist_like(9, dim=dim, seed=22)
    whole = vdb.GpuIndex(dim, dist)
    whole.batch_add(base)
    whole.pq_build(n_bits=4, m=16, train_n=min(n, 2000), max_iter=4, seed=5)
    pq = whole.pq_export()
    parts = []
    for r in range(shards):
        r0, r1 = shard_bounds(n, shards, r)
        ix = vdb.GpuIndex(dim, dist)
        ix.set_id_offset(r0)
        ix.batch_add(base[r0:r1])
        ix.pq_attach(4, 16, pq["centroids"], None)  # centroids replicated, codes encoded per shard
        parts.append(ix)
    opq = O.PQ.from_centroids(dim, 16, 4, kind, pq["centroids"])
    opq.encode_all(base)
    for k, ef in ((10, 100), (10, 10), (3, 1000), (20, 5)):
        rows = [ix.knn_pq_shard(qs, k, ef) for ix in parts]
        adc = np.stack([r[0] for r in rows])
        ex = np.stack([r[1] for r in rows])
        hi, hd, hc = pq_merge_resort(adc, ex, k)
        d_adc = torch.from_numpy(adc.view(np.int64)).cuda()
        d_ex = torch.from_numpy(ex.view(np.int64)).cuda()
        nq, efk = qs.shape[0], max(k, ef)
        o_idx = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        o_dist = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        o_cnt = torch.empty((nq,), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        parts[0].pq_merge_resort_device(d_adc.data_ptr(), d_ex.data_ptr(), shards, nq, efk, k, o_idx.data_ptr(),
                                        o_dist.data_ptr(), o_cnt.data_ptr())
        gi, gd, gc = o_idx.cpu().numpy(), o_dist.cpu().numpy(), o_cnt.cpu().numpy()
        wi, wd, wc = whole.knn_pq(qs, k, ef)
        for q in range(nq):
            oi, od = O.flat_knn_pq(base, opq, qs[q], k, ef, kind)
            c = len(oi)
            assert int(hc[q]) == c and int(gc[q]) == c and int(wc[q]) == c
            assert hi[q, :c].tolist() == oi.tolist(), (k, ef, q)
            assert gi[q, :c].tolist() == oi.tolist(), (k, ef, q)
            assert wi[q, :c].tolist() == oi.tolist(), (k, ef, q)
            assert np.array_equal(hd[q, :c], od) and np.array_equal(gd[q, :c], od)


def test_gistlike_knn_pq_large(mods):
    vdb, O = mods
    base = gist_like(30000, seed=1806)
    qs = gist_like(12, seed=1807)
    ix = vdb.GpuIndex(960, "l2sqr")
    ix.batch_add(base)
    ix.pq_build(n_bits=4, m=320, train_n=1000, max_iter=5, seed=42)
    pq, opq = _oracle_pq(O, ix, base, 0)
    assert np.array_equal(pq["codes"], opq.codes)
    idx, d, cnt = ix.knn_pq(qs, 10, 128)
    for q in range(qs.shape[0]):
        oi, od = O.flat_knn_pq(base, opq, qs[q], 10, 128)
        assert idx[q].tolist() == oi.tolist() and np.array_equal(d[q], od)


def test_add_after_pq_build_clears_table(mods):
    """MetadataVecTable::add / batch_add clear the PQ table first (metadata_vec_table.rs:65,77): at the index level an
    add after pq_build must drop the codes, so that knn_pq reports "needs a PQ table" instead of scanning code rows that
    do not exist (device out-of-bounds read before this rule was enforced in the library)."""
    vdb, O = mods
    base = gist_like(3000, dim=64, seed=11)
    ix = vdb.GpuIndex(64, "l2sqr")
    ix.batch_add(base[:2000])
    ix.pq_build(n_bits=4, m=16, train_n=500, max_iter=3, seed=1)
    assert ix.has_pq()
    ix.batch_add(base[2000:])
    assert len(ix) == 3000 and not ix.has_pq()
    with pytest.raises(vdb.VdbError):
        ix.knn_pq(base[0], 5, 32)
    with pytest.raises(vdb.VdbError):
        ix.knn_pq_shard(base[:2], 5, 32)
    # rebuilt table covers all rows again and matches the oracle
    ix.pq_build(n_bits=4, m=16, train_n=500, max_iter=3, seed=1)
    pq, opq = _oracle_pq(O, ix, base, 0)
    assert pq["codes"].shape[0] == 3000 and np.array_equal(pq["codes"], opq.codes)
    gi, gd = ix.knn_pq(base[7], 5, 32)
    oi, od = O.flat_knn_pq(base, opq, base[7], 5, 32, 0)
    assert gi.tolist() == oi.tolist() and np.array_equal(gd, od)


def test_attach_rejects_wrong_sizes(mods):
    """the wrappers check array sizes before the C side copies n * enc_dim / (1 << n_bits) * dim elements"""
    vdb, O = mods
    base = gist_like(200, dim=32, seed=12)
    ix = vdb.GpuIndex(32, "l2sqr")
    ix.batch_add(base)
    cent = np.zeros(16 * 32, dtype=np.float32)
    with pytest.raises(vdb.VdbError):
        ix.pq_attach(4, 8, cent[:-1])
    with pytest.raises(vdb.VdbError):
        ix.pq_attach(4, 8, cent, np.zeros((199, 4), dtype=np.uint8))
    with pytest.raises(vdb.VdbError):
        ix.pq_attach(5, 8, cent)
    with pytest.raises(vdb.VdbError):
        ix.ivf_attach(np.zeros((4, 31), dtype=np.float32))
    with pytest.raises(vdb.VdbError):
        ix.ivf_attach(np.zeros((4, 32), dtype=np.float32), np.zeros(199, dtype=np.uint64))
    ix.hnsw_build(M=4, ef_construction=16, seed=1)
    g = ix.hnsw_export()
    bad = dict(g)
    bad["level0"] = g["level0"][:-1]
    with pytest.raises(vdb.VdbError):
        ix.hnsw_attach(4, 16, bad)
    bad = dict(g)
    bad["upper_len"] = np.concatenate([g["upper_len"], [0]]).astype(np.uint64)
    with pytest.raises(vdb.VdbError):
        ix.hnsw_attach(4, 16, bad)
    ix.hnsw_attach(4, 16, g)  # the untouched export still attaches


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
@pytest.mark.parametrize("n_bits", [4, 8])
def test_pq_training_gpu_assignment_equals_host_kmeans(mods, dist, kind, n_bits):
    """PQ training runs Lloyd's assignment step (k_means.rs:117-120) on the GPU for all groups at once (the encoder
    kernel on the training rows) and the update on the host.  Given the same seeding stream the centroids must equal
    the oracle's all-host k-means bit for bit: same (distance, index) argmin, same row-order sums, same stopping rule.
    (The stream itself is this build's -- splitmix64 per group -- the reference's ChaCha stream is unpinned)"""
    vdb, O = mods
    n, dim, m = 700, 30, 7   # uneven groups: 5,5,4,4,4,4,4
    base = gist_like(n, dim=dim, seed=33)
    base[100:120] = base[:20]  # duplicate rows: equal distances in the argmin
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base)
    seed = 12345
    ix.pq_build(n_bits=n_bits, m=m, train_n=0, max_iter=6, tol=1e-6, seed=seed)
    got = ix.pq_export()["centroids"]
    gs = O.pq_groups(dim, m)
    kc = 1 << n_bits
    for g in range(m):
        gseed = (seed ^ (0xD1B54A32D192ED03 * (g + 1))) & ((1 << 64) - 1)
        c0, c1 = gs[g]
        want = O.kmeans(base, c0, c1, kc, 6, 1e-6, kind, gseed)
        have = got[kc * c0: kc * c1].reshape(kc, -1)
        assert np.array_equal(have.view(np.uint32), np.asarray(want, dtype=np.float32).reshape(kc, -1).view(np.uint32)), g
