"""GPU parity: IVFIndex (index_algorithm/ivf_index.rs) through the C ABI vs the CPU oracle.

Centroids are RNG-dependent in the reference (parity unpinned) and therefore an INPUT: built once by the library,
exported, and handed to the oracle.  Given the centroids, the cluster assignment (integer work) and the search results
(indices and distance bits, including the ResultSet::add replay order on ties) must be identical.
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


def test_ivf_index_test_restated(mods, gist_base):
    """ivf_index.rs:161-234: gist_1000 clipped to 12 dims, k = 7 clusters trained on len/10 rows, L2Sqr; knn(row 200,
    6) with the default 4 probes must return the same indices as FlatIndex, ascending distances."""
    vdb, O = mods
    base = np.ascontiguousarray(gist_base[:, :12])
    ix = vdb.GpuIndex(12, "l2sqr")
    ix.batch_add(base)
    ix.ivf_build(7, train_n=len(base) // 10, max_iter=20, tol=1e-6, seed=42)
    assert ix.has_ivf() and ix.ivf_info() == {"present": True, "k": 7, "default_n_probes": 4}
    gi, gd = ix.ivf_knn(base[200], 6)
    fi, fd = ix.flat_knn(base[200], 6)
    assert gi.tolist() == fi.tolist()
    assert len(gi) == 6 and np.all(np.diff(gd) >= 0)
    # and it equals the oracle's IVF on the same centroids
    ex = ix.ivf_export()
    iv = O.IVF(base, ex["centroids"], 0)
    assert np.array_equal(ex["assign"], iv.assign), "GPU cluster assignment differs from find_nearest"
    oi, od = iv.knn(base[200], 6, 4)
    assert gi.tolist() == oi.tolist() and np.array_equal(gd, od)


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
def test_ivf_parity_with_ties(mods, dist, kind):
    vdb, O = mods
    rng = np.random.default_rng(17)
    n, dim = 6000, 48
    base = rng.standard_normal((n, dim)).astype(np.float32)
    base[5000:5040] = base[100:140]  # exact duplicates: equal distances, usually in the same cluster, sometimes not
    base[5100] = base[7] * np.float32(1.0)  # and a lone duplicate
    qs = rng.standard_normal((25, dim)).astype(np.float32)
    qs[:5] = base[100:105] + np.float32(0.01)  # queries next to duplicated rows: ties at the cut
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base)
    ix.ivf_build(23, train_n=1000, max_iter=10, seed=5)
    ex = ix.ivf_export()
    iv = O.IVF(base, ex["centroids"], kind)
    assert np.array_equal(ex["assign"], iv.assign)
    for k, n_probes in ((1, 1), (10, 4), (10, 0), (40, 3), (7, 23), (5, 100), (1000, 2)):
        idx, d, cnt = ix.ivf_knn(qs, k, n_probes)
        for q in range(qs.shape[0]):
            oi, od = iv.knn(qs[q], k, n_probes if n_probes else 4)
            c = int(cnt[q])
            assert c == len(oi), (k, n_probes, q)
            assert idx[q, :c].tolist() == oi.tolist(), (k, n_probes, q)
            assert np.array_equal(d[q, :c], od), (k, n_probes, q)


def test_ivf_attach_external_and_invalidation(mods):
    """Reference-built indices come as centroids + clusters (bincode): attach both; writes drop the index."""
    vdb, O = mods
    base = gist_like(3000, dim=96, seed=3)
    cents = O.kmeans(base[:500], 0, 96, 9, 8, 1e-6, 0, 7)
    iv = O.IVF(base, cents, 0)
    ix = vdb.GpuIndex(96, "l2sqr")
    ix.batch_add(base)
    ix.ivf_attach(cents, iv.assign)
    qs = gist_like(8, dim=96, seed=4)
    idx, d, cnt = ix.ivf_knn(qs, 10, 2)
    for q in range(8):
        oi, od = iv.knn(qs[q], 10, 2)
        assert idx[q].tolist() == oi.tolist() and np.array_equal(d[q], od)
    # attach without clusters: assignment on the GPU
    ix.ivf_attach(cents)
    assert np.array_equal(ix.ivf_export()["assign"], iv.assign)
    with pytest.raises(vdb.VdbError):
        ix.ivf_attach(cents, np.full(3000, 9, dtype=np.uint64))  # cluster id out of range
    ix.add(base[0])
    assert not ix.has_ivf()
    with pytest.raises(vdb.VdbError):
        ix.ivf_knn(qs[0], 3)
    with pytest.raises(vdb.VdbError):
        ix.ivf_build(0)


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
@pytest.mark.parametrize("data", ["gist", "cancel", "decades", "dups"])
def test_ivf_half_prepass_keeps_results(mods, data, dist, kind):
    """The scan's certified half-precision pre-pass (ivf.hip: k_ivf_half_bounds / k_ivf_keep) drops offers that cannot be
    among the k nearest before the f32 rows are fetched; the ResultSet::add replay over what is left must give the oracle's
    answer, ties at the cut included (duplicated rows), on the corpora that stress the bound."""
    vdb, O = mods
    rng = np.random.default_rng(321)
    n, dim, nq = 24000, 128, 24  # 30 clusters of ~800 rows: 9 and 30 probes give lists long enough for the 8-bit tier
    if data == "gist":
        base, qs = gist_like(n, dim=dim, seed=5), gist_like(nq, dim=dim, seed=6)
    elif data == "cancel":
        c = rng.standard_normal(dim).astype(np.float32)
        c *= np.float32(50.0) / np.linalg.norm(c)
        base = (c[None, :] + np.float32(1e-2) * rng.standard_normal((n, dim))).astype(np.float32)
        qs = (c[None, :] + np.float32(1e-2) * rng.standard_normal((nq, dim))).astype(np.float32)
    elif data == "decades":
        base = (rng.standard_normal((n, dim)) * 10.0 ** rng.uniform(-2, 2, size=(n, 1))).astype(np.float32)
        qs = (rng.standard_normal((nq, dim)) * 10.0 ** rng.uniform(-2, 2, size=(nq, 1))).astype(np.float32)
    else:  # every row four times, interleaved: equal distances straddle every cut
        u = gist_like(n // 4, dim=dim, seed=7)
        base = np.concatenate([u, u, u, u])
        qs = gist_like(nq, dim=dim, seed=8)
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base)
    ix.ivf_build(30, train_n=2000, max_iter=5, seed=3)
    ex = ix.ivf_export()
    iv = O.IVF(base, ex["centroids"], kind, assign=ex["assign"])
    for k, npb in ((10, 4), (1, 2), (40, 9), (3, 30)):
        ix.set_param("ivf_half", 0)
        try:
            i0, d0, c0 = ix.ivf_knn(qs, k, npb)
        finally:
            ix.set_param("ivf_half", 1)
        ix.set_param("ivf_q8", 0)  # fp16 tier alone
        try:
            i2, d2, c2 = ix.ivf_knn(qs, k, npb)
        finally:
            ix.set_param("ivf_q8", 1)
        i1, d1, c1 = ix.ivf_knn(qs, k, npb)  # 8-bit tier (lists of at least 4 x max(1024, 64 k) offers) + fp16 tier
        assert np.array_equal(i0, i1) and np.array_equal(d0, d1) and np.array_equal(c0, c1), (k, npb)
        assert np.array_equal(i0, i2) and np.array_equal(d0, d2) and np.array_equal(c0, c2), (k, npb)
        for q in range(nq):
            oi, od = iv.knn(qs[q], k, npb)
            c = int(c1[q])
            assert c == len(oi)
            assert i1[q, :c].tolist() == oi.tolist() and np.array_equal(d1[q, :c], od), (k, npb, q)
