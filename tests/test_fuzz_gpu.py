"""GPU: seeded random configurations of the Flat path (dim, rows, queries, k, distance, kernel choice, data scale)
against the oracle -- the shapes nobody thought of.  Deterministic: the seeds are the test ids."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import lab_1806_vec_db_amd as vdb
    from oracle import oracle as O
    return vdb, O


def _check(gi, gd, oi, od):
    assert gi.tolist() == oi.tolist()
    assert np.array_equal(gd, od)


@pytest.mark.parametrize("seed", list(range(24)))
def test_flat_random_configuration(mods, seed):
    vdb, O = mods
    rng = np.random.default_rng(1000 + seed)
    dim = int(rng.choice([3, 8, 17, 48, 64, 100, 128, 200, 384, 515, 960, 1024, 1100, 1536]))
    n = int(rng.integers(40, 60000)) if dim <= 400 else int(rng.integers(40, 22000))
    nq = int(rng.choice([1, 2, 31, 33, 64, 65, 127, 129, 200]))
    k = int(rng.choice([1, 3, 10, 16, 33, 100]))
    dist, kind = (("l2sqr", 0), ("cosine", 1))[int(rng.integers(0, 2))]
    scale = float(rng.choice([1e-3, 1.0, 300.0]))
    style = int(rng.integers(0, 3))
    if style == 0:
        base = rng.standard_normal((n, dim)).astype(np.float32)
    elif style == 1:
        base = np.abs(rng.standard_normal((n, dim)) * 0.05 + 0.07).astype(np.float32)
    else:  # few distinct rows: many exact ties
        proto = rng.standard_normal((7, dim)).astype(np.float32)
        base = proto[rng.integers(0, 7, n)] + (rng.standard_normal((n, dim)) * (rng.integers(0, 2) * 1e-3)).astype(np.float32)
    base = (base * np.float32(scale)).astype(np.float32)
    qs = (base[rng.integers(0, n, nq)] + rng.standard_normal((nq, dim)).astype(np.float32) * np.float32(0.05 * scale)).astype(np.float32)
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base[: n // 2])
    ix.batch_add(base[n // 2:])
    mode = int(rng.choice([0, 1, 2]))
    ix.set_flat_mode(mode)
    ix.set_param("flat_gemm", int(rng.choice([0, 1, 2])))
    try:
        idx, d, cnt = ix.flat_knn(qs, k)
    finally:
        ix.set_param("flat_gemm", 0)
    oi, od, oc = O.flat_knn_batch(base, qs, k, kind, nthreads=8)
    for q in range(nq):
        c = int(cnt[q])
        assert c == int(oc[q]) == min(k, n)
        _check(idx[q, :c], d[q, :c], oi[q, :c], od[q, :c])


@pytest.mark.parametrize("seed", list(range(10)))
def test_pq_ivf_hnsw_random_configuration(mods, seed):
    """PQ-Flat, IVF and HNSW on one random corpus: centroids / graph built by the library, handed to the oracle."""
    vdb, O = mods
    rng = np.random.default_rng(5000 + seed)
    dim = int(rng.choice([12, 31, 64, 96, 130]))
    n = int(rng.integers(300, 9000))
    nq = int(rng.choice([1, 5, 40]))
    k = int(rng.choice([1, 5, 10, 25]))
    dist, kind = (("l2sqr", 0), ("cosine", 1))[int(rng.integers(0, 2))]
    base = (rng.standard_normal((n, dim)) * rng.uniform(0.1, 3.0, dim)).astype(np.float32)
    base[n - 5:] = base[:5]
    qs = (base[rng.integers(0, n, nq)] + rng.standard_normal((nq, dim)).astype(np.float32) * np.float32(0.1)).astype(np.float32)
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base)
    # PQ
    n_bits = int(rng.choice([4, 8]))
    m = int(rng.integers(1, min(dim, 40) + 1))
    ef = int(rng.choice([k, 3 * k, 64, 300]))
    ix.pq_build(n_bits=n_bits, m=m, train_n=min(n, 500), max_iter=3, seed=seed)
    pq = ix.pq_export()
    opq = O.PQ.from_centroids(dim, m, n_bits, kind, pq["centroids"])
    opq.encode_all(base)
    assert np.array_equal(pq["codes"], opq.codes)
    idx, d, cnt = ix.knn_pq(qs, k, ef)
    for q in range(nq):
        oi, od = O.flat_knn_pq(base, opq, qs[q], k, ef, kind)
        c = int(cnt[q])
        assert c == len(oi)
        _check(idx[q, :c], d[q, :c], oi, od)
    # IVF
    kc = int(rng.integers(1, 40))
    npb = int(rng.choice([1, 2, 4, 9, 64]))
    ix.ivf_build(kc, train_n=min(n, 400), max_iter=4, seed=seed)
    ex = ix.ivf_export()
    iv = O.IVF(base, ex["centroids"], kind)
    assert np.array_equal(ex["assign"], iv.assign)
    idx, d, cnt = ix.ivf_knn(qs, k, npb)
    for q in range(nq):
        oi, od = iv.knn(qs[q], k, npb)
        c = int(cnt[q])
        assert c == len(oi)
        _check(idx[q, :c], d[q, :c], oi, od)
    # HNSW (small graphs: the builder is serial below 1000 rows and batched above)
    if n <= 4000:
        M = int(rng.choice([4, 8, 16]))
        efc = int(rng.choice([20, 60]))
        batch = int(rng.choice([1, 16]))
        ix.hnsw_build(M=M, ef_construction=efc, seed=seed, batch=batch, nthreads=4)
        oh = O.HNSW.from_graph(base, kind, M, efc, ix.hnsw_export())
        efs = int(rng.choice([k, 40, 200]))
        idx, d, cnt = ix.knn_with_ef(qs, k, efs)
        for q in range(nq):
            oi, od = oh.knn(qs[q], k, efs)
            c = int(cnt[q])
            assert c == len(oi)
            _check(idx[q, :c], d[q, :c], oi, od)
