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


def test_flat_soak_configuration_87_of_round_2(mods):
    """Configuration #87 of tools/fuzz_flat.py seed 77 -- the one in flight when round 2's soak of an (uncommitted, since
    deleted) small-call cascade ended in a GPU memory-access fault (gpurun_out/fuzz_fs.log:88).  The
    generator state in front of it was recovered on the CPU (tools/replay_fuzz_flat.py -> tests/golden/...state.json), so
    these are the very rows and the very query: dim 192, 19 051 standard-normal rows, ONE query, k = 1, Cosine, one-part
    batch_add.  Runs it through every Flat tier choice of the shipped tree, and through the IVF scan, whose 8-bit / fp16
    tier kernels and Index::ensure_rows_q8 the deleted cascade had reused."""
    import json
    import os
    vdb, O = mods
    st = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fuzz_flat_seed77_cfg87_state.json")))
    rng = np.random.default_rng(0)
    rng.bit_generator.state = st["state"]
    dim = int(rng.choice([64, 96, 100, 128, 192, 256, 320, 384, 512, 768, 960, 1000, 1024, 1536]))
    n = int(rng.integers(17000, 60000))
    nq = int(rng.choice([1, 3, 17, 64, 65, 100, 128, 129, 200, 257]))
    k = int(rng.choice([1, 2, 5, 10, 16, 17, 33, 64, 70]))
    dist = str(rng.choice(["l2sqr", "cosine"]))
    style = int(rng.integers(0, 4))
    assert (dim, n, nq, k, dist, style) == (192, 19051, 1, 1, "cosine", 0)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    qs = rng.standard_normal((nq, dim)).astype(np.float32)
    oi, od, oc = O.flat_knn_batch(base, qs, k, 1, nthreads=8)
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base)
    idx, d, cnt = ix.flat_knn(qs, k)  # as a caller gets it today: one query on 19 051 rows takes the one-launch exact kernel
    assert cnt.tolist() == [1]
    _check(idx[0, :1], d[0, :1], oi[0][:1], od[0][:1])
    ix.set_param("flat_small", 1)  # ... and through the MFMA tiers the soak of round 2 was about
    for mode, half, tail in ((0, 0, 0), (0, 1, 0), (0, 2, 1), (2, 0, 1), (2, 1, 1), (1, 0, 0)):
        ix.set_flat_mode(mode)
        ix.set_param("flat_half", half)
        ix.set_param("flat_tail", tail)
        idx, d, cnt = ix.flat_knn(qs, k)
        assert cnt.tolist() == oc.tolist() == [1], (mode, half, tail)
        _check(idx[0, :1], d[0, :1], oi[0][:1], od[0][:1])
    ix.set_flat_mode(0)
    ix.set_param("flat_half", 0)
    ix.set_param("flat_tail", 0)
    ix.set_param("flat_small", 0)
    # the IVF scan over the same rows: few clusters and all of them probed -> probe lists of the whole table, the shape the
    # cascade gave the shared tier kernels (8-bit tier forced query-major = 2, cluster-major = 1, off = 0)
    ix.ivf_build(3, train_n=2000, max_iter=3, seed=1)
    ex = ix.ivf_export()
    iv = O.IVF(base, ex["centroids"], 1, assign=ex["assign"])
    want_i, want_d = iv.knn(qs[0], k, 3)
    try:
        for q8 in (2, 1, 0):
            ix.set_param("ivf_q8", q8)
            idx, d, cnt = ix.ivf_knn(qs, k, 3)
            assert int(cnt[0]) == len(want_i)
            _check(idx[0, :1], d[0, :1], want_i, want_d)
    finally:
        ix.set_param("ivf_q8", 1)
