"""GPU parity of the HNSW walk's certified half-precision pre-pass (hnsw.hip: hnsw_half_dots / hnsw_half_rejects).

The pre-pass scores every fresh neighbour from a row-major fp16 image and drops the ones whose reference distance is
certainly above the worst result; the others are scored with the reference's strict f32 fold.  A wrong drop would change
the result list or the counters, so parity with the oracle (hnsw_index.rs:258-291, same graph) IS the check of the bound.
The corpora are the ones that stress it: rows that differ in the last bits (cancellation: every distance is the
difference of two large numbers), row norms over four decades (the image's absolute error against small rows), and
plain gist-like rows; L2Sqr and Cosine; rows added after the image was built; the image after a scale change.
"""
import numpy as np
import pytest

from conftest import gist_like

pytestmark = pytest.mark.gpu

DIM = 128  # dim % 64 == 0: the pre-pass fetches whole 128-B lines of the fp16 image


@pytest.fixture(scope="module")
def mods():
    import lab_1806_vec_db_amd as vdb
    from oracle import oracle as O
    return vdb, O


def _corpus(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "gist":
        return gist_like(n, dim=DIM, seed=seed), gist_like(24, dim=DIM, seed=seed + 1)
    if kind == "cancel":  # one vector of norm 50 plus 1e-2 noise: distances ~1e-2 from sums ~5000
        c = rng.standard_normal(DIM).astype(np.float32)
        c *= np.float32(50.0) / np.linalg.norm(c)
        base = (c[None, :] + np.float32(1e-2) * rng.standard_normal((n, DIM))).astype(np.float32)
        qs = (c[None, :] + np.float32(1e-2) * rng.standard_normal((24, DIM))).astype(np.float32)
        return base, qs
    if kind == "decades":  # row norms from 1e-2 to 1e2
        base = rng.standard_normal((n, DIM)).astype(np.float32)
        base *= (10.0 ** rng.uniform(-2, 2, size=(n, 1))).astype(np.float32)
        qs = rng.standard_normal((24, DIM)).astype(np.float32)
        qs *= (10.0 ** rng.uniform(-2, 2, size=(24, 1))).astype(np.float32)
        return base, qs
    raise ValueError(kind)


def _check(vdb, O, ix, base, qs, kind, k, ef, M, efc):
    oh = O.HNSW.from_graph(base, kind, M, efc, ix.hnsw_export())
    ix.set_param("hnsw_half", 2)  # (2: also for calls of a few queries; 1 = auto leaves those to the exact stage alone)
    idx, d, cnt = ix.knn_with_ef(qs, k, ef)
    st_on = ix.hnsw_last_stats()
    dropped = ix.get_stat("hnsw_half_dropped")
    ix.set_param("hnsw_half", 0)
    try:
        idx0, d0, cnt0 = ix.knn_with_ef(qs, k, ef)
        st_off = ix.hnsw_last_stats()
        assert ix.get_stat("hnsw_half_dropped") == 0
    finally:
        ix.set_param("hnsw_half", 1)
    assert st_on == st_off
    assert np.array_equal(idx, idx0) and np.array_equal(d, d0) and np.array_equal(cnt, cnt0)
    oi, od, oc, nd, ne = oh.knn_batch(qs, k, ef, nthreads=4)
    assert cnt.tolist() == oc.tolist()
    for q in range(qs.shape[0]):
        c = int(cnt[q])
        assert idx[q, :c].tolist() == oi[q, :c].tolist(), q
        assert np.array_equal(d[q, :c], od[q, :c]), q
    assert st_on == (nd, ne), "distance-evaluation / expansion counts differ from the oracle"
    return dropped, nd


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
@pytest.mark.parametrize("data", ["gist", "cancel", "decades"])
def test_prepass_keeps_results_and_counters(mods, data, dist, kind):
    vdb, O = mods
    base, qs = _corpus(data, 4000, 4242)
    ix = vdb.GpuIndex(DIM, dist)
    ix.batch_add(base)
    ix.hnsw_build(M=16, ef_construction=60, seed=3, batch=32, nthreads=8)
    dropped, nd = _check(vdb, O, ix, base, qs, kind, 10, 48, 16, 60)
    if data != "cancel":  # (there the bound is as large as the distances: nothing may be ruled out, and nothing is needed)
        assert dropped > nd // 4, "the pre-pass is not ruling anything out: is it running?"
    _check(vdb, O, ix, base, qs, kind, 5, 5, 16, 60)  # a one-lane result list: the worst result moves with every admission


def test_image_follows_added_rows_and_scale_changes(mods):
    """rows added after the first walk extend the fp16 image; a row with a much larger norm changes the scale of the Flat
    mirror (half_refresh) and the image is rebuilt with it"""
    vdb, O = mods
    base, qs = _corpus("gist", 3000, 99)
    ix = vdb.GpuIndex(DIM, "l2sqr")
    ix.batch_add(base[:2500])
    ix.hnsw_build(M=12, ef_construction=40, seed=9, batch=16, nthreads=4)
    _check(vdb, O, ix, base[:2500], qs, 0, 8, 32, 12, 40)
    ix.batch_add(base[2500:])  # HNSWIndex::add per row
    assert len(ix) == 3000 and ix.has_hnsw()
    _check(vdb, O, ix, base, qs, 0, 8, 32, 12, 40)
    big = (base[:40] * np.float32(300.0)).astype(np.float32)  # norms 300x the largest so far: new scale
    ix.batch_add(big)
    allrows = np.concatenate([base, big])
    _check(vdb, O, ix, allrows, qs, 0, 8, 32, 12, 40)
    _check(vdb, O, ix, allrows, big[:8] + np.float32(0.5), 0, 8, 32, 12, 40)


@pytest.mark.parametrize("seed", list(range(8)))
def test_prepass_random_configuration(mods, seed):
    """random dims (multiples of 64), per-dimension scales, duplicates, graph widths and result-list sizes"""
    vdb, O = mods
    rng = np.random.default_rng(8800 + seed)
    dim = int(rng.choice([64, 128, 192, 320]))
    n = int(rng.integers(600, 5000))
    nq = int(rng.choice([3, 17, 40]))
    k = int(rng.choice([1, 5, 10, 25]))
    dist, kind = (("l2sqr", 0), ("cosine", 1))[int(rng.integers(0, 2))]
    base = (rng.standard_normal((n, dim)) * rng.uniform(0.05, 4.0, dim)).astype(np.float32)
    if seed % 3 == 0:  # clusters: many near-ties around the worst result
        base = (base[rng.integers(0, 20, n)] + np.float32(0.02) * rng.standard_normal((n, dim))).astype(np.float32)
    base[n - 5:] = base[:5]
    qs = (base[rng.integers(0, n, nq)] + rng.standard_normal((nq, dim)).astype(np.float32) * np.float32(0.1)).astype(np.float32)
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base)
    M = int(rng.choice([4, 8, 16]))
    efc = int(rng.choice([20, 60]))
    ix.hnsw_build(M=M, ef_construction=efc, seed=seed, batch=int(rng.choice([1, 16])), nthreads=4)
    ef = int(rng.choice([k, 40, 200]))
    _check(vdb, O, ix, base, qs, kind, k, ef, M, efc)
