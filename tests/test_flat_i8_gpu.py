"""GPU parity: the 8-bit first pass of Flat L2Sqr searches (k_flat_gemm8 / k_i8.hip / k_flat_tail_lb).

The pass streams a centred int8 mirror (1 B/element) and produces keys that are LOWER BOUNDS of the distances
(D(r, q) >= key(r, q) + O_q for every row); the exact stage walks the hit list in key order and stops when the k-th exact
distance is below the next bound; what it cannot close goes on to the fp16 / split-bf16 / exact tiers.  Tested here:
(1) the bound itself, for every (row, query) pair of several corpora, against float64; (2) bit-equality of whole searches
with the oracle, whichever tier answers; (3) the upkeep of the mirror (rows added after the first search, swap_remove, the
re-centring when the table has doubled); (4) degenerate inputs.
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


def _check_all(idx, d, cnt, oi, od, oc):
    assert cnt.tolist() == oc.tolist()
    for q in range(idx.shape[0]):
        assert idx[q].tolist() == oi[q].tolist(), (q, idx[q], oi[q])
        assert np.array_equal(d[q], od[q]), (q, d[q], od[q])


def _corpus(name, n, dim, rng):
    if name == "gist":
        return gist_like(n, dim=dim, seed=int(rng.integers(1 << 30)))
    if name == "normal":
        return rng.standard_normal((n, dim)).astype(np.float32)
    if name == "offset":  # cancellation-heavy: one large common vector + small noise (what the centring is for)
        c = (rng.standard_normal(dim) * 50 / np.sqrt(dim)).astype(np.float32)
        return (c[None, :] + 1e-2 * rng.standard_normal((n, dim))).astype(np.float32)
    if name == "decades":  # row norms over four decades
        s = np.exp(rng.uniform(np.log(1e-2), np.log(1e2), size=(n, 1))).astype(np.float32)
        return (rng.standard_normal((n, dim)) * s).astype(np.float32)
    if name == "sparse":  # a few large coordinates per row: the per-row scale is set by outliers
        x = 0.01 * rng.standard_normal((n, dim))
        for r in range(n):
            x[r, rng.integers(dim, size=3)] += rng.standard_normal(3) * 5
        return x.astype(np.float32)
    if name == "integers":  # exactly representable rows (rounding error zero for many of them)
        return rng.integers(-3, 4, size=(n, dim)).astype(np.float32)
    raise ValueError(name)


@pytest.mark.parametrize("name", ["gist", "normal", "offset", "decades", "sparse", "integers"])
@pytest.mark.parametrize("dim", [960, 192, 128, 320])
def test_keys_are_lower_bounds(mods, name, dim):
    """key + O_q <= D for EVERY (row, query) pair, queries from the corpus' distribution, from another one, and rows themselves"""
    vdb, _ = mods
    rng = np.random.default_rng(hash((name, dim)) % (1 << 31))
    n, nq = 6000 + int(rng.integers(0, 50)), 48
    base = _corpus(name, n, dim, rng)
    qs = np.concatenate([_corpus(name, nq // 3, dim, rng), _corpus("normal", nq // 3, dim, rng), base[: nq // 3] * np.float32(1.0)])
    ix = vdb.GpuIndex(dim, "l2sqr")
    ix.batch_add(base)
    keys, qsq, qoff, info = ix.flat_shortlist_keys(qs, 2)
    D = ((qs.astype(np.float64)[:, None, :] - base.astype(np.float64)[None, :, :]) ** 2).sum(-1)
    lb = keys.astype(np.float64) + qoff.astype(np.float64)[:, None]
    # the two roundings of the key's own evaluation are part of the certification's margin (flat_certify_lb, k_exact.hip):
    # 4 u (|x| + |q| + 2 |mu|)^2 -- restated here with the same norms
    xn = np.sqrt((base.astype(np.float64) ** 2).sum(1))
    qn = np.sqrt((qs.astype(np.float64) ** 2).sum(1))
    slack = 4 * 2.0 ** -24 * (xn[None, :] + qn[:, None] + 2 * info["xsq_max"]) ** 2 + 1e-9 * D
    bad = lb > D + slack
    assert not bad.any(), (name, dim, int(bad.sum()), float((lb - D)[bad].max()))
    # the bound is not vacuous: on average within a few percent of the distance where the data are not degenerate
    if name in ("gist", "normal"):
        far = D > 0.1 * D.mean()  # (a third of the queries are rows: D = 0 there)
        assert float(((D - lb)[far] / D[far]).mean()) < 0.08


@pytest.mark.parametrize("dim,n,nq", [(960, 40000, 200), (128, 50000, 130), (192, 30011, 97), (320, 20000, 70), (1024, 20000, 129), (2048, 17000, 40)])
def test_i8_pass_parity(mods, dim, n, nq):
    """KB (64-column k-blocks) = 15, 2, 3, 5, 16, 32: chunks of 3, 2, 3, 5, 2, 2; ragged last group; rows not a multiple of a unit; the widest
    dimension the pass takes (its query preparation then asks for 64 KB of dynamic LDS)"""
    vdb, O = mods
    if dim == 960:
        base, qs = gist_like(n, seed=41), gist_like(nq, seed=42)
    else:
        rng = np.random.default_rng(dim + 7)
        base = rng.standard_normal((n, dim)).astype(np.float32)
        qs = rng.standard_normal((nq, dim)).astype(np.float32)
    base[n - 1] = base[0]
    ix = vdb.GpuIndex(dim, "l2sqr")
    ix.batch_add(base)
    ix.set_flat_mode(2)
    idx, d, cnt = ix.flat_knn(qs, 10)
    assert ix.get_stat("flat_i8_valid") == 1 and ix.get_stat("flat_i8_queries") == nq
    redo = ix.get_stat("flat_i8_redo")
    if redo == 0:  # nothing needed the fp16 mirror yet: it has not been built (rows + norms + 1 B/element + 8 B of constants)
        assert ix.get_stat("flat_half_valid") == 0
        assert ix.get_stat("hbm_bytes_per_row") == dim * 4 + 4 + (dim + 63) // 64 * 64 + 8
    oi, od, oc = O.flat_knn_batch(base, qs, 10, 0, nthreads=8)
    _check_all(idx, d, cnt, oi, od, oc)
    print(f"dim {dim}: 8-bit pass passed on {redo} of {nq} queries")
    assert redo <= nq // 4
    # every kernel variant the dimension allows: the query group's image resident in LDS (res 0: dims up to 960; the chunk length is
    # then the depth of the row ring) and staged chunk by chunk through two buffers (res 1; per k-block or in one burst per chunk)
    for res, kc, burst in ((0, 5, 0), (0, 3, 0), (0, 2, 0), (1, 5, 0), (1, 3, 1), (1, 3, 2), (1, 2, 1), (1, 2, 2), (1, 0, 0)):
        ix.set_param("flat_gemm8_res", res)
        ix.set_param("flat_gemm8_kc", kc)
        ix.set_param("flat_gemm8_burst", burst)
        idx2, d2, cnt2 = ix.flat_knn(qs, 10)
        np.testing.assert_array_equal(idx, idx2)
        np.testing.assert_array_equal(d, d2)
    ix.set_param("flat_gemm8_res", 0)
    ix.set_param("flat_gemm8_kc", 0)
    ix.set_param("flat_gemm8_burst", 0)
    for nt in (1, 2):
        ix.set_param("flat_gemm8_nt", nt)
        idx2, d2, _ = ix.flat_knn(qs, 10)
        np.testing.assert_array_equal(idx, idx2)
        np.testing.assert_array_equal(d, d2)
    ix.set_param("flat_gemm8_nt", 0)
    for nw in (40, 41, 8, 4, 2, 1):  # exact stage: 4 waves with one chain per lane (40); 8 / 16 / 32 / 64 rows of a round per wave
        ix.set_param("flat_tail_lb_nw", nw)
        r0 = ix.get_stat("flat_i8_redo")
        idx2, d2, cnt2 = ix.flat_knn(qs, 10)
        np.testing.assert_array_equal(idx, idx2)
        np.testing.assert_array_equal(d, d2)
        np.testing.assert_array_equal(cnt, cnt2)
        assert ix.get_stat("flat_i8_redo") - r0 == redo  # the same queries close in the same rounds
        for k in (1, 64):
            idx3, d3, cnt3 = ix.flat_knn(qs[:40], k)
            oi3, od3, oc3 = O.flat_knn_batch(base, qs[:40], k, 0, nthreads=8)
            _check_all(idx3, d3, cnt3, oi3, od3, oc3)
    ix.set_param("flat_tail_lb_nw", 0)
    ix.set_param("flat_i8", 1)  # off: the fp16 pass answers (its mirror is built by this very call)
    q0 = ix.get_stat("flat_i8_queries")
    idx2, d2, _ = ix.flat_knn(qs, 10)
    assert ix.get_stat("flat_i8_queries") == q0
    if dim not in (320, 2048):  # (5 fp16 k-blocks / beyond the fp16 kernel's widths: no fp16 mirror for this dimension, the split-bf16 pass answers)
        assert ix.get_stat("flat_half_valid") == 1 and ix.get_stat("flat_half_queries") >= nq
    np.testing.assert_array_equal(idx, idx2)
    np.testing.assert_array_equal(d, d2)
    ix.set_param("flat_i8", 0)
    for k in (1, 33, 64):
        idx3, d3, cnt3 = ix.flat_knn(qs[:40], k)
        oi, od, oc = O.flat_knn_batch(base, qs[:40], k, 0, nthreads=8)
        _check_all(idx3, d3, cnt3, oi, od, oc)
    q0 = ix.get_stat("flat_i8_queries")
    idx3, d3, cnt3 = ix.flat_knn(qs[:20], 65)  # beyond the exact stage's 64 results: another tier
    assert ix.get_stat("flat_i8_queries") == q0
    oi, od, oc = O.flat_knn_batch(base, qs[:20], 65, 0, nthreads=8)
    _check_all(idx3, d3, cnt3, oi, od, oc)


def test_i8_pass_small_calls_and_rows_walked(mods):
    """calls of 1 .. 130 queries all take the pass; the number of rounds the exact stage may walk is a parameter"""
    vdb, O = mods
    n, dim = 30000, 960
    base, qs = gist_like(n, seed=5), gist_like(130, seed=6)
    ix = vdb.GpuIndex(dim, "l2sqr")
    ix.batch_add(base)
    ix.set_flat_mode(2)
    ix.set_param("flat_small", 1)
    oi, od, oc = O.flat_knn_batch(base, qs, 10, 0, nthreads=8)
    for nq in (1, 3, 64, 65, 128, 129, 130):
        q0 = ix.get_stat("flat_i8_queries")
        idx, d, cnt = ix.flat_knn(qs[:nq], 10)
        assert ix.get_stat("flat_i8_queries") == q0 + nq
        _check_all(idx, d, cnt, oi[:nq], od[:nq], oc[:nq])
    # one round of 63 rows only: more queries are passed on, the answers stay
    ix.set_param("flat_i8", 2)
    ix.set_param("flat_i8_rows", 64)
    r0 = ix.get_stat("flat_i8_redo")
    idx, d, cnt = ix.flat_knn(qs, 10)
    print("passed on with one round:", ix.get_stat("flat_i8_redo") - r0, "of", len(qs))
    _check_all(idx, d, cnt, oi, od, oc)
    ix.set_param("flat_i8_rows", 1024)
    idx, d, cnt = ix.flat_knn(qs, 10)
    _check_all(idx, d, cnt, oi, od, oc)


def test_i8_pass_redo_tiers(mods):
    """clusters of near-duplicates (margins far below the 8-bit bound) and exact duplicates across the cut: the pass cannot
    close them and hands the queries on; the fp16 / split-bf16 / exact tiers answer; auto mode switches the pass off"""
    vdb, O = mods
    rng = np.random.default_rng(77)
    dim, n, nq = 192, 30000, 140
    centers = rng.standard_normal((30, dim)).astype(np.float32)
    base = (centers[rng.integers(30, size=n)] + 1e-4 * rng.standard_normal((n, dim))).astype(np.float32)
    base[100:140] = base[99]  # 41 identical rows
    qs = (centers[rng.integers(30, size=nq)] + 1e-4 * rng.standard_normal((nq, dim))).astype(np.float32)
    qs[0] = base[99]
    ix = vdb.GpuIndex(dim, "l2sqr")
    ix.batch_add(base)
    ix.set_flat_mode(2)
    oi, od, oc = O.flat_knn_batch(base, qs, 10, 0, nthreads=8)
    # (i) the way a caller gets it: the first walk (252 rows) cannot close a cluster of ~1000 members; the second 8-bit attempt -- thresholds
    # from the k-th distances the first walk found, the whole list walked (k_redo.hip) -- does, and little is left for the other tiers
    idx, d, cnt = ix.flat_knn(qs, 10)
    _check_all(idx, d, cnt, oi, od, oc)
    second, left = ix.get_stat("flat_i8_second_queries"), ix.get_stat("flat_i8_redo")
    print("near-duplicate clusters: second attempt for", second, "of", nq, "; passed on to the fp16 tier:", left)
    assert second > nq // 2 and left == ix.get_stat("flat_i8_second_redo") and left <= nq // 8
    for _ in range(10):  # ... so auto mode keeps the pass for this index
        idx, d, cnt = ix.flat_knn(qs, 10)
        _check_all(idx, d, cnt, oi, od, oc)
    assert ix.get_stat("flat_i8_queries") == 11 * nq
    ix.close()
    # (ii) without the second attempt (flat_i8_second = 1): the fp16 / split-bf16 / exact tiers answer, and auto mode switches the pass off
    ix = vdb.GpuIndex(dim, "l2sqr")
    ix.batch_add(base)
    ix.set_flat_mode(2)
    ix.set_param("flat_i8_second", 1)
    idx, d, cnt = ix.flat_knn(qs, 10)
    _check_all(idx, d, cnt, oi, od, oc)
    redo = ix.get_stat("flat_i8_redo")
    print("near-duplicate clusters, no second attempt: passed on", redo, "of", nq)
    assert redo > nq // 2 and ix.get_stat("flat_i8_second_queries") == 0
    for _ in range(10):  # auto mode gives up on this index once 1/8 of >= 1024 queries were passed on
        idx, d, cnt = ix.flat_knn(qs, 10)
        _check_all(idx, d, cnt, oi, od, oc)
    q0 = ix.get_stat("flat_i8_queries")
    ix.flat_knn(qs, 10)
    assert ix.get_stat("flat_i8_queries") == q0
    ix.set_param("flat_i8", 2)  # forced: still right
    idx, d, cnt = ix.flat_knn(qs, 10)
    assert ix.get_stat("flat_i8_queries") == q0 + nq
    _check_all(idx, d, cnt, oi, od, oc)


def test_i8_mirror_upkeep(mods):
    """rows added after the first search (extension), a table that doubles (re-centring), swap_remove of rows in the middle,
    at the end and down to a ragged tile -- every state against the oracle"""
    vdb, O = mods
    dim = 320
    rng = np.random.default_rng(9)
    allrows = (rng.standard_normal((70000, dim)) + rng.standard_normal(dim) * 3).astype(np.float32)
    qs = (rng.standard_normal((70, dim)) + 1.5).astype(np.float32)
    ix = vdb.GpuIndex(dim, "l2sqr")
    ix.set_flat_mode(2)
    ix.set_param("flat_i8", 2)
    have = 0
    for upto in (20000, 20001, 20017, 33000, 70000):  # 33000 < 2 x 20000: extension; 70000: rebuild with a new centre
        ix.batch_add(allrows[have:upto])
        have = upto
        idx, d, cnt = ix.flat_knn(qs, 7)
        oi, od, oc = O.flat_knn_batch(allrows[:have], qs, 7, 0, nthreads=8)
        _check_all(idx, d, cnt, oi, od, oc)
    cur = allrows.copy()
    nn = have
    for victim in (5, nn - 2, 12345, 40000, 16 * 1000 + 15):
        ix.swap_remove(victim)
        cur[victim] = cur[nn - 1]
        nn -= 1
        if victim in (5, 40000):
            continue
        idx, d, cnt = ix.flat_knn(qs, 7)
        oi, od, oc = O.flat_knn_batch(cur[:nn], qs, 7, 0, nthreads=8)
        _check_all(idx, d, cnt, oi, od, oc)
    # removal right after an add (mirror behind the table): rebuilt by the next search
    ix.batch_add(allrows[:33])
    cur = np.concatenate([cur[:nn], allrows[:33]])
    nn += 33
    ix.swap_remove(3)
    cur[3] = cur[nn - 1]
    nn -= 1
    idx, d, cnt = ix.flat_knn(qs, 7)
    oi, od, oc = O.flat_knn_batch(cur[:nn], qs, 7, 0, nthreads=8)
    _check_all(idx, d, cnt, oi, od, oc)
    assert ix.get_stat("flat_i8_queries") > 0


def test_i8_degenerate_inputs(mods):
    """NaN / inf rows and queries, zero rows, zero queries, a constant table, tiny and huge magnitudes"""
    vdb, O = mods
    dim, n = 128, 20000
    rng = np.random.default_rng(3)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    base[7] = 0
    base[8, 5] = np.nan
    base[9, 6] = np.inf
    base[10] = 1e-20
    base[11] = 3e18
    base[12] = base[13]
    qs = rng.standard_normal((70, dim)).astype(np.float32)
    qs[0] = 0
    qs[1, 3] = np.nan
    qs[2, 4] = np.inf
    qs[3] = base[7]
    qs[4] = base[10]
    qs[5] = base[11]
    qs[6] = 1e-25
    ix = vdb.GpuIndex(dim, "l2sqr")
    ix.batch_add(base)
    ix.set_flat_mode(2)
    ix.set_param("flat_i8", 2)
    for k in (1, 10):
        idx, d, cnt = ix.flat_knn(qs, k)
        oi, od, oc = O.flat_knn_batch(base, qs, k, 0, nthreads=8)
        assert cnt.tolist() == oc.tolist()
        for q in range(len(qs)):
            assert idx[q].tolist() == oi[q].tolist(), (q, idx[q], oi[q])
            assert np.array_equal(d[q], od[q], equal_nan=True), (q, d[q], od[q])
    const = np.full((20000, dim), 0.25, dtype=np.float32)
    const[::7] += np.float32(1e-3)
    ix2 = vdb.GpuIndex(dim, "l2sqr")
    ix2.batch_add(const)
    ix2.set_flat_mode(2)
    ix2.set_param("flat_i8", 2)
    idx, d, cnt = ix2.flat_knn(qs[10:80], 5)
    oi, od, oc = O.flat_knn_batch(const, qs[10:80], 5, 0, nthreads=8)
    _check_all(idx, d, cnt, oi, od, oc)


def test_i8_not_for_u8(mods):
    """u8 tables keep their own path (Cosine takes the pass since round 4: tests/test_flat_i8_cosine_gpu.py)"""
    vdb, _ = mods
    rng = np.random.default_rng(1)
    base = rng.integers(0, 255, size=(20000, 128)).astype(np.uint8)
    ix = vdb.GpuIndex(128, "l2sqr", scalar="u8")
    ix.batch_add_u8(base)
    ix.set_flat_mode(2)
    ix.flat_knn(base[:70].astype(np.float32), 5)
    assert ix.get_stat("flat_i8_queries") == 0 and ix.get_stat("flat_i8_valid") == 0


@pytest.mark.parametrize("dim,n,nq", [(128, 130000, 1024), (960, 100000, 512), (192, 99000, 256 + 128)])
def test_i8_cooperative_sets(mods, dim, n, nq):
    """the resident filter kernel with the workgroups of an XCD in sets that share one row stream (8 / 4 groups per set; 3 groups: no
    sets): same hit lists, so the same answers as with the sets switched off, bit for bit, and as the oracle on a sample; the hit
    buffer is handed over in blocks (ragged last unit, rows not a multiple of a unit, a wave without units at the end of a slice)"""
    vdb, O = mods
    rng = np.random.default_rng(n + nq)
    if dim == 960:
        base, qs = gist_like(n, seed=77), gist_like(nq, seed=78)
    else:
        base = rng.standard_normal((n, dim)).astype(np.float32)
        qs = rng.standard_normal((nq, dim)).astype(np.float32)
    ix = vdb.GpuIndex(dim, "l2sqr")
    ix.batch_add(base)
    ix.set_flat_mode(2)
    ix.set_param("flat_gemm8_coop", 1)  # off
    idx0, d0, cnt0 = ix.flat_knn(qs, 10)
    assert ix.get_stat("flat_gemm8_coop_sets") <= 1  # the plain resident form ran
    r0 = ix.get_stat("flat_i8_redo")
    ix.set_param("flat_gemm8_coop", 0)  # auto: sets of gcd(groups, 8) workgroups
    idx1, d1, cnt1 = ix.flat_knn(qs, 10)
    groups = (nq + 127) // 128
    assert ix.get_stat("flat_gemm8_coop_sets") == (8 if groups % 8 == 0 else 4 if groups % 4 == 0 else 2 if groups % 2 == 0 else 0)
    assert ix.get_stat("flat_i8_queries") == 2 * nq
    np.testing.assert_array_equal(idx0, idx1)
    np.testing.assert_array_equal(d0, d1)
    np.testing.assert_array_equal(cnt0, cnt1)
    assert ix.get_stat("flat_i8_redo") - r0 == r0  # the same queries passed on
    assert r0 <= nq // 8
    sel = rng.choice(nq, 48, replace=False)
    oi, od, oc = O.flat_knn_batch(base, qs[sel], 10, 0, nthreads=8)
    _check_all(idx1[sel], d1[sel], cnt1[sel], oi, od, oc)
    for k in (1, 64):
        idx3, d3, cnt3 = ix.flat_knn(qs, k)
        oi, od, oc = O.flat_knn_batch(base, qs[sel[:16]], k, 0, nthreads=8)
        _check_all(idx3[sel[:16]], d3[sel[:16]], cnt3[sel[:16]], oi, od, oc)
    ix.close()


@pytest.mark.parametrize("dist,spread", [("l2sqr", 0.15), ("l2sqr", 0.6), ("cosine", 0.15)])
def test_i8_on_clustered_rows_and_the_auto_off_rule(mods, dist, spread):
    """tight Gaussian clusters (bench.py --data clustered): the k-th neighbour and hundreds of other cluster members lie within the 8-bit
    bound's gap of each other, so the first walk (252 rows) cannot close most queries; the second 8-bit attempt must (k_redo.hip), the answers
    must be the oracle's whichever tier gives them, and the auto-off rule must follow what left the 8-bit tier for good; looser clusters
    (0.6 sigma: still ~1900 members within a few gaps of each other) are reported and held to the same rule"""
    from conftest import gist_clustered

    vdb, O = mods
    n, dim, nq = 120_000, 960, 384
    # 64 clusters: ~1900 members each, far more than the 252 rows the exact stage may walk (1M rows in 1024 clusters: ~980)
    base = gist_clustered(n, dim=dim, seed=11, clusters=64, spread=spread)
    qs = gist_clustered(nq, dim=dim, seed=12, clusters=64, spread=spread)
    kind = O.L2SQR if dist == "l2sqr" else O.COSINE
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base)
    ix.set_param("flat_i8_stats", 1)
    ix.set_param("flat_i8_refine", 1)  # (this test is about the second attempt and the auto-off rule: the hit keys stay the 8-bit pass's)
    idx, d, cnt = ix.flat_knn(qs, 10)
    assert ix.get_stat("flat_i8_queries") == nq
    sel = np.arange(0, nq, 8)
    oi, od, oc = O.flat_knn_batch(base, qs[sel], 10, kind, nthreads=8)
    _check_all(idx[sel], d[sel], cnt[sel], oi, od, oc)
    redo, second = ix.get_stat("flat_i8_redo"), ix.get_stat("flat_i8_second_queries")
    hist = {r: ix.get_stat(f"flat_i8_rounds_{r}") for r in range(9)}
    print(f"{dist} spread {spread}: second attempt for {second} of {nq}, passed on to the fp16 tier {redo}; queries by rounds of the first walk "
          f"{hist}; hits per query mean {ix.get_stat('flat_i8_hits_sum') / nq:.0f} max {ix.get_stat('flat_i8_hits_max')}")
    assert sum(hist.values()) == nq
    if spread < 0.3:
        assert second > nq // 4  # tight clusters: the first walk cannot close them ...
        assert redo <= nq // 8   # ... the second attempt (thresholds from the first walk's k-th distances, whole list walked) does
    for _ in range(4):  # 5 x 384 queries: past the 1024 the auto rule wants to have seen
        idx2, d2, _ = ix.flat_knn(qs, 10)
        np.testing.assert_array_equal(idx, idx2)
        np.testing.assert_array_equal(d, d2)
    q0 = ix.get_stat("flat_i8_queries")
    idx2, d2, _ = ix.flat_knn(qs, 10)
    np.testing.assert_array_equal(idx, idx2)
    np.testing.assert_array_equal(d, d2)
    # the rule follows what LEFT the 8-bit tier: off once more than 1/8 of the (>= 1024) queries seen were passed on, on otherwise
    if ix.get_stat("flat_i8_redo") * 8 > 6 * nq:
        assert ix.get_stat("flat_i8_queries") == q0
    else:
        assert ix.get_stat("flat_i8_queries") == q0 + nq
    # without the second attempt (and without the fp16 refinement of the hit keys, which the long walks above have switched on by now: see
    # test_i8_hit_keys_refined_from_the_fp16_image) the same index hands (nearly) everything on and the answers stay
    ix.set_param("flat_i8", 2)
    ix.set_param("flat_i8_second", 1)
    ix.set_param("flat_i8_refine", 1)
    r0 = ix.get_stat("flat_i8_redo")
    idx2, d2, _ = ix.flat_knn(qs, 10)
    np.testing.assert_array_equal(idx, idx2)
    np.testing.assert_array_equal(d, d2)
    if spread < 0.3:
        assert ix.get_stat("flat_i8_redo") - r0 > nq // 4
    ix.close()


@pytest.mark.parametrize("dist", ["l2sqr", "cosine"])
def test_i8_threshold_sample_by_unit_minima(mods, dist):
    """the threshold sample hands the selection one value per (query, sampled unit) -- the unit's smallest key -- and the selection of a short
    sample is one wave per query (k_select_tau_tiny): production uses it from ~786k rows on (>= 16 x rank sampled units); forced here on 130k
    rows.  The threshold is >= the dense sample's (at least as many hits), the answers are the same bits and the oracle's."""
    vdb, O = mods
    n, dim, nq = 130000, 128, 300
    rng = np.random.default_rng(21)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    qs = rng.standard_normal((nq, dim)).astype(np.float32)
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base)
    ix.set_flat_mode(2)
    ix.set_param("flat_i8_stats", 1)
    ix.set_param("flat_i8_unit_min", 1)  # dense sample
    idx0, d0, cnt0 = ix.flat_knn(qs, 10)
    hits0 = ix.get_stat("flat_i8_hits_sum")
    ix.set_param("flat_i8_stats", 1)  # (resets the counters)
    ix.set_param("flat_i8_unit_min", 2)  # unit minima, forced
    idx1, d1, cnt1 = ix.flat_knn(qs, 10)
    hits1 = ix.get_stat("flat_i8_hits_sum")
    assert ix.get_stat("flat_i8_queries") == 2 * nq and ix.get_stat("flat_i8_redo") <= nq // 8
    np.testing.assert_array_equal(idx0, idx1)
    np.testing.assert_array_equal(d0, d1)
    np.testing.assert_array_equal(cnt0, cnt1)
    print(f"{dist}: hits per query dense sample {hits0 / nq:.0f}, unit minima {hits1 / nq:.0f}")
    assert hits1 >= hits0  # the r-th smallest unit minimum is never below the r-th smallest sampled key
    assert hits1 <= 3 * hits0
    sel = np.arange(0, nq, 6)
    oi, od, oc = O.flat_knn_batch(base, qs[sel], 10, O.L2SQR if dist == "l2sqr" else O.COSINE, nthreads=8)
    _check_all(idx1[sel], d1[sel], cnt1[sel], oi, od, oc)
    ix.close()


@pytest.mark.parametrize("dist", ["l2sqr", "cosine"])
def test_i8_hit_keys_refined_from_the_fp16_image(mods, dist):
    """tight clusters: the 8-bit keys' worst-case slack (a dot-product bound: ~2 |dx||q|) is wider than the spread of a cluster's distances, so
    the walk evaluates hundreds of rows in key order; k_flat_refine_half (k_redo.hip) replaces every hit's key by the larger lower bound the
    row-major fp16 image gives in DIFFERENCE form (sqrt(D) >= |x~ - q| - |dx_r|; Cosine: the same for the unit vectors) before the walk.
    Same answers as without and as the oracle, far fewer rounds; the auto rule turns it on from the rounds the walks take and leaves it off on
    separable rows; rows and queries whose norms are not plain numbers keep their keys."""
    from conftest import gist_clustered

    vdb, O = mods
    n, dim, nq = 120_000, 960, 384
    base = gist_clustered(n, dim=dim, seed=21, clusters=64, spread=0.15)
    qs = gist_clustered(nq, dim=dim, seed=22, clusters=64, spread=0.15)
    base[777] = 0.0          # a zero row (Cosine: clamp active: its key is -FLT_MAX and stays) ...
    if dist == "l2sqr":      # ... and one whose squared norm underflows (a Cosine index with such a row certifies nothing on the 8-bit pass:
        base[778] = base[5] * np.float32(1e-20)  # flat_certify_lb checks the index's smallest positive norm)
    qs[3] = 0.0              # a zero query
    kind = O.L2SQR if dist == "l2sqr" else O.COSINE
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base)
    ix.set_param("flat_i8_stats", 1)

    def rounds():
        return sum(min(r, 8) * ix.get_stat(f"flat_i8_rounds_{r}") for r in range(9))

    ix.set_param("flat_i8_refine", 1)  # off
    a = ix.flat_knn(qs, 10)
    r_off = rounds()
    assert ix.get_stat("flat_i8_refine_queries") == 0
    ix.set_param("flat_i8_refine", 2)  # always
    b = ix.flat_knn(qs, 10)
    r_on = rounds() - r_off
    assert ix.get_stat("flat_i8_refine_queries") == nq
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    sel = np.arange(0, nq, 6)
    oi, od, oc = O.flat_knn_batch(base, qs[sel], 10, kind, nthreads=8)
    _check_all(b[0][sel], b[1][sel], b[2][sel], oi, od, oc)
    print(f"{dist}: rounds of the first walk (capped at 8 per query) without / with refined keys: {r_off} / {r_on} for {nq} queries")
    assert r_on * 2 < r_off
    # auto: the first call sees the long walks, the next ones run refined; every 32nd call of the on state is a probe without
    ix.set_param("flat_i8_refine", 0)
    c = ix.flat_knn(qs, 10)
    assert ix.get_stat("flat_i8_refine_on") == 1 and ix.get_stat("flat_i8_refine_queries") == nq
    c = ix.flat_knn(qs, 10)
    assert ix.get_stat("flat_i8_refine_queries") == 2 * nq
    for x, y in zip(a, c):
        np.testing.assert_array_equal(x, y)
    ix.close()
    # separable rows: never on
    base2, qs2 = gist_like(100_000, dim=dim, seed=23), gist_like(128, dim=dim, seed=24)
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base2)
    for _ in range(3):
        g = ix.flat_knn(qs2, 10)
    assert ix.get_stat("flat_i8_refine_on") == 0 and ix.get_stat("flat_i8_refine_queries") == 0
    oi, od, oc = O.flat_knn_batch(base2, qs2[:32], 10, kind, nthreads=8)
    _check_all(g[0][:32], g[1][:32], g[2][:32], oi, od, oc)
    ix.close()
