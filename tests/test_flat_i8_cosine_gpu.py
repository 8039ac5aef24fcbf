"""GPU parity: the 8-bit first pass of Flat COSINE searches (the reference's default metric: pyo3/mod.rs:73, distance/mod.rs:60-69).

1 - <x, q> / (|x||q|) = |x/|x| - q/|q||^2 / 2, so the pass is the L2Sqr construction on UNIT rows and UNIT queries (k_i8.hip, COS):
key(r, q) + O_q <= 2 (1 - cos(x_r, q)) for every row, and the exact stage (k_flat_tail_lb<FOLD_DOT>) certifies against half of it minus the
rounding of the reference's own f32 evaluation.  Tested: (1) the bound for every (row, query) pair of six corpora against float64; (2)
bit-equality of whole searches with the oracle through every kernel variant; (3) mirror upkeep; (4) degenerate norms (zero rows and
queries, norms that overflow / underflow the reference's f32 fold, the 1e-10 clamp, NaN / inf); (5) the cooperative sets.
"""
import numpy as np
import pytest

from conftest import gist_like
from test_flat_i8_gpu import _check_all, _corpus

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import lab_1806_vec_db_amd as vdb
    from oracle import oracle as O
    return vdb, O


@pytest.mark.parametrize("name", ["gist", "normal", "offset", "decades", "sparse", "integers"])
@pytest.mark.parametrize("dim", [960, 192, 128, 320])
def test_cosine_keys_are_lower_bounds(mods, name, dim):
    """(key + O_q) / 2 <= 1 - cos for EVERY (row, query) pair: queries of the corpus' distribution, of another one, rows themselves"""
    vdb, _ = mods
    rng = np.random.default_rng(hash((name, dim, "cos")) % (1 << 31))
    n, nq = 6000 + int(rng.integers(0, 50)), 48
    base = _corpus(name, n, dim, rng)
    if name == "integers":
        base[(base == 0).all(1)] = 1.0  # (zero rows are a degenerate case of their own: test below)
    qs = np.concatenate([_corpus(name, nq // 3, dim, rng), _corpus("normal", nq // 3, dim, rng), base[: nq // 3] * np.float32(1.0)])
    ix = vdb.GpuIndex(dim, "cosine")
    ix.batch_add(base)
    keys, qsq, qoff, info = ix.flat_shortlist_keys(qs, 2)
    b64, q64 = base.astype(np.float64), qs.astype(np.float64)
    bu = b64 / np.sqrt((b64 ** 2).sum(1))[:, None]
    qu = q64 / np.sqrt((q64 ** 2).sum(1))[:, None]
    D = 0.5 * ((qu[:, None, :] - bu[None, :, :]) ** 2).sum(-1)  # = 1 - cos, without the cancellation of 1 - <.,.> for near-parallel pairs
    lb = 0.5 * (keys.astype(np.float64) + qoff.astype(np.float64)[:, None])
    # the key's own two roundings are part of the certification's margin (flat_certify_lb: 2 u (2 + 2 |mu|)^2), restated here
    slack = 2 * 2.0 ** -24 * (2 + 2 * info["xsq_max"]) ** 2 + 1e-9 * D
    bad = lb > D + slack
    assert not bad.any(), (name, dim, int(bad.sum()), float((lb - D)[bad].max()))
    if name in ("gist", "normal"):  # not vacuous
        far = D > 0.1 * D.mean()
        assert float(((D - lb)[far] / D[far]).mean()) < 0.08
    ix.close()


@pytest.mark.parametrize("dim,n,nq", [(960, 40000, 200), (128, 50000, 130), (192, 30011, 97), (320, 20000, 70), (1024, 20000, 129), (1536, 17000, 70)])
def test_cosine_i8_pass_parity(mods, dim, n, nq):
    vdb, O = mods
    if dim == 960:
        base, qs = gist_like(n, seed=41), gist_like(nq, seed=42)
    else:
        rng = np.random.default_rng(dim + 11)
        base = (rng.standard_normal((n, dim)) * np.exp(rng.uniform(-2, 2, size=(n, 1)))).astype(np.float32)  # norms over two decades
        qs = rng.standard_normal((nq, dim)).astype(np.float32)
    base[n - 1] = base[0]
    base[n - 2] = base[0] * np.float32(3.0)  # same direction, another norm: a tie or a near-tie of the cosine
    ix = vdb.GpuIndex(dim, "cosine")
    ix.batch_add(base)
    ix.set_flat_mode(2)
    idx, d, cnt = ix.flat_knn(qs, 10)
    assert ix.get_stat("flat_i8_valid") == 1 and ix.get_stat("flat_i8_queries") == nq
    redo = ix.get_stat("flat_i8_redo")
    oi, od, oc = O.flat_knn_batch(base, qs, 10, O.COSINE, nthreads=8)
    _check_all(idx, d, cnt, oi, od, oc)
    print(f"cosine dim {dim}: 8-bit pass passed on {redo} of {nq} queries")
    assert redo <= nq // 4
    for res, kc, burst in ((0, 5, 0), (0, 3, 0), (0, 2, 0), (1, 3, 2), (1, 2, 1), (1, 0, 0)):
        ix.set_param("flat_gemm8_res", res)
        ix.set_param("flat_gemm8_kc", kc)
        ix.set_param("flat_gemm8_burst", burst)
        idx2, d2, cnt2 = ix.flat_knn(qs, 10)
        np.testing.assert_array_equal(idx, idx2)
        np.testing.assert_array_equal(d, d2)
    ix.set_param("flat_gemm8_res", 0)
    ix.set_param("flat_gemm8_kc", 0)
    ix.set_param("flat_gemm8_burst", 0)
    for nw in (40, 41, 8):
        ix.set_param("flat_tail_lb_nw", nw)
        r0 = ix.get_stat("flat_i8_redo")
        idx2, d2, cnt2 = ix.flat_knn(qs, 10)
        np.testing.assert_array_equal(idx, idx2)
        np.testing.assert_array_equal(d, d2)
        np.testing.assert_array_equal(cnt, cnt2)
        assert ix.get_stat("flat_i8_redo") - r0 == redo
        for k in (1, 64):
            idx3, d3, cnt3 = ix.flat_knn(qs[:40], k)
            oi3, od3, oc3 = O.flat_knn_batch(base, qs[:40], k, O.COSINE, nthreads=8)
            _check_all(idx3, d3, cnt3, oi3, od3, oc3)
    ix.set_param("flat_tail_lb_nw", 0)
    for nqc in (1, 3, 65):  # small calls take the pass too
        q0 = ix.get_stat("flat_i8_queries")
        idx3, d3, cnt3 = ix.flat_knn(qs[:nqc], 10)
        assert ix.get_stat("flat_i8_queries") == q0 + nqc
        _check_all(idx3, d3, cnt3, oi[:nqc], od[:nqc], oc[:nqc])
    ix.set_param("flat_i8", 1)  # off: the fp16 pass answers the same
    q0 = ix.get_stat("flat_i8_queries")
    idx2, d2, _ = ix.flat_knn(qs, 10)
    assert ix.get_stat("flat_i8_queries") == q0
    np.testing.assert_array_equal(idx, idx2)
    np.testing.assert_array_equal(d, d2)
    ix.close()


def test_cosine_i8_degenerate_norms(mods):
    """zero rows (distance exactly 1), rows whose |x|^2 overflows / underflows the reference's f32 fold, tiny norms under the 1e-10
    clamp, NaN / inf rows, and the same kinds of queries: whatever tier ends up answering, the oracle's bits"""
    vdb, O = mods
    dim, n = 128, 20000
    rng = np.random.default_rng(5)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    base[7] = 0
    base[8, 5] = np.nan
    base[9, 6] = np.inf
    base[12] = base[13]
    base[14] = base[13] * np.float32(0.5)
    qs = rng.standard_normal((70, dim)).astype(np.float32)
    qs[3] = base[13]
    ix = vdb.GpuIndex(dim, "cosine")
    ix.batch_add(base)
    ix.set_flat_mode(2)
    ix.set_param("flat_i8", 2)
    for k in (1, 10):
        idx, d, cnt = ix.flat_knn(qs, k)
        oi, od, oc = O.flat_knn_batch(base, qs, k, O.COSINE, nthreads=8)
        _check_all(idx, d, cnt, oi, od, oc)
    assert ix.get_stat("flat_i8_redo") <= 8  # ordinary queries close in the pass although the table holds zero / NaN / inf rows
    # queries the real-number cosine does not describe: handed on, answered by the other tiers
    qd = qs[:8].copy()
    qd[0] = 0
    qd[1, 3] = np.nan
    qd[2, 4] = np.inf
    qd[4] = 1e-25
    qd[5] = 3e18
    qd[6] = 1e-12
    for k in (1, 10):
        idx, d, cnt = ix.flat_knn(qd, k)
        oi, od, oc = O.flat_knn_batch(base, qd, k, O.COSINE, nthreads=8)
        assert cnt.tolist() == oc.tolist()
        for q in range(len(qd)):
            assert idx[q].tolist() == oi[q].tolist(), (q, idx[q], oi[q])
            assert np.array_equal(d[q], od[q], equal_nan=True), (q, d[q], od[q])
    ix.close()
    # rows with norms the f32 fold cannot hold: always evaluated exactly (-FLT_MAX keys); tiny rows put the clamp in play -> nothing is
    # certified by the pass, the answers stay
    base2 = base.copy()
    base2[8] = base[20]
    base2[9] = base[21]
    base2[10] = 1e-20
    base2[11] = 3e18
    base2[15] = base[13] * np.float32(1e-18)
    ix2 = vdb.GpuIndex(dim, "cosine")
    ix2.batch_add(base2)
    ix2.set_flat_mode(2)
    ix2.set_param("flat_i8", 2)
    for k in (1, 10):
        idx, d, cnt = ix2.flat_knn(qs, k)
        oi, od, oc = O.flat_knn_batch(base2, qs, k, O.COSINE, nthreads=8)
        assert cnt.tolist() == oc.tolist()
        for q in range(len(qs)):
            assert idx[q].tolist() == oi[q].tolist(), (q, idx[q], oi[q])
            assert np.array_equal(d[q], od[q], equal_nan=True), (q, d[q], od[q])
    ix2.close()
    # the overflow row alone (no tiny norms): the pass certifies, the row is in every hit list
    base3 = base.copy()
    base3[8] = base[20]
    base3[9] = base[21]
    base3[11] = 3e18
    ix3 = vdb.GpuIndex(dim, "cosine")
    ix3.batch_add(base3)
    ix3.set_flat_mode(2)
    ix3.set_param("flat_i8", 2)
    idx, d, cnt = ix3.flat_knn(qs, 10)
    oi, od, oc = O.flat_knn_batch(base3, qs, 10, O.COSINE, nthreads=8)
    _check_all(idx, d, cnt, oi, od, oc)
    assert ix3.get_stat("flat_i8_redo") <= 8
    ix3.close()


def test_cosine_i8_mirror_upkeep(mods):
    vdb, O = mods
    dim = 320
    rng = np.random.default_rng(19)
    allrows = (rng.standard_normal((70000, dim)) + rng.standard_normal(dim) * 3).astype(np.float32)
    qs = (rng.standard_normal((70, dim)) + 1.5).astype(np.float32)
    ix = vdb.GpuIndex(dim, "cosine")
    ix.set_flat_mode(2)
    ix.set_param("flat_i8", 2)
    have = 0
    for upto in (20000, 20017, 33000, 70000):
        ix.batch_add(allrows[have:upto])
        have = upto
        idx, d, cnt = ix.flat_knn(qs, 7)
        oi, od, oc = O.flat_knn_batch(allrows[:have], qs, 7, O.COSINE, nthreads=8)
        _check_all(idx, d, cnt, oi, od, oc)
    cur = allrows.copy()
    nn = have
    for victim in (5, nn - 2, 12345, 16 * 1000 + 15):
        ix.swap_remove(victim)
        cur[victim] = cur[nn - 1]
        nn -= 1
        idx, d, cnt = ix.flat_knn(qs, 7)
        oi, od, oc = O.flat_knn_batch(cur[:nn], qs, 7, O.COSINE, nthreads=8)
        _check_all(idx, d, cnt, oi, od, oc)
    assert ix.get_stat("flat_i8_queries") > 0
    ix.close()


@pytest.mark.parametrize("dim,n,nq", [(960, 100000, 512), (128, 130000, 1024)])
def test_cosine_i8_cooperative_sets(mods, dim, n, nq):
    vdb, O = mods
    rng = np.random.default_rng(n + nq + 1)
    if dim == 960:
        base, qs = gist_like(n, seed=77), gist_like(nq, seed=78)
    else:
        base = rng.standard_normal((n, dim)).astype(np.float32)
        qs = rng.standard_normal((nq, dim)).astype(np.float32)
    ix = vdb.GpuIndex(dim, "cosine")
    ix.batch_add(base)
    ix.set_flat_mode(2)
    ix.set_param("flat_gemm8_coop", 1)  # off
    idx0, d0, cnt0 = ix.flat_knn(qs, 10)
    assert ix.get_stat("flat_gemm8_coop_sets") <= 1
    ix.set_param("flat_gemm8_coop", 0)
    idx1, d1, cnt1 = ix.flat_knn(qs, 10)
    assert ix.get_stat("flat_gemm8_coop_sets") == (8 if (nq // 128) % 8 == 0 else 4)
    assert ix.get_stat("flat_i8_queries") == 2 * nq
    np.testing.assert_array_equal(idx0, idx1)
    np.testing.assert_array_equal(d0, d1)
    np.testing.assert_array_equal(cnt0, cnt1)
    assert ix.get_stat("flat_i8_redo") <= nq // 4
    sel = rng.choice(nq, 48, replace=False)
    oi, od, oc = O.flat_knn_batch(base, qs[sel], 10, O.COSINE, nthreads=8)
    _check_all(idx1[sel], d1[sel], cnt1[sel], oi, od, oc)
    ix.close()
