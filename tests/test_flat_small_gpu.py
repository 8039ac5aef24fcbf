"""GPU: the one-launch Flat search of small tables (k_small.hip, the db.search() shape: pyo3/mod.rs:199-214 ->
flat_index.rs:48-57) against the oracle, bit for bit: every rows-per-workgroup geometry (16 / 32 / 64), chunk remainders, ragged
last workgroups, k > len, ties, NaN rows, both metrics, host pointers (pinned block, queries read over PCIe or staged) and
device pointers, and equality with the general exact path (flat_small = 1)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import lab_1806_vec_db_amd as vdb
    from oracle import oracle as O
    return vdb, O


def _check(ix, O, base, qs, k, kind):
    idx, d, cnt = ix.flat_knn(qs, k)
    oi, od, oc = O.flat_knn_batch(base, qs, k, kind, nthreads=8)
    for q in range(len(qs)):
        c = int(cnt[q])
        assert c == int(oc[q]) == min(k, len(base))
        assert idx[q, :c].tolist() == oi[q][:c].tolist(), (q, idx[q, :c], oi[q][:c])
        assert np.array_equal(d[q, :c], od[q][:c], equal_nan=True)
        assert not idx[q, c:].any() and not d[q, c:].any()  # slots beyond min(k, len) are defined (zero)


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
@pytest.mark.parametrize("n,dim", [(1, 4), (5, 12), (16, 960), (17, 100), (63, 128), (64, 516), (65, 1024), (1000, 960), (4097, 96),
                                   (8193, 64), (9000, 260), (16384, 32), (16385, 48), (3000, 2052)])
def test_small_tables_match_oracle(mods, n, dim, dist, kind):
    vdb, O = mods
    rng = np.random.default_rng(n * 31 + dim)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    if n > 40:
        base[n - 7:] = base[:7]  # exact ties, far apart in the table
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base)
    ix.set_param("flat_small", 2)  # also beyond the auto threshold (16385 rows)
    for nq in (1, 3, 31):
        qs = (base[rng.integers(0, n, nq)] + 0.1 * rng.standard_normal((nq, dim))).astype(np.float32)
        qs[0] = base[0]  # distance 0 / cosine ~0 and a tie with its duplicate
        for k in (1, 10, 64):
            _check(ix, O, base, qs, k, kind)
    assert ix.prof_get("flat_small")["launches"] == 0  # (profiling is off by default)
    ix.prof_enable(True)
    ix.flat_knn(base[:1], 3)
    assert ix.prof_get("flat_small")["launches"] == 1 and ix.prof_get("flat_exact")["launches"] == 0
    ix.close()


def test_small_equals_general_exact_path_and_auto_rule(mods):
    vdb, O = mods
    rng = np.random.default_rng(5)
    n, dim = 2500, 200
    base = np.round(np.abs(rng.normal(0.07, 0.045, (n, dim))), 4).astype(np.float32)
    qs = np.round(np.abs(rng.normal(0.07, 0.045, (40, dim))), 4).astype(np.float32)
    ix = vdb.GpuIndex(dim, "l2sqr")
    ix.batch_add(base)
    ix.prof_enable(True)
    a = ix.flat_knn(qs[:9], 10)          # auto: small table, < 32 queries -> k_flat_small
    assert ix.prof_get("flat_small")["launches"] == 1
    ix.set_param("flat_small", 1)
    b = ix.flat_knn(qs[:9], 10)
    assert ix.prof_get("flat_small")["launches"] == 1 and ix.prof_get("flat_exact")["launches"] >= 1
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    ix.set_param("flat_small", 0)
    ix.flat_knn(qs, 10)                  # 40 queries: the many-queries path, not this kernel
    ix.flat_knn(qs[:4], 100)             # k > 64: general path
    assert ix.prof_get("flat_small")["launches"] == 1
    ix.set_flat_mode(1)                  # exact scan requested explicitly: the scan kernels
    ix.flat_knn(qs[:4], 10)
    assert ix.prof_get("flat_small")["launches"] == 1
    ix.close()
    # beyond 16 384 rows the kernel competes with the MFMA pipeline: one query on 40 000 rows takes it, six queries do not
    big = np.round(np.abs(rng.normal(0.07, 0.045, (40000, 64))), 4).astype(np.float32)
    jx = vdb.GpuIndex(64, "l2sqr")
    jx.batch_add(big)
    jx.prof_enable(True)
    a = jx.flat_knn(big[:1] + np.float32(0.001), 10)
    assert jx.prof_get("flat_small")["launches"] == 1
    b6 = jx.flat_knn(big[:6] + np.float32(0.001), 10)
    assert jx.prof_get("flat_small")["launches"] == 1 and jx.get_stat("flat_i8_queries") + jx.get_stat("flat_half_queries") + jx.prof_get("flat_mfma")["launches"] > 0
    assert np.array_equal(a[0][0], b6[0][0]) and np.array_equal(a[1][0], b6[1][0])
    jx.close()


def test_small_nan_rows_and_swap_remove_and_offset(mods):
    vdb, O = mods
    rng = np.random.default_rng(9)
    n, dim = 300, 64
    base = rng.standard_normal((n, dim)).astype(np.float32)
    base[17, 3] = np.nan
    base[200] = 0.0  # zero row: cosine denominator clamp (distance/mod.rs:60-69)
    qs = rng.standard_normal((2, dim)).astype(np.float32)
    for dist, kind in (("l2sqr", 0), ("cosine", 1)):
        ix = vdb.GpuIndex(dim, dist)
        ix.batch_add(base)
        _check(ix, O, base, qs, 64, kind)
        jx = vdb.GpuIndex(dim, dist)  # k > len: every row comes back, the NaN distance last (ordered-float: NaN greatest)
        jx.batch_add(base[:40])
        _check(jx, O, base[:40], qs, 64, kind)
        jx.close()
        ix.swap_remove(5)  # vec_set.rs:131-137: the last row moves into slot 5
        b2 = base.copy()
        b2[5] = b2[-1]
        b2 = b2[:-1]
        _check(ix, O, b2, qs, 10, kind)
        ix.set_id_offset(1000)
        idx, d, cnt = ix.flat_knn(qs, 5)
        oi, od, oc = O.flat_knn_batch(b2, qs, 5, kind, nthreads=2)
        assert (idx - 1000).tolist() == [o[:5].tolist() for o in oi]
        ix.close()


def test_small_device_pointers(mods):
    torch = pytest.importorskip("torch")
    vdb, O = mods
    rng = np.random.default_rng(11)
    n, dim, nq, k = 1500, 960, 5, 10
    base = np.round(np.abs(rng.normal(0.07, 0.045, (n, dim))), 4).astype(np.float32)
    qs = np.round(np.abs(rng.normal(0.07, 0.045, (nq, dim))), 4).astype(np.float32)
    ix = vdb.GpuIndex(dim, "l2sqr")
    ix.batch_add(base)
    dq = torch.from_numpy(qs).cuda()
    o_i = torch.zeros((nq, k), dtype=torch.int64, device="cuda")
    o_d = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
    o_c = torch.zeros((nq,), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):  # the arrival counters are left zero by every launch
        ix.flat_knn_device(dq.data_ptr(), nq, k, o_i.data_ptr(), o_d.data_ptr(), o_c.data_ptr())
    oi, od, oc = O.flat_knn_batch(base, qs, k, 0, nthreads=4)
    assert o_i.cpu().numpy().tolist() == [o.tolist() for o in oi]
    assert np.array_equal(o_d.cpu().numpy(), np.stack(od))
    assert o_c.cpu().tolist() == [k] * nq
    ix.close()
