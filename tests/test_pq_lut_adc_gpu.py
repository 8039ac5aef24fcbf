"""GPU parity, directly on the intermediate values of the PQ path:

  a11  PQTable::create_lookup (pq_table.rs:195-224): the lookup table k_pq_lut builds, bit for bit against the
       oracle's orc_pq_lookup, and PQLookupTable::dist_cache (0 / |q|);
  a12  the ADC adapter (pq_table.rs:239-301): the value of EVERY code row as the scan kernels compute it (dense
       mode) against orc_pq_adc_all, and the fused threshold-filter path through what it exports -- the shard key
       rows of vdb_flat_knn_pq_shard: the max(ef,k) smallest (ADC value, global id) pair keys with the exact
       distance of the same row beside each.

A compensating error pair (a wrong table entry and a wrong sum that cancel in the final top-k) would pass the
end-to-end knn_pq tests; it cannot pass these.
"""
import numpy as np
import pytest

from conftest import gist_like

pytestmark = pytest.mark.gpu

NONE = np.uint64(0xFFFFFFFFFFFFFFFF)


@pytest.fixture(scope="module")
def mods():
    import lab_1806_vec_db_amd as vdb
    from oracle import oracle as O
    return vdb, O


def _build(vdb, O, base, dist, kind, n_bits, m, seed=5):
    ix = vdb.GpuIndex(base.shape[1], dist)
    ix.batch_add(base)
    ix.pq_build(n_bits=n_bits, m=m, train_n=min(len(base), 400), max_iter=4, seed=seed)
    pq = ix.pq_export()
    opq = O.PQ.from_centroids(ix.dim, pq["m"], pq["n_bits"], kind, pq["centroids"])
    opq.set_codes(pq["codes"])
    return ix, opq


CASES = [  # dim, m, n_bits: the Gist1M table shape, 4-dim groups, odd m (last byte half used), uneven groups, 8-bit in / out of LDS
    (960, 320, 4), (960, 240, 4), (96, 31, 4), (100, 7, 4), (64, 64, 4), (96, 24, 8), (960, 320, 8), (13, 5, 8),
]


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
@pytest.mark.parametrize("dim,m,n_bits", CASES)
def test_lookup_table_and_adc_values_bit_exact(mods, dist, kind, dim, m, n_bits):
    vdb, O = mods
    n = 3000 if dim >= 960 else 5000
    base = gist_like(n, dim=dim, seed=21)
    if kind == 1:
        base[17] = 0.0  # a zero row: cosine ADC denominators hit the 1e-10 clamp only through the query side, but its code row must still score like the oracle's
    qs = gist_like(9, dim=dim, seed=22)
    qs[3] = -qs[3]
    qs[4] = 0.0        # zero query: L2 table = |c|^2, cosine ADC = 1 - 0 / max(.., 1e-10)
    ix, opq = _build(vdb, O, base, dist, kind, n_bits, m)
    lut, qc = ix.pq_create_lookup(qs)
    assert lut.shape == (9, m << n_bits)
    adc = ix.pq_adc_all(qs)
    assert adc.shape == (9, n)
    for q in range(9):
        olut, oqc = opq.lookup(qs[q])
        assert np.array_equal(lut[q].view(np.uint32), olut.view(np.uint32)), (q, "lookup table differs from orc_pq_lookup")
        assert np.float32(qc[q]).view(np.uint32) == np.float32(oqc).view(np.uint32), (q, qc[q], oqc)
        oadc = opq.adc_all(qs[q], n)
        same = (adc[q].view(np.uint32) == oadc.view(np.uint32)) | (np.isnan(adc[q]) & np.isnan(oadc))
        assert same.all(), (q, np.flatnonzero(~same)[:5], adc[q][~same][:5], oadc[~same][:5])


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
@pytest.mark.parametrize("n,dim,m,n_bits,ef", [(70000, 96, 32, 4, 100), (66000, 960, 320, 4, 64), (5000, 64, 16, 8, 200), (80000, 64, 64, 4, 1000)])
def test_shard_key_rows_equal_oracle_adc(mods, dist, kind, n, dim, m, n_bits, ef):
    """vdb_flat_knn_pq_shard = what a shard hands to the all-gather; n >= 65536 goes through the sampled-threshold filter
    scan (k_pq_adc MODE 1), the small case through the dense scan.  Duplicated rows give equal codes -> ADC ties that the
    (value, id) order must break by id."""
    vdb, O = mods
    base = gist_like(n, dim=dim, seed=31)
    base[n // 2:n // 2 + 40] = base[:40]
    qs = gist_like(6, dim=dim, seed=32)
    qs[5] = base[7]
    ix, opq = _build(vdb, O, base, dist, kind, n_bits, m)
    off = 1000
    ix.set_id_offset(off)
    k = 10
    a, e = ix.knn_pq_shard(qs, k, ef)
    efk = max(ef, k)
    for q in range(len(qs)):
        oadc = opq.adc_all(qs[q], n)
        ok = np.sort(O.pair_keys(oadc, np.arange(off, off + n, dtype=np.uint64)))[:efk]
        assert np.array_equal(a[q], ok), (q, np.flatnonzero(a[q] != ok)[:5])
        sel = (ok & np.uint64(0xFFFFFFFF)).astype(np.int64) - off
        ex = np.array([O.dist(kind, base[i], qs[q]) for i in sel], dtype=np.float32)
        assert np.array_equal(e[q], O.pair_keys(ex, (sel + off).astype(np.uint64))), q


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
def test_quantised_scan_fallbacks_on_degenerate_queries(mods, dist, kind):
    """The 16-bit first pass of the ADC scan (k_pq_adc16) only handles tables of finite non-negative entries; queries whose
    table holds a NaN / inf (NaN or huge query components), constant tables (all centroids of every group equal -> step
    D = 1) and thresholds of +inf must take the f32 scan per query -- while the ordinary queries of the same call stay on
    the fast path.  Answers equal the oracle's either way."""
    vdb, O = mods
    n, dim, m = 70000, 64, 16
    base = gist_like(n, dim=dim, seed=41)
    ix, opq = _build(vdb, O, base, dist, kind, 4, m)
    qs = gist_like(12, dim=dim, seed=42)
    qs[1, 5] = np.nan
    qs[2, :] = 0.0
    qs[3, 7] = 3.0e19          # (x - c)^2 overflows to +inf in the table
    qs[4, :] = 1.0e-30         # denormal-scale products
    qs[5] = base[123]
    qs[6, 9] = -np.inf
    qs[8] = -base[77]          # Cosine: every row lies in the opposite half-space (tau >= 1: the f32 scan answers)
    qs[9] = base[5] * 1e-25    # Cosine: |q| so small that the reference's 1e-10 clamp acts
    for k, ef in ((10, 100), (5, 1000)):
        idx, d, cnt = ix.knn_pq(qs, k, ef)
        for q in range(len(qs)):
            oi, od = O.flat_knn_pq(base, opq, qs[q], k, ef, kind)
            assert idx[q, :len(oi)].tolist() == oi.tolist(), (k, ef, q)
            assert np.array_equal(d[q, :len(od)], od, equal_nan=True), (k, ef, q)
    # the A/B switch: same answers with the quantised pass off
    a = ix.knn_pq(qs[7:], 10, 64)
    try:
        ix.set_param("pq_adc16", 1)
        b = ix.knn_pq(qs[7:], 10, 64)
    finally:
        ix.set_param("pq_adc16", 0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_quantised_threshold_sample_keeps_results():
    """The threshold sample of the quantised ADC scan on the quantised tables themselves (k_pq_adc16<.., SAMPLE>, tau = M + D (s* +
    m/2)) against the exact f32 sample: the threshold only decides how many rows the scan keeps (the count is checked), so the
    answers must be identical -- and equal the oracle's (pq_table.rs:239-301, flat_index.rs:84-104)."""
    import lab_1806_vec_db_amd as vdb
    from oracle import oracle as O
    from conftest import gist_like

    n, dim, m = 140000, 64, 32
    rng = np.random.default_rng(8)
    base = gist_like(n, dim=dim, seed=33)
    base[70000:70020] = base[:20]  # ADC ties across the table
    qs = gist_like(70, dim=dim, seed=34)
    ix = vdb.GpuIndex(dim, "l2sqr")
    ix.batch_add(base)
    ix.pq_build(n_bits=4, m=m, train_n=3000, max_iter=4, seed=2)
    pq = ix.pq_export()
    opq = O.PQ.from_centroids(dim, m, 4, 0, pq["centroids"])
    opq.set_codes(pq["codes"])
    res = {}
    try:
        for v in (0, 1):
            ix.set_param("pq_sample16", v)
            for ef in (10, 100, 700):
                res[(v, ef)] = ix.knn_pq(qs, 10, ef)
    finally:
        ix.set_param("pq_sample16", 0)
    for ef in (10, 100, 700):
        a, b = res[(0, ef)], res[(1, ef)]
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), ef
        for q in (0, 33, 69):
            oi, od = O.flat_knn_pq(base, opq, qs[q], 10, ef)
            assert a[0][q, :len(oi)].tolist() == oi.tolist() and np.array_equal(a[1][q, :len(od)], od)
    ix.close()


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
@pytest.mark.parametrize("dim,m", [(64, 43), (192, 171), (344, 342), (96, 33), (100, 31), (128, 63)])
def test_quantised_scan_with_padded_code_words(dim, m, dist, kind):
    """Tables whose code rows are not whole 16-B words (odd m; the DB's default m = ceil(dim / 3): 171 at dim 512, 342 at dim
    1024) on the quantised scan: rows padded with zero bytes, zero tables for the padded groups.  Same answers as the f32 scan
    (pq_adc16 = 1) and as the oracle (pq_table.rs:239-301), and the quantised kernel really ran.  m = 31 / 63: whole code words whose
    last high nibble has no group (tools/fuzz_pq.py #127 / #349, seed 4242: the exact stage's word loop added what sits behind the
    query's table -- the |centroid|^2 table under Cosine)."""
    import lab_1806_vec_db_amd as vdb
    from oracle import oracle as O

    n = 66000
    rng = np.random.default_rng(dim + m)
    base = (rng.standard_normal((n, dim)) * rng.uniform(0.2, 2.0, dim)).astype(np.float32)
    base[40000:40010] = base[:10]
    qs = (base[rng.integers(0, n, 12)] + 0.1 * rng.standard_normal((12, dim))).astype(np.float32)
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base)
    ix.pq_build(n_bits=4, m=m, train_n=2000, max_iter=3, seed=1)
    pq = ix.pq_export()
    opq = O.PQ.from_centroids(dim, m, 4, kind, pq["centroids"])
    opq.set_codes(pq["codes"])
    a = ix.knn_pq(qs, 10, 100)
    ran = ix.get_stat("pq_adc16_queries")
    assert ran >= 12
    ix.set_param("pq_adc16", 1)
    try:
        b = ix.knn_pq(qs, 10, 100)
    finally:
        ix.set_param("pq_adc16", 0)
    assert ix.get_stat("pq_adc16_queries") == ran
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    for q in range(12):
        oi, od = O.flat_knn_pq(base, opq, qs[q], 10, 100, kind)
        assert a[0][q, :len(oi)].tolist() == oi.tolist() and np.array_equal(a[1][q, :len(od)], od)
    ix.close()


@pytest.mark.parametrize("dim,m", [(64, 16), (80, 20), (96, 33), (128, 64), (384, 96)])
def test_quantised_scan_of_8_bit_codes(dim, m):
    """n_bits = 8 (256 centroids per group, pq_table.rs:142-145) on its own quantised passes: sixteen queries per pass on one-byte tables
    cut into slices of 32 groups (k_pq_adc8x16; whole and missing second code words of a slice, one to three slices, two query groups, the
    second with 3 queries), eight per pass on 16-bit tables (k_pq_adc16x8) and one query per pass on a one-byte table (k_pq_adc8), exact f32
    sums for the candidates (k_pq_adc_exact8).  Same answers as the f32 scan (pq_adc16 = 1) and as the oracle, whole and padded code words,
    ties, a degenerate query; Cosine tables keep the f32 scan."""
    import lab_1806_vec_db_amd as vdb
    from oracle import oracle as O

    n = 70000
    rng = np.random.default_rng(dim * 7 + m)
    base = (rng.standard_normal((n, dim)) * rng.uniform(0.2, 2.0, dim)).astype(np.float32)
    base[35000:35010] = base[:10]
    qs = (base[rng.integers(0, n, 19)] + 0.1 * rng.standard_normal((19, dim))).astype(np.float32)  # three query groups, the last with 3
    qs[3, 2] = np.nan  # unquantisable table -> the f32 scan answers that query
    for dist, kind in (("l2sqr", 0), ("cosine", 1)):
        ix = vdb.GpuIndex(dim, dist)
        ix.batch_add(base)
        ix.pq_build(n_bits=8, m=m, train_n=3000, max_iter=2, seed=1)
        pq = ix.pq_export()
        opq = O.PQ.from_centroids(dim, m, 8, kind, pq["centroids"])
        opq.set_codes(pq["codes"])
        for ef in (100, 700):
            a = ix.knn_pq(qs, 10, ef)  # (sixteen queries per pass on sliced one-byte tables: k_pq_adc8x16, round 4)
            ran = ix.get_stat("pq_adc16_queries")
            assert (ran > 0) == (kind == 0)
            for variant in (1, 2):  # one query per pass on a byte table (k_pq_adc8) | eight per pass on sliced 16-bit tables (k_pq_adc16x8)
                ix.set_param("pq_adc8_sliced", variant)
                try:
                    a1 = ix.knn_pq(qs, 10, ef)
                finally:
                    ix.set_param("pq_adc8_sliced", 0)
                assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, a1)), (dist, ef, variant)
            ran = ix.get_stat("pq_adc16_queries")
            ix.set_param("pq_adc16", 1)
            try:
                b = ix.knn_pq(qs, 10, ef)
            finally:
                ix.set_param("pq_adc16", 0)
            assert ix.get_stat("pq_adc16_queries") == ran
            assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b)), (dist, ef)
            for q in range(19):
                oi, od = O.flat_knn_pq(base, opq, qs[q], 10, ef, kind)
                assert a[0][q, :len(oi)].tolist() == oi.tolist(), (dist, ef, q)
                assert np.array_equal(a[1][q, :len(od)], od, equal_nan=True)
        ix.close()


def test_quantised_scan_short_shares_configuration_727():
    """tools/fuzz_pq.py seed 4242 #727 -- dim 32, 74 205 rows, 4-bit m = 9 (5-byte code rows), one query, k = 1, ef = 64, L2Sqr, f32
    threshold sample: a memory-access fault.  With fewer than 1024 rows per CU a workgroup's share of the quantised scan is shorter
    than its 1024 lanes, and the idle lanes of the LAST workgroup formed code-word addresses up to 1023 rows past the table -- past
    the word-major mirror, which here ended on a page boundary (k_pq_adc16 since round 2; idle lanes now re-read a valid row).
    The shape as a regression test (whether an over-read faults depends on what the allocator put behind the mirror)."""
    import lab_1806_vec_db_amd as vdb
    from oracle import oracle as O

    n, dim, m = 74205, 32, 9
    rng = np.random.default_rng(727)
    base = np.round(np.abs(rng.normal(0.07, 0.045, (n, dim))), 4).astype(np.float32)
    qs = (base[rng.integers(0, n, 3)] + 0.05 * rng.standard_normal((3, dim))).astype(np.float32)
    ix = vdb.GpuIndex(dim, "l2sqr")
    ix.batch_add(base)
    ix.pq_build(n_bits=4, m=m, train_n=1500, max_iter=2, seed=727)
    pq = ix.pq_export()
    opq = O.PQ.from_centroids(dim, m, 4, 0, pq["centroids"])
    opq.set_codes(pq["codes"])
    try:
        for s16 in (1, 0):
            ix.set_param("pq_sample16", s16)
            for nq in (1, 3):
                idx, d, cnt = ix.knn_pq(qs[:nq], 1, 64)
                for q in range(nq):
                    oi, od = O.flat_knn_pq(base, opq, qs[q], 1, 64, 0)
                    assert idx[q, :1].tolist() == oi.tolist() and np.array_equal(d[q, :1], od)
    finally:
        ix.set_param("pq_sample16", 0)
    assert ix.get_stat("pq_adc16_queries") > 0
    ix.close()
