"""GPU parity: Flat brute force through the C ABI vs the CPU oracle.

Bar (north_star): neighbour indices identical, f32 distances within 1e-5 relative.  The HIP path is
stricter than that by construction: the distances that leave the library are recomputed in the
reference's summation order, so the tests assert bit-exact equality and only fall back to the
tolerance if that ever fails (the assert message says which).
"""
import numpy as np
import pytest

from conftest import gist_like

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5  # north_star tolerance for f32 distances


def _check(idx_g, dist_g, idx_o, dist_o):
    assert idx_g.tolist() == idx_o.tolist(), f"neighbour indices differ: {idx_g} vs {idx_o}"
    if not np.array_equal(dist_g, dist_o):
        np.testing.assert_allclose(dist_g, dist_o, rtol=REL_TOL, atol=0)
        pytest.fail("distances within 1e-5 but not bit-exact")


@pytest.fixture(scope="module")
def mods():
    import lab_1806_vec_db_amd as vdb
    from oracle import oracle as O
    return vdb, O


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("dist", ["l2sqr", "cosine"])
def test_gist1000_flat_parity(mods, gist_base, gist_test, mode, dist):
    """BASELINE config 1: Flat on gist_1000 x gist_test, k=10 (ground-truth protocol of gen_gnd.rs:54-72)."""
    vdb, O = mods
    ix = vdb.GpuIndex(960, dist)
    ix.batch_add(gist_base)
    ix.set_flat_mode(mode)
    nq = 100
    idx, d, cnt = ix.flat_knn(gist_test[:nq], 10)
    kind = 0 if dist == "l2sqr" else 1
    oi, od, oc = O.flat_knn_batch(gist_base, gist_test[:nq], 10, kind, nthreads=8)
    assert cnt.tolist() == oc.tolist()
    for q in range(nq):
        _check(idx[q], d[q], oi[q], od[q])


def test_survey_golden_vectors(mods, gist_base, gist_test):
    """SURVEY 8c golden candidates (numpy emulation of the strict f32 fold)."""
    vdb, _ = mods
    ix = vdb.GpuIndex(960, "l2sqr")
    ix.batch_add(gist_base)
    i0, d0 = ix.flat_knn(gist_test[0], 10)
    assert i0.tolist() == [918, 467, 725, 988, 207, 56, 18, 348, 27, 632]
    assert float(d0[0]).hex() == "0x1.00b9120000000p+0"
    i999, d999 = ix.flat_knn(gist_test[999], 10)
    assert i999.tolist() == [313, 167, 127, 168, 198, 165, 68, 269, 82, 43]
    assert float(d999[0]).hex() == "0x1.f5bc520000000p-3"


def test_flat_index_test_restated(mods, gist_base):
    """flat_index.rs:117-170: 12-dim clip, L2Sqr, query = row 200, k = 4."""
    vdb, O = mods
    b12 = np.ascontiguousarray(gist_base[:, :12])
    ix = vdb.GpuIndex(12, "l2sqr")
    ix.batch_add(b12)
    idx, d = ix.flat_knn(b12[200], 4)
    assert len(idx) == 4 and idx[0] == 200 and abs(d[0]) < 1e-6
    assert all(d[i] <= d[i + 1] for i in range(3))
    assert idx.tolist() == [200, 750, 471, 793]
    oi, od = O.flat_knn(b12, b12[200], 4)
    _check(idx, d, oi, od)


@pytest.mark.parametrize("dim", [4, 13, 32, 64, 96, 100, 960, 1088, 1536, 2048, 2100])
@pytest.mark.parametrize("dist", ["l2sqr", "cosine"])
def test_dims_and_edges(mods, dim, dist):
    vdb, O = mods
    rng = np.random.default_rng(dim)
    base = rng.standard_normal((777, dim)).astype(np.float32)
    qs = rng.standard_normal((9, dim)).astype(np.float32)
    kind = 0 if dist == "l2sqr" else 1
    ix = vdb.GpuIndex(dim, dist)
    # empty index -> empty result (flat_index.rs:163)
    i, d = ix.flat_knn(qs[0], 5)
    assert len(i) == 0
    ix.batch_add(base[:300])
    ix.batch_add(base[300:])  # growth path
    assert len(ix) == 777
    np.testing.assert_array_equal(ix[776], base[776])
    for mode, gemm in ((1, 0), (2, 1), (2, 2)):  # exact scan; MFMA small-batch kernel; 128-query kernel forced
        ix.set_flat_mode(mode)
        ix.set_param("flat_gemm", gemm)
        for k in (1, 10, 65, 777, 1000):
            idx, dd, cnt = ix.flat_knn(qs, k)
            assert (cnt == min(k, 777)).all()
            for q in range(qs.shape[0]):
                oi, od = O.flat_knn(base, qs[q], k, kind)
                c = int(cnt[q])
                _check(idx[q, :c], dd[q, :c], oi, od)
    i, d = ix.flat_knn(qs[0], 0)
    assert len(i) == 0
    # dimension mismatch is an error, not a silent truncation
    with pytest.raises(vdb.VdbError):
        ix.flat_knn(np.zeros(dim + 1, np.float32), 3)


def test_ties_and_duplicates(mods):
    """Duplicated rows: equal distances must come back in ascending index order (candidate_pair.rs:36-41)."""
    vdb, O = mods
    rng = np.random.default_rng(7)
    dim = 64
    uniq = rng.standard_normal((50, dim)).astype(np.float32)
    base = np.concatenate([uniq] * 400)  # 20000 rows, every row present 400 times
    q = uniq[3] + 0.01
    for mode in (1, 2):
        ix = vdb.GpuIndex(dim, "l2sqr")
        ix.batch_add(base)
        ix.set_flat_mode(mode)
        idx, d = ix.flat_knn(q, 25)
        oi, od = O.flat_knn(base, q, 25)
        _check(idx, d, oi, od)
    assert ix.flat_fallback_count() >= 1  # the MFMA shortlist cannot certify 400-way ties


def test_nan_and_inf_rows(mods):
    vdb, O = mods
    rng = np.random.default_rng(11)
    base = rng.standard_normal((500, 32)).astype(np.float32)
    base[17, 3] = np.nan
    base[40, 0] = np.inf
    q = rng.standard_normal(32).astype(np.float32)
    ix = vdb.GpuIndex(32, "l2sqr")
    ix.batch_add(base)
    idx, d = ix.flat_knn(q, 500)
    oi, od = O.flat_knn(base, q, 500)
    assert idx.tolist() == oi.tolist()
    np.testing.assert_array_equal(np.isnan(d), np.isnan(od))
    m = ~np.isnan(od)
    np.testing.assert_array_equal(d[m], od[m])


@pytest.mark.parametrize("n", [20000, 70001])
@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
def test_gistlike_mfma_parity(mods, n, dist, kind):
    """Gist1M-shaped synthetic rows at a size the oracle finishes in seconds; MFMA path forced."""
    vdb, O = mods
    base = gist_like(n, seed=1806)
    qs = gist_like(40, seed=1807)
    ix = vdb.GpuIndex(960, dist)
    ix.batch_add(base)
    ix.set_flat_mode(2)
    idx, d, cnt = ix.flat_knn(qs, 10)
    oi, od, oc = O.flat_knn_batch(base, qs, 10, kind, nthreads=8)
    for q in range(qs.shape[0]):
        _check(idx[q], d[q], oi[q], od[q])
    # auto mode must agree too
    ix.set_flat_mode(0)
    idx2, d2, _ = ix.flat_knn(qs, 10)
    np.testing.assert_array_equal(idx, idx2)
    np.testing.assert_array_equal(d, d2)
    print("fallbacks:", ix.flat_fallback_count())


@pytest.mark.parametrize("dim,n,nq", [(960, 70001, 200), (128, 30000, 129), (100, 20011, 70), (1536, 20000, 130), (2240, 17000, 66)])
@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
def test_gemm_filter_parity(mods, dim, n, nq, dist, kind):
    """More than 64 queries per call: the filter pass is k_flat_gemm (128 queries per corpus pass, k_gemm.hip).
    Covers KC = 3 and KC = 2 chunking (dim_pad/32 divisible by 3 or not), ragged last group, rows not a multiple of
    a unit, duplicated rows (ties at the cut)."""
    vdb, O = mods
    if dim == 960:
        base, qs = gist_like(n, seed=31), gist_like(nq, seed=32)
    else:
        rng = np.random.default_rng(dim)
        base = rng.standard_normal((n, dim)).astype(np.float32)
        qs = rng.standard_normal((nq, dim)).astype(np.float32)
    base[n - 1] = base[0]
    base[n // 2] = base[1]
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base)
    ix.set_flat_mode(2)
    ix.set_param("flat_half", 1)  # this test compares the split-bf16 kernels (the fp16 first pass: test_flat_half_gpu.py)
    fb = []
    for tw in (3, 2):
        ix.set_param("flat_gemm_tw", tw)
        f0 = ix.flat_fallback_count()
        idx, d, cnt = ix.flat_knn(qs, 10)
        fb.append(ix.flat_fallback_count() - f0)
        if tw == 3:
            oi, od, oc = O.flat_knn_batch(base, qs, 10, kind, nthreads=8)
        for q in range(nq):
            _check(idx[q], d[q], oi[q], od[q])
    ix.set_param("flat_gemm_tw", 3)
    ix.set_param("flat_gemm", 1)  # the small-batch kernel must give the same answer ...
    f0 = ix.flat_fallback_count()
    idx2, d2, _ = ix.flat_knn(qs, 10)
    fb.append(ix.flat_fallback_count() - f0)
    np.testing.assert_array_equal(idx, idx2)
    np.testing.assert_array_equal(d, d2)
    # ... and certify the same queries: both kernels add the same products in the same order per accumulator
    assert fb[0] == fb[1] == fb[2], fb


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
def test_dim1536_mfma_16_query_batches(mods, dist, kind):
    """1024 < dim <= 2048: the Q image of 32 queries no longer fits LDS, batches are 16 queries (k_mfma.hip)."""
    vdb, O = mods
    rng = np.random.default_rng(1536)
    base = rng.standard_normal((30000, 1536)).astype(np.float32)
    qs = rng.standard_normal((37, 1536)).astype(np.float32)
    ix = vdb.GpuIndex(1536, dist)
    ix.batch_add(base)
    idx, d, cnt = ix.flat_knn(qs, 10)  # auto mode: n >= 16384 -> MFMA path
    oi, od, _ = O.flat_knn_batch(base, qs, 10, kind, nthreads=8)
    for q in range(qs.shape[0]):
        _check(idx[q], d[q], oi[q], od[q])
    assert ix.flat_fallback_count() == 0


def test_cosine_degenerate_norms_mfma(mods):
    """Cosine through the MFMA path with zero rows, rows so small that the reference's max(|a||b|, 1e-10)
    clamp (distance/mod.rs:68) is active, a zero query and a tiny query: certification must refuse what it
    cannot bound and the exact scan must take over."""
    vdb, O = mods
    rng = np.random.default_rng(21)
    base = rng.standard_normal((3000, 64)).astype(np.float32)
    base[5] = 0.0
    base[77] = 0.0
    base[100] *= 1e-9
    base[2000] *= 1e-12
    qs = rng.standard_normal((6, 64)).astype(np.float32)
    qs[1] = 0.0
    qs[2] *= 1e-8
    qs[3] = -base[9]  # far side: distances near 2
    ix = vdb.GpuIndex(64, "cosine")
    ix.batch_add(base)
    for mode in (1, 2):
        ix.set_flat_mode(mode)
        for k in (1, 10, 40):
            idx, d, cnt = ix.flat_knn(qs, k)
            for q in range(qs.shape[0]):
                oi, od = O.flat_knn(base, qs[q], k, 1)
                _check(idx[q], d[q], oi, od)
    assert ix.flat_fallback_count() > 0


def test_mfma_nan_rows_and_queries(mods):
    vdb, O = mods
    rng = np.random.default_rng(31)
    base = rng.standard_normal((5000, 128)).astype(np.float32)
    base[17, 3] = np.nan
    base[4000, 0] = np.inf
    qs = rng.standard_normal((4, 128)).astype(np.float32)
    qs[2, 5] = np.nan
    ix = vdb.GpuIndex(128, "l2sqr")
    ix.batch_add(base)
    ix.set_flat_mode(2)
    idx, d, cnt = ix.flat_knn(qs, 10)
    for q in range(4):
        oi, od = O.flat_knn(base, qs[q], 10)
        assert idx[q].tolist() == oi.tolist(), q
        m = ~np.isnan(od)
        np.testing.assert_array_equal(np.isnan(d[q]), np.isnan(od))
        np.testing.assert_array_equal(d[q][m], od[m])


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
def test_large_k_sort_path(mods, dist, kind):
    """k > 1024 (e.g. db.search(k=len)): full (distance, index) sort of the exact distances."""
    vdb, O = mods
    rng = np.random.default_rng(77)
    base = rng.standard_normal((5000, 48)).astype(np.float32)
    base[100] = base[7]  # exact tie
    qs = rng.standard_normal((3, 48)).astype(np.float32)
    ix = vdb.GpuIndex(48, dist)
    ix.batch_add(base)
    for k in (1025, 3000, 5000, 7000):
        idx, d, cnt = ix.flat_knn(qs, k)
        assert (cnt == min(k, 5000)).all()
        for q in range(3):
            oi, od = O.flat_knn(base, qs[q], k, kind)
            _check(idx[q, :len(oi)], d[q, :len(od)], oi, od)


def test_swap_remove_and_offset(mods):
    vdb, O = mods
    rng = np.random.default_rng(3)
    base = rng.standard_normal((100, 16)).astype(np.float32)
    ix = vdb.GpuIndex(16, "l2sqr")
    ix.batch_add(base)
    ix.swap_remove(10)  # vec_set.rs:131-137
    ref = base.copy()
    ref[10] = ref[99]
    ref = ref[:99]
    assert len(ix) == 99
    np.testing.assert_array_equal(ix[10], base[99])
    idx, d = ix.flat_knn(base[5], 7)
    oi, od = O.flat_knn(ref, base[5], 7)
    _check(idx, d, oi, od)
    ix.set_id_offset(1000)
    idx2, _ = ix.flat_knn(base[5], 7)
    assert (idx2 == idx + 1000).all()


def test_calc_dist(mods):
    vdb, O = mods
    assert abs(vdb.calc_dist([1, 2, 3], [4, 5, 6], "l2sqr") - 27.0) < 1e-6  # distance/mod.rs:138-143
    assert abs(vdb.calc_dist([1, 2, 3], [2, 4, 6], "cosine")) < 1e-6       # distance/mod.rs:145-150 (f32 here)
    with pytest.raises(ValueError):
        vdb.calc_dist([1.0], [1.0], "manhattan")
    rng = np.random.default_rng(5)
    a, b = rng.standard_normal(960).astype(np.float32), rng.standard_normal(960).astype(np.float32)
    for name, kind in (("l2sqr", 0), ("cosine", 1)):
        assert vdb.calc_dist(a, b, name) == O.dist(kind, a, b)


def test_u8_scalar_path(mods):
    """DistanceScalar for u8 (distance/mod.rs:79-95): the KAT of :145-150 and a u8 Flat index against the oracle."""
    vdb, O = mods
    assert abs(vdb.calc_dist_u8([1, 2, 3], [2, 4, 6], "cosine") - 0.0) < 1e-6
    rng = np.random.default_rng(8)
    a = rng.integers(0, 256, 300, dtype=np.uint8)
    b = rng.integers(0, 256, 300, dtype=np.uint8)
    for dist, kind in (("l2sqr", 0), ("cosine", 1)):
        assert np.float32(vdb.calc_dist_u8(a, b, dist)) == np.float32(O.dist_u8(kind, a, b))
    base = rng.integers(0, 256, (500, 32), dtype=np.uint8)
    qs = rng.integers(0, 256, (6, 32), dtype=np.uint8)
    for dist, kind in (("l2sqr", 0), ("cosine", 1)):
        ix = vdb.GpuIndex(32, dist)
        ix.batch_add_u8(base)
        idx, d, cnt = ix.flat_knn_u8(qs, 7)
        for q in range(6):
            ref = sorted((np.float32(O.dist_u8(kind, base[i], qs[q])), i) for i in range(500))[:7]
            assert idx[q].tolist() == [i for _, i in ref]
            assert np.array_equal(d[q], np.array([x for x, _ in ref], dtype=np.float32))


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
def test_gemm_filter_adversarial_duplicates(mods, dist, kind):
    """15000 copies of one row: a query near it gets more hits than a candidate list holds (and the workgroup hit
    buffers fill).  Such queries must be flagged and redone by the exact scan, never answered from a truncated list;
    the other queries of the same 128-query group may be flagged with them but must stay exact too."""
    vdb, O = mods
    rng = np.random.default_rng(99)
    n, dim, nq = 40000, 64, 70
    base = rng.standard_normal((n, dim)).astype(np.float32)
    base[10000:25000] = base[7]
    qs = rng.standard_normal((nq, dim)).astype(np.float32)
    qs[0] = base[7]
    qs[1] = base[7] + np.float32(0.001)
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base)
    ix.set_flat_mode(2)
    idx, d, cnt = ix.flat_knn(qs, 10)
    assert ix.flat_fallback_count() >= 2
    oi, od, oc = O.flat_knn_batch(base, qs, 10, kind, nthreads=8)
    for q in range(nq):
        _check(idx[q], d[q], oi[q], od[q])


def test_gemm_no_hit_detection(mods):
    """flat_gemm_debug = 1 makes the filter pass return nothing: every query has fewer than k' hits, which must be
    detected (the thinned threshold sample only makes k' hits overwhelmingly likely, not certain) and redone exactly."""
    vdb, O = mods
    base = gist_like(30000, dim=96, seed=41)
    qs = gist_like(100, dim=96, seed=42)
    ix = vdb.GpuIndex(96, "l2sqr")
    ix.batch_add(base)
    ix.set_flat_mode(2)
    ix.set_param("flat_gemm_debug", 1)
    idx, d, cnt = ix.flat_knn(qs, 10)
    assert ix.flat_fallback_count() == 100
    oi, od, oc = O.flat_knn_batch(base, qs, 10, 0, nthreads=8)
    for q in range(100):
        _check(idx[q], d[q], oi[q], od[q])


def test_merge_topk_gathered_blocks(mods):
    """vdb_merge_topk_gathered reads the per-rank blocks of the all-gather buffer in place (ShardExchange layout); it
    must equal the host merge of the same three shard results, cross-shard ties included."""
    import torch
    vdb, O = mods
    from lab_1806_vec_db_amd.index import merge_topk
    from lab_1806_vec_db_amd.shard import ShardExchange, shard_bounds
    rng = np.random.default_rng(12)
    n, dim, nq, k, S = 3001, 24, 37, 10, 3
    base = rng.standard_normal((n, dim)).astype(np.float32)
    base[2500] = base[3]  # duplicate across shards
    qs = rng.standard_normal((nq, dim)).astype(np.float32)
    qs[0] = base[3]
    dq = torch.from_numpy(qs).cuda()
    blocks, parts = [], []
    for r in range(S):
        r0, r1 = shard_bounds(n, S, r)
        ix = vdb.GpuIndex(dim, "l2sqr")
        ix.batch_add(base[r0:r1])
        ix.set_id_offset(r0)
        ex = ShardExchange(nq, k, torch.device("cuda", 0), 1)
        ix.flat_knn_device(dq.data_ptr(), nq, k, ex.idx.data_ptr(), ex.dist.data_ptr(), ex.cnt.data_ptr())
        blocks.append(ex.send.clone())
        parts.append((ex.idx.cpu().numpy().copy(), ex.dist.cpu().numpy().copy(), ex.cnt.cpu().numpy().copy()))
        last = (ix, ex)
    ix, ex = last
    recv = torch.cat(blocks)
    o_idx = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    o_dist = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    o_cnt = torch.empty((nq,), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ix.merge_topk_gathered(recv.data_ptr(), ex.block, ex.off_ids, ex.off_dists, ex.off_counts, S, nq, k,
                           o_idx.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr())
    hi, hd, hc = merge_topk(np.stack([p[1] for p in parts]), np.stack([p[0] for p in parts]).astype(np.uint64),
                            np.stack([p[2] for p in parts]).astype(np.uint64), k)
    assert np.array_equal(o_idx.cpu().numpy().astype(np.uint64), hi)
    assert np.array_equal(o_dist.cpu().numpy(), hd)
    assert np.array_equal(o_cnt.cpu().numpy().astype(np.uint64), hc)
    for q in range(nq):
        oi, od = O.flat_knn(base, qs[q], k)
        assert hi[q].tolist() == oi.tolist() and np.array_equal(hd[q], od)


def test_merge_topk_gathered_async_and_pipelined_exchange(mods):
    """vdb_merge_topk_gathered_async enqueues the same merge on the caller's stream without a host synchronisation; the
    pipelined ShardExchange (two rotating sets of buffers, exchange of step i under the search of step i+1) returns, for
    every step, what the synchronous merge returns."""
    import torch
    vdb, O = mods
    from lab_1806_vec_db_amd.shard import ShardExchange
    rng = np.random.default_rng(13)
    n, dim, nq, k = 20000, 64, 130, 10
    base = rng.standard_normal((n, dim)).astype(np.float32)
    ix = vdb.GpuIndex(dim, "l2sqr")
    ix.batch_add(base)
    dev = torch.device("cuda", 0)
    ex = ShardExchange(nq, k, dev, 1)
    qsets = [torch.from_numpy(rng.standard_normal((nq, dim)).astype(np.float32)).cuda() for _ in range(5)]
    # (a) the enqueued merge equals the synchronous one on the same (one-block) receive buffer
    ix.flat_knn_device(qsets[0].data_ptr(), nq, k, ex.idx.data_ptr(), ex.dist.data_ptr(), ex.cnt.data_ptr())
    outs = [[torch.zeros((nq, k), dtype=torch.int64, device=dev), torch.zeros((nq, k), dtype=torch.float32, device=dev),
             torch.zeros((nq,), dtype=torch.int64, device=dev)] for _ in range(2)]
    args = (ex.send.data_ptr(), ex.block, ex.off_ids, ex.off_dists, ex.off_counts, 1, nq, k)
    ix.merge_topk_gathered(*args, *[t.data_ptr() for t in outs[0]])
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ix.merge_topk_gathered_async(*args, *[t.data_ptr() for t in outs[1]], stream=st.cuda_stream)
    st.synchronize()
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    # (b) rotating buffers: every step's results stay intact while the next step runs
    class OneRank(ShardExchange):  # the exchange of a single "rank" without a process group: the gathered buffer IS the send block
        def exchange_merge(self, gpu_index, group=None):
            b = self._bufs[self._cur]
            b["recv"].copy_(b["send"])
            stream = torch.cuda.current_stream()
            gpu_index.merge_topk_gathered_async(b["recv"].data_ptr(), self.block, self.off_ids, self.off_dists, self.off_counts, 1,
                                                self.nq, self.k, b["m_idx"].data_ptr(), b["m_dist"].data_ptr(), b["m_cnt"].data_ptr(),
                                                stream=stream.cuda_stream)
            if b["event"] is None:
                b["event"] = torch.cuda.Event()
            b["event"].record(stream)
            return b["m_idx"], b["m_dist"], b["m_cnt"]
    px = OneRank(nq, k, dev, 1, force=True)
    got = []
    for dq in qsets:
        bi, bd, bc = px.begin_step()
        ix.flat_knn_device(dq.data_ptr(), nq, k, bi.data_ptr(), bd.data_ptr(), bc.data_ptr())
        mi, md, mc = px.exchange_merge(ix)
        got.append((mi, md, mc))
        if len(got) >= 2:  # the previous step's results are complete by now (its buffers are reused only at the NEXT begin_step)
            px._bufs[(px._cur + 1) % 2]["event"].synchronize()
            pi, pd, _ = got[-2]
            ref_i, ref_d = ix.flat_knn(qsets[len(got) - 2].cpu().numpy(), k)[:2]
            assert np.array_equal(pi.cpu().numpy(), ref_i.astype(np.int64)) and np.array_equal(pd.cpu().numpy(), ref_d)
    px.wait()
    ref_i, ref_d = ix.flat_knn(qsets[-1].cpu().numpy(), k)[:2]
    assert np.array_equal(got[-1][0].cpu().numpy(), ref_i.astype(np.int64)) and np.array_equal(got[-1][1].cpu().numpy(), ref_d)
