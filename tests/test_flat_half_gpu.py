"""GPU parity: the fp16 first pass of large query batches (k_flat_gemm<GEMM_F16>, k_half.hip).

Calls with more than 64 queries first run with scaled fp16 operands (half the HBM bytes, a third of the matrix work),
certify against the MEASURED rounding error of the operands, and redo what they cannot certify with the split-bf16
pass (which in turn falls back to the exact scan).  Whatever tier answers, the results must equal the oracle's bit for
bit; the tests below drive every tier and the upkeep of the mirror (scale regrow, swap_remove, unsupported dims).
"""
import numpy as np
import pytest

from conftest import gist_like

pytestmark = pytest.mark.gpu


def _check_all(idx, d, cnt, oi, od, oc):
    assert cnt.tolist() == oc.tolist()
    for q in range(idx.shape[0]):
        assert idx[q].tolist() == oi[q].tolist(), (q, idx[q], oi[q])
        assert np.array_equal(d[q], od[q]), (q, d[q], od[q])


@pytest.fixture(scope="module")
def mods():
    import lab_1806_vec_db_amd as vdb
    from oracle import oracle as O
    return vdb, O


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
@pytest.mark.parametrize("dim,n,nq", [(960, 40000, 200), (128, 50000, 130), (192, 30011, 97), (2048, 20000, 70)])
def test_half_pass_parity(mods, dist, kind, dim, n, nq):
    """KB (64-column k-blocks) = 15, 2, 3, 32: chunks of 3 and of 2; ragged last group; rows not a multiple of a unit."""
    vdb, O = mods
    if dim == 960:
        base, qs = gist_like(n, seed=41), gist_like(nq, seed=42)
    else:
        rng = np.random.default_rng(dim + 7)
        base = rng.standard_normal((n, dim)).astype(np.float32)
        qs = rng.standard_normal((nq, dim)).astype(np.float32)
    base[n - 1] = base[0]
    ix = vdb.GpuIndex(dim, dist)
    ix.set_param("flat_i8", 1)  # (this file drives the fp16 / split-bf16 tiers; the 8-bit pass in front of them: test_flat_i8_gpu.py)
    ix.batch_add(base)
    ix.set_flat_mode(2)
    assert ix.get_stat("flat_half_valid") == 1
    idx, d, cnt = ix.flat_knn(qs, 10)
    assert ix.get_stat("flat_half_queries") == nq
    redo = ix.get_stat("flat_half_redo")
    oi, od, oc = O.flat_knn_batch(base, qs, 10, kind, nthreads=8)
    _check_all(idx, d, cnt, oi, od, oc)
    ix.set_param("flat_half", 1)  # split-bf16 only: same answer
    idx2, d2, cnt2 = ix.flat_knn(qs, 10)
    assert ix.get_stat("flat_half_queries") == nq
    np.testing.assert_array_equal(idx, idx2)
    np.testing.assert_array_equal(d, d2)
    print(f"dim {dim} {dist}: fp16 pass redid {redo} of {nq} queries")
    assert redo <= nq // 4  # random / gist-like data certify almost always
    ix.set_param("flat_half", 0)
    for k in (1, 50, 200):  # shortlists of 64, 200 and 800 rows
        h0 = ix.get_stat("flat_half_queries")
        idx, d, cnt = ix.flat_knn(qs[:66], k)
        assert ix.get_stat("flat_half_queries") == h0 + 66
        _check_all(idx, d, cnt, *O.flat_knn_batch(base, qs[:66], k, kind, nthreads=8))


def test_half_pass_redo_tier(mods):
    """Near-duplicate rows: the gaps between the 10th and the 64th neighbour are far below the fp16 rounding error, so
    the first pass cannot certify; the split-bf16 pass (and for exact ties the exact scan) must take over."""
    vdb, O = mods
    rng = np.random.default_rng(5)
    n, dim, nq = 30000, 256, 100
    centers = rng.standard_normal((n // 100, dim)).astype(np.float32)
    base = np.repeat(centers, 100, axis=0) + (1e-4 * rng.standard_normal((n, dim))).astype(np.float32)
    qs = centers[:nq] + (1e-4 * rng.standard_normal((nq, dim))).astype(np.float32)
    ix = vdb.GpuIndex(dim, "l2sqr")
    ix.set_param("flat_i8", 1)  # (this file drives the fp16 / split-bf16 tiers; the 8-bit pass in front of them: test_flat_i8_gpu.py)
    ix.batch_add(base)
    ix.set_flat_mode(2)
    ix.set_param("flat_half", 2)
    idx, d, cnt = ix.flat_knn(qs, 10)
    assert ix.get_stat("flat_half_redo") > 0
    oi, od, oc = O.flat_knn_batch(base, qs, 10, 0, nthreads=8)
    _check_all(idx, d, cnt, oi, od, oc)
    # auto mode gives up on the first pass once most queries had to be redone
    ix.set_param("flat_half", 0)
    for _ in range(12):
        ix.flat_knn(qs, 10)
    before = ix.get_stat("flat_half_queries")
    idx3, d3, _ = ix.flat_knn(qs, 10)
    if ix.get_stat("flat_half_redo") * 8 > before:
        assert ix.get_stat("flat_half_queries") == before
    np.testing.assert_array_equal(idx, idx3)
    np.testing.assert_array_equal(d, d3)


def test_half_mirror_rescale_and_swap_remove(mods):
    """Rows with 1000x larger norms arrive later: the mirror is rewritten with a new scale; swap_remove rewrites tiles."""
    vdb, O = mods
    rng = np.random.default_rng(9)
    dim, nq = 192, 80
    a = (0.01 * rng.standard_normal((20000, dim))).astype(np.float32)
    b = (10.0 * rng.standard_normal((5000, dim))).astype(np.float32)
    qs = np.concatenate([a[:40] + (0.003 * rng.standard_normal((40, dim))).astype(np.float32),
                         b[:40] + (3.0 * rng.standard_normal((40, dim))).astype(np.float32)])
    ix = vdb.GpuIndex(dim, "l2sqr")
    ix.set_flat_mode(2)
    ix.set_param("flat_i8", 1)  # (this file drives the fp16 / split-bf16 tiers; the 8-bit pass in front of them: test_flat_i8_gpu.py)
    ix.batch_add(a)
    idx, d, cnt = ix.flat_knn(qs, 10)
    _check_all(idx, d, cnt, *O.flat_knn_batch(a, qs, 10, 0, nthreads=8))
    ix.batch_add(b)
    base = np.concatenate([a, b])
    idx, d, cnt = ix.flat_knn(qs, 10)
    _check_all(idx, d, cnt, *O.flat_knn_batch(base, qs, 10, 0, nthreads=8))
    for i in (3, 24990, 17):
        ix.swap_remove(i)
        base[i] = base[-1]
        base = base[:-1]
    idx, d, cnt = ix.flat_knn(qs, 10)
    _check_all(idx, d, cnt, *O.flat_knn_batch(base, qs, 10, 0, nthreads=8))
    assert ix.get_stat("flat_half_queries") == 3 * nq


def test_half_pass_unsupported_inputs(mods):
    """dim 320 (5 k-blocks: no chunking) and extreme norms: no fp16 mirror, the split-bf16 pass serves the call."""
    vdb, O = mods
    rng = np.random.default_rng(11)
    base = rng.standard_normal((20000, 320)).astype(np.float32)
    qs = rng.standard_normal((70, 320)).astype(np.float32)
    ix = vdb.GpuIndex(320, "l2sqr")
    ix.set_param("flat_i8", 1)  # (this file drives the fp16 / split-bf16 tiers; the 8-bit pass in front of them: test_flat_i8_gpu.py)
    ix.batch_add(base)
    ix.set_flat_mode(2)
    assert ix.get_stat("flat_half_valid") == 0
    idx, d, cnt = ix.flat_knn(qs, 10)
    _check_all(idx, d, cnt, *O.flat_knn_batch(base, qs, 10, 0, nthreads=8))
    big = (1e15 * rng.standard_normal((20000, 128))).astype(np.float32)  # norms^2 ~ 1e32 > 2^80
    ix2 = vdb.GpuIndex(128, "cosine")
    ix2.set_param("flat_i8", 1)  # (this file drives the fp16 / split-bf16 tiers; the 8-bit pass in front of them: test_flat_i8_gpu.py)
    ix2.batch_add(big)
    ix2.set_flat_mode(2)
    assert ix2.get_stat("flat_half_valid") == 0
    qb = (1e15 * rng.standard_normal((70, 128))).astype(np.float32)
    idx, d, cnt = ix2.flat_knn(qb, 10)
    _check_all(idx, d, cnt, *O.flat_knn_batch(big, qb, 10, 1, nthreads=8))


def test_half_pass_odd_queries(mods):
    """Zero query, huge query, tiny query and a NaN query inside a large batch: per-query scales / error terms must
    either be exact or refuse certification."""
    vdb, O = mods
    rng = np.random.default_rng(13)
    base = rng.standard_normal((30000, 128)).astype(np.float32)
    qs = rng.standard_normal((90, 128)).astype(np.float32)
    qs[3] = 0.0
    qs[5] *= 1e20
    qs[7] *= 1e-20
    qs[11] *= 1e6
    qs[13] *= 1e-6
    for dist, kind in (("l2sqr", 0), ("cosine", 1)):
        ix = vdb.GpuIndex(128, dist)
        ix.set_param("flat_i8", 1)  # (this file drives the fp16 / split-bf16 tiers; the 8-bit pass in front of them: test_flat_i8_gpu.py)
        ix.batch_add(base)
        ix.set_flat_mode(2)
        idx, d, cnt = ix.flat_knn(qs, 10)
        assert ix.get_stat("flat_half_queries") == 90
        _check_all(idx, d, cnt, *O.flat_knn_batch(base, qs, 10, kind, nthreads=8))


def test_fused_exact_stage_equals_separate_kernels(mods):
    """k_flat_tail64 (counted select + re-rank + sort + certification in one launch, shortlists of up to 64 rows) against the
    four separate kernels, through both first passes and for Cosine's norm epilogue; ragged k, small calls."""
    vdb, O = mods
    base, qs = gist_like(30000, seed=51), gist_like(150, seed=52)
    base[29999] = base[5]
    for dist, kind in (("l2sqr", 0), ("cosine", 1)):
        ix = vdb.GpuIndex(960, dist)
        ix.set_param("flat_i8", 1)  # (this file drives the fp16 / split-bf16 tiers; the 8-bit pass in front of them: test_flat_i8_gpu.py)
        ix.batch_add(base)
        ix.set_flat_mode(2)
        for half in (0, 1):
            ix.set_param("flat_half", half)
            for nq, k in ((150, 10), (150, 16), (7, 3), (70, 1)):
                ix.set_param("flat_tail", 0)
                i0, d0, c0 = ix.flat_knn(qs[:nq], k)
                ix.set_param("flat_tail", 1)
                i1, d1, c1 = ix.flat_knn(qs[:nq], k)
                np.testing.assert_array_equal(i0, i1)
                np.testing.assert_array_equal(d0, d1)
                np.testing.assert_array_equal(c0, c1)
                if (nq, k) == (150, 10):
                    _check_all(i0, d0, c0, *O.flat_knn_batch(base, qs[:nq], k, kind, nthreads=8))
        assert ix.flat_fallback_count() == 0


def test_stat_names(mods):
    vdb, _ = mods
    ix = vdb.GpuIndex(128, "l2sqr")
    for name in ("flat_fallback", "flat_half_queries", "flat_half_redo", "flat_half_valid"):
        assert ix.get_stat(name) == 0
    with pytest.raises(vdb.VdbError):
        ix.get_stat("no_such_counter")


def test_row_blocked_filter_pass(mods):
    """flat_gemm_block_rows scans the mirror in row blocks, one launch per block over all query groups (global row ids =
    block base + local id; a measurement switch, off by default).  Small corpus: ragged last block, both precisions."""
    vdb, O = mods
    base, qs = gist_like(50000, seed=61), gist_like(300, seed=62)
    oi, od, oc = O.flat_knn_batch(base, qs, 10, 0, nthreads=8)
    ix = vdb.GpuIndex(960, "l2sqr")
    ix.set_param("flat_i8", 1)  # (this file drives the fp16 / split-bf16 tiers; the 8-bit pass in front of them: test_flat_i8_gpu.py)
    ix.batch_add(base)
    ix.set_flat_mode(2)
    try:
        for half in (0, 1):
            ix.set_param("flat_half", half)
            for block in (0, 384, 9600, 1 << 20):
                ix.set_param("flat_gemm_block_rows", block)
                idx, d, cnt = ix.flat_knn(qs, 10)
                _check_all(idx, d, cnt, oi, od, oc)
    finally:
        ix.set_param("flat_gemm_block_rows", 0)
    assert ix.flat_fallback_count() == 0


@pytest.mark.parametrize("dist,kind,dim,n,nq,half", [("cosine", 1, 960, 100000, 512, 0), ("l2sqr", 0, 128, 120000, 1024, 0), ("cosine", 1, 192, 99000, 256, 1)])
def test_cooperative_sets_of_the_fp16_and_split_bf16_filter(mods, dist, kind, dim, n, nq, half):
    """k_flat_gemm with the workgroups of an XCD in sets that share one row stream (k_gemm8.hip's scheme; sets of 4 / 8 / 2 here): the
    same answers as with the sets switched off, bit for bit, and as the oracle on a sample -- fp16 pass (Cosine, and L2Sqr with the
    8-bit pass off) and the split-bf16 pass alone (flat_half = 1)"""
    vdb, O = mods
    rng = np.random.default_rng(n + nq)
    if dim == 960:
        base, qs = gist_like(n, seed=91), gist_like(nq, seed=92)
    else:
        base = rng.standard_normal((n, dim)).astype(np.float32)
        qs = rng.standard_normal((nq, dim)).astype(np.float32)
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base)
    ix.set_flat_mode(2)
    ix.set_param("flat_i8", 1)
    ix.set_param("flat_half", 1 if half else 0)
    ix.set_param("flat_gemm_coop", 1)  # off
    idx0, d0, cnt0 = ix.flat_knn(qs, 10)
    assert ix.get_stat("flat_gemm_coop_sets") <= 1  # the plain form ran
    ix.set_param("flat_gemm_coop", 0)  # auto
    idx1, d1, cnt1 = ix.flat_knn(qs, 10)
    groups = (nq + 127) // 128
    assert ix.get_stat("flat_gemm_coop_sets") == (8 if groups % 8 == 0 else 4 if groups % 4 == 0 else 2)  # the sets really ran
    assert ix.get_stat("flat_i8_queries") == 0
    if not half:
        assert ix.get_stat("flat_half_queries") == 2 * nq
    np.testing.assert_array_equal(idx0, idx1)
    np.testing.assert_array_equal(d0, d1)
    np.testing.assert_array_equal(cnt0, cnt1)
    sel = rng.choice(nq, 32, replace=False)
    oi, od, oc = O.flat_knn_batch(base, qs[sel], 10, kind, nthreads=8)
    _check_all(idx1[sel], d1[sel], cnt1[sel], oi, od, oc)
    ix.close()
