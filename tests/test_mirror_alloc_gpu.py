"""GPU: a mirror that cannot be allocated is not a failed search (the fp16 mirror is built inside the first read that needs it; a
hipMalloc failure there used to throw out of the query although the next tier could have answered).  `debug_alloc_fail_over` makes every
device allocation of at least N bytes fail like an out-of-memory; the tiers must fall through -- 8-bit -> fp16 -> split-bf16 -> exact scan
over the rows -- with the oracle's answers, and vdb_index_prepare must build (or skip) the mirrors ahead of the first search."""
import os

import numpy as np
import pytest

from conftest import gist_like

# (VDB_EFENCE=1 gives every buffer an exact-size allocation of its own: the rows buffer then reallocates on every add and the size
# threshold of the failure injection hits it too -- the cascade is tested in the ordinary mode)
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("VDB_EFENCE") == "1", reason="failure injection is by allocation size")]


@pytest.fixture(scope="module")
def mods():
    import lab_1806_vec_db_amd as vdb
    from oracle import oracle as O
    return vdb, O


def _check(idx, d, cnt, oi, od, oc):
    assert cnt.tolist() == oc.tolist()
    assert np.array_equal(idx, oi) and np.array_equal(d, od)


@pytest.mark.parametrize("dist", ["l2sqr", "cosine"])
def test_search_survives_mirror_allocation_failures(mods, dist):
    vdb, O = mods
    n, dim, nq = 40000, 192, 96
    rng = np.random.default_rng(8)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    qs = rng.standard_normal((nq, dim)).astype(np.float32)
    oi, od, oc = O.flat_knn_batch(base, qs, 10, O.L2SQR if dist == "l2sqr" else O.COSINE, nthreads=8)
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base)
    ix.set_flat_mode(2)
    i8_bytes, bf16_bytes = n * dim, n * dim * 4  # the mirrors: 1 B, (2 B,) 4 B per element
    try:
        # (a) nothing of a mirror's size can be allocated: every tier falls through to the exact scan over the rows
        ix.set_param("debug_alloc_fail_over", i8_bytes // 2)
        idx, d, cnt = ix.flat_knn(qs, 10)
        _check(idx, d, cnt, oi, od, oc)
        assert ix.get_stat("flat_i8_valid") == 0 and ix.get_stat("flat_half_valid") == 0 and ix.get_stat("flat_bf16_mirror") == 0
        fails = ix.get_stat("mirror_alloc_failures")
        assert fails >= 2
        idx, d, cnt = ix.flat_knn(qs, 10)  # not retried per call while the table is unchanged
        _check(idx, d, cnt, oi, od, oc)
        assert ix.get_stat("mirror_alloc_failures") == fails
        # (b) room for the 8-bit mirror only: it is built by this call (rows changed -> retried) and answers
        ix.batch_add(base[:16])
        base2 = np.concatenate([base, base[:16]])
        oi2, od2, oc2 = O.flat_knn_batch(base2, qs, 10, O.L2SQR if dist == "l2sqr" else O.COSINE, nthreads=8)
        ix.set_param("debug_alloc_fail_over", i8_bytes * 3 // 2)
        idx, d, cnt = ix.flat_knn(qs, 10)
        _check(idx, d, cnt, oi2, od2, oc2)
        assert ix.get_stat("flat_i8_valid") == 1 and ix.get_stat("flat_i8_queries") == nq
        # (c) the redo path of an 8-bit-only index: one round of 63 rows and no second attempt hands queries on; the fp16 mirror cannot be
        # allocated, the split-bf16 mirror neither -> the exact scan answers them
        ix.set_param("flat_i8", 2)
        ix.set_param("flat_i8_rows", 64)
        ix.set_param("flat_i8_second", 1)
        ix.set_param("flat_gemm_debug", 1)  # thresholds of -inf: nothing passes the filter, every query is handed on
        r0 = ix.get_stat("flat_i8_redo")
        idx, d, cnt = ix.flat_knn(qs, 10)
        _check(idx, d, cnt, oi2, od2, oc2)
        assert ix.get_stat("flat_i8_redo") - r0 == nq and ix.get_stat("flat_half_valid") == 0
        ix.set_param("flat_gemm_debug", 0)
        # (d) memory is back: prepare builds the redo tiers ahead of time and the same redo runs on the fp16 pass
        ix.set_param("debug_alloc_fail_over", 0)
        ix.batch_add(base[16:32])
        ix.prepare(all_tiers=True)
        assert ix.get_stat("flat_i8_valid") == 1 and ix.get_stat("flat_half_valid") == 1 and ix.get_stat("flat_bf16_mirror") == 1
        base3 = np.concatenate([base2, base[16:32]])
        oi3, od3, oc3 = O.flat_knn_batch(base3, qs, 10, O.L2SQR if dist == "l2sqr" else O.COSINE, nthreads=8)
        ix.set_param("flat_gemm_debug", 1)
        h0 = ix.get_stat("flat_half_queries")
        idx, d, cnt = ix.flat_knn(qs, 10)
        _check(idx, d, cnt, oi3, od3, oc3)
        assert ix.get_stat("flat_half_queries") >= h0  # (the -inf thresholds starve the fp16 tier too: it hands on to the exact scan)
        ix.set_param("flat_gemm_debug", 0)
        idx, d, cnt = ix.flat_knn(qs, 10)
        _check(idx, d, cnt, oi3, od3, oc3)
    finally:
        ix.set_param("debug_alloc_fail_over", 0)
        ix.close()


def test_prepare_on_a_fresh_index(mods):
    """prepare() before the first search: the first tier's mirror exists, the first search builds nothing; empty and tiny tables are no-ops"""
    vdb, O = mods
    base, qs = gist_like(30000, seed=3), gist_like(40, seed=4)
    ix = vdb.GpuIndex(960, "cosine")
    ix.prepare()  # empty: nothing to do
    ix.batch_add(base)
    assert ix.get_stat("flat_i8_valid") == 0
    ix.prepare()
    assert ix.get_stat("flat_i8_valid") == 1 and ix.get_stat("flat_half_valid") == 0
    ix.set_flat_mode(2)
    idx, d, cnt = ix.flat_knn(qs, 10)
    oi, od, oc = O.flat_knn_batch(base, qs, 10, O.COSINE, nthreads=8)
    _check(idx, d, cnt, oi, od, oc)
    ix.close()
    small = vdb.GpuIndex(12, "l2sqr")
    small.batch_add(np.ones((5, 12), dtype=np.float32))
    small.prepare(all_tiers=True)
    small.close()


def test_key_refinement_without_room_for_the_fp16_image(mods):
    """the row-major fp16 image behind the refinement of the 8-bit pass's hit keys (k_flat_refine_half) is an accelerator: when it cannot be
    allocated the search goes on with the 8-bit keys (same answers), the failure is counted once and not retried until the table changes"""
    from conftest import gist_clustered

    vdb, O = mods
    n, dim, nq = 60_000, 192, 128
    base = gist_clustered(n, dim=dim, seed=31, clusters=32, spread=0.15)
    qs = gist_clustered(nq, dim=dim, seed=32, clusters=32, spread=0.15)
    oi, od, oc = O.flat_knn_batch(base, qs, 10, O.L2SQR, nthreads=8)
    ix = vdb.GpuIndex(dim, "l2sqr")
    ix.batch_add(base)
    try:
        ix.flat_knn(qs[:8], 10)  # the 8-bit mirror exists now
        ix.set_param("flat_i8_refine", 2)  # always: wants the fp16 images (2 x 2 B per element)
        ix.set_param("debug_alloc_fail_over", n * dim * 3 // 2)
        f0 = ix.get_stat("mirror_alloc_failures")
        idx, d, cnt = ix.flat_knn(qs, 10)
        _check(idx, d, cnt, oi, od, oc)
        assert ix.get_stat("flat_i8_refine_queries") == 0 and ix.get_stat("mirror_alloc_failures") > f0
        f1 = ix.get_stat("mirror_alloc_failures")
        idx, d, cnt = ix.flat_knn(qs, 10)
        _check(idx, d, cnt, oi, od, oc)
        assert ix.get_stat("mirror_alloc_failures") == f1
        ix.set_param("debug_alloc_fail_over", 0)  # memory is back, the table changes: built and used
        ix.batch_add(base[:8])
        idx, d, cnt = ix.flat_knn(qs, 10)
        assert ix.get_stat("flat_i8_refine_queries") == nq
        oi2, od2, oc2 = O.flat_knn_batch(np.concatenate([base, base[:8]]), qs, 10, O.L2SQR, nthreads=8)
        _check(idx, d, cnt, oi2, od2, oc2)
    finally:
        ix.set_param("debug_alloc_fail_over", 0)
        ix.close()
