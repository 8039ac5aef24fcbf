"""GPU: BASELINE.json's full size (1,000,000 x 960 f32) through size-independent properties, because the CPU
oracle needs ~1 s per query per core at this size: (a) the MFMA path equals the strict-order exact scan,
(b) four row shards merged by (distance, index) equal the unsharded answer, (c) a row queried against the
corpus finds itself first at distance exactly 0, (d) results ascend in (distance, index), (e) a handful of
queries are checked against the oracle itself.  The 1M case runs 256 queries (two groups: cooperative sets of 2, the hit buffer
handed over in blocks, the exact stage of the 8-bit pass over full-size hit lists), L2Sqr and Cosine."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, DIM, K = 1_000_000, 960, 10


@pytest.fixture(scope="module")
def world():
    import torch

    import lab_1806_vec_db_amd as vdb
    from bench import gist_like_gpu

    dev = torch.device("cuda", 0)
    base = gist_like_gpu(torch, N, DIM, 1806, dev)
    qs = gist_like_gpu(torch, 256, DIM, 1807, dev)
    ix = vdb.GpuIndex(DIM, "l2sqr")
    ix.add_device(base.data_ptr(), N)
    return vdb, torch, base, qs.cpu().numpy(), ix


def test_mfma_equals_exact_scan_and_properties(world):
    vdb, torch, base, qs, ix = world
    ix.set_flat_mode(0)
    q0 = ix.get_stat("flat_i8_queries")
    idx, d, cnt = ix.flat_knn(qs, K)
    assert (cnt == K).all()
    assert ix.flat_fallback_count() == 0
    # the headline path answered: the 8-bit pass with cooperative sets (2 groups -> sets of 2), nothing passed on
    assert ix.get_stat("flat_i8_queries") == q0 + len(qs) and ix.get_stat("flat_gemm8_coop_sets") == 2
    assert ix.get_stat("flat_i8_redo") == 0
    ix.set_param("flat_gemm8_coop", 1)  # the plain resident form: same hit lists, same answers
    idx_p, d_p, _ = ix.flat_knn(qs, K)
    ix.set_param("flat_gemm8_coop", 0)
    assert np.array_equal(idx, idx_p) and np.array_equal(d, d_p)
    ix.set_flat_mode(1)
    e_idx, e_d, _ = ix.flat_knn(qs[:24], K)
    ix.set_flat_mode(0)
    assert np.array_equal(idx[:24], e_idx) and np.array_equal(d[:24], e_d)
    for q in range(qs.shape[0]):  # ascending by (distance, index)
        pairs = list(zip(d[q].tolist(), idx[q].tolist()))
        assert pairs == sorted(pairs)
    # self-query: rows of the corpus as queries (flat_index.rs:163-165)
    rows = [0, 12345, 999_999]
    sq = base[rows].cpu().numpy()
    s_idx, s_d, _ = ix.flat_knn(sq, 4)
    for j, r in enumerate(rows):
        assert s_idx[j, 0] == r and s_d[j, 0] == 0.0


def test_four_shards_equal_unsharded(world):
    vdb, torch, base, qs, ix = world
    from lab_1806_vec_db_amd.shard import shard_bounds

    full_idx, full_d, _ = ix.flat_knn(qs[:64], K)
    parts_i, parts_d, parts_c = [], [], []
    for s in range(4):
        r0, r1 = shard_bounds(N, 4, s)
        sh = vdb.GpuIndex(DIM, "l2sqr")
        sh.add_device(base[r0:r1].contiguous().data_ptr(), r1 - r0)
        sh.set_id_offset(r0)
        i, dd, c = sh.flat_knn(qs[:64], K)
        parts_i.append(i); parts_d.append(dd); parts_c.append(c)
        sh.close()
    mi, md, mc = vdb.merge_topk(np.stack(parts_d), np.stack(parts_i), np.stack(parts_c), K)
    assert np.array_equal(mi, full_idx) and np.array_equal(md, full_d) and (mc == K).all()


def test_against_oracle_sample(world):
    vdb, torch, base, qs, ix = world
    from oracle import oracle as O

    host = base.cpu().numpy()
    oi, od, oc = O.flat_knn_batch(host, qs[:16], K, O.L2SQR, nthreads=16)
    gi, gd, _ = ix.flat_knn(qs[:16], K)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)


def test_cosine_full_size(world):
    """the reference's default metric at full size: the 8-bit pass on unit rows (sets of 2) equals the strict-order exact scan and the
    oracle; results ascend"""
    vdb, torch, base, qs, ix = world
    from oracle import oracle as O

    cx = vdb.GpuIndex(DIM, "cosine")
    cx.add_device(base.data_ptr(), N)
    idx, d, cnt = cx.flat_knn(qs, K)
    assert (cnt == K).all() and cx.get_stat("flat_i8_queries") == len(qs) and cx.get_stat("flat_gemm8_coop_sets") == 2
    assert cx.get_stat("flat_i8_redo") <= len(qs) // 8
    cx.set_flat_mode(1)
    e_idx, e_d, _ = cx.flat_knn(qs[:16], K)
    cx.set_flat_mode(0)
    assert np.array_equal(idx[:16], e_idx) and np.array_equal(d[:16], e_d)
    for q in range(qs.shape[0]):
        pairs = list(zip(d[q].tolist(), idx[q].tolist()))
        assert pairs == sorted(pairs)
    host = base.cpu().numpy()
    oi, od, oc = O.flat_knn_batch(host, qs[248:256], K, O.COSINE, nthreads=16)
    assert np.array_equal(idx[248:256], oi) and np.array_equal(d[248:256], od)
    cx.close()
