"""CPU: pins the oracle (oracle/vdb_oracle.c) before anything trusts it.

Every known-answer / property test the reference holds for the hot path is restated
here against the oracle, plus the committed golden vectors (tests/golden/flat_golden.json, produced by the
independent numpy emulation in oracle/np_ref.py on the reference's own gist fixtures).
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import np_ref as R
from oracle import oracle as O

EPS = 1e-6


def test_l2_kat():  # distance/mod.rs:138-143
    assert abs(O.dist(O.L2SQR, [1, 2, 3], [4, 5, 6]) - 27.0) < EPS


def test_cosine_u8_kat():  # distance/mod.rs:145-150
    assert abs(O.dist_u8(O.COSINE, [1, 2, 3], [2, 4, 6]) - 0.0) < EPS


def test_pq_groups_kat():  # pq_table.rs:312-322
    assert O.pq_groups(6, 2) == [(0, 3), (3, 6)]
    assert O.pq_groups(7, 3) == [(0, 3), (3, 5), (5, 7)]
    g = O.pq_groups(960, 320)
    assert len(g) == 320 and all(b - a == 3 for a, b in g)
    g = O.pq_groups(960, 240)
    assert len(g) == 240 and all(b - a == 4 for a, b in g)


@pytest.mark.parametrize("kind", [O.L2SQR, O.COSINE])
def test_pq_precise_property(kind):  # pq_table.rs:324-372: n < k, centroids == points -> ADC == exact
    rng = np.random.default_rng(42)
    src = rng.uniform(-1, 1, (5, 8)).astype(np.float32)
    pq = O.PQ.train(src, m=2, n_bits=4, kind=kind, k_means_size=0, max_iter=20, tol=1e-6, seed=42)
    codes = pq.codes
    for i in range(5):
        lut, qc = pq.lookup(src[i])
        for j in range(5):
            assert abs(O.dist(kind, src[i], src[j]) - pq.adc(codes[j], lut, qc)) < 1e-6


@pytest.mark.parametrize("kind", [O.L2SQR, O.COSINE])
def test_pq_p90_error(gist_base, kind):  # pq_table.rs:374-438: gist 64 rows x 13 dims, p90 < 0.2
    vs = np.ascontiguousarray(gist_base[:64, :13])
    pq = O.PQ.train(vs, m=-(-13 // 3), n_bits=4, kind=kind, k_means_size=0, max_iter=20, tol=1e-6, seed=42)
    codes = pq.codes
    rng = np.random.default_rng(42)
    errs = []
    for _ in range(20):
        i0, i1 = rng.integers(0, 64, 2)
        lut, qc = pq.lookup(vs[i1])
        d = pq.adc(codes[i0], lut, qc)
        e = O.dist(kind, vs[i0], vs[i1])
        errs.append(abs(d - e) / max(e, 1.0))
    errs.sort()
    assert errs[int(np.ceil(len(errs) * 0.9)) - 1] < 0.2


def test_pq_nibble_order():  # pq_table.rs:55-61,72-84: low nibble = even group; odd m -> last byte has one code
    cent = np.zeros(16 * 3, dtype=np.float32)
    # 3 groups of 1 dim; centroid c of every group sits at value c
    for g in range(3):
        cent[16 * g:16 * g + 16] = np.arange(16)
    pq = O.PQ.from_centroids(3, 3, 4, O.L2SQR, cent)
    code = pq.encode_row(np.array([2.2, 9.9, 14.6], np.float32))
    assert code.tolist() == [2 | (10 << 4), 15]


def test_kmeans_find_nearest(gist_base):  # k_means.rs:241-277
    c = O.kmeans(gist_base[:400], 0, 5, 3, seed=42)
    assert c.shape == (3, 5)
    pq = O.PQ.from_centroids(5, 1, 4, O.L2SQR, np.concatenate([c.ravel(), np.full(13 * 5, 1e6, np.float32)]))
    assert pq.encode_row(c[1])[0] == 1


def test_flat_index_test(gist_base):  # flat_index.rs:117-170
    b12 = np.ascontiguousarray(gist_base[:, :12])
    idx, d = O.flat_knn(b12, b12[200], 4)
    assert len(idx) == 4 and idx[0] == 200 and abs(d[0]) < 1e-6
    assert all(d[i] <= d[i + 1] for i in range(3))


@pytest.mark.parametrize("kind", [O.L2SQR, O.COSINE])
def test_hnsw_equals_flat(gist_base, kind):  # hnsw_index.rs:713-790
    b12 = np.ascontiguousarray(gist_base[:, :12])
    h = O.HNSW.build(b12, kind=kind, M=16, ef_construction=200, seed=42, batch=1)
    hi, hd = h.knn(b12[200], 6)
    fi, fd = O.flat_knn(b12, b12[200], 6, kind)
    assert hi.tolist() == fi.tolist()
    assert all(hd[i] <= hd[i + 1] for i in range(5))
    # graph round trip (the reference test saves/loads the index twice)
    h2 = O.HNSW.from_graph(b12, kind, 16, 200, h.graph())
    hi2, hd2 = h2.knn(b12[200], 6)
    assert hi2.tolist() == hi.tolist() and np.array_equal(hd, hd2)


def test_database_cosine_search_restated():  # database/mod.rs:551-607 (arithmetic part)
    rows = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], np.float32)
    pq = O.PQ.train(rows, m=2, n_bits=4, kind=O.COSINE, k_means_size=1, seed=1)
    idx, d = O.flat_knn_pq(rows, pq, np.array([0, 0, 1, 0], np.float32), 3, 3, O.COSINE)
    keep = [int(i) for i, x in zip(idx, d) if x <= 0.5]
    assert keep == [2]


def test_golden_vectors(gist_base, gist_test):
    g = json.load(open(os.path.join(GOLDEN, "flat_golden.json")))
    for q, e in g["flat_l2"].items():
        idx, d = O.flat_knn(gist_base, gist_test[int(q)], 10)
        assert idx.tolist() == e["idx"] and [float(x).hex() for x in d] == e["dist"]
    for q, e in g["flat_cosine"].items():
        idx, d = O.flat_knn(gist_base, gist_test[int(q)], 10, O.COSINE)
        assert idx.tolist() == e["idx"] and [float(x).hex() for x in d] == e["dist"]
    for q, e in g["cached_l2"].items():
        qv = gist_test[int(q)]
        qc = O.dist_cache(O.L2SQR, qv)
        got = [float(np.float32(O.dist_cached(O.L2SQR, gist_base[i], qv, O.dist_cache(O.L2SQR, gist_base[i]), qc))).hex()
               for i in range(16)]
        assert got == e
    b12 = np.ascontiguousarray(gist_base[:, :12])
    idx, d = O.flat_knn(b12, b12[200], 4)
    assert idx.tolist() == g["clip12_row200_k4"]["idx"] == [200, 750, 471, 793]
    assert [float(x).hex() for x in d] == g["clip12_row200_k4"]["dist"]
    assert [float(np.float32(O.dist_cache(O.L2SQR, gist_base[i]))).hex() for i in range(8)] == g["selfdot_first8"]


def test_survey_golden_candidates(gist_base, gist_test):
    i0, d0 = O.flat_knn(gist_base, gist_test[0], 10)
    assert i0.tolist() == [918, 467, 725, 988, 207, 56, 18, 348, 27, 632]
    assert float(d0[0]).hex() == "0x1.00b9120000000p+0"
    i1, d1 = O.flat_knn(gist_base, gist_test[1], 10)
    assert i1.tolist() == [152, 889, 194, 40, 449, 99, 205, 565, 761, 612]
    assert float(d1[0]).hex() == "0x1.cbddf20000000p-1"


def test_oracle_vs_numpy_emulation_random():
    rng = np.random.default_rng(1)
    base = rng.standard_normal((300, 77)).astype(np.float32)
    for q in rng.standard_normal((5, 77)).astype(np.float32):
        for cos in (False, True):
            ni, nd = R.flat_knn(base, q, 9, cosine=cos)
            oi, od = O.flat_knn(base, q, 9, O.COSINE if cos else O.L2SQR)
            assert ni.tolist() == oi.tolist() and np.array_equal(nd, od)


def test_result_set_semantics():
    # candidate_pair.rs:36-41 total order; NaN greatest; -0 == +0
    L = O.lib()
    assert L.orc_pair_cmp(1.0, 5, 1.0, 7) < 0 and L.orc_pair_cmp(1.0, 5, 0.5, 9) > 0
    assert L.orc_pair_cmp(float("nan"), 0, 1e30, 9) > 0 and L.orc_pair_cmp(float("nan"), 1, float("nan"), 1) == 0
    assert L.orc_pair_cmp(-0.0, 3, 0.0, 3) == 0
    # k larger than n, k == 0
    base = np.eye(4, dtype=np.float32)
    idx, d = O.flat_knn(base, base[1], 10)
    assert idx.tolist() == [1, 0, 2, 3]
    idx, d = O.flat_knn(base, base[1], 0)
    assert len(idx) == 0
    # recall (candidate_pair.rs:127-140)
    assert O.recall([1, 2, 3, 4], [4, 9, 1]) == 0.5
