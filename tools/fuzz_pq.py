"""Randomised parity run of FlatIndex::knn_pq (flat_index.rs:84-104) against the oracle on tables large enough for the fused
threshold-filter scans (>= 65 536 rows): 4- and 8-bit codes, any m (whole, padded and odd code rows), both metrics, ef below and
above the sample rank rule, ties, degenerate queries; the quantised passes and their threshold sample randomly on / off.
usage: python tools/fuzz_pq.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lab_1806_vec_db_amd as vdb
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1806)
t_end = time.time() + budget
it = bad = 0
while time.time() < t_end:
    it += 1
    dim = int(rng.choice([32, 48, 64, 96, 100, 128, 192, 256]))
    n = int(rng.integers(66000, 140000))
    n_bits = int(rng.choice([4, 4, 8]))
    m = int(rng.integers(1, min(dim, 80) + 1))
    nq = int(rng.choice([1, 7, 8, 9, 33, 64]))
    k = int(rng.choice([1, 5, 10, 40]))
    ef = int(rng.choice([k, 10, 64, 100, 300, 1000]))
    dist, kind = (("l2sqr", 0), ("cosine", 1))[int(rng.integers(0, 2))]
    style = int(rng.integers(0, 3))
    if style == 0:
        base = (rng.standard_normal((n, dim)) * rng.uniform(0.1, 3.0, dim)).astype(np.float32)
    elif style == 1:
        base = np.round(np.abs(rng.normal(0.07, 0.045, (n, dim))), 4).astype(np.float32)
    else:  # few distinct rows: equal codes -> ADC ties broken by id
        proto = rng.standard_normal((300, dim)).astype(np.float32)
        base = proto[rng.integers(0, 300, n)]
    qs = (base[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(np.float32)
    if nq > 2 and rng.random() < 0.3:
        qs[1, 0] = np.nan
    if nq > 3 and rng.random() < 0.3:
        qs[2] = 0.0
    ix = vdb.GpuIndex(dim, dist)
    ix.batch_add(base)
    ix.pq_build(n_bits=n_bits, m=m, train_n=3000 if n_bits == 8 else 1500, max_iter=2, seed=it)
    pq = ix.pq_export()
    opq = O.PQ.from_centroids(dim, m, n_bits, kind, pq["centroids"])
    opq.set_codes(pq["codes"])
    a16, s16 = int(rng.choice([0, 0, 1])), int(rng.choice([0, 0, 1]))
    v8 = int(rng.choice([0, 0, 1, 2]))  # 8-bit codes: scan variant (k_pq_adc8x16 / k_pq_adc8 / k_pq_adc16x8)
    ix.set_param("pq_adc16", a16)
    ix.set_param("pq_sample16", s16)
    ix.set_param("pq_adc8_sliced", v8)
    try:
        idx, d, cnt = ix.knn_pq(qs, k, ef)
    finally:
        ix.set_param("pq_adc16", 0)
        ix.set_param("pq_sample16", 0)
        ix.set_param("pq_adc8_sliced", 0)
    ok = True
    for q in range(nq):
        oi, od = O.flat_knn_pq(base, opq, qs[q], k, ef, kind)
        c = int(cnt[q])
        ok = ok and c == len(oi) and idx[q, :c].tolist() == oi.tolist() and np.array_equal(d[q, :c], od, equal_nan=True)
    print(f"#{it} dim {dim} n {n} bits {n_bits} m {m} nq {nq} k {k} ef {ef} {dist} style {style} adc16 {a16} sample16 {s16} scan8 {v8}: "
          f"{'ok' if ok else 'MISMATCH'} quantised {ix.get_stat('pq_adc16_queries')}", flush=True)
    bad += 0 if ok else 1
    ix.close()
print(f"done: {it} configurations, {bad} mismatches")
sys.exit(1 if bad else 0)
